def checkpoint_wrapper(m, *a, **k):
    return m
