class MelSpectrogram:  # mel branch only (n_mels > 0); never built by shipped configs
    def __init__(self, *a, **k):
        raise NotImplementedError
