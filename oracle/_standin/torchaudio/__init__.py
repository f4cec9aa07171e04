from . import transforms  # noqa


def load(*a, **k):
    raise NotImplementedError("torchaudio is not installed; stand-in")
