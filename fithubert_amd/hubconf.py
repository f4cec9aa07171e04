"""S3PRL hub entry (reference fithubert/hubconf.py:3-13)."""
from .expert import UpstreamExpert as _UpstreamExpert


def fithubert(ckpt, model_config, **kwargs):
    """FitHuBERT student on the MI355X-native engine.

    ckpt: Lightning checkpoint path with ``student_model.*`` weights; model_config: the yaml used for distillation."""
    return _UpstreamExpert(ckpt, model_config, **kwargs)
