"""fithubert_amd: MI355X-native FitHuBERT distillation hot path.

HIP/CDNA4 kernels + a native student engine (``csrc/``, built into ``libfithubert_hip.so``) behind the reference's
own Python surface: ``CustomStudentModelConfig`` / ``CustomStudentModel`` (modules/model.py), ``W2V2Distil`` (train.py),
``TeacherWrapper`` (utils/utils.py), ``UpstreamExpert`` / ``hubconf.fithubert`` (fithubert/).
"""
from .config import CustomStudentModelConfig, load_yaml  # noqa: F401

__all__ = ["CustomStudentModelConfig", "load_yaml"]
