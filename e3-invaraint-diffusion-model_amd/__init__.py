"""MI355X-native denoising hot path of LabJunBMI/E3-invaraint-diffusion-model.

Sub-packages mirror the reference's two script directories (same file names, entry points,
tensor layouts and checkpoint keys); the compute runs in hand-written gfx950 HIP kernels behind
the C-ABI of include/e3d_hip.h (csrc/, loaded by hip.py).  No CPU fallback.
"""
from . import hip  # noqa: F401

__all__ = ["hip", "ops", "bert", "blocks", "training", "optim", "sharding", "autograd", "biolip", "structure_model", "sequence_model"]


def __getattr__(name):
    if name in __all__:
        import importlib
        return importlib.import_module(f"{__name__}.{name}")
    raise AttributeError(name)
