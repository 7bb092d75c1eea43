"""resshift_amd — MI355X-native ResShift sampling hot path (HIP kernels behind a C ABI, drop-in Python host)."""
from .config import load_config  # noqa: F401

__all__ = ["load_config", "UNetModelSwin", "VQModelTorch", "create_gaussian_diffusion", "ResShiftSampler", "Engine"]


def __getattr__(name):  # lazy: importing the package must not require torch/HIP until a class is used
    if name == "UNetModelSwin":
        from .unet import UNetModelSwin as v
    elif name == "VQModelTorch":
        from .autoencoder import VQModelTorch as v
    elif name == "create_gaussian_diffusion":
        from .gaussian_diffusion import create_gaussian_diffusion as v
    elif name == "ResShiftSampler":
        from .sampler import ResShiftSampler as v
    elif name == "Engine":
        from .engine import Engine as v
    else:
        raise AttributeError(name)
    return v
