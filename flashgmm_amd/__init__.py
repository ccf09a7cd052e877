"""flashgmm_amd — MI355X-native GMM entropy-coding path (drop-in for tokkiwa/FlashGMM's
``GaussianMixtureConditional.compress/decompress`` and ``compressai.ans`` GMM coder calls).

    flashgmm_amd.ans               RansEncoder / BufferedRansEncoder / RansDecoder   (mirror of compressai.ans)
    flashgmm_amd.entropy_models    GaussianMixtureConditional                        (mirror of the Python boundary)
    flashgmm_amd._lib              ctypes binding of libflashgmm_amd.so (include/flashgmm_amd.h)


All floating-point work runs in hand-written HIP kernels (flashgmm_amd/csrc); there is no CPU fallback.
"""
from . import _lib  # noqa: F401
from .entropy_models import CheckpointedBytes, CompressedBatch, EntropyBottleneckCoder, GaussianMixtureConditional, ParameterHead  # noqa: F401
from . import ans  # noqa: F401

__version__ = "0.1.0"
