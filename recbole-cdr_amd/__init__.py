"""recbole-cdr_amd: the MI355X-native cross-domain recommendation hot path.

Host-side mirror of the reference's plugin interface for ONE path -- ``CrossDomainRecommender.calculate_loss /
predict / full_sort_predict`` driven by ``CrossDomainTrainer.fit`` -- on top of the C-ABI library ``libcdrhip.so``
(hand-written gfx950 HIP kernels, ``csrc/``).  There is no CPU or eager-PyTorch fallback: every model call goes
through the native library and raises if it is missing or if tensors are not on a ROCm device.

The directory name carries a hyphen (``recbole-cdr_amd``); import it as ``recbole_cdr_amd`` through the one-file
shim at the repository root.
"""
from . import binding  # noqa: F401
from .binding import NativeLibraryError, lib_path, build  # noqa: F401
from .utils import get_model, get_trainer, ModelType, InputType, CrossDomainDataLoaderState, train_mode2state  # noqa: F401

__version__ = '0.1.0'
