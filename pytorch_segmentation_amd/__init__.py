"""MI355X-native (gfx950) hot path of WoodsGao/pytorch_segmentation: conv-BN-ReLU blocks, ASPP, UNet decoder,
per-pixel cross-entropy and data-parallel gradient exchange on hand-written HIP kernels behind a C ABI
(include/pseg_amd.h).  No CPU / eager fallback exists: without libpseg_amd.so every compute call raises."""
from . import _lib  # noqa: F401
from .arena import ParamArena, prepare

__all__ = ['prepare', 'ParamArena']
__version__ = '0.1.0'
