"""vampnet_amd — MI355X-native engine for the VampNet `Interface.vamp()` hot path.

Python host code over libvampnet_hip.so (hand-written gfx950 HIP kernels behind the C ABI in
include/vampnet_hip.h).  torch is used only as the device-memory/stream container."""
from ._lib import VnError, LIB_PATH  # noqa: F401


def __getattr__(name):
    if name in ("Engine", "VampNetModel", "DEFAULT_PRECISION", "PrecisionFallbackWarning"):
        from . import engine
        return getattr(engine, name)
    if name == "Interface":
        from .interface import Interface
        return Interface
    raise AttributeError(name)
