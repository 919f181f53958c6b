"""pysp_amd -- MI355X (gfx950) implementation of pySP's debayer -> WB -> CCM -> sRGB hot path.

Host code mirrors the reference's Python interface for this path (same names, argument meaning and
error behaviour); all pixel work happens in hand-written HIP kernels reached through the C ABI of
`pysp_amd/csrc/libpysp_hip.so` (include/pysp_hip.h).  There is no CPU fallback.
"""
from .const import QualityDemosaic, PatternDemosaic  # noqa: F401
from .device_array import DeferredImage, DeviceArray, deferred_enabled, lazy_enabled, set_lazy  # noqa: F401

__all__ = ["QualityDemosaic", "PatternDemosaic", "DeviceArray", "DeferredImage", "set_lazy", "lazy_enabled", "deferred_enabled"]
__version__ = "0.1.0"
