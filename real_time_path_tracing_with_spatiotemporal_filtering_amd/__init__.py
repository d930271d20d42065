"""MI355X-native hot path of OnurBasci/Real_Time_Path_Tracing_With_SpatioTemporal_Filtering:
G-buffer -> temporal gradient -> 1-spp path trace -> N edge-stopping a-trous passes with
reprojection + temporal blend, as hand-written gfx950 HIP kernels behind the C ABI of
``include/rtpt.h`` (``librtpt_hip.so``).

* ``abi``    ctypes binding of the C ABI (the only route to the kernels; no CPU fallback)
* ``app``    headless mirror of the reference's PathTracingApplication render loop
* ``strips`` row-strip sharding across ranks + RCCL halo exchange
"""
from . import abi, strips  # noqa: F401
from .abi import Context, PushConstants, RtptError, RtptLibraryMissing, Ubo  # noqa: F401
from .strips import StripPlan  # noqa: F401

__all__ = ["abi", "strips", "Context", "PushConstants", "Ubo", "RtptError", "RtptLibraryMissing", "StripPlan"]
