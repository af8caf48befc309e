"""kmcp_amd — MI355X-native `kmcp search` hot path (ntHash k-mer generation + COBS index query).

The product is kmcp_amd/libkmcpgpu.so (HIP/gfx950 behind the C ABI in include/kmcp_gpu.h); this package
is the thin Python plumbing used by the tests, bench.py and the torch.distributed multi-GPU driver.
"""
from . import lib  # noqa: F401
from .lib import Database, KmcpGpuError, default_params  # noqa: F401

__all__ = ["Database", "KmcpGpuError", "default_params", "lib"]
