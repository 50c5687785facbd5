"""flac_amd -- MI355X-native FLAC frame engine behind the libFLAC encoder API.

csrc/    HIP kernels + C ABI (include/flacgpu.h) and the host C layer (libFLAC API mirror)
lib/     in-tree build outputs (libflacgpu.so, libFLACgpu.so)
engine   ctypes binding used by tests and bench.py
"""
from .engine import FrameEngine, FlacGpuError, make_settings, host_windows, raw_format, RawFormat, VerifyResult  # noqa: F401
