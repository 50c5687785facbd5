"""flac_amd -- MI355X-native FLAC frame engine behind the libFLAC encoder API.

csrc/    HIP kernels + C ABI (include/flacgpu.h) and the host C layer (libFLAC API mirror)
lib/     in-tree build outputs (libflacgpu.so, libFLACgpu.so)
engine   ctypes binding used by tests and bench.py
stream_decoder  ctypes binding of the device decoder for streams the engine did not write
"""
from .engine import FrameEngine, FlacGpuError, make_settings, host_windows, raw_format, RawFormat, VerifyResult  # noqa: F401
from .stream_decoder import StreamDecoder  # noqa: F401,E402
