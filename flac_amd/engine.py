"""flac_amd.engine -- thin ctypes binding of the C ABI (include/flacgpu.h + the host C layer).

Python is plumbing only: settings are resolved and window tables computed by the host C library
(libFLACgpu.so), frames are encoded by the HIP engine (libflacgpu.so).  There is no CPU encode path
here: if the HIP library or a GPU is missing, construction fails loudly.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBDIR = os.path.join(_HERE, "lib")
# (FLACGPU_ENGINE_SO: another build of the same library, for same-box A/B runs of a kernel change -- scripts/ab_engine.py)
ENGINE_SO = os.environ.get("FLACGPU_ENGINE_SO") or os.path.join(_LIBDIR, "libflacgpu.so")
HOST_SO = os.path.join(_LIBDIR, "libFLACgpu.so")

FLACGPU_MAX_APODIZATIONS = 32
FGH_MAX_APODIZATIONS = 32


class RawFormat(C.Structure):
    """flacgpu_raw_format (include/flacgpu.h)"""
    _fields_ = [("container_bits", C.c_uint32), ("big_endian", C.c_uint32), ("is_unsigned", C.c_uint32), ("shift", C.c_uint32),
                ("use_channel_map", C.c_uint32), ("channel_map", C.c_uint8 * 8)]


def raw_format(container_bits, big_endian=False, is_unsigned=False, shift=0, channel_map=None):
    f = RawFormat(container_bits, int(big_endian), int(is_unsigned), shift, 0)
    if channel_map is not None:
        f.use_channel_map = 1
        for i, c in enumerate(channel_map):
            f.channel_map[i] = c
    return f


class FlacGpuError(RuntimeError):
    pass


class _Apod(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("parts", C.c_uint32)]


class EngineConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32), ("channels", C.c_uint32), ("bits_per_sample", C.c_uint32),
        ("sample_rate", C.c_uint32), ("blocksize", C.c_uint32), ("do_mid_side_stereo", C.c_uint32),
        ("loose_mid_side_stereo", C.c_uint32), ("max_lpc_order", C.c_uint32), ("qlp_coeff_precision", C.c_uint32),
        ("min_residual_partition_order", C.c_uint32), ("max_residual_partition_order", C.c_uint32),
        ("num_apodizations", C.c_uint32), ("apodizations", _Apod * FLACGPU_MAX_APODIZATIONS),
        ("disable_constant_subframes", C.c_uint32), ("disable_fixed_subframes", C.c_uint32),
        ("disable_verbatim_subframes", C.c_uint32), ("limit_min_bitrate", C.c_uint32),
        ("device", C.c_int32), ("max_batch_frames", C.c_uint32),
        ("do_exhaustive_model_search", C.c_uint32), ("do_qlp_coeff_prec_search", C.c_uint32),
    ]


class _HostApod(C.Structure):
    _fields_ = [("type", C.c_int), ("p", C.c_float), ("start", C.c_float), ("end", C.c_float), ("parts", C.c_int32)]


class HostSettings(C.Structure):
    _fields_ = [
        ("channels", C.c_uint32), ("bits_per_sample", C.c_uint32), ("sample_rate", C.c_uint32), ("blocksize", C.c_uint32),
        ("streamable_subset", C.c_int), ("do_md5", C.c_int), ("verify", C.c_int),
        ("do_mid_side_stereo", C.c_int), ("loose_mid_side_stereo", C.c_int),
        ("max_lpc_order", C.c_uint32), ("qlp_coeff_precision", C.c_uint32),
        ("do_qlp_coeff_prec_search", C.c_int), ("do_escape_coding", C.c_int), ("do_exhaustive_model_search", C.c_int),
        ("min_residual_partition_order", C.c_uint32), ("max_residual_partition_order", C.c_uint32),
        ("rice_parameter_search_dist", C.c_uint32),
        ("num_apodizations", C.c_uint32), ("apodizations", _HostApod * FGH_MAX_APODIZATIONS),
        ("limit_min_bitrate", C.c_int),
        ("disable_constant_subframes", C.c_int), ("disable_fixed_subframes", C.c_int), ("disable_verbatim_subframes", C.c_int),
        ("total_samples_estimate", C.c_uint64),
    ]


class SubframeInfo(C.Structure):
    _fields_ = [("type", C.c_uint8), ("order", C.c_uint8), ("wasted_bits", C.c_uint8), ("partition_order", C.c_uint8),
                ("rice2", C.c_uint8), ("precision", C.c_uint8), ("shift", C.c_int8), ("pad", C.c_uint8),
                ("bits", C.c_uint32)]


class VerifyResult(C.Structure):
    """flacgpu_verify_result (include/flacgpu.h)"""
    _fields_ = [("status", C.c_int32), ("frame_number", C.c_uint32), ("channel", C.c_uint32), ("sample", C.c_uint32),
                ("absolute_sample", C.c_uint64), ("expected", C.c_int32), ("got", C.c_int32)]


_host = None
_engine = None


def load_host():
    """The host C layer; loading it also loads libflacgpu.so (it links against it)."""
    global _host
    if _host is None:
        if not os.path.exists(HOST_SO):
            raise FlacGpuError("%s not built: run `python -c 'import __graft_entry__ as g; g.build()'`" % HOST_SO)
        lib = C.CDLL(HOST_SO)      # RTLD_LOCAL: it exports the libFLAC encoder names, which must not interpose on a real libFLAC
        lib.flacgpu_host_settings_defaults.argtypes = [C.POINTER(HostSettings)]
        lib.flacgpu_host_settings_level.argtypes = [C.POINTER(HostSettings), C.c_uint32]
        lib.flacgpu_host_settings_apodization.argtypes = [C.POINTER(HostSettings), C.c_char_p]
        lib.flacgpu_host_settings_resolve.restype = C.c_int
        lib.flacgpu_host_settings_resolve.argtypes = [C.POINTER(HostSettings)]
        lib.flacgpu_host_engine_config.restype = C.c_int
        lib.flacgpu_host_engine_config.argtypes = [C.POINTER(HostSettings), C.c_int, C.c_uint32, C.POINTER(EngineConfig)]
        lib.flacgpu_host_windows.argtypes = [C.POINTER(HostSettings), C.c_uint32, C.c_void_p]
        _host = lib
    return _host


def load_engine():
    global _engine
    if _engine is None:
        if not os.path.exists(ENGINE_SO):
            raise FlacGpuError("%s not built: the HIP extension is required (no CPU fallback)" % ENGINE_SO)
        lib = C.CDLL(ENGINE_SO)
        lib.flacgpu_create.restype = C.c_int
        lib.flacgpu_create.argtypes = [C.POINTER(EngineConfig), C.c_void_p, C.POINTER(C.c_void_p)]
        lib.flacgpu_destroy.argtypes = [C.c_void_p]
        lib.flacgpu_max_output_bytes.restype = C.c_size_t
        lib.flacgpu_max_output_bytes.argtypes = [C.c_void_p, C.c_uint32]
        lib.flacgpu_encode_batch.restype = C.c_int64
        lib.flacgpu_encode_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint32, C.c_void_p,
                                             C.c_void_p, C.c_size_t, C.c_void_p]
        lib.flacgpu_encode_batch_device.restype = C.c_int
        lib.flacgpu_encode_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint32,
                                                    C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                                    C.c_void_p]
        lib.flacgpu_last_batch_info.restype = C.c_int
        lib.flacgpu_last_batch_info.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        lib.flacgpu_last_batch_kernel_ms.restype = C.c_int
        lib.flacgpu_last_batch_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                                     C.POINTER(C.c_float)]
        lib.flacgpu_last_batch_phase_ms.restype = C.c_int
        lib.flacgpu_last_batch_phase_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float * 6)]
        lib.flacgpu_batch_phase_ms.restype = C.c_int
        lib.flacgpu_batch_phase_ms.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_float * 6)]
        lib.flacgpu_stage_raw_device.restype = C.c_int
        lib.flacgpu_stage_raw_device.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(RawFormat), C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.flacgpu_encode_batch_raw.restype = C.c_int64
        lib.flacgpu_encode_batch_raw.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(RawFormat), C.c_uint32, C.c_uint64, C.c_uint32,
                                                 C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        lib.flacgpu_verify_batch_device.restype = C.c_int
        lib.flacgpu_verify_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.flacgpu_set_verify.restype = C.c_int
        lib.flacgpu_set_verify.argtypes = [C.c_void_p, C.c_uint32]
        lib.flacgpu_last_verify_result.restype = C.c_int
        lib.flacgpu_last_verify_result.argtypes = [C.c_void_p, C.POINTER(VerifyResult)]
        lib.flacgpu_set_subbatches.restype = C.c_int
        lib.flacgpu_set_subbatches.argtypes = [C.c_void_p, C.c_uint32]
        lib.flacgpu_strerror.restype = C.c_char_p
        lib.flacgpu_strerror.argtypes = [C.c_int]
        lib.flacgpu_submit_batch_raw.restype = C.c_int
        lib.flacgpu_submit_batch_raw.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(RawFormat), C.c_uint32, C.c_uint64, C.c_uint32,
                                                 C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        lib.flacgpu_collect.restype = C.c_int64
        lib.flacgpu_collect.argtypes = [C.c_void_p]
        lib.flacgpu_in_flight.restype = C.c_int
        lib.flacgpu_in_flight.argtypes = [C.c_void_p]
        lib.flacgpu_device_count.restype = C.c_int
        _engine = lib
    return _engine


def make_settings(channels=2, bps=16, rate=44100, level=5, blocksize=0, apodization=None, limit_min_bitrate=0,
                  max_lpc_order=None, max_partition_order=None, min_partition_order=None, mid_side=None,
                  loose_mid_side=None, qlp_coeff_precision=None, streamable_subset=1, exhaustive=0, prec_search=0,
                  disable=(0, 0, 0)):
    """Mirrors the order of the FLAC__stream_encoder_set_* calls a client makes before init."""
    h = load_host()
    s = HostSettings()
    h.flacgpu_host_settings_defaults(C.byref(s))
    s.channels, s.bits_per_sample, s.sample_rate = channels, bps, rate
    s.streamable_subset = streamable_subset
    h.flacgpu_host_settings_level(C.byref(s), level)
    if blocksize:
        s.blocksize = blocksize
    if max_lpc_order is not None:
        s.max_lpc_order = max_lpc_order
    if qlp_coeff_precision is not None:
        s.qlp_coeff_precision = qlp_coeff_precision
    if min_partition_order is not None:
        s.min_residual_partition_order = min_partition_order
    if max_partition_order is not None:
        s.max_residual_partition_order = max_partition_order
    if mid_side is not None:
        s.do_mid_side_stereo = mid_side
    if loose_mid_side is not None:
        s.loose_mid_side_stereo = loose_mid_side
    if apodization:
        h.flacgpu_host_settings_apodization(C.byref(s), apodization.encode())
    s.limit_min_bitrate = limit_min_bitrate
    s.do_exhaustive_model_search = 1 if exhaustive else 0
    s.do_qlp_coeff_prec_search = 1 if prec_search else 0
    s.disable_constant_subframes, s.disable_fixed_subframes, s.disable_verbatim_subframes = disable
    st = h.flacgpu_host_settings_resolve(C.byref(s))
    if st != 0:
        raise FlacGpuError("invalid encoder settings: FLAC__StreamEncoderInitStatus %d" % st)
    return s


def host_windows(settings, blocksize):
    h = load_host()
    w = np.empty((settings.num_apodizations, blocksize), dtype=np.float32)
    h.flacgpu_host_windows(C.byref(settings), blocksize, w.ctypes.data)
    return w


def md5_many_device(d_base_ptr, offsets, lengths, device=0, stream=None):
    """MD5 digests of len(offsets) byte ranges of device memory (flacgpu_md5_many_device: one lane per stream, for a corpus of many
    streams whose sample bytes are staged in HBM anyway).  d_base_ptr: raw device address; returns a list of 16-byte digests."""
    import numpy as np
    lib = load_engine()
    lib.flacgpu_md5_many_device.restype = C.c_int
    lib.flacgpu_md5_many_device.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    n = len(offsets)
    offs = np.ascontiguousarray(offsets, dtype=np.uint64)
    lens = np.ascontiguousarray(lengths, dtype=np.uint64)
    out = np.zeros((max(n, 1), 16), dtype=np.uint8)
    r = lib.flacgpu_md5_many_device(device, d_base_ptr, offs.ctypes.data, lens.ctypes.data, n, out.ctypes.data, stream)
    if r != 0:
        raise FlacGpuError("flacgpu_md5_many_device: %s" % lib.flacgpu_strerror(r).decode())
    return [out[i].tobytes() for i in range(n)]


class FrameEngine:
    """One GPU frame engine for one stream configuration (flacgpu_create .. flacgpu_destroy)."""

    def __init__(self, settings, device=0, max_batch_frames=4096):
        self.lib = load_engine()
        self.host = load_host()
        self.settings = settings
        self.cfg = EngineConfig()
        r = self.host.flacgpu_host_engine_config(C.byref(settings), device, max_batch_frames, C.byref(self.cfg))
        if r != 0:
            raise FlacGpuError("engine config: %s" % self.lib.flacgpu_strerror(r).decode())
        self.windows = host_windows(settings, settings.blocksize) if settings.max_lpc_order > 0 else None
        self.ctx = C.c_void_p()
        r = self.lib.flacgpu_create(C.byref(self.cfg), self.windows.ctypes.data if self.windows is not None else None,
                                    C.byref(self.ctx))
        if r != 0:
            self.ctx = None
            raise FlacGpuError("flacgpu_create: %s" % self.lib.flacgpu_strerror(r).decode())
        self.channels = settings.channels
        self.blocksize = settings.blocksize
        self.max_batch_frames = max_batch_frames

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.flacgpu_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def max_output_bytes(self, nframes):
        return self.lib.flacgpu_max_output_bytes(self.ctx, nframes)

    def _tail_windows(self, tail):
        if tail and self.settings.max_lpc_order > 0:
            return host_windows(self.settings, tail)
        return None

    def encode(self, pcm, first_frame_number=0):
        """pcm: int32 [nsamples, channels] in host memory -> (bytes, frame_bytes[nframes]).
        Splits into batches of max_batch_frames; a short final block is encoded as the last frame."""
        pcm = np.ascontiguousarray(pcm, dtype=np.int32)
        n, ch = pcm.shape
        assert ch == self.channels
        N = self.blocksize
        nframes_total = (n + N - 1) // N
        out_parts, fbs = [], []
        f0 = 0
        while f0 < nframes_total:
            nf = min(self.max_batch_frames, nframes_total - f0)
            s0, s1 = f0 * N, min(n, (f0 + nf) * N)
            tail = (s1 - s0) - (nf - 1) * N
            tail = 0 if tail == N else tail
            tw = self._tail_windows(tail)
            cap = self.max_output_bytes(nf)
            out = np.empty(cap, dtype=np.uint8)
            fb = np.empty(nf, dtype=np.uint32)
            chunk = pcm[s0:s1]
            r = self.lib.flacgpu_encode_batch(self.ctx, chunk.ctypes.data, nf, first_frame_number + f0, tail,
                                              tw.ctypes.data if tw is not None else None, out.ctypes.data, cap,
                                              fb.ctypes.data)
            if r < 0:
                raise FlacGpuError("flacgpu_encode_batch: %s" % self.lib.flacgpu_strerror(int(r)).decode())
            out_parts.append(out[:r].tobytes())
            fbs.append(fb)
            f0 += nf
        return b"".join(out_parts), np.concatenate(fbs) if fbs else np.zeros(0, np.uint32)

    def encode_raw(self, raw, fmt, first_frame_number=0):
        """raw: the sample bytes of ONE batch (<= max_batch_frames blocks) as they sit in a WAVE/AIFF/raw file;
        fmt: RawFormat.  Staged to int32 on the device (format_input of the reference's CLI), then encoded."""
        raw = np.frombuffer(raw, dtype=np.uint8) if not isinstance(raw, np.ndarray) else np.ascontiguousarray(raw).view(np.uint8)
        bytes_per_wide = (fmt.container_bits // 8) * self.channels
        n = raw.size // bytes_per_wide
        N = self.blocksize
        nf = (n + N - 1) // N
        assert 0 < nf <= self.max_batch_frames
        tail = n - (nf - 1) * N
        tail = 0 if tail == N else tail
        tw = self._tail_windows(tail)
        cap = self.max_output_bytes(nf)
        out = np.empty(cap, dtype=np.uint8)
        fb = np.empty(nf, dtype=np.uint32)
        r = self.lib.flacgpu_encode_batch_raw(self.ctx, raw.ctypes.data, C.byref(fmt), nf, first_frame_number, tail,
                                              tw.ctypes.data if tw is not None else None, out.ctypes.data, cap, fb.ctypes.data)
        if r < 0:
            raise FlacGpuError("flacgpu_encode_batch_raw: %s" % self.lib.flacgpu_strerror(int(r)).decode())
        return out[:r].tobytes(), fb

    def stage_raw_device(self, d_raw_ptr, fmt, wide_samples, d_pcm_ptr, d_err_ptr=None, stream=None):
        r = self.lib.flacgpu_stage_raw_device(self.ctx, d_raw_ptr, C.byref(fmt), wide_samples, d_pcm_ptr, d_err_ptr, stream)
        if r != 0:
            raise FlacGpuError("flacgpu_stage_raw_device: %s" % self.lib.flacgpu_strerror(r).decode())

    def encode_device(self, d_pcm_ptr, nframes, d_out_ptr, out_cap, d_frame_bytes_ptr, d_total_ptr,
                      first_frame_number=0, tail=0, stream=None):
        """All pointers are raw device addresses (e.g. torch.Tensor.data_ptr()). Asynchronous."""
        tw = self._tail_windows(tail)
        r = self.lib.flacgpu_encode_batch_device(self.ctx, d_pcm_ptr, nframes, first_frame_number, tail,
                                                 tw.ctypes.data if tw is not None else None, d_out_ptr, out_cap,
                                                 d_frame_bytes_ptr, d_total_ptr, stream)
        if r != 0:
            raise FlacGpuError("flacgpu_encode_batch_device: %s" % self.lib.flacgpu_strerror(r).decode())

    def set_verify(self, on=True):
        """every batch of encode()/encode_raw() is decoded again on the device and compared with its input"""
        r = self.lib.flacgpu_set_verify(self.ctx, 1 if on else 0)
        if r != 0:
            raise FlacGpuError("flacgpu_set_verify: %s" % self.lib.flacgpu_strerror(r).decode())

    def last_verify_result(self):
        v = VerifyResult()
        r = self.lib.flacgpu_last_verify_result(self.ctx, C.byref(v))
        if r != 0:
            raise FlacGpuError("flacgpu_last_verify_result: %s" % self.lib.flacgpu_strerror(r).decode())
        return v

    def verify_device(self, d_frames_ptr, d_frame_bytes_ptr, nframes, d_pcm_ptr, d_result_ptr, first_frame_number=0, tail=0, stream=None):
        """flacgpu_verify_batch_device: all pointers are raw device addresses; d_result_ptr receives a VerifyResult. Asynchronous."""
        r = self.lib.flacgpu_verify_batch_device(self.ctx, d_frames_ptr, d_frame_bytes_ptr, nframes, first_frame_number, tail, d_pcm_ptr, d_result_ptr, stream)
        if r != 0:
            raise FlacGpuError("flacgpu_verify_batch_device: %s" % self.lib.flacgpu_strerror(r).decode())

    def verify_hinted_frames(self):
        """development aid: frames of the most recent verify call that the thread-per-run pass vouched for (synchronises)"""
        n = C.c_uint32(0)
        self.lib.flacgpu_debug_verify_hinted_frames.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        r = self.lib.flacgpu_debug_verify_hinted_frames(self.ctx, C.byref(n))
        if r != 0:
            raise FlacGpuError("flacgpu_debug_verify_hinted_frames: %s" % self.lib.flacgpu_strerror(r).decode())
        return int(n.value)

    def last_batch_info(self, nframes):
        sub = (SubframeInfo * (nframes * self.channels))()
        ca = np.zeros(nframes, dtype=np.uint8)
        r = self.lib.flacgpu_last_batch_info(self.ctx, nframes, sub, ca.ctypes.data)
        if r != 0:
            raise FlacGpuError("flacgpu_last_batch_info: %s" % self.lib.flacgpu_strerror(r).decode())
        return sub, ca

    def last_batch_kernels(self):
        """names of the kernels (families / flavours) the most recent batch launched (flacgpu_last_batch_kernels)"""
        m = C.c_uint32(0)
        self.lib.flacgpu_last_batch_kernels.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        self.lib.flacgpu_kernel_bit_name.argtypes = [C.c_uint32]
        self.lib.flacgpu_kernel_bit_name.restype = C.c_char_p
        r = self.lib.flacgpu_last_batch_kernels(self.ctx, C.byref(m))
        if r != 0:
            raise FlacGpuError("flacgpu_last_batch_kernels: %s" % self.lib.flacgpu_strerror(r).decode())
        return {self.lib.flacgpu_kernel_bit_name(b).decode() for b in range(32) if m.value >> b & 1 and self.lib.flacgpu_kernel_bit_name(b)}

    def fused_fallbacks(self):
        """(frames that gave up waiting in the fused output since the engine was created, the most of any one batch); synchronises"""
        t, mx = C.c_uint32(0), C.c_uint32(0)
        self.lib.flacgpu_fused_fallbacks.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        r = self.lib.flacgpu_fused_fallbacks(self.ctx, C.byref(t), C.byref(mx))
        if r != 0:
            raise FlacGpuError("flacgpu_fused_fallbacks: %s" % self.lib.flacgpu_strerror(r).decode())
        return int(t.value), int(mx.value)

    def set_subbatches(self, n):
        r = self.lib.flacgpu_set_subbatches(self.ctx, n)
        if r != 0:
            raise FlacGpuError("flacgpu_set_subbatches: %s" % self.lib.flacgpu_strerror(r).decode())

    def last_phase_ms(self, batches_ago=0):
        """per-kernel ms of a recent batch (0 = the last): prep, autoc, model, eval, pack, scan+compact"""
        ms = (C.c_float * 6)()
        r = self.lib.flacgpu_batch_phase_ms(self.ctx, batches_ago, C.byref(ms))
        if r == 1:
            return None                                  # that batch carried no timing events (set_phase_timing)
        if r != 0:
            raise FlacGpuError("flacgpu_batch_phase_ms: %s" % self.lib.flacgpu_strerror(r).decode())
        return dict(zip(("prep", "autoc", "model", "eval", "pack", "scan_compact"), [float(v) for v in ms]))

    def set_phase_timing(self, every):
        """phase timing on every n-th batch (1: all, the default; 0: none): the event records are instrumentation, not work"""
        self.lib.flacgpu_set_phase_timing.restype = C.c_int
        self.lib.flacgpu_set_phase_timing.argtypes = [C.c_void_p, C.c_uint32]
        r = self.lib.flacgpu_set_phase_timing(self.ctx, every)
        if r != 0:
            raise FlacGpuError("flacgpu_set_phase_timing: %s" % self.lib.flacgpu_strerror(r).decode())

    def last_kernel_ms(self):
        a, p, k = C.c_float(), C.c_float(), C.c_float()
        r = self.lib.flacgpu_last_batch_kernel_ms(self.ctx, C.byref(a), C.byref(p), C.byref(k))
        if r != 0:
            raise FlacGpuError("flacgpu_last_batch_kernel_ms: %s" % self.lib.flacgpu_strerror(r).decode())
        return a.value, p.value, k.value
