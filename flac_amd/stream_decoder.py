"""flac_amd.stream_decoder -- ctypes binding of the device stream decoder (include/flacgpu.h: flacgpu_decode_stream_device): FLAC
streams this engine did not write, found by their sync codes and decoded a lane per frame on the GPU, with the errors the reference's
decoder would report (what `flac -t` / `flac -d` need; src/libFLAC/stream_decoder.c:1168).  torch is used for
device memory only.  No CPU path: without the HIP library or a GPU, construction fails."""
import ctypes as C

import numpy as np

from .engine import FlacGpuError, load_engine

ERROR_NAMES = {1: "LOST_SYNC", 2: "BAD_HEADER", 3: "FRAME_CRC_MISMATCH", 4: "UNPARSEABLE_STREAM", 5: "BAD_METADATA", 6: "OUT_OF_BOUNDS",
               7: "MISSING_FRAME"}


class StreamInfo(C.Structure):
    """flacgpu_stream_info"""
    _fields_ = [("has_streaminfo", C.c_uint32), ("min_blocksize", C.c_uint32), ("max_blocksize", C.c_uint32), ("sample_rate", C.c_uint32),
                ("channels", C.c_uint32), ("bits_per_sample", C.c_uint32)]


class DecodeEvent(C.Structure):
    _fields_ = [("status", C.c_uint32), ("pad", C.c_uint32), ("byte_offset", C.c_uint64)]


class DecodeResult(C.Structure):
    """flacgpu_decode_result"""
    _fields_ = [("samples", C.c_uint64), ("frames", C.c_uint64), ("silence_samples", C.c_uint64), ("candidates", C.c_uint64),
                ("deferred_decoded", C.c_uint64), ("redecoded_frames", C.c_uint64), ("nevents", C.c_uint32), ("end_in_header", C.c_uint32), ("format_changes", C.c_uint32),
                ("long_rice_codes", C.c_uint32), ("channels", C.c_uint32), ("bits_per_sample", C.c_uint32), ("sample_rate", C.c_uint32),
                ("errors_by_status", C.c_uint32 * 8), ("ms_scan", C.c_float), ("ms_decode", C.c_float), ("ms_place", C.c_float),
                ("ms_total", C.c_float)]


def _lib():
    lib = load_engine()
    if not getattr(lib, "_sd_bound", False):
        lib.flacgpu_decoder_create.restype = C.c_int
        lib.flacgpu_decoder_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        lib.flacgpu_decoder_destroy.argtypes = [C.c_void_p]
        lib.flacgpu_probe_stream.restype = C.c_int
        lib.flacgpu_probe_stream.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(StreamInfo), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_void_p]
        lib.flacgpu_decode_stream_device.restype = C.c_int
        lib.flacgpu_decode_stream_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(StreamInfo), C.c_void_p, C.c_uint64,
                                                     C.POINTER(DecodeResult), C.c_void_p, C.c_uint32, C.c_void_p]
        lib.flacgpu_pack_samples_device.restype = C.c_int
        lib.flacgpu_pack_samples_device.argtypes = [C.c_int, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p]
        lib._sd_bound = True
    return lib


def probe(stream):
    """(StreamInfo, first_frame_offset, total_samples, md5 bytes) of a FLAC file's first bytes (host; flacgpu_probe_stream)."""
    lib = _lib()
    buf = np.frombuffer(bytes(stream), dtype=np.uint8)
    si = StreamInfo()
    first = C.c_uint64(0)
    total = C.c_uint64(0)
    md5 = np.zeros(16, dtype=np.uint8)
    r = lib.flacgpu_probe_stream(buf.ctypes.data, len(buf), C.byref(si), C.byref(first), C.byref(total), md5.ctypes.data)
    if r != 0:
        raise FlacGpuError("flacgpu_probe_stream: %s" % lib.flacgpu_strerror(r).decode())
    return si, first.value, total.value, md5.tobytes()


class StreamDecoder:
    def __init__(self, device=0):
        self._lib = _lib()
        self._h = C.c_void_p()
        self.device = device
        r = self._lib.flacgpu_decoder_create(device, C.byref(self._h))
        if r != 0:
            self._h = None
            raise FlacGpuError("flacgpu_decoder_create: %s" % self._lib.flacgpu_strerror(r).decode())

    def close(self):
        if getattr(self, "_h", None):
            self._lib.flacgpu_decoder_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def decode_device(self, d_stream_ptr, nbytes, first_frame_offset, info, d_pcm_ptr, capacity_values, max_events=4096, stream=None):
        """Raw entry: device pointers in, (rc, DecodeResult, [(status, byte_offset)...]) out."""
        res = DecodeResult()
        ev = (DecodeEvent * max(max_events, 1))()
        rc = self._lib.flacgpu_decode_stream_device(self._h, d_stream_ptr, nbytes, first_frame_offset, C.byref(info) if info is not None else None,
                                                    d_pcm_ptr, capacity_values, C.byref(res), ev, max_events, stream)
        events = [(ev[i].status, ev[i].byte_offset) for i in range(min(res.nevents, max_events))]
        return rc, res, events

    def decode(self, stream, want_pcm=True, info=None, first_frame_offset=None, capacity_values=None, max_events=4096):
        """A whole stream from host bytes: copies it to the device, decodes, returns dict(pcm [samples][channels] int32 or None,
        events [status...], event_offsets, result fields...).  The output is sized from STREAMINFO's total, or by a verdict-only
        first call when that is unknown or too small."""
        import torch
        data = bytes(stream)
        si, first, total, md5 = probe(data)
        if info is None:
            info = si
        if first_frame_offset is None:
            first_frame_offset = first
        dev = torch.device("cuda", self.device)
        n = len(data)
        d_stream = torch.zeros(((n + 3) // 4) * 4 + 64, dtype=torch.uint8, device=dev)
        if n:
            d_stream[:n] = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(dev)
        torch.cuda.synchronize(dev)
        pcm = None
        if not want_pcm:
            rc, res, events = self.decode_device(d_stream.data_ptr(), n, first_frame_offset, info, None, 0, max_events)
        else:
            values = capacity_values if capacity_values is not None else (int(total) * int(info.channels) if info.has_streaminfo else 0)
            for attempt in range(3):
                if values <= 0:                                         # size unknown: a verdict-only pass tells
                    rc, res, events = self.decode_device(d_stream.data_ptr(), n, first_frame_offset, info, None, 0, max_events)
                    values = int(res.samples) * int(res.channels)
                    if rc != 0 or values == 0:
                        break
                d_pcm = torch.empty(values, dtype=torch.int32, device=dev)
                d_pcm.fill_(0x5a5a5a5a)
                rc, res, events = self.decode_device(d_stream.data_ptr(), n, first_frame_offset, info, d_pcm.data_ptr(), values, max_events)
                if rc == -4:                                            # FLACGPU_ERR_OUTPUT_TOO_SMALL: result says what is needed
                    values = int(res.samples) * int(res.channels)
                    continue
                break
            if rc == 0:
                c = max(int(res.channels), 1)
                ns = int(res.samples)
                pcm = d_pcm[:ns * c].cpu().numpy().reshape(ns, c) if ns else np.zeros((0, c), dtype=np.int32)
        if rc != 0:
            raise FlacGpuError("flacgpu_decode_stream_device: %s" % self._lib.flacgpu_strerror(rc).decode())
        return dict(pcm=pcm, events=[e[0] for e in events], event_offsets=[e[1] for e in events], nevents=int(res.nevents), ok=not bool(res.end_in_header),
                    samples=int(res.samples), frames=int(res.frames), silence=int(res.silence_samples), channels=int(res.channels),
                    bps=int(res.bits_per_sample), sample_rate=int(res.sample_rate), format_changes=int(res.format_changes),
                    long_rice_codes=int(res.long_rice_codes), candidates=int(res.candidates), redecoded=int(res.redecoded_frames), deferred_decoded=int(res.deferred_decoded),
                    ms=dict(scan=res.ms_scan, decode=res.ms_decode, place=res.ms_place, total=res.ms_total), md5=md5, total_samples=total)

    def pack_samples(self, d_pcm_ptr, nvalues, bps, d_out_ptr, stream=None):
        r = self._lib.flacgpu_pack_samples_device(self.device, d_pcm_ptr, nvalues, bps, d_out_ptr, stream)
        if r != 0:
            raise FlacGpuError("flacgpu_pack_samples_device: %s" % self._lib.flacgpu_strerror(r).decode())
