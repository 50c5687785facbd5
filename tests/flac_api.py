"""ctypes driver for the libFLAC stream-encoder API (include/FLACgpu_stream_encoder.h ==
include/FLAC/stream_encoder.h of the reference).  The SAME call sequence can be run against
flac_amd/lib/libFLACgpu.so (the product) and oracle/_ref/libFLAC_ref.so (the unmodified reference), so the
tests read like a client program and compare whole .flac files byte for byte."""
import ctypes as C
import io
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GPU_SO = os.path.join(ROOT, "flac_amd", "lib", "libFLACgpu.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libFLAC_ref.so")

WRITE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_ubyte), C.c_size_t, C.c_uint32, C.c_uint32, C.c_void_p)
SEEK_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_void_p)
TELL_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p)
META_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p)
PROGRESS_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p)

# metadata structures (format.h:505-895) -- only what the tests build by hand
class SeekPoint(C.Structure):
    _fields_ = [("sample_number", C.c_uint64), ("stream_offset", C.c_uint64), ("frame_samples", C.c_uint32)]


class SeekTable(C.Structure):
    _fields_ = [("num_points", C.c_uint32), ("points", C.POINTER(SeekPoint))]


class VCEntry(C.Structure):
    _fields_ = [("length", C.c_uint32), ("entry", C.c_char_p)]


class VorbisComment(C.Structure):
    _fields_ = [("vendor_string", VCEntry), ("num_comments", C.c_uint32), ("comments", C.POINTER(VCEntry))]


class Application(C.Structure):
    _fields_ = [("id", C.c_ubyte * 4), ("data", C.c_char_p)]


class StreamInfo(C.Structure):
    _fields_ = [("min_blocksize", C.c_uint32), ("max_blocksize", C.c_uint32), ("min_framesize", C.c_uint32),
                ("max_framesize", C.c_uint32), ("sample_rate", C.c_uint32), ("channels", C.c_uint32),
                ("bits_per_sample", C.c_uint32), ("total_samples", C.c_uint64), ("md5sum", C.c_ubyte * 16)]


class Picture(C.Structure):
    _fields_ = [("type", C.c_int), ("mime_type", C.c_char_p), ("description", C.c_char_p), ("width", C.c_uint32),
                ("height", C.c_uint32), ("depth", C.c_uint32), ("colors", C.c_uint32), ("data_length", C.c_uint32),
                ("data", C.c_char_p)]


class _Data(C.Union):
    _fields_ = [("stream_info", StreamInfo), ("seek_table", SeekTable), ("vorbis_comment", VorbisComment),
                ("application", Application), ("picture", Picture), ("raw", C.c_ubyte * 160)]


class StreamMetadata(C.Structure):
    _fields_ = [("type", C.c_int), ("is_last", C.c_int), ("length", C.c_uint32), ("data", _Data)]


def padding(nbytes):
    m = StreamMetadata(); m.type = 1; m.length = nbytes
    return m


def application(app_id, payload):
    m = StreamMetadata(); m.type = 2; m.length = 4 + len(payload)
    m.data.application.id = (C.c_ubyte * 4)(*app_id)
    m.data.application.data = payload
    m._keep = payload
    return m


def seektable(sample_numbers):
    pts = (SeekPoint * len(sample_numbers))()
    for i, s in enumerate(sample_numbers):
        pts[i].sample_number = s
    m = StreamMetadata(); m.type = 3; m.length = 18 * len(sample_numbers)
    m.data.seek_table.num_points = len(sample_numbers)
    m.data.seek_table.points = pts
    m._keep = pts
    return m


def vorbis_comment(comments, vendor=b"someone else's vendor string"):
    ents = (VCEntry * len(comments))()
    for i, c in enumerate(comments):
        ents[i].length = len(c); ents[i].entry = c
    m = StreamMetadata(); m.type = 4
    m.data.vorbis_comment.vendor_string.length = len(vendor)
    m.data.vorbis_comment.vendor_string.entry = vendor
    m.data.vorbis_comment.num_comments = len(comments)
    m.data.vorbis_comment.comments = ents
    m.length = 4 + len(vendor) + 4 + sum(4 + len(c) for c in comments)
    m._keep = (ents, comments, vendor)
    return m


def picture(ptype, mime, desc, w, h, data):
    m = StreamMetadata(); m.type = 6
    p = m.data.picture
    p.type, p.mime_type, p.description, p.width, p.height, p.depth, p.colors = ptype, mime, desc, w, h, 24, 0
    p.data_length, p.data = len(data), data
    m.length = 4 + 4 + len(mime) + 4 + len(desc) + 16 + 4 + len(data)
    m._keep = (mime, desc, data)
    return m


def load(which):
    lib = C.CDLL(GPU_SO if which == "gpu" else REF_SO)      # RTLD_LOCAL: the two libraries export the same names
    lib.FLAC__stream_encoder_new.restype = C.c_void_p
    for n in ("delete", "finish", "get_state", "get_blocksize", "get_channels", "get_bits_per_sample", "get_sample_rate",
              "get_max_lpc_order", "get_qlp_coeff_precision", "get_do_mid_side_stereo", "get_loose_mid_side_stereo",
              "get_min_residual_partition_order", "get_max_residual_partition_order", "get_num_threads", "get_verify",
              "get_streamable_subset", "get_limit_min_bitrate", "get_do_exhaustive_model_search",
              "get_do_qlp_coeff_prec_search", "get_do_escape_coding", "get_rice_parameter_search_dist",
              "get_verify_decoder_state"):
        getattr(lib, "FLAC__stream_encoder_" + n).argtypes = [C.c_void_p]
    lib.FLAC__stream_encoder_get_total_samples_estimate.argtypes = [C.c_void_p]
    lib.FLAC__stream_encoder_get_total_samples_estimate.restype = C.c_uint64
    lib.FLAC__stream_encoder_get_resolved_state_string.argtypes = [C.c_void_p]
    lib.FLAC__stream_encoder_get_resolved_state_string.restype = C.c_char_p
    for n in ("set_verify", "set_streamable_subset", "set_channels", "set_bits_per_sample", "set_sample_rate",
              "set_compression_level", "set_blocksize", "set_do_mid_side_stereo", "set_loose_mid_side_stereo",
              "set_max_lpc_order", "set_qlp_coeff_precision", "set_do_qlp_coeff_prec_search", "set_do_escape_coding",
              "set_do_exhaustive_model_search", "set_min_residual_partition_order", "set_max_residual_partition_order",
              "set_num_threads", "set_rice_parameter_search_dist", "set_limit_min_bitrate", "set_do_md5",
              "disable_constant_subframes", "disable_fixed_subframes", "disable_verbatim_subframes",
              "disable_instruction_set"):
        getattr(lib, "FLAC__stream_encoder_" + n).argtypes = [C.c_void_p, C.c_uint32]
    lib.FLAC__stream_encoder_set_ogg_serial_number.argtypes = [C.c_void_p, C.c_long]
    lib.FLAC__stream_encoder_set_total_samples_estimate.argtypes = [C.c_void_p, C.c_uint64]
    lib.FLAC__stream_encoder_set_apodization.argtypes = [C.c_void_p, C.c_char_p]
    lib.FLAC__stream_encoder_set_metadata.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    lib.FLAC__stream_encoder_init_stream.argtypes = [C.c_void_p, WRITE_CB, SEEK_CB, TELL_CB, META_CB, C.c_void_p]
    lib.FLAC__stream_encoder_init_ogg_stream.argtypes = [C.c_void_p, C.c_void_p, WRITE_CB, SEEK_CB, TELL_CB, META_CB, C.c_void_p]
    lib.FLAC__stream_encoder_init_file.argtypes = [C.c_void_p, C.c_char_p, PROGRESS_CB, C.c_void_p]
    lib.FLAC__stream_encoder_init_ogg_file.argtypes = [C.c_void_p, C.c_char_p, PROGRESS_CB, C.c_void_p]
    lib.FLAC__stream_encoder_process_interleaved.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    lib.FLAC__stream_encoder_process.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    return lib


_libs = {}


def lib_for(which):
    if which not in _libs:
        _libs[which] = load(which)
    return _libs[which]


class Sink:
    """an in-memory seekable output: what a client's write/seek/tell callbacks would do with a file"""

    def __init__(self, seekable=True):
        self.buf = io.BytesIO()
        self.seekable = seekable
        self.calls = []          # (bytes, samples, current_frame) per write callback
        self.streaminfo = None

    def callbacks(self):
        def w(enc, data, n, samples, frame, cd):
            self.buf.write(C.string_at(data, n))
            self.calls.append((n, samples, frame))
            return 0

        def s(enc, off, cd):
            self.buf.seek(off)
            return 0

        def t(enc, poff, cd):
            poff[0] = self.buf.tell()
            return 0

        def m(enc, meta, cd):
            self.streaminfo = bytes(C.string_at(meta, C.sizeof(StreamMetadata)))

        self._cbs = (WRITE_CB(w), SEEK_CB(s) if self.seekable else SEEK_CB(), TELL_CB(t) if self.seekable else TELL_CB(), META_CB(m))
        return self._cbs


READ_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_size_t), C.c_void_p)


def encode(which, pcm, bps, rate, level=5, chunk=None, planar=False, metadata=None, settings=(), seekable=True,
           total_samples_estimate=None, to_file=None, progress=None, ogg=False):
    """One complete client session: new -> set_* -> init -> process* -> finish -> delete.
    settings: sequence of (setter-name-without-prefix, value) applied in order after the compression level.
    Returns (file bytes, Sink or None)."""
    lib = lib_for(which)
    pcm = np.ascontiguousarray(pcm, dtype=np.int32)
    n, ch = pcm.shape
    e = lib.FLAC__stream_encoder_new()
    assert e
    try:
        assert lib.FLAC__stream_encoder_set_channels(e, ch)
        assert lib.FLAC__stream_encoder_set_bits_per_sample(e, bps)
        assert lib.FLAC__stream_encoder_set_sample_rate(e, rate)
        assert lib.FLAC__stream_encoder_set_compression_level(e, level)
        for name, val in settings:
            fn = getattr(lib, "FLAC__stream_encoder_" + name)
            fn(e, val)
        if total_samples_estimate is not None:
            assert lib.FLAC__stream_encoder_set_total_samples_estimate(e, total_samples_estimate)
        keep = None
        if metadata:
            keep = (C.POINTER(StreamMetadata) * len(metadata))(*[C.pointer(m) for m in metadata])
            assert lib.FLAC__stream_encoder_set_metadata(e, keep, len(metadata))
        sink = None
        if to_file:
            pcb = PROGRESS_CB(progress) if progress else PROGRESS_CB()
            st = (lib.FLAC__stream_encoder_init_ogg_file if ogg else lib.FLAC__stream_encoder_init_file)(e, to_file.encode(), pcb, None)
        elif ogg:
            sink = Sink(seekable)

            def rd(enc, buf, pbytes, cd):              # FLAC__StreamEncoderReadCallback: 0 CONTINUE, 1 END_OF_STREAM
                data = sink.buf.read(pbytes[0])
                C.memmove(buf, data, len(data))
                pbytes[0] = len(data)
                return 0 if data else 1

            keep_rd = READ_CB(rd)
            lib.FLAC__stream_encoder_init_ogg_stream.argtypes = [C.c_void_p, READ_CB, WRITE_CB, SEEK_CB, TELL_CB, META_CB, C.c_void_p]
            st = lib.FLAC__stream_encoder_init_ogg_stream(e, keep_rd, *sink.callbacks(), None)
        else:
            sink = Sink(seekable)
            st = lib.FLAC__stream_encoder_init_stream(e, *sink.callbacks(), None)
        if st != 0:
            raise RuntimeError("init status %d, state %s" % (st, lib.FLAC__stream_encoder_get_resolved_state_string(e).decode()))
        step = chunk or max(n, 1)
        pos = 0
        k = 0
        while pos < n:
            c = step if isinstance(step, int) else step[k % len(step)]
            k += 1
            part = pcm[pos:pos + c]
            if planar:
                cols = [np.ascontiguousarray(part[:, i]) for i in range(ch)]
                ptrs = (C.c_void_p * ch)(*[col.ctypes.data for col in cols])
                ok = lib.FLAC__stream_encoder_process(e, ptrs, len(part))
            else:
                part = np.ascontiguousarray(part)
                ok = lib.FLAC__stream_encoder_process_interleaved(e, part.ctypes.data, len(part))
            if not ok:
                raise RuntimeError("process failed: %s" % lib.FLAC__stream_encoder_get_resolved_state_string(e).decode())
            pos += len(part)
        if not lib.FLAC__stream_encoder_finish(e):
            raise RuntimeError("finish failed")
        if to_file:
            with open(to_file, "rb") as f:
                return f.read(), None
        return sink.buf.getvalue(), sink
    finally:
        lib.FLAC__stream_encoder_delete(e)
