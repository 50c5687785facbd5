/* include/flacgpu.h -- C ABI of the MI355X FLAC frame engine (libflacgpu.so).
 *
 * This is the drop-in boundary: it replaces exactly what the reference's
 * per-frame worker does,
 *     process_subframes_() + zero-pad + CRC-16        src/libFLAC/stream_encoder.c:3705-3744
 *     (-> process_subframe_ :4045, apply_apodization_ :4318, evaluate_*_subframe_ :4466-4699,
 *         find_best_partition_order_ :4701, FLAC__frame_add_header framing.c:245,
 *         FLAC__subframe_add_* framing.c:393-520, FLAC__bitwriter_write_rice_signed_block
 *         bitwriter.c:575)
 * for a BATCH of independent frames instead of one frame per thread-pool task
 * (stream_encoder.c:3490-3614).  Input is what FLAC__stream_encoder_process_interleaved()
 * (stream_encoder.c:2570) receives -- int32 inter-channel samples -- and the output is the
 * byte-aligned frames that write_bitbuffer_()/write_frame_() (stream_encoder.c:2988,3038) hand
 * to the client's write callback, in stream order.
 *
 * Plain C: pointers and sizes only.  Host-side libFLAC API (FLAC__stream_encoder_*) lives in
 * libFLACgpu.so (flac_amd/csrc/host/), which calls these entry points.
 */
#ifndef FLACGPU_H
#define FLACGPU_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FLACGPU_ABI_VERSION 5
#define FLACGPU_MAX_CHANNELS 8
#define FLACGPU_MAX_APODIZATIONS 32     /* FLAC__MAX_APODIZATION_FUNCTIONS */

/* error codes (negative returns) */
enum {
	FLACGPU_OK = 0,
	FLACGPU_ERR_UNSUPPORTED = -1,   /* configuration outside the engine's range: channels / bits / block size / orders beyond the format's
	                                   limits, or an apodization list that expands to more than 1024 window jobs or 2048 LPC analyses
	                                   per subframe (e.g. two subdivide_tukey(32) in one list: the reference accepts any list of up to
	                                   32 functions, stream_encoder.c:1940-2070; one subdivide_tukey(32) is 528 jobs / 1053 analyses) */
	FLACGPU_ERR_NO_DEVICE = -2,     /* no HIP device / kernel image not loadable: there is NO CPU fallback */
	FLACGPU_ERR_ALLOC = -3,         /* -> FLAC__STREAM_ENCODER_MEMORY_ALLOCATION_ERROR */
	FLACGPU_ERR_OUTPUT_TOO_SMALL = -4,
	FLACGPU_ERR_LAUNCH = -5,        /* -> FLAC__STREAM_ENCODER_FRAMING_ERROR */
	FLACGPU_ERR_BAD_ARG = -6,
	FLACGPU_ERR_INPUT = -7,         /* raw input violates its declared format (non-zero bits below `shift`) */
	FLACGPU_ERR_BUSY = -8           /* flacgpu_submit_batch_raw: FLACGPU_ASYNC_SLOTS batches are in flight -- collect one first */
};

/* apodization kinds as the frame engine sees them (stream_encoder.c:4318-4392): every window
 * function is a host-computed float table (window.c stays host side); SUBDIVIDE_TUKEY adds the
 * partial / punch-out state machine of set_next_subdivide_tukey (stream_encoder.c:4293). */
enum { FLACGPU_APOD_WINDOW = 0, FLACGPU_APOD_SUBDIVIDE_TUKEY = 1 };

typedef struct {
	uint32_t kind;      /* FLACGPU_APOD_* */
	uint32_t parts;     /* SUBDIVIDE_TUKEY only */
} flacgpu_apodization;

/* Immutable encoder settings: the fields of FLAC__StreamEncoderProtected that
 * process_subframes_ reads (src/libFLAC/include/protected/stream_encoder.h:91-130), already
 * resolved the way init_stream_internal_ resolves them (stream_encoder.c:723-829). */
typedef struct {
	uint32_t abi_version;            /* FLACGPU_ABI_VERSION */
	uint32_t channels;               /* 1..8 */
	uint32_t bits_per_sample;        /* 4..32 (at 32 the side channel has 33 bits, stream_encoder.c:3831-3835) */
	uint32_t sample_rate;
	uint32_t blocksize;              /* 16..65535 */
	uint32_t do_mid_side_stereo;
	uint32_t loose_mid_side_stereo;
	uint32_t max_lpc_order;          /* 0..32; below 16 the FMA autocorrelation routines (stream_encoder.c:1058-1066), from 16 the C loop (lpc.c:133) */
	uint32_t qlp_coeff_precision;    /* resolved, 5..15 */
	uint32_t min_residual_partition_order;
	uint32_t max_residual_partition_order; /* <= 8 */
	uint32_t num_apodizations;       /* 1..FLACGPU_MAX_APODIZATIONS */
	flacgpu_apodization apodizations[FLACGPU_MAX_APODIZATIONS];
	uint32_t disable_constant_subframes, disable_fixed_subframes, disable_verbatim_subframes;
	uint32_t limit_min_bitrate;
	int32_t  device;                 /* HIP device ordinal */
	uint32_t max_batch_frames;       /* capacity of one encode call */
	/* ABI 2: the wider model searches of process_subframe_ (stream_encoder.c:4155-4163, 4220-4243); ABI 3: 32 apodizations */
	uint32_t do_exhaustive_model_search;  /* -e: every fixed order 0..4 and every LPC order 1..max_lpc_order per analysis */
	uint32_t do_qlp_coeff_prec_search;    /* -p: every coefficient precision 5..max per LPC order */
} flacgpu_config;

typedef struct flacgpu_ctx flacgpu_ctx;

/* per-frame diagnostics (optional output), mirrors what FLAC__Frame/FLAC__Subframe would hold */
typedef struct {
	uint8_t  type;        /* 0 CONSTANT 1 VERBATIM 2 FIXED 3 LPC */
	uint8_t  order, wasted_bits, partition_order, rice2, precision;
	int8_t   shift;
	uint8_t  pad;
	uint32_t bits;
} flacgpu_subframe_info;

/* Creates an engine for one stream configuration.
 * windows: num_apodizations tables of `blocksize` floats each, concatenated -- what
 * resize_buffers_ computes with FLAC__window_* (stream_encoder.c:2913-2977).
 * Returns 0 or a negative FLACGPU_ERR_*. */
int flacgpu_create(const flacgpu_config *cfg, const float *windows, flacgpu_ctx **out);
void flacgpu_destroy(flacgpu_ctx *ctx);

/* Bytes an output buffer must hold for `nframes` frames in the worst case (all VERBATIM). */
size_t flacgpu_max_output_bytes(const flacgpu_ctx *ctx, uint32_t nframes);

/* Encode a batch of consecutive frames from HOST memory.
 *   pcm                interleaved int32 [(nframes-1)*blocksize + last][channels], stream order
 *   nframes            <= max_batch_frames
 *   first_frame_number frame number of the first frame in the batch
 *   last_block_samples 0: every frame has `blocksize` samples; else the LAST frame of the batch
 *                      has this many (the short final block, stream_encoder.c:1703-1711)
 *   tail_windows       windows recomputed for last_block_samples (NULL when it is 0)
 *   out/out_cap        receives the frames back to back
 *   frame_bytes        [nframes] length of each frame
 * Returns total bytes written (>= 0) or a negative FLACGPU_ERR_*. */
int64_t flacgpu_encode_batch(flacgpu_ctx *ctx, const int32_t *pcm, uint32_t nframes,
                             uint64_t first_frame_number, uint32_t last_block_samples,
                             const float *tail_windows, uint8_t *out, size_t out_cap,
                             uint32_t *frame_bytes);

/* Same, with every buffer already resident in DEVICE memory (HBM): d_pcm, d_out and
 * d_frame_bytes are device pointers; d_total_bytes (device, 8 bytes) receives the byte total.
 * The kernels write both in place (round 5): d_frame_bytes needs 4-byte, d_total_bytes 8-byte alignment, nothing beyond the
 * batch's nframes entries is touched, and their contents are undefined until the batch has completed on `stream` (either may be
 * NULL: the engine then writes its own arrays, which no entry point exposes).  Later kernels of the SAME batch read both back
 * (offsets are built from the lengths): a caller that overwrites them while the batch is in flight corrupts that batch's layout.
 * Misaligned pointers: FLACGPU_ERR_BAD_ARG.
 * Asynchronous on `stream` (a hipStream_t passed as void*, NULL = default stream); returns 0 or
 * a negative code.  This is the entry bench.py times. */
int flacgpu_encode_batch_device(flacgpu_ctx *ctx, const int32_t *d_pcm, uint32_t nframes,
                                uint64_t first_frame_number, uint32_t last_block_samples,
                                const float *tail_windows_host, uint8_t *d_out, size_t out_cap,
                                uint32_t *d_frame_bytes, uint64_t *d_total_bytes, void *stream);

/* ---- input staging on the device (what format_input() of the reference's `flac` tool does on the host,
 * src/flac/encode.c:2352-2492): raw interleaved sample bytes as they sit in a WAVE/AIFF/raw file -> int32 ---- */
typedef struct {
	uint32_t container_bits;         /* 8, 16, 24 or 32: bits each sample occupies in the raw data */
	uint32_t big_endian;             /* byte order of the raw data */
	uint32_t is_unsigned;            /* unsigned samples: the mid-point is subtracted */
	uint32_t shift;                  /* samples are left-justified: low `shift` bits must be zero and are dropped */
	uint32_t use_channel_map;        /* 0: identity */
	uint8_t  channel_map[FLACGPU_MAX_CHANNELS]; /* input channel c goes to output channel channel_map[c] */
} flacgpu_raw_format;

/* d_raw (device) holds wide_samples * channels samples in `fmt`; writes the interleaved int32 block to d_pcm (device).
 * d_error (device uint32, may be NULL, caller zeroes it) gets bit 0 set when a sample has non-zero bits below
 * `shift` (encode.c:2479-2488).  Asynchronous on `stream`. */
int flacgpu_stage_raw_device(flacgpu_ctx *ctx, const void *d_raw, const flacgpu_raw_format *fmt, uint64_t wide_samples,
                             int32_t *d_pcm, uint32_t *d_error, void *stream);

/* The MD5 digests of n byte ranges of DEVICE memory, d_base + offsets[i] .. + lengths[i] (any alignment, any length, zero
 * included) -- the sample bytes of n streams as they lie staged in HBM: what FLAC__stream_encoder_finish() puts into each
 * stream's STREAMINFO (the reference hashes them on the host as they pass, src/libFLAC/stream_encoder.c:3448, :3666-3686;
 * src/libFLAC/md5.c:60-222 is the transform).  One lane per stream: worth it for a corpus of many streams, not for one.
 * offsets, lengths: host arrays [n]; digests: host [n][16].  Synchronous on `stream` (may be NULL).
 * d_base must be 4-byte aligned (FLACGPU_ERR_BAD_ARG otherwise; the ranges themselves may start anywhere); the ranges must
 * lie inside the caller's allocation.  device: 0..63 (FLACGPU_ERR_BAD_ARG beyond: the per-device scratch is a fixed table). */
int flacgpu_md5_many_device(int device, const void *d_base, const uint64_t *offsets, const uint64_t *lengths, uint32_t n,
                            uint8_t *digests, void *stream);

/* flacgpu_encode_batch() from raw HOST sample bytes: copies the raw bytes (2 bytes per 16-bit sample instead of 4),
 * stages them on the device and encodes.  Returns FLACGPU_ERR_INPUT for a shift violation. */
int64_t flacgpu_encode_batch_raw(flacgpu_ctx *ctx, const void *raw, const flacgpu_raw_format *fmt, uint32_t nframes,
                                 uint64_t first_frame_number, uint32_t last_block_samples,
                                 const float *tail_windows, uint8_t *out, size_t out_cap,
                                 uint32_t *frame_bytes);

/* ---- the encoder's self check on the device (FLAC__stream_encoder_set_verify: the reference decodes every frame it wrote
 * with its own stream decoder and compares, write_bitbuffer_ stream_encoder.c:3000-3018, verify_write_callback_ :5155-5230;
 * the frame reader it uses is stream_decoder.c:2373-3400, the restoration lpc.c:978-1578 / fixed.c:571-667) ---- */
typedef struct {
	int32_t  status;           /* 0 every frame decodes back to its input; 1 audio mismatch (the fields below locate the first one
	                              in stream order, as FLAC__stream_encoder_get_verify_decoder_error_stats reports it);
	                              2 a frame does not decode (bad CRC, malformed header or subframe, wrong frame number / length) */
	uint32_t frame_number;     /* of the first bad frame */
	uint32_t channel, sample;  /* status 1: output channel and sample index within the block */
	uint64_t absolute_sample;  /* frame_number * blocksize + sample */
	int32_t  expected, got;
} flacgpu_verify_result;

/* Decode `nframes` frames lying back to back at d_frames (lengths d_frame_bytes, device) and compare them with d_pcm, the
 * interleaved int32 input of exactly these frames (device; the short last block as in flacgpu_encode_batch).  Frames must
 * carry the numbers first_frame_number, first_frame_number + 1, ...  The verdict goes to d_result (device).  Asynchronous
 * on `stream`.  Returns 0 or a negative FLACGPU_ERR_*. */
int flacgpu_verify_batch_device(flacgpu_ctx *ctx, const uint8_t *d_frames, const uint32_t *d_frame_bytes, uint32_t nframes,
                                uint64_t first_frame_number, uint32_t last_block_samples, const int32_t *d_pcm,
                                flacgpu_verify_result *d_result, void *stream);
/* on != 0: every batch encoded through flacgpu_encode_batch / flacgpu_encode_batch_raw is verified on the device before the
 * call returns; flacgpu_last_verify_result gives the verdict of the most recent batch (status 0 when verification is off). */
int flacgpu_set_verify(flacgpu_ctx *ctx, uint32_t on);
int flacgpu_last_verify_result(flacgpu_ctx *ctx, flacgpu_verify_result *out);

/* ---- decoding streams the engine did NOT write (`flac -t`, `flac -d`): what the reference's
 * FLAC__stream_decoder_process_until_end_of_stream (src/libFLAC/stream_decoder.c:1168) hands its client for a stream -- every
 * sample of every write callback, every error callback in order -- for a byte range in DEVICE memory, PCM left in DEVICE memory.
 * Frames are found by their sync codes (frame_sync_ :2321), headers checked with their CRC-8 (read_frame_header_ :2624), bodies
 * decoded one lane per frame (read_subframe_* :2949-3360, lpc.c:978-1578, fixed.c:571-667), CRC-16 and sample bounds checked
 * (read_frame_ :2373-2483), frames that are missing made up for with silence as the reference does (:2485-2554).  Any bytes are
 * legal input; a damaged stream yields the reference's errors (one documented exception: FLACGPU_DECODE long_rice_codes).
 * No FLAC__stream_decoder_* layer on top (out of scope) -- metadata is the caller's (flacgpu_probe_stream). ---- */
typedef struct {
	uint32_t has_streaminfo;         /* 0: the fields below are unknown (frames must then carry rate and sample size themselves) */
	uint32_t min_blocksize, max_blocksize, sample_rate, channels, bits_per_sample;
} flacgpu_stream_info;

/* error callbacks, as FLAC__StreamDecoderErrorStatus + 1 (include/FLAC/stream_decoder.h:277-310) */
enum {
	FLACGPU_DECODE_LOST_SYNC = 1, FLACGPU_DECODE_BAD_HEADER = 2, FLACGPU_DECODE_FRAME_CRC_MISMATCH = 3, FLACGPU_DECODE_UNPARSEABLE_STREAM = 4,
	FLACGPU_DECODE_BAD_METADATA = 5, FLACGPU_DECODE_OUT_OF_BOUNDS = 6, FLACGPU_DECODE_MISSING_FRAME = 7
};
typedef struct {
	uint32_t status;                 /* FLACGPU_DECODE_* */
	uint32_t pad;
	uint64_t byte_offset;            /* the sync code the error belongs to, or where the search that skipped bytes began */
} flacgpu_decode_event;

typedef struct {
	uint64_t samples;                /* inter-channel samples the client receives (= written to d_pcm), silence included */
	uint64_t frames;                 /* frames decoded into them */
	uint64_t silence_samples;        /* of `samples`: stand-ins for missing frames */
	uint64_t candidates;             /* sync codes looked at */
	uint64_t deferred_decoded;       /* sync codes inside frame data whose header happens to hold but does not look like this stream's, decoded after
	                                    all because the search got to one (0 for a whole stream) */
	uint64_t redecoded_frames;       /* frames decoded a second time, into a place other than the one their number implies (0 for a whole stream) */
	uint32_t nevents;                /* error callbacks; the first max_events of them are in `events` */
	uint32_t end_in_header;          /* the stream ended inside a frame header: the reference's process call returns false */
	uint32_t format_changes;         /* good frames whose channel count / sample size is not the stream's: reported, not written */
	uint32_t long_rice_codes;        /* frames given up for a Rice code whose unary part exceeds what a 32-bit residual allows.  The
	                                    reference applies this limit only to codes that do not straddle a refill of its 8 KiB read
	                                    buffer (bitreader_read_rice_signed_block.c); this decoder always.  Non-zero: the error list may
	                                    differ from the reference's for those frames. */
	uint32_t channels, bits_per_sample, sample_rate;   /* the stream's format: STREAMINFO, else the first good frame */
	uint32_t errors_by_status[8];    /* count per FLACGPU_DECODE_* */
	float ms_scan, ms_decode, ms_place, ms_total;       /* HIP-event times: sync scan; decode + CRC + finish; re-decodes + silence; all */
} flacgpu_decode_result;

typedef struct flacgpu_decoder flacgpu_decoder;
int flacgpu_decoder_create(int device, flacgpu_decoder **out);    /* scratch buffers grow with use and are kept between calls */
void flacgpu_decoder_destroy(flacgpu_decoder *dec);

/* HOST: where the audio frames of a FLAC file begin and what its STREAMINFO says, from the file's first bytes (ID3v2 tags in front
 * are stepped over; a stream that does not start with "fLaC" is taken as bare frames from byte 0).  total_samples, md5 (16 bytes) may
 * be NULL.  FLACGPU_ERR_INPUT: the metadata runs past nbytes. */
int flacgpu_probe_stream(const uint8_t *head, size_t nbytes, flacgpu_stream_info *si, uint64_t *first_frame_offset, uint64_t *total_samples, uint8_t *md5);

/* Decode d_stream[first_frame_offset .. nbytes) (DEVICE memory; d_stream 4-byte aligned, its allocation extending to the next multiple
 * of 4 bytes) into d_pcm (DEVICE, interleaved int32 [samples][channels], room for pcm_capacity_values int32 values -- the channel count is
 * the stream's, which without STREAMINFO only the decode tells; NULL: verdict only).  si may be NULL.  events: host array [max_events] (may be NULL).  Synchronous.
 * Returns 0, FLACGPU_ERR_OUTPUT_TOO_SMALL (result->samples x result->channels says what is needed; d_pcm's contents are then undefined) or another
 * negative FLACGPU_ERR_*. */
int flacgpu_decode_stream_device(flacgpu_decoder *dec, const uint8_t *d_stream, uint64_t nbytes, uint64_t first_frame_offset, const flacgpu_stream_info *si,
                                 int32_t *d_pcm, uint64_t pcm_capacity_values, flacgpu_decode_result *result, flacgpu_decode_event *events, uint32_t max_events,
                                 void *stream);
/* int32 samples (DEVICE) -> the bytes a WAVE file and the MD5 of STREAMINFO hold: little endian, (bits_per_sample + 7) / 8 bytes each
 * (what FLAC__MD5Accumulate is fed, src/libFLAC/md5.c:497; flacgpu_stage_raw_device's inverse).  Asynchronous on `stream`. */
int flacgpu_pack_samples_device(int device, const int32_t *d_pcm, uint64_t nvalues, uint32_t bits_per_sample, uint8_t *d_out, void *stream);

/* Diagnostics of the most recent batch: [nframes][channels] subframe choices and the channel
 * assignment per frame (0 independent, 1 left/side, 2 right/side, 3 mid/side). Host arrays. */
int flacgpu_last_batch_info(flacgpu_ctx *ctx, uint32_t nframes, flacgpu_subframe_info *sub,
                            uint8_t *channel_assignment);

/* Wall-clock-free timing of the last batch, measured with HIP events on the engine's stream:
 * milliseconds spent in the analysis kernel, the pack kernel and the compaction kernels. */
int flacgpu_last_batch_kernel_ms(flacgpu_ctx *ctx, float *analyze_ms, float *pack_ms, float *compact_ms);
/* The same per kernel, in launch order: ms[0] prep (wasted bits, fixed-predictor guess), ms[1] autocorrelation,
 * ms[2] LPC model (Levinson-Durbin, quantisation), ms[3] residual candidate evaluation + Rice search,
 * ms[4] pack, ms[5] scan + compaction. */
int flacgpu_last_batch_phase_ms(flacgpu_ctx *ctx, float ms[6]);
/* The same for the batch launched `batches_ago` batches before the last one (0 = the last; the engine keeps the
 * events of its 64 most recent batches), so that a run of batches can be timed without a host sync in between. */
int flacgpu_batch_phase_ms(flacgpu_ctx *ctx, uint32_t batches_ago, float ms[6]);
/* The phase times are instrumentation: six event records per batch, ~4.6 us of the stream each (1.4 % of a -8 step of 16384 frames,
 * a fifth of a -0 step).  every = 1 (default): each batch carries them; n: every n-th batch; 0: none.  flacgpu_batch_phase_ms
 * returns 1 (and zeros) for a batch that carried none.  The libFLAC API layer, which never reads them, runs with 0. */
int flacgpu_set_phase_timing(flacgpu_ctx *ctx, uint32_t every);
/* Which kernels the most recent batch launched: a bit per kernel (family / flavour); flacgpu_kernel_bit_name(bit) names bit
 * `bit` (NULL past the last).  The engine picks kernels by stream shape and batch size; tests that pin a
 * selection -- "the full-size -8 batch runs autoc3_kernel<SETS,PLANES> and the fused output" -- assert on this. */
int flacgpu_last_batch_kernels(const flacgpu_ctx *ctx, uint32_t *mask);
const char *flacgpu_kernel_bit_name(uint32_t bit);
/* The fused output (frames written once, at their final place) never waits unboundedly: a frame whose predecessors' lengths do
 * not turn up within FLACGPU_FUSED_SPIN_LIMIT polls goes to its slot and is placed by a kernel behind the pack kernel.  Such
 * frames cost time, never bytes; this counts them: all since the context was created, and the most of any one batch (zero in a
 * healthy run).  Synchronises the device. */
int flacgpu_fused_fallbacks(flacgpu_ctx *ctx, uint32_t *total, uint32_t *max_in_a_batch);
/* Split every batch into n (1..8) sub-batches that run on separate HIP streams inside the engine and join before the
 * frame compaction (default 1, or the FLACGPU_SUBBATCHES environment variable).  Output is identical. */
int flacgpu_set_subbatches(flacgpu_ctx *ctx, uint32_t n);

/* Page-locked host memory for PCM staging buffers handed to flacgpu_encode_batch (the H2D copy then runs
 * at full PCIe rate and asynchronously).  NULL when no device/runtime is available. */
void *flacgpu_alloc_pinned(size_t bytes);
void flacgpu_free_pinned(void *p);
/* The same for memory the caller already owns (page-locks it in place; contents untouched, writers may carry on).  The host API
 * layer fills ordinary memory while the HIP runtime is still starting on another thread and registers it afterwards. */
int flacgpu_host_register(void *p, size_t bytes);
void flacgpu_host_unregister(void *p);
/* ---- the asynchronous entry: what the task ring of the reference's frame-parallel encoder is for (stream_encoder.c:1134-1238,
 * 3530-3574: 2*threads+2 frames in flight, drained in order) -- here per BATCH.  flacgpu_submit_batch_raw() enqueues everything a
 * batch needs and returns at once: the raw bytes go to the device on a copy stream of their own, the kernels run on the engine's
 * stream behind them, the frame lengths and the byte total come back on a third.  Up to FLACGPU_ASYNC_SLOTS batches may be in
 * flight (each has its own device buffers for input and output; the kernels' scratch is shared: they run one batch after the
 * other anyway), so the copy of batch k+1 and the read-back of batch k-1 run beside the kernels of batch k.
 * flacgpu_collect() waits for the OLDEST batch in flight, copies its frames to the `out` given at submission and returns their
 * byte total (or the batch's error).  `raw`, `out` and `frame_bytes` must stay valid until the batch is collected; page-locked
 * memory (flacgpu_alloc_pinned / flacgpu_host_register) makes the copies truly asynchronous.  Arguments as for
 * flacgpu_encode_batch_raw.  Submissions and collections of one engine must come from one thread at a time. */
#define FLACGPU_ASYNC_SLOTS 4
int flacgpu_submit_batch_raw(flacgpu_ctx *ctx, const void *raw, const flacgpu_raw_format *fmt, uint32_t nframes,
                             uint64_t first_frame_number, uint32_t last_block_samples, const float *tail_windows,
                             uint8_t *out, size_t out_cap, uint32_t *frame_bytes);     /* 0, or a negative FLACGPU_ERR_* (nothing enqueued) */
int64_t flacgpu_collect(flacgpu_ctx *ctx);         /* bytes of the oldest batch, or a negative FLACGPU_ERR_* (FLACGPU_ERR_BAD_ARG: none in flight) */
int flacgpu_in_flight(const flacgpu_ctx *ctx);     /* batches submitted and not yet collected */

/* The configuration checks of flacgpu_create without a device: FLACGPU_OK, or the FLACGPU_ERR_UNSUPPORTED / _BAD_ARG that
 * flacgpu_create would return for this configuration.  Touches no HIP state. */
int flacgpu_config_check(const flacgpu_config *cfg);
/* flacgpu_max_output_bytes() before an engine exists: the same worst case from the configuration alone (0: cfg is NULL). */
size_t flacgpu_config_max_output_bytes(const flacgpu_config *cfg, uint32_t nframes);
/* 1 when the kernel driver's device node can be opened by this process -- a cheap hint that flacgpu_create may succeed, taken
 * WITHOUT starting the HIP runtime (which costs a fresh process 0.1-0.2 s); 0 means flacgpu_create will report NO_DEVICE. */
int flacgpu_device_probe(void);

/* Test hook (tests/test_log_pin.py): evaluates on the device, for n host arguments, mode 0 the engine's log (glibc 2.35's
 * algorithm restated, flac_amd/csrc/flacgpu_log.h), mode 1 the expected-bits expression of lpc.c:1591-1606 on
 * (lpc_error a, error_scale b), mode 2 the fixed-predictor estimate of fixed.c:284-288 on (total_error a, data_len b)
 * widened to double, mode 3 the device library's own log (informational).  Returns 0 or a negative FLACGPU_ERR_*. */
int flacgpu_debug_log_kat(int device, uint32_t mode, const double *a, const double *b, size_t n, double *out);
/* Development aid: how many frames of the most recent verify call were verified by the thread-per-run pass; the
 * rest went through the sequential decoder.  Synchronises the device. */
int flacgpu_debug_verify_hinted_frames(flacgpu_ctx *ctx, uint32_t *out);

/* Development aid (bench.py: the clock under each workload): one wavefront that, nsamples times, times a sleep of 130 048 shader
 * cycles against the constant 100 MHz counter.  d_out: device buffer [nsamples][4] uint64 {start, sleep in 100 MHz ticks, sleep in
 * s_memtime ticks, nominal cycles}.  Asynchronous on `stream`: launch it on a stream of its own beside the work to be observed. */
int flacgpu_debug_clock_probe(int device, void *stream, uint32_t nsamples, uint64_t *d_out);

const char *flacgpu_strerror(int code);
int flacgpu_device_count(void);

#ifdef __cplusplus
}
#endif
#endif
