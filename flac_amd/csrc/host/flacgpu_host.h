/* flac_amd/csrc/host/flacgpu_host.h -- internal header of the host-side C layer (libFLACgpu.so):
 * encoder settings as the reference's FLAC__StreamEncoderProtected holds them
 * (src/libFLAC/include/protected/stream_encoder.h:40-130), their resolution at init time, the
 * window tables, MD5 and the stream/metadata writer.  Everything here is host C; the per-block hot
 * path is behind include/flacgpu.h. */
#ifndef FLACGPU_HOST_H
#define FLACGPU_HOST_H
#include <stdint.h>
#include <stddef.h>
#include "flacgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* same set and order as FLAC__ApodizationFunction (protected/stream_encoder.h:40-60) */
typedef enum {
	FGH_APOD_BARTLETT, FGH_APOD_BARTLETT_HANN, FGH_APOD_BLACKMAN, FGH_APOD_BLACKMAN_HARRIS_4TERM_92DB_SIDELOBE,
	FGH_APOD_CONNES, FGH_APOD_FLATTOP, FGH_APOD_GAUSS, FGH_APOD_HAMMING, FGH_APOD_HANN, FGH_APOD_KAISER_BESSEL,
	FGH_APOD_NUTTALL, FGH_APOD_RECTANGLE, FGH_APOD_TRIANGLE, FGH_APOD_TUKEY, FGH_APOD_PARTIAL_TUKEY,
	FGH_APOD_PUNCHOUT_TUKEY, FGH_APOD_SUBDIVIDE_TUKEY, FGH_APOD_WELCH
} flacgpu_host_apod_type;

typedef struct {
	flacgpu_host_apod_type type;
	float p;            /* tukey p / gauss stddev / multiple_tukey p / subdivide_tukey p (already divided by parts) */
	float start, end;   /* partial / punchout tukey */
	int32_t parts;      /* subdivide_tukey */
} flacgpu_host_apodization;

#define FGH_MAX_APODIZATIONS 32

typedef struct {
	uint32_t channels, bits_per_sample, sample_rate, blocksize;
	int streamable_subset, do_md5, verify;
	int do_mid_side_stereo, loose_mid_side_stereo;
	uint32_t max_lpc_order, qlp_coeff_precision;
	int do_qlp_coeff_prec_search, do_escape_coding, do_exhaustive_model_search;
	uint32_t min_residual_partition_order, max_residual_partition_order, rice_parameter_search_dist;
	uint32_t num_apodizations;
	flacgpu_host_apodization apodizations[FGH_MAX_APODIZATIONS];
	int limit_min_bitrate;
	int disable_constant_subframes, disable_fixed_subframes, disable_verbatim_subframes;
	uint64_t total_samples_estimate;
} flacgpu_host_settings;

/* init status codes: numerically equal to FLAC__StreamEncoderInitStatus (stream_encoder.h:283-337) */
enum {
	FGH_INIT_OK = 0, FGH_INIT_ENCODER_ERROR = 1, FGH_INIT_UNSUPPORTED_CONTAINER = 2, FGH_INIT_INVALID_CALLBACKS = 3,
	FGH_INIT_INVALID_NUMBER_OF_CHANNELS = 4, FGH_INIT_INVALID_BITS_PER_SAMPLE = 5, FGH_INIT_INVALID_SAMPLE_RATE = 6,
	FGH_INIT_INVALID_BLOCK_SIZE = 7, FGH_INIT_INVALID_MAX_LPC_ORDER = 8, FGH_INIT_INVALID_QLP_COEFF_PRECISION = 9,
	FGH_INIT_BLOCK_SIZE_TOO_SMALL_FOR_LPC_ORDER = 10, FGH_INIT_NOT_STREAMABLE = 11, FGH_INIT_INVALID_METADATA = 12,
	FGH_INIT_ALREADY_INITIALIZED = 13
};

void flacgpu_host_settings_defaults(flacgpu_host_settings *s);              /* set_defaults_, stream_encoder.c:2628 */
void flacgpu_host_settings_level(flacgpu_host_settings *s, uint32_t level); /* set_compression_level, :1873 */
void flacgpu_host_settings_apodization(flacgpu_host_settings *s, const char *spec); /* set_apodization, :1940 */
int  flacgpu_host_settings_resolve(flacgpu_host_settings *s);               /* checks/defaults of init_stream_internal_, :723-829 */

/* maps resolved settings onto the engine configuration; returns 0 or FLACGPU_ERR_UNSUPPORTED when the
 * settings need a feature the GPU engine does not implement (there is no CPU fallback) */
int  flacgpu_host_engine_config(const flacgpu_host_settings *s, int device, uint32_t max_batch_frames, flacgpu_config *out);

/* window tables for `blocksize`: out[num_apodizations][blocksize] (resize_buffers_, :2913-2977) */
void flacgpu_host_window(const flacgpu_host_apodization *a, float *w, int32_t L);
void flacgpu_host_windows(const flacgpu_host_settings *s, uint32_t blocksize, float *out);

/* MD5 (RFC 1321) of the interleaved little-endian PCM, as STREAMINFO wants it (md5.c:497, :280) */
typedef struct { uint32_t state[4]; uint64_t nbytes; uint8_t block[64]; } flacgpu_host_md5;
void flacgpu_host_md5_init(flacgpu_host_md5 *m);
void flacgpu_host_md5_update(flacgpu_host_md5 *m, const void *data, size_t len);
void flacgpu_host_md5_final(flacgpu_host_md5 *m, uint8_t digest[16]);
/* eight chains at once (AVX2, one per 32-bit lane): a corpus is many streams, each with its own digest */
int flacgpu_host_md5_x8_available(void);
void flacgpu_host_md5_x8_blocks(flacgpu_host_md5 *const m[8], const void *const data[8], size_t nblocks);    /* every context at a block boundary */
void flacgpu_host_md5_many(const void *const *data, const size_t *len, uint32_t n, uint8_t (*digest)[16]);
int flacgpu_host_md5_x16_available(void);
void flacgpu_host_md5_many_mt(const void *const *data, const size_t *len, uint32_t n, uint8_t (*digest)[16], uint32_t nthreads);   /* the same on host threads, a group of chains per work item */
/* feeds `samples` inter-channel samples of interleaved int32 PCM as bytes_per_sample-byte little-endian */
void flacgpu_host_md5_pcm(flacgpu_host_md5 *m, const int32_t *interleaved, uint32_t channels, size_t samples, uint32_t bytes_per_sample);

/* the encoder's self check (set_verify): decode the frames of a batch again and compare them with the samples that
 * went in (`raw`: little endian, `width` bytes per sample, interleaved).  status 0 ok, 1 audio mismatch (the fields
 * locate the first one in stream order), 2 a frame does not decode.  Returns the status. */
typedef struct { int status; uint32_t frame_number, channel, sample; uint64_t absolute_sample; int32_t expected, got; } flacgpu_host_verify_result;
int flacgpu_host_verify_batch(const flacgpu_host_settings *s, const uint8_t *frames, const uint32_t *frame_bytes, uint32_t nframes, uint32_t tail,
                              uint32_t first_frame, const uint8_t *raw, uint32_t width, uint32_t nthreads, flacgpu_host_verify_result *out);

/* CRC-16 footers of `nframes` frames laid out back to back: index of the first wrong one, or -1 */
int64_t flacgpu_host_check_frame_crcs(const uint8_t *frames, const uint32_t *frame_bytes, uint64_t nframes, uint32_t nthreads);

#ifdef __cplusplus
}
#endif
#endif
