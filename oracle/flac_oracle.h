/* oracle/flac_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement ("spec in code") of the reference libFLAC per-block encode path
 * (SURVEY.md section 8a).  Scalar C, strict IEEE (-ffp-contract=off), explicit reduction
 * trees where the reference's *compiled* behaviour defines the result.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * The product (flac_amd/) never links or calls it.
 *
 * Parity status: PINNED -- every function is differentially tested against the real
 * reference binary (oracle/_ref/libFLAC_ref.so, built by oracle/Makefile from
 * /root/reference) in tests/test_oracle_vs_ref.py, and against committed golden
 * digests of reference output in tests/golden/.
 */
#ifndef FLAC_ORACLE_H
#define FLAC_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FO_MAX_LPC_ORDER 32
#define FO_MAX_APODIZATIONS 32
#define FO_MAX_CHANNELS 8

/* apodization kinds the frame engine distinguishes (stream_encoder.c:4318-4392):
 * everything except SUBDIVIDE_TUKEY is "one full-length window table". */
enum { FO_APOD_WINDOW = 0, FO_APOD_SUBDIVIDE_TUKEY = 1 };

/* which compiled autocorrelation routine the reference would dispatch to
 * (stream_encoder.c:1058-1066 on an FMA-capable x86-64 host) */
enum { FO_AUTOC_FMA_LAG8 = 8, FO_AUTOC_FMA_LAG12 = 12, FO_AUTOC_FMA_LAG16 = 16, FO_AUTOC_GENERIC = 0 };

typedef struct {
	uint32_t kind;          /* FO_APOD_* */
	uint32_t parts;         /* SUBDIVIDE_TUKEY: number of parts */
	const float *window;    /* [blocksize] table for THIS frame's blocksize */
} fo_apodization;

typedef struct {
	uint32_t channels;              /* 1..8 */
	uint32_t bits_per_sample;       /* 4..24 in this restatement */
	uint32_t sample_rate;
	uint32_t blocksize;             /* samples in THIS frame (the last frame may be short) */
	uint32_t do_mid_side;           /* stream_encoder.c:3777 */
	uint32_t loose_mid_side;
	uint32_t max_lpc_order;         /* 0 => fixed predictors only */
	uint32_t qlp_coeff_precision;   /* resolved (non-zero) value, stream_encoder.c:764-795 */
	uint32_t min_partition_order, max_partition_order;
	uint32_t num_apodizations;
	fo_apodization apodizations[FO_MAX_APODIZATIONS];
	uint32_t autoc_variant;         /* FO_AUTOC_*; blocksize<=32 forces GENERIC (stream_encoder.c:2978) */
	uint32_t disable_constant, disable_fixed, disable_verbatim;
	uint32_t limit_min_bitrate;
	uint32_t exhaustive;            /* -e: every fixed order 0..4 and every LPC order 1..max (stream_encoder.c:4155-4163,4220-4226) */
	uint32_t prec_search;           /* -p: every coefficient precision 5..max (stream_encoder.c:4230-4243) */
} fo_config;

typedef struct {
	uint32_t type;        /* 0 CONSTANT, 1 VERBATIM, 2 FIXED, 3 LPC */
	uint32_t order;
	uint32_t wasted_bits;
	uint32_t bits;        /* estimate used for selection */
	uint32_t partition_order;
	uint32_t rice2;
	uint32_t precision;
	int32_t  shift;
	int32_t  qlp[FO_MAX_LPC_ORDER];
} fo_subframe_info;

typedef struct {
	uint32_t channel_assignment; /* 0 independent, 1 left/side, 2 right/side, 3 mid/side */
	uint32_t frame_bytes;
	fo_subframe_info sub[FO_MAX_CHANNELS];
} fo_frame_info;

/* ---- whole frame ----------------------------------------------------------------- */
/* pcm[ch] points at blocksize samples. Returns bytes written, <0 on error
 * (-1 unsupported configuration, -2 output too small). */
int64_t fo_encode_frame(const fo_config *cfg, const int32_t *const pcm[], uint64_t frame_number,
                        uint8_t *out, size_t cap, fo_frame_info *info);

/* Convenience for tests: encode nframes full blocks + optional short tail of planar-per-stream
 * PCM (pcm[ch][total]) with ONE config; windows for the short tail are taken from tail_cfg. */
int64_t fo_encode_frames(const fo_config *cfg, const fo_config *tail_cfg, const int32_t *const pcm[],
                         uint64_t total_samples, uint64_t first_frame_number,
                         uint8_t *out, size_t cap, uint32_t *frame_bytes, uint32_t *nframes);

/* ---- stages (exposed for known-answer / differential tests) ------------------------ */
void     fo_window_tukey(float *w, int32_t L, float p);                         /* window.c:199 */
void     fo_window_data(const int32_t *in, const float *w, float *out, uint32_t n); /* lpc.c:68 */
void     fo_window_data_partial(const int32_t *in, const float *w, float *out, uint32_t n,
                                uint32_t part_size, uint32_t data_shift);       /* lpc.c:82 */
void     fo_autocorrelation(uint32_t variant, const float *d, uint32_t n, uint32_t lag, double *autoc);
void     fo_lp_coefficients(const double *autoc, uint32_t *max_order, float lp[][FO_MAX_LPC_ORDER], double *err); /* lpc.c:176 */
uint32_t fo_best_order(const double *err, uint32_t max_order, uint32_t total_samples, uint32_t overhead); /* lpc.c:1608 */
double   fo_expected_bits_per_residual_sample(double lpc_error, uint32_t total_samples); /* lpc.c:1580 */
int      fo_quantize_coefficients(const float *lp, uint32_t order, uint32_t precision, int32_t *q, int *shift); /* lpc.c:220 */
uint32_t fo_fixed_best_predictor(const int32_t *data, uint32_t n, float rbps[5]); /* fixed.c:222; data[-4..n) */
uint32_t fo_fixed_best_predictor_ex(const int32_t *data, uint32_t n, float rbps[5], int wide); /* wide: fixed_intrin_avx2.c:57 */
uint32_t fo_rice_search(const int32_t *residual, uint32_t residual_samples, uint32_t predictor_order,
                        uint32_t rice_limit, uint32_t min_po, uint32_t max_po, uint32_t bps,
                        uint32_t *best_po, uint32_t *params /*[1<<max_po]*/);     /* stream_encoder.c:4701 */
uint8_t  fo_crc8(const uint8_t *p, size_t n);    /* crc.c:366 */
uint16_t fo_crc16(const uint8_t *p, size_t n);   /* crc.c:376 */
double   fo_log(double x);                       /* libm log as the reference binary calls it */

#ifdef __cplusplus
}
#endif
#endif
