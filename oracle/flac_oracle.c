/* oracle/flac_oracle.c -- TEST INFRASTRUCTURE ONLY; see flac_oracle.h.
 *
 * A from-scratch restatement of the reference's per-block encode path. Every function
 * cites the reference code (path:line under /root/reference) whose behaviour it states.
 * Where the reference's result is defined by its compiled binary rather than its source
 * (gcc -O3 -fassociative-math ... ) the compiled association order is
 * written out explicitly; this file must be compiled with -ffp-contract=off and without
 * any fast-math flag (oracle/Makefile does that).
 */
#include "flac_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------
 * small integer helpers            (include/private/bitmath.h:156,172; bitmath.c:63)
 * ---------------------------------------------------------------------------------- */
static uint32_t ilog2_u32(uint32_t v) { return 31u - (uint32_t)__builtin_clz(v); }
static uint32_t ilog2_u64(uint64_t v) { return 63u - (uint32_t)__builtin_clzll(v); }
static uint32_t silog2_i64(int64_t v)
{
	if(v == 0) return 0;
	if(v == -1) return 2;
	if(v < 0) v = -(v + 1);
	return ilog2_u64((uint64_t)v) + 2;
}
static uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }

double fo_log(double x) { return log(x); }

/* ------------------------------------------------------------------------------------
 * CRC-8 (poly x^8+x^2+x+1) and CRC-16 (poly x^16+x^15+x^2+1), both MSB first, init 0
 * (crc.c:366,376 -- table-driven there, bitwise here; same function of the bytes)
 * ---------------------------------------------------------------------------------- */
uint8_t fo_crc8(const uint8_t *p, size_t n)
{
	uint32_t c = 0;
	while(n--) {
		c ^= *p++;
		for(int b = 0; b < 8; b++) c = (c & 0x80) ? ((c << 1) ^ 0x07) & 0xff : (c << 1) & 0xff;
	}
	return (uint8_t)c;
}
uint16_t fo_crc16(const uint8_t *p, size_t n)
{
	uint32_t c = 0;
	while(n--) {
		c ^= (uint32_t)(*p++) << 8;
		for(int b = 0; b < 8; b++) c = (c & 0x8000) ? ((c << 1) ^ 0x8005) & 0xffff : (c << 1) & 0xffff;
	}
	return (uint16_t)c;
}

/* ------------------------------------------------------------------------------------
 * MSB-first bit writer              (bitwriter.c:316-437; big-endian word packing there)
 * ---------------------------------------------------------------------------------- */
typedef struct { uint8_t *buf; size_t cap; uint64_t nbits; int overflow; } bitw;

static void bw_bits(bitw *w, uint64_t v, uint32_t n) /* low n bits of v, n <= 64 */
{
	while(n) {
		size_t byte = (size_t)(w->nbits >> 3);
		uint32_t used = (uint32_t)(w->nbits & 7), room = 8 - used, take = n < room ? n : room;
		if(byte >= w->cap) { w->overflow = 1; return; }
		if(used == 0) w->buf[byte] = 0;
		uint32_t chunk = (uint32_t)((v >> (n - take)) & ((1u << take) - 1u));
		w->buf[byte] |= (uint8_t)(chunk << (room - take));
		w->nbits += take;
		n -= take;
	}
}
static void bw_signed(bitw *w, int64_t v, uint32_t n) { bw_bits(w, (uint64_t)v, n); } /* two's complement, low n bits (bitwriter.c:366,387) */
static void bw_unary(bitw *w, uint32_t zeros) /* `zeros` 0-bits then a 1 (bitwriter.c:429) */
{
	while(zeros >= 32) { bw_bits(w, 0, 32); zeros -= 32; }
	bw_bits(w, 1, zeros + 1);
}
static void bw_utf8_u32(bitw *w, uint32_t v) /* bitwriter.c:832 */
{
	if(v < 0x80) bw_bits(w, v, 8);
	else if(v < 0x800) { bw_bits(w, 0xC0 | (v >> 6), 8); bw_bits(w, 0x80 | (v & 0x3F), 8); }
	else if(v < 0x10000) { bw_bits(w, 0xE0 | (v >> 12), 8); bw_bits(w, 0x80 | ((v >> 6) & 0x3F), 8); bw_bits(w, 0x80 | (v & 0x3F), 8); }
	else if(v < 0x200000) { bw_bits(w, 0xF0 | (v >> 18), 8); bw_bits(w, 0x80 | ((v >> 12) & 0x3F), 8); bw_bits(w, 0x80 | ((v >> 6) & 0x3F), 8); bw_bits(w, 0x80 | (v & 0x3F), 8); }
	else if(v < 0x4000000) { bw_bits(w, 0xF8 | (v >> 24), 8); bw_bits(w, 0x80 | ((v >> 18) & 0x3F), 8); bw_bits(w, 0x80 | ((v >> 12) & 0x3F), 8); bw_bits(w, 0x80 | ((v >> 6) & 0x3F), 8); bw_bits(w, 0x80 | (v & 0x3F), 8); }
	else { bw_bits(w, 0xFC | (v >> 30), 8); bw_bits(w, 0x80 | ((v >> 24) & 0x3F), 8); bw_bits(w, 0x80 | ((v >> 18) & 0x3F), 8); bw_bits(w, 0x80 | ((v >> 12) & 0x3F), 8); bw_bits(w, 0x80 | ((v >> 6) & 0x3F), 8); bw_bits(w, 0x80 | (v & 0x3F), 8); }
}
/* Rice code of one residual (bitwriter.c:575-706): zig-zag fold, quotient in unary, k LSBs */
static void bw_rice(bitw *w, int32_t r, uint32_t k)
{
	uint32_t u = ((uint32_t)r << 1) ^ (uint32_t)(r >> 31);
	bw_unary(w, u >> k);
	if(k) bw_bits(w, u & ((1u << k) - 1u), k);
}

/* ------------------------------------------------------------------------------------
 * windows                                           (window.c:199-222, host-side tables)
 * ---------------------------------------------------------------------------------- */
void fo_window_tukey(float *w, int32_t L, float p)
{
	int32_t n;
	if(!(p > 0.0f && p < 1.0f)) {
		if(p <= 0.0f) { for(n = 0; n < L; n++) w[n] = 1.0f; return; }
		if(p >= 1.0f) { /* hann, window.c:144 */
			const int32_t N = L - 1;
			for(n = 0; n < L; n++) w[n] = (float)(0.5f - 0.5f * cosf(2.0f * M_PI * n / N));
			return;
		}
		fo_window_tukey(w, L, 0.5f); /* NaN */
		return;
	}
	{
		const int32_t Np = (int32_t)(p / 2.0f * L) - 1;
		for(n = 0; n < L; n++) w[n] = 1.0f;
		if(Np > 0) {
			for(n = 0; n <= Np; n++) {
				w[n] = (float)(0.5f - 0.5f * cosf(M_PI * n / Np));
				w[L - Np - 1 + n] = (float)(0.5f - 0.5f * cosf(M_PI * (n + Np) / Np));
			}
		}
	}
}

/* out[i] = (float)in[i] * w[i]  -- one int->float rounding, one float multiply (lpc.c:68) */
void fo_window_data(const int32_t *in, const float *w, float *out, uint32_t n)
{
	for(uint32_t i = 0; i < n; i++) out[i] = (float)in[i] * w[i];
}
/* lpc.c:82-94: first part_size taps ramp up with w[0..part), the next part_size ramp down
 * with w[n-part..n), one trailing zero. Writes (at most) 2*part_size+1 outputs. */
void fo_window_data_partial(const int32_t *in, const float *w, float *out, uint32_t n, uint32_t part_size, uint32_t data_shift)
{
	uint32_t i, j;
	if(part_size + data_shift < n) {
		for(i = 0; i < part_size; i++) out[i] = (float)in[data_shift + i] * w[i];
		i = umin(i, n - part_size - data_shift);
		for(j = n - part_size; j < n; i++, j++) out[i] = (float)in[data_shift + i] * w[j];
		if(i < n) out[i] = 0.0f;
	}
}

/* the same on the 64-bit signal arrays this restatement keeps internally: lpc.c:75,96 for the 33-bit side channel;
 * for every narrower channel (float)(int64)v == (float)(int32)v, one rounding either way */
static void window_data64(const int64_t *in, const float *w, float *out, uint32_t n)
{
	for(uint32_t i = 0; i < n; i++) out[i] = (float)in[i] * w[i];
}
static void window_data_partial64(const int64_t *in, const float *w, float *out, uint32_t n, uint32_t part_size, uint32_t data_shift)
{
	uint32_t i, j;
	if(part_size + data_shift < n) {
		for(i = 0; i < part_size; i++) out[i] = (float)in[data_shift + i] * w[i];
		i = umin(i, n - part_size - data_shift);
		for(j = n - part_size; j < n; i++, j++) out[i] = (float)in[data_shift + i] * w[j];
		if(i < n) out[i] = 0.0f;
	}
}

/* ------------------------------------------------------------------------------------
 * autocorrelation, in the association order of the reference's compiled routines
 * ---------------------------------------------------------------------------------- */
#define D(k) ((double)d[k])

/* lpc.c:133-157: the "data locality" loop, used when data_len < 32 or lag > 16.
 * Plain sequential accumulation per lag, in increasing sample order. */
static void autoc_generic_small(const float *d, uint32_t n, uint32_t lag, double *a)
{
	uint32_t s, c;
	const uint32_t limit = n - lag;
	for(c = 0; c < lag; c++) a[c] = 0.0;
	for(s = 0; s <= limit; s++) { double x = D(s); for(c = 0; c < lag; c++) a[c] += x * D(s + c); }
	for(; s < n; s++) { double x = D(s); for(c = 0; c < n - s; c++) a[c] += x * D(s + c); }
}

/* Tail shared by the three FMA routines, from sample i to n-1 for lag j:
 * one 4-sample step if >= 4 samples remain, then <= 3 sequential fma steps. */
static double autoc_fma_tail(const float *d, uint32_t i, uint32_t n, uint32_t j, double a)
{
	if(n - i >= 4) {
		double hi = fma(D(i + 1), D(i + 1 - j), D(i + 3) * D(i + 3 - j));
		double lo = fma(D(i), D(i - j), D(i + 2) * D(i + 2 - j));
		a = (hi + lo) + a;
		i += 4;
	}
	for(; i < n; i++) a = fma(D(i), D(i - j), a);
	return a;
}

/* lpc_intrin_fma.c:46,61 (MAX_LAG 8 and 16) <- deduplication/lpc_compute_autocorrelation_intrin.c,
 * as gcc 11.4 -O3 -fassociative-math target("fma") vectorised it: 4 double lanes, two
 * vectors per iteration (8 samples), lanes combined (3+1)+(2+0). SURVEY.md 5.9. */
static void autoc_fma_lag8_16(const float *d, uint32_t n, uint32_t L, double *autoc)
{
	for(uint32_t j = 0; j < L; j++) {
		double a = 0.0, acc[4] = {0.0, 0.0, 0.0, 0.0};
		uint32_t i;
		for(i = j; i < L; i++) a += D(i) * D(i - j);
		const uint32_t nb = (n - L) / 8;
		i = L;
		for(uint32_t k = 0; k < nb; k++, i += 8)
			for(uint32_t l = 0; l < 4; l++)
				acc[l] += fma(D(i + l), D(i + l - j), D(i + 4 + l) * D(i + 4 + l - j));
		if(nb) a = ((acc[3] + acc[1]) + (acc[2] + acc[0])) + a;
		autoc[j] = autoc_fma_tail(d, i, n, j, a);
	}
}

/* lpc_intrin_fma.c:54 (MAX_LAG 12): as above but gcc additionally unrolled the 8-sample
 * body x2 (predictive commoning) and, for lag 8 only, factored x*y0 + x*y2 -> x*(y0+y2)
 * across the two halves (-fassociative-math). SURVEY.md 5.9. */
static void autoc_fma_lag12(const float *d, uint32_t n, double *autoc)
{
	const uint32_t L = 12;
	for(uint32_t j = 0; j < L; j++) {
		double a = 0.0, acc[4] = {0.0, 0.0, 0.0, 0.0};
		uint32_t i;
		for(i = j; i < L; i++) a += D(i) * D(i - j);
		const uint32_t nb = (n - L) / 8;
		const uint32_t npairs = nb > 2 ? ((nb - 3) & ~1u) / 2 + 1 : 0;
		uint32_t k = 0;
		i = L;
		for(uint32_t p = 0; p < npairs; p++, k += 2, i += 16) {
			for(uint32_t l = 0; l < 4; l++) {
				if(j == 8) {
					double x0 = D(i + l), x1 = D(i + 4 + l), x2 = D(i + 8 + l), x3 = D(i + 12 + l);
					double y0 = D(i + l - 8), y1 = D(i + 4 + l - 8);
					acc[l] += fma(x0, (y0 + x2), x1 * (y1 + x3));
				}
				else {
					double t0 = fma(D(i + l), D(i + l - j), D(i + 4 + l) * D(i + 4 + l - j));
					double t1 = fma(D(i + 8 + l), D(i + 8 + l - j), D(i + 12 + l) * D(i + 12 + l - j));
					acc[l] += (t1 + t0);
				}
			}
		}
		for(; k < nb; k++, i += 8)
			for(uint32_t l = 0; l < 4; l++)
				acc[l] += fma(D(i + l), D(i + l - j), D(i + 4 + l) * D(i + 4 + l - j));
		if(nb) a = ((acc[3] + acc[1]) + (acc[2] + acc[0])) + a;
		autoc[j] = autoc_fma_tail(d, i, n, j, a);
	}
}
#undef D

void fo_autocorrelation(uint32_t variant, const float *d, uint32_t n, uint32_t lag, double *autoc)
{
	switch(variant) {
		case FO_AUTOC_FMA_LAG8:  autoc_fma_lag8_16(d, n, 8, autoc); break;
		case FO_AUTOC_FMA_LAG12: autoc_fma_lag12(d, n, autoc); break;
		case FO_AUTOC_FMA_LAG16: autoc_fma_lag8_16(d, n, 16, autoc); break;
		default: autoc_generic_small(d, n, lag, autoc); break;
	}
}

/* ------------------------------------------------------------------------------------
 * Levinson-Durbin (lpc.c:176-218) as compiled: source order, no contraction, true divide,
 * except the odd-order middle term which -fassociative-math factored to (r+1.0)*lpc[j].
 * ---------------------------------------------------------------------------------- */
void fo_lp_coefficients(const double *autoc, uint32_t *max_order, float lp[][FO_MAX_LPC_ORDER], double *error)
{
	double lpc[FO_MAX_LPC_ORDER], err = autoc[0];
	for(uint32_t i = 0; i < *max_order; i++) {
		double r = -autoc[i + 1];
		uint32_t j;
		for(j = 0; j < i; j++) r -= lpc[j] * autoc[i - j];
		r /= err;
		lpc[i] = r;
		for(j = 0; j < (i >> 1); j++) {
			double tmp = lpc[j];
			lpc[j] += r * lpc[i - 1 - j];
			lpc[i - 1 - j] += r * tmp;
		}
		if(i & 1) lpc[j] = (r + 1.0) * lpc[j];
		err *= (1.0 - r * r);
		for(j = 0; j <= i; j++) lp[i][j] = (float)(-lpc[j]);
		error[i] = err;
		if(err == 0.0) { *max_order = i + 1; return; }
	}
}

/* lpc.c:1580-1606. 0.5*log(x)/M_LN2 was folded by -freciprocal-math into one multiply by
 * 0.5/ln2 = 0.72134752044448169 */
static double expected_bits_scaled(double lpc_error, double error_scale)
{
	if(lpc_error > 0.0) {
		double bps = log(error_scale * lpc_error) * 0.7213475204444817;
		return bps >= 0.0 ? bps : 0.0;
	}
	if(lpc_error < 0.0) return 1e32;
	return 0.0;
}
double fo_expected_bits_per_residual_sample(double lpc_error, uint32_t total_samples)
{
	return expected_bits_scaled(lpc_error, 0.5 / (double)total_samples);
}
/* lpc.c:1608-1630: argmin over orders, first minimum wins */
uint32_t fo_best_order(const double *err, uint32_t max_order, uint32_t total_samples, uint32_t overhead)
{
	const double scale = 0.5 / (double)total_samples;
	double best_bits = (double)4294967295u;
	uint32_t best = 0;
	for(uint32_t idx = 0, order = 1; idx < max_order; idx++, order++) {
		double bits = expected_bits_scaled(err[idx], scale) * (double)(total_samples - order) + (double)(order * overhead);
		if(bits < best_bits) { best = idx; best_bits = bits; }
	}
	return best + 1;
}

/* lpc.c:220-314 */
int fo_quantize_coefficients(const float *lp, uint32_t order, uint32_t precision, int32_t *q, int *shift)
{
	double cmax = 0.0;
	int32_t qmax, qmin;
	uint32_t i;
	precision--;
	qmax = (int32_t)1 << precision; qmin = -qmax; qmax--;
	for(i = 0; i < order; i++) { double a = fabs((double)lp[i]); if(a > cmax) cmax = a; }
	if(cmax <= 0.0) return 2;
	{
		int e;
		(void)frexp(cmax, &e);
		e--;
		*shift = (int)precision - e - 1;
		if(*shift > 15) *shift = 15;
		else if(*shift < -16) return 1;
	}
	if(*shift >= 0) {
		double error = 0.0;
		for(i = 0; i < order; i++) {
			error += (double)(lp[i] * (float)(1 << *shift));
			int32_t v = (int32_t)lround(error);
			if(v > qmax) v = qmax; else if(v < qmin) v = qmin;
			error -= v;
			q[i] = v;
		}
	}
	else {
		const int nshift = -*shift;
		double error = 0.0;
		for(i = 0; i < order; i++) {
			error += (double)(lp[i] / (float)(1 << nshift));
			int32_t v = (int32_t)lround(error);
			if(v > qmax) v = qmax; else if(v < qmin) v = qmin;
			error -= v;
			q[i] = v;
		}
		*shift = 0;
	}
	return 0;
}

/* lpc.c:942-976 */
static uint64_t max_prediction_value_before_shift(uint32_t bps, const int32_t *q, uint32_t order)
{
	uint32_t s = 0;
	for(uint32_t i = 0; i < order; i++) s += (uint32_t)abs(q[i]);
	return ((uint64_t)1 << (bps - 1)) * s;
}
static uint32_t max_prediction_before_shift_bps(uint32_t bps, const int32_t *q, uint32_t order)
{
	return silog2_i64((int64_t)max_prediction_value_before_shift(bps, q, order));
}
static uint32_t max_residual_bps(uint32_t bps, const int32_t *q, uint32_t order, int shift)
{
	uint64_t maxabs = (uint64_t)1 << (bps - 1);
	uint64_t after = (uint64_t)(-1 * ((-1 * (int64_t)max_prediction_value_before_shift(bps, q, order)) >> shift));
	return silog2_i64((int64_t)(maxabs + after));
}

/* ------------------------------------------------------------------------------------
 * fixed predictors                                                (fixed.c:222-374,470-561)
 * ---------------------------------------------------------------------------------- */
static float fixed_rbps(uint64_t err, uint32_t n)
{
	/* fixed.c:284-288 as compiled: log(M_LN2*err/n)/M_LN2 with the division by M_LN2 folded
	 * to a multiply by 1/ln2 (-freciprocal-math) */
	if(err == 0) return 0.0f;
	return (float)(log(((double)err * M_LN2) / (double)n) * 1.4426950408889634);
}
/* data points at sample 4 of the block (data[-4..-1] are valid); n = blocksize-4.
 *
 * narrow (wide=0): the reference dispatches to ..._intrin_ssse3 (fixed_intrin_ssse3.c:62,
 *   selected at stream_encoder.c:1085 when bps+ilog2(17n) < 32): a 4-way split of the
 *   block plus a scalar remainder -- exactly the plain sums of fixed.c:222 for every n.
 * wide (wide=1): ..._wide_intrin_avx2 (fixed_intrin_avx2.c:57, selected at
 *   stream_encoder.c:1095).  Its four 64-bit lanes each process q = n/4 samples; lane l
 *   takes its 4-sample HISTORY from offset l*q but its DATA from offset (l*n)/4, and the
 *   n%4 trailing samples are ignored ("Ignore the remainder").  For n%4 == 0 this is the
 *   plain sum; otherwise lanes 2 and 3 difference across a small gap.  Restated literally:
 *   lane l sees the virtual sequence  data[l*q-4 .. l*q-1] ++ data[(l*n)/4 .. (l*n)/4+q). */
uint32_t fo_fixed_best_predictor_ex(const int32_t *data, uint32_t n, float rbps[5], int wide)
{
	uint64_t e[5] = {0, 0, 0, 0, 0};
	uint32_t order;
	if(!wide) {
		for(int32_t i = 0; i < (int32_t)n; i++) {
			uint32_t a = (uint32_t)data[i], b = (uint32_t)data[i - 1], c = (uint32_t)data[i - 2], d = (uint32_t)data[i - 3], f = (uint32_t)data[i - 4];
			int32_t dd[5] = { (int32_t)a, (int32_t)(a - b), (int32_t)(a - 2u * b + c), (int32_t)(a - 3u * b + 3u * c - d), (int32_t)(a - 4u * b + 6u * c - 4u * d + f) };
			for(int k = 0; k < 5; k++) e[k] += (uint32_t)(dd[k] < 0 ? -(uint32_t)dd[k] : (uint32_t)dd[k]);
		}
		for(int k = 0; k < 5; k++) e[k] = (uint32_t)e[k]; /* 32-bit totals there; cannot overflow by the selector's bound */
	}
	else {
		const int32_t q = (int32_t)(n / 4);
		for(int32_t l = 0; l < 4; l++) {
			const int32_t hist = l * q, start = (int32_t)(((uint64_t)l * n) / 4);
			for(int32_t i = 0; i < q; i++) {
				int64_t v[5];
				for(int32_t j = 0; j < 5; j++) { int32_t m = i - j; v[j] = data[m >= 0 ? start + m : hist + m]; }
				int64_t dd[5] = { v[0], v[0] - v[1], v[0] - 2 * v[1] + v[2], v[0] - 3 * v[1] + 3 * v[2] - v[3], v[0] - 4 * v[1] + 6 * v[2] - 4 * v[3] + v[4] };
				for(int k = 0; k < 5; k++) e[k] += (uint64_t)(dd[k] < 0 ? -dd[k] : dd[k]);
			}
		}
	}
#define MIN2(x, y) ((x) < (y) ? (x) : (y))
	if(e[0] <= MIN2(MIN2(MIN2(e[1], e[2]), e[3]), e[4])) order = 0;
	else if(e[1] <= MIN2(MIN2(e[2], e[3]), e[4])) order = 1;
	else if(e[2] <= MIN2(e[3], e[4])) order = 2;
	else if(e[3] <= e[4]) order = 3;
	else order = 4;
#undef MIN2
	for(int k = 0; k < 5; k++) rbps[k] = fixed_rbps(e[k], n);
	return order;
}
uint32_t fo_fixed_best_predictor(const int32_t *data, uint32_t n, float rbps[5])
{
	return fo_fixed_best_predictor_ex(data, n, rbps, 0);
}

static uint64_t abs64(int64_t v) { return (uint64_t)(v < 0 ? -v : v); }
#define RBPS_OF(err, n) ((float)(((err) > 0) ? log(M_LN2 * (double)(err) / (double)(n)) / M_LN2 : 0.0))
/* subframe_bps 28..32: FLAC__fixed_compute_best_predictor_limit_residual_intrin_avx2 (fixed_intrin_avx2.c:187, selected at
 * stream_encoder.c:1027/1088).  Unlike the estimators above the sums start at sample 0 of the block (orders that can be
 * formed there), four 64-bit lanes take n/4 samples each -- history from offset l*(n/4), data from offset (l*n)/4, the
 * same quirk as the _wide flavour -- and the n%4 trailing samples are added afterwards.  An order whose residual
 * leaves the int32 range anywhere is marked with 34 bits per sample and cannot be picked. */
static uint32_t fixed_best_predictor_limit_avx2(const int32_t *data, uint32_t n, float rbps[5])
{
	uint64_t tot[5] = {0, 0, 0, 0, 0}, sh[5] = {0, 0, 0, 0, 0}, smallest = UINT64_MAX;
	const int32_t q = (int32_t)(n / 4);
	uint32_t order = 0;
	for(int32_t i = -4; i < 0; i++) {
		uint64_t e[4];
		e[0] = abs64((int64_t)data[i]);
		e[1] = i > -4 ? abs64((int64_t)data[i] - data[i - 1]) : 0;
		e[2] = i > -3 ? abs64((int64_t)data[i] - 2 * (int64_t)data[i - 1] + data[i - 2]) : 0;
		e[3] = i > -2 ? abs64((int64_t)data[i] - 3 * (int64_t)data[i - 1] + 3 * (int64_t)data[i - 2] - data[i - 3]) : 0;
		for(int k = 0; k < 4; k++) { tot[k] += e[k]; sh[k] |= e[k]; }
	}
	for(int32_t l = 0; l < 4; l++) {
		const int32_t h = l * q, start = (int32_t)(((uint64_t)l * n) / 4);
		int64_t p0 = data[-1 + h], p1 = (int64_t)data[-1 + h] - data[-2 + h], p2 = p1 - ((int64_t)data[-2 + h] - data[-3 + h]),
		        p3 = p2 - ((int64_t)data[-2 + h] - 2 * (int64_t)data[-3 + h] + data[-4 + h]);
		for(int32_t i = 0; i < q; i++) {
			const int64_t d0 = data[i + start], d1 = d0 - p0, d2 = d1 - p1, d3 = d2 - p2, d4 = d3 - p3;
			const uint64_t e[5] = { abs64(d0), abs64(d1), abs64(d2), abs64(d3), abs64(d4) };
			for(int k = 0; k < 5; k++) { tot[k] += e[k]; sh[k] |= e[k]; }
			p0 = d0; p1 = d1; p2 = d2; p3 = d3;
		}
	}
	for(int32_t i = 4 * q; i < (int32_t)n; i++) {
		const uint64_t e[5] = {
			abs64((int64_t)data[i]), abs64((int64_t)data[i] - data[i - 1]), abs64((int64_t)data[i] - 2 * (int64_t)data[i - 1] + data[i - 2]),
			abs64((int64_t)data[i] - 3 * (int64_t)data[i - 1] + 3 * (int64_t)data[i - 2] - data[i - 3]),
			abs64((int64_t)data[i] - 4 * (int64_t)data[i - 1] + 6 * (int64_t)data[i - 2] - 4 * (int64_t)data[i - 3] + data[i - 4]) };
		for(int k = 0; k < 5; k++) { tot[k] += e[k]; sh[k] |= e[k]; }
	}
	for(uint32_t k = 0; k < 5; k++) {                      /* CHECK_ORDER_IS_VALID, fixed_intrin_avx2.c:172 */
		if(sh[k] <= INT32_MAX) {
			if(tot[k] < smallest) { order = k; smallest = tot[k]; }
			rbps[k] = RBPS_OF(tot[k], n);
		}
		else rbps[k] = 34.0f;
	}
	return order;
}
/* subframe_bps 33: FLAC__fixed_compute_best_predictor_limit_residual_33bit (fixed.c:424, plain C only).  Its
 * CHECK_ORDER_IS_VALID (fixed.c:360) gives 34 bits per sample to every order that is not a new minimum as well. */
static uint32_t fixed_best_predictor_limit_33bit(const int64_t *data, uint32_t n, float rbps[5])
{
	uint64_t tot[5] = {0, 0, 0, 0, 0}, smallest = UINT64_MAX;
	int valid[5] = {1, 1, 1, 1, 1};
	uint32_t order = 0;
	for(int32_t i = -4; i < (int32_t)n; i++) {
		uint64_t e[5];
		e[0] = abs64(data[i]);
		e[1] = i > -4 ? abs64(data[i] - data[i - 1]) : 0;
		e[2] = i > -3 ? abs64(data[i] - 2 * data[i - 1] + data[i - 2]) : 0;
		e[3] = i > -2 ? abs64(data[i] - 3 * data[i - 1] + 3 * data[i - 2] - data[i - 3]) : 0;
		e[4] = i > -1 ? abs64(data[i] - 4 * data[i - 1] + 6 * data[i - 2] - 4 * data[i - 3] + data[i - 4]) : 0;
		for(int k = 0; k < 5; k++) { tot[k] += e[k]; if(e[k] > INT32_MAX) valid[k] = 0; }
	}
	for(uint32_t k = 0; k < 5; k++) {
		if(valid[k] && tot[k] < smallest) { order = k; smallest = tot[k]; rbps[k] = RBPS_OF(tot[k], n); }
		else rbps[k] = 34.0f;
	}
	return order;
}

/* fixed.c:470-499 (32-bit wrapping, subframe_bps + order <= 32), fixed.c:501 / :532 (64-bit differences truncated to 32
 * bits: stream_encoder.c:4511-4516).  x -> sample `order` */
static void fixed_residual(const int64_t *x, uint32_t n, uint32_t order, int wide, int32_t *r)
{
	if(wide) {
		for(int32_t i = 0; i < (int32_t)n; i++) {
			switch(order) {
				case 0: r[i] = (int32_t)x[i]; break;
				case 1: r[i] = (int32_t)(x[i] - x[i - 1]); break;
				case 2: r[i] = (int32_t)(x[i] - 2 * x[i - 1] + x[i - 2]); break;
				case 3: r[i] = (int32_t)(x[i] - 3 * x[i - 1] + 3 * x[i - 2] - x[i - 3]); break;
				default: r[i] = (int32_t)(x[i] - 4 * x[i - 1] + 6 * x[i - 2] - 4 * x[i - 3] + x[i - 4]); break;
			}
		}
		return;
	}
	for(int32_t i = 0; i < (int32_t)n; i++) {
		uint32_t a = (uint32_t)x[i];
		switch(order) {
			case 0: r[i] = (int32_t)a; break;
			case 1: r[i] = (int32_t)(a - (uint32_t)x[i - 1]); break;
			case 2: r[i] = (int32_t)(a - 2u * (uint32_t)x[i - 1] + (uint32_t)x[i - 2]); break;
			case 3: r[i] = (int32_t)(a - 3u * (uint32_t)x[i - 1] + 3u * (uint32_t)x[i - 2] - (uint32_t)x[i - 3]); break;
			default: r[i] = (int32_t)(a - 4u * (uint32_t)x[i - 1] + 6u * (uint32_t)x[i - 2] - 4u * (uint32_t)x[i - 3] + (uint32_t)x[i - 4]); break;
		}
	}
}

/* lpc.c:321 (32-bit wrapping accumulate), lpc.c:582 (64-bit accumulate), and -- wide == 2 -- lpc.c:832 / :886, the
 * overflow-checked flavours taken when the residual may not fit 32 bits (stream_encoder.c:4601-4609): they return 0,
 * and the candidate is dropped, as soon as one residual leaves (INT32_MIN, INT32_MAX] */
static int lpc_residual(const int64_t *x, uint32_t n, const int32_t *q, uint32_t order, int shift, int wide, int32_t *r)
{
	for(int32_t i = 0; i < (int32_t)n; i++) {
		if(wide == 2) {
			int64_t s = 0, v;
			for(uint32_t j = 0; j < order; j++) s += (int64_t)q[j] * x[i - 1 - (int32_t)j];
			v = x[i] - (s >> shift);
			if(v <= INT32_MIN || v > INT32_MAX) return 0;
			r[i] = (int32_t)v;
		}
		else if(wide) {
			int64_t s = 0;
			for(uint32_t j = 0; j < order; j++) s += (int64_t)q[j] * (int64_t)x[i - 1 - (int32_t)j];
			r[i] = (int32_t)((int64_t)x[i] - (s >> shift));
		}
		else {
			uint32_t s = 0;
			for(uint32_t j = 0; j < order; j++) s += (uint32_t)q[j] * (uint32_t)x[i - 1 - (int32_t)j];
			r[i] = (int32_t)((uint32_t)x[i] - (uint32_t)((int32_t)s >> shift));
		}
	}
	return 1;
}
static int lpc_residual_mode(uint32_t bps, const int32_t *q, uint32_t order, int shift)   /* stream_encoder.c:4601-4617 */
{
	if(max_residual_bps(bps, q, order, shift) > 32) return 2;
	return max_prediction_before_shift_bps(bps, q, order) > 32 ? 1 : 0;
}

/* ------------------------------------------------------------------------------------
 * Rice partition search     (stream_encoder.c:4701-4795,4797-4852,4929-4951,4954-5075)
 * returns estimated residual bits (incl. the 2+4 method/order bits); fills best_po/params
 * ---------------------------------------------------------------------------------- */
static uint32_t limited_max_po(uint32_t limit, uint32_t blocksize, uint32_t order) /* format.c:550 */
{
	uint32_t po = limit;
	while(po > 0 && (blocksize >> po) <= order) po--;
	return po;
}
static uint32_t max_po_from_blocksize(uint32_t blocksize) /* format.c:540 */
{
	uint32_t po = 0;
	while(!(blocksize & 1)) { po++; blocksize >>= 1; }
	return umin(15, po);
}
static uint32_t rice_partition_bits(uint32_t k, uint32_t n, uint64_t sum) /* :4929-4951 */
{
	uint64_t b = 4 + (uint64_t)(1 + k) * n + (k ? (sum >> (k - 1)) : (sum << 1)) - (n >> 1);
	return (uint32_t)(b < 0xffffffffu ? b : 0xffffffffu);
}
uint32_t fo_rice_search(const int32_t *residual, uint32_t residual_samples, uint32_t predictor_order,
                        uint32_t rice_limit, uint32_t min_po, uint32_t max_po, uint32_t bps,
                        uint32_t *best_po_out, uint32_t *params_out)
{
	const uint32_t blocksize = residual_samples + predictor_order;
	uint32_t best_bits = 0, best_po = 0;
	max_po = limited_max_po(max_po, blocksize, predictor_order);
	min_po = umin(min_po, max_po);

	uint64_t *sums = (uint64_t *)malloc(sizeof(uint64_t) * (2u << max_po));
	uint32_t *cand = (uint32_t *)malloc(sizeof(uint32_t) * (1u << max_po));
	{ /* sums at max_po: 32-bit wrapping accumulator when the reference uses one (:4814-4834) */
		const uint32_t dps = blocksize >> max_po;
		const uint32_t threshold = 32 - ilog2_u32(dps);
		const int narrow = (bps + 4 < threshold);
		uint32_t rs = 0, end = (uint32_t)(-(int32_t)predictor_order);
		for(uint32_t p = 0; p < (1u << max_po); p++) {
			uint64_t s = 0;
			end += dps;
			for(; rs < end; rs++) { int32_t v = residual[rs]; s += (uint32_t)(v < 0 ? -(uint32_t)v : (uint32_t)v); }
			sums[p] = narrow ? (uint32_t)s : s;
		}
		uint32_t from = 0, to = 1u << max_po;
		for(int po = (int)max_po - 1; po >= (int)min_po; po--)
			for(uint32_t i = 0; i < (1u << po); i++, from += 2) sums[to++] = sums[from] + sums[from + 1];
	}
	{
		uint32_t off = 0;
		for(int po = (int)max_po; po >= (int)min_po; po--) {
			const uint32_t parts = 1u << po, base = blocksize >> po;
			uint32_t bits = 2 + 4;
			int ok = 1;
			for(uint32_t p = 0; p < parts; p++) {
				uint32_t n = base, k;
				if(p == 0) { if(n <= predictor_order) { ok = 0; break; } n -= predictor_order; }
				const uint32_t div = 0x40000 / n;
				const uint64_t mean = sums[off + p];
				if(mean < 2 || (((mean - 1) * div) >> 18) == 0) k = 0;
				else k = ilog2_u64(((mean - 1) * div) >> 18) + 1;
				if(k >= rice_limit) k = rice_limit - 1;
				const uint32_t pb = rice_partition_bits(k, n, mean);
				cand[p] = k;
				bits = (pb < 0xffffffffu - bits) ? bits + pb : 0xffffffffu;
			}
			if(!ok) break;
			off += parts;
			if(best_bits == 0 || bits < best_bits) {
				best_bits = bits; best_po = (uint32_t)po;
				memcpy(params_out, cand, sizeof(uint32_t) * parts);
			}
		}
	}
	free(sums); free(cand);
	*best_po_out = best_po;
	return best_bits;
}

/* ------------------------------------------------------------------------------------
 * one subframe: model search                           (stream_encoder.c:4045-4290)
 * ---------------------------------------------------------------------------------- */
typedef struct {
	fo_subframe_info s;
	uint32_t *params;   /* [1 << partition_order] */
	int64_t constant;
} subframe_t;

/* apply_apodization_ state machine (stream_encoder.c:4293-4392) */
typedef struct { uint32_t a, b, c; double autoc[FO_MAX_LPC_ORDER + 1], root[FO_MAX_LPC_ORDER + 1]; } apod_state;

static void next_subdivide(int32_t parts, uint32_t *a, uint32_t *depth, uint32_t *part)
{
	if(*depth == 2) {
		if(*part == 0) *part = 2;
		else { *part = 0; (*depth)++; }
	}
	else if(*part < 2 * (*depth) - 1) (*part)++;
	else { *part = 0; (*depth)++; }
	if(*depth > (uint32_t)parts) { (*a)++; *depth = 1; *part = 0; }
}

static int apply_apodization(const fo_config *cfg, apod_state *st, const int64_t *sig, float *windowed,
                             uint32_t *max_order, uint32_t subframe_bps, float lp[][FO_MAX_LPC_ORDER],
                             double *lpc_error, uint32_t *guess)
{
	const uint32_t N = cfg->blocksize;
	const fo_apodization *ap = &cfg->apodizations[st->a];
	const uint32_t variant = N <= FO_MAX_LPC_ORDER ? FO_AUTOC_GENERIC : cfg->autoc_variant;
	if(st->b == 1) {
		window_data64(sig, ap->window, windowed, N);
		fo_autocorrelation(variant, windowed, N, *max_order + 1, st->autoc);
		if(ap->kind == FO_APOD_SUBDIVIDE_TUKEY) {
			memcpy(st->root, st->autoc, *max_order * sizeof(double)); /* note: max_order, not +1 (:4340) */
			st->b++;
		}
		else st->a++;
	}
	else {
		if(N / st->b <= FO_MAX_LPC_ORDER) {
			next_subdivide((int32_t)ap->parts, &st->a, &st->b, &st->c);
			return 0;
		}
		if(!(st->c % 2)) {
			window_data_partial64(sig, ap->window, windowed, N, N / st->b / 2, (st->c / 2 * N) / st->b);
			fo_autocorrelation(variant, windowed, N / st->b, *max_order + 1, st->autoc);
		}
		else {
			for(uint32_t i = 0; i < *max_order; i++) st->autoc[i] = st->root[i] - st->autoc[i];
		}
		next_subdivide((int32_t)ap->parts, &st->a, &st->b, &st->c);
	}
	if(st->autoc[0] == 0.0) return 0;
	fo_lp_coefficients(st->autoc, max_order, lp, lpc_error);
	*guess = fo_best_order(lpc_error, *max_order, N, subframe_bps + (cfg->prec_search ? 5 : cfg->qlp_coeff_precision) /* :4384-4388 */);
	return 1;
}

static void process_subframe(const fo_config *cfg, const int64_t *sig /* already >> wasted */, uint32_t subframe_bps,
                             uint32_t wasted, int disable_constant, uint32_t min_po, uint32_t max_po,
                             subframe_t *best, subframe_t *cand, int32_t *residual, float *windowed)
{
	const uint32_t N = cfg->blocksize;
	const uint32_t rice_limit = cfg->bits_per_sample > 16 ? 31 : 15; /* :4076 */
	const uint32_t hdr = 8 + wasted;                                  /* 1+6+1 (+unary wasted) */
	uint32_t best_bits;

	/* VERBATIM baseline (:4669) */
	best->s.type = 1; best->s.order = 0; best->s.wasted_bits = wasted;
	best_bits = (cfg->disable_verbatim && N >= 4) ? 0xffffffffu : hdr + N * subframe_bps;

	if(N > 4) {
		float rbps[5];
		uint32_t guess_fixed;
		/* selector: stream_encoder.c:4098-4108 */
		if(subframe_bps <= 32) {
			int32_t *s32 = (int32_t *)malloc(sizeof(int32_t) * N);
			for(uint32_t i = 0; i < N; i++) s32[i] = (int32_t)sig[i];
			if(subframe_bps < 28) guess_fixed = fo_fixed_best_predictor_ex(s32 + 4, N - 4, rbps, !(subframe_bps + ilog2_u32((N - 4) * 17) < 32));
			else guess_fixed = fixed_best_predictor_limit_avx2(s32 + 4, N - 4, rbps);
			free(s32);
		}
		else guess_fixed = fixed_best_predictor_limit_33bit(sig + 4, N - 4, rbps);
		int constant = 0;
		if(!disable_constant && rbps[1] == 0.0f) {
			constant = 1;
			for(uint32_t i = 1; i < N; i++) if(sig[0] != sig[i]) { constant = 0; break; }
		}
		if(constant) {
			const uint32_t bits = hdr + subframe_bps;
			if(bits < best_bits) { best->s.type = 0; best->constant = sig[0]; best_bits = bits; }
		}
		else {
			if(!cfg->disable_fixed || (cfg->max_lpc_order == 0 && best_bits == 0xffffffffu)) {
				/* -e tries every fixed order, else the guessed one (stream_encoder.c:4155-4166) */
				uint32_t min_fixed = cfg->exhaustive ? 0 : guess_fixed, max_fixed = cfg->exhaustive ? 4 : guess_fixed;
				if(max_fixed >= N) max_fixed = N - 1;
				for(uint32_t order = min_fixed; order <= max_fixed; order++) {
					uint32_t po, rbits, est;
					if(rbps[order] >= (float)subframe_bps) continue;
					fixed_residual(sig + order, N - order, order, subframe_bps + order > 32, residual);
					rbits = fo_rice_search(residual, N - order, order, rice_limit, min_po, max_po, subframe_bps, &po, cand->params);
					est = hdr + order * subframe_bps;
					est = rbits < 0xffffffffu - est ? est + rbits : 0xffffffffu;
					if(est < best_bits) {
						uint32_t *t = best->params; best->params = cand->params; cand->params = t;
						best->s.type = 2; best->s.order = order; best->s.partition_order = po; best_bits = est;
					}
				}
			}
			if(cfg->max_lpc_order > 0) {
				uint32_t max_lpc = cfg->max_lpc_order >= N ? N - 1 : cfg->max_lpc_order;
				if(max_lpc > 0) {
					apod_state st;
					float lp[FO_MAX_LPC_ORDER][FO_MAX_LPC_ORDER];
					double lpc_error[FO_MAX_LPC_ORDER];
					memset(&st, 0, sizeof st);
					st.b = 1;
					while(st.a < cfg->num_apodizations) {
						uint32_t max_this = max_lpc, guess = 0;
						if(!apply_apodization(cfg, &st, sig, windowed, &max_this, subframe_bps, lp, lpc_error, &guess))
							continue;
						/* -e: every order 1..max_this (which the recursion may have shortened), else the guessed one (:4220-4226) */
						for(uint32_t order = cfg->exhaustive ? 1 : guess; order <= (cfg->exhaustive ? max_this : guess); order++) {
							uint32_t min_prec, max_prec;
							if(fo_expected_bits_per_residual_sample(lpc_error[order - 1], N - order) >= (double)subframe_bps)
								continue;
							if(cfg->prec_search) {           /* :4230-4240 */
								min_prec = 5;
								if(subframe_bps <= 17) {
									max_prec = umin(32 - subframe_bps - ilog2_u32(order), 15);
									if(max_prec < min_prec) max_prec = min_prec;
								}
								else max_prec = 15;
							}
							else min_prec = max_prec = cfg->qlp_coeff_precision;
							for(uint32_t prec = min_prec; prec <= max_prec; prec++) {
								uint32_t precision = prec, po, rbits, est;
								int32_t q[FO_MAX_LPC_ORDER];
								int shift;
								if(subframe_bps <= 17) precision = umin(precision, 32 - subframe_bps - ilog2_u32(order)); /* :4591 */
								memset(q, 0, sizeof q);
								if(fo_quantize_coefficients(lp[order - 1], order, precision, q, &shift) != 0)
									continue;
								if(!lpc_residual(sig + order, N - order, q, order, shift, lpc_residual_mode(subframe_bps, q, order, shift), residual))
									continue; /* a residual does not fit 32 bits (:4603-4608) */
								rbits = fo_rice_search(residual, N - order, order, rice_limit, min_po, max_po, subframe_bps, &po, cand->params);
								est = hdr + 4 + 5 + order * (precision + subframe_bps);
								est = rbits < 0xffffffffu - est ? est + rbits : 0xffffffffu;
								if(est > 0 && est < best_bits) {
									uint32_t *t = best->params; best->params = cand->params; cand->params = t;
									best->s.type = 3; best->s.order = order; best->s.partition_order = po;
									best->s.precision = precision; best->s.shift = shift;
									memcpy(best->s.qlp, q, sizeof q);
									best_bits = est;
								}
							}
						}
					}
				}
			}
		}
	}
	if(best_bits == 0xffffffffu) { best->s.type = 1; best_bits = hdr + N * subframe_bps; } /* :4281 */
	best->s.bits = best_bits;
	best->s.rice2 = 0;
	if(best->s.type >= 2)
		for(uint32_t p = 0; p < (1u << best->s.partition_order); p++)
			if(best->params[p] >= 15) { best->s.rice2 = 1; break; } /* :4786 */
}

/* ------------------------------------------------------------------------------------
 * frame assembly                        (stream_encoder_framing.c:245-594; Appendix B)
 * ---------------------------------------------------------------------------------- */
static void write_frame_header(bitw *w, const fo_config *cfg, uint32_t channel_assignment, uint64_t frame_number)
{
	const uint32_t bs = cfg->blocksize, sr = cfg->sample_rate;
	uint32_t u, bs_hint = 0, sr_hint = 0;
	const size_t start = (size_t)(w->nbits >> 3);
	bw_bits(w, 0x3ffe, 14); bw_bits(w, 0, 1); bw_bits(w, 0, 1);
	switch(bs) {
		case 192: u = 1; break; case 576: u = 2; break; case 1152: u = 3; break; case 2304: u = 4; break;
		case 4608: u = 5; break; case 256: u = 8; break; case 512: u = 9; break; case 1024: u = 10; break;
		case 2048: u = 11; break; case 4096: u = 12; break; case 8192: u = 13; break; case 16384: u = 14; break;
		case 32768: u = 15; break;
		default: bs_hint = u = (bs <= 0x100) ? 6 : 7; break;
	}
	bw_bits(w, u, 4);
	switch(sr) {
		case 88200: u = 1; break; case 176400: u = 2; break; case 192000: u = 3; break; case 8000: u = 4; break;
		case 16000: u = 5; break; case 22050: u = 6; break; case 24000: u = 7; break; case 32000: u = 8; break;
		case 44100: u = 9; break; case 48000: u = 10; break; case 96000: u = 11; break;
		default:
			if(sr <= 255000 && sr % 1000 == 0) sr_hint = u = 12;
			else if(sr <= 655350 && sr % 10 == 0) sr_hint = u = 14;
			else if(sr <= 0xffff) sr_hint = u = 13;
			else u = 0;
			break;
	}
	bw_bits(w, u, 4);
	bw_bits(w, channel_assignment == 0 ? cfg->channels - 1 : 7 + channel_assignment, 4);
	switch(cfg->bits_per_sample) {
		case 8: u = 1; break; case 12: u = 2; break; case 16: u = 4; break; case 20: u = 5; break;
		case 24: u = 6; break; case 32: u = 7; break; default: u = 0; break;
	}
	bw_bits(w, u, 3); bw_bits(w, 0, 1);
	bw_utf8_u32(w, (uint32_t)frame_number);
	if(bs_hint) bw_bits(w, bs - 1, bs_hint == 6 ? 8 : 16);
	if(sr_hint == 12) bw_bits(w, sr / 1000, 8);
	else if(sr_hint == 13) bw_bits(w, sr, 16);
	else if(sr_hint == 14) bw_bits(w, sr / 10, 16);
	if(!w->overflow) bw_bits(w, fo_crc8(w->buf + start, (size_t)(w->nbits >> 3) - start), 8);
}

static void write_subframe(bitw *w, const fo_config *cfg, const subframe_t *sf, const int64_t *sig, uint32_t bps, int32_t *residual)
{
	const uint32_t N = cfg->blocksize, wasted = sf->s.wasted_bits, order = sf->s.order;
	uint32_t type_bits;
	switch(sf->s.type) {
		case 0: type_bits = 0x00; break;
		case 1: type_bits = 0x02; break;
		case 2: type_bits = 0x10 | (order << 1); break;
		default: type_bits = 0x40 | ((order - 1) << 1); break;
	}
	bw_bits(w, type_bits | (wasted ? 1 : 0), 8);
	if(wasted) bw_unary(w, wasted - 1);
	if(sf->s.type == 0) { bw_signed(w, sf->constant, bps); return; }
	if(sf->s.type == 1) { for(uint32_t i = 0; i < N; i++) bw_signed(w, sig[i], bps); return; }
	for(uint32_t i = 0; i < order; i++) bw_signed(w, sig[i], bps);
	if(sf->s.type == 3) {
		bw_bits(w, sf->s.precision - 1, 4);
		bw_signed(w, sf->s.shift, 5);
		for(uint32_t i = 0; i < order; i++) bw_signed(w, sf->s.qlp[i], sf->s.precision);
		(void)lpc_residual(sig + order, N - order, sf->s.qlp, order, sf->s.shift, lpc_residual_mode(bps, sf->s.qlp, order, sf->s.shift), residual);
	}
	else fixed_residual(sig + order, N - order, order, bps + order > 32, residual);
	{
		const uint32_t po = sf->s.partition_order, plen = sf->s.rice2 ? 5 : 4;
		uint32_t k = 0;
		bw_bits(w, sf->s.rice2 ? 1 : 0, 2);
		bw_bits(w, po, 4);
		for(uint32_t p = 0; p < (1u << po); p++) {
			uint32_t n = (N >> po) - (p == 0 ? order : 0);
			bw_bits(w, sf->params[p], plen);
			for(uint32_t i = 0; i < n; i++) bw_rice(w, residual[k + i], sf->params[p]);
			k += n;
		}
	}
}

/* stream_encoder.c:5077 get_wasted_bits_, :5103 get_wasted_bits_wide_ (the side channel of a 32-bit stream; an all-zero
 * one loses 1 bit there) */
static uint32_t wasted_bits_of(const int64_t *sig, uint32_t n, int wide)
{
	uint64_t x = 0;
	for(uint32_t i = 0; i < n; i++) x |= (uint64_t)sig[i];
	return x ? (uint32_t)__builtin_ctzll(x) : (wide ? 1 : 0);
}

int64_t fo_encode_frame(const fo_config *cfg, const int32_t *const pcm[], uint64_t frame_number,
                        uint8_t *out, size_t cap, fo_frame_info *info)
{
	const uint32_t N = cfg->blocksize, C = cfg->channels, bps = cfg->bits_per_sample;
	if(C < 1 || C > FO_MAX_CHANNELS || bps < 4 || bps > 32 || N < 1 || N > 65535) return -1;
	if(cfg->max_lpc_order > FO_MAX_LPC_ORDER) return -1;
	const int ms = cfg->do_mid_side && C == 2;
	int do_indep = 1, do_ms = 0, loose_pick_ms = 0;
	uint32_t max_po = umin(max_po_from_blocksize(N), cfg->max_partition_order);
	uint32_t min_po = umin(cfg->min_partition_order, max_po);

	/* signals: [0..C) independent, [C], [C+1] mid and side; 4 leading zeros like the
	 * reference's buffers (stream_encoder.c:2846-2850) -- never read by valid orders */
	const uint32_t nsig = C + (ms ? 2 : 0);
	int64_t *store = (int64_t *)calloc((size_t)nsig * (N + 4), sizeof(int64_t));
	int64_t *sig[FO_MAX_CHANNELS + 2];
	uint32_t sbps[FO_MAX_CHANNELS + 2], wasted[FO_MAX_CHANNELS + 2];
	subframe_t best[FO_MAX_CHANNELS + 2], cand;
	int32_t *residual = (int32_t *)malloc(sizeof(int32_t) * (N + 1));
	float *windowed = (float *)malloc(sizeof(float) * (N + 1));
	int64_t ret;
	for(uint32_t c = 0; c < nsig; c++) sig[c] = store + (size_t)c * (N + 4) + 4;
	for(uint32_t c = 0; c < C; c++) for(uint32_t i = 0; i < N; i++) sig[c][i] = pcm[c][i];
	memset(best, 0, sizeof best); memset(&cand, 0, sizeof cand);
	cand.params = (uint32_t *)malloc(sizeof(uint32_t) << max_po);
	for(uint32_t c = 0; c < nsig; c++) best[c].params = (uint32_t *)malloc(sizeof(uint32_t) << max_po);

	if(ms) {
		if(cfg->loose_mid_side) { /* :3778-3807 */
			uint64_t lr = 0, msum = 0;
			for(uint32_t i = 1; i < N; i++) {      /* 64-bit from 25 bits per sample up; the same sums below that */
				int64_t pl = sig[0][i] - sig[0][i - 1], pr = sig[1][i] - sig[1][i - 1];
				lr += abs64(pl) + abs64(pr);
				msum += abs64((pl + pr) >> 1) + abs64(pl - pr);
			}
			if(lr < msum) { do_indep = 1; do_ms = 0; }
			else { do_indep = 0; do_ms = 1; loose_pick_ms = 1; }
		}
		else { do_indep = 1; do_ms = 1; }
	}
	if(do_ms) {
		for(uint32_t i = 0; i < N; i++) {
			sig[C + 1][i] = sig[0][i] - sig[1][i];
			sig[C][i] = (sig[0][i] + sig[1][i]) >> 1;
		}
	}
	if(do_indep) for(uint32_t c = 0; c < C; c++) {
		uint32_t w = wasted_bits_of(sig[c], N, 0);
		if(w > bps) w = bps;
		if(w) for(uint32_t i = 0; i < N; i++) sig[c][i] >>= w;
		wasted[c] = w; sbps[c] = bps - w;
	}
	if(do_ms) for(uint32_t c = 0; c < 2; c++) {
		uint32_t w = wasted_bits_of(sig[C + c], N, bps == 32 && c == 1);
		if(w > bps) w = bps;
		if(w) for(uint32_t i = 0; i < N; i++) sig[C + c][i] >>= w;
		wasted[C + c] = w; sbps[C + c] = bps - w + c;
	}
	{
		int disable_constant = (int)cfg->disable_constant, all_constant = 1;
		if(do_indep) for(uint32_t c = 0; c < C; c++) {
			if(cfg->limit_min_bitrate && all_constant && c + 1 == C) disable_constant = 1; /* :3874 */
			process_subframe(cfg, sig[c], sbps[c], wasted[c], disable_constant, min_po, max_po, &best[c], &cand, residual, windowed);
			if(best[c].s.type != 0) all_constant = 0;
		}
		if(do_ms) for(uint32_t c = 0; c < 2; c++)
			process_subframe(cfg, sig[C + c], sbps[C + c], wasted[C + c], disable_constant, min_po, max_po, &best[C + c], &cand, residual, windowed);
	}
	{
		uint32_t ca = 0, left = 0, right = 1;
		bitw w = { out, cap, 0, 0 };
		if(ms) {
			if(!cfg->loose_mid_side) {
				uint32_t bits[4], mn;
				bits[0] = best[0].s.bits + best[1].s.bits;
				bits[1] = best[0].s.bits + best[3].s.bits;
				bits[2] = best[1].s.bits + best[3].s.bits;
				bits[3] = best[2].s.bits + best[3].s.bits;
				mn = bits[0];
				for(uint32_t k = 1; k <= 3; k++) if(bits[k] < mn) { mn = bits[k]; ca = k; }
			}
			else ca = loose_pick_ms ? 3 : 0;
			switch(ca) {
				case 0: left = 0; right = 1; break;
				case 1: left = 0; right = 3; break;
				case 2: left = 3; right = 1; break;
				default: left = 2; right = 3; break;
			}
		}
		write_frame_header(&w, cfg, ca, frame_number);
		if(ms) {
			write_subframe(&w, cfg, &best[left], sig[left], sbps[left], residual);
			write_subframe(&w, cfg, &best[right], sig[right], sbps[right], residual);
			if(info) { info->sub[0] = best[left].s; info->sub[1] = best[right].s; }
		}
		else for(uint32_t c = 0; c < C; c++) {
			write_subframe(&w, cfg, &best[c], sig[c], sbps[c], residual);
			if(info) info->sub[c] = best[c].s;
		}
		if(w.nbits & 7) bw_bits(&w, 0, 8 - (uint32_t)(w.nbits & 7));       /* :3720 zero-pad */
		if(!w.overflow) bw_bits(&w, fo_crc16(out, (size_t)(w.nbits >> 3)), 16); /* :3727 */
		ret = w.overflow ? -2 : (int64_t)(w.nbits >> 3);
		if(info) { info->channel_assignment = ca; info->frame_bytes = (uint32_t)(w.nbits >> 3); }
	}
	for(uint32_t c = 0; c < nsig; c++) free(best[c].params);
	free(cand.params); free(windowed); free(residual); free(store);
	return ret;
}

int64_t fo_encode_frames(const fo_config *cfg, const fo_config *tail_cfg, const int32_t *const pcm[],
                         uint64_t total_samples, uint64_t first_frame_number,
                         uint8_t *out, size_t cap, uint32_t *frame_bytes, uint32_t *nframes)
{
	const uint32_t N = cfg->blocksize;
	uint64_t pos = 0, fn = first_frame_number;
	size_t used = 0;
	uint32_t nf = 0;
	while(pos < total_samples) {
		const uint64_t left = total_samples - pos;
		const fo_config *c = left >= N ? cfg : tail_cfg;
		const int32_t *ch[FO_MAX_CHANNELS];
		if(!c || (left < N && c->blocksize != left)) return -1;
		for(uint32_t k = 0; k < cfg->channels; k++) ch[k] = pcm[k] + pos;
		int64_t r = fo_encode_frame(c, ch, fn, out + used, cap - used, 0);
		if(r < 0) return r;
		if(frame_bytes) frame_bytes[nf] = (uint32_t)r;
		used += (size_t)r; nf++; fn++;
		pos += c->blocksize;
	}
	if(nframes) *nframes = nf;
	return (int64_t)used;
}
