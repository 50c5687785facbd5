// flac_amd/csrc/flacgpu_kernels.hip -- CDNA4 (gfx950) kernels of the FLAC frame engine.
//
// One launch encodes a batch of thousands of independent frames (the reference encodes one frame
// per thread-pool task, src/libFLAC/stream_encoder.c:3627-3744):
//
//   analyze_kernel : one workgroup per (frame, candidate channel).  Channel signal lives in LDS for
//                    the whole model search of process_subframe_ (stream_encoder.c:4045-4290):
//                    wasted bits (:5077), fixed-predictor sums (fixed.c:222 / fixed_intrin_avx2.c:57),
//                    windowing + autocorrelation in the reference's compiled association order
//                    (lpc_intrin_fma.c:46-72), Levinson-Durbin / order guess /
//                    quantisation (lpc.c:176,1608,220) and, per candidate, an integer FIR straight
//                    out of LDS feeding per-partition |residual| sums (LDS atomics) and the closed-form
//                    Rice parameter / bit estimate (stream_encoder.c:4701-5075).  Emits one small
//                    decision record; residuals never touch HBM.
//   pack_kernel    : one workgroup per frame.  Channel-assignment argmin (stream_encoder.c:3944-3972),
//                    recomputes only the winning residuals, per-symbol bit lengths -> workgroup prefix
//                    sum -> bits OR-ed into an LDS frame image (stream_encoder_framing.c:245-594,
//                    bitwriter.c:575), CRC-8 / parallel CRC-16 (crc.c:366,376), coalesced store.
//   scan/compact   : exclusive prefix sum of frame lengths, frames packed back to back.
//
// Integer/byte work, HBM/LDS/VALU bound: no MFMA by design.  Compile with -ffp-contract=off: the
// fp64 sections must round exactly like the reference binary.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "flacgpu_dev.h"

namespace flacgpu {

// ---------------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t ilog2_u32(uint32_t v) { return 31u - (uint32_t)__clz((int)v); }
__device__ __forceinline__ uint32_t ilog2_u64(uint64_t v) { return 63u - (uint32_t)__clzll((long long)v); }
__device__ __forceinline__ uint32_t silog2_i64(int64_t v)
{
	if(v == 0) return 0;
	if(v == -1) return 2;
	if(v < 0) v = -(v + 1);
	return ilog2_u64((uint64_t)v) + 2;
}
__device__ __forceinline__ uint32_t umin32(uint32_t a, uint32_t b) { return a < b ? a : b; }

// LDS signal layout: rows of 16 samples padded to 18 words so that the per-thread sliding window
// (thread t owns samples [16t,16t+16)) reads conflict-free; 32 zero samples in front so that a
// zero-padded FIR of up to 32 taps never needs a bounds check.
__device__ __forceinline__ int sigidx(int i) { const int m = i + 32; return m + ((m >> 4) << 1); }

__device__ __forceinline__ uint64_t wave_reduce_add_u64(uint64_t v)
{
#pragma unroll
	for(int off = 32; off >= 1; off >>= 1) {
		uint32_t lo = __shfl_xor((uint32_t)v, off), hi = __shfl_xor((uint32_t)(v >> 32), off);
		v += ((uint64_t)hi << 32) | lo;
	}
	return v;
}
__device__ __forceinline__ uint32_t wave_reduce_or_u32(uint32_t v)
{
#pragma unroll
	for(int off = 32; off >= 1; off >>= 1) v |= __shfl_xor(v, off);
	return v;
}
// workgroup reductions through a small LDS scratch (8 x u64)
__device__ __forceinline__ uint64_t block_reduce_add_u64(uint64_t v, uint64_t *scratch, int tid)
{
	v = wave_reduce_add_u64(v);
	__syncthreads();
	if((tid & 63) == 0) scratch[tid >> 6] = v;
	__syncthreads();
	uint64_t r = 0;
	for(int w = 0; w < TPB / 64; w++) r += scratch[w];
	return r;
}
__device__ __forceinline__ uint32_t block_reduce_or_u32(uint32_t v, uint64_t *scratch, int tid)
{
	v = wave_reduce_or_u32(v);
	__syncthreads();
	if((tid & 63) == 0) scratch[tid >> 6] = v;
	__syncthreads();
	uint32_t r = 0;
	for(int w = 0; w < TPB / 64; w++) r |= (uint32_t)scratch[w];
	return r;
}

// ---------------------------------------------------------------------------------------------
// fp64 model stage: Levinson-Durbin + order guess + quantisation, ONE LANE PER ANALYSIS with the whole
// recursion in registers (fully unrolled, statically indexed) -- lpc.c:176-314,1580-1630 as the
// reference binary computes them (see oracle/flac_oracle.c for the compiled-behaviour notes)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double expected_bits_scaled(double lpc_error, double error_scale)
{
	if(lpc_error > 0.0) {
		// 0.5*log(x)/M_LN2 folded by -freciprocal-math into log(x) * (0.5/ln 2)
		double bps = log(error_scale * lpc_error) * 0.7213475204444817;
		return bps >= 0.0 ? bps : 0.0;
	}
	if(lpc_error < 0.0) return 1e32;
	return 0.0;
}

// One Levinson-Durbin recursion up to `upto` orders (lpc.c:188-217). lpc[] / errs[] live in registers.
// Returns the number of orders actually produced (stops early when err == 0, lpc.c:213).
template <int MAXORD, bool ROWS>
__device__ __forceinline__ uint32_t levinson(const double (&a)[MAXORD + 1], uint32_t upto, double (&lpc)[MAXORD], double (&errs)[MAXORD], float *rows)
{
	double err = a[0];
	uint32_t used = upto;
#pragma unroll
	for(int i = 0; i < MAXORD; i++) {
		if((uint32_t)i < used) {
			double r = -a[i + 1];
#pragma unroll
			for(int j = 0; j < i; j++) r -= lpc[j] * a[i - j];
			r /= err;
			lpc[i] = r;
#pragma unroll
			for(int j = 0; j < (i >> 1); j++) {
				const double tmp = lpc[j];
				lpc[j] += r * lpc[i - 1 - j];
				lpc[i - 1 - j] += r * tmp;
			}
			if(i & 1) lpc[i >> 1] = (r + 1.0) * lpc[i >> 1];   // compiled form of lpc[j] += lpc[j]*r
			err *= (1.0 - r * r);
			errs[i] = err;
			if(ROWS) {
				// lp_coeff[i][0..i] of lpc.c:208-209, kept so that the guessed order needs no second pass
#pragma unroll
				for(int j = 0; j <= i; j++) rows[i * MAXORD + j] = (float)(-lpc[j]);
			}
			if(err == 0.0) used = (uint32_t)i + 1;
		}
	}
	return used;
}

// autoc: lag values of this analysis (already punched-out when applicable). rows: MAXORD*MAXORD floats of
// LDS scratch private to this lane, or null (then the recursion is simply run twice). Returns 0 when no LPC
// candidate results (autoc[0]==0, estimate >= bps, quantiser failure, residual would need the >32-bit
// "limit_residual" flavour).
template <int MAXORD>
__device__ int lpc_model(const double (&a)[MAXORD + 1], uint32_t max_order, uint32_t n, uint32_t sbps,
                         uint32_t cfg_precision, float *rows, Candidate *out)
{
	double lpc[MAXORD], errs[MAXORD];
	if(a[0] == 0.0) return 0;
#pragma unroll
	for(int i = 0; i < MAXORD; i++) { lpc[i] = 0.0; errs[i] = 0.0; }
	const uint32_t used = rows ? levinson<MAXORD, true>(a, max_order, lpc, errs, rows) : levinson<MAXORD, false>(a, max_order, lpc, errs, rows);
	// FLAC__lpc_compute_best_order (lpc.c:1608): total_samples is the full blocksize
	uint32_t order = 1;
	double err_order = errs[0];
	{
		const double scale = 0.5 / (double)n;
		const uint32_t overhead = sbps + cfg_precision;
		double best_bits = 4294967295.0;
#pragma unroll
		for(int idx = 0; idx < MAXORD; idx++) {
			if((uint32_t)idx < used) {
				const uint32_t o = (uint32_t)idx + 1;
				const double bits = expected_bits_scaled(errs[idx], scale) * (double)(n - o) + (double)(o * overhead);
				if(bits < best_bits) { order = o; best_bits = bits; err_order = errs[idx]; }
			}
		}
	}
	// stream_encoder.c:4227-4229
	if(expected_bits_scaled(err_order, 0.5 / (double)(n - order)) >= (double)sbps) return 0;
	float coef[MAXORD];
	if(rows) {
#pragma unroll
		for(int j = 0; j < MAXORD; j++) coef[j] = (uint32_t)j < order ? rows[(order - 1) * MAXORD + j] : 0.0f;
	}
	else {
		// coefficients of `order`: rerun the (deterministic) recursion up to that order
		(void)levinson<MAXORD, false>(a, order, lpc, errs, rows);
#pragma unroll
		for(int j = 0; j < MAXORD; j++) coef[j] = (uint32_t)j < order ? (float)(-lpc[j]) : 0.0f;
	}
	// stream_encoder.c:4591-4595 then FLAC__lpc_quantize_coefficients (lpc.c:220)
	uint32_t precision = cfg_precision;
	if(sbps <= 17) precision = umin32(precision, 32 - sbps - ilog2_u32(order));
	int shift;
	int32_t q[MAXORD];
	{
		const uint32_t p1 = precision - 1;
		int32_t qmax = (int32_t)1 << p1, qmin = -qmax;
		qmax--;
		double cmax = 0.0;
#pragma unroll
		for(int i = 0; i < MAXORD; i++) { const double v = fabs((double)coef[i]); if(v > cmax) cmax = v; }
		if(cmax <= 0.0) return 0;
		int e;
		(void)frexp(cmax, &e);
		e--;
		shift = (int)p1 - e - 1;
		if(shift > 15) shift = 15;
		else if(shift < -16) return 0;
		double error = 0.0;
		const bool neg = shift < 0;
		const float scale = (float)(1 << (neg ? -shift : shift));
#pragma unroll
		for(int i = 0; i < MAXORD; i++) {
			int32_t v = 0;
			if((uint32_t)i < order) {
				error += neg ? (double)(coef[i] / scale) : (double)(coef[i] * scale);
				v = (int32_t)lround(error);
				if(v > qmax) v = qmax; else if(v < qmin) v = qmin;
				error -= v;
			}
			q[i] = v;
		}
		if(neg) shift = 0;
	}
	// residual kernel selector (stream_encoder.c:4601-4617, lpc.c:942-976)
	{
		uint32_t abs_sum = 0;
#pragma unroll
		for(int i = 0; i < MAXORD; i++) abs_sum += (uint32_t)abs(q[i]);
		const uint64_t maxabs = (uint64_t)1 << (sbps - 1);
		const uint64_t before = maxabs * abs_sum;
		const uint64_t after = (uint64_t)(-1 * ((-1 * (int64_t)before) >> shift));
		if(silog2_i64((int64_t)(maxabs + after)) > 32) return 0;
		out->wide = silog2_i64((int64_t)before) > 32;
	}
#pragma unroll
	for(int i = 0; i < MAX_ORDER; i++) out->q[i] = i < MAXORD ? q[i < MAXORD ? i : 0] : 0;
	out->order = order;
	out->precision = precision;
	out->shift = shift;
	return 1;
}

// ---------------------------------------------------------------------------------------------
// autocorrelation in the association order of the reference's compiled routines:
// one lane per chain = (window job, lag j, vector lane l); the 4 lane accumulators of a lag are
// combined as (acc3+acc1)+(acc2+acc0) afterwards, then the scalar head/tail of lpc_intrin_fma.c.
// d = windowed data of the job in LDS, nd = its data_len.
// ---------------------------------------------------------------------------------------------
#define DD(k) ((double)d[k])
// One lane per chain (window job, lag j, vector lane l); a wavefront holds all 64 chains of one job, so its
// LDS reads are broadcasts (x) or a run of <= 19 consecutive words (y): conflict free.  The loads of several
// steps are issued together so LDS latency overlaps the fp64 work of the previous steps.
// body of lpc_intrin_fma.c:46,61 (lag 8 / lag 16): acc_l += fma(d[i],d[i-j], d[i+4]*d[i+4-j]), i = L+8k+l
__device__ __forceinline__ double autoc_chain_8_16(const float *d, uint32_t nd, uint32_t L, uint32_t j, uint32_t l)
{
	const uint32_t nb = (nd - L) / 8;
	const float *px = d + L + l, *py = px - j;
	double acc = 0.0;
	uint32_t k = 0;
	for(; k + 4 <= nb; k += 4, px += 32, py += 32) {
		float x[8], y[8];
#pragma unroll
		for(int u = 0; u < 8; u++) { x[u] = px[4 * u]; y[u] = py[4 * u]; }
#pragma unroll
		for(int u = 0; u < 4; u++)
			acc += fma((double)x[2 * u], (double)y[2 * u], (double)x[2 * u + 1] * (double)y[2 * u + 1]);
	}
	for(; k < nb; k++, px += 8, py += 8)
		acc += fma((double)px[0], (double)py[0], (double)px[4] * (double)py[4]);
	return acc;
}
// body of lpc_intrin_fma.c:54 (lag 12): gcc unrolled the 8-sample body x2 (acc += t1+t0 per 16 samples) and,
// for lag 8 only, factored x*y0+x*y2 -> x*(y0+y2) across the two halves (y2 == x0 of the next half there)
__device__ __forceinline__ double autoc_chain_12(const float *d, uint32_t nd, uint32_t j, uint32_t l)
{
	const uint32_t L = 12;
	const uint32_t nb = (nd - L) / 8;
	const uint32_t npairs = nb > 2 ? ((nb - 3) & ~1u) / 2 + 1 : 0;
	const float *px = d + L + l, *py = px - j;
	double acc = 0.0;
	uint32_t k = 0, p = 0;
	if(j == 8) {
		for(; p + 2 <= npairs; p += 2, k += 4, px += 32, py += 32) {
			float x[8], y[4];
#pragma unroll
			for(int u = 0; u < 8; u++) x[u] = px[4 * u];
			y[0] = py[0]; y[1] = py[4]; y[2] = py[16]; y[3] = py[20];
			acc += fma((double)x[0], ((double)y[0] + (double)x[2]), (double)x[1] * ((double)y[1] + (double)x[3]));
			acc += fma((double)x[4], ((double)y[2] + (double)x[6]), (double)x[5] * ((double)y[3] + (double)x[7]));
		}
		for(; p < npairs; p++, k += 2, px += 16, py += 16)
			acc += fma((double)px[0], ((double)py[0] + (double)px[8]), (double)px[4] * ((double)py[4] + (double)px[12]));
	}
	else {
		for(; p + 2 <= npairs; p += 2, k += 4, px += 32, py += 32) {
			float x[8], y[8];
#pragma unroll
			for(int u = 0; u < 8; u++) { x[u] = px[4 * u]; y[u] = py[4 * u]; }
#pragma unroll
			for(int h = 0; h < 2; h++) {
				const double t0 = fma((double)x[4 * h], (double)y[4 * h], (double)x[4 * h + 1] * (double)y[4 * h + 1]);
				const double t1 = fma((double)x[4 * h + 2], (double)y[4 * h + 2], (double)x[4 * h + 3] * (double)y[4 * h + 3]);
				acc += (t1 + t0);
			}
		}
		for(; p < npairs; p++, k += 2, px += 16, py += 16) {
			const double t0 = fma((double)px[0], (double)py[0], (double)px[4] * (double)py[4]);
			const double t1 = fma((double)px[8], (double)py[8], (double)px[12] * (double)py[12]);
			acc += (t1 + t0);
		}
	}
	for(; k < nb; k++, px += 8, py += 8)
		acc += fma((double)px[0], (double)py[0], (double)px[4] * (double)py[4]);
	return acc;
}
// scalar head (samples j..L-1), lane combine and tail for lag j
__device__ __forceinline__ double autoc_finish(const float *d, uint32_t nd, uint32_t L, uint32_t j, const double *acc4)
{
	double a = 0.0;
	for(uint32_t h = j; h < L; h++) a += DD(h) * DD(h - j);
	const uint32_t nb = (nd - L) / 8;
	if(nb) a = ((acc4[3] + acc4[1]) + (acc4[2] + acc4[0])) + a;
	uint32_t i = L + 8 * nb;
	if(nd - i >= 4) {
		const double hi = fma(DD(i + 1), DD(i + 1 - j), DD(i + 3) * DD(i + 3 - j));
		const double lo = fma(DD(i), DD(i - j), DD(i + 2) * DD(i + 2 - j));
		a = (hi + lo) + a;
		i += 4;
	}
	for(; i < nd; i++) a = fma(DD(i), DD(i - j), a);
	return a;
}
// lpc.c:133-157 (blocksize <= 32): plain sequential accumulation per lag
__device__ __forceinline__ double autoc_small(const float *d, uint32_t nd, uint32_t c)
{
	double a = 0.0;
	for(uint32_t s = 0; s + c < nd; s++) a += DD(s) * DD(s + c);
	return a;
}
#undef DD
// ---------------------------------------------------------------------------------------------
// shared between analyze and pack: build the candidate channel's signal in LDS
// returns the OR of all samples (for wasted bits) reduced over the workgroup
// ---------------------------------------------------------------------------------------------
// which: 0..C-1 independent channel, C = mid, C+1 = side
__device__ __forceinline__ int32_t pick_channel(const int32_t *frame_pcm, uint32_t C, uint32_t i, uint32_t which)
{
	if(which < C) return frame_pcm[(size_t)i * C + which];
	const int32_t l = frame_pcm[(size_t)i * 2], r = frame_pcm[(size_t)i * 2 + 1];
	return which == C ? ((l + r) >> 1) : (l - r);
}

__device__ void load_signal(int32_t *sig, const int32_t *frame_pcm, uint32_t C, uint32_t n, uint32_t which,
                            uint32_t *or_out, int tid)
{
	uint32_t orv = 0;
	// zero the 32-sample front pad and the tail up to the next chunk boundary + one chunk
	if(tid < 32) sig[sigidx(tid - 32)] = 0;
	const uint32_t nround = ((n + 15u) & ~15u) + 16u;
	for(uint32_t i = n + (uint32_t)tid; i < nround; i += TPB) sig[sigidx((int)i)] = 0;
	if(C == 2) {
		const int2 *p = (const int2 *)frame_pcm;
		for(uint32_t i = (uint32_t)tid; i < n; i += TPB) {
			const int2 lr = p[i];
			int32_t v = which == 0 ? lr.x : which == 1 ? lr.y : which == 2 ? ((lr.x + lr.y) >> 1) : (lr.x - lr.y);
			sig[sigidx((int)i)] = v;
			orv |= (uint32_t)v;
		}
	}
	else {
		for(uint32_t i = (uint32_t)tid; i < n; i += TPB) {
			int32_t v = pick_channel(frame_pcm, C, i, which);
			sig[sigidx((int)i)] = v;
			orv |= (uint32_t)v;
		}
	}
	*or_out = orv;
}

// residual of CHUNK consecutive samples starting at `base` with a zero-padded MAXORD-tap FIR
// (lpc.c:321 32-bit wrapping / lpc.c:582 64-bit accumulate; fixed.c:470 is the same FIR with binomial taps)
// MODE 0: 32-bit wrapping accumulate with 24-bit multiplies (lpc.c:321; valid when samples fit 24 bits signed and
//         |tap| < 2^23: the low 32 bits of the product are the same, at the full VALU rate of v_mad_i32_i24)
// MODE 1: 32-bit wrapping accumulate, full 32-bit multiplies (lpc.c:321)
// MODE 2: 64-bit accumulate (lpc.c:582)
template <int MAXORD, int MODE>
__device__ __forceinline__ void fir_chunk(const int32_t *sig, int base, const int32_t *q, int shift, int32_t *r)
{
	int32_t x[MAXORD + CHUNK];
#pragma unroll
	for(int k = 0; k < MAXORD + CHUNK; k++) x[k] = sig[sigidx(base - MAXORD + k)];
#pragma unroll
	for(int s = 0; s < CHUNK; s++) {
		if(MODE == 2) {
			int64_t sum = 0;
#pragma unroll
			for(int j = 0; j < MAXORD; j++) sum += (int64_t)q[j] * (int64_t)x[MAXORD + s - 1 - j];
			r[s] = (int32_t)((int64_t)x[MAXORD + s] - (sum >> shift));
		}
		else {
			uint32_t sum = 0;
#pragma unroll
			for(int j = 0; j < MAXORD; j++)
				sum += MODE == 0 ? (uint32_t)__mul24(q[j], x[MAXORD + s - 1 - j]) : (uint32_t)q[j] * (uint32_t)x[MAXORD + s - 1 - j];
			r[s] = (int32_t)((uint32_t)x[MAXORD + s] - (uint32_t)((int32_t)sum >> shift));
		}
	}
}

// mode: 0/1/2 as above (wave-uniform)
template <int MAXORD>
__device__ __forceinline__ void fir_chunk_dispatch(const int32_t *sig, int base, const int32_t *q, int shift, int mode, int32_t *r)
{
	if(mode == 0) fir_chunk<MAXORD, 0>(sig, base, q, shift, r);
	else if(mode == 1) fir_chunk<MAXORD, 1>(sig, base, q, shift, r);
	else fir_chunk<MAXORD, 2>(sig, base, q, shift, r);
}
__device__ __forceinline__ int fir_mode(bool wide, uint32_t sbps) { return wide ? 2 : (sbps <= 24 ? 0 : 1); }

// ---------------------------------------------------------------------------------------------
// analyze_kernel
// ---------------------------------------------------------------------------------------------
// fixed-size part of the workgroup's LDS state; the large arrays are carved dynamically (analyze_layout)
struct AnalyzeSmall {
	uint64_t scratch[8];
	uint64_t pob[TPB / 64][MAX_PO + 1];           // slow path: per-wave bit totals per partition order
	uint32_t divtab[(MAX_PO + 1) * (MAX_ORDER + 1)]; // 0x40000 / ((n >> po) - order)
	int cand_valid[MAX_ANALYSES + 1];
	uint32_t wbest_bits[TPB / 64], wbest_ci[TPB / 64], wbest_po[TPB / 64];
};

struct AnalyzeLayout { uint32_t wsums, accs, autoc, cands, kbestw, kcandw, small, total; };
__host__ __device__ inline AnalyzeLayout analyze_layout(const DevParams &P)
{
	AnalyzeLayout L;
	uint32_t o = P.sig_bytes + P.wnd_bytes;
	L.wsums = o;  o += (TPB / 64) * (2u << P.max_po) * 8;                   // per-wave partition sums
	L.accs = o;   o += P.max_jobs * (P.max_lpc_order + 1) * 4 * 8;          // chain accumulators
	L.autoc = o;  o += P.max_jobs * MAX_ORDER * 8;                          // finished autocorrelations
	L.cands = o;  o += (P.max_analyses + 1) * (uint32_t)sizeof(Candidate);  // [0] fixed, [1+a] LPC analysis a
	L.kbestw = o; o += (TPB / 64) * 2 * (1u << P.max_po);                   // per wave: Rice parameters of its best candidate + scratch
	L.kcandw = o; o += P.max_po > 6 ? (TPB / 64) * (2u << P.max_po) : 0;    // slow path only
	o = (o + 15u) & ~15u;
	L.small = o;  o += (uint32_t)sizeof(AnalyzeSmall);
	L.total = (o + 15u) & ~15u;
	return L;
}

__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int mask)
{
	const uint32_t lo = __shfl_xor((uint32_t)v, mask), hi = __shfl_xor((uint32_t)(v >> 32), mask);
	return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ uint32_t sat_add_u32(uint32_t est, uint32_t rbits)
{
	return rbits < 0xffffffffu - est ? est + rbits : 0xffffffffu;
}

// One WAVEFRONT evaluates one residual candidate (fixed or LPC) without any workgroup barrier:
// integer FIR out of LDS (lane owns S consecutive samples), |residual| per partition, the flat tree of
// merged sums, Rice parameter and bit estimate per partition, best partition order
// (find_best_partition_order_ / precompute_partition_info_sums_ / set_partitioned_rice_,
// stream_encoder.c:4701-5075).  Returns the estimated residual bits; Rice parameters of the best
// partition order go to kout[0 .. 2^best_po).
template <int MAXORD>
__device__ uint32_t eval_candidate_wave(uint64_t *wsums, uint8_t *kcand, uint64_t *pob, uint8_t *kout, const uint32_t *divtab,
                                        const int32_t *sig, uint32_t n, uint32_t order, const int32_t *q, int shift, bool wide,
                                        uint32_t sbps, const DevParams &P, uint32_t frame_max_po, uint32_t frame_min_po,
                                        uint32_t *best_po_out, int lane)
{
	uint32_t max_po = frame_max_po;
	while(max_po > 0 && (n >> max_po) <= order) max_po--;                   // format.c:550
	const uint32_t min_po = umin32(frame_min_po, max_po);
	const uint32_t psize = n >> max_po, nparts = 1u << max_po;
	const bool narrow = (sbps + 4) < (32 - ilog2_u32(psize));               // stream_encoder.c:4814-4817
	int32_t qr[MAXORD];
#pragma unroll
	for(int j = 0; j < MAXORD; j++) qr[j] = q[j];
	const int fmode = fir_mode(wide, sbps);

	// Lane `lane` owns chunks lane, lane+64, ... of CHUNK consecutive samples: adjacent lanes read adjacent
	// 18-word rows of the padded signal, i.e. conflict-free LDS reads.
	const uint32_t nchunks = (n + CHUNK - 1) / CHUNK;
	const uint32_t g = psize / CHUNK;                                       // chunks per leaf partition
	// fast path: a leaf partition is g = 2^a adjacent lanes of one pass (e.g. 4096 samples: 64 partitions of 4 chunks)
	const bool direct = max_po <= 6 && psize % CHUNK == 0 && g >= 1 && g <= 64 && (g & (g - 1)) == 0;
	uint64_t vdirect = 0;             // direct path: leaf sum of partition `lane`
	if(!direct) {
		for(uint32_t p = (uint32_t)lane; p < nparts; p += 64) wsums[p] = 0;
		__builtin_amdgcn_wave_barrier();
	}
	if(direct) {
		const uint32_t lp = 64u / g;                      // leaves per pass (power of two)
		const uint32_t src = ((uint32_t)lane & (lp - 1)) * g, want = (uint32_t)lane / lp;
#pragma unroll 1
		for(uint32_t pass = 0; pass * 64 < nchunks; pass++) {
			const uint32_t base = (pass * 64 + (uint32_t)lane) * CHUNK;
			uint64_t mine = 0;
			if(base < n) {
				int32_t r[CHUNK];
				fir_chunk_dispatch<MAXORD>(sig, (int)base, qr, shift, fmode, r);
				uint32_t acc32 = 0;
				uint64_t acc64 = 0;
#pragma unroll
				for(int s2 = 0; s2 < CHUNK; s2++) {
					const uint32_t i = base + s2;
					if(i >= order && i < n) {
						const int32_t v = r[s2];
						const uint32_t av = (uint32_t)(v < 0 ? -(uint32_t)v : (uint32_t)v);
						if(narrow) acc32 += av; else acc64 += av;
					}
				}
				mine = narrow ? (uint64_t)acc32 : acc64;
			}
			for(uint32_t m = 1; m < g; m <<= 1) mine += shfl_xor_u64(mine, (int)m);
			// leaf p = pass*lp + lane/g sits in every lane of its group; lane L wants leaf L
			const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)mine, (int)src), hi = (uint32_t)__shfl((int)(uint32_t)(mine >> 32), (int)src);
			if(want == pass) vdirect = ((uint64_t)hi << 32) | lo;
		}
		if((uint32_t)lane >= nparts) vdirect = 0;
	}
	else {
		for(uint32_t cidx = (uint32_t)lane; cidx < nchunks; cidx += 64) {
			const uint32_t base = cidx * CHUNK;
			int32_t r[CHUNK];
			fir_chunk_dispatch<MAXORD>(sig, (int)base, qr, shift, fmode, r);
			uint32_t part = base / psize, next = (part + 1) * psize;
			uint64_t run = 0;
#pragma unroll
			for(int s2 = 0; s2 < CHUNK; s2++) {
				const uint32_t i = base + s2;
				if(i == next) {
					if(run) atomicAdd((unsigned long long *)&wsums[part], (unsigned long long)run);
					run = 0; part++; next += psize;
				}
				if(i >= order && i < n) { const int32_t v = r[s2]; run += (uint32_t)(v < 0 ? -(uint32_t)v : (uint32_t)v); }
			}
			if(run && part < nparts) atomicAdd((unsigned long long *)&wsums[part], (unsigned long long)run);
		}
	}
	uint32_t best_bits = 0, best_po = 0;
	if(max_po <= 6) {
		// leaves into lanes 0..nparts-1, merged level by level with a butterfly; every lane of a group holds
		// the group's sum, the group's first lane speaks for the partition
		uint64_t v;
		if(direct) v = vdirect;
		else { __builtin_amdgcn_wave_barrier(); v = (uint32_t)lane < nparts ? wsums[lane] : 0; }
		if(narrow) v = (uint32_t)v;
		// (1) merged sums of every level: 6 dependent butterfly stages
		uint64_t vlev[7];
		vlev[0] = v;
#pragma unroll
		for(int d = 1; d <= 6; d++) {
			if((uint32_t)d <= max_po - min_po) v += shfl_xor_u64(v, 1 << (d - 1));
			vlev[d] = v;
		}
		// (2) Rice parameter and bit estimate of this lane's partition at every level (independent VALU work)
		uint32_t klev[7];
		uint64_t blev[7];
		bool big = false;
#pragma unroll
		for(int d = 0; d <= 6; d++) {
			klev[d] = 0; blev[d] = 0;
			if((uint32_t)d <= max_po - min_po) {
				const uint32_t po = max_po - (uint32_t)d;
				const uint32_t pidx = (uint32_t)lane >> d;
				const bool rep = ((uint32_t)lane & ((1u << d) - 1u)) == 0 && (uint32_t)lane < nparts;
				const uint32_t o = pidx == 0 ? order : 0;
				const uint32_t ns = (n >> po) - o;
				const uint32_t div = divtab[po * (MAX_ORDER + 1) + o];
				const uint64_t sum = vlev[d];
				uint32_t k;
				if(sum < 2 || (((sum - 1) * div) >> 18) == 0) k = 0;
				else k = ilog2_u64(((sum - 1) * div) >> 18) + 1;
				if(k >= P.rice_limit) k = P.rice_limit - 1;
				uint64_t bb = 4 + (uint64_t)(1 + k) * ns + (k ? (sum >> (k - 1)) : (sum << 1)) - (ns >> 1);
				if(bb > 0xffffffffull) bb = 0xffffffffull;
				klev[d] = k;
				blev[d] = rep ? bb : 0;
				big |= bb >= (1ull << 25);
			}
		}
		// (3) totals per level: all levels reduced together so the cross-lane latency overlaps.
		// 64 terms below 2^25 fit 32 bits -- the usual case; otherwise reduce in 64 bits.
		if(!__any((int)big)) {
			uint32_t t[7];
#pragma unroll
			for(int d = 0; d <= 6; d++) t[d] = (uint32_t)blev[d];
#pragma unroll
			for(int off = 32; off >= 1; off >>= 1) {
#pragma unroll
				for(int d = 0; d <= 6; d++) t[d] += __shfl_xor(t[d], off);
			}
#pragma unroll
			for(int d = 0; d <= 6; d++) blev[d] = t[d];
		}
		else {
#pragma unroll
			for(int d = 0; d <= 6; d++) blev[d] = wave_reduce_add_u64(blev[d]);
		}
#pragma unroll
		for(int d = 0; d <= 6; d++) {
			if((uint32_t)d <= max_po - min_po) {
				const uint64_t tot = 6 + blev[d];
				const uint32_t bits = tot >= 0xffffffffull ? 0xffffffffu : (uint32_t)tot;
				if(best_bits == 0 || bits < best_bits) { best_bits = bits; best_po = max_po - (uint32_t)d; }
			}
		}
		const uint32_t db = max_po - best_po;
		uint32_t kk = 0;
#pragma unroll
		for(int d = 0; d <= 6; d++) if((uint32_t)d == db) kk = klev[d];
		if((((uint32_t)lane & ((1u << db) - 1u)) == 0) && (uint32_t)lane < nparts) kout[(uint32_t)lane >> db] = (uint8_t)kk;
	}
	else {
		// partition orders 7/8: same computation through LDS (wave-private arrays)
		__builtin_amdgcn_wave_barrier();
		if(lane <= MAX_PO) pob[lane] = 0;
		__builtin_amdgcn_wave_barrier();
		uint32_t total_nodes = 0;
		for(int po = (int)max_po; po >= (int)min_po; po--) total_nodes += 1u << po;
		for(uint32_t node = (uint32_t)lane; node < total_nodes; node += 64) {
			uint32_t po = max_po, off = 0;
			while(node - off >= (1u << po)) { off += 1u << po; po--; }
			const uint32_t p = node - off, nleaf = 1u << (max_po - po);
			uint64_t sum = 0;
			for(uint32_t k = 0; k < nleaf; k++) { const uint64_t t = wsums[p * nleaf + k]; sum += narrow ? (uint64_t)(uint32_t)t : t; }
			const uint32_t o = p == 0 ? order : 0, ns = (n >> po) - o, div = divtab[po * (MAX_ORDER + 1) + o];
			uint32_t k;
			if(sum < 2 || (((sum - 1) * div) >> 18) == 0) k = 0;
			else k = ilog2_u64(((sum - 1) * div) >> 18) + 1;
			if(k >= P.rice_limit) k = P.rice_limit - 1;
			uint64_t b = 4 + (uint64_t)(1 + k) * ns + (k ? (sum >> (k - 1)) : (sum << 1)) - (ns >> 1);
			if(b > 0xffffffffull) b = 0xffffffffull;
			kcand[node] = (uint8_t)k;
			atomicAdd((unsigned long long *)&pob[po], (unsigned long long)b);
		}
		__builtin_amdgcn_wave_barrier();
		uint32_t off = 0, best_off = 0;
		for(int po = (int)max_po; po >= (int)min_po; po--) {
			const uint64_t b = 6 + pob[po];
			const uint32_t bits = b >= 0xffffffffull ? 0xffffffffu : (uint32_t)b;
			if(best_bits == 0 || bits < best_bits) { best_bits = bits; best_po = (uint32_t)po; best_off = off; }
			off += 1u << po;
		}
		for(uint32_t p = (uint32_t)lane; p < (1u << best_po); p += 64) kout[p] = kcand[best_off + p];
	}
	__builtin_amdgcn_wave_barrier();
	*best_po_out = best_po;
	return best_bits;
}

template <int MAXORD>
__global__ __launch_bounds__(TPB, 2) void analyze_kernel(const DevParams P, const int32_t *__restrict__ pcm,
                                                      const float *__restrict__ windows,
                                                      const float *__restrict__ tail_windows,
                                                      uint32_t nframes, uint32_t tail_n,
                                                      const JobTable *__restrict__ jt_main, const JobTable *__restrict__ jt_tail,
                                                      SubDecision *__restrict__ decisions,
                                                      unsigned long long *__restrict__ dbg)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
	// optional phase stamps (FLACGPU_DEBUG_TIMING=1): s_memtime at phase boundaries of every workgroup
#define STAMP(k) do { if(dbg && tid == 0) dbg[(size_t)blockIdx.x * 16 + (k)] = (unsigned long long)clock64(); } while(0)
	STAMP(0);
	const uint32_t C = P.channels, N = P.blocksize;

	// XCD-aware mapping: consecutive workgroup ids round-robin over the 8 XCDs; keep the candidate
	// channels of one frame on one XCD so their shared PCM lines hit that XCD's L2.
	uint32_t f, cand;
	{
		const uint32_t b = blockIdx.x, total = nframes * P.ncand;
		const uint32_t per_xcd_full = (total / (8 * P.ncand)) * P.ncand;   // blocks per XCD in the evenly divisible part
		const uint32_t head = per_xcd_full * 8;
		uint32_t lin;
		if(b < head) { const uint32_t x = b & 7, k = b >> 3; lin = x * per_xcd_full + k; }
		else lin = b;
		f = lin / P.ncand; cand = lin % P.ncand;
	}
	const bool is_tail = tail_n != 0 && f == nframes - 1;
	const uint32_t n = is_tail ? tail_n : N;
	const float *win = is_tail ? tail_windows : windows;
	const JobTable *jt = is_tail ? jt_tail : jt_main;
	const int32_t *frame_pcm = pcm + (size_t)f * N * C;

	const AnalyzeLayout LY = analyze_layout(P);
	int32_t *sig = (int32_t *)smem;
	float *wnd = (float *)(smem + P.sig_bytes);
	uint64_t *wsums_all = (uint64_t *)(smem + LY.wsums);
	double *accs = (double *)(smem + LY.accs);
	double *autoc_job = (double *)(smem + LY.autoc);
	Candidate *cands = (Candidate *)(smem + LY.cands);
	uint8_t *kbestw_all = smem + LY.kbestw;
	uint8_t *kcandw_all = smem + LY.kcandw;
	AnalyzeSmall *sh = (AnalyzeSmall *)(smem + LY.small);

	SubDecision *dec = decisions + (size_t)f * P.ncand + cand;

	// ---- which signal does this workgroup model? ----------------------------------------------
	uint32_t which = cand;          // index into {ch0..chC-1, mid, side}
	if(P.ms_mode == 2) {
		// loose mid/side (stream_encoder.c:3778-3807): both workgroups of the frame compute the decision
		uint64_t lr = 0, ms = 0;
		const int2 *p = (const int2 *)frame_pcm;
		for(uint32_t i = 1 + (uint32_t)tid; i < n; i += TPB) {
			const int2 a = p[i], b = p[i - 1];
			const int32_t pl = a.x - b.x, pr = a.y - b.y;
			lr += (uint64_t)(uint32_t)(abs(pl) + abs(pr));
			ms += (uint64_t)(uint32_t)(abs((pl + pr) >> 1) + abs(pl - pr));
		}
		lr = block_reduce_add_u64(lr, sh->scratch, tid);
		ms = block_reduce_add_u64(ms, sh->scratch, tid);
		if(!(lr < ms)) which = 2 + cand;   // mid, side
	}

	// limit_min_bitrate (stream_encoder.c:3874-3879): the last independent channel (and then mid/side)
	// may not be CONSTANT when every earlier channel is
	bool disable_constant = P.disable_constant != 0;
	if(P.limit_min_bitrate && !disable_constant && (P.ms_mode == 2 ? which == 1 : which >= C - 1)) {
		// are channels 0..C-2 all constant over this block? (their best subframe is CONSTANT iff so)
		uint32_t diff = 0;
		for(uint32_t i = (uint32_t)tid; i < n; i += TPB)
			for(uint32_t c = 0; c + 1 < C; c++) diff |= (uint32_t)(frame_pcm[(size_t)i * C + c] ^ frame_pcm[c]);
		diff = block_reduce_or_u32(diff, sh->scratch, tid);
		if(diff == 0) disable_constant = true;
	}

	// ---- signal into LDS, wasted bits (stream_encoder.c:3842-3867,5077) --------------------------
	uint32_t orv;
	load_signal(sig, frame_pcm, C, n, which, &orv, tid);
	orv = block_reduce_or_u32(orv, sh->scratch, tid);
	uint32_t wasted = orv ? (uint32_t)(__ffs((int)orv) - 1) : 0;
	if(wasted > P.bps) wasted = P.bps;
	if(wasted) {
		for(uint32_t i = (uint32_t)tid; i < n; i += TPB) sig[sigidx((int)i)] >>= wasted;
	}
	const uint32_t sbps = P.bps - wasted + (which == C + 1 ? 1 : 0);
	const uint32_t hdr = 8 + wasted;
	STAMP(1);
	// reciprocal table of set_partitioned_rice_ (stream_encoder.c:4997,5009)
	for(uint32_t t = (uint32_t)tid; t < (MAX_PO + 1) * (MAX_ORDER + 1); t += TPB) {
		const uint32_t po = t / (MAX_ORDER + 1), o = t - po * (MAX_ORDER + 1);
		const uint32_t ps = n >> po;
		sh->divtab[t] = ps > o ? 0x40000u / (ps - o) : 0;
	}
	__syncthreads();

	// partition order limits of the frame (stream_encoder.c:3759-3761)
	uint32_t frame_max_po = 0;
	{ uint32_t b = n; while(!(b & 1)) { frame_max_po++; b >>= 1; } if(frame_max_po > 15) frame_max_po = 15; }
	frame_max_po = umin32(frame_max_po, P.max_po);
	const uint32_t frame_min_po = umin32(P.min_po, frame_max_po);

	// ---- baseline: VERBATIM (stream_encoder.c:4081-4086) ------------------------------------------
	uint32_t best_type = 1, best_order = 0, best_po = 0, best_precision = 0, best_ci = 0, best_wave = 0;
	int32_t best_shift = 0;
	int32_t best_constant = 0;
	uint32_t best_bits = (P.disable_verbatim && n >= 4) ? 0xffffffffu : hdr + n * sbps;
	uint32_t nan = 0;               // LPC analyses
	bool fixed_valid = false;
	uint32_t fixed_order = 0;

	if(n > 4) {
		// ---- fixed predictor estimate (fixed.c:222 / fixed_intrin_avx2.c:57) ----------------------
		const uint32_t n4 = n - 4;
		const bool fwide = !(sbps + ilog2_u32(n4 * 17) < 32);
		uint64_t e0 = 0, e1 = 0, e2 = 0, e3 = 0, e4 = 0;
		if(!fwide || (n4 & 3) == 0) {
			// plain sums over samples 4..n-1 (the AVX2 routine's four lanes tile the range exactly when
			// (n-4) % 4 == 0): per-thread sliding window, 32-bit differences (|d4| < 2^29 for bps <= 25)
			for(uint32_t base = CHUNK * (uint32_t)tid; base < n; base += CHUNK * TPB) {
				int32_t x[CHUNK + 4];
#pragma unroll
				for(int k = 0; k < CHUNK + 4; k++) x[k] = sig[sigidx((int)base - 4 + k)];
				uint32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0;   // 16 terms < 2^29 each: no overflow
#pragma unroll
				for(int t = 0; t < CHUNK; t++) {
					const uint32_t i = base + t;
					if(i >= 4 && i < n) {
						const int32_t a0 = x[t + 4], a1 = x[t + 3], a2 = x[t + 2], a3 = x[t + 1], a4 = x[t];
						const int32_t d1 = a0 - a1, d2 = a0 - 2 * a1 + a2, d3 = a0 - 3 * a1 + 3 * a2 - a3, d4 = a0 - 4 * a1 + 6 * a2 - 4 * a3 + a4;
						s0 += (uint32_t)abs(a0); s1 += (uint32_t)abs(d1); s2 += (uint32_t)abs(d2); s3 += (uint32_t)abs(d3); s4 += (uint32_t)abs(d4);
					}
				}
				e0 += s0; e1 += s1; e2 += s2; e3 += s3; e4 += s4;
			}
		}
		else {
			// fixed_intrin_avx2.c:57 with (n-4) % 4 != 0 (short last blocks): lane l takes its history from l*q but
			// its data from (l*(n-4))/4 and the remainder is dropped -- restated literally
			const uint32_t q = n4 / 4;
			for(uint32_t l = 0; l < 4; l++) {
				const int hist = (int)(l * q), start = (int)(((uint64_t)l * n4) / 4);
				for(uint32_t i = (uint32_t)tid; i < q; i += TPB) {
					int64_t v[5];
#pragma unroll
					for(int j = 0; j < 5; j++) { const int m = (int)i - j; v[j] = sig[sigidx(4 + (m >= 0 ? start + m : hist + m))]; }
					const int64_t d0 = v[0], d1 = v[0] - v[1], d2 = v[0] - 2 * v[1] + v[2], d3 = v[0] - 3 * v[1] + 3 * v[2] - v[3],
					              d4 = v[0] - 4 * v[1] + 6 * v[2] - 4 * v[3] + v[4];
					e0 += (uint64_t)(d0 < 0 ? -d0 : d0); e1 += (uint64_t)(d1 < 0 ? -d1 : d1); e2 += (uint64_t)(d2 < 0 ? -d2 : d2);
					e3 += (uint64_t)(d3 < 0 ? -d3 : d3); e4 += (uint64_t)(d4 < 0 ? -d4 : d4);
				}
			}
		}
		{
			// one workgroup reduction for the five totals
			e0 = wave_reduce_add_u64(e0); e1 = wave_reduce_add_u64(e1); e2 = wave_reduce_add_u64(e2);
			e3 = wave_reduce_add_u64(e3); e4 = wave_reduce_add_u64(e4);
			uint64_t *red = wsums_all;      // free at this point
			__syncthreads();
			if(lane == 0) { red[wave * 5 + 0] = e0; red[wave * 5 + 1] = e1; red[wave * 5 + 2] = e2; red[wave * 5 + 3] = e3; red[wave * 5 + 4] = e4; }
			__syncthreads();
			e0 = e1 = e2 = e3 = e4 = 0;
			for(int w = 0; w < TPB / 64; w++) { e0 += red[w * 5 + 0]; e1 += red[w * 5 + 1]; e2 += red[w * 5 + 2]; e3 += red[w * 5 + 3]; e4 += red[w * 5 + 4]; }
			__syncthreads();
		}
		uint32_t guess_fixed;
		{
			const uint64_t m34 = e3 < e4 ? e3 : e4, m234 = e2 < m34 ? e2 : m34, m1234 = e1 < m234 ? e1 : m234;
			if(e0 <= m1234) guess_fixed = 0;
			else if(e1 <= m234) guess_fixed = 1;
			else if(e2 <= m34) guess_fixed = 2;
			else if(e3 <= e4) guess_fixed = 3;
			else guess_fixed = 4;
		}
		const uint64_t eg = guess_fixed == 0 ? e0 : guess_fixed == 1 ? e1 : guess_fixed == 2 ? e2 : guess_fixed == 3 ? e3 : e4;
		// rbps = (float)(log(M_LN2*err/n)/M_LN2) as compiled (fixed.c:284-288)
		const float rbps_guess = eg ? (float)(log(((double)eg * 0.69314718055994530942) / (double)n4) * 1.4426950408889634) : 0.0f;
		const bool rbps1_zero = e1 == 0 || (float)(log(((double)e1 * 0.69314718055994530942) / (double)n4) * 1.4426950408889634) == 0.0f;

		STAMP(2);
		bool is_constant = false;
		if(!disable_constant && rbps1_zero) {
			uint32_t diff = 0;
			const int32_t first = sig[sigidx(0)];
			for(uint32_t i = (uint32_t)tid; i < n; i += TPB) diff |= (uint32_t)(sig[sigidx((int)i)] ^ first);
			diff = block_reduce_or_u32(diff, sh->scratch, tid);
			is_constant = diff == 0;
		}
		if(is_constant) {
			const uint32_t bits = hdr + sbps;
			if(bits < best_bits) { best_type = 0; best_constant = sig[sigidx(0)]; best_bits = bits; }
		}
		else {
			// ---- FIXED candidate (stream_encoder.c:4153-4196, 4489): the same FIR with binomial taps ----
			if(!P.disable_fixed || (P.max_lpc_order == 0 && best_bits == 0xffffffffu)) {
				fixed_order = guess_fixed;   // n > 4 so order <= n-1
				fixed_valid = !(rbps_guess >= (float)sbps);
				if(fixed_valid && tid < MAX_ORDER) {
					const uint32_t order = fixed_order;
					int32_t c = 0;
					if(order == 1) c = tid == 0 ? 1 : 0;
					else if(order == 2) c = tid == 0 ? 2 : tid == 1 ? -1 : 0;
					else if(order == 3) c = tid == 0 ? 3 : tid == 1 ? -3 : tid == 2 ? 1 : 0;
					else if(order == 4) c = tid == 0 ? 4 : tid == 1 ? -6 : tid == 2 ? 4 : tid == 3 ? -1 : 0;
					cands[0].q[tid] = c;
				}
			}
			// ---- LPC analyses (stream_encoder.c:4199-4275; apply_apodization_ :4318-4392) --------------
			// The reference walks its apodization state machine sequentially; every step is a pure
			// function of the block, so here all window jobs are windowed and autocorrelated at once, all
			// analyses are modelled at once (one lane each) and all candidates are evaluated concurrently
			// (one wavefront each); the winner is the first minimum in the reference's order (its strict '<').
			if(P.max_lpc_order > 0) {
				const uint32_t max_lpc = P.max_lpc_order >= n ? n - 1 : P.max_lpc_order;
				const uint32_t variant = n <= 32 ? 0u : P.autoc_variant;
				if(max_lpc > 0) {
					const uint32_t lag = max_lpc + 1;
					const uint32_t njobs = jt->njobs;
					nan = jt->nanalyses;
					// ---- windowing: out[i] = (float)x[i] * w[i] (lpc.c:68-94), all jobs, loads batched --------
					for(uint32_t jb = 0; jb < njobs; jb++) {
						const WindowJob jbv = jt->jobs[jb];
						const float *w = win + (size_t)jbv.apod * n;
						float *o = wnd + jbv.off;
						if(jbv.full) {
							uint32_t i = (uint32_t)tid;
							for(; i + 3 * TPB < n; i += 4 * TPB) {
								const float w0 = w[i], w1 = w[i + TPB], w2 = w[i + 2 * TPB], w3 = w[i + 3 * TPB];
								o[i] = (float)sig[sigidx((int)i)] * w0;
								o[i + TPB] = (float)sig[sigidx((int)(i + TPB))] * w1;
								o[i + 2 * TPB] = (float)sig[sigidx((int)(i + 2 * TPB))] * w2;
								o[i + 3 * TPB] = (float)sig[sigidx((int)(i + 3 * TPB))] * w3;
							}
							for(; i < n; i += TPB) o[i] = (float)sig[sigidx((int)i)] * w[i];
						}
						else {
							// first `part` taps ramp up with w[0..part), the next `part` ramp down with w[n-part..n), then one 0
							const uint32_t part = jbv.part, dshift = jbv.dshift, i0 = jbv.i0;
							const uint32_t total = i0 + part + 1;
							uint32_t i = (uint32_t)tid;
							for(; i + TPB < total; i += 2 * TPB) {
								const uint32_t i2 = i + TPB;
								const bool hiA = i >= i0 && i < i0 + part, hiB = i2 >= i0 && i2 < i0 + part;
								const bool loA = !hiA && i < part, loB = !hiB && i2 < part;
								const float wa = hiA ? w[n - part + (i - i0)] : loA ? w[i] : 0.0f;
								const float wb = hiB ? w[n - part + (i2 - i0)] : loB ? w[i2] : 0.0f;
								if(hiA || loA) o[i] = (float)sig[sigidx((int)(dshift + i))] * wa; else if(i == i0 + part && i < n) o[i] = 0.0f;
								if(hiB || loB) o[i2] = (float)sig[sigidx((int)(dshift + i2))] * wb; else if(i2 == i0 + part && i2 < n) o[i2] = 0.0f;
							}
							for(; i < total; i += TPB) {
								const bool hi = i >= i0 && i < i0 + part, lo = !hi && i < part;
								if(hi) o[i] = (float)sig[sigidx((int)(dshift + i))] * w[n - part + (i - i0)];
								else if(lo) o[i] = (float)sig[sigidx((int)(dshift + i))] * w[i];
								else if(i == i0 + part && i < n) o[i] = 0.0f;
							}
						}
					}
					__syncthreads();
					STAMP(3);
					// ---- autocorrelation: one wavefront per job (64 chains = 16 lags x 4 vector lanes) ------------
					if(variant == 0) {
						for(uint32_t t = (uint32_t)tid; t < njobs * lag; t += TPB) {
							const uint32_t jb = t / lag, j = t - jb * lag;
							autoc_job[jb * MAX_ORDER + j] = autoc_small(wnd + jt->jobs[jb].off, jt->jobs[jb].nd, j);
						}
					}
					else {
						const uint32_t j = (uint32_t)lane >> 2, l = (uint32_t)lane & 3;
						const uint32_t mine = jt->wave_njobs[wave];
						for(uint32_t q = 0; q < mine; q++) {
							const uint32_t jb = jt->wave_jobs[wave][q];
							const float *d = wnd + jt->jobs[jb].off;
							const uint32_t nd = jt->jobs[jb].nd;
							if(j < lag) accs[(jb * lag + j) * 4 + l] = variant == 12 ? autoc_chain_12(d, nd, j, l) : autoc_chain_8_16(d, nd, variant, j, l);
						}
						__syncthreads();
						for(uint32_t t = (uint32_t)tid; t < njobs * lag; t += TPB) {
							const uint32_t jb = t / lag, jj = t - jb * lag;
							autoc_job[jb * MAX_ORDER + jj] = autoc_finish(wnd + jt->jobs[jb].off, jt->jobs[jb].nd, variant, jj, &accs[(jb * lag + jj) * 4]);
						}
					}
					__syncthreads();
					STAMP(4);
					// ---- one lane per analysis: Levinson-Durbin, order guess, quantisation -----------------
					// (the window buffer is dead now: it holds each lane's lp_coeff rows)
					if((uint32_t)tid < nan) {
						const uint32_t jb = jt->an_job[tid], rt = jt->an_root[tid];
						const bool punch = jt->an_punch[tid] != 0;
						double av[MAXORD + 1];
#pragma unroll
						for(int j = 0; j <= MAXORD; j++) {
							double v = 0.0;
							if((uint32_t)j < lag) {
								v = autoc_job[jb * MAX_ORDER + j];
								// punch-out: root - partial for lags < max_order only; lag max_order keeps the
								// partial's value (stream_encoder.c:4339-4340,4370-4371)
								if(punch && (uint32_t)j < max_lpc) v = autoc_job[rt * MAX_ORDER + j] - v;
							}
							av[j] = v;
						}
						float *rows = (nan * MAXORD * MAXORD * 4u <= P.wnd_bytes) ? wnd + (size_t)tid * MAXORD * MAXORD : nullptr;
						sh->cand_valid[1 + tid] = lpc_model<MAXORD>(av, max_lpc, n, sbps, P.precision, rows, &cands[1 + tid]);
					}
				}
			}
		}
	}
	if(tid == 0) {
		sh->cand_valid[0] = fixed_valid ? 1 : 0;
		if(fixed_valid) { cands[0].order = fixed_order; cands[0].precision = 0; cands[0].shift = 0; cands[0].wide = 0; }
	}
	__syncthreads();
	STAMP(5);

	// ---- candidates: one wavefront each, no workgroup barriers -------------------------------------
	{
		const uint32_t kstride = 1u << P.max_po;
		uint64_t *wsums = wsums_all + (size_t)wave * (2u << P.max_po);
		uint8_t *kbw = kbestw_all + (size_t)wave * 2 * kstride, *ktmp = kbw + kstride;
		uint8_t *kcw = kcandw_all + (size_t)wave * (2u << P.max_po);
		uint32_t wb_bits = 0xffffffffu, wb_ci = 0xffffffffu, wb_po = 0;
		for(uint32_t ci = (uint32_t)wave; ci <= nan; ci += TPB / 64) {
			if(!sh->cand_valid[ci]) continue;
			const Candidate *cd = &cands[ci];
			const uint32_t order = cd->order;
			uint32_t po;
			const uint32_t rbits = eval_candidate_wave<MAXORD>(wsums, kcw, sh->pob[wave], ktmp, sh->divtab, sig, n, order, cd->q, cd->shift,
			                                                   cd->wide != 0, sbps, P, frame_max_po, frame_min_po, &po, lane);
			const uint32_t est = ci == 0 ? sat_add_u32(hdr + order * sbps, rbits)
			                             : sat_add_u32(hdr + 4 + 5 + order * (cd->precision + sbps), rbits);
			if(est > 0 && est < wb_bits) {      // strict: the earlier candidate keeps a tie (stream_encoder.c:4191,4266)
				wb_bits = est; wb_ci = ci; wb_po = po;
				for(uint32_t p = (uint32_t)lane; p < (1u << po); p += 64) kbw[p] = ktmp[p];
				__builtin_amdgcn_wave_barrier();
			}
		}
		if(lane == 0) { sh->wbest_bits[wave] = wb_bits; sh->wbest_ci[wave] = wb_ci; sh->wbest_po[wave] = wb_po; }
	}
	__syncthreads();
	STAMP(6);
	// ---- winner: first minimum in the reference's evaluation order ---------------------------------------
	{
		uint32_t cb = 0xffffffffu, cci = 0xffffffffu, cw = 0;
		for(uint32_t w = 0; w < TPB / 64; w++) {
			const uint32_t b = sh->wbest_bits[w], ci = sh->wbest_ci[w];
			if(ci != 0xffffffffu && (b < cb || (b == cb && ci < cci))) { cb = b; cci = ci; cw = w; }
		}
		if(cci != 0xffffffffu && cb < best_bits) {
			best_bits = cb; best_ci = cci; best_wave = cw; best_po = sh->wbest_po[cw];
			best_type = cci == 0 ? 2 : 3;
			best_order = cands[cci].order; best_precision = cands[cci].precision; best_shift = cands[cci].shift;
		}
	}
	if(best_bits == 0xffffffffu) { best_type = 1; best_bits = hdr + n * sbps; }   // stream_encoder.c:4281

	// ---- decision record ---------------------------------------------------------------------------
	uint32_t rice2 = 0;
	if(best_type >= 2) {
		const uint8_t *kb = kbestw_all + (size_t)best_wave * 2 * (1u << P.max_po);
		uint32_t big = 0;
		for(uint32_t p = (uint32_t)tid; p < (1u << best_po); p += TPB) {
			const uint8_t k = kb[p];
			dec->params[p] = k;
			if(k >= 15) big = 1;
		}
		rice2 = block_reduce_or_u32(big, sh->scratch, tid);   // stream_encoder.c:4786-4791
	}
	if(tid < MAX_ORDER) dec->q[tid] = best_type == 3 ? cands[best_ci].q[tid] : 0;
	if(tid == 0) {
		dec->bits = best_bits;
		dec->type = (uint8_t)best_type; dec->order = (uint8_t)best_order; dec->wasted = (uint8_t)wasted;
		dec->po = (uint8_t)best_po; dec->rice2 = (uint8_t)rice2; dec->precision = (uint8_t)best_precision;
		dec->shift = (int8_t)best_shift; dec->which = (uint8_t)which;
		dec->constant = best_constant;
	}
	STAMP(7);
#undef STAMP
}

// ---------------------------------------------------------------------------------------------
// pack_kernel
// ---------------------------------------------------------------------------------------------
// Frame image in LDS as 32-bit words holding the stream MSB first (word = big-endian view).
__device__ __forceinline__ void put_bits(uint32_t *buf, uint32_t cap_words, uint32_t pos, uint32_t v, uint32_t len)
{
	if(len == 0) return;
	const uint32_t w = pos >> 5, o = pos & 31;
	const uint64_t val = (uint64_t)(len == 32 ? v : (v & ((1u << len) - 1u))) << (64 - len - o);
	const uint32_t hi = (uint32_t)(val >> 32), lo = (uint32_t)val;
	if(hi && w < cap_words) atomicOr(&buf[w], hi);
	if(lo && w + 1 < cap_words) atomicOr(&buf[w + 1], lo);
}

__device__ __forceinline__ uint32_t crc16_step_byte(uint32_t c, uint32_t byte)
{
	c ^= byte << 8;
#pragma unroll
	for(int b = 0; b < 8; b++) c = (c & 0x8000u) ? ((c << 1) ^ 0x8005u) & 0xffffu : (c << 1) & 0xffffu;
	return c;
}
// multiply two GF(2) polynomials of degree < 16 modulo x^16+x^15+x^2+1
__device__ __forceinline__ uint32_t gf16_mul(uint32_t a, uint32_t b)
{
	uint32_t r = 0;
#pragma unroll
	for(int i = 0; i < 16; i++) {
		if(b & 0x8000u) r ^= a;       // process b from its top bit: r = r*x + (bit? a : 0) done below
		b <<= 1;
		if(i != 15) r = (r & 0x8000u) ? ((r << 1) ^ 0x8005u) & 0xffffu : (r << 1) & 0xffffu;
	}
	return r;
}
// x^(8*nbytes) mod P
__device__ uint32_t gf16_xpow8(uint32_t nbytes)
{
	uint32_t result = 1, base = 0x0100u;    // x^8
	while(nbytes) {
		if(nbytes & 1) result = gf16_mul(result, base);
		base = gf16_mul(base, base);
		nbytes >>= 1;
	}
	return result;
}

struct PackShared {
	uint64_t scratch[8];
	uint8_t params[1u << MAX_PO];
	uint32_t scan[TPB / 64 + 1];
	uint32_t ca, left, right;
	uint32_t crc_parts[TPB];
	uint32_t bitpos;
	uint32_t overflow;
};

template <int MAXORD>
__global__ __launch_bounds__(TPB) void pack_kernel(const DevParams P, const int32_t *__restrict__ pcm,
                                                   uint32_t nframes, uint32_t tail_n, uint64_t first_frame_number,
                                                   const SubDecision *__restrict__ decisions,
                                                   uint8_t *__restrict__ slots, uint32_t *__restrict__ frame_bytes,
                                                   FrameInfo *__restrict__ info)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int tid = (int)threadIdx.x;
	const uint32_t C = P.channels, N = P.blocksize;
	const uint32_t f = blockIdx.x;
	const bool is_tail = tail_n != 0 && f == nframes - 1;
	const uint32_t n = is_tail ? tail_n : N;
	const int32_t *frame_pcm = pcm + (size_t)f * N * C;
	const SubDecision *dec = decisions + (size_t)f * P.ncand;

	int32_t *sig = (int32_t *)smem;
	uint32_t *img = (uint32_t *)(smem + P.sig_bytes);
	PackShared *sh = (PackShared *)(smem + P.sig_bytes + P.slot_bytes);
	const uint32_t cap_words = P.slot_bytes / 4;

	for(uint32_t w = (uint32_t)tid; w < cap_words; w += TPB) img[w] = 0;
	if(tid == 0) {
		// channel assignment (stream_encoder.c:3944-3972)
		uint32_t ca = 0, left = 0, right = 1;
		if(P.ms_mode == 1) {
			const uint32_t b0 = dec[0].bits + dec[1].bits, b1 = dec[0].bits + dec[3].bits,
			               b2 = dec[1].bits + dec[3].bits, b3 = dec[2].bits + dec[3].bits;
			uint32_t mn = b0;
			if(b1 < mn) { mn = b1; ca = 1; }
			if(b2 < mn) { mn = b2; ca = 2; }
			if(b3 < mn) { mn = b3; ca = 3; }
			left = ca == 2 ? 3 : ca == 3 ? 2 : 0;
			right = ca == 0 ? 1 : ca == 2 ? 1 : 3;
		}
		else if(P.ms_mode == 2) {
			ca = dec[0].which >= 2 ? 3 : 0;
		}
		sh->ca = ca; sh->left = left; sh->right = right; sh->overflow = 0;
	}
	__syncthreads();
	const uint32_t ca = sh->ca;

	// ---- frame header (stream_encoder_framing.c:245-391), one lane -------------------------------
	if(tid == 0) {
		uint8_t hb[16];
		uint32_t nb = 0;
		uint32_t bs_code, bs_hint = 0, sr_code, sr_hint = 0;
		switch(n) {
			case 192: bs_code = 1; break; case 576: bs_code = 2; break; case 1152: bs_code = 3; break;
			case 2304: bs_code = 4; break; case 4608: bs_code = 5; break; case 256: bs_code = 8; break;
			case 512: bs_code = 9; break; case 1024: bs_code = 10; break; case 2048: bs_code = 11; break;
			case 4096: bs_code = 12; break; case 8192: bs_code = 13; break; case 16384: bs_code = 14; break;
			case 32768: bs_code = 15; break;
			default: bs_hint = bs_code = (n <= 0x100) ? 6 : 7; break;
		}
		const uint32_t sr = P.sample_rate;
		switch(sr) {
			case 88200: sr_code = 1; break; case 176400: sr_code = 2; break; case 192000: sr_code = 3; break;
			case 8000: sr_code = 4; break; case 16000: sr_code = 5; break; case 22050: sr_code = 6; break;
			case 24000: sr_code = 7; break; case 32000: sr_code = 8; break; case 44100: sr_code = 9; break;
			case 48000: sr_code = 10; break; case 96000: sr_code = 11; break;
			default:
				if(sr <= 255000 && sr % 1000 == 0) sr_hint = sr_code = 12;
				else if(sr <= 655350 && sr % 10 == 0) sr_hint = sr_code = 14;
				else if(sr <= 0xffff) sr_hint = sr_code = 13;
				else sr_code = 0;
				break;
		}
		uint32_t bps_code;
		switch(P.bps) {
			case 8: bps_code = 1; break; case 12: bps_code = 2; break; case 16: bps_code = 4; break;
			case 20: bps_code = 5; break; case 24: bps_code = 6; break; case 32: bps_code = 7; break;
			default: bps_code = 0; break;
		}
		hb[nb++] = 0xff; hb[nb++] = 0xf8;
		hb[nb++] = (uint8_t)((bs_code << 4) | sr_code);
		hb[nb++] = (uint8_t)(((ca == 0 ? C - 1 : 7 + ca) << 4) | (bps_code << 1));
		{
			const uint32_t v = (uint32_t)(first_frame_number + f);   // bitwriter.c:832
			if(v < 0x80) hb[nb++] = (uint8_t)v;
			else if(v < 0x800) { hb[nb++] = (uint8_t)(0xC0 | (v >> 6)); hb[nb++] = (uint8_t)(0x80 | (v & 0x3F)); }
			else if(v < 0x10000) { hb[nb++] = (uint8_t)(0xE0 | (v >> 12)); hb[nb++] = (uint8_t)(0x80 | ((v >> 6) & 0x3F)); hb[nb++] = (uint8_t)(0x80 | (v & 0x3F)); }
			else if(v < 0x200000) { hb[nb++] = (uint8_t)(0xF0 | (v >> 18)); hb[nb++] = (uint8_t)(0x80 | ((v >> 12) & 0x3F)); hb[nb++] = (uint8_t)(0x80 | ((v >> 6) & 0x3F)); hb[nb++] = (uint8_t)(0x80 | (v & 0x3F)); }
			else if(v < 0x4000000) { hb[nb++] = (uint8_t)(0xF8 | (v >> 24)); hb[nb++] = (uint8_t)(0x80 | ((v >> 18) & 0x3F)); hb[nb++] = (uint8_t)(0x80 | ((v >> 12) & 0x3F)); hb[nb++] = (uint8_t)(0x80 | ((v >> 6) & 0x3F)); hb[nb++] = (uint8_t)(0x80 | (v & 0x3F)); }
			else { hb[nb++] = (uint8_t)(0xFC | (v >> 30)); hb[nb++] = (uint8_t)(0x80 | ((v >> 24) & 0x3F)); hb[nb++] = (uint8_t)(0x80 | ((v >> 18) & 0x3F)); hb[nb++] = (uint8_t)(0x80 | ((v >> 12) & 0x3F)); hb[nb++] = (uint8_t)(0x80 | ((v >> 6) & 0x3F)); hb[nb++] = (uint8_t)(0x80 | (v & 0x3F)); }
		}
		if(bs_hint == 6) hb[nb++] = (uint8_t)(n - 1);
		else if(bs_hint == 7) { hb[nb++] = (uint8_t)((n - 1) >> 8); hb[nb++] = (uint8_t)(n - 1); }
		if(sr_hint == 12) hb[nb++] = (uint8_t)(sr / 1000);
		else if(sr_hint == 13) { hb[nb++] = (uint8_t)(sr >> 8); hb[nb++] = (uint8_t)sr; }
		else if(sr_hint == 14) { hb[nb++] = (uint8_t)((sr / 10) >> 8); hb[nb++] = (uint8_t)(sr / 10); }
		uint32_t crc = 0;
		for(uint32_t k = 0; k < nb; k++) {
			crc ^= hb[k];
			for(int b = 0; b < 8; b++) crc = (crc & 0x80u) ? ((crc << 1) ^ 0x07u) & 0xffu : (crc << 1) & 0xffu;
		}
		hb[nb++] = (uint8_t)crc;
		for(uint32_t k = 0; k < nb; k++) put_bits(img, cap_words, 8 * k, hb[k], 8);
		sh->bitpos = 8 * nb;
	}
	__syncthreads();

	// ---- subframes (stream_encoder_framing.c:393-594) -------------------------------------------
	const uint32_t nsub = C;
	for(uint32_t s = 0; s < nsub; s++) {
		uint32_t di;   // decision index
		if(P.ms_mode == 1) di = s == 0 ? sh->left : sh->right;
		else di = s;
		const SubDecision *d = dec + di;
		const uint32_t which = d->which, type = d->type, order = d->order, wasted = d->wasted;
		const uint32_t sbps = P.bps - wasted + (which == C + 1 ? 1 : 0);
		uint32_t pos = sh->bitpos;
		__syncthreads();

		uint32_t orv;
		load_signal(sig, frame_pcm, C, n, which, &orv, tid);
		__syncthreads();
		if(wasted) for(uint32_t i = (uint32_t)tid; i < n; i += TPB) sig[sigidx((int)i)] >>= wasted;
		__syncthreads();

		// subframe header byte (+ unary wasted bits)
		uint32_t type_bits = type == 0 ? 0x00u : type == 1 ? 0x02u : type == 2 ? (0x10u | (order << 1)) : (0x40u | ((order - 1) << 1));
		if(tid == 0) {
			put_bits(img, cap_words, pos, type_bits | (wasted ? 1u : 0u), 8);
			if(wasted) put_bits(img, cap_words, pos + 8 + (wasted - 1), 1, 1);
		}
		pos += 8 + wasted;

		if(type == 0) {
			if(tid == 0) put_bits(img, cap_words, pos, (uint32_t)d->constant, sbps);
			pos += sbps;
		}
		else if(type == 1) {
			for(uint32_t i = (uint32_t)tid; i < n; i += TPB) put_bits(img, cap_words, pos + i * sbps, (uint32_t)sig[sigidx((int)i)], sbps);
			pos += n * sbps;
		}
		else {
			// warm-up, (precision, shift, coefficients), entropy coding header
			if((uint32_t)tid < order) put_bits(img, cap_words, pos + (uint32_t)tid * sbps, (uint32_t)sig[sigidx(tid)], sbps);
			pos += order * sbps;
			const int shift = type == 3 ? d->shift : 0;
			bool wide = false;
			if(type == 3) {
				const uint32_t precision = d->precision;
				if(tid == 0) {
					put_bits(img, cap_words, pos, precision - 1, 4);
					put_bits(img, cap_words, pos + 4, (uint32_t)shift, 5);
				}
				if((uint32_t)tid < order) put_bits(img, cap_words, pos + 9 + (uint32_t)tid * precision, (uint32_t)d->q[tid], precision);
				pos += 9 + order * precision;
				uint32_t abs_sum = 0;
				for(uint32_t i = 0; i < order; i++) abs_sum += (uint32_t)abs(d->q[i]);
				wide = silog2_i64((int64_t)(((uint64_t)1 << (sbps - 1)) * abs_sum)) > 32;
			}
			const uint32_t po = d->po, plen = d->rice2 ? 5u : 4u;
			if(tid == 0) {
				put_bits(img, cap_words, pos, d->rice2 ? 1u : 0u, 2);
				put_bits(img, cap_words, pos + 2, po, 4);
			}
			pos += 6;
			// taps
			int32_t q[MAXORD];
#pragma unroll
			for(int j = 0; j < MAXORD; j++) {
				int32_t c = 0;
				if(type == 3) c = j < MAX_ORDER ? d->q[j] : 0;
				else if(order == 1) c = j == 0 ? 1 : 0;
				else if(order == 2) c = j == 0 ? 2 : j == 1 ? -1 : 0;
				else if(order == 3) c = j == 0 ? 3 : j == 1 ? -3 : j == 2 ? 1 : 0;
				else if(order == 4) c = j == 0 ? 4 : j == 1 ? -6 : j == 2 ? 4 : j == 3 ? -1 : 0;
				q[j] = c;
			}
			const uint32_t psize = n >> po;
			for(uint32_t p = (uint32_t)tid; p < (1u << po); p += TPB) sh->params[p] = d->params[p];
			__syncthreads();
			// residual bits: per-thread chunk lengths -> exclusive scan -> write
			for(uint32_t pass = 0; pass < n; pass += CHUNK * TPB) {
				const uint32_t base = pass + CHUNK * (uint32_t)tid;
				int32_t r[CHUNK];
				uint32_t mybits = 0;
				if(base < n) {
					fir_chunk_dispatch<MAXORD>(sig, (int)base, q, shift, fir_mode(wide, sbps), r);
					uint32_t part = base / psize, next = (part + 1) * psize;
					uint32_t k = sh->params[part];
#pragma unroll
					for(int t = 0; t < CHUNK; t++) {
						const uint32_t i = base + t;
						if(i == next) { part++; next += psize; if(i < n) k = sh->params[part]; }
						if(i < n && i >= order) {
							if(i == (part == 0 ? order : part * psize)) mybits += plen;
							const uint32_t u = ((uint32_t)r[t] << 1) ^ (uint32_t)(r[t] >> 31);
							mybits += (u >> k) + 1 + k;
						}
					}
				}
				// workgroup exclusive scan of mybits
				uint32_t incl = mybits;
#pragma unroll
				for(int off = 1; off < 64; off <<= 1) { uint32_t t = __shfl_up(incl, off); if((tid & 63) >= off) incl += t; }
				__syncthreads();
				if((tid & 63) == 63) sh->scan[tid >> 6] = incl;
				__syncthreads();
				uint32_t wave_off = 0, total = 0;
				for(int w = 0; w < TPB / 64; w++) { if(w < (tid >> 6)) wave_off += sh->scan[w]; total += sh->scan[w]; }
				uint32_t p = pos + wave_off + incl - mybits;
				if(base < n) {
					uint32_t part = base / psize, next = (part + 1) * psize;
					uint32_t k = sh->params[part];
#pragma unroll
					for(int t = 0; t < CHUNK; t++) {
						const uint32_t i = base + t;
						if(i == next) { part++; next += psize; if(i < n) k = sh->params[part]; }
						if(i < n && i >= order) {
							if(i == (part == 0 ? order : part * psize)) { put_bits(img, cap_words, p, k, plen); p += plen; }
							const uint32_t u = ((uint32_t)r[t] << 1) ^ (uint32_t)(r[t] >> 31);
							const uint32_t msbs = u >> k;
							put_bits(img, cap_words, p + msbs, (1u << k) | (u & ((1u << k) - 1u)), k + 1);
							p += msbs + 1 + k;
						}
					}
				}
				pos += total;
				__syncthreads();
			}
		}
		if(tid == 0) {
			sh->bitpos = pos;
			if(info) {
				flacgpu_subframe_info *si = &info[f].sub[s];
				si->type = (uint8_t)type; si->order = (uint8_t)order; si->wasted_bits = (uint8_t)wasted;
				si->partition_order = d->po; si->rice2 = d->rice2; si->precision = d->precision; si->shift = d->shift;
				si->pad = 0; si->bits = d->bits;
			}
		}
		__syncthreads();
	}

	// ---- zero-pad to a byte, CRC-16 over the whole frame, footer (stream_encoder.c:3720-3734) --------
	const uint32_t body_bytes = (sh->bitpos + 7) >> 3;
	const uint32_t total_bytes = body_bytes + 2;
	if(total_bytes > P.slot_bytes) { if(tid == 0) { sh->overflow = 1; } }
	__syncthreads();
	{
		// each thread CRCs a contiguous span, then spans are combined: crc(A||B) = crc(A)*x^(8|B|) + crc(B)
		const uint32_t span = (body_bytes + TPB - 1) / TPB;
		const uint32_t lo = umin32((uint32_t)tid * span, body_bytes), hi = umin32(lo + span, body_bytes);
		uint32_t c = 0;
		for(uint32_t k = lo; k < hi; k++) c = crc16_step_byte(c, (img[k >> 2] >> (24 - 8 * (k & 3))) & 0xffu);
		c = gf16_mul(c, gf16_xpow8(body_bytes - hi));
		// xor-reduce
#pragma unroll
		for(int off = 32; off >= 1; off >>= 1) c ^= __shfl_xor(c, off);
		if((tid & 63) == 0) sh->crc_parts[tid >> 6] = c;
		__syncthreads();
		if(tid == 0) {
			uint32_t crc = 0;
			for(int w = 0; w < TPB / 64; w++) crc ^= sh->crc_parts[w];
			put_bits(img, cap_words, body_bytes * 8, crc, 16);
		}
		__syncthreads();
	}
	// ---- store: image words are big-endian views, slots are byte arrays ---------------------------
	{
		uint32_t *dst = (uint32_t *)(slots + (size_t)f * P.slot_bytes);
		const uint32_t words = (umin32(total_bytes, P.slot_bytes) + 3) >> 2;
		for(uint32_t w = (uint32_t)tid; w < words; w += TPB) dst[w] = __builtin_bswap32(img[w]);
		if(tid == 0) {
			frame_bytes[f] = sh->overflow ? 0xffffffffu : total_bytes;
			if(info) info[f].channel_assignment = (uint8_t)ca;
		}
	}
}

// ---------------------------------------------------------------------------------------------
// scan + compaction
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void scan_kernel(const uint32_t *__restrict__ frame_bytes, uint32_t nframes,
                                                    uint64_t *__restrict__ offsets, uint64_t *__restrict__ total)
{
	__shared__ uint64_t wave_tot[16];
	__shared__ uint64_t carry;
	const int tid = (int)threadIdx.x;
	if(tid == 0) carry = 0;
	__syncthreads();
	for(uint32_t base = 0; base < nframes; base += 1024) {
		const uint32_t i = base + (uint32_t)tid;
		uint64_t v = i < nframes ? (frame_bytes[i] == 0xffffffffu ? 0 : frame_bytes[i]) : 0, incl = v;
#pragma unroll
		for(int off = 1; off < 64; off <<= 1) {
			uint32_t lo = __shfl_up((uint32_t)incl, off), hi = __shfl_up((uint32_t)(incl >> 32), off);
			if((tid & 63) >= off) incl += ((uint64_t)hi << 32) | lo;
		}
		if((tid & 63) == 63) wave_tot[tid >> 6] = incl;
		__syncthreads();
		uint64_t woff = 0, tot = 0;
		for(int w = 0; w < 16; w++) { if(w < (tid >> 6)) woff += wave_tot[w]; tot += wave_tot[w]; }
		if(i < nframes) offsets[i] = carry + woff + incl - v;
		__syncthreads();
		if(tid == 0) carry += tot;
		__syncthreads();
	}
	if(tid == 0) { offsets[nframes] = carry; *total = carry; }
}

__global__ __launch_bounds__(TPB) void compact_kernel(const uint8_t *__restrict__ slots, uint32_t slot_bytes,
                                                      const uint32_t *__restrict__ frame_bytes,
                                                      const uint64_t *__restrict__ offsets,
                                                      uint8_t *__restrict__ out, uint64_t out_cap)
{
	const uint32_t f = blockIdx.x;
	const uint32_t nb = frame_bytes[f];
	if(nb == 0xffffffffu) return;
	const uint64_t off = offsets[f];
	if(off + nb > out_cap) return;
	const uint8_t *src = slots + (size_t)f * slot_bytes;
	uint8_t *dst = out + off;
	// head bytes up to 4-byte alignment of dst, then aligned words assembled from two source words
	const uint32_t mis = (uint32_t)((4 - ((uintptr_t)dst & 3)) & 3);
	const uint32_t head = umin32(mis, nb);
	if(threadIdx.x < head) dst[threadIdx.x] = src[threadIdx.x];
	const uint32_t words = (nb - head) >> 2;
	const uint32_t *sw = (const uint32_t *)src;
	uint32_t *dw = (uint32_t *)(dst + head);
	const uint32_t sh = head * 8;
	for(uint32_t w = threadIdx.x; w < words; w += TPB) {
		uint32_t v;
		if(sh == 0) v = sw[w];
		else v = (sw[w] >> sh) | (sw[w + 1] << (32 - sh));
		dw[w] = v;
	}
	const uint32_t done = head + words * 4;
	if(threadIdx.x < nb - done) dst[done + threadIdx.x] = src[done + threadIdx.x];
}

} // namespace flacgpu

// ---------------------------------------------------------------------------------------------
// launch wrappers (called from flacgpu_api.cpp)
// ---------------------------------------------------------------------------------------------
using namespace flacgpu;

template <int MAXORD>
static hipError_t launch_analyze_t(const DevParams &P, const int32_t *pcm, const float *win, const float *tailwin,
                                   uint32_t nframes, uint32_t tail_n, const JobTable *jtm, const JobTable *jtt, SubDecision *dec, unsigned long long *dbg, size_t lds, hipStream_t s)
{
	static bool attr_set = false;
	if(!attr_set) {
		hipError_t e = hipFuncSetAttribute((const void *)analyze_kernel<MAXORD>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
		if(e != hipSuccess) return e;
		attr_set = true;
	}
	hipLaunchKernelGGL(analyze_kernel<MAXORD>, dim3(nframes * P.ncand), dim3(TPB), lds, s, P, pcm, win, tailwin, nframes, tail_n, jtm, jtt, dec, dbg);
	return hipGetLastError();
}
template <int MAXORD>
static hipError_t launch_pack_t(const DevParams &P, const int32_t *pcm, uint32_t nframes, uint32_t tail_n, uint64_t first,
                                const SubDecision *dec, uint8_t *slots, uint32_t *fb, FrameInfo *info, size_t lds, hipStream_t s)
{
	static bool attr_set = false;
	if(!attr_set) {
		hipError_t e = hipFuncSetAttribute((const void *)pack_kernel<MAXORD>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
		if(e != hipSuccess) return e;
		attr_set = true;
	}
	hipLaunchKernelGGL(pack_kernel<MAXORD>, dim3(nframes), dim3(TPB), lds, s, P, pcm, nframes, tail_n, first, dec, slots, fb, info);
	return hipGetLastError();
}

namespace flacgpu {
size_t analyze_lds_bytes(const DevParams &P) { return analyze_layout(P).total; }
size_t pack_lds_bytes(const DevParams &P) { return (size_t)P.sig_bytes + P.slot_bytes + sizeof(PackShared); }

hipError_t launch_analyze(const DevParams &P, const int32_t *pcm, const float *win, const float *tailwin,
                          uint32_t nframes, uint32_t tail_n, const JobTable *jtm, const JobTable *jtt, SubDecision *dec,
                          unsigned long long *dbg, hipStream_t s)
{
	const size_t lds = analyze_lds_bytes(P);
	const uint32_t m = P.max_lpc_order > 4 ? P.max_lpc_order : 4;
	if(m <= 8) return launch_analyze_t<8>(P, pcm, win, tailwin, nframes, tail_n, jtm, jtt, dec, dbg, lds, s);
	if(m <= 12) return launch_analyze_t<12>(P, pcm, win, tailwin, nframes, tail_n, jtm, jtt, dec, dbg, lds, s);
	return launch_analyze_t<16>(P, pcm, win, tailwin, nframes, tail_n, jtm, jtt, dec, dbg, lds, s);
}
hipError_t launch_pack(const DevParams &P, const int32_t *pcm, uint32_t nframes, uint32_t tail_n, uint64_t first,
                       const SubDecision *dec, uint8_t *slots, uint32_t *fb, FrameInfo *info, hipStream_t s)
{
	const size_t lds = pack_lds_bytes(P);
	const uint32_t m = P.max_lpc_order > 4 ? P.max_lpc_order : 4;
	if(m <= 8) return launch_pack_t<8>(P, pcm, nframes, tail_n, first, dec, slots, fb, info, lds, s);
	if(m <= 12) return launch_pack_t<12>(P, pcm, nframes, tail_n, first, dec, slots, fb, info, lds, s);
	return launch_pack_t<16>(P, pcm, nframes, tail_n, first, dec, slots, fb, info, lds, s);
}
hipError_t launch_scan(const uint32_t *fb, uint32_t nframes, uint64_t *offsets, uint64_t *total, hipStream_t s)
{
	hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, s, fb, nframes, offsets, total);
	return hipGetLastError();
}
hipError_t launch_compact(const uint8_t *slots, uint32_t slot_bytes, const uint32_t *fb, const uint64_t *offsets,
                          uint8_t *out, uint64_t out_cap, uint32_t nframes, hipStream_t s)
{
	hipLaunchKernelGGL(compact_kernel, dim3(nframes), dim3(TPB), 0, s, slots, slot_bytes, fb, offsets, out, out_cap);
	return hipGetLastError();
}
} // namespace flacgpu
