// flac_amd/csrc/flacgpu_devfn.h -- device helper functions shared by the analysis and pack kernels:
// integer helpers, wavefront/workgroup reductions, the fp64 model stage (Levinson-Durbin, order guess,
// quantisation), the autocorrelation chains in the reference's compiled association order, signal staging
// into LDS, the integer FIR and the generic one-wavefront residual candidate evaluation.
// Every function cites the reference code (file:line under src/libFLAC/) whose behaviour it restates.
#ifndef FLACGPU_DEVFN_H
#define FLACGPU_DEVFN_H
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "flacgpu_dev.h"
// log() as the reference binary gets it from glibc 2.35 on an FMA host, restated operation by operation (not the device
// library's log: a last-place difference could flip a comparison of the model search)
#define FLACGPU_LOG_FN __device__ __forceinline__
#define FLACGPU_LOG_TABQ static __device__ const
#include "flacgpu_log.h"

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
__device__ __forceinline__ uint32_t umax32(uint32_t a, uint32_t b) { return a > b ? a : b; }

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
// v + (value of the lane's partner group) at exchange stage M; groups of 2^M lanes hold equal values.
// (s_nop 1: a DPP read needs two wait states after the VALU write of its source, and the compiler does not see
// into an asm statement)
template <int M>
__device__ __forceinline__ uint32_t bfly_add(uint32_t v)
{
	uint32_t d;
	if(M == 0) { asm("s_nop 1\n\tv_add_u32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(v)); return d; }
	if(M == 1) { asm("s_nop 1\n\tv_add_u32_dpp %0, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(v)); return d; }
	if(M == 2) { asm("s_nop 1\n\tv_add_u32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(v)); return d; }     // the other quad of the 8
	if(M == 3) { asm("s_nop 1\n\tv_add_u32_dpp %0, %1, %1 row_mirror row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(v)); return d; }          // the other half of the row
	if(M == 4) return v + (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x401F);                                                // lane ^ 16
	// both halves of the wavefront are uniform by now: two scalar reads
	return (uint32_t)__builtin_amdgcn_readlane((int)v, 0) + (uint32_t)__builtin_amdgcn_readlane((int)v, 32);
}
template <int M>
__device__ __forceinline__ uint32_t bfly_or(uint32_t v)
{
	uint32_t d;
	if(M == 0) { asm("s_nop 1\n\tv_or_b32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(v)); return d; }
	if(M == 1) { asm("s_nop 1\n\tv_or_b32_dpp %0, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(v)); return d; }
	if(M == 2) { asm("s_nop 1\n\tv_or_b32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(v)); return d; }
	if(M == 3) { asm("s_nop 1\n\tv_or_b32_dpp %0, %1, %1 row_mirror row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(v)); return d; }
	if(M == 4) return v | (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x401F);
	return (uint32_t)__builtin_amdgcn_readlane((int)v, 0) | (uint32_t)__builtin_amdgcn_readlane((int)v, 32);
}
// wavefront-wide sum / or, result in every lane: DPP butterflies, no LDS round trips
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
	v = bfly_add<0>(v); v = bfly_add<1>(v); v = bfly_add<2>(v); v = bfly_add<3>(v); v = bfly_add<4>(v); return bfly_add<5>(v);
}
__device__ __forceinline__ uint32_t wave_or_u32(uint32_t v)
{
	v = bfly_or<0>(v); v = bfly_or<1>(v); v = bfly_or<2>(v); v = bfly_or<3>(v); v = bfly_or<4>(v); return bfly_or<5>(v);
}
// 64-bit sum of lane values below 2^50 as two 32-bit butterflies (24-bit low limb: 64 of them cannot overflow)
__device__ __forceinline__ uint64_t wave_sum_u50(uint64_t v)
{
	const uint32_t lo = wave_sum_u32((uint32_t)v & 0xffffffu), hi = wave_sum_u32((uint32_t)(v >> 24));
	return (uint64_t)lo + ((uint64_t)hi << 24);
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
__device__ __forceinline__ uint64_t abs_i64(int64_t v) { return (uint64_t)(v < 0 ? -v : v); }
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
__device__ __forceinline__ double expected_bits_scaled(double lpc_error, double error_scale, const uint64_t *logtab = flacgpu_log_tab)
{
	if(lpc_error > 0.0) {
		// 0.5*log(x)/M_LN2 folded by -freciprocal-math into log(x) * (0.5/ln 2)
		double bps = flacgpu_log_with(error_scale * lpc_error, logtab) * 0.7213475204444817;
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

// evaluate_lpc_subframe_'s front end for one (order, precision): precision clamp (stream_encoder.c:4591-4595),
// FLAC__lpc_quantize_coefficients (lpc.c:220-314), residual-width selector (stream_encoder.c:4601-4617, lpc.c:942-976).
// Returns 0 when no candidate results (quantiser failure).
template <int MAXORD>
__device__ __forceinline__ int quantize_candidate(const float (&coef)[MAXORD], uint32_t order, uint32_t precision, uint32_t sbps, Candidate *out)
{
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
	{
		uint32_t abs_sum = 0;
#pragma unroll
		for(int i = 0; i < MAXORD; i++) abs_sum += (uint32_t)abs(q[i]);
		const uint64_t maxabs = (uint64_t)1 << (sbps - 1);
		const uint64_t before = maxabs * abs_sum;
		const uint64_t after = (uint64_t)(-1 * ((-1 * (int64_t)before) >> shift));
		if(silog2_i64((int64_t)(maxabs + after)) > 32) out->wide = 2;          // the residual may not fit 32 bits: checked while it is computed
		else out->wide = silog2_i64((int64_t)before) > 32 ? 1u : 0u;
	}
#pragma unroll
	for(int i = 0; i < MAXORD; i++) out->q[i] = q[i];                // taps past MAXORD are never read
	out->order = order;
	out->precision = precision;
	out->shift = shift;
	return 1;
}
// the candidates of one LPC order: the configured precision, or with -p every precision 5..max (stream_encoder.c:4230-4243)
template <int MAXORD>
__device__ __forceinline__ void emit_order_candidates(const float (&coef)[MAXORD], uint32_t order, uint32_t sbps, const DevParams &P, Candidate *slots, int *vslots)
{
	if(!P.prec_search) { vslots[0] = quantize_candidate<MAXORD>(coef, order, P.precision, sbps, &slots[0]); return; }
	uint32_t maxp = 15;
	if(sbps <= 17) { maxp = umin32(32 - sbps - ilog2_u32(order), 15); if(maxp < 5) maxp = 5; }
	for(uint32_t prec = 5; prec <= maxp; prec++) vslots[prec - 5] = quantize_candidate<MAXORD>(coef, order, prec, sbps, &slots[prec - 5]);
}

// One LPC analysis (autocorrelation a[], already punched-out when applicable) -> its candidate slots
// [norders][nprec] and their valid flags.  Normally one slot: the guessed order (lpc.c:1608) at the configured
// precision; -e emits every order 1..max as the recursion produces it; -p every precision.
template <int MAXORD>
__device__ void lpc_model(const double (&a)[MAXORD + 1], uint32_t max_order, uint32_t n, uint32_t sbps, const DevParams &P,
                          Candidate *slots, int *vslots, const uint64_t *logtab)
{
	const uint32_t nslots = P.norders * P.nprec;
	for(uint32_t s = 0; s < nslots; s++) vslots[s] = 0;
	if(a[0] == 0.0) return;
	double lpc[MAXORD], errs[MAXORD];
#pragma unroll
	for(int i = 0; i < MAXORD; i++) { lpc[i] = 0.0; errs[i] = 0.0; }
	if(P.exhaustive) {
		// the recursion of levinson<> with the candidates of order i+1 emitted as soon as its coefficients exist
		double err = a[0];
		uint32_t used = max_order;
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
				if(i & 1) lpc[i >> 1] = (r + 1.0) * lpc[i >> 1];
				err *= (1.0 - r * r);
				if(err == 0.0) used = (uint32_t)i + 1;
				const uint32_t order = (uint32_t)i + 1;
				// stream_encoder.c:4227-4229
				if(!(expected_bits_scaled(err, 0.5 / (double)(n - order)) >= (double)sbps)) {
					float coef[MAXORD];
#pragma unroll
					for(int j = 0; j < MAXORD; j++) coef[j] = j <= i ? (float)(-lpc[j]) : 0.0f;
					emit_order_candidates<MAXORD>(coef, order, sbps, P, slots + (size_t)i * P.nprec, vslots + (size_t)i * P.nprec);
				}
			}
		}
		return;
	}
	const uint32_t used = levinson<MAXORD, false>(a, max_order, lpc, errs, nullptr);
	// FLAC__lpc_compute_best_order (lpc.c:1608): total_samples is the full blocksize; with -p the overhead is priced
	// at the smallest precision (stream_encoder.c:4384-4388)
	uint32_t order = 1;
	double err_order = errs[0];
	{
		const double scale = 0.5 / (double)n;
		const uint32_t overhead = sbps + (P.prec_search ? 5u : P.precision);
		double best_bits = 4294967295.0;
#pragma unroll
		for(int idx = 0; idx < MAXORD; idx++) {
			if((uint32_t)idx < used) {
				const uint32_t o = (uint32_t)idx + 1;
				const double bits = expected_bits_scaled(errs[idx], scale, logtab) * (double)(n - o) + (double)(o * overhead);
				if(bits < best_bits) { order = o; best_bits = bits; err_order = errs[idx]; }
			}
		}
	}
	// stream_encoder.c:4227-4229
	if(expected_bits_scaled(err_order, 0.5 / (double)(n - order), logtab) >= (double)sbps) return;
	// coefficients of `order`: rerun the (deterministic) recursion up to that order
	(void)levinson<MAXORD, false>(a, order, lpc, errs, nullptr);
	float coef[MAXORD];
#pragma unroll
	for(int j = 0; j < MAXORD; j++) coef[j] = (uint32_t)j < order ? (float)(-lpc[j]) : 0.0f;
	emit_order_candidates<MAXORD>(coef, order, sbps, P, slots, vslots);
}

// ---------------------------------------------------------------------------------------------
// autocorrelation in the association order of the reference's compiled routines:
// one lane per chain = (window job, lag j, vector lane l); the 4 lane accumulators of a lag are
// combined as (acc3+acc1)+(acc2+acc0) afterwards, then the scalar head/tail of lpc_intrin_fma.c.
// d = windowed data of the job in LDS, nd = its data_len.
// ---------------------------------------------------------------------------------------------
#define DD(k) ((double)d[k])
// the same with the data given as two plain windows: head = d[0,32), tail = d[tail_lo, nd)
__device__ __forceinline__ double autoc_finish2(const float *head, const float *tail, uint32_t tail_lo, uint32_t nd, uint32_t L, uint32_t j, const double *acc4)
{
#define D2(k) ((double)((uint32_t)(k) >= tail_lo ? tail[(uint32_t)(k) - tail_lo] : head[(k)]))
	double a = 0.0;
	for(uint32_t h = j; h < L; h++) a += D2(h) * D2(h - j);
	const uint32_t nb = (nd - L) / 8;
	if(nb) a = ((acc4[3] + acc4[1]) + (acc4[2] + acc4[0])) + a;
	uint32_t i = L + 8 * nb;
	if(nd - i >= 4) {
		const double hi = fma(D2(i + 1), D2(i + 1 - j), D2(i + 3) * D2(i + 3 - j));
		const double lo = fma(D2(i), D2(i - j), D2(i + 2) * D2(i + 2 - j));
		a = (hi + lo) + a;
		i += 4;
	}
	for(; i < nd; i++) a = fma(D2(i), D2(i - j), a);
	return a;
#undef D2
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
	return which == C ? (int32_t)(((int64_t)l + (int64_t)r) >> 1) : (l - r);      // (the 33-bit side of a 32-bit stream: side64())
}
// side channel of a 32-bit stream: 33 bits (stream_encoder.c:3833)
__device__ __forceinline__ int64_t side64(int2 lr) { return (int64_t)lr.x - (int64_t)lr.y; }

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
			int32_t v = which == 0 ? lr.x : which == 1 ? lr.y : which == 2 ? (int32_t)(((int64_t)lr.x + (int64_t)lr.y) >> 1) : (lr.x - lr.y);
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
// MODE 2: 64-bit accumulate (lpc.c:582; with 64-bit samples fixed.c:532)
// MODE 3: 64-bit accumulate, and the residual must lie in (INT32_MIN, INT32_MAX] (lpc.c:832; 64-bit samples lpc.c:886):
//         returns true when a residual of a sample in [lo, hi) does not
// ST: int32_t samples, or int64_t (the 33-bit side channel of a 32-bit stream; modes 2 and 3 only)
// LOAD: sample i of the channel (0 in front of the block and behind it)
template <int MAXORD, int MODE, typename ST, typename LOAD>
__device__ __forceinline__ bool fir_chunk_core(LOAD load, int base, const int32_t *q, int shift, int32_t *r, uint32_t lo, uint32_t hi)
{
	ST x[MAXORD + CHUNK];
	bool bad = false;
#pragma unroll
	for(int k = 0; k < MAXORD + CHUNK; k++) x[k] = (ST)load(base - MAXORD + k);
#pragma unroll
	for(int s = 0; s < CHUNK; s++) {
		if(MODE >= 2) {
			uint64_t sum = 0;                                             // wraps like the reference's int64 does
#pragma unroll
			for(int j = 0; j < MAXORD; j++) sum += (uint64_t)((int64_t)q[j] * (int64_t)x[MAXORD + s - 1 - j]);
			const int64_t v = (int64_t)((uint64_t)(int64_t)x[MAXORD + s] - (uint64_t)((int64_t)sum >> shift));
			r[s] = (int32_t)v;
			if(MODE == 3) { const uint32_t i = (uint32_t)(base + s); bad = bad || (i >= lo && i < hi && (v <= (int64_t)INT32_MIN || v > (int64_t)INT32_MAX)); }
		}
		else {
			uint32_t sum = 0;
#pragma unroll
			for(int j = 0; j < MAXORD; j++)
				sum += MODE == 0 ? (uint32_t)__mul24(q[j], (int32_t)x[MAXORD + s - 1 - j]) : (uint32_t)q[j] * (uint32_t)x[MAXORD + s - 1 - j];
			r[s] = (int32_t)((uint32_t)x[MAXORD + s] - (uint32_t)((int32_t)sum >> shift));
		}
	}
	return bad;
}
template <int MAXORD, int MODE, typename ST>
__device__ __forceinline__ bool fir_chunk(const ST *sig, int base, const int32_t *q, int shift, int32_t *r, uint32_t lo, uint32_t hi)
{
	return fir_chunk_core<MAXORD, MODE, ST>([&](int i) { return sig[sigidx(i)]; }, base, q, shift, r, lo, hi);
}

// Where a general kernel finds the samples of a channel: the padded LDS array (32- or 64-bit samples, sigidx layout), or --
// blocks too long for the LDS -- the planar copy in HBM (ChanPrep::fmt), read through the caches
struct SigRef {
	const void *p;
	uint32_t kind;      // 0: LDS int32, 1: LDS int64, 2: HBM plane
	uint32_t fmt, n;    // kind 2: ChanPrep::fmt and the block length
};
__device__ __forceinline__ int64_t plane_sample(const SigRef S, int i)
{
	if(i < 0 || (uint32_t)i >= S.n) return 0;
	return S.fmt == 1 ? (int64_t)((const int16_t *)S.p)[i] : S.fmt == 2 ? ((const int64_t *)S.p)[i] : (int64_t)((const int32_t *)S.p)[i];
}

// mode: 0..3 as above (wave-uniform).  lo/hi: the samples the overflow check of mode 3 looks at (predictor order ..
// block length).  Returns the mode-3 verdict (false otherwise).
template <int MAXORD>
__device__ __forceinline__ bool fir_chunk_dispatch(const SigRef S, int base, const int32_t *q, int shift, int mode, int32_t *r, uint32_t lo, uint32_t hi)
{
	const void *sig = S.p;
	const bool s64 = S.kind == 1;
	if(S.kind == 2) {
		auto ld = [&](int i) { return plane_sample(S, i); };
		if(mode == 3) return fir_chunk_core<MAXORD, 3, int64_t>(ld, base, q, shift, r, lo, hi);
		if(mode == 2 || S.fmt == 2) return fir_chunk_core<MAXORD, 2, int64_t>(ld, base, q, shift, r, lo, hi);
		return fir_chunk_core<MAXORD, 1, int64_t>(ld, base, q, shift, r, lo, hi);       // (mode 0 and 1 give the same residuals)
	}
	if(s64) {
		if(mode == 3) return fir_chunk<MAXORD, 3, int64_t>((const int64_t *)sig, base, q, shift, r, lo, hi);
		return fir_chunk<MAXORD, 2, int64_t>((const int64_t *)sig, base, q, shift, r, lo, hi);
	}
	if(mode == 0) return fir_chunk<MAXORD, 0, int32_t>((const int32_t *)sig, base, q, shift, r, lo, hi);
	if(mode == 1) return fir_chunk<MAXORD, 1, int32_t>((const int32_t *)sig, base, q, shift, r, lo, hi);
	if(mode == 2) return fir_chunk<MAXORD, 2, int32_t>((const int32_t *)sig, base, q, shift, r, lo, hi);
	return fir_chunk<MAXORD, 3, int32_t>((const int32_t *)sig, base, q, shift, r, lo, hi);
}
// Candidate::wide (0 / 1 / 2) -> FIR mode
__device__ __forceinline__ int fir_mode(uint32_t wide, uint32_t sbps) { return wide == 2 ? 3 : wide ? 2 : (sbps <= 24 ? 0 : 1); }

// ---------------------------------------------------------------------------------------------
// fixed-predictor candidates of a subframe (stream_encoder.c:4153-4190): the guessed order, or with -e every
// order 0..4, each skipped when its estimate rbps[order] = (float)(log(M_LN2*err/n)/M_LN2) (as compiled,
// fixed.c:284-288) is not below the sample width.  Called by all lanes of one wavefront / the first 64 threads.
// e[k]: error sums of the SHIFTED signal.  Returns true when at least one candidate is valid.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float fixed_rbps(uint64_t e, uint32_t n4)
{
	return e ? (float)(flacgpu_log(((double)e * 0.69314718055994530942) / (double)n4) * 1.4426950408889634) : 0.0f;
}
// invalid: bit k set = order k got 34 bits per sample from an overflow-checked estimator (fixed_intrin_avx2.c:172, fixed.c:360)
__device__ __forceinline__ bool emit_fixed_candidates(const DevParams &P, Candidate *c0, int *v0, const uint64_t (&e)[5], uint32_t n4, uint32_t guess,
                                                      bool allowed, uint32_t sbps, int lane, uint32_t invalid = 0)
{
	bool any = false;
	for(uint32_t k = 0; k < P.nfixed; k++) {
		const uint32_t order = P.exhaustive ? k : guess;
		const uint64_t eo = order == 0 ? e[0] : order == 1 ? e[1] : order == 2 ? e[2] : order == 3 ? e[3] : e[4];
		const bool ok = allowed && !((invalid >> order) & 1u) && !(fixed_rbps(eo, n4) >= (float)sbps);
		any = any || ok;
		if(lane < MAX_ORDER) {                                        // ALL taps of the record: the 32-tap flavours of the evaluation and pack kernels read
			int32_t c = 0;                                            // them all (taps 16..31 were left as they lay: zero in fresh memory, anything in reused memory)
			if(order == 1) c = lane == 0 ? 1 : 0;
			else if(order == 2) c = lane == 0 ? 2 : lane == 1 ? -1 : 0;
			else if(order == 3) c = lane == 0 ? 3 : lane == 1 ? -3 : lane == 2 ? 1 : 0;
			else if(order == 4) c = lane == 0 ? 4 : lane == 1 ? -6 : lane == 2 ? 4 : lane == 3 ? -1 : 0;
			c0[k].q[lane] = c;
		}
		// 64-bit differences once the 32-bit ones could wrap (stream_encoder.c:4511-4516)
		if(lane == 0) { c0[k].order = order; c0[k].precision = 0; c0[k].shift = 0; c0[k].wide = sbps + order > 32 ? 1u : 0u; v0[k] = ok ? 1 : 0; }
	}
	return any;
}

// ---------------------------------------------------------------------------------------------
// integer FIR building blocks shared by the evaluation and pack kernels
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t sad_u32(uint32_t a, uint32_t b, uint32_t c)       // |a - b| + c, a and b unsigned
{
	uint32_t d;
	asm("v_sad_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
	return d;
}

// ---- difference sums of a run of samples (the prep kernels, flacgpu_prep.hip, and ff_kernel, flacgpu_kernels.hip) ----
struct Prep2Acc {
	uint32_t orv, diff;
	uint32_t mag;              // OR of x ^ (x >> 31): every sample fits int16 iff (mag >> wasted) < 2^15
	uint64_t e[5];
};


// statistics of the 16 samples x[4..19] of a chunk (x[0..3] = the four samples in front of them).
// Sums are taken on the UNSHIFTED signal: every |difference| is a multiple of 2^wasted, so the sums of the shifted
// signal the reference computes (it shifts in place first) are these sums >> wasted, exactly.
// |d_k[i]| = |d_(k-1)[i] - d_(k-1)[i-1]| is one v_sad_u32 on the sign-flipped (order preserving) operands.
// MAG: also collect Prep2Acc::mag (the side channel: 17 bits wide, but quiet enough for the packed 16-bit kernels most of the time)
// PARTS (the presets without an LPC search, prep2_kernel<.,.,true>): cs[k] = this chunk's sum for order k, as added to A.e[k];
// ex[k] = what the residual of order k has IN FRONT of sample 4 (samples k..3: the predictor estimate skips them, the residual
// of the chosen order does not, stream_encoder.c:4100 vs :4456)
template <bool WIDE, bool MAG = false, bool PARTS = false, int CH = CHUNK, bool FUSE = false>
__device__ __forceinline__ void prep2_chunk(const int32_t (&x)[CH + 4], bool first_chunk, int32_t first, Prep2Acc &A, uint32_t *cs = nullptr, uint32_t *ex = nullptr,
                                            bool maybe_first = true /* wave-uniform: false = no lane of this wavefront holds the block's first chunk */)
{
	constexpr uint32_t M = 0x80000000u;
	uint32_t s[5] = {0, 0, 0, 0, 0};
	// differences at the three samples in front of the chunk
	int32_t d1p = x[3] - x[2], d2p = (x[3] - x[2]) - (x[2] - x[1]), d3p = ((x[3] - x[2]) - (x[2] - x[1])) - ((x[2] - x[1]) - (x[1] - x[0]));
	uint32_t xbp = (uint32_t)x[3] ^ M;
#pragma unroll
	for(int t = 0; t < CH; t++) {
		const int32_t a0 = x[t + 4];
		A.orv |= (uint32_t)a0; A.diff |= (uint32_t)(a0 ^ first);
		if(MAG) A.mag |= (uint32_t)(a0 ^ (a0 >> 31));
		const uint32_t xb = (uint32_t)a0 ^ M;
		const int32_t d1 = a0 - x[t + 3], d2 = d1 - d1p, d3 = d2 - d2p;
		if(FUSE && !WIDE && (t >= 4 || !maybe_first)) {
			// (round 6, FUSE: the running sum is the instruction's own third operand -- |a - b| + c -- instead of a sum formed behind it:
			//  five additions per sample and channel less, a seventh of this loop; the same 32-bit sums.  prep3_kernel: -10 % instructions,
			//  -2.5 % time (it is HBM bound next to that), -5 +1.1 %; ff_kernel, whose lanes run five chains of eighteen dependent
			//  v_sad_u32 that way, got 2.5 % SLOWER and keeps the sums apart: profiles/r06_z_*)
			s[0] = sad_u32(xb, M, s[0]); s[1] = sad_u32(xb, xbp, s[1]); s[2] = sad_u32((uint32_t)d1 ^ M, (uint32_t)d1p ^ M, s[2]);
			s[3] = sad_u32((uint32_t)d2 ^ M, (uint32_t)d2p ^ M, s[3]); s[4] = sad_u32((uint32_t)d3 ^ M, (uint32_t)d3p ^ M, s[4]);
		}
		else {
			uint32_t t0 = sad_u32(xb, M, 0), t1 = sad_u32(xb, xbp, 0), t2 = sad_u32((uint32_t)d1 ^ M, (uint32_t)d1p ^ M, 0),
			         t3 = sad_u32((uint32_t)d2 ^ M, (uint32_t)d2p ^ M, 0), t4 = sad_u32((uint32_t)d3 ^ M, (uint32_t)d3p ^ M, 0);
			if(t < 4) {
				if(first_chunk) {
					if(PARTS) { ex[0] += t0; if(t >= 1) ex[1] += t1; if(t >= 2) ex[2] += t2; if(t >= 3) ex[3] += t3; }
					t0 = t1 = t2 = t3 = t4 = 0;                                        // the sums start at sample 4 (stream_encoder.c:4100)
				}
			}
			if(WIDE) { A.e[0] += t0; A.e[1] += t1; A.e[2] += t2; A.e[3] += t3; A.e[4] += t4; }
			else { s[0] += t0; s[1] += t1; s[2] += t2; s[3] += t3; s[4] += t4; }
		}
		d1p = d1; d2p = d2; d3p = d3; xbp = xb;
	}
	if(!WIDE) {
#pragma unroll
		for(int k = 0; k < 5; k++) A.e[k] += s[k];
		if(PARTS) {
#pragma unroll
			for(int k = 0; k < 5; k++) cs[k] = s[k];
		}
	}
}


// NP dependent v_dot2_i32_i16 and the logical shift as ONE asm statement: between separate asm statements the
// compiler pads every dependent pair with an s_nop, and its own v_dot2c form costs a v_mov per sample.
template <int NP>
__device__ __forceinline__ uint32_t dot2_chain_lshr(const uint32_t (&W)[NP], const uint32_t (&Q)[NP], int32_t sum0, uint32_t shift)
{
	uint32_t d;
	if constexpr(NP == 1) asm("v_dot2_i32_i16 %0, %2, %3, %1\n\tv_lshrrev_b32 %0, %4, %0" : "=&v"(d) : "v"(sum0), "v"(W[0]), "v"(Q[0]), "v"(shift));
	if constexpr(NP == 2) asm("v_dot2_i32_i16 %0, %2, %3, %1\n\tv_dot2_i32_i16 %0, %4, %5, %0\n\tv_lshrrev_b32 %0, %6, %0" : "=&v"(d) : "v"(sum0), "v"(W[0]), "v"(Q[0]), "v"(W[1]), "v"(Q[1]), "v"(shift));
	if constexpr(NP == 3) asm("v_dot2_i32_i16 %0, %2, %3, %1\n\tv_dot2_i32_i16 %0, %4, %5, %0\n\tv_dot2_i32_i16 %0, %6, %7, %0\n\tv_lshrrev_b32 %0, %8, %0" : "=&v"(d) : "v"(sum0), "v"(W[0]), "v"(Q[0]), "v"(W[1]), "v"(Q[1]), "v"(W[2]), "v"(Q[2]), "v"(shift));
	if constexpr(NP == 4) asm("v_dot2_i32_i16 %0, %2, %3, %1\n\tv_dot2_i32_i16 %0, %4, %5, %0\n\tv_dot2_i32_i16 %0, %6, %7, %0\n\tv_dot2_i32_i16 %0, %8, %9, %0\n\tv_lshrrev_b32 %0, %10, %0" : "=&v"(d) : "v"(sum0), "v"(W[0]), "v"(Q[0]), "v"(W[1]), "v"(Q[1]), "v"(W[2]), "v"(Q[2]), "v"(W[3]), "v"(Q[3]), "v"(shift));
	if constexpr(NP == 5) asm("v_dot2_i32_i16 %0, %2, %3, %1\n\tv_dot2_i32_i16 %0, %4, %5, %0\n\tv_dot2_i32_i16 %0, %6, %7, %0\n\tv_dot2_i32_i16 %0, %8, %9, %0\n\tv_dot2_i32_i16 %0, %10, %11, %0\n\tv_lshrrev_b32 %0, %12, %0" : "=&v"(d) : "v"(sum0), "v"(W[0]), "v"(Q[0]), "v"(W[1]), "v"(Q[1]), "v"(W[2]), "v"(Q[2]), "v"(W[3]), "v"(Q[3]), "v"(W[4]), "v"(Q[4]), "v"(shift));
	if constexpr(NP == 6) asm("v_dot2_i32_i16 %0, %2, %3, %1\n\tv_dot2_i32_i16 %0, %4, %5, %0\n\tv_dot2_i32_i16 %0, %6, %7, %0\n\tv_dot2_i32_i16 %0, %8, %9, %0\n\tv_dot2_i32_i16 %0, %10, %11, %0\n\tv_dot2_i32_i16 %0, %12, %13, %0\n\tv_lshrrev_b32 %0, %14, %0" : "=&v"(d) : "v"(sum0), "v"(W[0]), "v"(Q[0]), "v"(W[1]), "v"(Q[1]), "v"(W[2]), "v"(Q[2]), "v"(W[3]), "v"(Q[3]), "v"(W[4]), "v"(Q[4]), "v"(W[5]), "v"(Q[5]), "v"(shift));
	if constexpr(NP == 7) asm("v_dot2_i32_i16 %0, %2, %3, %1\n\tv_dot2_i32_i16 %0, %4, %5, %0\n\tv_dot2_i32_i16 %0, %6, %7, %0\n\tv_dot2_i32_i16 %0, %8, %9, %0\n\tv_dot2_i32_i16 %0, %10, %11, %0\n\tv_dot2_i32_i16 %0, %12, %13, %0\n\tv_dot2_i32_i16 %0, %14, %15, %0\n\tv_lshrrev_b32 %0, %16, %0" : "=&v"(d) : "v"(sum0), "v"(W[0]), "v"(Q[0]), "v"(W[1]), "v"(Q[1]), "v"(W[2]), "v"(Q[2]), "v"(W[3]), "v"(Q[3]), "v"(W[4]), "v"(Q[4]), "v"(W[5]), "v"(Q[5]), "v"(W[6]), "v"(Q[6]), "v"(shift));
	if constexpr(NP == 8) asm("v_dot2_i32_i16 %0, %2, %3, %1\n\tv_dot2_i32_i16 %0, %4, %5, %0\n\tv_dot2_i32_i16 %0, %6, %7, %0\n\tv_dot2_i32_i16 %0, %8, %9, %0\n\tv_dot2_i32_i16 %0, %10, %11, %0\n\tv_dot2_i32_i16 %0, %12, %13, %0\n\tv_dot2_i32_i16 %0, %14, %15, %0\n\tv_dot2_i32_i16 %0, %16, %17, %0\n\tv_lshrrev_b32 %0, %18, %0" : "=&v"(d) : "v"(sum0), "v"(W[0]), "v"(Q[0]), "v"(W[1]), "v"(Q[1]), "v"(W[2]), "v"(Q[2]), "v"(W[3]), "v"(Q[3]), "v"(W[4]), "v"(Q[4]), "v"(W[5]), "v"(Q[5]), "v"(W[6]), "v"(Q[6]), "v"(W[7]), "v"(Q[7]), "v"(shift));
	return d;
}

// the same for the 24-bit multiply-add chain on 32-bit samples: K taps per asm statement (30-operand limit), the
// last one followed by the arithmetic shift and the sign flip that make the prediction an order-preserving unsigned
template <int K, bool LAST>
__device__ __forceinline__ uint32_t mad24_chain(const int32_t *xr /* xr[-j] is the sample tap j reads */, const int32_t *q, uint32_t sum, uint32_t shift)
{
	uint32_t d;
	if constexpr(K == 4 && !LAST) asm("v_mad_i32_i24 %0, %2, %3, %1\n\tv_mad_i32_i24 %0, %4, %5, %0\n\tv_mad_i32_i24 %0, %6, %7, %0\n\tv_mad_i32_i24 %0, %8, %9, %0" : "=&v"(d) : "v"(sum), "v"(q[0]), "v"(xr[-0]), "v"(q[1]), "v"(xr[-1]), "v"(q[2]), "v"(xr[-2]), "v"(q[3]), "v"(xr[-3]));
	if constexpr(K == 4 && LAST) asm("v_mad_i32_i24 %0, %2, %3, %1\n\tv_mad_i32_i24 %0, %4, %5, %0\n\tv_mad_i32_i24 %0, %6, %7, %0\n\tv_mad_i32_i24 %0, %8, %9, %0\n\tv_ashrrev_i32 %0, %10, %0\n\tv_xor_b32 %0, 0x80000000, %0" : "=&v"(d) : "v"(sum), "v"(q[0]), "v"(xr[-0]), "v"(q[1]), "v"(xr[-1]), "v"(q[2]), "v"(xr[-2]), "v"(q[3]), "v"(xr[-3]), "v"(shift));
	if constexpr(K == 8 && !LAST) asm("v_mad_i32_i24 %0, %2, %3, %1\n\tv_mad_i32_i24 %0, %4, %5, %0\n\tv_mad_i32_i24 %0, %6, %7, %0\n\tv_mad_i32_i24 %0, %8, %9, %0\n\tv_mad_i32_i24 %0, %10, %11, %0\n\tv_mad_i32_i24 %0, %12, %13, %0\n\tv_mad_i32_i24 %0, %14, %15, %0\n\tv_mad_i32_i24 %0, %16, %17, %0" : "=&v"(d) : "v"(sum), "v"(q[0]), "v"(xr[-0]), "v"(q[1]), "v"(xr[-1]), "v"(q[2]), "v"(xr[-2]), "v"(q[3]), "v"(xr[-3]), "v"(q[4]), "v"(xr[-4]), "v"(q[5]), "v"(xr[-5]), "v"(q[6]), "v"(xr[-6]), "v"(q[7]), "v"(xr[-7]));
	if constexpr(K == 8 && LAST) asm("v_mad_i32_i24 %0, %2, %3, %1\n\tv_mad_i32_i24 %0, %4, %5, %0\n\tv_mad_i32_i24 %0, %6, %7, %0\n\tv_mad_i32_i24 %0, %8, %9, %0\n\tv_mad_i32_i24 %0, %10, %11, %0\n\tv_mad_i32_i24 %0, %12, %13, %0\n\tv_mad_i32_i24 %0, %14, %15, %0\n\tv_mad_i32_i24 %0, %16, %17, %0\n\tv_ashrrev_i32 %0, %18, %0\n\tv_xor_b32 %0, 0x80000000, %0" : "=&v"(d) : "v"(sum), "v"(q[0]), "v"(xr[-0]), "v"(q[1]), "v"(xr[-1]), "v"(q[2]), "v"(xr[-2]), "v"(q[3]), "v"(xr[-3]), "v"(q[4]), "v"(xr[-4]), "v"(q[5]), "v"(xr[-5]), "v"(q[6]), "v"(xr[-6]), "v"(q[7]), "v"(xr[-7]), "v"(shift));
	if constexpr(K == 12 && !LAST) asm("v_mad_i32_i24 %0, %2, %3, %1\n\tv_mad_i32_i24 %0, %4, %5, %0\n\tv_mad_i32_i24 %0, %6, %7, %0\n\tv_mad_i32_i24 %0, %8, %9, %0\n\tv_mad_i32_i24 %0, %10, %11, %0\n\tv_mad_i32_i24 %0, %12, %13, %0\n\tv_mad_i32_i24 %0, %14, %15, %0\n\tv_mad_i32_i24 %0, %16, %17, %0\n\tv_mad_i32_i24 %0, %18, %19, %0\n\tv_mad_i32_i24 %0, %20, %21, %0\n\tv_mad_i32_i24 %0, %22, %23, %0\n\tv_mad_i32_i24 %0, %24, %25, %0" : "=&v"(d) : "v"(sum), "v"(q[0]), "v"(xr[-0]), "v"(q[1]), "v"(xr[-1]), "v"(q[2]), "v"(xr[-2]), "v"(q[3]), "v"(xr[-3]), "v"(q[4]), "v"(xr[-4]), "v"(q[5]), "v"(xr[-5]), "v"(q[6]), "v"(xr[-6]), "v"(q[7]), "v"(xr[-7]), "v"(q[8]), "v"(xr[-8]), "v"(q[9]), "v"(xr[-9]), "v"(q[10]), "v"(xr[-10]), "v"(q[11]), "v"(xr[-11]));
	if constexpr(K == 12 && LAST) asm("v_mad_i32_i24 %0, %2, %3, %1\n\tv_mad_i32_i24 %0, %4, %5, %0\n\tv_mad_i32_i24 %0, %6, %7, %0\n\tv_mad_i32_i24 %0, %8, %9, %0\n\tv_mad_i32_i24 %0, %10, %11, %0\n\tv_mad_i32_i24 %0, %12, %13, %0\n\tv_mad_i32_i24 %0, %14, %15, %0\n\tv_mad_i32_i24 %0, %16, %17, %0\n\tv_mad_i32_i24 %0, %18, %19, %0\n\tv_mad_i32_i24 %0, %20, %21, %0\n\tv_mad_i32_i24 %0, %22, %23, %0\n\tv_mad_i32_i24 %0, %24, %25, %0\n\tv_ashrrev_i32 %0, %26, %0\n\tv_xor_b32 %0, 0x80000000, %0" : "=&v"(d) : "v"(sum), "v"(q[0]), "v"(xr[-0]), "v"(q[1]), "v"(xr[-1]), "v"(q[2]), "v"(xr[-2]), "v"(q[3]), "v"(xr[-3]), "v"(q[4]), "v"(xr[-4]), "v"(q[5]), "v"(xr[-5]), "v"(q[6]), "v"(xr[-6]), "v"(q[7]), "v"(xr[-7]), "v"(q[8]), "v"(xr[-8]), "v"(q[9]), "v"(xr[-9]), "v"(q[10]), "v"(xr[-10]), "v"(q[11]), "v"(xr[-11]), "v"(shift));
	return d;
}


// ---------------------------------------------------------------------------------------------
// generic one-wavefront residual candidate evaluation (any block length / partition order)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int mask)
{
	const uint32_t lo = __shfl_xor((uint32_t)v, mask), hi = __shfl_xor((uint32_t)(v >> 32), mask);
	return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ uint32_t sat_add_u32(uint32_t est, uint32_t rbits)
{
	return rbits < 0xffffffffu - est ? est + rbits : 0xffffffffu;
}

// The same search when every lane sum is small (< 2^23, so that nothing can wrap or saturate and 32-bit arithmetic
// is exact): the 2^(max_po+1)-1 nodes of the partition tree are spread over the lanes -- leaf p on lane p, the
// merged partitions of the lower orders on the lanes after each other -- their sums come from ONE prefix sum over
// the lanes, and each lane evaluates at most two nodes instead of one node per partition order.
// set_partitioned_rice_ (stream_encoder.c:4997-5046) without branches, for sum < 2^23: mean-based parameter
// k = ilog2(((sum-1)*div) >> 18) + 1 (0 when that quotient is 0 or sum < 2), then the closed-form bit count
__device__ __forceinline__ void rice_node_small(uint32_t sum, uint32_t ns, uint32_t div, uint32_t rice_limit_m1, uint32_t &k, uint32_t &bits)
{
	const uint32_t sm1 = (sum > 1u ? sum : 1u) - 1u;
	const uint32_t x = (uint32_t)(((uint64_t)sm1 * div) >> 18);                  // < 2^23
	uint32_t kk = 32u - umin32((uint32_t)__clz((int)x), 32u);   // __clz(0) = 32
	kk = umin32(kk, rice_limit_m1);
	k = kk;
	bits = 4 + (1 + kk) * ns + ((sum << 1) >> kk) - (ns >> 1);
}
// Rice search over the partition orders as ONE butterfly: after exchange stage m every lane holds the |residual|
// sum of its aligned group of 2^m lanes, i.e. of "its" partition at order max_po - (m - e); it evaluates that node
// right there (all lanes of a group redundantly -- redundancy is free in SIMD) and the per-order bit totals ride along
// through the remaining stages, so that at the end every lane holds every order's total.  No prefix sums, no
// gathers; DPP adds for the first four stages.  Requires every lane sum < 2^23 (32-bit arithmetic exact).
// CE / CD: e and D when they are known at compile time (-1: not) -- the levels outside [e, e + D] and the butterflies that would carry
// their (zero) totals along then vanish from the code
template <int CE = -1, int CD = -1>
__device__ __forceinline__ uint32_t rice_search_nodes(uint32_t v, uint32_t e_rt, uint32_t n, uint32_t order, uint32_t max_po, uint32_t min_po,
                                                      uint32_t rice_limit, const uint32_t *divtab, uint8_t *kout, uint32_t *best_po_out, int lane)
{
	const uint32_t e = CE >= 0 ? (uint32_t)CE : e_rt;
	const uint32_t D = CD >= 0 ? (uint32_t)CD : max_po - min_po;                     // number of lower orders searched
	const uint32_t rl1 = rice_limit - 1;
	uint32_t tm[7], km[7];
#pragma unroll
	for(int m = 0; m < 7; m++) { tm[m] = 0; km[m] = 0; }
#pragma unroll
	for(int m = 0; m < 7; m++) {
		if((uint32_t)m >= e && (uint32_t)m - e <= D) {
			// node of this lane at partition order po: partition index lane >> m; partition 0 is `order` samples short
			const uint32_t po = max_po - ((uint32_t)m - e);
			const uint32_t nsf = n >> po;
			const uint32_t d0 = divtab[po * (MAX_ORDER + 1)], d1 = divtab[po * (MAX_ORDER + 1) + order];
			const bool p0 = (uint32_t)lane < (1u << m);
			rice_node_small(v, p0 ? nsf - order : nsf, p0 ? d1 : d0, rl1, km[m], tm[m]);
		}
		if(m < 6) {
			if(m == 0) { v = bfly_add<0>(v); }
			if(m == 1) { v = bfly_add<1>(v); }
			if(m == 2) { v = bfly_add<2>(v); }
			if(m == 3) { v = bfly_add<3>(v); }
			if(m == 4) { v = bfly_add<4>(v); }
			if(m == 5) { v = bfly_add<5>(v); }
#pragma unroll
			for(int j = 0; j <= m; j++) {
				if(CE >= 0 && (j < CE || j > CE + CD)) continue;       // (a level that is not searched has no total to carry)
				if(m == 0) tm[j] = bfly_add<0>(tm[j]);
				if(m == 1) tm[j] = bfly_add<1>(tm[j]);
				if(m == 2) tm[j] = bfly_add<2>(tm[j]);
				if(m == 3) tm[j] = bfly_add<3>(tm[j]);
				if(m == 4) tm[j] = bfly_add<4>(tm[j]);
				if(m == 5) tm[j] = bfly_add<5>(tm[j]);
			}
		}
	}
	// strict <, highest order first: ties keep the higher order (stream_encoder.c:4735-4763); all uniform by now
	uint32_t best_bits = 0, best_m = 0;
	bool have = false;
#pragma unroll
	for(int m = 0; m < 7; m++) {
		if((uint32_t)m >= e && (uint32_t)m - e <= D) {
			const uint32_t bits = 6 + (uint32_t)__builtin_amdgcn_readfirstlane((int)tm[m]);
			if(!have || bits < best_bits) { best_bits = bits; best_m = (uint32_t)m; have = true; }
		}
	}
	uint32_t kk = km[0];
#pragma unroll
	for(int m = 1; m < 7; m++) if((uint32_t)m == best_m) kk = km[m];
	if(((uint32_t)lane & ((1u << best_m) - 1u)) == 0) kout[(uint32_t)lane >> best_m] = (uint8_t)kk;
	__builtin_amdgcn_wave_barrier();
	*best_po_out = max_po - (best_m - e);
	return best_bits;
}


// Rice search over the partition orders for leaf sums held one per lane group (stream_encoder.c:4701-5075).
// v: this lane's |residual| sum over its S samples; e = 6 - max_po: 2^e adjacent lanes form a leaf partition.
__device__ __forceinline__ uint32_t rice_search_owner(uint64_t v, bool narrow, uint32_t e, uint32_t n, uint32_t order, uint32_t max_po, uint32_t min_po,
                                                      uint32_t rice_limit, const uint32_t *divtab, uint8_t *kout, uint32_t *best_po_out, int lane)
{
	for(uint32_t m = 0; m < e; m++) v += shfl_xor_u64(v, 1 << m);
	if(narrow) v = (uint32_t)v;
	uint64_t vlev[7];
	vlev[0] = v;
#pragma unroll
	for(int d = 1; d <= 6; d++) {
		if((uint32_t)d <= max_po - min_po) v += shfl_xor_u64(v, 1 << (e + d - 1));
		vlev[d] = v;
	}
	uint32_t klev[7];
	uint64_t blev[7];
	bool big = false;
#pragma unroll
	for(int d = 0; d <= 6; d++) {
		klev[d] = 0; blev[d] = 0;
		if((uint32_t)d <= max_po - min_po) {
			const uint32_t po = max_po - (uint32_t)d;
			const uint32_t g = e + (uint32_t)d;                        // log2 lanes per partition at this order
			const uint32_t pidx = (uint32_t)lane >> g;
			const bool rep = ((uint32_t)lane & ((1u << g) - 1u)) == 0;
			const uint32_t o = pidx == 0 ? order : 0;
			const uint32_t ns = (n >> po) - o;
			const uint32_t div = divtab[po * (MAX_ORDER + 1) + o];
			const uint64_t sum = vlev[d];
			uint32_t k;
			if(sum < 2 || (((sum - 1) * div) >> 18) == 0) k = 0;
			else k = ilog2_u64(((sum - 1) * div) >> 18) + 1;
			if(k >= rice_limit) k = rice_limit - 1;
			uint64_t bb = 4 + (uint64_t)(1 + k) * ns + (k ? (sum >> (k - 1)) : (sum << 1)) - (ns >> 1);
			if(bb > 0xffffffffull) bb = 0xffffffffull;
			klev[d] = k;
			blev[d] = rep ? bb : 0;
			big |= rep && bb >= (1ull << 25);
		}
	}
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
	uint32_t best_bits = 0, best_po = 0;
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
	const uint32_t g = e + db;
	if(((uint32_t)lane & ((1u << g) - 1u)) == 0) kout[(uint32_t)lane >> g] = (uint8_t)kk;
	__builtin_amdgcn_wave_barrier();
	*best_po_out = best_po;
	return best_bits;
}


// One WAVEFRONT evaluates one residual candidate (fixed or LPC) without any workgroup barrier:
// integer FIR out of LDS (lane owns S consecutive samples), |residual| per partition, the flat tree of
// merged sums, Rice parameter and bit estimate per partition, best partition order
// (find_best_partition_order_ / precompute_partition_info_sums_ / set_partitioned_rice_,
// stream_encoder.c:4701-5075).  Returns the estimated residual bits; Rice parameters of the best
// partition order go to kout[0 .. 2^best_po).
template <int MAXORD>
__device__ __forceinline__ uint32_t eval_candidate_wave(uint64_t *wsums, uint8_t *kcand, uint64_t *pob, uint8_t *kout, const uint32_t *divtab,
                                        const SigRef sig /* by value: this function is not always inlined */, uint32_t n, uint32_t order, const int32_t *q, int shift, uint32_t wide,
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
	bool bad = false;                 // mode 3: a residual left the 32-bit range -> no candidate (stream_encoder.c:4603-4608)

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
	{
		// one pass = 64 chunks, chunk `lane` of the pass to this lane; the FIR has a single call site (the kernel carries
		// seven flavours of it), what happens to the 16 residuals depends on the partition geometry
		const uint32_t lp = direct ? 64u / g : 1u;        // direct: leaves per pass (power of two)
		const uint32_t src = direct ? ((uint32_t)lane & (lp - 1)) * g : 0u, want = direct ? (uint32_t)lane / lp : 0u;
#pragma unroll 1
		for(uint32_t pass = 0; pass * 64 < nchunks; pass++) {
			const uint32_t base = (pass * 64 + (uint32_t)lane) * CHUNK;
			int32_t r[CHUNK];
			const bool active = base < n;
			if(active) bad = fir_chunk_dispatch<MAXORD>(sig, (int)base, qr, shift, fmode, r, order, n) || bad;
			if(direct) {
				uint64_t mine = 0;
				if(active) {
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
			else if(active) {
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
		if(direct && (uint32_t)lane >= nparts) vdirect = 0;
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
	return __any((int)bad) ? 0xffffffffu : best_bits;
}

} // namespace flacgpu
#endif
