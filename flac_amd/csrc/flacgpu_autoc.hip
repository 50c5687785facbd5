// flac_amd/csrc/flacgpu_autoc.hip -- autocorrelation of the windowed block (apply_apodization_ ->
// FLAC__lpc_window_data{,_partial} -> FLAC__lpc_compute_autocorrelation, stream_encoder.c:4318-4392, lpc.c:68-94,
// lpc_intrin_fma.c:46-72) for all frames of nominal length.
//
// Two kernels with the same arithmetic: autoc2_kernel (below; lane = (subframe, vector lane of the reference's accumulators)) and
// autoc3_kernel (further down; lane = subframe: a quarter of the conversions and LDS reads, the one the full-size batches of the
// presets with several window jobs run -- launch_autoc2 picks).
//
// autoc2_kernel's parallel shape: a WAVEFRONT takes one window job (whole block, a half, a third ...) of 16 consecutive
// (frame, candidate channel) subframes.  Lane = (subframe s = lane/4, vector lane l = lane%4) and carries the
// accumulators of ALL lags of "its" AVX lane l of the reference routine: the reference keeps, per lag j, a 4-wide
// fp64 vector acc_j and steps it 8 samples at a time,
//     acc_j[l] += fma(d[i], d[i-j], d[i+4] * d[i+4-j]),   i = L + 8k + l,
// so a lane needs a sliding window of its own block d[i-15 .. i+4] and nothing from other lanes.  Per step a lane
// converts 8 new floats to fp64 once and runs 3 fp64 operations per lag on them (the rounding sequence of the
// compiled reference: mul, fma, add); with one lane per (lag, l) instead, every chain step would convert its four
// operands again and read them from LDS again (4.5x the LDS traffic, 1.8x the VALU work).
//
// The block streams through a small per-wavefront LDS tile: coalesced loads of the interleaved PCM (each line of
// a stereo frame is loaded ONCE and feeds the left/right/mid/side subframes of that frame), wasted-bits shift,
// int->float, window multiply, one float per (subframe, sample).  Subframe regions are 100 floats apart, so the
// 32 lanes of an LDS access group (8 subframes x 4 l) hit 32 different banks.  No workgroup barriers anywhere.
//
// fp64 VALU bound (39 fp64 operations + 11 conversions per 8 samples and lane at -8).  -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "flacgpu_dev.h"
#include "flacgpu_devfn.h"

namespace flacgpu {

#ifndef AUTOC2_WAVES_PER_SIMD
#define AUTOC2_WAVES_PER_SIMD 3
#endif

constexpr int A2_T = 64;            // new samples per tile = 8 chain steps
constexpr int A2_H = 16;            // history samples in front of a tile (>= any lag)
constexpr int A2_IST = 100;         // floats per subframe region (80 + 3 + slack; = 4 mod 32)
constexpr int A2_ITEMS = 16;        // subframes per wavefront

struct A2Job {
	const float *w;
	uint32_t n, nd, full, part, dshift, i0;
};
// source sample and window weight of job element i (lpc.c:68 whole block; lpc.c:82-94 partial window);
// weight 0 (and a harmless source index) outside the job's data
__device__ __forceinline__ void a2_index(const A2Job &J, int32_t i, uint32_t &src, float &wt)
{
	src = 0; wt = 0.0f;
	if(i >= 0 && (uint32_t)i < J.nd) {
		uint32_t widx = 0;
		bool on = true;
		if(J.full) { src = (uint32_t)i; widx = (uint32_t)i; }
		else {
			const uint32_t ui = (uint32_t)i;
			if(ui >= J.i0 && ui < J.i0 + J.part) widx = J.n - J.part + (ui - J.i0);
			else if(ui < J.part) widx = ui;
			else on = false;
			if(on) src = J.dshift + ui;
		}
		if(on) wt = J.w[widx];
	}
}
// (float)sample * window, with a +0 where the weight is 0 (the reference writes 0.0f there)
__device__ __forceinline__ float a2_value(int32_t v, uint32_t wasted, float wt) { return __builtin_fmaf((float)(v >> wasted), wt, 0.0f); }

struct A2Items {
	const int32_t *ptr[A2_ITEMS];     // SRC 1: ptr[0..3] = the four frames; SRC 2: ptr[0..7] = the eight frames; SRC 0: one per subframe
	uint32_t which[A2_ITEMS];
	uint32_t wasted[A2_ITEMS];
};

// SRC: how the 16 subframes of a wavefront map to the interleaved PCM.
//   0  anything: one load per subframe and sample (pick_channel)
//   1  stereo with a full mid/side search: four frames x {L, R, M, S}, one 8-byte load per frame and sample feeds four subframes
//   2  stereo with two candidate channels per frame -- no mid/side search (-3), or the loose one (-1, -4: prep has picked L/R or
//      M/S per frame) --: eight frames x 2, one 8-byte load per frame and sample feeds both
template <int SRC>
struct A2Fetch {
	int32_t v[SRC == 1 ? 8 : A2_ITEMS];
	float wt;
};
__device__ __forceinline__ int32_t a2_pick2(int32_t l, int32_t r, uint32_t which) { return which == 0 ? l : which == 1 ? r : which == 2 ? ((l + r) >> 1) : (l - r); }
template <int SRC>
__device__ __forceinline__ void a2_fetch(const A2Job &J, const A2Items &I, uint32_t C, int32_t i, A2Fetch<SRC> &F)
{
	uint32_t src;
	a2_index(J, i, src, F.wt);
	if(SRC == 1) {
#pragma unroll
		for(int fr = 0; fr < 4; fr++) { const int2 lr = ((const int2 *)I.ptr[fr])[src]; F.v[2 * fr] = lr.x; F.v[2 * fr + 1] = lr.y; }
	}
	else if(SRC == 2) {
#pragma unroll
		for(int fr = 0; fr < 8; fr++) { const int2 lr = ((const int2 *)I.ptr[fr])[src]; F.v[2 * fr] = lr.x; F.v[2 * fr + 1] = lr.y; }
	}
	else {
#pragma unroll
		for(int t = 0; t < A2_ITEMS; t++) F.v[t] = pick_channel(I.ptr[t], C, src, I.which[t]);
	}
}
template <int SRC>
__device__ __forceinline__ void a2_store(float *tile, const A2Items &I, const A2Fetch<SRC> &F, int slot)
{
	if(SRC == 2) {
#pragma unroll
		for(int fr = 0; fr < 8; fr++) {
			const int32_t l = F.v[2 * fr], r = F.v[2 * fr + 1];
			tile[(2 * fr + 0) * A2_IST + slot] = a2_value(a2_pick2(l, r, I.which[2 * fr + 0]), I.wasted[2 * fr + 0], F.wt);
			tile[(2 * fr + 1) * A2_IST + slot] = a2_value(a2_pick2(l, r, I.which[2 * fr + 1]), I.wasted[2 * fr + 1], F.wt);
		}
	}
	else if(SRC == 1) {
#pragma unroll
		for(int fr = 0; fr < 4; fr++) {
			const int32_t l = F.v[2 * fr], r = F.v[2 * fr + 1];
			tile[(4 * fr + 0) * A2_IST + slot] = a2_value(l, I.wasted[4 * fr + 0], F.wt);
			tile[(4 * fr + 1) * A2_IST + slot] = a2_value(r, I.wasted[4 * fr + 1], F.wt);
			tile[(4 * fr + 2) * A2_IST + slot] = a2_value((l + r) >> 1, I.wasted[4 * fr + 2], F.wt);
			tile[(4 * fr + 3) * A2_IST + slot] = a2_value(l - r, I.wasted[4 * fr + 3], F.wt);
		}
	}
	else {
#pragma unroll
		for(int t = 0; t < A2_ITEMS; t++) tile[t * A2_IST + slot] = a2_value(F.v[t], I.wasted[t], F.wt);
	}
}

// one chain step of lpc_intrin_fma.c:46,61 for every lag; w[HB + c] = d[i + c] of this lane, HB = LAG-1 samples of history
#define A2_STEP(c) _Pragma("unroll") for(int j = 0; j < LAG; j++) acc[j] += fma(w[HB + (c)], w[HB + (c) - j], w[HB + (c) + 4] * w[HB + (c) + 4 - j])
// two steps of the lag-12 routine as compiled (lpc_intrin_fma.c:54): acc += t1 + t0 per 16 samples and, for lag 8
// only, x*y0 + x*y2 factored into x*(y0+y2) across the two halves
#define A2_PAIR(c) _Pragma("unroll") for(int j = 0; j < LAG; j++) { \
		if(j == 8) acc[j] += fma(w[HB + (c)], (w[HB + (c) - 8] + w[HB + (c) + 8]), w[HB + (c) + 4] * (w[HB + (c) - 4] + w[HB + (c) + 12])); \
		else { const double t0 = fma(w[HB + (c)], w[HB + (c) - j], w[HB + (c) + 4] * w[HB + (c) + 4 - j]); \
		       const double t1 = fma(w[HB + (c) + 8], w[HB + (c) + 8 - j], w[HB + (c) + 12] * w[HB + (c) + 12 - j]); acc[j] += (t1 + t0); } }

// GROUPED: a one-wavefront workgroup takes one window-job SET of JobTable (the whole block | the halves | the thirds ...:
// every set covers the block once, its jobs run one after the other) of one group of 16 subframes, and the sets of a group are
// consecutive workgroups of the SAME XCD (workgroups go round-robin over the 8 XCDs, so those are blockIdx b, b+8, b+16 ...).
// They start together and sweep the block at the same pace: a PCM line is fetched from HBM once and found in that XCD's L2 by
// the other sets, instead of once per pass of an unrelated wavefront elsewhere on the chip (-8: three passes).
template <int VARIANT, int LAG, int SRC, bool GROUPED>
__global__ __launch_bounds__(TPB, (VARIANT == 8 && SRC != 0) ? 4 : AUTOC2_WAVES_PER_SIMD) void autoc2_kernel(const DevParams P, const int32_t *__restrict__ pcm, const float *__restrict__ windows,
                                                                            uint32_t nmain, const JobTable *__restrict__ jt, const ChanPrep *__restrict__ preps,
                                                                            double *__restrict__ autoc_out)
{
	__shared__ float sh[GROUPED ? 1 : TPB / 64][A2_ITEMS * A2_IST];
	const int lane = (int)threadIdx.x & 63;
	const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	float *tile = sh[wave];
	const uint32_t nfc = nmain * P.ncand;
	const uint32_t ngroups = (nfc + A2_ITEMS - 1) / A2_ITEMS;
	uint32_t jb_lo, jb_hi, fc0;
	if(GROUPED) {
		const uint32_t nsets = jt->nsets, b = blockIdx.x;
		const uint32_t per_xcd = (ngroups / 8) * nsets, head = per_xcd * 8;      // blocks every XCD gets in full
		uint32_t group, set;
		if(b < head) { const uint32_t slot = b >> 3; group = (slot / nsets) * 8 + (b & 7); set = slot % nsets; }
		else { const uint32_t r = b - head; group = (ngroups / 8) * 8 + r / nsets; set = r % nsets; }
		if(group >= ngroups) return;
		fc0 = group * A2_ITEMS;
		jb_lo = jt->set_first[set]; jb_hi = jb_lo + jt->set_count[set];
	}
	else {
		const uint32_t wi = blockIdx.x * (TPB / 64) + wave;
		if(wi >= jt->njobs * ngroups) return;
		// jobs are enumerated longest first (whole block, halves, thirds ...): the long wavefronts start first
		jb_lo = wi / ngroups; jb_hi = jb_lo + 1;
		fc0 = (wi - jb_lo * ngroups) * A2_ITEMS;
	}
	const uint32_t N = P.blocksize, C = P.channels;
	constexpr uint32_t L = VARIANT;
	constexpr int HB = LAG - 1;

	A2Items I;
	uint32_t any_lpc = 0;
#pragma unroll
	for(int t = 0; t < A2_ITEMS; t++) {
		const uint32_t fc = fc0 + (uint32_t)t < nfc ? fc0 + (uint32_t)t : nfc - 1;
		const ChanPrep pr = preps[fc];
		I.wasted[t] = (uint32_t)__builtin_amdgcn_readfirstlane((int)pr.wasted);
		I.which[t] = (uint32_t)__builtin_amdgcn_readfirstlane((int)pr.which);
		any_lpc |= pr.flags & PREP_LPC;
		if(SRC == 0) I.ptr[t] = pcm + (size_t)(fc / P.ncand) * N * C;
	}
	if(SRC != 0) {
		constexpr int PER = SRC == 1 ? 4 : 2;                      // subframes per frame
#pragma unroll
		for(int fr = 0; fr < A2_ITEMS / PER; fr++) {
			const uint32_t f = fc0 / PER + (uint32_t)fr < nmain ? fc0 / PER + (uint32_t)fr : nmain - 1;
			I.ptr[fr] = pcm + (size_t)f * N * C;
		}
	}
	if(!__builtin_amdgcn_readfirstlane((int)any_lpc)) return;      // 16 constant subframes: nothing to analyse
	const int item = lane >> 2, l = lane & 3;
	const float *rd = tile + item * A2_IST + A2_H + l;                // rd[t] = d[tile base + t + l]

	for(uint32_t jb = jb_lo; jb < jb_hi; jb++) {
	const WindowJob jv = jt->jobs[jb];
	A2Job J;
	J.w = windows + (size_t)jv.apod * N;
	J.n = N; J.nd = jv.nd; J.full = jv.full; J.part = jv.part; J.dshift = jv.dshift; J.i0 = jv.i0;
	const uint32_t nd = jv.nd;
	const uint32_t nb = (nd - L) / 8;
	const uint32_t npairs12 = nb > 2 ? ((nb - 3) & ~1u) / 2 + 1 : 0;
	const uint32_t ntiles = (nb + 7) / 8;

	double acc[LAG];
#pragma unroll
	for(int j = 0; j < LAG; j++) acc[j] = 0.0;

	// history of tile 0: d[L-16, L)
	{
		A2Fetch<SRC> H;
		a2_fetch<SRC>(J, I, C, (int32_t)L - A2_H + (lane & 15), H);
		if(lane < A2_H) a2_store<SRC>(tile, I, H, lane);
	}
	A2Fetch<SRC> F;
	a2_fetch<SRC>(J, I, C, (int32_t)L + lane, F);
	for(uint32_t t = 0; t < ntiles; t++) {
		if(t) {
			// the last 16 samples become the history of the next tile (one wavefront: LDS operations execute in order)
			float hv[4];
#pragma unroll
			for(int u = 0; u < 4; u++) hv[u] = tile[item * A2_IST + A2_T + 4 * l + u];
			__builtin_amdgcn_wave_barrier();
#pragma unroll
			for(int u = 0; u < 4; u++) tile[item * A2_IST + 4 * l + u] = hv[u];
		}
		a2_store<SRC>(tile, I, F, A2_H + lane);
		if(t + 1 < ntiles) a2_fetch<SRC>(J, I, C, (int32_t)(L + A2_T * (t + 1)) + lane, F);
		__builtin_amdgcn_wave_barrier();
		const uint32_t k0 = 8 * t;
		const uint32_t ksteps = nb - k0 < 8 ? nb - k0 : 8;
		if(ksteps == 8 && (VARIANT != 12 || k0 + 8 <= 2 * npairs12)) {
#pragma unroll
			for(int h = 0; h < 2; h++) {
				double w[HB + 32];
#pragma unroll
				for(int u = 0; u < HB + 32; u++) w[u] = (double)rd[32 * h - HB + u];
				if(VARIANT != 12) { A2_STEP(0); A2_STEP(8); A2_STEP(16); A2_STEP(24); }
				else { A2_PAIR(0); A2_PAIR(16); }
			}
		}
		else {
			uint32_t kk = 0;
			while(kk < ksteps) {
				double w[HB + 16];
				const float *p = rd + 8 * kk;
#pragma unroll
				for(int u = 0; u < HB + 16; u++) w[u] = (double)p[u - HB];
				if(VARIANT == 12 && kk + 2 <= ksteps && k0 + kk + 2 <= 2 * npairs12) { A2_PAIR(0); kk += 2; }
				else { A2_STEP(0); kk += 1; }
			}
		}
		__builtin_amdgcn_wave_barrier();
	}

	// ---- head d[0,16) and tail d[nd-24, nd) of every subframe as plain copies (the tile is dead now) --------------
	const uint32_t tail_lo = nd - 24;               // nd > 32
	{
		A2Fetch<SRC> H;
		const int u = lane < 40 ? lane : 39;
		a2_fetch<SRC>(J, I, C, u < 16 ? u : (int32_t)tail_lo + (u - 16), H);
		if(lane < 40) a2_store<SRC>(tile, I, H, lane);
	}
	__builtin_amdgcn_wave_barrier();
	// lane (subframe, l) finishes lags l, l+4, l+8, l+12: the four lane accumulators of a lag sit in one quad
	const uint32_t max_lpc = P.max_lpc_order >= N ? N - 1 : P.max_lpc_order;
	const uint32_t lag = max_lpc + 1;
	const uint32_t fc = fc0 + (uint32_t)item;
	double *out = autoc_out + ((size_t)fc * P.max_jobs + jb) * AUTOC_STRIDE;
	const float *head = tile + item * A2_IST, *tail = head + 16;
#pragma unroll
	for(int m = 0; m < (LAG + 3) / 4; m++) {
		double a4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
		for(int jj = 0; jj < 4; jj++) {
			const int j = 4 * m + jj;                  // lag whose owner is quad lane jj
			if(j < LAG) {
#pragma unroll
				for(int q = 0; q < 4; q++) {
					const double v = __shfl(acc[j], (lane & ~3) | q);
					if(l == jj) a4[q] = v;
				}
			}
		}
		const uint32_t j = 4 * (uint32_t)m + (uint32_t)l;
		if(j < lag && fc < nfc) out[j] = autoc_finish2(head, tail, tail_lo, nd, L, j, a4);
	}
	__builtin_amdgcn_wave_barrier();          // the tile is reused by the next job of this wavefront
	}
}
#undef A2_STEP
#undef A2_PAIR

// ---------------------------------------------------------------------------------------------------------------------------
// autoc3_kernel: the same routine with a lane per SUBFRAME (round 4).  The four vector lanes l of the reference's accumulators
// (above: four GPU lanes of a quartet, each converting and reading the same floats again) are one GPU lane's acc[lag][l]: a sample
// is read from LDS and converted to fp64 ONCE per subframe, not once per l -- a quarter of the conversions and LDS reads, and the
// fill of the tile (wasted-bits shift, int -> float, window multiply) is a quarter per lane of compute as well.  Same operations
// on the same values in the same order per accumulator; 23 % fewer instructions per sample.
// A wavefront takes one window job of 64 consecutive subframes = 16 stereo frames x {L, R, M, S}; its tile holds 32 new samples
// per subframe (row stride 41 words: the 64 lanes reading "their" sample hit 32 banks twice), the history of a lag chain stays
// in registers.  Two wavefronts per SIMD (104 registers of accumulators + 88 of window), each with 52 independent chains.
// Stereo with a full mid/side search only (every preset from -5 up on stereo input); the other sources keep autoc2_kernel.
constexpr int A3_T = 32;             // new samples per tile = 4 chain steps
constexpr int A3_ST = 41;            // words per subframe row (>= 40: head + tail of the finish)
constexpr int A3_ITEMS = 64;         // subframes per wavefront
struct A3Fetch { int2 v[8]; float wt; };
// PLANES: left and right come from the planar channels the prep kernel left behind (16-bit input: 16-bit pairs, wasted bits
// shifted out -- a quarter of the interleaved 32-bit PCM's bytes per frame and sweep) instead of from the PCM
// (addresses: a wave-uniform base -- the group's first frame -- plus a 32-bit byte offset per lane, 16 frames at most: the loads
//  take the scalar base and a one-register offset instead of 64-bit address arithmetic per load)
template <bool PLANES>
__device__ __forceinline__ void a3_fetch(const A2Job &J, const int2 *__restrict__ pcm2, uint32_t N, uint32_t f0, uint32_t nmain, uint32_t half, int32_t i, A3Fetch &F)
{
	uint32_t src;
	a2_index(J, i, src, F.wt);
	const uint32_t last = nmain - 1u - f0;                  // (f0 < nmain)
	if(PLANES) {
		const char *gl = (const char *)((const int32_t *)pcm2 + (size_t)f0 * 4u * N), *gr = gl + (size_t)N * 4u;      // (pcm2: the planar channels here)
#pragma unroll
		for(int q = 0; q < 8; q++) {
			const uint32_t fr = umin32(2u * (uint32_t)q + half, last), off = fr * 16u * N + 2u * src;
			F.v[q] = make_int2((int)*(const int16_t *)(gl + off), (int)*(const int16_t *)(gr + off));
		}
	}
	else {
		const char *g = (const char *)(pcm2 + (size_t)f0 * N);
#pragma unroll
		for(int q = 0; q < 8; q++) {
			const uint32_t fr = umin32(2u * (uint32_t)q + half, last), off = (fr * N + src) * 8u;
			F.v[q] = *(const int2 *)(g + off);
		}
	}
}
// any_wasted: some subframe of the wavefront has wasted bits (wave-uniform; without, the shifts and their counts are not issued)
template <bool PLANES>
__device__ __forceinline__ void a3_store(float *tile, const uint32_t *wasted4 /* [16]: the four wasted-bits counts of a frame, a byte each */, bool any_wasted, uint32_t half, const A3Fetch &F, uint32_t col, bool tune_no_flat)
{
	if(any_wasted) {
#pragma unroll
		for(int q = 0; q < 8; q++) {
			const uint32_t fr = 2u * (uint32_t)q + half, w4 = wasted4[fr];
			// (a plane's samples come without their channel's wasted bits: put the zeros back for mid and side)
			const int32_t l = PLANES ? (int32_t)((uint32_t)F.v[q].x << (w4 & 0xffu)) : F.v[q].x, r = PLANES ? (int32_t)((uint32_t)F.v[q].y << ((w4 >> 8) & 0xffu)) : F.v[q].y;
			float *row = tile + fr * 4u * A3_ST + col;
			row[0 * A3_ST] = a2_value(l, w4 & 0xffu, F.wt);
			row[1 * A3_ST] = a2_value(r, (w4 >> 8) & 0xffu, F.wt);
			row[2 * A3_ST] = a2_value((l + r) >> 1, (w4 >> 16) & 0xffu, F.wt);
			row[3 * A3_ST] = a2_value(l - r, w4 >> 24, F.wt);
		}
	}
	else if(__all((int)(F.wt == 1.0f)) && !tune_no_flat) {
		// every column of this tile lies on the window's flat part (five sixths of a tukey(0.5 / 3) block, all of a partial window's
		// middle): fma(x, 1.0f, +0) is x bit for bit -- (float) of an integer is never -0 -- so the 32 multiplies of the lane are not
		// issued (round 6; wave-uniform: one compare and a branch per tile)
#pragma unroll
		for(int q = 0; q < 8; q++) {
			const uint32_t fr = 2u * (uint32_t)q + half;
			const int32_t l = F.v[q].x, r = F.v[q].y;
			int32_t lr = l + r;
			if(PLANES) asm("v_add_u32 %0, %1, %2" : "=v"(lr) : "v"(l), "v"(r));
			float *row = tile + fr * 4u * A3_ST + col;
			row[0 * A3_ST] = (float)l;
			row[1 * A3_ST] = (float)r;
			row[2 * A3_ST] = (float)(lr >> 1);
			row[3 * A3_ST] = (float)(l - r);
		}
	}
	else {
#pragma unroll
		for(int q = 0; q < 8; q++) {
			const uint32_t fr = 2u * (uint32_t)q + half;
			const int32_t l = F.v[q].x, r = F.v[q].y;
			int32_t lr = l + r;
			if(PLANES) asm("v_add_u32 %0, %1, %2" : "=v"(lr) : "v"(l), "v"(r));      // (the compiler, knowing both fit 16 bits, builds the mid sample from four 16-bit operations instead of an add and a shift)
			float *row = tile + fr * 4u * A3_ST + col;
			row[0 * A3_ST] = a2_value(l, 0u, F.wt);
			row[1 * A3_ST] = a2_value(r, 0u, F.wt);
			row[2 * A3_ST] = a2_value(lr >> 1, 0u, F.wt);
			row[3 * A3_ST] = a2_value(l - r, 0u, F.wt);
		}
	}
}
// IND (round 5): the 64 subframes of a wavefront are 64 INDEPENDENT candidate channels -- mono, stereo without a mid/side search or
// with the loose one, 3..8 channels: rows fc0 .. fc0 + 63 of the batch's (frame, candidate channel) list, whatever frames they
// belong to -- each read from its own planar copy (the prep kernels leave every candidate channel behind, wasted bits shifted
// out: 16-bit pairs where the subframe fits, ChanPrep::fmt, else 32-bit words).  Lane (half, sl) fills column sl of the rows
// 2 q + half: 32 loads of 2 or 4 bytes where the mid/side flavours have 8 of 8 or 4 -- next to 203 fp64-bound VALU instructions
// per 8 samples and subframe that is noise, and every channel layout gets the lane-per-subframe arithmetic (mono's
// autocorrelation took 0.59 ms per 16384 frames for ONE channel with autoc2_kernel's general source, stereo's four take 0.70).
struct A3FetchInd { int32_t v[32]; float wt; };
// kind (wave-uniform): 1 every row of the group holds 16-bit pairs and the group is whole, 0 the same with 32-bit words, 2 anything
// else (a mix of the two -- wasted bits bring a 24-bit channel under 16 --, or the batch's last group: rows beyond it repeat the last).
// The whole-group paths address a row as the group's scalar base + one 32-bit offset that moves by a scalar step per row pair: an add
// and a load per row (the general path: a clamp, a 64-bit multiply-add, a select and two predicated loads -- as much VALU work again
// as the arithmetic of the tile); is16: bit q set = row 2 q + half holds 16-bit pairs (this lane's half)
__device__ __forceinline__ void a3_fetch_ind(const A2Job &J, const int32_t *__restrict__ chan, uint32_t stride_words, uint32_t fc0, uint32_t nfc, uint32_t half, uint32_t is16, uint32_t kind,
                                             int32_t i, A3FetchInd &F)
{
	uint32_t src;
	a2_index(J, i, src, F.wt);
	const char *g = (const char *)(chan + (size_t)fc0 * stride_words);
	const uint32_t step = stride_words * 8u;                // two rows on
	if(kind == 1u) {
		uint32_t off = half * stride_words * 4u + 2u * src;
#pragma unroll
		for(int q = 0; q < 32; q++) { F.v[q] = (int32_t)*(const int16_t *)(g + off); off += step; }
	}
	else if(kind == 0u) {
		uint32_t off = half * stride_words * 4u + 4u * src;
#pragma unroll
		for(int q = 0; q < 32; q++) { F.v[q] = *(const int32_t *)(g + off); off += step; }
	}
	else {
		const uint32_t last = nfc - 1u - fc0;               // (fc0 < nfc)
#pragma unroll
		for(int q = 0; q < 32; q++) {
			const uint32_t row = umin32(2u * (uint32_t)q + half, last);
			const char *p = g + (size_t)row * stride_words * 4u;
			F.v[q] = (is16 >> q) & 1u ? (int32_t)*(const int16_t *)(p + 2u * src) : *(const int32_t *)(p + 4u * src);
		}
	}
}
__device__ __forceinline__ void a3_store_ind(float *tile, uint32_t half, const A3FetchInd &F, uint32_t col)
{
#pragma unroll
	for(int q = 0; q < 32; q++) tile[(2u * (uint32_t)q + half) * A3_ST + col] = a2_value(F.v[q], 0u, F.wt);
}
// one chain step (lpc_intrin_fma.c:46,61) / two steps of the lag-12 routine as compiled (:54), for the four vector lanes l:
// W(c) = d[first sample of the step + c] of this lane's subframe
#define A3_STEP(c) _Pragma("unroll") for(int l = 0; l < 4; l++) { _Pragma("unroll") for(int j = 0; j < LAG; j++) \
		acc[j][l] += fma(w[HB + (c) + l], w[HB + (c) + l - j], w[HB + (c) + l + 4] * w[HB + (c) + l + 4 - j]); }
#define A3_PAIR(c) _Pragma("unroll") for(int l = 0; l < 4; l++) { _Pragma("unroll") for(int j = 0; j < LAG; j++) { \
		if(j == 8) acc[j][l] += fma(w[HB + (c) + l], (w[HB + (c) + l - 8] + w[HB + (c) + l + 8]), w[HB + (c) + l + 4] * (w[HB + (c) + l - 4] + w[HB + (c) + l + 12])); \
		else { const double t0 = fma(w[HB + (c) + l], w[HB + (c) + l - j], w[HB + (c) + l + 4] * w[HB + (c) + l + 4 - j]); \
		       const double t1 = fma(w[HB + (c) + l + 8], w[HB + (c) + l + 8 - j], w[HB + (c) + l + 12] * w[HB + (c) + l + 12 - j]); acc[j][l] += (t1 + t0); } } }

// SETS: a wavefront takes a window-job SET (the whole block | the halves | the thirds: each covers the block once, JobTable) and
// the sets of a group of subframes are consecutive workgroups of the same XCD that sweep the block side by side, as in
// autoc2_kernel<..., GROUPED>: a PCM line is fetched from HBM once and found in that XCD's L2 by the other sets.  Without: a
// wavefront per job (twice the wavefronts, half as long: the chip's two-per-SIMD slots fill evenly, and every job fetches its own
// lines -- 75 KB per frame instead of 42).
// SRC: 0 stereo with a full mid/side search from the interleaved PCM, 1 the same from the left / right planes (PLANES), 2 independent
// subframes from their planes (IND)
// (independent subframes by sets: launched only while its wavefronts fit one per SIMD -- the whole register file is the wavefront's)
template <int VARIANT, int LAG, bool SETS, int SRC>
__global__ __launch_bounds__(64, (SETS && SRC == 2) ? 1 : 2) void autoc3_kernel(const DevParams P, const int32_t *__restrict__ pcm, const float *__restrict__ windows,
                                                       uint32_t nmain, const JobTable *__restrict__ jt, const ChanPrep *__restrict__ preps,
                                                       double *__restrict__ autoc_out, uint32_t no_flat_arg)
{
	const bool no_flat = no_flat_arg != 0;                          // (FLACGPU_NO_FLAT=1: the window multiply on every tile, for A/B runs)
	__shared__ float tile[A3_ITEMS * A3_ST];
	__shared__ uint32_t wasted4[A3_ITEMS / 4];
	constexpr bool PLANES = SRC == 1, IND = SRC == 2;
	const int lane = (int)threadIdx.x;
	const uint32_t nfc = nmain * (IND ? P.ncand : 4u), ngroups = (nfc + A3_ITEMS - 1) / A3_ITEMS;
	uint32_t jb_lo, jb_hi, fc0;
	if(SETS) {
		const uint32_t nsets = jt->nsets, b = blockIdx.x;
		const uint32_t per_xcd = (ngroups / 8) * nsets, head = per_xcd * 8;      // blocks every XCD gets in full
		uint32_t group, set;
		if(b < head) { const uint32_t slot = b >> 3; group = (slot / nsets) * 8 + (b & 7); set = slot % nsets; }
		else { const uint32_t r = b - head; group = (ngroups / 8) * 8 + r / nsets; set = r % nsets; }
		if(group >= ngroups) return;
		fc0 = group * A3_ITEMS;
		jb_lo = jt->set_first[set]; jb_hi = jb_lo + jt->set_count[set];
	}
	else {
		// jobs are enumerated longest first (whole block, halves, thirds ...): the long wavefronts start first
		jb_lo = blockIdx.x / ngroups; jb_hi = jb_lo + 1;
		fc0 = (blockIdx.x - jb_lo * ngroups) * A3_ITEMS;
	}
	// (round 6) IND: the batch's last group is rarely whole, and a partial group takes the general fetch -- as much VALU work again as the
	// arithmetic -- in EVERY job of that group: its whole-block job then is the launch's critical path (stereo without mid/side, 3640
	// blocks of 4608: 0.78 ms for 0.27 of work).  The last group starts 64 rows before the end instead: the rows it shares with the
	// group before are computed twice, the same values stored twice.
	if(IND && fc0 + A3_ITEMS > nfc && nfc >= A3_ITEMS) fc0 = nfc - A3_ITEMS;
	const uint32_t f0 = fc0 / 4u;
	const uint32_t N = P.blocksize;
	constexpr uint32_t L = VARIANT;
	constexpr int HB = LAG - 1;
	const uint32_t fc = fc0 + (uint32_t)lane;
	bool any_wasted;
	uint32_t is16 = 0, kind = 2;
	const uint32_t half = (uint32_t)lane >> 5, sl = (uint32_t)lane & 31u;
	{
		const ChanPrep pr = preps[fc < nfc ? fc : nfc - 1];
		((uint8_t *)wasted4)[lane] = (uint8_t)pr.wasted;
		if(!__any((int)(pr.flags & PREP_LPC))) return;             // 64 constant subframes: nothing to analyse
		any_wasted = __any((int)(pr.wasted != 0)) != 0;
		if(IND) {
			// which rows hold 16-bit pairs: bit q of this lane's word = row 2 q + half
			const uint64_t b = __ballot((int)(pr.fmt == 1u));
			uint32_t ev = 0, od = 0;
#pragma unroll
			for(int q = 0; q < 32; q++) { ev |= (uint32_t)((b >> (2 * q)) & 1ull) << q; od |= (uint32_t)((b >> (2 * q + 1)) & 1ull) << q; }
			is16 = half ? od : ev;
			// (a row beyond the batch's last subframe reads as a copy of the last: its lane's result is not stored)
			if(fc0 + A3_ITEMS <= nfc) kind = b == ~0ull ? 1u : b == 0ull ? 0u : 2u;
		}
	}
	const int2 *pcm2 = (const int2 *)pcm;
	const float *row = tile + lane * A3_ST;
	__builtin_amdgcn_wave_barrier();

#pragma unroll 1
	for(uint32_t jb = jb_lo; jb < jb_hi; jb++) {
	const WindowJob jv = jt->jobs[jb];
	A2Job J;
	J.w = windows + (size_t)jv.apod * N;
	J.n = N; J.nd = jv.nd; J.full = jv.full; J.part = jv.part; J.dshift = jv.dshift; J.i0 = jv.i0;
	const uint32_t nd = jv.nd;
	const uint32_t nb = (nd - L) / 8;
	const uint32_t npairs12 = nb > 2 ? ((nb - 3) & ~1u) / 2 + 1 : 0;
	const uint32_t ntiles = (nb + 3) / 4;

	double acc[LAG][4];
#pragma unroll
	for(int j = 0; j < LAG; j++) { acc[j][0] = 0.0; acc[j][1] = 0.0; acc[j][2] = 0.0; acc[j][3] = 0.0; }
#define A3_FETCH(idx) do { if constexpr(IND) a3_fetch_ind(J, pcm, P.chan_stride, fc0, nfc, half, is16, kind, (idx), G); else a3_fetch<PLANES>(J, pcm2, N, f0, nmain, half, (idx), F); } while(0)
#define A3_STORE(col) do { if constexpr(IND) a3_store_ind(tile, half, G, (col)); else a3_store<PLANES>(tile, wasted4, any_wasted, half, F, (col), no_flat); } while(0)
	double w[HB + A3_T];              // w[HB + c] = d[first sample of the tile + c]
	A3Fetch F;
	A3FetchInd G;
	(void)F; (void)G;
	// the samples in front of the first step: d[L - 16, L)
	A3_FETCH((int32_t)L - 16 + (int32_t)(sl & 15u));
	if(sl < 16) A3_STORE(sl);
	A3_FETCH((int32_t)L + (int32_t)sl);
	__builtin_amdgcn_wave_barrier();
#pragma unroll
	for(int u = 0; u < HB; u++) w[A3_T + u] = (double)row[16 - HB + u];
	for(uint32_t t = 0; t < ntiles; t++) {
		__builtin_amdgcn_wave_barrier();
		A3_STORE(sl);
		if(t + 1 < ntiles) A3_FETCH((int32_t)(L + A3_T * (t + 1)) + (int32_t)sl);
		__builtin_amdgcn_wave_barrier();
#pragma unroll
		for(int u = 0; u < HB; u++) w[u] = w[A3_T + u];
		// (the tile's samples are converted where the chain first wants them, eight -- sixteen -- at a time: converted all at once up
		//  front, the 32 doubles of the tile were live next to the 104 accumulators through the whole tile: 256 registers and spills
		//  in the flavours with the wider fetch.  Same values, same operations.)
		const uint32_t k0 = 4 * t;
		const uint32_t ksteps = nb - k0 < 4 ? nb - k0 : 4;
		if(VARIANT == 12) {
#pragma unroll
			for(int kk = 0; kk < 4; kk += 2) {
#pragma unroll
				for(int u = 0; u < 16; u++) w[HB + 8 * kk + u] = (double)row[8 * kk + u];
				if((uint32_t)kk + 2 <= ksteps && k0 + (uint32_t)kk + 2 <= 2 * npairs12) { A3_PAIR(8 * kk); }
				else {
					if((uint32_t)kk < ksteps) { A3_STEP(8 * kk); }
					if((uint32_t)kk + 1 < ksteps) { A3_STEP(8 * kk + 8); }
				}
			}
		}
		else {
#pragma unroll
			for(int kk = 0; kk < 4; kk++) {
#pragma unroll
				for(int u = 0; u < 8; u++) w[HB + 8 * kk + u] = (double)row[8 * kk + u];
				if((uint32_t)kk < ksteps) { A3_STEP(8 * kk); }
			}
		}
	}
	__builtin_amdgcn_wave_barrier();

	// ---- head d[0,16) and tail d[nd-24, nd) of every subframe as plain copies (the tile is dead now) --------------
	const uint32_t tail_lo = nd - 24;               // nd > 32
	{
		A3_FETCH(sl < 16 ? (int32_t)sl : (int32_t)(tail_lo + (sl - 16)));
		A3_STORE(sl);
		A3_FETCH((int32_t)(tail_lo + 16 + (sl & 7u)));
		if(sl < 8) A3_STORE(32 + sl);
	}
	__builtin_amdgcn_wave_barrier();
#undef A3_FETCH
#undef A3_STORE
	const uint32_t max_lpc = P.max_lpc_order >= N ? N - 1 : P.max_lpc_order;
	const uint32_t lag = max_lpc + 1;
	double *out = autoc_out + ((size_t)(fc < nfc ? fc : nfc - 1) * P.max_jobs + jb) * AUTOC_STRIDE;
#pragma unroll
	for(int j = 0; j < LAG; j++) {
		if((uint32_t)j < lag) {
			const double a4[4] = {acc[j][0], acc[j][1], acc[j][2], acc[j][3]};
			const double r = autoc_finish2(row, row + 16, tail_lo, nd, L, (uint32_t)j, a4);
			if(fc < nfc) out[j] = r;
		}
	}
	__builtin_amdgcn_wave_barrier();          // the tile is reused by the next job of this wavefront
	}
}
#undef A3_STEP
#undef A3_PAIR

// ---------------------------------------------------------------------------------------------------------------------------
// autoc4_kernel (round 5): the PLAIN loop of lpc.c:133-157 with a lane per subframe -- what the reference runs when lag > 16, i.e.
// from -l 16 up (stream_encoder.c:1058-1066 picks an FMA routine only below):
//     for sample: d = data[sample]; for coeff < lag: autoc[coeff] += d * data[sample + coeff]
// per lag a plain sequence of additions in increasing sample order (every product of two floats is exact in a double, so one
// v_fma_f64 is the reference's multiply and add).  Written backwards, autoc[c] += data[i - c] * data[i] for i = 0 .. n - 1 with
// zeros in front of the block, it is the same terms in the same order per lag, and the shape of autoc3_kernel: lane = subframe, LAG
// accumulators per lane, a sliding window of LAG - 1 earlier samples in registers, the samples streaming through the same 32-sample
// tile filled from the candidate channels' planes (IND source: every layout, stereo with mid/side included -- its four planes are
// there).  Samples past the job's end arrive as +0 (weight 0) and add nothing: no head, no tail, no finish.  LAG FMAs per sample
// and subframe: 17 at -l 16 (the -8 routines: 19.5 operations), 33 at -l 32.  The wavefront-per-job kernel this replaces ran one
// LANE per lag, 4096 dependent FMAs deep: 2.24 ms per 4096 frames at -8 -l 16 against 0.24 for -8 (profiles/r05_i_order_rate.txt).
template <int LAG>
__global__ __launch_bounds__(64, 2) void autoc4_kernel(const DevParams P, const int32_t *__restrict__ chan, const float *__restrict__ windows,
                                                       uint32_t nmain, const JobTable *__restrict__ jt, const ChanPrep *__restrict__ preps,
                                                       double *__restrict__ autoc_out)
{
	__shared__ float tile[A3_ITEMS * A3_ST];
	const int lane = (int)threadIdx.x;
	const uint32_t nfc = nmain * P.ncand, ngroups = (nfc + A3_ITEMS - 1) / A3_ITEMS;
	const uint32_t jb = blockIdx.x / ngroups;                             // jobs longest first: the long wavefronts start first
	uint32_t fc0 = (blockIdx.x - jb * ngroups) * A3_ITEMS;
	if(fc0 + A3_ITEMS > nfc && nfc >= A3_ITEMS) fc0 = nfc - A3_ITEMS;      // (the last group whole: see autoc3_kernel)
	const uint32_t N = P.blocksize;
	constexpr int HB = LAG - 1;
	const uint32_t fc = fc0 + (uint32_t)lane;
	const uint32_t half = (uint32_t)lane >> 5, sl = (uint32_t)lane & 31u;
	uint32_t is16 = 0, kind = 2;
	{
		const ChanPrep pr = preps[fc < nfc ? fc : nfc - 1];
		if(!__any((int)(pr.flags & PREP_LPC))) return;
		const uint64_t b = __ballot((int)(pr.fmt == 1u));
		uint32_t ev = 0, od = 0;
#pragma unroll
		for(int q = 0; q < 32; q++) { ev |= (uint32_t)((b >> (2 * q)) & 1ull) << q; od |= (uint32_t)((b >> (2 * q + 1)) & 1ull) << q; }
		is16 = half ? od : ev;
		if(fc0 + A3_ITEMS <= nfc) kind = b == ~0ull ? 1u : b == 0ull ? 0u : 2u;
	}
	const float *row = tile + lane * A3_ST;
	const WindowJob jv = jt->jobs[jb];
	A2Job J;
	J.w = windows + (size_t)jv.apod * N;
	J.n = N; J.nd = jv.nd; J.full = jv.full; J.part = jv.part; J.dshift = jv.dshift; J.i0 = jv.i0;
	const uint32_t nd = jv.nd, ntiles = (nd + A3_T - 1) / A3_T;
	double acc[LAG];
#pragma unroll
	for(int j = 0; j < LAG; j++) acc[j] = 0.0;
	double w[HB + A3_T];                  // w[HB + c] = d[first sample of the tile + c]; w[0 .. HB) the samples in front of it
#pragma unroll
	for(int u = 0; u < HB; u++) w[A3_T + u] = 0.0;
	A3FetchInd G;
	a3_fetch_ind(J, chan, P.chan_stride, fc0, nfc, half, is16, kind, (int32_t)sl, G);
	for(uint32_t t = 0; t < ntiles; t++) {
		__builtin_amdgcn_wave_barrier();
		a3_store_ind(tile, half, G, sl);
		if(t + 1 < ntiles) a3_fetch_ind(J, chan, P.chan_stride, fc0, nfc, half, is16, kind, (int32_t)(A3_T * (t + 1) + sl), G);
		__builtin_amdgcn_wave_barrier();
#pragma unroll
		for(int u = 0; u < HB; u++) w[u] = w[A3_T + u];
#pragma unroll
		for(int k8 = 0; k8 < A3_T; k8 += 8) {
#pragma unroll
			for(int u = 0; u < 8; u++) w[HB + k8 + u] = (double)row[k8 + u];
#pragma unroll
			for(int u = 0; u < 8; u++) {
#pragma unroll
				for(int c = 0; c < LAG; c++) acc[c] = fma(w[HB + k8 + u - c], w[HB + k8 + u], acc[c]);
			}
		}
	}
	const uint32_t max_lpc = P.max_lpc_order >= N ? N - 1 : P.max_lpc_order;
	const uint32_t lag = max_lpc + 1;
	double *out = autoc_out + ((size_t)(fc < nfc ? fc : nfc - 1) * P.max_jobs + jb) * AUTOC_STRIDE;
	if(fc < nfc) {
#pragma unroll
		for(int j = 0; j < LAG; j++) if((uint32_t)j < lag) out[j] = acc[j];
	}
}
// from -l 16 up, blocks longer than 32 samples, every candidate channel's plane there
bool autoc4_applicable(const DevParams &P) { return P.blocksize >= 64 && P.max_lpc_order >= 16 && P.autoc_variant == 0 && !P.wide_samples && !tune().no_fast1; }
hipError_t launch_autoc4(const DevParams &P, const int32_t *chan, const float *win, uint32_t nmain, uint32_t njobs, const JobTable *jt, const ChanPrep *preps, double *autoc, hipStream_t s)
{
	if(nmain == 0 || njobs == 0) return hipSuccess;
	const uint32_t max_lpc = P.max_lpc_order >= P.blocksize ? P.blocksize - 1 : P.max_lpc_order;
	const uint32_t lag = max_lpc + 1, ngroups = (nmain * P.ncand + A3_ITEMS - 1) / A3_ITEMS;
	note_launch(K_AUTOC4);
	if(lag <= 17) hipLaunchKernelGGL(autoc4_kernel<17>, dim3(njobs * ngroups), dim3(64), 0, s, P, chan, win, nmain, jt, preps, autoc);
	else if(lag <= 25) hipLaunchKernelGGL(autoc4_kernel<25>, dim3(njobs * ngroups), dim3(64), 0, s, P, chan, win, nmain, jt, preps, autoc);
	else hipLaunchKernelGGL(autoc4_kernel<33>, dim3(njobs * ngroups), dim3(64), 0, s, P, chan, win, nmain, jt, preps, autoc);
	return hipGetLastError();
}

// the lane-per-subframe kernel: stereo with a full mid/side search and enough subframes for two wavefronts per SIMD (below that
// autoc2_kernel's mid/side flavour, four times as fine-grained, is the faster one); every other channel layout from the planes
// (IND) as soon as half the SIMDs get a wavefront -- autoc2_kernel's general source is three times slower per channel
static bool autoc3_ms(const DevParams &P) { return P.channels == 2 && P.ms_mode == 1 && P.ncand == 4; }
static bool autoc3_wanted(const DevParams &P, const int32_t *chan, uint32_t nmain, uint32_t njobs)
{
	const int mode = tune().autoc3_mode;      // FLACGPU_AUTOC3 = 0: never, 1: whenever it applies, 2: when it fills the chip
	if(mode == 0 || P.blocksize < 64) return false;
	if(!autoc3_ms(P) && (!chan || tune().no_fast1)) return false;
	const uint32_t waves = njobs * ((nmain * P.ncand + A3_ITEMS - 1) / A3_ITEMS);
	return mode == 1 || waves >= (autoc3_ms(P) ? 2048u : 128u);
}
template <int VARIANT, int LAG>
static void launch_autoc3_t(const DevParams &P, const int32_t *pcm, const int32_t *chan, const float *win, uint32_t nmain, uint32_t njobs, uint32_t nsets, const JobTable *jt, const ChanPrep *preps, double *autoc, hipStream_t s)
{
	const uint32_t ngroups = (nmain * P.ncand + A3_ITEMS - 1) / A3_ITEMS;
	const int sets = tune().autoc3_sets;
	if(!autoc3_ms(P)) {
		// independent subframes from their planes.  A wavefront per job fills the chip better when the subframes are few (six short
		// wavefronts per group at -8).  Where the jobs no longer fit one per SIMD but the SETS (whole | halves | thirds: one sweep of the
		// block each, equal lengths) still do -- 171..341 groups -- a wavefront per set, with the whole register file to itself (launch
		// bounds 1: no spills, and the dispatcher cannot put two on one SIMD): mono's 16384 frames 0.320 -> 0.304 ms
		// (profiles/r05_o_chan_rate_ind_sets_ab.txt).  The floor of either shape is one wavefront's sweep of the block ALONE on a SIMD,
		// 0.27-0.30 ms (two per SIMD hide each other's LDS and conversion latencies: 0.21 ms per sweep in the stereo flavour).
		const int isets = tune().autoc3_ind_sets;
		const uint32_t simds = 1024;
		const bool by_sets = nsets >= 2 && nsets <= 8 && (isets == 1 || (isets == 2 && njobs * ngroups > simds && nsets * ngroups <= simds));
		note_launch(K_AUTOC3 | K_AUTOC3_PLANES | K_AUTOC1 | (by_sets ? K_AUTOC3_SETS : 0u));
		if(by_sets) hipLaunchKernelGGL((autoc3_kernel<VARIANT, LAG, true, 2>), dim3(nsets * ngroups), dim3(64), 0, s, P, chan, win, nmain, jt, preps, autoc, (uint32_t)tune().no_flat);
		else hipLaunchKernelGGL((autoc3_kernel<VARIANT, LAG, false, 2>), dim3(njobs * ngroups), dim3(64), 0, s, P, chan, win, nmain, jt, preps, autoc, (uint32_t)tune().no_flat);
		return;
	}
	// 16-bit input: the prep kernel's left and right planes are 16-bit pairs (ChanPrep::fmt = 1 whenever sbps <= 16) -- read those
	const int planes = tune().autoc3_planes;
	const bool pl = planes && chan && P.bps <= 16;
	const int32_t *src = pl ? chan : pcm;
	// By SETS (a wavefront sweeps the block once per set: equal wavefronts, a PCM line fetched once per group) or by JOBS (six wavefronts
	// of three lengths per group at -8, longest first).  Equal wavefronts that do not fill the chip's 2048 slots a whole number of times
	// leave SIMDs idle while the ones with two wavefronts finish: 8192 frames = 1536 wavefronts took one full round, 0.425 ms, for 0.75
	// of work; by jobs 0.319 ms (the step +10 %), 12288 frames 0.610 -> 0.504 (+6 %), equal at 16384, by sets ahead from 32768 frames on
	// (profiles/r06_ay_ab_autoc3_sets_by_batch.txt).  FLACGPU_AUTOC3_SETS = 0: never by sets, 2: always, 1: by sets from 1.5 rounds up.
	const bool by_sets = sets && nsets >= 2 && nsets <= 8 && (sets == 2 || nsets * ngroups >= 3072u);
	note_launch(K_AUTOC3 | (pl ? K_AUTOC3_PLANES : 0u) | (by_sets ? K_AUTOC3_SETS : 0u));
	if(by_sets) {
		if(pl) hipLaunchKernelGGL((autoc3_kernel<VARIANT, LAG, true, 1>), dim3(nsets * ngroups), dim3(64), 0, s, P, src, win, nmain, jt, preps, autoc, (uint32_t)tune().no_flat);
		else hipLaunchKernelGGL((autoc3_kernel<VARIANT, LAG, true, 0>), dim3(nsets * ngroups), dim3(64), 0, s, P, src, win, nmain, jt, preps, autoc, (uint32_t)tune().no_flat);
	}
	else if(pl) hipLaunchKernelGGL((autoc3_kernel<VARIANT, LAG, false, 1>), dim3(njobs * ngroups), dim3(64), 0, s, P, src, win, nmain, jt, preps, autoc, (uint32_t)tune().no_flat);
	else hipLaunchKernelGGL((autoc3_kernel<VARIANT, LAG, false, 0>), dim3(njobs * ngroups), dim3(64), 0, s, P, src, win, nmain, jt, preps, autoc, (uint32_t)tune().no_flat);
}

template <int VARIANT, int LAG>
static void launch_autoc2_t(const DevParams &P, const int32_t *pcm, const float *win, uint32_t nmain, uint32_t njobs, uint32_t nsets, const JobTable *jt,
                            const ChanPrep *preps, double *autoc, hipStream_t s)
{
	const uint32_t nfc = nmain * P.ncand, ngroups = (nfc + A2_ITEMS - 1) / A2_ITEMS;
	const bool ms4 = P.channels == 2 && P.ms_mode == 1, st2 = P.channels == 2 && P.ncand == 2;
	const int nogroup = tune().autoc2_ungrouped;
	note_launch(K_AUTOC2);
	if(nsets >= 2 && nsets <= 8 && !nogroup) {
		// one single-wavefront workgroup per (group of subframes, job set)
		const dim3 grid(ngroups * nsets), block(64);
		if(ms4) hipLaunchKernelGGL((autoc2_kernel<VARIANT, LAG, 1, true>), grid, block, 0, s, P, pcm, win, nmain, jt, preps, autoc);
		else if(st2) hipLaunchKernelGGL((autoc2_kernel<VARIANT, LAG, 2, true>), grid, block, 0, s, P, pcm, win, nmain, jt, preps, autoc);
		else hipLaunchKernelGGL((autoc2_kernel<VARIANT, LAG, 0, true>), grid, block, 0, s, P, pcm, win, nmain, jt, preps, autoc);
		return;
	}
	const uint32_t waves = njobs * ngroups;
	const dim3 grid((waves + TPB / 64 - 1) / (TPB / 64)), block(TPB);
	if(ms4) hipLaunchKernelGGL((autoc2_kernel<VARIANT, LAG, 1, false>), grid, block, 0, s, P, pcm, win, nmain, jt, preps, autoc);
	else if(st2) hipLaunchKernelGGL((autoc2_kernel<VARIANT, LAG, 2, false>), grid, block, 0, s, P, pcm, win, nmain, jt, preps, autoc);
	else hipLaunchKernelGGL((autoc2_kernel<VARIANT, LAG, 0, false>), grid, block, 0, s, P, pcm, win, nmain, jt, preps, autoc);
}

// true when the streaming kernel serves the nominal-length frames of this configuration
bool autoc2_applicable(const DevParams &P) { return P.blocksize > 32 && P.max_lpc_order > 0 && P.autoc_variant != 0 && !P.wide_samples; }

hipError_t launch_autoc2(const DevParams &P, const int32_t *pcm, const int32_t *chan, const float *win, uint32_t nmain, uint32_t njobs, uint32_t nsets, const JobTable *jt,
                         const ChanPrep *preps, double *autoc, hipStream_t s)
{
	if(nmain == 0 || njobs == 0) return hipSuccess;
	const uint32_t max_lpc = P.max_lpc_order >= P.blocksize ? P.blocksize - 1 : P.max_lpc_order;
	const uint32_t lag = max_lpc + 1;
	// (lags 10..12 of the lag-12 routine and 14..16 of the lag-16 one would spill: they stay with autoc2_kernel)
	if(autoc3_wanted(P, chan, nmain, njobs) && (P.autoc_variant == 8 || (P.autoc_variant == 12 && lag <= 9) || (P.autoc_variant == 16 && lag <= 13))) {
		if(P.autoc_variant == 8) launch_autoc3_t<8, 8>(P, pcm, chan, win, nmain, njobs, nsets, jt, preps, autoc, s);
		else if(P.autoc_variant == 12) launch_autoc3_t<12, 9>(P, pcm, chan, win, nmain, njobs, nsets, jt, preps, autoc, s);
		else launch_autoc3_t<16, 13>(P, pcm, chan, win, nmain, njobs, nsets, jt, preps, autoc, s);
		return hipGetLastError();
	}
	if(P.autoc_variant == 8) launch_autoc2_t<8, 8>(P, pcm, win, nmain, njobs, nsets, jt, preps, autoc, s);
	else if(P.autoc_variant == 12) { if(lag <= 9) launch_autoc2_t<12, 9>(P, pcm, win, nmain, njobs, nsets, jt, preps, autoc, s); else launch_autoc2_t<12, 12>(P, pcm, win, nmain, njobs, nsets, jt, preps, autoc, s); }
	else { if(lag <= 13) launch_autoc2_t<16, 13>(P, pcm, win, nmain, njobs, nsets, jt, preps, autoc, s); else launch_autoc2_t<16, 16>(P, pcm, win, nmain, njobs, nsets, jt, preps, autoc, s); }
	return hipGetLastError();
}

} // namespace flacgpu
