// flac_amd/csrc/flacgpu_analyze.hip -- the model search of process_subframe_ (stream_encoder.c:4045-4290)
// as FOUR phase-specialised kernels, each with the parallel shape and occupancy its phase wants:
//
//   prep_kernel   workgroup per (frame, candidate channel): wasted bits (stream_encoder.c:5077), loose mid/side
//                 (:3778), fixed-predictor sums + order guess (fixed.c:222 / fixed_intrin_avx2.c:57), CONSTANT
//                 detection (:4111-4140).                                      -> ChanPrep, Candidate[0]
//   autoc_kernel  WAVEFRONT per (frame, channel, window job): streams the block through a small LDS tile:
//                 coalesced int32 loads -> window multiply (lpc.c:68-94) -> floats de-interleaved by (i mod 4) so
//                 that each lane's two operand streams are consecutive 8-byte LDS reads -> the reference's fp64
//                 chains (lpc_intrin_fma.c:46-72).                              -> autoc[job][lag]
//   model_kernel  LANE per (frame, channel, analysis): punch-out subtraction (stream_encoder.c:4370), Levinson-
//                 Durbin, order guess, quantisation (lpc.c:176,1608,220).       -> Candidate[1+analysis]
//   eval_kernel   workgroup per (frame, channel), one wavefront per residual candidate: each lane OWNS n/64
//                 consecutive samples (so a lane's |residual| sum is a whole leaf partition, no cross-lane
//                 reduction), 16-bit channels run the FIR as v_dot2_i32_i16 on packed sample pairs, Rice search
//                 as a butterfly over partition orders (stream_encoder.c:4701-5075), first-minimum winner.
//                                                                               -> SubDecision
//
// The hand-off records between the kernels are a few hundred bytes per channel; residuals and windowed data never
// reach HBM.  Integer + fp64 VALU work: no MFMA by design.  Compile with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "flacgpu_dev.h"
#include "flacgpu_devfn.h"

namespace flacgpu {

#ifndef AUTOC_WAVES_PER_SIMD
#define AUTOC_WAVES_PER_SIMD 8
#endif
#ifndef EVAL_WAVES_PER_SIMD
#define EVAL_WAVES_PER_SIMD 4
#endif

// XCD-aware mapping (blocks round-robin over the 8 XCDs): keep the candidate channels of one frame on one XCD so
// that their shared PCM lines hit that XCD's L2.
__device__ __forceinline__ void map_block(uint32_t b, uint32_t nframes, uint32_t ncand, uint32_t &f, uint32_t &cand)
{
	const uint32_t total = nframes * ncand;
	const uint32_t per_xcd_full = (total / (8 * ncand)) * ncand;
	const uint32_t head = per_xcd_full * 8;
	const uint32_t lin = b < head ? (b & 7) * per_xcd_full + (b >> 3) : b;
	f = lin / ncand;
	// rotate the channel order from frame to frame: workgroups reach the CUs of an XCD round-robin, and with a fixed
	// order the slowest channel (side: 17-bit samples, no dot2) of every frame would land on the same quarter of the CUs
	cand = (lin + f + (f >> 3) + (f >> 6)) % ncand;
}

// ---------------------------------------------------------------------------------------------------------
// prep_kernel
// ---------------------------------------------------------------------------------------------------------
template <int DUMMY>
__global__ __launch_bounds__(TPB) void prep_kernel(const DevParams P, const int32_t *__restrict__ pcm, uint32_t nframes, uint32_t tail_n, uint32_t f_lo,
                                                   ChanPrep *__restrict__ preps, Candidate *__restrict__ cands, int *__restrict__ valid, int32_t *__restrict__ chan)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	__shared__ uint64_t scratch[8];
	__shared__ uint64_t red[(TPB / 64) * 5];
	const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const uint32_t C = P.channels, N = P.blocksize;
	uint32_t f, cand;
	if(f_lo) { f = f_lo + blockIdx.x / P.ncand; cand = blockIdx.x % P.ncand; }      // frames [f_lo, nframes) only
	else map_block(blockIdx.x, nframes, P.ncand, f, cand);
	const bool is_tail = tail_n != 0 && f == nframes - 1;
	const uint32_t n = is_tail ? tail_n : N;
	const int32_t *frame_pcm = pcm + (size_t)f * N * C;
	int32_t *sig = (int32_t *)smem;
	const size_t fc = (size_t)f * P.ncand + cand;
	const uint32_t cstride = P.ncslots;

	uint32_t which = cand;
	if(P.ms_mode == 2) {
		// loose mid/side (stream_encoder.c:3778-3807): both workgroups of the frame compute the decision
		uint64_t lr = 0, ms = 0;
		const int2 *p = (const int2 *)frame_pcm;
		if(!P.wide_samples) {
			for(uint32_t i = 1 + (uint32_t)tid; i < n; i += TPB) {
				const int2 a = p[i], b = p[i - 1];
				const int32_t pl = a.x - b.x, pr = a.y - b.y;
				lr += (uint64_t)(uint32_t)(abs(pl) + abs(pr));
				ms += (uint64_t)(uint32_t)(abs((pl + pr) >> 1) + abs(pl - pr));
			}
		}
		else {
			// 25 bits per sample and more: 64-bit differences (stream_encoder.c:3789-3796)
			for(uint32_t i = 1 + (uint32_t)tid; i < n; i += TPB) {
				const int2 a = p[i], b = p[i - 1];
				const int64_t pl = (int64_t)a.x - (int64_t)b.x, pr = (int64_t)a.y - (int64_t)b.y;
				lr += abs_i64(pl) + abs_i64(pr);
				ms += abs_i64((pl + pr) >> 1) + abs_i64(pl - pr);
			}
		}
		lr = block_reduce_add_u64(lr, scratch, tid);
		ms = block_reduce_add_u64(ms, scratch, tid);
		if(!(lr < ms)) which = 2 + cand;
	}
	// limit_min_bitrate (stream_encoder.c:3874-3879)
	bool disable_constant = P.disable_constant != 0;
	if(P.limit_min_bitrate && !disable_constant && (P.ms_mode == 2 ? which == 1 : which >= C - 1)) {
		uint32_t diff = 0;
		for(uint32_t i = (uint32_t)tid; i < n; i += TPB)
			for(uint32_t c = 0; c + 1 < C; c++) diff |= (uint32_t)(frame_pcm[(size_t)i * C + c] ^ frame_pcm[c]);
		diff = block_reduce_or_u32(diff, scratch, tid);
		if(diff == 0) disable_constant = true;
	}
	uint32_t wasted;
	bool s64 = false;                                   // this channel keeps 64-bit samples (33 bits after the shift)
	int64_t *sig64 = (int64_t *)smem;
	const bool stream = P.stream_sig != 0;              // the block does not fit the LDS: every pass reads the PCM again
	const bool side33 = P.bps == 32 && C == 2 && which == 3;
	// unshifted sample i of this channel (stream mode)
	auto RAW = [&](uint32_t i) -> int64_t {
		if(C == 2) { const int2 lr = ((const int2 *)frame_pcm)[i]; return which == 0 ? (int64_t)lr.x : which == 1 ? (int64_t)lr.y : which == 2 ? (((int64_t)lr.x + (int64_t)lr.y) >> 1) : side64(lr); }
		return (int64_t)pick_channel(frame_pcm, C, i, which);
	};
	if(stream) {
		uint32_t olo = 0, ohi = 0;
		for(uint32_t i = (uint32_t)tid; i < n; i += TPB) { const uint64_t v = (uint64_t)RAW(i); olo |= (uint32_t)v; ohi |= (uint32_t)(v >> 32); }
		olo = block_reduce_or_u32(olo, scratch, tid);
		ohi = block_reduce_or_u32(ohi, scratch, tid);
		wasted = olo ? (uint32_t)(__ffs((int)olo) - 1) : (side33 ? (ohi ? 32u : 1u) : 0u);
		if(wasted > P.bps) wasted = P.bps;
		s64 = side33 && wasted == 0;
	}
	else if(side33) {
		// side channel of a 32-bit stream: 33 bits; get_wasted_bits_wide_ (stream_encoder.c:5103), all-zero loses 1 bit
		const int2 *p = (const int2 *)frame_pcm;
		uint32_t olo = 0, ohi = 0;
		for(uint32_t i = (uint32_t)tid; i < n; i += TPB) { const uint64_t v = (uint64_t)side64(p[i]); olo |= (uint32_t)v; ohi |= (uint32_t)(v >> 32); }
		olo = block_reduce_or_u32(olo, scratch, tid);
		ohi = block_reduce_or_u32(ohi, scratch, tid);
		wasted = olo ? (uint32_t)(__ffs((int)olo) - 1) : ohi ? 32u : 1u;
		if(wasted > P.bps) wasted = P.bps;
		s64 = wasted == 0;
		const uint32_t nround = ((n + 15u) & ~15u) + 16u;
		if(s64) {
			if(tid < 32) sig64[sigidx(tid - 32)] = 0;
			for(uint32_t i = n + (uint32_t)tid; i < nround; i += TPB) sig64[sigidx((int)i)] = 0;
			for(uint32_t i = (uint32_t)tid; i < n; i += TPB) sig64[sigidx((int)i)] = side64(p[i]);
		}
		else {
			if(tid < 32) sig[sigidx(tid - 32)] = 0;
			for(uint32_t i = n + (uint32_t)tid; i < nround; i += TPB) sig[sigidx((int)i)] = 0;
			for(uint32_t i = (uint32_t)tid; i < n; i += TPB) sig[sigidx((int)i)] = (int32_t)(side64(p[i]) >> wasted);
		}
	}
	else {
		uint32_t orv;
		load_signal(sig, frame_pcm, C, n, which, &orv, tid);
		orv = block_reduce_or_u32(orv, scratch, tid);
		wasted = orv ? (uint32_t)(__ffs((int)orv) - 1) : 0;
		if(wasted > P.bps) wasted = P.bps;
		if(wasted) for(uint32_t i = (uint32_t)tid; i < n; i += TPB) sig[sigidx((int)i)] >>= wasted;
	}
	const uint32_t sbps = P.bps - wasted + (which == C + 1 ? 1 : 0);
	__syncthreads();
	// sample i of the shifted channel, whatever its width
	auto SV = [&](int i) -> int64_t {
		if(stream) return (i < 0 || (uint32_t)i >= n) ? 0 : (RAW((uint32_t)i) >> wasted);
		return s64 ? sig64[sigidx(i)] : (int64_t)sig[sigidx(i)];
	};
	// planar copy of the shifted channel for the evaluation and pack kernels
	const uint32_t fmt = s64 ? 2u : sbps <= 16 ? 1u : 0u;
	{
		int32_t *dst = chan + fc * (size_t)P.chan_stride;
		if(fmt == 2) for(uint32_t i = (uint32_t)tid; i < n; i += TPB) ((int64_t *)dst)[i] = SV((int)i);
		else if(fmt) for(uint32_t i = (uint32_t)tid; i < n; i += TPB) ((uint16_t *)dst)[i] = (uint16_t)SV((int)i);
		else for(uint32_t i = (uint32_t)tid; i < n; i += TPB) dst[i] = (int32_t)SV((int)i);
	}

	uint32_t flags = 0, fixed_order = 0;
	int64_t constant = 0;
	const uint32_t verbatim_bits = (P.disable_verbatim && n >= 4) ? 0xffffffffu : 8 + wasted + n * sbps;
	if(n > 4) {
		// fixed predictor estimate (fixed.c:222 / fixed_intrin_avx2.c:57; from 28 bits the overflow-checked flavours
		// fixed_intrin_avx2.c:187 and, for 33-bit samples, fixed.c:424)
		const uint32_t n4 = n - 4;
		const bool fwide = !(sbps + ilog2_u32(n4 * 17) < 32);
		const bool flimit = sbps >= 28;
		uint64_t e0 = 0, e1 = 0, e2 = 0, e3 = 0, e4 = 0;
		uint32_t over = 0;            // flimit: bit k = a residual of order k does not fit 31 bits + sign
		if(flimit) {
			const bool lanes4 = !s64 && (n4 & 3) != 0;       // the four-lane AVX2 routine on a length that is no multiple of 4
			const uint32_t q4 = n4 / 4;
			auto acc = [&](int64_t d0, int64_t d1, int64_t d2, int64_t d3, int64_t d4, uint32_t upto) {
				const uint64_t a0 = abs_i64(d0), a1 = abs_i64(d1), a2 = abs_i64(d2), a3 = abs_i64(d3), a4 = abs_i64(d4);
				e0 += a0; if(a0 > 0x7fffffffull) over |= 1u;
				if(upto >= 1) { e1 += a1; if(a1 > 0x7fffffffull) over |= 2u; }
				if(upto >= 2) { e2 += a2; if(a2 > 0x7fffffffull) over |= 4u; }
				if(upto >= 3) { e3 += a3; if(a3 > 0x7fffffffull) over |= 8u; }
				if(upto >= 4) { e4 += a4; if(a4 > 0x7fffffffull) over |= 16u; }
			};
			auto full = [&](int i, uint32_t upto) {
				const int64_t v0 = SV(i), v1 = SV(i - 1), v2 = SV(i - 2), v3 = SV(i - 3), v4 = SV(i - 4);
				acc(v0, v0 - v1, v0 - 2 * v1 + v2, v0 - 3 * v1 + 3 * v2 - v3, v0 - 4 * v1 + 6 * v2 - 4 * v3 + v4, upto);
			};
			if(!lanes4) {
				// the sums run over the whole block, each order from the first sample it can be formed at
				for(uint32_t i = (uint32_t)tid; i < n; i += TPB) full((int)i, i < 4 ? i : 4u);
			}
			else {
				// fixed_intrin_avx2.c:219-340 restated literally: the four samples in front (orders 0..3 as far as they can
				// be formed), four lanes of q4 samples (history at l*q4, data at (l*n4)/4), then the n4 % 4 samples left
				if(tid < 4) full(tid, (uint32_t)tid);
				for(uint32_t l = 0; l < 4; l++) {
					const int hist = (int)(l * q4), start = (int)(((uint64_t)l * n4) / 4);
					for(uint32_t i = (uint32_t)tid; i < q4; i += TPB) {
						int64_t v[5];
#pragma unroll
						for(int j = 0; j < 5; j++) { const int m = (int)i - j; v[j] = SV(4 + (m >= 0 ? start + m : hist + m)); }
						acc(v[0], v[0] - v[1], v[0] - 2 * v[1] + v[2], v[0] - 3 * v[1] + 3 * v[2] - v[3], v[0] - 4 * v[1] + 6 * v[2] - 4 * v[3] + v[4], 4u);
					}
				}
				for(uint32_t i = 4 * q4 + (uint32_t)tid; i < n4; i += TPB) full((int)(4 + i), 4u);
			}
			over = block_reduce_or_u32(over, scratch, tid);
		}
		else if(!fwide || (n4 & 3) == 0) {
			for(uint32_t base = CHUNK * (uint32_t)tid; base < n; base += CHUNK * TPB) {
				int32_t x[CHUNK + 4];
#pragma unroll
				for(int k = 0; k < CHUNK + 4; k++) x[k] = stream ? (int32_t)SV((int)base - 4 + k) : sig[sigidx((int)base - 4 + k)];
				uint32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0;
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
			// fixed_intrin_avx2.c:57 with (n-4) % 4 != 0 (short last blocks): restated literally
			const uint32_t q = n4 / 4;
			for(uint32_t l = 0; l < 4; l++) {
				const int hist = (int)(l * q), start = (int)(((uint64_t)l * n4) / 4);
				for(uint32_t i = (uint32_t)tid; i < q; i += TPB) {
					int64_t v[5];
#pragma unroll
					for(int j = 0; j < 5; j++) { const int m = (int)i - j; v[j] = SV(4 + (m >= 0 ? start + m : hist + m)); }
					const int64_t d0 = v[0], d1 = v[0] - v[1], d2 = v[0] - 2 * v[1] + v[2], d3 = v[0] - 3 * v[1] + 3 * v[2] - v[3],
					              d4 = v[0] - 4 * v[1] + 6 * v[2] - 4 * v[3] + v[4];
					e0 += (uint64_t)(d0 < 0 ? -d0 : d0); e1 += (uint64_t)(d1 < 0 ? -d1 : d1); e2 += (uint64_t)(d2 < 0 ? -d2 : d2);
					e3 += (uint64_t)(d3 < 0 ? -d3 : d3); e4 += (uint64_t)(d4 < 0 ? -d4 : d4);
				}
			}
		}
		e0 = wave_reduce_add_u64(e0); e1 = wave_reduce_add_u64(e1); e2 = wave_reduce_add_u64(e2);
		e3 = wave_reduce_add_u64(e3); e4 = wave_reduce_add_u64(e4);
		if(lane == 0) { red[wave * 5 + 0] = e0; red[wave * 5 + 1] = e1; red[wave * 5 + 2] = e2; red[wave * 5 + 3] = e3; red[wave * 5 + 4] = e4; }
		__syncthreads();
		e0 = e1 = e2 = e3 = e4 = 0;
		for(int w = 0; w < TPB / 64; w++) { e0 += red[w * 5 + 0]; e1 += red[w * 5 + 1]; e2 += red[w * 5 + 2]; e3 += red[w * 5 + 3]; e4 += red[w * 5 + 4]; }
		uint32_t guess_fixed, invalid = 0;
		if(!flimit) {
			const uint64_t m34 = e3 < e4 ? e3 : e4, m234 = e2 < m34 ? e2 : m34, m1234 = e1 < m234 ? e1 : m234;
			if(e0 <= m1234) guess_fixed = 0;
			else if(e1 <= m234) guess_fixed = 1;
			else if(e2 <= m34) guess_fixed = 2;
			else if(e3 <= e4) guess_fixed = 3;
			else guess_fixed = 4;
		}
		else {
			// CHECK_ORDER_IS_VALID: first minimum among the orders whose residuals all fit; the others count 34 bits per
			// sample (fixed_intrin_avx2.c:172).  The plain C flavour behind 33-bit samples (fixed.c:360) gives 34 bits to
			// every order that is not a new minimum, too.
			const uint64_t ev[5] = {e0, e1, e2, e3, e4};
			uint64_t smallest = ~0ull;
			guess_fixed = 0; invalid = over;
			for(uint32_t k = 0; k < 5; k++) {
				if((over >> k) & 1u) continue;
				if(ev[k] < smallest) { guess_fixed = k; smallest = ev[k]; }
				else if(s64) invalid |= 1u << k;
			}
		}
		const bool rbps1_zero = !(invalid & 2u) && (e1 == 0 || fixed_rbps(e1, n4) == 0.0f);
		bool is_constant = false;
		if(!disable_constant && rbps1_zero) {
			uint32_t diff = 0;
			const int64_t first = SV(0);
			for(uint32_t i = (uint32_t)tid; i < n; i += TPB) { const uint64_t x = (uint64_t)(SV((int)i) ^ first); diff |= (uint32_t)x | (uint32_t)(x >> 32); }
			diff = block_reduce_or_u32(diff, scratch, tid);
			is_constant = diff == 0;
		}
		if(is_constant) { flags |= PREP_CONSTANT; constant = SV(0); }
		else if(P.max_lpc_order > 0) flags |= PREP_LPC;          // n > 4, so at least order 4 is possible
		const bool fixed_allowed = !is_constant && (!P.disable_fixed || (P.max_lpc_order == 0 && verbatim_bits == 0xffffffffu));
		fixed_order = fixed_allowed ? guess_fixed : 0;
		const uint64_t es[5] = {e0, e1, e2, e3, e4};
		if(tid < 64 && emit_fixed_candidates(P, &cands[fc * cstride], &valid[fc * cstride], es, n4, guess_fixed, fixed_allowed, sbps, tid, invalid)) flags |= PREP_FIXED_VALID;
	}
	else if(tid < 64) {
		// n <= 4: no fixed or LPC candidate at all
		for(uint32_t k = (uint32_t)tid; k < P.nfixed; k += 64) valid[fc * cstride + k] = 0;
	}
	// hand-off
	if(tid == 0) {
		ChanPrep pr;
		pr.which = which; pr.wasted = wasted; pr.sbps = sbps; pr.n = n; pr.flags = flags; pr.fixed_order = fixed_order;
		pr.constant = (int32_t)constant; pr.constant_hi = (int32_t)(constant >> 32); pr.verbatim_bits = verbatim_bits; pr.fmt = fmt; pr.handled = 0; pr.pad = 0;
		preps[fc] = pr;
	}
}

// ---------------------------------------------------------------------------------------------------------
// autoc_kernel
// ---------------------------------------------------------------------------------------------------------
// LDS tile of one wavefront: windowed floats of AT block samples + 16 samples of history, de-interleaved into four
// planes by (i mod 4).  With i = origin + 4*(e - AHE) + r, plane r element e.  Two copies: A at r*APS + e, and B
// shifted by one float, so that a pair (e, e+1) is an aligned 8-byte read in one of them whatever the parity of e.
constexpr int AT = 256;                 // new samples per tile = 32 chain steps
constexpr int AHE = 4;                  // history elements per plane (16 samples >= any lag)
constexpr int APS = 72;                 // plane stride in floats (AHE + AT/4 = 68, padded: planes 8 banks apart)
constexpr int ATILE = 8 * APS + 2;      // floats: copy A (4 planes), copy B (4 planes, +1)
constexpr int AHEAD = 32, ATAIL = 40;   // plain copies of d[0,32) and d[nd-40,nd) for the scalar head/tail code
__device__ __forceinline__ int aoffA(int r, int e) { return r * APS + e; }
__device__ __forceinline__ int aoffB(int r, int e) { return 4 * APS + r * APS + e + 1; }

struct AutocWave {
	float tile[ATILE + 6];
	float head[AHEAD];
	float tail[ATAIL];
	double acc4[MAX_ORDER][4];
};

// windowed sample i of a job (lpc.c:68 full block, lpc.c:82-94 partial window), 0 beyond the job's data
struct JobView {
	const int32_t *frame_pcm; const float *w; uint32_t C, which, wasted, n, nd;
	uint32_t full, part, dshift, i0;
	uint32_t side33;        // the side channel of a 32-bit stream: 33-bit values (lpc.c:75,96)
};
// v: the unshifted sample (64 bits wide only for side33)
__device__ __forceinline__ void job_fetch(const JobView &J, uint32_t i, int64_t &v, float &wt)
{
	v = 0; wt = 0.0f;
	if(i < J.nd) {
		uint32_t src, widx; bool on = true;
		if(J.full) { src = i; widx = i; }
		else {
			src = J.dshift + i;
			const bool hi = i >= J.i0 && i < J.i0 + J.part;
			if(hi) widx = J.n - J.part + (i - J.i0);
			else if(i < J.part) widx = i;
			else { widx = 0; on = false; }
		}
		if(on) {
			wt = J.w[widx];
			if(J.C == 2) {
				const int2 lr = ((const int2 *)J.frame_pcm)[src];
				v = J.which == 0 ? (int64_t)lr.x : J.which == 1 ? (int64_t)lr.y : J.which == 2 ? (((int64_t)lr.x + (int64_t)lr.y) >> 1) : J.side33 ? side64(lr) : (int64_t)(lr.x - lr.y);
			}
			else v = pick_channel(J.frame_pcm, J.C, src, J.which);
		}
	}
}
// (float)sample * window: one integer -> float rounding, one multiply.  A 33-bit sample goes through double, which
// holds it exactly, so that the conversion rounds once like the reference's cvtsi2ss does.
__device__ __forceinline__ float job_value(const JobView &J, int64_t v, float wt)
{
	if(J.side33) return (float)(double)(v >> J.wasted) * wt;
	return (float)(int32_t)(v >> J.wasted) * wt;
}

// one chain step of lpc_intrin_fma.c:46,61 on stream elements: x = d[i], d[i+4]; y = d[i-j], d[i+4-j]
#define STEP816(acc, xs, ys, u) acc += fma((double)(xs)[2 * (u)], (double)(ys)[2 * (u)], (double)(xs)[2 * (u) + 1] * (double)(ys)[2 * (u) + 1])

template <int DUMMY>
__global__ __launch_bounds__(TPB, AUTOC_WAVES_PER_SIMD) void autoc_kernel(const DevParams P, const int32_t *__restrict__ pcm, const float *__restrict__ windows,
                                                    const float *__restrict__ tail_windows, uint32_t nframes, uint32_t tail_n, uint32_t f_lo,
                                                    const JobTable *__restrict__ jt_main, const JobTable *__restrict__ jt_tail,
                                                    const ChanPrep *__restrict__ preps, double *__restrict__ autoc_out)
{
	__shared__ AutocWave sh[TPB / 64];
	const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
	AutocWave &W = sh[wave];
	const uint32_t njobs_main = P.max_jobs;
	const uint32_t item = blockIdx.x * (TPB / 64) + (uint32_t)wave;
	const uint32_t total = (nframes - f_lo) * P.ncand * njobs_main;       // frames [f_lo, nframes)
	if(item >= total) return;
	// longest jobs first within a frame-channel: jobs are enumerated full, halves, thirds ... already
	const uint32_t fc = f_lo * P.ncand + item / njobs_main, jb = item % njobs_main;
	const uint32_t f = fc / P.ncand;
	const bool is_tail = tail_n != 0 && f == nframes - 1;
	const JobTable *jt = is_tail ? jt_tail : jt_main;
	const ChanPrep pr = preps[fc];
	if(!(pr.flags & PREP_LPC) || jb >= jt->njobs) return;
	const uint32_t n = pr.n;
	const uint32_t max_lpc = P.max_lpc_order >= n ? n - 1 : P.max_lpc_order;
	const uint32_t lag = max_lpc + 1;
	const uint32_t variant = n <= 32 ? 0u : P.autoc_variant;      // 0: the plain loop of lpc.c:133-157
	const WindowJob jv = jt->jobs[jb];
	JobView J;
	J.frame_pcm = pcm + (size_t)f * P.blocksize * P.channels;
	J.w = (is_tail ? tail_windows : windows) + (size_t)jv.apod * n;
	J.C = P.channels; J.which = pr.which; J.wasted = pr.wasted; J.n = n; J.nd = jv.nd;
	J.full = jv.full; J.part = jv.part; J.dshift = jv.dshift; J.i0 = jv.i0;
	J.side33 = (P.bps == 32 && P.channels == 2 && pr.which == 3) ? 1u : 0u;
	const uint32_t nd = jv.nd;
	const uint32_t tail_lo = nd > (uint32_t)ATAIL ? nd - ATAIL : 0;
	double *out = autoc_out + ((size_t)fc * P.max_jobs + jb) * AUTOC_STRIDE;

	// plain head / tail copies
	if(lane < AHEAD) { int64_t v; float wt; job_fetch(J, (uint32_t)lane, v, wt); W.head[lane] = job_value(J, v, wt); }
	if(lane < ATAIL) { int64_t v; float wt; job_fetch(J, tail_lo + (uint32_t)lane, v, wt); W.tail[lane] = job_value(J, v, wt); }

	if(variant == 0 && n <= 32) {
		// lpc.c:133-157 (blocksize <= 32): the whole job sits in W.tail (tail_lo == 0)
		__builtin_amdgcn_wave_barrier();
		if((uint32_t)lane < lag) out[lane] = autoc_small(W.tail, nd, (uint32_t)lane);
		return;
	}
	if(variant == 0) {
		// lpc.c:133-157 with lag > 16 (max_lpc_order >= 16): autoc[c] = sum over s of d[s]*d[s+c] in increasing s, one lane
		// per lag.  Every product of two floats is exact in double, so the chain is a plain sequence of additions; the
		// samples stream through the tile as doubles, 256 at a time plus the 32 the largest lag looks ahead (zeros
		// beyond the job's data add nothing).
		double *td = (double *)W.tile;                       // 288 doubles
		double acc = 0.0;
		for(uint32_t t0 = 0; t0 < nd; t0 += 256) {
			__builtin_amdgcn_wave_barrier();
#pragma unroll
			for(int u = 0; u < 5; u++) {
				const uint32_t k = (uint32_t)lane + 64u * (uint32_t)u;
				if(k < 288) { int64_t v; float wt; job_fetch(J, t0 + k, v, wt); td[k] = (double)job_value(J, v, wt); }
			}
			__builtin_amdgcn_wave_barrier();
			if((uint32_t)lane < lag) {
				const double *py = td + lane;
#pragma unroll 8
				for(int sidx = 0; sidx < 256; sidx++) acc = fma(td[sidx], py[sidx], acc);
			}
		}
		if((uint32_t)lane < lag) out[lane] = acc;
		return;
	}
	const uint32_t L = variant;
	const uint32_t nb = (nd - L) / 8;
	const uint32_t j = (uint32_t)lane >> 2, l = (uint32_t)lane & 3;
	// this lane's operand streams inside a tile
	const int fl = (int)l - (int)j >= 0 ? 0 : -(((int)j - (int)l + 3) / 4);     // floor((l-j)/4)
	const int ry = (((int)l - (int)j) % 4 + 4) % 4;
	const int ey = AHE + fl;
	const float *xs = W.tile + aoffA((int)l, AHE);
	const float *ys = W.tile + ((ey & 1) ? aoffB(ry, ey) : aoffA(ry, ey));
	double acc = 0.0;
	const uint32_t npairs12 = nb > 2 ? ((nb - 3) & ~1u) / 2 + 1 : 0;            // lag-12 routine: steps taken two at a time
	const uint32_t ntiles = (nb + 31) / 32;

	// tile 0 history: d[L-16, L)
	if(lane < 16) {
		const int i = (int)L - 16 + lane;
		int64_t v = 0; float wt = 0.0f;
		if(i >= 0) job_fetch(J, (uint32_t)i, v, wt);
		const float d = job_value(J, v, wt);
		const int r = i & 3, e = (i - ((int)L - 16)) >> 2;       // L multiple of 4
		W.tile[aoffA(r, e)] = d; W.tile[aoffB(r, e)] = d;
	}
	// prefetch tile 0
	int64_t pv[4]; float pw[4];
#pragma unroll
	for(int u = 0; u < 4; u++) job_fetch(J, L + (uint32_t)lane + 64u * (uint32_t)u, pv[u], pw[u]);

	for(uint32_t t = 0; t < ntiles; t++) {
		const uint32_t tb = L + AT * t;
		if(t) {
			// carry the last 16 samples over as history (single wavefront: LDS operations execute in order)
			float hv = 0.0f;
			const int c = lane >> 4, r = (lane >> 2) & 3, h = lane & 3;
			if(lane < 32) hv = W.tile[c ? aoffB(r, AT / 4 + h) : aoffA(r, AT / 4 + h)];
			__builtin_amdgcn_wave_barrier();
			if(lane < 32) W.tile[c ? aoffB(r, h) : aoffA(r, h)] = hv;
		}
#pragma unroll
		for(int u = 0; u < 4; u++) {
			const float d = job_value(J, pv[u], pw[u]);
			const int r = lane & 3, e = AHE + (lane >> 2) + 16 * u;
			W.tile[aoffA(r, e)] = d; W.tile[aoffB(r, e)] = d;
		}
		if(t + 1 < ntiles) {
#pragma unroll
			for(int u = 0; u < 4; u++) job_fetch(J, tb + AT + (uint32_t)lane + 64u * (uint32_t)u, pv[u], pw[u]);
		}
		__builtin_amdgcn_wave_barrier();
		const uint32_t k0 = 32 * t;
		const uint32_t ksteps = nb - k0 < 32 ? nb - k0 : 32;
		if(j < lag) {
			if(variant != 12) {
				if(ksteps == 32) {
#pragma unroll
					for(int g = 0; g < 8; g++) {
						float x[8], y[8];
#pragma unroll
						for(int u = 0; u < 8; u++) { x[u] = xs[8 * g + u]; y[u] = ys[8 * g + u]; }
#pragma unroll
						for(int u = 0; u < 4; u++) STEP816(acc, x, y, u);
					}
				}
				else for(uint32_t kk = 0; kk < ksteps; kk++) STEP816(acc, xs + 2 * kk, ys + 2 * kk, 0);
			}
			else {
				// lpc_intrin_fma.c:54 (lag 12) as compiled: the 8-sample body unrolled x2 (acc += t1+t0 per 16 samples); for
				// lag 8 only, x*y0 + x*y2 factored into x*(y0+y2) across the two halves (y2 == x0 of the next half)
				uint32_t kk = 0;
				if(k0 + 32 <= 2 * npairs12) {
					if(j == 8) {
#pragma unroll
						for(int g = 0; g < 16; g++) {
							const float *px = xs + 4 * g, *py = ys + 4 * g;
							acc += fma((double)px[0], ((double)py[0] + (double)px[2]), (double)px[1] * ((double)py[1] + (double)px[3]));
						}
					}
					else {
#pragma unroll
						for(int g = 0; g < 16; g++) {
							const float *px = xs + 4 * g, *py = ys + 4 * g;
							const double t0 = fma((double)px[0], (double)py[0], (double)px[1] * (double)py[1]);
							const double t1 = fma((double)px[2], (double)py[2], (double)px[3] * (double)py[3]);
							acc += (t1 + t0);
						}
					}
					kk = 32;
				}
				else {
					for(; kk + 2 <= ksteps && k0 + kk + 2 <= 2 * npairs12; kk += 2) {
						const float *px = xs + 2 * kk, *py = ys + 2 * kk;
						if(j == 8) acc += fma((double)px[0], ((double)py[0] + (double)px[2]), (double)px[1] * ((double)py[1] + (double)px[3]));
						else {
							const double t0 = fma((double)px[0], (double)py[0], (double)px[1] * (double)py[1]);
							const double t1 = fma((double)px[2], (double)py[2], (double)px[3] * (double)py[3]);
							acc += (t1 + t0);
						}
					}
					for(; kk < ksteps; kk++) STEP816(acc, xs + 2 * kk, ys + 2 * kk, 0);
				}
			}
		}
		__builtin_amdgcn_wave_barrier();
	}
	if(j < lag) W.acc4[j][l] = acc;
	__builtin_amdgcn_wave_barrier();
	if((uint32_t)lane < lag) out[lane] = autoc_finish2(W.head, W.tail, tail_lo, nd, L, (uint32_t)lane, W.acc4[lane]);
}
#undef STEP816

// ---------------------------------------------------------------------------------------------------------
// model_kernel
// ---------------------------------------------------------------------------------------------------------
template <int MAXORD>
// A lane runs a whole Levinson-Durbin recursion and the quantisation searches behind it: a long dependent fp64 chain, latency bound.
// Left alone the compiler takes 189 registers for the 12-tap instance (two wavefronts per SIMD); bounded to 128 it spills nothing
// that matters and four wavefronts hide each other's latency: 0.087 -> 0.070 ms per 16384 frames (five: spills, 0.138;
// profiles/r03_z_model_waves_ab.txt).  The instances for more than 12 taps keep their registers (they would spill their tap arrays).
#ifndef MODEL_WAVES_PER_SIMD
#define MODEL_WAVES_PER_SIMD 4
#endif
__global__ __launch_bounds__(TPB, MAXORD <= 12 ? MODEL_WAVES_PER_SIMD : 2) void model_kernel(const DevParams P, uint32_t nframes, uint32_t tail_n,
                                                    const JobTable *__restrict__ jt_main, const JobTable *__restrict__ jt_tail,
                                                    const ChanPrep *__restrict__ preps, const double *__restrict__ autoc_in,
                                                    Candidate *__restrict__ cands, int *__restrict__ valid, uint32_t *__restrict__ nleft)
{
	if(nleft && blockIdx.x == 0 && threadIdx.x == 0) { nleft[0] = 0; nleft[1] = 0; }       // the lists of the evaluation kernels start empty
	// the { invc, logc } table of the log (flacgpu_log.h) in LDS: a lane looks it up a dozen times, each at its own index
	__shared__ uint64_t logtab[256];
	logtab[threadIdx.x] = flacgpu_log_tab[threadIdx.x];
	__syncthreads();
	const uint32_t na_main = P.max_analyses;
	const uint32_t id = blockIdx.x * TPB + threadIdx.x;
	if(na_main == 0 || id >= nframes * P.ncand * na_main) return;
	const uint32_t fc = id / na_main, a = id - fc * na_main;
	const uint32_t f = fc / P.ncand;
	const bool is_tail = tail_n != 0 && f == nframes - 1;
	const JobTable *jt = is_tail ? jt_tail : jt_main;
	const ChanPrep pr = preps[fc];
	const uint32_t cstride = P.ncslots, aslots = P.norders * P.nprec;
	Candidate *slots = &cands[(size_t)fc * cstride + P.nfixed + (size_t)a * aslots];
	int *vslots = &valid[(size_t)fc * cstride + P.nfixed + (size_t)a * aslots];
	if((pr.flags & PREP_LPC) && a < jt->nanalyses) {
		const uint32_t n = pr.n;
		const uint32_t max_lpc = P.max_lpc_order >= n ? n - 1 : P.max_lpc_order;
		const uint32_t lag = max_lpc + 1;
		const uint32_t jb = jt->an_job[a], rt = jt->an_root[a];
		const bool punch = jt->an_punch[a] != 0;
		const double *aj = autoc_in + ((size_t)fc * P.max_jobs + jb) * AUTOC_STRIDE;
		const double *ar = autoc_in + ((size_t)fc * P.max_jobs + rt) * AUTOC_STRIDE;
		double av[MAXORD + 1], rv[MAXORD + 1];
		// (every lag is loaded, wanted or not -- a row of the table has AUTOC_STRIDE > MAXORD entries -- so that the loads go out
		//  together: a load under a condition is followed by a wait for it, and there were 2 (MAXORD + 1) of those in a row)
		static_assert(MAXORD < (int)AUTOC_STRIDE, "a row holds every lag of this flavour");
#pragma unroll
		for(int j = 0; j <= MAXORD; j++) av[j] = aj[j];
#pragma unroll
		for(int j = 0; j <= MAXORD; j++) rv[j] = 0.0;
		if(punch) {                                                  // (one block under one condition: its loads still go out together)
#pragma unroll
			for(int j = 0; j <= MAXORD; j++) rv[j] = ar[j];
		}
#pragma unroll
		for(int j = 0; j <= MAXORD; j++) {
			double v = 0.0;
			if((uint32_t)j < lag) {
				v = av[j];
				// punch-out: root - partial for lags < max_order only; lag max_order keeps the partial's value
				// (stream_encoder.c:4339-4340,4370-4371)
				if(punch && (uint32_t)j < max_lpc) v = rv[j] - v;
			}
			av[j] = v;
		}
		lpc_model<MAXORD>(av, max_lpc, n, pr.sbps, P, slots, vslots, logtab);
	}
	else for(uint32_t s = 0; s < aslots; s++) vslots[s] = 0;
}

// ---------------------------------------------------------------------------------------------------------
// eval_kernel
// ---------------------------------------------------------------------------------------------------------
// "owner" layout of the block in LDS: lane L owns samples [L*S, (L+1)*S), S = n/64, stored with 16 samples of
// history in front at a stride that is odd in words, so that all lanes reading "their sample s" hit 64 different
// banks.  PACKED: two 16-bit samples per word (channels with <= 16 bits after the wasted-bits shift).
constexpr int OH = 16;                  // history samples per lane region (>= MAXORD)
__device__ __forceinline__ uint32_t owner_stride_words(uint32_t S, bool packed)
{
	const uint32_t w = packed ? (S + OH) / 2 : S + OH;
	return w | 1u;
}

// (rice_search_owner, the same search on 64-bit leaf sums: flacgpu_devfn.h since round 6 -- the deciding prep kernel runs it on wide samples)
// (rice_node_small / rice_search_nodes: flacgpu_devfn.h -- the prep kernel of the fixed-predictor presets runs them, too)
// |residual| without forming the residual.  With p = sum >> shift (arithmetic) the reference's residual is x - p.
// Start the wrapping tap sum at 2^31: (sum + 2^31) >> shift LOGICAL = p + 2^(31-shift) =: p + bias, exactly, and
// never negative; x + bias is not negative either as long as |x| <= bias (16-bit samples, shift <= 15).  Then
// |x - p| = |(x + bias) - (p + bias)| as UNSIGNED numbers: one v_sad_u32, which also accumulates.
// ---- residual magnitude of one lane's S samples, packed 16-bit samples -------------------------------------------
// NP coefficient pairs (order <= 2*NP): sample t of the piece window is predicted from the NP pairs in front of it;
// even t take their pairs straight from the LDS words (A), odd t from the words shifted by one sample (B).
// Same low 32 bits as the wrapping sum of lpc.c:321.  FIRST: piece 0, where lane 0 skips its `order` warm-up samples.
template <int NP, bool NARROW, bool FIRST, bool MASKED>
__device__ __forceinline__ void fir_piece_packed(const uint32_t *w /* word of sample (piece start - 2*NP) */, const uint32_t (&Q)[NP], uint32_t shift, uint32_t bias, uint32_t order,
                                                 uint32_t rem, bool lane0, int32_t sum0, uint32_t &acc32, uint64_t &acc64)
{
	constexpr int NW = NP + CHUNK / 2;
	uint32_t A[NW + 1], B[NW];
#pragma unroll
	for(int m = 0; m < NW; m++) A[m] = w[m];
	A[NW] = 0;
#pragma unroll
	for(int m = 0; m < NW; m++) B[m] = __builtin_amdgcn_alignbit(A[m + 1], A[m], 16);
#pragma unroll
	for(int s = 0; s < CHUNK; s++) {
		const int t = 2 * NP + s;
		uint32_t W[NP];
#pragma unroll
		for(int p = 0; p < NP; p++) W[p] = (t & 1) ? B[(t - 3) / 2 - p] : A[(t - 2) / 2 - p];
		const uint32_t pb = dot2_chain_lshr<NP>(W, Q, sum0, shift);
		const uint32_t xb = bias + (uint32_t)((t & 1) ? ((int32_t)A[(t - 1) / 2] >> 16) : (int32_t)(int16_t)(A[t / 2] & 0xffffu));
		uint32_t pbm = pb;
		if(FIRST && s < 2 * NP) { if(lane0 && (uint32_t)s < order) pbm = xb; }
		if(MASKED) { if((uint32_t)s >= rem) pbm = xb; }
		if(NARROW) acc32 = sad_u32(xb, pbm, acc32); else acc64 += sad_u32(xb, pbm, 0);
	}
}
template <int NP, bool NARROW>
__device__ __forceinline__ uint64_t fir_abs_packed(const uint32_t *reg, uint32_t S, uint32_t order, const uint32_t (&Q)[NP], int shift_, int lane)
{
	uint32_t acc32 = 0;
	uint64_t acc64 = 0;
	const uint32_t nfull = S / CHUNK;
	const uint32_t *w = reg + (OH - 2 * NP) / 2;
	const uint32_t shift = (uint32_t)shift_, bias = 0x80000000u >> shift;                               // 0 <= shift <= 15 (lpc.c:220-314)
	const int32_t sum0 = (int32_t)0x80000000;
	fir_piece_packed<NP, NARROW, true, false>(w, Q, shift, bias, order, CHUNK, lane == 0, sum0, acc32, acc64);      // S >= 16
#pragma unroll 1
	for(uint32_t c = 1; c < nfull; c++) fir_piece_packed<NP, NARROW, false, false>(w + (CHUNK / 2) * c, Q, shift, bias, order, CHUNK, false, sum0, acc32, acc64);
	if(S % CHUNK) fir_piece_packed<NP, NARROW, false, true>(w + (CHUNK / 2) * nfull, Q, shift, bias, order, S % CHUNK, false, sum0, acc32, acc64);
	return NARROW ? (uint64_t)acc32 : acc64;
}
template <int MAXORD, bool NARROW>
__device__ __forceinline__ uint64_t fir_abs_packed_dispatch(const uint32_t *reg, uint32_t S, uint32_t order, const int32_t *q, int shift, int lane)
{
	// pair p = (q[2p] for the nearer sample, q[2p+1] for the farther one); taps beyond `order` are zero
	uint32_t Q[MAXORD / 2];
#pragma unroll
	for(int p = 0; p < MAXORD / 2; p++) Q[p] = ((uint32_t)q[2 * p] << 16) | ((uint32_t)q[2 * p + 1] & 0xffffu);
	const uint32_t np = (order + 1) / 2;
#define FAP(N_) { uint32_t Qn[N_]; _Pragma("unroll") for(int p = 0; p < N_; p++) Qn[p] = Q[p]; return fir_abs_packed<N_, NARROW>(reg, S, order, Qn, shift, lane); }
	if(MAXORD >= 16 && np > 6) { if(np == 7) FAP(7) else FAP(8) }
	if(MAXORD >= 12 && np > 4) { if(np == 5) FAP(5) else FAP(6) }
	if(np > 2) { if(np == 3) FAP(3) else FAP(4) }
	if(np == 2) FAP(2)
	FAP(1)
#undef FAP
}

// ---- the same on 32-bit samples: NT taps (order <= NT); FMODE 0 v_mad_i32_i24, 1 32-bit multiplies (both lpc.c:321),
// 2 64-bit accumulate (lpc.c:582)
template <int NT, int FMODE, bool NARROW, bool FIRST, bool MASKED>
__device__ __forceinline__ void fir_piece_i32(const int32_t *w /* sample (piece start - NT) */, const int32_t (&q)[NT], int shift, uint32_t order,
                                              uint32_t rem, bool lane0, uint32_t &acc32, uint64_t &acc64)
{
	int32_t x[NT + CHUNK];
#pragma unroll
	for(int k = 0; k < NT + CHUNK; k++) x[k] = w[k];
#pragma unroll
	for(int s = 0; s < CHUNK; s++) {
		uint32_t av;
		if(FMODE == 2) {
			int64_t sum = 0;
#pragma unroll
			for(int jj = 0; jj < NT; jj++) sum += (int64_t)q[jj] * (int64_t)x[NT + s - 1 - jj];
			const int32_t r = (int32_t)((int64_t)x[NT + s] - (sum >> shift));
			av = (uint32_t)(r < 0 ? -(uint32_t)r : (uint32_t)r);
		}
		else {
			// |x - p| as an unsigned difference of the sign-flipped operands (order preserving): one v_sad_u32
			uint32_t pb;
			if(FMODE == 0) {
				if(NT == 16) pb = mad24_chain<8, true>(&x[NT + s - 9], &q[8], mad24_chain<8, false>(&x[NT + s - 1], &q[0], 0, 0), (uint32_t)shift);
				else pb = mad24_chain<NT == 16 ? 8 : NT, true>(&x[NT + s - 1], &q[0], 0, (uint32_t)shift);
			}
			else {
				uint32_t sum = 0;
#pragma unroll
				for(int jj = 0; jj < NT; jj++) sum += (uint32_t)q[jj] * (uint32_t)x[NT + s - 1 - jj];
				pb = (uint32_t)((int32_t)sum >> shift) ^ 0x80000000u;
			}
			av = sad_u32((uint32_t)x[NT + s] ^ 0x80000000u, pb, 0);
		}
		if(FIRST && s < NT) { if(lane0 && (uint32_t)s < order) av = 0; }
		if(MASKED) { if((uint32_t)s >= rem) av = 0; }
		if(NARROW) acc32 += av; else acc64 += av;
	}
}
template <int NT, int FMODE, bool NARROW>
__device__ __forceinline__ uint64_t fir_abs_i32(const uint32_t *reg, uint32_t S, uint32_t order, const int32_t *qall, int shift, int lane)
{
	int32_t q[NT];
#pragma unroll
	for(int jj = 0; jj < NT; jj++) q[jj] = qall[jj];
	uint32_t acc32 = 0;
	uint64_t acc64 = 0;
	const uint32_t nfull = S / CHUNK;
	const int32_t *w = (const int32_t *)reg + (OH - NT);
	fir_piece_i32<NT, FMODE, NARROW, true, false>(w, q, shift, order, CHUNK, lane == 0, acc32, acc64);
#pragma unroll 1
	for(uint32_t c = 1; c < nfull; c++) fir_piece_i32<NT, FMODE, NARROW, false, false>(w + CHUNK * c, q, shift, order, CHUNK, false, acc32, acc64);
	if(S % CHUNK) fir_piece_i32<NT, FMODE, NARROW, false, true>(w + CHUNK * nfull, q, shift, order, S % CHUNK, false, acc32, acc64);
	return NARROW ? (uint64_t)acc32 : acc64;
}
template <int MAXORD, int FMODE, bool NARROW>
__device__ __forceinline__ uint64_t fir_abs_i32_dispatch(const uint32_t *reg, uint32_t S, uint32_t order, const int32_t *q, int shift, int lane)
{
	if(MAXORD >= 16 && order > 12) return fir_abs_i32<MAXORD >= 16 ? 16 : MAXORD, FMODE, NARROW>(reg, S, order, q, shift, lane);
	if(MAXORD >= 12 && order > 8) return fir_abs_i32<MAXORD >= 12 ? 12 : MAXORD, FMODE, NARROW>(reg, S, order, q, shift, lane);
	if(order > 4) return fir_abs_i32<8, FMODE, NARROW>(reg, S, order, q, shift, lane);
	return fir_abs_i32<4, FMODE, NARROW>(reg, S, order, q, shift, lane);
}

// the rare candidates, plain code: the 64-bit accumulate (lpc.c:582) on a packed 16-bit channel, and -- check -- the
// overflow-checked flavour (lpc.c:832) on either layout: bad comes back true when a residual leaves (INT32_MIN, INT32_MAX]
template <int MAXORD>
__device__ uint64_t fir_abs_plain64(const uint32_t *reg, bool packed, uint32_t S, uint32_t order, const int32_t *q, int shift, int lane, bool check, bool &bad)
{
	const int16_t *x16 = (const int16_t *)reg + OH;        // x[s] = this lane's sample s, x[-1..-OH] its history
	const int32_t *x32 = (const int32_t *)reg + OH;
	uint64_t acc = 0;
	bad = false;
	for(uint32_t s = (lane == 0 ? order : 0); s < S; s++) {
		int64_t sum = 0;
#pragma unroll
		for(int j = 0; j < MAXORD; j++) sum += (int64_t)q[j] * (int64_t)(packed ? (int32_t)x16[(int)s - 1 - j] : x32[(int)s - 1 - j]);
		const int64_t v = (int64_t)(packed ? (int32_t)x16[s] : x32[s]) - (sum >> shift);
		if(check && (v <= (int64_t)INT32_MIN || v > (int64_t)INT32_MAX)) bad = true;
		const int32_t r = (int32_t)v;
		acc += (uint32_t)(r < 0 ? -(uint32_t)r : (uint32_t)r);
	}
	return acc;
}

// One wavefront evaluates one residual candidate on the owner layout; requires n == 64*S, S >= 16, max_po <= 6.
template <int MAXORD>
__device__ __forceinline__ uint32_t eval_candidate_owner(const uint32_t *reg /* this lane's region */, bool packed, uint32_t S, uint32_t n, uint32_t order, const int32_t *q, int shift,
                                         uint32_t wide /* Candidate::wide */, uint32_t sbps, uint32_t rice_limit, uint32_t max_po, uint32_t min_po, const uint32_t *divtab,
                                         uint8_t *kout, uint32_t *best_po_out, int lane)
{
	const uint32_t psize = n >> max_po;
	const bool narrow = (sbps + 4) < (32 - ilog2_u32(psize));               // stream_encoder.c:4814-4817
	uint64_t v;
	bool bad = false;
	if(wide == 2) {
		v = fir_abs_plain64<MAXORD>(reg, packed, S, order, q, shift, lane, true, bad);
		if(__any((int)bad)) return 0xffffffffu;                         // no candidate (stream_encoder.c:4603-4604)
		if(narrow) v = (uint32_t)v;
	}
	else if(packed) {
		if(!wide) v = narrow ? fir_abs_packed_dispatch<MAXORD, true>(reg, S, order, q, shift, lane) : fir_abs_packed_dispatch<MAXORD, false>(reg, S, order, q, shift, lane);
		else { v = fir_abs_plain64<MAXORD>(reg, true, S, order, q, shift, lane, false, bad); if(narrow) v = (uint32_t)v; }
	}
	else {
		const int fmode = fir_mode(wide, sbps);
		if(fmode == 0) v = narrow ? fir_abs_i32_dispatch<MAXORD, 0, true>(reg, S, order, q, shift, lane) : fir_abs_i32_dispatch<MAXORD, 0, false>(reg, S, order, q, shift, lane);
		else if(fmode == 1) v = fir_abs_i32_dispatch<MAXORD, 1, false>(reg, S, order, q, shift, lane);     // > 24-bit samples are never "narrow"
		else v = fir_abs_i32_dispatch<MAXORD, 2, false>(reg, S, order, q, shift, lane);
		if(fmode != 0 && narrow) v = (uint32_t)v;
	}
	if(!__any((int)(v >= (1u << 23))))
		return rice_search_nodes((uint32_t)v, 6 - max_po, n, order, max_po, min_po, rice_limit, divtab, kout, best_po_out, lane);
	return rice_search_owner(v, narrow, 6 - max_po, n, order, max_po, min_po, rice_limit, divtab, kout, best_po_out, lane);
}

// ---- workgroup state of the evaluation kernel ---------------------------------------------------------------------
constexpr int EVAL_CPW_MAX = 4;          // candidate channels of one frame served by one workgroup
struct EvalChan {                         // per channel, written once by one lane, then read by everybody
	ChanPrep pr;
	uint32_t nan, any, mine, packed, stride, ctx, pad[2];
};
struct EvalSmall {
	uint32_t divtab[(MAX_PO + 1) * (MAX_ORDER + 1)];
	uint64_t pob[EVAL_MAX_WAVES][MAX_PO + 1];
	uint32_t wbest_bits[EVAL_CPW_MAX][EVAL_MAX_WAVES], wbest_ci[EVAL_CPW_MAX][EVAL_MAX_WAVES], wbest_po[EVAL_CPW_MAX][EVAL_MAX_WAVES];
	EvalChan ch[EVAL_CPW_MAX];
};
// LDS bytes of one lane-owner channel image: 64 lane regions of S samples + OH history at an odd word stride
__host__ __device__ inline uint32_t owner_chan_bytes(uint32_t N, bool packed)
{
	const uint32_t S = N / 64, w = (packed ? (S + OH) / 2 : S + OH) | 1u;
	return (64 * w * 4 + (CHUNK + MAX_ORDER) * 4 + 15u) & ~15u;
}
__host__ __device__ inline bool owner_possible(const DevParams &P) { return P.blocksize % 64 == 0 && P.blocksize / 64 >= (uint32_t)OH && P.max_lpc_order <= (uint32_t)OH && !P.wide_samples && !P.stream_sig; }
// candidate records are staged in LDS next to the channel image when there are few of them (every preset); the wide
// searches (-e, -p: hundreds of slots per channel) read them from global memory and keep only the valid flags in LDS
__host__ __device__ inline bool eval_cands_in_lds(const DevParams &P) { return P.ncslots <= 48 && !(P.tune_flags & 1u); }
__host__ __device__ inline uint32_t eval_cand_bytes(const DevParams &P)
{
	return (eval_cands_in_lds(P) ? P.ncslots * (uint32_t)sizeof(Candidate) : 0u) + ((P.ncslots * 4 + 15u) & ~15u);
}
// worst-case subframe width of candidate channel `cand` (the side channel carries one bit more)
__host__ __device__ inline uint32_t cand_max_sbps(const DevParams &P, uint32_t cand)
{
	const bool side = (P.ms_mode == 1 && cand == 3) || (P.ms_mode == 2 && cand == 1);
	return P.bps + (side ? 1 : 0);
}
// [channel contexts: image | candidate records | valid flags] [wsums] [kbestw] [kcandw] [EvalSmall]
struct EvalLayout { uint32_t ctx_bytes, wsums, kbestw, kcandw, small, total; };
__host__ __device__ inline EvalLayout eval_layout(const DevParams &P, uint32_t waves, uint32_t cpw, bool generic)
{
	EvalLayout L;
	uint32_t o = 0;
	if(generic) o = cpw * (P.sig_bytes + eval_cand_bytes(P));
	else if(!owner_possible(P)) o = 0;      // VARIANT 0 then only decides the channels that have no residual candidate: no image
	else {
		// the most demanding group of cpw consecutive candidate channels
		const bool s_even = (P.blocksize / 64) % 2 == 0;
		for(uint32_t g = 0; g * cpw < P.ncand; g++) {
			uint32_t t = 0;
			for(uint32_t c = g * cpw; c < (g + 1) * cpw && c < P.ncand; c++) t += owner_chan_bytes(P.blocksize, s_even && cand_max_sbps(P, c) <= 16) + eval_cand_bytes(P);
			if(t > o) o = t;
		}
	}
	L.ctx_bytes = o;
	L.wsums = o;  o += generic ? waves * (2u << P.max_po) * 8 : 0;
	L.kbestw = o; o += cpw * waves * 2 * (1u << P.max_po);
	L.kcandw = o; o += generic && P.max_po > 6 ? waves * (2u << P.max_po) : 0;
	o = (o + 15u) & ~15u;
	L.small = o;  o += (uint32_t)sizeof(EvalSmall);
	L.total = (o + 15u) & ~15u;
	return L;
}

// VARIANT selects which (frame, channel)s a launch serves (the others are left alone), so that each flavour of the
// residual evaluation gets its own register allocation:
//   0  owner layout: packed 16-bit samples + dot2 FIR, or 32-bit samples (17..25-bit channels), chosen per channel;
//      also every channel that has no residual candidate at all
//   2  any other block length / partition order: generic chunked evaluation with LDS partition sums
// A workgroup serves cpw consecutive candidate channels of one frame with nwaves wavefronts: the work items
// (channel, candidate) are dealt to the wavefronts round-robin, rotating the channel from round to round so that the
// expensive channel (side: 17-bit samples, no dot2) is spread evenly.  nwaves is a multiple of 4 wherever possible:
// a 5-wavefront workgroup puts two wavefronts on one SIMD and the dispatcher then fits only 2 such workgroups per CU.
// (the body of the kernel as a function of the (frame, channel group) it serves: eval_kernel runs it once per workgroup,
//  eval_list_kernel in a loop over the channels evalg_kernel left behind)
template <int MAXORD, int VARIANT>
__device__ __forceinline__ void eval_body(const DevParams &P, const int32_t *__restrict__ chan, uint32_t nframes, uint32_t tail_n, uint32_t cpw,
                                          const JobTable *__restrict__ jt_main, const JobTable *__restrict__ jt_tail,
                                          const ChanPrep *__restrict__ preps, const Candidate *__restrict__ cands,
                                          const int *__restrict__ valid, SubDecision *__restrict__ decisions, unsigned long long *__restrict__ dbg,
                                          uint32_t prefetch_ahead, uint32_t f, uint32_t grp, unsigned char *smem)
{
	const int tid = (int)threadIdx.x, lane = tid & 63;
	const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane(tid >> 6);
#define STAMP(k) do { if(dbg && tid == 0) dbg[(size_t)blockIdx.x * 16 + (k)] = (unsigned long long)clock64(); } while(0)
	const unsigned long long t_start = dbg ? (unsigned long long)clock64() : 0ull;
	const uint32_t nthreads = blockDim.x, nwaves = nthreads >> 6;
	const uint32_t N = P.blocksize;
	const uint32_t ngrp = P.ncand / cpw;
	uint32_t pf_tmp = 0;                  // destination of the prefetch load (see below)
	const size_t fc0 = (size_t)f * P.ncand + (size_t)grp * cpw;
	const bool is_tail = tail_n != 0 && f == nframes - 1;
	const JobTable *jt = is_tail ? jt_tail : jt_main;
	const uint32_t n = is_tail ? tail_n : N;
	const uint32_t cstride = P.ncslots, aslots = P.norders * P.nprec;
	const bool cands_lds = eval_cands_in_lds(P);

	const EvalLayout LY = eval_layout(P, nwaves, cpw, VARIANT == 2);
	uint64_t *wsums_all = (uint64_t *)(smem + LY.wsums);
	uint8_t *kbestw_all = smem + LY.kbestw;
	uint8_t *kcandw_all = smem + LY.kcandw;
	EvalSmall *sh = (EvalSmall *)(smem + LY.small);
	const uint32_t kstride = 1u << P.max_po;

	// partition order limits of the frame (stream_encoder.c:3759-3761)
	uint32_t frame_max_po = 0;
	{ uint32_t b = n; while(!(b & 1)) { frame_max_po++; b >>= 1; } if(frame_max_po > 15) frame_max_po = 15; }
	frame_max_po = umin32(frame_max_po, P.max_po);
	const uint32_t frame_min_po = umin32(P.min_po, frame_max_po);
	const uint32_t S = n / 64;
	// (a short last block whose lane runs are odd while the nominal ones are even would need a wider image than the
	// launch reserved: it takes the generic path)
	const bool owner = MAXORD <= OH && owner_possible(P) && (n % 64 == 0) && S >= (uint32_t)OH && frame_max_po <= 6 && (S % 2 == 0 || (N / 64) % 2 == 1);
	const uint32_t cand_bytes = eval_cand_bytes(P), cand_valid_off = cands_lds ? P.ncslots * (uint32_t)sizeof(Candidate) : 0u;

	// ---- channel facts ---------------------------------------------------------------------------------------
	if(tid < (int)cpw) {
		EvalChan &E = sh->ch[tid];
		const ChanPrep pr = preps[fc0 + (size_t)tid];
		E.pr = pr;
		E.nan = P.nfixed + ((pr.flags & PREP_LPC) ? jt->nanalyses * aslots : 0);       // candidate slots of this channel
		E.any = (!(pr.flags & PREP_CONSTANT) && ((pr.flags & PREP_FIXED_VALID) || E.nan > P.nfixed)) ? 1u : 0u;
		const int kind = !E.any ? 0 : owner ? 0 : 2;
		E.mine = (kind == VARIANT && pr.handled != EVG_HANDLED) ? 1u : 0u;       // (evalg_kernel ran first and took what it could)
		E.packed = (VARIANT == 0 && pr.fmt && (S % 2 == 0)) ? 1u : 0u;
		E.stride = owner_stride_words(S, E.packed != 0);
	}
	if(tid < EVAL_CPW_MAX * EVAL_MAX_WAVES) { (&sh->wbest_bits[0][0])[tid] = 0xffffffffu; (&sh->wbest_ci[0][0])[tid] = 0xffffffffu; (&sh->wbest_po[0][0])[tid] = 0; }
	__syncthreads();
	STAMP(6);
	uint32_t nmine = 0, nwork = 0;
	{
		uint32_t o = 0;
		for(uint32_t c = 0; c < cpw; c++) {
			if(tid == 0) sh->ch[c].ctx = o;
			if(sh->ch[c].mine) {
				nmine++;
				if(sh->ch[c].any) { nwork++; o += (VARIANT == 2 ? P.sig_bytes : owner_chan_bytes(n, sh->ch[c].packed != 0)) + cand_bytes; }
			}
		}
	}
	if(nmine == 0) return;
	if(dbg && tid == 0) { dbg[(size_t)blockIdx.x * 16] = t_start; dbg[(size_t)blockIdx.x * 16 + 8] = (unsigned long long)clock64(); dbg[(size_t)blockIdx.x * 16 + 9] = (unsigned long long)VARIANT + 1; }

	if(nwork) {
		for(uint32_t t = (uint32_t)tid; t < (MAX_PO + 1) * (MAX_ORDER + 1); t += nthreads) {
			const uint32_t po = t / (MAX_ORDER + 1), o = t - po * (MAX_ORDER + 1);
			const uint32_t ps = n >> po;
			sh->divtab[t] = ps > o ? 0x40000u / (ps - o) : 0;
		}
		__syncthreads();         // ctx offsets visible
		STAMP(7);
		// ---- candidate records and the channel signals into LDS -------------------------------------------------------
		for(uint32_t c = 0; c < cpw; c++) {
			const EvalChan &E = sh->ch[c];
			if(!E.mine || !E.any) continue;
			const size_t fc = fc0 + c;
			unsigned char *ctx = smem + E.ctx;
			const uint32_t img_bytes = VARIANT == 2 ? P.sig_bytes : owner_chan_bytes(n, E.packed != 0);
			{
				const uint32_t *src = (const uint32_t *)(cands + fc * cstride);
				uint32_t *dst = (uint32_t *)(ctx + img_bytes);
				if(cands_lds) for(uint32_t t = (uint32_t)tid; t < E.nan * (uint32_t)(sizeof(Candidate) / 4); t += nthreads) dst[t] = src[t];
				int *vd = (int *)(ctx + img_bytes + cand_valid_off);
				for(uint32_t t = (uint32_t)tid; t < E.nan; t += nthreads) vd[t] = valid[fc * cstride + t];
			}
			if(c < 2) STAMP(10 + 2 * c);
			const uint32_t *src = (const uint32_t *)(chan + fc * (size_t)P.chan_stride);       // planar channel, already shifted (ChanPrep::fmt)
			const uint32_t srcfmt = E.pr.fmt;
			if(VARIANT != 2) {
				uint32_t *sigw = (uint32_t *)ctx;
				const bool packed = E.packed != 0;
				const uint32_t stride = E.stride;
				// zero lane 0's history
				if(tid < OH) { if(packed) { if(tid < OH / 2) sigw[tid] = 0; } else sigw[tid] = 0; }
				const bool spow2 = (S & (S - 1)) == 0;
				const uint32_t slog = ilog2_u32(S);
				const uint32_t nvec = srcfmt ? (n + 7) / 8 : (n + 3) / 4;       // 16-byte pieces: 8 packed / 4 plain samples each
				for(uint32_t m = (uint32_t)tid; m < nvec; m += nthreads) {
					const uint4 pv = ((const uint4 *)src)[m];
					const uint32_t wv[4] = {pv.x, pv.y, pv.z, pv.w};
					if(srcfmt && packed) {
						// word = two samples; S is even, so a pair never straddles two lanes
#pragma unroll
						for(int k = 0; k < 4; k++) {
							const uint32_t i = 8 * m + 2 * (uint32_t)k;
							const uint32_t Lo = spow2 ? i >> slog : i / S, s = i - Lo * S;
							sigw[Lo * stride + (OH + s) / 2] = wv[k];
							if(s + OH >= S && Lo + 1 < 64) sigw[(Lo + 1) * stride + (s + OH - S) / 2] = wv[k];
						}
					}
					else {
						// 32-bit lane regions (from either source format)
#pragma unroll
						for(int k = 0; k < 8; k++) {
							if(!srcfmt && k >= 4) break;
							const uint32_t i = (srcfmt ? 8 : 4) * m + (uint32_t)k;
							const int32_t v = srcfmt ? ((k & 1) ? ((int32_t)wv[k >> 1] >> 16) : (int32_t)(int16_t)(wv[k >> 1] & 0xffffu)) : (int32_t)wv[k & 3];
							const uint32_t Lo = spow2 ? i >> slog : i / S, s = i - Lo * S;
							sigw[Lo * stride + OH + s] = (uint32_t)v;
							if(s + OH >= S && Lo + 1 < 64) sigw[(Lo + 1) * stride + (s + OH - S)] = (uint32_t)v;
						}
					}
				}
			}
			else if(P.stream_sig) { /* the candidates read the plane itself */ }
			else if(srcfmt == 2) {
				// 33-bit side channel: 64-bit samples, same row layout
				int64_t *sig = (int64_t *)ctx;
				if(tid < 32) sig[sigidx(tid - 32)] = 0;
				const uint32_t nround = ((n + 15u) & ~15u) + 16u;
				for(uint32_t i = n + (uint32_t)tid; i < nround; i += nthreads) sig[sigidx((int)i)] = 0;
				for(uint32_t i = (uint32_t)tid; i < n; i += nthreads) sig[sigidx((int)i)] = ((const int64_t *)src)[i];
			}
			else {
				int32_t *sig = (int32_t *)ctx;
				if(tid < 32) sig[sigidx(tid - 32)] = 0;
				const uint32_t nround = ((n + 15u) & ~15u) + 16u;
				for(uint32_t i = n + (uint32_t)tid; i < nround; i += nthreads) sig[sigidx((int)i)] = 0;
				if(srcfmt) for(uint32_t i = (uint32_t)tid; i < n; i += nthreads) sig[sigidx((int)i)] = (int32_t)((const int16_t *)src)[i];
				else for(uint32_t i = (uint32_t)tid; i < n; i += nthreads) sig[sigidx((int)i)] = (int32_t)src[i];
			}
		}
		__syncthreads();
		STAMP(1);
		// ---- the inputs of the workgroup that will take this one's place, pulled into this XCD's L2 -------------------------------
		// A workgroup starts with two dependent round trips to HBM (channel records; then candidate records and planar samples)
		// before its first FIR.  Workgroups reach an XCD in launch order, prefetch_ahead of them at a time: while this one
		// computes, it touches every 128-byte line its successor on the same XCD (block index + 8 * prefetch_ahead) will read --
		// one load per thread, result never used -- so that the successor's round trips end in the L2.  No extra HBM traffic: a
		// line is fetched once, by whoever asks first.
		if(VARIANT == 0 && prefetch_ahead) {
			const uint32_t bp = blockIdx.x + 8u * prefetch_ahead;
			if(bp < gridDim.x) {
				uint32_t fp, gp;
				map_block(bp, nframes, ngrp, fp, gp);
				const size_t fcp = (size_t)fp * P.ncand + (size_t)gp * cpw;
				const uint32_t per = nthreads / cpw, c = (uint32_t)tid / per, t = (uint32_t)tid - c * per;       // threads of channel c
				if(c < cpw) {
					const size_t fc = fcp + c;
					const uint32_t plane_lines = (N * 2 + 127) / 128, cand_lines = (cstride * (uint32_t)sizeof(Candidate) + 127) / 128, valid_lines = (cstride * 4 + 127) / 128;
					const unsigned char *a = nullptr;
					if(t < plane_lines) a = (const unsigned char *)(chan + fc * (size_t)P.chan_stride) + (size_t)t * 128;
					else if(t < plane_lines + cand_lines) a = (const unsigned char *)(cands + fc * cstride) + (size_t)(t - plane_lines) * 128;
					else if(t < plane_lines + cand_lines + valid_lines) a = (const unsigned char *)(valid + fc * cstride) + (size_t)(t - plane_lines - cand_lines) * 128;
					else if(t == plane_lines + cand_lines + valid_lines) a = (const unsigned char *)(preps + fc);
					if(a) asm volatile("global_load_dword %0, %1, off" : "=v"(pf_tmp) : "v"(a) : "memory");
				}
			}
		}

		// ---- work items (candidate, channel): one wavefront each ---------------------------------------------------------
		{
			uint64_t *wsums = wsums_all + (size_t)wave * (2u << P.max_po);
			uint8_t *kcw = kcandw_all + (size_t)wave * (2u << P.max_po);
			uint8_t *ktmp = kbestw_all + ((size_t)wave * 2 + 1) * kstride;           // scratch of this wavefront (slot of channel 0)
			const uint32_t ncmax = P.nfixed + jt->nanalyses * aslots;
			const bool rotate = nwaves % cpw == 0;
			for(uint32_t item = wave, k = 0; item < cpw * ncmax; item += nwaves, k++) {
				const uint32_t ci = item / cpw;
				const uint32_t c = (item - ci * cpw + (rotate ? k : 0)) % cpw;
				const EvalChan &E = sh->ch[c];
				if(!E.mine || !E.any || ci >= E.nan) continue;
				unsigned char *ctx = smem + E.ctx;
				const uint32_t img_bytes = VARIANT == 2 ? P.sig_bytes : owner_chan_bytes(n, E.packed != 0);
				const Candidate *cd = (cands_lds ? (const Candidate *)(ctx + img_bytes) : cands + (fc0 + c) * cstride) + ci;
				if(!((const int *)(ctx + img_bytes + cand_valid_off))[ci]) continue;
				const uint32_t order = cd->order, sbps = E.pr.sbps, hdr = 8 + E.pr.wasted;
				uint32_t po = 0, rbits;
				if constexpr(VARIANT == 0 && MAXORD > OH) rbits = 0;      // predictors of more than 16 taps are evaluated by VARIANT 2
				else if constexpr(VARIANT == 0)
					rbits = eval_candidate_owner<MAXORD>((const uint32_t *)ctx + (uint32_t)lane * E.stride, E.packed != 0, S, n, order, cd->q, cd->shift, cd->wide, sbps, P.rice_limit,
					                                     frame_max_po, frame_min_po, sh->divtab, ktmp, &po, lane);
				else {
					int32_t q[MAXORD];
#pragma unroll
					for(int jj = 0; jj < MAXORD; jj++) q[jj] = cd->q[jj];
					SigRef sref;
					if(P.stream_sig) { sref.p = (const void *)(chan + (fc0 + c) * (size_t)P.chan_stride); sref.kind = 2; }
					else { sref.p = (const void *)ctx; sref.kind = E.pr.fmt == 2 ? 1u : 0u; }
					sref.fmt = E.pr.fmt; sref.n = n;
					rbits = eval_candidate_wave<MAXORD>(wsums, kcw, sh->pob[wave], ktmp, sh->divtab, sref, n, order, q, cd->shift,
					                                    cd->wide, sbps, P, frame_max_po, frame_min_po, &po, lane);
				}
				const uint32_t est = ci < P.nfixed ? sat_add_u32(hdr + order * sbps, rbits)
				                             : sat_add_u32(hdr + 4 + 5 + order * (cd->precision + sbps), rbits);
				// a wavefront meets the candidates of a channel in increasing order; strict <: the earlier candidate keeps a
				// tie (stream_encoder.c:4191,4266)
				if(est > 0 && est < sh->wbest_bits[c][wave]) {
					__builtin_amdgcn_wave_barrier();
					if(lane == 0) { sh->wbest_bits[c][wave] = est; sh->wbest_ci[c][wave] = ci; sh->wbest_po[c][wave] = po; }
					uint8_t *kbw = kbestw_all + (((size_t)c * nwaves + wave) * 2) * kstride;
					for(uint32_t p = (uint32_t)lane; p < (1u << po); p += 64) kbw[p] = ktmp[p];
					__builtin_amdgcn_wave_barrier();
				}
				if(item == wave) STAMP(2);
			}
		}
		__syncthreads();
		STAMP(4);
	}

	// ---- decisions: first minimum in the reference's evaluation order; one wavefront per channel ---------------------------
	for(uint32_t c = wave; c < cpw; c += nwaves) {
		const EvalChan &E = sh->ch[c];
		if(!E.mine) continue;
		const ChanPrep &pr = E.pr;
		const uint32_t sbps = pr.sbps, wasted = pr.wasted, hdr = 8 + wasted;
		SubDecision *dec = decisions + fc0 + c;
		uint32_t best_type = 1, best_order = 0, best_po = 0, best_precision = 0, best_ci = 0, best_wave = 0;
		int32_t best_shift = 0, best_constant = 0, best_constant_hi = 0;
		uint32_t best_bits = pr.verbatim_bits;
		if(pr.flags & PREP_CONSTANT) {
			const uint32_t bits = hdr + sbps;
			if(bits < best_bits) { best_type = 0; best_constant = pr.constant; best_constant_hi = pr.constant_hi; best_bits = bits; }
		}
		const Candidate *mycands = nullptr;
		if(E.any) {
			mycands = cands_lds ? (const Candidate *)(smem + E.ctx + (VARIANT == 2 ? P.sig_bytes : owner_chan_bytes(n, E.packed != 0))) : cands + (fc0 + c) * cstride;
			uint32_t cb = 0xffffffffu, cci = 0xffffffffu, cw = 0;
			for(uint32_t w = 0; w < nwaves; w++) {
				const uint32_t b = sh->wbest_bits[c][w], ci = sh->wbest_ci[c][w];
				if(ci != 0xffffffffu && (b < cb || (b == cb && ci < cci))) { cb = b; cci = ci; cw = w; }
			}
			if(cci != 0xffffffffu && cb < best_bits) {
				best_bits = cb; best_ci = cci; best_wave = cw; best_po = sh->wbest_po[c][cw];
				best_type = cci < P.nfixed ? 2 : 3;
				best_order = mycands[cci].order; best_precision = mycands[cci].precision; best_shift = mycands[cci].shift;
			}
		}
		if(best_bits == 0xffffffffu) { best_type = 1; best_bits = hdr + n * sbps; }   // stream_encoder.c:4281
		uint32_t rice2 = 0;
		if(best_type >= 2) {
			const uint8_t *kb = kbestw_all + (((size_t)c * nwaves + best_wave) * 2) * kstride;
			uint32_t big = 0;
			for(uint32_t p = (uint32_t)lane; p < (1u << best_po); p += 64) {
				const uint8_t kk = kb[p];
				dec->params[p] = kk;
				if(kk >= 15) big = 1;
			}
			rice2 = __any((int)big) ? 1u : 0u;                         // stream_encoder.c:4786-4791
		}
		if(lane < MAX_ORDER) dec->q[lane] = best_type == 3 && lane < MAXORD ? mycands[best_ci].q[lane] : 0;
		if(lane == 0) {
			dec->bits = best_bits;
			dec->type = (uint8_t)best_type; dec->order = (uint8_t)best_order; dec->wasted = (uint8_t)wasted;
			dec->po = (uint8_t)best_po; dec->rice2 = (uint8_t)rice2; dec->precision = (uint8_t)best_precision;
			dec->shift = (int8_t)best_shift; dec->which = (uint8_t)pr.which;
			dec->constant = best_constant; dec->constant_hi = best_constant_hi; dec->fmt = pr.fmt;
		}
	}
	STAMP(5);
#undef STAMP
	if(VARIANT == 0 && prefetch_ahead) asm volatile("s_waitcnt vmcnt(0)" : "+v"(pf_tmp));      // (the prefetch's destination register stays reserved until its data is back)
}
template <int MAXORD, int VARIANT>
__global__ __launch_bounds__(EVAL_MAX_WAVES * 64, VARIANT == 0 ? EVAL_WAVES_PER_SIMD : 2) void eval_kernel(const DevParams P, const int32_t *__restrict__ chan, uint32_t nframes, uint32_t tail_n, uint32_t cpw,
                                                                   const JobTable *__restrict__ jt_main, const JobTable *__restrict__ jt_tail,
                                                                   const ChanPrep *__restrict__ preps, const Candidate *__restrict__ cands,
                                                                   const int *__restrict__ valid, SubDecision *__restrict__ decisions, unsigned long long *__restrict__ dbg,
                                                                   uint32_t prefetch_ahead)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	uint32_t f, grp;
	map_block(blockIdx.x, nframes, P.ncand / cpw, f, grp);
	eval_body<MAXORD, VARIANT>(P, chan, nframes, tail_n, cpw, jt_main, jt_tail, preps, cands, valid, decisions, dbg, prefetch_ahead, f, grp, smem);
}
// the channels of the short last block go on the list, too (host-side launch: their prep kernel is the general one)
__global__ void list_append_kernel(uint32_t fc0, uint32_t count, uint32_t *__restrict__ left, uint32_t *__restrict__ nleft)
{
	if(threadIdx.x == 0) { const uint32_t at = atomicAdd(nleft, count); for(uint32_t i = 0; i < count; i++) left[at + i] = fc0 + i; }
}
// The channels evalg_kernel (flacgpu_evalg.hip) did not decide -- it lists them -- one at a time, a fixed grid looping over the
// list: normally the list is empty or short, and a grid sized for the worst case would cost more to dispatch (88 us per 16384
// frames) than the work it finds.
template <int MAXORD>
__global__ __launch_bounds__(EVAL_MAX_WAVES * 64, EVAL_WAVES_PER_SIMD) void eval_list_kernel(const DevParams P, const int32_t *__restrict__ chan, uint32_t nframes, uint32_t tail_n,
                                                                   const JobTable *__restrict__ jt_main, const JobTable *__restrict__ jt_tail,
                                                                   const ChanPrep *__restrict__ preps, const Candidate *__restrict__ cands,
                                                                   const int *__restrict__ valid, SubDecision *__restrict__ decisions,
                                                                   const uint32_t *__restrict__ left, const uint32_t *__restrict__ nleft)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const uint32_t count = *nleft;
	for(uint32_t e = blockIdx.x; e < count; e += gridDim.x) {
		const uint32_t fc = left[e];
		eval_body<MAXORD, 0>(P, chan, nframes, tail_n, 1u, jt_main, jt_tail, preps, cands, valid, decisions, nullptr, 0u, fc / P.ncand, fc % P.ncand, smem);
		__syncthreads();
	}
}

} // namespace flacgpu

// ---------------------------------------------------------------------------------------------------------
// launch
// ---------------------------------------------------------------------------------------------------------
using namespace flacgpu;

namespace flacgpu {
static uint32_t ilog2_host(uint32_t v) { uint32_t l = 0; while(v >>= 1) l++; return l; }
// (channels per workgroup, wavefronts per workgroup) of the owner-layout evaluation
static void eval_shape(const DevParams &P, uint32_t &cpw, uint32_t &waves)
{
	const uint32_t nc = P.ncslots;
	// prefer a multiple of 4 wavefronts with every wavefront busy in every round, and the smallest workgroup that does it
	cpw = 1; waves = 0;
	for(uint32_t c = 1; c <= P.ncand && c <= (uint32_t)EVAL_CPW_MAX; c *= 2) {
		if(P.ncand % c) continue;
		const uint32_t items = c * nc;
		for(uint32_t w = 4; w <= (uint32_t)EVAL_MAX_WAVES; w += 4)
			if(items % w == 0 && w % c == 0 && waves == 0) { cpw = c; waves = w; }
		if(waves) break;
	}
	if(!waves) {
		// no exact fit: as many wavefronts as items, rounded to full rounds
		cpw = 1;
		const uint32_t rounds = (nc + EVAL_MAX_WAVES - 1) / EVAL_MAX_WAVES;
		waves = (nc + rounds - 1) / rounds;
		if(nc >= 4) waves = waves >= 8 ? 8 : 4;
	}
	const int fw = tune().eval_waves, fc = tune().eval_cpw;
	if(fc > 0 && fc <= EVAL_CPW_MAX && P.ncand % (uint32_t)fc == 0) cpw = (uint32_t)fc;
	if(fw > 0 && fw <= EVAL_MAX_WAVES) waves = (uint32_t)fw;
	while(owner_possible(P) && cpw > 1 && eval_layout(P, waves, cpw, false).total > 64 * 1024) cpw /= 2;     // keep at least 2 workgroups per CU
}
static uint32_t eval_waves_generic(const DevParams &P)
{
	const uint32_t nc = P.ncslots;
	const uint32_t rounds = (nc + EVAL_MAX_WAVES - 1) / EVAL_MAX_WAVES;
	uint32_t w = (nc + rounds - 1) / rounds;
	return w < 1 ? 1 : w;
}
size_t analyze_lds_bytes(const DevParams &P)
{
	uint32_t cpw, waves;
	eval_shape(P, cpw, waves);
	const size_t a = P.sig_bytes, d = owner_possible(P) ? eval_layout(P, waves, cpw, false).total : 0, g = eval_layout(P, eval_waves_generic(P), 1, true).total;
	return a > d ? (a > g ? a : g) : (d > g ? d : g);
}

// A 16-bit stream whose leaf partitions are so long that the reference sums them in 64 bits (stream_encoder.c:4814: few partitions
// of a long block) has no channel evalg_kernel takes (it wants the 32-bit sums) and none evalw_kernel takes (it wants 32-bit
// planes): everything would travel through both lists to eval_list_kernel's fixed grid.  Such streams keep the round-2 launch.
static bool evalg_worthwhile(const DevParams &P)
{
	if(P.bps > 16) return true;
	uint32_t fmax = 0;
	{ uint32_t b = P.blocksize; while(b && !(b & 1)) { fmax++; b >>= 1; } }
	if(fmax > P.max_po) fmax = P.max_po;
	const uint32_t psize = P.blocksize >> fmax;
	return (P.bps + 4) < 32 - ilog2_host(psize);
}
template <int MAXORD>
static hipError_t launch_model_eval(const DevParams &P, const int32_t *pcm, uint32_t nframes, uint32_t tail_n, const JobTable *jtm, const JobTable *jtt,
                                    const AnalyzeBuffers &B, SubDecision *dec, hipEvent_t *pev, hipStream_t s)
{
	static AttrFlags attr_set;
	if(AttrOnce once{attr_set}) {
		hipError_t e = hipFuncSetAttribute((const void *)eval_kernel<MAXORD, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
		if(e == hipSuccess) e = hipFuncSetAttribute((const void *)eval_kernel<MAXORD, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
		if(e == hipSuccess) e = hipFuncSetAttribute((const void *)eval_list_kernel<MAXORD>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
		if(e != hipSuccess) return e;
		once.ok();
	}
	if(P.max_analyses) {
		note_launch(K_MODEL);
		const uint32_t lanes = nframes * P.ncand * P.max_analyses;
		hipLaunchKernelGGL(model_kernel<MAXORD>, dim3((lanes + TPB - 1) / TPB), dim3(TPB), 0, s, P, nframes, tail_n, jtm, jtt, B.prep, B.autoc, B.cands, B.valid, B.nleft);
	}
	else if(!(prep2_applicable(P) && prep2_decides(P))) (void)hipMemsetAsync(B.nleft, 0, 2 * sizeof(uint32_t), s);      // (a deciding prep kernel has started the list)
	sync_debug("model", s);
	if(pev && P.max_analyses) (void)hipEventRecord(pev[2], s);       // (no LPC analyses: no kernel since pev[0]; flacgpu_batch_phase_ms knows)
	uint32_t cpw, waves;
	eval_shape(P, cpw, waves);
	const uint32_t gwaves = eval_waves_generic(P);
	const bool op = owner_possible(P);
	const size_t lds = op ? eval_layout(P, waves, cpw, false).total : 0, lds_generic = eval_layout(P, gwaves, 1, true).total;
	if(B.dbg) {
		static bool said = false;
		if(!said) {
			said = true;
			int nb0 = -1, nb1 = -1;
			(void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb0, (const void *)eval_kernel<MAXORD, 0>, (int)(waves * 64), lds);
			(void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb1, (const void *)eval_kernel<MAXORD, 2>, (int)(gwaves * 64), lds_generic);
			fprintf(stderr, "[flacgpu] eval: %u channels x %u waves per WG, %zu B LDS/WG, occupancy API: %d / %d WGs per CU\n", cpw, waves, lds, nb0, nb1);
		}
	}
	// which flavours can occur in this batch at all (each launch serves only its own channels)
	// how many workgroups ahead the one is that takes a finishing workgroup's place on its XCD: the workgroups an XCD holds
	int ahead = tune().eval_prefetch;
	if(ahead < 0) {
		static thread_local int derived = -1;        // (per thread: no shared write; the figure is a property of the kernel and the chip)
		if(derived < 0) {
			int nb = 0;
			if(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)eval_kernel<MAXORD, 0>, (int)(waves * 64), lds) != hipSuccess || nb < 1) nb = 2;
			derived = nb * 16;      // (measured on MI355X, profiles/r02_g_prefetch_ab.txt: 32..96 workgroups ahead are equally good, 256 is too early)
		}
		ahead = derived;
	}
	const bool pdz = prep2_applicable(P) && prep2_decides(P) && op && !B.dbg;
	// (evalw_kernel fills its image in 16-byte pieces of a run and ends a run with a whole or a half piece: runs of 16 k or 16 k + 8
	//  samples.  The 18- and 36-sample runs evalg_kernel takes since round 6 -- blocks of 1152 and 2304 -- are not its: what evalg_kernel
	//  lists there goes straight to the lane-owner body, and a stream of more than 16 bits at such a block size keeps the general kernel)
	const bool evalw_ok = ((P.blocksize / 64u) % 8u) == 0;
	if(pdz) {
		// the prep kernel has decided the frames of nominal length (flacgpu_prep.hip: DECIDE); the short last block and what it listed
		// go through the lane-owner body
		note_launch(K_EVAL_LIST);
		if(tail_n) hipLaunchKernelGGL(list_append_kernel, dim3(1), dim3(64), 0, s, (nframes - 1) * P.ncand, P.ncand, B.left, B.nleft);
		const uint32_t grid = nframes * P.ncand < 1024u ? nframes * P.ncand : 1024u;
		hipLaunchKernelGGL((eval_list_kernel<MAXORD>), dim3(grid), dim3(4 * 64), eval_layout(P, 4, 1, false).total, s, P, B.chan, nframes, tail_n, jtm, jtt, B.prep, B.cands, B.valid, dec, B.left, B.nleft);
	}
	else if(op && evalg_applicable(P) && !B.dbg && evalg_worthwhile(P) && (evalw_ok || P.bps <= 16) && (P.max_lpc_order <= 12 || P.bps <= 16)) {      // (13..16 taps: evalg_kernel<32> has them, evalw_kernel stops at 12)
		// one wavefront per channel: 16-bit pairs (flacgpu_evalg.hip), then the 32-bit channels it listed (flacgpu_evalw.hip) -- or
		// those straight away when the stream has more than 16 bits --, and what neither takes through the workgroup-per-channel
		// body above, a fixed grid looping over the last list
		const uint32_t *last_list, *last_count;
		if(P.bps <= 16) {
			hipError_t e = launch_evalg(P, nframes, tail_n, jtm, B, dec, s);
			if(e == hipSuccess && evalw_ok) e = launch_evalw(P, nframes, tail_n, jtm, B, dec, B.left, B.nleft, B.left2, B.nleft + 1, s);
			if(e != hipSuccess) return e;
			last_list = evalw_ok ? B.left2 : B.left; last_count = evalw_ok ? B.nleft + 1 : B.nleft;
		}
		else {
			const hipError_t e = launch_evalw(P, nframes, tail_n, jtm, B, dec, nullptr, nullptr, B.left, B.nleft, s);
			if(e != hipSuccess) return e;
			last_list = B.left; last_count = B.nleft;
		}
		note_launch(K_EVAL_LIST);
		const uint32_t lw = 4u;
		const uint32_t grid = nframes * P.ncand < 1024u ? nframes * P.ncand : 1024u;
		hipLaunchKernelGGL((eval_list_kernel<MAXORD>), dim3(grid), dim3(lw * 64), eval_layout(P, lw, 1, false).total, s, P, B.chan, nframes, tail_n, jtm, jtt, B.prep, B.cands, B.valid, dec, last_list, last_count);
	}
	else if(!op && P.bps <= 16 && P.max_lpc_order > 12 && evalg_applicable(P) && !B.dbg && evalg_worthwhile(P)) {
		// predictors of 17..32 taps on 16-bit input (round 6): the wavefront-per-channel kernel takes what its 32-bit chain is exact for
		// (and the channels without residual candidates); the general kernel below skips what it marked (ChanPrep::handled)
		const hipError_t e = launch_evalg(P, nframes, tail_n, jtm, B, dec, s);
		if(e != hipSuccess) return e;
		// (the short last block is not that kernel's: its channels without a residual candidate are decided by the one-wavefront form
		//  of the general kernel, as in the branch below)
		if(tail_n) { note_launch(K_EVAL); hipLaunchKernelGGL((eval_kernel<MAXORD, 0>), dim3(nframes * P.ncand), dim3(64), eval_layout(P, 1, 1, false).total, s, P, B.chan, nframes, tail_n, 1u, jtm, jtt, B.prep, B.cands, B.valid, dec, B.dbg, 0u); }
	}
	else if(op) { note_launch(K_EVAL); hipLaunchKernelGGL((eval_kernel<MAXORD, 0>), dim3(nframes * (P.ncand / cpw)), dim3(waves * 64), lds, s, P, B.chan, nframes, tail_n, cpw, jtm, jtt, B.prep, B.cands, B.valid, dec, B.dbg, (uint32_t)ahead); }
	else { note_launch(K_EVAL); hipLaunchKernelGGL((eval_kernel<MAXORD, 0>), dim3(nframes * P.ncand), dim3(64), eval_layout(P, 1, 1, false).total /* VARIANT 0 lays out as such */, s, P, B.chan, nframes, tail_n, 1u, jtm, jtt, B.prep, B.cands, B.valid, dec, B.dbg, 0u); }
	if(!op || tail_n || P.max_po > 6)
		{ note_launch(K_EVAL); hipLaunchKernelGGL((eval_kernel<MAXORD, 2>), dim3(nframes * P.ncand), dim3(gwaves * 64), lds_generic, s, P, B.chan, nframes, tail_n, 1u, jtm, jtt, B.prep, B.cands, B.valid, dec, B.dbg, 0u); }
	sync_debug("eval", s);
	return hipGetLastError();
}

// FLACGPU_SYNC_DEBUG=1: wait for every kernel and name the one that faults (development aid)
void sync_debug(const char *what, hipStream_t s)
{
	if(!tune().sync_debug) return;
	fprintf(stderr, "[flacgpu] %s ...", what); fflush(stderr);
	const hipError_t e = hipStreamSynchronize(s);
	fprintf(stderr, " %s\n", e == hipSuccess ? "ok" : hipGetErrorString(e)); fflush(stderr);
}
hipError_t launch_analyze(const DevParams &P, const int32_t *pcm, const float *win, const float *tailwin, uint32_t nframes, uint32_t tail_n,
                          const JobTable *jtm, const JobTable *jtt, uint32_t nsets_main, const AnalyzeBuffers &B, SubDecision *dec, hipEvent_t *pev, hipStream_t s)
{
	static AttrFlags attr_set;
	if(AttrOnce once{attr_set}) {
		hipError_t e = hipFuncSetAttribute((const void *)prep_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
		if(e != hipSuccess) return e;
		once.ok();
	}
	{
		// frames of nominal length: one workgroup per frame (flacgpu_prep.hip); the short last block, and block sizes that
		// kernel does not take, go through the workgroup-per-subframe kernel above
		uint32_t f_lo = 0;
		if(prep2_applicable(P)) {
			f_lo = tail_n ? nframes - 1 : nframes;
			const hipError_t e = launch_prep2(P, pcm, f_lo, B, dec, s);
			if(e != hipSuccess) return e;
		}
		if(f_lo < nframes)
			{ note_launch(K_PREP); hipLaunchKernelGGL(prep_kernel<0>, dim3((nframes - f_lo) * P.ncand), dim3(TPB), P.sig_bytes, s, P, pcm, nframes, tail_n, f_lo, B.prep, B.cands, B.valid, B.chan); }
	}
	sync_debug("prep", s);
	if(pev) (void)hipEventRecord(pev[0], s);
	if(P.max_analyses) {
		// frames of nominal length: the streaming kernel (flacgpu_autoc.hip); the short last block, and tiny blocks
		// (lpc.c:133-157), go through the wavefront-per-job kernel above
		uint32_t f_lo = 0;
		// autoc2 runs a whole window job per lane quartet: few subframes in the batch (mono, small batches) leave most
		// SIMDs without a wavefront, and the batch then takes one full job's time.  Below a wavefront or two per SIMD the
		// wavefront-per-job kernel (16x the wavefronts for 1.8x the arithmetic) is the faster one.
		const int force = tune().autoc2_force;      // FLACGPU_AUTOC2 + 1 -- 0: decide here, 1: never, 2: always
		const uint32_t nmain2 = tail_n ? nframes - 1 : nframes;
		const uint32_t waves2 = P.max_jobs * ((nmain2 * P.ncand + 15) / 16);
		// measured break-even (scripts/chan_rate.py): ~640 wavefronts for the stereo mid/side flavour (four channels share
		// the loads of a frame), ~2048 for the others
		const bool ms4 = P.channels == 2 && P.ms_mode == 1;
		const bool use2 = force == 2 || (force == 0 && waves2 >= (ms4 ? 640u : 2048u));
		if(autoc2_applicable(P) && use2) {
			f_lo = tail_n ? nframes - 1 : nframes;
			const hipError_t e = launch_autoc2(P, pcm, B.chan, win, f_lo, P.max_jobs, nsets_main, jtm, B.prep, B.autoc, s);
			if(e != hipSuccess) return e;
		}
		else if(autoc4_applicable(P) && (force == 2 || (force == 0 && P.max_jobs * ((nmain2 * P.ncand + 63) / 64) >= 128u))) {
			// -l 16 and up: the reference's plain loop, a lane per subframe (from 128 wavefronts up; the wavefront-per-job kernel below)
			f_lo = tail_n ? nframes - 1 : nframes;
			const hipError_t e = launch_autoc4(P, B.chan, win, f_lo, P.max_jobs, jtm, B.prep, B.autoc, s);
			if(e != hipSuccess) return e;
		}
		if(f_lo < nframes) {
			const uint32_t items = (nframes - f_lo) * P.ncand * P.max_jobs;
			note_launch(K_AUTOC);
			hipLaunchKernelGGL(autoc_kernel<0>, dim3((items + TPB / 64 - 1) / (TPB / 64)), dim3(TPB), 0, s, P, pcm, win, tailwin, nframes, tail_n, f_lo, jtm, jtt, B.prep, B.autoc);
		}
	}
	sync_debug("autoc", s);
	if(pev && P.max_analyses) (void)hipEventRecord(pev[1], s);       // (an event record costs the stream ~4 us: a tenth of a -0 step for two empty phases)
	const uint32_t m = P.max_lpc_order > 4 ? P.max_lpc_order : 4;
	if(m <= 8) return launch_model_eval<8>(P, pcm, nframes, tail_n, jtm, jtt, B, dec, pev, s);
	if(m <= 12) return launch_model_eval<12>(P, pcm, nframes, tail_n, jtm, jtt, B, dec, pev, s);
	if(m <= 16) return launch_model_eval<16>(P, pcm, nframes, tail_n, jtm, jtt, B, dec, pev, s);
	return launch_model_eval<32>(P, pcm, nframes, tail_n, jtm, jtt, B, dec, pev, s);
}
} // namespace flacgpu
