// flac_amd/csrc/flacgpu_prep.hip -- per-subframe preparation for all frames of nominal length:
// mid/side build (stream_encoder.c:3823-3836), loose mid/side decision (:3778-3807), wasted bits (:5077, :3842-3867),
// fixed-predictor error sums and order guess (fixed.c:222-299 / fixed_intrin_avx2.c:57), CONSTANT detection
// (:4111-4140), limit_min_bitrate (:3874-3879).  Writes the hand-off record (ChanPrep, Candidate[0]) and the
// PLANAR, already wasted-bits-shifted channel signal that the evaluation and pack kernels read (16-bit pairs when
// the subframe fits 16 bits, else 32-bit), so that those kernels stream 2-4 bytes per sample instead of re-reading
// and re-deriving the interleaved frame.
//
// One workgroup per frame, one WAVEFRONT per candidate channel: the frame is staged once in LDS (coalesced loads),
// then every wavefront derives its own channel (L, R, (L+R)>>1, L-R) from the staged data, each lane owning
// runs of 16 consecutive samples (conflict-free 18-word rows), reduces its statistics with wavefront shuffles only,
// decides, and stores its channel.  HBM/L2 bound: reads 4*C bytes, writes ~10 bytes per inter-channel sample.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "flacgpu_dev.h"
#include "flacgpu_devfn.h"

namespace flacgpu {

#ifndef P2_WAVES
#define P2_WAVES 3
#endif

constexpr uint32_t P2_DIVTAB_BYTES = ((MAX_PO + 1) * (MAX_ORDER + 1) * 4 + 15) & ~15u;
// LDS image of one raw channel, transposed: sample i sits in row i%16, column i/16 + 1 (column 0 holds zeros: the
// samples in front of the block), rows P2_TS(n) words apart.  A lane that owns the 16 samples of chunk t reads
// row r at column t+1 -- consecutive lanes, consecutive words, no address arithmetic (ds_read offsets); the staging
// store of 32 consecutive samples hits 32 banks because the row stride is 2 mod 32.
__host__ __device__ inline uint32_t p2_ts(uint32_t n, uint32_t ch = CHUNK) { const uint32_t nch = (n + ch - 1) / ch; return ((nch - 1 + 31) / 32) * 32 + 2; }
__host__ __device__ inline uint32_t p2_chan_bytes(uint32_t n, uint32_t ch = CHUNK) { return ch * p2_ts(n, ch) * 4; }
// The chunk a lane owns in prep2_kernel: 16 samples, except for the 1152-sample blocks of the presets -0 .. -2, where 16 makes 72
// chunks -- a full pass of the wavefront and one with eight lanes -- and 18 makes 64: one pass, every lane busy.  (1152 = 18 * 64, and
// every partition the Rice search may ask for, 1152 >> 0..6, is a whole number of 18-sample chunks.)
__host__ __device__ constexpr uint32_t p2_chunk_len(uint32_t blocksize) { return blocksize == 1152 ? 18u : (uint32_t)CHUNK; }

// WIDE: per-run partial sums may exceed 32 bits (more than 20 bits per sample)
// NFIX: the block size when it is one of the presets' (4096, 1152), else 0 = read it from the parameters.  A constant block size
// makes the LDS tile stride a constant and folds the tile addresses into instruction offsets: with the stride in a register the
// kernel spilled 27 VGPRs (the story of prep3_kernel below, which serves stereo with a mid/side search; this one serves the
// presets without one, -0 and -3, the loose ones, -1 and -4 at 1152, and everything that is not stereo).
// DECIDE (prep2_decides()): no LPC search and one fixed order per subframe (-0, -1, -2): the residual of the guessed order is the
// very difference signal whose sums this kernel has just taken, chunk by chunk -- so it also runs the Rice search on the partition
// sums it can add up from them (find_best_partition_order_, stream_encoder.c:4701-5075), compares with VERBATIM / CONSTANT and
// writes the SubDecision itself: no evaluation kernel, no second pass over the block.  A channel whose sums leave the 32-bit node
// arithmetic goes on eval_list_kernel's list instead.
template <bool WIDE, uint32_t NFIX, bool DECIDE>
__global__ __launch_bounds__(TPB, P2_WAVES) void prep2_kernel(const DevParams P, const int32_t *__restrict__ pcm, uint32_t nmain,
                                                    ChanPrep *__restrict__ preps, Candidate *__restrict__ cands, int *__restrict__ valid,
                                                    int32_t *__restrict__ chan, SubDecision *__restrict__ decisions, uint32_t *__restrict__ left, uint32_t *__restrict__ nleft)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	__shared__ uint32_t sh_alleq[FLACGPU_MAX_CHANNELS];
	__shared__ uint32_t sh_loose_ms;
	const int tid = (int)threadIdx.x, lane = tid & 63;
	const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane(tid >> 6), nthreads = blockDim.x;
	const uint32_t C = P.channels, N = NFIX ? NFIX : P.blocksize, n = N;          // n % 16 == 0 (prep2_applicable)
	const uint32_t f = blockIdx.x;
	const int32_t *frame_pcm = pcm + (size_t)f * N * C;
	const bool stereo_ms = C == 2 && P.ms_mode != 0;
	const uint32_t G = stereo_ms ? 2u : (C < 4 ? C : 4u);            // raw channels staged per round
	const uint32_t cstride = P.ncslots;
	constexpr uint32_t CH = p2_chunk_len(NFIX);                          // samples a lane owns per pass (18 for the 1152-sample blocks)
	const uint32_t nchunks = n / CH;
	const uint32_t TS = NFIX ? ((NFIX / CH - 1 + 31) / 32) * 32 + 2 : p2_ts(n), cbytes = CH * TS * 4;       // (p2_ts, p2_chan_bytes)
	const bool need_flags = P.limit_min_bitrate || P.ms_mode == 2;
	// DECIDE: [divisor table][per wavefront: chunk sums 5 x nchunks | the first chunk's extras 8 | Rice parameters 64 B] behind the staged channels
	const uint32_t dz_base = (stereo_ms ? 2u : (C < 4 ? C : 4u)) * cbytes;
	uint32_t *dz_divtab = (uint32_t *)(smem + dz_base);
	// (WIDE: a chunk's sum of fourth differences of 25-bit samples does not fit 32 bits: 64-bit chunk sums, the extras and parameters behind them)
	constexpr uint32_t CSB = WIDE ? 8u : 4u;
	const uint32_t dz_wbytes = 5 * nchunks * CSB + 8 * 4 + 64;
	uint32_t *dz_csum = (uint32_t *)(smem + dz_base + P2_DIVTAB_BYTES + (size_t)wave * dz_wbytes);
	uint64_t *dz_csum64 = (uint64_t *)dz_csum;
	uint32_t *dz_extra = (uint32_t *)((unsigned char *)dz_csum + (size_t)5 * nchunks * CSB);
	uint8_t *dz_kout = (uint8_t *)(dz_extra + 8);

	for(uint32_t c0 = 0; c0 < C; c0 += G) {
		const uint32_t nraw = C - c0 < G ? C - c0 : G;
		__syncthreads();
		if(DECIDE && c0 == 0) {
			for(uint32_t t = (uint32_t)tid; t < (MAX_PO + 1) * (MAX_ORDER + 1); t += nthreads) {
				const uint32_t po = t / (MAX_ORDER + 1), o = t - po * (MAX_ORDER + 1), ps = n >> po;
				dz_divtab[t] = ps > o ? 0x40000u / (ps - o) : 0;
			}
		}
		// ---- stage the raw channels -------------------------------------------------------------------------
		if(tid < (int)CH) for(uint32_t r = 0; r < nraw; r++) ((int32_t *)(smem + (size_t)r * cbytes))[(uint32_t)tid * TS] = 0;
		if(CH != CHUNK) {
			// sample i: row i % CH, column i / CH + 1 (the division by a constant is a multiply; nine trips per thread at 1152 / 128)
			if(C == 2) {
				int32_t *sl = (int32_t *)smem, *sr = (int32_t *)(smem + cbytes);
				const int2 *p = (const int2 *)frame_pcm;
#pragma unroll 9
				for(uint32_t i = (uint32_t)tid; i < n; i += nthreads) { const uint32_t c = i / CH, a = (i - c * CH) * TS + c + 1; const int2 lr = p[i]; sl[a] = lr.x; sr[a] = lr.y; }
			}
			else {
				int32_t *s0 = (int32_t *)smem, *s1 = (int32_t *)(smem + cbytes), *s2 = (int32_t *)(smem + 2 * (size_t)cbytes), *s3 = (int32_t *)(smem + 3 * (size_t)cbytes);
#pragma unroll 4
				for(uint32_t i = (uint32_t)tid; i < n; i += nthreads) {
					const uint32_t c = i / CH, a = (i - c * CH) * TS + c + 1;
					const int32_t *p = frame_pcm + (size_t)i * C + c0;
					const int32_t v0 = p[0], v1 = nraw > 1 ? p[1] : 0, v2 = nraw > 2 ? p[2] : 0, v3 = nraw > 3 ? p[3] : 0;
					s0[a] = v0;
					if(nraw > 1) s1[a] = v1;
					if(nraw > 2) s2[a] = v2;
					if(nraw > 3) s3[a] = v3;
				}
			}
		}
		else {
			const uint32_t a0 = ((uint32_t)tid & 15u) * TS + ((uint32_t)tid >> 4) + 1;     // sample i = tid; i += nthreads moves nthreads/16 columns
			if(C == 2) {
				int32_t *sl = (int32_t *)smem, *sr = (int32_t *)(smem + cbytes);
				const int2 *p = (const int2 *)frame_pcm;
				uint32_t a = a0;
				// (eight loads in flight per thread: rolled, this loop was a load, a wait and two LDS stores per trip)
#pragma unroll 8
				for(uint32_t i = (uint32_t)tid; i < n; i += nthreads, a += nthreads / 16) { const int2 lr = p[i]; sl[a] = lr.x; sr[a] = lr.y; }
			}
			else {
				// mono, 3..8 channels (up to four raw channels per round): a thread takes sample i of every channel of the round --
				// neighbouring words of one line --, four samples in flight per thread (round 4: a rolled loop of single loads by one
				// wavefront per channel: mono's prep took longer than stereo's four channels, profiles/archive/r04_t_chan_rate.txt)
				uint32_t a = a0;
				int32_t *s0 = (int32_t *)smem, *s1 = (int32_t *)(smem + cbytes), *s2 = (int32_t *)(smem + 2 * (size_t)cbytes), *s3 = (int32_t *)(smem + 3 * (size_t)cbytes);
#pragma unroll 4
				for(uint32_t i = (uint32_t)tid; i < n; i += nthreads, a += nthreads / 16) {
					const int32_t *p = frame_pcm + (size_t)i * C + c0;
					const int32_t v0 = p[0], v1 = nraw > 1 ? p[1] : 0, v2 = nraw > 2 ? p[2] : 0, v3 = nraw > 3 ? p[3] : 0;
					s0[a] = v0;
					if(nraw > 1) s1[a] = v1;
					if(nraw > 2) s2[a] = v2;
					if(nraw > 3) s3[a] = v3;
				}
			}
		}
		__syncthreads();

		// ---- this wavefront's channel ---------------------------------------------------------------------
		const bool active = stereo_ms ? true : wave < nraw;
		const uint32_t which = stereo_ms ? wave : c0 + wave;            // 0..C-1 channel, C mid, C+1 side
		const int32_t *sa = (const int32_t *)(smem + (size_t)((stereo_ms || !active) ? 0 : wave) * cbytes);
		const int32_t *sb = (const int32_t *)(smem + (size_t)(stereo_ms ? 1 : 0) * cbytes);
		const int mode = !stereo_ms ? 0 : (int)which;                   // 0 take a, 1 take b, 2 mid, 3 side
		const bool loose_here = P.ms_mode == 2 && wave == 0;
		Prep2Acc A;
		A.orv = 0; A.diff = 0; A.mag = 0xffffffffu;
#pragma unroll
		for(int k = 0; k < 5; k++) A.e[k] = 0;
		uint64_t lr_sum = 0, ms_sum = 0;
		int32_t first = 0;
		if(mode == 3) A.mag = 0;
		if(active) {
			{
				const int32_t a = sa[1], b = sb[1];          // sample 0: row 0, column 1
				first = mode == 0 ? a : mode == 1 ? b : mode == 2 ? ((a + b) >> 1) : (a - b);
			}
			for(uint32_t ch = (uint32_t)lane; ch < nchunks; ch += 64) {
				// x[k] = sample CH*ch - 4 + k: the last four rows of column ch, then the CH rows of column ch+1
				const int32_t *pa = sa + ch, *pb = sb + ch;
				int32_t x[CH + 4];
				if(mode == 0 && !loose_here) {
#pragma unroll
					for(int k = 0; k < (int)CH + 4; k++) x[k] = k < 4 ? pa[(CH - 4 + k) * TS] : pa[(k - 4) * TS + 1];
				}
				else {
					int32_t a[CH + 4], b[CH + 4];
#pragma unroll
					for(int k = 0; k < (int)CH + 4; k++) { a[k] = k < 4 ? pa[(CH - 4 + k) * TS] : pa[(k - 4) * TS + 1]; b[k] = k < 4 ? pb[(CH - 4 + k) * TS] : pb[(k - 4) * TS + 1]; }
					if(loose_here) {
						// loose mid/side (stream_encoder.c:3778-3807), bps < 25
#pragma unroll
						for(int t = 0; t < (int)CH; t++) {
							if(t > 0 || ch > 0) {
								const int32_t pl = a[t + 4] - a[t + 3], pr = b[t + 4] - b[t + 3];
								lr_sum += (uint64_t)(uint32_t)(abs(pl) + abs(pr));
								ms_sum += (uint64_t)(uint32_t)(abs((pl + pr) >> 1) + abs(pl - pr));
							}
						}
					}
#pragma unroll
					for(int k = 0; k < (int)CH + 4; k++) x[k] = mode == 0 ? a[k] : mode == 1 ? b[k] : mode == 2 ? ((a[k] + b[k]) >> 1) : (a[k] - b[k]);
				}
				if(DECIDE) {
					uint32_t cs[5], ex[5] = {0, 0, 0, 0, 0};
					// A lane's chunk is summed in 32 bits wherever that cannot wrap: |d4| <= 16 max|x|, so 16 (18) of them stay below 2^32 for
					// samples of up to 24 bits -- and for the 25-bit side channel at 16 per chunk (256 (2^24 - 1)), not at 18.  Only that one
					// case adds every |difference| to the 64-bit totals (two instructions instead of one, twenty times per sample; until
					// round 6 every channel of a > 20-bit stream paid that).
					constexpr bool SIDE64 = WIDE && CH > 16;
					uint64_t before[5];
					if(SIDE64 && mode == 3) {
#pragma unroll
						for(int k = 0; k < 5; k++) before[k] = A.e[k];
					}
					if(mode == 3) prep2_chunk<SIDE64, true, true, (int)CH>(x, ch == 0, first, A, cs, ex);
					else prep2_chunk<false, false, true, (int)CH>(x, ch == 0, first, A, cs, ex);
					if(WIDE) {
#pragma unroll
						for(int k = 0; k < 5; k++) dz_csum64[(uint32_t)k * nchunks + ch] = (SIDE64 && mode == 3) ? A.e[k] - before[k] : (uint64_t)cs[k];
					}
					else {
#pragma unroll
						for(int k = 0; k < 5; k++) dz_csum[(uint32_t)k * nchunks + ch] = cs[k];
					}
					if(ch == 0) {
#pragma unroll
						for(int k = 0; k < 5; k++) dz_extra[k] = ex[k];
					}
				}
				else if(mode == 3) prep2_chunk<(WIDE && CH > 16), true, false, (int)CH>(x, ch == 0, first, A);      // (see above: only the side channel at 18 per chunk)
				else prep2_chunk<false, false, false, (int)CH>(x, ch == 0, first, A);
			}
			A.orv = wave_or_u32(A.orv);
			A.diff = wave_or_u32(A.diff);
			if(mode == 3) A.mag = wave_or_u32(A.mag);
#pragma unroll
			for(int k = 0; k < 5; k++) A.e[k] = wave_sum_u50(A.e[k]);
		}
		uint32_t cand = stereo_ms ? wave : which;
		bool emit = active;
		if(need_flags) {
			if(P.ms_mode == 2 && wave == 0) { lr_sum = wave_sum_u50(lr_sum); ms_sum = wave_sum_u50(ms_sum); if(lane == 0) sh_loose_ms = lr_sum < ms_sum ? 0u : 1u; }
			if(active && lane == 0 && which < C) sh_alleq[which] = A.diff == 0 ? 1u : 0u;
			__syncthreads();
			if(P.ms_mode == 2) {
				const uint32_t ms = sh_loose_ms;
				emit = ms ? wave >= 2 : wave < 2;
				cand = ms ? wave - 2 : wave;
			}
		}
		else if(P.ms_mode == 2) emit = false;     // unreachable (need_flags covers loose mode)
		if(!emit) continue;

		// ---- decisions (every lane holds the reduced values) --------------------------------------------------------
		bool disable_constant = P.disable_constant != 0;
		if(P.limit_min_bitrate && !disable_constant && (P.ms_mode == 2 ? which == 1 : which >= C - 1)) {
			bool all = true;
			for(uint32_t c = 0; c + 1 < C; c++) all = all && sh_alleq[c] != 0;
			if(all) disable_constant = true;
		}
		uint32_t wasted = A.orv ? (uint32_t)(__ffs((int)A.orv) - 1) : 0;
		if(wasted > P.bps) wasted = P.bps;
		const uint32_t sbps = P.bps - wasted + (which == C + 1 ? 1 : 0);
		uint32_t flags = 0, fixed_order = 0;
		int32_t constant = 0;
		const uint32_t verbatim_bits = (P.disable_verbatim && n >= 4) ? 0xffffffffu : 8 + wasted + n * sbps;
		if(n > 4) {
			const uint32_t n4 = n - 4;
			const uint64_t e0 = A.e[0] >> wasted, e1 = A.e[1] >> wasted, e2 = A.e[2] >> wasted, e3 = A.e[3] >> wasted, e4 = A.e[4] >> wasted;
			uint32_t guess_fixed;
			{
				const uint64_t m34 = e3 < e4 ? e3 : e4, m234 = e2 < m34 ? e2 : m34, m1234 = e1 < m234 ? e1 : m234;
				if(e0 <= m1234) guess_fixed = 0;
				else if(e1 <= m234) guess_fixed = 1;
				else if(e2 <= m34) guess_fixed = 2;
				else if(e3 <= e4) guess_fixed = 3;
				else guess_fixed = 4;
			}
			// CONSTANT needs rbps[1] == 0 and all samples equal (stream_encoder.c:4111-4140); all equal implies e1 == 0
			const bool is_constant = !disable_constant && A.diff == 0;
			const size_t fcx = (size_t)f * P.ncand + cand;
			if(is_constant) { flags |= PREP_CONSTANT; constant = first >> wasted; }
			else if(P.max_lpc_order > 0) flags |= PREP_LPC;
			const bool fixed_allowed = !is_constant && (!P.disable_fixed || (P.max_lpc_order == 0 && verbatim_bits == 0xffffffffu));
			fixed_order = fixed_allowed ? guess_fixed : 0;
			const uint64_t es[5] = {e0, e1, e2, e3, e4};
			if(emit_fixed_candidates(P, &cands[fcx * cstride], &valid[fcx * cstride], es, n4, guess_fixed, fixed_allowed, sbps, lane)) flags |= PREP_FIXED_VALID;
		}
		const uint32_t fmt = (sbps <= 16 || (A.mag >> wasted) < 32768u) ? 1u : 0u;
		const size_t fc = (size_t)f * P.ncand + cand;
		uint32_t handled = 0;
		if(DECIDE) {
			// first minimum in the reference's evaluation order (verbatim -> constant | the fixed order), as eval_kernel decides it
			const uint32_t hdr = 8 + wasted;
			uint32_t best_type = 1, best_bits = verbatim_bits, dpo = 0, rice2 = 0;
			int32_t best_constant = 0;
			bool leave = false;
			if(flags & PREP_CONSTANT) {
				const uint32_t bits = hdr + sbps;
				if(bits < best_bits) { best_type = 0; best_constant = constant; best_bits = bits; }
			}
			if(flags & PREP_FIXED_VALID) {
				uint32_t fmax = 0;
				{ uint32_t b = n; while(!(b & 1)) { fmax++; b >>= 1; } if(fmax > 15) fmax = 15; }
				fmax = umin32(fmax, P.max_po);                                         // (<= 6: prep2_decides)
				const uint32_t fmin = umin32(P.min_po, fmax), e = 6 - fmax, cpp = (n >> fmax) / CH;
				__builtin_amdgcn_wave_barrier();                                       // this wavefront's chunk sums are in LDS
				uint32_t v = 0;
				uint64_t v64 = 0;
				if(((uint32_t)lane & ((1u << e) - 1u)) == 0) {
					const uint32_t pidx = (uint32_t)lane >> e;
					uint64_t sum = 0;
					if(WIDE) { const uint64_t *cs = dz_csum64 + fixed_order * nchunks + pidx * cpp; for(uint32_t c = 0; c < cpp; c++) sum += cs[c]; }
					else { const uint32_t *cs = dz_csum + fixed_order * nchunks + pidx * cpp; for(uint32_t c = 0; c < cpp; c++) sum += cs[c]; }
					if(pidx == 0) sum += dz_extra[fixed_order];
					sum >>= wasted;
					v64 = sum;
					v = sum >= (1u << 23) ? (1u << 23) : (uint32_t)sum;
				}
				const bool bigleaf = __any((int)(v >= (1u << 23))) != 0;
				if(bigleaf && !WIDE) leave = true;
				else {
					uint32_t po = 0, rbits;
					if(!bigleaf) rbits = rice_search_nodes(v, e, n, fixed_order, fmax, fmin, P.rice_limit, dz_divtab, dz_kout, &po, lane);
					else {
						// leaf sums beyond the 32-bit node arithmetic (round 6: 17..24-bit input at -0..-2 used to go to eval_list_kernel's list
						// here, every channel of it): the search on 64-bit sums, the leaf's lanes holding its sum between them
						// (find_best_partition_order_ / set_partitioned_rice_, stream_encoder.c:4701-5075; :4814-4817 for `narrow`)
						const bool narrow = (sbps + 4) < (32 - ilog2_u32(n >> fmax));
						rbits = rice_search_owner(v64, narrow, e, n, fixed_order, fmax, fmin, P.rice_limit, dz_divtab, dz_kout, &po, lane);
					}
					const uint32_t est = sat_add_u32(hdr + fixed_order * sbps, rbits);
					if(est > 0 && est < best_bits) { best_type = 2; best_bits = est; dpo = po; }
				}
			}
			if(leave) { if(lane == 0) left[atomicAdd(nleft, 1u)] = (uint32_t)fc; }
			else {
				if(best_bits == 0xffffffffu) { best_type = 1; best_bits = hdr + n * sbps; }   // stream_encoder.c:4281
				SubDecision *dec = decisions + fc;
				if(best_type == 2) {
					uint32_t big = 0;
					if((uint32_t)lane < (1u << dpo)) { const uint8_t kk = dz_kout[lane]; dec->params[lane] = kk; if(kk >= 15) big = 1; }
					rice2 = __any((int)big) ? 1u : 0u;                         // stream_encoder.c:4786-4791
				}
				if(lane < MAX_ORDER) dec->q[lane] = 0;
				if(lane == 0) {
					dec->bits = best_bits;
					dec->type = (uint8_t)best_type; dec->order = (uint8_t)(best_type == 2 ? fixed_order : 0); dec->wasted = (uint8_t)wasted;
					dec->po = (uint8_t)dpo; dec->rice2 = (uint8_t)rice2; dec->precision = 0;
					dec->shift = 0; dec->which = (uint8_t)which;
					dec->constant = best_constant; dec->constant_hi = best_constant >> 31; dec->fmt = fmt;
				}
				handled = EVG_HANDLED;
			}
		}
		if(lane == 0) {
			ChanPrep pr;
			pr.which = which; pr.wasted = wasted; pr.sbps = sbps; pr.n = n; pr.flags = flags; pr.fixed_order = fixed_order;
			pr.constant = constant; pr.verbatim_bits = verbatim_bits; pr.fmt = fmt; pr.constant_hi = constant >> 31; pr.handled = handled; pr.pad = 0;
			preps[fc] = pr;
		}
		// ---- planar channel, shifted ---------------------------------------------------------------------------
		uint32_t *dst = (uint32_t *)(chan + fc * (size_t)N);
		for(uint32_t ch = (uint32_t)lane; ch < nchunks; ch += 64) {
			const uint32_t base = ch * CH;
			const int32_t *pa = sa + ch + 1, *pb = sb + ch + 1;
			int32_t x[CH];
#pragma unroll
			for(int k = 0; k < (int)CH; k++) {
				const int32_t a = pa[k * TS];
				int32_t v = a;
				if(mode != 0) { const int32_t b = pb[k * TS]; v = mode == 1 ? b : mode == 2 ? ((a + b) >> 1) : (a - b); }
				x[k] = v >> wasted;
			}
			if(CH != CHUNK) {
				// 18 samples: nine words of pairs (or eighteen words) at a 36-byte (72-byte) lane stride: plain word stores
				if(fmt) {
#pragma unroll
					for(int j = 0; j < (int)CH / 2; j++) dst[base / 2 + j] = ((uint32_t)x[2 * j] & 0xffffu) | ((uint32_t)x[2 * j + 1] << 16);
				}
				else {
#pragma unroll
					for(int j = 0; j < (int)CH; j++) dst[base + j] = (uint32_t)x[j];
				}
			}
			else if(fmt) {
				uint4 w0, w1;
				w0.x = ((uint32_t)x[0] & 0xffffu) | ((uint32_t)x[1] << 16); w0.y = ((uint32_t)x[2] & 0xffffu) | ((uint32_t)x[3] << 16);
				w0.z = ((uint32_t)x[4] & 0xffffu) | ((uint32_t)x[5] << 16); w0.w = ((uint32_t)x[6] & 0xffffu) | ((uint32_t)x[7] << 16);
				w1.x = ((uint32_t)x[8] & 0xffffu) | ((uint32_t)x[9] << 16); w1.y = ((uint32_t)x[10] & 0xffffu) | ((uint32_t)x[11] << 16);
				w1.z = ((uint32_t)x[12] & 0xffffu) | ((uint32_t)x[13] << 16); w1.w = ((uint32_t)x[14] & 0xffffu) | ((uint32_t)x[15] << 16);
				uint4 *d4 = (uint4 *)(dst + base / 2);
				d4[0] = w0; d4[1] = w1;
			}
			else {
				uint4 *d4 = (uint4 *)(dst + base);
#pragma unroll
				for(int k = 0; k < 4; k++) { uint4 w; w.x = (uint32_t)x[4 * k]; w.y = (uint32_t)x[4 * k + 1]; w.z = (uint32_t)x[4 * k + 2]; w.w = (uint32_t)x[4 * k + 3]; d4[k] = w; }
			}
		}
	}
}

// ---------------------------------------------------------------------------------------------------------
// prep3_kernel: stereo frames of 4096 samples with a mid/side search (what the presets from -4 up encode)
// ---------------------------------------------------------------------------------------------------------
// Same results as prep2_kernel, different split of the work: wavefront w owns the QUARTER [w*N/4, (w+1)*N/4) of the
// frame for ALL candidate channels.  (Round 6: <., NW, CH> -- NW wavefronts, each a PART of 64 chunks of CH samples: blocks of
// NW x 64 x CH samples, i.e. 1024, 2048, 4096, 8192 with chunks of 16 and 1152, 2304, 4608 with chunks of 18 -- the block sizes
// of `-b` on the LPC presets ran prep2_kernel, 2.7x this kernel's time per sample: VERDICT r05 #7's -8 -b 1152.)  It stages its own quarter (coalesced loads -> its own transposed LDS tile, no
// workgroup barrier in front of the compute), reads left and right ONCE for the four channels derived from them, and
// the workgroup meets only twice: to add up the four partial statistics and to learn the wasted bits before the planar
// channels are written.
struct Prep3Part {
	uint32_t orv[4], diff[4], mag;
	int32_t first[4];
	uint64_t e[4][5];
	uint64_t lr, ms;
};
struct Prep3Out { uint32_t wasted[4]; int32_t slot[4]; uint32_t fmt[4]; };

template <bool WIDE, int NW = 4, int CH = CHUNK>
__global__ __launch_bounds__(64 * NW, 4) void prep3_kernel(const DevParams P, const int32_t *__restrict__ pcm, uint32_t nmain,
                                                       ChanPrep *__restrict__ preps, Candidate *__restrict__ cands, int *__restrict__ valid,
                                                       int32_t *__restrict__ chan)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	__shared__ Prep3Part part[NW];
	__shared__ Prep3Out outp;
	const int tid = (int)threadIdx.x, lane = tid & 63;
	const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane(tid >> 6);
	constexpr uint32_t Q = 64 * CH, N = NW * Q;                           // prep3_applicable(): one CH-sample chunk per lane; a
	                                                                      // constant tile stride folds every LDS address into an offset
	const uint32_t f = blockIdx.x;
	const uint32_t q0 = wave * Q;
	const int2 *p = (const int2 *)(pcm + (size_t)f * N * 2) + q0;
	constexpr uint32_t TS = ((Q / CH - 1 + 31) / 32) * 32 + 2, cbytes = CH * TS * 4;     // p2_ts(Q, CH), p2_chan_bytes(Q, CH)
	int32_t *sa = (int32_t *)(smem + (size_t)wave * 2 * cbytes), *sb = (int32_t *)(smem + (size_t)wave * 2 * cbytes + cbytes);
	const uint32_t cstride = P.ncslots;

	// ---- stage this quarter: sample q0 + i -> row i%CH, column i/CH + 1; column 0 = the four samples in front ---------
	{
		int2 v[CH];
#pragma unroll
		for(int r = 0; r < CH; r++) v[r] = p[(uint32_t)lane + 64u * (uint32_t)r];
#pragma unroll
		for(int r = 0; r < CH; r++) {
			const uint32_t i = (uint32_t)lane + 64u * (uint32_t)r;
			const uint32_t col = CH == 16 ? i >> 4 : i / (uint32_t)CH, row = i - col * (uint32_t)CH;
			const uint32_t a = row * TS + col + 1;
			sa[a] = v[r].x; sb[a] = v[r].y;
		}
		if(lane < CH) {
			int2 h = make_int2(0, 0);
			if(lane >= CH - 4 && wave) h = p[lane - CH];
			sa[(uint32_t)lane * TS] = h.x; sb[(uint32_t)lane * TS] = h.y;
		}
	}
	__builtin_amdgcn_wave_barrier();

	// ---- statistics of the four channels over this quarter: one CH-sample chunk per lane ------------------------------
	int32_t a[CH + 4], b[CH + 4];
	{
		const int32_t *pa = sa + lane, *pb = sb + lane;
#pragma unroll
		for(int k = 0; k < CH + 4; k++) { a[k] = k < 4 ? pa[(CH - 4 + k) * TS] : pa[(k - 4) * TS + 1]; b[k] = k < 4 ? pb[(CH - 4 + k) * TS] : pb[(k - 4) * TS + 1]; }
	}
	const bool first_chunk = wave == 0 && lane == 0;
	if(P.ms_mode == 2) {
		// loose mid/side (stream_encoder.c:3778-3807), bps < 25
		uint64_t lr_sum = 0, ms_sum = 0;
#pragma unroll
		for(int t = 0; t < CH; t++) {
			if(t > 0 || !first_chunk) {
				const int32_t pl = a[t + 4] - a[t + 3], pr = b[t + 4] - b[t + 3];
				lr_sum += (uint64_t)(uint32_t)(abs(pl) + abs(pr));
				ms_sum += (uint64_t)(uint32_t)(abs((pl + pr) >> 1) + abs(pl - pr));
			}
		}
		lr_sum = wave_sum_u50(lr_sum); ms_sum = wave_sum_u50(ms_sum);
		if(lane == 0) { part[wave].lr = lr_sum; part[wave].ms = ms_sum; }
	}
#pragma unroll
	for(int c = 0; c < 4; c++) {
		int32_t x[CH + 4];
#pragma unroll
		for(int k = 0; k < CH + 4; k++) x[k] = c == 0 ? a[k] : c == 1 ? b[k] : c == 2 ? ((a[k] + b[k]) >> 1) : (a[k] - b[k]);
		const int32_t first = __builtin_amdgcn_readfirstlane(x[4]);      // this quarter's first sample
		Prep2Acc A;
		A.orv = 0; A.diff = 0; A.mag = 0;
#pragma unroll
		for(int k = 0; k < 5; k++) A.e[k] = 0;
		// (a lane's 16 samples are summed in 32 bits whatever the sample width this kernel serves: a fourth difference of 25-bit
		//  samples -- the side channel of a 24-bit stream -- is below 2^28, sixteen of them below 2^32; only the totals across lanes
		//  and wavefronts need more.  Round 3 added every |difference| in 64 bits for such streams: two instructions instead of one,
		//  twenty times per sample and channel)
		// (eighteen fourth differences of a 25-bit side channel do not: that one chunk flavour adds in 64 bits, as prep2_kernel's does)
		if(c == 3) prep2_chunk<(WIDE && CH > 16), true, false, CH, true>(x, first_chunk, first, A, nullptr, nullptr, wave == 0);
		else prep2_chunk<false, false, false, CH, true>(x, first_chunk, first, A, nullptr, nullptr, wave == 0);
		A.orv = wave_or_u32(A.orv);
		A.diff = wave_or_u32(A.diff);
		if(c == 3) A.mag = wave_or_u32(A.mag);
#pragma unroll
		for(int k = 0; k < 5; k++) A.e[k] = WIDE ? wave_sum_u50(A.e[k]) : (uint64_t)wave_sum_u32((uint32_t)A.e[k]);   // !WIDE (bps <= 17, launch_prep2): 1152 x 2^21 at most
		if(lane == 0) {
			Prep3Part &pt = part[wave];
			pt.orv[c] = A.orv; pt.diff[c] = A.diff; pt.first[c] = first;
			if(c == 3) pt.mag = A.mag;
#pragma unroll
			for(int k = 0; k < 5; k++) pt.e[c][k] = A.e[k];
		}
		__builtin_amdgcn_sched_barrier(0);                             // one channel at a time: interleaving them spills
	}
	__syncthreads();

	// ---- wavefront c decides channel c (every lane holds the totals; fewer than four wavefronts: c, c + NW, ...) -----------------
#pragma unroll 1
	for(uint32_t c = wave; c < 4; c += NW) {
		const uint32_t which = c;                                   // 0 left, 1 right, 2 mid, 3 side
		uint32_t orv = 0, diff = 0, mag = 0;
		uint64_t e[5] = {0, 0, 0, 0, 0}, lr = 0, ms = 0;
		bool alleq0 = true;                                         // the LEFT channel is constant (limit_min_bitrate)
		const int32_t f0 = part[0].first[c];
		constexpr int UW = NW > 4 ? 2 : NW;                         // (eight parts' sums asked for at once: 25 registers spilled)
#pragma unroll UW
		for(int w = 0; w < NW; w++) {
			orv |= part[w].orv[c]; mag |= part[w].mag;
			diff |= part[w].diff[c] | (uint32_t)(part[w].first[c] ^ f0);
			for(int k = 0; k < 5; k++) e[k] += part[w].e[c][k];
			lr += part[w].lr; ms += part[w].ms;
			alleq0 = alleq0 && part[w].diff[0] == 0 && part[w].first[0] == part[0].first[0];
		}
		int32_t slot = (int32_t)c;
		if(P.ms_mode == 2) { const bool use_ms = !(lr < ms); slot = use_ms ? (c >= 2 ? (int32_t)c - 2 : -1) : (c < 2 ? (int32_t)c : -1); }
		uint32_t wasted = orv ? (uint32_t)(__ffs((int)orv) - 1) : 0;
		if(wasted > P.bps) wasted = P.bps;
		const uint32_t sbps = P.bps - wasted + (which == 3 ? 1 : 0);
		const uint32_t fmt = (sbps <= 16 || (which == 3 && (mag >> wasted) < 32768u)) ? 1u : 0u;
		if(lane == 0) { outp.wasted[c] = wasted; outp.slot[c] = slot; outp.fmt[c] = fmt; }
		if(slot >= 0) {
			bool disable_constant = P.disable_constant != 0;
			if(P.limit_min_bitrate && !disable_constant && (P.ms_mode == 2 ? which == 1 : which >= 1) && alleq0) disable_constant = true;
			uint32_t flags = 0, fixed_order = 0;
			int32_t constant = 0;
			const uint32_t verbatim_bits = P.disable_verbatim ? 0xffffffffu : 8 + wasted + N * sbps;
			const uint32_t n4 = N - 4;
			const uint64_t es[5] = {e[0] >> wasted, e[1] >> wasted, e[2] >> wasted, e[3] >> wasted, e[4] >> wasted};
			uint32_t guess_fixed;
			{
				const uint64_t m34 = es[3] < es[4] ? es[3] : es[4], m234 = es[2] < m34 ? es[2] : m34, m1234 = es[1] < m234 ? es[1] : m234;
				if(es[0] <= m1234) guess_fixed = 0;
				else if(es[1] <= m234) guess_fixed = 1;
				else if(es[2] <= m34) guess_fixed = 2;
				else if(es[3] <= es[4]) guess_fixed = 3;
				else guess_fixed = 4;
			}
			const bool is_constant = !disable_constant && diff == 0;
			const size_t fc = (size_t)f * P.ncand + (uint32_t)slot;
			if(is_constant) { flags |= PREP_CONSTANT; constant = f0 >> wasted; }
			else if(P.max_lpc_order > 0) flags |= PREP_LPC;
			const bool fixed_allowed = !is_constant && (!P.disable_fixed || (P.max_lpc_order == 0 && verbatim_bits == 0xffffffffu));
			fixed_order = fixed_allowed ? guess_fixed : 0;
			if(emit_fixed_candidates(P, &cands[fc * cstride], &valid[fc * cstride], es, n4, guess_fixed, fixed_allowed, sbps, lane)) flags |= PREP_FIXED_VALID;
			if(lane == 0) {
				ChanPrep pr;
				pr.which = which; pr.wasted = wasted; pr.sbps = sbps; pr.n = N; pr.flags = flags; pr.fixed_order = fixed_order;
				pr.constant = constant; pr.verbatim_bits = verbatim_bits; pr.fmt = fmt; pr.constant_hi = constant >> 31; pr.handled = 0; pr.pad = 0;
				preps[fc] = pr;
			}
		}
	}
	__syncthreads();

	// ---- planar channels of this quarter, shifted, straight from the registers ----------------------------------------
	{
		const uint32_t base = q0 + (uint32_t)lane * CH;
		// the quarter is read again from the LDS tile (conflict-free, behind the barrier) rather than kept in registers across
		// the decision phase: holding a[], b[] alive there cost 23 spilled VGPRs = 23 KB of scratch written and read per frame
		int32_t ra[CH], rb[CH];
		{
			const int32_t *pa = sa + lane, *pb = sb + lane;
#pragma unroll
			for(int k = 0; k < CH; k++) { ra[k] = pa[k * TS + 1]; rb[k] = pb[k * TS + 1]; }
		}
#pragma unroll
		for(int c = 0; c < 4; c++) {
			const int32_t slot = outp.slot[c];
			if(slot < 0) continue;
			const uint32_t wasted = outp.wasted[c];
			int32_t x[CH];
#pragma unroll
			for(int k = 0; k < CH; k++) x[k] = (c == 0 ? ra[k] : c == 1 ? rb[k] : c == 2 ? ((ra[k] + rb[k]) >> 1) : (ra[k] - rb[k])) >> wasted;
			uint32_t *dst = (uint32_t *)(chan + ((size_t)f * P.ncand + (uint32_t)slot) * (size_t)N);
			if constexpr(CH != CHUNK) {
				// 18 samples: nine words of pairs (or eighteen words) at a 36-byte (72-byte) lane stride: plain word stores
				if(outp.fmt[c]) {
#pragma unroll
					for(int j = 0; j < CH / 2; j++) dst[base / 2 + j] = ((uint32_t)x[2 * j] & 0xffffu) | ((uint32_t)x[2 * j + 1] << 16);
				}
				else {
#pragma unroll
					for(int j = 0; j < CH; j++) dst[base + j] = (uint32_t)x[j];
				}
			}
			else if(outp.fmt[c]) {
				uint4 w0, w1;
				w0.x = ((uint32_t)x[0] & 0xffffu) | ((uint32_t)x[1] << 16); w0.y = ((uint32_t)x[2] & 0xffffu) | ((uint32_t)x[3] << 16);
				w0.z = ((uint32_t)x[4] & 0xffffu) | ((uint32_t)x[5] << 16); w0.w = ((uint32_t)x[6] & 0xffffu) | ((uint32_t)x[7] << 16);
				w1.x = ((uint32_t)x[8] & 0xffffu) | ((uint32_t)x[9] << 16); w1.y = ((uint32_t)x[10] & 0xffffu) | ((uint32_t)x[11] << 16);
				w1.z = ((uint32_t)x[12] & 0xffffu) | ((uint32_t)x[13] << 16); w1.w = ((uint32_t)x[14] & 0xffffu) | ((uint32_t)x[15] << 16);
				uint4 *d4 = (uint4 *)(dst + base / 2);
				d4[0] = w0; d4[1] = w1;
			}
			else {
				uint4 *d4 = (uint4 *)(dst + base);
#pragma unroll
				for(int k = 0; k < 4; k++) { uint4 w; w.x = (uint32_t)x[4 * k]; w.y = (uint32_t)x[4 * k + 1]; w.z = (uint32_t)x[4 * k + 2]; w.w = (uint32_t)x[4 * k + 3]; d4[k] = w; }
			}
		}
	}
}
// NW x 64 x CH samples: 4096 as ever; round 6: 1024, 2048, 8192 (chunks of 16) and 1152, 2304, 4608 (chunks of 18) where an LPC search
// follows (without one the deciding prep2_kernel / ff_kernel own these sizes: prep2_decides)
static bool prep3_shape(uint32_t n, uint32_t &nw, uint32_t &ch)
{
	for(uint32_t c = 16; c <= 18; c += 2) for(uint32_t w = 1; w <= 8; w *= 2) if(n == w * 64 * c && !(c == 18 && w == 8)) { nw = w; ch = c; return true; }
	return false;
}
bool prep3_applicable(const DevParams &P)
{
	uint32_t nw, ch;
	if(!(P.channels == 2 && P.ms_mode != 0 && !P.wide_samples && prep3_shape(P.blocksize, nw, ch))) return false;
	return P.blocksize == 4096 || (P.max_lpc_order != 0 && !tune().no_prep3n);
}

// ---------------------------------------------------------------------------------------------------------
// prep4_kernel (round 5): frames of 4096 samples of 1..8 INDEPENDENT channels -- mono, stereo without a mid/side search, 3..8
// channels (the reference treats channels one by one, stream_encoder.c:3872-3900)
// ---------------------------------------------------------------------------------------------------------
// prep3_kernel's split of the work for the layouts prep2_kernel served with one wavefront per channel (mono: ONE wavefront walks the
// whole frame four chunks per lane, three passes over it, behind a workgroup-wide staging loop -- 0.31 ms per 16384 frames, more
// than stereo's four candidate channels take in prep3_kernel; 5.1: 1.25 ms): wavefront w owns the quarter [1024 w, 1024 w + 1024)
// of the frame for ALL channels -- stages it with coalesced word loads into its own transposed tiles (no barrier in front of the
// compute), takes the statistics of its one 16-sample chunk per lane channel by channel, and the workgroup meets twice: partial
// statistics, then the wasted bits before the planar channels are written.  Channel c is decided by wavefront c % 4.
struct Prep4Part { uint32_t orv[FLACGPU_MAX_CHANNELS], diff[FLACGPU_MAX_CHANNELS]; int32_t first[FLACGPU_MAX_CHANNELS]; uint64_t e[FLACGPU_MAX_CHANNELS][5]; };
struct Prep4Out { uint32_t wasted[FLACGPU_MAX_CHANNELS], fmt[FLACGPU_MAX_CHANNELS]; };

// (round 6: <., NW, CH> as prep3_kernel -- blocks of NW x 64 x CH samples)
template <bool WIDE, int NW = 4, int CH = CHUNK>
__global__ __launch_bounds__(64 * NW, 4) void prep4_kernel(const DevParams P, const int32_t *__restrict__ pcm, uint32_t nmain, uint32_t G,
                                                       ChanPrep *__restrict__ preps, Candidate *__restrict__ cands, int *__restrict__ valid,
                                                       int32_t *__restrict__ chan)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	__shared__ Prep4Part part[NW];
	__shared__ Prep4Out outp;
	const int tid = (int)threadIdx.x, lane = tid & 63;
	const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane(tid >> 6);
	constexpr uint32_t Q = 64 * CH, N = NW * Q;
	const uint32_t C = P.channels;
	const uint32_t f = blockIdx.x;
	const uint32_t q0 = wave * Q;
	const int32_t *p = pcm + ((size_t)f * N + q0) * C;                      // this quarter: Q * C consecutive words
	constexpr uint32_t TS = ((Q / CH - 1 + 31) / 32) * 32 + 2, cbytes = CH * TS * 4;     // p2_ts(Q, CH), p2_chan_bytes(Q, CH)
	unsigned char *tiles = smem + (size_t)wave * G * cbytes;               // [G] tiles of this wavefront
	const uint32_t cstride = P.ncslots;
	const bool first_chunk = wave == 0 && lane == 0;

	// The channels in rounds of G (all of them at once up to four; 5.1 in two rounds of three, 7 and 8 channels in two of four): the
	// tiles of a round are 17 KB per channel -- six channels at once left ONE workgroup per CU, one wavefront per SIMD, and the kernel
	// waited for its loads (1.16 ms per 16384 frames of 5.1; profiles/r05_f_chan_rate.txt)
#pragma unroll 1
	for(uint32_t cg = 0; cg < C; cg += G) {
		const uint32_t ng = C - cg < G ? C - cg : G;
		if(cg) __syncthreads();                                             // (the planes of the round before have been written from the tiles)
		// ---- stage this quarter's words of the round's channels: sample i -> tile[channel - cg] row i % 16, column i / 16 + 1 --------
		{
			const uint32_t nwords = Q * ng;
			const uint32_t ginv = ng > 1 ? 0xffffffffu / ng + 1u : 0u;           // ceil(2^32 / ng)
			constexpr int RB = CH == 16 ? 16 : CH / 2;                              // loads in flight per lane (eighteen at once spilled 14 registers)
			for(uint32_t m0 = 0; m0 < nwords; m0 += 64 * RB) {
				int32_t v[RB];
#pragma unroll
				for(int r = 0; r < RB; r++) {
					const uint32_t m = m0 + (uint32_t)lane + 64u * (uint32_t)r;   // (Q * ng is a whole number of passes of 64 x RB words)
					const uint32_t i = ng == 1 ? m : __umulhi(m, ginv), c = m - i * ng;      // m / ng (m < 4608: the reciprocal is exact)
					v[r] = p[i * C + cg + c];
				}
#pragma unroll
				for(int r = 0; r < RB; r++) {
					const uint32_t m = m0 + (uint32_t)lane + 64u * (uint32_t)r;
					const uint32_t i = ng == 1 ? m : __umulhi(m, ginv), c = m - i * ng;
					const uint32_t col = CH == 16 ? i >> 4 : i / (uint32_t)CH, row = i - col * (uint32_t)CH;
					((int32_t *)(tiles + (size_t)c * cbytes))[row * TS + col + 1] = v[r];
				}
			}
			// column 0 = the four samples in front of the quarter (rows CH - 4 .. CH - 1), zeros in front of the block
			for(uint32_t t = (uint32_t)lane; t < (uint32_t)CH * ng; t += 64) {
				const uint32_t c = t / (uint32_t)CH, r = t - c * (uint32_t)CH;
				int32_t h = 0;
				if(r >= (uint32_t)CH - 4 && wave) h = p[((int32_t)r - CH) * (int32_t)C + (int32_t)(cg + c)];
				((int32_t *)(tiles + (size_t)c * cbytes))[r * TS] = h;
			}
		}
		__builtin_amdgcn_wave_barrier();

		// ---- statistics over this quarter, channel by channel: one 16-sample chunk per lane ---------------------------------------
#pragma unroll 1
		for(uint32_t cl = 0; cl < ng; cl++) {
			const uint32_t c = cg + cl;
			const int32_t *pa = (const int32_t *)(tiles + (size_t)cl * cbytes) + lane;
			int32_t x[CH + 4];
#pragma unroll
			for(int k = 0; k < CH + 4; k++) x[k] = k < 4 ? pa[(CH - 4 + k) * TS] : pa[(k - 4) * TS + 1];
			const int32_t first = __builtin_amdgcn_readfirstlane(x[4]);      // this quarter's first sample
			Prep2Acc A;
			A.orv = 0; A.diff = 0; A.mag = 0;
#pragma unroll
			for(int k = 0; k < 5; k++) A.e[k] = 0;
			prep2_chunk<false, false, false, CH>(x, first_chunk, first, A);    // (a lane's sixteen or eighteen differences fit 32 bits at any width served here: 18 x 2^27)
			A.orv = wave_or_u32(A.orv);
			A.diff = wave_or_u32(A.diff);
#pragma unroll
			for(int k = 0; k < 5; k++) A.e[k] = WIDE ? wave_sum_u50(A.e[k]) : (uint64_t)wave_sum_u32((uint32_t)A.e[k]);
			if(lane == 0) {
				Prep4Part &pt = part[wave];
				pt.orv[c] = A.orv; pt.diff[c] = A.diff; pt.first[c] = first;
#pragma unroll
				for(int k = 0; k < 5; k++) pt.e[c][k] = A.e[k];
			}
		}
		__syncthreads();

		// ---- wavefront w decides the round's channels cg + w (every lane holds the totals) -------------------------------------------
		for(uint32_t c = cg + wave; c < cg + ng; c += NW) {
			uint32_t orv = 0, diff = 0;
			uint64_t e[5] = {0, 0, 0, 0, 0};
			const int32_t f0 = part[0].first[c];
			constexpr int UW = NW > 4 ? 2 : NW;
#pragma unroll UW
			for(int w = 0; w < NW; w++) {
				orv |= part[w].orv[c];
				diff |= part[w].diff[c] | (uint32_t)(part[w].first[c] ^ f0);
				for(int k = 0; k < 5; k++) e[k] += part[w].e[c][k];
			}
			uint32_t wasted = orv ? (uint32_t)(__ffs((int)orv) - 1) : 0;
			if(wasted > P.bps) wasted = P.bps;
			const uint32_t sbps = P.bps - wasted;
			const uint32_t fmt = sbps <= 16 ? 1u : 0u;
			if(lane == 0) { outp.wasted[c] = wasted; outp.fmt[c] = fmt; }
			bool disable_constant = P.disable_constant != 0;
			if(P.limit_min_bitrate && !disable_constant && c + 1 == C) {
				// limit_min_bitrate (stream_encoder.c:3874-3879): the last channel is not CONSTANT when every channel in front of it is
				// (the last channel is in the last round: the records of all the others are there)
				bool others_constant = true;
				for(uint32_t o = 0; o + 1 < C; o++)
					for(int w = 0; w < NW; w++) others_constant = others_constant && part[w].diff[o] == 0 && part[w].first[o] == part[0].first[o];
				if(others_constant) disable_constant = true;
			}
			uint32_t flags = 0, fixed_order = 0;
			int32_t constant = 0;
			const uint32_t verbatim_bits = P.disable_verbatim ? 0xffffffffu : 8 + wasted + N * sbps;
			const uint32_t n4 = N - 4;
			const uint64_t es[5] = {e[0] >> wasted, e[1] >> wasted, e[2] >> wasted, e[3] >> wasted, e[4] >> wasted};
			uint32_t guess_fixed;
			{
				const uint64_t m34 = es[3] < es[4] ? es[3] : es[4], m234 = es[2] < m34 ? es[2] : m34, m1234 = es[1] < m234 ? es[1] : m234;
				if(es[0] <= m1234) guess_fixed = 0;
				else if(es[1] <= m234) guess_fixed = 1;
				else if(es[2] <= m34) guess_fixed = 2;
				else if(es[3] <= es[4]) guess_fixed = 3;
				else guess_fixed = 4;
			}
			const bool is_constant = !disable_constant && diff == 0;
			const size_t fc = (size_t)f * P.ncand + c;
			if(is_constant) { flags |= PREP_CONSTANT; constant = f0 >> wasted; }
			else if(P.max_lpc_order > 0) flags |= PREP_LPC;
			const bool fixed_allowed = !is_constant && (!P.disable_fixed || (P.max_lpc_order == 0 && verbatim_bits == 0xffffffffu));
			fixed_order = fixed_allowed ? guess_fixed : 0;
			if(emit_fixed_candidates(P, &cands[fc * cstride], &valid[fc * cstride], es, n4, guess_fixed, fixed_allowed, sbps, lane)) flags |= PREP_FIXED_VALID;
			if(lane == 0) {
				ChanPrep pr;
				pr.which = c; pr.wasted = wasted; pr.sbps = sbps; pr.n = N; pr.flags = flags; pr.fixed_order = fixed_order;
				pr.constant = constant; pr.verbatim_bits = verbatim_bits; pr.fmt = fmt; pr.constant_hi = constant >> 31; pr.handled = 0; pr.pad = 0;
				preps[fc] = pr;
			}
		}
		__syncthreads();

		// ---- planar channels of this quarter, shifted: the tiles are read again (conflict-free, behind the barrier) -----------------
		{
			const uint32_t base = q0 + (uint32_t)lane * CH;
#pragma unroll 1
			for(uint32_t cl = 0; cl < ng; cl++) {
				const uint32_t c = cg + cl;
				const int32_t *pa = (const int32_t *)(tiles + (size_t)cl * cbytes) + lane;
				const uint32_t wasted = outp.wasted[c];
				int32_t x[CH];
#pragma unroll
				for(int k = 0; k < CH; k++) x[k] = pa[k * TS + 1] >> wasted;
				uint32_t *dst = (uint32_t *)(chan + ((size_t)f * P.ncand + c) * (size_t)N);
				if constexpr(CH != CHUNK) {
					if(outp.fmt[c]) {
#pragma unroll
						for(int j2 = 0; j2 < CH / 2; j2++) dst[base / 2 + j2] = ((uint32_t)x[2 * j2] & 0xffffu) | ((uint32_t)x[2 * j2 + 1] << 16);
					}
					else {
#pragma unroll
						for(int j2 = 0; j2 < CH; j2++) dst[base + j2] = (uint32_t)x[j2];
					}
				}
				else if(outp.fmt[c]) {
					uint4 w0, w1;
					w0.x = ((uint32_t)x[0] & 0xffffu) | ((uint32_t)x[1] << 16); w0.y = ((uint32_t)x[2] & 0xffffu) | ((uint32_t)x[3] << 16);
					w0.z = ((uint32_t)x[4] & 0xffffu) | ((uint32_t)x[5] << 16); w0.w = ((uint32_t)x[6] & 0xffffu) | ((uint32_t)x[7] << 16);
					w1.x = ((uint32_t)x[8] & 0xffffu) | ((uint32_t)x[9] << 16); w1.y = ((uint32_t)x[10] & 0xffffu) | ((uint32_t)x[11] << 16);
					w1.z = ((uint32_t)x[12] & 0xffffu) | ((uint32_t)x[13] << 16); w1.w = ((uint32_t)x[14] & 0xffffu) | ((uint32_t)x[15] << 16);
					uint4 *d4 = (uint4 *)(dst + base / 2);
					d4[0] = w0; d4[1] = w1;
				}
				else {
					uint4 *d4 = (uint4 *)(dst + base);
#pragma unroll
					for(int k = 0; k < 4; k++) { uint4 w; w.x = (uint32_t)x[4 * k]; w.y = (uint32_t)x[4 * k + 1]; w.z = (uint32_t)x[4 * k + 2]; w.w = (uint32_t)x[4 * k + 3]; d4[k] = w; }
				}
			}
		}
	}
}
// every channel a candidate channel of its own (no mid/side: ncand == channels), the block size the tiles are built for, and the whole
// frame in the workgroup's LDS (8 channels: 135 KB)
bool prep4_applicable(const DevParams &P)
{
	uint32_t nw, ch;
	if(!(P.ms_mode == 0 && P.ncand == P.channels && !P.wide_samples && !P.stream_sig && prep3_shape(P.blocksize, nw, ch))) return false;
	return P.blocksize == 4096 || !tune().no_prep3n;                  // (round 6: the other sizes of prep3_shape)
}


// the kernel above serves frames of nominal length when every lane run is whole and the AVX2 short-tail quirk of
// the reference's wide fixed-predictor routine cannot occur (fixed_intrin_avx2.c:57 with (n-4) % 4 != 0)
bool prep2_applicable(const DevParams &P)
{
	const uint32_t nraw = (P.channels == 2 && P.ms_mode != 0) ? 2u : (P.channels < 4 ? P.channels : 4u);
	return P.blocksize % 16 == 0 && P.blocksize > 4 && (size_t)nraw * p2_chan_bytes(P.blocksize) <= 150 * 1024 && !P.wide_samples && !P.stream_sig;
}

// prep2_kernel<.,.,true>: the presets whose only residual candidate is the guessed fixed order.  The leaf partitions must be whole
// 16-sample chunks, the partition sums the reference's 32-bit ones (stream_encoder.c:4814), and eval_list_kernel (the lane-owner
// evaluation) must be able to take what this kernel leaves behind.
// the deciding kernel's wide flavour: chunk sums and leaf sums in 64 bits (17..24-bit input; the side channel has 25)
static bool prep2_decide_wide(const DevParams &P) { return P.bps > 16; }
static size_t prep2_decide_lds(const DevParams &P, uint32_t nraw, uint32_t waves)
{
	const uint32_t ch = p2_chunk_len(P.blocksize);
	return (size_t)nraw * p2_chan_bytes(P.blocksize, ch) + P2_DIVTAB_BYTES + (size_t)waves * (5 * (P.blocksize / ch) * (prep2_decide_wide(P) ? 8 : 4) + 8 * 4 + 64);
}
bool prep2_decides(const DevParams &P)
{
	const int off = tune().no_prep_decide;
	if(off || !prep2_applicable(P) || prep3_applicable(P)) return false;
	if(P.max_lpc_order != 0 || P.nfixed != 1 || P.ncslots != 1 || P.bps > 24 || P.tune_flags) return false;
	if(P.bps > 16 && tune().no_wide_decide) return false;
	const uint32_t n = P.blocksize;
	uint32_t fmax = 0;
	{ uint32_t b = n; while(!(b & 1)) { fmax++; b >>= 1; } }
	if(fmax > P.max_po) fmax = P.max_po;
	if(fmax > 6 || n % 64 != 0 || n / 64 < 16 || (n / 64) % 2 != 0) return false;
	const uint32_t psize = n >> fmax;
	uint32_t lg = 0;
	while((2u << lg) <= psize) lg++;
	// (the 32-bit flavour: the partition sums are the reference's 32-bit ones, stream_encoder.c:4814; the wide one carries 64 bits)
	if(!(psize % p2_chunk_len(n) == 0 && (prep2_decide_wide(P) || (P.bps + 1 + 4) < 32 - lg))) return false;
	// the chunk sums of every wavefront behind the staged channels must still fit the LDS (6 channels x 8192 samples do not)
	const bool stereo_ms = P.channels == 2 && P.ms_mode != 0;
	const uint32_t nraw = stereo_ms ? 2u : (P.channels < 4 ? P.channels : 4u), waves = stereo_ms ? 4u : nraw;
	return prep2_decide_lds(P, nraw, waves) <= 150 * 1024;
}

hipError_t launch_prep2(const DevParams &P, const int32_t *pcm, uint32_t nmain, const AnalyzeBuffers &B, SubDecision *dec, hipStream_t s)
{
	if(nmain == 0) {
		// a batch that is only a short last block: the list the deciding prep kernel would have started must still start empty
		if(prep2_decides(P)) (void)hipMemsetAsync(B.nleft, 0, 2 * sizeof(uint32_t), s);
		return hipSuccess;
	}
	static AttrFlags attr_set;
	if(AttrOnce once{attr_set}) {
		hipError_t e = hipSuccess;
#define P2ATTR(W, NF, DZ) if(e == hipSuccess) e = hipFuncSetAttribute((const void *)prep2_kernel<W, NF, DZ>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024)
		P2ATTR(false, 0, false); P2ATTR(false, 1152, false); P2ATTR(false, 4096, false); P2ATTR(true, 0, false); P2ATTR(true, 1152, false); P2ATTR(true, 4096, false);
		P2ATTR(false, 0, true); P2ATTR(false, 1152, true); P2ATTR(true, 0, true); P2ATTR(true, 1152, true);
#undef P2ATTR
		if(e != hipSuccess) return e;
		once.ok();
	}
	if(prep4_applicable(P) && !tune().no_fast1 && !tune().no_prep4 && !prep2_decides(P)) {
		uint32_t nw = 4, ch = 16;
		(void)prep3_shape(P.blocksize, nw, ch);
		// channels per round: all of them up to four, else the rounds as even as they come (5, 6 -> 3; 7, 8 -> 4); the eight-wavefront
		// instance two at most (its tiles are 34 KB per channel of a round)
		const uint32_t C = P.channels;
		uint32_t G = C <= 4 ? C : (C + 1) / 2;
		if(nw == 8 && G > 2) G = 2;
		const size_t lds4 = (size_t)nw * G * p2_chan_bytes(64 * ch, ch);
		static AttrFlags attr4;
		if(AttrOnce once{attr4}) {
			// (at most four channels' tiles at a time: 68 KB, 76 KB with 18-sample chunks; the kernel's static LDS -- the partial records -- is 1.7 KB, 3.4 with eight wavefronts)
			hipError_t e = hipSuccess;
#define P4ATTR(W, NWV, CHV) if(e == hipSuccess) e = hipFuncSetAttribute((const void *)prep4_kernel<W, NWV, CHV>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024)
			P4ATTR(false, 1, 16); P4ATTR(false, 2, 16); P4ATTR(false, 4, 16); P4ATTR(false, 8, 16); P4ATTR(false, 1, 18); P4ATTR(false, 2, 18); P4ATTR(false, 4, 18);
			P4ATTR(true, 1, 16); P4ATTR(true, 2, 16); P4ATTR(true, 4, 16); P4ATTR(true, 8, 16); P4ATTR(true, 1, 18); P4ATTR(true, 2, 18); P4ATTR(true, 4, 18);
#undef P4ATTR
			if(e != hipSuccess) return e;
		once.ok();
		}
		note_launch(K_PREP1);
		// (the wavefront's sum of 1024 (1152) fourth differences: |d4| < 2^(bps+3), so 32 bits hold it up to 18-bit samples only -- ADVICE r05: at
		//  20 bits a Nyquist alternation at half of full scale wrapped e[4] and fixed order 4 was guessed instead of 0)
		const bool wide = P.bps > 18;
#define P4GO(NWV, CHV) do { if(wide) hipLaunchKernelGGL((prep4_kernel<true, NWV, CHV>), dim3(nmain), dim3(64 * NWV), lds4, s, P, pcm, nmain, G, B.prep, B.cands, B.valid, B.chan); \
		                    else hipLaunchKernelGGL((prep4_kernel<false, NWV, CHV>), dim3(nmain), dim3(64 * NWV), lds4, s, P, pcm, nmain, G, B.prep, B.cands, B.valid, B.chan); } while(0)
		if(ch == 16) { if(nw == 1) P4GO(1, 16); else if(nw == 2) P4GO(2, 16); else if(nw == 4) P4GO(4, 16); else P4GO(8, 16); }
		else { if(nw == 1) P4GO(1, 18); else if(nw == 2) P4GO(2, 18); else P4GO(4, 18); }
#undef P4GO
		return hipGetLastError();
	}
	if(prep3_applicable(P) && !tune().no_prep3) {
		uint32_t nw = 4, ch = 16;
		(void)prep3_shape(P.blocksize, nw, ch);
		static AttrFlags attr3;
		if(AttrOnce once{attr3}) {
			hipError_t e = hipSuccess;
			// (what the instance asks for, no more: the eight-wavefront one has 1.9 KB of static LDS beside it, and 159 KB + that is over the CU's)
#define P3ATTR(W, NWV, CHV) if(e == hipSuccess) e = hipFuncSetAttribute((const void *)prep3_kernel<W, NWV, CHV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * NWV * p2_chan_bytes(64 * CHV, CHV)))
			P3ATTR(false, 1, 16); P3ATTR(false, 2, 16); P3ATTR(false, 4, 16); P3ATTR(false, 8, 16); P3ATTR(false, 1, 18); P3ATTR(false, 2, 18); P3ATTR(false, 4, 18);
			P3ATTR(true, 1, 16); P3ATTR(true, 2, 16); P3ATTR(true, 4, 16); P3ATTR(true, 8, 16); P3ATTR(true, 1, 18); P3ATTR(true, 2, 18); P3ATTR(true, 4, 18);
#undef P3ATTR
			if(e != hipSuccess) return e;
		once.ok();
		}
		note_launch(K_PREP3);
		const size_t lds3 = 2 * (size_t)nw * p2_chan_bytes(64 * ch, ch);
		// (the side channel has bps + 1 bits: its part's sum of fourth differences is below 2^(bps+14) -- 1152 of them: 1.125 x that --, 32 bits up to 17-bit input)
		const bool wide = P.bps > 17;
#define P3GO(NWV, CHV) do { if(wide) hipLaunchKernelGGL((prep3_kernel<true, NWV, CHV>), dim3(nmain), dim3(64 * NWV), lds3, s, P, pcm, nmain, B.prep, B.cands, B.valid, B.chan); \
		                    else hipLaunchKernelGGL((prep3_kernel<false, NWV, CHV>), dim3(nmain), dim3(64 * NWV), lds3, s, P, pcm, nmain, B.prep, B.cands, B.valid, B.chan); } while(0)
		if(ch == 16) { if(nw == 1) P3GO(1, 16); else if(nw == 2) P3GO(2, 16); else if(nw == 4) P3GO(4, 16); else P3GO(8, 16); }
		else { if(nw == 1) P3GO(1, 18); else if(nw == 2) P3GO(2, 18); else P3GO(4, 18); }
#undef P3GO
		return hipGetLastError();
	}
	const bool stereo_ms = P.channels == 2 && P.ms_mode != 0;
	const uint32_t nraw = stereo_ms ? 2u : (P.channels < 4 ? P.channels : 4u);
	// wavefronts that own a channel (one per raw channel of a round; four with mid/side: L, R, M, S) = the workgroup.  (Round 5 tried
	// four wavefronts always, so that 256 threads stage a frame whatever the channel count: mono at -0 went 0.164 -> 0.252 ms per 33 M
	// samples -- the three idle wavefronts take the register file's room from nine more one-wavefront workgroups per CU.  What this
	// kernel keeps from that round is the staging loop with four samples in flight per thread.)
	const uint32_t active = stereo_ms ? 4u : nraw;
	const uint32_t waves = active;
	const size_t lds = (size_t)nraw * p2_chan_bytes(P.blocksize, p2_chunk_len(P.blocksize));
#define P2GO(W, NF) hipLaunchKernelGGL((prep2_kernel<W, NF, false>), dim3(nmain), dim3(64 * waves), lds, s, P, pcm, nmain, B.prep, B.cands, B.valid, B.chan, dec, B.left, B.nleft)
	note_launch(K_PREP2);
	if(prep2_decides(P)) {
		note_launch(K_PREP2_DECIDE);
		// (launch_model_eval then runs eval_list_kernel on what is left, and nothing else)
		(void)hipMemsetAsync(B.nleft, 0, 2 * sizeof(uint32_t), s);
		const size_t ldz = prep2_decide_lds(P, nraw, active);
		if(prep2_decide_wide(P)) {
			if(P.blocksize == 1152) hipLaunchKernelGGL((prep2_kernel<true, 1152, true>), dim3(nmain), dim3(64 * waves), ldz, s, P, pcm, nmain, B.prep, B.cands, B.valid, B.chan, dec, B.left, B.nleft);
			else hipLaunchKernelGGL((prep2_kernel<true, 0, true>), dim3(nmain), dim3(64 * waves), ldz, s, P, pcm, nmain, B.prep, B.cands, B.valid, B.chan, dec, B.left, B.nleft);
		}
		else if(P.blocksize == 1152) hipLaunchKernelGGL((prep2_kernel<false, 1152, true>), dim3(nmain), dim3(64 * waves), ldz, s, P, pcm, nmain, B.prep, B.cands, B.valid, B.chan, dec, B.left, B.nleft);
		else hipLaunchKernelGGL((prep2_kernel<false, 0, true>), dim3(nmain), dim3(64 * waves), ldz, s, P, pcm, nmain, B.prep, B.cands, B.valid, B.chan, dec, B.left, B.nleft);
		return hipGetLastError();
	}
	if(P.bps > 20) { if(P.blocksize == 4096) P2GO(true, 4096); else if(P.blocksize == 1152) P2GO(true, 1152); else P2GO(true, 0); }
	else { if(P.blocksize == 4096) P2GO(false, 4096); else if(P.blocksize == 1152) P2GO(false, 1152); else P2GO(false, 0); }
#undef P2GO
	return hipGetLastError();
}

} // namespace flacgpu
