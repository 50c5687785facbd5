// flac_amd/csrc/flacgpu_evalg.hip -- the residual evaluation of process_subframe_ (stream_encoder.c:4147-4290) for the
// case every preset produces: one WAVEFRONT per (frame, candidate channel), every residual candidate of the channel
// evaluated by that one wavefront.
//
// Why (round 3): the wavefront-per-candidate kernel (flacgpu_analyze.hip: eval_kernel<.,0>) is bound by the number of VALU
// instructions it issues, and a third of them were not arithmetic the reference asks for: every candidate formed the
// same shifted sample words from the same LDS image, formed x + bias per sample, evaluated 448 Rice nodes for a tree
// of 127, and paid its own dispatch.  Here
//   * lane L owns samples [L*S, (L+1)*S) of the block (same owner layout), and the candidates of a channel are taken
//     TWO AT A TIME through the block: the sample words and their one-sample-shifted companions are loaded / formed once
//     per pair;
//   * the sample itself rides in the tap chain: with t = sum_k q_k x[i-k] - 2^s x[i] (one more 16-bit tap, -2^s fits
//     int16 for s <= 15) the residual is -(t >> s) exactly (arithmetic shift; 2^s x[i] is a multiple of 2^s), so
//     |residual| = |((t + 2^31) >> s logical) - 2^(31-s)| is one v_sad_u32 against a constant -- no x + bias per sample.
//     Exact as long as t does not leave 32 bits: checked per candidate from its coefficients
//     (2^(sbps-1) * (sum|q| + 2^s) < 2^31); a channel with a candidate that fails it is left to eval_kernel;
//   * the taps, shifts and biases of the pair live in SGPRs (v_dot2_i32_i16 takes one scalar operand);
//   * the Rice search (stream_encoder.c:4701-5075) of a pair evaluates every node of both partition trees ONCE: the leaf
//     sums go through one prefix sum over the lanes into LDS, a node's sum is the difference of two entries, and the
//     2 x 127 nodes are spread over four lane passes (leaves of candidate 0 | leaves of candidate 1 | the two
//     32-node levels | everything above) instead of seven passes per candidate;
//   * no workgroup: a wavefront has its own LDS image and meets nobody (no barriers).
// Same integers as eval_kernel throughout (same FIR low 32 bits, same |residual| sums, same set_partitioned_rice_
// arithmetic, same first-minimum tie rule).  Channels this kernel does not take -- 17..25-bit samples, 64-bit or
// overflow-checked candidates, sums that leave the 32-bit Rice arithmetic, block lengths whose lane runs are no
// multiple of 16 -- stay with eval_kernel, which is launched after it and skips what ChanPrep::handled marks.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "flacgpu_evalg.h"
#include "flacgpu_evalg_chain.h"

namespace flacgpu {

// pairs of folded taps a kernel instance provides for: 7 (predictors of at most 12 taps + the sample: every preset), 17 (-l 13..32,
// round 6).  The window of a 16-sample piece is NPM words of history in front of its 8.
template <int MAXORD> constexpr int eg_npm() { return MAXORD <= 12 ? 7 : 17; }

// ---- the FIR of one candidate over one 16-sample piece of every lane's run ------------------------------------------
// AA[j]: the word holding samples (2j - 14, 2j - 13) relative to the piece start, BB[m] = samples (2m - 13, 2m - 12).
// Folded taps c_0 = -2^shift, c_j = q[j-1]; Q[p] = (c_2p << 16) | (c_2p+1 & 0xffff): the high half multiplies the nearer
// sample.  Sample s of the piece: pairs p = 0..NPF-1 are (x[s-2p], x[s-2p-1]) = AA[(s+13)/2 - p] for odd s,
// BB[s/2 + 6 - p] for even s.  NPF dependent v_dot2_i32_i16 and the logical shift are ONE asm statement (between separate
// statements the compiler pads every dependent pair with an s_nop, flacgpu_devfn.h: dot2_chain_lshr).
template <int NPF, int NQ>
__device__ __forceinline__ uint32_t dot2_chain_s(const uint32_t (&W)[NPF], const uint32_t (&Q)[NQ], uint32_t sum0, uint32_t shift)
{
	if constexpr(NPF >= 8) return dot2_chain_long<NPF, NQ>(W, Q, sum0, shift);
	uint32_t d;
	if constexpr(NPF == 1) asm("v_dot2_i32_i16 %0, %2, %3, %1\n\tv_lshrrev_b32 %0, %4, %0" : "=&v"(d) : "v"(sum0), "v"(W[0]), "s"(Q[0]), "s"(shift));
	if constexpr(NPF == 2) asm("v_dot2_i32_i16 %0, %2, %3, %1\n\tv_dot2_i32_i16 %0, %4, %5, %0\n\tv_lshrrev_b32 %0, %6, %0" : "=&v"(d) : "v"(sum0), "v"(W[0]), "s"(Q[0]), "v"(W[1]), "s"(Q[1]), "s"(shift));
	if constexpr(NPF == 3) asm("v_dot2_i32_i16 %0, %2, %3, %1\n\tv_dot2_i32_i16 %0, %4, %5, %0\n\tv_dot2_i32_i16 %0, %6, %7, %0\n\tv_lshrrev_b32 %0, %8, %0" : "=&v"(d) : "v"(sum0), "v"(W[0]), "s"(Q[0]), "v"(W[1]), "s"(Q[1]), "v"(W[2]), "s"(Q[2]), "s"(shift));
	if constexpr(NPF == 4) asm("v_dot2_i32_i16 %0, %2, %3, %1\n\tv_dot2_i32_i16 %0, %4, %5, %0\n\tv_dot2_i32_i16 %0, %6, %7, %0\n\tv_dot2_i32_i16 %0, %8, %9, %0\n\tv_lshrrev_b32 %0, %10, %0" : "=&v"(d) : "v"(sum0), "v"(W[0]), "s"(Q[0]), "v"(W[1]), "s"(Q[1]), "v"(W[2]), "s"(Q[2]), "v"(W[3]), "s"(Q[3]), "s"(shift));
	if constexpr(NPF == 5) asm("v_dot2_i32_i16 %0, %2, %3, %1\n\tv_dot2_i32_i16 %0, %4, %5, %0\n\tv_dot2_i32_i16 %0, %6, %7, %0\n\tv_dot2_i32_i16 %0, %8, %9, %0\n\tv_dot2_i32_i16 %0, %10, %11, %0\n\tv_lshrrev_b32 %0, %12, %0" : "=&v"(d) : "v"(sum0), "v"(W[0]), "s"(Q[0]), "v"(W[1]), "s"(Q[1]), "v"(W[2]), "s"(Q[2]), "v"(W[3]), "s"(Q[3]), "v"(W[4]), "s"(Q[4]), "s"(shift));
	if constexpr(NPF == 6) asm("v_dot2_i32_i16 %0, %2, %3, %1\n\tv_dot2_i32_i16 %0, %4, %5, %0\n\tv_dot2_i32_i16 %0, %6, %7, %0\n\tv_dot2_i32_i16 %0, %8, %9, %0\n\tv_dot2_i32_i16 %0, %10, %11, %0\n\tv_dot2_i32_i16 %0, %12, %13, %0\n\tv_lshrrev_b32 %0, %14, %0" : "=&v"(d) : "v"(sum0), "v"(W[0]), "s"(Q[0]), "v"(W[1]), "s"(Q[1]), "v"(W[2]), "s"(Q[2]), "v"(W[3]), "s"(Q[3]), "v"(W[4]), "s"(Q[4]), "v"(W[5]), "s"(Q[5]), "s"(shift));
	if constexpr(NPF == 7) asm("v_dot2_i32_i16 %0, %2, %3, %1\n\tv_dot2_i32_i16 %0, %4, %5, %0\n\tv_dot2_i32_i16 %0, %6, %7, %0\n\tv_dot2_i32_i16 %0, %8, %9, %0\n\tv_dot2_i32_i16 %0, %10, %11, %0\n\tv_dot2_i32_i16 %0, %12, %13, %0\n\tv_dot2_i32_i16 %0, %14, %15, %0\n\tv_lshrrev_b32 %0, %16, %0" : "=&v"(d) : "v"(sum0), "v"(W[0]), "s"(Q[0]), "v"(W[1]), "s"(Q[1]), "v"(W[2]), "s"(Q[2]), "v"(W[3]), "s"(Q[3]), "v"(W[4]), "s"(Q[4]), "v"(W[5]), "s"(Q[5]), "v"(W[6]), "s"(Q[6]), "s"(shift));
	return d;
}
__device__ __forceinline__ uint32_t sad_u32_s(uint32_t a, uint32_t b_uniform, uint32_t c)       // |a - b| + c, b in an SGPR
{
	uint32_t d;
	asm("v_sad_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(b_uniform), "v"(c));
	return d;
}
// PC: the piece's number when it is one of the first (lane 0's first `order` samples are warm-up, not residual: pieces 0 and, from 17
// taps up, 1), -1 for every other piece
// NV: samples of the piece that belong to the lane's run (16; 8 for the half piece that ends a run of 16 k + 8 samples: blocks of
// 4608 samples)
// AA[j]: the word holding samples (2 j - 2 NPM, 2 j - 2 NPM + 1) relative to the piece start, BB[m] = samples (2 m - 2 NPM + 1, 2 m - 2 NPM + 2)
template <int NPF, int PC, int NV, int NPM>
__device__ __forceinline__ uint32_t fir16_folded(const uint32_t (&AA)[NPM + 8], const uint32_t (&BB)[NPM + 7], const uint32_t (&Q)[NPM], uint32_t shift, uint32_t bias, uint32_t order,
                                                 bool lane0, uint32_t sum0, uint32_t acc)
{
#pragma unroll
	for(int s = 0; s < NV; s++) {
		uint32_t W[NPF];
#pragma unroll
		for(int p = 0; p < NPF; p++) W[p] = (s & 1) ? AA[(s + 2 * NPM - 1) / 2 - p] : BB[s / 2 + NPM - 1 - p];
		uint32_t pb = dot2_chain_s<NPF, NPM>(W, Q, sum0, shift);
		if(PC >= 0 && 16 * PC + s < 2 * NPF - 1) { if(lane0 && (uint32_t)(16 * PC + s) < order) pb = bias; }
		acc = sad_u32_s(pb, bias, acc);
	}
	return acc;
}
template <int PC, int NV, int NPM>
__device__ __forceinline__ uint32_t fir16_dispatch(uint32_t npf, const uint32_t (&AA)[NPM + 8], const uint32_t (&BB)[NPM + 7], const uint32_t (&Q)[NPM], uint32_t shift, uint32_t bias,
                                                   uint32_t order, bool lane0, uint32_t sum0, uint32_t acc)
{
#define EG_CASE(n) case n: return fir16_folded<n, PC, NV, NPM>(AA, BB, Q, shift, bias, order, lane0, sum0, acc);
	if constexpr(NPM == 7) {
		switch(npf) {
		EG_CASE(1) EG_CASE(2) EG_CASE(3) EG_CASE(4) EG_CASE(5) EG_CASE(6)
		default: return fir16_folded<7, PC, NV, NPM>(AA, BB, Q, shift, bias, order, lane0, sum0, acc);
		}
	}
	else {
		switch(npf) {
		EG_CASE(1) EG_CASE(2) EG_CASE(3) EG_CASE(4) EG_CASE(5) EG_CASE(6) EG_CASE(7) EG_CASE(8) EG_CASE(9) EG_CASE(10) EG_CASE(11) EG_CASE(12)
		EG_CASE(13) EG_CASE(14) EG_CASE(15) EG_CASE(16)
		default: return fir16_folded<17, PC, NV, NPM>(AA, BB, Q, shift, bias, order, lane0, sum0, acc);
		}
	}
#undef EG_CASE
}
// base: byte address of word (first sample of the piece - 2 NPM samples) in the lane's column; rows at immediate offsets
template <int NPM>
__device__ __forceinline__ void load_piece(const unsigned char *base, uint32_t (&AA)[NPM + 8], uint32_t (&BB)[NPM + 7])
{
#pragma unroll
	for(int j = 0; j < NPM + 8; j++) AA[j] = *(const uint32_t *)(base + j * EG_ROW);
#pragma unroll
	for(int m = 0; m < NPM + 7; m++) BB[m] = __builtin_amdgcn_alignbit(AA[m + 1], AA[m], 16);
}
// one of the pieces that open the block (piece PC, 8 PC < NPM): the words in front of the run come from the previous lane's column
template <int NPM, int PC>
__device__ __forceinline__ void load_piece_head(const unsigned char *own /* word 0 of the run */, const unsigned char *prev_end /* word `run words` of the previous column: one past its last */,
                                                uint32_t (&AA)[NPM + 8], uint32_t (&BB)[NPM + 7])
{
#pragma unroll
	for(int j = 0; j < NPM + 8; j++) {
		constexpr int dummy = 0; (void)dummy;
		const int w = 8 * PC - NPM + j;                      // word of the run (negative: of the previous lane's run, counted from its end)
		AA[j] = w < 0 ? *(const uint32_t *)(prev_end + w * (int)EG_ROW) : *(const uint32_t *)(own + w * (int)EG_ROW);
	}
#pragma unroll
	for(int m = 0; m < NPM + 7; m++) BB[m] = __builtin_amdgcn_alignbit(AA[m + 1], AA[m], 16);
}

// one slot of the pair: a residual candidate's folded taps and scalars, all wave-uniform
template <int NPM> struct EgSlot { uint32_t Q[NPM]; uint32_t shift, bias, order, npf, precision, ci; };

// LDS of one wavefront: [image (S/2 rows of 65 words)][prefix sums | divisor table | best parameters (flacgpu_evalg.h)]
// WPC wavefronts share a channel's image and split its candidates between them (each has its own search state; the better of their
// first minima wins): an experiment in occupancy (the LDS image allows four one-wavefront channels per SIMD), opt-in, see launch_evalg.
template <int MAXORD>
__host__ __device__ inline uint32_t evalg_lds_bytes(uint32_t N, uint32_t wpc = 1) { return (N / 128) * EG_ROW + wpc * eg_tail_bytes<MAXORD>() + 64 + 128 + ((N / 64) % 8 ? 8 * EG_ROW : 0); }      // (+128: the half piece of a 16 k + 8 run loads four rows behind the image; runs of 16 k + 2 or + 4 samples: up to seven)
constexpr int EG_PIECES_AHEAD = 8;        // 16-byte pieces of the planar channel a lane has in flight before its first use (8 = a 4096-sample block at WPC 1)

// returns false when the channel is not this kernel's (the caller lists it)
template <int MAXORD, int WPC>
__device__ __forceinline__ bool evalg_body(const DevParams &P, const int32_t *__restrict__ chan, const JobTable *__restrict__ jt, ChanPrep *__restrict__ preps,
                                           const Candidate *__restrict__ cands, const int *__restrict__ valid, SubDecision *__restrict__ decisions, uint32_t fc,
                                           unsigned char *smem, int tid)
{
	const int lane = tid & 63;
	const uint32_t wave = WPC > 1 ? (uint32_t)__builtin_amdgcn_readfirstlane(tid >> 6) : 0u;
	constexpr int AHEAD = EG_PIECES_AHEAD / WPC;
	constexpr int NPM = eg_npm<MAXORD>(), NQ = 2 * NPM - 1;               // folded tap pairs; taps of a candidate record a lane holds (13 | 33)
	static_assert(MAXORD < NQ, "a record holds the predictor's taps");
	const uint32_t n = P.blocksize, S = n / 64;
	const uint32_t aslots = P.norders * P.nprec, cstride = P.ncslots;
	// ---- every load from HBM goes out before the first use: one round trip, not three ------------------------------------
	const ChanPrep pr = preps[fc];
	const uint32_t nanalyses = jt->nanalyses;
	int c_vflag = 0;
	uint32_t c_order = 0, c_shift = 0, c_prec = 0, c_wide = 0;
	int32_t cq[NQ];
#pragma unroll
	for(int j = 0; j < NQ; j++) cq[j] = 0;
	if((uint32_t)lane < cstride) {                                            // lane c holds candidate c (cstride <= EG_MAXC)
		const size_t ix = (size_t)fc * cstride + (uint32_t)lane;
		c_vflag = valid[ix];
		const Candidate *cd = cands + ix;
		c_order = cd->order; c_shift = (uint32_t)cd->shift; c_prec = cd->precision; c_wide = cd->wide;
#pragma unroll
		for(int j = 0; j < MAXORD; j++) cq[j] = cd->q[j];
	}
	const uint4 *src = (const uint4 *)(chan + (size_t)fc * P.chan_stride);
	// (runs that are no whole number of 16-byte pieces -- 18 or 36 samples: blocks of 1152 and 2304 -- are filled word by word below)
	const bool by_words = (S % 8u) != 0;
	const uint32_t nvec = by_words ? 0u : n / 8, vps = S / 8;                 // 16-byte pieces of the block, of a lane's run
	uint4 pv[AHEAD];
#pragma unroll
	for(int i = 0; i < AHEAD; i++) { const uint32_t m = (uint32_t)tid + 64u * WPC * (uint32_t)i; if(m < nvec) pv[i] = src[m]; }

	const uint32_t nan = P.nfixed + ((pr.flags & PREP_LPC) ? nanalyses * aslots : 0);
	const bool any = !(pr.flags & PREP_CONSTANT) && ((pr.flags & PREP_FIXED_VALID) || nan > P.nfixed);
	// partition order limits of the frame (stream_encoder.c:3759-3761)
	uint32_t frame_max_po = 0;
	{ uint32_t b = n; while(!(b & 1)) { frame_max_po++; b >>= 1; } if(frame_max_po > 15) frame_max_po = 15; }
	frame_max_po = umin32(frame_max_po, P.max_po);
	const uint32_t frame_min_po = umin32(P.min_po, frame_max_po);
	const uint32_t psize = n >> frame_max_po;
	const bool narrow = (pr.sbps + 4) < (32 - ilog2_u32(psize));               // stream_encoder.c:4814-4817
	const uint32_t sbps = pr.sbps, hdr = 8 + pr.wasted;
	const uint32_t tail_off = (S / 2) * EG_ROW + wave * eg_tail_bytes<MAXORD>();           // this wavefront's search state behind the shared image
	uint8_t *kbest = smem + tail_off + eg_tail_bytes<MAXORD>() - 64;
	uint32_t *merge = (uint32_t *)(smem + (S / 2) * EG_ROW + WPC * eg_tail_bytes<MAXORD>());      // [WPC][4]: best estimate, candidate, left?, -
	EgSearch R;
	R.best_est = 0xffffffffu; R.best_ci = 0xffffffffu; R.best_po = 0;

	if(any) {
	if(pr.fmt != 1 || !narrow || frame_max_po > 6) return false;

	// ---- candidate records: folded taps, and whether this kernel's arithmetic is exact for them --------------------------------
	uint32_t Qv[NPM];
	const bool c_valid = (uint32_t)lane < nan && c_vflag != 0;
	bool c_ok = true;
	{
		int32_t t[2 * NPM];
		uint32_t abs_sum = 0;
#pragma unroll
		for(int j = 0; j < NQ; j++) {
			const int32_t q = (uint32_t)j < c_order ? cq[j] : 0;
			t[j + 1] = q;
			abs_sum += (uint32_t)(q < 0 ? -q : q);
		}
		t[0] = -(int32_t)(1u << (c_shift & 15u));
#pragma unroll
		for(int p = 0; p < NPM; p++) Qv[p] = ((uint32_t)t[2 * p] << 16) | ((uint32_t)t[2 * p + 1] & 0xffffu);
		// (from 13 taps up the reference runs the 64-bit routine on 16-bit input -- bps + precision + ilog2(order) > 32, lpc.c:942-976 --:
		//  where the bound below holds, the sum it forms fits 32 bits and the chain here forms the same integer; the overflow-checked
		//  flavour, wide == 2, stays out)
		if(c_valid) c_ok = (MAXORD <= 12 ? c_wide == 0 : c_wide <= 1) && c_order <= (uint32_t)MAXORD && c_shift <= 15u
		                   && (((uint64_t)abs_sum + (1u << (c_shift & 15u))) << (sbps - 1)) < (1ull << 31);
	}
	if(__any((int)!c_ok)) return false;
	uint64_t vmask = __ballot((int)c_valid);

	// ---- LDS of this wavefront --------------------------------------------------------------------------------------------
	const uint32_t rows = S / 2;
	const uint32_t img_bytes = rows * EG_ROW;
	// the planar channel (16-bit pairs) into the image: a 16-byte piece is four consecutive words of ONE lane's run (S is a
	// multiple of 16)
	{
		const bool spow2 = (vps & (vps - 1)) == 0;
		const uint32_t vlog = ilog2_u32(vps);
		if(tid < NPM + 1) *(uint32_t *)(smem + (rows - (NPM + 1) + (uint32_t)tid) * EG_ROW) = 0;      // column 0: lane 0's history
#pragma unroll
		for(int i = 0; i < AHEAD; i++) {
			const uint32_t m = (uint32_t)tid + 64u * WPC * (uint32_t)i;
			if(m < nvec) {
				const uint32_t Lo = spow2 ? m >> vlog : m / vps, r = m - Lo * vps;          // run, piece of the run
				unsigned char *d = smem + (Lo + 1) * 4 + 4 * r * EG_ROW;
				*(uint32_t *)(d) = pv[i].x; *(uint32_t *)(d + EG_ROW) = pv[i].y; *(uint32_t *)(d + 2 * EG_ROW) = pv[i].z; *(uint32_t *)(d + 3 * EG_ROW) = pv[i].w;
			}
		}
		if(by_words) {
			// word w of the block is word w % (S/2) of run w / (S/2)  (round 6: 1152- and 2304-sample blocks at the LPC presets)
			const uint32_t wpr = S / 2, nwords = n / 2;
			const uint32_t *srcw = (const uint32_t *)src;
			for(uint32_t w = (uint32_t)tid; w < nwords; w += 64 * WPC) {
				const uint32_t Lo = w / wpr, r = w - Lo * wpr;
				*(uint32_t *)(smem + (Lo + 1) * 4 + r * EG_ROW) = srcw[w];
			}
		}
		for(uint32_t m = (uint32_t)tid + 64u * WPC * AHEAD; m < nvec; m += 64 * WPC) {          // blocks of more than 4096 samples
			const uint4 v = src[m];
			const uint32_t Lo = spow2 ? m >> vlog : m / vps, r = m - Lo * vps;
			unsigned char *d = smem + (Lo + 1) * 4 + 4 * r * EG_ROW;
			*(uint32_t *)(d) = v.x; *(uint32_t *)(d + EG_ROW) = v.y; *(uint32_t *)(d + 2 * EG_ROW) = v.z; *(uint32_t *)(d + 3 * EG_ROW) = v.w;
		}
	}
	eg_search_setup<MAXORD>(R, smem, tail_off, S, frame_max_po, frame_min_po, P.rice_limit, lane, jt);
	(void)img_bytes;
	const unsigned char *own = smem + ((uint32_t)lane + 1) * 4;                      // word 0 of this lane's run
	const unsigned char *prev_end = smem + (uint32_t)lane * 4 + rows * EG_ROW;       // one past the previous column's last word: the words in front of the run lie below it
	const uint32_t npieces = S / 16;                                                   // whole 16-sample pieces of a run; S % 16 == 8: a half piece behind them
	const uint32_t sum0 = 0x80000000u;
	if(WPC > 1) {
		__syncthreads();                                                          // the image is whole
		// the channel's candidates in halves: this wavefront's are the `mine` lowest (wave 0) or the rest (wave 1) of the valid ones
		const uint32_t nv = (uint32_t)__builtin_popcountll(vmask), first = (nv + 1) / 2;
		uint64_t m = vmask, lo = 0;
		for(uint32_t i = 0; i < first; i++) { lo |= m & (0 - m); m &= m - 1; }
		vmask = wave == 0 ? lo : m;
	}
	else __builtin_amdgcn_wave_barrier();
	bool leave = false;

	// ---- the candidates, two at a time ----------------------------------------------------------------------------------------
	while(vmask) {
		EgSlot<NPM> A, B;
		const uint32_t ci0 = (uint32_t)__builtin_ctzll(vmask);
		vmask &= vmask - 1;
		const bool two = vmask != 0;
		const uint32_t ci1 = two ? (uint32_t)__builtin_ctzll(vmask) : ci0;
		if(two) vmask &= vmask - 1;
#pragma unroll
		for(int p = 0; p < NPM; p++) { A.Q[p] = rdlane(Qv[p], ci0); B.Q[p] = rdlane(Qv[p], ci1); }
		A.shift = rdlane(c_shift, ci0); B.shift = rdlane(c_shift, ci1);
		A.order = rdlane(c_order, ci0); B.order = rdlane(c_order, ci1);
		A.precision = rdlane(c_prec, ci0); B.precision = rdlane(c_prec, ci1);
		A.bias = 0x80000000u >> A.shift; B.bias = 0x80000000u >> B.shift;
		A.npf = (A.order + 2) / 2; B.npf = (B.order + 2) / 2;
		A.ci = ci0; B.ci = ci1;

		uint32_t v0 = 0, v1 = 0;
		{
			uint32_t AA[NPM + 8], BB[NPM + 7];
			load_piece_head<NPM, 0>(own, prev_end, AA, BB);
			v0 = fir16_dispatch<0, 16, NPM>(A.npf, AA, BB, A.Q, A.shift, A.bias, A.order, lane == 0, sum0, v0);
			if(two) v1 = fir16_dispatch<0, 16, NPM>(B.npf, AA, BB, B.Q, B.shift, B.bias, B.order, lane == 0, sum0, v1);
		}
		if constexpr(NPM > 8) {
			// (17 pairs: the windows of pieces 1 and 2 still reach in front of the run, and warm-up samples 16..31 lie in piece 1; runs of
			//  at least 48 samples, evalg_applicable)
			{
				uint32_t AA[NPM + 8], BB[NPM + 7];
				load_piece_head<NPM, 1>(own, prev_end, AA, BB);
				v0 = fir16_dispatch<1, 16, NPM>(A.npf, AA, BB, A.Q, A.shift, A.bias, A.order, lane == 0, sum0, v0);
				if(two) v1 = fir16_dispatch<1, 16, NPM>(B.npf, AA, BB, B.Q, B.shift, B.bias, B.order, lane == 0, sum0, v1);
			}
			{
				uint32_t AA[NPM + 8], BB[NPM + 7];
				load_piece_head<NPM, 2>(own, prev_end, AA, BB);
				v0 = fir16_dispatch<-1, 16, NPM>(A.npf, AA, BB, A.Q, A.shift, A.bias, A.order, false, sum0, v0);
				if(two) v1 = fir16_dispatch<-1, 16, NPM>(B.npf, AA, BB, B.Q, B.shift, B.bias, B.order, false, sum0, v1);
			}
		}
#pragma unroll 1
		for(uint32_t c = NPM > 8 ? 3 : 1; c < npieces; c++) {
			uint32_t AA[NPM + 8], BB[NPM + 7];
			load_piece<NPM>(own + (int)(8 * c - NPM) * (int)EG_ROW, AA, BB);
			v0 = fir16_dispatch<-1, 16, NPM>(A.npf, AA, BB, A.Q, A.shift, A.bias, A.order, false, sum0, v0);
			if(two) v1 = fir16_dispatch<-1, 16, NPM>(B.npf, AA, BB, B.Q, B.shift, B.bias, B.order, false, sum0, v1);
		}
		if constexpr(NPM == 7) {
		if(S & 15u) {
			// the short piece that ends the run: 8 samples (4608-sample blocks), 4 (2304) or 2 (1152) -- the words it loads behind the run
			// are never used
			uint32_t AA[NPM + 8], BB[NPM + 7];
			load_piece<NPM>(own + (int)(8 * npieces - NPM) * (int)EG_ROW, AA, BB);
			if((S & 15u) == 8u) {
				v0 = fir16_dispatch<-1, 8, NPM>(A.npf, AA, BB, A.Q, A.shift, A.bias, A.order, false, sum0, v0);
				if(two) v1 = fir16_dispatch<-1, 8, NPM>(B.npf, AA, BB, B.Q, B.shift, B.bias, B.order, false, sum0, v1);
			}
			else if((S & 15u) == 4u) {
				v0 = fir16_dispatch<-1, 4, NPM>(A.npf, AA, BB, A.Q, A.shift, A.bias, A.order, false, sum0, v0);
				if(two) v1 = fir16_dispatch<-1, 4, NPM>(B.npf, AA, BB, B.Q, B.shift, B.bias, B.order, false, sum0, v1);
			}
			else {
				v0 = fir16_dispatch<-1, 2, NPM>(A.npf, AA, BB, A.Q, A.shift, A.bias, A.order, false, sum0, v0);
				if(two) v1 = fir16_dispatch<-1, 2, NPM>(B.npf, AA, BB, B.Q, B.shift, B.bias, B.order, false, sum0, v1);
			}
		}
		}
		// sums that leave the 32-bit arithmetic of the node passes: the channel is eval_list_kernel's (nothing was written yet)
		if(__any((int)((v0 | v1) >= (1u << 23)))) { if(WPC == 1) return false; leave = true; break; }
		EgCand CA, CB;
		CA.order = A.order; CA.precision = A.precision; CA.ci = A.ci; CB.order = B.order; CB.precision = B.precision; CB.ci = B.ci;
		eg_pair_search(R, smem, kbest, v0, v1, CA, CB, two, P.nfixed, hdr, sbps, lane);
	}
	if(WPC > 1) {
		// the better first minimum of the two halves (an equal estimate: the earlier candidate, stream_encoder.c:4191,4266); a half
		// that met sums beyond the node arithmetic sends the whole channel to the list
		if(lane == 0) { merge[4 * wave] = R.best_est; merge[4 * wave + 1] = R.best_ci; merge[4 * wave + 2] = leave ? 1u : 0u; }
		__syncthreads();
		if(merge[2] | merge[6]) return false;
		const uint32_t oe = merge[4 * (wave ^ 1u)], oc = merge[4 * (wave ^ 1u) + 1];
		const bool mine = R.best_est < oe || (R.best_est == oe && (R.best_ci < oc || (R.best_ci == oc && wave == 0)));
		if(!mine) return true;                                                    // (the other wavefront writes the decision)
	}
	}       // any
	else if(WPC > 1 && wave != 0) return true;                                    // a channel without candidates: wavefront 0 decides it

	eg_decide<MAXORD, NQ>(R, P, pr, n, kbest, c_order, c_prec, c_shift, cq, decisions + fc, preps + fc, lane);
	return true;
}

template <int MAXORD, int WPC>
__global__ __launch_bounds__(64 * WPC, MAXORD > 12 ? 3 : EVALG_WAVES_PER_SIMD) void evalg_kernel(      // (17 tap pairs: a window of 49 words and 50 registers of records)
                                                                          const DevParams P, const int32_t *__restrict__ chan, uint32_t nframes, uint32_t tail_n,
                                                                          const JobTable *__restrict__ jt, ChanPrep *__restrict__ preps, const Candidate *__restrict__ cands,
                                                                          const int *__restrict__ valid, SubDecision *__restrict__ decisions,
                                                                          uint32_t *__restrict__ left, uint32_t *__restrict__ nleft)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int tid = (int)threadIdx.x;
	const uint32_t fc = blockIdx.x;
	// what this kernel does not take -- the short last block first of all -- goes on the next kernel's list
	const bool tail = tail_n != 0 && fc / P.ncand == nframes - 1;
	if(tail || !evalg_body<MAXORD, WPC>(P, chan, jt, preps, cands, valid, decisions, fc, smem, tid)) { if(tid == 0) left[atomicAdd(nleft, 1u)] = fc; }
}

// ---------------------------------------------------------------------------------------------------------
// launch
// ---------------------------------------------------------------------------------------------------------
bool evalg_applicable(const DevParams &P)
{
	const int off = tune().no_evalg;
	const uint32_t S = P.blocksize / 64;
	const uint32_t tail = S % 16;                                           // a run's last piece: whole, or 8, 4 or 2 samples
	if(off || P.blocksize % 64 != 0 || P.wide_samples || P.stream_sig || P.ncslots > (uint32_t)EG_MAXC || P.blocksize > 16384) return false;
	// predictors of 13..32 taps (round 6): runs of whole pieces, three of them at least (the windows of the first three reach in front of the run)
	if(P.max_lpc_order > 12) return P.max_lpc_order <= 32 && S >= 48 && tail == 0 && !tune().no_evalg32;
	return S >= 16 && (tail == 0 || tail == 8 || tail == 4 || tail == 2);
}
template <int MAXORD, int WPC>
static hipError_t launch_evalg_t(const DevParams &P, uint32_t nframes, uint32_t tail_n, const JobTable *jt, const AnalyzeBuffers &B, SubDecision *dec, hipStream_t s)
{
	const uint32_t lds = evalg_lds_bytes<MAXORD>(P.blocksize, WPC);
	static AttrFlags set;
	if(AttrOnce once{set}) { const hipError_t e = hipFuncSetAttribute((const void *)evalg_kernel<MAXORD, WPC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024); if(e != hipSuccess) return e;
		once.ok(); }
	note_launch(K_EVALG);
	hipLaunchKernelGGL((evalg_kernel<MAXORD, WPC>), dim3(nframes * P.ncand), dim3(64 * WPC), lds, s, P, B.chan, nframes, tail_n, jt, B.prep, B.cands, B.valid, dec, B.left, B.nleft);
	return hipGetLastError();
}
hipError_t launch_evalg(const DevParams &P, uint32_t nframes, uint32_t tail_n, const JobTable *jt, const AnalyzeBuffers &B, SubDecision *dec, hipStream_t s)
{
	if(nframes == 0) return hipSuccess;
	// FLACGPU_EVAL_WPC=2: two wavefronts share a channel's image and halve its candidates (5 wavefronts per SIMD instead of 4).
	// Measured and left off: same bytes, 2 % slower at -8 (profiles/r03_s_wpc_ab.txt: 0.923-0.929 ms against 0.882-0.910) -- the
	// kernel's idle quarter is not a lack of wavefronts
	const int wpc = tune().eval_wpc;
	const bool two = wpc == 2 && P.ncslots >= 4;
	if(P.max_lpc_order > 12) return launch_evalg_t<32, 1>(P, nframes, tail_n, jt, B, dec, s);
	if(P.max_lpc_order <= 8) return two ? launch_evalg_t<8, 2>(P, nframes, tail_n, jt, B, dec, s) : launch_evalg_t<8, 1>(P, nframes, tail_n, jt, B, dec, s);
	return two ? launch_evalg_t<12, 2>(P, nframes, tail_n, jt, B, dec, s) : launch_evalg_t<12, 1>(P, nframes, tail_n, jt, B, dec, s);
}

} // namespace flacgpu
