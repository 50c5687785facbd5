// flac_amd/csrc/flacgpu_evalw.hip -- the wavefront-per-channel evaluation (see flacgpu_evalg.hip) for channels whose planar
// copy holds 32-BIT samples: the 17..25-bit subframes of 24-bit streams (BASELINE config 4: every channel), and the side channel
// of a loud 16-bit frame (white noise: one channel in four).  Same shape -- lane L owns samples [L*S, (L+1)*S) in a transposed
// LDS image, the candidates of the channel two at a time with their taps in SGPRs, the packed Rice node search of
// flacgpu_evalg.h -- with the two FIR flavours such channels need (lpc.c:942-976 picks per candidate):
//   * 32-bit sum (lpc.c:321; Candidate::wide == 0, samples of at most 24 bits): one v_mad_i32_i24 per tap, the sample itself as one
//     more tap of -2^shift (exact under the same coefficient bound as in flacgpu_evalg.hip), one logical shift, one v_sad_u32
//     against 2^(31-shift);
//   * 64-bit sum (lpc.c:582; wide == 1 -- nearly every candidate of a 24-bit stream, whose 15-bit coefficients times 24-bit
//     samples leave 32 bits): one v_mad_i64_i32 per tap, one v_alignbit for the low word of (sum >> shift), and |x - p| as one
//     v_sad_u32 on the sign-flipped operands (x ^ 2^31 is formed once per sample for the pair).
// The lanes' |residual| sums are carried in 64 bits across the 16-sample pieces (such subframes are beyond the reference's 32-bit
// partition sums, stream_encoder.c:4814): a piece's 32-bit partial sum cannot wrap because the kernel only takes candidates whose
// residual bound (lpc.c:962-967) is below 2^28; sums of 2^23 and more leave for eval_list_kernel as everywhere.
// Same integers as eval_kernel<.,0>'s fir_abs_i32 flavours; what this kernel does not take (overflow-checked candidates,
// 25-bit samples under a 32-bit sum) goes on the list.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "flacgpu_evalg.h"

namespace flacgpu {

#ifndef EVALW_WAVES_PER_SIMD
#define EVALW_WAVES_PER_SIMD 3
#endif

// x[k] = sample (piece start - 12 + k), k = 0..27; taps q[0..NT) in SGPRs (zero beyond the order)
#define EW_M24 "v_mad_i32_i24 %0, "
template <int NT>
__device__ __forceinline__ uint32_t mad24_chain_s(const int32_t *xr /* xr[-1 - j] is the sample tap j reads, xr[0] the sample itself */, const uint32_t (&q)[12], uint32_t negpow,
                                                  uint32_t sum0, uint32_t shift)
{
	uint32_t d;
	if constexpr(NT == 4)
		asm("v_mad_i32_i24 %0, %2, %3, %1\n\tv_mad_i32_i24 %0, %4, %5, %0\n\tv_mad_i32_i24 %0, %6, %7, %0\n\tv_mad_i32_i24 %0, %8, %9, %0\n\t"
		    "v_mad_i32_i24 %0, %10, %11, %0\n\tv_lshrrev_b32 %0, %12, %0"
		    : "=&v"(d) : "v"(sum0), "v"(xr[-1]), "s"(q[0]), "v"(xr[-2]), "s"(q[1]), "v"(xr[-3]), "s"(q[2]), "v"(xr[-4]), "s"(q[3]), "v"(xr[0]), "s"(negpow), "s"(shift));
	if constexpr(NT == 8)
		asm("v_mad_i32_i24 %0, %2, %3, %1\n\tv_mad_i32_i24 %0, %4, %5, %0\n\tv_mad_i32_i24 %0, %6, %7, %0\n\tv_mad_i32_i24 %0, %8, %9, %0\n\t"
		    "v_mad_i32_i24 %0, %10, %11, %0\n\tv_mad_i32_i24 %0, %12, %13, %0\n\tv_mad_i32_i24 %0, %14, %15, %0\n\tv_mad_i32_i24 %0, %16, %17, %0\n\t"
		    "v_mad_i32_i24 %0, %18, %19, %0\n\tv_lshrrev_b32 %0, %20, %0"
		    : "=&v"(d) : "v"(sum0), "v"(xr[-1]), "s"(q[0]), "v"(xr[-2]), "s"(q[1]), "v"(xr[-3]), "s"(q[2]), "v"(xr[-4]), "s"(q[3]), "v"(xr[-5]), "s"(q[4]), "v"(xr[-6]), "s"(q[5]),
		      "v"(xr[-7]), "s"(q[6]), "v"(xr[-8]), "s"(q[7]), "v"(xr[0]), "s"(negpow), "s"(shift));
	if constexpr(NT == 10)
		asm("v_mad_i32_i24 %0, %2, %3, %1\n\tv_mad_i32_i24 %0, %4, %5, %0\n\tv_mad_i32_i24 %0, %6, %7, %0\n\tv_mad_i32_i24 %0, %8, %9, %0\n\t"
		    "v_mad_i32_i24 %0, %10, %11, %0\n\tv_mad_i32_i24 %0, %12, %13, %0\n\tv_mad_i32_i24 %0, %14, %15, %0\n\tv_mad_i32_i24 %0, %16, %17, %0\n\t"
		    "v_mad_i32_i24 %0, %18, %19, %0\n\tv_mad_i32_i24 %0, %20, %21, %0\n\tv_mad_i32_i24 %0, %22, %23, %0\n\tv_lshrrev_b32 %0, %24, %0"
		    : "=&v"(d) : "v"(sum0), "v"(xr[-1]), "s"(q[0]), "v"(xr[-2]), "s"(q[1]), "v"(xr[-3]), "s"(q[2]), "v"(xr[-4]), "s"(q[3]), "v"(xr[-5]), "s"(q[4]), "v"(xr[-6]), "s"(q[5]),
		      "v"(xr[-7]), "s"(q[6]), "v"(xr[-8]), "s"(q[7]), "v"(xr[-9]), "s"(q[8]), "v"(xr[-10]), "s"(q[9]), "v"(xr[0]), "s"(negpow), "s"(shift));
	if constexpr(NT == 12) {
		uint32_t t;
		asm("v_mad_i32_i24 %0, %2, %3, %1\n\tv_mad_i32_i24 %0, %4, %5, %0\n\tv_mad_i32_i24 %0, %6, %7, %0\n\tv_mad_i32_i24 %0, %8, %9, %0\n\t"
		    "v_mad_i32_i24 %0, %10, %11, %0\n\tv_mad_i32_i24 %0, %12, %13, %0\n\tv_mad_i32_i24 %0, %14, %15, %0\n\tv_mad_i32_i24 %0, %16, %17, %0"
		    : "=&v"(t) : "v"(sum0), "v"(xr[-1]), "s"(q[0]), "v"(xr[-2]), "s"(q[1]), "v"(xr[-3]), "s"(q[2]), "v"(xr[-4]), "s"(q[3]), "v"(xr[-5]), "s"(q[4]), "v"(xr[-6]), "s"(q[5]),
		      "v"(xr[-7]), "s"(q[6]), "v"(xr[-8]), "s"(q[7]));
		asm("v_mad_i32_i24 %0, %2, %3, %1\n\tv_mad_i32_i24 %0, %4, %5, %0\n\tv_mad_i32_i24 %0, %6, %7, %0\n\tv_mad_i32_i24 %0, %8, %9, %0\n\t"
		    "v_mad_i32_i24 %0, %10, %11, %0\n\tv_lshrrev_b32 %0, %12, %0"
		    : "=&v"(d) : "v"(t), "v"(xr[-9]), "s"(q[8]), "v"(xr[-10]), "s"(q[9]), "v"(xr[-11]), "s"(q[10]), "v"(xr[-12]), "s"(q[11]), "v"(xr[0]), "s"(negpow), "s"(shift));
	}
	return d;
}
template <int NT>
__device__ __forceinline__ uint64_t mad64_chain_s(const int32_t *xr, const uint32_t (&q)[12], uint64_t init)
{
	uint64_t d;
	if constexpr(NT == 4)
		asm("v_mad_i64_i32 %0, vcc, %1, %2, %9\n\tv_mad_i64_i32 %0, vcc, %3, %4, %0\n\tv_mad_i64_i32 %0, vcc, %5, %6, %0\n\tv_mad_i64_i32 %0, vcc, %7, %8, %0"
		    : "=&v"(d) : "v"(xr[-1]), "s"(q[0]), "v"(xr[-2]), "s"(q[1]), "v"(xr[-3]), "s"(q[2]), "v"(xr[-4]), "s"(q[3]), "v"(init) : "vcc");
	if constexpr(NT == 8)
		asm("v_mad_i64_i32 %0, vcc, %1, %2, %17\n\tv_mad_i64_i32 %0, vcc, %3, %4, %0\n\tv_mad_i64_i32 %0, vcc, %5, %6, %0\n\tv_mad_i64_i32 %0, vcc, %7, %8, %0\n\t"
		    "v_mad_i64_i32 %0, vcc, %9, %10, %0\n\tv_mad_i64_i32 %0, vcc, %11, %12, %0\n\tv_mad_i64_i32 %0, vcc, %13, %14, %0\n\tv_mad_i64_i32 %0, vcc, %15, %16, %0"
		    : "=&v"(d) : "v"(xr[-1]), "s"(q[0]), "v"(xr[-2]), "s"(q[1]), "v"(xr[-3]), "s"(q[2]), "v"(xr[-4]), "s"(q[3]), "v"(xr[-5]), "s"(q[4]), "v"(xr[-6]), "s"(q[5]),
		      "v"(xr[-7]), "s"(q[6]), "v"(xr[-8]), "s"(q[7]), "v"(init) : "vcc");
	if constexpr(NT == 10)
		asm("v_mad_i64_i32 %0, vcc, %1, %2, %21\n\tv_mad_i64_i32 %0, vcc, %3, %4, %0\n\tv_mad_i64_i32 %0, vcc, %5, %6, %0\n\tv_mad_i64_i32 %0, vcc, %7, %8, %0\n\t"
		    "v_mad_i64_i32 %0, vcc, %9, %10, %0\n\tv_mad_i64_i32 %0, vcc, %11, %12, %0\n\tv_mad_i64_i32 %0, vcc, %13, %14, %0\n\tv_mad_i64_i32 %0, vcc, %15, %16, %0\n\t"
		    "v_mad_i64_i32 %0, vcc, %17, %18, %0\n\tv_mad_i64_i32 %0, vcc, %19, %20, %0"
		    : "=&v"(d) : "v"(xr[-1]), "s"(q[0]), "v"(xr[-2]), "s"(q[1]), "v"(xr[-3]), "s"(q[2]), "v"(xr[-4]), "s"(q[3]), "v"(xr[-5]), "s"(q[4]), "v"(xr[-6]), "s"(q[5]),
		      "v"(xr[-7]), "s"(q[6]), "v"(xr[-8]), "s"(q[7]), "v"(xr[-9]), "s"(q[8]), "v"(xr[-10]), "s"(q[9]), "v"(init) : "vcc");
	if constexpr(NT == 12)
		asm("v_mad_i64_i32 %0, vcc, %1, %2, %25\n\tv_mad_i64_i32 %0, vcc, %3, %4, %0\n\tv_mad_i64_i32 %0, vcc, %5, %6, %0\n\tv_mad_i64_i32 %0, vcc, %7, %8, %0\n\t"
		    "v_mad_i64_i32 %0, vcc, %9, %10, %0\n\tv_mad_i64_i32 %0, vcc, %11, %12, %0\n\tv_mad_i64_i32 %0, vcc, %13, %14, %0\n\tv_mad_i64_i32 %0, vcc, %15, %16, %0\n\t"
		    "v_mad_i64_i32 %0, vcc, %17, %18, %0\n\tv_mad_i64_i32 %0, vcc, %19, %20, %0\n\tv_mad_i64_i32 %0, vcc, %21, %22, %0\n\tv_mad_i64_i32 %0, vcc, %23, %24, %0"
		    : "=&v"(d) : "v"(xr[-1]), "s"(q[0]), "v"(xr[-2]), "s"(q[1]), "v"(xr[-3]), "s"(q[2]), "v"(xr[-4]), "s"(q[3]), "v"(xr[-5]), "s"(q[4]), "v"(xr[-6]), "s"(q[5]),
		      "v"(xr[-7]), "s"(q[6]), "v"(xr[-8]), "s"(q[7]), "v"(xr[-9]), "s"(q[8]), "v"(xr[-10]), "s"(q[9]), "v"(xr[-11]), "s"(q[10]), "v"(xr[-12]), "s"(q[11]), "v"(init) : "vcc");
	return d;
}
__device__ __forceinline__ uint32_t sad_u32_vs(uint32_t a, uint32_t b_uniform, uint32_t c)
{
	uint32_t d;
	asm("v_sad_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(b_uniform), "v"(c));
	return d;
}

// one slot of the pair (wide: 0 = the 24-bit multiplier chain with the folded sample tap, 1 = the 64-bit chain)
struct EwSlot { uint32_t q[12]; uint32_t shift, bias, negpow, order, nt, wide, precision, ci; };

// the |residual| sum of one candidate over one 16-sample piece of every lane's run
// NV: samples of the piece that belong to the lane's run (16; 8 for the half piece that ends a run of 16 k + 8 samples)
template <int NT, bool WIDE, bool FIRST, int NV = 16>
__device__ __forceinline__ uint32_t fir16_w(const int32_t (&x)[28], const uint32_t (&xm)[16], const EwSlot &C, bool lane0, uint32_t sum0)
{
	uint32_t acc = 0;
	const uint64_t init64 = (uint64_t)0x80000000u << C.shift;
#pragma unroll
	for(int s = 0; s < NV; s++) {
		if constexpr(!WIDE) {
			uint32_t pb = mad24_chain_s<NT>(&x[12 + s], C.q, C.negpow, sum0, C.shift);
			if(FIRST && s < NT) { if(lane0 && (uint32_t)s < C.order) pb = C.bias; }
			acc = sad_u32_vs(pb, C.bias, acc);
		}
		else {
			// the chain starts at 2^31 << shift: the low word of (sum >> shift) then comes out with its sign bit flipped, ready for the
			// unsigned |x - p| (adding 2^31 modulo 2^32 is that flip)
			const uint64_t sum = mad64_chain_s<NT>(&x[12 + s], C.q, init64);
			uint32_t pm = __builtin_amdgcn_alignbit((uint32_t)(sum >> 32), (uint32_t)sum, C.shift);
			if(FIRST && s < NT) { if(lane0 && (uint32_t)s < C.order) pm = xm[s]; }
			acc = sad_u32(xm[s], pm, acc);
		}
	}
	return acc;
}
template <bool FIRST, int NV = 16>
__device__ __forceinline__ uint32_t fir16_w_dispatch(const int32_t (&x)[28], const uint32_t (&xm)[16], const EwSlot &C, bool lane0, uint32_t sum0)
{
	if(C.wide) {
		if(C.nt == 4) return fir16_w<4, true, FIRST, NV>(x, xm, C, lane0, sum0);
		if(C.nt == 8) return fir16_w<8, true, FIRST, NV>(x, xm, C, lane0, sum0);
		if(C.nt == 10) return fir16_w<10, true, FIRST, NV>(x, xm, C, lane0, sum0);
		return fir16_w<12, true, FIRST, NV>(x, xm, C, lane0, sum0);
	}
	if(C.nt == 4) return fir16_w<4, false, FIRST, NV>(x, xm, C, lane0, sum0);
	if(C.nt == 8) return fir16_w<8, false, FIRST, NV>(x, xm, C, lane0, sum0);
	if(C.nt == 10) return fir16_w<10, false, FIRST, NV>(x, xm, C, lane0, sum0);      // (96 kHz / 24-bit music at -8: three in five winners have order 10)
	return fir16_w<12, false, FIRST, NV>(x, xm, C, lane0, sum0);
}

// LDS of a channel: [image (S rows of 65 words)][per wavefront: prefix sums | divisor table | best parameters (flacgpu_evalg.h)][merge]
// WPC = 2: two wavefronts share a channel's image and halve its candidates between them (each with its own search state; the better
// of their first minima wins).  A 4096-sample image of 32-bit samples is 16.6 KB: one wavefront per image is 2.25 wavefronts per
// SIMD, and the kernel -- a dependent chain of multiply-adds per sample, like flacgpu_evalg.hip's -- then issues at 0.64 of the
// chip's rate (profiles/archive/r04_a_pmc_counters_hires_before.txt); two per image are four per SIMD.  (The 16-bit kernel has the same
// option and does not need it: its images are half the size.)
template <int MAXORD>
__host__ __device__ inline uint32_t evalw_lds_bytes(uint32_t N, uint32_t wpc = 1) { return (N / 64) * EG_ROW + wpc * eg_tail_bytes<MAXORD>() + 64; }
constexpr int EW_PIECES_AHEAD = 8;

// returns false when the channel is not this kernel's (the caller lists it)
template <int MAXORD, int WPC>
__device__ __forceinline__ bool evalw_body(const DevParams &P, const int32_t *__restrict__ chan, const JobTable *__restrict__ jt, ChanPrep *__restrict__ preps,
                                           const Candidate *__restrict__ cands, const int *__restrict__ valid, SubDecision *__restrict__ decisions, uint32_t fc,
                                           unsigned char *smem, int tid)
{
	const int lane = tid & 63;
	const uint32_t wave = WPC > 1 ? (uint32_t)__builtin_amdgcn_readfirstlane(tid >> 6) : 0u;
	const uint32_t n = P.blocksize, S = n / 64;
	const uint32_t aslots = P.norders * P.nprec, cstride = P.ncslots;
	// ---- every load from HBM goes out before the first use ------------------------------------------------------------------
	const ChanPrep pr = preps[fc];
	const uint32_t nanalyses = jt->nanalyses;
	int c_vflag = 0;
	uint32_t c_order = 0, c_shift = 0, c_prec = 0, c_wide = 0;
	int32_t cq[13];
#pragma unroll
	for(int j = 0; j < 13; j++) cq[j] = 0;
	if((uint32_t)lane < cstride) {                                            // lane c holds candidate c (cstride <= EG_MAXC)
		const size_t ix = (size_t)fc * cstride + (uint32_t)lane;
		c_vflag = valid[ix];
		const Candidate *cd = cands + ix;
		c_order = cd->order; c_shift = (uint32_t)cd->shift; c_prec = cd->precision; c_wide = cd->wide;
#pragma unroll
		for(int j = 0; j < MAXORD; j++) cq[j] = cd->q[j];
	}
	const uint4 *src = (const uint4 *)(chan + (size_t)fc * P.chan_stride);
	const uint32_t nvec = n / 4, vps = S / 4;                                 // 16-byte pieces (4 samples) of the block, of a lane's run
	uint4 pv[EW_PIECES_AHEAD];
#pragma unroll
	for(int i = 0; i < EW_PIECES_AHEAD; i++) { const uint32_t m = (uint32_t)tid + 64u * WPC * (uint32_t)i; if(m < nvec) pv[i] = src[m]; }

	const uint32_t nan = P.nfixed + ((pr.flags & PREP_LPC) ? nanalyses * aslots : 0);
	const bool any = !(pr.flags & PREP_CONSTANT) && ((pr.flags & PREP_FIXED_VALID) || nan > P.nfixed);
	uint32_t frame_max_po = 0;                                                // stream_encoder.c:3759-3761
	{ uint32_t b = n; while(!(b & 1)) { frame_max_po++; b >>= 1; } if(frame_max_po > 15) frame_max_po = 15; }
	frame_max_po = umin32(frame_max_po, P.max_po);
	const uint32_t frame_min_po = umin32(P.min_po, frame_max_po);
	const uint32_t sbps = pr.sbps, hdr = 8 + pr.wasted;
	const uint32_t tail_off = S * EG_ROW + wave * eg_tail_bytes<MAXORD>();      // this wavefront's search state behind the shared image
	uint8_t *kbest = smem + tail_off + eg_tail_bytes<MAXORD>() - 64;
	uint32_t *merge = (uint32_t *)(smem + S * EG_ROW + WPC * eg_tail_bytes<MAXORD>());     // [WPC][4]: best estimate, candidate, left?, -
	EgSearch R;
	R.best_est = 0xffffffffu; R.best_ci = 0xffffffffu; R.best_po = 0;

	if(any) {
	// (round 6: a plane of 16-bit PAIRS is taken too -- ChanPrep::fmt = 1: 16-bit audio in a 24-bit container (eight wasted bits in every
	//  subframe), a quiet side channel, or what evalg_kernel listed because one of its candidates wants the 64-bit sum.  These went to the
	//  general body: 16-bit audio in 24 bits at -8 evaluated in 2.73 ms per 16384 frames against 0.95 as a 16-bit stream.  The image is
	//  the same -- a sample per word --, filled from 8-byte pieces of four samples; the arithmetic is exact for any width)
	const bool pairs = pr.fmt == 1;
	if(pr.fmt > 1 || frame_max_po > 6 || sbps > 32) return false;

	// ---- candidate records: which flavour, and whether this kernel's arithmetic is exact for them ------------------------------
	const bool c_valid = (uint32_t)lane < nan && c_vflag != 0;
	bool c_ok = true;
	if(c_valid) {
		uint32_t abs_sum = 0;
#pragma unroll
		for(int j = 0; j < 13; j++) { const int32_t q = (uint32_t)j < c_order ? cq[j] : 0; abs_sum += (uint32_t)(q < 0 ? -q : q); }
		const uint32_t sh = c_shift & 15u;
		const uint64_t maxabs = (uint64_t)1 << (sbps - 1), before = maxabs * abs_sum;
		const uint64_t rmax = maxabs + ((before + ((uint64_t)1 << sh) - 1) >> sh);                      // lpc.c:962-967
		c_ok = c_order <= (uint32_t)MAXORD && c_shift <= 15u && rmax < ((uint64_t)1 << 28) && c_wide <= 1;       // (2: the overflow-checked flavour, lpc.c:832)
		// the 32-bit sum of lpc.c:321 is the true sum for a wide == 0 candidate (that is what wide == 0 says), so the 64-bit chain
		// gives the same integers: it takes the candidates the 24-bit multiplier cannot (25-bit samples) or whose folded sample tap
		// could leave 32 bits
		if(c_wide == 0 && !(sbps <= 24 && (((uint64_t)abs_sum + (1u << sh)) << (sbps - 1)) < (1ull << 31))) c_wide = 1;
	}
#pragma unroll
	for(int j = 0; j < 13; j++) if((uint32_t)j >= c_order) cq[j] = 0;
	if(__any((int)!c_ok)) return false;
	uint64_t vmask = __ballot((int)c_valid);

	// ---- LDS of this wavefront: word j of lane L's run (one sample) at (j * 65 + L + 1); column 0 = lane 0's history: zero ----------
	const uint32_t rows = S;
	{
		const bool spow2 = (vps & (vps - 1)) == 0;
		const uint32_t vlog = ilog2_u32(vps);
		if(tid < 16) *(uint32_t *)(smem + (rows - 16 + (uint32_t)tid) * EG_ROW) = 0;
		if(pairs) {
			const uint2 *src2 = (const uint2 *)src;
			for(uint32_t m = (uint32_t)tid; m < nvec; m += 64 * WPC) {
				const uint2 v = src2[m];
				const uint32_t Lo = spow2 ? m >> vlog : m / vps, r = m - Lo * vps;
				unsigned char *d = smem + (Lo + 1) * 4 + 4 * r * EG_ROW;
				*(int32_t *)(d) = (int32_t)(int16_t)(v.x & 0xffffu); *(int32_t *)(d + EG_ROW) = (int32_t)v.x >> 16;
				*(int32_t *)(d + 2 * EG_ROW) = (int32_t)(int16_t)(v.y & 0xffffu); *(int32_t *)(d + 3 * EG_ROW) = (int32_t)v.y >> 16;
			}
		}
		else {
#pragma unroll
		for(int i = 0; i < EW_PIECES_AHEAD; i++) {
			const uint32_t m = (uint32_t)tid + 64u * WPC * (uint32_t)i;
			if(m < nvec) {
				const uint32_t Lo = spow2 ? m >> vlog : m / vps, r = m - Lo * vps;
				unsigned char *d = smem + (Lo + 1) * 4 + 4 * r * EG_ROW;
				*(uint32_t *)(d) = pv[i].x; *(uint32_t *)(d + EG_ROW) = pv[i].y; *(uint32_t *)(d + 2 * EG_ROW) = pv[i].z; *(uint32_t *)(d + 3 * EG_ROW) = pv[i].w;
			}
		}
		for(uint32_t m = (uint32_t)tid + 64u * WPC * EW_PIECES_AHEAD; m < nvec; m += 64 * WPC) {
			const uint4 v = src[m];
			const uint32_t Lo = spow2 ? m >> vlog : m / vps, r = m - Lo * vps;
			unsigned char *d = smem + (Lo + 1) * 4 + 4 * r * EG_ROW;
			*(uint32_t *)(d) = v.x; *(uint32_t *)(d + EG_ROW) = v.y; *(uint32_t *)(d + 2 * EG_ROW) = v.z; *(uint32_t *)(d + 3 * EG_ROW) = v.w;
		}
		}
	}
	eg_search_setup<MAXORD>(R, smem, tail_off, S, frame_max_po, frame_min_po, P.rice_limit, lane, jt);
	const unsigned char *own = smem + ((uint32_t)lane + 1) * 4;                      // sample 0 of this lane's run
	const unsigned char *hist = smem + (uint32_t)lane * 4 + (rows - 12) * EG_ROW;    // the 12 samples in front of it: the previous column's last
	const uint32_t npieces = S / 16;
	const uint32_t sum0 = 0x80000000u;
	if(WPC > 1) {
		__syncthreads();                                                          // the image is whole
		// the channel's candidates in halves: this wavefront's are the lowest (wave 0) or the rest (wave 1) of the valid ones
		const uint32_t nv = (uint32_t)__builtin_popcountll(vmask), first = (nv + 1) / 2;
		uint64_t m = vmask, lo = 0;
		for(uint32_t i = 0; i < first; i++) { lo |= m & (0 - m); m &= m - 1; }
		vmask = wave == 0 ? lo : m;
	}
	else __builtin_amdgcn_wave_barrier();
	bool leave = false;

	// ---- the candidates, two at a time ----------------------------------------------------------------------------------------
	while(vmask) {
		EwSlot A, B;
		const uint32_t ci0 = (uint32_t)__builtin_ctzll(vmask);
		vmask &= vmask - 1;
		const bool two = vmask != 0;
		const uint32_t ci1 = two ? (uint32_t)__builtin_ctzll(vmask) : ci0;
		if(two) vmask &= vmask - 1;
#pragma unroll
		for(int j = 0; j < 12; j++) { A.q[j] = rdlane((uint32_t)cq[j], ci0); B.q[j] = rdlane((uint32_t)cq[j], ci1); }
		A.shift = rdlane(c_shift, ci0); B.shift = rdlane(c_shift, ci1);
		A.order = rdlane(c_order, ci0); B.order = rdlane(c_order, ci1);
		A.precision = rdlane(c_prec, ci0); B.precision = rdlane(c_prec, ci1);
		A.wide = rdlane(c_wide, ci0); B.wide = rdlane(c_wide, ci1);
		A.bias = 0x80000000u >> A.shift; B.bias = 0x80000000u >> B.shift;
		A.negpow = 0u - (1u << A.shift); B.negpow = 0u - (1u << B.shift);
		A.nt = A.order <= 4 ? 4u : (MAXORD <= 8 || A.order <= 8) ? 8u : A.order <= 10 ? 10u : 12u; B.nt = B.order <= 4 ? 4u : (MAXORD <= 8 || B.order <= 8) ? 8u : B.order <= 10 ? 10u : 12u;
		A.ci = ci0; B.ci = ci1;

		uint64_t s0 = 0, s1 = 0;
		{
			int32_t x[28];
			uint32_t xm[16];
#pragma unroll
			for(int k = 0; k < 12; k++) x[k] = *(const int32_t *)(hist + k * EG_ROW);
#pragma unroll
			for(int k = 0; k < 16; k++) { x[12 + k] = *(const int32_t *)(own + k * EG_ROW); xm[k] = (uint32_t)x[12 + k] ^ 0x80000000u; }
			s0 += fir16_w_dispatch<true>(x, xm, A, lane == 0, sum0);
			if(two) s1 += fir16_w_dispatch<true>(x, xm, B, lane == 0, sum0);
		}
#pragma unroll 1
		for(uint32_t c = 1; c < npieces; c++) {
			int32_t x[28];
			uint32_t xm[16];
			const unsigned char *b = own + (16 * c - 12) * EG_ROW;
#pragma unroll
			for(int k = 0; k < 28; k++) x[k] = *(const int32_t *)(b + k * EG_ROW);
#pragma unroll
			for(int k = 0; k < 16; k++) xm[k] = (uint32_t)x[12 + k] ^ 0x80000000u;
			s0 += fir16_w_dispatch<false>(x, xm, A, false, sum0);
			if(two) s1 += fir16_w_dispatch<false>(x, xm, B, false, sum0);
		}
		if(S & 8u) {
			// the half piece that ends a run of 16 k + 8 samples (blocks of 4608): its eight samples and the twelve in front of them
			int32_t x[28];
			uint32_t xm[16];
			const unsigned char *b = own + (16 * npieces - 12) * EG_ROW;
#pragma unroll
			for(int k = 0; k < 28; k++) x[k] = k < 20 ? *(const int32_t *)(b + k * EG_ROW) : 0;
#pragma unroll
			for(int k = 0; k < 16; k++) xm[k] = (uint32_t)x[12 + k] ^ 0x80000000u;
			s0 += fir16_w_dispatch<false, 8>(x, xm, A, false, sum0);
			if(two) s1 += fir16_w_dispatch<false, 8>(x, xm, B, false, sum0);
		}
		// sums that leave the 32-bit arithmetic of the node passes: the channel is eval_list_kernel's (nothing was written yet)
		if(__any((int)((s0 | s1) >= (1ull << 23)))) { if(WPC == 1) return false; leave = true; break; }
		EgCand CA, CB;
		CA.order = A.order; CA.precision = A.precision; CA.ci = A.ci; CB.order = B.order; CB.precision = B.precision; CB.ci = B.ci;
		eg_pair_search(R, smem, kbest, (uint32_t)s0, (uint32_t)s1, CA, CB, two, P.nfixed, hdr, sbps, lane);
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

	eg_decide<MAXORD>(R, P, pr, n, kbest, c_order, c_prec, c_shift, cq, decisions + fc, preps + fc, lane);
	return true;
}

// LIST: the channels come from a list (what flacgpu_evalg.hip's kernel left: the 32-bit channels of a 16-bit stream), a fixed grid
// of wavefronts looping over it; otherwise one wavefront per channel of the batch (24-bit streams: every channel is one of these)
template <int MAXORD, bool LIST, int WPC>
__global__ __launch_bounds__(64 * WPC, WPC > 1 ? 4 : EVALW_WAVES_PER_SIMD) void evalw_kernel(const DevParams P, const int32_t *__restrict__ chan, uint32_t nframes, uint32_t tail_n,
                                                                          const JobTable *__restrict__ jt, ChanPrep *__restrict__ preps, const Candidate *__restrict__ cands,
                                                                          const int *__restrict__ valid, SubDecision *__restrict__ decisions,
                                                                          const uint32_t *__restrict__ in_list, const uint32_t *__restrict__ in_count,
                                                                          uint32_t *__restrict__ left, uint32_t *__restrict__ nleft)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int tid = (int)threadIdx.x;
	static_assert(!(LIST && WPC > 1), "the list flavour runs one wavefront per channel");
	if(LIST) {
		const uint32_t count = *in_count;
		for(uint32_t e = blockIdx.x; e < count; e += gridDim.x) {
			const uint32_t fc = in_list[e];
			const bool tail = tail_n != 0 && fc / P.ncand == nframes - 1;
			if(tail || !evalw_body<MAXORD, WPC>(P, chan, jt, preps, cands, valid, decisions, fc, smem, tid)) { if(tid == 0) left[atomicAdd(nleft, 1u)] = fc; }
			__builtin_amdgcn_wave_barrier();
		}
	}
	else {
		const uint32_t fc = blockIdx.x;
		const bool tail = tail_n != 0 && fc / P.ncand == nframes - 1;
		if(tail || !evalw_body<MAXORD, WPC>(P, chan, jt, preps, cands, valid, decisions, fc, smem, tid)) { if(tid == 0) left[atomicAdd(nleft, 1u)] = fc; }
	}
}

// ---------------------------------------------------------------------------------------------------------
// launch
// ---------------------------------------------------------------------------------------------------------
template <int MAXORD>
static hipError_t launch_evalw_t(const DevParams &P, uint32_t nframes, uint32_t tail_n, const JobTable *jt, const AnalyzeBuffers &B, SubDecision *dec,
                                 const uint32_t *in_list, const uint32_t *in_count, uint32_t *out_list, uint32_t *out_count, hipStream_t s)
{
	static AttrFlags set;
	if(AttrOnce once{set}) {
		hipError_t e = hipFuncSetAttribute((const void *)evalw_kernel<MAXORD, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
		if(e == hipSuccess) e = hipFuncSetAttribute((const void *)evalw_kernel<MAXORD, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
		if(e == hipSuccess) e = hipFuncSetAttribute((const void *)evalw_kernel<MAXORD, true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
		if(e != hipSuccess) return e;
		once.ok();
	}
	note_launch(K_EVALW);
	const uint32_t nchan = nframes * P.ncand;
	if(in_list) {
		// as many wavefronts as the chip holds of them: an empty list costs a few microseconds, a full one (white noise: the side
		// channel of every frame) keeps every SIMD busy
		const uint32_t lds = evalw_lds_bytes<MAXORD>(P.blocksize);
		uint32_t per_cu = (160u * 1024u) / lds;
		if(per_cu > 4u * EVALW_WAVES_PER_SIMD) per_cu = 4u * EVALW_WAVES_PER_SIMD;
		if(per_cu < 1) per_cu = 1;
		uint32_t grid = 256u * per_cu;
		if(grid > nchan) grid = nchan;
		hipLaunchKernelGGL((evalw_kernel<MAXORD, true, 1>), dim3(grid), dim3(64), lds, s, P, B.chan, nframes, tail_n, jt, B.prep, B.cands, B.valid, dec, in_list, in_count, out_list, out_count);
	}
	else {
		// every channel of the batch (streams of more than 16 bits): two wavefronts per channel (FLACGPU_EVALW_WPC=1: one, for A/B runs)
		const int wpc = tune().evalw_wpc;
		if(wpc == 2 && P.ncslots >= 4)
			hipLaunchKernelGGL((evalw_kernel<MAXORD, false, 2>), dim3(nchan), dim3(128), evalw_lds_bytes<MAXORD>(P.blocksize, 2), s, P, B.chan, nframes, tail_n, jt, B.prep, B.cands, B.valid, dec, nullptr, nullptr, out_list, out_count);
		else
			hipLaunchKernelGGL((evalw_kernel<MAXORD, false, 1>), dim3(nchan), dim3(64), evalw_lds_bytes<MAXORD>(P.blocksize), s, P, B.chan, nframes, tail_n, jt, B.prep, B.cands, B.valid, dec, nullptr, nullptr, out_list, out_count);
	}
	return hipGetLastError();
}
hipError_t launch_evalw(const DevParams &P, uint32_t nframes, uint32_t tail_n, const JobTable *jt, const AnalyzeBuffers &B, SubDecision *dec,
                        const uint32_t *in_list, const uint32_t *in_count, uint32_t *out_list, uint32_t *out_count, hipStream_t s)
{
	if(nframes == 0) return hipSuccess;
	if(P.max_lpc_order <= 8) return launch_evalw_t<8>(P, nframes, tail_n, jt, B, dec, in_list, in_count, out_list, out_count, s);
	return launch_evalw_t<12>(P, nframes, tail_n, jt, B, dec, in_list, in_count, out_list, out_count, s);
}

} // namespace flacgpu
