// flac_amd/csrc/flacgpu_evalg.h -- what the wavefront-per-channel evaluation kernels share (flacgpu_evalg.hip: 16-bit pairs,
// v_dot2_i32_i16; flacgpu_evalw.hip: 32-bit samples, v_mad_i32_i24 / v_mad_i64_i32): the transposed LDS image, the DPP scans, the
// packed Rice node search of a PAIR of candidates (find_best_partition_order_ / set_partitioned_rice_, stream_encoder.c:4701-5075),
// the first-minimum bookkeeping and the decision record (stream_encoder.c:4147-4290).
#ifndef FLACGPU_EVALG_H
#define FLACGPU_EVALG_H
#include "flacgpu_dev.h"
#include "flacgpu_devfn.h"

namespace flacgpu {

#ifndef EVALG_WAVES_PER_SIMD
#define EVALG_WAVES_PER_SIMD 4
#endif

constexpr int EG_MAXC = 32;               // candidate slots of a channel this kernel takes (lane c holds candidate c's record)

// ---- DPP helpers (the compiler sees these, so it places the wait states itself) --------------------------------------
template <int CTRL, int ROWMASK>
__device__ __forceinline__ uint32_t dpp_add(uint32_t v)
{
	return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWMASK, 0xf, false);
}
// inclusive prefix sum over the 64 lanes: Hillis-Steele inside a row of 16, then the row totals
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v)
{
	v = dpp_add<0x111, 0xf>(v);            // row_shr:1
	v = dpp_add<0x112, 0xf>(v);            // row_shr:2
	v = dpp_add<0x114, 0xf>(v);            // row_shr:4
	v = dpp_add<0x118, 0xf>(v);            // row_shr:8
	v = dpp_add<0x142, 0xa>(v);            // row_bcast:15 into rows 1 and 3
	v = dpp_add<0x143, 0xc>(v);            // row_bcast:31 into rows 2 and 3
	return v;
}
// the same without the last step: lanes 31 and 63 end with the totals of their halves
__device__ __forceinline__ uint32_t half_scan_incl(uint32_t v)
{
	v = dpp_add<0x111, 0xf>(v);
	v = dpp_add<0x112, 0xf>(v);
	v = dpp_add<0x114, 0xf>(v);
	v = dpp_add<0x118, 0xf>(v);
	v = dpp_add<0x142, 0xa>(v);
	return v;
}
__device__ __forceinline__ uint32_t rdlane(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }

// LDS image of a channel, TRANSPOSED: word j of lane L's run at (j * 65 + L + 1) -- the 64 lanes reading "their word j" read 64
// consecutive words (no bank conflict, no padding per lane), and the word in front of a run is the last word of the run of
// lane L - 1: column L.  Column 0 is lane 0's history: zero.  8.3 KB per 4096-sample channel (the per-lane regions with their
// own history copies took 10.5 KB: one wavefront more per SIMD).
constexpr uint32_t EG_ROW = 65 * 4;      // bytes per row

// ---- one node pass of the Rice search ---------------------------------------------------------------------------
// lane constants of a pass: LDS byte offsets of the two prefix-sum entries whose difference is twice the node's
// |residual| sum; nsf9 = samples of a full partition at this level + 9 (0: the lane has no node in this pass);
// dtoff = byte offset of the level's row of the divisor table; p0 = the node is partition 0 (`order` samples short)
struct EgPass { uint32_t a_start, a_end, nsf9, dtoff; bool p0; };
// set_partitioned_rice_ (stream_encoder.c:4997-5046) on sum2 = 2 * sum, sum < 2^29, without branches:
//   k    = ilog2(((sum - 1) * div) >> 18) + 1 for sum >= 2 and a non-zero quotient, else 0; div = 0x40000 / ns
//   bits = 4 + (1 + k) * ns + (k ? sum >> (k - 1) : sum << 1) - (ns >> 1)              (:4929-4950, the estimate)
// with (sum - 1) * div >> 18 == mul_hi(2 * (sum - 1), div << 13), and ilog2(x) + 1 == the binary exponent of (float)x
// (x < 2^20 here: exact).  ns - (ns >> 1) + 4 == (ns + 9) >> 1.
__device__ __forceinline__ void rice_pass(const unsigned char *lds, const EgPass &C, uint32_t ord_lane /* this lane's candidate's order */, uint32_t rl1, uint32_t &k, uint32_t &bits)
{
	const uint32_t o = C.p0 ? ord_lane : 0u;
	const uint32_t ns9 = C.nsf9 - o;                                       // (a lane without a node has nsf9 = 0, p0 = false and a zero sum:
	const uint32_t ns = ns9 - 9u;                                          //  k = 0 and bits = 0 whatever ns wraps to)
	const uint32_t dsh = *(const uint32_t *)(lds + C.dtoff + o * 4u);
	const uint32_t sum2 = *(const uint32_t *)(lds + C.a_end) - *(const uint32_t *)(lds + C.a_start);
	const uint32_t a2 = (sum2 > 2u ? sum2 : 2u) - 2u;
	const uint32_t x = __umulhi(a2, dsh);
	// ilog2(x) + 1, 0 for x == 0 (x < 2^20): the leading-zero count of 2 x + 1 is that of x less one, and 31 for x == 0.  (Round 4 took
	// the exponent of (float)x: the compiler saw a 64-bit product behind x and built the float with its 64-bit sequence, seven
	// instructions where this is three -- profiles/r05_evalg_isa_histogram.txt)
	uint32_t kk = 31u - (uint32_t)__builtin_clz((x << 1) | 1u);
	kk = umin32(kk, rl1);
	k = kk;
	bits = __umul24(kk, ns) + (ns9 >> 1) + (sum2 >> kk);
}


// ---- the search state of a channel -----------------------------------------------------------------------------------------
// LDS behind the image: [prefix sums 2 x 66][divisor table 7 x (MAXORD + 1)][best parameters 64 B]
template <int MAXORD>
__host__ __device__ inline uint32_t eg_tail_bytes() { return 2 * 66 * 4 + 7 * (MAXORD + 1) * 4 + 64; }
struct EgSearch {
	EgPass PA, PC, PD;
	uint32_t mD, e, D, rl1, ps_off, frame_max_po;
	uint32_t best_est, best_ci, best_po;
};
// tables and lane constants; level m: nodes of 2^m lanes, partition order frame_max_po - (m - e); searched when e <= m <= e + D
template <int MAXORD>
__device__ __forceinline__ void eg_search_setup(EgSearch &R, unsigned char *smem, uint32_t img_bytes, uint32_t S, uint32_t frame_max_po, uint32_t frame_min_po, uint32_t rice_limit, int lane,
                                                const JobTable *__restrict__ jt)
{
	uint32_t *ps = (uint32_t *)(smem + img_bytes);                           // [2][66]
	uint32_t *dt = ps + 2 * 66;                                              // [7][MAXORD + 1]: (0x40000 / ((S << m) - o)) << 13
	const uint32_t ps_off = img_bytes, dt_off = img_bytes + 2 * 66 * 4;
	const uint32_t e = 6 - frame_max_po, D = frame_max_po - frame_min_po;
	for(uint32_t t = (uint32_t)lane; t < 7 * (MAXORD + 1); t += 64) {
		const uint32_t m = t / (MAXORD + 1), o = t - m * (MAXORD + 1);
		dt[t] = jt->eg_div[m][o];                                            // (0x40000 / ((S << m) - o)) << 13, from the host (JobTable)
	}
	if(lane < 2) ps[lane * 66] = 0;
	const uint32_t half = (uint32_t)lane >> 5, j = (uint32_t)lane & 31u;
	const bool a0 = e == 0;
	R.PA.a_start = ps_off + (uint32_t)lane * 4; R.PA.a_end = R.PA.a_start + 4;          // (pass B: + 66 * 4)
	R.PA.nsf9 = a0 ? S + 9 : 0; R.PA.dtoff = dt_off; R.PA.p0 = a0 && lane == 0;
	const bool a1 = e <= 1 && 1 <= e + D;
	R.PC.a_start = ps_off + (half * 66 + 2 * j) * 4; R.PC.a_end = R.PC.a_start + 8;
	R.PC.nsf9 = a1 ? 2 * S + 9 : 0; R.PC.dtoff = dt_off + (MAXORD + 1) * 4; R.PC.p0 = a1 && j == 0;
	uint32_t mD = j < 16 ? 2u : j < 24 ? 3u : j < 28 ? 4u : j < 30 ? 5u : j < 31 ? 6u : 7u;
	const uint32_t idx = j - (32u - (128u >> mD));                                 // (mD == 7: unused)
	const bool aD = mD <= 6 && e <= mD && mD <= e + D;
	R.PD.a_start = ps_off + (half * 66 + (aD ? idx << mD : 0u)) * 4; R.PD.a_end = R.PD.a_start + (aD ? (4u << mD) : 0u);
	R.PD.nsf9 = aD ? (S << mD) + 9 : 0; R.PD.dtoff = dt_off + (aD ? mD : 0u) * (MAXORD + 1) * 4; R.PD.p0 = aD && idx == 0;
	R.mD = aD ? mD : 7u;
	R.e = e; R.D = D; R.rl1 = rice_limit - 1; R.ps_off = ps_off; R.frame_max_po = frame_max_po;
	R.best_est = 0xffffffffu; R.best_ci = 0xffffffffu; R.best_po = 0;
}
// one candidate of the pair as the search sees it
struct EgCand { uint32_t order, precision, ci; };
// The Rice search of a pair from the lanes' |residual| sums v0 / v1 (< 2^23), the estimates, and the bookkeeping of the channel's
// first minimum (the Rice parameters of a new best go to kbest).  Candidates are met in increasing order; strict <: the earlier
// one keeps a tie (stream_encoder.c:4191,4266).
__device__ __forceinline__ void eg_pair_search(EgSearch &R, unsigned char *smem, uint8_t *kbest, uint32_t v0, uint32_t v1, const EgCand &A, const EgCand &B, bool two,
                                               uint32_t nfixed, uint32_t hdr, uint32_t sbps, int lane)
{
	uint32_t *ps = (uint32_t *)(smem + R.ps_off);
	const uint32_t e = R.e, D = R.D, rl1 = R.rl1;
	ps[1 + lane] = wave_scan_incl(v0 << 1);
	ps[66 + 1 + lane] = wave_scan_incl(v1 << 1);
	__builtin_amdgcn_wave_barrier();
	uint32_t tot0[7], tot1[7];                                                // per level m: total bits of the level (uniform)
#pragma unroll
	for(int m = 0; m < 7; m++) { tot0[m] = 0; tot1[m] = 0; }
	uint32_t kA = 0, kB = 0, kC = 0, kD = 0, b;
	if(e == 0) {
		rice_pass(smem, R.PA, A.order, rl1, kA, b);
		tot0[0] = rdlane(wave_scan_incl(b), 63);
		EgPass PB = R.PA; PB.a_start += 66 * 4; PB.a_end += 66 * 4;
		rice_pass(smem, PB, B.order, rl1, kB, b);
		tot1[0] = rdlane(wave_scan_incl(b), 63);
	}
	const uint32_t ord_lane = lane < 32 ? A.order : B.order;
	if(e <= 1 && 1 <= e + D) {
		rice_pass(smem, R.PC, ord_lane, rl1, kC, b);
		b = half_scan_incl(b);
		tot0[1] = rdlane(b, 31); tot1[1] = rdlane(b, 63);
	}
	if(e + D >= 2) {
		rice_pass(smem, R.PD, ord_lane, rl1, kD, b);
		// the levels sit in aligned lane groups of their own size (16 | 8 | 4 | 2 | 1): each total is read at the butterfly
		// stage that has summed exactly its group
		tot0[6] = rdlane(b, 30); tot1[6] = rdlane(b, 62);
		b = bfly_add<0>(b); tot0[5] = rdlane(b, 28); tot1[5] = rdlane(b, 60);
		b = bfly_add<1>(b); tot0[4] = rdlane(b, 24); tot1[4] = rdlane(b, 56);
		b = bfly_add<2>(b); tot0[3] = rdlane(b, 16); tot1[3] = rdlane(b, 48);
		b = bfly_add<3>(b); tot0[2] = rdlane(b, 0); tot1[2] = rdlane(b, 32);
	}
	__builtin_amdgcn_wave_barrier();                                          // (the prefix sums are rewritten by the next pair)

	// strict <, highest order first: ties keep the higher order (stream_encoder.c:4735-4763)
#pragma unroll
	for(int slot = 0; slot < 2; slot++) {
		if(slot == 1 && !two) break;
		const EgCand &X = slot ? B : A;
		uint32_t bb = 0, bm = 0;
		bool have = false;
#pragma unroll
		for(int m = 0; m < 7; m++) {
			if((uint32_t)m >= e && (uint32_t)m - e <= D) {
				const uint32_t bits = 6 + (slot ? tot1[m] : tot0[m]);
				if(!have || bits < bb) { bb = bits; bm = (uint32_t)m; have = true; }
			}
		}
		const uint32_t est = X.ci < nfixed ? sat_add_u32(hdr + X.order * sbps, bb) : sat_add_u32(hdr + 4 + 5 + X.order * (X.precision + sbps), bb);
		if(est > 0 && est < R.best_est) {
			R.best_est = est; R.best_ci = X.ci; R.best_po = R.frame_max_po - (bm - e);
			const uint32_t half = (uint32_t)lane >> 5, j = (uint32_t)lane & 31u;
			if(bm == 0) kbest[lane] = (uint8_t)(slot ? kB : kA);
			else if(bm == 1) { if(half == (uint32_t)slot) kbest[j] = (uint8_t)kC; }
			else if(half == (uint32_t)slot && R.mD == bm) kbest[j - (32u - (128u >> bm))] = (uint8_t)kD;
			__builtin_amdgcn_wave_barrier();
		}
	}
}
// the decision: first minimum in the reference's evaluation order (verbatim -> constant | fixed -> LPC); lane c holds candidate c's
// order / precision / shift / taps (cq[0..MAXORD))
template <int MAXORD, int NQ>
__device__ __forceinline__ void eg_decide(const EgSearch &R, const DevParams &P, const ChanPrep &pr, uint32_t n, const uint8_t *kbest, uint32_t c_order, uint32_t c_prec, uint32_t c_shift,
                                          const int32_t (&cq)[NQ], SubDecision *dec, ChanPrep *prep_out, int lane)
{
	const uint32_t wasted = pr.wasted, sbps = pr.sbps, hdr = 8 + wasted;
	const uint32_t best_ci = R.best_ci;
	uint32_t best_type = 1, best_order = 0, dpo = 0, best_precision = 0;
	int32_t best_shift = 0, best_constant = 0, best_constant_hi = 0;
	uint32_t best_bits = pr.verbatim_bits;
	if(pr.flags & PREP_CONSTANT) {
		const uint32_t bits = hdr + sbps;
		if(bits < best_bits) { best_type = 0; best_constant = pr.constant; best_constant_hi = pr.constant_hi; best_bits = bits; }
	}
	if(best_ci != 0xffffffffu && R.best_est < best_bits) {
		best_bits = R.best_est; dpo = R.best_po;
		best_type = best_ci < P.nfixed ? 2 : 3;
		best_order = rdlane(c_order, best_ci); best_precision = rdlane(c_prec, best_ci); best_shift = (int32_t)rdlane(c_shift, best_ci);
	}
	if(best_bits == 0xffffffffu) { best_type = 1; best_bits = hdr + n * sbps; }   // stream_encoder.c:4281
	uint32_t rice2 = 0;
	if(best_type >= 2) {
		uint32_t big = 0;
		if((uint32_t)lane < (1u << dpo)) {
			const uint8_t kk = kbest[lane];
			dec->params[lane] = kk;
			if(kk >= 15) big = 1;
		}
		rice2 = __any((int)big) ? 1u : 0u;                         // stream_encoder.c:4786-4791
	}
	if(lane < MAX_ORDER) {
		int32_t qv = 0;
#pragma unroll
		for(int j = 0; j < MAXORD; j++) { const int32_t t = (int32_t)rdlane((uint32_t)cq[j], best_ci & 63u); if(lane == j) qv = t; }
		dec->q[lane] = best_type == 3 ? qv : 0;
	}
	if(lane == 0) {
		dec->bits = best_bits;
		dec->type = (uint8_t)best_type; dec->order = (uint8_t)best_order; dec->wasted = (uint8_t)wasted;
		dec->po = (uint8_t)dpo; dec->rice2 = (uint8_t)rice2; dec->precision = (uint8_t)best_precision;
		dec->shift = (int8_t)best_shift; dec->which = (uint8_t)pr.which;
		dec->constant = best_constant; dec->constant_hi = best_constant_hi; dec->fmt = pr.fmt;
		prep_out->handled = EVG_HANDLED;
	}
}

} // namespace flacgpu
#endif
