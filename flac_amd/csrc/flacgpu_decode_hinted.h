// flac_amd/csrc/flacgpu_decode_hinted.h -- verifying a frame the pack kernel has just written, 16 samples per thread.
//
// Rice decoding is serial only because the START of a code is known once the code before it has been read.  The pack kernel
// (pack2_kernel, flacgpu_kernels.hip) knows where every 16-sample run of every subframe starts -- it computed those bit
// offsets to write the frame -- and hands them over as HINTS.  With them a workgroup verifies a frame with one thread per
// run instead of one lane per frame (flacgpu_decode.h, ~100 dependent instructions per sample):
//   * the frame header is parsed as before; every fixed-width field of a subframe header (type, wasted bits, warm-up
//     samples, precision, shift, coefficients, partition order) sits at a position that is arithmetic in the fields in
//     front of it, so the threads PEEK them;
//   * thread t takes the Rice parameter of its partition from the position the hint of the partition's first run names,
//     decodes its 16 codes from its own hint, and reports where it ended;
//   * THE HINTS ARE NOT TRUSTED.  The chain is checked: run 0 must start where the subframe header ends, run t must end
//     where run t+1 starts, the last run where the next subframe starts, the last subframe at the zero padding in front
//     of the CRC.  If the chain holds, every code was read at the position and with the parameter a sequential decoder
//     reads it at -- by induction over the runs -- whatever the hints' origin;
//   * the decoded RESIDUALS are compared with the residuals the input implies, e[i] = y[i] - (sum_j q[j] y[i-1-j] >> shift)
//     in exact arithmetic (y = the coded channel the input implies, shifted down by the wasted bits): if the warm-up samples
//     and all residuals agree, the decoder's recurrence x[i] = r[i] + (sum_j q[j] x[i-1-j] >> shift) reproduces y sample by
//     sample (induction over i; the decoder's 32-bit sums are exact for in-range samples, stream_decoder.c:3224-3232).
// The verdict of this pass is only ever "this frame is verified" or "SUSPECT": a suspect frame -- a real mismatch, a damaged
// frame, an escape-coded partition, a frame this pass does not cover -- is decoded again by the sequential decoder, which
// alone produces error reports.  So the pass must be SOUND (never accept what the sequential decoder would not) and need not
// be complete.  tests/test_decode_pin.py drives the host instantiation of this header against honest, damaged and random
// hints; the kernel (flacgpu_verify.hip: verify_hinted_kernel) is built from the same functions.
#ifndef FLACGPU_DECODE_HINTED_H
#define FLACGPU_DECODE_HINTED_H
#include "flacgpu_decode.h"

namespace flacgpu {

constexpr uint32_t HINT_RUN = 16;                 // samples per run (CHUNK of the pack kernel)
constexpr uint32_t HINT_MAX_RUNS = 256;           // runs per subframe the pass covers (blocks of up to 4096 samples)
constexpr uint32_t HINT_MAX_ORDER = 16;           // predictor orders the pack kernel that writes hints handles

// where the frame's bytes can be read as aligned big-endian words: `w` may point to global memory or to an LDS copy
struct PeekSrc {
	const uint32_t *w0;            // the aligned word that holds the first byte of the frame
	uint32_t nwords;               // words that may be read (index clamp: what lies beyond is never consumed legitimately)
	uint32_t skip;                 // bits of w0 in front of the frame
	uint32_t limit;                // bits of the frame body (CRC-16 excluded)
};
FLACGPU_HD inline uint32_t peek_word(const PeekSrc &S, uint32_t i) { return __builtin_bswap32(S.w0[i < S.nwords ? i : S.nwords - 1]); }
// n bits (0..32) at bit `pos` of the frame
FLACGPU_HD inline uint32_t peek_bits(const PeekSrc &S, uint32_t pos, uint32_t n)
{
	if(n == 0) return 0;
	const uint32_t a = S.skip + pos, wi = a >> 5, o = a & 31u;
	const uint64_t v = ((uint64_t)peek_word(S, wi) << 32) | peek_word(S, wi + 1);
	return (uint32_t)((v << o) >> (64 - n));
}
FLACGPU_HD inline int32_t peek_signed(const PeekSrc &S, uint32_t pos, uint32_t n)      // 1..32
{
	const uint32_t v = peek_bits(S, pos, n);
	return n >= 32 ? (int32_t)v : (int32_t)(v << (32 - n)) >> (32 - n);
}

// a bit reader that starts at bit `pos` of the frame (br_pos stays relative to the frame's first byte)
FLACGPU_HD inline void br_init_at(BitReader &b, const PeekSrc &S, uint32_t pos)
{
	const uint32_t tot = S.skip + pos;
	b.w0 = S.w0;
	b.wlast = S.w0 + (S.nwords - 1);
	b.wp = S.w0 + (tot >> 5);
	b.ah = 0; b.al = 0; b.nb = 0; b.skip = S.skip; b.limit = S.limit; b.bad = 0;
	b.pre = br_fetch(b);
	br_refill(b);
	const uint32_t r = tot & 31u;
	if(r) { b.ah <<= r; b.nb -= r; }
	br_refill(b);
}

// ---- the frame header from five words loaded at once (decode_frame_header reads it through the bit reader and checks its CRC-8
// byte by byte from memory: a dozen dependent round trips, which was a third of a hinted workgroup's life).  Same checks, same
// verdicts (tests/test_decode_pin.py compares the two on valid, damaged and random headers); a header is at most 16 bytes, which
// with up to three bytes of misalignment in front lie in five aligned words.
struct HeadWords { uint32_t w[5]; };
FLACGPU_HD inline HeadWords hinted_head_words(const PeekSrc &S)
{
	HeadWords W;
#pragma unroll
	for(int k = 0; k < 5; k++) W.w[k] = peek_word(S, (uint32_t)k);
	return W;
}
// n bits (1..32) at bit a (misalignment included) of the five words, a + n <= 160
FLACGPU_HD inline uint32_t head_bits(const HeadWords &W, uint32_t a, uint32_t n)
{
	const uint32_t wi = a >> 5, o = a & 31u;
	uint32_t hi = W.w[0], lo = W.w[1];
#pragma unroll
	for(int k = 1; k < 4; k++) if(wi == (uint32_t)k) { hi = W.w[k]; lo = W.w[k + 1]; }
	if(wi >= 4) { hi = W.w[4]; lo = 0; }
	const uint64_t v = ((uint64_t)hi << 32) | lo;
	return (uint32_t)((v << o) >> (64 - n));
}
// one byte through CRC-8 (poly 0x07): c * x^8 mod P, the columns of the matrix written out
FLACGPU_HD inline uint32_t crc8_byte(uint32_t c)
{
	uint32_t t = 0;
#pragma unroll
	for(int k = 0; k < 6; k++) t ^= ((c >> k) & 1u) ? (0x07u << k) : 0u;
	t ^= (c & 0x40u) ? 0xC7u : 0u;
	t ^= (c & 0x80u) ? 0x89u : 0u;
	return t & 0xffu;
}
FLACGPU_HD inline int hinted_frame_header_w(const HeadWords &W, const PeekSrc &S, const DecodeExpect &E, FrameHead &H, uint32_t *pos_out)
{
	uint32_t pos = 0;                                               // bits consumed, from the frame's first byte
#define HGET(n) (pos += (n), head_bits(W, S.skip + pos - (n), (n)))
	if(HGET(15) != 0x7ffcu || HGET(1) != 0) return DEC_ERROR;
	const uint32_t bs_code = HGET(4), sr_code = HGET(4), ca = HGET(4), bps_code = HGET(3);
	if(HGET(1) != 0) return DEC_ERROR;
	uint64_t fn;
	{
		const uint32_t b0 = HGET(8);
		uint32_t extra;
		if(b0 < 0x80u) { fn = b0; extra = 0; }
		else if((b0 & 0xe0u) == 0xc0u) { fn = b0 & 0x1fu; extra = 1; }
		else if((b0 & 0xf0u) == 0xe0u) { fn = b0 & 0x0fu; extra = 2; }
		else if((b0 & 0xf8u) == 0xf0u) { fn = b0 & 0x07u; extra = 3; }
		else if((b0 & 0xfcu) == 0xf8u) { fn = b0 & 0x03u; extra = 4; }
		else if((b0 & 0xfeu) == 0xfcu) { fn = b0 & 0x01u; extra = 5; }
		else return DEC_ERROR;
		for(uint32_t k = 0; k < extra; k++) { const uint32_t c = HGET(8); if((c & 0xc0u) != 0x80u) return DEC_ERROR; fn = (fn << 6) | (c & 0x3fu); }
	}
	uint32_t bs;
	if(bs_code == 0) return DEC_ERROR;
	else if(bs_code == 1) bs = 192;
	else if(bs_code <= 5) bs = 576u << (bs_code - 2);
	else if(bs_code == 6) bs = HGET(8) + 1;
	else if(bs_code == 7) bs = HGET(16) + 1;
	else bs = 256u << (bs_code - 8);
	if(sr_code == 12) (void)HGET(8);
	else if(sr_code == 13 || sr_code == 14) (void)HGET(16);
	else if(sr_code == 15) return DEC_ERROR;
	const uint32_t hdr_bytes = pos >> 3;                             // at most 15
	uint32_t crc = 0;
	for(uint32_t i = 0; i < hdr_bytes; i++) crc = crc8_byte(crc ^ head_bits(W, S.skip + 8 * i, 8));
	if(HGET(8) != crc || pos > S.limit) return DEC_ERROR;
#undef HGET
	const uint32_t bps_of = bps_code == 1 ? 8u : bps_code == 2 ? 12u : bps_code == 4 ? 16u : bps_code == 5 ? 20u : bps_code == 6 ? 24u : bps_code == 7 ? 32u : 0u;
	if(bps_code == 3) return DEC_ERROR;
	if(fn != E.frame_number || bs != E.n || (bps_code && bps_of != E.bps)) return DEC_ERROR;
	if((ca < 8 && ca + 1 != E.channels) || ca > 10 || (ca >= 8 && E.channels != 2)) return DEC_ERROR;
	H.ca = ca; H.n = bs;
	*pos_out = pos;
	return DEC_OK;
}

FLACGPU_HD inline int hinted_frame_header(const PeekSrc &S, const DecodeExpect &E, FrameHead &H, uint32_t *pos_out)
{
	const HeadWords W = hinted_head_words(S);
	return hinted_frame_header_w(W, S, E, H, pos_out);
}

// what a subframe header says, read by peeking (every field's position follows from the fields in front of it)
struct HintedSub {
	uint32_t ok;                   // 0: malformed, or outside what this pass covers -> the frame is suspect
	uint32_t type;                 // 0 constant, 1 verbatim, 2 fixed, 3 lpc
	uint32_t wasted, sb, order;
	uint32_t prec; int32_t shift;
	uint32_t pos_body;             // constant value / first verbatim or warm-up sample
	uint32_t pos_q;                // first coefficient (lpc)
	uint32_t r0;                   // first bit of the residual section's first partition (its parameter field)
	uint32_t plen, esc, po, psize;
	uint32_t wide_sum, narrow24;
	uint32_t end_fixed;            // constant / verbatim: first bit behind the subframe
};
FLACGPU_HD inline HintedSub hinted_subframe_head(const PeekSrc &S, uint32_t pos, uint32_t sbps_nominal, uint32_t n)
{
	HintedSub H;
	H.ok = 0; H.type = 0; H.wasted = 0; H.sb = 0; H.order = 0; H.prec = 0; H.shift = 0; H.pos_body = 0; H.pos_q = 0; H.r0 = 0;
	H.plen = 4; H.esc = 15; H.po = 0; H.psize = n; H.wide_sum = 1; H.narrow24 = 0; H.end_fixed = 0;
	if(pos + 8 > S.limit) return H;
	const uint32_t hb = peek_bits(S, pos, 8);
	if(hb & 0x80u) return H;
	const uint32_t t = (hb >> 1) & 0x3fu;
	uint32_t p = pos + 8;
	if(hb & 1u) {
		const uint32_t z = peek_bits(S, p, 32);
		if(z == 0) return H;                                     // (more than 32 wasted bits cannot be: sbps <= 33)
		const uint32_t lz = (uint32_t)__builtin_clz(z);
		H.wasted = lz + 1; p += lz + 1;
	}
	if(H.wasted >= sbps_nominal) return H;
	H.sb = sbps_nominal - H.wasted;
	if(H.sb > 32) return H;                                       // the 33-bit side channel: not covered
	H.pos_body = p;
	if(t == 0) { H.type = 0; H.end_fixed = p + H.sb; H.ok = H.end_fixed <= S.limit; return H; }
	if(t == 1) { H.type = 1; const uint64_t e = (uint64_t)p + (uint64_t)n * H.sb; H.end_fixed = (uint32_t)e; H.ok = e <= S.limit; return H; }
	bool lpc;
	if(t >= 8 && t <= 12) { H.order = t - 8; lpc = false; }
	else if(t >= 32) { H.order = t - 31; lpc = true; }
	else return H;
	if(H.order > n || H.order > HINT_MAX_ORDER) return H;
	H.type = lpc ? 3 : 2;
	p += H.order * H.sb;
	H.narrow24 = H.sb <= 24;
	if(lpc) {
		if(p + 9 > S.limit) return H;
		H.prec = peek_bits(S, p, 4) + 1;
		if(H.prec == 16) return H;
		H.shift = peek_signed(S, p + 4, 5);
		if(H.shift < 0) return H;
		H.pos_q = p + 9;
		p += 9 + H.order * H.prec;
		if(p > S.limit) return H;
		uint64_t abs_sum = 0;
		for(uint32_t j = 0; j < H.order; j++) { const int32_t t = peek_signed(S, H.pos_q + j * H.prec, H.prec); abs_sum += (uint32_t)(t < 0 ? -t : t); }
		H.wide_sum = dec_lpc_needs_wide_sum(H.sb, abs_sum, H.shift);      // stream_decoder.c:3240-3246
	}
	else H.wide_sum = H.sb + H.order > 32;                          // fixed.c:571-667
	if(p + 6 > S.limit) return H;
	const uint32_t method = peek_bits(S, p, 2);
	if(method > 1) return H;
	H.plen = method ? 5u : 4u; H.esc = method ? 31u : 15u;
	H.po = peek_bits(S, p + 2, 4);
	H.psize = n >> H.po;
	if(H.po && ((H.psize << H.po) != n || H.psize < H.order)) return H;
	if(H.psize % HINT_RUN != 0 || n % HINT_RUN != 0 || n / HINT_RUN > HINT_MAX_RUNS) return H;      // runs must not straddle partitions
	H.r0 = p + 6;
	H.ok = 1;
	return H;
}
// tap j of a fixed predictor of the given order (fixed.c:571: FIRs with binomial taps and shift 0)
FLACGPU_HD inline int32_t hinted_fixed_tap(uint32_t order, uint32_t j)
{
	if(order == 1) return j == 0 ? 1 : 0;
	if(order == 2) return j == 0 ? 2 : j == 1 ? -1 : 0;
	if(order == 3) return j == 0 ? 3 : j == 1 ? -3 : j == 2 ? 1 : 0;
	if(order == 4) return j == 0 ? 4 : j == 1 ? -6 : j == 2 ? 4 : j == 3 ? -1 : 0;
	return 0;
}

// Run t of a subframe: decode its codes (samples [16t, 16t+16) from `first` on: run 0 starts behind the warm-up samples) from
// bit `start` with parameter k, compare them with the residuals the signal implies.  yw[0..31] = y[16t-16 .. 16t+15] (values in
// front of sample 0 are never used).  Returns 0 when every residual agrees; *end = the bit behind the run's last code.
// MODE 0: 64-bit prediction sum (H.wide_sum); 1: 32-bit sum from the 24-bit multiplier (H.narrow24); 2: 32-bit sum, 32-bit multiplies
template <int MAXORD, typename ST, int MODE>
FLACGPU_HD inline uint32_t hinted_run_mode(const PeekSrc &S, uint32_t start, uint32_t k, uint32_t first, const ST (&yw)[32], const int32_t (&q)[MAXORD],
                                           const HintedSub &H, uint32_t *end)
{
	BitReader b;
	br_init_at(b, S, start);
	uint32_t bad = 0;
#pragma unroll
	for(int s = 0; s < (int)HINT_RUN; s++) {
		if((uint32_t)s >= first) {
			const uint32_t u = br_rice(b, k);
			const int32_t r = (int32_t)((u >> 1) ^ (0u - (u & 1u)));
			if(MODE == 0) {
				int64_t sum = 0;
#pragma unroll
				for(int j = 0; j < MAXORD; j++) sum += (int64_t)q[j] * (int64_t)yw[16 + s - 1 - j];
				bad |= (uint32_t)((int64_t)yw[16 + s] - (sum >> H.shift) != (int64_t)r);
			}
			else {
				uint32_t s32 = 0;
				if(MODE == 1) {
					int32_t hh[MAXORD];
#pragma unroll
					for(int j = 0; j < MAXORD; j++) hh[j] = (int32_t)yw[16 + s - 1 - j];
					s32 = FLACGPU_DOT24(q, hh);
				}
				else {
#pragma unroll
					for(int j = 0; j < MAXORD; j++) s32 += (uint32_t)q[j] * (uint32_t)(int32_t)yw[16 + s - 1 - j];
				}
				// y - pred == r as integers: the 32-bit difference must not have wrapped
				const int32_t pred = (int32_t)s32 >> H.shift;
				int32_t d;
				const bool ovf = __builtin_sub_overflow((int32_t)yw[16 + s], pred, &d);
				bad |= (uint32_t)ovf | (uint32_t)(d != r);
			}
		}
	}
	bad |= b.bad | (uint32_t)br_over(b);
	*end = (uint32_t)br_pos(b);
	return bad;
}
template <int MAXORD, typename ST>
FLACGPU_HD inline uint32_t hinted_run(const PeekSrc &S, uint32_t start, uint32_t k, uint32_t first, const ST (&yw)[32], const int32_t (&q)[MAXORD],
                                      const HintedSub &H, uint32_t *end)
{
	// (one decision per run, three straight-line bodies: the choice inside the unrolled loop cost two branches per sample)
	if(H.wide_sum) return hinted_run_mode<MAXORD, ST, 0>(S, start, k, first, yw, q, H, end);
	if(H.narrow24) return hinted_run_mode<MAXORD, ST, 1>(S, start, k, first, yw, q, H, end);
	return hinted_run_mode<MAXORD, ST, 2>(S, start, k, first, yw, q, H, end);
}

#ifndef __HIPCC__
// ---- the whole pass over one frame as ONE thread of control: the host pin (oracle/decode_pin.cpp).  The kernel runs the same
// steps with a thread per run; every decision it takes is one of the functions above.  hints: [C][HINT_MAX_RUNS].
// Returns 0: verified, 1: suspect.
template <int MAXORD>
inline int verify_frame_hinted_host(const uint8_t *p, size_t len, const uint8_t *buf_hi, const DecodeExpect &E, const int32_t *pcm, const uint32_t *hints)
{
	if(len < 6 || E.n % HINT_RUN != 0 || E.n / HINT_RUN > HINT_MAX_RUNS || E.n > E.blocksize) return 1;
	BitReader b;
	br_init(b, p, len - 2, buf_hi);
	PeekSrc S;
	S.w0 = b.w0; S.nwords = (uint32_t)(b.wlast - b.w0) + 1; S.skip = b.skip; S.limit = (uint32_t)b.limit;
	FrameHead FH;
	uint32_t pos = 0;
	if(hinted_frame_header(S, E, FH, &pos) != DEC_OK) return 1;
	const uint32_t C = E.channels, n = FH.n, nruns = n / HINT_RUN;
	int32_t *y = new int32_t[n + 16];
	uint32_t *ends = new uint32_t[HINT_MAX_RUNS];
	int suspect = 0;
	for(uint32_t ch = 0; ch < C && !suspect; ch++) {
		const HintedSub H = hinted_subframe_head(S, pos, coded_bps(E.bps, FH.ca, ch), n);
		if(!H.ok) { suspect = 1; break; }
		// the signal the input implies for this coded channel, shifted down by the wasted bits (which must be zero in it)
		for(uint32_t i = 0; i < 16; i++) y[i] = 0;
		for(uint32_t i = 0; i < n; i++) {
			const int64_t v = coded_expectation(pcm + (size_t)i * C, FH.ca, ch);
			if(v & (((int64_t)1 << H.wasted) - 1)) suspect = 1;
			const int64_t ys = v >> H.wasted;
			if(ys != (int64_t)(int32_t)ys) suspect = 1;              // a 33-bit value: not covered
			y[16 + i] = (int32_t)ys;
		}
		if(suspect) break;
		if(H.type == 0) {
			const int32_t v = peek_signed(S, H.pos_body, H.sb);
			for(uint32_t i = 0; i < n; i++) if(y[16 + i] != v) suspect = 1;
			pos = H.end_fixed;
		}
		else if(H.type == 1) {
			for(uint32_t i = 0; i < n; i++) if(peek_signed(S, H.pos_body + i * H.sb, H.sb) != y[16 + i]) suspect = 1;
			pos = H.end_fixed;
		}
		else {
			if(H.order > (uint32_t)MAXORD) { suspect = 1; break; }
			int32_t q[MAXORD];
			for(int j = 0; j < MAXORD; j++) q[j] = (uint32_t)j >= H.order ? 0 : H.type == 3 ? peek_signed(S, H.pos_q + (uint32_t)j * H.prec, H.prec) : hinted_fixed_tap(H.order, (uint32_t)j);
			for(uint32_t i = 0; i < H.order; i++) if(peek_signed(S, H.pos_body + i * H.sb, H.sb) != y[16 + i]) suspect = 1;
			const uint32_t *hs = hints + (size_t)ch * HINT_MAX_RUNS;
			if(hs[0] != H.r0) suspect = 1;
			for(uint32_t t = 0; t < nruns && !suspect; t++) {
				const uint32_t part = (t * HINT_RUN) / H.psize, t0 = part * H.psize / HINT_RUN;
				const uint32_t kpos = hs[t0];
				if(kpos + H.plen > S.limit || hs[t] > S.limit) { suspect = 1; break; }
				const uint32_t k = peek_bits(S, kpos, H.plen);
				if(k == H.esc) { suspect = 1; break; }                 // raw partitions: the sequential decoder's business
				int32_t yw[32];
				for(int u = 0; u < 32; u++) yw[u] = y[t * HINT_RUN + (uint32_t)u];
				const uint32_t first = t == 0 ? H.order : 0;
				if(hinted_run<MAXORD, int32_t>(S, hs[t] + (t == t0 ? H.plen : 0), k, first, yw, q, H, &ends[t])) suspect = 1;
			}
			for(uint32_t t = 0; t + 1 < nruns && !suspect; t++) if(ends[t] != hs[t + 1]) suspect = 1;
			if(!suspect) pos = ends[nruns - 1];
		}
	}
	if(!suspect) {
		// zero bits up to the byte boundary, and the body ends exactly where the CRC-16 starts
		const uint32_t rem = pos & 7u;
		if(pos > S.limit) suspect = 1;
		else if(rem && peek_bits(S, pos, 8 - rem) != 0) suspect = 1;
		else if(pos + (rem ? 8 - rem : 0) != S.limit) suspect = 1;
	}
	delete[] y; delete[] ends;
	return suspect;
}

// honest hints for the tests: the run starts a sequential reading of the frame finds (what pack2_kernel exports).
// Returns 0 when the frame parses and every residual subframe is within what the hinted pass covers.
inline int make_hints_host(const uint8_t *p, size_t len, const uint8_t *buf_hi, const DecodeExpect &E, uint32_t *hints)
{
	if(len < 6 || E.n % HINT_RUN != 0 || E.n / HINT_RUN > HINT_MAX_RUNS) return 1;
	BitReader b;
	br_init(b, p, len - 2, buf_hi);
	FrameHead FH;
	if(decode_frame_header(b, p, E, FH) != DEC_OK) return 1;
	PeekSrc S;
	S.w0 = b.w0; S.nwords = (uint32_t)(b.wlast - b.w0) + 1; S.skip = b.skip; S.limit = (uint32_t)b.limit;
	uint32_t pos = (uint32_t)br_pos(b);
	const uint32_t n = FH.n;
	for(uint32_t ch = 0; ch < E.channels; ch++) {
		const HintedSub H = hinted_subframe_head(S, pos, coded_bps(E.bps, FH.ca, ch), n);
		uint32_t *hs = hints + (size_t)ch * HINT_MAX_RUNS;
		for(uint32_t t = 0; t < HINT_MAX_RUNS; t++) hs[t] = 0;
		if(!H.ok) return 1;
		if(H.type < 2) { pos = H.end_fixed; continue; }
		BitReader r;
		br_init_at(r, S, H.r0);
		uint32_t k = 0;
		for(uint32_t i = 0; i < n; i++) {
			if(i % HINT_RUN == 0) hs[i / HINT_RUN] = (uint32_t)br_pos(r);
			if(i % H.psize == 0) { k = br_get(r, H.plen); if(k == H.esc) return 1; }
			if(i >= H.order) (void)br_rice(r, k);
			if(r.bad || br_over(r)) return 1;
		}
		pos = (uint32_t)br_pos(r);
	}
	return 0;
}
#endif

} // namespace flacgpu
#endif
