// flac_amd/csrc/flacgpu_decode.h -- a FLAC frame decoder as ONE thread of control per frame, written so that the same
// source runs as a lane of a GPU wavefront (flacgpu_verify.hip: 64 frames per wavefront, thousands of wavefronts per
// batch) and, for tests, as plain host code (oracle/decode_pin.cpp).  It is the encoder's self check (SURVEY.md 8f row 3:
// FLAC__stream_encoder_set_verify): what the reference does with its own stream decoder inside write_bitbuffer_
// (src/libFLAC/stream_encoder.c:3000-3018, verify_write_callback_ :5155-5230), i.e. the frame reader of
// src/libFLAC/stream_decoder.c (read_frame_ :2373, read_frame_header_ :2609, read_subframe_ :2866, read_subframe_fixed_ :3082,
// read_subframe_lpc_ :3155, read_residual_partitioned_rice_ :3299), the restoration lpc.c:978-1578 / fixed.c:571-667 and the
// inter-channel undo stream_decoder.c:2503-2553.  Written from the format, not from that code.
//
// Rice decoding is serial within a subframe and the subframes of a frame are found one after the other, so the unit of
// parallelism is the frame: a 16384-frame batch is 256 wavefronts.  The decoder pushes every decoded CODED-channel
// sample (after the wasted-bits shift) into a SINK:
//   * the batch kernel's sink compares it with the value the input PCM implies for that coded channel (left, right,
//     (L+R)>>1, L-R: the decorrelation is a bijection, so "every coded channel equals its expectation" <=> "every
//     output sample equals the input") -- no decoded sample is ever stored;
//   * the detail pass (run for the first bad frame only) stores the channels, undoes the decorrelation and locates the
//     first differing output sample exactly as the reference reports it.
#ifndef FLACGPU_DECODE_H
#define FLACGPU_DECODE_H
#include <stdint.h>
#include <stddef.h>

#ifndef FLACGPU_HD
#define FLACGPU_HD
#endif
// low 32 bits of sum_j a[j] * b[j] for values that fit 24 bits signed (the GPU has a full-rate 24-bit multiply-add)
#ifndef FLACGPU_DOT24
namespace flacgpu {
template <int M> inline uint32_t dot24_plain(const int32_t (&a)[M], const int32_t (&b)[M]) { uint32_t s = 0; for(int j = 0; j < M; j++) s += (uint32_t)a[j] * (uint32_t)b[j]; return s; }
}
#define FLACGPU_DOT24(a, b) flacgpu::dot24_plain(a, b)
#endif

namespace flacgpu {

// what the decoder must find in a frame header for it to belong to this stream at this position
struct DecodeExpect {
	uint32_t channels, bps, blocksize;       // of the stream
	uint32_t n;                              // samples this frame must hold (the short last block has fewer)
	uint64_t frame_number;
};
enum { DEC_OK = 0, DEC_MISMATCH = 1, DEC_ERROR = 2 };

// ---- MSB-first bit reader over global memory ---------------------------------------------------------------------------
// A 64-bit window kept as two 32-bit halves (64-bit shifts cost a lane twice as much as 32-bit ones and the lane is
// latency bound): `ah` always holds the next 32 unread bits once nb >= 32, `al` what follows, nb = valid bits in ah:al,
// zeros behind them.  The next word is always in flight (a lane has no second wavefront to hide its load latency
// behind): `pre` holds it as loaded, the byte swap happens when it enters the window.  Loads never leave [first aligned
// word of the frame, the aligned word holding the buffer's last byte]: beyond that the address is clamped (what comes
// back is never consumed legitimately -- a frame that reads past its own end is an error -- it only must not fault).
struct BitReader {
	const uint32_t *wp;            // next aligned word to load
	const uint32_t *wlast;         // the aligned word that holds the last byte of the buffer the frames lie in
	const uint32_t *w0;            // the aligned word that holds the first byte of the frame
	uint32_t pre;                  // the word in front of wp, as loaded (little endian)
	uint32_t ah, al;               // the window
	uint32_t nb;                   // valid bits in it
	uint32_t skip;                 // bits of w0 in front of the frame
	uint64_t limit;                // bits the frame body holds: reading beyond is an error
	uint32_t bad;
};
// shifts by 0..32 (a machine shift takes its count modulo 32)
FLACGPU_HD inline uint32_t dec_shl(uint32_t v, uint32_t s) { const uint32_t a = s >> 1; return (v << a) << (s - a); }
FLACGPU_HD inline uint32_t dec_shr(uint32_t v, uint32_t s) { const uint32_t a = s >> 1; return (v >> a) >> (s - a); }
FLACGPU_HD inline uint32_t br_fetch(BitReader &b)
{
	const uint32_t *q = b.wp < b.wlast ? b.wp : b.wlast;
	b.wp++;
	return *q;
}
// takes the word in flight into the window when the low half is empty: afterwards nb >= 33
FLACGPU_HD inline void br_refill(BitReader &b)
{
	if(b.nb <= 32) {
		const uint32_t w = __builtin_bswap32(b.pre);
		b.pre = br_fetch(b);
		b.ah |= dec_shr(w, b.nb);                                      // (al is empty: all valid bits sit in ah)
		b.al = dec_shl(w, 32 - b.nb);
		b.nb += 32;
	}
}
// drops t bits, 1 <= t <= 32, t <= nb
FLACGPU_HD inline void br_drop(BitReader &b, uint32_t t)
{
	const uint32_t s = 32 - t;                                          // 0..31
	b.ah = s ? (b.ah << t) | (b.al >> s) : b.al;                        // (one funnel shift on the GPU: v_alignbit_b32)
	b.al = (b.al << (t - 1)) << 1;
	b.nb -= t;
}
// bits consumed so far, counted from the first byte of the frame: words taken into the window (one more is in flight)
FLACGPU_HD inline uint64_t br_pos(const BitReader &b) { return (uint64_t)(b.wp - b.w0 - 1) * 32 - b.nb - b.skip; }
FLACGPU_HD inline bool br_over(const BitReader &b) { return br_pos(b) > b.limit; }
// frame bytes [p, p + nbytes) inside a buffer that ends at buf_hi
FLACGPU_HD inline void br_init(BitReader &b, const uint8_t *p, size_t nbytes, const uint8_t *buf_hi)
{
	const uint32_t mis = (uint32_t)((uintptr_t)p & 3);
	b.w0 = (const uint32_t *)(p - mis);                                 // (pointer arithmetic, not an integer cast: the address space survives)
	b.wlast = (const uint32_t *)((buf_hi - 1) - ((uintptr_t)(buf_hi - 1) & 3));
	b.wp = b.w0;
	b.ah = 0; b.al = 0; b.nb = 0; b.skip = mis * 8; b.limit = (uint64_t)nbytes * 8; b.bad = 0;
	b.pre = br_fetch(b);
	br_refill(b);                                                       // nb = 32: the first word in ah
	if(b.skip) { b.ah <<= b.skip; b.nb -= b.skip; }
	br_refill(b);
}
FLACGPU_HD inline uint32_t br_get(BitReader &b, uint32_t n)            // 0 <= n <= 32
{
	if(n == 0) return 0;
	br_refill(b);                                                      // now nb >= 33
	const uint32_t v = b.ah >> (32 - n);
	br_drop(b, n);
	return v;
}
FLACGPU_HD inline int32_t br_get_signed(BitReader &b, uint32_t n)      // 1 <= n <= 32
{
	const uint32_t v = br_get(b, n);
	return n >= 32 ? (int32_t)v : (int32_t)(v << (32 - n)) >> (32 - n);
}
FLACGPU_HD inline int64_t br_get_sample(BitReader &b, uint32_t n)      // up to 33 bits (the side channel of a 32-bit stream)
{
	if(n <= 32) return (int64_t)br_get_signed(b, n);
	const uint64_t hi = br_get(b, n - 32), lo = br_get(b, 32);
	const uint64_t v = (hi << 32) | lo;
	return (int64_t)(v << (64 - n)) >> (64 - n);
}
FLACGPU_HD inline uint32_t br_unary(BitReader &b)                      // zeros in front of the next one bit
{
	uint32_t z = 0;
	for(;;) {
		br_refill(b);                                                  // nb >= 33: ah is all valid
		if(b.ah != 0) {
			const uint32_t lz = (uint32_t)__builtin_clz(b.ah);
			br_drop(b, lz + 1);
			return z + lz;
		}
		z += 32; br_drop(b, 32);
		if(br_over(b)) { b.bad = 1; return z; }                            // ran off the frame: stop
	}
}
// one Rice-coded residual with parameter k (0..30): unary quotient, stop bit, k low bits -- in one step when the whole code
// lies in the 32 bits at hand (nearly always), else piecewise
FLACGPU_HD inline uint32_t br_rice(BitReader &b, uint32_t k)
{
	br_refill(b);
	if(b.ah != 0) {
		const uint32_t lz = (uint32_t)__builtin_clz(b.ah), total = lz + 1 + k;
		if(total <= 32) {
			const uint32_t low = k ? (b.ah >> (32 - total)) & ((1u << k) - 1u) : 0u;      // (v_bfe_u32)
			br_drop(b, total);
			return (lz << k) | low;
		}
	}
	const uint32_t msbs = br_unary(b);
	return (msbs << k) | br_get(b, k);
}

FLACGPU_HD inline uint32_t dec_ilog2(uint32_t v) { return 31u - (uint32_t)__builtin_clz(v); }
// FLAC__bitmath_silog2 (bitmath.c): bits of a two's complement number
FLACGPU_HD inline uint32_t dec_silog2(int64_t v)
{
	if(v == 0) return 0;
	if(v == -1) return 2;
	if(v < 0) v = -(v + 1);
	return 63u - (uint32_t)__builtin_clzll((unsigned long long)v) + 2;
}
// The reference decoder's choice between 32-bit wrap-around restoration and the 64-bit sum (stream_decoder.c:3240-3246, 1.5.0):
// FLAC__lpc_restore_signal when FLAC__lpc_max_residual_bps and FLAC__lpc_max_prediction_before_shift_bps (lpc.c:942-968, both
// built on the sum of the taps' magnitudes) are at most 32, FLAC__lpc_restore_signal_wide otherwise.  For the in-range audio
// an encoder produces the two restorations agree wherever either rule picks the narrow one; the rule is restated exactly so
// that a frame from anywhere (flacgpu_verify_batch_device takes any bytes) decodes as the reference decodes it.
FLACGPU_HD inline bool dec_lpc_needs_wide_sum(uint32_t sb, uint64_t abs_sum_of_taps, int32_t shift)
{
	const uint64_t maxabs = (uint64_t)1 << (sb - 1);
	const uint64_t before = maxabs * abs_sum_of_taps;                                   // lpc.c:942-950
	const uint64_t after = (uint64_t)(-1 * ((-1 * (int64_t)before) >> shift));         // lpc.c:965
	return !(dec_silog2((int64_t)(maxabs + after)) <= 32 && dec_silog2((int64_t)before) <= 32);
}

// CRC-8 of the frame header (poly 0x07), bit by bit: at most 16 bytes per frame
FLACGPU_HD inline uint32_t dec_crc8(const uint8_t *p, uint32_t n)
{
	uint32_t c = 0;
	for(uint32_t i = 0; i < n; i++) { c ^= p[i]; for(int k = 0; k < 8; k++) c = (c & 0x80u) ? ((c << 1) ^ 0x07u) & 0xffu : (c << 1) & 0xffu; }
	return c;
}

// CRC-16 of a whole frame body (poly 0x8005, init 0, crc.c:376), bit by bit: the detail pass only (one frame per batch at
// most); the batch pass checks the footers with the span-parallel kernel of flacgpu_verify.hip
FLACGPU_HD inline uint32_t dec_crc16(const uint8_t *p, size_t n)
{
	uint32_t c = 0;
	for(size_t i = 0; i < n; i++) { c ^= (uint32_t)p[i] << 8; for(int k = 0; k < 8; k++) c = (c & 0x8000u) ? ((c << 1) ^ 0x8005u) & 0xffffu : (c << 1) & 0xffffu; }
	return c;
}

struct FrameHead { uint32_t ca; uint32_t n; };

// frame header: sync, blocking strategy, block size / sample rate / channel assignment / sample size codes, UTF-8 frame
// number, optional block size and sample rate fields, CRC-8 (format: SURVEY.md appendix B)
FLACGPU_HD inline int decode_frame_header(BitReader &b, const uint8_t *p, const DecodeExpect &E, FrameHead &H)
{
	if(br_get(b, 15) != 0x7ffcu || br_get(b, 1) != 0) return DEC_ERROR;           // sync + reserved, fixed-blocksize stream
	const uint32_t bs_code = br_get(b, 4), sr_code = br_get(b, 4), ca = br_get(b, 4), bps_code = br_get(b, 3);
	if(br_get(b, 1) != 0) return DEC_ERROR;
	uint64_t fn;
	{
		const uint32_t b0 = br_get(b, 8);
		uint32_t extra;
		if(b0 < 0x80u) { fn = b0; extra = 0; }
		else if((b0 & 0xe0u) == 0xc0u) { fn = b0 & 0x1fu; extra = 1; }
		else if((b0 & 0xf0u) == 0xe0u) { fn = b0 & 0x0fu; extra = 2; }
		else if((b0 & 0xf8u) == 0xf0u) { fn = b0 & 0x07u; extra = 3; }
		else if((b0 & 0xfcu) == 0xf8u) { fn = b0 & 0x03u; extra = 4; }
		else if((b0 & 0xfeu) == 0xfcu) { fn = b0 & 0x01u; extra = 5; }
		else return DEC_ERROR;
		for(uint32_t k = 0; k < extra; k++) { const uint32_t c = br_get(b, 8); if((c & 0xc0u) != 0x80u) return DEC_ERROR; fn = (fn << 6) | (c & 0x3fu); }
	}
	uint32_t bs;
	if(bs_code == 0) return DEC_ERROR;
	else if(bs_code == 1) bs = 192;
	else if(bs_code <= 5) bs = 576u << (bs_code - 2);
	else if(bs_code == 6) bs = br_get(b, 8) + 1;
	else if(bs_code == 7) bs = br_get(b, 16) + 1;
	else bs = 256u << (bs_code - 8);
	if(sr_code == 12) (void)br_get(b, 8);
	else if(sr_code == 13 || sr_code == 14) (void)br_get(b, 16);
	else if(sr_code == 15) return DEC_ERROR;
	const uint32_t hdr_bytes = (uint32_t)(br_pos(b) >> 3);
	if(br_get(b, 8) != dec_crc8(p, hdr_bytes) || br_over(b)) return DEC_ERROR;
	const uint32_t bps_of = bps_code == 1 ? 8u : bps_code == 2 ? 12u : bps_code == 4 ? 16u : bps_code == 5 ? 20u : bps_code == 6 ? 24u : bps_code == 7 ? 32u : 0u;
	if(bps_code == 3) return DEC_ERROR;
	if(fn != E.frame_number || bs != E.n || (bps_code && bps_of != E.bps)) return DEC_ERROR;
	if((ca < 8 && ca + 1 != E.channels) || ca > 10 || (ca >= 8 && E.channels != 2)) return DEC_ERROR;
	H.ca = ca; H.n = bs;
	return DEC_OK;
}

// One subframe: header, then sample by sample -- constant / verbatim / warm-up / predicted (fixed predictors are FIRs with
// binomial taps and shift 0, fixed.c:571) -- each handed to sink(i, value << wasted).  MAXORD: taps kept in registers
// (orders above it are a decode error for this instantiation: the caller picks MAXORD from the stream's settings; 32 covers
// the format).  ST: int32_t, or int64_t when a 33-bit sample can occur (side channel of a 32-bit stream).
template <int MAXORD, typename ST, class SINK>
FLACGPU_HD inline int decode_subframe(BitReader &b, uint32_t n, uint32_t sbps_nominal, SINK &sink)
{
	if(br_get(b, 1) != 0) return DEC_ERROR;
	const uint32_t type = br_get(b, 6);
	uint32_t wasted = 0;
	if(br_get(b, 1)) wasted = br_unary(b) + 1;
	if(wasted >= sbps_nominal) return DEC_ERROR;
	const uint32_t sb = sbps_nominal - wasted;
	if(type == 0) {                                             // CONSTANT
		const int64_t v = br_get_sample(b, sb);
		for(uint32_t i = 0; i < n; i++) sink(i, (int64_t)((uint64_t)v << wasted));
		return b.bad ? DEC_ERROR : DEC_OK;
	}
	if(type == 1) {                                             // VERBATIM
		for(uint32_t i = 0; i < n; i++) {
			const int64_t v = br_get_sample(b, sb);
			sink(i, (int64_t)((uint64_t)v << wasted));
		}
		return br_over(b) ? DEC_ERROR : DEC_OK;
	}
	uint32_t order;
	bool lpc;
	if(type >= 8 && type <= 12) { order = type - 8; lpc = false; }
	else if(type >= 32) { order = type - 31; lpc = true; }
	else return DEC_ERROR;                                      // reserved
	if(order > n || order > (uint32_t)MAXORD) return DEC_ERROR;
	// the last MAXORD samples, in slots that ROTATE instead of shifting: the residual loop below is unrolled MAXORD times and
	// iteration s of a pass writes slot s, so every index is a constant after unrolling and no sample is ever moved
	// (slot t holds sample i0 - MAXORD + t at the start of a pass that begins with sample i0)
	ST h[MAXORD];
	int32_t q[MAXORD];
#pragma unroll
	for(int j = 0; j < MAXORD; j++) { h[j] = 0; q[j] = 0; }
	for(uint32_t i = 0; i < order; i++) {
		const int64_t v = br_get_sample(b, sb);
		sink(i, (int64_t)((uint64_t)v << wasted));
		const uint32_t slot = i + (uint32_t)MAXORD - order;         // the first pass begins with sample `order`
#pragma unroll
		for(int t = 0; t < MAXORD; t++) if((uint32_t)t == slot) h[t] = (ST)v;
	}
	int32_t shift = 0;
	bool wide_sum = true;                                       // 64-bit prediction sum
	const bool narrow24 = sb <= 24;                             // (taps have at most 15 bits)
	if(lpc) {
		const uint32_t prec = br_get(b, 4) + 1;
		if(prec == 16) return DEC_ERROR;
		shift = br_get_signed(b, 5);
		if(shift < 0) return DEC_ERROR;
#pragma unroll
		for(int j = 0; j < MAXORD; j++) if((uint32_t)j < order) q[j] = br_get_signed(b, prec);
		uint64_t abs_sum = 0;
#pragma unroll
		for(int j = 0; j < MAXORD; j++) if((uint32_t)j < order) abs_sum += (uint32_t)(q[j] < 0 ? -q[j] : q[j]);
		wide_sum = dec_lpc_needs_wide_sum(sb, abs_sum, shift);
	}
	else {
		if(order == 1) { q[0] = 1; }
		else if(order == 2) { q[0] = 2; q[1] = -1; }
		else if(order == 3) { q[0] = 3; q[1] = -3; q[2] = 1; }
		else if(order == 4) { q[0] = 4; q[1] = -6; q[2] = 4; q[3] = -1; }
		wide_sum = sb + order > 32;                             // fixed.c:571-667: 64-bit restoration when the differences may need it
	}
	// residual: coding method, partition order, then per partition the Rice parameter (or the escape code and a raw width)
	const uint32_t method = br_get(b, 2);
	if(method > 1) return DEC_ERROR;
	const uint32_t plen = method ? 5u : 4u, esc = method ? 31u : 15u;
	const uint32_t po = br_get(b, 4);
	const uint32_t psize = n >> po;
	if(po && ((psize << po) != n || psize < order)) return DEC_ERROR;
	if(br_over(b)) return DEC_ERROR;
	uint32_t next_part = order, k = 0, raw = 0;                 // sample index at which the next partition starts
	bool escaped = false;
	uint32_t part = 0;
	bool err = false;
	for(uint32_t i = order; i < n && !err; ) {
		// (the pass has ONE exit, at its end: the 24-bit multiply-add is inline asm, which the compiler treats as convergent, and
		//  a loop with a convergent operation and a second exit is not unrolled -- the slots would become a dynamically indexed
		//  array in scratch memory)
#pragma unroll
		for(int s = 0; s < MAXORD; s++) {
			if(i < n && !err) {
				while(i == next_part && !err) {                     // (a partition 0 that holds no residual at all is legal: psize == order)
					k = br_get(b, plen);
					escaped = k == esc;
					if(escaped) raw = br_get(b, 5);
					part++;
					next_part = po ? part * psize : n;
					err = br_over(b);                               // (checked per partition, not per sample: a lane that runs off its
				}                                                   //  frame reads zeros / clamped words until the partition ends)
				int64_t r;
				if(escaped) r = raw ? (int64_t)br_get_signed(b, raw) : 0;
				else {
					const uint32_t u = br_rice(b, k);
					r = (int64_t)(int32_t)((u >> 1) ^ (0u - (u & 1u)));
				}
				// tap j multiplies sample i-1-j: written earlier in this pass (slot s-1-j) or left from the pass before (slot MAXORD-1-j+s)
				ST hs[MAXORD];
#pragma unroll
				for(int j = 0; j < MAXORD; j++) hs[j] = j < s ? h[s - 1 - j] : h[MAXORD - 1 - j + s];
				int64_t sum = 0;
				if(wide_sum) {
#pragma unroll
					for(int j = 0; j < MAXORD; j++) sum += (int64_t)q[j] * (int64_t)hs[j];
				}
				else if(narrow24) {
					// samples and taps below 2^23 in magnitude: the low 32 bits of every product come out of the 24-bit multiplier
					// (full rate on the GPU; a 32-bit multiply runs at a quarter of it)
					int32_t hh[MAXORD];
#pragma unroll
					for(int j = 0; j < MAXORD; j++) hh[j] = (int32_t)hs[j];
					sum = (int64_t)(int32_t)FLACGPU_DOT24(q, hh);
				}
				else {
					uint32_t s32 = 0;
#pragma unroll
					for(int j = 0; j < MAXORD; j++) s32 += (uint32_t)q[j] * (uint32_t)(int32_t)hs[j];
					sum = (int64_t)(int32_t)s32;
				}
				const int64_t v = r + (sum >> shift);
				sink(i, (int64_t)((uint64_t)v << wasted));
				h[s] = (ST)v;
				i++;
			}
		}
	}
	if(err) return DEC_ERROR;
	return (b.bad || br_over(b)) ? DEC_ERROR : DEC_OK;
}

// after the last subframe: zero bits up to the byte boundary, and the body must end exactly where the CRC-16 starts
FLACGPU_HD inline int decode_frame_tail(BitReader &b)
{
	const uint32_t rem = (uint32_t)(br_pos(b) & 7);
	if(rem && br_get(b, 8 - rem) != 0) return DEC_ERROR;
	return (b.bad || br_pos(b) != b.limit) ? DEC_ERROR : DEC_OK;
}
// nominal width of coded channel ch under channel assignment ca (the side channel carries one bit more)
FLACGPU_HD inline uint32_t coded_bps(uint32_t bps, uint32_t ca, uint32_t ch)
{
	const bool side = (ca == 8 && ch == 1) || (ca == 9 && ch == 0) || (ca == 10 && ch == 1);
	return bps + (side ? 1u : 0u);
}
// the value the input implies for coded channel ch at one inter-channel sample (x = the C input samples of it)
FLACGPU_HD inline int64_t coded_expectation(const int32_t *x, uint32_t ca, uint32_t ch)
{
	if(ca < 8) return (int64_t)x[ch];
	const int64_t l = x[0], r = x[1];
	if(ca == 8) return ch == 0 ? l : l - r;
	if(ca == 9) return ch == 0 ? l - r : r;
	return ch == 0 ? (l + r) >> 1 : l - r;
}

// ---- the fast pass: one frame against its input; DEC_OK / DEC_MISMATCH / DEC_ERROR -------------------------------------
// pcm: the frame's input, interleaved int32 [n][C]
template <int MAXORD, typename ST>
FLACGPU_HD inline int verify_frame_fast(const uint8_t *p, size_t len, const uint8_t *buf_hi, const DecodeExpect &E, const int32_t *pcm)
{
	if(len < 6) return DEC_ERROR;
	BitReader b;
	br_init(b, p, len - 2, buf_hi);
	FrameHead H;
	if(decode_frame_header(b, p, E, H) != DEC_OK) return DEC_ERROR;
	const uint32_t C = E.channels;
	uint32_t differ = 0;
	for(uint32_t ch = 0; ch < C; ch++) {
		const uint32_t ca = H.ca;
		auto sink = [&](uint32_t i, int64_t v) { differ |= (uint32_t)(v != coded_expectation(pcm + (size_t)i * C, ca, ch)); };
		if(decode_subframe<MAXORD, ST>(b, H.n, coded_bps(E.bps, H.ca, ch), sink) != DEC_OK) return DEC_ERROR;
	}
	if(decode_frame_tail(b) != DEC_OK) return DEC_ERROR;
	return differ ? DEC_MISMATCH : DEC_OK;
}

// ---- the detail pass: decode into x[C][stride] (int64), undo the decorrelation (stream_decoder.c:2503-2553), find the
// first output sample that differs from the input in stream order (sample-major, then channel) ----------------------------
struct DecodeDetail { int32_t status; uint32_t channel, sample; int32_t expected, got; };
template <int MAXORD, typename ST>
FLACGPU_HD inline void verify_frame_detail(const uint8_t *p, size_t len, const uint8_t *buf_hi, const DecodeExpect &E, const int32_t *pcm,
                                           int64_t *x, size_t stride, DecodeDetail &D)
{
	D.status = DEC_ERROR; D.channel = 0; D.sample = 0; D.expected = 0; D.got = 0;
	if(len < 6 || dec_crc16(p, len - 2) != (((uint32_t)p[len - 2] << 8) | p[len - 1])) return;
	BitReader b;
	br_init(b, p, len - 2, buf_hi);
	FrameHead H;
	if(decode_frame_header(b, p, E, H) != DEC_OK) return;
	const uint32_t C = E.channels, n = H.n;
	for(uint32_t ch = 0; ch < C; ch++) {
		int64_t *xc = x + (size_t)ch * stride;
		auto sink = [&](uint32_t i, int64_t v) { xc[i] = v; };
		if(decode_subframe<MAXORD, ST>(b, n, coded_bps(E.bps, H.ca, ch), sink) != DEC_OK) return;
	}
	if(decode_frame_tail(b) != DEC_OK) return;
	if(H.ca == 8) for(uint32_t i = 0; i < n; i++) x[stride + i] = x[i] - x[stride + i];
	else if(H.ca == 9) for(uint32_t i = 0; i < n; i++) x[i] += x[stride + i];
	else if(H.ca == 10) for(uint32_t i = 0; i < n; i++) {
		const int64_t sd = x[stride + i];
		const int64_t mid = (int64_t)(((uint64_t)x[i] << 1) | ((uint64_t)sd & 1));
		x[i] = (mid + sd) >> 1; x[stride + i] = (mid - sd) >> 1;
	}
	D.status = DEC_OK;
	for(uint32_t i = 0; i < n; i++)
		for(uint32_t ch = 0; ch < C; ch++) {
			const int32_t want = pcm[(size_t)i * C + ch];
			if(x[(size_t)ch * stride + i] != (int64_t)want) {
				D.status = DEC_MISMATCH; D.channel = ch; D.sample = i; D.expected = want; D.got = (int32_t)x[(size_t)ch * stride + i];
				return;
			}
		}
}

} // namespace flacgpu
#endif
