// flac_amd/csrc/flacgpu_stream_decode.h -- decoding FLAC streams this engine did NOT write (`flac -t`, `flac -d`):
// the per-position and per-frame work of the reference's stream decoder as functions that run as ONE LANE of a GPU wavefront
// (flacgpu_stream_decode.hip) and, for tests, as plain host code (oracle/stream_decode_pin.cpp).
//
// What the reference does sequentially, byte by byte (src/libFLAC/stream_decoder.c):
//   frame_sync_          :2321  find 0xFF followed by 0xF8 / 0xF9
//   read_frame_header_   :2624  parse what follows, CRC-8; a header that does not hold makes the search go on a few bytes later
//   read_subframe_*      :2949-3297, read_residual_partitioned_rice_ :3299, read_zero_padding_ :3362
//   restoration          lpc.c:978-1578 (32-bit wrap-around, 64-bit sum, 33-bit), fixed.c:571-667
//   read_frame_          :2373  CRC-16, undo_channel_coding :3476, the bounds check :2466-2483, missing-frame silence :2485-2554
// is split here into
//   sd_parse_candidate   everything that can be said about a sync code from the <= 18 bytes behind it (every byte position in
//                        parallel: the scan kernel);
//   sd_decode_frame      one frame body per lane from a candidate whose header holds: subframes in the reference's order of
//                        checks, so that a damaged frame fails with the reference's error, and with the reference's integer
//                        semantics (32-bit wrap-around where its buffers are 32-bit), so that ANY bytes decode as it decodes them;
//   sd_undo_channels     the inter-channel step and the bounds check on one inter-channel sample (the thread-parallel finish kernel);
// and the sequential part that remains -- which candidate the search reaches after which -- is flacgpu_stream_walk.h: host code over
// the candidate table, a few nanoseconds per frame.
// Written from the format and the behaviour of the functions cited, not from their code.
#ifndef FLACGPU_STREAM_DECODE_H
#define FLACGPU_STREAM_DECODE_H
#include "flacgpu_decode.h"

namespace flacgpu {

// what a candidate / a frame body came to: FLAC__StreamDecoderErrorStatus + 1 (include/FLAC/stream_decoder.h), then states of this decoder
enum : uint8_t {
	SD_OK = 0, SD_LOST_SYNC = 1, SD_BAD_HEADER = 2, SD_CRC_MISMATCH = 3, SD_UNPARSEABLE = 4, SD_BAD_METADATA = 5, SD_OUT_OF_BOUNDS = 6, SD_MISSING_FRAME = 7,
	SD_EOS = 8,           // the stream ends inside (the reference's read callback reports END_OF_STREAM)
	SD_RETRY = 9,         // internal: decode again with the instance that keeps 32 taps and multiplies in 32 bits
	SD_NOT_DECODED = 10,  // internal: header did not hold, nothing to decode
	SD_DEFERRED = 11      // internal: header holds but does not look like one of this stream's: decoded only if the search gets there
};

// the STREAMINFO fields the frame reader consults (stream_decoder.c:2706-2711, 2775-2780, 2808-2811, 2919-2934)
struct SdInfo {
	uint32_t has_streaminfo;
	uint32_t min_blocksize, max_blocksize, sample_rate, channels, bps;
};

// one sync code: 32 bytes
struct StreamCand {
	uint64_t pos;          // byte offset of the 0xFF
	uint64_t number;       // the coded number: a sample number when `variable`, else a frame number
	uint32_t sample_rate;
	uint16_t blocksize;    // (0: the header does not say -- reserved code)
	uint8_t  hdr_len;      // bytes up to and including the CRC-8 (when the header was read that far)
	uint8_t  hstat;        // SD_OK, SD_BAD_HEADER, SD_UNPARSEABLE, SD_EOS
	uint8_t  resume;       // hstat != SD_OK: the search goes on at pos + resume
	uint8_t  channels, ca /* 0 independent, 1 left/side, 2 right/side, 3 mid/side */, bps;
	uint8_t  variable;
	uint8_t  pad[3];
};
static_assert(sizeof(StreamCand) == 32, "StreamCand is 32 bytes");

// what decoding a candidate's frame came to: 16 bytes
struct StreamBody {
	uint64_t spec;         // where the finish kernel put the frame's first sample (inter-channel sample index), ~0: nowhere
	uint32_t len;          // bytes of the frame incl. the CRC-16 (bstat SD_OK / SD_CRC_MISMATCH / SD_OUT_OF_BOUNDS)
	uint8_t  bstat;        // SD_OK, SD_LOST_SYNC, SD_UNPARSEABLE, SD_EOS, SD_CRC_MISMATCH, SD_OUT_OF_BOUNDS, SD_NOT_DECODED
	uint8_t  oob_mask;     // channels with a sample outside the frame's sample width (one error each, stream_decoder.c:2470-2481)
	uint8_t  wrote;        // the finish kernel wrote the frame at `spec`
	uint8_t  pad_error;    // bstat SD_LOST_SYNC / SD_UNPARSEABLE: bit 0 the bits up to the byte boundary behind the failure are not zero -- one more LOST_SYNC; bit 1: the failure was an over-long Rice code
};
static_assert(sizeof(StreamBody) == 16, "StreamBody is 16 bytes");

FLACGPU_HD inline bool sd_is_sync(uint32_t b0, uint32_t b1) { return b0 == 0xffu && (b1 >> 1) == 0x7cu; }

// CRC-8 over n bytes fetched through `get`
template <class GET>
FLACGPU_HD inline uint32_t sd_crc8(GET &get, uint64_t c, uint32_t n)
{
	uint32_t crc = 0;
	for(uint32_t i = 0; i < n; i++) { crc ^= get(c + i); for(int k = 0; k < 8; k++) crc = (crc & 0x80u) ? ((crc << 1) ^ 0x07u) & 0xffu : (crc << 1) & 0xffu; }
	return crc;
}

// The header behind the sync code at byte c of a stream of nbytes bytes (get(i) = byte i), as read_frame_header_ goes through it
// (stream_decoder.c:2624-2944): which error it ends with, and where the search goes on then.
//   * a 0xFF as third or fourth byte: BAD_HEADER, and that byte is looked at again as a possible sync (:2669-2676);
//   * sample-rate code 15: BAD_HEADER at once (:2759-2762);
//   * a number that is not UTF-8, or block size 65536: BAD_HEADER, the last byte read is looked at again (:2823-2829, :2872-2878);
//   * CRC-8 mismatch: BAD_HEADER, the search goes on behind the CRC byte (:2911-2915);
//   * reserved codes: the header is read to its CRC all the same, then UNPARSEABLE_STREAM (:2936-2940);
//   * the stream ends inside the header: the decoder stops (END_OF_STREAM, :2664).
template <class GET>
FLACGPU_HD inline void sd_parse_candidate(GET &get, uint64_t nbytes, uint64_t c, const SdInfo &I, StreamCand &R)
{
	R.pos = c; R.number = 0; R.sample_rate = 0; R.blocksize = 0; R.hdr_len = 0; R.hstat = SD_OK; R.resume = 0;
	R.channels = 0; R.ca = 0; R.bps = 0; R.variable = 0; R.pad[0] = R.pad[1] = R.pad[2] = 0;
	bool unparseable = false;
	uint32_t len = 2;                                           // bytes read so far
#define SD_NEED(k) do { if(c + len + (k) > nbytes) { R.hstat = SD_EOS; R.hdr_len = (uint8_t)len; return; } } while(0)
#define SD_BAD(res) do { R.hstat = SD_BAD_HEADER; R.resume = (uint8_t)(res); R.hdr_len = (uint8_t)len; return; } while(0)
	const uint32_t h1 = get(c + 1);
	SD_NEED(1);
	const uint32_t h2 = get(c + 2); len = 3;
	if(h2 == 0xffu) SD_BAD(2);
	SD_NEED(1);
	const uint32_t h3 = get(c + 3); len = 4;
	if(h3 == 0xffu) SD_BAD(3);
	uint32_t bs = 0, bs_hint = 0, sr = 0, sr_hint = 0;
	const uint32_t bs_code = h2 >> 4, sr_code = h2 & 15u;
	if(bs_code == 0) unparseable = true;
	else if(bs_code == 1) bs = 192;
	else if(bs_code <= 5) bs = 576u << (bs_code - 2);
	else if(bs_code <= 7) bs_hint = bs_code;
	else bs = 256u << (bs_code - 8);
	if(sr_code == 0) { if(I.has_streaminfo) sr = I.sample_rate; else unparseable = true; }
	else if(sr_code <= 11)
		sr = sr_code == 1 ? 88200u : sr_code == 2 ? 176400u : sr_code == 3 ? 192000u : sr_code == 4 ? 8000u : sr_code == 5 ? 16000u : sr_code == 6 ? 22050u
		   : sr_code == 7 ? 24000u : sr_code == 8 ? 32000u : sr_code == 9 ? 44100u : sr_code == 10 ? 48000u : 96000u;
	else if(sr_code <= 14) sr_hint = sr_code;
	else SD_BAD(4);
	const uint32_t cx = h3 >> 4;
	if(cx & 8u) { R.channels = 2; if((cx & 7u) <= 2) R.ca = (uint8_t)((cx & 7u) + 1); else unparseable = true; }
	else { R.channels = (uint8_t)(cx + 1); R.ca = 0; }
	const uint32_t bx = (h3 & 0x0eu) >> 1;
	if(bx == 0) { if(I.has_streaminfo) R.bps = (uint8_t)I.bps; else unparseable = true; }
	else if(bx == 3) unparseable = true;
	else R.bps = (uint8_t)(bx == 1 ? 8 : bx == 2 ? 12 : bx == 4 ? 16 : bx == 5 ? 20 : bx == 6 ? 24 : 32);
	if(h3 & 1u) unparseable = true;
	// the number: UTF-8 style, 36 bits when it counts samples, 31 when it counts frames (bitreader.c:928-1040)
	const bool variable = (h1 & 1u) || (I.has_streaminfo && I.min_blocksize != I.max_blocksize);
	R.variable = variable ? 1 : 0;
	{
		SD_NEED(1);
		const uint32_t b0 = get(c + len); len++;
		uint64_t v; uint32_t extra;
		if(!(b0 & 0x80u)) { v = b0; extra = 0; }
		else if((b0 & 0xe0u) == 0xc0u) { v = b0 & 0x1fu; extra = 1; }
		else if((b0 & 0xf0u) == 0xe0u) { v = b0 & 0x0fu; extra = 2; }
		else if((b0 & 0xf8u) == 0xf0u) { v = b0 & 0x07u; extra = 3; }
		else if((b0 & 0xfcu) == 0xf8u) { v = b0 & 0x03u; extra = 4; }
		else if((b0 & 0xfeu) == 0xfcu) { v = b0 & 0x01u; extra = 5; }
		else if(variable && b0 == 0xfeu) { v = 0; extra = 6; }
		else SD_BAD(len - 1);
		for(; extra; extra--) {
			SD_NEED(1);
			const uint32_t x = get(c + len); len++;
			if(!(x & 0x80u) || (x & 0x40u)) SD_BAD(len - 1);
			v = (v << 6) | (x & 0x3fu);
		}
		R.number = v;
	}
	if(bs_hint) {
		SD_NEED(1);
		uint32_t x = get(c + len); len++;
		if(bs_hint == 7) { SD_NEED(1); x = (x << 8) | get(c + len); len++; }
		bs = x + 1;
		if(bs > 65535u) SD_BAD(len - 1);
	}
	if(sr_hint) {
		SD_NEED(1);
		uint32_t x = get(c + len); len++;
		if(sr_hint != 12) { SD_NEED(1); x = (x << 8) | get(c + len); len++; }
		sr = sr_hint == 12 ? x * 1000u : sr_hint == 13 ? x : x * 10u;
	}
	SD_NEED(1);
	const uint32_t crc = get(c + len);
	const uint32_t want = sd_crc8(get, c, len);
	len++;
	R.hdr_len = (uint8_t)len;
	R.blocksize = (uint16_t)bs; R.sample_rate = sr;
	if(crc != want) { R.hstat = SD_BAD_HEADER; R.resume = (uint8_t)len; return; }
	if(unparseable) { R.hstat = SD_UNPARSEABLE; R.resume = (uint8_t)len; return; }
#undef SD_NEED
#undef SD_BAD
}

// Where the finish kernel puts a frame before the walk has run: the sample number its header implies (exact for every stream whose
// frames are all there; the walk checks, and what it finds elsewhere is decoded again into place)
FLACGPU_HD inline uint64_t sd_spec_sample(const StreamCand &K, const SdInfo &I)
{
	if(K.variable) return K.number;
	return K.number * (uint64_t)(I.has_streaminfo ? I.min_blocksize : K.blocksize);
}
FLACGPU_HD inline uint32_t sd_nominal_bps(uint32_t bps, uint32_t ca, uint32_t ch)
{
	const bool side = (ca == 1 && ch == 1) || (ca == 2 && ch == 0) || (ca == 3 && ch == 1);
	return bps + (side ? 1u : 0u);
}

// one Rice code with parameter k; *bad: the unary part is longer than a 32-bit residual allows (bitreader_read_rice_signed_block.c:
// "limit = UINT32_MAX >> parameter", not applied for parameter 0)
FLACGPU_HD inline uint32_t sd_rice(BitReader &b, uint32_t k, uint32_t &bad)
{
	br_refill(b);
	if(b.ah != 0) {
		const uint32_t lz = (uint32_t)__builtin_clz(b.ah), total = lz + 1 + k;
		if(total <= 32) {
			const uint32_t low = k ? (b.ah >> (32 - total)) & ((1u << k) - 1u) : 0u;
			br_drop(b, total);
			return (lz << k) | low;
		}
	}
	const uint32_t msbs = br_unary(b);
	if(k && msbs > (0xffffffffu >> k)) bad = 1;
	return (msbs << k) | br_get(b, k);
}

// One subframe of n samples whose channel is `nominal` bits wide, every sample handed to sink(i, value) as the reference holds it
// after read_subframe_ (the wasted bits shifted back in: in 32 bits, or in 64 for the 33-bit side channel, stream_decoder.c:3027-3047).
// MAXORD: taps kept in registers; EXACT: no 24-bit multiplier shortcut.  A subframe this instance cannot take (order > MAXORD, or
// a value outside 24 bits met the shortcut) returns SD_RETRY.  ST: int32_t, or int64_t when a 33-bit channel can occur.
// *fail_pos: where the reference's reader stands when it reports SD_LOST_SYNC / SD_UNPARSEABLE, in bits from the frame's start --
// read_frame_ goes on to read_zero_padding_ from there (:2423-2425) and a second LOST_SYNC follows when those bits are not zero.
template <int MAXORD, bool EXACT, typename ST, class SINK>
FLACGPU_HD inline int sd_decode_subframe(BitReader &b, uint32_t n, uint32_t nominal, SINK &sink, uint64_t *fail_pos)
{
#define SD_FAIL(code) do { *fail_pos = br_pos(b); return (b.bad || br_over(b)) ? (int)SD_EOS : (int)(code); } while(0)
	const uint32_t x0 = br_get(b, 8);
	const uint32_t x = x0 & 0xfeu;
	uint32_t wasted = 0;
	if(x0 & 1u) {
		wasted = br_unary(b) + 1;
		if(b.bad || br_over(b)) return SD_EOS;
		if(wasted >= nominal) SD_FAIL(SD_LOST_SYNC);                       // :2966-2970
	}
	if(br_over(b)) return SD_EOS;
	const uint32_t sb = nominal - wasted;
	if(x & 0x80u) SD_FAIL(SD_LOST_SYNC);                                   // :2979
	const bool r33 = sizeof(ST) == 8 && sb == 33;                          // (then wasted == 0)
	// a value as the reference's buffers hold it: int32 (wrap-around) unless the channel really has 33 bits; then the wasted bits
	auto emit = [&](uint32_t i, int64_t v) {
		if(r33) sink(i, v);
		else if(nominal <= 32) sink(i, (int64_t)(int32_t)((uint32_t)(int32_t)v << wasted));
		else sink(i, (int64_t)((uint64_t)(int64_t)(int32_t)v << wasted));
	};
	if(x == 0) {                                                           // CONSTANT :3050
		const int64_t v = br_get_sample(b, sb);
		if(b.bad || br_over(b)) return SD_EOS;
		for(uint32_t i = 0; i < n; i++) emit(i, v);
		return SD_OK;
	}
	if(x == 2) {                                                           // VERBATIM :3258
		for(uint32_t i = 0; i < n; i++) emit(i, br_get_sample(b, sb));
		return (b.bad || br_over(b)) ? (int)SD_EOS : (int)SD_OK;
	}
	uint32_t order;
	bool lpc;
	if(x < 16) SD_FAIL(SD_UNPARSEABLE);                                    // :2990
	else if(x <= 24) { order = (x >> 1) & 7u; lpc = false; }
	else if(x < 64) SD_FAIL(SD_UNPARSEABLE);                               // :3009
	else { order = ((x >> 1) & 31u) + 1; lpc = true; }
	if(n <= order) SD_FAIL(SD_LOST_SYNC);                                  // :2997, :3016
	if(order > (uint32_t)MAXORD) return SD_RETRY;
	ST h[MAXORD];
	int32_t q[MAXORD];
#pragma unroll
	for(int j = 0; j < MAXORD; j++) { h[j] = 0; q[j] = 0; }
	uint32_t wild = 0;                                                     // a value outside 24 bits was seen (the shortcut's condition)
	for(uint32_t i = 0; i < order; i++) {
		int64_t v = br_get_sample(b, sb);
		if(!r33) v = (int64_t)(int32_t)v;
		emit(i, v);
		wild |= (uint32_t)((uint32_t)(int32_t)v + 0x800000u) >> 24;
		const uint32_t slot = i + (uint32_t)MAXORD - order;
#pragma unroll
		for(int t = 0; t < MAXORD; t++) if((uint32_t)t == slot) h[t] = (ST)v;
	}
	if(b.bad || br_over(b)) return SD_EOS;
	int32_t shift = 0;
	bool wide_sum = false;
	if(lpc) {
		const uint32_t prec = br_get(b, 4) + 1;
		if(br_over(b)) return SD_EOS;
		if(prec == 16) SD_FAIL(SD_LOST_SYNC);                              // :3185
		shift = br_get_signed(b, 5);
		if(br_over(b)) return SD_EOS;
		if(shift < 0) SD_FAIL(SD_LOST_SYNC);                               // :3195
#pragma unroll
		for(int j = 0; j < MAXORD; j++) if((uint32_t)j < order) q[j] = br_get_signed(b, prec);
		if(br_over(b)) return SD_EOS;
		uint64_t abs_sum = 0;
#pragma unroll
		for(int j = 0; j < MAXORD; j++) if((uint32_t)j < order) abs_sum += (uint32_t)(q[j] < 0 ? -q[j] : q[j]);
		wide_sum = r33 || dec_lpc_needs_wide_sum(sb, abs_sum, shift);      // :3240-3253
	}
	else {
		if(order == 1) { q[0] = 1; }
		else if(order == 2) { q[0] = 2; q[1] = -1; }
		else if(order == 3) { q[0] = 3; q[1] = -3; q[2] = 1; }
		else if(order == 4) { q[0] = 4; q[1] = -6; q[2] = 4; q[3] = -1; }
		wide_sum = r33;                    // (fixed.c:571-667: the 32-bit and the 64-bit restoration agree in their low 32 bits; 33-bit channels sum in 64)
	}
	const uint32_t method = br_get(b, 2);
	if(br_over(b)) return SD_EOS;
	if(method > 1) SD_FAIL(SD_UNPARSEABLE);                                // :3117-3120, :3225-3228
	const uint32_t plen = method ? 5u : 4u, esc = method ? 31u : 15u;
	const uint32_t po = br_get(b, 4);
	if(br_over(b)) return SD_EOS;
	if((n >> po) < order || (n & ((1u << po) - 1u)) != 0) SD_FAIL(SD_LOST_SYNC);       // :3107-3112
	const uint32_t psize = n >> po;
	const bool narrow24 = !EXACT && !wide_sum;             // (taps have at most 15 bits; the history is watched: `wild`)
	uint32_t next_part = order, k = 0, raw = 0, part = 0, ricebad = 0;
	uint64_t part_pos = 0, bad_pos = 0;                    // where the partition's codes begin; the same for the partition with the bad code
	bool escaped = false, err = false;
	for(uint32_t i = order; i < n && !err; ) {
#pragma unroll
		for(int s = 0; s < MAXORD; s++) {
			if(i < n && !err) {
				while(i == next_part && !err) {
					k = br_get(b, plen);
					escaped = k == esc;
					if(escaped) raw = br_get(b, 5);
					part_pos = br_pos(b);
					part++;
					next_part = po ? part * psize : n;
					err = br_over(b) || b.bad || ricebad || (narrow24 && wild);    // (per partition, not per sample)
				}
				int64_t r;
				if(escaped) r = raw ? (int64_t)br_get_signed(b, raw) : 0;
				else {
					const uint32_t u = sd_rice(b, k, ricebad);
					if(ricebad) {                                         // (the reference gives up at this code: nothing behind it counts)
						bad_pos = part_pos; err = true;
						if(b.bad || br_over(b)) ricebad = 0;              // ... unless the stream ended inside the code: END_OF_STREAM
					}
					r = (int64_t)(int32_t)((u >> 1) ^ (0u - (u & 1u)));
				}
				ST hs[MAXORD];
#pragma unroll
				for(int j = 0; j < MAXORD; j++) hs[j] = j < s ? h[s - 1 - j] : h[MAXORD - 1 - j + s];
				int64_t v;
				if(wide_sum) {
					int64_t sum = 0;
#pragma unroll
					for(int j = 0; j < MAXORD; j++) sum += (int64_t)q[j] * (int64_t)hs[j];
					v = r + (sum >> shift);                                 // lpc.c:1267 (then cut to 32 bits), :1522
				}
				else if(narrow24) {
					int32_t hh[MAXORD];
#pragma unroll
					for(int j = 0; j < MAXORD; j++) hh[j] = (int32_t)hs[j];
					v = (int64_t)(int32_t)((uint32_t)(int32_t)r + (uint32_t)((int32_t)FLACGPU_DOT24(q, hh) >> shift));
				}
				else {
					uint32_t s32 = 0;
#pragma unroll
					for(int j = 0; j < MAXORD; j++) s32 += (uint32_t)q[j] * (uint32_t)(int32_t)hs[j];
					v = (int64_t)(int32_t)((uint32_t)(int32_t)r + (uint32_t)((int32_t)s32 >> shift));     // lpc.c:1014
				}
				if(!r33) v = (int64_t)(int32_t)v;
				emit(i, v);
				wild |= (uint32_t)((uint32_t)(int32_t)v + 0x800000u) >> 24;
				h[s] = (ST)v;
				i++;
			}
		}
	}
	if(!ricebad && (b.bad || br_over(b))) return SD_EOS;
	// (the reference's Rice reader keeps its position in locals and leaves the bit reader where the partition's codes began when it
	//  gives up on a code, bitreader_read_rice_signed_block.c: "if(x > limit) return false")
	if(ricebad) { *fail_pos = bad_pos | (1ull << 63); return SD_LOST_SYNC; }   // :3326-3332 (bit 63: this was the reason)
	if(narrow24 && wild) return SD_RETRY;
	return SD_OK;
#undef SD_FAIL
}

// One frame body from a candidate whose header holds.  p: the frame's first byte (the 0xFF); avail: bytes from there to the end of
// the stream.  On SD_OK *len = bytes of the frame including its CRC-16 (which is NOT checked here: a span-parallel kernel does).
// SINK(ch, i, value).
template <int MAXORD, bool EXACT, typename ST, class SINK>
FLACGPU_HD inline int sd_decode_frame(const uint8_t *p, uint64_t avail, const uint8_t *buf_hi, const StreamCand &K, SINK &sink, uint32_t *len, uint32_t *pad_error)
{
	*pad_error = 0;
	BitReader b;
	br_init(b, p, (size_t)avail, buf_hi);
	for(uint32_t k = 0; k < K.hdr_len; k += 4) (void)br_get(b, (K.hdr_len - k >= 4 ? 4u : K.hdr_len - k) * 8);     // (the header: parsed by the scan)
	const uint32_t n = K.blocksize;
	for(uint32_t ch = 0; ch < K.channels; ch++) {
		auto s1 = [&](uint32_t i, int64_t v) { sink(ch, i, v); };
		uint64_t fail_pos = 0;
		const int st = sd_decode_subframe<MAXORD, EXACT, ST>(b, n, sd_nominal_bps(K.bps, K.ca, ch), s1, &fail_pos);
		if(st == SD_LOST_SYNC || st == SD_UNPARSEABLE) {
			// read_zero_padding_ runs all the same, from where the reader stood (fail_pos <= the stream's end here)
			// (bit 1: a Rice code longer than a 32-bit residual allows was the reason.  The reference applies that limit only to codes that
			//  do not straddle a refill of its 8 KiB read buffer -- the one place where its verdict depends on more than the stream's bytes;
			//  this decoder applies it always and says so, so that a caller, or a test, can tell)
			if(fail_pos >> 63) *pad_error |= 2;
			fail_pos &= ~(1ull << 63);
			const uint32_t rem = (uint32_t)(fail_pos & 7);
			if(rem && (p[fail_pos >> 3] & (0xffu >> rem)) != 0) *pad_error |= 1;
		}
		if(st != SD_OK) return st;
	}
	const uint32_t rem = (uint32_t)(br_pos(b) & 7);
	if(rem) {
		const uint32_t z = br_get(b, 8 - rem);
		if(b.bad || br_over(b)) return SD_EOS;
		if(z != 0) return SD_LOST_SYNC;                                    // :3369-3372
	}
	const uint64_t bytes = (br_pos(b) >> 3) + 2;
	if(bytes > avail) return SD_EOS;                                       // (the footer is cut off: :2432-2436, no error of its own)
	*len = (uint32_t)bytes;
	return SD_OK;
}

// the inter-channel step on one inter-channel sample of a stereo frame (undo_channel_coding, stream_decoder.c:3476-3526): c0, c1 as
// the subframes left them; bps < 32: everything in 32-bit wrap-around arithmetic; bps == 32: the side channel has 33 bits
FLACGPU_HD inline void sd_undo_channels(uint32_t ca, uint32_t bps, int64_t c0, int64_t c1, int32_t &o0, int32_t &o1)
{
	if(bps < 32) {
		const uint32_t a = (uint32_t)(int32_t)c0, s = (uint32_t)(int32_t)c1;
		if(ca == 1) { o0 = (int32_t)a; o1 = (int32_t)(a - s); }
		else if(ca == 2) { o0 = (int32_t)(a + s); o1 = (int32_t)s; }
		else if(ca == 3) {
			const uint32_t mid = (a << 1) | (s & 1u);
			o0 = (int32_t)(mid + s) >> 1; o1 = (int32_t)(mid - s) >> 1;
		}
		else { o0 = (int32_t)a; o1 = (int32_t)s; }
	}
	else {
		if(ca == 1) { o0 = (int32_t)c0; o1 = (int32_t)((int64_t)(int32_t)c0 - c1); }
		else if(ca == 2) { o1 = (int32_t)c1; o0 = (int32_t)((int64_t)(int32_t)c1 + c0); }
		else if(ca == 3) {
			const int64_t mid = (int64_t)(((uint64_t)(int64_t)(int32_t)c0 << 1) | ((uint64_t)c1 & 1u));
			o0 = (int32_t)((mid + c1) >> 1); o1 = (int32_t)((mid - c1) >> 1);
		}
		else { o0 = (int32_t)c0; o1 = (int32_t)c1; }
	}
}
FLACGPU_HD inline bool sd_out_of_bounds(int32_t v, uint32_t bps)
{
	const int sh = 32 - (int)bps;
	return v < (INT32_MIN >> sh) || v > (INT32_MAX >> sh);
}

} // namespace flacgpu
#endif
