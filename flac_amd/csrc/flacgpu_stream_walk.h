// flac_amd/csrc/flacgpu_stream_walk.h -- the sequential remainder of the reference's stream decoder, as host code over the table
// the GPU passes leave behind (flacgpu_stream_decode.h): which sync code the search reaches after which, which errors the client
// hears in which order, where every frame's samples go and where silence stands in for frames that are missing.
// One pass, a few nanoseconds per frame (the reference's FLAC__stream_decoder_process_until_end_of_stream loop,
// src/libFLAC/stream_decoder.c:1168-1201, with everything per-byte and per-sample taken out):
//   search       frame_sync_ :2321 -- the next sync code at or behind the search position; skipped bytes are one LOST_SYNC (:2369-2372),
//                also when the stream ends while searching
//   header       read_frame_header_ :2624 -- a header that does not hold is its own error and says where the search goes on
//                (no rewind: read_frame_ :2393-2394 returns at once)
//   body         read_frame_ :2395-2483 -- a body that does not decode, a CRC-16 that does not match, a sample out of bounds: the
//                error, and the search goes on right behind the sync code (:2558-2587; the rewind needs the tell / seek callbacks
//                or the sync code still in the reader's buffer -- a file or a memory stream has them)
//   numbering    :2917-2934 -- frame numbers become sample numbers through the stream's fixed block size, known from STREAMINFO or
//                from the first good frame
//   missing      :2485-2554 -- a good frame that starts later than the last one ended: MISSING_FRAME unless an error was already
//                sent for the gap, and silence of at most 5 s / 50 blocks when the two frames agree on the format
// Host only, no HIP: shared by libflacgpu.so and the CPU pin harness of the tests (oracle/stream_decode_pin.cpp).
#ifndef FLACGPU_STREAM_WALK_H
#define FLACGPU_STREAM_WALK_H
#include <stdint.h>
#include <vector>
#include "flacgpu_stream_decode.h"

namespace flacgpu {

struct WalkEvent { uint32_t status; uint64_t pos; };                       // status: SD_* (= FLAC__StreamDecoderErrorStatus + 1); pos: the sync code (or search start) it belongs to
struct WalkPlace { int64_t cand; uint64_t out, n; uint64_t sample_number; };   // cand >= 0: that candidate's frame goes to out .. out + n; -1: n samples of silence
struct WalkResult {
	std::vector<WalkEvent> events;
	std::vector<WalkPlace> places;
	uint64_t samples = 0;          // inter-channel samples the client receives
	uint64_t frames = 0, silence_samples = 0;
	uint32_t end_in_header = 0;    // the stream ended inside a frame header: process_until_end_of_stream returns false (:1186-1187)
	uint32_t long_rice_codes = 0;  // frames the search reached that were given up for a Rice code longer than a 32-bit residual allows (flacgpu_stream_decode.h)
	uint32_t need_deferred = 0;    // the search reached a candidate whose frame has not been decoded yet (SD_DEFERRED): decode those, walk again
	uint32_t format_changes = 0;   // good frames whose channel count or sample width is not the stream's (they get no room in the output)
	uint32_t channels = 0, bps = 0, sample_rate = 0;   // the stream's format: STREAMINFO, else the first good frame
};

// last_byte: the stream's final byte (decides whether a search that runs into the end has skipped anything, frame_sync_ :2339-2372)
inline void sd_walk(const SdInfo &I, uint64_t first_pos, uint64_t nbytes, uint32_t last_byte, const StreamCand *cand, const StreamBody *body, size_t ncand, WalkResult &R)
{
	uint64_t pos = first_pos;
	size_t ci = 0;
	uint32_t fixed_bs = 0;
	bool error_sent = false;
	bool last_set = false;
	uint64_t last_sn = 0; uint32_t last_bs = 0, last_sr = 0, last_ch = 0, last_bps = 0;
	uint64_t out = 0;
	R.channels = I.has_streaminfo ? I.channels : 0; R.bps = I.has_streaminfo ? I.bps : 0; R.sample_rate = I.has_streaminfo ? I.sample_rate : 0;
	auto send = [&](uint32_t st, uint64_t p) { R.events.push_back(WalkEvent{st, p}); error_sent = true; };
	for(;;) {
		while(ci < ncand && cand[ci].pos < pos) ci++;
		if(ci == ncand) {
			// the search runs into the end of the stream: an error if it got past at least one byte that is no sync code
			const uint64_t r = nbytes > pos ? nbytes - pos : 0;
			if(r >= 2 || (r == 1 && last_byte != 0xffu)) send(SD_LOST_SYNC, pos);
			break;
		}
		const StreamCand &K = cand[ci];
		const StreamBody &B = body[ci];
		if(K.pos > pos) send(SD_LOST_SYNC, pos);
		if(K.hstat == SD_EOS) { R.end_in_header = 1; break; }
		if(K.hstat != SD_OK) { send(K.hstat, K.pos); pos = K.pos + K.resume; continue; }
		if(B.bstat == SD_DEFERRED) { R.need_deferred = 1; return; }
		// the header holds: its number becomes a sample number (:2917-2934)
		uint32_t next_fixed = 0;
		uint64_t sn;
		if(K.variable) sn = K.number;
		else if(fixed_bs) sn = (uint64_t)fixed_bs * K.number;
		else if(I.has_streaminfo) { sn = (uint64_t)I.min_blocksize * K.number; next_fixed = I.max_blocksize; }
		else if(K.number == 0) { sn = 0; next_fixed = K.blocksize; }
		else sn = (uint64_t)K.blocksize * K.number;
		bool good = false;
		if(B.bstat == SD_OK) good = true;
		else if(B.bstat == SD_OUT_OF_BOUNDS) { for(uint32_t ch = 0; ch < K.channels; ch++) if(B.oob_mask & (1u << ch)) send(SD_OUT_OF_BOUNDS, K.pos); }
		else if(B.bstat == SD_EOS) { /* the stream ends inside the frame: no error of its own (:2417-2420, :2432-2436) */ }
		else { send(B.bstat, K.pos); if(B.pad_error & 1) send(SD_LOST_SYNC, K.pos); if(B.pad_error & 2) R.long_rice_codes++; }
		if(good) {
			// frames missing in front of this one (:2485-2554)
			if(last_set && last_sn + last_bs < sn) {
				uint64_t need = sn - (last_sn + last_bs);
				if(!error_sent) send(SD_MISSING_FRAME, K.pos);
				if(last_sr == K.sample_rate && last_ch == K.channels && last_bps == K.bps && last_bs >= 16) {
					if(need > 5ull * last_sr) need = 5ull * last_sr;
					if(need > 50ull * last_bs) need = 50ull * last_bs;
					if(need) {
						const bool fits = last_ch == R.channels && last_bps == R.bps;
						if(fits) { R.places.push_back(WalkPlace{-1, out, need, last_sn + last_bs}); out += need; R.silence_samples += need; }
					}
				}
			}
			error_sent = false;
			if(next_fixed) fixed_bs = next_fixed;
			if(!R.channels) { R.channels = K.channels; R.bps = K.bps; R.sample_rate = K.sample_rate; }
			if(K.channels == R.channels && K.bps == R.bps) { R.places.push_back(WalkPlace{(int64_t)ci, out, K.blocksize, sn}); out += K.blocksize; R.frames++; }
			else R.format_changes++;
			last_set = true; last_sn = sn; last_bs = K.blocksize; last_sr = K.sample_rate; last_ch = K.channels; last_bps = K.bps;
			pos = K.pos + B.len;
		}
		else {
			error_sent = false;
			pos = K.pos + 2;
		}
	}
	R.samples = out;
}

// Where the audio frames of a FLAC file begin and what its STREAMINFO says (format: "fLaC", then metadata blocks of a 4-byte header
// each -- last flag, type, 24-bit length -- the first of which is STREAMINFO, src/libFLAC/stream_decoder.c:1654-1717, :1719-1990;
// ID3v2 tags in front are stepped over, :2296-2319).  A stream that does not start that way is taken as bare frames from byte 0.
// Returns false when the metadata runs past the end of the buffer.  md5 (16 bytes) and total_samples may be null.
inline bool sd_probe_metadata(const uint8_t *s, uint64_t n, SdInfo &I, uint64_t &first_pos, uint64_t *total_samples, uint8_t *md5)
{
	I.has_streaminfo = 0; I.min_blocksize = I.max_blocksize = I.sample_rate = I.channels = I.bps = 0;
	first_pos = 0;
	if(total_samples) *total_samples = 0;
	uint64_t p = 0;
	while(p + 10 <= n && s[p] == 'I' && s[p + 1] == 'D' && s[p + 2] == '3') {
		const uint64_t skip = ((uint64_t)(s[p + 6] & 0x7f) << 21) | ((uint64_t)(s[p + 7] & 0x7f) << 14) | ((uint64_t)(s[p + 8] & 0x7f) << 7) | (s[p + 9] & 0x7f);
		p += 10 + skip;
	}
	if(p + 4 > n || s[p] != 'f' || s[p + 1] != 'L' || s[p + 2] != 'a' || s[p + 3] != 'C') return true;
	p += 4;
	for(;;) {
		if(p + 4 > n) return false;
		const bool last = (s[p] & 0x80) != 0;
		const uint32_t type = s[p] & 0x7f;
		const uint64_t len = ((uint64_t)s[p + 1] << 16) | ((uint64_t)s[p + 2] << 8) | s[p + 3];
		p += 4;
		if(p + len > n) return false;
		if(type == 0 && len >= 34 && !I.has_streaminfo) {
			const uint8_t *q = s + p;
			I.has_streaminfo = 1;
			I.min_blocksize = ((uint32_t)q[0] << 8) | q[1];
			I.max_blocksize = ((uint32_t)q[2] << 8) | q[3];
			I.sample_rate = ((uint32_t)q[10] << 12) | ((uint32_t)q[11] << 4) | (q[12] >> 4);
			I.channels = ((q[12] >> 1) & 7u) + 1;
			I.bps = (((uint32_t)(q[12] & 1u) << 4) | (q[13] >> 4)) + 1;
			if(total_samples) *total_samples = ((uint64_t)(q[13] & 15u) << 32) | ((uint64_t)q[14] << 24) | ((uint64_t)q[15] << 16) | ((uint64_t)q[16] << 8) | q[17];
			if(md5) for(int k = 0; k < 16; k++) md5[k] = q[18 + k];
		}
		p += len;
		if(last) break;
	}
	first_pos = p;
	return true;
}

} // namespace flacgpu
#endif
