// oracle/stream_decode_pin.cpp -- TEST INFRASTRUCTURE ONLY.
// The product's stream decoder for FLAC streams it did not write (flac_amd/csrc/flacgpu_stream_decode.h: what a lane of the GPU
// kernels runs; flacgpu_stream_walk.h: the host pass over their table) compiled for the host and driven the way
// flacgpu_stream_decode.hip drives it -- every byte position looked at for a sync code, every candidate whose header holds decoded
// (first by the instance that keeps 12 taps and uses the 24-bit multiplier, again by the exact one when that says so), CRC-16,
// inter-channel step, bounds, then the walk -- so that tests/test_stream_decode_cpu.py can hold its logic to the reference's
// decoder (oracle/_ref/libFLAC_ref.so: ref_decode_stream) without a GPU.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../flac_amd/csrc/flacgpu_stream_walk.h"

using namespace flacgpu;

template <typename ST>
static void decode_all(const uint8_t *s, uint64_t n, std::vector<StreamCand> &cand, std::vector<StreamBody> &body, std::vector<std::vector<int32_t>> &pcm, uint32_t *retries)
{
	for(size_t i = 0; i < cand.size(); i++) {
		const StreamCand &K = cand[i];
		StreamBody &B = body[i];
		B.spec = ~0ull; B.len = 0; B.bstat = SD_NOT_DECODED; B.oob_mask = 0; B.wrote = 0; B.pad_error = 0;
		if(K.hstat != SD_OK) continue;
		std::vector<ST> coded((size_t)K.channels * K.blocksize);
		auto sink = [&](uint32_t ch, uint32_t j, int64_t v) { coded[(size_t)ch * K.blocksize + j] = (ST)v; };
		uint32_t len = 0, pad_error = 0;
		int st = sd_decode_frame<12, false, ST>(s + K.pos, n - K.pos, s + n, K, sink, &len, &pad_error);
		if(st == SD_RETRY) { (*retries)++; st = sd_decode_frame<32, true, ST>(s + K.pos, n - K.pos, s + n, K, sink, &len, &pad_error); }
		B.bstat = (uint8_t)st; B.pad_error = (uint8_t)pad_error;
		if(st != SD_OK) continue;
		B.len = len;
		if(dec_crc16(s + K.pos, len - 2) != (((uint32_t)s[K.pos + len - 2] << 8) | s[K.pos + len - 1])) { B.bstat = SD_CRC_MISMATCH; continue; }
		std::vector<int32_t> &out = pcm[i];
		out.resize((size_t)K.channels * K.blocksize);
		for(uint32_t j = 0; j < K.blocksize; j++) {
			if(K.channels == 2) {
				int32_t o0, o1;
				sd_undo_channels(K.ca, K.bps, (int64_t)coded[j], (int64_t)coded[K.blocksize + j], o0, o1);
				out[(size_t)j * 2] = o0; out[(size_t)j * 2 + 1] = o1;
			}
			else for(uint32_t ch = 0; ch < K.channels; ch++) out[(size_t)j * K.channels + ch] = (int32_t)coded[(size_t)ch * K.blocksize + j];
			for(uint32_t ch = 0; ch < K.channels; ch++) if(sd_out_of_bounds(out[(size_t)j * K.channels + ch], K.bps)) B.oob_mask |= (uint8_t)(1u << ch);
		}
		if(B.oob_mask) B.bstat = SD_OUT_OF_BOUNDS;
	}
}

// info: has_streaminfo, min_blocksize, max_blocksize, sample_rate, channels, bps; result: samples, frames, silence_samples, nevents,
// end_in_header, format_changes, channels, bps, candidates, retries
extern "C" int sdpin_decode(const uint8_t *stream, uint64_t nbytes, uint64_t first_pos, const uint32_t *info, int32_t *pcm, uint64_t cap_samples,
                            uint32_t *ev_status, uint64_t *ev_pos, uint32_t max_events, uint64_t *result)
{
	// the lane code reads the aligned words its bytes lie in: give it an aligned copy that ends on a word boundary
	const uint64_t padded = (nbytes + 3) & ~3ull;
	uint8_t *s = (uint8_t *)aligned_alloc(64, (size_t)((padded + 63) & ~63ull) + 64);
	memset(s, 0, (size_t)((padded + 63) & ~63ull) + 64);
	memcpy(s, stream, (size_t)nbytes);
	SdInfo I = {info[0], info[1], info[2], info[3], info[4], info[5]};
	std::vector<StreamCand> cand;
	bool wide = false;
	auto get = [&](uint64_t i) -> uint32_t { return s[i]; };
	for(uint64_t c = first_pos; c + 1 < nbytes; c++) {
		if(!sd_is_sync(s[c], s[c + 1])) continue;
		StreamCand K;
		sd_parse_candidate(get, nbytes, c, I, K);
		if(K.hstat == SD_OK && K.bps == 32 && K.ca != 0) wide = true;
		cand.push_back(K);
	}
	std::vector<StreamBody> body(cand.size());
	std::vector<std::vector<int32_t>> fpcm(cand.size());
	uint32_t retries = 0;
	if(wide) decode_all<int64_t>(s, nbytes, cand, body, fpcm, &retries);
	else decode_all<int32_t>(s, nbytes, cand, body, fpcm, &retries);
	WalkResult W;
	sd_walk(I, first_pos, nbytes, nbytes ? s[nbytes - 1] : 0, cand.data(), body.data(), cand.size(), W);
	const uint32_t C = W.channels;
	int rc = 0;
	if(W.samples > cap_samples) rc = -4;
	else for(const WalkPlace &P : W.places) {
		if(P.cand < 0) memset(pcm + P.out * C, 0, (size_t)P.n * C * 4);
		else memcpy(pcm + P.out * C, fpcm[(size_t)P.cand].data(), (size_t)P.n * C * 4);
	}
	for(size_t e = 0; e < W.events.size() && e < max_events; e++) { ev_status[e] = W.events[e].status; ev_pos[e] = W.events[e].pos; }
	result[0] = W.samples; result[1] = W.frames; result[2] = W.silence_samples; result[3] = W.events.size(); result[4] = W.end_in_header;
	result[5] = W.format_changes; result[6] = W.channels; result[7] = W.bps; result[8] = cand.size(); result[9] = retries; result[10] = W.long_rice_codes;
	free(s);
	return rc;
}

extern "C" int sdpin_probe(const uint8_t *stream, uint64_t nbytes, uint32_t *info, uint64_t *first_pos, uint64_t *total_samples, uint8_t *md5)
{
	SdInfo I;
	const bool ok = sd_probe_metadata(stream, nbytes, I, *first_pos, total_samples, md5);
	info[0] = I.has_streaminfo; info[1] = I.min_blocksize; info[2] = I.max_blocksize; info[3] = I.sample_rate; info[4] = I.channels; info[5] = I.bps;
	return ok ? 0 : -1;
}
