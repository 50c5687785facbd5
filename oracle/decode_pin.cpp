// oracle/decode_pin.cpp -- TEST INFRASTRUCTURE ONLY.
// The product's frame decoder (flac_amd/csrc/flacgpu_decode.h: the code every lane of the GPU verify kernels runs)
// compiled for the host and driven the way flacgpu_verify.hip drives it: the fast pass over every frame, the first
// frame that is not OK, the detail pass on that frame.  tests/test_decode_pin.py checks it against the host frame
// decoder (host/verify.c) and against injected damage, so that the decoder's logic is pinned without a GPU.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../flac_amd/csrc/flacgpu_decode.h"
#include "../flac_amd/csrc/flacgpu_decode_hinted.h"

using namespace flacgpu;

struct pin_result { int32_t status; uint32_t frame_number, channel, sample; uint64_t absolute_sample; int32_t expected, got; };

template <int MAXORD, typename ST>
static int run(const uint8_t *frames, const uint32_t *fb, uint32_t nframes, uint32_t C, uint32_t bps, uint32_t N, uint32_t tail, uint64_t first,
               const int32_t *pcm, size_t total_bytes, pin_result *out)
{
	size_t off = 0;
	memset(out, 0, sizeof *out);
	for(uint32_t f = 0; f < nframes; f++) {
		DecodeExpect E = {C, bps, N, (f + 1 == nframes && tail) ? tail : N, first + f};
		const int32_t *fp = pcm + (size_t)f * N * C;
		int st = verify_frame_fast<MAXORD, ST>(frames + off, fb[f], frames + total_bytes, E, fp);
		// (on the GPU the footers are checked by their own kernel; a frame it flags is a frame that does not decode)
		if(fb[f] >= 6 && dec_crc16(frames + off, fb[f] - 2) != (((uint32_t)frames[off + fb[f] - 2] << 8) | frames[off + fb[f] - 1])) st = DEC_ERROR;
		if(st != DEC_OK) {
			int64_t *x = (int64_t *)malloc(sizeof(int64_t) * (size_t)C * N);
			DecodeDetail D;
			verify_frame_detail<MAXORD, ST>(frames + off, fb[f], frames + total_bytes, E, fp, x, N, D);
			free(x);
			out->status = D.status; out->frame_number = (uint32_t)(first + f); out->channel = D.channel; out->sample = D.sample;
			out->absolute_sample = (first + f) * N + D.sample; out->expected = D.expected; out->got = D.got;
			// the two passes must agree on what kind of problem this is
			if(D.status != st) out->status = 100 + 10 * st + D.status;
			return out->status;
		}
		off += fb[f];
	}
	return 0;
}

extern "C" int decodepin_verify_batch(const uint8_t *frames, const uint32_t *fb, uint32_t nframes, uint32_t channels, uint32_t bps, uint32_t blocksize,
                                      uint32_t tail, uint64_t first_frame, const int32_t *pcm, uint32_t maxord, pin_result *out)
{
	size_t total = 0;
	for(uint32_t f = 0; f < nframes; f++) total += fb[f];
	const bool wide = bps == 32 && channels == 2;
#define GO(M) (wide ? run<M, int64_t>(frames, fb, nframes, channels, bps, blocksize, tail, first_frame, pcm, total, out) \
                    : run<M, int32_t>(frames, fb, nframes, channels, bps, blocksize, tail, first_frame, pcm, total, out))
	if(maxord <= 8) return GO(8);
	if(maxord <= 12) return GO(12);
	return GO(32);
#undef GO
}

// ---- the hinted pass (flacgpu_decode_hinted.h): per-frame verdicts 0 verified / 1 suspect, with the hints given -----------------
// hints: [nframes][channels][HINT_MAX_RUNS]
extern "C" int decodepin_make_hints(const uint8_t *frames, const uint32_t *fb, uint32_t nframes, uint32_t channels, uint32_t bps, uint32_t blocksize,
                                    uint32_t tail, uint64_t first_frame, uint32_t *hints, uint8_t *covered)
{
	size_t total = 0, off = 0;
	for(uint32_t f = 0; f < nframes; f++) total += fb[f];
	int n = 0;
	for(uint32_t f = 0; f < nframes; f++) {
		DecodeExpect E = {channels, bps, blocksize, (f + 1 == nframes && tail) ? tail : blocksize, first_frame + f};
		covered[f] = make_hints_host(frames + off, fb[f], frames + total, E, hints + (size_t)f * channels * HINT_MAX_RUNS) == 0;
		n += covered[f];
		off += fb[f];
	}
	return n;
}
extern "C" int decodepin_verify_hinted(const uint8_t *frames, const uint32_t *fb, uint32_t nframes, uint32_t channels, uint32_t bps, uint32_t blocksize,
                                       uint32_t tail, uint64_t first_frame, const int32_t *pcm, const uint32_t *hints, uint32_t maxord, uint8_t *suspect)
{
	size_t total = 0, off = 0;
	for(uint32_t f = 0; f < nframes; f++) total += fb[f];
	int n = 0;
	for(uint32_t f = 0; f < nframes; f++) {
		DecodeExpect E = {channels, bps, blocksize, (f + 1 == nframes && tail) ? tail : blocksize, first_frame + f};
		const int32_t *fp = pcm + (size_t)f * blocksize * channels;
		const uint32_t *h = hints + (size_t)f * channels * HINT_MAX_RUNS;
		int st;
		if(maxord <= 8) st = verify_frame_hinted_host<8>(frames + off, fb[f], frames + total, E, fp, h);
		else if(maxord <= 12) st = verify_frame_hinted_host<12>(frames + off, fb[f], frames + total, E, fp, h);
		else st = verify_frame_hinted_host<16>(frames + off, fb[f], frames + total, E, fp, h);
		suspect[f] = (uint8_t)st;
		n += st;
		off += fb[f];
	}
	return n;
}

// the two frame-header parsers (bit reader + bytes from memory; five words at once) on one frame: returns
// (status of decode_frame_header) | (status of hinted_frame_header << 8) | (1 << 16 when both accept but disagree on ca / n / position)
extern "C" int decodepin_header_both(const uint8_t *frame, uint32_t len, uint32_t lead, uint32_t channels, uint32_t bps, uint32_t blocksize, uint32_t n, uint64_t frame_number)
{
	// `frame` holds `lead` stray bytes in front of the frame (its alignment is the caller's choice) and slack behind it
	const uint8_t *p = frame + lead;
	DecodeExpect E = {channels, bps, blocksize, n, frame_number};
	BitReader b;
	br_init(b, p, len - 2, p + len + 16);
	FrameHead A, B;
	A.ca = A.n = B.ca = B.n = 0;
	const int sa = decode_frame_header(b, p, E, A);
	PeekSrc S;
	S.w0 = b.w0; S.nwords = (uint32_t)(b.wlast - b.w0) + 1; S.skip = b.skip; S.limit = (uint32_t)b.limit;
	uint32_t pos = 0;
	const int sb = hinted_frame_header(S, E, B, &pos);
	int r = sa | (sb << 8);
	if(sa == DEC_OK && sb == DEC_OK && (A.ca != B.ca || A.n != B.n || pos != (uint32_t)br_pos(b))) r |= 1 << 16;
	return r;
}
