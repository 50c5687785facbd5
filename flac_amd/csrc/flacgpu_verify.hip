// flac_amd/csrc/flacgpu_verify.hip -- the encoder's self check on the device (SURVEY.md 8f row 3;
// FLAC__stream_encoder_set_verify, src/libFLAC/stream_encoder.c:3000-3018, 5155-5230): every frame of a batch is decoded
// again and compared with the samples that went in, without leaving HBM.
//
//   verify_kernel    ONE LANE PER FRAME (flacgpu_decode.h): a wavefront walks 64 frames in lockstep through header, subframes,
//                    Rice codes and predictor restoration; each decoded coded-channel sample is compared on the spot with the
//                    value the input implies for it, nothing is stored.  Rice decoding is a serial bit-dependency chain, so a
//                    lane runs at instruction-issue latency; what is parallel is the batch: 16384 frames = 256 wavefronts, one
//                    per CU.  The bit window and the expected sample are fetched one step ahead (no second wavefront on the
//                    SIMD to hide a load behind).
//   (crc_check_kernel, flacgpu_kernels.hip: the CRC-16 footers, one wavefront per frame, spans in parallel.)
//   verify_detail_kernel  one lane, only when a frame failed: decodes the FIRST bad frame of the batch into a scratch buffer,
//                    undoes the inter-channel decorrelation and reports {frame, channel, sample, expected, got} of the first
//                    differing output sample, or "does not decode" -- what get_verify_decoder_error_stats returns.
// Integer / bit work, latency bound by construction; no MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "flacgpu.h"
#include "flacgpu_dev.h"
#define FLACGPU_HD __device__
#include "flacgpu_decode.h"

namespace flacgpu {

template <int MAXORD, typename ST>
__global__ __launch_bounds__(64) void verify_kernel(const DevParams P, const uint8_t *__restrict__ frames, const uint32_t *__restrict__ frame_bytes,
                                                    const uint64_t *__restrict__ offsets, uint32_t nframes, uint32_t tail_n, uint64_t first_frame_number,
                                                    const int32_t *__restrict__ pcm, VerifyState *__restrict__ state)
{
	const uint32_t f = blockIdx.x * 64u + threadIdx.x;
	if(f >= nframes) return;
	const uint32_t fb = frame_bytes[f];
	const uint32_t C = P.channels, N = P.blocksize;
	DecodeExpect E;
	E.channels = C; E.bps = P.bps; E.blocksize = N; E.n = (tail_n && f + 1 == nframes) ? tail_n : N; E.frame_number = first_frame_number + f;
	int st = DEC_ERROR;
	if(fb != 0xffffffffu) {
		const uint8_t *p = frames + offsets[f];
		const uint8_t *hi = frames + offsets[nframes];
		const int32_t *x = pcm + (size_t)f * N * C;
		st = DEC_OK;
		if(fb < 6) st = DEC_ERROR;
		else {
			BitReader b;
			br_init(b, p, fb - 2, hi);
			FrameHead H;
			if(decode_frame_header(b, p, E, H) != DEC_OK) st = DEC_ERROR;
			else {
				uint32_t differ = 0;
				for(uint32_t ch = 0; ch < C && st == DEC_OK; ch++) {
					const uint32_t ca = H.ca;
					// the expectation of sample i is fetched while sample i-1 is being decoded
					int64_t e_next = coded_expectation(x, ca, ch);
					auto sink = [&](uint32_t i, int64_t v) {
						differ |= (uint32_t)(v != e_next);
						const uint32_t j = i + 1 < H.n ? i + 1 : i;
						e_next = coded_expectation(x + (size_t)j * C, ca, ch);
					};
					if(decode_subframe<MAXORD, ST>(b, H.n, coded_bps(E.bps, ca, ch), sink) != DEC_OK) st = DEC_ERROR;
				}
				if(st == DEC_OK && decode_frame_tail(b) != DEC_OK) st = DEC_ERROR;
				if(st == DEC_OK && differ) st = DEC_MISMATCH;
			}
		}
	}
	if(st != DEC_OK) atomicMin(&state->first_bad, f);
}

template <int MAXORD, typename ST>
__global__ __launch_bounds__(64) void verify_detail_kernel(const DevParams P, const uint8_t *__restrict__ frames, const uint32_t *__restrict__ frame_bytes,
                                                           const uint64_t *__restrict__ offsets, uint32_t nframes, uint32_t tail_n, uint64_t first_frame_number,
                                                           const int32_t *__restrict__ pcm, int64_t *__restrict__ scratch, VerifyState *__restrict__ state,
                                                           flacgpu_verify_result *__restrict__ result)
{
	if(threadIdx.x != 0 || blockIdx.x != 0) return;
	flacgpu_verify_result R;
	R.status = 0; R.frame_number = 0; R.channel = 0; R.sample = 0; R.absolute_sample = 0; R.expected = 0; R.got = 0;
	const uint32_t f = state->first_bad;
	if(f < nframes) {
		const uint32_t C = P.channels, N = P.blocksize;
		DecodeExpect E;
		E.channels = C; E.bps = P.bps; E.blocksize = N; E.n = (tail_n && f + 1 == nframes) ? tail_n : N; E.frame_number = first_frame_number + f;
		DecodeDetail D;
		D.status = DEC_ERROR; D.channel = 0; D.sample = 0; D.expected = 0; D.got = 0;
		const uint32_t fb = frame_bytes[f];
		if(fb != 0xffffffffu) verify_frame_detail<MAXORD, ST>(frames + offsets[f], fb, frames + offsets[nframes], E, pcm + (size_t)f * N * C, scratch, N, D);
		// a frame the batch pass (or the CRC pass) rejected but the detail pass accepts cannot be: report it as undecodable
		R.status = D.status == DEC_OK ? DEC_ERROR : D.status;
		R.frame_number = (uint32_t)(first_frame_number + f);
		R.channel = D.channel; R.sample = D.sample;
		R.absolute_sample = (first_frame_number + f) * (uint64_t)N + D.sample;      // stream_encoder.c:5186-5196
		R.expected = D.expected; R.got = D.got;
	}
	*result = R;
}

__global__ void verify_reset_kernel(VerifyState *state) { state->first_bad = 0xffffffffu; }

template <int MAXORD, typename ST>
static hipError_t launch_verify_t(const DevParams &P, const uint8_t *frames, const uint32_t *fb, const uint64_t *offsets, uint32_t nframes, uint32_t tail_n,
                                  uint64_t first, const int32_t *pcm, int64_t *scratch, VerifyState *state, flacgpu_verify_result *result, hipStream_t s)
{
	hipLaunchKernelGGL((verify_kernel<MAXORD, ST>), dim3((nframes + 63) / 64), dim3(64), 0, s, P, frames, fb, offsets, nframes, tail_n, first, pcm, state);
	hipLaunchKernelGGL((verify_detail_kernel<MAXORD, ST>), dim3(1), dim3(64), 0, s, P, frames, fb, offsets, nframes, tail_n, first, pcm, scratch, state, result);
	return hipGetLastError();
}

hipError_t launch_verify(const DevParams &P, const uint8_t *frames, const uint32_t *fb, const uint64_t *offsets, uint32_t nframes, uint32_t tail_n,
                         uint64_t first, const int32_t *pcm, int64_t *scratch, VerifyState *state, flacgpu_verify_result *result, hipStream_t s)
{
	hipLaunchKernelGGL(verify_reset_kernel, dim3(1), dim3(1), 0, s, state);
	hipError_t e = launch_crc_check(frames, fb, offsets, nframes, state, s);
	if(e != hipSuccess) return e;
	const bool wide = P.bps == 32 && P.channels == 2;            // a 33-bit side channel can occur (stream_encoder.c:3831-3835)
	const uint32_t m = P.max_lpc_order;
#define GO(M) (wide ? launch_verify_t<M, int64_t>(P, frames, fb, offsets, nframes, tail_n, first, pcm, scratch, state, result, s) \
                    : launch_verify_t<M, int32_t>(P, frames, fb, offsets, nframes, tail_n, first, pcm, scratch, state, result, s))
	if(m <= 8) e = GO(8);
	else if(m <= 12) e = GO(12);
	else e = GO(32);
#undef GO
	sync_debug("verify", s);
	return e;
}

} // namespace flacgpu
