// flac_amd/csrc/flacgpu_stage.hip -- input staging on the device: raw interleaved sample bytes as they sit in a
// WAVE / AIFF / raw file  ->  the interleaved int32 block FLAC__stream_encoder_process_interleaved() takes.
// Restates format_input() of the reference's command-line tool (src/flac/encode.c:2352-2492): 8/16/24/32-bit containers,
// either byte order, signed or unsigned (unsigned: subtract the mid-point), optional channel permutation, and the
// "left-justified in a wider container" case (shift; low bits that are not zero are an input error, :2479-2488).
// At 16 bits this moves 2 bytes per sample over PCIe instead of 4.  One thread per 4 consecutive samples
// (16-byte stores); pure streaming: HBM bound (reads container bytes, writes 4 bytes per sample).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "flacgpu_dev.h"

namespace flacgpu {

__device__ __forceinline__ int32_t stage_one(const uint8_t *p, const StageParams &S)
{
	uint32_t t;
	if(S.bytes == 1) t = p[0];
	else if(S.bytes == 2) t = S.big_endian ? ((uint32_t)p[0] << 8) | p[1] : ((uint32_t)p[1] << 8) | p[0];
	else if(S.bytes == 3) t = S.big_endian ? ((uint32_t)p[0] << 16) | ((uint32_t)p[1] << 8) | p[2] : ((uint32_t)p[2] << 16) | ((uint32_t)p[1] << 8) | p[0];
	else t = S.big_endian ? ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]
	                      : ((uint32_t)p[3] << 24) | ((uint32_t)p[2] << 16) | ((uint32_t)p[1] << 8) | p[0];
	const uint32_t bits = 8 * S.bytes;
	if(S.is_unsigned) return (int32_t)(t - (1u << (bits - 1)));            // encode.c: - 0x80 / 0x8000 / 0x800000 / 0x80000000
	return bits == 32 ? (int32_t)t : (int32_t)(t << (32 - bits)) >> (32 - bits);
}

__global__ __launch_bounds__(256) void stage_raw_kernel(const StageParams S, const uint8_t *__restrict__ raw, uint64_t nvalues,
                                                         int32_t *__restrict__ pcm, uint32_t *__restrict__ err)
{
	const uint64_t i0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 4;
	if(i0 >= nvalues) return;
	const uint32_t C = S.channels, mask = (1u << S.shift) - 1u;
	int32_t v[4];
	uint32_t bad = 0;
#pragma unroll
	for(int k = 0; k < 4; k++) {
		const uint64_t i = i0 + (uint64_t)k;
		v[k] = 0;
		if(i < nvalues) {
			// value i of the OUTPUT (wide sample i / C, output channel i % C) comes from the input channel mapped onto it
			uint64_t src = i;
			if(S.use_map) {
				const uint64_t ws = i / C;
				const uint32_t oc = (uint32_t)(i - ws * C);
				uint32_t ic = oc;
				for(uint32_t c = 0; c < C; c++) if(S.map[c] == oc) ic = c;
				src = ws * C + ic;
			}
			int32_t x = stage_one(raw + src * S.bytes, S);
			if(S.shift) { bad |= (uint32_t)x & mask; x >>= S.shift; }
			v[k] = x;
		}
	}
	if(i0 + 4 <= nvalues) *(int4 *)(pcm + i0) = make_int4(v[0], v[1], v[2], v[3]);
	else for(int k = 0; k < 4; k++) if(i0 + (uint64_t)k < nvalues) pcm[i0 + k] = v[k];
	if(bad && err) atomicOr(err, 1u);
}

hipError_t launch_stage_raw(const StageParams &S, const void *d_raw, uint64_t nvalues, int32_t *d_pcm, uint32_t *d_err, hipStream_t s)
{
	if(nvalues == 0) return hipSuccess;
	const uint64_t threads = (nvalues + 3) / 4;
	hipLaunchKernelGGL(stage_raw_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, S, (const uint8_t *)d_raw, nvalues, d_pcm, d_err);
	return hipGetLastError();
}

} // namespace flacgpu
