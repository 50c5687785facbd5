// flac_amd/csrc/flacgpu_md5.hip -- MD5 of MANY streams' sample bytes on the device (md5.c:60-222 is RFC 1321's
// transform, stream_encoder.c:3448 / :3666-3686 what it is fed: the interleaved samples as little-endian bytes, one digest per
// stream in STREAMINFO).
//
// One chain is serial -- 64 dependent steps per 64 bytes -- so a single stream gains nothing from a GPU.  A CORPUS is
// many streams: one LANE per stream, each lane running RFC 1321 on its own bytes where they already lie in HBM (the staged input
// of the encode: no copy back to the host, no host threads).  A lane reads its 64-byte block as aligned words (four 16-byte loads
// and a word, shifted into place when the stream does not start on a word boundary), so any byte offset works; lanes whose streams
// are shorter idle at the end.  Throughput is the chain's latency times the number of streams: 30-70 MB/s per lane -- the ten-hour
// corpus as 1000 tracks hashes in 0.09 s (70 GB/s), as 120 tracks (two wavefronts on an otherwise idle chip) in 1.1-1.8 s, which
// four host threads of the AVX2 eight-chain routine beat (0.78 s): flac_amd/corpus.py picks by the number of tracks
// (profiles/archive/r04_s_md5_device.txt).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "flacgpu.h"

namespace flacgpu {

__device__ __forceinline__ uint32_t md5_rotl(uint32_t x, uint32_t s) { return __builtin_amdgcn_alignbit(x, x, 32u - s); }
// RFC 1321, section 3.4: the four auxiliary functions and one step  a = b + ((a + f(b,c,d) + X[k] + T[i]) <<< s)
#define MD5_F(x, y, z) (((x) & (y)) | (~(x) & (z)))
#define MD5_G(x, y, z) (((x) & (z)) | ((y) & ~(z)))
#define MD5_H(x, y, z) ((x) ^ (y) ^ (z))
#define MD5_I(x, y, z) ((y) ^ ((x) | ~(z)))
// (X[k] + T[i] does not depend on the chain: the dependent path of a step is f, one three-operand add, the rotate, one add)
#define MD5_STEP(f, a, b, c, d, k, s, t) a = b + md5_rotl(a + (X[k] + (t)) + f(b, c, d), s)
__device__ __forceinline__ void md5_block(uint32_t (&st)[4], const uint32_t (&X)[16])
{
	uint32_t a = st[0], b = st[1], c = st[2], d = st[3];
	MD5_STEP(MD5_F, a, b, c, d, 0, 7, 0xd76aa478u); MD5_STEP(MD5_F, d, a, b, c, 1, 12, 0xe8c7b756u); MD5_STEP(MD5_F, c, d, a, b, 2, 17, 0x242070dbu); MD5_STEP(MD5_F, b, c, d, a, 3, 22, 0xc1bdceeeu);
	MD5_STEP(MD5_F, a, b, c, d, 4, 7, 0xf57c0fafu); MD5_STEP(MD5_F, d, a, b, c, 5, 12, 0x4787c62au); MD5_STEP(MD5_F, c, d, a, b, 6, 17, 0xa8304613u); MD5_STEP(MD5_F, b, c, d, a, 7, 22, 0xfd469501u);
	MD5_STEP(MD5_F, a, b, c, d, 8, 7, 0x698098d8u); MD5_STEP(MD5_F, d, a, b, c, 9, 12, 0x8b44f7afu); MD5_STEP(MD5_F, c, d, a, b, 10, 17, 0xffff5bb1u); MD5_STEP(MD5_F, b, c, d, a, 11, 22, 0x895cd7beu);
	MD5_STEP(MD5_F, a, b, c, d, 12, 7, 0x6b901122u); MD5_STEP(MD5_F, d, a, b, c, 13, 12, 0xfd987193u); MD5_STEP(MD5_F, c, d, a, b, 14, 17, 0xa679438eu); MD5_STEP(MD5_F, b, c, d, a, 15, 22, 0x49b40821u);
	MD5_STEP(MD5_G, a, b, c, d, 1, 5, 0xf61e2562u); MD5_STEP(MD5_G, d, a, b, c, 6, 9, 0xc040b340u); MD5_STEP(MD5_G, c, d, a, b, 11, 14, 0x265e5a51u); MD5_STEP(MD5_G, b, c, d, a, 0, 20, 0xe9b6c7aau);
	MD5_STEP(MD5_G, a, b, c, d, 5, 5, 0xd62f105du); MD5_STEP(MD5_G, d, a, b, c, 10, 9, 0x02441453u); MD5_STEP(MD5_G, c, d, a, b, 15, 14, 0xd8a1e681u); MD5_STEP(MD5_G, b, c, d, a, 4, 20, 0xe7d3fbc8u);
	MD5_STEP(MD5_G, a, b, c, d, 9, 5, 0x21e1cde6u); MD5_STEP(MD5_G, d, a, b, c, 14, 9, 0xc33707d6u); MD5_STEP(MD5_G, c, d, a, b, 3, 14, 0xf4d50d87u); MD5_STEP(MD5_G, b, c, d, a, 8, 20, 0x455a14edu);
	MD5_STEP(MD5_G, a, b, c, d, 13, 5, 0xa9e3e905u); MD5_STEP(MD5_G, d, a, b, c, 2, 9, 0xfcefa3f8u); MD5_STEP(MD5_G, c, d, a, b, 7, 14, 0x676f02d9u); MD5_STEP(MD5_G, b, c, d, a, 12, 20, 0x8d2a4c8au);
	MD5_STEP(MD5_H, a, b, c, d, 5, 4, 0xfffa3942u); MD5_STEP(MD5_H, d, a, b, c, 8, 11, 0x8771f681u); MD5_STEP(MD5_H, c, d, a, b, 11, 16, 0x6d9d6122u); MD5_STEP(MD5_H, b, c, d, a, 14, 23, 0xfde5380cu);
	MD5_STEP(MD5_H, a, b, c, d, 1, 4, 0xa4beea44u); MD5_STEP(MD5_H, d, a, b, c, 4, 11, 0x4bdecfa9u); MD5_STEP(MD5_H, c, d, a, b, 7, 16, 0xf6bb4b60u); MD5_STEP(MD5_H, b, c, d, a, 10, 23, 0xbebfbc70u);
	MD5_STEP(MD5_H, a, b, c, d, 13, 4, 0x289b7ec6u); MD5_STEP(MD5_H, d, a, b, c, 0, 11, 0xeaa127fau); MD5_STEP(MD5_H, c, d, a, b, 3, 16, 0xd4ef3085u); MD5_STEP(MD5_H, b, c, d, a, 6, 23, 0x04881d05u);
	MD5_STEP(MD5_H, a, b, c, d, 9, 4, 0xd9d4d039u); MD5_STEP(MD5_H, d, a, b, c, 12, 11, 0xe6db99e5u); MD5_STEP(MD5_H, c, d, a, b, 15, 16, 0x1fa27cf8u); MD5_STEP(MD5_H, b, c, d, a, 2, 23, 0xc4ac5665u);
	MD5_STEP(MD5_I, a, b, c, d, 0, 6, 0xf4292244u); MD5_STEP(MD5_I, d, a, b, c, 7, 10, 0x432aff97u); MD5_STEP(MD5_I, c, d, a, b, 14, 15, 0xab9423a7u); MD5_STEP(MD5_I, b, c, d, a, 5, 21, 0xfc93a039u);
	MD5_STEP(MD5_I, a, b, c, d, 12, 6, 0x655b59c3u); MD5_STEP(MD5_I, d, a, b, c, 3, 10, 0x8f0ccc92u); MD5_STEP(MD5_I, c, d, a, b, 10, 15, 0xffeff47du); MD5_STEP(MD5_I, b, c, d, a, 1, 21, 0x85845dd1u);
	MD5_STEP(MD5_I, a, b, c, d, 8, 6, 0x6fa87e4fu); MD5_STEP(MD5_I, d, a, b, c, 15, 10, 0xfe2ce6e0u); MD5_STEP(MD5_I, c, d, a, b, 6, 15, 0xa3014314u); MD5_STEP(MD5_I, b, c, d, a, 13, 21, 0x4e0811a1u);
	MD5_STEP(MD5_I, a, b, c, d, 4, 6, 0xf7537e82u); MD5_STEP(MD5_I, d, a, b, c, 11, 10, 0xbd3af235u); MD5_STEP(MD5_I, c, d, a, b, 2, 15, 0x2ad7d2bbu); MD5_STEP(MD5_I, b, c, d, a, 9, 21, 0xeb86d391u);
	st[0] += a; st[1] += b; st[2] += c; st[3] += d;
}

// NB consecutive blocks (64 NB bytes) that start `done` bytes into the stream, as aligned words: W[0 .. 16 NB] (the word behind the
// last block included), sh = the byte of W[0] the first block starts at.  Nothing is read beyond the aligned word that holds the
// stream's last byte (indices are clamped: every load is unconditional and all of them are in flight at once).
template <int NB>
__device__ __forceinline__ void md5_fetch(const uint8_t *base, uint64_t done, uint64_t len, const void *safe, uint32_t (&W)[16 * NB + 1])
{
	const uint8_t *p = base + done;
	const uint32_t sh = (uint32_t)((uintptr_t)p & 3u);
	const uint64_t left = len > done ? len - done : 0;
	// (nothing left: the loads go to a word that is certainly there and what they bring is never used -- a load under a condition
	//  would be a branch with a wait behind it, one round trip after the other)
	const uint32_t *wp = left ? (const uint32_t *)(p - sh) : (const uint32_t *)safe;
	const uint32_t nwords = left ? (uint32_t)(((left > 64u * NB ? 64u * NB : left) + sh + 3) >> 2) : 1u;
#pragma unroll
	for(int k = 0; k < 16 * NB + 1; k++) W[k] = wp[(uint32_t)k < nwords ? (uint32_t)k : nwords - 1];
}
// block `b` of a fetched group as the 16 message words; bytes at and behind the stream's end read as zero
template <int NB>
__device__ __forceinline__ void md5_words(const uint32_t (&W)[16 * NB + 1], int b, uint32_t sh, uint64_t left /* bytes of the stream from this block on */, uint32_t (&X)[16])
{
#pragma unroll
	for(int k = 0; k < 16; k++) X[k] = __builtin_amdgcn_alignbyte(W[16 * b + k + 1], W[16 * b + k], sh);
	if(left < 64) {
		const uint32_t nb = (uint32_t)left;
#pragma unroll
		for(int k = 0; k < 16; k++) {
			const uint32_t lo = 4u * (uint32_t)k;
			if(nb <= lo) X[k] = 0;
			else if(nb < lo + 4) X[k] &= 0xffffffffu >> (8u * (lo + 4 - nb));
		}
	}
}

// A lane's chain is serial and alone on its SIMD most of the time (120 tracks are two wavefronts), so what it must never do is wait
// for memory: the next group of four blocks is fetched while the current one is hashed.
constexpr int MD5_NB = 4;
__global__ __launch_bounds__(64) void md5_many_kernel(const uint8_t *__restrict__ base, const uint64_t *__restrict__ offs, const uint64_t *__restrict__ lens, uint32_t n,
                                                      uint8_t *__restrict__ digests)
{
	const uint32_t i = blockIdx.x * 64 + threadIdx.x;
	const bool live = i < n;
	const uint8_t *p = base + (live ? offs[i] : 0);
	const uint64_t len = live ? lens[i] : 0;
	const uint32_t sh = (uint32_t)((uintptr_t)p & 3u);
	uint32_t st[4] = {0x67452301u, 0xefcdab89u, 0x98badcfeu, 0x10325476u};
	uint64_t done = 0;
	uint32_t X[16];
	uint32_t Wa[16 * MD5_NB + 1], Wb[16 * MD5_NB + 1];
	md5_fetch<MD5_NB>(p, 0, len, lens, Wa);
	// whole groups, two per pass (the buffers swap roles without being copied)
	while(len - done >= 64u * MD5_NB) {
		md5_fetch<MD5_NB>(p, done + 64u * MD5_NB, len, lens, Wb);
#pragma unroll
		for(int b = 0; b < MD5_NB; b++) { md5_words<MD5_NB>(Wa, b, sh, 64, X); md5_block(st, X); }
		done += 64u * MD5_NB;
		if(len - done < 64u * MD5_NB) {
#pragma unroll
			for(int k = 0; k < 16 * MD5_NB + 1; k++) Wa[k] = Wb[k];
			break;
		}
		md5_fetch<MD5_NB>(p, done + 64u * MD5_NB, len, lens, Wa);
#pragma unroll
		for(int b = 0; b < MD5_NB; b++) { md5_words<MD5_NB>(Wb, b, sh, 64, X); md5_block(st, X); }
		done += 64u * MD5_NB;
	}
	// what is left (fewer than four blocks) is in Wa: its whole blocks, then the last one or two with the padding --
	// 0x80, zeros, the length in bits (RFC 1321, 3.1-3.2)
	uint32_t rem = (uint32_t)(len - done);
#pragma unroll
	for(int b = 0; b < MD5_NB - 1; b++) {
		if(rem >= 64) { md5_words<MD5_NB>(Wa, b, sh, 64, X); md5_block(st, X); rem -= 64; done += 64; }
	}
	const uint32_t tb = (uint32_t)((len % (64u * MD5_NB)) / 64u);           // index in Wa of the block that holds the stream's last (partial) bytes
	if(rem) {
		// (a switch on tb with constant indices: the group lives in registers)
		if(tb == 0) md5_words<MD5_NB>(Wa, 0, sh, rem, X);
		else if(tb == 1) md5_words<MD5_NB>(Wa, 1, sh, rem, X);
		else if(tb == 2) md5_words<MD5_NB>(Wa, 2, sh, rem, X);
		else md5_words<MD5_NB>(Wa, 3, sh, rem, X);
	}
	else {
#pragma unroll
		for(int k = 0; k < 16; k++) X[k] = 0;
	}
#pragma unroll
	for(int k = 0; k < 16; k++) if((rem >> 2) == (uint32_t)k) X[k] |= 0x80u << (8u * (rem & 3u));
	if(rem >= 56) {
		md5_block(st, X);
#pragma unroll
		for(int k = 0; k < 16; k++) X[k] = 0;
	}
	X[14] = (uint32_t)(len << 3); X[15] = (uint32_t)(len >> 29);
	md5_block(st, X);
	if(live) {
		uint32_t *o = (uint32_t *)(digests + (size_t)i * 16);
		o[0] = st[0]; o[1] = st[1]; o[2] = st[2]; o[3] = st[3];
	}
}

} // namespace flacgpu

using namespace flacgpu;

// include/flacgpu.h: digests of n byte ranges of device memory, d_base + offsets[i] .. + lengths[i]; arrays on the host, digests to the host.
// d_base must be 4-byte aligned (the kernel reads the aligned word a range starts in: with an aligned base that word lies inside the
// caller's allocation whatever the range's own alignment); that the ranges lie inside the allocation is the caller's business, as
// with any pointer + length.  The small device scratch (offsets, lengths, digests) is kept per device and grows on demand: no
// hipMalloc / hipFree -- a device-wide synchronisation -- per call.
#include <mutex>
namespace {
// (a lock per device: calls for different devices do not wait for each other -- ADVICE r05; the scratch goes with the process)
struct Md5Scratch { void *p = nullptr; size_t cap = 0; std::mutex mu; };
Md5Scratch g_md5_scratch[64];
}
extern "C" int flacgpu_md5_many_device(int device, const void *d_base, const uint64_t *offsets, const uint64_t *lengths, uint32_t n, uint8_t *digests, void *stream)
{
	if(n == 0) return FLACGPU_OK;
	if(!d_base || !offsets || !lengths || !digests || ((uintptr_t)d_base & 3u) || device < 0 || device >= 64) return FLACGPU_ERR_BAD_ARG;
	if(hipSetDevice(device) != hipSuccess) return FLACGPU_ERR_NO_DEVICE;
	hipStream_t s = (hipStream_t)stream;
	// one call per device at a time uses the scratch (the call is synchronous on its stream anyway)
	Md5Scratch &sc = g_md5_scratch[device];
	std::lock_guard<std::mutex> lock(sc.mu);
	const size_t need = (size_t)n * 32;
	if(sc.cap < need) {
		if(sc.p) (void)hipFree(sc.p);
		sc.p = nullptr; sc.cap = 0;
		const size_t cap = need < 65536 ? 65536 : need;
		if(hipMalloc(&sc.p, cap) != hipSuccess) return FLACGPU_ERR_ALLOC;
		sc.cap = cap;
	}
	uint64_t *d_meta = (uint64_t *)sc.p;
	uint8_t *d_dig = (uint8_t *)sc.p + (size_t)n * 16;
	int r = FLACGPU_OK;
	if(hipMemcpyAsync(d_meta, offsets, (size_t)n * 8, hipMemcpyHostToDevice, s) != hipSuccess || hipMemcpyAsync(d_meta + n, lengths, (size_t)n * 8, hipMemcpyHostToDevice, s) != hipSuccess) r = FLACGPU_ERR_LAUNCH;
	if(r == FLACGPU_OK) {
		hipLaunchKernelGGL(md5_many_kernel, dim3((n + 63) / 64), dim3(64), 0, s, (const uint8_t *)d_base, d_meta, d_meta + n, n, d_dig);
		if(hipGetLastError() != hipSuccess || hipMemcpyAsync(digests, d_dig, (size_t)n * 16, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) r = FLACGPU_ERR_LAUNCH;
	}
	return r;
}
