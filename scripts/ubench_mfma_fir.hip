// scripts/ubench_mfma_fir.hip -- does the idle matrix pipe pay for the FIR residual of the -8 evaluation?  
//
// The evaluation kernel (flac_amd/csrc/flacgpu_evalg.hip) spends 80 % of its VALU instructions on t[i] = sum_k c_k x[i-k] (folded
// taps c_0 = -2^shift, c_k = q_{k-1}; 16-bit samples, taps of up to 12 bits) as chains of v_dot2_i32_i16, one shift and one
// v_sad_u32 per candidate-sample.  This program measures, on the same data and with the same result (sum over the block of
// |t >> shift|, checked against a scalar host loop), two ways of getting there:
//   chain   the kernel's way: a wavefront owns a 4096-sample channel (lane = 64 consecutive samples), the candidates two at a time,
//           per 16-sample piece 15 LDS words + 14 shifted words per pair, then NPF dot2 + shift + sad per candidate-sample;
//   mfma    v_mfma_i32_16x16x32_i8: a tile is 16 candidates x 16 samples.  Samples split into a signed high byte and a low byte
//           offset by 128 (x = 256 xh + xl' + 128), taps into balanced signed bytes (c = 256 ch + cl): three products per tile
//           with ONE B operand [xl' window | xh window] --  HH = [0 | ch] B,  MID = [ch | cl] B,  LL = [cl | 0] B  --
//           t = (HH << 16) + (MID << 8) + LL + 128 sum(c); then shift and sad on the VALU, 4 result registers per lane.
//           B comes from two byte planes in LDS stored back to front (the window of a sample is 16 ascending bytes).
// Reported: time per launch, and nanoseconds of SIMD time per USEFUL candidate-sample for 10, 12 and 16 candidates (the matrix
// tile always computes 16 rows).   build: hipcc --offload-arch=gfx950 -O3 -o build/ubench_mfma_fir scripts/ubench_mfma_fir.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while(0)

constexpr int N = 4096;            // samples per channel
constexpr int NTAP = 13;           // folded taps of an order-12 candidate
constexpr int MAXC = 16;
typedef int v4i __attribute__((ext_vector_type(4)));

struct Cands { int16_t c[MAXC][16]; uint32_t shift[MAXC]; };      // c[.][0] = -2^shift, c[.][1..12] the quantised coefficients, rest 0

// ---------------------------------------------------------------------------------------------------------------------------
// chain: the kernel's inner loop, reduced to what costs (no Rice search, no first-piece special case: lane 0's history is zeros)
// ---------------------------------------------------------------------------------------------------------------------------
constexpr uint32_t ROW = 65 * 4;
__device__ __forceinline__ uint32_t chain7(const uint32_t (&W)[7], const uint32_t (&Q)[7], uint32_t sum0, uint32_t shift)
{
	uint32_t d;
	asm("v_dot2_i32_i16 %0, %2, %3, %1\n\tv_dot2_i32_i16 %0, %4, %5, %0\n\tv_dot2_i32_i16 %0, %6, %7, %0\n\tv_dot2_i32_i16 %0, %8, %9, %0\n\tv_dot2_i32_i16 %0, %10, %11, %0\n\tv_dot2_i32_i16 %0, %12, %13, %0\n\tv_dot2_i32_i16 %0, %14, %15, %0\n\tv_lshrrev_b32 %0, %16, %0"
	    : "=&v"(d) : "v"(sum0), "v"(W[0]), "s"(Q[0]), "v"(W[1]), "s"(Q[1]), "v"(W[2]), "s"(Q[2]), "v"(W[3]), "s"(Q[3]), "v"(W[4]), "s"(Q[4]), "v"(W[5]), "s"(Q[5]), "v"(W[6]), "s"(Q[6]), "s"(shift));
	return d;
}
__device__ __forceinline__ uint32_t sad_s(uint32_t a, uint32_t b, uint32_t c) { uint32_t d; asm("v_sad_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(b), "v"(c)); return d; }
__device__ __forceinline__ uint32_t piece16(const uint32_t (&AA)[15], const uint32_t (&BB)[14], const uint32_t (&Q)[7], uint32_t shift, uint32_t bias, uint32_t acc)
{
#pragma unroll
	for(int s = 0; s < 16; s++) {
		uint32_t W[7];
#pragma unroll
		for(int p = 0; p < 7; p++) W[p] = (s & 1) ? AA[(s + 13) / 2 - p] : BB[s / 2 + 6 - p];
		acc = sad_s(chain7(W, Q, 0x80000000u, shift), bias, acc);
	}
	return acc;
}
__global__ __launch_bounds__(64, 4) void chain_kernel(const int16_t *__restrict__ x, const Cands *__restrict__ cands, uint32_t ncand, uint64_t *__restrict__ out)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int lane = (int)threadIdx.x;
	const uint32_t ch = blockIdx.x;
	const uint32_t S = N / 64, rows = S / 2;
	// image: word j of lane L's run at j * 65 + L + 1; column 0 (lane 0's history): zeros
	const uint4 *src = (const uint4 *)(x + (size_t)ch * N);
	if(lane < 8) *(uint32_t *)(smem + (rows - 8 + (uint32_t)lane) * ROW) = 0;
	for(uint32_t m = (uint32_t)lane; m < N / 8; m += 64) {
		const uint4 v = src[m];
		const uint32_t Lo = m / (S / 8), r = m - Lo * (S / 8);
		unsigned char *d = smem + (Lo + 1) * 4 + 4 * r * ROW;
		*(uint32_t *)d = v.x; *(uint32_t *)(d + ROW) = v.y; *(uint32_t *)(d + 2 * ROW) = v.z; *(uint32_t *)(d + 3 * ROW) = v.w;
	}
	__builtin_amdgcn_wave_barrier();
	__syncthreads();
	const unsigned char *own = smem + ((uint32_t)lane + 1) * 4;
	const unsigned char *hist = smem + (uint32_t)lane * 4 + (rows - 7) * ROW;
	for(uint32_t c0 = 0; c0 < ncand; c0 += 2) {
		uint32_t QA[7], QB[7];
		const uint32_t c1 = c0 + 1 < ncand ? c0 + 1 : c0;
#pragma unroll
		for(int p = 0; p < 7; p++) {
			QA[p] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(((uint32_t)(uint16_t)cands->c[c0][2 * p] << 16) | (uint16_t)cands->c[c0][2 * p + 1]));
			QB[p] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(((uint32_t)(uint16_t)cands->c[c1][2 * p] << 16) | (uint16_t)cands->c[c1][2 * p + 1]));
		}
		const uint32_t sA = (uint32_t)__builtin_amdgcn_readfirstlane((int)cands->shift[c0]), sB = (uint32_t)__builtin_amdgcn_readfirstlane((int)cands->shift[c1]);
		const uint32_t bA = 0x80000000u >> sA, bB = 0x80000000u >> sB;
		uint32_t v0 = 0, v1 = 0;
		{
			uint32_t AA[15], BB[14];
#pragma unroll
			for(int j = 0; j < 7; j++) AA[j] = *(const uint32_t *)(hist + j * ROW);
#pragma unroll
			for(int j = 7; j < 15; j++) AA[j] = *(const uint32_t *)(own + (j - 7) * ROW);
#pragma unroll
			for(int m = 0; m < 14; m++) BB[m] = __builtin_amdgcn_alignbit(AA[m + 1], AA[m], 16);
			v0 = piece16(AA, BB, QA, sA, bA, v0);
			v1 = piece16(AA, BB, QB, sB, bB, v1);
		}
#pragma unroll 1
		for(uint32_t c = 1; c < S / 16; c++) {
			uint32_t AA[15], BB[14];
			const unsigned char *base = own + (8 * c - 7) * ROW;
#pragma unroll
			for(int j = 0; j < 15; j++) AA[j] = *(const uint32_t *)(base + j * ROW);
#pragma unroll
			for(int m = 0; m < 14; m++) BB[m] = __builtin_amdgcn_alignbit(AA[m + 1], AA[m], 16);
			v0 = piece16(AA, BB, QA, sA, bA, v0);
			v1 = piece16(AA, BB, QB, sB, bB, v1);
		}
		// the lanes' sums -> the block's (the real kernel runs the Rice search on them instead)
		uint64_t t0 = v0, t1 = v1;
		for(int o = 32; o; o >>= 1) { t0 += __shfl_xor(t0, o); t1 += __shfl_xor(t1, o); }
		if(lane == 0) { out[(size_t)ch * MAXC + c0] = t0; if(c1 != c0) out[(size_t)ch * MAXC + c1] = t1; }
	}
}

// ---------------------------------------------------------------------------------------------------------------------------
// mfma
// ---------------------------------------------------------------------------------------------------------------------------
constexpr uint32_t PLANE = N + 64;          // bytes per reversed byte plane: rev[p] = byte of sample N - 1 - p; the 48 behind it: history = 0 (xh) / -128 (xl')
__global__ __launch_bounds__(64, 4) void mfma_kernel(const int16_t *__restrict__ x, const Cands *__restrict__ cands, uint64_t *__restrict__ out)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	unsigned char *lo = smem, *hi = smem + PLANE;
	const int lane = (int)threadIdx.x;
	const uint32_t ch = blockIdx.x;
	// ---- byte planes, back to front: a sample's window x[i], x[i-1], ... is ascending bytes ------------------------------------
	{
		const uint4 *src = (const uint4 *)(x + (size_t)ch * N);
		for(uint32_t m = (uint32_t)lane; m < N / 8; m += 64) {
			const uint4 v = src[m];                                    // samples 8 m .. 8 m + 7
			const uint32_t w[4] = {v.x, v.y, v.z, v.w};
			uint32_t l0 = 0, l1 = 0, h0 = 0, h1 = 0;
#pragma unroll
			for(int k = 0; k < 8; k++) {
				const int32_t s = (int16_t)(w[k >> 1] >> (16 * (k & 1)));
				const uint32_t xl = ((uint32_t)s & 0xffu) ^ 0x80u;      // xl - 128 as a signed byte
				const uint32_t xh = ((uint32_t)(s >> 8)) & 0xffu;
				// sample 8 m + k goes to reversed position N - 1 - 8 m - k: byte (7 - k) of the 8-byte group at N - 8 - 8 m
				if(7 - k < 4) { l0 |= xl << (8 * (7 - k)); h0 |= xh << (8 * (7 - k)); } else { l1 |= xl << (8 * (3 - k)); h1 |= xh << (8 * (3 - k)); }
			}
			*(uint2 *)(lo + (N - 8 - 8 * m)) = make_uint2(l0, l1);
			*(uint2 *)(hi + (N - 8 - 8 * m)) = make_uint2(h0, h1);
		}
		if(lane < 16) { *(uint32_t *)(lo + N + 4 * lane) = 0x80808080u; *(uint32_t *)(hi + N + 4 * lane) = 0; }      // samples in front of the block: 0 = 256 * 0 + (-128 + 128)
	}
	// ---- A operands (constant for the channel): lane (i = lane % 16 candidate, kb = lane / 16): bytes k = 8 kb .. 8 kb + 7 of a 32-deep row
	//      MID row = [ch_0..ch_15 | cl_0..cl_15], HH row = [0 | ch], LL row = [cl | 0];  B column = [xl' window (16) | xh window (16)]
	const uint32_t ci = (uint32_t)lane & 15u, kb = (uint32_t)lane >> 4;
	uint64_t a_mid = 0, a_hh = 0, a_ll = 0;
	int32_t csum = 0;
	{
#pragma unroll
		for(int k = 0; k < 16; k++) csum += cands->c[ci][k];
#pragma unroll
		for(int kk = 0; kk < 8; kk++) {
			const uint32_t k = 8 * (kb & 1u) + (uint32_t)kk;           // tap index
			const int32_t c = cands->c[ci][k];
			const int32_t chh = (c + 128) >> 8, cll = c - chh * 256;   // balanced split: cl in [-128, 127]
			const uint64_t bh = (uint64_t)((uint32_t)chh & 0xffu) << (8 * kk), bl = (uint64_t)((uint32_t)cll & 0xffu) << (8 * kk);
			if(kb < 2) { a_mid |= bh; a_ll |= bl; } else { a_mid |= bl; a_hh |= bh; }
		}
	}
	// per result register r: candidate 4 (lane / 16) + r -- its shift, its bias, and the constant 128 sum(c) + 2^31 the LL product starts from
	uint32_t sh[4], bias[4];
	v4i kinit;
#pragma unroll
	for(int r = 0; r < 4; r++) {
		const uint32_t cr = 4u * kb + (uint32_t)r;
		sh[r] = cands->shift[cr]; bias[r] = 0x80000000u >> sh[r];
		int32_t cs = 0;
#pragma unroll
		for(int k = 0; k < 16; k++) cs += cands->c[cr][k];
		kinit[r] = (int32_t)(0x80000000u + (uint32_t)(128 * cs));
	}
	(void)csum;
	__syncthreads();
	uint32_t acc[4] = {0, 0, 0, 0};
	const v4i zero = {0, 0, 0, 0};
	const uint32_t j = (uint32_t)lane & 15u;
	const unsigned char *plane = (kb < 2 ? lo : hi) + 8 * (kb & 1u);
#pragma unroll 4
	for(uint32_t i0 = 0; i0 < N; i0 += 16) {
		// B: 8 ascending bytes of the reversed plane from the position of sample i0 + j (two aligned words and the one behind them, aligned)
		const uint32_t pos = (uint32_t)(N - 1) - i0 - j;
		const unsigned char *p = plane + pos;
		const uint32_t al = (uint32_t)(uintptr_t)p & 3u;
		const uint32_t *pw = (const uint32_t *)(p - al);
		const uint32_t w0 = pw[0], w1 = pw[1], w2 = pw[2];
		const uint32_t b0 = __builtin_amdgcn_alignbyte(w1, w0, al), b1 = __builtin_amdgcn_alignbyte(w2, w1, al);
		const long b = (long)(((uint64_t)b1 << 32) | b0);
		const v4i hh = __builtin_amdgcn_mfma_i32_16x16x32_i8((long)a_hh, b, zero, 0, 0, 0);
		const v4i mid = __builtin_amdgcn_mfma_i32_16x16x32_i8((long)a_mid, b, zero, 0, 0, 0);
		const v4i ll = __builtin_amdgcn_mfma_i32_16x16x32_i8((long)a_ll, b, kinit, 0, 0, 0);
#pragma unroll
		for(int r = 0; r < 4; r++) {
			const uint32_t t = ((uint32_t)hh[r] << 16) + (((uint32_t)mid[r] << 8) + (uint32_t)ll[r]);
			const uint32_t pb = t >> sh[r];
			acc[r] += pb > bias[r] ? pb - bias[r] : bias[r] - pb;           // (v_sad_u32)
		}
	}
	// the block's sums: over the 16 sample lanes of a candidate row group (the real kernel would do this per 64-sample leaf)
#pragma unroll
	for(int r = 0; r < 4; r++) {
		uint64_t t = acc[r];
		for(int o = 8; o; o >>= 1) t += __shfl_xor(t, o);
		if(j == 0) out[(size_t)ch * MAXC + 4 * kb + (uint32_t)r] = t;
	}
}

// ---------------------------------------------------------------------------------------------------------------------------
// mfma4: the same products with the 16 rows of a tile used as 4 candidates x 4 SAMPLE PHASES.  Column j of B is the window that ends
// at sample i0 + 4 j + 3 (16 bytes per plane: 13 taps + 3 phases), row 4 c + ph of A holds candidate c's taps moved 3 - ph places
// down the window, so D[4 c + ph][j] is t of sample i0 + 4 j + ph: a tile covers 64 SAMPLES x 4 candidates, a lane (j, c) ends up with
// four consecutive samples of ONE candidate in its four result registers (one shift count, one bias per lane), B is loaded once per
// 64 samples for all candidate sets, and 10 candidates fill 10 of 12 rows-of-four instead of 10 of 16 rows.  64 samples are one
// leaf partition of the -8 Rice search at 4096-sample blocks: the tile's sum over the 16 lanes of a candidate is a leaf sum.
// ---------------------------------------------------------------------------------------------------------------------------
template <int NSETS>
__global__ __launch_bounds__(64, 4) void mfma4_kernel(const int16_t *__restrict__ x, const Cands *__restrict__ cands, uint64_t *__restrict__ out)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	unsigned char *lo = smem, *hi = smem + PLANE;
	uint32_t *leaf = (uint32_t *)(smem + 2 * PLANE);               // [4 * NSETS candidates][64 leaves]
	const int lane = (int)threadIdx.x;
	const uint32_t ch = blockIdx.x;
	{
		const uint4 *src = (const uint4 *)(x + (size_t)ch * N);
		for(uint32_t m = (uint32_t)lane; m < N / 8; m += 64) {
			const uint4 v = src[m];
			const uint32_t w[4] = {v.x, v.y, v.z, v.w};
			uint32_t l0 = 0, l1 = 0, h0 = 0, h1 = 0;
#pragma unroll
			for(int k = 0; k < 8; k++) {
				const int32_t s = (int16_t)(w[k >> 1] >> (16 * (k & 1)));
				const uint32_t xl = ((uint32_t)s & 0xffu) ^ 0x80u, xh = ((uint32_t)(s >> 8)) & 0xffu;
				if(7 - k < 4) { l0 |= xl << (8 * (7 - k)); h0 |= xh << (8 * (7 - k)); } else { l1 |= xl << (8 * (3 - k)); h1 |= xh << (8 * (3 - k)); }
			}
			*(uint2 *)(lo + (N - 8 - 8 * m)) = make_uint2(l0, l1);
			*(uint2 *)(hi + (N - 8 - 8 * m)) = make_uint2(h0, h1);
		}
		if(lane < 16) { *(uint32_t *)(lo + N + 4 * lane) = 0x80808080u; *(uint32_t *)(hi + N + 4 * lane) = 0; }
	}
	// A operands: lane (row = lane % 16 = 4 c + ph, kb = lane / 16): window places k'' = 8 (kb & 1) .. + 7; tap index k = k'' - (3 - ph)
	const uint32_t row = (uint32_t)lane & 15u, kb = (uint32_t)lane >> 4, ac = row >> 2, ph = row & 3u;
	uint64_t a_mid[NSETS], a_hh[NSETS], a_ll[NSETS];
	v4i kinit[NSETS];
	uint32_t sh[NSETS], bias[NSETS];
#pragma unroll
	for(int st = 0; st < NSETS; st++) {
		a_mid[st] = 0; a_hh[st] = 0; a_ll[st] = 0;
		const uint32_t ci = 4u * (uint32_t)st + ac;
#pragma unroll
		for(int kk = 0; kk < 8; kk++) {
			const int32_t k = (int32_t)(8 * (kb & 1u)) + kk - (3 - (int32_t)ph);
			const int32_t c = k >= 0 && k < 16 ? (int32_t)cands->c[ci][k] : 0;
			const int32_t chh = (c + 128) >> 8, cll = c - chh * 256;
			const uint64_t bh = (uint64_t)((uint32_t)chh & 0xffu) << (8 * kk), bl = (uint64_t)((uint32_t)cll & 0xffu) << (8 * kk);
			if(kb < 2) { a_mid[st] |= bh; a_ll[st] |= bl; } else { a_mid[st] |= bl; a_hh[st] |= bh; }
		}
		// this lane's results: candidate 4 st + kb (D row 4 kb + r = candidate kb of the set, phase r)
		const uint32_t cr = 4u * (uint32_t)st + kb;
		sh[st] = cands->shift[cr]; bias[st] = 0x80000000u >> sh[st];
		int32_t cs = 0;
#pragma unroll
		for(int k = 0; k < 16; k++) cs += cands->c[cr][k];
		const int32_t k0 = (int32_t)(0x80000000u + (uint32_t)(128 * cs));
		kinit[st] = v4i{k0, k0, k0, k0};
	}
	__syncthreads();
	const v4i zero = {0, 0, 0, 0};
	const uint32_t j = (uint32_t)lane & 15u;
	const unsigned char *plane = (kb < 2 ? lo : hi) + 8 * (kb & 1u);
#pragma unroll 2
	for(uint32_t i0 = 0; i0 < N; i0 += 64) {
		const uint32_t pos = (uint32_t)(N - 1) - (i0 + 4u * j + 3u);             // the window of column j ends at sample i0 + 4 j + 3
		const unsigned char *p = plane + pos;
		const uint32_t al = (uint32_t)(uintptr_t)p & 3u;
		const uint32_t *pw = (const uint32_t *)(p - al);
		const uint32_t w0 = pw[0], w1 = pw[1], w2 = pw[2];
		const uint32_t b0 = __builtin_amdgcn_alignbyte(w1, w0, al), b1 = __builtin_amdgcn_alignbyte(w2, w1, al);
		const long b = (long)(((uint64_t)b1 << 32) | b0);
#pragma unroll
		for(int st = 0; st < NSETS; st++) {
			const v4i hh = __builtin_amdgcn_mfma_i32_16x16x32_i8((long)a_hh[st], b, zero, 0, 0, 0);
			const v4i mid = __builtin_amdgcn_mfma_i32_16x16x32_i8((long)a_mid[st], b, zero, 0, 0, 0);
			const v4i ll = __builtin_amdgcn_mfma_i32_16x16x32_i8((long)a_ll[st], b, kinit[st], 0, 0, 0);
			uint32_t acc = 0;
#pragma unroll
			for(int r = 0; r < 4; r++) {
				const uint32_t t = ((uint32_t)hh[r] << 16) + (((uint32_t)mid[r] << 8) + (uint32_t)ll[r]);
				const uint32_t pb = t >> sh[st];
				acc += pb > bias[st] ? pb - bias[st] : bias[st] - pb;
			}
			// the leaf's sum of this lane's candidate: over the 16 lanes of its row group
			acc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)acc, 0x111, 0xf, 0xf, false);      // row_shr:1
			acc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)acc, 0x112, 0xf, 0xf, false);      // row_shr:2
			acc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)acc, 0x114, 0xf, 0xf, false);      // row_shr:4
			acc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)acc, 0x118, 0xf, 0xf, false);      // row_shr:8
			if(j == 15) leaf[(4u * (uint32_t)st + kb) * 64u + (i0 >> 6)] = acc;
		}
	}
	__syncthreads();
	// (the real kernel hands leaf[c][lane] to the Rice search; here: the block's totals)
	for(uint32_t c = 0; c < 4u * NSETS; c++) {
		uint64_t t = leaf[c * 64u + (uint32_t)lane];
		for(int o = 32; o; o >>= 1) t += __shfl_xor(t, o);
		if(lane == 0) out[(size_t)ch * MAXC + c] = t;
	}
}

// ---------------------------------------------------------------------------------------------------------------------------
int main(int argc, char **argv)
{
	const uint32_t nch = argc > 1 ? (uint32_t)atoi(argv[1]) : 65536;          // channels = wavefronts (16384 stereo frames x L, R, M, S)
	std::vector<int16_t> hx((size_t)nch * N);
	uint32_t seed = 12345;
	auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return seed >> 8; };
	for(uint32_t c = 0; c < nch; c++) {
		int32_t v = 0;
		for(int i = 0; i < N; i++) { v += (int32_t)(rnd() % 2001) - 1000; if(v > 30000) v = 30000; if(v < -30000) v = -30000; hx[(size_t)c * N + i] = (int16_t)v; }
	}
	hx[5] = 32767; hx[6] = -32768; hx[7] = 255; hx[8] = -129; hx[9] = 128;          // byte-boundary cases
	Cands hc;
	memset(&hc, 0, sizeof hc);
	for(int c = 0; c < MAXC; c++) {
		hc.shift[c] = 9 + (uint32_t)(c % 3);
		hc.c[c][0] = (int16_t)-(1 << hc.shift[c]);
		for(int k = 1; k < NTAP; k++) hc.c[c][k] = (int16_t)((int32_t)(rnd() % 1201) - 600 + (k == 1 ? 900 : 0));
	}
	hc.c[3][4] = 2047; hc.c[3][5] = -2048; hc.c[4][2] = 127; hc.c[4][3] = 128; hc.c[4][6] = -128; hc.c[4][7] = -129;
	// reference on the first few channels
	const uint32_t ncheck = nch < 8 ? nch : 8;
	std::vector<uint64_t> ref((size_t)ncheck * MAXC);
	for(uint32_t ch = 0; ch < ncheck; ch++)
		for(int c = 0; c < MAXC; c++) {
			uint64_t s = 0;
			for(int i = 0; i < N; i++) {
				int32_t t = 0;
				for(int k = 0; k < NTAP; k++) t += (int32_t)hc.c[c][k] * (i - k >= 0 ? (int32_t)hx[(size_t)ch * N + i - k] : 0);
				const int32_t r = -(t >> hc.shift[c]);
				s += (uint64_t)(r < 0 ? -(int64_t)r : r);
			}
			ref[(size_t)ch * MAXC + c] = s;
		}
	int16_t *dx; Cands *dc; uint64_t *dout;
	CK(hipMalloc(&dx, hx.size() * 2)); CK(hipMalloc(&dc, sizeof hc)); CK(hipMalloc(&dout, (size_t)nch * MAXC * 8));
	CK(hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dc, &hc, sizeof hc, hipMemcpyHostToDevice));
	std::vector<uint64_t> got((size_t)nch * MAXC);
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	const size_t lds_chain = (N / 128) * ROW + 256, lds_mfma = 2 * PLANE;
	auto check = [&](const char *what, uint32_t nc) {
		CK(hipMemcpy(got.data(), dout, got.size() * 8, hipMemcpyDeviceToHost));
		int bad = 0;
		for(uint32_t ch = 0; ch < ncheck; ch++) for(uint32_t c = 0; c < nc; c++) if(got[(size_t)ch * MAXC + c] != ref[(size_t)ch * MAXC + c]) { if(bad < 4) fprintf(stderr, "%s: channel %u candidate %u: %llu, reference %llu\n", what, ch, c, (unsigned long long)got[(size_t)ch * MAXC + c], (unsigned long long)ref[(size_t)ch * MAXC + c]); bad++; }
		return bad;
	};
	printf("FIR of %u channels x %d samples, order-12 candidates (13 folded taps); SIMD time = launch time x 1024 SIMDs; per USEFUL candidate-sample\n", nch, N);
	for(uint32_t nc : {10u, 12u, 16u}) {
		CK(hipMemset(dout, 0, (size_t)nch * MAXC * 8));
		for(int w = 0; w < 2; w++) hipLaunchKernelGGL(chain_kernel, dim3(nch), dim3(64), lds_chain, 0, dx, dc, nc, dout);
		CK(hipEventRecord(e0));
		const int reps = 5;
		for(int w = 0; w < reps; w++) hipLaunchKernelGGL(chain_kernel, dim3(nch), dim3(64), lds_chain, 0, dx, dc, nc, dout);
		CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
		float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
		const int bad = check("chain", nc);
		printf("chain  %2u candidates: %7.3f ms per launch   %6.3f ns SIMD time per candidate-sample   %s\n", nc, ms, ms * 1e6 * 1024.0 / ((double)nch * N * nc), bad ? "RESULTS DIFFER" : "results = reference");
	}
	{
		CK(hipMemset(dout, 0, (size_t)nch * MAXC * 8));
		for(int w = 0; w < 2; w++) hipLaunchKernelGGL(mfma_kernel, dim3(nch), dim3(64), lds_mfma, 0, dx, dc, dout);
		CK(hipEventRecord(e0));
		const int reps = 5;
		for(int w = 0; w < reps; w++) hipLaunchKernelGGL(mfma_kernel, dim3(nch), dim3(64), lds_mfma, 0, dx, dc, dout);
		CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
		float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
		const int bad = check("mfma", 16);
		printf("mfma   16-row tiles : %7.3f ms per launch   %s\n", ms, bad ? "RESULTS DIFFER" : "results = reference");
		for(uint32_t nc : {10u, 12u, 16u}) printf("mfma   %2u useful rows: %6.3f ns SIMD time per candidate-sample\n", nc, ms * 1e6 * 1024.0 / ((double)nch * N * nc));
	}
	for(int nsets : {3, 4}) {
		CK(hipMemset(dout, 0, (size_t)nch * MAXC * 8));
		const size_t lds4 = 2 * PLANE + (size_t)4 * nsets * 64 * 4;
		auto go = [&]() { if(nsets == 3) hipLaunchKernelGGL(mfma4_kernel<3>, dim3(nch), dim3(64), lds4, 0, dx, dc, dout); else hipLaunchKernelGGL(mfma4_kernel<4>, dim3(nch), dim3(64), lds4, 0, dx, dc, dout); };
		for(int w = 0; w < 2; w++) go();
		CK(hipEventRecord(e0));
		const int reps = 5;
		for(int w = 0; w < reps; w++) go();
		CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
		float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
		const int bad = check("mfma4", 4u * (uint32_t)nsets);
		printf("mfma4  %d sets of 4 candidates x 4 sample phases (64-sample tiles): %7.3f ms per launch   %s\n", nsets, ms, bad ? "RESULTS DIFFER" : "results = reference");
		for(uint32_t nc : {10u, 12u, 16u}) if(nc <= 4u * (uint32_t)nsets && nc > 4u * (uint32_t)(nsets - 1)) printf("mfma4  %2u useful candidates of %d: %6.3f ns SIMD time per candidate-sample\n", nc, 4 * nsets, ms * 1e6 * 1024.0 / ((double)nch * N * nc));
	}
	return 0;
}
