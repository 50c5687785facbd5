// scripts/ubench_chain.hip -- how fast ONE wavefront issues the evaluation kernel's dependent chain (7 x v_dot2_i32_i16 ->
// v_lshrrev -> v_sad_u32 per sample), alone and with 2..8 wavefronts per SIMD, as one chain or as two interleaved chains.
// hipcc --offload-arch=gfx950 -O3 -o build/ubc scripts/ubench_chain.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITER 2048
template <int MODE>
__global__ __launch_bounds__(64) void k(uint32_t *out, const uint32_t *in, int n)
{
	const int t = threadIdx.x;
	uint32_t w[8];
	for(int u = 0; u < 8; u++) w[u] = in[(t + u) & 255];
	const uint32_t q0 = __builtin_amdgcn_readfirstlane(in[1]), q1 = __builtin_amdgcn_readfirstlane(in[2]), sh = 3, bias = 0x10000000u, s0 = 0x80000000u;
	uint32_t acc0 = 0, acc1 = 0;
	for(int it = 0; it < n; it++) {
#pragma unroll
		for(int s = 0; s < 8; s++) {
			if(MODE == 0) {
				uint32_t d, e;
				asm volatile("v_dot2_i32_i16 %0, %2, %3, %1\n\tv_dot2_i32_i16 %0, %4, %3, %0\n\tv_dot2_i32_i16 %0, %5, %3, %0\n\tv_dot2_i32_i16 %0, %6, %3, %0\n\tv_dot2_i32_i16 %0, %7, %3, %0\n\tv_dot2_i32_i16 %0, %8, %3, %0\n\tv_dot2_i32_i16 %0, %9, %3, %0\n\tv_lshrrev_b32 %0, %10, %0"
				             : "=&v"(d) : "v"(s0), "v"(w[0]), "s"(q0), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "s"(sh));
				asm volatile("v_sad_u32 %0, %1, %2, %0" : "+v"(acc0) : "v"(d), "s"(bias));
				asm volatile("v_dot2_i32_i16 %0, %2, %3, %1\n\tv_dot2_i32_i16 %0, %4, %3, %0\n\tv_dot2_i32_i16 %0, %5, %3, %0\n\tv_dot2_i32_i16 %0, %6, %3, %0\n\tv_dot2_i32_i16 %0, %7, %3, %0\n\tv_dot2_i32_i16 %0, %8, %3, %0\n\tv_dot2_i32_i16 %0, %9, %3, %0\n\tv_lshrrev_b32 %0, %10, %0"
				             : "=&v"(e) : "v"(s0), "v"(w[1]), "s"(q1), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]), "s"(sh));
				asm volatile("v_sad_u32 %0, %1, %2, %0" : "+v"(acc1) : "v"(e), "s"(bias));
			}
			else {
				uint32_t d, e;
				asm volatile("v_dot2_i32_i16 %0, %3, %4, %2\n\tv_dot2_i32_i16 %1, %5, %12, %2\n\tv_dot2_i32_i16 %0, %5, %4, %0\n\tv_dot2_i32_i16 %1, %6, %12, %1\n\tv_dot2_i32_i16 %0, %6, %4, %0\n\tv_dot2_i32_i16 %1, %7, %12, %1\n\t"
				             "v_dot2_i32_i16 %0, %7, %4, %0\n\tv_dot2_i32_i16 %1, %8, %12, %1\n\tv_dot2_i32_i16 %0, %8, %4, %0\n\tv_dot2_i32_i16 %1, %9, %12, %1\n\tv_dot2_i32_i16 %0, %9, %4, %0\n\tv_dot2_i32_i16 %1, %10, %12, %1\n\t"
				             "v_dot2_i32_i16 %0, %10, %4, %0\n\tv_dot2_i32_i16 %1, %11, %12, %1\n\tv_lshrrev_b32 %0, %13, %0\n\tv_lshrrev_b32 %1, %13, %1"
				             : "=&v"(d), "=&v"(e) : "v"(s0), "v"(w[0]), "s"(q0), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]), "s"(q1), "s"(sh));
				asm volatile("v_sad_u32 %0, %2, %4, %0\n\tv_sad_u32 %1, %3, %4, %1" : "+v"(acc0), "+v"(acc1) : "v"(d), "v"(e), "s"(bias));
			}
		}
	}
	out[blockIdx.x * 64 + t] = acc0 + acc1;
}
template <int MODE>
static void run(const char *name, uint32_t *out, const uint32_t *in, int wps)
{
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	const int blocks = 256 * 4 * wps;
	hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, in, 16);
	(void)hipDeviceSynchronize();
	(void)hipEventRecord(e0);
	hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, in, ITER);
	(void)hipEventRecord(e1);
	(void)hipEventSynchronize(e1);
	float ms;
	(void)hipEventElapsedTime(&ms, e0, e1);
	const double inst_per_wave = (double)ITER * 8 * 18;
	printf("%-12s %d waves/SIMD: %8.3f ms   %.2f cycles per instruction of one wave, %.2f SIMD-cycles per wave-instruction (@2.4GHz)\n", name, wps, ms, ms * 1e-3 * 2.4e9 / inst_per_wave, ms * 1e-3 * 2.4e9 / inst_per_wave / wps);
}
int main()
{
	uint32_t *out, *in;
	(void)hipMalloc(&out, 256 * 4 * 8 * 64 * 4);
	(void)hipMalloc(&in, 256 * 4);
	uint32_t h[256];
	for(int i = 0; i < 256; i++) h[i] = 1000 + i * 7;
	(void)hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
	for(int w = 1; w <= 8; w++) { run<0>("one chain", out, in, w); run<1>("interleaved", out, in, w); }
	return 0;
}
