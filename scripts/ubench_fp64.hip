// scripts/ubench_fp64.hip -- SIMD cycles per wavefront instruction of the autocorrelation kernel's operations (v_fma_f64, v_mul_f64,
// v_add_f64, v_cvt_f64_f32, v_cvt_f32_i32, v_fma_f32), eight independent chains per wavefront, 1..5 wavefronts per SIMD.
// hipcc --offload-arch=gfx950 -O3 -o build/ubf scripts/ubench_fp64.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITER 4096
template <int OP>
__global__ __launch_bounds__(64) void k(double *out, const double *in, int n)
{
	const int t = threadIdx.x;
	double a[8], b = in[t & 31], c = in[(t + 1) & 31];
	float f[8];
	for(int u = 0; u < 8; u++) { a[u] = in[(t + u) & 31]; f[u] = (float)a[u]; }
	for(int it = 0; it < n; it++) {
#pragma unroll
		for(int u = 0; u < 8; u++) {
			if(OP == 0) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[u]) : "v"(b), "v"(c));
			if(OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[u]) : "v"(b));
			if(OP == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[u]) : "v"(b));
			if(OP == 3) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[u]) : "v"(f[u]));
			if(OP == 4) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(f[u]));
			if(OP == 5) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[u]) : "v"(f[(u + 1) & 7]));
			if(OP == 6) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[u]) : "v"(a[u]));
			if(OP == 7) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(a[u]) : "v"(b));
		}
	}
	double s = 0;
	for(int u = 0; u < 8; u++) s += a[u] + (double)f[u];
	out[blockIdx.x * 64 + t] = s;
}
template <int OP>
static void run(const char *name, double *out, const double *in, int wps)
{
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	const int blocks = 256 * 4 * wps;
	hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, out, in, 16);
	(void)hipDeviceSynchronize();
	(void)hipEventRecord(e0);
	hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, out, in, ITER);
	(void)hipEventRecord(e1);
	(void)hipEventSynchronize(e1);
	float ms;
	(void)hipEventElapsedTime(&ms, e0, e1);
	const double inst_per_wave = (double)ITER * 8;
	printf("%-14s %d waves/SIMD: %8.3f ms   %.2f SIMD-cycles per wave-instruction (@2.4GHz)\n", name, wps, ms, ms * 1e-3 * 2.4e9 / inst_per_wave / wps);
}
int main()
{
	double *out, *in;
	(void)hipMalloc(&out, 256 * 4 * 8 * 64 * 8);
	(void)hipMalloc(&in, 32 * 8);
	double h[32];
	for(int i = 0; i < 32; i++) h[i] = 1.0 + i * 1e-9;
	(void)hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
	for(int w = 1; w <= 5; w += 2) {
		run<0>("v_fma_f64", out, in, w); run<1>("v_mul_f64", out, in, w); run<2>("v_add_f64", out, in, w); run<3>("v_cvt_f64_f32", out, in, w);
		run<4>("v_cvt_f32_i32", out, in, w); run<5>("v_fma_f32", out, in, w); run<6>("v_cvt_f32_f64", out, in, w); run<7>("v_pk_fma_f32", out, in, w);
	}
	return 0;
}
