// scripts/ubench_valu.hip -- issue-rate microbenchmarks of the VALU ops the analysis kernels are built from
// (development aid; numbers are quoted in DESIGN.md).  hipcc --offload-arch=gfx950 -O3 -o /tmp/ub scripts/ubench_valu.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define ITER 4096
#define UNR 16

template <int OP>
__global__ __launch_bounds__(256) void k(double *out, const float *in, int n)
{
	const int t = threadIdx.x;
	float f[UNR];
	double d[UNR];
	int32_t a[UNR];
#pragma unroll
	for(int u = 0; u < UNR; u++) { f[u] = in[(t + u) & 255]; d[u] = (double)in[(t + 2 * u) & 255]; a[u] = (int32_t)(in[(t + 3 * u) & 255] * 1000.f); }
	double acc[UNR];
#pragma unroll
	for(int u = 0; u < UNR; u++) acc[u] = 0.0;
	int32_t iacc[UNR];
#pragma unroll
	for(int u = 0; u < UNR; u++) iacc[u] = u;
	for(int it = 0; it < n; it++) {
#pragma unroll
		for(int u = 0; u < UNR; u++) {
			if(OP == 0) { acc[u] = fma(d[u], d[(u + 1) % UNR], acc[u]); }                       // v_fma_f64
			else if(OP == 1) { acc[u] = acc[u] + d[u]; }                                          // v_add_f64
			else if(OP == 2) { acc[u] = acc[u] * d[u]; }                                          // v_mul_f64
			else if(OP == 3) { f[u] = f[u] * 1.0001f + 0.5f; acc[u] += (double)f[u]; }            // cvt + add (+1 f32 fma)
			else if(OP == 4) { iacc[u] = __mul24(iacc[u], a[u]) + a[(u + 1) % UNR]; }             // v_mad_i32_i24
			else if(OP == 5) { iacc[u] = iacc[u] * a[u] + a[(u + 1) % UNR]; }                      // v_mad_u32 / mul_lo + add
			else if(OP == 6) {                                                                    // v_dot2_i32_i16
				typedef short s2 __attribute__((ext_vector_type(2)));
				s2 x, y; x = __builtin_bit_cast(s2, a[u]); y = __builtin_bit_cast(s2, a[(u + 1) % UNR]);
				iacc[u] = __builtin_amdgcn_sdot2(x, y, iacc[u], false);
			}
			else if(OP == 7) { uint32_t r_; asm("v_sad_u32 %0, %1, %2, %3" : "=v"(r_) : "v"(iacc[u]), "v"(a[u]), "v"(a[(u + 1) % UNR])); iacc[u] = (int32_t)r_; }   // v_sad_u32
			else if(OP == 8) { f[u] = f[u] * 1.0001f + 0.5f; }                                     // v_fma_f32 (baseline of OP 3)
			else if(OP == 9) { int64_t w = (int64_t)iacc[u] * (int64_t)a[u]; iacc[u] = (int32_t)(w >> 7) + 1; }  // v_mad_i64_i32 / mul_hi
		}
	}
	double r = 0;
#pragma unroll
	for(int u = 0; u < UNR; u++) r += acc[u] + (double)iacc[u] + (double)f[u];
	out[blockIdx.x * 256 + t] = r;
}

template <int OP>
static void run(const char *name, double *out, const float *in, double ops_per_iter)
{
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	const int blocks = 256 * 8;     // 8 workgroups (32 waves) per CU: full occupancy
	hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, in, 16);
	hipDeviceSynchronize();
	hipEventRecord(e0);
	hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, in, ITER);
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	float ms;
	hipEventElapsedTime(&ms, e0, e1);
	const double winst = (double)blocks * 4 * ITER * UNR * ops_per_iter;   // wave-instructions of the op under test
	// cycles per wave-instruction per SIMD at an assumed 2.4 GHz, 1024 SIMDs
	const double cyc = ms * 1e-3 * 2.4e9 * 1024 / winst;
	printf("%-28s %8.3f ms   %.2f SIMD-cycles per wave-instruction-group (@2.4GHz)\n", name, ms, cyc);
}

int main()
{
	double *out; float *in;
	hipMalloc(&out, 256 * 8 * 256 * sizeof(double));
	hipMalloc(&in, 256 * sizeof(float));
	float h[256];
	for(int i = 0; i < 256; i++) h[i] = 1.0f + i * 1e-3f;
	hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
	run<0>("v_fma_f64", out, in, 1);
	run<1>("v_add_f64", out, in, 1);
	run<2>("v_mul_f64", out, in, 1);
	run<8>("v_fma_f32", out, in, 1);
	run<3>("fma_f32+cvt_f64_f32+add_f64", out, in, 1);
	run<4>("v_mad_i32_i24", out, in, 1);
	run<5>("mul_lo_u32+add", out, in, 1);
	run<6>("v_dot2_i32_i16", out, in, 1);
	run<7>("v_sad_u32", out, in, 1);
	run<9>("mul i64 (mad_i64_i32)", out, in, 1);
	return 0;
}
