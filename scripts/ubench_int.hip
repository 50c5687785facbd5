// scripts/ubench_int.hip -- issue rates of single VALU instructions (inline asm, 16 independent chains per lane).
// hipcc --offload-arch=gfx950 -O3 -o build/ubi scripts/ubench_int.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITER 4096
#define UNR 16
#define OP3(name, str) struct name { static __device__ __forceinline__ void f(uint32_t &a, uint32_t b, uint32_t c) { asm volatile(str : "+v"(a) : "v"(b), "v"(c)); } };
OP3(AddU32, "v_add_u32 %0, %0, %1")
OP3(SubU32, "v_sub_u32 %0, %1, %0")
OP3(Ashr, "v_ashrrev_i32 %0, 3, %0")
OP3(MaxI32, "v_max_i32 %0, %0, %1")
OP3(BfeI32, "v_bfe_i32 %0, %0, 0, 16")
OP3(Alignbit, "v_alignbit_b32 %0, %0, %1, 16")
OP3(Add3, "v_add3_u32 %0, %0, %1, %2")
OP3(MulLo, "v_mul_lo_u32 %0, %0, %1")
OP3(MadU24, "v_mad_u32_u24 %0, %0, %1, %2")
OP3(MadI24, "v_mad_i32_i24 %0, %0, %1, %2")
OP3(CvtF32I32, "v_cvt_f32_i32 %0, %0")
OP3(MulF32, "v_mul_f32 %0, %0, %1")
OP3(Xor, "v_xor_b32 %0, %0, %1")
OP3(LshlOr, "v_lshl_or_b32 %0, %0, 1, %1")
OP3(LshlAdd, "v_lshl_add_u32 %0, %0, 1, %1")
OP3(Perm, "v_perm_b32 %0, %0, %1, %2")
OP3(Dot2, "v_dot2_i32_i16 %0, %1, %2, %0")
OP3(Dot4, "v_dot4_i32_i8 %0, %1, %2, %0")
OP3(PkMad16, "v_pk_mad_i16 %0, %0, %1, %2")
OP3(PkAdd16, "v_pk_add_i16 %0, %0, %1")
OP3(SadU32, "v_sad_u32 %0, %0, %1, %2")
OP3(MulHi, "v_mul_hi_i32 %0, %0, %1")
OP3(Mov, "v_mov_b32 %0, %1")
OP3(Cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
OP3(AddCo, "v_add_co_u32 %0, vcc, %0, %1")
OP3(Ffbh, "v_ffbh_u32 %0, %0")
OP3(Lshrrev, "v_lshrrev_b32 %0, %1, %0")
OP3(Max3, "v_max3_i32 %0, %0, %1, %2")
OP3(Med3, "v_med3_i32 %0, %0, %1, %2")
OP3(MadU64, "v_mad_u64_u32 %0, vcc, %1, %2, 0")   // dummy: handled separately
template <class O>
__global__ __launch_bounds__(256) void k(uint32_t *out, const uint32_t *in, int n)
{
	const int t = threadIdx.x;
	uint32_t a[UNR];
#pragma unroll
	for(int u = 0; u < UNR; u++) a[u] = in[(t + u) & 255];
	const uint32_t b = in[(t + 77) & 255], c = in[(t + 99) & 255];
	for(int it = 0; it < n; it++) {
#pragma unroll
		for(int u = 0; u < UNR; u++) O::f(a[u], b, c);
	}
	uint32_t r = 0;
#pragma unroll
	for(int u = 0; u < UNR; u++) r += a[u];
	out[blockIdx.x * 256 + t] = r;
}
struct Mad64 {};
template <>
__global__ __launch_bounds__(256) void k<Mad64>(uint32_t *out, const uint32_t *in, int n)
{
	const int t = threadIdx.x;
	uint64_t a[UNR];
#pragma unroll
	for(int u = 0; u < UNR; u++) a[u] = in[(t + u) & 255];
	const uint32_t b = in[(t + 77) & 255], c = in[(t + 99) & 255];
	for(int it = 0; it < n; it++) {
#pragma unroll
		for(int u = 0; u < UNR; u++) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(a[u]) : "v"(b), "v"(c) : "vcc");
	}
	uint64_t r = 0;
#pragma unroll
	for(int u = 0; u < UNR; u++) r += a[u];
	out[blockIdx.x * 256 + t] = (uint32_t)r;
}
struct CvtF64 {};
template <>
__global__ __launch_bounds__(256) void k<CvtF64>(uint32_t *out, const uint32_t *in, int n)
{
	const int t = threadIdx.x;
	double a[UNR]; float f[UNR];
#pragma unroll
	for(int u = 0; u < UNR; u++) { f[u] = (float)in[(t + u) & 255]; a[u] = 0; }
	for(int it = 0; it < n; it++) {
#pragma unroll
		for(int u = 0; u < UNR; u++) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[u]) : "v"(f[u]));
	}
	double r = 0;
#pragma unroll
	for(int u = 0; u < UNR; u++) r += a[u];
	out[blockIdx.x * 256 + t] = (uint32_t)r;
}
#define F64OP(name, str) struct name {}; \
template <> __global__ __launch_bounds__(256) void k<name>(uint32_t *out, const uint32_t *in, int n) \
{ \
	const int t = threadIdx.x; \
	double a[UNR]; \
	for(int u = 0; u < UNR; u++) a[u] = 1.0 + 1e-9 * (double)in[(t + u) & 255]; \
	const double b = 1.0 + 1e-12 * (double)in[(t + 77) & 255], c = 1e-12 * (double)in[(t + 99) & 255]; \
	for(int it = 0; it < n; it++) { \
		_Pragma("unroll") for(int u = 0; u < UNR; u++) asm volatile(str : "+v"(a[u]) : "v"(b), "v"(c)); \
	} \
	double r = 0; \
	for(int u = 0; u < UNR; u++) r += a[u]; \
	out[blockIdx.x * 256 + t] = (uint32_t)r; \
}
F64OP(FmaF64, "v_fma_f64 %0, %0, %1, %2")
F64OP(MulF64, "v_mul_f64 %0, %0, %1")
F64OP(AddF64, "v_add_f64 %0, %0, %1")
template <class O>
static void run(const char *name, uint32_t *out, const uint32_t *in)
{
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	const int blocks = 256 * 8;
	hipLaunchKernelGGL(k<O>, dim3(blocks), dim3(256), 0, 0, out, in, 16);
	(void)hipDeviceSynchronize();
	(void)hipEventRecord(e0);
	hipLaunchKernelGGL(k<O>, dim3(blocks), dim3(256), 0, 0, out, in, ITER);
	(void)hipEventRecord(e1);
	(void)hipEventSynchronize(e1);
	float ms;
	(void)hipEventElapsedTime(&ms, e0, e1);
	const double winst = (double)blocks * 4 * ITER * UNR;
	printf("%-16s %8.3f ms   %.2f SIMD-cycles per wave-instruction (@2.4GHz)\n", name, ms, ms * 1e-3 * 2.4e9 * 1024 / winst);
}
#define R(x) run<x>(#x, out, in)
int main()
{
	uint32_t *out, *in;
	(void)hipMalloc(&out, 256 * 8 * 256 * 4);
	(void)hipMalloc(&in, 256 * 4);
	uint32_t h[256];
	for(int i = 0; i < 256; i++) h[i] = 1000 + i * 7;
	(void)hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
	R(AddU32); R(SubU32); R(Ashr); R(MaxI32); R(BfeI32); R(Alignbit); R(Add3); R(MulLo); R(MadU24); R(MadI24); R(CvtF32I32); R(MulF32); R(Xor);
	R(LshlOr); R(LshlAdd); R(Perm); R(Dot2); R(Dot4); R(PkMad16); R(PkAdd16); R(SadU32); R(MulHi); R(Mov); R(Cndmask); R(AddCo); R(Ffbh); R(Lshrrev); R(Max3); R(Med3);
	R(Mad64); R(CvtF64); R(FmaF64); R(MulF64); R(AddF64);
	return 0;
}
