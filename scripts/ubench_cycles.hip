// scripts/ubench_cycles.hip -- shader cycles a SIMD spends per wavefront-instruction, per instruction class, measured INSIDE the
// kernel (round 2's ubench_valu.hip timed with HIP events and divided by an assumed 2.4 GHz).  Every SIMD of the chip gets W
// wavefronts (workgroups of four wavefronts, W per compute unit; HW_ID of every wavefront is recorded and the host checks that each
// of the 1024 SIMDs got exactly W) that run the same loop of N instructions of one class.  Three figures per class and W:
//   * SIMD: (last end - first start of the SIMD's wavefronts, on the chip-wide 100 MHz s_memrealtime) x the clock the loop ran at
//     (s_memtime ticks with the shader clock; its ratio to s_memrealtime inside the same loop) / (W x N) -- THE figure;
//   * events: the whole launch between two HIP events, same conversion -- agrees with it to a few per cent;
//   * own: a wavefront's own elapsed s_memtime / N -- the interval between ITS instructions.  NOT a SIMD rate: the wavefronts of a
//     SIMD do not all start together, and own / W understates the SIMD's cycles per instruction by up to 2x (this file's first
//     version reported that and concluded "2 cycles per v_dot2 at eight wavefronts": wrong, retracted in DESIGN.md section 7).
// the VALU roofline against per-class measured cycles.
//     hipcc --offload-arch=gfx950 -O3 -o /tmp/ubc scripts/ubench_cycles.hip && /tmp/ubc [json path]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

#define UNR 8
#define ITER 1024
#define REP 8

enum { ADD = 0, LSHR, XOR, MOV, ADD3, DOT2, SAD, PERM, ALIGNBIT, MAD24, MADLO, MAD64, FMA64, ADD64, MUL64, CVT_F64_I32, FMA32, DPP_ADD, READLANE, CNDMASK, BFE, MED3, DOT2_C1, DOT2_C2, DOT2_C4, DOT2_BLK7, ADD_C1, ADD_C2, FMA64_C1, FMA64_C2, FMA64_C4, DOT2_SGPR, DOT2_SGPR_C2, EVALG_LIKE, NOPS };
static const char *NAMES[] = {"v_add_u32", "v_lshrrev_b32", "v_xor_b32", "v_mov_b32", "v_add3_u32", "v_dot2_i32_i16", "v_sad_u32", "v_perm_b32", "v_alignbit_b32", "v_mad_i32_i24",
                              "v_mul_lo_u32", "v_mad_i64_i32", "v_fma_f64", "v_add_f64", "v_mul_f64", "v_cvt_f64_i32", "v_fma_f32", "v_add_u32 dpp", "v_readlane_b32", "v_cndmask_b32", "v_bfe_i32", "v_med3_i32", "dot2, 1 chain", "dot2, 2 chains", "dot2, 4 chains", "dot2 6+lshr blocks, 2 alternating", "v_add_u32, 1 chain", "v_add_u32, 2 chains", "fma_f64, 1 chain", "fma_f64, 2 chains", "fma_f64, 4 chains", "dot2, a tap from an SGPR, 8 chains", "dot2, a tap from an SGPR, 2 chains", "evalg-like: 2x(6 dot2 sgpr + lshr sgpr) + 2 sad sgpr"};

template <int OP>
__global__ __launch_bounds__(256, 8) void k(uint64_t *cyc, uint64_t *real, uint32_t *sink, int iters)
{
	uint32_t a[UNR], b[UNR];
	double d[UNR], e[UNR];
	uint64_t w[UNR];
#pragma unroll
	for(int u = 0; u < UNR; u++) w[u] = threadIdx.x * 0x9E3779B97F4A7C15ull + u;
#pragma unroll
	for(int u = 0; u < UNR; u++) { a[u] = threadIdx.x * 2654435761u + u * 40503u; b[u] = a[u] ^ 0x5bd1e995u; d[u] = 1.0 + 1e-9 * (threadIdx.x + u); e[u] = 0.999999 + 1e-10 * u; }
	__builtin_amdgcn_s_barrier();
	const uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
	for(int it = 0; it < iters; it++) {
		// (REP x UNR instructions per trip: a taken branch costs a wavefront tens of cycles, which eight instructions do not hide)
#pragma unroll
		for(int uu = 0; uu < REP * UNR; uu++) {
			const int u = uu % UNR, v = (u + 1) % UNR;
			if(OP == ADD) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[u]) : "v"(b[u]));
			if(OP == LSHR) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(a[u]));
			if(OP == XOR) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[u]) : "v"(b[u]));
			if(OP == MOV) asm volatile("v_mov_b32 %0, %1" : "+v"(a[u]) : "v"(b[v]));
			if(OP == ADD3) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[u]) : "v"(b[u]), "v"(b[v]));
			if(OP == DOT2) asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(a[u]) : "v"(b[u]), "v"(b[v]));
			if(OP == SAD) asm volatile("v_sad_u32 %0, %1, %2, %0" : "+v"(a[u]) : "v"(b[u]), "v"(b[v]));
			if(OP == PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[u]) : "v"(b[u]), "v"(b[v]));
			if(OP == ALIGNBIT) asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(a[u]) : "v"(b[u]), "v"(b[v]));
			if(OP == MAD24) asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(a[u]) : "v"(b[u]), "v"(b[v]));
			if(OP == MADLO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[u]) : "v"(b[u]));
			if(OP == MAD64) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(w[u]) : "v"(b[u]), "v"(b[v]) : "vcc");
			if(OP == FMA64) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[u]) : "v"(e[u]), "v"(e[v]));
			if(OP == ADD64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[u]) : "v"(e[u]));
			if(OP == MUL64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[u]) : "v"(e[u]));
			if(OP == CVT_F64_I32) asm volatile("v_cvt_f64_i32 %0, %1" : "+v"(d[u]) : "v"(a[u]));
			if(OP == FMA32) { float f = __uint_as_float(a[u]); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f) : "v"(b[u]), "v"(b[v])); a[u] = __float_as_uint(f); }
			if(OP == DPP_ADD) asm volatile("v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[u]) : "v"(b[u]));
			if(OP == READLANE) { uint32_t s_; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s_) : "v"(a[u])); b[u] ^= s_ & (it == -1); }
			if(OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[u]) : "v"(b[u]) : );
			if(OP == BFE) asm volatile("v_bfe_i32 %0, %0, 3, 17" : "+v"(a[u]));
			if(OP == MED3) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(a[u]) : "v"(b[u]), "v"(b[v]));
			// dependent chains: every instruction waits for the one before it in its chain (1, 2 or 4 chains side by side)
			if(OP == DOT2_C1) asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(a[0]) : "v"(b[u]), "v"(b[v]));
			if(OP == DOT2_C2) asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(a[u & 1]) : "v"(b[u]), "v"(b[v]));
			if(OP == DOT2_C4) asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(a[u & 3]) : "v"(b[u]), "v"(b[v]));
			// the evaluation kernel's shape: a block of six dependent dot2 and a shift per candidate, two candidates alternating
			// (8 instructions per block here: 6 dot2 + shift + sad, the sad on the other block's result)
			if(OP == DOT2_BLK7) {
				const int blk = (uu / 8) & 1, pos = uu % 8;
				if(pos < 6) asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(a[blk]) : "v"(b[u]), "v"(b[v]));
				else if(pos == 6) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(a[blk]));
				else asm volatile("v_sad_u32 %0, %1, %2, %0" : "+v"(a[2 + blk]) : "v"(a[blk]), "v"(b[v]));
			}
			if(OP == DOT2_SGPR) { uint32_t sq = (uint32_t)iters * 77u + (uint32_t)u; asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(a[u]) : "v"(b[u]), "s"(sq)); }
			if(OP == DOT2_SGPR_C2) { uint32_t sq = (uint32_t)iters * 77u + (uint32_t)u; asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(a[u & 1]) : "v"(b[u]), "s"(sq)); }
			if(OP == EVALG_LIKE) {
				// 16 instructions: two interleaved chains of six dot2 (tap in an SGPR) and a shift by an SGPR, then two v_sad_u32 against an SGPR into one sum
				const int pos = uu % 16;
				uint32_t sq = (uint32_t)iters * 77u + (uint32_t)(pos >> 1), sh = (uint32_t)iters & 7u, bias = (uint32_t)iters << 8;
				if(pos < 12) asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(a[pos & 1]) : "v"(b[u]), "s"(sq));
				else if(pos < 14) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(a[pos & 1]) : "s"(sh));
				else asm volatile("v_sad_u32 %0, %1, %2, %0" : "+v"(a[2]) : "v"(a[pos & 1]), "s"(bias));
			}
			if(OP == ADD_C1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[0]) : "v"(b[u]));
			if(OP == ADD_C2) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[u & 1]) : "v"(b[u]));
			if(OP == FMA64_C1) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[0]) : "v"(e[u]), "v"(e[v]));
			if(OP == FMA64_C2) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[u & 1]) : "v"(e[u]), "v"(e[v]));
			if(OP == FMA64_C4) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[u & 3]) : "v"(e[u]), "v"(e[v]));
		}
	}
	const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
	// (only what the loop of this OP touched is folded into the result: the other arrays are dead and take no registers -- all W
	//  wavefronts of a SIMD must be resident together, 8 of them: at most 64 registers)
	constexpr bool USES_D = OP == FMA64 || OP == ADD64 || OP == MUL64 || OP == CVT_F64_I32 || OP == FMA64_C1 || OP == FMA64_C2 || OP == FMA64_C4, USES_W = OP == MAD64;
	uint32_t x = 0;
#pragma unroll
	for(int u = 0; u < UNR; u++) {
		x ^= a[u] ^ b[u];
		if(USES_D) x ^= (uint32_t)(int64_t)d[u];
		if(USES_W) x ^= (uint32_t)w[u] ^ (uint32_t)(w[u] >> 32);
	}
	if((threadIdx.x & 63) == 0) {
		cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0; real[blockIdx.x * 4 + (threadIdx.x >> 6)] = r1 - r0;
		// absolute start and end on the 100 MHz counter (one counter for the chip): the host checks that the W wavefronts of a SIMD really
		// ran side by side -- a SIMD's rate from the span between its first start and its last end, next to the per-wavefront figure
		real[(1 << 16) + 2 * (blockIdx.x * 4 + (threadIdx.x >> 6))] = r0; real[(1 << 16) + 2 * (blockIdx.x * 4 + (threadIdx.x >> 6)) + 1] = r1;
		// where the wavefront ran: HW_ID (SIMD, compute unit, shader array, shader engine) and the XCC -- the host counts wavefronts per SIMD
		uint32_t hw, xcc;
		asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
		asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
		sink[1 + blockIdx.x * 4 + (threadIdx.x >> 6)] = (hw & 0xfff0u) | ((xcc & 0xfu) << 16);
	}
	if(x == 0x12345678u) sink[0] = x;
}

struct Res { double cyc_per_inst, mhz, waves_per_simd_seen; int simds_seen; double span_cyc_per_inst, event_cyc_per_inst; };
template <int OP>
static Res run(int waves_per_simd, uint64_t *d_cyc, uint64_t *d_real, uint32_t *d_sink)
{
	hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
	// workgroups of four wavefronts (one per SIMD of the compute unit that takes the workgroup), W workgroups per compute unit
	const int wgs = prop.multiProcessorCount * waves_per_simd, blocks = wgs * 4;
	hipLaunchKernelGGL(k<OP>, dim3(wgs), dim3(256), 0, 0, d_cyc, d_real, d_sink, 64);         // warm-up
	hipDeviceSynchronize();
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	hipEventRecord(e0);
	hipLaunchKernelGGL(k<OP>, dim3(wgs), dim3(256), 0, 0, d_cyc, d_real, d_sink, ITER);
	hipEventRecord(e1);
	hipDeviceSynchronize();
	float ev_ms = 0;
	hipEventElapsedTime(&ev_ms, e0, e1);
	hipEventDestroy(e0); hipEventDestroy(e1);
	std::vector<uint64_t> c(blocks), r(blocks);
	hipMemcpy(c.data(), d_cyc, blocks * 8, hipMemcpyDeviceToHost);
	hipMemcpy(r.data(), d_real, blocks * 8, hipMemcpyDeviceToHost);
	std::vector<double> per(blocks), mhz(blocks);
	for(int i = 0; i < blocks; i++) { per[i] = (double)c[i] / ((double)waves_per_simd * ITER * UNR * REP); mhz[i] = (double)c[i] / ((double)r[i] / 100.0); }
	std::sort(per.begin(), per.end()); std::sort(mhz.begin(), mhz.end());
	std::vector<uint32_t> hw(blocks);
	hipMemcpy(hw.data(), d_sink + 1, blocks * 4, hipMemcpyDeviceToHost);
	std::sort(hw.begin(), hw.end());
	int simds = 0, most = 0, run_ = 0;
	for(int i = 0; i < blocks; i++) { if(i == 0 || hw[i] != hw[i - 1]) { simds++; run_ = 0; } run_++; if(run_ > most) most = run_; }
	// per SIMD: (last end - first start) of its wavefronts on the 100 MHz counter, in shader cycles at the median clock, per instruction
	std::vector<uint64_t> ab(2 * blocks);
	hipMemcpy(ab.data(), d_real + (1 << 16), 2 * blocks * 8, hipMemcpyDeviceToHost);
	std::vector<uint32_t> hw2(blocks);
	hipMemcpy(hw2.data(), d_sink + 1, blocks * 4, hipMemcpyDeviceToHost);
	std::vector<std::pair<uint32_t, int>> order(blocks);
	for(int i = 0; i < blocks; i++) order[i] = {hw2[i], i};
	std::sort(order.begin(), order.end());
	std::vector<double> spans;
	for(int i = 0; i < blocks;) {
		int j = i; uint64_t lo = ~0ull, hi = 0; int n = 0;
		while(j < blocks && order[j].first == order[i].first) { const int w = order[j].second; if(ab[2 * w] < lo) lo = ab[2 * w]; if(ab[2 * w + 1] > hi) hi = ab[2 * w + 1]; n++; j++; }
		spans.push_back((double)(hi - lo) / 100.0 * mhz[blocks / 2] / ((double)n * ITER * UNR * REP));
		i = j;
	}
	std::sort(spans.begin(), spans.end());
	// ... and the whole launch between two HIP events (launch overhead included), at the median clock
	const double ev = (double)ev_ms * 1e3 * mhz[blocks / 2] / ((double)waves_per_simd * ITER * UNR * REP);
	return Res{per[blocks / 2], mhz[blocks / 2], (double)most, simds, spans[spans.size() / 2], ev};
}

int main(int argc, char **argv)
{
	uint64_t *d_cyc, *d_real; uint32_t *d_sink;
	hipMalloc(&d_cyc, 1 << 20); hipMalloc(&d_real, 1 << 21); hipMalloc(&d_sink, 1 << 20);
	FILE *js = argc > 1 ? fopen(argv[1], "w") : nullptr;
	if(js) fprintf(js, "{\"what\": \"shader cycles a SIMD spends per wavefront-instruction with W wavefronts running the same loop: from the first start to the last end of the SIMD's wavefronts (s_memrealtime) at the clock measured inside the loop (s_memtime / s_memrealtime), median over the 1024 SIMDs; events_w8: the launch between two HIP events; own_interval: cycles between ONE wavefront's own instructions\", \"classes\": {\n");
	printf("%-58s %34s | %20s | %20s | %4s  %s\n", "cycles per wavefront-instruction", "a SIMD, W = 1 2 4 5 8 (first start to last end)", "HIP events, W = 2 4 8", "a wavefront's own, W = 1 4 8", "MHz", "SIMDs seen, most wavefronts on one (W = 2 / 8)");
	bool first = true;
#define ROW(OP) do { Res r1 = run<OP>(1, d_cyc, d_real, d_sink), r2 = run<OP>(2, d_cyc, d_real, d_sink), r4 = run<OP>(4, d_cyc, d_real, d_sink), r5 = run<OP>(5, d_cyc, d_real, d_sink), r8 = run<OP>(8, d_cyc, d_real, d_sink); \
		printf("%-58s %6.2f %6.2f %6.2f %6.2f %6.2f | %6.2f %6.2f %6.2f | %6.2f %6.2f %6.2f | %4.0f  %d/%d %.0f/%.0f\n", NAMES[OP], r1.span_cyc_per_inst, r2.span_cyc_per_inst, r4.span_cyc_per_inst, r5.span_cyc_per_inst, r8.span_cyc_per_inst, \
		       r2.event_cyc_per_inst, r4.event_cyc_per_inst, r8.event_cyc_per_inst, r1.cyc_per_inst, 4 * r4.cyc_per_inst, 8 * r8.cyc_per_inst, r8.mhz, r2.simds_seen, r8.simds_seen, r2.waves_per_simd_seen, r8.waves_per_simd_seen); \
		if(js) { fprintf(js, "%s  \"%s\": {\"w1\": %.4f, \"w2\": %.4f, \"w4\": %.4f, \"w5\": %.4f, \"w8\": %.4f, \"events_w8\": %.4f, \"own_interval_w1\": %.4f, \"own_interval_w8\": %.4f, \"mhz_w8\": %.1f}", first ? "" : ",\n", NAMES[OP], \
		                 r1.span_cyc_per_inst, r2.span_cyc_per_inst, r4.span_cyc_per_inst, r5.span_cyc_per_inst, r8.span_cyc_per_inst, r8.event_cyc_per_inst, r1.cyc_per_inst, 8 * r8.cyc_per_inst, r8.mhz); first = false; } } while(0)
	ROW(ADD); ROW(LSHR); ROW(XOR); ROW(MOV); ROW(ADD3); ROW(DOT2); ROW(SAD); ROW(PERM); ROW(ALIGNBIT); ROW(MAD24); ROW(MADLO); ROW(MAD64);
	ROW(FMA64); ROW(ADD64); ROW(MUL64); ROW(CVT_F64_I32); ROW(FMA32); ROW(DPP_ADD); ROW(READLANE); ROW(CNDMASK); ROW(BFE); ROW(MED3);
	ROW(DOT2_C1); ROW(DOT2_C2); ROW(DOT2_C4); ROW(DOT2_BLK7); ROW(ADD_C1); ROW(ADD_C2); ROW(FMA64_C1); ROW(FMA64_C2); ROW(FMA64_C4); ROW(DOT2_SGPR); ROW(DOT2_SGPR_C2); ROW(EVALG_LIKE);
	if(js) { fprintf(js, "\n}}\n"); fclose(js); }
	return 0;
}
