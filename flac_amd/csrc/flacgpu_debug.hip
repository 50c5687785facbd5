// flac_amd/csrc/flacgpu_debug.hip -- test hooks of the C ABI (include/flacgpu.h: flacgpu_debug_*): device-side
// evaluation of the libm-dependent expressions of the model search, so that a test can pin them against the host libm
// the reference binary links (tests/test_log_pin.py).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "flacgpu.h"
#include "flacgpu_dev.h"
#include "flacgpu_devfn.h"

namespace flacgpu {
__global__ __launch_bounds__(TPB) void log_kat_kernel(uint32_t mode, const double *__restrict__ a, const double *__restrict__ b, size_t n, double *__restrict__ out)
{
	const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
	if(i >= n) return;
	double v;
	if(mode == 0) v = flacgpu_log(a[i]);                                              // the restated glibc log itself
	else if(mode == 1) v = expected_bits_scaled(a[i], b[i]);                          // lpc.c:1591-1606 as compiled
	else if(mode == 2) v = (double)fixed_rbps((uint64_t)a[i], (uint32_t)b[i]);        // fixed.c:284-288 as compiled
	else v = log(a[i]);                                                               // the device library's log (informational)
	out[i] = v;
}
}

extern "C" int flacgpu_debug_log_kat(int device, uint32_t mode, const double *a, const double *b, size_t n, double *out)
{
	using namespace flacgpu;
	if(!a || !out || n == 0 || mode > 3 || ((mode == 1 || mode == 2) && !b)) return FLACGPU_ERR_BAD_ARG;
	int ndev = 0;
	if(hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev || hipSetDevice(device) != hipSuccess) return FLACGPU_ERR_NO_DEVICE;
	double *da = nullptr, *db = nullptr, *dout = nullptr;
	int rc = FLACGPU_OK;
	if(hipMalloc(&da, n * 8) != hipSuccess || hipMalloc(&dout, n * 8) != hipSuccess || (b && hipMalloc(&db, n * 8) != hipSuccess)) rc = FLACGPU_ERR_ALLOC;
	if(rc == FLACGPU_OK && (hipMemcpy(da, a, n * 8, hipMemcpyHostToDevice) != hipSuccess || (b && hipMemcpy(db, b, n * 8, hipMemcpyHostToDevice) != hipSuccess))) rc = FLACGPU_ERR_LAUNCH;
	if(rc == FLACGPU_OK) {
		hipLaunchKernelGGL(log_kat_kernel, dim3((unsigned)((n + TPB - 1) / TPB)), dim3(TPB), 0, 0, mode, da, db, n, dout);
		if(hipGetLastError() != hipSuccess || hipMemcpy(out, dout, n * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = FLACGPU_ERR_LAUNCH;
	}
	if(da) (void)hipFree(da);
	if(db) (void)hipFree(db);
	if(dout) (void)hipFree(dout);
	return rc;
}

// ---- the engine clock the chip really holds ------------------------------------------------------------------
// One wavefront on a side stream, beside whatever the chip is doing: per sample it reads the constant 100 MHz counter
// (s_memrealtime) and the shader-side counter (s_memtime), sleeps through 16 x s_sleep 127 -- 16 x 127 x 64 = 130 048 cycles of the
// clock the SIMD runs on, by the ISA manual's definition of s_sleep -- and reads both again.  130 048 / (real time of the sleep) is the
// clock; the s_memtime delta is kept next to it (on parts where that counter ticks with the shader clock the two agree; where it is a
// constant-frequency counter, too, the sleep is the measurement).  A sample takes ~55-65 us: ~16 k samples per second.
namespace flacgpu {
__global__ __launch_bounds__(64) void clock_probe_kernel(uint64_t *__restrict__ out, uint32_t nsamples)
{
	if(threadIdx.x != 0) return;
	for(uint32_t i = 0; i < nsamples; i++) {
		const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
		const uint64_t c0 = __builtin_readcyclecounter();
#pragma unroll
		for(int k = 0; k < 16; k++) __builtin_amdgcn_s_sleep(127);
		const uint64_t t1 = __builtin_amdgcn_s_memrealtime();
		const uint64_t c1 = __builtin_readcyclecounter();
		out[4 * (size_t)i + 0] = t0; out[4 * (size_t)i + 1] = t1 - t0; out[4 * (size_t)i + 2] = c1 - c0; out[4 * (size_t)i + 3] = 16ull * 127ull * 64ull;
	}
}
}
// d_out: device buffer of nsamples x 4 uint64 {start (100 MHz ticks), 100 MHz ticks of the sleep, s_memtime ticks of the sleep, nominal
// cycles of the sleep}.  Asynchronous on `stream` (give it a stream of its own: the probe runs BESIDE the work it observes).
extern "C" int flacgpu_debug_clock_probe(int device, void *stream, uint32_t nsamples, uint64_t *d_out)
{
	using namespace flacgpu;
	if(!d_out || nsamples == 0) return FLACGPU_ERR_BAD_ARG;
	if(hipSetDevice(device) != hipSuccess) return FLACGPU_ERR_NO_DEVICE;
	hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d_out, nsamples);
	return hipGetLastError() == hipSuccess ? FLACGPU_OK : FLACGPU_ERR_LAUNCH;
}
