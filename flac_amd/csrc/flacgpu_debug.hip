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
