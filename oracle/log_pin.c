/* oracle/log_pin.c -- TEST INFRASTRUCTURE ONLY.
 * The product's restatement of glibc's log (flac_amd/csrc/flacgpu_log.h, the function the kernels call on the device)
 * compiled for the host, next to the libm of the box, so that tests/test_log_pin.py can compare the two bit for bit.
 * Built with -ffp-contract=off: every operation of the restatement is the single IEEE operation written there. */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include "../flac_amd/csrc/flacgpu_log.h"

#define BODY \
	size_t bad = 0; \
	for(size_t i = 0; i < n; i++) { \
		const double a = flacgpu_log(in[i]), b = log(in[i]); \
		uint64_t ua, ub; memcpy(&ua, &a, 8); memcpy(&ub, &b, 8); \
		if(out) out[i] = a; \
		if(ua != ub && !(a != a && b != b)) { if(!bad && first_bad) *first_bad = in[i]; bad++; } \
	} \
	return bad;

__attribute__((target("fma"))) static size_t run_fma(const double *in, size_t n, double *out, double *first_bad) { BODY }
static size_t run_plain(const double *in, size_t n, double *out, double *first_bad) { BODY }

/* number of arguments on which the restatement and libm's log differ; out (may be NULL) receives the restatement's values */
size_t logpin_compare(const double *in, size_t n, double *out, double *first_bad)
{
	return __builtin_cpu_supports("fma") ? run_fma(in, n, out, first_bad) : run_plain(in, n, out, first_bad);
}
void logpin_libm(const double *in, size_t n, double *out) { for(size_t i = 0; i < n; i++) out[i] = log(in[i]); }
/* the two expressions as the reference binary evaluates them (oracle/flac_oracle.c:280-291,374-377) through libm */
void logpin_expected_bits(const double *err, const double *scale, size_t n, double *out)
{
	for(size_t i = 0; i < n; i++) {
		double v = 0.0;
		if(err[i] > 0.0) { const double bps = log(scale[i] * err[i]) * 0.7213475204444817; v = bps >= 0.0 ? bps : 0.0; }
		else if(err[i] < 0.0) v = 1e32;
		out[i] = v;
	}
}
void logpin_fixed_rbps(const uint64_t *e, const uint32_t *n4, size_t n, float *out)
{
	for(size_t i = 0; i < n; i++) out[i] = e[i] ? (float)(log(((double)e[i] * 0.69314718055994530942) / (double)n4[i]) * 1.4426950408889634) : 0.0f;
}
int logpin_host_has_fma(void) { return __builtin_cpu_supports("fma") && __builtin_cpu_supports("avx2"); }
