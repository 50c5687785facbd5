/* flac_amd/csrc/flacgpu_log.h -- log() as the reference binary computes it on an x86-64 host with AVX2+FMA.
 *
 * Two comparisons of the model search go through libm's double-precision log:
 *     FLAC__lpc_compute_expected_bits_per_residual_sample   src/libFLAC/lpc.c:1594
 *     FLAC__fixed_compute_best_predictor*                   src/libFLAC/fixed.c:284-288,339-343
 * and a log that differs in the last place can flip a `bits < best_bits` tie.  libm is not part of
 * /root/reference: it is glibc 2.35 (Ubuntu GLIBC 2.35-0ubuntu3.11 in the image the reference is built in).  This file
 * restates glibc's algorithm (sysdeps/ieee754/dbl-64/e_log.c: Szabolcs Nagy's table-driven log, 128 subintervals,
 * log(x) = log1p(z/c - 1) + log(c) + k ln2; separate polynomial near 1) in the operation order of the variant glibc's
 * ifunc selects on an FMA+AVX2 host (sysdeps/x86_64/fpu/multiarch/e_log-fma.c, i.e. e_log.c compiled with -mfma:
 * the products feeding sums are contracted; the sequence below was read off the disassembly of __log_fma in the
 * image's libm.so.6).  Every operation is a single IEEE-754 binary64 operation (fma / mul / add / sub), so the result
 * is the same on any IEEE machine -- the device included: no dependence on the ROCm device library's log.
 * Constants: flacgpu_log_data.h (scripts/extract_glibc_log_data.py).
 *
 * Provenance and licence: the algorithm and its constants are the GNU C Library's (glibc 2.35, sysdeps/ieee754/dbl-64/
 * e_log.c and e_log_data.c, Copyright (C) 2018-2022 Free Software Foundation, Inc., contributed by Arm Ltd.), distributed
 * under the GNU Lesser General Public License, version 2.1 or later.  This restatement and the extracted table
 * (flacgpu_log_data.h) are a derived work of that code and are offered under the same terms (LGPL-2.1-or-later), whatever
 * licence the rest of this repository carries; the reference itself (libFLAC) is BSD-3-Clause and contains none of it.
 *
 * Pinned: tests/test_log_pin.py compares this function, compiled for the host, with the libm of the box on >= 10^8
 * arguments bit for bit (CPU), and the device instantiation with the same libm on >= 10^7 (GPU).
 *
 * Include with FLACGPU_LOG_FN defined to the function qualifiers (`__device__ __forceinline__` in the kernels,
 * `static inline` for the host pin test) and FLACGPU_LOG_TABQ to the qualifier of the constant tables. */
#ifndef FLACGPU_LOG_H
#define FLACGPU_LOG_H
#include <stdint.h>
#include <string.h>
#include "flacgpu_log_data.h"

#ifndef FLACGPU_LOG_FN
#define FLACGPU_LOG_FN static inline
#endif
#ifndef FLACGPU_LOG_TABQ
#define FLACGPU_LOG_TABQ static const
#endif

FLACGPU_LOG_TABQ uint64_t flacgpu_log_poly[5] = FLACGPU_LOG_POLY;       /* A[0..4] */
FLACGPU_LOG_TABQ uint64_t flacgpu_log_poly1[11] = FLACGPU_LOG_POLY1;    /* B[0..10] */
FLACGPU_LOG_TABQ uint64_t flacgpu_log_tab[256] = FLACGPU_LOG_TAB;       /* { invc, logc } x 128 */

FLACGPU_LOG_FN double flacgpu_log_asdouble(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
FLACGPU_LOG_FN uint64_t flacgpu_log_asuint(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }

/* tab: the { invc, logc } table, flacgpu_log_tab or a copy of it in faster memory (the model kernel keeps one in LDS) */
FLACGPU_LOG_FN double flacgpu_log_with(double x, const uint64_t *tab)
{
#define A(i) flacgpu_log_asdouble(flacgpu_log_poly[i])
#define B(i) flacgpu_log_asdouble(flacgpu_log_poly1[i])
	uint64_t ix = flacgpu_log_asuint(x);
	const uint32_t top = (uint32_t)(ix >> 48);
	/* 1 - 2^-4 <= x < 1 + 0x1.09p-4: log1p(r) by a degree-12 polynomial, the leading terms r - r^2/2 in double-double */
	if(ix - 0x3FEE000000000000ull < 0x0003090000000000ull) {
		if(ix == 0x3FF0000000000000ull) return 0.0;
		const double r = x - 1.0;
		const double a12 = __builtin_fma(r, B(2), B(1)), a45 = __builtin_fma(r, B(5), B(4)), a78 = __builtin_fma(r, B(8), B(7));
		const double r2 = r * r;
		const double a123 = __builtin_fma(r2, B(3), a12), a456 = __builtin_fma(r2, B(6), a45);
		const double r3 = r * r2;
		const double a789 = __builtin_fma(r2, B(9), a78);
		const double a7t = __builtin_fma(r3, B(10), a789);
		const double a4t = __builtin_fma(a7t, r3, a456);
		const double p = __builtin_fma(a4t, r3, a123);
		const double t1 = __builtin_fma(r, 0x1p27, r);
		const double rhi = __builtin_fma(-0x1p27, r, t1);
		const double rhi2 = rhi * rhi;
		const double rlo = r - rhi;
		const double hi = __builtin_fma(rhi2, B(0), r);
		const double d = r - hi;
		const double s = r + rhi;
		const double lo0 = __builtin_fma(rhi2, B(0), d);
		const double lo = __builtin_fma(B(0) * rlo, s, lo0);
		return hi + __builtin_fma(p, r3, lo);
	}
	if(top - 0x0010u >= 0x7ff0u - 0x0010u) {
		/* not a positive normal number */
		if((ix << 1) == 0) return -1.0 / 0.0;                                   /* log(+-0) = -inf */
		if(ix == 0x7FF0000000000000ull) return x;                                /* log(inf) = inf */
		if((top & 0x8000u) || (top & 0x7ff0u) == 0x7ff0u) return (x - x) / 0.0;   /* negative or NaN */
		ix = flacgpu_log_asuint(x * 0x1p52);                                      /* subnormal: normalise */
		ix -= 52ull << 52;
	}
	/* x = 2^k z, z in [0x1.6p-1, 0x1.6p0); c = centre of z's subinterval */
	const uint64_t tmp = ix - 0x3FE6000000000000ull;
	const uint32_t i = (uint32_t)(tmp >> 45) & 127u;
	const int32_t k = (int32_t)((int64_t)tmp >> 52);
	const double z = flacgpu_log_asdouble(ix - (tmp & 0xFFF0000000000000ull));
	const double invc = flacgpu_log_asdouble(tab[2 * i]), logc = flacgpu_log_asdouble(tab[2 * i + 1]);
	const double ln2hi = flacgpu_log_asdouble(FLACGPU_LOG_LN2HI), ln2lo = flacgpu_log_asdouble(FLACGPU_LOG_LN2LO);
	const double r = __builtin_fma(z, invc, -1.0);
	const double kd = (double)k;
	const double w = __builtin_fma(kd, ln2hi, logc);
	const double p12 = __builtin_fma(r, A(2), A(1));
	const double hi = r + w;
	const double r2 = r * r;
	const double lo = __builtin_fma(kd, ln2lo, (w - hi) + r);
	const double r3 = r * r2;
	const double p34 = __builtin_fma(r, A(4), A(3));
	const double t = __builtin_fma(r2, A(0), lo);
	const double q = __builtin_fma(p34, r2, p12);
	return __builtin_fma(r3, q, t) + hi;
#undef A
#undef B
}
FLACGPU_LOG_FN double flacgpu_log(double x) { return flacgpu_log_with(x, flacgpu_log_tab); }
#endif
