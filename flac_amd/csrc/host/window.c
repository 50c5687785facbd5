/* flac_amd/csrc/host/window.c -- apodization window tables, computed once per encoder on the host
 * and uploaded to the GPU engine (init-time work, stays host C).
 *
 * States the same formulas as the reference's src/libFLAC/window.c:50-300 (float arithmetic with the
 * argument of cosf formed in double where the reference's M_PI promotes it), so that
 * out[i] = (float)x[i] * w[i] on the device sees bit-identical tables.
 */
#include <math.h>
#include <stdint.h>
#include "flacgpu_host.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

static void w_rectangle(float *w, int32_t L) { for(int32_t n = 0; n < L; n++) w[n] = 1.0f; }

static void w_bartlett(float *w, int32_t L) /* window.c:50 */
{
	const int32_t N = L - 1;
	int32_t n, half = (L & 1) ? N / 2 : L / 2 - 1;
	for(n = 0; n <= half; n++) w[n] = 2.0f * n / (float)N;
	for(; n <= N; n++) w[n] = 2.0f - 2.0f * n / (float)N;
}
static void w_bartlett_hann(float *w, int32_t L) /* window.c:70 */
{
	const int32_t N = L - 1;
	for(int32_t n = 0; n < L; n++)
		w[n] = (float)((0.62f - 0.38f * cosf(2.0f * M_PI * ((float)n / (float)N))) - 0.48f * fabsf((float)n / (float)N - 0.5f));   /* as regrouped by the reference's build, see below */
}
/* generalised cosine-sum windows: a0 - a1 cos(2 pi n/N) + a2 cos(4 pi n/N) - a3 cos(6 pi n/N) + a4 cos(8 pi n/N).
 * The reference is built with -fassociative-math (configure.ac / CMakeLists.txt), and its compiler regroups the terms:
 * positive and negative ones are summed separately -- (a0 + a2 c2) - (a1 c1 + a3 c3), and for the five terms of flattop
 * a0 + ((a2 c2 + a4 c4) - (a1 c1 + a3 c3)).  This file is compiled strictly, so the grouping is written out; the tables
 * are compared bit for bit with the reference's in tests/test_oracle_vs_ref.py::test_host_window_tables_bit_exact. */
static void w_blackman(float *w, int32_t L) /* window.c:79 */
{
	const int32_t N = L - 1;
	for(int32_t n = 0; n < L; n++) w[n] = (float)((0.42f + 0.08f * cosf(4.0f * M_PI * n / N)) - 0.5f * cosf(2.0f * M_PI * n / N));
}
static void w_blackman_harris(float *w, int32_t L) /* window.c:89 */
{
	const int32_t N = L - 1;
	for(int32_t n = 0; n <= N; n++)
		w[n] = (float)((0.35875f + 0.14128f * cosf(4.0f * M_PI * n / N)) - (0.48829f * cosf(2.0f * M_PI * n / N) + 0.01168f * cosf(6.0f * M_PI * n / N)));
}
static void w_connes(float *w, int32_t L) /* window.c:98 */
{
	const int32_t N = L - 1;
	const double N2 = (double)N / 2.;
	for(int32_t n = 0; n <= N; n++) {
		double k = ((double)n - N2) / N2;
		k = 1.0f - k * k;
		w[n] = (float)(k * k);
	}
}
static void w_flattop(float *w, int32_t L) /* window.c:111 */
{
	const int32_t N = L - 1;
	for(int32_t n = 0; n < L; n++)
		w[n] = (float)(0.21557895f + ((0.277263158f * cosf(4.0f * M_PI * n / N) + 0.006947368f * cosf(8.0f * M_PI * n / N)) - (0.41663158f * cosf(2.0f * M_PI * n / N) + 0.083578947f * cosf(6.0f * M_PI * n / N))));
}
static void w_gauss(float *w, int32_t L, float stddev) /* window.c:120 */
{
	const int32_t N = L - 1;
	const double N2 = (double)N / 2.;
	if(!(stddev > 0.0f && stddev <= 0.5f)) { w_gauss(w, L, 0.25f); return; }
	for(int32_t n = 0; n <= N; n++) {
		const double k = ((double)n - N2) / (stddev * N2);
		w[n] = (float)exp(-0.5f * k * k);
	}
}
static void w_hamming(float *w, int32_t L) /* window.c:138 */
{
	const int32_t N = L - 1;
	for(int32_t n = 0; n < L; n++) w[n] = (float)(0.54f - 0.46f * cosf(2.0f * M_PI * n / N));
}
static void w_hann(float *w, int32_t L) /* window.c:147 */
{
	const int32_t N = L - 1;
	for(int32_t n = 0; n < L; n++) w[n] = (float)(0.5f - 0.5f * cosf(2.0f * M_PI * n / N));
}
static void w_kaiser_bessel(float *w, int32_t L) /* window.c:156 */
{
	const int32_t N = L - 1;
	for(int32_t n = 0; n < L; n++)
		w[n] = (float)((0.402f + 0.098f * cosf(4.0f * M_PI * n / N)) - (0.498f * cosf(2.0f * M_PI * n / N) + 0.001f * cosf(6.0f * M_PI * n / N)));
}
static void w_nuttall(float *w, int32_t L) /* window.c:165 */
{
	const int32_t N = L - 1;
	for(int32_t n = 0; n < L; n++)
		w[n] = (float)((0.3635819f + 0.1365995f * cosf(4.0f * M_PI * n / N)) - (0.4891775f * cosf(2.0f * M_PI * n / N) + 0.0106411f * cosf(6.0f * M_PI * n / N)));
}
static void w_triangle(float *w, int32_t L) /* window.c:182 */
{
	int32_t n, half = (L & 1) ? (L + 1) / 2 : L / 2;
	for(n = 1; n <= half; n++) w[n - 1] = 2.0f * n / ((float)L + 1.0f);
	for(; n <= L; n++) w[n - 1] = (float)(2 * (L - n + 1)) / ((float)L + 1.0f);
}
static void w_tukey(float *w, int32_t L, float p) /* window.c:199 */
{
	if(p <= 0.0) w_rectangle(w, L);
	else if(p >= 1.0) w_hann(w, L);
	else if(!(p > 0.0f && p < 1.0f)) w_tukey(w, L, 0.5f);
	else {
		const int32_t Np = (int32_t)(p / 2.0f * L) - 1;
		w_rectangle(w, L);
		if(Np > 0)
			for(int32_t n = 0; n <= Np; n++) {
				w[n] = (float)(0.5f - 0.5f * cosf(M_PI * n / Np));
				w[L - Np - 1 + n] = (float)(0.5f - 0.5f * cosf(M_PI * (n + Np) / Np));
			}
	}
}
static void w_partial_tukey(float *w, int32_t L, float p, float start, float end) /* window.c:224 */
{
	const int32_t start_n = (int32_t)(start * L), end_n = (int32_t)(end * L), N = end_n - start_n;
	int32_t Np, n, i;
	if(p <= 0.0f) { w_partial_tukey(w, L, 0.05f, start, end); return; }
	if(p >= 1.0f) { w_partial_tukey(w, L, 0.95f, start, end); return; }
	if(!(p > 0.0f && p < 1.0f)) { w_partial_tukey(w, L, 0.5f, start, end); return; }
	Np = (int32_t)(p / 2.0f * N);
	for(n = 0; n < start_n && n < L; n++) w[n] = 0.0f;
	for(i = 1; n < (start_n + Np) && n < L; n++, i++) w[n] = (float)(0.5f - 0.5f * cosf(M_PI * i / Np));
	for(; n < (end_n - Np) && n < L; n++) w[n] = 1.0f;
	for(i = Np; n < end_n && n < L; n++, i--) w[n] = (float)(0.5f - 0.5f * cosf(M_PI * i / Np));
	for(; n < L; n++) w[n] = 0.0f;
}
static void w_punchout_tukey(float *w, int32_t L, float p, float start, float end) /* window.c:256 */
{
	const int32_t start_n = (int32_t)(start * L), end_n = (int32_t)(end * L);
	int32_t Ns, Ne, n, i;
	if(p <= 0.0f) { w_punchout_tukey(w, L, 0.05f, start, end); return; }
	if(p >= 1.0f) { w_punchout_tukey(w, L, 0.95f, start, end); return; }
	if(!(p > 0.0f && p < 1.0f)) { w_punchout_tukey(w, L, 0.5f, start, end); return; }
	Ns = (int32_t)(p / 2.0f * start_n);
	Ne = (int32_t)(p / 2.0f * (L - end_n));
	for(n = 0, i = 1; n < Ns && n < L; n++, i++) w[n] = (float)(0.5f - 0.5f * cosf(M_PI * i / Ns));
	for(; n < start_n - Ns && n < L; n++) w[n] = 1.0f;
	for(i = Ns; n < start_n && n < L; n++, i--) w[n] = (float)(0.5f - 0.5f * cosf(M_PI * i / Ns));
	for(; n < end_n && n < L; n++) w[n] = 0.0f;
	for(i = 1; n < end_n + Ne && n < L; n++, i++) w[n] = (float)(0.5f - 0.5f * cosf(M_PI * i / Ne));
	for(; n < L - (Ne) && n < L; n++) w[n] = 1.0f;
	for(i = Ne; n < L; n++, i--) w[n] = (float)(0.5f - 0.5f * cosf(M_PI * i / Ne));
}
static void w_welch(float *w, int32_t L) /* window.c:292 */
{
	const int32_t N = L - 1;
	const double N2 = (double)N / 2.;
	for(int32_t n = 0; n <= N; n++) {
		const double k = ((double)n - N2) / N2;
		w[n] = (float)(1.0f - k * k);
	}
}

/* the switch of resize_buffers_ (stream_encoder.c:2913-2977) */
void flacgpu_host_window(const flacgpu_host_apodization *a, float *w, int32_t L)
{
	switch(a->type) {
		case FGH_APOD_BARTLETT: w_bartlett(w, L); break;
		case FGH_APOD_BARTLETT_HANN: w_bartlett_hann(w, L); break;
		case FGH_APOD_BLACKMAN: w_blackman(w, L); break;
		case FGH_APOD_BLACKMAN_HARRIS_4TERM_92DB_SIDELOBE: w_blackman_harris(w, L); break;
		case FGH_APOD_CONNES: w_connes(w, L); break;
		case FGH_APOD_FLATTOP: w_flattop(w, L); break;
		case FGH_APOD_GAUSS: w_gauss(w, L, a->p); break;
		case FGH_APOD_HAMMING: w_hamming(w, L); break;
		case FGH_APOD_HANN: w_hann(w, L); break;
		case FGH_APOD_KAISER_BESSEL: w_kaiser_bessel(w, L); break;
		case FGH_APOD_NUTTALL: w_nuttall(w, L); break;
		case FGH_APOD_RECTANGLE: w_rectangle(w, L); break;
		case FGH_APOD_TRIANGLE: w_triangle(w, L); break;
		case FGH_APOD_TUKEY: w_tukey(w, L, a->p); break;
		case FGH_APOD_PARTIAL_TUKEY: w_partial_tukey(w, L, a->p, a->start, a->end); break;
		case FGH_APOD_PUNCHOUT_TUKEY: w_punchout_tukey(w, L, a->p, a->start, a->end); break;
		case FGH_APOD_SUBDIVIDE_TUKEY: w_tukey(w, L, a->p); break;
		case FGH_APOD_WELCH: w_welch(w, L); break;
		default: w_hann(w, L); break;
	}
}
