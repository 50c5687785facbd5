/* scripts/api_rate.c -- end-to-end rate of the drop-in encoder API from a C caller: PCM in host memory through
 * FLAC__stream_encoder_process_interleaved, frames to a write callback.  The same source runs against the reference
 * library when built with -DUSE_REF against oracle/_ref (see scripts/api_rate.sh).
 *   cc -O2 -Iinclude scripts/api_rate.c -o build/api_rate -Lflac_amd/lib -lFLACgpu -lm -Wl,-rpath,'$ORIGIN/../flac_amd/lib' */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef USE_REF
#include "FLAC/stream_encoder.h"
#else
#include "FLACgpu_stream_encoder.h"
#endif

static size_t g_bytes, g_frames;
static FLAC__StreamEncoderWriteStatus wcb(const FLAC__StreamEncoder *e, const FLAC__byte b[], size_t n, uint32_t samples, uint32_t frame, void *cd)
{
	(void)e; (void)b; (void)frame; (void)cd;
	g_bytes += n; if(samples) g_frames++;
	return FLAC__STREAM_ENCODER_WRITE_STATUS_OK;
}
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main(int argc, char **argv)
{
	const size_t nframes = argc > 1 ? strtoul(argv[1], 0, 10) : 8192;
	const int level = argc > 2 ? atoi(argv[2]) : 8, md5 = argc > 3 ? atoi(argv[3]) : 1, threads = argc > 4 ? atoi(argv[4]) : 0;
	const size_t n = nframes * 4096;
	int32_t *pcm = malloc(n * 2 * sizeof *pcm);
	uint32_t lcg = 12345;
	double s1 = 0, s2 = 0;
	for(size_t i = 0; i < n; i++) {           /* tones + smoothed noise, channels correlated */
		lcg = lcg * 1664525u + 1013904223u; s1 = 0.9 * s1 + 0.1 * ((int32_t)(lcg >> 8) % 4096 - 2048);
		lcg = lcg * 1664525u + 1013904223u; s2 = 0.9 * s2 + 0.1 * ((int32_t)(lcg >> 8) % 4096 - 2048);
		const double t = 9000.0 * sin(2 * M_PI * 441.0 * i / 44100.0) + 5000.0 * sin(2 * M_PI * 1234.5 * i / 44100.0 + 0.3);
		pcm[2 * i] = (int32_t)lrint(t + 3.0 * s1); pcm[2 * i + 1] = (int32_t)lrint(0.8 * t + 3.0 * s2 + 1.5 * s1);
	}
	for(int rep = 0; rep < 3; rep++) {      /* first: HIP runtime start-up included; then: the engine parked by the stream before */
		g_bytes = g_frames = 0;
		const double t0 = now();
		FLAC__StreamEncoder *e = FLAC__stream_encoder_new();
		FLAC__stream_encoder_set_channels(e, 2); FLAC__stream_encoder_set_bits_per_sample(e, 16); FLAC__stream_encoder_set_sample_rate(e, 44100);
		FLAC__stream_encoder_set_compression_level(e, (uint32_t)level);
		FLAC__stream_encoder_set_do_md5(e, md5);
		if(threads > 1) FLAC__stream_encoder_set_num_threads(e, (uint32_t)threads);
		if(FLAC__stream_encoder_init_stream(e, wcb, 0, 0, 0, 0) != 0) { fprintf(stderr, "init failed\n"); return 1; }
		const size_t chunk = 1u << 20;
		for(size_t i = 0; i < n; i += chunk)
			if(!FLAC__stream_encoder_process_interleaved(e, pcm + 2 * i, (uint32_t)(n - i < chunk ? n - i : chunk))) { fprintf(stderr, "process failed\n"); return 1; }
		if(!FLAC__stream_encoder_finish(e)) { fprintf(stderr, "finish failed\n"); return 1; }
		FLAC__stream_encoder_delete(e);
		const double dt = now() - t0;
		printf("level %d md5 %d threads %d: %9.1f M samples/s (%zu samples, %zu frames, %zu bytes, %.3f s incl. init)\n", level, md5, threads, n / dt / 1e6, n, g_frames, g_bytes, dt);
	}
	free(pcm);
	return 0;
}
