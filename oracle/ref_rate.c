/* oracle/ref_rate.c -- TEST INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg).
 * Times the UNMODIFIED reference libFLAC (oracle/_ref/libFLAC_ref.so, driven through its public API only,
 * include/FLAC/stream_encoder.h) on this box's host cores the way SURVEY.md 8d prescribes for the CPU baseline:
 * input = a raw 16-bit stereo file on tmpfs, output discarded in the write callback, `reps` encodes back to back,
 * MD5 on (as `flac` has it).  bench.py starts one of these per core for the independent-process figure, and one
 * with -j N for the library's own thread pool (FLAC__stream_encoder_set_num_threads, stream_encoder.c:2151).
 *   ref_rate <raw16le-stereo-file> <level> <threads> <reps> [bps rate channels]
 * prints: samples reps seconds_total seconds_best */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <time.h>
#include "FLAC/stream_encoder.h"

static FLAC__StreamEncoderWriteStatus sink(const FLAC__StreamEncoder *e, const FLAC__byte b[], size_t n, uint32_t s, uint32_t f, void *c)
{
	(void)e; (void)b; (void)s; (void)f;
	*(uint64_t *)c += n;
	return FLAC__STREAM_ENCODER_WRITE_STATUS_OK;
}
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

int main(int argc, char **argv)
{
	if(argc < 5) { fprintf(stderr, "usage: ref_rate file level threads reps [bps rate channels]\n"); return 2; }
	const int level = atoi(argv[2]), threads = atoi(argv[3]), reps = atoi(argv[4]);
	const unsigned bps = argc > 5 ? (unsigned)atoi(argv[5]) : 16, rate = argc > 6 ? (unsigned)atoi(argv[6]) : 44100, ch = argc > 7 ? (unsigned)atoi(argv[7]) : 2;
	FILE *f = fopen(argv[1], "rb");
	if(!f) { perror(argv[1]); return 1; }
	fseek(f, 0, SEEK_END);
	const long bytes = ftell(f);
	fseek(f, 0, SEEK_SET);
	const size_t width = bps > 16 ? 4 : 2;                       /* 16-bit samples as int16, wider ones as int32 */
	const size_t nval = (size_t)bytes / width, nsamp = nval / ch;
	int32_t *pcm = malloc(nval * sizeof *pcm);
	void *raw = malloc((size_t)bytes);
	if(!pcm || !raw || fread(raw, 1, (size_t)bytes, f) != (size_t)bytes) { fprintf(stderr, "read failed\n"); return 1; }
	fclose(f);
	for(size_t i = 0; i < nval; i++) pcm[i] = width == 2 ? ((int16_t *)raw)[i] : ((int32_t *)raw)[i];
	free(raw);
	double total = 0, best = 1e30;
	for(int r = 0; r < reps; r++) {
		uint64_t out = 0;
		FLAC__StreamEncoder *e = FLAC__stream_encoder_new();
		FLAC__stream_encoder_set_channels(e, ch);
		FLAC__stream_encoder_set_bits_per_sample(e, bps);
		FLAC__stream_encoder_set_sample_rate(e, rate);
		FLAC__stream_encoder_set_compression_level(e, (uint32_t)level);
		FLAC__stream_encoder_set_total_samples_estimate(e, nsamp);
		if(threads > 1) FLAC__stream_encoder_set_num_threads(e, (uint32_t)threads);
		const double t0 = now();
		if(FLAC__stream_encoder_init_stream(e, sink, 0, 0, 0, &out) != FLAC__STREAM_ENCODER_INIT_STATUS_OK) { fprintf(stderr, "init failed\n"); return 1; }
		/* the CLI feeds 2048-sample reads (src/flac/encode.c); bigger chunks only help the reference */
		for(size_t i = 0; i < nsamp; i += 65536) {
			const size_t n = nsamp - i < 65536 ? nsamp - i : 65536;
			if(!FLAC__stream_encoder_process_interleaved(e, pcm + i * ch, (uint32_t)n)) { fprintf(stderr, "process failed\n"); return 1; }
		}
		FLAC__stream_encoder_finish(e);
		const double dt = now() - t0;
		FLAC__stream_encoder_delete(e);
		total += dt;
		if(dt < best) best = dt;
	}
	printf("%zu %d %.6f %.6f\n", nsamp, reps, total, best);
	return 0;
}
