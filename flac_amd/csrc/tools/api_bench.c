/* flac_amd/csrc/tools/api_bench.c -- the drop-in boundary timed from a C client: PCM in host memory through
 * FLAC__stream_encoder_process_interleaved (src/libFLAC/stream_encoder.c:2570), frames to a write callback that keeps them in memory
 * (no file), FLAC__stream_encoder_finish -- what a program linked against libFLAC does.  bench.py runs it for its `libflac_api`
 * side figures: one stream with MD5 off and on (the API's default), and K streams at once, a thread and an encoder
 * each, MD5 on -- the shape in which a serial per-stream hash stops being the bound.
 * The same source builds against the reference library (-DUSE_REF: oracle/_ref/api_bench_ref) so that both can be asked for the
 * same stream and their bytes compared.
 *   usage: api_bench <pcm.i32> <samples> <blocksize, 0 = the preset's> <level> <md5 0|1> <streams K> <reps> [dump.flac]
 *   pcm.i32: interleaved int32 stereo 16-bit samples (samples x 2 values); every stream encodes all of it.
 *   prints one JSON line; with dump.flac, stream 0 of the last repetition is written there. */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>
#ifdef USE_REF
#include "FLAC/stream_encoder.h"
extern FLAC__bool FLAC__stream_encoder_set_do_md5(FLAC__StreamEncoder *encoder, FLAC__bool value);
#else
#include "FLACgpu_stream_encoder.h"
#endif

typedef struct { unsigned char *buf; size_t len, cap, pos; size_t frames; } sink_t;
static FLAC__StreamEncoderWriteStatus wcb(const FLAC__StreamEncoder *e, const FLAC__byte b[], size_t n, uint32_t samples, uint32_t frame, void *cd)
{
	sink_t *s = (sink_t *)cd; (void)e; (void)frame;
	if(s->pos + n > s->cap) { size_t c = s->cap ? s->cap * 2 : (size_t)1 << 24; while(c < s->pos + n) c *= 2; unsigned char *p = realloc(s->buf, c); if(!p) return FLAC__STREAM_ENCODER_WRITE_STATUS_FATAL_ERROR; s->buf = p; s->cap = c; }
	memcpy(s->buf + s->pos, b, n);
	s->pos += n; if(s->pos > s->len) s->len = s->pos;
	if(samples) s->frames++;
	return FLAC__STREAM_ENCODER_WRITE_STATUS_OK;
}
/* seek / tell: the STREAMINFO block is fixed up at finish, as for a file (stream_encoder.c:3251-3298) */
static FLAC__StreamEncoderSeekStatus scb(const FLAC__StreamEncoder *e, FLAC__uint64 off, void *cd) { (void)e; ((sink_t *)cd)->pos = (size_t)off; return FLAC__STREAM_ENCODER_SEEK_STATUS_OK; }
static FLAC__StreamEncoderTellStatus tcb(const FLAC__StreamEncoder *e, FLAC__uint64 *off, void *cd) { (void)e; *off = ((sink_t *)cd)->pos; return FLAC__STREAM_ENCODER_TELL_STATUS_OK; }
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

typedef struct { const int32_t *pcm; size_t n; uint32_t block; int level, md5; sink_t sink; int ok; pthread_barrier_t *bar; } job_t;
static void *run(void *arg)
{
	job_t *j = (job_t *)arg;
	j->ok = 0; j->sink.len = j->sink.pos = 0; j->sink.frames = 0;
	if(j->bar) pthread_barrier_wait(j->bar);
	FLAC__StreamEncoder *e = FLAC__stream_encoder_new();
	if(!e) return 0;
	FLAC__stream_encoder_set_channels(e, 2); FLAC__stream_encoder_set_bits_per_sample(e, 16); FLAC__stream_encoder_set_sample_rate(e, 44100);
	FLAC__stream_encoder_set_compression_level(e, (uint32_t)j->level);
	if(j->block) FLAC__stream_encoder_set_blocksize(e, j->block);
	FLAC__stream_encoder_set_do_md5(e, j->md5);
	FLAC__stream_encoder_set_total_samples_estimate(e, j->n);
	if(FLAC__stream_encoder_init_stream(e, wcb, scb, tcb, 0, &j->sink) != 0) { FLAC__stream_encoder_delete(e); return 0; }
	const size_t chunk = (size_t)1 << 20;
	int ok = 1;
	for(size_t i = 0; i < j->n && ok; i += chunk) ok = FLAC__stream_encoder_process_interleaved(e, j->pcm + 2 * i, (uint32_t)(j->n - i < chunk ? j->n - i : chunk)) ? 1 : 0;
	if(!FLAC__stream_encoder_finish(e)) ok = 0;
	FLAC__stream_encoder_delete(e);
	j->ok = ok;
	return 0;
}

int main(int argc, char **argv)
{
	if(argc < 8) { fprintf(stderr, "usage: api_bench <pcm.i32> <samples> <blocksize> <level> <md5> <streams> <reps> [dump]\n"); return 2; }
	const size_t n = strtoul(argv[2], 0, 10);
	const uint32_t block = (uint32_t)atoi(argv[3]);
	const int level = atoi(argv[4]), md5 = atoi(argv[5]), K = atoi(argv[6]), reps = atoi(argv[7]);
	const char *dump = argc > 8 ? argv[8] : 0;
	const int fd = open(argv[1], O_RDONLY);
	struct stat st;
	if(fd < 0 || fstat(fd, &st) != 0 || (size_t)st.st_size < n * 2 * sizeof(int32_t)) { fprintf(stderr, "api_bench: cannot use %s\n", argv[1]); return 2; }
	const int32_t *pcm = mmap(0, n * 2 * sizeof(int32_t), PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
	if(pcm == MAP_FAILED) { perror("mmap"); return 2; }
	job_t *jobs = calloc((size_t)K, sizeof *jobs);
	pthread_t *th = calloc((size_t)K, sizeof *th);
	double best = 1e30, first = 0;
	int all_ok = 1;
	for(int r = 0; r < reps; r++) {
		pthread_barrier_t bar;
		pthread_barrier_init(&bar, 0, (unsigned)K + 1);
		for(int k = 0; k < K; k++) { jobs[k].pcm = pcm; jobs[k].n = n; jobs[k].block = block; jobs[k].level = level; jobs[k].md5 = md5; jobs[k].bar = &bar; pthread_create(&th[k], 0, run, &jobs[k]); }
		pthread_barrier_wait(&bar);
		const double t0 = now();
		for(int k = 0; k < K; k++) pthread_join(th[k], 0);
		const double dt = now() - t0;
		pthread_barrier_destroy(&bar);
		for(int k = 0; k < K; k++) if(!jobs[k].ok) all_ok = 0;
		if(r == 0) first = dt;
		if(dt < best) best = dt;
	}
	/* every stream got the same input: the same bytes must have come out */
	int same = 1;
	for(int k = 1; k < K; k++) if(jobs[k].sink.len != jobs[0].sink.len || memcmp(jobs[k].sink.buf, jobs[0].sink.buf, jobs[0].sink.len) != 0) same = 0;
	if(dump) { FILE *f = fopen(dump, "wb"); if(f) { fwrite(jobs[0].sink.buf, 1, jobs[0].sink.len, f); fclose(f); } }
	printf("{\"ok\": %s, \"streams\": %d, \"md5\": %d, \"level\": %d, \"samples_per_stream\": %zu, \"frames_per_stream\": %zu, \"bytes_per_stream\": %zu, "
	       "\"seconds_best\": %.6f, \"seconds_first\": %.6f, \"reps\": %d, \"Msamples_per_s\": %.1f, \"streams_identical\": %s}\n",
	       all_ok ? "true" : "false", K, md5, level, n, jobs[0].sink.frames, jobs[0].sink.len, best, first, reps, (double)K * (double)n / best / 1e6, same ? "true" : "false");
	return all_ok && same ? 0 : 1;
}
