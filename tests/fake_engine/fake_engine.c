/* tests/fake_engine/fake_engine.c -- TEST INFRASTRUCTURE, not an encoder.
 *
 * A stand-in for libflacgpu.so (include/flacgpu.h) that lets the host API layer (flac_amd/csrc/host/stream_encoder.c: batch
 * slots, worker thread, the MD5 chain that runs ahead of the submissions, the helper threads of the narrowing copy, delivery
 * order, STREAMINFO fix-up) run on a machine without a GPU.  It does NOT produce FLAC: a "frame" is a 16-byte record
 *     'F' 'K' | u16 pad | u32 frame number | u32 samples | u32 FNV-1a of the frame's raw sample bytes
 * followed by (frame number % 5) bytes of 0xEE, so that frame lengths vary.  tests/test_host_pipeline_cpu.py puts a directory
 * holding this library (built as libflacgpu.so) in front of the product's on LD_LIBRARY_PATH, in a subprocess of its own, and
 * checks every record, their order and the STREAMINFO against values computed in Python.  Nothing in flac_amd/ refers to it. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "flacgpu.h"

struct fake_job { const void *raw; flacgpu_raw_format fmt; uint32_t nframes; uint64_t first; uint32_t last; uint8_t *out; size_t out_cap; uint32_t *fb; };
struct flacgpu_ctx { flacgpu_config cfg; uint32_t verify; struct fake_job q[FLACGPU_ASYNC_SLOTS]; uint64_t sub, col; unsigned submit_calls; };
static int g_creates, g_destroys;
int fake_engine_creates(void) { return g_creates; }
int fake_engine_destroys(void) { return g_destroys; }

static void pause_us(const char *env)
{
	const char *d = getenv(env);
	if(d) { struct timespec ts = {atol(d) / 1000000, 1000L * (atol(d) % 1000000)}; nanosleep(&ts, 0); }
}
int flacgpu_create(const flacgpu_config *cfg, const float *windows, flacgpu_ctx **out)
{
	(void)windows;
	if(!cfg || !out || cfg->abi_version != FLACGPU_ABI_VERSION) return FLACGPU_ERR_BAD_ARG;
	pause_us("FAKE_ENGINE_CREATE_DELAY_US");                      /* a fresh process spends 0.1-0.2 s here on the real thing */
	if(getenv("FAKE_ENGINE_FAIL_CREATE")) return FLACGPU_ERR_NO_DEVICE;
	flacgpu_ctx *c = calloc(1, sizeof *c);
	if(!c) return FLACGPU_ERR_ALLOC;
	c->cfg = *cfg;
	*out = c;
	g_creates++;
	return FLACGPU_OK;
}
void flacgpu_destroy(flacgpu_ctx *ctx) { if(ctx) g_destroys++; free(ctx); }
size_t flacgpu_max_output_bytes(const flacgpu_ctx *ctx, uint32_t nframes) { (void)ctx; return (size_t)nframes * 32; }
void *flacgpu_alloc_pinned(size_t bytes) { void *p = 0; if(posix_memalign(&p, 4096, bytes ? bytes : 1) != 0) return 0; memset(p, 0, bytes); return p; }   /* touched, as page-locked memory is */
void flacgpu_free_pinned(void *p) { free(p); }
int flacgpu_host_register(void *p, size_t bytes) { (void)p; (void)bytes; return getenv("FAKE_ENGINE_FAIL_REGISTER") ? FLACGPU_ERR_ALLOC : FLACGPU_OK; }
void flacgpu_host_unregister(void *p) { (void)p; }
int flacgpu_device_probe(void) { return getenv("FAKE_ENGINE_NO_PROBE") ? 0 : 1; }
int flacgpu_config_check(const flacgpu_config *cfg) { return cfg && cfg->abi_version == FLACGPU_ABI_VERSION ? FLACGPU_OK : FLACGPU_ERR_BAD_ARG; }
size_t flacgpu_config_max_output_bytes(const flacgpu_config *cfg, uint32_t nframes) { (void)cfg; return (size_t)nframes * 32; }
const char *flacgpu_strerror(int code) { (void)code; return "fake engine"; }
int flacgpu_set_verify(flacgpu_ctx *ctx, uint32_t on) { ctx->verify = on; return FLACGPU_OK; }
int flacgpu_set_phase_timing(flacgpu_ctx *ctx, uint32_t every) { (void)ctx; (void)every; return FLACGPU_OK; }
int flacgpu_last_verify_result(flacgpu_ctx *ctx, flacgpu_verify_result *out) { (void)ctx; memset(out, 0, sizeof *out); return FLACGPU_OK; }

int64_t flacgpu_encode_batch_raw(flacgpu_ctx *ctx, const void *raw, const flacgpu_raw_format *fmt, uint32_t nframes,
                                 uint64_t first_frame_number, uint32_t last_block_samples, const float *tail_windows,
                                 uint8_t *out, size_t out_cap, uint32_t *frame_bytes)
{
	(void)tail_windows;
	const uint32_t N = ctx->cfg.blocksize, C = ctx->cfg.channels, w = fmt->container_bits / 8;
	const uint8_t *p = raw;
	size_t total = 0;
	if(nframes > ctx->cfg.max_batch_frames) return FLACGPU_ERR_BAD_ARG;
	/* a GPU batch takes a few milliseconds: leave the other threads time to run ahead (FAKE_ENGINE_DELAY_US) */
	pause_us("FAKE_ENGINE_DELAY_US");
	for(uint32_t f = 0; f < nframes; f++) {
		const uint32_t n = (f + 1 == nframes && last_block_samples) ? last_block_samples : N;
		const size_t bytes = (size_t)n * C * w;
		uint32_t h = 2166136261u;
		for(size_t i = 0; i < bytes; i++) h = (h ^ p[i]) * 16777619u;
		p += bytes;
		const uint32_t fn = (uint32_t)(first_frame_number + f), len = 16 + fn % 5;
		if(total + len > out_cap) return FLACGPU_ERR_OUTPUT_TOO_SMALL;
		uint8_t *q = out + total;
		q[0] = 'F'; q[1] = 'K'; q[2] = q[3] = 0;
		memcpy(q + 4, &fn, 4); memcpy(q + 8, &n, 4); memcpy(q + 12, &h, 4);
		memset(q + 16, 0xEE, len - 16);
		frame_bytes[f] = len;
		total += len;
	}
	return (int64_t)total;
}

/* the asynchronous entry: a submission is only NOTED -- the raw bytes are read when the batch is collected, as late as the real
 * engine's copy engine may read them, so that a host layer that reuses a slot before collecting it is caught by the records */
int flacgpu_submit_batch_raw(flacgpu_ctx *ctx, const void *raw, const flacgpu_raw_format *fmt, uint32_t nframes, uint64_t first_frame_number, uint32_t last_block_samples,
                             const float *tail_windows, uint8_t *out, size_t out_cap, uint32_t *frame_bytes)
{
	(void)tail_windows;
	if(ctx->sub - ctx->col >= FLACGPU_ASYNC_SLOTS) return FLACGPU_ERR_BUSY;
	if(nframes > ctx->cfg.max_batch_frames) return FLACGPU_ERR_BAD_ARG;
	if(getenv("FAKE_ENGINE_FAIL_SUBMIT")) return FLACGPU_ERR_LAUNCH;
	{
		/* FAKE_ENGINE_FAIL_SUBMIT_NTH=n: only the n-th submission (1-based) of this engine is refused, the ones around it go through */
		const char *nth = getenv("FAKE_ENGINE_FAIL_SUBMIT_NTH");
		ctx->submit_calls++;
		if(nth && ctx->submit_calls == (unsigned)atoi(nth)) return FLACGPU_ERR_LAUNCH;
	}
	struct fake_job *j = &ctx->q[ctx->sub % FLACGPU_ASYNC_SLOTS];
	j->raw = raw; j->fmt = *fmt; j->nframes = nframes; j->first = first_frame_number; j->last = last_block_samples; j->out = out; j->out_cap = out_cap; j->fb = frame_bytes;
	ctx->sub++;
	return FLACGPU_OK;
}
int64_t flacgpu_collect(flacgpu_ctx *ctx)
{
	if(ctx->sub == ctx->col) return FLACGPU_ERR_BAD_ARG;
	const struct fake_job j = ctx->q[ctx->col % FLACGPU_ASYNC_SLOTS];
	ctx->col++;
	return flacgpu_encode_batch_raw(ctx, j.raw, &j.fmt, j.nframes, j.first, j.last, 0, j.out, j.out_cap, j.fb);
}
int flacgpu_in_flight(const flacgpu_ctx *ctx) { return (int)(ctx->sub - ctx->col); }
