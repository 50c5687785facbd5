/* flac_amd/csrc/host/stream_encoder.c -- the libFLAC stream-encoder API on top of the GPU frame engine.
 *
 * Mirrors the reference's public encoder interface (include/FLAC/stream_encoder.h:704-1896; behaviour
 * of src/libFLAC/stream_encoder.c:570-2626 for the object life cycle, :2988-3300 for stream output and
 * metadata fix-up, stream_encoder_framing.c:46-243 for metadata block serialisation).
 *
 * What is different by design: the reference encodes one block per process_frame_() call
 * (stream_encoder.c:3627); here whole blocks are STAGED in a pinned host buffer and handed to the HIP
 * engine `batch_frames` at a time (include/flacgpu.h).  Frames come back byte-aligned and in order;
 * the per-frame bookkeeping of write_frame_() (seek points, min/max frame size, callbacks) then runs
 * on the calling thread exactly as it would have.  The one-sample "overread" of the reference
 * (stream_encoder.c:565-576) is kept: a block is only released once one more sample exists, so the
 * final block -- full or short -- is always produced by finish().
 *
 * No CPU encode path: if the engine cannot be created init fails (ENCODER_ERROR) and says why on stderr.
 */
#define _FILE_OFFSET_BITS 64
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include <limits.h>
#include <stdatomic.h>
#include <time.h>
#include <unistd.h>
#include <sched.h>
#include "flacgpu_host.h"
#include "ogg.h"
#include "FLACgpu_stream_encoder.h"

/* format.c:57: the reference writes its vendor string into every VORBIS_COMMENT block.  Files are
 * required to be byte-identical to the reference's, so the same string is written here. */
const char *FLAC__VENDOR_STRING = "reference libFLAC 1.5.0 20250211";

const char * const FLAC__StreamEncoderStateString[] = {
	"FLAC__STREAM_ENCODER_OK", "FLAC__STREAM_ENCODER_UNINITIALIZED", "FLAC__STREAM_ENCODER_OGG_ERROR",
	"FLAC__STREAM_ENCODER_VERIFY_DECODER_ERROR", "FLAC__STREAM_ENCODER_VERIFY_MISMATCH_IN_AUDIO_DATA",
	"FLAC__STREAM_ENCODER_CLIENT_ERROR", "FLAC__STREAM_ENCODER_IO_ERROR", "FLAC__STREAM_ENCODER_FRAMING_ERROR",
	"FLAC__STREAM_ENCODER_MEMORY_ALLOCATION_ERROR"
};
const char * const FLAC__StreamEncoderInitStatusString[] = {
	"FLAC__STREAM_ENCODER_INIT_STATUS_OK", "FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR",
	"FLAC__STREAM_ENCODER_INIT_STATUS_UNSUPPORTED_CONTAINER", "FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_CALLBACKS",
	"FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_NUMBER_OF_CHANNELS", "FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_BITS_PER_SAMPLE",
	"FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_SAMPLE_RATE", "FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_BLOCK_SIZE",
	"FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_MAX_LPC_ORDER", "FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_QLP_COEFF_PRECISION",
	"FLAC__STREAM_ENCODER_INIT_STATUS_BLOCK_SIZE_TOO_SMALL_FOR_LPC_ORDER", "FLAC__STREAM_ENCODER_INIT_STATUS_NOT_STREAMABLE",
	"FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_METADATA", "FLAC__STREAM_ENCODER_INIT_STATUS_ALREADY_INITIALIZED"
};
const char * const FLAC__StreamEncoderReadStatusString[] = {
	"FLAC__STREAM_ENCODER_READ_STATUS_CONTINUE", "FLAC__STREAM_ENCODER_READ_STATUS_END_OF_STREAM",
	"FLAC__STREAM_ENCODER_READ_STATUS_ABORT", "FLAC__STREAM_ENCODER_READ_STATUS_UNSUPPORTED"
};
const char * const FLAC__StreamEncoderWriteStatusString[] = {
	"FLAC__STREAM_ENCODER_WRITE_STATUS_OK", "FLAC__STREAM_ENCODER_WRITE_STATUS_FATAL_ERROR"
};
const char * const FLAC__StreamEncoderSeekStatusString[] = {
	"FLAC__STREAM_ENCODER_SEEK_STATUS_OK", "FLAC__STREAM_ENCODER_SEEK_STATUS_ERROR", "FLAC__STREAM_ENCODER_SEEK_STATUS_UNSUPPORTED"
};
const char * const FLAC__StreamEncoderTellStatusString[] = {
	"FLAC__STREAM_ENCODER_TELL_STATUS_OK", "FLAC__STREAM_ENCODER_TELL_STATUS_ERROR", "FLAC__STREAM_ENCODER_TELL_STATUS_UNSUPPORTED"
};

/* what the reference keeps in FLAC__StreamEncoderProtected: the client-visible settings and state */
struct FLAC__StreamEncoderProtected {
	FLAC__StreamEncoderState state;
	flacgpu_host_settings s;
	uint32_t num_threads;
	FLAC__StreamMetadata **metadata;
	uint32_t num_metadata_blocks;
	FLAC__uint64 streaminfo_offset, seektable_offset, audio_offset;
};

/* ... and in FLAC__StreamEncoderPrivate: callbacks, stream bookkeeping, and here the batch staging */
#define NSLOT 4                              /* batch slots (<= FLACGPU_ASYNC_SLOTS + 1: one is always the caller's) */
#define FLACGPU_PRE_MAX_SEGS 1024            /* "fLaC" + STREAMINFO + a VORBIS_COMMENT + the client's blocks: one segment each (beyond: written by init itself) */
struct FLAC__StreamEncoderPrivate {
	FLAC__StreamEncoderWriteCallback write_cb;
	FLAC__StreamEncoderReadCallback read_cb;  /* Ogg FLAC only: the STREAMINFO page is read back at finish */
	int is_ogg, final_batch, emit_is_last;                  /* final_batch: the batch being delivered ends the stream (its last frame closes the Ogg stream) */
	fgh_ogg_aspect ogg;
	FLAC__StreamEncoderSeekCallback seek_cb;
	FLAC__StreamEncoderTellCallback tell_cb;
	FLAC__StreamEncoderMetadataCallback metadata_cb;
	FLAC__StreamEncoderProgressCallback progress_cb;
	void *client_data;
	FILE *file;
	FLAC__uint64 bytes_written, samples_written;
	uint32_t frames_written, total_frames_estimate;
	uint32_t current_frame_number;
	uint32_t frame_blocksize;                 /* samples of the frame being written (get_blocksize during callbacks) */
	FLAC__StreamMetadata streaminfo;
	FLAC__StreamMetadata_SeekTable *seek_table;
	uint32_t first_seekpoint_to_check;
	flacgpu_host_md5 md5;
	int is_being_deleted;
	/* GPU batches.  A ring of NSLOT slots: the caller's thread fills one with sample BYTES (little endian, ceil(bps/8) wide: at
	 * once the input format of the MD5 (:3448) and a raw format the engine stages on the device), while the worker thread
	 * runs the MD5 chain and hands the slots behind it to the engine's asynchronous entry (flacgpu_submit_batch_raw: up to
	 * NSLOT - 1 batches in flight, their input copies and read-backs beside the kernels of the batches in front) and collects
	 * them in order; frames are delivered on the caller's thread, in stream order -- the reference's ring of 2*threads+2 frame
	 * tasks (stream_encoder.c:1134-1238, 3530-3574), per batch. */
	flacgpu_ctx *gpu;
	uint32_t batch_frames;
	uint32_t width;                           /* bytes per staged sample */
	flacgpu_raw_format rawfmt;
	struct batch_slot {
		uint8_t *raw;                         /* pinned [batch_frames*blocksize + 1][channels][width] */
		uint8_t *out;                         /* pinned, frames back to back */
		uint32_t *frame_bytes;
		uint32_t nframes, tail, first_frame;  /* the submitted batch */
		int64_t total;                        /* result: bytes, or a negative FLACGPU_ERR_* */
		flacgpu_host_verify_result vres;      /* set_verify: the first frame of the batch that does not decode back to its input */
		int state;                            /* 0 being filled / free, 1 handed to the worker, 3 in flight on the engine, 2 done */
	} slot[NSLOT];
	int inflight, col_slot;                   /* worker: batches on the engine, and the slot of the oldest of them */
	int submit_failed;                        /* worker: a submission failed (its error code): nothing goes to the engine after it, so the
	                                           * batches in flight always occupy consecutive ring slots from col_slot on */
	int cur;                                  /* slot the caller is filling */
	size_t staged;                            /* inter-channel samples in slot[cur] */
	size_t out_cap;
	uint32_t next_frame_number;               /* of the next batch to submit */
	float *tail_windows;
	pthread_t worker;
	int worker_started, worker_quit;
	flacgpu_host_verify_result verify_stats;  /* get_verify_decoder_error_stats */
	pthread_mutex_t mu;
	pthread_cond_t cv;
	/* The MD5 is one serial chain over the whole stream (md5.c:497) and the slowest stage of a 16-bit stream (~0.65 GB/s): the
	 * worker runs it AHEAD of the submissions, over the samples the caller has published from the slot it is still filling
	 * (pub_slot/pub_staged, under mu), so that the chain starts with the first samples and not with the first full batch. */
	int md5_slot;                             /* slot of the batch the chain is in */
	size_t md5_done;                          /* inter-channel samples of that batch already hashed */
	size_t md5_ahead;                         /* ... and of the batch after it (only while the first batch waits for the engine) */
	int pub_slot; size_t pub_staged, pub_last;
	struct stage_pool *pool;                  /* helper threads of the narrowing copy (big process() calls only) */
	/* Bring-up.  Starting the HIP runtime, creating the engine and page-locking the slots takes a fresh process 0.2-0.3 s -- as
	 * long as a 30-minute stream then takes to encode -- so init_*() hands it to a thread of its own and returns; the caller
	 * stages into the (not yet page-locked) slots and the worker runs the MD5 chain meanwhile, and the first batch waits for the
	 * engine.  A failure after init_*() has returned surfaces as a failed process()/finish() with an error state and a message
	 * on stderr.  When the device node is missing, or with FLACGPU_SYNC_INIT=1, bring-up runs inside init_*() and fails there. */
	int engine_on;                            /* init_*() succeeded and finish() has not run yet */
	pthread_t bring_th;
	int bring_started, bring_done, bring_result;
	/* Asynchronous bring-up and the client's output: init_*() has not written "fLaC" and the metadata blocks yet when it
	 * returns with the engine still coming up -- they go out in front of the first frame (emit()), or from finish() for a
	 * stream without one, once the engine is known to be there.  An engine that fails to come up (device memory, a kernel
	 * image the device cannot load) then fails the stream before a single byte reached the client's file, as the reference's
	 * init would have (a client that branches on the init status must not be left with a half-written file). */
	int preamble_pending;
	uint8_t *pre_buf;                         /* the head, serialised by init_*() -- a client may free its metadata objects as soon as init has */
	size_t pre_len[FLACGPU_PRE_MAX_SEGS];     /* returned (the reference has written them by then; `flac` does exactly that) -- one write per segment */
	uint32_t pre_nseg;
	flacgpu_config bring_cfg;
	size_t raw_bytes;
	int registered[2 * NSLOT];
	float *windows; size_t wcount;            /* the engine's window tables (host copy) */
	int engine_failed;                        /* an engine call failed: this engine is not parked */
	double t_bring_wait;
	/* FLACGPU_HOST_TIMING=1: where a stream's wall time went, printed by finish() */
	int timing;
	double t_init_engine, t_init_pinned, t_stage, t_wait, t_emit, t_md5, t_encode, t_release, t_start;
};
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }
#define MD5_PIECE ((size_t)1 << 16)           /* samples: the caller publishes, and the worker hashes, in pieces of at least this */

/* export.h:107 -- 1: this library writes Ogg FLAC (host/ogg.c) */
int FLAC_API_SUPPORTS_OGG_FLAC = 1;

#define PROT(e) ((e)->protected_)
#define PRIV(e) ((e)->private_)

/* ------------------------------------------------------------------------------------------------
 * object life cycle (stream_encoder.c:570-700)
 * ---------------------------------------------------------------------------------------------- */
static void set_defaults(FLAC__StreamEncoder *e)
{
	flacgpu_host_settings_defaults(&PROT(e)->s);      /* set_defaults_ :2628-2700, ends with compression level 5 */
	PROT(e)->num_threads = 1;
	PROT(e)->metadata = 0;
	PROT(e)->num_metadata_blocks = 0;
	struct FLAC__StreamEncoderPrivate *p = PRIV(e);
	p->write_cb = 0; p->seek_cb = 0; p->tell_cb = 0; p->metadata_cb = 0; p->progress_cb = 0; p->client_data = 0;
	p->read_cb = 0; p->is_ogg = 0; p->final_batch = 0;
	fgh_ogg_aspect_set_defaults(&p->ogg);
	p->seek_table = 0;
}

FLAC__StreamEncoder *FLAC__stream_encoder_new(void)
{
	FLAC__StreamEncoder *e = calloc(1, sizeof *e);
	if(!e) return 0;
	e->protected_ = calloc(1, sizeof *e->protected_);
	e->private_ = calloc(1, sizeof *e->private_);
	if(!e->protected_ || !e->private_) { free(e->protected_); free(e->private_); free(e); return 0; }
	set_defaults(e);
	PROT(e)->state = FLAC__STREAM_ENCODER_UNINITIALIZED;
	return e;
}

/* ------------------------------------------------------------------------------------------------
 * helper threads of the narrowing copy.  process*() is the one pass a host thread makes over the client's samples; with the
 * MD5 off it sets the rate of the whole single-stream API (the GPU behind it is an order of magnitude faster), so calls that
 * bring >= 256 K values are cut into one range per thread.  A helper spins for some tens of microseconds after a job (calls
 * arrive back to back) and sleeps on the condition variable otherwise.
 * ---------------------------------------------------------------------------------------------- */
#define STAGE_MAX_THREADS 16
struct stage_job { int (*fn)(const void *args, size_t lo, size_t hi); const void *args; size_t lo, hi; int ok; };
struct stage_pool {
	int n;                                    /* helpers */
	pthread_t th[STAGE_MAX_THREADS];
	struct stage_job job[STAGE_MAX_THREADS];
	pthread_mutex_t mu;
	pthread_cond_t cv;
	_Atomic uint64_t gen;
	_Atomic int pending, quit;
	struct stage_pool_arg { struct stage_pool *sp; int i; } arg[STAGE_MAX_THREADS];
};
static inline void cpu_relax(void)
{
#if defined(__x86_64__)
	__builtin_ia32_pause();
#endif
}
static void *stage_helper_main(void *a)
{
	struct stage_pool *sp = ((struct stage_pool_arg *)a)->sp;
	const int i = ((struct stage_pool_arg *)a)->i;
	uint64_t seen = 0;
	for(;;) {
		for(int spin = 0; spin < 4000 && atomic_load_explicit(&sp->gen, memory_order_acquire) == seen && !atomic_load_explicit(&sp->quit, memory_order_relaxed); spin++) cpu_relax();
		if(atomic_load_explicit(&sp->gen, memory_order_acquire) == seen) {
			pthread_mutex_lock(&sp->mu);
			while(atomic_load_explicit(&sp->gen, memory_order_acquire) == seen && !atomic_load_explicit(&sp->quit, memory_order_relaxed)) pthread_cond_wait(&sp->cv, &sp->mu);
			pthread_mutex_unlock(&sp->mu);
		}
		if(atomic_load_explicit(&sp->quit, memory_order_relaxed)) break;
		seen = atomic_load_explicit(&sp->gen, memory_order_acquire);
		struct stage_job *j = &sp->job[i];
		if(j->hi > j->lo) j->ok = j->fn(j->args, j->lo, j->hi);
		atomic_fetch_sub_explicit(&sp->pending, 1, memory_order_release);
	}
	return 0;
}
static struct stage_pool *stage_pool_create(int helpers)
{
	struct stage_pool *sp = calloc(1, sizeof *sp);
	if(!sp) return 0;
	pthread_mutex_init(&sp->mu, 0);
	pthread_cond_init(&sp->cv, 0);
	for(int i = 0; i < helpers; i++) {
		sp->arg[i].sp = sp; sp->arg[i].i = i;
		if(pthread_create(&sp->th[i], 0, stage_helper_main, &sp->arg[i]) != 0) break;
		sp->n++;
	}
	if(sp->n == 0) { pthread_mutex_destroy(&sp->mu); pthread_cond_destroy(&sp->cv); free(sp); return 0; }
	return sp;
}
static void stage_pool_destroy(struct stage_pool *sp)
{
	if(!sp) return;
	pthread_mutex_lock(&sp->mu);
	atomic_store(&sp->quit, 1);
	pthread_cond_broadcast(&sp->cv);
	pthread_mutex_unlock(&sp->mu);
	for(int i = 0; i < sp->n; i++) pthread_join(sp->th[i], 0);
	pthread_mutex_destroy(&sp->mu);
	pthread_cond_destroy(&sp->cv);
	free(sp);
}
/* fn(args, lo, hi) over [0, count) in ranges whose bounds are multiples of `grain`; returns the AND of the results */
static int stage_pool_run(struct stage_pool *sp, int (*fn)(const void *, size_t, size_t), const void *args, size_t count, size_t grain)
{
	const int parts = sp ? sp->n + 1 : 1;
	if(parts == 1 || count < ((size_t)1 << 18)) return fn(args, 0, count);
	size_t per = (count / (size_t)parts + grain - 1) / grain * grain;
	size_t lo = per < count ? per : count;                 /* the caller's own range is [0, lo) */
	const size_t mine = lo;
	for(int i = 0; i < sp->n; i++) {
		size_t hi = i + 1 == sp->n ? count : (lo + per < count ? lo + per : count);
		sp->job[i].fn = fn; sp->job[i].args = args; sp->job[i].lo = lo; sp->job[i].hi = hi; sp->job[i].ok = 1;
		lo = hi;
	}
	atomic_store_explicit(&sp->pending, sp->n, memory_order_relaxed);
	pthread_mutex_lock(&sp->mu);
	atomic_fetch_add_explicit(&sp->gen, 1, memory_order_release);
	pthread_cond_broadcast(&sp->cv);
	pthread_mutex_unlock(&sp->mu);
	int ok = fn(args, 0, mine);
	for(unsigned spin = 0; atomic_load_explicit(&sp->pending, memory_order_acquire) > 0; spin++) { if(spin < 20000) cpu_relax(); else sched_yield(); }
	for(int i = 0; i < sp->n; i++) ok &= sp->job[i].ok;
	return ok;
}

/* ------------------------------------------------------------------------------------------------
 * Parked engines.  Creating an engine costs 30 ms in a warm process (device buffers, 17 streams, 500 events),
 * releasing it and its page-locked slots 40 ms: together as long as 130 M samples take to pass through it.  A program that
 * encodes one stream after another with the same settings -- `flac *.wav` is one -- gets the previous stream's engine and
 * slots back instead.  finish() parks them here when the stream ended without an engine error; init_*() takes them when the
 * configuration is the same (window tables compared float by float) and the parked capacity is at least what it needs.  A
 * parked engine is never destroyed by the library: it is replaced by a newer one, or goes with the process.
 * FLACGPU_ENGINE_CACHE=0 switches the parking off.
 * ---------------------------------------------------------------------------------------------- */
struct engine_bundle {
	flacgpu_ctx *gpu;
	flacgpu_config cfg;
	float *windows; size_t wcount;
	uint8_t *raw[NSLOT], *out[NSLOT];
	uint32_t *fb[NSLOT];
	int registered[2 * NSLOT];
	size_t raw_bytes, out_cap;
};
/* Round 6: a pool, not one -- a program that encodes K streams at once (a thread and an encoder each: bench.py's libflac_api figure,
 * a ripper's worker pool) parks K engines when they finish and gets K back for its next K streams; with one place to park, K - 1
 * of them were created and destroyed per round, 70 ms each and one after the other inside the HIP runtime: 16 streams of 67 M
 * samples took 1.97 s, 1.6 of them that.  A sequential program still holds exactly one.  FLACGPU_ENGINE_CACHE=n: at most n (default
 * 16, 0: none). */
#define PARK_MAX 16
static pthread_mutex_t g_park_mu = PTHREAD_MUTEX_INITIALIZER;
static struct engine_bundle g_park[PARK_MAX];
static int g_park_valid[PARK_MAX];
static unsigned g_park_clock, g_park_age[PARK_MAX];
static int parking_limit(void) { const char *v = getenv("FLACGPU_ENGINE_CACHE"); int n = v ? atoi(v) : PARK_MAX; return n < 0 ? 0 : n > PARK_MAX ? PARK_MAX : n; }
static int parking_enabled(void) { return parking_limit() > 0; }
static int park_any(void) { int any = 0; for(int i = 0; i < PARK_MAX; i++) any |= g_park_valid[i]; return any; }      /* (under g_park_mu) */
static void bundle_destroy(struct engine_bundle *b)
{
	for(int i = 0; i < NSLOT; i++) {
		if(b->registered[2 * i]) flacgpu_host_unregister(b->raw[i]);
		if(b->registered[2 * i + 1]) flacgpu_host_unregister(b->out[i]);
		free(b->raw[i]); free(b->out[i]); free(b->fb[i]);
	}
	free(b->windows);
	if(b->gpu) flacgpu_destroy(b->gpu);
	memset(b, 0, sizeof *b);
}
/* a parked engine that serves `cfg` (capacity: at least cfg->max_batch_frames) with these window tables */
static int park_take(const flacgpu_config *cfg, const float *windows, size_t wcount, size_t raw_bytes, size_t out_cap, struct engine_bundle *out)
{
	int hit = 0;
	pthread_mutex_lock(&g_park_mu);
	for(int i = 0; i < PARK_MAX && !hit; i++) {
		if(!g_park_valid[i]) continue;
		flacgpu_config a = g_park[i].cfg, b = *cfg;
		const int roomy = a.max_batch_frames >= b.max_batch_frames && g_park[i].raw_bytes >= raw_bytes && g_park[i].out_cap >= out_cap;
		a.max_batch_frames = b.max_batch_frames = 0;
		if(roomy && memcmp(&a, &b, sizeof a) == 0 && g_park[i].wcount == wcount && (wcount == 0 || memcmp(g_park[i].windows, windows, wcount * sizeof(float)) == 0)) {
			*out = g_park[i];
			g_park_valid[i] = 0;
			hit = 1;
		}
	}
	pthread_mutex_unlock(&g_park_mu);
	return hit;
}
static int bundle_same_settings(const struct engine_bundle *x, const struct engine_bundle *y)
{
	flacgpu_config a = x->cfg, b = y->cfg;
	a.max_batch_frames = b.max_batch_frames = 0;
	return memcmp(&a, &b, sizeof a) == 0 && x->wcount == y->wcount && (x->wcount == 0 || memcmp(x->windows, y->windows, x->wcount * sizeof(float)) == 0);
}
/* engines of OTHER settings go when one is parked (a program that changes its settings from stream to stream holds one engine, as
 * before); engines of the same settings gather, up to the limit: those are streams that ran side by side */
static void park_put(struct engine_bundle *b)
{
	struct engine_bundle gone[PARK_MAX];
	int ngone = 0, slot = -1;
	const int limit = parking_limit();
	pthread_mutex_lock(&g_park_mu);
	for(int i = 0; i < PARK_MAX; i++) if(g_park_valid[i] && (i >= limit || !bundle_same_settings(&g_park[i], b))) { gone[ngone++] = g_park[i]; g_park_valid[i] = 0; }
	for(int i = 0; i < limit && slot < 0; i++) if(!g_park_valid[i]) slot = i;
	if(slot < 0) {          /* full: the one parked longest makes room */
		slot = 0;
		for(int i = 1; i < limit; i++) if((int)(g_park_age[i] - g_park_age[slot]) < 0) slot = i;
		gone[ngone++] = g_park[slot];
	}
	g_park[slot] = *b; g_park_valid[slot] = 1; g_park_age[slot] = ++g_park_clock;
	pthread_mutex_unlock(&g_park_mu);
	for(int i = 0; i < ngone; i++) bundle_destroy(&gone[i]);
	memset(b, 0, sizeof *b);
}

static void release_engine(FLAC__StreamEncoder *e)
{
	struct FLAC__StreamEncoderPrivate *p = PRIV(e);
	if(p->bring_started) { pthread_join(p->bring_th, 0); p->bring_started = 0; }
	if(p->worker_started) {
		pthread_mutex_lock(&p->mu);
		p->worker_quit = 1;
		pthread_cond_broadcast(&p->cv);
		pthread_mutex_unlock(&p->mu);
		pthread_join(p->worker, 0);
		pthread_mutex_destroy(&p->mu);
		pthread_cond_destroy(&p->cv);
		p->worker_started = 0; p->worker_quit = 0;
	}
	stage_pool_destroy(p->pool); p->pool = 0;
	{
		struct engine_bundle b;
		memset(&b, 0, sizeof b);
		b.gpu = p->gpu; b.cfg = p->bring_cfg; b.windows = p->windows; b.wcount = p->wcount; b.raw_bytes = p->raw_bytes; b.out_cap = p->out_cap;
		int whole = 1;
		for(int i = 0; i < NSLOT; i++) {
			b.raw[i] = p->slot[i].raw; b.out[i] = p->slot[i].out; b.fb[i] = p->slot[i].frame_bytes;
			if(!b.raw[i] || !b.out[i] || !b.fb[i]) whole = 0;
			b.registered[2 * i] = p->registered[2 * i]; b.registered[2 * i + 1] = p->registered[2 * i + 1];
			p->slot[i].raw = 0; p->slot[i].out = 0; p->slot[i].frame_bytes = 0; p->slot[i].state = 0;
			p->registered[2 * i] = p->registered[2 * i + 1] = 0;
		}
		p->gpu = 0; p->windows = 0; p->wcount = 0;
		if(b.gpu && !p->engine_failed && whole && parking_enabled()) park_put(&b);
		else bundle_destroy(&b);
	}
	p->engine_on = 0; p->bring_done = 0; p->engine_failed = 0;
	p->out_cap = 0;
	free(p->tail_windows); p->tail_windows = 0;
	p->staged = 0; p->cur = 0; p->inflight = 0; p->col_slot = 0; p->submit_failed = 0;
	if(PROT(e)->metadata) { free(PROT(e)->metadata); PROT(e)->metadata = 0; PROT(e)->num_metadata_blocks = 0; }
}

void FLAC__stream_encoder_delete(FLAC__StreamEncoder *e)
{
	if(!e) return;
	PRIV(e)->is_being_deleted = 1;            /* finish() then only releases, it does not flush (:623-630) */
	(void)FLAC__stream_encoder_finish(e);
	release_engine(e);
	free(e->private_);
	free(e->protected_);
	free(e);
}

/* ------------------------------------------------------------------------------------------------
 * setters: legal only while UNINITIALIZED (stream_encoder.c:1782-2297)
 * ---------------------------------------------------------------------------------------------- */
#define SETTER(name, type, stmt) \
	FLAC__bool FLAC__stream_encoder_##name(FLAC__StreamEncoder *e, type value) { \
		if(PROT(e)->state != FLAC__STREAM_ENCODER_UNINITIALIZED) return 0; \
		stmt; return 1; }

SETTER(set_verify, FLAC__bool, PROT(e)->s.verify = value)
SETTER(set_streamable_subset, FLAC__bool, PROT(e)->s.streamable_subset = value)
SETTER(set_do_md5, FLAC__bool, PROT(e)->s.do_md5 = value)
SETTER(set_channels, uint32_t, PROT(e)->s.channels = value)
SETTER(set_bits_per_sample, uint32_t, PROT(e)->s.bits_per_sample = value)
SETTER(set_sample_rate, uint32_t, PROT(e)->s.sample_rate = value)
SETTER(set_blocksize, uint32_t, PROT(e)->s.blocksize = value)
SETTER(set_do_mid_side_stereo, FLAC__bool, PROT(e)->s.do_mid_side_stereo = value)
SETTER(set_loose_mid_side_stereo, FLAC__bool, PROT(e)->s.loose_mid_side_stereo = value)
SETTER(set_max_lpc_order, uint32_t, PROT(e)->s.max_lpc_order = value)
SETTER(set_qlp_coeff_precision, uint32_t, PROT(e)->s.qlp_coeff_precision = value)
SETTER(set_do_qlp_coeff_prec_search, FLAC__bool, PROT(e)->s.do_qlp_coeff_prec_search = value)
SETTER(set_do_escape_coding, FLAC__bool, (void)value)                 /* deprecated in the reference too (:2107-2114) */
SETTER(set_do_exhaustive_model_search, FLAC__bool, PROT(e)->s.do_exhaustive_model_search = value)
SETTER(set_min_residual_partition_order, uint32_t, PROT(e)->s.min_residual_partition_order = value)
SETTER(set_max_residual_partition_order, uint32_t, PROT(e)->s.max_residual_partition_order = value)
SETTER(set_rice_parameter_search_dist, uint32_t, (void)value)         /* deprecated (:2174-2188) */
SETTER(set_limit_min_bitrate, FLAC__bool, PROT(e)->s.limit_min_bitrate = value)
SETTER(disable_instruction_set, uint32_t, (void)value)                /* CPU dispatch mask: meaningless here */
SETTER(disable_constant_subframes, FLAC__bool, PROT(e)->s.disable_constant_subframes = value)
SETTER(disable_fixed_subframes, FLAC__bool, PROT(e)->s.disable_fixed_subframes = value)
SETTER(disable_verbatim_subframes, FLAC__bool, PROT(e)->s.disable_verbatim_subframes = value)

FLAC__bool FLAC__stream_encoder_set_ogg_serial_number(FLAC__StreamEncoder *e, long value)
{
	if(PROT(e)->state != FLAC__STREAM_ENCODER_UNINITIALIZED) return 0;      /* :1781-1797 */
	PRIV(e)->ogg.serial_number = value;
	return 1;
}

FLAC__bool FLAC__stream_encoder_set_compression_level(FLAC__StreamEncoder *e, uint32_t value)
{
	if(PROT(e)->state != FLAC__STREAM_ENCODER_UNINITIALIZED) return 0;
	flacgpu_host_settings_level(&PROT(e)->s, value);
	return 1;
}

FLAC__bool FLAC__stream_encoder_set_apodization(FLAC__StreamEncoder *e, const char *specification)
{
	if(PROT(e)->state != FLAC__STREAM_ENCODER_UNINITIALIZED || !specification) return 0;
	flacgpu_host_settings_apodization(&PROT(e)->s, specification);
	return 1;
}

uint32_t FLAC__stream_encoder_set_num_threads(FLAC__StreamEncoder *e, uint32_t value)
{
	/* accepted and recorded; parallelism is the GPU batch, not a host thread pool (:2151-2172) */
	if(PROT(e)->state != FLAC__STREAM_ENCODER_UNINITIALIZED) return FLAC__STREAM_ENCODER_SET_NUM_THREADS_ALREADY_INITIALIZED;
	if(value > 64) return FLAC__STREAM_ENCODER_SET_NUM_THREADS_TOO_MANY_THREADS;
	PROT(e)->num_threads = value ? value : 1;
	return FLAC__STREAM_ENCODER_SET_NUM_THREADS_OK;
}

FLAC__bool FLAC__stream_encoder_set_total_samples_estimate(FLAC__StreamEncoder *e, FLAC__uint64 value)
{
	if(PROT(e)->state != FLAC__STREAM_ENCODER_UNINITIALIZED) return 0;
	const FLAC__uint64 cap = ((FLAC__uint64)1 << 36) - 1;
	PROT(e)->s.total_samples_estimate = value < cap ? value : cap;
	return 1;
}

FLAC__bool FLAC__stream_encoder_set_metadata(FLAC__StreamEncoder *e, FLAC__StreamMetadata **metadata, uint32_t num_blocks)
{
	if(PROT(e)->state != FLAC__STREAM_ENCODER_UNINITIALIZED) return 0;
	if(!metadata) num_blocks = 0;
	free(PROT(e)->metadata);                  /* the pointer ARRAY is copied, the blocks stay the client's (:2202-2231) */
	PROT(e)->metadata = 0; PROT(e)->num_metadata_blocks = 0;
	if(num_blocks) {
		FLAC__StreamMetadata **m = malloc(sizeof *m * num_blocks);
		if(!m) return 0;
		memcpy(m, metadata, sizeof *m * num_blocks);
		PROT(e)->metadata = m; PROT(e)->num_metadata_blocks = num_blocks;
	}
	if(num_blocks >= (1u << 16)) return 0;    /* the Ogg mapping's 16-bit header-packet count (:2228, ogg_encoder_aspect.c:75) */
	PRIV(e)->ogg.num_metadata = num_blocks;
	return 1;
}

/* ------------------------------------------------------------------------------------------------
 * getters (stream_encoder.c:2299-2511)
 * ---------------------------------------------------------------------------------------------- */
FLAC__StreamEncoderState FLAC__stream_encoder_get_state(const FLAC__StreamEncoder *e) { return PROT(e)->state; }
/* :2308-2319.  Verify requested but the encoder not initialised: the reference has no decoder object yet and answers
 * MEMORY_ALLOCATION_ERROR (8).  Afterwards the decoder here is a frame decoder without a state machine of its own:
 * "searching for the next frame" (FLAC__STREAM_DECODER_SEARCH_FOR_FRAME_SYNC, 2). */
FLAC__StreamDecoderState FLAC__stream_encoder_get_verify_decoder_state(const FLAC__StreamEncoder *e)
{
	if(!PROT(e)->s.verify) return FLAC__STREAM_DECODER_UNINITIALIZED;
	return PRIV(e)->engine_on ? 2 : 8;            /* (not on the engine pointer: the bring-up thread sets it, under the lock, some time after init) */
}
/* :2318-2330: for VERIFY_DECODER_ERROR the reference answers with its verify decoder's state string; the frame decoder here
 * has one state to report, the one get_verify_decoder_state gives */
const char *FLAC__stream_encoder_get_resolved_state_string(const FLAC__StreamEncoder *e)
{
	if(PROT(e)->state != FLAC__STREAM_ENCODER_VERIFY_DECODER_ERROR) return FLAC__StreamEncoderStateString[PROT(e)->state];
	return "FLAC__STREAM_DECODER_SEARCH_FOR_FRAME_SYNC";
}
void FLAC__stream_encoder_get_verify_decoder_error_stats(const FLAC__StreamEncoder *e, FLAC__uint64 *absolute_sample, uint32_t *frame_number, uint32_t *channel, uint32_t *sample, FLAC__int32 *expected, FLAC__int32 *got)
{
	const flacgpu_host_verify_result *v = &PRIV(e)->verify_stats;           /* :2327-2343 */
	if(absolute_sample) *absolute_sample = v->absolute_sample;
	if(frame_number) *frame_number = v->frame_number;
	if(channel) *channel = v->channel;
	if(sample) *sample = v->sample;
	if(expected) *expected = v->expected;
	if(got) *got = v->got;
}
#define GETTER(name, type, expr) type FLAC__stream_encoder_##name(const FLAC__StreamEncoder *e) { return (type)(expr); }
GETTER(get_verify, FLAC__bool, PROT(e)->s.verify)
GETTER(get_streamable_subset, FLAC__bool, PROT(e)->s.streamable_subset)
GETTER(get_do_md5, FLAC__bool, PROT(e)->s.do_md5)
GETTER(get_channels, uint32_t, PROT(e)->s.channels)
GETTER(get_bits_per_sample, uint32_t, PROT(e)->s.bits_per_sample)
GETTER(get_sample_rate, uint32_t, PROT(e)->s.sample_rate)
GETTER(get_do_mid_side_stereo, FLAC__bool, PROT(e)->s.do_mid_side_stereo)
GETTER(get_loose_mid_side_stereo, FLAC__bool, PROT(e)->s.loose_mid_side_stereo)
GETTER(get_max_lpc_order, uint32_t, PROT(e)->s.max_lpc_order)
GETTER(get_qlp_coeff_precision, uint32_t, PROT(e)->s.qlp_coeff_precision)
GETTER(get_do_qlp_coeff_prec_search, FLAC__bool, PROT(e)->s.do_qlp_coeff_prec_search)
GETTER(get_do_escape_coding, FLAC__bool, PROT(e)->s.do_escape_coding)
GETTER(get_do_exhaustive_model_search, FLAC__bool, PROT(e)->s.do_exhaustive_model_search)
GETTER(get_min_residual_partition_order, uint32_t, PROT(e)->s.min_residual_partition_order)
GETTER(get_max_residual_partition_order, uint32_t, PROT(e)->s.max_residual_partition_order)
GETTER(get_num_threads, uint32_t, PROT(e)->num_threads)
GETTER(get_rice_parameter_search_dist, uint32_t, PROT(e)->s.rice_parameter_search_dist)
GETTER(get_total_samples_estimate, FLAC__uint64, PROT(e)->s.total_samples_estimate)
GETTER(get_limit_min_bitrate, FLAC__bool, PROT(e)->s.limit_min_bitrate)
/* during a write callback this is the block size of the frame being delivered (the reference shrinks
 * protected_->blocksize for the final short block, stream_encoder.c:1704) */
uint32_t FLAC__stream_encoder_get_blocksize(const FLAC__StreamEncoder *e)
{
	return PROT(e)->state == FLAC__STREAM_ENCODER_OK && PRIV(e)->frame_blocksize ? PRIV(e)->frame_blocksize : PROT(e)->s.blocksize;
}

/* ------------------------------------------------------------------------------------------------
 * metadata serialisation (stream_encoder_framing.c:46-243): big-endian fields, byte aligned throughout
 * ---------------------------------------------------------------------------------------------- */
typedef struct { uint8_t *p; size_t n, cap; int bad; } bytebuf;

static void bb_put(bytebuf *b, const void *src, size_t len)
{
	if(b->bad) return;
	if(b->n + len > b->cap) {
		size_t nc = b->cap ? b->cap * 2 : 256;
		while(nc < b->n + len) nc *= 2;
		uint8_t *q = realloc(b->p, nc);
		if(!q) { b->bad = 1; return; }
		b->p = q; b->cap = nc;
	}
	if(src) memcpy(b->p + b->n, src, len); else memset(b->p + b->n, 0, len);
	b->n += len;
}
static void bb_be(bytebuf *b, uint64_t v, unsigned bytes) { uint8_t t[8]; for(unsigned i = 0; i < bytes; i++) t[i] = (uint8_t)(v >> (8 * (bytes - 1 - i))); bb_put(b, t, bytes); }
static void bb_le32(bytebuf *b, uint32_t v) { uint8_t t[4] = {(uint8_t)v, (uint8_t)(v >> 8), (uint8_t)(v >> 16), (uint8_t)(v >> 24)}; bb_put(b, t, 4); }

/* returns 0 when the block cannot be framed (length field disagrees with the content, or >= 2^24) */
static int serialise_block(const FLAC__StreamMetadata *m, bytebuf *b)
{
	const uint32_t vlen = (uint32_t)strlen(FLAC__VENDOR_STRING);
	uint32_t length = m->length;
	if(m->type == FLAC__METADATA_TYPE_VORBIS_COMMENT) length = length - m->data.vorbis_comment.vendor_string.length + vlen;
	if(length >= (1u << 24)) return 0;
	const size_t start = b->n;
	bb_be(b, ((uint32_t)(m->is_last ? 1 : 0) << 7) | ((uint32_t)m->type & 0x7f), 1);
	bb_be(b, length, 3);
	switch(m->type) {
		case FLAC__METADATA_TYPE_STREAMINFO: {
			const FLAC__StreamMetadata_StreamInfo *si = &m->data.stream_info;
			const uint64_t total = si->total_samples >= ((uint64_t)1 << 36) ? 0 : si->total_samples;
			bb_be(b, si->min_blocksize, 2); bb_be(b, si->max_blocksize, 2);
			bb_be(b, si->min_framesize, 3); bb_be(b, si->max_framesize, 3);
			/* 20 bits rate | 3 bits channels-1 | 5 bits bps-1 | 36 bits total samples */
			bb_be(b, ((uint64_t)si->sample_rate << 44) | ((uint64_t)(si->channels - 1) << 41) | ((uint64_t)(si->bits_per_sample - 1) << 36) | total, 8);
			bb_put(b, si->md5sum, 16);
			break;
		}
		case FLAC__METADATA_TYPE_PADDING: bb_put(b, 0, m->length); break;
		case FLAC__METADATA_TYPE_APPLICATION:
			if(m->length < 4) return 0;
			bb_put(b, m->data.application.id, 4);
			bb_put(b, m->data.application.data, m->length - 4);
			break;
		case FLAC__METADATA_TYPE_SEEKTABLE:
			for(uint32_t i = 0; i < m->data.seek_table.num_points; i++) {
				const FLAC__StreamMetadata_SeekPoint *sp = &m->data.seek_table.points[i];
				bb_be(b, sp->sample_number, 8); bb_be(b, sp->stream_offset, 8); bb_be(b, sp->frame_samples, 2);
			}
			break;
		case FLAC__METADATA_TYPE_VORBIS_COMMENT:
			bb_le32(b, vlen); bb_put(b, FLAC__VENDOR_STRING, vlen);       /* our vendor string replaces the client's */
			bb_le32(b, m->data.vorbis_comment.num_comments);
			for(uint32_t i = 0; i < m->data.vorbis_comment.num_comments; i++) {
				bb_le32(b, m->data.vorbis_comment.comments[i].length);
				bb_put(b, m->data.vorbis_comment.comments[i].entry, m->data.vorbis_comment.comments[i].length);
			}
			break;
		case FLAC__METADATA_TYPE_CUESHEET: {
			const FLAC__StreamMetadata_CueSheet *cs = &m->data.cue_sheet;
			bb_put(b, cs->media_catalog_number, 128);
			bb_be(b, cs->lead_in, 8);
			bb_be(b, cs->is_cd ? 0x80 : 0, 1); bb_put(b, 0, 258);             /* 1 bit is_cd + 7+258*8 reserved bits */
			bb_be(b, cs->num_tracks, 1);
			for(uint32_t i = 0; i < cs->num_tracks; i++) {
				const FLAC__StreamMetadata_CueSheet_Track *t = &cs->tracks[i];
				bb_be(b, t->offset, 8); bb_be(b, t->number, 1); bb_put(b, t->isrc, 12);
				bb_be(b, ((uint32_t)t->type << 7) | ((uint32_t)t->pre_emphasis << 6), 1); bb_put(b, 0, 13);   /* 2 flags + 6+13*8 reserved bits */
				bb_be(b, t->num_indices, 1);
				for(uint32_t j = 0; j < t->num_indices; j++) { bb_be(b, t->indices[j].offset, 8); bb_be(b, t->indices[j].number, 1); bb_put(b, 0, 3); }
			}
			break;
		}
		case FLAC__METADATA_TYPE_PICTURE: {
			const FLAC__StreamMetadata_Picture *pic = &m->data.picture;
			const size_t ml = strlen(pic->mime_type), dl = strlen((const char *)pic->description);
			bb_be(b, (uint32_t)pic->type, 4);
			bb_be(b, ml, 4); bb_put(b, pic->mime_type, ml);
			bb_be(b, dl, 4); bb_put(b, pic->description, dl);
			bb_be(b, pic->width, 4); bb_be(b, pic->height, 4); bb_be(b, pic->depth, 4); bb_be(b, pic->colors, 4);
			bb_be(b, pic->data_length, 4); bb_put(b, pic->data, pic->data_length);
			break;
		}
		default: bb_put(b, m->data.unknown.data, m->length); break;
	}
	if(b->bad) return 0;
	return b->n - start == (size_t)length + 4;    /* the declared length must be the written length (framing.c:233-240) */
}

/* ------------------------------------------------------------------------------------------------
 * legality checks of client metadata (format.c:241-520), as far as init uses them
 * ---------------------------------------------------------------------------------------------- */
static unsigned utf8_len(const uint8_t *u)           /* shortest-form UTF-8 only; 0 = invalid (format.c:322) */
{
	if(!(u[0] & 0x80)) return 1;
	unsigned n = (u[0] & 0xE0) == 0xC0 ? 2 : (u[0] & 0xF0) == 0xE0 ? 3 : (u[0] & 0xF8) == 0xF0 ? 4 : (u[0] & 0xFC) == 0xF8 ? 5 : (u[0] & 0xFE) == 0xFC ? 6 : 0;
	if(!n) return 0;
	for(unsigned i = 1; i < n; i++) if((u[i] & 0xC0) != 0x80) return 0;
	if(n == 2 && (u[0] & 0xFE) == 0xC0) return 0;
	if(n == 3 && ((u[0] == 0xE0 && (u[1] & 0xE0) == 0x80) || (u[0] == 0xED && (u[1] & 0xE0) == 0xA0) || (u[0] == 0xEF && u[1] == 0xBF && (u[2] & 0xFE) == 0xBE))) return 0;
	if(n > 3 && u[0] == (uint8_t)(0xFF << (8 - n)) && (u[1] & (0xFF << (8 - n) & 0xFF)) == 0x80) return 0;   /* overlong 4/5/6 */
	return n;
}
static int seektable_is_legal(const FLAC__StreamMetadata_SeekTable *st)
{
	if((uint64_t)st->num_points * 18 >= (1u << 24)) return 0;
	for(uint32_t i = 1; i < st->num_points; i++)
		if(st->points[i].sample_number != FLAC__STREAM_METADATA_SEEKPOINT_PLACEHOLDER && st->points[i].sample_number <= st->points[i - 1].sample_number) return 0;
	return 1;
}
static int seekpoint_cmp(const void *a, const void *b)
{
	const FLAC__uint64 l = ((const FLAC__StreamMetadata_SeekPoint *)a)->sample_number, r = ((const FLAC__StreamMetadata_SeekPoint *)b)->sample_number;
	return l == r ? 0 : l < r ? -1 : 1;
}
static void seektable_sort(FLAC__StreamMetadata_SeekTable *st)    /* sort, drop duplicates, pad with placeholders (format.c:283-313) */
{
	if(!st->num_points) return;
	qsort(st->points, st->num_points, sizeof *st->points, seekpoint_cmp);
	uint32_t j = 0;
	for(uint32_t i = 0; i < st->num_points; i++) {
		if(st->points[i].sample_number != FLAC__STREAM_METADATA_SEEKPOINT_PLACEHOLDER && j > 0 && st->points[i].sample_number == st->points[j - 1].sample_number) continue;
		st->points[j++] = st->points[i];
	}
	for(; j < st->num_points; j++) { st->points[j].sample_number = FLAC__STREAM_METADATA_SEEKPOINT_PLACEHOLDER; st->points[j].stream_offset = 0; st->points[j].frame_samples = 0; }
}
static int cuesheet_is_legal(const FLAC__StreamMetadata_CueSheet *cs)
{
	const int cd = cs->is_cd;
	if(cd && (cs->lead_in < 2 * 44100 || cs->lead_in % 588)) return 0;
	if(cs->num_tracks == 0) return 0;
	if(cd && cs->tracks[cs->num_tracks - 1].number != 170) return 0;
	for(uint32_t i = 0; i < cs->num_tracks; i++) {
		const FLAC__StreamMetadata_CueSheet_Track *t = &cs->tracks[i];
		if(t->number == 0) return 0;
		if(cd && !((t->number >= 1 && t->number <= 99) || t->number == 170)) return 0;
		if(cd && t->offset % 588) return 0;
		if(i + 1 < cs->num_tracks && (t->num_indices == 0 || t->indices[0].number > 1)) return 0;
		for(uint32_t j = 0; j < t->num_indices; j++) {
			if(cd && t->indices[j].offset % 588) return 0;
			if(j && t->indices[j].number != t->indices[j - 1].number + 1) return 0;
		}
	}
	return 1;
}
static int picture_is_legal(const FLAC__StreamMetadata_Picture *pic)
{
	for(const char *c = pic->mime_type; *c; c++) if(*c < 0x20 || *c > 0x7e) return 0;
	for(const uint8_t *d = pic->description; *d;) { const unsigned n = utf8_len(d); if(!n) return 0; d += n; }
	return 1;
}

/* ------------------------------------------------------------------------------------------------
 * stream output: the bookkeeping of write_frame_ / write_bitbuffer_ (stream_encoder.c:2988-3136)
 * ---------------------------------------------------------------------------------------------- */
static int ogg_write_proxy(void *encoder, const uint8_t *buf, size_t bytes, uint32_t samples, uint32_t current_frame, void *client_data)
{
	FLAC__StreamEncoder *e = encoder;
	return PRIV(e)->write_cb(e, buf, bytes, samples, current_frame, client_data) == FLAC__STREAM_ENCODER_WRITE_STATUS_OK;
}
static FLAC__StreamEncoderInitStatus write_preamble(FLAC__StreamEncoder *e);
static int emit(FLAC__StreamEncoder *e, const uint8_t *buf, size_t bytes, uint32_t samples)
{
	struct FLAC__StreamEncoderPrivate *p = PRIV(e);
	FLAC__uint64 pos = 0;
	if(p->preamble_pending) {            /* the first frame of a stream whose engine came up beside init_*(): the stream's head first */
		/* (the metadata packets are not the stream's last block and have no block size, whatever the frame that triggered them is:
		 * an Ogg stream whose first frame is also its last must not close on its STREAMINFO page) */
		const int is_last = p->emit_is_last;
		const uint32_t fbs = p->frame_blocksize;
		p->preamble_pending = 0; p->emit_is_last = 0; p->frame_blocksize = 0;
		const FLAC__StreamEncoderInitStatus st = write_preamble(e);
		p->emit_is_last = is_last; p->frame_blocksize = fbs;
		if(st != FLAC__STREAM_ENCODER_INIT_STATUS_OK) return 0;
	}
	/* (the tell callback is called where write_frame_ calls it: for a metadata write, :3054, and -- once, lazily -- for a frame
	 * that holds a seek point of the template, :3083; not in front of every frame) */
	if(samples == 0) {
		/* watch STREAMINFO and the first SEEKTABLE go by to learn their offsets */
		const unsigned type = buf[0] & 0x7f;
		if(p->tell_cb && p->tell_cb(e, &pos, p->client_data) == FLAC__STREAM_ENCODER_TELL_STATUS_ERROR) { PROT(e)->state = FLAC__STREAM_ENCODER_CLIENT_ERROR; return 0; }
		if(type == FLAC__METADATA_TYPE_STREAMINFO) PROT(e)->streaminfo_offset = pos;
		else if(type == FLAC__METADATA_TYPE_SEEKTABLE && PROT(e)->seektable_offset == 0) PROT(e)->seektable_offset = pos;
	}
	if(p->seek_table && PROT(e)->audio_offset > 0 && p->seek_table->num_points > 0) {
		/* fill every template seek point that falls into this frame (:3070-3103) */
		const uint32_t bs = FLAC__stream_encoder_get_blocksize(e);
		const FLAC__uint64 first = p->samples_written, last = first + bs - 1;
		for(uint32_t i = p->first_seekpoint_to_check; i < p->seek_table->num_points; i++) {
			const FLAC__uint64 t = p->seek_table->points[i].sample_number;
			if(t > last) break;
			if(t >= first) {
				if(pos == 0 && p->tell_cb && p->tell_cb(e, &pos, p->client_data) == FLAC__STREAM_ENCODER_TELL_STATUS_ERROR) { PROT(e)->state = FLAC__STREAM_ENCODER_CLIENT_ERROR; return 0; }
				p->seek_table->points[i].sample_number = first;
				p->seek_table->points[i].stream_offset = pos - PROT(e)->audio_offset;
				p->seek_table->points[i].frame_samples = bs;
			}
			p->first_seekpoint_to_check++;
		}
	}
	if(p->is_ogg) {
		/* one packet per write; pages go to the client as (header, body) write pairs (:3106-3118) */
		if(!fgh_ogg_aspect_write(&p->ogg, buf, bytes, samples, p->current_frame_number, p->emit_is_last, ogg_write_proxy, e, p->client_data)) {
			PROT(e)->state = FLAC__STREAM_ENCODER_CLIENT_ERROR;
			return 0;
		}
	}
	else if(p->write_cb(e, buf, bytes, samples, p->current_frame_number, p->client_data) != FLAC__STREAM_ENCODER_WRITE_STATUS_OK) {
		PROT(e)->state = FLAC__STREAM_ENCODER_CLIENT_ERROR;
		return 0;
	}
	p->bytes_written += bytes;
	p->samples_written += samples;
	if(p->current_frame_number + 1 > p->frames_written) p->frames_written = p->current_frame_number + 1;
	if(samples > 0) {
		FLAC__StreamMetadata_StreamInfo *si = &p->streaminfo.data.stream_info;
		if(bytes < si->min_framesize) si->min_framesize = (uint32_t)bytes;
		if(bytes > si->max_framesize) si->max_framesize = (uint32_t)bytes;
	}
	return 1;
}

static int emit_block(FLAC__StreamEncoder *e, const FLAC__StreamMetadata *m)
{
	bytebuf b = {0, 0, 0, 0};
	int ok = serialise_block(m, &b);
	if(!ok) PROT(e)->state = b.bad ? FLAC__STREAM_ENCODER_MEMORY_ALLOCATION_ERROR : FLAC__STREAM_ENCODER_FRAMING_ERROR;
	else ok = emit(e, b.p, b.n, 0);
	free(b.p);
	return ok;
}

/* The worker hands a batch to the engine's asynchronous entry (the MD5 chain has been over exactly its sample bytes, in stream
 * order, :3448).  Returns 0 when nothing went in flight (b->total holds the error). */
static int engine_submit_slot(FLAC__StreamEncoder *e, struct batch_slot *b)
{
	struct FLAC__StreamEncoderPrivate *p = PRIV(e);
	const flacgpu_host_settings *s = &PROT(e)->s;
	const double t0 = p->timing ? now_s() : 0;
	if(p->bring_result != FLACGPU_OK) { b->total = p->bring_result; return 0; }       /* (the worker has waited for bring_done) */
	const float *tw = 0;
	if(b->tail && s->max_lpc_order > 0) {
		/* windows are recomputed for the short block, as resize_buffers_ does at finish (:1703-1711) */
		float *w = realloc(p->tail_windows, sizeof(float) * s->num_apodizations * b->tail);
		if(!w) { b->total = FLACGPU_ERR_ALLOC; return 0; }
		p->tail_windows = w;
		flacgpu_host_windows(s, b->tail, w);
		tw = w;
	}
	b->vres.status = 0;
	const int r = flacgpu_submit_batch_raw(p->gpu, b->raw, &p->rawfmt, b->nframes, b->first_frame, b->tail, tw, b->out, p->out_cap, b->frame_bytes);
	if(p->timing) p->t_encode += now_s() - t0;
	if(r != FLACGPU_OK) { b->total = r; return 0; }
	return 1;
}
/* the oldest batch in flight comes back (the worker; blocks until the engine has it) */
static void engine_collect_slot(FLAC__StreamEncoder *e, struct batch_slot *b)
{
	struct FLAC__StreamEncoderPrivate *p = PRIV(e);
	const flacgpu_host_settings *s = &PROT(e)->s;
	const double t0 = p->timing ? now_s() : 0;
	b->total = flacgpu_collect(p->gpu);
	if(p->timing) p->t_encode += now_s() - t0;
	if(b->total >= 0 && s->verify) {
		/* write_bitbuffer_ verifies every frame before it is written (:3000-3018): here the engine has decoded the batch again on
		 * the device, next to the staged input (flacgpu_set_verify); this is its verdict */
		flacgpu_verify_result v;
		if(flacgpu_last_verify_result(p->gpu, &v) != FLACGPU_OK) { b->total = FLACGPU_ERR_LAUNCH; return; }
		b->vres.status = v.status; b->vres.frame_number = v.frame_number; b->vres.channel = v.channel; b->vres.sample = v.sample;
		b->vres.absolute_sample = v.absolute_sample; b->vres.expected = v.expected; b->vres.got = v.got;
	}
}
/* HIP runtime, engine, page-locked slots: on a thread of its own (see bring_* in the private struct), or inside init_*() */
static void *bringup_main(void *arg)
{
	FLAC__StreamEncoder *e = arg;
	struct FLAC__StreamEncoderPrivate *p = PRIV(e);
	const flacgpu_host_settings *s = &PROT(e)->s;
	const double t0 = now_s();
	flacgpu_ctx *gpu = 0;
	int r = FLACGPU_OK;
	float *windows = 0;
	if(s->max_lpc_order > 0) {
		windows = malloc(sizeof(float) * s->num_apodizations * s->blocksize);
		if(!windows) r = FLACGPU_ERR_ALLOC; else flacgpu_host_windows(s, s->blocksize, windows);
	}
	if(r == FLACGPU_OK) r = flacgpu_create(&p->bring_cfg, windows, &gpu);
	p->windows = windows; p->wcount = windows ? (size_t)s->num_apodizations * s->blocksize : 0;     /* kept: the key of a parked engine */
	if(r == FLACGPU_OK) (void)flacgpu_set_phase_timing(gpu, 0);      /* nobody reads per-kernel times through the libFLAC API: no event records */
	if(r == FLACGPU_OK && s->verify) r = flacgpu_set_verify(gpu, 1);
	const double t1 = now_s();
	if(r == FLACGPU_OK)
		for(int i = 0; i < NSLOT; i++) {    /* a slot that cannot be page-locked still works: its copies are staged by the runtime */
			p->registered[2 * i] = flacgpu_host_register(p->slot[i].raw, p->raw_bytes) == FLACGPU_OK;
			p->registered[2 * i + 1] = flacgpu_host_register(p->slot[i].out, p->out_cap) == FLACGPU_OK;
		}
	if(r != FLACGPU_OK) {
		fprintf(stderr, "libFLACgpu: cannot create the GPU frame engine: %s\n", flacgpu_strerror(r));
		if(gpu) { flacgpu_destroy(gpu); gpu = 0; }
	}
	p->t_init_engine = t1 - t0; p->t_init_pinned = now_s() - t1;
	pthread_mutex_lock(&p->mu);
	p->gpu = gpu; p->bring_result = r; p->bring_done = 1;
	pthread_cond_broadcast(&p->cv);
	pthread_mutex_unlock(&p->mu);
	return 0;
}

static void *worker_main(void *arg)
{
	FLAC__StreamEncoder *e = arg;
	struct FLAC__StreamEncoderPrivate *p = PRIV(e);
	const flacgpu_host_settings *s = &PROT(e)->s;
	const size_t full = (size_t)p->batch_frames * s->blocksize, sample_bytes = (size_t)s->channels * p->width;
	pthread_mutex_lock(&p->mu);
	for(;;) {
		int k = -1;
		if(s->do_md5) {
			/* Stream order: the batch the chain is in (md5_slot) is also the next one to be submitted.  While it is being filled the
			 * chain follows the caller's published progress; once it is submitted the chain finishes it; and while the engine is
			 * still coming up the chain carries on into the other slot (md5_ahead), which the caller is filling by then. */
			const int cur = p->md5_slot;
			const struct batch_slot *b = &p->slot[cur];
			int slot_to_hash = -1;
			size_t from = 0, n = 0;
			if(b->state == 1) {
				const size_t nsamp = (size_t)(b->nframes - 1) * s->blocksize + (b->tail ? b->tail : s->blocksize);
				if(p->md5_done < nsamp) { slot_to_hash = cur; from = p->md5_done; n = nsamp - from; }
				else if(p->bring_done) k = cur;
				else {
					const int o = (cur + 1) % NSLOT;
					size_t avail = p->slot[o].state == 0 && p->pub_slot == o ? p->pub_staged : 0;
					if(avail > full) avail = full;
					if(avail >= p->md5_ahead + MD5_PIECE || (avail == full && avail > p->md5_ahead)) { slot_to_hash = o; from = p->md5_ahead; n = avail - from; }
				}
			}
			else {
				size_t avail = b->state == 0 && p->pub_slot == cur ? p->pub_staged : 0;
				if(avail > full) avail = full;                 /* the sample beyond the batch belongs to the next one */
				if(avail >= p->md5_done + MD5_PIECE || (avail == full && avail > p->md5_done)) { slot_to_hash = cur; from = p->md5_done; n = avail - from; }
			}
			if(slot_to_hash >= 0) {
				if(p->slot[slot_to_hash].state != 1 && n > 16 * MD5_PIECE) n = 16 * MD5_PIECE;      /* look for a submission again every few milliseconds */
				const uint8_t *src = p->slot[slot_to_hash].raw + from * sample_bytes;
				pthread_mutex_unlock(&p->mu);
				const double t0 = p->timing ? now_s() : 0;
				flacgpu_host_md5_update(&p->md5, src, n * sample_bytes);
				if(p->timing) p->t_md5 += now_s() - t0;
				pthread_mutex_lock(&p->mu);
				if(slot_to_hash == cur) p->md5_done += n; else p->md5_ahead += n;
				continue;
			}
		}
		else {
			k = p->slot[p->md5_slot].state == 1 ? p->md5_slot : -1;       /* stream order: md5_slot is the next batch for the engine, MD5 or not */
			if(k >= 0 && !p->bring_done) k = -2;              /* a batch is waiting for the engine */
		}
		if(k >= 0 && p->inflight < NSLOT - 1) {
			/* into flight: its input copy starts now, beside the kernels of the batches in front */
			int went = 0;
			if(p->submit_failed) p->slot[k].total = p->submit_failed;      /* the stream has failed: every later batch fails the same way, unsubmitted */
			else {
				pthread_mutex_unlock(&p->mu);
				went = engine_submit_slot(e, &p->slot[k]);
				pthread_mutex_lock(&p->mu);
			}
			if(went) { if(p->inflight++ == 0) p->col_slot = k; p->slot[k].state = 3; }
			else {
				/* (b->total says why.)  The batches in front stay in flight and are collected in order; this one and all behind it are
				 * failed results, so that col_slot + 1 is always the next batch in flight */
				if(!p->submit_failed) p->submit_failed = p->slot[k].total < 0 ? (int)p->slot[k].total : FLACGPU_ERR_LAUNCH;
				p->slot[k].state = 2;
			}
			p->md5_slot = (k + 1) % NSLOT; p->md5_done = p->md5_ahead; p->md5_ahead = 0;
			pthread_cond_broadcast(&p->cv);
			continue;
		}
		if(p->inflight > 0) {
			/* nothing to hand over (or the engine's ring is full): the oldest batch in flight comes back */
			const int c = p->col_slot;
			pthread_mutex_unlock(&p->mu);
			engine_collect_slot(e, &p->slot[c]);
			pthread_mutex_lock(&p->mu);
			p->slot[c].state = 2;
			p->inflight--; p->col_slot = (c + 1) % NSLOT;
			pthread_cond_broadcast(&p->cv);
			continue;
		}
		if(p->worker_quit && k == -1) break;
		{
			const double t0 = p->timing ? now_s() : 0;
			pthread_cond_wait(&p->cv, &p->mu);
			if(p->timing && !p->bring_done) p->t_bring_wait += now_s() - t0;
		}
	}
	pthread_mutex_unlock(&p->mu);
	return 0;
}
/* hand slot k (nframes staged blocks; `tail` = samples of a short last block, 0: all full) to the worker */
static void submit_slot(FLAC__StreamEncoder *e, int k, uint32_t nframes, uint32_t tail)
{
	struct FLAC__StreamEncoderPrivate *p = PRIV(e);
	struct batch_slot *b = &p->slot[k];
	b->nframes = nframes; b->tail = tail; b->first_frame = p->next_frame_number;
	p->next_frame_number += nframes;
	pthread_mutex_lock(&p->mu);
	b->state = 1;
	p->pub_slot = (k + 1) % NSLOT; p->pub_staged = 1; p->pub_last = 1;      /* the caller goes on in the next slot, behind the overread sample */
	pthread_cond_broadcast(&p->cv);
	pthread_mutex_unlock(&p->mu);
}
/* tell the worker how far the slot being filled has got (MD5 streams only) */
static inline void publish_staged(FLAC__StreamEncoder *e)
{
	struct FLAC__StreamEncoderPrivate *p = PRIV(e);
	if(!PROT(e)->s.do_md5 || p->staged < p->pub_last + MD5_PIECE) return;
	pthread_mutex_lock(&p->mu);
	p->pub_staged = p->pub_last = p->staged;
	pthread_cond_broadcast(&p->cv);
	pthread_mutex_unlock(&p->mu);
}
/* wait for slot k (if it is in flight) and deliver its frames in order on this thread */
static int collect_slot(FLAC__StreamEncoder *e, int k)
{
	struct FLAC__StreamEncoderPrivate *p = PRIV(e);
	struct batch_slot *b = &p->slot[k];
	double t0 = p->timing ? now_s() : 0;
	pthread_mutex_lock(&p->mu);
	if(b->state == 0) { pthread_mutex_unlock(&p->mu); return 1; }
	while(b->state != 2) pthread_cond_wait(&p->cv, &p->mu);
	b->state = 0;
	pthread_mutex_unlock(&p->mu);
	if(p->timing) { const double t1 = now_s(); p->t_wait += t1 - t0; t0 = t1; }
	if(b->total < 0) {
		p->engine_failed = 1;
		fprintf(stderr, "libFLACgpu: the GPU frame engine failed: %s\n", flacgpu_strerror((int)b->total));
		PROT(e)->state = b->total == FLACGPU_ERR_ALLOC ? FLAC__STREAM_ENCODER_MEMORY_ALLOCATION_ERROR : FLAC__STREAM_ENCODER_FRAMING_ERROR;
		return 0;
	}
	const uint32_t N = PROT(e)->s.blocksize;
	const uint8_t *q = b->out;
	for(uint32_t f = 0; f < b->nframes; f++) {
		if(b->vres.status && b->first_frame + f == b->vres.frame_number) {
			/* the frames in front of it were fine and are out; this one is not written (:3005-3016) */
			p->verify_stats = b->vres;
			PROT(e)->state = b->vres.status == 1 ? FLAC__STREAM_ENCODER_VERIFY_MISMATCH_IN_AUDIO_DATA : FLAC__STREAM_ENCODER_VERIFY_DECODER_ERROR;
			p->frame_blocksize = 0;
			return 0;
		}
		const uint32_t samples = (f + 1 == b->nframes && b->tail) ? b->tail : N;
		p->frame_blocksize = samples;
		p->emit_is_last = p->final_batch && f + 1 == b->nframes;            /* is_last_block of write_frame_ (:3038) */
		const int ok = emit(e, q, b->frame_bytes[f], samples);
		p->emit_is_last = 0;
		if(!ok) return 0;
		q += b->frame_bytes[f];
		p->current_frame_number++;
		p->streaminfo.data.stream_info.total_samples += samples;
	}
	p->frame_blocksize = 0;
	if(p->timing) p->t_emit += now_s() - t0;
	return 1;
}

/* ------------------------------------------------------------------------------------------------
 * init (stream_encoder.c:707-1440)
 * ---------------------------------------------------------------------------------------------- */
static FLAC__StreamEncoderInitStatus init_common(FLAC__StreamEncoder *e, FLAC__StreamEncoderReadCallback rcb, FLAC__StreamEncoderWriteCallback wcb, FLAC__StreamEncoderSeekCallback scb,
                                                 FLAC__StreamEncoderTellCallback tcb, FLAC__StreamEncoderMetadataCallback mcb, void *client_data, int is_ogg)
{
	struct FLAC__StreamEncoderPrivate *p = PRIV(e);
	flacgpu_host_settings *s = &PROT(e)->s;
	if(PROT(e)->state != FLAC__STREAM_ENCODER_UNINITIALIZED) return FLAC__STREAM_ENCODER_INIT_STATUS_ALREADY_INITIALIZED;
	if(!wcb || (scb && !tcb)) return FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_CALLBACKS;
	const int st = flacgpu_host_settings_resolve(s);          /* the checks and defaults of :729-829; codes are the enum's */
	if(st != FGH_INIT_OK) return (FLAC__StreamEncoderInitStatus)st;

	if(is_ogg && PROT(e)->metadata) {
		/* no seek table in Ogg FLAC, and a VORBIS_COMMENT goes first (:831-857) */
		uint32_t n = PROT(e)->num_metadata_blocks;
		for(uint32_t i = 0; i < n; i++)
			if(PROT(e)->metadata[i] && PROT(e)->metadata[i]->type == FLAC__METADATA_TYPE_SEEKTABLE) {
				for(n--; i < n; i++) PROT(e)->metadata[i] = PROT(e)->metadata[i + 1];
				break;
			}
		PROT(e)->num_metadata_blocks = n;
		for(uint32_t i = 1; i < n; i++)
			if(PROT(e)->metadata[i] && PROT(e)->metadata[i]->type == FLAC__METADATA_TYPE_VORBIS_COMMENT) {
				FLAC__StreamMetadata *vc = PROT(e)->metadata[i];
				for(; i > 0; i--) PROT(e)->metadata[i] = PROT(e)->metadata[i - 1];
				PROT(e)->metadata[0] = vc;
				break;
			}
	}
	/* client metadata (:866-925) */
	p->seek_table = 0;
	if(!PROT(e)->metadata && PROT(e)->num_metadata_blocks) return FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_METADATA;
	int has_seektable = 0, has_vc = 0, icon1 = 0, icon2 = 0;
	for(uint32_t i = 0; i < PROT(e)->num_metadata_blocks; i++) {
		FLAC__StreamMetadata *m = PROT(e)->metadata[i];
		if(!m) return FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_METADATA;
		switch(m->type) {
			case FLAC__METADATA_TYPE_STREAMINFO: return FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_METADATA;
			case FLAC__METADATA_TYPE_SEEKTABLE:
				if(has_seektable++ || !seektable_is_legal(&m->data.seek_table)) return FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_METADATA;
				p->seek_table = &m->data.seek_table;
				break;
			case FLAC__METADATA_TYPE_VORBIS_COMMENT:
				if(has_vc++) return FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_METADATA;
				break;
			case FLAC__METADATA_TYPE_CUESHEET:
				if(!cuesheet_is_legal(&m->data.cue_sheet)) return FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_METADATA;
				break;
			case FLAC__METADATA_TYPE_PICTURE:
				if(!picture_is_legal(&m->data.picture)) return FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_METADATA;
				if(m->data.picture.type == 1) {     /* the one 32x32 PNG file icon */
					if(icon1++) return FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_METADATA;
					if((strcmp(m->data.picture.mime_type, "image/png") && strcmp(m->data.picture.mime_type, "-->")) || m->data.picture.width != 32 || m->data.picture.height != 32)
						return FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_METADATA;
				}
				else if(m->data.picture.type == 2 && icon2++) return FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_METADATA;
				break;
			default: break;
		}
	}

	/* the GPU engine; features it does not implement are refused, never approximated */
	{
		/* Blocks per GPU batch: FLACGPU_BATCH_FRAMES, or what a budget of staged sample bytes per slot buys (FLACGPU_BATCH_BYTES,
		 * default 16 MiB -- 64 MiB for a stream that announces at least a GiB of samples --: 1024 blocks of 16-bit stereo at 4096 samples,
		 * 7 of 8 x 32-bit x 65535) -- page-locked input slots of
		 * that size and two output slots exist per encoder -- and never more than the stream is said to hold.  Measured on one
		 * 16-bit stereo stream: page-locking and releasing the slots costs more than bigger batches win back on the GPU, which is
		 * an order of magnitude ahead of the host either way (64 MiB: 0.88 G samples/s, 16 MiB: 1.37 G).  No frame is delivered
		 * before its batch is full or finish() is called: INTEGRATION.md, "output latency". */
		const char *env = getenv("FLACGPU_BATCH_FRAMES");
		long bf;
		if(env) bf = strtol(env, 0, 10);
		else {
			const char *eb = getenv("FLACGPU_BATCH_BYTES");
			const double per_block = (double)s->blocksize * s->channels * ((s->bits_per_sample + 7) / 8);
			/* a stream that says it is long (total_samples_estimate: the reference's tool always says) gets 64 MiB slots: the kernels reach
			 * their large-batch rate there (profiles/r03_api_batch.txt: 16 / 32 / 64 / 128 MiB = 6.7 / 8.1 / 8.5 / 7.7 G samples/s), and
			 * four slots' worth of page-locking is paid once in sixteen batches or more */
			const int is_long = s->total_samples_estimate && (double)s->total_samples_estimate * s->channels * ((s->bits_per_sample + 7) / 8) >= 16.0 * 64.0 * 1024 * 1024;
			const double budget = eb ? strtod(eb, 0) : (is_long ? 64.0 : 16.0) * 1024 * 1024;
			bf = (long)(budget / per_block);
			if(bf > 16384) bf = 16384;
			if(bf < 4) bf = 4;
		}
		if(s->total_samples_estimate) {
			const uint64_t need = (s->total_samples_estimate + s->blocksize - 1) / s->blocksize;
			if(need < (uint64_t)bf) bf = (long)need;
		}
		if(bf < 1) bf = 1; else if(bf > 65536) bf = 65536;
		p->batch_frames = (uint32_t)bf;
		memset(&p->verify_stats, 0, sizeof p->verify_stats);
		env = getenv("FLACGPU_DEVICE");
		const int device = env ? atoi(env) : 0;
		p->timing = getenv("FLACGPU_HOST_TIMING") != 0;
		p->t_init_engine = p->t_init_pinned = p->t_stage = p->t_wait = p->t_emit = p->t_md5 = p->t_encode = p->t_release = p->t_bring_wait = 0;
		p->t_start = now_s();
		int parked = 0;
		int r = flacgpu_host_engine_config(s, device, p->batch_frames, &p->bring_cfg);
		if(r == FLACGPU_OK) r = flacgpu_config_check(&p->bring_cfg);       /* what the engine refuses, it refuses here, not on the bring-up thread */
		if(r == FLACGPU_OK) {
			p->width = (s->bits_per_sample + 7) / 8;
			memset(&p->rawfmt, 0, sizeof p->rawfmt);
			p->rawfmt.container_bits = 8 * p->width;            /* little endian, signed, right-justified */
			p->out_cap = flacgpu_config_max_output_bytes(&p->bring_cfg, p->batch_frames);
			p->raw_bytes = (size_t)p->width * s->channels * ((size_t)p->batch_frames * s->blocksize + 1);
			p->windows = 0; p->wcount = 0; p->engine_failed = 0;
			pthread_mutex_lock(&g_park_mu);
			const int maybe = park_any() && parking_enabled();
			pthread_mutex_unlock(&g_park_mu);
			if(maybe) {
				const size_t wcount = s->max_lpc_order > 0 ? (size_t)s->num_apodizations * s->blocksize : 0;
				float *w = wcount ? malloc(sizeof(float) * wcount) : 0;
				struct engine_bundle b;
				if(wcount && w) flacgpu_host_windows(s, s->blocksize, w);
				if((!wcount || w) && park_take(&p->bring_cfg, w, wcount, p->raw_bytes, p->out_cap, &b)) {
					parked = 1;
					p->gpu = b.gpu; p->bring_cfg = b.cfg; p->windows = b.windows; p->wcount = b.wcount; p->raw_bytes = b.raw_bytes; p->out_cap = b.out_cap;
					for(int i = 0; i < NSLOT; i++) {
						p->slot[i].raw = b.raw[i]; p->slot[i].out = b.out[i]; p->slot[i].frame_bytes = b.fb[i]; p->slot[i].state = 0;
						p->registered[2 * i] = b.registered[2 * i]; p->registered[2 * i + 1] = b.registered[2 * i + 1];
					}
					r = flacgpu_set_verify(p->gpu, s->verify ? 1 : 0);
					if(r != FLACGPU_OK) p->engine_failed = 1;       /* do not park it again: the next stream gets a fresh engine */
				}
				free(w);
			}
			for(int i = 0; i < NSLOT && !parked; i++) {
				void *a = 0, *b = 0;
				if(posix_memalign(&a, 4096, p->raw_bytes) != 0) a = 0;
				if(posix_memalign(&b, 4096, p->out_cap) != 0) b = 0;
				p->slot[i].raw = a; p->slot[i].out = b;
				p->slot[i].frame_bytes = malloc(sizeof(uint32_t) * p->batch_frames);
				p->slot[i].state = 0;
				p->registered[2 * i] = p->registered[2 * i + 1] = 0;
				if(!p->slot[i].raw || !p->slot[i].out || !p->slot[i].frame_bytes) r = FLACGPU_ERR_ALLOC;
			}
		}
		p->md5_slot = 0; p->md5_done = 0; p->md5_ahead = 0; p->pub_slot = 0; p->pub_staged = p->pub_last = 0; p->inflight = 0; p->col_slot = 0; p->submit_failed = 0;
		p->bring_started = 0; p->bring_done = 0; p->bring_result = FLACGPU_OK;
		if(r == FLACGPU_OK) {
			pthread_mutex_init(&p->mu, 0);
			pthread_cond_init(&p->cv, 0);
			p->worker_quit = 0;
			if(pthread_create(&p->worker, 0, worker_main, e) == 0) p->worker_started = 1;
			else { pthread_mutex_destroy(&p->mu); pthread_cond_destroy(&p->cv); r = FLACGPU_ERR_ALLOC; }
		}
		if(r == FLACGPU_OK) {
			/* on its own thread when a device node is there to be opened; otherwise here, so that the error is init's */
			const char *es = getenv("FLACGPU_SYNC_INIT");
			const int async = !(es && atoi(es)) && flacgpu_device_probe();
			if(parked) p->bring_done = 1;                       /* the previous stream's engine: nothing to bring up */
			else if(async && pthread_create(&p->bring_th, 0, bringup_main, e) == 0) p->bring_started = 1;
			else {
				bringup_main(e);
				r = p->bring_result;
				if(r != FLACGPU_OK) {          /* (bringup_main has said why) */
					release_engine(e);
					PROT(e)->state = r == FLACGPU_ERR_ALLOC ? FLAC__STREAM_ENCODER_MEMORY_ALLOCATION_ERROR : FLAC__STREAM_ENCODER_FRAMING_ERROR;
					return FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR;
				}
			}
			/* helper threads for the narrowing copy of big process() calls: FLACGPU_STAGE_THREADS (the caller's thread included;
			 * default 4, 1 = none) */
			const char *et = getenv("FLACGPU_STAGE_THREADS");
			long nt = et ? strtol(et, 0, 10) : 4;
			const long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
			if(nt > ncpu) nt = ncpu;
			if(nt > STAGE_MAX_THREADS) nt = STAGE_MAX_THREADS;
			p->pool = nt > 1 ? stage_pool_create((int)nt - 1) : 0;          /* no pool is not an error: the caller's thread does it all */
			p->engine_on = 1;
		}
		if(r != FLACGPU_OK) {
			fprintf(stderr, "libFLACgpu: cannot create the GPU frame engine: %s\n", flacgpu_strerror(r));
			release_engine(e);
			PROT(e)->state = r == FLACGPU_ERR_ALLOC ? FLAC__STREAM_ENCODER_MEMORY_ALLOCATION_ERROR : FLAC__STREAM_ENCODER_FRAMING_ERROR;
			/* state must read UNINITIALIZED-or-error; the reference leaves the error state set here too */
			return FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR;
		}
	}

	p->read_cb = rcb; p->write_cb = wcb; p->seek_cb = scb; p->tell_cb = tcb; p->metadata_cb = mcb; p->client_data = client_data;    /* only once every check has passed (:1126-1131) */
	p->is_ogg = is_ogg; p->final_batch = 0; p->emit_is_last = 0;
	if(is_ogg && !fgh_ogg_aspect_init(&p->ogg)) {                   /* :1121-1124 */
		release_engine(e);
		PROT(e)->state = FLAC__STREAM_ENCODER_OGG_ERROR;
		return FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR;
	}
	p->staged = 0; p->cur = 0; p->current_frame_number = 0; p->next_frame_number = 0; p->frame_blocksize = 0;
	p->first_seekpoint_to_check = 0; p->samples_written = 0;
	PROT(e)->streaminfo_offset = PROT(e)->seektable_offset = PROT(e)->audio_offset = 0;
	PROT(e)->state = FLAC__STREAM_ENCODER_OK;

	memset(&p->streaminfo, 0, sizeof p->streaminfo);
	if(s->do_md5) flacgpu_host_md5_init(&p->md5);
	/* "fLaC", STREAMINFO with the unknowns zeroed, a VORBIS_COMMENT if the client gave none, client blocks (:1335-1425): serialised
	 * now, written now -- or, with the engine still coming up on its own thread, in front of the first frame (see preamble_pending) */
	{
		bytebuf b = {0, 0, 0, 0};
		int ok = 1;
		p->pre_nseg = 0;
		bb_put(&b, "fLaC", 4);
		p->pre_len[p->pre_nseg++] = 4;
		p->streaminfo.type = FLAC__METADATA_TYPE_STREAMINFO;
		p->streaminfo.is_last = 0;
		p->streaminfo.length = 34;
		FLAC__StreamMetadata_StreamInfo *si = &p->streaminfo.data.stream_info;
		si->min_blocksize = si->max_blocksize = s->blocksize;
		si->sample_rate = s->sample_rate; si->channels = s->channels; si->bits_per_sample = s->bits_per_sample;
		si->total_samples = s->total_samples_estimate;
		size_t at = b.n;
		ok = serialise_block(&p->streaminfo, &b);
		p->pre_len[p->pre_nseg++] = b.n - at;
		si->min_framesize = (1u << 24) - 1;
		si->total_samples = 0;
		if(ok && !has_vc) {
			FLAC__StreamMetadata vc;
			memset(&vc, 0, sizeof vc);
			vc.type = FLAC__METADATA_TYPE_VORBIS_COMMENT;
			vc.is_last = PROT(e)->num_metadata_blocks == 0;
			vc.length = 8;
			at = b.n;
			ok = serialise_block(&vc, &b);
			p->pre_len[p->pre_nseg++] = b.n - at;
		}
		int deferrable = PROT(e)->num_metadata_blocks + 3 <= FLACGPU_PRE_MAX_SEGS;
		for(uint32_t i = 0; ok && deferrable && i < PROT(e)->num_metadata_blocks; i++) {
			PROT(e)->metadata[i]->is_last = i + 1 == PROT(e)->num_metadata_blocks;
			at = b.n;
			ok = serialise_block(PROT(e)->metadata[i], &b);
			p->pre_len[p->pre_nseg++] = b.n - at;
		}
		if(!ok) {
			PROT(e)->state = b.bad ? FLAC__STREAM_ENCODER_MEMORY_ALLOCATION_ERROR : FLAC__STREAM_ENCODER_FRAMING_ERROR;
			free(b.p);
			return FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR;
		}
		p->pre_buf = b.p;
		p->preamble_pending = 0;
		if(p->bring_started && deferrable) { p->preamble_pending = 1; return FLAC__STREAM_ENCODER_INIT_STATUS_OK; }
		if(write_preamble(e) != FLAC__STREAM_ENCODER_INIT_STATUS_OK) return FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR;
		if(!deferrable) {
			/* (more blocks than segments: written here, one by one; such a stream waits for its engine first) */
			if(p->bring_started) { pthread_join(p->bring_th, 0); p->bring_started = 0; if(p->bring_result != FLACGPU_OK) { PROT(e)->state = FLAC__STREAM_ENCODER_FRAMING_ERROR; return FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR; } }
			for(uint32_t i = 0; i < PROT(e)->num_metadata_blocks; i++) {
				PROT(e)->metadata[i]->is_last = i + 1 == PROT(e)->num_metadata_blocks;
				if(!emit_block(e, PROT(e)->metadata[i])) return FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR;
			}
			if(p->tell_cb && p->tell_cb(e, &PROT(e)->audio_offset, p->client_data) == FLAC__STREAM_ENCODER_TELL_STATUS_ERROR) {
				PROT(e)->state = FLAC__STREAM_ENCODER_CLIENT_ERROR;
				return FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR;
			}
		}
		return FLAC__STREAM_ENCODER_INIT_STATUS_OK;
	}
}
/* the serialised head goes to the client, one write per block as the reference writes them */
static FLAC__StreamEncoderInitStatus write_preamble(FLAC__StreamEncoder *e)
{
	struct FLAC__StreamEncoderPrivate *p = PRIV(e);
	size_t at = 0;
	int ok = 1;
	for(uint32_t i = 0; ok && i < p->pre_nseg; i++) { ok = emit(e, p->pre_buf + at, p->pre_len[i], 0); at += p->pre_len[i]; }
	free(p->pre_buf); p->pre_buf = 0;
	const int all = p->pre_nseg >= 2 + PROT(e)->num_metadata_blocks;          /* (every block was a segment) */
	p->pre_nseg = 0;
	if(!ok) return FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR;
	if(all && p->tell_cb && p->tell_cb(e, &PROT(e)->audio_offset, p->client_data) == FLAC__STREAM_ENCODER_TELL_STATUS_ERROR) {
		PROT(e)->state = FLAC__STREAM_ENCODER_CLIENT_ERROR;
		return FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR;
	}
	return FLAC__STREAM_ENCODER_INIT_STATUS_OK;
}

FLAC__StreamEncoderInitStatus FLAC__stream_encoder_init_stream(FLAC__StreamEncoder *e, FLAC__StreamEncoderWriteCallback wcb, FLAC__StreamEncoderSeekCallback scb,
                                                               FLAC__StreamEncoderTellCallback tcb, FLAC__StreamEncoderMetadataCallback mcb, void *client_data)
{
	return init_common(e, 0, wcb, scb, tcb, mcb, client_data, 0);
}
FLAC__StreamEncoderInitStatus FLAC__stream_encoder_init_ogg_stream(FLAC__StreamEncoder *e, FLAC__StreamEncoderReadCallback rcb, FLAC__StreamEncoderWriteCallback wcb,
                                                                   FLAC__StreamEncoderSeekCallback scb, FLAC__StreamEncoderTellCallback tcb,
                                                                   FLAC__StreamEncoderMetadataCallback mcb, void *client_data)
{
	return init_common(e, rcb, wcb, scb, tcb, mcb, client_data, 1);
}

/* FILE* flavour: our own write/seek/tell callbacks plus the progress callback per frame (:5258-5327) */
static FLAC__StreamEncoderWriteStatus file_write(const FLAC__StreamEncoder *e, const FLAC__byte buf[], size_t bytes, uint32_t samples, uint32_t frame, void *cd)
{
	(void)cd; (void)frame;
	struct FLAC__StreamEncoderPrivate *p = PRIV(e);
	if(fwrite(buf, 1, bytes, p->file) != bytes) return FLAC__STREAM_ENCODER_WRITE_STATUS_FATAL_ERROR;
	/* Ogg FLAC: `samples` is always 0 at this point of the reference's callback chain (ogg_encoder_aspect.c), so it calls the
	 * progress callback on every write, header and body pages included; the counters are only advanced after we return */
	if(p->progress_cb && (p->is_ogg || samples > 0))
		p->progress_cb(e, p->bytes_written + bytes, p->samples_written + samples, p->frames_written + (samples ? 1 : 0), p->total_frames_estimate, p->client_data);
	return FLAC__STREAM_ENCODER_WRITE_STATUS_OK;
}
static FLAC__StreamEncoderSeekStatus file_seek(const FLAC__StreamEncoder *e, FLAC__uint64 off, void *cd)
{
	(void)cd;
	return fseeko(PRIV(e)->file, (off_t)off, SEEK_SET) < 0 ? FLAC__STREAM_ENCODER_SEEK_STATUS_ERROR : FLAC__STREAM_ENCODER_SEEK_STATUS_OK;
}
static FLAC__StreamEncoderTellStatus file_tell(const FLAC__StreamEncoder *e, FLAC__uint64 *off, void *cd)
{
	(void)cd;
	const off_t o = ftello(PRIV(e)->file);
	if(o < 0) return FLAC__STREAM_ENCODER_TELL_STATUS_ERROR;
	*off = (FLAC__uint64)o;
	return FLAC__STREAM_ENCODER_TELL_STATUS_OK;
}

static FLAC__StreamEncoderReadStatus file_read(const FLAC__StreamEncoder *e, FLAC__byte buf[], size_t *bytes, void *cd)
{
	(void)cd;                                  /* :5245-5256 */
	*bytes = fread(buf, 1, *bytes, PRIV(e)->file);
	if(*bytes == 0) return feof(PRIV(e)->file) ? FLAC__STREAM_ENCODER_READ_STATUS_END_OF_STREAM : FLAC__STREAM_ENCODER_READ_STATUS_ABORT;
	return FLAC__STREAM_ENCODER_READ_STATUS_CONTINUE;
}
static FLAC__StreamEncoderInitStatus init_FILE_common(FLAC__StreamEncoder *e, FILE *file, FLAC__StreamEncoderProgressCallback pcb, void *client_data, int is_ogg)
{
	struct FLAC__StreamEncoderPrivate *p = PRIV(e);
	if(PROT(e)->state != FLAC__STREAM_ENCODER_UNINITIALIZED) return FLAC__STREAM_ENCODER_INIT_STATUS_ALREADY_INITIALIZED;
	if(!file) { PROT(e)->state = FLAC__STREAM_ENCODER_IO_ERROR; return FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR; }
	p->file = file;                            /* owned from here on: closed by finish() even when init fails (:1497-1502) */
	p->progress_cb = pcb;
	p->bytes_written = 0; p->samples_written = 0; p->frames_written = 0;
	const int to_stdout = file == stdout;
	const FLAC__StreamEncoderInitStatus st = init_common(e, (is_ogg && !to_stdout) ? file_read : 0, file_write, to_stdout ? 0 : file_seek, to_stdout ? 0 : file_tell, 0, client_data, is_ogg);
	if(st != FLAC__STREAM_ENCODER_INIT_STATUS_OK) return st;
	p->progress_cb = pcb;
	const uint32_t bs = PROT(e)->s.blocksize;
	p->total_frames_estimate = (uint32_t)((PROT(e)->s.total_samples_estimate + bs - 1) / bs);
	return st;
}
FLAC__StreamEncoderInitStatus FLAC__stream_encoder_init_FILE(FLAC__StreamEncoder *e, FILE *file, FLAC__StreamEncoderProgressCallback pcb, void *cd) { return init_FILE_common(e, file, pcb, cd, 0); }
FLAC__StreamEncoderInitStatus FLAC__stream_encoder_init_ogg_FILE(FLAC__StreamEncoder *e, FILE *file, FLAC__StreamEncoderProgressCallback pcb, void *cd) { return init_FILE_common(e, file, pcb, cd, 1); }

static FLAC__StreamEncoderInitStatus init_file_common(FLAC__StreamEncoder *e, const char *filename, FLAC__StreamEncoderProgressCallback pcb, void *cd, int is_ogg)
{
	if(PROT(e)->state != FLAC__STREAM_ENCODER_UNINITIALIZED) return FLAC__STREAM_ENCODER_INIT_STATUS_ALREADY_INITIALIZED;
	FILE *f = filename ? fopen(filename, "w+b") : stdout;      /* NULL filename = stdout (:1578) */
	if(!f) { PROT(e)->state = FLAC__STREAM_ENCODER_IO_ERROR; return FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR; }
	return init_FILE_common(e, f, pcb, cd, is_ogg);
}
FLAC__StreamEncoderInitStatus FLAC__stream_encoder_init_file(FLAC__StreamEncoder *e, const char *fn, FLAC__StreamEncoderProgressCallback pcb, void *cd) { return init_file_common(e, fn, pcb, cd, 0); }
FLAC__StreamEncoderInitStatus FLAC__stream_encoder_init_ogg_file(FLAC__StreamEncoder *e, const char *fn, FLAC__StreamEncoderProgressCallback pcb, void *cd) { return init_file_common(e, fn, pcb, cd, 1); }

/* ------------------------------------------------------------------------------------------------
 * process (stream_encoder.c:2513-2626): range check, stage, release full batches
 * ---------------------------------------------------------------------------------------------- */
static int release_if_full(FLAC__StreamEncoder *e)
{
	struct FLAC__StreamEncoderPrivate *p = PRIV(e);
	const uint32_t N = PROT(e)->s.blocksize, C = PROT(e)->s.channels;
	const size_t full = (size_t)p->batch_frames * N;
	if(p->staged <= full) return 1;            /* one sample beyond the batch must exist (the overread) */
	const int k = p->cur, o = (k + 1) % NSLOT;
	/* the next slot of the ring holds the oldest batch still out (NSLOT - 1 batches back): deliver it, then this one goes to
	 * the worker and the caller carries on filling that slot, starting with the overread sample */
	if(!collect_slot(e, o)) return 0;
	memcpy(p->slot[o].raw, p->slot[k].raw + full * C * p->width, (size_t)C * p->width);
	submit_slot(e, k, p->batch_frames, 0);
	p->cur = o;
	p->staged = 1;
	return 1;
}

/* range check (stream_encoder.c:2544-2547) and narrowing copy of `count` values spaced `sstride` apart in src to
 * sample slots spaced `dstride` samples apart in dst */
#if defined(__x86_64__)
#include <immintrin.h>
/* 16-bit streams, the common case, eight or sixteen values per step: this loop is the one pass the caller's thread makes over
 * its samples, so it sets the rate of the whole single-stream API once the MD5 is off (the GPU behind it is ~15x faster) */
__attribute__((target("avx2"))) static int stage16_flat_avx2(int16_t *d, const int32_t *src, size_t count, int32_t smin, int32_t smax)
{
	__m256i vlo = _mm256_setzero_si256(), vhi = _mm256_setzero_si256();
	size_t k = 0;
	int32_t l = 0, h = 0;
	if(count >= 4096) for(; ((uintptr_t)(d + k) & 31) != 0; k++) { const int32_t v = src[k]; d[k] = (int16_t)v; l = v < l ? v : l; h = v > h ? v : h; }
	/* (the saturating pack equals truncation for every value that passes the range check; the others fail the call) */
	if(((uintptr_t)(d + k) & 31) == 0 && count >= 4096) {
		/* the staging buffer is written once and read by the DMA engine: streaming stores spare the read-for-ownership */
		for(; k + 16 <= count; k += 16) {
			const __m256i a = _mm256_loadu_si256((const __m256i *)(src + k)), b = _mm256_loadu_si256((const __m256i *)(src + k + 8));
			vlo = _mm256_min_epi32(vlo, _mm256_min_epi32(a, b)); vhi = _mm256_max_epi32(vhi, _mm256_max_epi32(a, b));
			_mm256_stream_si256((__m256i *)(d + k), _mm256_permute4x64_epi64(_mm256_packs_epi32(a, b), 0xD8));
		}
		_mm_sfence();
	}
	for(; k + 16 <= count; k += 16) {
		const __m256i a = _mm256_loadu_si256((const __m256i *)(src + k)), b = _mm256_loadu_si256((const __m256i *)(src + k + 8));
		vlo = _mm256_min_epi32(vlo, _mm256_min_epi32(a, b)); vhi = _mm256_max_epi32(vhi, _mm256_max_epi32(a, b));
		_mm256_storeu_si256((__m256i *)(d + k), _mm256_permute4x64_epi64(_mm256_packs_epi32(a, b), 0xD8));
	}
	int32_t lo[8], hi[8];
	_mm256_storeu_si256((__m256i *)lo, vlo); _mm256_storeu_si256((__m256i *)hi, vhi);
	for(int i = 0; i < 8; i++) { l = lo[i] < l ? lo[i] : l; h = hi[i] > h ? hi[i] : h; }
	for(; k < count; k++) { const int32_t v = src[k]; d[k] = (int16_t)v; l = v < l ? v : l; h = v > h ? v : h; }
	return !(l < smin || h > smax);
}
/* planar stereo (FLAC__stream_encoder_process): left and right arrays -> interleaved 16-bit pairs */
__attribute__((target("avx2"))) static int stage16_stereo_avx2(int16_t *d, const int32_t *L, const int32_t *R, size_t count, int32_t smin, int32_t smax)
{
	__m256i vlo = _mm256_setzero_si256(), vhi = _mm256_setzero_si256();
	const __m256i m = _mm256_set1_epi32(0xffff);
	size_t k = 0;
	for(; k + 8 <= count; k += 8) {
		const __m256i a = _mm256_loadu_si256((const __m256i *)(L + k)), b = _mm256_loadu_si256((const __m256i *)(R + k));
		vlo = _mm256_min_epi32(vlo, _mm256_min_epi32(a, b)); vhi = _mm256_max_epi32(vhi, _mm256_max_epi32(a, b));
		_mm256_storeu_si256((__m256i *)(d + 2 * k), _mm256_or_si256(_mm256_and_si256(a, m), _mm256_slli_epi32(b, 16)));
	}
	int32_t lo[8], hi[8], l = 0, h = 0;
	_mm256_storeu_si256((__m256i *)lo, vlo); _mm256_storeu_si256((__m256i *)hi, vhi);
	for(int i = 0; i < 8; i++) { l = lo[i] < l ? lo[i] : l; h = hi[i] > h ? hi[i] : h; }
	for(; k < count; k++) { const int32_t a = L[k], b = R[k]; d[2 * k] = (int16_t)a; d[2 * k + 1] = (int16_t)b; l = a < l ? a : l; l = b < l ? b : l; h = a > h ? a : h; h = b > h ? b : h; }
	return !(l < smin || h > smax);
}
static int have_avx2(void) { return __builtin_cpu_supports("avx2") ? 1 : 0; }      /* reads the flags libgcc cached at load time */
#else
static int have_avx2(void) { return 0; }
#endif

static int stage_values(uint8_t *dst, size_t dstride, const int32_t *src, size_t sstride, size_t count, uint32_t width, int32_t smin, int32_t smax)
{
	int32_t lo = 0, hi = 0;
	if(width == 2) {
		int16_t *d = (int16_t *)dst;
#if defined(__x86_64__)
		if(dstride == 1 && sstride == 1 && have_avx2()) return stage16_flat_avx2(d, src, count, smin, smax);
#endif
		if(dstride == 1 && sstride == 1) for(size_t k = 0; k < count; k++) { const int32_t v = src[k]; d[k] = (int16_t)v; lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
		else for(size_t k = 0; k < count; k++) { const int32_t v = src[k * sstride]; d[k * dstride] = (int16_t)v; lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
	}
	else if(width == 1) {
		for(size_t k = 0; k < count; k++) { const int32_t v = src[k * sstride]; dst[k * dstride] = (uint8_t)v; lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
	}
	else if(width == 3) {
		for(size_t k = 0; k < count; k++) {
			const int32_t v = src[k * sstride];
			uint8_t *q = dst + 3 * k * dstride;
			q[0] = (uint8_t)v; q[1] = (uint8_t)(v >> 8); q[2] = (uint8_t)(v >> 16);
			lo = v < lo ? v : lo; hi = v > hi ? v : hi;
		}
	}
	else {
		int32_t *d = (int32_t *)dst;
		for(size_t k = 0; k < count; k++) { const int32_t v = src[k * sstride]; d[k * dstride] = v; lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
	}
	return !(lo < smin || hi > smax);
}

/* the two shapes of a process() call as ranges for stage_pool_run */
struct stage_flat_args { uint8_t *dst; const int32_t *src; uint32_t width; int32_t smin, smax; };
static int stage_flat_range(const void *a_, size_t lo, size_t hi)
{
	const struct stage_flat_args *a = a_;
	return stage_values(a->dst + lo * a->width, 1, a->src + lo, 1, hi - lo, a->width, a->smin, a->smax);
}
struct stage_planar_args { uint8_t *dst; const int32_t * const *src; uint32_t width, channels; int32_t smin, smax; };
static int stage_planar_range(const void *a_, size_t lo, size_t hi)
{
	const struct stage_planar_args *a = a_;
	const uint32_t C = a->channels;
#if defined(__x86_64__)
	if(C == 2 && a->width == 2 && have_avx2()) return stage16_stereo_avx2((int16_t *)(a->dst + lo * 4), a->src[0] + lo, a->src[1] + lo, hi - lo, a->smin, a->smax);
#endif
	int ok = 1;
	for(uint32_t c = 0; c < C && ok; c++) ok = stage_values(a->dst + (lo * C + c) * a->width, C, a->src[c] + lo, 1, hi - lo, a->width, a->smin, a->smax);
	return ok;
}

FLAC__bool FLAC__stream_encoder_process_interleaved(FLAC__StreamEncoder *e, const FLAC__int32 buffer[], uint32_t samples)
{
	struct FLAC__StreamEncoderPrivate *p = PRIV(e);
	if(PROT(e)->state != FLAC__STREAM_ENCODER_OK) return 0;
	const uint32_t N = PROT(e)->s.blocksize, C = PROT(e)->s.channels, bps = PROT(e)->s.bits_per_sample;
	const int32_t smax = INT32_MAX >> (32 - bps), smin = INT32_MIN >> (32 - bps);
	const size_t cap = (size_t)p->batch_frames * N + 1;
	uint32_t j = 0;
	while(j < samples) {
		size_t n = cap - p->staged;
		if(n > samples - j) n = samples - j;
		const double t0 = p->timing ? now_s() : 0;
		const struct stage_flat_args a = { p->slot[p->cur].raw + p->staged * C * p->width, buffer + (size_t)j * C, p->width, smin, smax };
		const int ok = stage_pool_run(p->pool, stage_flat_range, &a, n * C, 64);
		if(p->timing) p->t_stage += now_s() - t0;
		if(!ok) {
			PROT(e)->state = FLAC__STREAM_ENCODER_CLIENT_ERROR;
			return 0;
		}
		p->staged += n; j += (uint32_t)n;
		publish_staged(e);
		if(!release_if_full(e)) return 0;
	}
	return 1;
}

FLAC__bool FLAC__stream_encoder_process(FLAC__StreamEncoder *e, const FLAC__int32 * const buffer[], uint32_t samples)
{
	struct FLAC__StreamEncoderPrivate *p = PRIV(e);
	if(PROT(e)->state != FLAC__STREAM_ENCODER_OK) return 0;
	const uint32_t N = PROT(e)->s.blocksize, C = PROT(e)->s.channels, bps = PROT(e)->s.bits_per_sample;
	const int32_t smax = INT32_MAX >> (32 - bps), smin = INT32_MIN >> (32 - bps);
	const size_t cap = (size_t)p->batch_frames * N + 1;
	for(uint32_t c = 0; c < C; c++) if(!buffer[c]) return 0;
	const int32_t *chan[FLACGPU_MAX_CHANNELS];
	uint32_t j = 0;
	while(j < samples) {
		size_t n = cap - p->staged;
		if(n > samples - j) n = samples - j;
		const double t0 = p->timing ? now_s() : 0;
		for(uint32_t c = 0; c < C; c++) chan[c] = buffer[c] + j;
		const struct stage_planar_args a = { p->slot[p->cur].raw + p->staged * C * p->width, chan, p->width, C, smin, smax };
		const int ok = stage_pool_run(p->pool, stage_planar_range, &a, n, 64);
		if(p->timing) p->t_stage += now_s() - t0;
		if(!ok) {
			PROT(e)->state = FLAC__STREAM_ENCODER_CLIENT_ERROR;
			return 0;
		}
		p->staged += n; j += (uint32_t)n;
		publish_staged(e);
		if(!release_if_full(e)) return 0;
	}
	return 1;
}

/* ------------------------------------------------------------------------------------------------
 * finish (stream_encoder.c:1625-1778) and the STREAMINFO / SEEKTABLE fix-up (update_metadata_, :3139-3298)
 * ---------------------------------------------------------------------------------------------- */
static int rewrite(FLAC__StreamEncoder *e, FLAC__uint64 offset, const uint8_t *b, size_t n)
{
	struct FLAC__StreamEncoderPrivate *p = PRIV(e);
	const FLAC__StreamEncoderSeekStatus ss = p->seek_cb(e, offset, p->client_data);
	if(ss != FLAC__STREAM_ENCODER_SEEK_STATUS_OK) { if(ss == FLAC__STREAM_ENCODER_SEEK_STATUS_ERROR) PROT(e)->state = FLAC__STREAM_ENCODER_CLIENT_ERROR; return 0; }
	if(p->write_cb(e, b, n, 0, 0, p->client_data) != FLAC__STREAM_ENCODER_WRITE_STATUS_OK) { PROT(e)->state = FLAC__STREAM_ENCODER_CLIENT_ERROR; return 0; }
	return 1;
}

static void update_metadata(FLAC__StreamEncoder *e)
{
	struct FLAC__StreamEncoderPrivate *p = PRIV(e);
	const FLAC__StreamMetadata_StreamInfo *si = &p->streaminfo.data.stream_info;
	const FLAC__uint64 base = PROT(e)->streaminfo_offset;
	uint8_t b[18];
	/* MD5 at byte 4+18, then the 5 bytes holding (bps-1)<<4 | total_samples[35:32] .. total_samples[7:0] at 4+13,
	 * then min/max frame size at 4+4 -- the order the reference seeks in */
	if(!rewrite(e, base + 4 + 18, si->md5sum, 16)) return;
	{
		FLAC__uint64 t = si->total_samples;
		if(t > ((FLAC__uint64)1 << 36)) t = 0;
		b[0] = (uint8_t)(((si->bits_per_sample - 1) << 4) | ((t >> 32) & 0x0F));
		b[1] = (uint8_t)(t >> 24); b[2] = (uint8_t)(t >> 16); b[3] = (uint8_t)(t >> 8); b[4] = (uint8_t)t;
		if(!rewrite(e, base + 4 + 13, b, 5)) return;
	}
	b[0] = (uint8_t)(si->min_framesize >> 16); b[1] = (uint8_t)(si->min_framesize >> 8); b[2] = (uint8_t)si->min_framesize;
	b[3] = (uint8_t)(si->max_framesize >> 16); b[4] = (uint8_t)(si->max_framesize >> 8); b[5] = (uint8_t)si->max_framesize;
	if(!rewrite(e, base + 4 + 4, b, 6)) return;
	if(p->seek_table && p->seek_table->num_points > 0 && PROT(e)->seektable_offset > 0) {
		for(uint32_t i = 0; i < p->seek_table->num_points; i++)      /* template points beyond the end become placeholders */
			if(p->seek_table->points[i].sample_number > si->total_samples) p->seek_table->points[i].sample_number = FLAC__STREAM_METADATA_SEEKPOINT_PLACEHOLDER;
		seektable_sort(p->seek_table);
		{
			const FLAC__StreamEncoderSeekStatus ss = p->seek_cb(e, PROT(e)->seektable_offset + 4, p->client_data);      /* :3262-3266 */
			if(ss != FLAC__STREAM_ENCODER_SEEK_STATUS_OK) { if(ss == FLAC__STREAM_ENCODER_SEEK_STATUS_ERROR) PROT(e)->state = FLAC__STREAM_ENCODER_CLIENT_ERROR; return; }
		}
		for(uint32_t i = 0; i < p->seek_table->num_points; i++) {
			const FLAC__StreamMetadata_SeekPoint *sp = &p->seek_table->points[i];
			for(int k = 0; k < 8; k++) { b[k] = (uint8_t)(sp->sample_number >> (56 - 8 * k)); b[8 + k] = (uint8_t)(sp->stream_offset >> (56 - 8 * k)); }
			b[16] = (uint8_t)(sp->frame_samples >> 8); b[17] = (uint8_t)sp->frame_samples;
			if(p->write_cb(e, b, 18, 0, 0, p->client_data) != FLAC__STREAM_ENCODER_WRITE_STATUS_OK) { PROT(e)->state = FLAC__STREAM_ENCODER_CLIENT_ERROR; return; }
		}
	}
}

/* update_ogg_metadata_ (:3303-3440) with the page helpers of ogg_helper.c: read the first page back, patch MD5, total
 * samples and min/max frame size inside its body (the STREAMINFO sits 13 bytes into the first packet), new page CRC, write it back */
static void update_ogg_metadata(FLAC__StreamEncoder *e)
{
	struct FLAC__StreamEncoderPrivate *p = PRIV(e);
	const FLAC__StreamMetadata_StreamInfo *si = &p->streaminfo.data.stream_info;
	const size_t prefix = 1 + 4 + 1 + 1 + 2 + 4;                       /* 0x7F "FLAC" major minor count "fLaC" */
	uint8_t page[27 + 255 + 255 * 255];
	if(p->seek_cb(e, 0, p->client_data) == FLAC__STREAM_ENCODER_SEEK_STATUS_UNSUPPORTED) return;
	if(!p->read_cb) return;
	{
		const FLAC__StreamEncoderSeekStatus ss = p->seek_cb(e, PROT(e)->streaminfo_offset, p->client_data);
		if(ss != FLAC__STREAM_ENCODER_SEEK_STATUS_OK) { if(ss == FLAC__STREAM_ENCODER_SEEK_STATUS_ERROR) PROT(e)->state = FLAC__STREAM_ENCODER_CLIENT_ERROR; return; }
	}
	size_t have = 0, want = 27;
	int stage = 0;                                                   /* 0 fixed header, 1 segment table, 2 body */
	size_t header_len = 0, body_len = 0;
	for(;;) {
		while(have < want) {
			size_t n = want - have;
			const FLAC__StreamEncoderReadStatus rs = p->read_cb(e, page + have, &n, p->client_data);
			if(rs == FLAC__STREAM_ENCODER_READ_STATUS_ABORT) { PROT(e)->state = FLAC__STREAM_ENCODER_CLIENT_ERROR; return; }
			if(rs == FLAC__STREAM_ENCODER_READ_STATUS_END_OF_STREAM || n == 0) { PROT(e)->state = FLAC__STREAM_ENCODER_OGG_ERROR; return; }
			have += n;
		}
		if(stage == 0) {
			if(memcmp(page, "OggS", 4) || (page[5] & 0x01) || memcmp(page + 6, "\0\0\0\0\0\0\0\0", 8) || page[26] == 0) { PROT(e)->state = FLAC__STREAM_ENCODER_OGG_ERROR; return; }
			header_len = 27 + page[26]; want = header_len; stage = 1;
		}
		else if(stage == 1) {
			for(size_t i = 0; i < page[26]; i++) body_len += page[27 + i];
			want = header_len + body_len; stage = 2;
		}
		else break;
	}
	uint8_t *body = page + header_len;
	if(prefix + 4 + 18 + 16 > body_len) { PROT(e)->state = FLAC__STREAM_ENCODER_OGG_ERROR; return; }
	memcpy(body + prefix + 4 + 18, si->md5sum, 16);
	{
		const FLAC__uint64 t = si->total_samples;
		uint8_t *b = body + prefix + 4 + 13;
		b[0] = (uint8_t)((b[0] & 0xF0) | ((t >> 32) & 0x0F));
		b[1] = (uint8_t)(t >> 24); b[2] = (uint8_t)(t >> 16); b[3] = (uint8_t)(t >> 8); b[4] = (uint8_t)t;
	}
	{
		uint8_t *b = body + prefix + 4 + 4;
		b[0] = (uint8_t)(si->min_framesize >> 16); b[1] = (uint8_t)(si->min_framesize >> 8); b[2] = (uint8_t)si->min_framesize;
		b[3] = (uint8_t)(si->max_framesize >> 16); b[4] = (uint8_t)(si->max_framesize >> 8); b[5] = (uint8_t)si->max_framesize;
	}
	fgh_ogg_page_checksum_set(page, header_len, body, body_len);
	{
		const FLAC__StreamEncoderSeekStatus ss = p->seek_cb(e, PROT(e)->streaminfo_offset, p->client_data);
		if(ss != FLAC__STREAM_ENCODER_SEEK_STATUS_OK) { if(ss == FLAC__STREAM_ENCODER_SEEK_STATUS_ERROR) PROT(e)->state = FLAC__STREAM_ENCODER_CLIENT_ERROR; return; }
	}
	if(p->write_cb(e, page, header_len, 0, 0, p->client_data) != FLAC__STREAM_ENCODER_WRITE_STATUS_OK ||
	   p->write_cb(e, body, body_len, 0, 0, p->client_data) != FLAC__STREAM_ENCODER_WRITE_STATUS_OK) { PROT(e)->state = FLAC__STREAM_ENCODER_CLIENT_ERROR; return; }
}

FLAC__bool FLAC__stream_encoder_finish(FLAC__StreamEncoder *e)
{
	if(!e) return 0;
	struct FLAC__StreamEncoderPrivate *p = PRIV(e);
	int error = 0;
	if(PROT(e)->state == FLAC__STREAM_ENCODER_UNINITIALIZED) {
		if(PROT(e)->metadata) { free(PROT(e)->metadata); PROT(e)->metadata = 0; PROT(e)->num_metadata_blocks = 0; }
		if(p->file) { if(p->file != stdout) fclose(p->file); p->file = 0; }
		return 1;
	}
	if(PROT(e)->state == FLAC__STREAM_ENCODER_OK && !p->is_being_deleted && p->engine_on) {
		/* the batch in flight, then everything still staged: full blocks and the final one (short, or exactly full) */
		for(int i = 1; i < NSLOT && !error; i++) if(!collect_slot(e, (p->cur + i) % NSLOT)) error = 1;      /* oldest first */
		if(!error && p->staged) {
			const uint32_t N = PROT(e)->s.blocksize;
			const uint32_t nframes = (uint32_t)((p->staged + N - 1) / N);
			uint32_t tail = (uint32_t)(p->staged - (size_t)(nframes - 1) * N);
			if(tail == N) tail = 0;
			submit_slot(e, p->cur, nframes, tail);
			p->final_batch = 1;
			if(!collect_slot(e, p->cur)) error = 1;
			p->final_batch = 0;
		}
		p->staged = 0;
	}
	else if(p->worker_started) {
		/* an encoder that is torn down or already failed: let the worker drain, deliver nothing */
		pthread_mutex_lock(&p->mu);
		for(;;) {
			int busy = 0;
			for(int i = 0; i < NSLOT; i++) if(p->slot[i].state == 1 || p->slot[i].state == 3) busy = 1;
			if(!busy) break;
			pthread_cond_wait(&p->cv, &p->mu);
		}
		pthread_mutex_unlock(&p->mu);
	}
	if(p->preamble_pending && !p->is_being_deleted && PROT(e)->state == FLAC__STREAM_ENCODER_OK) {
		/* no frame went out (an empty stream): its head still has to, provided the engine did come up */
		if(p->bring_started) { pthread_join(p->bring_th, 0); p->bring_started = 0; }
		p->preamble_pending = 0;
		if(p->bring_result != FLACGPU_OK) {
			PROT(e)->state = p->bring_result == FLACGPU_ERR_ALLOC ? FLAC__STREAM_ENCODER_MEMORY_ALLOCATION_ERROR : FLAC__STREAM_ENCODER_FRAMING_ERROR;
			error = 1;
		}
		else if(write_preamble(e) != FLAC__STREAM_ENCODER_INIT_STATUS_OK) error = 1;
	}
	p->preamble_pending = 0;
	free(p->pre_buf); p->pre_buf = 0; p->pre_nseg = 0;
	if(PROT(e)->s.do_md5 && p->engine_on) flacgpu_host_md5_final(&p->md5, p->streaminfo.data.stream_info.md5sum);
	if(!p->is_being_deleted && PROT(e)->state == FLAC__STREAM_ENCODER_OK) {
		p->current_frame_number = 0;
		if(p->seek_cb) {
			if(p->is_ogg) update_ogg_metadata(e); else update_metadata(e);
			if(PROT(e)->state != FLAC__STREAM_ENCODER_OK) error = 1;
		}
		if(p->metadata_cb) p->metadata_cb(e, &p->streaminfo, p->client_data);
	}
	if(p->file) { if(p->file != stdout) fclose(p->file); p->file = 0; }
	if(p->is_ogg) fgh_ogg_aspect_finish(&p->ogg);
	const int timing = p->timing;
	const double t_rel = timing ? now_s() : 0;
	release_engine(e);
	if(timing) {
		const double t1 = now_s();
		fprintf(stderr, "libFLACgpu timing (s): total %.4f | init: engine %.4f pinned %.4f | caller: stage %.4f wait %.4f deliver %.4f | worker: md5 %.4f wait-for-engine %.4f encode %.4f | release %.4f\n",
		        t1 - p->t_start, p->t_init_engine, p->t_init_pinned, p->t_stage, p->t_wait, p->t_emit, p->t_md5, p->t_bring_wait, p->t_encode, t1 - t_rel);
		p->timing = 0;
	}
	set_defaults(e);
	if(!error) PROT(e)->state = FLAC__STREAM_ENCODER_UNINITIALIZED;
	return !error;
}
