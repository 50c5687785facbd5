// flac_amd/csrc/flacgpu_api.cpp -- C-ABI entry points of libflacgpu.so (include/flacgpu.h).
//
// Host-side plumbing only: device buffers, streams, events, launches.  There is deliberately NO
// CPU implementation behind these entry points: without a usable HIP device flacgpu_create()
// fails with FLACGPU_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <unistd.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <new>
#include "flacgpu.h"
#include "flacgpu_dev.h"

using namespace flacgpu;

constexpr int TIMING_RING = 64;
struct flacgpu_ctx {
	flacgpu_config cfg;
	DevParams P;
	int device;
	hipStream_t stream;          // engine-owned stream (host-buffer entry point)
	// HIP events of the last TIMING_RING batches (so that a caller can read per-kernel times of a run of batches
	// afterwards, without a host sync in between); ev/pev point at the set of the current batch
	hipEvent_t ev_ring[TIMING_RING][5], pev_ring[TIMING_RING][3];
	bool timed_ring[TIMING_RING]; uint32_t timing_every;      // which of them carry events (flacgpu_set_phase_timing)
	hipEvent_t seq_ring[TIMING_RING][7];   // the seven phase boundaries of each batch: which of its events closes which phase (an event record costs
	                                       // the stream ~4 us, so phases that launch nothing share the event of the phase in front)
	hipEvent_t *ev;              // start, after analyze, after pack, after compact, spare
	hipEvent_t *pev;             // inside the analysis: after prep, after autoc, after model
	uint64_t batch_seq;          // batches launched so far
	uint32_t nsub;               // sub-batches of a batch, each on its own stream (1: everything on one stream)
	hipStream_t sub_stream[FLACGPU_MAX_SUBBATCHES];
	hipEvent_t sub_done[FLACGPU_MAX_SUBBATCHES], ev_fork;
	AnalyzeBuffers ab;           // hand-off records between the analysis kernels
	float *d_windows;            // [num_apod][blocksize]
	float *d_tail_windows;       // [num_apod][blocksize] scratch for the short last block
	SubDecision *d_decisions;    // [max_batch][ncand]
	uint8_t *d_plan;             // [max_batch] pack_plan_stride(P): what pack2_kernel packs from (flacgpu_dev.h: PackSub)
	uint8_t *d_slots;            // [max_batch][slot_bytes]
	bool ff_ok;                  // ff_kernel (one kernel per batch: -0 .. -2 on 16-bit stereo) can take this stream
	uint32_t *d_frame_bytes;     // [max_batch]
	uint32_t *last_fb;           // where the last batch's frame lengths are: d_frame_bytes, or the caller's array
	uint64_t *d_offsets;         // [max_batch+1]
	uint64_t *d_total;
	// the fused output (flacgpu_kernels.hip: PackOut): tagged frame lengths, segment totals and starts, arrival counters, the ticket
	uint64_t *d_fo_fstate, *d_fo_sstate, *d_fo_sprefix, *d_fo_scount;
	uint32_t *d_fo_fall, *d_fo_nfall;
	size_t fo_frames, fo_segs;
	uint32_t fo_epoch, fo_spin_limit;
	uint32_t fo_last_fused;      // epoch of the last batch whose frames really went through the fused output (its fo_place_kernel zeroed the
	                             // fall-back counter of the epoch behind it; an epoch spent on a batch that did not fuse breaks that chain)
	bool fo_dirty;               // a batch that used the fused-output words ended in an error: the next one starts from clean words
	int ff_lag;                  // FLACGPU_FF_LAG (launch_ff): -1 not set -- ff_kernel batches take the two-kernel compaction
	FrameInfo *d_info;           // [max_batch]
	int32_t *d_pcm;              // staging for the host entry point
	uint8_t *d_raw;              // raw sample bytes of flacgpu_encode_batch_raw
	size_t d_raw_bytes;
	uint32_t *d_stage_err;
	uint8_t *d_out;
	size_t d_pcm_bytes, d_out_bytes;
	uint32_t last_nframes;
	bool timing_valid;
	// the self check (flacgpu_verify.hip): state word, verdict, offsets of a caller's frames, scratch of the detail pass
	VerifyState *d_vstate;
	flacgpu_verify_result *d_vresult;
	uint64_t *d_voffsets, *d_vtotal;
	int64_t *d_vscratch;
	void *d_vdecoded;            // decoded coded-channel samples of a batch, lane-interleaved (flacgpu_verify.hip)
	uint32_t *d_vfinfo;          // [max_batch] per-frame verdict of the decode pass
	uint32_t *d_vhints;          // [max_batch][channels][HINT_RUNS] run starts written by the pack kernel (flacgpu_decode_hinted.h)
	uint32_t *d_vfstat;          // [max_batch] verdict of the hinted pass: 0 verified, 1 to be decoded sequentially
	// the batch the hints in d_vhints describe: they are used when exactly these frames come back to be verified
	const uint8_t *hint_out; uint32_t hint_nframes, hint_count; uint64_t hint_first;
	int force_hints;             // FLACGPU_VERIFY_FORCE_HINTS=1 (tests): use them whatever frames are handed in; =0: never
	uint32_t verify_on;
	flacgpu_verify_result last_verify;
	JobTable h_jobtab[2];        // [0] nominal blocksize, [1] the short last block of the current batch
	JobTable *d_jobtab;          // device copies of both
	// the asynchronous entry (flacgpu_submit_batch_raw / flacgpu_collect): a ring of batches in flight
	struct AsyncSlot {
		uint8_t *d_raw, *d_out;                 // this batch's input bytes and finished frames on the device
		size_t d_raw_bytes, d_out_bytes;
		uint32_t *d_fb; uint64_t *d_total; uint32_t *d_err; flacgpu_verify_result *d_vres;
		struct Host { uint64_t total; uint32_t err; uint32_t pad; flacgpu_verify_result vres; } *h;     // page-locked: where the small results land
		uint32_t *h_fb;                         // page-locked [max_batch]
		hipEvent_t ev_in, ev_done, ev_small, ev_pay;
		uint8_t *out; size_t out_cap; uint32_t *frame_bytes; uint32_t nframes; int check_shift;
		bool pay_by_kernel;                     // the frames were written to `out` by payload_copy_kernel (page-locked `out`)
	} as[FLACGPU_ASYNC_SLOTS];
	hipStream_t s_in, s_small, s_pay;           // input copies | lengths and totals back | payloads back
	uint64_t sub_seq, col_seq;
	bool async_ready;
	Tune tune;                   // the development switches of this context (environment at flacgpu_create) and what its last batch launched
	uint32_t last_launched;
};

namespace flacgpu {
static Tune read_tune(int device)
{
	Tune t;
	memset(&t, 0, sizeof t);
	auto num = [](const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; };
	auto set = [](const char *name) { return getenv(name) ? 1 : 0; };
	t.autoc3_mode = num("FLACGPU_AUTOC3", 2); t.autoc3_sets = num("FLACGPU_AUTOC3_SETS", 1); t.autoc3_planes = num("FLACGPU_AUTOC3_PLANES", 1);
	t.autoc3_ind_sets = num("FLACGPU_AUTOC3_IND_SETS", 2);
	t.event_fence = num("FLACGPU_EVENT_FENCE", 0);            // 1: the timing events with their system-scope fence (round 4's)
	t.copy_results = num("FLACGPU_COPY_RESULTS", 0);          // 1: frame lengths / total copied to the caller's arrays behind the last kernel (round 4's)
	t.autoc2_ungrouped = set("FLACGPU_AUTOC2_UNGROUPED");
	{ const char *e = getenv("FLACGPU_AUTOC2"); t.autoc2_force = e ? atoi(e) + 1 : 0; }
	t.no_ff = set("FLACGPU_NO_FF"); t.no_run18 = set("FLACGPU_NO_RUN18"); t.no_run18w = set("FLACGPU_NO_RUN18W"); t.no_prep3 = set("FLACGPU_NO_PREP3"); t.no_prep3n = set("FLACGPU_NO_PREP3N"); t.no_prep_decide = set("FLACGPU_NO_PREP_DECIDE");
	t.no_evalg = set("FLACGPU_NO_EVALG"); t.no_fast1 = set("FLACGPU_NO_FAST1"); t.no_prep4 = set("FLACGPU_NO_PREP4"); t.no_flat = set("FLACGPU_NO_FLAT"); t.no_wide_decide = set("FLACGPU_NO_WIDE_DECIDE"); t.no_evalg32 = set("FLACGPU_NO_EVALG32"); t.no_wide_ff = set("FLACGPU_NO_WIDE_FF");
	t.eval_wpc = num("FLACGPU_EVAL_WPC", 1) == 2 ? 2 : 1; t.evalw_wpc = num("FLACGPU_EVALW_WPC", 2) == 1 ? 1 : 2;
	t.eval_waves = num("FLACGPU_EVAL_WAVES", 0); t.eval_cpw = num("FLACGPU_EVAL_CPW", 0); t.eval_prefetch = num("FLACGPU_EVAL_PREFETCH", -1);
	t.sync_debug = set("FLACGPU_SYNC_DEBUG"); t.no_fused = set("FLACGPU_NO_FUSED_COMPACT"); t.no_copy_kernel = set("FLACGPU_NO_COPY_KERNEL");
	t.cands_global = set("FLACGPU_EVAL_CANDS_GLOBAL");
	t.device = device;
	return t;
}
static thread_local Tune *g_tune = nullptr;
Tune &tune()
{
	if(g_tune) return *g_tune;
	static thread_local Tune dflt = read_tune(0);
	return dflt;
}
std::mutex &attr_mutex() { static std::mutex m; return m; }
// an entry point of a context runs under its switches
struct TuneScope { Tune *prev; explicit TuneScope(Tune *t) : prev(g_tune) { g_tune = t; } ~TuneScope() { g_tune = prev; } };

// The apply_apodization_ state machine (stream_encoder.c:4293-4392) unrolled into a static schedule.
void build_job_table(const DevParams &P, uint32_t n, JobTable *jt)
{
	memset(jt, 0, sizeof *jt);
	uint32_t nj = 0, na = 0, woff = 0, ns = 0;
	auto add_set = [&](uint32_t first, uint32_t count) { if(count) { if(ns < 8) { jt->set_first[ns] = (uint16_t)first; jt->set_count[ns] = (uint16_t)count; } ns++; } };
	for(uint32_t a = 0; a < P.num_apod; a++) {
		const uint32_t root = nj;
		WindowJob &jr = jt->jobs[nj];
		jr.off = woff; jr.nd = n; jr.apod = a; jr.full = 1;
		woff += (n + 3u) & ~1u;
		jt->an_job[na] = (uint16_t)nj; jt->an_punch[na] = 0; jt->an_root[na] = (uint16_t)root; na++; nj++;
		add_set(root, 1);
		if(P.apod_kind[a] == FLACGPU_APOD_SUBDIVIDE_TUKEY) {
			for(uint32_t b = 2; b <= P.apod_parts[a]; b++) {
				if(n / b <= 32) continue;                               /* :4349-4357 */
				const uint32_t depth_first = nj;
				for(uint32_t pi = 0; pi < b; pi++) {
					if(nj >= (uint32_t)MAX_JOBS || na + 2 > (uint32_t)MAX_ANALYSES) continue;   /* create() rejects such configs */
					WindowJob &jp = jt->jobs[nj];
					jp.off = woff; jp.nd = n / b; jp.apod = a; jp.full = 0;
					jp.part = n / b / 2; jp.dshift = (pi * n) / b;      /* :4361 */
					jp.i0 = jp.part < n - jp.part - jp.dshift ? jp.part : n - jp.part - jp.dshift;
					woff += (jp.nd + 3u) & ~1u;
					jt->an_job[na] = (uint16_t)nj; jt->an_punch[na] = 0; jt->an_root[na] = (uint16_t)root; na++;
					if(b >= 3) { jt->an_job[na] = (uint16_t)nj; jt->an_punch[na] = 1; jt->an_root[na] = (uint16_t)root; na++; }   /* :4295-4308 */
					nj++;
				}
				add_set(depth_first, nj - depth_first);
			}
		}
	}
	jt->njobs = nj; jt->nanalyses = na; jt->wnd_floats = woff; jt->nsets = ns;
	for(uint32_t m = 0; m < 7; m++)
		for(uint32_t o = 0; o <= (uint32_t)MAX_ORDER; o++) { const uint32_t full = (n / 64u) << m; jt->eg_div[m][o] = full > o ? (0x40000u / (full - o)) << 13 : 0u; }
}
}

extern "C" int flacgpu_batch_phase_ms(flacgpu_ctx *c, uint32_t batches_ago, float ms[6])
{
	if(!c || !ms || !c->timing_valid || batches_ago >= (uint32_t)TIMING_RING || batches_ago >= c->batch_seq) return FLACGPU_ERR_BAD_ARG;
	if(hipSetDevice(c->device) != hipSuccess) return FLACGPU_ERR_NO_DEVICE;
	const size_t slot = (size_t)((c->batch_seq - 1 - batches_ago) % TIMING_RING);
	hipEvent_t *seq = c->seq_ring[slot];
	if(!c->timed_ring[slot]) { for(int i = 0; i < 6; i++) ms[i] = 0.0f; return 1; }      // that batch carried no events
	if(hipEventSynchronize(seq[6]) != hipSuccess) return FLACGPU_ERR_LAUNCH;
	for(int i = 0; i < 6; i++) {
		ms[i] = 0.0f;
		if(seq[i] != seq[i + 1] && hipEventElapsedTime(&ms[i], seq[i], seq[i + 1]) != hipSuccess) return FLACGPU_ERR_LAUNCH;
	}
	return FLACGPU_OK;
}
extern "C" int flacgpu_last_batch_phase_ms(flacgpu_ctx *c, float ms[6]) { return flacgpu_batch_phase_ms(c, 0, ms); }
extern "C" int flacgpu_set_phase_timing(flacgpu_ctx *c, uint32_t every)
{
	if(!c) return FLACGPU_ERR_BAD_ARG;
	c->timing_every = every;
	return FLACGPU_OK;
}
extern "C" int flacgpu_set_subbatches(flacgpu_ctx *c, uint32_t n)
{
	if(!c || n < 1 || n > (uint32_t)FLACGPU_MAX_SUBBATCHES) return FLACGPU_ERR_BAD_ARG;
	c->nsub = n;
	return FLACGPU_OK;
}

extern "C" void *flacgpu_alloc_pinned(size_t bytes)
{
	void *p = nullptr;
	if(hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
	return p;
}
extern "C" void flacgpu_free_pinned(void *p) { if(p) (void)hipHostFree(p); }
extern "C" int flacgpu_host_register(void *p, size_t bytes)
{
	if(!p || !bytes) return FLACGPU_ERR_BAD_ARG;
	return hipHostRegister(p, bytes, hipHostRegisterDefault) == hipSuccess ? FLACGPU_OK : FLACGPU_ERR_ALLOC;
}
extern "C" void flacgpu_host_unregister(void *p) { if(p) (void)hipHostUnregister(p); }
extern "C" int flacgpu_device_probe(void)
{
	const int fd = open("/dev/kfd", O_RDWR | O_CLOEXEC);
	if(fd < 0) return 0;
	close(fd);
	return 1;
}

extern "C" const char *flacgpu_strerror(int code)
{
	switch(code) {
		case FLACGPU_OK: return "ok";
		case FLACGPU_ERR_UNSUPPORTED: return "configuration outside the GPU engine's supported range";
		case FLACGPU_ERR_NO_DEVICE: return "no usable HIP device (there is no CPU fallback)";
		case FLACGPU_ERR_ALLOC: return "device memory allocation failed";
		case FLACGPU_ERR_OUTPUT_TOO_SMALL: return "output buffer too small";
		case FLACGPU_ERR_LAUNCH: return "kernel launch or execution failed";
		case FLACGPU_ERR_BAD_ARG: return "bad argument";
		case FLACGPU_ERR_INPUT: return "raw sample data has non-zero bits below its declared shift";
		case FLACGPU_ERR_BUSY: return "too many batches in flight: collect one first";
		default: return "unknown error";
	}
}

extern "C" int flacgpu_device_count(void)
{
	int n = 0;
	if(hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

static void free_ctx(flacgpu_ctx *c)
{
	if(!c) return;
	(void)hipSetDevice(c->device);
	if(c->d_windows) (void)hipFree(c->d_windows);
	if(c->d_tail_windows) (void)hipFree(c->d_tail_windows);
	if(c->d_decisions) (void)hipFree(c->d_decisions);
	if(c->d_plan) (void)hipFree(c->d_plan);
	if(c->d_slots) (void)hipFree(c->d_slots);
	if(c->d_frame_bytes) (void)hipFree(c->d_frame_bytes);
	if(c->d_offsets) (void)hipFree(c->d_offsets);
	if(c->d_total) (void)hipFree(c->d_total);
	if(c->d_fo_fstate) (void)hipFree(c->d_fo_fstate);
	if(c->d_fo_sstate) (void)hipFree(c->d_fo_sstate);
	if(c->d_fo_sprefix) (void)hipFree(c->d_fo_sprefix);
	if(c->d_fo_scount) (void)hipFree(c->d_fo_scount);
	if(c->d_fo_fall) (void)hipFree(c->d_fo_fall);
	if(c->d_fo_nfall) (void)hipFree(c->d_fo_nfall);
	if(c->d_info) (void)hipFree(c->d_info);
	if(c->d_pcm) (void)hipFree(c->d_pcm);
	if(c->d_raw) (void)hipFree(c->d_raw);
	if(c->d_stage_err) (void)hipFree(c->d_stage_err);
	if(c->d_out) (void)hipFree(c->d_out);
	if(c->ab.prep) (void)hipFree(c->ab.prep);
	if(c->ab.autoc) (void)hipFree(c->ab.autoc);
	if(c->ab.cands) (void)hipFree(c->ab.cands);
	if(c->ab.valid) (void)hipFree(c->ab.valid);
	if(c->ab.chan) (void)hipFree(c->ab.chan);
	for(int i = 0; i < FLACGPU_ASYNC_SLOTS; i++) {
		flacgpu_ctx::AsyncSlot &a = c->as[i];
		if(a.d_raw) (void)hipFree(a.d_raw);
		if(a.d_out) (void)hipFree(a.d_out);
		if(a.d_fb) (void)hipFree(a.d_fb);
		if(a.d_total) (void)hipFree(a.d_total);
		if(a.d_err) (void)hipFree(a.d_err);
		if(a.d_vres) (void)hipFree(a.d_vres);
		if(a.h) (void)hipHostFree(a.h);
		if(a.h_fb) (void)hipHostFree(a.h_fb);
		if(a.ev_in) (void)hipEventDestroy(a.ev_in);
		if(a.ev_done) (void)hipEventDestroy(a.ev_done);
		if(a.ev_small) (void)hipEventDestroy(a.ev_small);
		if(a.ev_pay) (void)hipEventDestroy(a.ev_pay);
	}
	if(c->s_in) (void)hipStreamDestroy(c->s_in);
	if(c->s_small) (void)hipStreamDestroy(c->s_small);
	if(c->s_pay) (void)hipStreamDestroy(c->s_pay);
	if(c->ab.left) (void)hipFree(c->ab.left);
	if(c->ab.left2) (void)hipFree(c->ab.left2);
	if(c->ab.nleft) (void)hipFree(c->ab.nleft);
	if(c->ab.dbg) (void)hipFree(c->ab.dbg);
	for(int r = 0; r < TIMING_RING; r++) for(int i = 0; i < 3; i++) if(c->pev_ring[r][i]) (void)hipEventDestroy(c->pev_ring[r][i]);
	for(int i = 0; i < FLACGPU_MAX_SUBBATCHES; i++) { if(c->sub_stream[i]) (void)hipStreamDestroy(c->sub_stream[i]); if(c->sub_done[i]) (void)hipEventDestroy(c->sub_done[i]); }
	if(c->ev_fork) (void)hipEventDestroy(c->ev_fork);
	if(c->d_jobtab) (void)hipFree(c->d_jobtab);
	if(c->d_vstate) (void)hipFree(c->d_vstate);
	if(c->d_vresult) (void)hipFree(c->d_vresult);
	if(c->d_voffsets) (void)hipFree(c->d_voffsets);
	if(c->d_vtotal) (void)hipFree(c->d_vtotal);
	if(c->d_vscratch) (void)hipFree(c->d_vscratch);
	if(c->d_vdecoded) (void)hipFree(c->d_vdecoded);
	if(c->d_vfinfo) (void)hipFree(c->d_vfinfo);
	if(c->d_vhints) (void)hipFree(c->d_vhints);
	if(c->d_vfstat) (void)hipFree(c->d_vfstat);
	for(int r = 0; r < TIMING_RING; r++) for(int i = 0; i < 5; i++) if(c->ev_ring[r][i]) (void)hipEventDestroy(c->ev_ring[r][i]);
	if(c->stream) (void)hipStreamDestroy(c->stream);
	delete c;
}

// worst-case frame: header + per channel (verbatim size + Rice estimate slack of N/2 bits + side info), a multiple of 16
static uint64_t worst_case_frame_bytes(uint32_t C, uint32_t N, uint32_t bps)
{
	const uint64_t per_ch_bits = (uint64_t)N * (bps + 1) + N / 2 + 8 + 32 + 16 * 33 + 9 + 6 + 5 * (1u << MAX_PO);
	const uint64_t bytes = 16 + C * ((per_ch_bits + 7) / 8) + 2 + 16;
	return (bytes + 15) & ~(uint64_t)15;
}

// ---- supported range (everything else is a documented, loud failure; no CPU fallback) ----
static int check_config(const flacgpu_config *cfg)
{
	if(!cfg || cfg->abi_version != FLACGPU_ABI_VERSION) return FLACGPU_ERR_BAD_ARG;
	if(cfg->channels < 1 || cfg->channels > FLACGPU_MAX_CHANNELS) return FLACGPU_ERR_UNSUPPORTED;
	if(cfg->bits_per_sample < 4 || cfg->bits_per_sample > 32) return FLACGPU_ERR_UNSUPPORTED;
	if(cfg->blocksize < 16 || cfg->blocksize > 65535) return FLACGPU_ERR_UNSUPPORTED;          // FLAC__MAX_BLOCK_SIZE
	if(cfg->max_lpc_order > (uint32_t)MAX_ORDER) return FLACGPU_ERR_UNSUPPORTED;       // FLAC__MAX_LPC_ORDER
	if(cfg->max_lpc_order > 0 && (cfg->qlp_coeff_precision < 5 || cfg->qlp_coeff_precision > 15)) return FLACGPU_ERR_UNSUPPORTED;
	if(cfg->max_residual_partition_order > MAX_PO) return FLACGPU_ERR_UNSUPPORTED;
	if(cfg->max_lpc_order > 0 && (cfg->num_apodizations < 1 || cfg->num_apodizations > FLACGPU_MAX_APODIZATIONS)) return FLACGPU_ERR_UNSUPPORTED;
	if(cfg->max_batch_frames < 1) return FLACGPU_ERR_BAD_ARG;
	for(uint32_t a = 0; a < cfg->num_apodizations && cfg->max_lpc_order > 0; a++)
		if(cfg->apodizations[a].kind == FLACGPU_APOD_SUBDIVIDE_TUKEY && cfg->apodizations[a].parts < 2) return FLACGPU_ERR_BAD_ARG;
	return FLACGPU_OK;
}

// the kernels' view of a (range-checked) configuration; UNSUPPORTED for what does not fit the engine's tables or the LDS
static int fill_params(const flacgpu_config *cfg, DevParams &P, JobTable *jobtab)
{
	memset(&P, 0, sizeof P);
	const uint32_t C = cfg->channels, N = cfg->blocksize, bps = cfg->bits_per_sample;
	P.channels = C; P.bps = bps; P.sample_rate = cfg->sample_rate; P.blocksize = N;
	// stream_encoder.c:737-741: mid/side only for stereo; loose only with mid/side
	const bool ms = cfg->do_mid_side_stereo && C == 2;
	P.ms_mode = ms ? (cfg->loose_mid_side_stereo ? 2u : 1u) : 0u;
	P.ncand = P.ms_mode == 1 ? 4u : C;
	P.wide_samples = bps > 24 ? 1u : 0u;
	P.chan_stride = (bps == 32 && P.ms_mode != 0) ? 2 * N : N;
	P.max_lpc_order = cfg->max_lpc_order;
	P.precision = cfg->qlp_coeff_precision;
	P.min_po = cfg->min_residual_partition_order;
	P.max_po = cfg->max_residual_partition_order;
	P.rice_limit = bps > 16 ? 31u : 15u;                       // stream_encoder.c:4076
	P.num_apod = cfg->max_lpc_order > 0 ? cfg->num_apodizations : 0;
	for(uint32_t a = 0; a < P.num_apod; a++) { P.apod_kind[a] = cfg->apodizations[a].kind; P.apod_parts[a] = cfg->apodizations[a].parts; }
	// stream_encoder.c:1058-1066 on an FMA-capable x86-64 host
	// 0: from order 16 up the plain C loop of lpc.c:133-157 does the work (lag > 16)
	P.autoc_variant = cfg->max_lpc_order < 8 ? 8u : cfg->max_lpc_order < 12 ? 12u : cfg->max_lpc_order < 16 ? 16u : 0u;
	P.disable_constant = cfg->disable_constant_subframes; P.disable_fixed = cfg->disable_fixed_subframes;
	P.disable_verbatim = cfg->disable_verbatim_subframes; P.limit_min_bitrate = cfg->limit_min_bitrate;
	P.slot_bytes = (uint32_t)worst_case_frame_bytes(C, N, bps);
	{
		const uint32_t maxidx = 32 + ((N + 15u) & ~15u) + 16u + 16u;
		P.sig_bytes = ((maxidx + ((maxidx >> 4) << 1) + 8) * 4 + 15) & ~15u;
		if(P.chan_stride != N) P.sig_bytes *= 2;               // 64-bit samples
		// window jobs of one subframe (same enumeration as analyze_kernel): all of them are windowed
		// into LDS at once so that their autocorrelation chains run concurrently
		uint32_t nj = 0, na = 0;
		for(uint32_t a = 0; a < P.num_apod; a++) {
			nj++; na++;
			if(P.apod_kind[a] == FLACGPU_APOD_SUBDIVIDE_TUKEY)
				for(uint32_t b = 2; b <= P.apod_parts[a]; b++) { if(N / b <= 32) continue; nj += b; na += b >= 3 ? 2 * b : b; }
		}
		if(nj > (uint32_t)MAX_JOBS || na > (uint32_t)MAX_ANALYSES) return FLACGPU_ERR_UNSUPPORTED;
		build_job_table(P, N, jobtab);
		P.max_jobs = nj ? nj : 1; P.max_analyses = na;
		P.exhaustive = cfg->do_exhaustive_model_search ? 1 : 0;
		P.prec_search = cfg->do_qlp_coeff_prec_search && cfg->max_lpc_order > 0 ? 1 : 0;
		P.nfixed = P.exhaustive ? 5 : 1;
		P.norders = P.exhaustive && cfg->max_lpc_order ? cfg->max_lpc_order : 1;
		P.nprec = P.prec_search ? 11 : 1;
		P.ncslots = P.nfixed + na * P.norders * P.nprec;
	}
	P.img_global = 0; P.stream_sig = 0;
	P.tune_flags = tune().cands_global ? 1u : 0u;
	if(N > 16384 || analyze_lds_bytes(P) > 160 * 1024 - 1024) { P.stream_sig = 1; P.sig_bytes = 0; }     // the block does not fit the LDS: the general kernels read HBM
	if(pack_lds_bytes(P) > 160 * 1024 - 1024) P.img_global = 1;            // many channels x long blocks: the frame is assembled in HBM
	if((uint64_t)P.slot_bytes > (uint64_t)4 * 1024 * 1024) return FLACGPU_ERR_UNSUPPORTED;      // CRC span table (flacgpu_kernels.hip)
	if(analyze_lds_bytes(P) > 160 * 1024 - 1024 || pack_lds_bytes(P) > 160 * 1024 - 1024) return FLACGPU_ERR_UNSUPPORTED;
	return FLACGPU_OK;
}

extern "C" int flacgpu_config_check(const flacgpu_config *cfg)
{
	int r = check_config(cfg);
	if(r != FLACGPU_OK) return r;
	DevParams P;
	JobTable *jt = new(std::nothrow) JobTable;
	if(!jt) return FLACGPU_ERR_ALLOC;
	r = fill_params(cfg, P, jt);
	delete jt;
	return r;
}

extern "C" int flacgpu_create(const flacgpu_config *cfg, const float *windows, flacgpu_ctx **out)
{
	if(!out) return FLACGPU_ERR_BAD_ARG;
	*out = nullptr;
	int r = check_config(cfg);
	if(r != FLACGPU_OK) return r;
	if(cfg->max_lpc_order > 0 && !windows) return FLACGPU_ERR_BAD_ARG;

	int ndev = 0;
	if(hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev) return FLACGPU_ERR_NO_DEVICE;
	if(hipSetDevice(cfg->device) != hipSuccess) return FLACGPU_ERR_NO_DEVICE;

	flacgpu_ctx *c = new(std::nothrow) flacgpu_ctx();
	if(!c) return FLACGPU_ERR_ALLOC;
	memset(c, 0, sizeof *c);
	c->cfg = *cfg;
	c->device = cfg->device;
	c->tune = read_tune(cfg->device);
	TuneScope tune_scope(&c->tune);
	r = fill_params(cfg, c->P, &c->h_jobtab[0]);
	if(r != FLACGPU_OK) { delete c; return r; }
	DevParams &P = c->P;
	const uint32_t N = cfg->blocksize;

	bool ok = true;
	ok = ok && hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess;
	for(int r = 0; r < TIMING_RING && ok; r++) {
		// (timing only: nothing synchronizes on these to READ memory -- without the system-scope fence a record stops costing the
		//  stream a cache write-back + invalidate, ~6 us of idle chip per event and six events a batch: profiles/r05_p_timeline_*.txt)
		const unsigned tflags = c->tune.event_fence ? 0u : hipEventDisableSystemFence;
		for(int i = 0; i < 5 && ok; i++) ok = hipEventCreateWithFlags(&c->ev_ring[r][i], tflags) == hipSuccess;
		for(int i = 0; i < 3 && ok; i++) ok = hipEventCreateWithFlags(&c->pev_ring[r][i], tflags) == hipSuccess;
	}
	c->ev = c->ev_ring[0]; c->pev = c->pev_ring[0];
	for(int i = 0; i < FLACGPU_MAX_SUBBATCHES && ok; i++) {
		ok = hipStreamCreateWithFlags(&c->sub_stream[i], hipStreamNonBlocking) == hipSuccess;
		ok = ok && hipEventCreateWithFlags(&c->sub_done[i], hipEventDisableTiming) == hipSuccess;
	}
	ok = ok && hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) == hipSuccess;
	c->nsub = 1;
	c->timing_every = 1;                                             // (every batch, until the caller says otherwise: flacgpu_set_phase_timing)
	if(const char *e = getenv("FLACGPU_TIMING_EVERY")) { const int v = atoi(e); if(v >= 0) c->timing_every = (uint32_t)v; }
	if(const char *e = getenv("FLACGPU_SUBBATCHES")) { const int v = atoi(e); if(v >= 1 && v <= FLACGPU_MAX_SUBBATCHES) c->nsub = (uint32_t)v; }
	const size_t B = cfg->max_batch_frames;
	const size_t wbytes = (size_t)(P.num_apod ? P.num_apod : 1) * N * sizeof(float);
	ok = ok && hipMalloc(&c->d_windows, wbytes) == hipSuccess;
	ok = ok && hipMalloc(&c->d_tail_windows, wbytes) == hipSuccess;
	ok = ok && hipMalloc(&c->d_decisions, B * P.ncand * sizeof(SubDecision)) == hipSuccess;
	ok = ok && hipMalloc(&c->d_plan, B * pack_plan_stride(P)) == hipSuccess;
	ok = ok && hipMalloc(&c->d_slots, B * P.slot_bytes + 64) == hipSuccess;       // (+64: fo_copy_slot reads whole 16-byte pieces behind a frame)
	ok = ok && hipMalloc(&c->d_frame_bytes, B * sizeof(uint32_t)) == hipSuccess;
	ok = ok && hipMalloc(&c->d_offsets, (B + 1) * sizeof(uint64_t)) == hipSuccess;
	ok = ok && hipMalloc(&c->d_total, sizeof(uint64_t)) == hipSuccess;
	{
		// fused output: zeroed here, once (epoch 0 is never used), and whenever the epoch wraps
		c->fo_frames = (B + 63) & ~(size_t)63; c->fo_segs = c->fo_frames / 64; c->fo_epoch = 0;
		ok = ok && hipMalloc(&c->d_fo_fstate, c->fo_frames * sizeof(uint64_t)) == hipSuccess && hipMemset(c->d_fo_fstate, 0, c->fo_frames * sizeof(uint64_t)) == hipSuccess;
		ok = ok && hipMalloc(&c->d_fo_sstate, c->fo_segs * sizeof(uint64_t)) == hipSuccess && hipMemset(c->d_fo_sstate, 0, c->fo_segs * sizeof(uint64_t)) == hipSuccess;
		ok = ok && hipMalloc(&c->d_fo_scount, c->fo_segs * sizeof(uint64_t)) == hipSuccess && hipMemset(c->d_fo_scount, 0, c->fo_segs * sizeof(uint64_t)) == hipSuccess;
		ok = ok && hipMalloc(&c->d_fo_sprefix, c->fo_segs * sizeof(uint64_t)) == hipSuccess && hipMemset(c->d_fo_sprefix, 0, c->fo_segs * sizeof(uint64_t)) == hipSuccess;
		ok = ok && hipMalloc(&c->d_fo_fall, c->fo_frames * sizeof(uint32_t)) == hipSuccess;
		ok = ok && hipMalloc(&c->d_fo_nfall, 4 * sizeof(uint32_t)) == hipSuccess && hipMemset(c->d_fo_nfall, 0, 4 * sizeof(uint32_t)) == hipSuccess;
		// a poll is three loads and a short sleep, ~1 us: a frame gives up after a few milliseconds (FLACGPU_FUSED_SPIN_LIMIT=0: at once
		// -- the tests' way to the slot + fo_fixup_kernel route)
		c->fo_spin_limit = 4096;
		if(const char *e = getenv("FLACGPU_FUSED_SPIN_LIMIT")) c->fo_spin_limit = (uint32_t)strtoul(e, nullptr, 10);
		c->ff_lag = -1;
		if(const char *e = getenv("FLACGPU_FF_LAG")) c->ff_lag = atoi(e);
	}
	ok = ok && hipMalloc(&c->d_info, B * sizeof(FrameInfo)) == hipSuccess;
	ok = ok && hipMalloc(&c->d_jobtab, 2 * sizeof(JobTable)) == hipSuccess;
	ok = ok && hipMemcpy(c->d_jobtab, c->h_jobtab, sizeof(JobTable), hipMemcpyHostToDevice) == hipSuccess;
	if(ok && P.num_apod) ok = hipMemcpy(c->d_windows, windows, wbytes, hipMemcpyHostToDevice) == hipSuccess;
	{
		const size_t nfc = B * P.ncand, ncs = P.ncslots;
		ok = ok && hipMalloc(&c->ab.prep, nfc * sizeof(ChanPrep)) == hipSuccess;
		ok = ok && hipMalloc(&c->ab.autoc, nfc * P.max_jobs * AUTOC_STRIDE * sizeof(double)) == hipSuccess;
		ok = ok && hipMalloc(&c->ab.cands, nfc * ncs * sizeof(Candidate)) == hipSuccess;
		ok = ok && hipMalloc(&c->ab.valid, nfc * ncs * sizeof(int)) == hipSuccess;
		ok = ok && hipMalloc(&c->ab.chan, nfc * P.chan_stride * sizeof(int32_t)) == hipSuccess;
		ok = ok && hipMalloc(&c->ab.left, nfc * sizeof(uint32_t)) == hipSuccess;
		ok = ok && hipMalloc(&c->ab.left2, nfc * sizeof(uint32_t)) == hipSuccess;
		ok = ok && hipMalloc(&c->ab.nleft, 2 * FLACGPU_MAX_SUBBATCHES * sizeof(uint32_t)) == hipSuccess;
		c->ff_ok = ff_applicable(P);
		if(ok && getenv("FLACGPU_DEBUG_TIMING")) { ok = hipMalloc(&c->ab.dbg, nfc * 16 * sizeof(unsigned long long)) == hipSuccess; if(ok) (void)hipMemset(c->ab.dbg, 0, nfc * 16 * sizeof(unsigned long long)); }
	}
	if(!ok) { free_ctx(c); return FLACGPU_ERR_ALLOC; }
	if(getenv("FLACGPU_POISON")) {
		// development aid (tests/test_gpu_parity.py): every scratch buffer starts out as garbage instead of the zeros fresh device
		// memory happens to hold, so that a kernel reading what no kernel wrote shows up as a parity failure
		const size_t nfc = B * P.ncand, ncs = P.ncslots;
		(void)hipMemset(c->d_decisions, 0xA5, B * P.ncand * sizeof(SubDecision));
		(void)hipMemset(c->d_plan, 0xA5, B * pack_plan_stride(P));
		(void)hipMemset(c->d_slots, 0xA5, B * P.slot_bytes);
		(void)hipMemset(c->d_frame_bytes, 0xA5, B * sizeof(uint32_t));
		(void)hipMemset(c->d_offsets, 0xA5, (B + 1) * sizeof(uint64_t));
		(void)hipMemset(c->d_info, 0xA5, B * sizeof(FrameInfo));
		(void)hipMemset(c->ab.prep, 0xA5, nfc * sizeof(ChanPrep));
		(void)hipMemset(c->ab.autoc, 0xA5, nfc * P.max_jobs * AUTOC_STRIDE * sizeof(double));
		(void)hipMemset(c->ab.cands, 0xA5, nfc * ncs * sizeof(Candidate));
		(void)hipMemset(c->ab.valid, 0xA5, nfc * ncs * sizeof(int));
		(void)hipMemset(c->ab.chan, 0xA5, nfc * P.chan_stride * sizeof(int32_t));
		(void)hipDeviceSynchronize();
	}
	*out = c;
	return FLACGPU_OK;
}

extern "C" void flacgpu_destroy(flacgpu_ctx *ctx) { free_ctx(ctx); }

extern "C" size_t flacgpu_max_output_bytes(const flacgpu_ctx *ctx, uint32_t nframes)
{
	return ctx ? (size_t)nframes * ctx->P.slot_bytes : 0;
}
extern "C" size_t flacgpu_config_max_output_bytes(const flacgpu_config *cfg, uint32_t nframes)
{
	return cfg ? (size_t)nframes * (size_t)worst_case_frame_bytes(cfg->channels, cfg->blocksize, cfg->bits_per_sample) : 0;
}

static int run_batch(flacgpu_ctx *c, const int32_t *d_pcm, uint32_t nframes, uint64_t first, uint32_t tail_n,
                     const float *tail_windows_host, uint8_t *d_out, size_t out_cap, uint32_t *d_fb_out,
                     uint64_t *d_total_out, hipStream_t s)
{
	if(!c || !d_pcm || !d_out || nframes == 0 || nframes > c->cfg.max_batch_frames) return FLACGPU_ERR_BAD_ARG;
	if(((uintptr_t)d_fb_out & 3u) || ((uintptr_t)d_total_out & 7u)) return FLACGPU_ERR_BAD_ARG;      // (the kernels write them in place: include/flacgpu.h)
	if(tail_n >= c->P.blocksize) tail_n = 0;
	if(hipSetDevice(c->device) != hipSuccess) return FLACGPU_ERR_NO_DEVICE;
	TuneScope tune_scope(&c->tune);
	c->tune.launched = 0;
	struct KeepLaunched { flacgpu_ctx *c; ~KeepLaunched() { c->last_launched = c->tune.launched; } } keep_launched{c};
	const DevParams &P = c->P;
	// frame lengths and the stream's length are written where the caller wants them (they were copied there behind the last kernel:
	// two more dispatches per batch -- a tenth of a -0 step)
	uint32_t *const fb = d_fb_out && !c->tune.copy_results ? d_fb_out : c->d_frame_bytes;
	uint64_t *const tot = d_total_out && !c->tune.copy_results ? d_total_out : c->d_total;
	c->last_fb = fb;
	if(tail_n) {
		// The short last block has its own job schedule and windows.  Both live in one device copy each, and the sources are
		// pageable host memory (this context; the caller's array): wait until an earlier batch that may still be reading the
		// device copies has drained, then copy synchronously.  Once per stream -- the price is nil.
		if(P.num_apod && !tail_windows_host) return FLACGPU_ERR_BAD_ARG;
		if(hipStreamSynchronize(s) != hipSuccess) return FLACGPU_ERR_LAUNCH;
		build_job_table(P, tail_n, &c->h_jobtab[1]);
		if(hipMemcpy(c->d_jobtab + 1, &c->h_jobtab[1], sizeof(JobTable), hipMemcpyHostToDevice) != hipSuccess) return FLACGPU_ERR_LAUNCH;
		if(P.num_apod && hipMemcpy(c->d_tail_windows, tail_windows_host, (size_t)P.num_apod * tail_n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return FLACGPU_ERR_LAUNCH;
	}
	c->ev = c->ev_ring[c->batch_seq % TIMING_RING]; c->pev = c->pev_ring[c->batch_seq % TIMING_RING];
	const size_t ring_slot = c->batch_seq % TIMING_RING;
	hipEvent_t *seq = c->seq_ring[ring_slot];
	// phase timing is instrumentation: every n-th batch carries the event records (flacgpu_set_phase_timing; an event record costs the
	// stream ~4.6 us, six of them 1.4 % of a -8 step of 16384 frames and a fifth of a -0 step)
	const bool timed = c->timing_every && (c->batch_seq % c->timing_every) == 0;
	c->timed_ring[ring_slot] = timed;
	c->batch_seq++;
	if(timed) (void)hipEventRecord(c->ev[0], s);
	bool fused = false;
	uint32_t nsub = c->nsub;
	// The fused output (flacgpu_kernels.hip, PackOut): the pack kernels (pack2_kernel, ff_kernel) write every frame once, at its
	// final place; no slots, no scan / compact kernels.  FLACGPU_NO_FUSED_COMPACT=1: the two-kernel compaction, for A/B runs and the
	// parity tests of that path.
	const int fuse = c->tune.no_fused ? 0 : 1;
	PackOutArgs po = {nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0};
	// ff_kernel (one kernel for the whole frame, flacgpu_kernels.hip) where it applies: not with the verify hints (the decoder wants
	// the pack kernel's run starts), not with the debug stamps; one stream
	const bool ff = c->ff_ok && !c->d_vhints && !c->ab.dbg;
	// (ff_kernel places its frames itself only when asked to, launch_ff: otherwise the batch has no use for the fused output's words)
	if(fuse && nframes <= c->fo_frames && !(ff && c->ff_lag < 0)) {
		uint32_t epoch = c->fo_epoch + 1;
		if(epoch >= (1u << 24) || c->fo_dirty) {
			// the tags have gone round: start again from clean words (once in sixteen million batches) -- or the batch before ended in an
			// error with kernels of it possibly run: counters its last users would have zeroed may not be zero
			if(hipMemsetAsync(c->d_fo_fstate, 0, c->fo_frames * sizeof(uint64_t), s) != hipSuccess || hipMemsetAsync(c->d_fo_sstate, 0, c->fo_segs * sizeof(uint64_t), s) != hipSuccess ||
			   hipMemsetAsync(c->d_fo_scount, 0, c->fo_segs * sizeof(uint64_t), s) != hipSuccess ||
			   hipMemsetAsync(c->d_fo_sprefix, 0, c->fo_segs * sizeof(uint64_t), s) != hipSuccess || hipMemsetAsync(c->d_fo_nfall, 0, 2 * sizeof(uint32_t), s) != hipSuccess) return FLACGPU_ERR_LAUNCH;
			c->fo_epoch = 0; epoch = 1; c->fo_last_fused = 0;
		}
		else if(c->fo_last_fused + 1 != epoch && hipMemsetAsync(c->d_fo_nfall, 0, 2 * sizeof(uint32_t), s) != hipSuccess) return FLACGPU_ERR_LAUNCH;
		po = PackOutArgs{d_out, out_cap, c->d_offsets, tot, c->d_fo_fstate, c->d_fo_sstate, c->d_fo_sprefix, c->d_fo_scount, c->d_fo_fall, c->d_fo_nfall, epoch, c->fo_spin_limit, c->ff_lag > 0 ? (uint32_t)c->ff_lag : 0u};
		// The epoch is spent from here on, whether or not the launches below succeed: a kernel that tagged words with it may have run
		// before a later launch fails, and the next batch must not take those words for its own.  An unused epoch costs
		// nothing: the tags only have to differ from batch to batch.  (The arrival counters are zeroed by their last user; a batch
		// that aborted half way may leave some non-zero: fo_dirty makes the next batch start from clean words.)
		c->fo_epoch = epoch;
		c->fo_dirty = true;          // (until every launch of this batch has been enqueued)
	}
	if(c->ab.dbg || nframes < 256 * nsub || ff) nsub = 1;
	const bool lpc = P.max_analyses != 0;
	if(ff) {
		// every frame of nominal length in ONE launch; a short last block goes through the general kernels as a batch of its own
		// (every buffer is indexed by frame: the same launches on offset pointers) and then, with the fused output, behind the others
		const uint32_t nmain = tail_n ? nframes - 1 : nframes;
		const bool ffpo = po.out != nullptr && c->ff_lag >= 0;          // (the kernel places its frames itself only when asked to: launch_ff)
		if(launch_ff(P, d_pcm, nmain, first, c->d_slots, fb, c->d_info, ffpo ? &po : nullptr, s) != hipSuccess) return FLACGPU_ERR_LAUNCH;
		fused = ffpo && nmain != 0;
		hipEvent_t after_main = c->ev[3];
		if(tail_n) {
			if(timed) (void)hipEventRecord(c->pev[0], s);
			after_main = c->pev[0];
			const size_t fc0 = (size_t)nmain * P.ncand, ncs = P.ncslots;
			AnalyzeBuffers B = c->ab;
			B.prep += fc0; B.autoc += fc0 * P.max_jobs * AUTOC_STRIDE; B.cands += fc0 * ncs; B.valid += fc0 * ncs; B.chan += fc0 * P.chan_stride; B.left += fc0; B.left2 += fc0; B.dbg = nullptr;
			if(launch_analyze(P, d_pcm + (size_t)nmain * P.blocksize * P.channels, c->d_windows, c->d_tail_windows, 1, tail_n, c->d_jobtab, c->d_jobtab + 1, c->h_jobtab[0].nsets, B,
			                  c->d_decisions + fc0, nullptr, s) != hipSuccess) return FLACGPU_ERR_LAUNCH;
			if(launch_pack(P, B.chan, 1, tail_n, first + nmain, c->d_decisions + fc0, c->d_plan + (size_t)nmain * pack_plan_stride(P), c->d_slots + (size_t)nmain * P.slot_bytes, fb + nmain, c->d_info + nmain, nullptr, nullptr, nullptr, nullptr, nullptr, s) != hipSuccess)
				return FLACGPU_ERR_LAUNCH;
			if(fused && launch_append_tail(c->d_slots + (size_t)nmain * P.slot_bytes, fb, nmain, &po, s) != hipSuccess) return FLACGPU_ERR_LAUNCH;
		}
		c->hint_count = 0;
		// (the kernel's time is booked under "prep", a short last block's under "pack", the two-kernel compaction's under its own name)
		seq[0] = c->ev[0]; seq[1] = seq[2] = seq[3] = seq[4] = after_main;
		if(fused) seq[5] = seq[6] = c->ev[3];
		else {
			if(tail_n) { if(timed) (void)hipEventRecord(c->ev[2], s); seq[5] = c->ev[2]; }
			else { if(timed) (void)hipEventRecord(c->pev[0], s); seq[1] = seq[2] = seq[3] = seq[4] = seq[5] = c->pev[0]; }
			seq[6] = c->ev[3];
		}
	}
	else if(nsub > 1) {
		// Independent sub-batches on their own streams: the latency-bound kernels of one (prep, model, pack) fill the
		// gaps of the VALU-bound kernels of another (autoc, eval).  Every buffer is indexed by frame, so a sub-batch
		// is the same launches on offset pointers; they join before the scan over all frame lengths.
		(void)hipEventRecord(c->ev_fork, s);
		for(uint32_t i = 0; i < nsub; i++) {
			const uint32_t f0 = (uint32_t)((uint64_t)nframes * i / nsub), f1 = (uint32_t)((uint64_t)nframes * (i + 1) / nsub), nf = f1 - f0;
			hipStream_t ss = c->sub_stream[i];
			(void)hipStreamWaitEvent(ss, c->ev_fork, 0);
			const size_t fc0 = (size_t)f0 * P.ncand, ncs = P.ncslots;
			AnalyzeBuffers B = c->ab;
			B.prep += fc0; B.autoc += fc0 * P.max_jobs * AUTOC_STRIDE; B.cands += fc0 * ncs; B.valid += fc0 * ncs; B.chan += fc0 * P.chan_stride; B.left += fc0; B.left2 += fc0; B.nleft += 2 * i; B.dbg = nullptr;
			const uint32_t tn = i + 1 == nsub ? tail_n : 0;
			if(launch_analyze(P, d_pcm + (size_t)f0 * P.blocksize * P.channels, c->d_windows, c->d_tail_windows, nf, tn, c->d_jobtab, c->d_jobtab + 1, c->h_jobtab[0].nsets, B,
			                  c->d_decisions + fc0, nullptr, ss) != hipSuccess) return FLACGPU_ERR_LAUNCH;
			if(launch_pack(P, B.chan, nf, tn, first + f0, c->d_decisions + fc0, c->d_plan + (size_t)f0 * pack_plan_stride(P), c->d_slots + (size_t)f0 * P.slot_bytes, fb + f0, c->d_info + f0, nullptr, nullptr, nullptr, nullptr, nullptr, ss) != hipSuccess)
				return FLACGPU_ERR_LAUNCH;
			(void)hipEventRecord(c->sub_done[i], ss);
			(void)hipStreamWaitEvent(s, c->sub_done[i], 0);
		}
		c->hint_count = 0;
		// the per-kernel events of the single-stream path are not meaningful here: everything is booked under "pack"
		if(timed) (void)hipEventRecord(c->ev[2], s);
		seq[0] = seq[1] = seq[2] = seq[3] = seq[4] = c->ev[0]; seq[5] = c->ev[2]; seq[6] = c->ev[3];
	}
	else {
		if(launch_analyze(P, d_pcm, c->d_windows, c->d_tail_windows, nframes, tail_n, c->d_jobtab, c->d_jobtab + 1, c->h_jobtab[0].nsets, c->ab, c->d_decisions, timed ? c->pev : nullptr, s) != hipSuccess) return FLACGPU_ERR_LAUNCH;
		if(c->ab.dbg) {
			// development aid: average shader cycles per phase of the eval workgroups of this launch
			const size_t nwg = (size_t)nframes * P.ncand;
			unsigned long long *h = (unsigned long long *)malloc(nwg * 16 * sizeof(unsigned long long));
			if(h && hipStreamSynchronize(s) == hipSuccess && hipMemcpy(h, c->ab.dbg, nwg * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess) {
				double acc[6] = {0}; size_t cnt[6] = {0};
				for(size_t w = 0; w < nwg; w++) for(int k = 1; k < 6; k++) if(h[w * 16 + k] && h[w * 16 + k - 1]) { acc[k] += (double)(h[w * 16 + k] - h[w * 16 + k - 1]); cnt[k]++; }
				double pro = 0; size_t npro = 0;
				for(size_t w = 0; w < nwg; w++) if(h[w * 16 + 8] && h[w * 16]) { pro += (double)(h[w * 16 + 8] - h[w * 16]); npro++; }
				fprintf(stderr, "[flacgpu] eval prologue %.0f ticks (%zu WGs)\n", npro ? pro / npro : 0, npro);
				for(int v = 1; v <= 3; v++) {
					unsigned long long t0 = ~0ull, t1 = 0; double dur = 0; size_t nv = 0;
					for(size_t w = 0; w < nwg; w++) if(h[w * 16 + 9] == (unsigned long long)v && h[w * 16 + 5]) {
						nv++; dur += (double)(h[w * 16 + 5] - h[w * 16]);
						if(h[w * 16] < t0) t0 = h[w * 16];
						if(h[w * 16 + 5] > t1) t1 = h[w * 16 + 5];
					}
					if(nv) fprintf(stderr, "[flacgpu] eval variant %d: %zu WGs, avg %.0f ticks each, first start -> last end %.0f ticks, mean concurrency %.1f WGs\n",
					               v - 1, nv, dur / nv, (double)(t1 - t0), dur / (double)(t1 - t0));
				}
				{
					double fa[16] = {0}; size_t fn[16] = {0};
					for(size_t w = 0; w < nwg; w++) for(int k = 1; k < 16; k++) if(k != 9 && h[w * 16 + k] && h[w * 16]) { fa[k] += (double)(h[w * 16 + k] - h[w * 16]); fn[k]++; }
					fprintf(stderr, "[flacgpu] eval stamps (avg ticks since the workgroup's start): facts+barrier 6:%.0f  offsets+divtab+barrier 7:%.0f  ch0 records copied 10:%.0f  ch1 records copied 12:%.0f  signals in LDS 1:%.0f  first candidate 2:%.0f  all candidates 4:%.0f  end 5:%.0f\n",
					        fn[6] ? fa[6] / fn[6] : 0, fn[7] ? fa[7] / fn[7] : 0, fn[10] ? fa[10] / fn[10] : 0, fn[12] ? fa[12] / fn[12] : 0, fn[1] ? fa[1] / fn[1] : 0, fn[2] ? fa[2] / fn[2] : 0, fn[4] ? fa[4] / fn[4] : 0, fn[5] ? fa[5] / fn[5] : 0);
				}
				fprintf(stderr, "[flacgpu] eval phases (avg s_memtime ticks per WG): load %.0f  cand0 %.0f  round1->round2 %.0f  rest %.0f  decide %.0f\n",
				        cnt[1] ? acc[1] / cnt[1] : 0, cnt[2] ? acc[2] / cnt[2] : 0, cnt[3] ? acc[3] / cnt[3] : 0, cnt[4] ? acc[4] / cnt[4] : 0, cnt[5] ? acc[5] / cnt[5] : 0);
			}
			free(h);
			(void)hipMemsetAsync(c->ab.dbg, 0, nwg * 16 * sizeof(unsigned long long), s);
		}
		if(timed) (void)hipEventRecord(c->ev[1], s);
		{
			uint32_t hinted = 0;
			// (not with the debug stamps: they are indexed by workgroup, the fused output takes frames in dispatch order)
			if(launch_pack(P, c->ab.chan, nframes, tail_n, first, c->d_decisions, c->d_plan, c->d_slots, fb, c->d_info, c->ab.dbg, po.out && !c->ab.dbg ? &po : nullptr, &fused, c->d_vhints, &hinted, s) != hipSuccess) return FLACGPU_ERR_LAUNCH;
			c->hint_out = d_out; c->hint_nframes = nframes; c->hint_first = first; c->hint_count = hinted;
		}
		if(c->ab.dbg) {
			// development aid: wall cycles of the pack2 workgroups between their stamps
			unsigned long long *h = (unsigned long long *)malloc((size_t)nframes * 16 * sizeof(unsigned long long));
			if(h && hipStreamSynchronize(s) == hipSuccess && hipMemcpy(h, c->ab.dbg, (size_t)nframes * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess) {
				double acc[13] = {0}; size_t cnt[13] = {0};
				for(size_t w = 0; w < nframes; w++) { int prev = 0; for(int k = 1; k < 13; k++) if(h[w * 16 + k] && h[w * 16 + prev]) { acc[k] += (double)(h[w * 16 + k] - h[w * 16 + prev]); cnt[k]++; prev = k; } }
				fprintf(stderr, "[flacgpu] pack2 stamps (avg ticks since previous stamp):");
				for(int k = 1; k < 13; k++) fprintf(stderr, " %d:%.0f", k, cnt[k] ? acc[k] / cnt[k] : 0.0);
				fprintf(stderr, "\n");
			}
			free(h);
			(void)hipMemsetAsync(c->ab.dbg, 0, (size_t)nframes * P.ncand * 16 * sizeof(unsigned long long), s);
		}
		// without LPC analyses nothing is launched between the prep and the evaluation phase, and no event is recorded there
		seq[0] = c->ev[0]; seq[1] = c->pev[0]; seq[2] = lpc ? c->pev[1] : c->pev[0]; seq[3] = lpc ? c->pev[2] : c->pev[0]; seq[4] = c->ev[1];
		if(fused) seq[5] = seq[6] = c->ev[3];
		else { if(timed) (void)hipEventRecord(c->ev[2], s); seq[5] = c->ev[2]; seq[6] = c->ev[3]; }
	}
	if(fused) { note_launch(K_FUSED_OUTPUT); c->fo_last_fused = po.epoch; }
	if(!fused) {
		if(launch_scan(fb, nframes, c->d_offsets, tot, s) != hipSuccess) return FLACGPU_ERR_LAUNCH;
		if(launch_compact(c->d_slots, P.slot_bytes, fb, c->d_offsets, d_out, out_cap, nframes, s) != hipSuccess) return FLACGPU_ERR_LAUNCH;
	}
	if(timed) (void)hipEventRecord(c->ev[3], s);
	if(d_fb_out && fb != d_fb_out && hipMemcpyAsync(d_fb_out, fb, nframes * sizeof(uint32_t), hipMemcpyDeviceToDevice, s) != hipSuccess) return FLACGPU_ERR_LAUNCH;
	if(d_total_out && tot != d_total_out && hipMemcpyAsync(d_total_out, tot, sizeof(uint64_t), hipMemcpyDeviceToDevice, s) != hipSuccess) return FLACGPU_ERR_LAUNCH;
	c->last_nframes = nframes;
	c->timing_valid = true;
	c->fo_dirty = false;
	return FLACGPU_OK;
}

extern "C" int flacgpu_encode_batch_device(flacgpu_ctx *ctx, const int32_t *d_pcm, uint32_t nframes,
                                           uint64_t first_frame_number, uint32_t last_block_samples,
                                           const float *tail_windows_host, uint8_t *d_out, size_t out_cap,
                                           uint32_t *d_frame_bytes, uint64_t *d_total_bytes, void *stream)
{
	return run_batch(ctx, d_pcm, nframes, first_frame_number, last_block_samples, tail_windows_host, d_out, out_cap,
	                 d_frame_bytes, d_total_bytes, stream ? (hipStream_t)stream : (ctx ? ctx->stream : nullptr));
}

static int make_stage_params(const flacgpu_ctx *c, const flacgpu_raw_format *fmt, StageParams *S)
{
	if(!fmt || (fmt->container_bits != 8 && fmt->container_bits != 16 && fmt->container_bits != 24 && fmt->container_bits != 32)) return FLACGPU_ERR_BAD_ARG;
	if(fmt->shift >= fmt->container_bits) return FLACGPU_ERR_BAD_ARG;
	memset(S, 0, sizeof *S);
	S->bytes = fmt->container_bits / 8; S->big_endian = fmt->big_endian ? 1 : 0; S->is_unsigned = fmt->is_unsigned ? 1 : 0;
	S->shift = fmt->shift; S->channels = c->P.channels; S->use_map = fmt->use_channel_map ? 1 : 0;
	uint32_t seen = 0;
	for(uint32_t ch = 0; ch < c->P.channels; ch++) {
		S->map[ch] = S->use_map ? fmt->channel_map[ch] : (uint8_t)ch;
		if(S->map[ch] >= c->P.channels || (seen & (1u << S->map[ch]))) return FLACGPU_ERR_BAD_ARG;      // must be a permutation
		seen |= 1u << S->map[ch];
	}
	return FLACGPU_OK;
}

extern "C" int flacgpu_stage_raw_device(flacgpu_ctx *c, const void *d_raw, const flacgpu_raw_format *fmt, uint64_t wide_samples,
                                        int32_t *d_pcm, uint32_t *d_error, void *stream)
{
	if(!c || !d_raw || !d_pcm) return FLACGPU_ERR_BAD_ARG;
	StageParams S;
	const int r = make_stage_params(c, fmt, &S);
	if(r != FLACGPU_OK) return r;
	if(hipSetDevice(c->device) != hipSuccess) return FLACGPU_ERR_NO_DEVICE;
	if(launch_stage_raw(S, d_raw, wide_samples * c->P.channels, d_pcm, d_error, stream ? (hipStream_t)stream : c->stream) != hipSuccess) return FLACGPU_ERR_LAUNCH;
	return FLACGPU_OK;
}

// the device buffers of the host entry points
static int ensure_host_staging(flacgpu_ctx *c, uint32_t nframes)
{
	const DevParams &P = c->P;
	// the kernels index frames at a fixed stride of blocksize*channels; staging is sized for full frames
	const size_t pcm_bytes = (size_t)nframes * P.blocksize * P.channels * sizeof(int32_t);
	const size_t out_bytes = (size_t)nframes * P.slot_bytes;
	if(c->d_pcm_bytes < pcm_bytes) {
		if(c->d_pcm) (void)hipFree(c->d_pcm);
		c->d_pcm = nullptr; c->d_pcm_bytes = 0;
		if(hipMalloc(&c->d_pcm, pcm_bytes) != hipSuccess) return FLACGPU_ERR_ALLOC;
		c->d_pcm_bytes = pcm_bytes;
	}
	if(c->d_out_bytes < out_bytes) {
		if(c->d_out) (void)hipFree(c->d_out);
		c->d_out = nullptr; c->d_out_bytes = 0;
		if(hipMalloc(&c->d_out, out_bytes) != hipSuccess) return FLACGPU_ERR_ALLOC;
		c->d_out_bytes = out_bytes;
	}
	return FLACGPU_OK;
}
// how many leading frames of this verify call the hints in d_vhints describe: they were written by the pack kernel for the last
// batch this engine encoded, and they are used when that batch comes back (same buffer, same frames).  The hinted pass does not
// trust them -- stale hints cost time, not correctness.
static uint32_t hints_for(const flacgpu_ctx *c, const uint8_t *d_frames, uint32_t nframes, uint64_t first)
{
	if(!c->d_vhints || c->hint_count == 0 || c->force_hints == 0) return 0;
	if(c->force_hints == 1) return c->hint_count < nframes ? c->hint_count : nframes;
	if(d_frames != c->hint_out || nframes != c->hint_nframes || first != c->hint_first) return 0;
	return c->hint_count;
}
// encode c->d_pcm (already filled on the engine's stream) and bring the frames back
static int64_t encode_staged(flacgpu_ctx *c, uint32_t nframes, uint64_t first_frame_number, uint32_t tail_n, const float *tail_windows,
                             uint8_t *out, size_t out_cap, uint32_t *frame_bytes)
{
	hipStream_t s = c->stream;
	TuneScope tune_scope(&c->tune);
	int r = run_batch(c, c->d_pcm, nframes, first_frame_number, tail_n, tail_windows, c->d_out, c->d_out_bytes, nullptr, nullptr, s);
	if(r != FLACGPU_OK) return r;
	memset(&c->last_verify, 0, sizeof c->last_verify);
	if(c->verify_on) {
		// the frames are decoded again where they lie and compared with the staged input (stream_encoder.c:3000-3018)
		if(launch_verify(c->P, c->d_out, c->d_frame_bytes, c->d_offsets, nframes, tail_n, first_frame_number, c->d_pcm, c->d_vscratch, c->d_vdecoded, c->d_vfinfo, c->d_vstate, c->d_vresult,
		                 c->d_vhints, hints_for(c, c->d_out, nframes, first_frame_number), c->d_vfstat, c->ab.dbg, s) != hipSuccess)
			return FLACGPU_ERR_LAUNCH;
		if(hipMemcpyAsync(&c->last_verify, c->d_vresult, sizeof c->last_verify, hipMemcpyDeviceToHost, s) != hipSuccess) return FLACGPU_ERR_LAUNCH;
	}
	uint64_t total = 0;
	if(hipMemcpyAsync(frame_bytes, c->d_frame_bytes, nframes * sizeof(uint32_t), hipMemcpyDeviceToHost, s) != hipSuccess) return FLACGPU_ERR_LAUNCH;
	if(hipMemcpyAsync(&total, c->d_total, sizeof total, hipMemcpyDeviceToHost, s) != hipSuccess) return FLACGPU_ERR_LAUNCH;
	if(hipStreamSynchronize(s) != hipSuccess) return FLACGPU_ERR_LAUNCH;
	for(uint32_t i = 0; i < nframes; i++) if(frame_bytes[i] == 0xffffffffu) return FLACGPU_ERR_LAUNCH;
	if(total > out_cap) return FLACGPU_ERR_OUTPUT_TOO_SMALL;
	if(hipMemcpy(out, c->d_out, total, hipMemcpyDeviceToHost) != hipSuccess) return FLACGPU_ERR_LAUNCH;
	return (int64_t)total;
}

extern "C" int64_t flacgpu_encode_batch(flacgpu_ctx *c, const int32_t *pcm, uint32_t nframes,
                                        uint64_t first_frame_number, uint32_t last_block_samples,
                                        const float *tail_windows, uint8_t *out, size_t out_cap,
                                        uint32_t *frame_bytes)
{
	if(!c || !pcm || !out || !frame_bytes || nframes == 0 || nframes > c->cfg.max_batch_frames) return FLACGPU_ERR_BAD_ARG;
	if(hipSetDevice(c->device) != hipSuccess) return FLACGPU_ERR_NO_DEVICE;
	const DevParams &P = c->P;
	const uint32_t tail_n = last_block_samples < P.blocksize ? last_block_samples : 0;
	const size_t nsamp = (size_t)(nframes - 1) * P.blocksize + (tail_n ? tail_n : P.blocksize);
	const int r = ensure_host_staging(c, nframes);
	if(r != FLACGPU_OK) return r;
	if(hipMemcpyAsync(c->d_pcm, pcm, nsamp * P.channels * sizeof(int32_t), hipMemcpyHostToDevice, c->stream) != hipSuccess) return FLACGPU_ERR_LAUNCH;
	return encode_staged(c, nframes, first_frame_number, tail_n, tail_windows, out, out_cap, frame_bytes);
}

extern "C" int64_t flacgpu_encode_batch_raw(flacgpu_ctx *c, const void *raw, const flacgpu_raw_format *fmt, uint32_t nframes,
                                            uint64_t first_frame_number, uint32_t last_block_samples,
                                            const float *tail_windows, uint8_t *out, size_t out_cap,
                                            uint32_t *frame_bytes)
{
	if(!c || !raw || !out || !frame_bytes || nframes == 0 || nframes > c->cfg.max_batch_frames) return FLACGPU_ERR_BAD_ARG;
	StageParams S;
	int r = make_stage_params(c, fmt, &S);
	if(r != FLACGPU_OK) return r;
	if(hipSetDevice(c->device) != hipSuccess) return FLACGPU_ERR_NO_DEVICE;
	const DevParams &P = c->P;
	const uint32_t tail_n = last_block_samples < P.blocksize ? last_block_samples : 0;
	const size_t nsamp = (size_t)(nframes - 1) * P.blocksize + (tail_n ? tail_n : P.blocksize);
	r = ensure_host_staging(c, nframes);
	if(r != FLACGPU_OK) return r;
	const size_t raw_bytes = nsamp * P.channels * S.bytes;
	if(c->d_raw_bytes < raw_bytes) {
		if(c->d_raw) (void)hipFree(c->d_raw);
		c->d_raw = nullptr; c->d_raw_bytes = 0;
		if(hipMalloc(&c->d_raw, raw_bytes) != hipSuccess) return FLACGPU_ERR_ALLOC;
		c->d_raw_bytes = raw_bytes;
	}
	if(!c->d_stage_err && hipMalloc(&c->d_stage_err, sizeof(uint32_t)) != hipSuccess) return FLACGPU_ERR_ALLOC;
	hipStream_t s = c->stream;
	uint32_t herr = 0;
	if(hipMemsetAsync(c->d_stage_err, 0, sizeof(uint32_t), s) != hipSuccess) return FLACGPU_ERR_LAUNCH;
	if(hipMemcpyAsync(c->d_raw, raw, raw_bytes, hipMemcpyHostToDevice, s) != hipSuccess) return FLACGPU_ERR_LAUNCH;
	if(launch_stage_raw(S, c->d_raw, (uint64_t)nsamp * P.channels, c->d_pcm, c->d_stage_err, s) != hipSuccess) return FLACGPU_ERR_LAUNCH;
	if(S.shift) {
		if(hipMemcpyAsync(&herr, c->d_stage_err, sizeof herr, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return FLACGPU_ERR_LAUNCH;
		if(herr) return FLACGPU_ERR_INPUT;
	}
	return encode_staged(c, nframes, first_frame_number, tail_n, tail_windows, out, out_cap, frame_bytes);
}

// ---- the asynchronous entry -------------------------------------------------------------------------------------------------
// The frames of a batch, device -> page-locked host memory, by a small grid of its own stream: the byte total is read on the device
// (no host round trip between the kernels and the read-back), and the copy engines stay with the input -- on this platform
// input and output copies queued as hipMemcpyAsync took turns on one engine (profiles/r03_j_async_trace.txt: 2.4 + 1.45 ms per
// 8192-frame batch, one after the other), whereas a copy kernel and a copy engine run side by side.
__global__ __launch_bounds__(256) void payload_copy_kernel(const uint8_t *__restrict__ src, const uint64_t *__restrict__ d_total, uint8_t *__restrict__ dst, uint64_t cap)
{
	uint64_t total = *d_total;
	if(total > cap) total = 0;                                          // (reported by flacgpu_collect; nothing is written)
	const uint64_t nvec = total / 16;
	const uint4 *s4 = (const uint4 *)src;
	uint4 *d4 = (uint4 *)dst;
	for(uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (uint64_t)gridDim.x * blockDim.x) d4[i] = s4[i];
	if(blockIdx.x == 0) for(uint64_t i = nvec * 16 + threadIdx.x; i < total; i += blockDim.x) dst[i] = src[i];
}
static int ensure_verify(flacgpu_ctx *c);
static int async_prepare(flacgpu_ctx *c)
{
	if(c->async_ready) return FLACGPU_OK;
	const size_t B = c->cfg.max_batch_frames;
	bool ok = hipStreamCreateWithFlags(&c->s_in, hipStreamNonBlocking) == hipSuccess;
	ok = ok && hipStreamCreateWithFlags(&c->s_small, hipStreamNonBlocking) == hipSuccess;
	ok = ok && hipStreamCreateWithFlags(&c->s_pay, hipStreamNonBlocking) == hipSuccess;
	for(int i = 0; i < FLACGPU_ASYNC_SLOTS && ok; i++) {
		flacgpu_ctx::AsyncSlot &a = c->as[i];
		ok = ok && hipMalloc(&a.d_fb, B * sizeof(uint32_t)) == hipSuccess;
		ok = ok && hipMalloc(&a.d_total, sizeof(uint64_t)) == hipSuccess;
		ok = ok && hipMalloc(&a.d_err, sizeof(uint32_t)) == hipSuccess;
		ok = ok && hipMalloc(&a.d_vres, sizeof(flacgpu_verify_result)) == hipSuccess;
		ok = ok && hipHostMalloc((void **)&a.h, sizeof *a.h, hipHostMallocDefault) == hipSuccess;
		ok = ok && hipHostMalloc((void **)&a.h_fb, B * sizeof(uint32_t), hipHostMallocDefault) == hipSuccess;
		ok = ok && hipEventCreateWithFlags(&a.ev_in, hipEventDisableTiming) == hipSuccess;
		ok = ok && hipEventCreateWithFlags(&a.ev_done, hipEventDisableTiming) == hipSuccess;
		ok = ok && hipEventCreateWithFlags(&a.ev_small, hipEventDisableTiming) == hipSuccess;
		ok = ok && hipEventCreateWithFlags(&a.ev_pay, hipEventDisableTiming) == hipSuccess;
	}
	if(!ok) {
		// all or nothing: a later call starts from null fields again instead of creating streams and buffers over the ones of this attempt
		for(int i = 0; i < FLACGPU_ASYNC_SLOTS; i++) {
			flacgpu_ctx::AsyncSlot &a = c->as[i];
			if(a.d_fb) (void)hipFree(a.d_fb);
			if(a.d_total) (void)hipFree(a.d_total);
			if(a.d_err) (void)hipFree(a.d_err);
			if(a.d_vres) (void)hipFree(a.d_vres);
			if(a.h) (void)hipHostFree(a.h);
			if(a.h_fb) (void)hipHostFree(a.h_fb);
			if(a.ev_in) (void)hipEventDestroy(a.ev_in);
			if(a.ev_done) (void)hipEventDestroy(a.ev_done);
			if(a.ev_small) (void)hipEventDestroy(a.ev_small);
			if(a.ev_pay) (void)hipEventDestroy(a.ev_pay);
			a.d_fb = nullptr; a.d_total = nullptr; a.d_err = nullptr; a.d_vres = nullptr; a.h = nullptr; a.h_fb = nullptr;
			a.ev_in = a.ev_done = a.ev_small = a.ev_pay = nullptr;
		}
		if(c->s_in) (void)hipStreamDestroy(c->s_in);
		if(c->s_small) (void)hipStreamDestroy(c->s_small);
		if(c->s_pay) (void)hipStreamDestroy(c->s_pay);
		c->s_in = c->s_small = c->s_pay = nullptr;
		(void)hipGetLastError();
		return FLACGPU_ERR_ALLOC;
	}
	c->async_ready = true;
	return FLACGPU_OK;
}
extern "C" int flacgpu_in_flight(const flacgpu_ctx *c) { return c ? (int)(c->sub_seq - c->col_seq) : 0; }

// a submission that fails half way: what it has already enqueued (copies from the caller's `raw`, kernels, the copy kernel into the
// caller's `out`) is waited for before the error goes back, so that "an error: nothing of this batch is in flight" holds
static int submit_drain(flacgpu_ctx *c, int code)
{
	if(c->s_in) (void)hipStreamSynchronize(c->s_in);
	(void)hipStreamSynchronize(c->stream);
	if(c->s_small) (void)hipStreamSynchronize(c->s_small);
	if(c->s_pay) (void)hipStreamSynchronize(c->s_pay);
	(void)hipGetLastError();
	return code;
}
extern "C" int flacgpu_submit_batch_raw(flacgpu_ctx *c, const void *raw, const flacgpu_raw_format *fmt, uint32_t nframes,
                                        uint64_t first_frame_number, uint32_t last_block_samples, const float *tail_windows,
                                        uint8_t *out, size_t out_cap, uint32_t *frame_bytes)
{
	if(!c || !raw || !out || !frame_bytes || nframes == 0 || nframes > c->cfg.max_batch_frames) return FLACGPU_ERR_BAD_ARG;
	if(c->sub_seq - c->col_seq >= FLACGPU_ASYNC_SLOTS) return FLACGPU_ERR_BUSY;
	StageParams S;
	int r = make_stage_params(c, fmt, &S);
	if(r != FLACGPU_OK) return r;
	if(hipSetDevice(c->device) != hipSuccess) return FLACGPU_ERR_NO_DEVICE;
	TuneScope tune_scope(&c->tune);
	r = async_prepare(c);
	if(r != FLACGPU_OK) return r;
	const DevParams &P = c->P;
	const uint32_t tail_n = last_block_samples < P.blocksize ? last_block_samples : 0;
	const size_t nsamp = (size_t)(nframes - 1) * P.blocksize + (tail_n ? tail_n : P.blocksize);
	r = ensure_host_staging(c, nframes);                 // (the shared int32 staging buffer; the engine's own d_out is not used here)
	if(r != FLACGPU_OK) return r;
	if(c->verify_on) { r = ensure_verify(c); if(r != FLACGPU_OK) return r; }
	flacgpu_ctx::AsyncSlot &a = c->as[c->sub_seq % FLACGPU_ASYNC_SLOTS];
	const size_t raw_bytes = nsamp * P.channels * S.bytes, out_bytes = (size_t)nframes * P.slot_bytes;
	if(a.d_raw_bytes < raw_bytes) {
		if(a.d_raw) (void)hipFree(a.d_raw);
		a.d_raw = nullptr; a.d_raw_bytes = 0;
		if(hipMalloc(&a.d_raw, raw_bytes) != hipSuccess) return FLACGPU_ERR_ALLOC;
		a.d_raw_bytes = raw_bytes;
	}
	if(a.d_out_bytes < out_bytes) {
		if(a.d_out) (void)hipFree(a.d_out);
		a.d_out = nullptr; a.d_out_bytes = 0;
		if(hipMalloc(&a.d_out, out_bytes) != hipSuccess) return FLACGPU_ERR_ALLOC;
		a.d_out_bytes = out_bytes;
	}
	hipStream_t s = c->stream;
	// input: its own stream, so that it runs beside the kernels of the batches in front
	if(hipMemcpyAsync(a.d_raw, raw, raw_bytes, hipMemcpyHostToDevice, c->s_in) != hipSuccess) return submit_drain(c, FLACGPU_ERR_LAUNCH);
	if(hipEventRecord(a.ev_in, c->s_in) != hipSuccess) return submit_drain(c, FLACGPU_ERR_LAUNCH);
	// kernels: the engine's stream, batch after batch
	if(hipStreamWaitEvent(s, a.ev_in, 0) != hipSuccess) return submit_drain(c, FLACGPU_ERR_LAUNCH);
	if(hipMemsetAsync(a.d_err, 0, sizeof(uint32_t), s) != hipSuccess) return submit_drain(c, FLACGPU_ERR_LAUNCH);
	if(launch_stage_raw(S, a.d_raw, (uint64_t)nsamp * P.channels, c->d_pcm, a.d_err, s) != hipSuccess) return submit_drain(c, FLACGPU_ERR_LAUNCH);
	r = run_batch(c, c->d_pcm, nframes, first_frame_number, tail_n, tail_windows, a.d_out, a.d_out_bytes, a.d_fb, a.d_total, s);
	if(r != FLACGPU_OK) return submit_drain(c, r);
	if(c->verify_on) {
		if(launch_verify(c->P, a.d_out, c->last_fb, c->d_offsets, nframes, tail_n, first_frame_number, c->d_pcm, c->d_vscratch, c->d_vdecoded, c->d_vfinfo, c->d_vstate, c->d_vresult,
		                 c->d_vhints, hints_for(c, a.d_out, nframes, first_frame_number), c->d_vfstat, c->ab.dbg, s) != hipSuccess) return submit_drain(c, FLACGPU_ERR_LAUNCH);
		if(hipMemcpyAsync(a.d_vres, c->d_vresult, sizeof(flacgpu_verify_result), hipMemcpyDeviceToDevice, s) != hipSuccess) return submit_drain(c, FLACGPU_ERR_LAUNCH);
	}
	if(hipEventRecord(a.ev_done, s) != hipSuccess) return submit_drain(c, FLACGPU_ERR_LAUNCH);
	// lengths, total, verdicts: back on a stream of their own (the payload follows from flacgpu_collect, once its size is known)
	if(hipStreamWaitEvent(c->s_small, a.ev_done, 0) != hipSuccess) return submit_drain(c, FLACGPU_ERR_LAUNCH);
	if(hipMemcpyAsync(a.h_fb, a.d_fb, nframes * sizeof(uint32_t), hipMemcpyDeviceToHost, c->s_small) != hipSuccess) return submit_drain(c, FLACGPU_ERR_LAUNCH);
	if(hipMemcpyAsync(&a.h->total, a.d_total, sizeof(uint64_t), hipMemcpyDeviceToHost, c->s_small) != hipSuccess) return submit_drain(c, FLACGPU_ERR_LAUNCH);
	if(hipMemcpyAsync(&a.h->err, a.d_err, sizeof(uint32_t), hipMemcpyDeviceToHost, c->s_small) != hipSuccess) return submit_drain(c, FLACGPU_ERR_LAUNCH);
	if(c->verify_on && hipMemcpyAsync(&a.h->vres, a.d_vres, sizeof(flacgpu_verify_result), hipMemcpyDeviceToHost, c->s_small) != hipSuccess) return submit_drain(c, FLACGPU_ERR_LAUNCH);
	if(hipEventRecord(a.ev_small, c->s_small) != hipSuccess) return submit_drain(c, FLACGPU_ERR_LAUNCH);
	// the frames themselves: by the copy kernel when `out` is page-locked (hipHostMalloc / hipHostRegister) and 16-byte aligned,
	// otherwise by a copy from flacgpu_collect once the total is known on the host
	a.pay_by_kernel = false;
	{
		void *dptr = nullptr;
		const int off = c->tune.no_copy_kernel;
		if(!off && ((uintptr_t)out & 15u) == 0 && hipHostGetDevicePointer(&dptr, out, 0) == hipSuccess && dptr) {
			if(hipStreamWaitEvent(c->s_pay, a.ev_done, 0) != hipSuccess) return submit_drain(c, FLACGPU_ERR_LAUNCH);
			hipLaunchKernelGGL(payload_copy_kernel, dim3(256), dim3(256), 0, c->s_pay, a.d_out, a.d_total, (uint8_t *)dptr, (uint64_t)out_cap);
			if(hipGetLastError() != hipSuccess || hipEventRecord(a.ev_pay, c->s_pay) != hipSuccess) return submit_drain(c, FLACGPU_ERR_LAUNCH);
			a.pay_by_kernel = true;
		}
		else (void)hipGetLastError();
	}
	a.out = out; a.out_cap = out_cap; a.frame_bytes = frame_bytes; a.nframes = nframes; a.check_shift = S.shift != 0;
	c->sub_seq++;
	return FLACGPU_OK;
}

extern "C" int64_t flacgpu_collect(flacgpu_ctx *c)
{
	if(!c || c->sub_seq == c->col_seq) return FLACGPU_ERR_BAD_ARG;
	if(hipSetDevice(c->device) != hipSuccess) return FLACGPU_ERR_NO_DEVICE;
	flacgpu_ctx::AsyncSlot &a = c->as[c->col_seq % FLACGPU_ASYNC_SLOTS];
	c->col_seq++;                                         // (whatever happens below, this batch is no longer in flight)
	// (the payload copy kernel may still be writing the caller's `out` while the small results are already here: whatever this call
	//  returns, it returns only when nothing of the batch touches the caller's memory any more -- ADVICE r03)
	const bool small_ok = hipEventSynchronize(a.ev_small) == hipSuccess;
	const bool pay_ok = !a.pay_by_kernel || hipEventSynchronize(a.ev_pay) == hipSuccess;
	if(!small_ok || !pay_ok) return FLACGPU_ERR_LAUNCH;
	memset(&c->last_verify, 0, sizeof c->last_verify);
	if(c->verify_on) c->last_verify = a.h->vres;
	if(a.check_shift && a.h->err) return FLACGPU_ERR_INPUT;
	memcpy(a.frame_bytes, a.h_fb, a.nframes * sizeof(uint32_t));
	for(uint32_t i = 0; i < a.nframes; i++) if(a.h_fb[i] == 0xffffffffu) return FLACGPU_ERR_LAUNCH;
	const uint64_t total = a.h->total;
	if(total > a.out_cap) return FLACGPU_ERR_OUTPUT_TOO_SMALL;
	if(!a.pay_by_kernel) {
		if(hipMemcpyAsync(a.out, a.d_out, total, hipMemcpyDeviceToHost, c->s_pay) != hipSuccess) return FLACGPU_ERR_LAUNCH;
		if(hipStreamSynchronize(c->s_pay) != hipSuccess) return FLACGPU_ERR_LAUNCH;
	}
	return (int64_t)total;
}

// buffers of the self check, allocated on first use
static int ensure_verify(flacgpu_ctx *c)
{
	if(c->d_vstate) return FLACGPU_OK;
	const size_t B = c->cfg.max_batch_frames;
	bool ok = hipMalloc(&c->d_vstate, sizeof(VerifyState)) == hipSuccess;
	ok = ok && hipMalloc(&c->d_vresult, sizeof(flacgpu_verify_result)) == hipSuccess;
	ok = ok && hipMalloc(&c->d_voffsets, (B + 1) * sizeof(uint64_t)) == hipSuccess;
	ok = ok && hipMalloc(&c->d_vtotal, sizeof(uint64_t)) == hipSuccess;
	ok = ok && hipMalloc(&c->d_vscratch, (size_t)c->P.channels * c->P.blocksize * sizeof(int64_t)) == hipSuccess;
	ok = ok && hipMalloc(&c->d_vdecoded, verify_decoded_bytes(c->P, (uint32_t)B)) == hipSuccess;
	ok = ok && hipMalloc(&c->d_vfinfo, B * sizeof(uint32_t)) == hipSuccess;
	ok = ok && hipMalloc(&c->d_vfstat, B * sizeof(uint32_t)) == hipSuccess;
	{
		const char *e = getenv("FLACGPU_VERIFY_FORCE_HINTS");
		c->force_hints = e ? (atoi(e) ? 1 : 0) : -1;
		// blocks of up to 4096 samples whose frame image and signal fit a workgroup's LDS share are verified a run per thread
		if(c->force_hints != 0 && verify_hinted_covers(c->P))
			ok = ok && hipMalloc(&c->d_vhints, B * c->P.channels * HINT_RUNS * sizeof(uint32_t)) == hipSuccess;
	}
	c->hint_count = 0;
	if(!ok) {
		// all or nothing: d_vstate is the "allocated" mark, and a later call must not find it set next to buffers that are not there
		void **bufs[] = {(void **)&c->d_vstate, (void **)&c->d_vresult, (void **)&c->d_voffsets, (void **)&c->d_vtotal, (void **)&c->d_vscratch, (void **)&c->d_vdecoded,
		                 (void **)&c->d_vfinfo, (void **)&c->d_vfstat, (void **)&c->d_vhints};
		for(void **b : bufs) { if(*b) (void)hipFree(*b); *b = nullptr; }
		(void)hipGetLastError();
	}
	return ok ? FLACGPU_OK : FLACGPU_ERR_ALLOC;
}

extern "C" int flacgpu_verify_batch_device(flacgpu_ctx *c, const uint8_t *d_frames, const uint32_t *d_frame_bytes, uint32_t nframes,
                                           uint64_t first_frame_number, uint32_t last_block_samples, const int32_t *d_pcm,
                                           flacgpu_verify_result *d_result, void *stream)
{
	if(!c || !d_frames || !d_frame_bytes || !d_pcm || !d_result || nframes == 0 || nframes > c->cfg.max_batch_frames) return FLACGPU_ERR_BAD_ARG;
	if(hipSetDevice(c->device) != hipSuccess) return FLACGPU_ERR_NO_DEVICE;
	const int r = ensure_verify(c);
	if(r != FLACGPU_OK) return r;
	TuneScope tune_scope(&c->tune);
	hipStream_t s = stream ? (hipStream_t)stream : c->stream;
	const uint32_t tail_n = last_block_samples < c->P.blocksize ? last_block_samples : 0;
	if(launch_scan(d_frame_bytes, nframes, c->d_voffsets, c->d_vtotal, s) != hipSuccess) return FLACGPU_ERR_LAUNCH;
	if(launch_verify(c->P, d_frames, d_frame_bytes, c->d_voffsets, nframes, tail_n, first_frame_number, d_pcm, c->d_vscratch, c->d_vdecoded, c->d_vfinfo, c->d_vstate, d_result,
	                 c->d_vhints, hints_for(c, d_frames, nframes, first_frame_number), c->d_vfstat, c->ab.dbg, s) != hipSuccess)
		return FLACGPU_ERR_LAUNCH;
	return FLACGPU_OK;
}

// frames of the last verify call that the thread-per-run pass vouched for (the others were decoded sequentially); synchronises
extern "C" int flacgpu_debug_verify_hinted_frames(flacgpu_ctx *c, uint32_t *out)
{
	if(!c || !out || !c->d_vstate) return FLACGPU_ERR_BAD_ARG;
	if(hipSetDevice(c->device) != hipSuccess) return FLACGPU_ERR_NO_DEVICE;
	VerifyState h;
	if(hipDeviceSynchronize() != hipSuccess || hipMemcpy(&h, c->d_vstate, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return FLACGPU_ERR_LAUNCH;
	*out = h.hinted_ok;
	return FLACGPU_OK;
}

extern "C" int flacgpu_set_verify(flacgpu_ctx *c, uint32_t on)
{
	if(!c) return FLACGPU_ERR_BAD_ARG;
	if(on) {
		if(hipSetDevice(c->device) != hipSuccess) return FLACGPU_ERR_NO_DEVICE;
		const int r = ensure_verify(c);
		if(r != FLACGPU_OK) return r;
	}
	c->verify_on = on ? 1 : 0;
	memset(&c->last_verify, 0, sizeof c->last_verify);
	return FLACGPU_OK;
}
extern "C" int flacgpu_last_verify_result(flacgpu_ctx *c, flacgpu_verify_result *out)
{
	if(!c || !out) return FLACGPU_ERR_BAD_ARG;
	*out = c->last_verify;
	return FLACGPU_OK;
}

extern "C" int flacgpu_last_batch_info(flacgpu_ctx *c, uint32_t nframes, flacgpu_subframe_info *sub, uint8_t *channel_assignment)
{
	if(!c || nframes == 0 || nframes > c->last_nframes) return FLACGPU_ERR_BAD_ARG;
	if(hipSetDevice(c->device) != hipSuccess) return FLACGPU_ERR_NO_DEVICE;
	FrameInfo *h = (FrameInfo *)malloc(sizeof(FrameInfo) * nframes);
	if(!h) return FLACGPU_ERR_ALLOC;
	if(hipMemcpy(h, c->d_info, sizeof(FrameInfo) * nframes, hipMemcpyDeviceToHost) != hipSuccess) { free(h); return FLACGPU_ERR_LAUNCH; }
	for(uint32_t f = 0; f < nframes; f++) {
		if(sub) for(uint32_t ch = 0; ch < c->P.channels; ch++) sub[(size_t)f * c->P.channels + ch] = h[f].sub[ch];
		if(channel_assignment) channel_assignment[f] = h[f].channel_assignment;
	}
	free(h);
	return FLACGPU_OK;
}

// which kernels the last batch launched (K_* of flacgpu_dev.h; flacgpu_kernel_bit_name names a bit) -- what the tests that pin a
// kernel selection assert on
extern "C" int flacgpu_last_batch_kernels(const flacgpu_ctx *c, uint32_t *mask)
{
	if(!c || !mask) return FLACGPU_ERR_BAD_ARG;
	*mask = c->last_launched;
	return FLACGPU_OK;
}
extern "C" const char *flacgpu_kernel_bit_name(uint32_t bit)
{
	static const char *const names[] = {"ff_kernel", "prep3_kernel", "prep2_kernel", "prep_kernel", "autoc3_kernel", "autoc2_kernel", "autoc_kernel", "model_kernel",
		"evalg_kernel", "evalw_kernel", "eval_list_kernel", "eval_kernel", "pack_plan_kernel", "pack2_kernel", "pack_kernel", "fo_place_kernel", "scan_kernel", "compact_kernel",
		"append_tail_kernel", "pack2_kernel<run18>", "autoc3_kernel<SETS>", "autoc3_kernel<PLANES>", "fused_output", "prep2_kernel<DECIDE>", "prep4_kernel", "autoc3_kernel<IND>", "autoc4_kernel"};
	return bit < sizeof names / sizeof names[0] ? names[bit] : nullptr;
}
// frames that gave up waiting in the fused output (PackOut, flacgpu_kernels.hip) and were placed from their slots by fo_place_kernel:
// all of them since the context was created, and the most of any one batch.  Zero in a healthy run; a non-zero count is time (up
// to spin_limit polls per frame), never bytes.  Synchronises the device.
extern "C" int flacgpu_fused_fallbacks(flacgpu_ctx *c, uint32_t *total, uint32_t *max_in_a_batch)
{
	if(!c) return FLACGPU_ERR_BAD_ARG;
	if(hipSetDevice(c->device) != hipSuccess) return FLACGPU_ERR_NO_DEVICE;
	uint32_t h[4] = {0, 0, 0, 0};
	if(hipDeviceSynchronize() != hipSuccess || hipMemcpy(h, c->d_fo_nfall, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return FLACGPU_ERR_LAUNCH;
	if(total) *total = h[2];
	if(max_in_a_batch) *max_in_a_batch = h[3];
	return FLACGPU_OK;
}

extern "C" int flacgpu_last_batch_kernel_ms(flacgpu_ctx *c, float *analyze_ms, float *pack_ms, float *compact_ms)
{
	if(!c || !c->timing_valid) return FLACGPU_ERR_BAD_ARG;
	if(hipSetDevice(c->device) != hipSuccess) return FLACGPU_ERR_NO_DEVICE;
	float ms[6];
	const int r = flacgpu_batch_phase_ms(c, 0, ms);
	if(r != FLACGPU_OK) return r;
	const float a = ms[0] + ms[1] + ms[2] + ms[3], p = ms[4], k = ms[5];
	if(analyze_ms) *analyze_ms = a;
	if(pack_ms) *pack_ms = p;
	if(compact_ms) *compact_ms = k;
	return FLACGPU_OK;
}
