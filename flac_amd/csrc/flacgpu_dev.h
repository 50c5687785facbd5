// flac_amd/csrc/flacgpu_dev.h -- structures shared by the kernels and the C-ABI host code.
#ifndef FLACGPU_DEV_H
#define FLACGPU_DEV_H
#include <stdint.h>
#include <stddef.h>
#include <hip/hip_runtime.h>
#include <atomic>
#include <mutex>
#include "flacgpu.h"

namespace flacgpu {

constexpr int TPB = 256;        // threads per workgroup (4 wavefronts of 64)
constexpr int CHUNK = 16;       // consecutive samples owned by one thread in FIR passes
constexpr int MAX_ORDER = 32;   // taps kept per candidate (FLAC__MAX_LPC_ORDER, format.h)
constexpr int AUTOC_STRIDE = 40;// doubles per autocorrelation record (lags 0..32)
constexpr int MAX_PO = 8;       // max residual partition order (FLAC subset limit)
constexpr int MAX_JOBS = 1024;  // windowed-data jobs per subframe (one subdivide_tukey(32) makes 528)
constexpr int MAX_ANALYSES = 2048; // LPC analyses per subframe (one subdivide_tukey(32): 1053)

// flattened, device-friendly copy of flacgpu_config
struct DevParams {
	uint32_t channels, bps, sample_rate, blocksize;
	uint32_t ms_mode;          // 0 none, 1 full mid/side search, 2 loose
	uint32_t ncand;            // candidate channels analysed per frame
	uint32_t max_lpc_order, precision, min_po, max_po, rice_limit;
	uint32_t num_apod;
	uint32_t apod_kind[FLACGPU_MAX_APODIZATIONS], apod_parts[FLACGPU_MAX_APODIZATIONS];
	uint32_t autoc_variant;    // 8 / 12 / 16: which compiled reference routine to follow
	uint32_t disable_constant, disable_fixed, disable_verbatim, limit_min_bitrate;
	uint32_t slot_bytes;       // bytes reserved per frame in the slot buffer / LDS frame image
	uint32_t sig_bytes;        // LDS bytes of the padded signal array
	uint32_t max_jobs, max_analyses; // window jobs / LPC analyses per subframe at the nominal blocksize
	// candidate slots of a subframe, in the reference's evaluation order (stream_encoder.c:4155-4266): nfixed fixed
	// orders, then per analysis norders LPC orders x nprec coefficient precisions
	uint32_t exhaustive, prec_search;
	uint32_t nfixed;           // 1 (the guessed order) or 5 (-e: orders 0..4)
	uint32_t norders;          // 1 (the guessed order) or max_lpc_order (-e: orders 1..max)
	uint32_t nprec;            // 1 or 11 (-p: precisions 5..15)
	uint32_t ncslots;          // nfixed + max_analyses * norders * nprec
	// more than 24 bits per sample: only the general kernels run; the side channel of a 32-bit stream has 33 bits
	// (stream_encoder.c:3831-3835) and is kept as 64-bit samples in HBM and LDS
	uint32_t wide_samples;     // bps > 24
	uint32_t chan_stride;      // 32-bit words per planar channel: blocksize, or 2 * blocksize when a 33-bit channel can occur
	uint32_t img_global;       // the worst-case frame does not fit the LDS next to the pack kernel's state: it is assembled in its HBM slot
	uint32_t tune_flags;       // development switches (environment): bit 0 FLACGPU_EVAL_CANDS_GLOBAL: the evaluation kernel reads the
	                           // candidate records from global memory instead of staging them in LDS
	uint32_t stream_sig;       // a block does not fit the LDS (more than 16384 samples, or 16384 64-bit ones): the general prep and
	                           // evaluation kernels read the samples from HBM instead of an LDS copy (sig_bytes = 0)
};

// analysis -> pack hand-off, one per (frame, candidate channel); 16-byte multiple
struct SubDecision {
	uint32_t bits;             // estimated subframe bits (selection metric)
	uint8_t type;              // 0 CONSTANT 1 VERBATIM 2 FIXED 3 LPC
	uint8_t order, wasted, po, rice2, precision;
	int8_t shift;
	uint8_t which;             // signal modelled: 0..C-1 channel, C mid, C+1 side
	int32_t constant;          // CONSTANT: the sample value (low 32 bits)
	int32_t q[MAX_ORDER];
	uint8_t params[1u << MAX_PO];
	int32_t constant_hi;       // bits 32.. of a 33-bit constant (sign extension otherwise)
	uint32_t fmt;              // ChanPrep::fmt of the channel: how its planar copy is stored
	uint32_t pad[2];
};

struct Candidate {
	uint32_t order, precision;
	int32_t shift;
	uint32_t wide;             // 0: 32-bit FIR (lpc.c:321), 1: 64-bit accumulate (lpc.c:582), 2: overflow-checked (lpc.c:832,886)
	int32_t q[MAX_ORDER];
};

// One windowed-data job of a subframe's LPC analysis (apply_apodization_, stream_encoder.c:4318-4392):
// the whole block under window table `apod`, or one partial window of a subdivide_tukey depth.
struct WindowJob {
	uint32_t off;       // float offset of this job's windowed data in the LDS window buffer
	uint32_t nd;        // data_len handed to the autocorrelation
	uint32_t apod;      // which window table
	uint32_t full;      // 1: whole block (lpc.c:68), 0: partial window (lpc.c:82)
	uint32_t part, dshift, i0;
	uint32_t pad;
};
// Host-built schedule for one blocksize: jobs (longest first) and the analyses derived from them in the reference's
// order (full, then per depth: partial [, punch-out]).
struct JobTable {
	uint32_t njobs, nanalyses, wnd_floats, pad;
	WindowJob jobs[MAX_JOBS];
	uint16_t an_job[MAX_ANALYSES], an_root[MAX_ANALYSES];
	uint8_t an_punch[MAX_ANALYSES];
	// job SETS: the whole-block job of an apodization, or all partial windows of one subdivide_tukey depth -- each set covers
	// the block exactly once (flacgpu_autoc.hip runs the sets of a group of subframes side by side); the first 8 are listed
	uint32_t nsets;
	uint16_t set_first[8], set_count[8];
	// the Rice search of the wavefront-per-channel evaluation kernels (flacgpu_evalg.h: rice_pass): for a node of 2^m lane runs of
	// S = n / 64 samples, `o` of them warm-up samples, (0x40000 / (ns)) << 13 with ns = (S << m) - o -- the integer reciprocal of
	// set_partitioned_rice_ (stream_encoder.c:5020), a function of the block size alone: built here, once, instead of 91 integer
	// divisions per channel in the kernels (round 5)
	uint32_t eg_div[7][MAX_ORDER + 1];
};
void build_job_table(const DevParams &P, uint32_t n, JobTable *jt);

// ---- hand-off records between the analysis kernels (flacgpu_analyze.hip) ------------------------------------
enum { PREP_LPC = 1, PREP_FIXED_VALID = 2, PREP_CONSTANT = 4 };
struct ChanPrep {
	uint32_t which;            // signal modelled: 0..C-1 channel, C mid, C+1 side
	uint32_t wasted, sbps, n;  // wasted bits, subframe bps after the shift, samples in this block
	uint32_t flags;            // PREP_*
	uint32_t fixed_order;      // guessed fixed-predictor order
	int32_t constant;          // sample value when PREP_CONSTANT
	uint32_t verbatim_bits;    // size of the VERBATIM baseline (0xffffffff: disabled)
	uint32_t fmt;              // planar channel copy: 1 = 16-bit pairs (every sample fits int16: sbps <= 16, or a quiet side
	                           // channel), 0 = 32-bit samples, 2 = 64-bit samples (sbps 33)
	int32_t constant_hi;       // bits 32.. of a 33-bit constant
	uint32_t handled;          // written 0 by the prep kernels; EVG_HANDLED once evalg_kernel (flacgpu_evalg.hip) has decided the channel
	uint32_t pad;
};
constexpr uint32_t EVG_HANDLED = 0x600Du;
struct AnalyzeBuffers {
	ChanPrep *prep;            // [frames*ncand]
	double *autoc;             // [frames*ncand][max_jobs][AUTOC_STRIDE]
	Candidate *cands;          // [frames*ncand][ncslots]: fixed orders, then analysis a / order / precision (DevParams::ncslots)
	int *valid;                // same shape
	int32_t *chan;             // [frames*ncand][blocksize] planar channel signals, wasted bits shifted out (ChanPrep::fmt)
	uint32_t *left, *left2, *nleft; // two lists [frames*ncand] of channels a wavefront-per-channel evaluation kernel left to the next kernel in line
	                           // (evalg -> evalw -> eval_list_kernel), and their counts nleft[0], nleft[1] (zeroed by the model kernel)
	unsigned long long *dbg;   // FLACGPU_DEBUG_TIMING=1: [frames*ncand][16] s_memtime stamps of the eval kernel (else null)
};
constexpr int FLACGPU_MAX_SUBBATCHES = 8;   // streams a batch may be split over (FLACGPU_SUBBATCHES / flacgpu_set_subbatches)
constexpr int EVAL_MAX_WAVES = 8;   // wavefronts per eval workgroup (one residual candidate each per round)

struct FrameInfo {
	flacgpu_subframe_info sub[FLACGPU_MAX_CHANNELS];
	uint8_t channel_assignment;
	uint8_t pad[3];
};

// ---- development / A-B switches and the record of what a batch launched -----------------------------------------------------
// The switches are read from the environment when a CONTEXT is created (flacgpu_create), not once per process: two engines of one
// process may differ, nothing is a function static shared by threads, and the per-device kernel attributes (dynamic LDS size) are
// set once per device, not once per process.  tune() is the calling thread's current context's copy while one of its
// entry points runs (flacgpu_api.cpp: TuneScope), else a per-thread copy read from the environment.
enum : uint32_t {
	K_FF = 1u << 0, K_PREP3 = 1u << 1, K_PREP2 = 1u << 2, K_PREP = 1u << 3, K_AUTOC3 = 1u << 4, K_AUTOC2 = 1u << 5, K_AUTOC = 1u << 6, K_MODEL = 1u << 7,
	K_EVALG = 1u << 8, K_EVALW = 1u << 9, K_EVAL_LIST = 1u << 10, K_EVAL = 1u << 11, K_PACK_PLAN = 1u << 12, K_PACK2 = 1u << 13, K_PACK = 1u << 14,
	K_FO_PLACE = 1u << 15, K_SCAN = 1u << 16, K_COMPACT = 1u << 17, K_APPEND_TAIL = 1u << 18, K_PACK2_RUN18 = 1u << 19, K_AUTOC3_SETS = 1u << 20,
	K_AUTOC3_PLANES = 1u << 21, K_FUSED_OUTPUT = 1u << 22, K_PREP2_DECIDE = 1u << 23, K_PREP1 = 1u << 24, K_AUTOC1 = 1u << 25 /* autoc3_kernel<IND> */, K_AUTOC4 = 1u << 26
};
struct Tune {
	int autoc3_mode;           // FLACGPU_AUTOC3: 0 never, 1 whenever it applies, 2 (default) when it fills the chip
	int autoc3_sets, autoc3_planes, autoc2_ungrouped;
	int event_fence, copy_results;      // (host side, flacgpu_api.cpp)
	int autoc3_ind_sets;          // FLACGPU_AUTOC3_IND_SETS: independent channels, a wavefront per window-job SET: 0 never, 1 always, 2 by the batch's size
	int autoc2_force;          // FLACGPU_AUTOC2: 0 decide per batch, 1 never, 2 always
	int no_ff, no_run18, no_run18w, no_prep3, no_prep3n, no_prep4, no_prep_decide, no_evalg, no_fast1, no_flat, no_wide_decide, no_evalg32, no_wide_ff;
	int eval_wpc, evalw_wpc, eval_waves, eval_cpw, eval_prefetch /* -1: derive */;
	int sync_debug, no_fused, no_copy_kernel, cands_global;
	int device;                // the context's device: index of the per-device "attributes set" flags
	uint32_t launched;         // K_* of every kernel launched since the batch began
};
Tune &tune();
inline void note_launch(uint32_t k) { tune().launched |= k; }
// The hipFuncSetAttribute calls of a launch function, once per (call site, device) and safe against a second thread with another
// context on the same device (the flag used to be set before the attributes were): `if(AttrOnce once{flags}) { ...set the
// attributes, return on error...; once.ok(); }` -- the first thread runs the block holding the lock, the others wait for it; the flag
// is set by ok() only, so a failed attempt is retried by the next launch.
struct AttrFlags { std::atomic<bool> done[64]; };
std::mutex &attr_mutex();
struct AttrOnce {
	AttrFlags &f; int d; bool first; std::unique_lock<std::mutex> lk;
	explicit AttrOnce(AttrFlags &flags) : f(flags), d(tune().device & 63), first(false)
	{
		if(f.done[d].load(std::memory_order_acquire)) return;
		lk = std::unique_lock<std::mutex>(attr_mutex());
		first = !f.done[d].load(std::memory_order_relaxed);
		if(!first) lk.unlock();
	}
	explicit operator bool() const { return first; }
	void ok() { f.done[d].store(true, std::memory_order_release); }
};

void sync_debug(const char *what, hipStream_t s);
size_t analyze_lds_bytes(const DevParams &P);
size_t pack_lds_bytes(const DevParams &P);
hipError_t launch_analyze(const DevParams &P, const int32_t *pcm, const float *win, const float *tailwin,
                          uint32_t nframes, uint32_t tail_n, const JobTable *jt_main, const JobTable *jt_tail, uint32_t nsets_main /* JobTable::nsets of jt_main */, const AnalyzeBuffers &B,
                          SubDecision *dec, hipEvent_t *phase_ev /* [3]: after prep, autoc, model; may be null */, hipStream_t s);
bool autoc2_applicable(const DevParams &P);
// chan: the planar channels of the prep kernel (AnalyzeBuffers::chan), or null: autoc3_kernel reads left / right from them when they are 16-bit pairs
hipError_t launch_autoc2(const DevParams &P, const int32_t *pcm, const int32_t *chan, const float *win, uint32_t nmain, uint32_t njobs, uint32_t nsets, const JobTable *jt,
                         const ChanPrep *preps, double *autoc, hipStream_t s);
// the plain autocorrelation loop of -l 16 and up with a lane per subframe (flacgpu_autoc.hip: autoc4_kernel); chan: AnalyzeBuffers::chan
bool autoc4_applicable(const DevParams &P);
hipError_t launch_autoc4(const DevParams &P, const int32_t *chan, const float *win, uint32_t nmain, uint32_t njobs, const JobTable *jt, const ChanPrep *preps, double *autoc, hipStream_t s);
bool evalg_applicable(const DevParams &P);
hipError_t launch_evalg(const DevParams &P, uint32_t nframes, uint32_t tail_n, const JobTable *jt, const AnalyzeBuffers &B, SubDecision *dec, hipStream_t s);
// 32-bit planar channels (flacgpu_evalw.hip): every channel of the batch (in_list == null), or the channels of a list
hipError_t launch_evalw(const DevParams &P, uint32_t nframes, uint32_t tail_n, const JobTable *jt, const AnalyzeBuffers &B, SubDecision *dec,
                        const uint32_t *in_list, const uint32_t *in_count, uint32_t *out_list, uint32_t *out_count, hipStream_t s);
bool prep2_applicable(const DevParams &P);
bool prep2_decides(const DevParams &P);      // prep2_kernel also evaluates and decides (no LPC search: -0 .. -2)
hipError_t launch_prep2(const DevParams &P, const int32_t *pcm, uint32_t nmain, const AnalyzeBuffers &B, SubDecision *dec, hipStream_t s);
// where the pack kernel / ff_kernel may put the frames directly (fused output: flacgpu_kernels.hip, PackOut); with po == null or
// po->out == null every frame goes to its slot and launch_scan + launch_compact must follow.  fstate / fall [max frames rounded up
// to 64], sstate / sprefix / scount [max frames / 64 rounded up], nfall [2]: zeroed once when they are allocated (the tagged words
// carry the epoch of their batch, the counters are zeroed by their last user); epoch: 1 .. 2^24 - 1, +1 for every launch;
// spin_limit: polls before a frame gives up waiting for the ones in front of it and takes the slot + fo_place_kernel route
struct PackOutArgs { uint8_t *out; uint64_t cap; uint64_t *offsets; uint64_t *total; uint64_t *fstate, *sstate, *sprefix, *scount; uint32_t *fall, *nfall; uint32_t epoch, spin_limit, lag /* launch_ff */; };
// What pack2_kernel packs from: pack_plan_kernel (one lane per frame) turns the decision records of the two (C) winning candidate
// channels of a frame into records whose fields are ready to use -- every field wave-uniform for the pack workgroup, which takes
// them with scalar loads: no decision logic, no LDS copy of the records, no single-lane header assembly in the pack kernel.
struct PackSub {                       // 192 bytes
	uint32_t di;                       // candidate channel the samples come from
	uint32_t type, order, wasted;      // SubDecision's
	uint32_t sbps, smask;              // sample width of the subframe, (1 << sbps) - 1
	uint32_t fmt16;                    // planar copy holds 16-bit pairs
	int32_t shift;                     // quantisation shift (0 for the fixed predictors)
	uint32_t fmode;                    // FIR arithmetic on 32-bit samples: 0 v_mad_i32_i24, 1 32-bit products, 2 64-bit sums (fir_mode)
	uint32_t po, rice2;
	uint32_t type_byte;                // the subframe header byte (stream_encoder_framing.c:393-594), wasted-bits flag included
	uint32_t constant;                 // CONSTANT: the sample, masked to sbps bits
	uint32_t b_bits;                   // bits of B
	uint32_t bits;                     // SubDecision::bits (flacgpu_subframe_info)
	uint32_t inv_psize;                // ceil(2^32 / partition size): sample index / partition size = mul_hi(index, inv_psize), exact below 2^16
	uint32_t QP[8];                    // taps 2p, 2p+1 as an int16 pair (the packed-sample chains)
	uint32_t B[8];                     // what stands between the warm-up samples and the first partition, MSB first: LPC precision, shift and
	                                   // coefficients (up to 16 of 15 bits), then the Rice method and the partition order
	int32_t q[16];                     // taps (zero from `order` on)
};                                     // (the Rice parameters stay in the decision record: SubDecision::params, a byte per thread)
struct PackHead { uint32_t hw[4]; uint32_t hdr_bytes; uint32_t ca; uint32_t pad[2]; };      // frame header as big-endian words, its length
static_assert(sizeof(PackSub) == 192 && sizeof(PackHead) == 32, "scalar-load friendly records");
inline size_t pack_plan_stride(const DevParams &P) { return sizeof(PackHead) + (size_t)P.channels * sizeof(PackSub); }
// plan: [nframes] records of pack_plan_stride(P) bytes (scratch of the pack kernels)
hipError_t launch_pack(const DevParams &P, const int32_t *chan, uint32_t nframes, uint32_t tail_n, uint64_t first,
                       const SubDecision *dec, uint8_t *plan, uint8_t *slots, uint32_t *fb, FrameInfo *info, unsigned long long *dbg,
                       const PackOutArgs *po, bool *fused_out, uint32_t *hints, uint32_t *hinted_frames, hipStream_t s);
// the one-kernel path of the presets without an LPC search on 16-bit stereo in 1152-sample blocks (flacgpu_kernels.hip: ff_kernel):
// every frame of nominal length of the batch; po as for launch_pack
bool ff_applicable(const DevParams &P);
hipError_t launch_ff(const DevParams &P, const int32_t *pcm, uint32_t nmain, uint64_t first, uint8_t *slots, uint32_t *fb, FrameInfo *info, const PackOutArgs *po, hipStream_t s);
// fused output, short last block: frame f, assembled in its slot by the general kernels, goes behind the frames of nominal length
hipError_t launch_append_tail(const uint8_t *slot, const uint32_t *fb, uint32_t f, const PackOutArgs *po, hipStream_t s);
// hints (null: none wanted): [frame][channel][HINT_RUNS] bit offset, from the frame's first byte, at which the codes of each
// 16-sample run of a residual-coded subframe start (the partition's parameter field when the run opens a partition) -- what the
// hinted verify pass decodes from (flacgpu_decode_hinted.h).  *hinted_frames: the leading frames that got them.
constexpr uint32_t HINT_RUNS = 256;
// raw sample bytes -> interleaved int32 (flacgpu_stage.hip)
struct StageParams {
	uint32_t bytes;          // container bytes per sample: 1, 2, 3, 4
	uint32_t big_endian, is_unsigned, shift, channels, use_map;
	uint8_t map[FLACGPU_MAX_CHANNELS];   // input channel c goes to output channel map[c]
};
hipError_t launch_stage_raw(const StageParams &S, const void *d_raw, uint64_t nvalues, int32_t *d_pcm, uint32_t *d_err, hipStream_t s);
// the self check (flacgpu_verify.hip, crc_check_kernel in flacgpu_kernels.hip)
struct VerifyState { uint32_t first_bad; uint32_t hinted_ok; uint32_t pad[2]; };      // index of the first frame of the batch that failed (0xffffffff: none); frames the hinted pass verified
hipError_t launch_crc_check(const uint8_t *frames, const uint32_t *fb, const uint64_t *offsets, uint32_t nframes, VerifyState *state, hipStream_t s, uint8_t *per_frame_bad = nullptr);
bool verify_hinted_covers(const DevParams &P);        // this configuration's frames can go through the thread-per-run verify pass
size_t verify_decoded_bytes(const DevParams &P, uint32_t max_frames);     // the lane-interleaved buffer of decoded coded-channel samples
// hints / nhinted: the pack kernel's run starts for the first nhinted frames (null / 0: none) -- those frames go through the
// thread-per-run pass first and only the ones it cannot vouch for are decoded sequentially; fstat: [nframes] scratch
hipError_t launch_verify(const DevParams &P, const uint8_t *frames, const uint32_t *fb, const uint64_t *offsets, uint32_t nframes, uint32_t tail_n,
                         uint64_t first, const int32_t *pcm, int64_t *scratch, void *decoded, uint32_t *finfo, VerifyState *state, flacgpu_verify_result *result,
                         const uint32_t *hints, uint32_t nhinted, uint32_t *fstat, unsigned long long *dbg, hipStream_t s);
hipError_t launch_scan(const uint32_t *fb, uint32_t nframes, uint64_t *offsets, uint64_t *total, hipStream_t s);
hipError_t launch_compact(const uint8_t *slots, uint32_t slot_bytes, const uint32_t *fb, const uint64_t *offsets,
                          uint8_t *out, uint64_t out_cap, uint32_t nframes, hipStream_t s);

} // namespace flacgpu
#endif
