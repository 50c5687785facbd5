// flac_amd/csrc/flacgpu_verify.hip -- the encoder's self check on the device (SURVEY.md 8f row 3;
// FLAC__stream_encoder_set_verify, src/libFLAC/stream_encoder.c:3000-3018, 5155-5230): every frame of a batch is decoded
// again and compared with the samples that went in, without leaving HBM.
//
//   verify_kernel    ONE LANE PER FRAME (flacgpu_decode.h): a wavefront walks 64 frames in lockstep through header, subframes,
//                    Rice codes and predictor restoration.  Rice decoding is a serial bit-dependency chain, so a lane runs at
//                    instruction-issue latency; what is parallel is the batch: 16384 frames = 256 wavefronts, one per CU.  A lane
//                    therefore does nothing that waits on memory besides its own bit window (fetched one word ahead): the decoded
//                    CODED-channel samples are stored lane-interleaved (sample i of the 64 frames of a wavefront = one 256-byte row,
//                    fire-and-forget stores), not compared here -- 64 lanes reading 64 inputs 32 KiB apart would alias one cache set.
//   compare_kernel   thread-parallel: rows of decoded samples through an LDS transpose against the value the input PCM implies
//                    for that coded channel (left, right, (L+R)>>1, L-R: the decorrelation is a bijection, flacgpu_decode.h).
//   (crc_check_kernel, flacgpu_kernels.hip: the CRC-16 footers, one wavefront per frame, spans in parallel.)
//   verify_detail_kernel  one lane, only when a frame failed: decodes the FIRST bad frame of the batch into a scratch buffer,
//                    undoes the inter-channel decorrelation and reports {frame, channel, sample, expected, got} of the first
//                    differing output sample, or "does not decode" -- what get_verify_decoder_error_stats returns.
// Integer / bit work, latency bound by construction; no MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "flacgpu.h"
#include "flacgpu_dev.h"
#define FLACGPU_HD __device__
// 24-bit multiply-adds as ONE asm statement per 4 taps (the compiler turns the __mul24 builtin back into a quarter-rate 32-bit
// multiply when it cannot prove the operands' range, and pads separate asm statements with s_nop)
namespace flacgpu {
template <int M>
__device__ __forceinline__ uint32_t dot24_asm(const int32_t (&a)[M], const int32_t (&b)[M])
{
	static_assert(M % 4 == 0, "taps in fours");
	uint32_t d = 0;
#pragma unroll
	for(int j = 0; j < M; j += 4)
		asm("v_mad_i32_i24 %0, %1, %2, %0\n\tv_mad_i32_i24 %0, %3, %4, %0\n\tv_mad_i32_i24 %0, %5, %6, %0\n\tv_mad_i32_i24 %0, %7, %8, %0"
		    : "+v"(d) : "v"(a[j]), "v"(b[j]), "v"(a[j + 1]), "v"(b[j + 1]), "v"(a[j + 2]), "v"(b[j + 2]), "v"(a[j + 3]), "v"(b[j + 3]));
	return d;
}
}
#define FLACGPU_DOT24(a, b) flacgpu::dot24_asm(a, b)
#include "flacgpu_decode.h"
#include "flacgpu_decode_hinted.h"

namespace flacgpu {
static_assert(HINT_RUNS == HINT_MAX_RUNS && HINT_RUN == CHUNK, "the pack kernel's runs are the hinted pass's runs");

// ---- the hinted pass: a workgroup per frame, a thread per 16-sample run (flacgpu_decode_hinted.h has the reasoning and every
// decision; this kernel is its steps with the per-run work spread over the threads).  fstat[f] = 0: the frame is verified;
// 1: it goes to the sequential decoder below.
struct HintedShared { uint32_t hs[HINT_RUNS + 1]; uint32_t ends[HINT_RUNS]; int32_t q[HINT_MAX_ORDER]; uint32_t head[4]; };
// the coded channels the input implies are staged in LDS: all of them in one pass over the PCM when they fit next to four other
// workgroups (stereo does), else channel by channel
__host__ __device__ inline bool hinted_stage_all(const DevParams &P) { return (size_t)P.channels * (16 + P.blocksize) * 4 <= 36 * 1024; }
static size_t hinted_lds_bytes(const DevParams &P) { return (size_t)(hinted_stage_all(P) ? P.channels : 1) * (16 + P.blocksize) * 4 + sizeof(HintedShared); }

template <int MAXORD>
__global__ __launch_bounds__(TPB) void verify_hinted_kernel(const DevParams P, const uint8_t *__restrict__ frames, const uint32_t *__restrict__ frame_bytes,
                                                            const uint64_t *__restrict__ offsets, uint32_t nframes, uint32_t nhinted, uint64_t first_frame_number,
                                                            const int32_t *__restrict__ pcm, const uint32_t *__restrict__ hints, uint32_t *__restrict__ fstat,
                                                            VerifyState *__restrict__ state, unsigned long long *__restrict__ dbg, uint32_t prefetch_ahead)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const uint32_t f = blockIdx.x, tid = threadIdx.x;
	// (The workgroup's first act is to wait for its 32 KB of input from HBM, a quarter of its life: it also touches the input of the
	// workgroup that will follow it on this CU -- one discarded load per 128-byte line -- so that that one's wait ends in the L2 / MALL.)
#define VSTAMP(k) do { if(dbg && tid == 0) dbg[(size_t)blockIdx.x * 16 + (k)] = (unsigned long long)clock64(); } while(0)
	VSTAMP(0);
	const uint32_t C = P.channels, N = P.blocksize;
	const bool stage_all = hinted_stage_all(P);
	const uint32_t fb = f < nhinted ? frame_bytes[f] : 0xffffffffu;
	if(fb == 0xffffffffu || fb < 6 || fb > P.slot_bytes) { if(tid == 0) fstat[f] = 1; return; }        // (the same for every thread)
	VSTAMP(13);
	int32_t *ybase = (int32_t *)smem;                               // [channels or 1][16 + N]: the value the input implies, NOT yet shifted by the wasted bits
	HintedShared *sh = (HintedShared *)(ybase + (size_t)(stage_all ? C : 1) * (16 + N));
	// The frame is read where it lies, as aligned words of global memory: it was written a moment ago and sits in the L2.  (An LDS
	// copy behind the generic pointers of BitReader / PeekSrc does not compile with this toolchain: the local-to-generic cast's null
	// check comes out as an instruction the assembler rejects.)
	const uint8_t *p = frames + offsets[f], *hi = frames + offsets[nframes];
	const uint32_t mis = (uint32_t)((uintptr_t)p & 3);
	const uint32_t *w0g = (const uint32_t *)(p - mis);
	const uint32_t nw = (mis + fb + 3) / 4, avail = (uint32_t)(((hi - 1) - (p - mis)) / 4) + 1;        // words up to the one holding the buffer's last byte
	for(uint32_t t = tid; t < 16 * (stage_all ? C : 1); t += TPB) ybase[(t >> 4) * (16 + N) + (t & 15)] = 0;
	PeekSrc S;
	S.w0 = w0g; S.nwords = nw + 2 < avail ? nw + 2 : avail; S.skip = mis * 8; S.limit = (fb - 2) * 8;
	DecodeExpect E;
	E.channels = C; E.bps = P.bps; E.blocksize = N; E.n = N; E.frame_number = first_frame_number + f;
	// Order of the loads: a wait for one load is a wait for every load issued before it.  The five words of the frame header go
	// first, then the thread's share of the frame's input (stereo: 16 independent loads), then the touch of the successor's input --
	// so that the header is parsed while the input is still on its way.
	const HeadWords HW = hinted_head_words(S);
	constexpr int PRE = (int)(HINT_MAX_RUNS * HINT_RUN / TPB);
	int2 lr[PRE];
	const bool pre = stage_all && C == 2;
	if(pre) {
#pragma unroll
		for(int k = 0; k < PRE; k++) {
			// (the index is clamped, not the load made conditional: a load under a condition is followed by a wait for it, sixteen
			//  times over -- that alone was 15 k of the workgroup's 90 k ticks)
			const uint32_t i = tid + (uint32_t)k * TPB;
			lr[k] = *(const int2 *)(pcm + ((size_t)f * N + (i < N ? i : N - 1)) * 2);
		}
	}
	uint32_t pf = 0;
	if(prefetch_ahead && f + prefetch_ahead < nhinted) {
		const unsigned char *a = (const unsigned char *)(pcm + (size_t)(f + prefetch_ahead) * P.blocksize * P.channels);
		const uint32_t lines = (P.blocksize * P.channels * 4 + 127) / 128;
		for(uint32_t l = tid; l < lines; l += TPB) { const unsigned char *q = a + (size_t)l * 128; asm volatile("global_load_dword %0, %1, off" : "+v"(pf) : "v"(q) : "memory"); }
	}
	// the frame header: the first wavefront reads it for all
	VSTAMP(14);
	if(tid < 64) {
		FrameHead FH0;
		FH0.ca = 0; FH0.n = 0;
		uint32_t hpos = 0;
		const int st = hinted_frame_header_w(HW, S, E, FH0, &hpos);
		if(tid == 0) { sh->head[0] = (uint32_t)st; sh->head[1] = FH0.ca; sh->head[2] = FH0.n; sh->head[3] = hpos; }
	}
	VSTAMP(15);
	__syncthreads();
	if(sh->head[0] != (uint32_t)DEC_OK) { if(tid == 0) fstat[f] = 1; asm volatile("s_waitcnt vmcnt(0)" :: "v"(pf)); return; }
	FrameHead FH;
	FH.ca = sh->head[1]; FH.n = sh->head[2];
	uint32_t pos = sh->head[3];
	const uint32_t n = FH.n, nruns = n / HINT_RUN;
	uint32_t suspect = 0;
	bool bail = false;
	VSTAMP(1);
	if(stage_all) {
		// one pass over the frame's input for all its coded channels
		if(pre) {
#pragma unroll
			for(int k = 0; k < PRE; k++) {
				const uint32_t i = tid + (uint32_t)k * TPB;
				if(i < n) {
					const int32_t xx[2] = {lr[k].x, lr[k].y};
#pragma unroll
					for(uint32_t ch = 0; ch < 2; ch++) {
						const int64_t v = coded_expectation(xx, FH.ca, ch);
						suspect |= (uint32_t)(v != (int64_t)(int32_t)v);
						ybase[ch * (16 + N) + 16 + i] = (int32_t)v;
					}
				}
			}
		}
		else for(uint32_t i = tid; i < n; i += TPB) {
			const int32_t *x = pcm + ((size_t)f * N + i) * C;
			for(uint32_t ch = 0; ch < C; ch++) {
				const int64_t v = coded_expectation(x, FH.ca, ch);
				suspect |= (uint32_t)(v != (int64_t)(int32_t)v);
				ybase[ch * (16 + N) + 16 + i] = (int32_t)v;
			}
		}
	}
	for(uint32_t ch = 0; ch < C; ch++) {
		const HintedSub H = hinted_subframe_head(S, pos, coded_bps(E.bps, FH.ca, ch), n);         // (every thread: the same answer)
		if(!H.ok || H.order > (uint32_t)MAXORD) { bail = true; break; }
		int32_t *y = ybase + (stage_all ? (size_t)ch * (16 + N) : 0);
		if(ch < 2) VSTAMP(2 + 5 * ch);
		if(!stage_all)
			for(uint32_t i = tid; i < n; i += TPB) {
				const int64_t v = coded_expectation(pcm + ((size_t)f * N + i) * C, FH.ca, ch);
				suspect |= (uint32_t)(v != (int64_t)(int32_t)v);
				y[16 + i] = (int32_t)v;
			}
		__syncthreads();
		if(ch < 2) VSTAMP(3 + 5 * ch);
		// the subframe's own signal is that value shifted down by the wasted bits, which must be zero in it (in place: a thread
		// shifts the samples it staged)
		if(H.wasted) {
			for(uint32_t i = tid; i < n; i += TPB) {
				const int32_t v = y[16 + i];
				suspect |= (uint32_t)((v & (int32_t)((1u << H.wasted) - 1u)) != 0);
				y[16 + i] = v >> H.wasted;
			}
		}
		if(H.type >= 2) {
			if(tid < nruns) sh->hs[tid] = hints[((size_t)f * C + ch) * HINT_RUNS + tid];
			if(tid < HINT_MAX_ORDER) sh->q[tid] = tid >= H.order ? 0 : H.type == 3 ? peek_signed(S, H.pos_q + tid * H.prec, H.prec) : hinted_fixed_tap(H.order, tid);
		}
		__syncthreads();
		if(H.type == 0) {
			const int32_t v = peek_signed(S, H.pos_body, H.sb);
			for(uint32_t i = tid; i < n; i += TPB) suspect |= (uint32_t)(y[16 + i] != v);
			pos = H.end_fixed;
		}
		else if(H.type == 1) {
			for(uint32_t i = tid; i < n; i += TPB) suspect |= (uint32_t)(peek_signed(S, H.pos_body + i * H.sb, H.sb) != y[16 + i]);
			pos = H.end_fixed;
		}
		else {
			if(ch < 2) VSTAMP(4 + 5 * ch);
			if(tid < H.order) suspect |= (uint32_t)(peek_signed(S, H.pos_body + tid * H.sb, H.sb) != y[16 + tid]);
			if(tid == 0) suspect |= (uint32_t)(sh->hs[0] != H.r0);
			if(tid < nruns) {
				const uint32_t t = tid, part = (t * HINT_RUN) / H.psize, t0 = part * H.psize / HINT_RUN;
				const uint32_t kpos = sh->hs[t0], mystart = sh->hs[t];
				uint32_t e = 0xffffffffu;
				if(kpos + H.plen > S.limit || mystart > S.limit) suspect = 1;
				else {
					const uint32_t k = peek_bits(S, kpos, H.plen);
					if(k == H.esc) suspect = 1;                     // raw partitions: the sequential decoder's business
					else {
						int32_t yw[32], q[MAXORD];
#pragma unroll
						for(int u = 0; u < 32; u++) yw[u] = y[t * HINT_RUN + (uint32_t)u];
#pragma unroll
						for(int j = 0; j < MAXORD; j++) q[j] = sh->q[j];
						suspect |= hinted_run<MAXORD, int32_t>(S, mystart + (t == t0 ? H.plen : 0), k, t == 0 ? H.order : 0, yw, q, H, &e);
					}
				}
				sh->ends[t] = e;
			}
			if(ch < 2) VSTAMP(5 + 5 * ch);
			__syncthreads();
			if(ch < 2) VSTAMP(6 + 5 * ch);
			if(tid + 1 < nruns) suspect |= (uint32_t)(sh->ends[tid] != sh->hs[tid + 1]);
			pos = sh->ends[nruns - 1];
		}
		__syncthreads();                                        // y, hs, ends, q are rewritten for the next channel
	}
	if(!bail) {
		// zero bits up to the byte boundary, and the body ends exactly where the CRC-16 starts
		const uint32_t rem = pos & 7u;
		if(pos > S.limit) suspect = 1;
		else if(rem && peek_bits(S, pos, 8 - rem) != 0) suspect = 1;
		else if(pos + (rem ? 8 - rem : 0) != S.limit) suspect = 1;
	}
	else suspect = 1;
	const int any = __syncthreads_or((int)suspect);
	if(tid == 0) {
		fstat[f] = any ? 1u : 0u;
		if(!any) atomicAdd(&state->hinted_ok, 1u);
	}
	VSTAMP(12);
	asm volatile("s_waitcnt vmcnt(0)" :: "v"(pf));
#undef VSTAMP
}

// per-frame verdict of the decode pass: [3:0] channel assignment, bit 8 set = decodes
template <int MAXORD, typename ST>
__global__ __launch_bounds__(64) void verify_kernel(const DevParams P, const uint8_t *__restrict__ frames, const uint32_t *__restrict__ frame_bytes,
                                                    const uint64_t *__restrict__ offsets, uint32_t nframes, uint32_t tail_n, uint64_t first_frame_number,
                                                    ST *__restrict__ decoded, uint32_t *__restrict__ finfo, VerifyState *__restrict__ state,
                                                    const uint32_t *__restrict__ fstat)
{
	const uint32_t f = blockIdx.x * 64u + threadIdx.x;
	if(f >= nframes) return;
	if(fstat && fstat[f] == 0) { finfo[f] = 0x200u; return; }          // verified by the hinted pass: nothing to decode, nothing to compare
	const uint32_t fb = frame_bytes[f];
	const uint32_t C = P.channels, N = P.blocksize;
	DecodeExpect E;
	E.channels = C; E.bps = P.bps; E.blocksize = N; E.n = (tail_n && f + 1 == nframes) ? tail_n : N; E.frame_number = first_frame_number + f;
	int st = DEC_ERROR;
	uint32_t ca = 0;
	if(fb != 0xffffffffu && fb >= 6) {
		const uint8_t *p = frames + offsets[f];
		const uint8_t *hi = frames + offsets[nframes];
		BitReader b;
		br_init(b, p, fb - 2, hi);
		FrameHead H;
		if(decode_frame_header(b, p, E, H) == DEC_OK) {
			st = DEC_OK;
			ca = H.ca;
			for(uint32_t ch = 0; ch < C && st == DEC_OK; ch++) {
				ST *row = decoded + ((size_t)blockIdx.x * C + ch) * N * 64 + threadIdx.x;      // sample i of this frame: row[i * 64]
				auto sink = [&](uint32_t, int64_t v) { *row = (ST)v; row += 64; };                // (samples arrive in order)
				if(decode_subframe<MAXORD, ST>(b, H.n, coded_bps(E.bps, ca, ch), sink) != DEC_OK) st = DEC_ERROR;
			}
			if(st == DEC_OK && decode_frame_tail(b) != DEC_OK) st = DEC_ERROR;
		}
	}
	finfo[f] = ca | (st == DEC_OK ? 0x100u : 0u);
	if(st != DEC_OK) atomicMin(&state->first_bad, f);
}

// decoded coded-channel samples against the input: block = the 64 frames of one decode wavefront x 64 samples
constexpr int CMP_T = 64;
template <typename ST>
__global__ __launch_bounds__(TPB) void verify_compare_kernel(const DevParams P, uint32_t nframes, uint32_t tail_n, const int32_t *__restrict__ pcm,
                                                             const ST *__restrict__ decoded, const uint32_t *__restrict__ finfo, VerifyState *__restrict__ state)
{
	__shared__ ST tile[CMP_T][65];
	__shared__ uint32_t info[64];
	const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const uint32_t wb = blockIdx.x, i0 = blockIdx.y * CMP_T;
	const uint32_t C = P.channels, N = P.blocksize;
	uint32_t mine = 0;
	if(tid < 64) { const uint32_t f = wb * 64u + (uint32_t)tid; mine = f < nframes ? finfo[f] : 0u; info[tid] = mine; }
	if(!__syncthreads_or((int)(mine & 0x100u))) return;            // none of these 64 frames was decoded here (hinted pass, or not decodable)
	uint32_t badmask_lo = 0, badmask_hi = 0;                        // frames (of this wavefront's share) with a differing sample
	for(uint32_t ch = 0; ch < C; ch++) {
		__syncthreads();
		const ST *rows = decoded + ((size_t)wb * C + ch) * N * 64;
		for(int r = wave; r < CMP_T; r += TPB / 64) { const uint32_t i = i0 + (uint32_t)r; tile[r][lane] = i < N ? rows[(size_t)i * 64 + lane] : (ST)0; }
		__syncthreads();
		for(int fl = wave; fl < 64; fl += TPB / 64) {
			const uint32_t f = wb * 64u + (uint32_t)fl, inf = info[fl];
			if(f >= nframes || !(inf & 0x100u)) continue;              // (frames that did not decode are already flagged)
			const uint32_t n = (tail_n && f + 1 == nframes) ? tail_n : N, i = i0 + (uint32_t)lane;
			bool differ = false;
			if(i < n) differ = (int64_t)tile[lane][fl] != coded_expectation(pcm + ((size_t)f * N + i) * C, inf & 15u, ch);
			if(__any((int)differ)) { if(fl < 32) badmask_lo |= 1u << fl; else badmask_hi |= 1u << (fl - 32); }
		}
	}
	if(lane == 0) {
		uint32_t first = 0xffffffffu;
		if(badmask_lo) first = (uint32_t)__ffs((int)badmask_lo) - 1;
		else if(badmask_hi) first = 32u + (uint32_t)__ffs((int)badmask_hi) - 1;
		if(first != 0xffffffffu) atomicMin(&state->first_bad, wb * 64u + first);
	}
}

template <int MAXORD, typename ST>
__global__ __launch_bounds__(64) void verify_detail_kernel(const DevParams P, const uint8_t *__restrict__ frames, const uint32_t *__restrict__ frame_bytes,
                                                           const uint64_t *__restrict__ offsets, uint32_t nframes, uint32_t tail_n, uint64_t first_frame_number,
                                                           const int32_t *__restrict__ pcm, int64_t *__restrict__ scratch, VerifyState *__restrict__ state,
                                                           flacgpu_verify_result *__restrict__ result)
{
	if(threadIdx.x != 0 || blockIdx.x != 0) return;
	flacgpu_verify_result R;
	R.status = 0; R.frame_number = 0; R.channel = 0; R.sample = 0; R.absolute_sample = 0; R.expected = 0; R.got = 0;
	const uint32_t f = state->first_bad;
	if(f < nframes) {
		const uint32_t C = P.channels, N = P.blocksize;
		DecodeExpect E;
		E.channels = C; E.bps = P.bps; E.blocksize = N; E.n = (tail_n && f + 1 == nframes) ? tail_n : N; E.frame_number = first_frame_number + f;
		DecodeDetail D;
		D.status = DEC_ERROR; D.channel = 0; D.sample = 0; D.expected = 0; D.got = 0;
		const uint32_t fb = frame_bytes[f];
		if(fb != 0xffffffffu) verify_frame_detail<MAXORD, ST>(frames + offsets[f], fb, frames + offsets[nframes], E, pcm + (size_t)f * N * C, scratch, N, D);
		// a frame the batch pass (or the CRC pass) rejected but the detail pass accepts cannot be: report it as undecodable
		R.status = D.status == DEC_OK ? DEC_ERROR : D.status;
		R.frame_number = (uint32_t)(first_frame_number + f);
		R.channel = D.channel; R.sample = D.sample;
		R.absolute_sample = (first_frame_number + f) * (uint64_t)N + D.sample;      // stream_encoder.c:5186-5196
		R.expected = D.expected; R.got = D.got;
	}
	*result = R;
}

__global__ void verify_reset_kernel(VerifyState *state) { state->first_bad = 0xffffffffu; state->hinted_ok = 0; }

template <int MAXORD, typename ST>
static hipError_t launch_verify_t(const DevParams &P, const uint8_t *frames, const uint32_t *fb, const uint64_t *offsets, uint32_t nframes, uint32_t tail_n,
                                  uint64_t first, const int32_t *pcm, int64_t *scratch, void *decoded, uint32_t *finfo, VerifyState *state, flacgpu_verify_result *result,
                                  const uint32_t *fstat, hipStream_t s)
{
	const uint32_t nwb = (nframes + 63) / 64;
	hipLaunchKernelGGL((verify_kernel<MAXORD, ST>), dim3(nwb), dim3(64), 0, s, P, frames, fb, offsets, nframes, tail_n, first, (ST *)decoded, finfo, state, fstat);
	hipLaunchKernelGGL((verify_compare_kernel<ST>), dim3(nwb, (P.blocksize + CMP_T - 1) / CMP_T), dim3(TPB), 0, s, P, nframes, tail_n, pcm, (const ST *)decoded, finfo, state);
	hipLaunchKernelGGL((verify_detail_kernel<MAXORD, ST>), dim3(1), dim3(64), 0, s, P, frames, fb, offsets, nframes, tail_n, first, pcm, scratch, state, result);
	return hipGetLastError();
}

size_t verify_decoded_bytes(const DevParams &P, uint32_t max_frames)
{
	const bool wide = P.bps == 32 && P.channels == 2;
	return (size_t)((max_frames + 63) / 64) * 64 * P.channels * P.blocksize * (wide ? 8 : 4);
}
bool verify_hinted_covers(const DevParams &P)
{
	const bool wide = P.bps == 32 && P.channels == 2;
	return !wide && P.blocksize % HINT_RUN == 0 && P.blocksize / HINT_RUN <= HINT_MAX_RUNS && P.max_lpc_order <= HINT_MAX_ORDER && hinted_lds_bytes(P) <= 64 * 1024;
}
template <int MAXORD>
static hipError_t launch_hinted_t(const DevParams &P, const uint8_t *frames, const uint32_t *fb, const uint64_t *offsets, uint32_t nframes, uint32_t nhinted, uint64_t first,
                                  const int32_t *pcm, const uint32_t *hints, uint32_t *fstat, VerifyState *state, unsigned long long *dbg, hipStream_t s)
{
	static AttrFlags attr_set;
	static uint32_t ahead_of[64];
	uint32_t &ahead = ahead_of[tune().device & 63];
	if(AttrOnce once{attr_set}) {
		// workgroups resident at a time = the distance to the one that takes this one's place
		int per_cu = 0, cus = 256;
		hipDeviceProp_t prop;
		int dev = 0;
		if(hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
		if(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)verify_hinted_kernel<MAXORD>, TPB, hinted_lds_bytes(P)) != hipSuccess || per_cu < 1) per_cu = 4;
		const char *e2 = getenv("FLACGPU_VERIFY_PREFETCH");
		ahead = e2 ? (uint32_t)atoi(e2) : (uint32_t)(per_cu * cus);
		const hipError_t e = hipFuncSetAttribute((const void *)verify_hinted_kernel<MAXORD>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
		if(e != hipSuccess) return e;
		once.ok();
	}
	hipLaunchKernelGGL((verify_hinted_kernel<MAXORD>), dim3(nframes), dim3(TPB), hinted_lds_bytes(P), s, P, frames, fb, offsets, nframes, nhinted, first, pcm, hints, fstat, state, dbg, ahead);
	if(dbg) {
		// development aid (FLACGPU_DEBUG_TIMING=1): shader-clock ticks between the stamps of the hinted workgroups
		unsigned long long *h = (unsigned long long *)malloc((size_t)nhinted * 16 * sizeof(unsigned long long));
		if(h && hipStreamSynchronize(s) == hipSuccess && hipMemcpy(h, dbg, (size_t)nhinted * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess) {
			double acc[16] = {0}; size_t cnt[16] = {0};
			for(size_t w = 0; w < nhinted; w++) for(int k = 1; k < 16; k++) if(h[w * 16 + k] && h[w * 16]) { acc[k] += (double)(h[w * 16 + k] - h[w * 16]); cnt[k]++; }
			fprintf(stderr, "[flacgpu] hinted verify stamps (avg ticks since the workgroup's start; 13-15 lie inside 0..1):");
			for(int k = 1; k < 16; k++) fprintf(stderr, " %d:%.0f", k, cnt[k] ? acc[k] / cnt[k] : 0.0);
			fprintf(stderr, "\n");
		}
		free(h);
		(void)hipMemsetAsync(dbg, 0, (size_t)nhinted * 16 * sizeof(unsigned long long), s);
	}
	return hipGetLastError();
}
hipError_t launch_verify(const DevParams &P, const uint8_t *frames, const uint32_t *fb, const uint64_t *offsets, uint32_t nframes, uint32_t tail_n,
                         uint64_t first, const int32_t *pcm, int64_t *scratch, void *decoded, uint32_t *finfo, VerifyState *state, flacgpu_verify_result *result,
                         const uint32_t *hints, uint32_t nhinted, uint32_t *fstat, unsigned long long *dbg, hipStream_t s)
{
	hipLaunchKernelGGL(verify_reset_kernel, dim3(1), dim3(1), 0, s, state);
	hipError_t e = launch_crc_check(frames, fb, offsets, nframes, state, s);
	if(e != hipSuccess) return e;
	const bool wide = P.bps == 32 && P.channels == 2;            // a 33-bit side channel can occur (stream_encoder.c:3831-3835)
	const uint32_t m = P.max_lpc_order;
	// frames the pack kernel left hints for: a thread per run first; what that pass cannot vouch for is decoded sequentially below
	if(tail_n && nhinted >= nframes) nhinted = nframes - 1;      // (the short last block never has hints)
	if(!hints || !fstat || !verify_hinted_covers(P)) nhinted = 0;
	if(nhinted) {
		if(m <= 8) e = launch_hinted_t<8>(P, frames, fb, offsets, nframes, nhinted, first, pcm, hints, fstat, state, dbg, s);
		else if(m <= 12) e = launch_hinted_t<12>(P, frames, fb, offsets, nframes, nhinted, first, pcm, hints, fstat, state, dbg, s);
		else e = launch_hinted_t<16>(P, frames, fb, offsets, nframes, nhinted, first, pcm, hints, fstat, state, dbg, s);
		if(e != hipSuccess) return e;
	}
	else fstat = nullptr;
#define GO(M) (wide ? launch_verify_t<M, int64_t>(P, frames, fb, offsets, nframes, tail_n, first, pcm, scratch, decoded, finfo, state, result, fstat, s) \
                    : launch_verify_t<M, int32_t>(P, frames, fb, offsets, nframes, tail_n, first, pcm, scratch, decoded, finfo, state, result, fstat, s))
	if(m <= 8) e = GO(8);
	else if(m <= 12) e = GO(12);
	else e = GO(32);
#undef GO
	sync_debug("verify", s);
	return e;
}

} // namespace flacgpu
