// flac_amd/csrc/flacgpu_stream_decode.hip -- decoding, on the device, FLAC streams this engine did not write (SURVEY.md 8f row 3:
// what `flac -t` / `flac -d` ask of FLAC__stream_decoder_process_until_end_of_stream, src/libFLAC/stream_decoder.c:1168): a byte
// range in HBM in, interleaved int32 PCM in HBM out, the errors the reference's decoder would have reported, in its order.
//
//   sd_scan_kernel<FILL>   every byte position looked at for a sync code (0xFF, 0xF8 | 0xF9): a wavefront per 16 KiB, coalesced words,
//                          a ballot per 256 bytes; a position that matches gets its header parsed (sd_parse_candidate).  Twice: a count
//                          per wavefront, a prefix sum (sd_prefix_kernel), then the records in stream order.  HBM bound: the stream
//                          is read once per pass.
//   sd_decode_kernel       ONE LANE PER CANDIDATE whose header holds (flacgpu_stream_decode.h: the subframes in the reference's order
//                          of checks, its integer semantics).  Rice decoding is a serial bit chain and where a frame ends is only known
//                          once it is decoded, so the unit of parallelism is the candidate: a wavefront walks 64 of them in lockstep.
//                          Decoded coded-channel samples go to a lane-interleaved scratch (sample i of the wavefront's 64 frames is
//                          one 256-byte row), not to the output: 64 lanes writing 64 frames 32 KiB apart would be 64 partial lines
//                          per store.  Latency / issue bound by construction.
//   crc_check_kernel       (flacgpu_kernels.hip) the CRC-16 of every frame that decoded, spans in parallel.
//   sd_finish_kernel       thread-parallel: rows of the scratch through an LDS transpose, the inter-channel step, the bounds check,
//                          interleaved PCM written coalesced at the place the frame's own number implies.
//   (host)                 flacgpu_stream_walk.h over the candidate table: which candidate the reference's search reaches after
//                          which, the errors in order, silence for missing frames; frames that are not where their number said
//                          (none, in a stream whose frames are all there) are decoded again into place by the same two kernels.
// No MFMA: bit and integer work.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <new>
#include <vector>
#include "flacgpu.h"
#include "flacgpu_dev.h"
#define FLACGPU_HD __host__ __device__
namespace flacgpu {
template <int M>
__device__ __forceinline__ uint32_t sd_dot24_asm(const int32_t (&a)[M], const int32_t (&b)[M])
{
	static_assert(M % 4 == 0, "taps in fours");
	uint32_t d = 0;
#pragma unroll
	for(int j = 0; j < M; j += 4)
		asm("v_mad_i32_i24 %0, %1, %2, %0\n\tv_mad_i32_i24 %0, %3, %4, %0\n\tv_mad_i32_i24 %0, %5, %6, %0\n\tv_mad_i32_i24 %0, %7, %8, %0"
		    : "+v"(d) : "v"(a[j]), "v"(b[j]), "v"(a[j + 1]), "v"(b[j + 1]), "v"(a[j + 2]), "v"(b[j + 2]), "v"(a[j + 3]), "v"(b[j + 3]));
	return d;
}
template <int M> inline uint32_t sd_dot24_host(const int32_t (&a)[M], const int32_t (&b)[M]) { uint32_t s = 0; for(int j = 0; j < M; j++) s += (uint32_t)a[j] * (uint32_t)b[j]; return s; }
template <int M>
__host__ __device__ __forceinline__ uint32_t sd_dot24(const int32_t (&a)[M], const int32_t (&b)[M])
{
#if defined(__HIP_DEVICE_COMPILE__)
	return sd_dot24_asm(a, b);
#else
	return sd_dot24_host(a, b);
#endif
}
}
#define FLACGPU_DOT24(a, b) flacgpu::sd_dot24(a, b)
#include "flacgpu_stream_walk.h"

namespace flacgpu {

constexpr uint32_t SC_ITERS = 64;                        // 64 lanes x 4 bytes x 64 = 16 KiB of stream per wavefront
constexpr uint32_t SC_BYTES = 64 * 4 * SC_ITERS;

struct ScanSummary { unsigned long long total; uint32_t pad[2]; };

// sync codes in the four byte positions of word w (little endian: byte j = bits 8j..8j+7), nx = the word behind it
__device__ __forceinline__ uint32_t sd_match4(uint32_t w, uint32_t nx)
{
	// quick reject: no 0xFF byte in the word
	const uint32_t inv = ~w;
	if((((inv - 0x01010101u) & ~inv) & 0x80808080u) == 0) return 0;
	uint32_t m = 0;
	const uint64_t v = ((uint64_t)nx << 32) | w;
#pragma unroll
	for(int j = 0; j < 4; j++) {
		const uint32_t b0 = (uint32_t)(v >> (8 * j)) & 0xffu, b1 = (uint32_t)(v >> (8 * j + 8)) & 0xffu;
		if(b0 == 0xffu && (b1 >> 1) == 0x7cu) m |= 1u << j;
	}
	return m;
}

// FILL = false: counts[wave] = sync codes in the wavefront's 16 KiB at or behind first_pos.  FILL = true: their records, in order,
// from base[wave] on.  words: the stream as aligned 32-bit words (d_stream is 4-byte aligned and its allocation extends to a multiple
// of four bytes).
template <bool FILL>
__global__ __launch_bounds__(64) void sd_scan_kernel(const uint32_t *__restrict__ words, const uint8_t *__restrict__ bytes, uint64_t nbytes, uint64_t first_pos, const SdInfo I,
                                                     uint32_t *__restrict__ counts, const unsigned long long *__restrict__ base, StreamCand *__restrict__ cand)
{
	const uint32_t lane = threadIdx.x;
	const uint64_t wave = blockIdx.x;
	const uint64_t w0 = wave * (SC_BYTES / 4);
	const uint64_t nwords = (nbytes + 3) / 4;
	uint32_t mine = 0;
	unsigned long long out = FILL ? base[wave] : 0ull;
	for(uint32_t it = 0; it < SC_ITERS; it++) {
		const uint64_t wi = w0 + (uint64_t)it * 64 + lane;
		const uint64_t wc = wi < nwords ? wi : nwords - 1, wn = wi + 1 < nwords ? wi + 1 : nwords - 1;      // (clamped, loaded unconditionally)
		const uint32_t w = words[wc];
		const uint32_t nx = words[wn];
		uint32_t m = wi < nwords ? sd_match4(w, wi + 1 < nwords ? nx : 0u) : 0u;
		if(m) {
			// positions in front of the first frame, and a 0xFF that is the stream's last byte, are no candidates
			const uint64_t p = wi * 4;
#pragma unroll
			for(int j = 0; j < 4; j++) if((m >> j) & 1u) { if(p + j < first_pos || p + j + 1 >= nbytes) m &= ~(1u << j); }
		}
		if(__ballot(m != 0) == 0ull) continue;                           // (the same for the whole wavefront)
		const uint32_t c = (uint32_t)__popc(m);
		if(!FILL) mine += c;
		else {
			// rank among the wavefront's matches of this round: lanes in order, positions within a lane in order
			uint32_t incl = c;
#pragma unroll
			for(int off = 1; off < 64; off <<= 1) { const uint32_t t = __shfl_up(incl, off); if((int)lane >= off) incl += t; }
			const uint32_t total = __shfl(incl, 63);
			uint32_t r = incl - c;
			if(m) {
				auto get = [&](uint64_t i) -> uint32_t { return bytes[i]; };
#pragma unroll
				for(int j = 0; j < 4; j++) if((m >> j) & 1u) {
					StreamCand K;
					sd_parse_candidate(get, nbytes, wi * 4 + j, I, K);
					cand[out + r] = K;
					r++;
				}
			}
			out += total;
		}
	}
	if(!FILL) {
#pragma unroll
		for(int off = 32; off >= 1; off >>= 1) mine += __shfl_xor(mine, off);
		if(lane == 0) counts[wave] = mine;
	}
}

// exclusive prefix sums of the per-wavefront counts, one workgroup: base[w], and the total
__global__ __launch_bounds__(1024) void sd_prefix_kernel(const uint32_t *__restrict__ counts, uint64_t n, unsigned long long *__restrict__ base, ScanSummary *__restrict__ sum)
{
	__shared__ unsigned long long part[1024];
	const uint32_t t = threadIdx.x;
	const uint64_t per = (n + 1023) / 1024, lo = (uint64_t)t * per, hi = lo + per < n ? lo + per : n;
	unsigned long long s = 0;
	for(uint64_t i = lo; i < hi; i++) s += counts[i];
	part[t] = s;
	__syncthreads();
	if(t == 0) { unsigned long long run = 0; for(int k = 0; k < 1024; k++) { const unsigned long long v = part[k]; part[k] = run; run += v; } sum->total = run; }
	__syncthreads();
	unsigned long long run = part[t];
	for(uint64_t i = lo; i < hi; i++) { base[i] = run; run += counts[i]; }
}

// ---- the decode pass --------------------------------------------------------------------------------------------------------
// One lane per entry of a list of candidate indices.  place == 0: the candidates' first decode -- status, length and what the CRC
// kernel needs are recorded; place != 0: frames the walk found good decoded again for their samples only (status to lstat[]).
// retry: take only candidates an earlier instance left at SD_RETRY.
// decoded: [wavefront of the launch][Cmax][Nmax][64 lanes]; d_len / d_off: what crc_check_kernel reads (0xffffffff: nothing to check).
template <int MAXORD, bool EXACT, typename ST>
__global__ __launch_bounds__(64) void sd_decode_kernel(const uint8_t *__restrict__ stream, uint64_t nbytes, const StreamCand *__restrict__ cand, StreamBody *__restrict__ body,
                                                       uint32_t count, const uint32_t *__restrict__ list, uint32_t place, uint32_t Cmax, uint32_t Nmax, ST *__restrict__ decoded,
                                                       uint32_t *__restrict__ d_len, uint64_t *__restrict__ d_off, uint8_t *__restrict__ lstat, uint32_t retry)
{
	const uint32_t g = blockIdx.x * 64u + threadIdx.x;
	if(g >= count) return;
	const uint64_t ci = list[g];
	const StreamCand K = cand[ci];
	uint8_t *stp = place ? lstat + g : &body[ci].bstat;
	if(retry && *stp != SD_RETRY) return;
	ST *base = decoded + (size_t)blockIdx.x * Cmax * Nmax * 64 + threadIdx.x;
	ST *row = base;
	uint32_t cur = 0;
	auto sink = [&](uint32_t ch, uint32_t, int64_t v) {
		if(ch != cur) { cur = ch; row = base + (size_t)ch * Nmax * 64; }
		*row = (ST)v; row += 64;                                         // (samples of a channel arrive in order)
	};
	uint32_t len = 0, pad_error = 0;
	const int st = sd_decode_frame<MAXORD, EXACT, ST>(stream + K.pos, nbytes - K.pos, stream + nbytes, K, sink, &len, &pad_error);
	if(place) { *stp = (uint8_t)st; return; }
	StreamBody B;
	B.spec = ~0ull; B.len = st == SD_OK ? len : 0u; B.bstat = (uint8_t)st; B.oob_mask = 0; B.wrote = 0; B.pad_error = (uint8_t)pad_error;
	body[ci] = B;
	d_len[g] = st == SD_OK ? len : 0xffffffffu;
	d_off[g] = K.pos;
}

// ---- the finish pass -----------------------------------------------------------------------------------------------------------
// block (x: a decode wavefront's 64 frames, y: a tile of T samples).  Frames that decoded and whose CRC-16 holds: inter-channel step,
// bounds check (a bit per offending channel into oob[]), and -- when the frame has the stream's format and its place lies inside the
// output -- interleaved PCM at `spec` = the sample number its header implies minus spec_base (chunk mode), or at place[] (list mode).
struct FinishArgs {
	uint32_t count; const uint32_t *list; const uint64_t *place; const uint8_t *lstat;      // place != null: frames decoded again into place
	uint32_t Cmax, Nmax;
	uint64_t spec_base, capacity;           // inter-channel samples
	uint32_t out_channels, out_bps;
	SdInfo I;
};
template <typename ST>
__global__ __launch_bounds__(TPB) void sd_finish_kernel(const FinishArgs A, const StreamCand *__restrict__ cand, StreamBody *__restrict__ body, const uint8_t *__restrict__ crcbad,
                                                        const ST *__restrict__ decoded, uint32_t *__restrict__ oob, int32_t *__restrict__ pcm)
{
	constexpr int T = sizeof(ST) == 4 ? 64 : 32;
	__shared__ ST tile[2][T][65];
	__shared__ uint64_t f_out[64];
	__shared__ uint32_t f_n[64], f_fmt[64];        // f_fmt: channels | ca << 8 | bps << 16 | ok << 24 | write << 25
	const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const uint32_t wb = blockIdx.x, i0 = blockIdx.y * T;
	if(tid < 64) {
		const uint32_t g = wb * 64u + (uint32_t)tid;
		uint32_t n = 0, fmt = 0;
		uint64_t outp = 0;
		if(g < A.count) {
			const uint64_t ci = A.list[g];
			const StreamCand K = cand[ci];
			bool ok, wr;
			if(A.place) { ok = A.lstat[g] == SD_OK; wr = ok; outp = A.place[g]; }
			else {
				const StreamBody B = body[ci];
				ok = K.hstat == SD_OK && B.bstat == SD_OK && !crcbad[g];
				const uint64_t sn = sd_spec_sample(K, A.I);
				wr = ok && K.channels == A.out_channels && K.bps == A.out_bps && sn >= A.spec_base && sn - A.spec_base + K.blocksize <= A.capacity;
				outp = sn - A.spec_base;
				if(blockIdx.y == 0 && K.hstat == SD_OK) {
					if(B.bstat == SD_OK && crcbad[g]) body[ci].bstat = SD_CRC_MISMATCH;
					if(wr) { body[ci].spec = outp; body[ci].wrote = 1; }
				}
			}
			n = K.blocksize;
			fmt = (uint32_t)K.channels | ((uint32_t)K.ca << 8) | ((uint32_t)K.bps << 16) | (ok ? 1u << 24 : 0u) | (wr ? 1u << 25 : 0u);
		}
		f_n[tid] = n; f_fmt[tid] = fmt; f_out[tid] = outp;
	}
	__syncthreads();
	// anything to do in this tile?
	bool any = false;
	for(int fl = 0; fl < 64; fl++) any = any || (((f_fmt[fl] >> 24) & 1u) && f_n[fl] > i0);
	if(!any) return;
	uint32_t Cm = 0;
	for(int fl = 0; fl < 64; fl++) if((f_fmt[fl] >> 24) & 1u) Cm = max(Cm, f_fmt[fl] & 0xffu);
	uint32_t oobacc[64 / (TPB / 64)];              // this wavefront's frames fl = wave, wave + 4, ...: offending-channel bits seen by this lane
#pragma unroll
	for(int k = 0; k < 64 / (TPB / 64); k++) oobacc[k] = 0;
	const uint32_t C = A.out_channels;
	for(uint32_t chp = 0; chp < Cm; chp += 2) {
		// two channels at a time (a stereo frame's pair goes through the inter-channel step together)
		__syncthreads();
		for(int cc = 0; cc < 2; cc++) {
			if(chp + cc >= Cm) break;
			const ST *rows = decoded + ((size_t)wb * A.Cmax + chp + cc) * A.Nmax * 64;
			for(int r = wave; r < T; r += TPB / 64) { const uint32_t i = i0 + (uint32_t)r; tile[cc][r][lane] = i < A.Nmax ? rows[(size_t)i * 64 + lane] : (ST)0; }
		}
		__syncthreads();
		for(int k = 0; k < 64 / (TPB / 64); k++) {
			const int fl = wave + k * (TPB / 64);
			const uint32_t fmt = f_fmt[fl], n = f_n[fl];
			if(!((fmt >> 24) & 1u)) continue;
			const uint32_t fc = fmt & 0xffu, ca = (fmt >> 8) & 0xffu, bps = (fmt >> 16) & 0xffu;
			const bool wr = (fmt >> 25) & 1u;
			if(chp >= fc) continue;
			{
				const uint32_t li = (uint32_t)lane, i = i0 + li;                 // (T = 32, the 33-bit instance: half the lanes idle)
				if(li >= (uint32_t)T || i >= n) continue;
				int32_t o0, o1 = 0;
				const bool two = chp + 1 < fc;
				if(fc == 2) sd_undo_channels(ca, bps, (int64_t)tile[0][li][fl], (int64_t)tile[1][li][fl], o0, o1);
				else { o0 = (int32_t)tile[0][li][fl]; if(two) o1 = (int32_t)tile[1][li][fl]; }
				if(sd_out_of_bounds(o0, bps)) oobacc[k] |= 1u << chp;
				if(two && sd_out_of_bounds(o1, bps)) oobacc[k] |= 1u << (chp + 1);
				if(wr) {
					int32_t *dst = pcm + (f_out[fl] + i) * C + chp;
					if(C == 2) *(int2 *)dst = make_int2(o0, o1);
					else { dst[0] = o0; if(two) dst[1] = o1; }
				}
			}
		}
	}
	if(!A.place) {
#pragma unroll
		for(int k = 0; k < 64 / (TPB / 64); k++) {
			uint32_t v = oobacc[k];
#pragma unroll
			for(int off = 32; off >= 1; off >>= 1) v |= __shfl_xor(v, off);
			const uint32_t g = wb * 64u + (uint32_t)(wave + k * (TPB / 64));
			if(lane == 0 && v && g < A.count) atomicOr(&oob[g], v);
		}
	}
}

// int32 PCM -> the sample bytes a WAVE file or the MD5 of STREAMINFO holds: little endian, ceil(bps / 8) bytes per sample
// (what FLAC__MD5Accumulate is fed, src/libFLAC/md5.c:497; the inverse of flacgpu_stage.hip)
__global__ __launch_bounds__(TPB) void sd_pack_kernel(const int32_t *__restrict__ pcm, uint64_t nvalues, uint32_t bytes_per, uint8_t *__restrict__ out)
{
	const uint64_t i = (uint64_t)blockIdx.x * TPB + threadIdx.x;
	if(i >= nvalues) return;
	const uint32_t v = (uint32_t)pcm[i];
	uint8_t *q = out + i * bytes_per;
	for(uint32_t k = 0; k < bytes_per; k++) q[k] = (uint8_t)(v >> (8 * k));
}

} // namespace flacgpu

using namespace flacgpu;

// a device buffer that grows and is kept between calls
struct SdBuf {
	void *p = nullptr; size_t bytes = 0;
	bool need(size_t n)
	{
		if(n <= bytes && p) return true;
		if(p) (void)hipFree(p);
		p = nullptr; bytes = 0;
		const size_t want = n + n / 4 + 256;
		if(hipMalloc(&p, want) != hipSuccess) { p = nullptr; return false; }
		bytes = want;
		return true;
	}
	void release() { if(p) (void)hipFree(p); p = nullptr; bytes = 0; }
};
struct flacgpu_decoder {
	int device;
	hipStream_t own_stream;
	SdBuf counts, base, cand, body, decoded, len, off, crcbad, oob, list, place, lstat;
	ScanSummary *d_sum;
	VerifyState *d_vstate;
};

extern "C" void flacgpu_decoder_destroy(flacgpu_decoder *d);
extern "C" int flacgpu_decoder_create(int device, flacgpu_decoder **out)
{
	if(!out) return FLACGPU_ERR_BAD_ARG;
	*out = nullptr;
	int ndev = 0;
	if(hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return FLACGPU_ERR_NO_DEVICE;
	if(hipSetDevice(device) != hipSuccess) return FLACGPU_ERR_NO_DEVICE;
	flacgpu_decoder *d = new (std::nothrow) flacgpu_decoder();
	if(!d) return FLACGPU_ERR_ALLOC;
	d->device = device; d->own_stream = nullptr; d->d_sum = nullptr; d->d_vstate = nullptr;
	if(hipStreamCreateWithFlags(&d->own_stream, hipStreamNonBlocking) != hipSuccess || hipMalloc((void **)&d->d_sum, sizeof(ScanSummary)) != hipSuccess ||
	   hipMalloc((void **)&d->d_vstate, sizeof(VerifyState)) != hipSuccess) { flacgpu_decoder_destroy(d); return FLACGPU_ERR_ALLOC; }
	*out = d;
	return FLACGPU_OK;
}
extern "C" void flacgpu_decoder_destroy(flacgpu_decoder *d)
{
	if(!d) return;
	(void)hipSetDevice(d->device);
	SdBuf *bs[] = {&d->counts, &d->base, &d->cand, &d->body, &d->decoded, &d->len, &d->off, &d->crcbad, &d->oob, &d->list, &d->place, &d->lstat};
	for(SdBuf *b : bs) b->release();
	if(d->d_sum) (void)hipFree(d->d_sum);
	if(d->d_vstate) (void)hipFree(d->d_vstate);
	if(d->own_stream) (void)hipStreamDestroy(d->own_stream);
	delete d;
}

extern "C" int flacgpu_probe_stream(const uint8_t *head, size_t nbytes, flacgpu_stream_info *si, uint64_t *first_frame_offset, uint64_t *total_samples, uint8_t *md5)
{
	if(!head || !si || !first_frame_offset) return FLACGPU_ERR_BAD_ARG;
	SdInfo I;
	uint64_t first = 0;
	const bool ok = sd_probe_metadata(head, nbytes, I, first, total_samples, md5);
	si->has_streaminfo = I.has_streaminfo; si->min_blocksize = I.min_blocksize; si->max_blocksize = I.max_blocksize; si->sample_rate = I.sample_rate;
	si->channels = I.channels; si->bits_per_sample = I.bps;
	*first_frame_offset = first;
	return ok ? FLACGPU_OK : FLACGPU_ERR_INPUT;
}

extern "C" int flacgpu_pack_samples_device(int device, const int32_t *d_pcm, uint64_t nvalues, uint32_t bits_per_sample, uint8_t *d_out, void *stream)
{
	if(!d_pcm || !d_out || bits_per_sample < 1 || bits_per_sample > 32) return FLACGPU_ERR_BAD_ARG;
	if(hipSetDevice(device) != hipSuccess) return FLACGPU_ERR_NO_DEVICE;
	if(nvalues == 0) return FLACGPU_OK;
	const uint64_t blocks = (nvalues + TPB - 1) / TPB;
	if(blocks > 0x7fffffffull) return FLACGPU_ERR_BAD_ARG;
	hipLaunchKernelGGL(sd_pack_kernel, dim3((uint32_t)blocks), dim3(TPB), 0, (hipStream_t)stream, d_pcm, nvalues, (bits_per_sample + 7) / 8, d_out);
	return hipGetLastError() == hipSuccess ? FLACGPU_OK : FLACGPU_ERR_LAUNCH;
}

namespace {
struct ChunkPlan { uint32_t Cmax, Nmax; bool wide; uint32_t frames_per_chunk; };

template <typename ST>
hipError_t run_decode(flacgpu_decoder *d, const uint8_t *stream, uint64_t nbytes, uint32_t count, bool place, const ChunkPlan &cp, hipStream_t s)
{
	const uint32_t nwb = (count + 63) / 64;
	hipLaunchKernelGGL((sd_decode_kernel<12, false, ST>), dim3(nwb), dim3(64), 0, s, stream, nbytes, (const StreamCand *)d->cand.p, (StreamBody *)d->body.p, count,
	                   (const uint32_t *)d->list.p, place ? 1u : 0u, cp.Cmax, cp.Nmax, (ST *)d->decoded.p, (uint32_t *)d->len.p, (uint64_t *)d->off.p, (uint8_t *)d->lstat.p, 0u);
	hipLaunchKernelGGL((sd_decode_kernel<32, true, ST>), dim3(nwb), dim3(64), 0, s, stream, nbytes, (const StreamCand *)d->cand.p, (StreamBody *)d->body.p, count,
	                   (const uint32_t *)d->list.p, place ? 1u : 0u, cp.Cmax, cp.Nmax, (ST *)d->decoded.p, (uint32_t *)d->len.p, (uint64_t *)d->off.p, (uint8_t *)d->lstat.p, 1u);
	return hipGetLastError();
}
template <typename ST>
hipError_t run_finish(flacgpu_decoder *d, const FinishArgs &A, int32_t *pcm, hipStream_t s)
{
	constexpr uint32_t T = sizeof(ST) == 4 ? 64 : 32;
	const uint32_t nwb = (A.count + 63) / 64;
	hipLaunchKernelGGL((sd_finish_kernel<ST>), dim3(nwb, (A.Nmax + T - 1) / T), dim3(TPB), 0, s, A, (const StreamCand *)d->cand.p, (StreamBody *)d->body.p, (const uint8_t *)d->crcbad.p, (const ST *)d->decoded.p, (uint32_t *)d->oob.p, pcm);
	return hipGetLastError();
}
}

#define SD_CK(x) do { if((x) != hipSuccess) return FLACGPU_ERR_LAUNCH; } while(0)

extern "C" int flacgpu_decode_stream_device(flacgpu_decoder *d, const uint8_t *d_stream, uint64_t nbytes, uint64_t first_frame_offset, const flacgpu_stream_info *si,
                                            int32_t *d_pcm, uint64_t pcm_capacity_values, flacgpu_decode_result *result, flacgpu_decode_event *events, uint32_t max_events,
                                            void *stream)
{
	if(!d || !result || (nbytes && !d_stream) || ((uintptr_t)d_stream & 3) || first_frame_offset > nbytes) return FLACGPU_ERR_BAD_ARG;
	if(hipSetDevice(d->device) != hipSuccess) return FLACGPU_ERR_NO_DEVICE;
	hipStream_t s = stream ? (hipStream_t)stream : d->own_stream;
	memset(result, 0, sizeof *result);
	SdInfo I;
	memset(&I, 0, sizeof I);
	if(si && si->has_streaminfo) { I.has_streaminfo = 1; I.min_blocksize = si->min_blocksize; I.max_blocksize = si->max_blocksize; I.sample_rate = si->sample_rate; I.channels = si->channels; I.bps = si->bits_per_sample; }
	hipEvent_t ev[4];
	for(auto &e : ev) if(hipEventCreate(&e) != hipSuccess) return FLACGPU_ERR_ALLOC;
	struct EvGuard { hipEvent_t *e; ~EvGuard() { for(int k = 0; k < 4; k++) (void)hipEventDestroy(e[k]); } } evg{ev};
	std::vector<StreamCand> cand;
	std::vector<StreamBody> body;
	uint32_t last_byte = 0;
	uint64_t ncand = 0;
	(void)hipEventRecord(ev[0], s);
	if(nbytes >= 2) {
		// ---- scan: count, prefix, fill
		const uint64_t waves = (nbytes + SC_BYTES - 1) / SC_BYTES;
		if(waves > 0x7fffffffull) return FLACGPU_ERR_UNSUPPORTED;
		if(!d->counts.need((size_t)waves * 4) || !d->base.need((size_t)waves * 8)) return FLACGPU_ERR_ALLOC;
		const uint32_t *words = (const uint32_t *)d_stream;
		hipLaunchKernelGGL((sd_scan_kernel<false>), dim3((uint32_t)waves), dim3(64), 0, s, words, d_stream, nbytes, first_frame_offset, I, (uint32_t *)d->counts.p, (const unsigned long long *)nullptr, (StreamCand *)nullptr);
		hipLaunchKernelGGL(sd_prefix_kernel, dim3(1), dim3(1024), 0, s, (const uint32_t *)d->counts.p, waves, (unsigned long long *)d->base.p, d->d_sum);
		ScanSummary sum;
		SD_CK(hipMemcpyAsync(&sum, d->d_sum, sizeof sum, hipMemcpyDeviceToHost, s));
		uint8_t lb = 0;
		SD_CK(hipMemcpyAsync(&lb, d_stream + nbytes - 1, 1, hipMemcpyDeviceToHost, s));
		SD_CK(hipStreamSynchronize(s));
		last_byte = lb;
		ncand = sum.total;
		if(ncand > 0xfffffff0ull) return FLACGPU_ERR_UNSUPPORTED;
		if(ncand) {
			if(!d->cand.need((size_t)ncand * sizeof(StreamCand)) || !d->body.need((size_t)ncand * sizeof(StreamBody))) return FLACGPU_ERR_ALLOC;
			hipLaunchKernelGGL((sd_scan_kernel<true>), dim3((uint32_t)waves), dim3(64), 0, s, words, d_stream, nbytes, first_frame_offset, I, (uint32_t *)d->counts.p, (const unsigned long long *)d->base.p, (StreamCand *)d->cand.p);
			cand.resize((size_t)ncand);
			SD_CK(hipMemcpyAsync(cand.data(), d->cand.p, (size_t)ncand * sizeof(StreamCand), hipMemcpyDeviceToHost, s));
			SD_CK(hipStreamSynchronize(s));
		}
	}
	else if(nbytes == 1) { uint8_t lb = 0; SD_CK(hipMemcpy(&lb, d_stream, 1, hipMemcpyDeviceToHost)); last_byte = lb; }
	(void)hipEventRecord(ev[1], s);
	// ---- which candidates are decoded at once.  A sync code inside a frame's data passes the header's CRC-8 once in 256 times and
	// then claims any block size and channel count; decoded with the rest it would set the scratch's shape (8 x 65535 samples per
	// lane) and keep one lane busy long after its wavefront's other 63 are done.  In a stream whose frames are all there the search
	// never reaches such a candidate.  So: the candidates that look like the stream's frames (STREAMINFO's format and block-size
	// range; without STREAMINFO: the format of the first header that holds) are decoded now, the others are DEFERRED -- only if
	// the walk reaches one of them are they decoded too, all of them, and the walk run again.
	uint32_t out_channels = I.has_streaminfo ? I.channels : 0, out_bps = I.has_streaminfo ? I.bps : 0;
	uint64_t spec_base = 0;
	bool have_first = false;
	for(const StreamCand &K : cand) {
		if(K.hstat != SD_OK) continue;
		if(!have_first) { have_first = true; spec_base = sd_spec_sample(K, I); if(!out_channels) { out_channels = K.channels; out_bps = K.bps; } }
		break;
	}
	body.resize((size_t)ncand);
	std::vector<uint32_t> main_list, deferred;
	ChunkPlan cpm = {1, 1, false, 0}, cpd = {1, 1, false, 0};
	for(size_t i = 0; i < (size_t)ncand; i++) {
		const StreamCand &K = cand[i];
		StreamBody &B = body[i];
		B.spec = ~0ull; B.len = 0; B.bstat = SD_NOT_DECODED; B.oob_mask = 0; B.wrote = 0; B.pad_error = 0;
		if(K.hstat != SD_OK) continue;
		const bool plausible = K.channels == out_channels && K.bps == out_bps && (!I.has_streaminfo || K.blocksize <= I.max_blocksize);
		ChunkPlan &cp = plausible ? cpm : cpd;
		cp.Cmax = std::max<uint32_t>(cp.Cmax, K.channels); cp.Nmax = std::max<uint32_t>(cp.Nmax, K.blocksize);
		if(K.bps == 32 && K.ca != 0) cp.wide = true;
		(plausible ? main_list : deferred).push_back((uint32_t)i);
		B.bstat = SD_DEFERRED;
	}
	if(ncand) SD_CK(hipMemcpyAsync(d->body.p, body.data(), (size_t)ncand * sizeof(StreamBody), hipMemcpyHostToDevice, s));
	std::vector<uint32_t> oob_list;                                       // per list entry, filled chunk by chunk
	// first decode of a list of candidates: decode, CRC-16, finish (speculative placement), statuses into d->body
	auto first_decode = [&](const std::vector<uint32_t> &list, ChunkPlan &cp) -> int {
		if(list.empty()) return FLACGPU_OK;
		const size_t elem = cp.wide ? 8 : 4;
		// chunks of candidates sized by the lane-interleaved scratch: as many wavefronts in one launch as a quarter of the free HBM
		// (at most 32 GiB of the 288) holds -- a lane is a serial bit chain, what fills the chip is the number of frames in flight:
		// 16384 frames are one wavefront per CU, a ten-hour stream's 390 000 are six per SIMD
		const size_t per64 = (size_t)cp.Cmax * cp.Nmax * 64 * elem;
		size_t budget = (size_t)1 << 30, mfree = 0, mtotal = 0;
		if(hipMemGetInfo(&mfree, &mtotal) == hipSuccess) budget = std::max(budget, std::min<size_t>((size_t)32 << 30, mfree / 4));
		if(d->decoded.bytes > budget) budget = d->decoded.bytes;          // (what an earlier call left is there to be used)
		size_t wpc = budget / per64;
		if(wpc < 1) wpc = 1;
		if(wpc > 8192) wpc = 8192;                                        // 524288 candidates
		const size_t total_wb = (list.size() + 63) / 64;
		if(wpc > total_wb) wpc = total_wb;
		cp.frames_per_chunk = (uint32_t)(wpc * 64);
		if(!d->decoded.need(wpc * per64) || !d->len.need((size_t)cp.frames_per_chunk * 4) || !d->off.need((size_t)cp.frames_per_chunk * 8) ||
		   !d->crcbad.need(cp.frames_per_chunk) || !d->oob.need((size_t)cp.frames_per_chunk * 4) || !d->lstat.need(64) || !d->list.need(list.size() * 4)) return FLACGPU_ERR_ALLOC;
		oob_list.assign(list.size(), 0);
		for(size_t l0 = 0; l0 < list.size(); l0 += cp.frames_per_chunk) {
			const uint32_t count = (uint32_t)std::min<size_t>(cp.frames_per_chunk, list.size() - l0);
			SD_CK(hipMemcpyAsync(d->list.p, list.data() + l0, (size_t)count * 4, hipMemcpyHostToDevice, s));
			SD_CK(hipMemsetAsync(d->crcbad.p, 0, count, s));
			SD_CK(hipMemsetAsync(d->oob.p, 0, (size_t)count * 4, s));
			SD_CK(cp.wide ? run_decode<int64_t>(d, d_stream, nbytes, count, false, cp, s) : run_decode<int32_t>(d, d_stream, nbytes, count, false, cp, s));
			SD_CK(launch_crc_check(d_stream, (const uint32_t *)d->len.p, (const uint64_t *)d->off.p, count, d->d_vstate, s, (uint8_t *)d->crcbad.p));
			FinishArgs A;
			A.count = count; A.list = (const uint32_t *)d->list.p; A.place = nullptr; A.lstat = nullptr; A.Cmax = cp.Cmax; A.Nmax = cp.Nmax; A.spec_base = spec_base;
			A.capacity = d_pcm && out_channels ? pcm_capacity_values / out_channels : 0; A.out_channels = out_channels; A.out_bps = out_bps; A.I = I;
			SD_CK(cp.wide ? run_finish<int64_t>(d, A, d_pcm, s) : run_finish<int32_t>(d, A, d_pcm, s));
			SD_CK(hipMemcpyAsync(oob_list.data() + l0, d->oob.p, (size_t)count * 4, hipMemcpyDeviceToHost, s));
		}
		// (the statuses of the listed candidates come back with the whole table: 16 bytes per sync code)
		std::vector<StreamBody> back((size_t)ncand);
		SD_CK(hipMemcpyAsync(back.data(), d->body.p, (size_t)ncand * sizeof(StreamBody), hipMemcpyDeviceToHost, s));
		SD_CK(hipStreamSynchronize(s));
		for(size_t k = 0; k < list.size(); k++) {
			StreamBody &B = body[list[k]];
			B = back[list[k]];
			if(oob_list[k] && B.bstat == SD_OK) { B.bstat = SD_OUT_OF_BOUNDS; B.oob_mask = (uint8_t)oob_list[k]; }
		}
		return FLACGPU_OK;
	};
	{ const int r = first_decode(main_list, cpm); if(r != FLACGPU_OK) return r; }
	(void)hipEventRecord(ev[2], s);
	// ---- the walk
	WalkResult W;
	sd_walk(I, first_frame_offset, nbytes, last_byte, cand.data(), body.data(), (size_t)ncand, W);
	if(W.need_deferred) {
		result->deferred_decoded = deferred.size();
		const int r = first_decode(deferred, cpd); if(r != FLACGPU_OK) return r;
		W = WalkResult();
		sd_walk(I, first_frame_offset, nbytes, last_byte, cand.data(), body.data(), (size_t)ncand, W);
	}
	result->samples = W.samples; result->frames = W.frames; result->silence_samples = W.silence_samples; result->candidates = ncand;
	result->nevents = (uint32_t)W.events.size(); result->end_in_header = W.end_in_header; result->format_changes = W.format_changes;
	result->long_rice_codes = W.long_rice_codes; result->channels = W.channels; result->bits_per_sample = W.bps; result->sample_rate = W.sample_rate;
	for(const WalkEvent &e : W.events) if(e.status < 8) result->errors_by_status[e.status]++;
	for(size_t k = 0; k < W.events.size() && k < max_events && events; k++) { events[k].status = W.events[k].status; events[k].pad = 0; events[k].byte_offset = W.events[k].pos; }
	int rc = FLACGPU_OK;
	if(d_pcm && W.samples * W.channels > pcm_capacity_values) rc = FLACGPU_ERR_OUTPUT_TOO_SMALL;
	else if(d_pcm && W.samples) {
		// ---- frames that are not yet where the walk puts them.  A candidate the finish kernel wrote that the walk does not place
		// there has scribbled over that range ("dirty"); good frames that were not written, or lie in a dirty range, are decoded again
		// into place; silence last.
		const bool format_as_guessed = W.channels == out_channels && W.bps == out_bps;
		std::vector<uint8_t> placed((size_t)ncand, 0);
		for(const WalkPlace &P : W.places) if(P.cand >= 0 && format_as_guessed && body[(size_t)P.cand].wrote && body[(size_t)P.cand].spec == P.out) placed[(size_t)P.cand] = 1;
		std::vector<std::pair<uint64_t, uint64_t>> dirty;
		for(size_t i = 0; i < (size_t)ncand; i++) if(body[i].wrote && !placed[i]) dirty.push_back({body[i].spec, body[i].spec + cand[i].blocksize});
		std::sort(dirty.begin(), dirty.end());
		// merged, so that one binary search answers "does [a, b) touch any"
		std::vector<std::pair<uint64_t, uint64_t>> merged;
		for(auto &iv : dirty) { if(!merged.empty() && iv.first <= merged.back().second) merged.back().second = std::max(merged.back().second, iv.second); else merged.push_back(iv); }
		auto touches = [&](uint64_t a, uint64_t b) {
			auto it = std::upper_bound(merged.begin(), merged.end(), std::make_pair(b, (uint64_t)0), [](const std::pair<uint64_t, uint64_t> &x, const std::pair<uint64_t, uint64_t> &y) { return x.first < y.first; });
			if(it == merged.begin()) return false;
			--it;
			return it->second > a && it->first < b;
		};
		std::vector<uint32_t> list;
		std::vector<uint64_t> place;
		ChunkPlan cp = {1, 1, false, 0};
		for(const WalkPlace &P : W.places) if(P.cand >= 0 && (!placed[(size_t)P.cand] || touches(P.out, P.out + P.n))) {
			const StreamCand &K = cand[(size_t)P.cand];
			list.push_back((uint32_t)P.cand); place.push_back(P.out);
			cp.Cmax = std::max<uint32_t>(cp.Cmax, K.channels); cp.Nmax = std::max<uint32_t>(cp.Nmax, K.blocksize);
			if(K.bps == 32 && K.ca != 0) cp.wide = true;
		}
		result->redecoded_frames = list.size();
		if(!list.empty()) {
			const size_t per64 = (size_t)cp.Cmax * cp.Nmax * 64 * (cp.wide ? 8 : 4);
			size_t wpc = std::max<size_t>(1, std::min<size_t>(256, ((size_t)1 << 30) / per64));
			cp.frames_per_chunk = (uint32_t)(wpc * 64);
			if(!d->decoded.need(wpc * per64) || !d->len.need((size_t)cp.frames_per_chunk * 4) || !d->off.need((size_t)cp.frames_per_chunk * 8)) return FLACGPU_ERR_ALLOC;
		}
		for(size_t l0 = 0; l0 < list.size(); l0 += cp.frames_per_chunk) {
			const uint32_t count = (uint32_t)std::min<size_t>(cp.frames_per_chunk, list.size() - l0);
			if(!d->list.need((size_t)count * 4) || !d->place.need((size_t)count * 8) || !d->lstat.need(count)) return FLACGPU_ERR_ALLOC;
			SD_CK(hipMemcpyAsync(d->list.p, list.data() + l0, (size_t)count * 4, hipMemcpyHostToDevice, s));
			SD_CK(hipMemcpyAsync(d->place.p, place.data() + l0, (size_t)count * 8, hipMemcpyHostToDevice, s));
			SD_CK(hipMemsetAsync(d->lstat.p, 0, count, s));
			SD_CK(cp.wide ? run_decode<int64_t>(d, d_stream, nbytes, count, true, cp, s) : run_decode<int32_t>(d, d_stream, nbytes, count, true, cp, s));
			FinishArgs A;
			A.count = count; A.list = (const uint32_t *)d->list.p; A.place = (const uint64_t *)d->place.p; A.lstat = (const uint8_t *)d->lstat.p; A.Cmax = cp.Cmax; A.Nmax = cp.Nmax; A.spec_base = 0;
			A.capacity = pcm_capacity_values / W.channels; A.out_channels = W.channels; A.out_bps = W.bps; A.I = I;
			SD_CK(cp.wide ? run_finish<int64_t>(d, A, d_pcm, s) : run_finish<int32_t>(d, A, d_pcm, s));
		}
		for(const WalkPlace &P : W.places) if(P.cand < 0) SD_CK(hipMemsetAsync(d_pcm + P.out * W.channels, 0, (size_t)P.n * W.channels * 4, s));
	}
	(void)hipEventRecord(ev[3], s);
	SD_CK(hipStreamSynchronize(s));
	float ms = 0;
	if(hipEventElapsedTime(&ms, ev[0], ev[1]) == hipSuccess) result->ms_scan = ms;
	if(hipEventElapsedTime(&ms, ev[1], ev[2]) == hipSuccess) result->ms_decode = ms;
	if(hipEventElapsedTime(&ms, ev[2], ev[3]) == hipSuccess) result->ms_place = ms;
	if(hipEventElapsedTime(&ms, ev[0], ev[3]) == hipSuccess) result->ms_total = ms;
	return rc;
}
