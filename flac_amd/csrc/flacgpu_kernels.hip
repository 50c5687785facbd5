// flac_amd/csrc/flacgpu_kernels.hip -- CDNA4 (gfx950) kernels of the FLAC frame engine.
//
// One launch encodes a batch of thousands of independent frames (the reference encodes one frame
// per thread-pool task, src/libFLAC/stream_encoder.c:3627-3744):
//
//   (the model search -- prep / autoc / model / eval kernels -- lives in flacgpu_analyze.hip and hands
//    one small SubDecision record per (frame, candidate channel) to the pack kernel)
//   pack_kernel    : one workgroup per frame.  Channel-assignment argmin (stream_encoder.c:3944-3972),
//                    recomputes only the winning residuals, per-symbol bit lengths -> workgroup prefix
//                    sum -> bits OR-ed into an LDS frame image (stream_encoder_framing.c:245-594,
//                    bitwriter.c:575), CRC-8 / parallel CRC-16 (crc.c:366,376), coalesced store.
//   scan/compact   : exclusive prefix sum of frame lengths, frames packed back to back.
//
// Integer/byte work, HBM/LDS/VALU bound: no MFMA by design.  Compile with -ffp-contract=off: the
// fp64 sections must round exactly like the reference binary.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "flacgpu_dev.h"

#include "flacgpu_devfn.h"

namespace flacgpu {

// ---------------------------------------------------------------------------------------------
// pack_kernel
// ---------------------------------------------------------------------------------------------
// Frame image in LDS as 32-bit words holding the stream MSB first (word = big-endian view).
__device__ __forceinline__ void put_bits(uint32_t *buf, uint32_t cap_words, uint32_t pos, uint32_t v, uint32_t len)
{
	if(len == 0) return;
	const uint32_t w = pos >> 5, o = pos & 31;
	const uint64_t val = (uint64_t)(len == 32 ? v : (v & ((1u << len) - 1u))) << (64 - len - o);
	const uint32_t hi = (uint32_t)(val >> 32), lo = (uint32_t)val;
	if(hi && w < cap_words) atomicOr(&buf[w], hi);
	if(lo && w + 1 < cap_words) atomicOr(&buf[w + 1], lo);
}

// ---- CRC-16 (poly 0x8005, init 0, crc.c:376) -----------------------------------------------------------------
// The frame image is cut into 64-byte spans from byte 0; a thread runs one span a 32-bit word at a time with four
// 256-entry tables (TAB[k][v] = v * x^(16+8k) mod P: one independent lookup per byte of the word, no serial
// byte chain), shifts its remainder past the whole spans behind it with a precomputed x^(512 m) mod P, and the
// spans are xor-reduced: crc(A||B) = crc(A) * x^(8|B|) + crc(B) in GF(2)[x]/(x^16+x^15+x^2+1).
constexpr uint32_t CRC_SPAN = 64;                       // bytes per span
constexpr uint32_t CRC_MAX_SPANS = 4 * 1024 * 1024 / 64;    // frames up to 4 MiB (8 channels x 65535 samples x 33 bits = 2.1 MiB)
// pack2_kernel cuts its LDS image into 44-byte spans instead: 11 words per lane is an ODD word stride, so the 64 lanes of a
// wavefront read 64 different banks (a 16-word stride puts them on 4 banks: every image read a 16-way conflict)
constexpr uint32_t CRC2_SPAN = 44, CRC2_WORDS = 11;
constexpr uint32_t CRC2_MAX_SPANS = 4096;                // x 44 bytes = 176 KiB: more than an LDS image can be
struct CrcTables { uint16_t tab[4][256]; uint16_t xspan[CRC_MAX_SPANS]; uint16_t xbyte[CRC_SPAN + 1]; uint16_t xspan44[CRC2_MAX_SPANS]; };
constexpr uint32_t crc_mulx(uint32_t c) { return (c & 0x8000u) ? ((c << 1) ^ 0x8005u) & 0xffffu : (c << 1) & 0xffffu; }
constexpr uint32_t crc_mulx8(uint32_t c) { for(int b = 0; b < 8; b++) c = crc_mulx(c); return c; }
constexpr CrcTables make_crc_tables()
{
	CrcTables t{};
	for(uint32_t v = 0; v < 256; v++) {
		uint32_t c = v << 8;                  // v * x^8
		c = crc_mulx8(c);                     // v * x^16 mod P
		for(int k = 0; k < 4; k++) { t.tab[k][v] = (uint16_t)c; c = crc_mulx8(c); }
	}
	uint32_t c = 1;
	for(uint32_t r = 0; r <= CRC_SPAN; r++) { t.xbyte[r] = (uint16_t)c; c = crc_mulx8(c); }      // x^(8r)
	const uint32_t xs = t.xbyte[CRC_SPAN];                                                     // x^512
	c = 1;
	for(uint32_t m = 0; m < CRC_MAX_SPANS; m++) {
		t.xspan[m] = (uint16_t)c;
		// c *= xs  (schoolbook product mod P)
		uint32_t r = 0, b = xs;
		for(int i = 0; i < 16; i++) { r = crc_mulx(r); if(b & 0x8000u) r ^= c; b = (b << 1) & 0xffffu; }
		c = r;
	}
	const uint32_t xs44 = t.xbyte[CRC2_SPAN];                                                  // x^352
	c = 1;
	for(uint32_t m = 0; m < CRC2_MAX_SPANS; m++) {
		t.xspan44[m] = (uint16_t)c;
		uint32_t r = 0, b = xs44;
		for(int i = 0; i < 16; i++) { r = crc_mulx(r); if(b & 0x8000u) r ^= c; b = (b << 1) & 0xffffu; }
		c = r;
	}
	return t;
}
__device__ const CrcTables g_crc_tables = make_crc_tables();

// multiply two GF(2) polynomials of degree < 16 modulo x^16+x^15+x^2+1
__device__ __forceinline__ uint32_t gf16_mul(uint32_t a, uint32_t b)
{
	uint32_t r = 0;
#pragma unroll
	for(int i = 0; i < 16; i++) {
		r = (r & 0x8000u) ? ((r << 1) ^ 0x8005u) & 0xffffu : (r << 1) & 0xffffu;
		if(b & 0x8000u) r ^= a;
		b <<= 1;
	}
	return r;
}

// frame header (stream_encoder_framing.c:245-391, bitwriter.c:832): every byte goes to sink(byte) as it is produced,
// the CRC-8 (poly 0x07) runs along; returns the number of bytes including the CRC.  No byte array: a dynamically
// indexed local array lives in scratch memory, and its round trips were the slowest part of the pack kernel.
template <class SINK>
__device__ __forceinline__ uint32_t frame_header_gen(const DevParams &P, uint32_t n, uint32_t ca, uint32_t frame_number, SINK sink)
{
	uint32_t nb = 0, crc = 0;
	auto put = [&](uint32_t byte) {
		byte &= 0xffu;
		// crc = (crc ^ byte) * x^8 mod x^8+x^2+x+1: x^8 = x^2+x+1, applied twice (the first product has 10 bits)
		const uint32_t v = crc ^ byte, w = v ^ (v << 1) ^ (v << 2), hi = w >> 8;
		crc = (w ^ hi ^ (hi << 1) ^ (hi << 2)) & 0xffu;
		sink(nb, byte);
		nb++;
	};
	const uint32_t C = P.channels;
	uint32_t bs_code, bs_hint = 0, sr_code, sr_hint = 0;
	switch(n) {
		case 192: bs_code = 1; break; case 576: bs_code = 2; break; case 1152: bs_code = 3; break;
		case 2304: bs_code = 4; break; case 4608: bs_code = 5; break; case 256: bs_code = 8; break;
		case 512: bs_code = 9; break; case 1024: bs_code = 10; break; case 2048: bs_code = 11; break;
		case 4096: bs_code = 12; break; case 8192: bs_code = 13; break; case 16384: bs_code = 14; break;
		case 32768: bs_code = 15; break;
		default: bs_hint = bs_code = (n <= 0x100) ? 6 : 7; break;
	}
	const uint32_t sr = P.sample_rate;
	switch(sr) {
		case 88200: sr_code = 1; break; case 176400: sr_code = 2; break; case 192000: sr_code = 3; break;
		case 8000: sr_code = 4; break; case 16000: sr_code = 5; break; case 22050: sr_code = 6; break;
		case 24000: sr_code = 7; break; case 32000: sr_code = 8; break; case 44100: sr_code = 9; break;
		case 48000: sr_code = 10; break; case 96000: sr_code = 11; break;
		default:
			if(sr <= 255000 && sr % 1000 == 0) sr_hint = sr_code = 12;
			else if(sr <= 655350 && sr % 10 == 0) sr_hint = sr_code = 14;
			else if(sr <= 0xffff) sr_hint = sr_code = 13;
			else sr_code = 0;
			break;
	}
	uint32_t bps_code;
	switch(P.bps) {
		case 8: bps_code = 1; break; case 12: bps_code = 2; break; case 16: bps_code = 4; break;
		case 20: bps_code = 5; break; case 24: bps_code = 6; break; case 32: bps_code = 7; break;
		default: bps_code = 0; break;
	}
	put(0xff); put(0xf8);
	put(((bs_code << 4) | sr_code));
	put((((ca == 0 ? C - 1 : 7 + ca) << 4) | (bps_code << 1)));
	{
		const uint32_t v = frame_number;   // bitwriter.c:832
		if(v < 0x80) put(v);
		else if(v < 0x800) { put((0xC0 | (v >> 6))); put((0x80 | (v & 0x3F))); }
		else if(v < 0x10000) { put((0xE0 | (v >> 12))); put((0x80 | ((v >> 6) & 0x3F))); put((0x80 | (v & 0x3F))); }
		else if(v < 0x200000) { put((0xF0 | (v >> 18))); put((0x80 | ((v >> 12) & 0x3F))); put((0x80 | ((v >> 6) & 0x3F))); put((0x80 | (v & 0x3F))); }
		else if(v < 0x4000000) { put((0xF8 | (v >> 24))); put((0x80 | ((v >> 18) & 0x3F))); put((0x80 | ((v >> 12) & 0x3F))); put((0x80 | ((v >> 6) & 0x3F))); put((0x80 | (v & 0x3F))); }
		else { put((0xFC | (v >> 30))); put((0x80 | ((v >> 24) & 0x3F))); put((0x80 | ((v >> 18) & 0x3F))); put((0x80 | ((v >> 12) & 0x3F))); put((0x80 | ((v >> 6) & 0x3F))); put((0x80 | (v & 0x3F))); }
	}
	if(bs_hint == 6) put((n - 1));
	else if(bs_hint == 7) { put(((n - 1) >> 8)); put((n - 1)); }
	if(sr_hint == 12) put((sr / 1000));
	else if(sr_hint == 13) { put((sr >> 8)); put(sr); }
	else if(sr_hint == 14) { put(((sr / 10) >> 8)); put((sr / 10)); }
	sink(nb, crc);
	return nb + 1;
}
// the same as four big-endian words of the frame image (hw[0] = bytes 0..3 ...; the header is 16 bytes at most: 4 + 6 + 2 + 2 + 1,
// zero behind its end), from wave-uniform inputs: every byte is appended to a 128-bit scalar accumulator, so the whole header is
// scalar code and the wavefront spends one select and one LDS store on it (a lane that builds it byte by byte into the image makes
// the whole wavefront issue a hundred vector instructions)
__device__ __forceinline__ uint32_t frame_header_words(const DevParams &P, uint32_t n, uint32_t ca, uint32_t frame_number, uint32_t (&hw)[4])
{
	uint64_t hi = 0, lo = 0;
	const uint32_t nb = frame_header_gen(P, n, ca, frame_number, [&](uint32_t, uint32_t byte) { hi = (hi << 8) | (lo >> 56); lo = (lo << 8) | (uint64_t)(byte & 0xffu); });
	// left-align the nb bytes in the 16
	const uint32_t sh = 8u * (16u - nb);                     // 24 .. 88 bits
	if(sh >= 64u) { hi = lo << (sh - 64u); lo = 0; }
	else { hi = (hi << sh) | (lo >> (64u - sh)); lo <<= sh; }
	hw[0] = (uint32_t)(hi >> 32); hw[1] = (uint32_t)hi; hw[2] = (uint32_t)(lo >> 32); hw[3] = (uint32_t)lo;
	return nb;
}
__device__ uint32_t frame_header_bytes(const DevParams &P, uint32_t n, uint32_t ca, uint32_t frame_number, uint8_t (&hb)[16])
{
	return frame_header_gen(P, n, ca, frame_number, [&](uint32_t k, uint32_t byte) { hb[k] = (uint8_t)byte; });
}

// CRC-16 of the first body_bytes of the frame image (see above); result valid in thread 0.  Ends with a barrier.
// a word of the frame image; an image in HBM was built with atomics at the L2, so the read must not be served by this CU's L1
__device__ __forceinline__ uint32_t img_word(const uint32_t *img, uint32_t w, bool global_img)
{
	return global_img ? __hip_atomic_load(img + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : img[w];
}
__device__ uint32_t frame_crc16(const uint32_t *img, uint32_t body_bytes, const uint16_t (*crc_tab)[256], uint32_t *crc_parts, int tid,
                                const uint16_t *xspan_lds = nullptr, uint32_t nxspan_lds = 0, const uint16_t *xbyte_lds = nullptr, bool global_img = false)
{
	// spans of 64 bytes; the last (possibly short) span is followed by nothing, span s by nsp-2-s whole spans and the last one:
	// crc = (sum over the whole spans of crc_s * x^(512 (nsp-2-s))) * x^(8 last_len) + crc_last -- the shift past the last span is
	// common to all whole spans and is applied once, after the reduction
	const uint32_t nsp = (body_bytes + CRC_SPAN - 1) / CRC_SPAN;
	const uint32_t last_len = body_bytes - (nsp ? nsp - 1 : 0) * CRC_SPAN;                 // 1..64 bytes
	uint32_t c = 0;                     // low half: whole spans, shifted among themselves; high half: the last span
	for(uint32_t sp = (uint32_t)tid; sp < nsp; sp += TPB) {
		const uint32_t *wp = img + sp * (CRC_SPAN / 4);
		uint32_t cs = 0;
		if(sp + 1 < nsp) {
#pragma unroll
			for(int k = 0; k < (int)(CRC_SPAN / 4); k++) {
				const uint32_t v = (cs << 16) ^ img_word(wp, (uint32_t)k, global_img);
				cs = (uint32_t)crc_tab[3][v >> 24] ^ crc_tab[2][(v >> 16) & 0xffu] ^ crc_tab[1][(v >> 8) & 0xffu] ^ crc_tab[0][v & 0xffu];
			}
			const uint32_t m = nsp - 2 - sp;
			const uint32_t xs = m < nxspan_lds ? xspan_lds[m] : g_crc_tables.xspan[m];
			c ^= m ? gf16_mul(cs, xs) : cs;
		}
		else {
			for(uint32_t k = 0; k < last_len; k++) {
				const uint32_t b = (img_word(wp, k >> 2, global_img) >> (24 - 8 * (k & 3))) & 0xffu;
				cs = ((cs << 8) & 0xffffu) ^ crc_tab[0][(cs >> 8) ^ b];
			}
			c ^= cs << 16;
		}
	}
#pragma unroll
	for(int off = 32; off >= 1; off >>= 1) c ^= __shfl_xor(c, off);
	if((tid & 63) == 0) crc_parts[tid >> 6] = c;
	__syncthreads();
	uint32_t crc = 0;
	for(int w = 0; w < TPB / 64; w++) crc ^= crc_parts[w];
	const uint32_t xb = xbyte_lds ? xbyte_lds[last_len] : g_crc_tables.xbyte[last_len];
	return gf16_mul(crc & 0xffffu, xb) ^ (crc >> 16);
}

// The same for pack2_kernel's LDS image: 44-byte spans (conflict-free reads, see CRC2_SPAN); every lane loads its eleven words
// at once and runs them under a predicate, the lane with the short last span included -- that span used to be a byte-serial
// chain of up to 64 dependent LDS round trips which the whole workgroup waited for at the barrier.
// SPANW: words per span, odd.  11 (the table g_crc_tables.xspan44 behind the LDS copy); ff_kernel: 13 with its own table, all of it in LDS
// (its frames of ~3 KB are 66..69 spans of 44 bytes -- a second pass of the wavefront for two to five lanes -- and 56..59 of 52)
template <int NT, int SPANW = (int)CRC2_WORDS>
__device__ __forceinline__ uint32_t frame_crc16_p2(const uint32_t *img, uint32_t body_bytes, const uint16_t (*crc_tab)[256], uint32_t *crc_parts, int tid,
                                                   const uint16_t *xspan_lds, uint32_t nxspan_lds, const uint16_t *xbyte_lds)
{
	constexpr uint32_t CRC2_SPAN = 4 * SPANW, CRC2_WORDS = SPANW;          // (shadow the 11-word constants)
	const uint32_t nsp = (body_bytes + CRC2_SPAN - 1) / CRC2_SPAN;
	const uint32_t last_len = body_bytes - (nsp ? nsp - 1 : 0) * CRC2_SPAN;                // 1..44 bytes
	uint32_t c = 0;                     // low half: whole spans, shifted among themselves; high half: the last span
	for(uint32_t sp = (uint32_t)tid; sp < nsp; sp += NT) {
		const uint32_t *wp = img + sp * CRC2_WORDS;
		uint32_t w[CRC2_WORDS];
#pragma unroll
		for(int k = 0; k < (int)CRC2_WORDS; k++) w[k] = wp[k];             // (past the last span: still inside the workgroup's LDS, never used)
		const bool last = sp + 1 == nsp;
		const uint32_t nw = last ? last_len >> 2 : CRC2_WORDS;
		uint32_t cs = 0, tailw = 0;
#pragma unroll
		for(int k = 0; k < (int)CRC2_WORDS; k++) {
			if((uint32_t)k < nw) {
				const uint32_t v = (cs << 16) ^ w[k];
				cs = (uint32_t)crc_tab[3][v >> 24] ^ crc_tab[2][(v >> 16) & 0xffu] ^ crc_tab[1][(v >> 8) & 0xffu] ^ crc_tab[0][v & 0xffu];
			}
			else if((uint32_t)k == nw) tailw = w[k];
		}
		if(last) {
			const uint32_t nb = last_len & 3u;
			for(uint32_t j = 0; j < nb; j++) {
				const uint32_t b = (tailw >> (24 - 8 * j)) & 0xffu;
				cs = ((cs << 8) & 0xffffu) ^ crc_tab[0][(cs >> 8) ^ b];
			}
			c ^= cs << 16;
		}
		else {
			const uint32_t m = nsp - 2 - sp;
			const uint32_t xs = (SPANW != 11 || m < nxspan_lds) ? xspan_lds[m] : g_crc_tables.xspan44[m];
			c ^= m ? gf16_mul(cs, xs) : cs;
		}
	}
#pragma unroll
	for(int off = 32; off >= 1; off >>= 1) c ^= __shfl_xor(c, off);
	if((tid & 63) == 0) crc_parts[tid >> 6] = c;
	__syncthreads();
	uint32_t crc = 0;
	for(int w = 0; w < NT / 64; w++) crc ^= crc_parts[w];
	return gf16_mul(crc & 0xffffu, xbyte_lds[last_len]) ^ (crc >> 16);
}

// Round 4's form of the same for pack2_kernel and ff_kernel: spans aligned to the END of the frame.  The body is padded with its
// zero bytes to a whole word (the image is zero behind it), spans of SPANW words are counted back from that word, and the first
// one reaches in front of the image, where zeros lie (P2_IMG_PAD: a CRC that starts at zero does not see leading zeros) -- so
// every span is whole, every lane runs the same SPANW unconditional steps, and there is neither a short last span nor a byte
// loop; the padding's factor x^(8 z) comes off at the end by its inverse (x is a unit modulo P: P(0) = 1).  A byte's table
// address is one SDWA shift (v_lshlrev_b32 with a byte select; the tables sit at a fixed LDS address, which goes into the
// instruction's offset field): 9 VALU instructions and 4 LDS reads per word where the round-3 form had 13 + predication
// (profiles/archive/r04_l_crc_ab.txt).
constexpr uint32_t crc_gfmul_c(uint32_t a, uint32_t b) { uint32_t r = 0; for(int i = 0; i < 16; i++) { r = crc_mulx(r); if(b & 0x8000u) r ^= a; b = (b << 1) & 0xffffu; } return r; }
constexpr uint32_t CRC_XINV1 = 0xC002u;                   // x^-1 = x^15 + x^14 + x  (x * that = x^16 + x^15 + x^2 = 1 mod P)
constexpr uint32_t crc_xinv8() { uint32_t r = 1; for(int i = 0; i < 8; i++) r = crc_gfmul_c(r, CRC_XINV1); return r; }
static_assert(crc_gfmul_c(2u, CRC_XINV1) == 1u, "x * x^-1 = 1 mod x^16+x^15+x^2+1");
constexpr uint32_t CRC_XINV8 = crc_xinv8(), CRC_XINV16 = crc_gfmul_c(CRC_XINV8, CRC_XINV8), CRC_XINV24 = crc_gfmul_c(CRC_XINV16, CRC_XINV8);
static_assert(crc_gfmul_c(CRC_XINV8, crc_mulx8(1)) == 1u, "x^8 * x^-8 = 1");
template <int B>
__device__ __forceinline__ uint32_t byte_times2(uint32_t v, uint32_t one)       // 2 * byte B of v
{
	uint32_t d;
	if constexpr(B == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(d) : "v"(one), "v"(v));
	if constexpr(B == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(d) : "v"(one), "v"(v));
	if constexpr(B == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(d) : "v"(one), "v"(v));
	if constexpr(B == 3) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(d) : "v"(one), "v"(v));
	return d;
}
// The four 256-entry tables (uint16_t [4][256]) sit at LDS ADDRESS 0 -- the start of the kernel's dynamic LDS, which has no static
// LDS in front of it (lds_base_is_zero() checks) --, so that a table address is the shifted byte itself and the table's offset
// goes into the instruction's offset field; img[-SPANW .. -1] must be readable zeros.  Ends with a barrier.
typedef __attribute__((address_space(3))) const uint16_t *lds_u16p;
__device__ __forceinline__ bool lds_base_is_zero(const unsigned char *smem) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char *)smem == 0u; }
// a * b modulo P for frame_crc16_end: the 31-bit carry-less product by Horner (a bit of b as a mask, one and-xor), then its high
// half folded back through the CRC tables at LDS address 0 (tab[0][v] = v x^16, tab[1][v] = v x^24 mod P): 3 instructions a bit
// and 6 for the fold, where the shift-and-reduce loop of gf16_mul takes 6 a bit
__device__ __forceinline__ uint32_t gf16_mul_tab(uint32_t a, uint32_t b, uint32_t one)
{
	uint32_t p = 0;
#pragma unroll
	for(int i = 15; i >= 0; i--) p = (p << 1) ^ (a & (uint32_t)__builtin_amdgcn_sbfe((int)b, i, 1));
	return (p & 0xffffu) ^ *(lds_u16p)(uintptr_t)byte_times2<2>(p, one) ^ *(lds_u16p)(uintptr_t)(byte_times2<3>(p, one) + 512);
}
template <int NT, int SPANW>
__device__ __forceinline__ uint32_t frame_crc16_end(const uint32_t *img, uint32_t body_bytes, uint32_t *crc_parts, int tid,
                                                    const uint16_t *xspan_lds, uint32_t nxspan_lds, const uint16_t *xspan_global)
{
	const uint32_t W = (body_bytes + 3) >> 2, z = W * 4 - body_bytes;
	const uint32_t nsp = (W + SPANW - 1) / SPANW;
	uint32_t one = 1;
	asm volatile("" : "+v"(one));                                            // (a register: the SDWA form takes no literal)
	uint32_t c = 0;
	for(uint32_t sp = (uint32_t)tid; sp < nsp; sp += NT) {
		const uint32_t *wp = img + (int32_t)W - (int32_t)(SPANW * (nsp - sp));
		uint32_t w[SPANW];
#pragma unroll
		for(int k = 0; k < SPANW; k++) w[k] = wp[k];
		uint32_t cs = 0;
#pragma unroll
		for(int k = 0; k < SPANW; k++) {
			const uint32_t v = (cs << 16) ^ w[k];
			cs = (uint32_t)*(lds_u16p)(uintptr_t)(byte_times2<3>(v, one) + 3 * 512) ^ *(lds_u16p)(uintptr_t)(byte_times2<2>(v, one) + 2 * 512)
			     ^ *(lds_u16p)(uintptr_t)(byte_times2<1>(v, one) + 512) ^ *(lds_u16p)(uintptr_t)byte_times2<0>(v, one);
		}
		const uint32_t m = nsp - 1 - sp;                                     // whole spans behind this one
		const uint32_t xs = m < nxspan_lds ? xspan_lds[m] : xspan_global[m];
		c ^= gf16_mul_tab(cs, xs, one);                                        // (the last span's factor is x^0 = 1)
	}
#pragma unroll
	for(int off = 32; off >= 1; off >>= 1) c ^= __shfl_xor(c, off);
	if(NT > 64) {
		if((tid & 63) == 0) crc_parts[tid >> 6] = c;
		__syncthreads();
		c = 0;
		for(int wv = 0; wv < NT / 64; wv++) c ^= crc_parts[wv];
	}
	else __syncthreads();
	return gf16_mul_tab(c, z == 0 ? 1u : z == 1 ? CRC_XINV8 : z == 2 ? CRC_XINV16 : CRC_XINV24, one);
}

// number of frame header bytes including the CRC-8, without building them (same cases as frame_header_bytes)
__device__ __forceinline__ uint32_t frame_header_len(const DevParams &P, uint32_t n, uint32_t v)
{
	uint32_t nb = 4 + 1;
	nb += v < 0x80 ? 1 : v < 0x800 ? 2 : v < 0x10000 ? 3 : v < 0x200000 ? 4 : v < 0x4000000 ? 5 : 6;
	const bool bs_std = n == 192 || n == 576 || n == 1152 || n == 2304 || n == 4608 || (n >= 256 && n <= 32768 && (n & (n - 1)) == 0);
	if(!bs_std) nb += n <= 0x100 ? 1 : 2;
	const uint32_t sr = P.sample_rate;
	const bool sr_std = sr == 88200 || sr == 176400 || sr == 192000 || sr == 8000 || sr == 16000 || sr == 22050 || sr == 24000 || sr == 32000 ||
	                    sr == 44100 || sr == 48000 || sr == 96000;
	if(!sr_std) {
		if(sr <= 255000 && sr % 1000 == 0) nb += 1;
		else if(sr <= 655350 && sr % 10 == 0) nb += 2;
		else if(sr <= 0xffff) nb += 2;
	}
	return nb;
}

// LDS bytes of the one-pass signal window of pack_kernel: 32 samples in front, CHUNK*TPB samples, 16 + 16 behind, 18-word rows
__host__ __device__ inline uint32_t pack_pass_sig_bytes(const DevParams &P)
{
	const uint32_t span = P.blocksize < (uint32_t)(CHUNK * TPB) ? ((P.blocksize + 15u) & ~15u) : (uint32_t)(CHUNK * TPB);
	const uint32_t maxidx = 32 + span + 16u + 16u;
	return (((maxidx + ((maxidx >> 4) << 1) + 8) * 4 + 15) & ~15u) * (P.chan_stride != P.blocksize ? 2u : 1u);     // 64-bit samples
}
// a sample of up to 33 bits, two's complement, MSB first
__device__ __forceinline__ void put_sample(uint32_t *buf, uint32_t cap_words, uint32_t pos, int64_t v, uint32_t len)
{
	if(len > 32) { put_bits(buf, cap_words, pos, (uint32_t)((uint64_t)v >> 32), len - 32); put_bits(buf, cap_words, pos + len - 32, (uint32_t)v, 32); }
	else put_bits(buf, cap_words, pos, (uint32_t)v, len);
}

struct PackShared {
	uint64_t scratch[8];
	uint8_t params[1u << MAX_PO];
	uint32_t scan[TPB / 64 + 1];
	uint32_t ca, left, right;
	uint32_t crc_parts[TPB / 64];
	uint16_t crc_tab[4][256];
	uint32_t bitpos;
	uint32_t overflow;
};

template <int MAXORD>
__global__ __launch_bounds__(TPB) void pack_kernel(const DevParams P, const int32_t *__restrict__ chan,
                                                   uint32_t nframes, uint32_t tail_n, uint32_t f_lo, uint64_t first_frame_number,
                                                   const SubDecision *__restrict__ decisions,
                                                   uint8_t *__restrict__ slots, uint32_t *__restrict__ frame_bytes,
                                                   FrameInfo *__restrict__ info)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int tid = (int)threadIdx.x;
	const uint32_t C = P.channels, N = P.blocksize;
	const uint32_t f = f_lo + blockIdx.x;                  // frames [f_lo, nframes)
	const bool is_tail = tail_n != 0 && f == nframes - 1;
	const uint32_t n = is_tail ? tail_n : N;
	const SubDecision *dec = decisions + (size_t)f * P.ncand;

	// LDS: the samples of ONE pass (4096 samples + 32 in front, 18-word rows) | frame image | small state
	int32_t *sig = (int32_t *)smem;
	const uint32_t sigb = pack_pass_sig_bytes(P);
	// (frames too large for the LDS are assembled in place in their HBM slot: same code, atomics at the L2)
	const bool gimg = P.img_global != 0;
	uint32_t *img = gimg ? (uint32_t *)(slots + (size_t)f * P.slot_bytes) : (uint32_t *)(smem + sigb);
	PackShared *sh = (PackShared *)(smem + sigb + (gimg ? 0 : P.slot_bytes));
	const uint32_t cap_words = P.slot_bytes / 4;

	for(uint32_t w = (uint32_t)tid; w < cap_words; w += TPB) img[w] = 0;
	for(uint32_t w = (uint32_t)tid; w < 4 * 256 / 2; w += TPB) ((uint32_t *)sh->crc_tab)[w] = ((const uint32_t *)g_crc_tables.tab)[w];
	if(tid == 0) {
		// channel assignment (stream_encoder.c:3944-3972)
		uint32_t ca = 0, left = 0, right = 1;
		if(P.ms_mode == 1) {
			const uint32_t b0 = dec[0].bits + dec[1].bits, b1 = dec[0].bits + dec[3].bits,
			               b2 = dec[1].bits + dec[3].bits, b3 = dec[2].bits + dec[3].bits;
			uint32_t mn = b0;
			if(b1 < mn) { mn = b1; ca = 1; }
			if(b2 < mn) { mn = b2; ca = 2; }
			if(b3 < mn) { mn = b3; ca = 3; }
			left = ca == 2 ? 3 : ca == 3 ? 2 : 0;
			right = ca == 0 ? 1 : ca == 2 ? 1 : 3;
		}
		else if(P.ms_mode == 2) {
			ca = dec[0].which >= 2 ? 3 : 0;
		}
		sh->ca = ca; sh->left = left; sh->right = right; sh->overflow = 0;
	}
	__syncthreads();
	const uint32_t ca = sh->ca;

	// ---- frame header (stream_encoder_framing.c:245-391), one lane -------------------------------
	if(tid == 0) {
		uint8_t hb[16];
		const uint32_t nb = frame_header_bytes(P, n, ca, (uint32_t)(first_frame_number + f), hb);
		for(uint32_t k = 0; k < nb; k++) put_bits(img, cap_words, 8 * k, hb[k], 8);
		sh->bitpos = 8 * nb;
	}
	__syncthreads();

	// ---- subframes (stream_encoder_framing.c:393-594) -------------------------------------------
	const uint32_t nsub = C;
	for(uint32_t s = 0; s < nsub; s++) {
		uint32_t di;   // decision index
		if(P.ms_mode == 1) di = s == 0 ? sh->left : sh->right;
		else di = s;
		const SubDecision *d = dec + di;
		const uint32_t which = d->which, type = d->type, order = d->order, wasted = d->wasted;
		const uint32_t sbps = P.bps - wasted + (which == C + 1 ? 1 : 0);
		uint32_t pos = sh->bitpos;
		__syncthreads();

		// the planar channel written by the prep kernels (wasted bits already shifted out, 16-bit pairs when sbps <= 16)
		// is staged one pass of CHUNK*TPB samples at a time: sample pass + k sits at sigidx(k), k = -32 .. CHUNK*TPB+16
		const uint32_t *src = (const uint32_t *)(chan + ((size_t)f * P.ncand + di) * P.chan_stride);
		const bool s64 = sbps > 32;                 // the 33-bit side channel of a 32-bit stream: 64-bit samples
		int64_t *sig64 = (int64_t *)smem;
		auto stage_pass = [&](uint32_t pass) {
			__syncthreads();
			const int span = (int)(N < (uint32_t)(CHUNK * TPB) ? ((N + 15u) & ~15u) : (uint32_t)(CHUNK * TPB));
			for(int k = tid - 32; k < span + 16; k += TPB) {
				const int64_t i = (int64_t)pass + k;
				if(s64) { sig64[sigidx(k)] = (i >= 0 && i < (int64_t)n) ? ((const int64_t *)src)[i] : 0; continue; }
				int32_t v = 0;
				if(i >= 0 && i < (int64_t)n) v = d->fmt == 1 ? (int32_t)((const int16_t *)src)[i] : (int32_t)src[i];
				sig[sigidx(k)] = v;
			}
			__syncthreads();
		};
		auto SV = [&](int k) -> int64_t { return s64 ? sig64[sigidx(k)] : (int64_t)sig[sigidx(k)]; };

		// subframe header byte (+ unary wasted bits)
		uint32_t type_bits = type == 0 ? 0x00u : type == 1 ? 0x02u : type == 2 ? (0x10u | (order << 1)) : (0x40u | ((order - 1) << 1));
		if(tid == 0) {
			put_bits(img, cap_words, pos, type_bits | (wasted ? 1u : 0u), 8);
			if(wasted) put_bits(img, cap_words, pos + 8 + (wasted - 1), 1, 1);
		}
		pos += 8 + wasted;

		if(type == 0) {
			if(tid == 0) put_sample(img, cap_words, pos, ((int64_t)d->constant_hi << 32) | (int64_t)(uint32_t)d->constant, sbps);
			pos += sbps;
		}
		else if(type == 1) {
			for(uint32_t pass = 0; pass < n; pass += CHUNK * TPB) {
				stage_pass(pass);
				for(uint32_t k = (uint32_t)tid; k < CHUNK * TPB && pass + k < n; k += TPB) put_sample(img, cap_words, pos + (pass + k) * sbps, SV((int)k), sbps);
			}
			pos += n * sbps;
		}
		else {
			// warm-up (written when pass 0 is staged), (precision, shift, coefficients), entropy coding header
			const uint32_t warm_pos = pos;
			pos += order * sbps;
			const int shift = type == 3 ? d->shift : 0;
			bool wide = sbps + order > 32;                 // FIXED: 64-bit differences (stream_encoder.c:4511-4516)
			if(type == 3) {
				const uint32_t precision = d->precision;
				if(tid == 0) {
					put_bits(img, cap_words, pos, precision - 1, 4);
					put_bits(img, cap_words, pos + 4, (uint32_t)shift, 5);
				}
				if((uint32_t)tid < order) put_bits(img, cap_words, pos + 9 + (uint32_t)tid * precision, (uint32_t)d->q[tid], precision);
				pos += 9 + order * precision;
				uint32_t abs_sum = 0;
				for(uint32_t i = 0; i < order; i++) abs_sum += (uint32_t)abs(d->q[i]);
				wide = silog2_i64((int64_t)(((uint64_t)1 << (sbps - 1)) * abs_sum)) > 32;
			}
			const uint32_t po = d->po, plen = d->rice2 ? 5u : 4u;
			if(tid == 0) {
				put_bits(img, cap_words, pos, d->rice2 ? 1u : 0u, 2);
				put_bits(img, cap_words, pos + 2, po, 4);
			}
			pos += 6;
			// taps
			int32_t q[MAXORD];
#pragma unroll
			for(int j = 0; j < MAXORD; j++) {
				int32_t c = 0;
				if(type == 3) c = j < MAX_ORDER ? d->q[j] : 0;
				else if(order == 1) c = j == 0 ? 1 : 0;
				else if(order == 2) c = j == 0 ? 2 : j == 1 ? -1 : 0;
				else if(order == 3) c = j == 0 ? 3 : j == 1 ? -3 : j == 2 ? 1 : 0;
				else if(order == 4) c = j == 0 ? 4 : j == 1 ? -6 : j == 2 ? 4 : j == 3 ? -1 : 0;
				q[j] = c;
			}
			const uint32_t psize = n >> po;
			for(uint32_t p = (uint32_t)tid; p < (1u << po); p += TPB) sh->params[p] = d->params[p];
			__syncthreads();
			// residual bits: per-thread chunk lengths -> exclusive scan -> write
			for(uint32_t pass = 0; pass < n; pass += CHUNK * TPB) {
				const uint32_t base = pass + CHUNK * (uint32_t)tid;
				int32_t r[CHUNK];
				uint32_t mybits = 0;
				stage_pass(pass);
				if(pass == 0 && (uint32_t)tid < order) put_sample(img, cap_words, warm_pos + (uint32_t)tid * sbps, SV(tid), sbps);
				if(base < n) {
					const SigRef sref = {(const void *)smem, s64 ? 1u : 0u, 0u, 0u};
					(void)fir_chunk_dispatch<MAXORD>(sref, (int)(CHUNK * (uint32_t)tid), q, shift, fir_mode(wide ? 1u : 0u, sbps), r, 0, 0);
					uint32_t part = base / psize, next = (part + 1) * psize;
					uint32_t k = sh->params[part];
#pragma unroll
					for(int t = 0; t < CHUNK; t++) {
						const uint32_t i = base + t;
						if(i == next) { part++; next += psize; if(i < n) k = sh->params[part]; }
						if(i < n && i >= order) {
							if(i == (part == 0 ? order : part * psize)) mybits += plen;
							const uint32_t u = ((uint32_t)r[t] << 1) ^ (uint32_t)(r[t] >> 31);
							mybits += (u >> k) + 1 + k;
						}
					}
				}
				// workgroup exclusive scan of mybits
				uint32_t incl = mybits;
#pragma unroll
				for(int off = 1; off < 64; off <<= 1) { uint32_t t = __shfl_up(incl, off); if((tid & 63) >= off) incl += t; }
				__syncthreads();
				if((tid & 63) == 63) sh->scan[tid >> 6] = incl;
				__syncthreads();
				uint32_t wave_off = 0, total = 0;
				for(int w = 0; w < TPB / 64; w++) { if(w < (tid >> 6)) wave_off += sh->scan[w]; total += sh->scan[w]; }
				uint32_t p = pos + wave_off + incl - mybits;
				if(base < n) {
					uint32_t part = base / psize, next = (part + 1) * psize;
					uint32_t k = sh->params[part];
#pragma unroll
					for(int t = 0; t < CHUNK; t++) {
						const uint32_t i = base + t;
						if(i == next) { part++; next += psize; if(i < n) k = sh->params[part]; }
						if(i < n && i >= order) {
							if(i == (part == 0 ? order : part * psize)) { put_bits(img, cap_words, p, k, plen); p += plen; }
							const uint32_t u = ((uint32_t)r[t] << 1) ^ (uint32_t)(r[t] >> 31);
							const uint32_t msbs = u >> k;
							put_bits(img, cap_words, p + msbs, (1u << k) | (u & ((1u << k) - 1u)), k + 1);
							p += msbs + 1 + k;
						}
					}
				}
				pos += total;
				__syncthreads();
			}
		}
		if(tid == 0) {
			sh->bitpos = pos;
			if(info) {
				flacgpu_subframe_info *si = &info[f].sub[s];
				si->type = (uint8_t)type; si->order = (uint8_t)order; si->wasted_bits = (uint8_t)wasted;
				si->partition_order = d->po; si->rice2 = d->rice2; si->precision = d->precision; si->shift = d->shift;
				si->pad = 0; si->bits = d->bits;
			}
		}
		__syncthreads();
	}

	// ---- zero-pad to a byte, CRC-16 over the whole frame, footer (stream_encoder.c:3720-3734) --------
	const uint32_t body_bytes = (sh->bitpos + 7) >> 3;
	const uint32_t total_bytes = body_bytes + 2;
	if(total_bytes > P.slot_bytes) { if(tid == 0) { sh->overflow = 1; } }
	__syncthreads();
	{
		const uint32_t crc = frame_crc16(img, body_bytes, sh->crc_tab, sh->crc_parts, tid, nullptr, 0, nullptr, gimg);
		if(tid == 0) put_bits(img, cap_words, body_bytes * 8, crc, 16);
		__syncthreads();
	}
	// ---- store: image words are big-endian views, slots are byte arrays ---------------------------
	{
		uint32_t *dst = (uint32_t *)(slots + (size_t)f * P.slot_bytes);
		const uint32_t words = (umin32(total_bytes, P.slot_bytes) + 3) >> 2;
		for(uint32_t w = (uint32_t)tid; w < words; w += TPB) dst[w] = __builtin_bswap32(img_word(img, w, gimg));      // (in place when the image is the slot)
		if(tid == 0) {
			frame_bytes[f] = sh->overflow ? 0xffffffffu : total_bytes;
			if(info) info[f].channel_assignment = (uint8_t)ca;
		}
	}
}


// ---------------------------------------------------------------------------------------------
// pack2_kernel: frames of nominal length whose partitions are whole 16-sample runs
// ---------------------------------------------------------------------------------------------
// One workgroup per frame; thread t owns samples [16t, 16t+16) of every subframe (per 4096-sample pass).  It loads its
// window of the planar channel straight from global memory (32 bytes of packed history + 32 bytes of its own samples:
// no LDS staging, no barrier), recomputes the winning residual with the same dot2 / mad24 chains as the evaluation
// kernel, sizes its 16 Rice codes, takes its bit offset from a DPP wavefront scan plus one LDS hop, and ORs the codes
// into the zeroed LDS frame image.  Only the non-zero part of a code is written (stop bit + low bits): the unary zeros
// are already there.
__device__ __forceinline__ uint32_t wave_scan_incl_dpp(uint32_t v)
{
	uint32_t d;
	asm("s_nop 1\n\t"
	    "v_add_u32_dpp %0, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
	    "v_add_u32_dpp %0, %1, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
	    "v_add_u32_dpp %0, %1, %0 row_shr:3 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
	    "s_nop 1\n\t"
	    "v_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xe\n\t"
	    "s_nop 1\n\t"
	    "v_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xc\n\t"
	    "s_nop 1\n\t"
	    "v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
	    "s_nop 1\n\t"
	    "v_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
	    : "=&v"(d) : "v"(v));
	return d;
}
// OR `len` bits (1..32) of v into the image at bit `pos`; positions past the image go to two dump words behind it
__device__ __forceinline__ void or_bits(uint32_t *buf, uint32_t cap_words, uint32_t pos, uint32_t v, uint32_t len)
{
	const uint32_t w = umin32(pos >> 5, cap_words), o = pos & 31;
	const uint64_t t = ((uint64_t)(v << (32 - len)) << 32) >> o;
	atomicOr(&buf[w], (uint32_t)(t >> 32));
	atomicOr(&buf[w + 1], (uint32_t)t);
}

// The same for a code inside a run that is known to end inside the image (no clamp), from 32-bit shifts that take their count
// from the low five bits of pos themselves: hi = X >> o, lo = {X, 0} >> o, X = the code left-aligned in a word
__device__ __forceinline__ void or_code_fit(uint32_t *buf, uint32_t pos, uint32_t x_left)
{
	uint32_t *w = (uint32_t *)((unsigned char *)buf + ((pos >> 3) & ~3u));
	atomicOr(w, x_left >> (pos & 31));
	atomicOr(w + 1, __builtin_amdgcn_alignbit(x_left, 0u, pos & 31));
}

// The same with the image at a known LDS byte address (the workgroup's LDS starts at 0, lds_base_is_zero: pack2_kernel, ff_kernel):
// the constant goes into the instruction's offset field instead of being added to every word address
typedef __attribute__((address_space(3))) uint32_t *lds_u32p;
__device__ __forceinline__ void or_code_abs(uint32_t img_abs, uint32_t pos, uint32_t x_left)
{
	lds_u32p w = (lds_u32p)(uintptr_t)(((pos >> 3) & ~3u) + img_abs);
	(void)__hip_atomic_fetch_or(w, x_left >> (pos & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	(void)__hip_atomic_fetch_or(w + 1, __builtin_amdgcn_alignbit(x_left, 0u, pos & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

constexpr uint32_t P2_XSPAN = 512;           // span shifts kept in LDS (frames up to 22 KiB; longer ones read the global table)
constexpr uint32_t P2_IMG_PAD = 64;          // zero bytes in front of the frame image (frame_crc16_end reads up to a span in front of it)
struct Pack2Shared {
	uint16_t crc_tab[4][256];                              // (first: at a fixed LDS address, see frame_crc16_end)
	uint32_t wtot[2][TPB / 64];
	uint32_t crc_parts[TPB / 64];
	uint16_t xspan[P2_XSPAN];
	uint16_t xbyte[CRC_SPAN + 2];
	uint32_t placed;           // fused output: the frames in front turned up in time, and
	uint32_t excl_lo, excl_hi; // the byte offset of this frame in the stream of the batch
};

// ---- fused output: every frame is written once, at its final place in the stream of the batch ---------------------------------
// A frame's place is the sum of the lengths of all frames in front of it.  Round 2 tried the textbook single-pass scan with
// decoupled look-back (Merrill & Garland) at the END of the pack workgroups and lost (pack 0.45 -> 0.60 ms): thousands of
// workgroups of one dispatch round reach their look-back together, none of them has a predecessor that already knows its prefix,
// and the prefix then travels through them as a chain.  This version has no chain:
//   * a frame PUBLISHES its length the moment it is known -- before the CRC and the store: a tagged word of its own, and one
//     64-bit atomic add of {1, length} to a counter of its SEGMENT of 64 frames that nobody polls;
//   * whoever finds 63 arrivals in front of it in that counter has the segment's total in the value that came back, and
//     publishes it (tagged) -- also long before anybody asks;
//   * at its end a frame adds up, with all its loads in flight at once and nothing to wait for in the normal case, the lengths in
//     front of it in its own segment and the totals of the segments in front, 64 per step, nearest first.  The first frame of a
//     segment leaves its offset behind as the segment's START: a walk that meets a known start stops there, so however long the
//     batch, a frame looks at one or two rows of 64 totals.
// What it buys depends on the kernel (profiles/archive/r04_g_fused_ab.txt, r04_h_ff_lag_ab.txt): pack2_kernel (-3 .. -8: five workgroups
// of four wavefronts per CU, a frame's length known four fifths through) 0.26 + 0.09 ms of scan and compaction -> 0.30 ms;
// ff_kernel (-0 .. -2: one wavefront per frame, 26 us from load to store) 0.106 + 0.039 -> 0.157 ms -- a hop between two CUs is
// 2-5 us under load (MI355X_MICROARCH.md, handoff-1to1), and a wavefront that waits for two of them at the end of 26 us is a SIMD
// with three wavefronts instead of four for a fifth of the time.  So ff_kernel keeps its slots and the two small kernels by
// default; what it can do instead (launch_ff) is place, from each wavefront, the frame of 2048 frames ago.
//   (Two versions that did not survive: frames taken in dispatch order off an atomic ticket -- one word saturates at ~88 M
//   fetch-adds/s, 0.19 ms for the 16384 workgroups of a -0 batch; and the segment totals as atomic accumulators that the readers
//   poll -- readers and atomics fight for the lines: pack2_kernel 0.30 -> 0.88 ms.)
// Nothing here depends on the order or the placement of the workgroups for its RESULT, and nothing for its termination either:
// a frame waits only for words that workgroups with lower indices publish without waiting for anybody -- on this chip they were
// dispatched earlier (observed, not promised) --, and every wait is bounded: a frame whose predecessors do not turn up in time
// (PackOut::spin_limit polls) goes to its SLOT, as without the fused output, and onto a list; fo_place_kernel behind the pack kernel
// places the listed frames (in a normal run: none, and the kernel is a handful of idle workgroups).
// Words are tagged with the batch's epoch instead of being zeroed per batch: [63:40] epoch (never 0), [39:0] value.
constexpr uint64_t FO_VAL = (1ull << 40) - 1;
// a frame's word, bit 39: its bytes are in its slot as well, for whoever places it from there (ff_kernel with PackOut::lag)
constexpr uint64_t FO_STORED = 1ull << 39, FO_LEN = FO_STORED - 1;
struct PackOut {
	uint8_t *out;              // frames back to back (null: every frame goes to its slot)
	uint64_t cap;
	uint64_t *offsets;         // [nframes + 1]
	uint64_t *total;
	uint64_t *fstate;          // [frames] tagged length of every frame (null: nothing is published, scan + compact kernels follow)
	uint64_t *sstate;          // [segments] tagged total of every segment of 64 frames
	uint64_t *sprefix;         // [segments] tagged offset of a segment's first frame
	uint64_t *scount;          // [segments] {arrivals, bytes} so far (whoever completes a segment zeroes it again)
	uint32_t *fall;            // [frames] the frames that gave up waiting and went to their slots
	uint32_t *nfall;           // [4] how many, by epoch parity (fo_place_kernel zeroes the other one for the next batch); [2] all of them since the
	                           // context was created, [3] the most of any one batch
	uint32_t epoch;            // 1 .. 2^24 - 1
	uint32_t spin_limit;       // polls before a frame gives up waiting for the frames in front of it
	uint32_t lag;              // ff_kernel: the wavefront of frame f places frame f - lag (0: nobody places anything inside the kernel)
};
__device__ __forceinline__ uint64_t fo_word(uint32_t epoch, uint64_t v) { return ((uint64_t)epoch << 40) | v; }
__device__ __forceinline__ bool fo_ready(uint64_t w, uint32_t epoch) { return (uint32_t)(w >> 40) == epoch; }
__device__ __forceinline__ uint64_t fo_load(const uint64_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void fo_store(uint64_t *p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// one lane: the frame's length is known.  Returns the segment's counter as it was in front of this frame.
__device__ __forceinline__ uint64_t fo_publish(const PackOut &O, uint32_t f, uint32_t bytes)
{
	fo_store(&O.fstate[f], fo_word(O.epoch, bytes));
	return __hip_atomic_fetch_add(&O.scount[f >> 6], (1ull << 40) | (uint64_t)bytes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the same lane, later (the counter's value has had time to come back): the frame that completes its segment publishes the total
__device__ __forceinline__ void fo_close_segment(const PackOut &O, uint32_t f, uint32_t nmain, uint32_t bytes, uint64_t before)
{
	const uint32_t seg = f >> 6, members = umin32(64u, nmain - seg * 64u);
	if((uint32_t)(before >> 40) + 1u != members) return;
	fo_store(&O.sstate[seg], fo_word(O.epoch, (before & FO_VAL) + bytes));
	fo_store(&O.scount[seg], 0ull);
}
// a whole wavefront, at the end of frame f: the sum of the lengths of all frames in front of it; false: they did not turn up in time
// WITH_OWN: the frame's own length is asked for as well (the deferred compaction of ff_kernel places somebody else's frame)
template <bool WITH_OWN = false>
__device__ __forceinline__ bool fo_exclusive(const PackOut &O, uint32_t f, int lane, uint64_t &excl_out, uint32_t *own_bytes = nullptr)
{
	const uint32_t seg = f >> 6, within = f & 63u, ep = O.epoch, nrow = within + (WITH_OWN ? 1u : 0u);
	const uint64_t none = fo_word(ep, 0);
	uint64_t a, b, c;
	int64_t s0 = (int64_t)seg - 1;
	uint64_t excl = 0;
	uint32_t polls = 0;
	// the frames in front of this one in its segment, and the first row of segments in front of it: one round of loads
	bool first_row = true;
	for(;;) {
		const int64_t idx = s0 - lane;               // this lane's segment of the row (below 0: in front of the batch -- a known start of 0)
		a = first_row && (uint32_t)lane < nrow ? fo_load(&O.fstate[(size_t)seg * 64 + (uint32_t)lane]) : none;
		b = idx >= 0 ? fo_load(&O.sstate[idx]) : none;
		c = idx >= 0 ? fo_load(&O.sprefix[idx]) : none;
		uint32_t j;
		for(;;) {
			const uint64_t cm = __ballot((int)fo_ready(c, ep)), bm = __ballot((int)fo_ready(b, ep));
			j = cm ? (uint32_t)(__ffsll((unsigned long long)cm) - 1) : 64u;      // the nearest segment whose start is known
			const uint64_t need = j >= 63u ? ~0ull : ((2ull << j) - 1ull);
			// (WITH_OWN: the frame itself must also have its bytes in its slot)
			const bool a_ok = fo_ready(a, ep) && !(WITH_OWN && first_row && (uint32_t)lane == within && !(a & FO_STORED));
			if((bm & need) == need && !__any((int)!a_ok)) break;
			if(polls++ >= O.spin_limit) return false;
			__builtin_amdgcn_s_sleep(8);
			// (only the words that are missing are asked for again: thousands of pollers on whole rows are traffic of their own)
			if(!a_ok) a = fo_load(&O.fstate[(size_t)seg * 64 + (uint32_t)lane]);
			if(!fo_ready(b, ep)) b = fo_load(&O.sstate[idx]);
			if(!fo_ready(c, ep) && (uint32_t)lane <= 4u) c = fo_load(&O.sprefix[idx]);
		}
		if(WITH_OWN && first_row) *own_bytes = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)a, (int)within);      // (a length has 32 bits)
		uint64_t v = (uint32_t)lane < within ? a & FO_LEN : 0ull;
		if((uint32_t)lane <= j) v += b & FO_VAL;
		if((uint32_t)lane == j) v += c & FO_VAL;
		excl += wave_reduce_add_u64(v);
		if(j < 64u) break;
		s0 -= 64; first_row = false;
	}
	if(within == 0 && lane == 0) fo_store(&O.sprefix[seg], fo_word(ep, excl));
	excl_out = excl;
	return true;
}
// 16 bytes straight to / from memory (sc0 sc1: write-through stores, loads that no cache of this CU serves): what a frame that
// another CU will read inside the same launch is stored and read with (MI355X_MICROARCH.md: "sc0 sc1 stores and loads both sides")
typedef uint32_t fo_u4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) volatile fo_u4 *fo_gv4;
typedef __attribute__((address_space(1))) const volatile fo_u4 *fo_cgv4;
typedef __attribute__((address_space(1))) const volatile uint32_t *fo_cgv1;

// a frame as it lies in its slot (little-endian bytes, 16-byte aligned, written by another wavefront of this launch) to byte
// address dst, whatever its alignment: a lane takes 12 words per pass -- three 16-byte loads and the word behind them -- shifts
// them into the destination's word grid and stores three times 16 bytes; head and tail bytes one by one
__device__ __forceinline__ void fo_copy_slot(const uint8_t *src, uint8_t *dst, uint32_t nb, int lane)
{
	const uint32_t head = umin32((uint32_t)((4 - ((uintptr_t)dst & 3)) & 3), nb);
	if((uint32_t)lane < head) dst[lane] = (uint8_t)(*(fo_cgv1)(src) >> (8 * (uint32_t)lane));
	const uint32_t words = (nb - head) >> 2, shb = head;               // destination word w = source bytes [head + 4 w, head + 4 w + 4)
	uint32_t *dw = (uint32_t *)(dst + head);
	for(uint32_t w0 = 12u * (uint32_t)lane; w0 < words; w0 += 12u * 64u) {
		const fo_u4 a = *(fo_cgv4)(src + 4 * w0), b = *(fo_cgv4)(src + 4 * w0 + 16), c = *(fo_cgv4)(src + 4 * w0 + 32);
		const uint32_t e = *(fo_cgv1)(src + 4 * w0 + 48);                // (inside the slot: slots end 16 bytes behind the longest frame)
		const uint32_t in[13] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, e};
		uint32_t o[12];
#pragma unroll
		for(int k = 0; k < 12; k++) o[k] = __builtin_amdgcn_alignbyte(in[k + 1], in[k], shb);
		if(w0 + 12 <= words) {
			uint4 *d4 = (uint4 *)(dw + w0);                                // (4-byte aligned: the hardware takes dword-aligned 16-byte stores)
			d4[0] = make_uint4(o[0], o[1], o[2], o[3]); d4[1] = make_uint4(o[4], o[5], o[6], o[7]); d4[2] = make_uint4(o[8], o[9], o[10], o[11]);
		}
		else {
#pragma unroll
			for(int k = 0; k < 12; k++) if(w0 + (uint32_t)k < words) dw[w0 + k] = o[k];
		}
	}
	const uint32_t done = head + words * 4;
	if((uint32_t)lane < nb - done) { const uint32_t k = done + (uint32_t)lane; dst[k] = (uint8_t)(*(fo_cgv1)(src + (k & ~3u)) >> (8 * (k & 3u))); }
}

// the finished frame image (big-endian word views in LDS) to byte address dst, whatever its alignment
template <int NT = TPB>
__device__ __forceinline__ void store_image(const uint32_t *img, uint8_t *dst, uint32_t nb, int tid)
{
	const uint32_t head = umin32((uint32_t)((4 - ((uintptr_t)dst & 3)) & 3), nb);
	if((uint32_t)tid < head) dst[tid] = (uint8_t)(img[0] >> (24 - 8 * tid));
	const uint32_t words = (nb - head) >> 2, sh = head * 8;
	uint32_t *dw = (uint32_t *)(dst + head);
	for(uint32_t w = (uint32_t)tid; w < words; w += NT) {
		const uint32_t be = sh ? (img[w] << sh) | (img[w + 1] >> (32 - sh)) : img[w];
		dw[w] = __builtin_bswap32(be);
	}
	const uint32_t done = head + words * 4;
	if((uint32_t)tid < nb - done) { const uint32_t k = done + (uint32_t)tid; dst[k] = (uint8_t)(img[k >> 2] >> (24 - 8 * (k & 3))); }
}

// dot2_chain_lshr (flacgpu_devfn.h) with the tap pairs in scalar registers: they are the same for the whole subframe
template <int NP>
__device__ __forceinline__ uint32_t dot2_chain_lshr_s(const uint32_t (&W)[NP], const uint32_t (&Q)[8], int32_t sum0, uint32_t shift)
{
	uint32_t d;
	static_assert(NP == 2 || NP == 4 || NP == 6 || NP == 8, "instantiated pair counts");
	if constexpr(NP == 2) asm("v_dot2_i32_i16 %0, %2, %3, %1\n\tv_dot2_i32_i16 %0, %4, %5, %0\n\tv_lshrrev_b32 %0, %6, %0" : "=&v"(d) : "v"(sum0), "v"(W[0]), "s"(Q[0]), "v"(W[1]), "s"(Q[1]), "s"(shift));
	if constexpr(NP == 4) asm("v_dot2_i32_i16 %0, %2, %3, %1\n\tv_dot2_i32_i16 %0, %4, %5, %0\n\tv_dot2_i32_i16 %0, %6, %7, %0\n\tv_dot2_i32_i16 %0, %8, %9, %0\n\tv_lshrrev_b32 %0, %10, %0" : "=&v"(d) : "v"(sum0), "v"(W[0]), "s"(Q[0]), "v"(W[1]), "s"(Q[1]), "v"(W[2]), "s"(Q[2]), "v"(W[3]), "s"(Q[3]), "s"(shift));
	if constexpr(NP == 6) asm("v_dot2_i32_i16 %0, %2, %3, %1\n\tv_dot2_i32_i16 %0, %4, %5, %0\n\tv_dot2_i32_i16 %0, %6, %7, %0\n\tv_dot2_i32_i16 %0, %8, %9, %0\n\tv_dot2_i32_i16 %0, %10, %11, %0\n\tv_dot2_i32_i16 %0, %12, %13, %0\n\tv_lshrrev_b32 %0, %14, %0" : "=&v"(d) : "v"(sum0), "v"(W[0]), "s"(Q[0]), "v"(W[1]), "s"(Q[1]), "v"(W[2]), "s"(Q[2]), "v"(W[3]), "s"(Q[3]), "v"(W[4]), "s"(Q[4]), "v"(W[5]), "s"(Q[5]), "s"(shift));
	if constexpr(NP == 8) asm("v_dot2_i32_i16 %0, %2, %3, %1\n\tv_dot2_i32_i16 %0, %4, %5, %0\n\tv_dot2_i32_i16 %0, %6, %7, %0\n\tv_dot2_i32_i16 %0, %8, %9, %0\n\tv_dot2_i32_i16 %0, %10, %11, %0\n\tv_dot2_i32_i16 %0, %12, %13, %0\n\tv_dot2_i32_i16 %0, %14, %15, %0\n\tv_dot2_i32_i16 %0, %16, %17, %0\n\tv_lshrrev_b32 %0, %18, %0" : "=&v"(d) : "v"(sum0), "v"(W[0]), "s"(Q[0]), "v"(W[1]), "s"(Q[1]), "v"(W[2]), "s"(Q[2]), "v"(W[3]), "s"(Q[3]), "v"(W[4]), "s"(Q[4]), "v"(W[5]), "s"(Q[5]), "v"(W[6]), "s"(Q[6]), "v"(W[7]), "s"(Q[7]), "s"(shift));
	return d;
}
// residuals of this thread's 16 samples from the packed window A[0..15] (A[0..7] = the 16 samples in front); Q[p] = taps 2p, 2p+1 as
// an int16 pair, wave-uniform
// RUN: samples of the run (16, or 18 for the 1152-sample blocks: an even number, so a run is whole words of pairs)
template <int NP, int RUN = CHUNK>
__device__ __forceinline__ void pack_fir_packed(const uint32_t (&A)[8 + RUN / 2], const uint32_t (&Q)[8], uint32_t shift, int32_t (&r)[RUN])
{
	uint32_t B[7 + RUN / 2];
#pragma unroll
	for(int m = 0; m < 7 + RUN / 2; m++) B[m] = __builtin_amdgcn_alignbit(A[m + 1], A[m], 16);
	const uint32_t bias = 0x80000000u >> shift;
	const int32_t sum0 = (int32_t)0x80000000;
#pragma unroll
	for(int s = 0; s < RUN; s++) {
		const int u = 16 + s;
		uint32_t W[NP];
#pragma unroll
		for(int p = 0; p < NP; p++) W[p] = (u & 1) ? B[(u - 3) / 2 - p] : A[(u - 2) / 2 - p];
		const uint32_t pb = dot2_chain_lshr_s<NP>(W, Q, sum0, shift);                   // prediction + bias (see flacgpu_devfn.h)
		const uint32_t xb = bias + (uint32_t)((u & 1) ? ((int32_t)A[u / 2] >> 16) : (int32_t)(int16_t)(A[u / 2] & 0xffffu));
		r[s] = (int32_t)(xb - pb);
	}
}
// the same from 32-bit samples x[0..31] (x[16+s] = own sample s): FMODE 0 v_mad_i32_i24, 1 32-bit multiplies (lpc.c:321),
// 2 64-bit accumulate (lpc.c:582)
template <int NT, int FMODE, int RUN = CHUNK>
__device__ __forceinline__ void pack_fir_i32(const int32_t (&x)[16 + RUN], const int32_t *qg, int shift, int32_t (&r)[RUN])
{
	int32_t q[NT];
#pragma unroll
	for(int j = 0; j < NT; j++) q[j] = qg[j];
#pragma unroll
	for(int s = 0; s < RUN; s++) {
		if(FMODE == 0) {
			uint32_t pb;
			if(NT == 16) pb = mad24_chain<8, true>(&x[16 + s - 9], &q[8], mad24_chain<8, false>(&x[16 + s - 1], &q[0], 0, 0), (uint32_t)shift);
			else pb = mad24_chain<NT == 16 ? 8 : NT, true>(&x[16 + s - 1], &q[0], 0, (uint32_t)shift);
			r[s] = x[16 + s] - (int32_t)(pb ^ 0x80000000u);
		}
		else if(FMODE == 1) {
			uint32_t sum = 0;
#pragma unroll
			for(int j = 0; j < NT; j++) sum += (uint32_t)q[j] * (uint32_t)x[16 + s - 1 - j];
			r[s] = (int32_t)((uint32_t)x[16 + s] - (uint32_t)((int32_t)sum >> shift));
		}
		else {
			int64_t sum = 0;
#pragma unroll
			for(int j = 0; j < NT; j++) sum += (int64_t)q[j] * (int64_t)x[16 + s - 1 - j];
			r[s] = (int32_t)((int64_t)x[16 + s] - (sum >> shift));
		}
	}
}

// HINTS: the instantiation that also leaves its run starts behind for the verify pass (four more registers, which cost the
// 8-tap instance its fifth workgroup per CU: only paid when verification is on)
// NT: threads of the workgroup = 16-sample runs per pass: 256 for blocks of up to 4096 samples, 128 for the 1152-sample blocks of
// -0 .. -2 (72 runs: 256 threads would idle three in four)
// RUN: samples a thread owns per pass.  16; 18 for the 1152-sample blocks of -0 .. -2 (64 runs: ONE wavefront per frame, every lane
// busy, instead of 72 runs on two wavefronts of which the second has eight lanes to do; without the verify hints, whose decoder
// counts in 16-sample runs)
// The frame header of a frame of nominal length, split into what the stream fixes (computed on the host per launch: the cases
// of frame_header_gen above) and what the frame brings (channel assignment, frame number, CRC-8)
struct HdrConst {
	uint32_t b2;               // block size code << 4 | sample rate code
	uint32_t bps_code2;        // sample size code << 1
	uint32_t tail;             // the bytes behind the frame number -- block size, sample rate: 0..4 of them -- left-aligned
	uint32_t tail_n;
};
static HdrConst make_hdr_const(const DevParams &P)
{
	HdrConst h;
	const uint32_t n = P.blocksize, sr = P.sample_rate;
	uint32_t bs_code, bs_hint = 0, sr_code, sr_hint = 0, bps_code;
	switch(n) {
		case 192: bs_code = 1; break; case 576: bs_code = 2; break; case 1152: bs_code = 3; break;
		case 2304: bs_code = 4; break; case 4608: bs_code = 5; break; case 256: bs_code = 8; break;
		case 512: bs_code = 9; break; case 1024: bs_code = 10; break; case 2048: bs_code = 11; break;
		case 4096: bs_code = 12; break; case 8192: bs_code = 13; break; case 16384: bs_code = 14; break;
		case 32768: bs_code = 15; break;
		default: bs_hint = bs_code = (n <= 0x100) ? 6 : 7; break;
	}
	switch(sr) {
		case 88200: sr_code = 1; break; case 176400: sr_code = 2; break; case 192000: sr_code = 3; break;
		case 8000: sr_code = 4; break; case 16000: sr_code = 5; break; case 22050: sr_code = 6; break;
		case 24000: sr_code = 7; break; case 32000: sr_code = 8; break; case 44100: sr_code = 9; break;
		case 48000: sr_code = 10; break; case 96000: sr_code = 11; break;
		default:
			if(sr <= 255000 && sr % 1000 == 0) sr_hint = sr_code = 12;
			else if(sr <= 655350 && sr % 10 == 0) sr_hint = sr_code = 14;
			else if(sr <= 0xffff) sr_hint = sr_code = 13;
			else sr_code = 0;
			break;
	}
	switch(P.bps) {
		case 8: bps_code = 1; break; case 12: bps_code = 2; break; case 16: bps_code = 4; break;
		case 20: bps_code = 5; break; case 24: bps_code = 6; break; case 32: bps_code = 7; break;
		default: bps_code = 0; break;
	}
	uint8_t t[4] = {0, 0, 0, 0};
	uint32_t tn = 0;
	if(bs_hint == 6) t[tn++] = (uint8_t)(n - 1);
	else if(bs_hint == 7) { t[tn++] = (uint8_t)((n - 1) >> 8); t[tn++] = (uint8_t)(n - 1); }
	if(sr_hint == 12) t[tn++] = (uint8_t)(sr / 1000);
	else if(sr_hint == 13) { t[tn++] = (uint8_t)(sr >> 8); t[tn++] = (uint8_t)sr; }
	else if(sr_hint == 14) { t[tn++] = (uint8_t)((sr / 10) >> 8); t[tn++] = (uint8_t)(sr / 10); }
	h.b2 = (bs_code << 4) | sr_code; h.bps_code2 = bps_code << 1;
	h.tail = ((uint32_t)t[0] << 24) | ((uint32_t)t[1] << 16) | ((uint32_t)t[2] << 8) | t[3]; h.tail_n = tn;
	return h;
}
// ... and assembled by sixteen lanes, lane j = byte j (4 + 6 + 4 + 1 bytes at most).  The CRC-8 (poly x^8 + x^2 + x + 1, crc.c:366) of
// a byte string is the sum of its bytes times x^(8 * bytes behind them, the CRC's own place included): every lane multiplies its
// byte up, the lanes add.  Returns this lane's word of the header (valid in lanes 0, 4, 8, 12: bytes j .. j+3, zero behind the
// header's end); nb: the header's length.
__device__ __forceinline__ uint32_t frame_header_lanes(const HdrConst &hc, uint32_t C, uint32_t ca, uint32_t v, uint32_t j, uint32_t &nb)
{
	const uint32_t nu = v < 0x80 ? 1u : v < 0x800 ? 2u : v < 0x10000 ? 3u : v < 0x200000 ? 4u : v < 0x4000000 ? 5u : 6u;      // bitwriter.c:832
	const uint32_t nbody = 4 + nu + hc.tail_n;
	uint32_t byte = 0;
	if(j == 0) byte = 0xff;
	else if(j == 1) byte = 0xf8;
	else if(j == 2) byte = hc.b2;
	else if(j == 3) byte = ((ca == 0 ? C - 1 : 7 + ca) << 4) | hc.bps_code2;
	else if(j < 4 + nu) {
		const uint32_t k = j - 4, sh = 6 * (nu - 1 - k);
		const uint32_t bits = sh < 32 ? v >> sh : 0u;
		byte = nu == 1 ? v : k == 0 ? ((0xffu << (8 - nu)) & 0xffu) | bits : 0x80u | (bits & 0x3fu);
	}
	else if(j < nbody) byte = (hc.tail >> (24 - 8 * (j - 4 - nu))) & 0xffu;
	// this byte's share of the CRC
	uint32_t c = j < nbody ? byte : 0u;
	for(uint32_t m = j; m < nbody; m++) {
		// c * x^8 mod x^8+x^2+x+1: x^8 = x^2+x+1, applied twice (the first product has 10 bits)
		const uint32_t w = c ^ (c << 1) ^ (c << 2), hi = w >> 8;
		c = (w ^ hi ^ (hi << 1) ^ (hi << 2)) & 0xffu;
	}
#pragma unroll
	for(int m = 1; m < 16; m <<= 1) c ^= (uint32_t)__shfl_xor((int)c, m, 16);
	if(j == nbody) byte = c;
	uint32_t word = byte << (24 - 8 * (j & 3));
	word |= (uint32_t)__shfl_xor((int)word, 1, 16);
	word |= (uint32_t)__shfl_xor((int)word, 2, 16);
	nb = nbody + 1;
	return word;
}

// pack_plan_kernel: sixteen lanes per frame of nominal length (lane j = tap j of a subframe's predictor), four frames per wavefront.
// Channel assignment (stream_encoder.c:3944-3972), the frame header (stream_encoder_framing.c:245-391) as four words, and per
// subframe a PackSub (flacgpu_dev.h): the winning record's fields, its taps as int16 pairs, and everything between the warm-up
// samples and the first Rice partition as a ready bit string.  pack2_kernel used to do this itself, four wavefronts each, as
// chains of LDS round trips and scalar decision trees between its barriers: a third of the kernel's wall time per workgroup.
// (Small on purpose: a kernel of this size runs each of its instructions once per CU, from a cold instruction cache -- the first
//  version, one lane per frame with its loops unrolled, took 14 us for 16384 frames of which a wavefront's own work was 2.5.)
constexpr uint32_t PLAN_LANES = 16, PLAN_FRAMES = 64 / PLAN_LANES;
__global__ __launch_bounds__(64) void pack_plan_kernel(const DevParams P, const HdrConst hc, uint32_t nmain, uint64_t first_frame_number,
                                                       const SubDecision *__restrict__ decisions, uint8_t *__restrict__ plan, uint32_t stride,
                                                       FrameInfo *__restrict__ info)
{
	__shared__ uint32_t shB[PLAN_FRAMES][10];                // a subframe's bit string under construction (8 words + 2 of or_bits' slack)
	const uint32_t j = threadIdx.x % PLAN_LANES, g = threadIdx.x / PLAN_LANES;
	const uint32_t f_raw = blockIdx.x * PLAN_FRAMES + g;
	const bool live = f_raw < nmain;
	const uint32_t f = live ? f_raw : nmain - 1;             // (lanes of a frame past the end go along and store nothing: shuffles below)
	const uint32_t C = P.channels, N = P.blocksize;
	const SubDecision *dec = decisions + (size_t)f * P.ncand;
	uint32_t ca = 0, left = 0, right = 1;
	if(P.ms_mode == 1) {
		const uint32_t b0 = dec[0].bits + dec[1].bits, b1 = dec[0].bits + dec[3].bits, b2 = dec[1].bits + dec[3].bits, b3 = dec[2].bits + dec[3].bits;
		uint32_t mn = b0;
		if(b1 < mn) { mn = b1; ca = 1; }
		if(b2 < mn) { mn = b2; ca = 2; }
		if(b3 < mn) { mn = b3; ca = 3; }
		left = ca == 2 ? 3 : ca == 3 ? 2 : 0;
		right = ca == 0 ? 1 : ca == 2 ? 1 : 3;
	}
	else if(P.ms_mode == 2) ca = dec[0].which >= 2 ? 3 : 0;
	PackHead *H = (PackHead *)(plan + (size_t)f * stride);
	{
		uint32_t nb;
		const uint32_t word = frame_header_lanes(hc, C, ca, (uint32_t)(first_frame_number + f), j, nb);
		if(live && (j & 3) == 0) H->hw[j >> 2] = word;
		if(live && j == 0) {
			*(uint4 *)&H->hdr_bytes = make_uint4(nb, ca, 0u, 0u);
			if(info) info[f].channel_assignment = (uint8_t)ca;
		}
	}
	PackSub *S = (PackSub *)(H + 1);
#pragma unroll 1
	for(uint32_t s = 0; s < C; s++, S++) {
		const uint32_t di = P.ms_mode == 1 ? (s == 0 ? left : right) : s;
		const SubDecision *d = dec + di;
		const uint32_t which = d->which, type = d->type, order = d->order, wasted = d->wasted, po = d->po, rice2 = d->rice2, precision = d->precision;
		const int32_t shift = type == 3 ? (int32_t)d->shift : 0;
		const uint32_t sbps = P.bps - wasted + (which == C + 1 ? 1 : 0);
		const uint32_t smask = sbps >= 32 ? 0xffffffffu : (1u << sbps) - 1u;
		const uint32_t type_bits = type == 0 ? 0x00u : type == 1 ? 0x02u : type == 2 ? (0x10u | (order << 1)) : (0x40u | ((order - 1) << 1));
		// this lane's tap (fixed.c:470: the fixed predictors are FIRs with binomial taps and shift 0)
		int32_t c = 0;
		if(type == 3) c = d->q[j];
		else if(type == 2) {
			// row `order` of (-1)^j * binomial(order, j + 1), orders 1..4: 1 | 2 -1 | 3 -3 1 | 4 -6 4 -1
			const int32_t tab = order == 1 ? 0x0001 : order == 2 ? 0x00f2 : order == 3 ? 0x01d3 : order == 4 ? 0xf4a4 : 0;
			c = j < 4 ? (int32_t)((uint32_t)tab << (28 - 4 * j)) >> 28 : 0;
		}
		if(j >= order) c = 0;
		// lpc.c:942-976's residual-width bound: the sum of the taps' magnitudes, over the frame's sixteen lanes
		uint32_t abs_sum = (uint32_t)abs(c);
#pragma unroll
		for(int m = 1; m < (int)PLAN_LANES; m <<= 1) abs_sum += (uint32_t)__shfl_xor((int)abs_sum, m, PLAN_LANES);
		const bool wide = type == 3 && silog2_i64((int64_t)(((uint64_t)1 << (sbps - 1)) * abs_sum)) > 32;
		const int32_t c_next = __shfl_down(c, 1, PLAN_LANES);
		// the bit string between the warm-up samples and the first partition: [precision-1:4][shift:5][coefficients] (LPC), then the
		// Rice method and the partition order; every lane ORs its piece into the wavefront's copy
		if(j < 10) shB[g][j] = 0;
		__builtin_amdgcn_wave_barrier();
		uint32_t b_bits = 0;
		if(type == 3) {
			if(j == 0) { or_bits(shB[g], 8, 0, precision - 1, 4); or_bits(shB[g], 8, 4, (uint32_t)shift & 31u, 5); }
			if(j < order) or_bits(shB[g], 8, 9 + j * precision, (uint32_t)c & ((1u << precision) - 1u), precision);
			b_bits = 9 + order * precision;
		}
		if(type >= 2) {
			if(j == 1) { or_bits(shB[g], 8, b_bits, rice2 ? 1u : 0u, 2); or_bits(shB[g], 8, b_bits + 2, po, 4); }
			b_bits += 6;
		}
		__builtin_amdgcn_wave_barrier();
		if(live) {
			if(j == 0) {
				*(uint4 *)&S->di = make_uint4(di, type, order, wasted);
				*(uint4 *)&S->sbps = make_uint4(sbps, smask, d->fmt == 1 ? 1u : 0u, (uint32_t)shift);
				*(uint4 *)&S->fmode = make_uint4((uint32_t)fir_mode(wide ? 1u : 0u, sbps), po, rice2, type_bits | (wasted ? 1u : 0u));
				*(uint4 *)&S->constant = make_uint4((uint32_t)d->constant & smask, b_bits, d->bits, 0xffffffffu / umax32(N >> (po & 15u), 1u) + 1u);
				if(info) {
					flacgpu_subframe_info *si = &info[f].sub[s];
					si->type = (uint8_t)type; si->order = (uint8_t)order; si->wasted_bits = (uint8_t)wasted;
					si->partition_order = (uint8_t)po; si->rice2 = (uint8_t)rice2; si->precision = (uint8_t)precision; si->shift = d->shift;
					si->pad = 0; si->bits = d->bits;
				}
			}
			if((j & 1) == 0) S->QP[j >> 1] = ((uint32_t)c << 16) | ((uint32_t)c_next & 0xffffu);
			S->q[j] = c;
			if(j < 8) S->B[j] = shB[g][j];
		}
		__builtin_amdgcn_wave_barrier();
	}
}

// HINTS / NT / RUN: see above
template <int MAXORD, bool HINTS, int NT, int RUN = CHUNK>
// Workgroups per CU (wavefronts per SIMD) the register budget is set for.  Round 3 settled on five ("without a spill"); round 6's
// stamps show a workgroup's wall time is 45 % barriers and first-touch latencies, which more workgroups side by side hide: six for
// predictors of up to 12 taps (80 registers, one spilled in the -8 instance), seven for up to 8 (72, none) -- pack 0.889 -> 0.838 ms
// per 65536 frames at -8, 0.895 -> 0.806 at -5, same box (profiles/r06_ap_abn_pack2_waves_*.txt; seven at 12 taps spills 12 and loses).
// The 17.6 KB image of a 16-bit stereo frame allows seven; longer predictors keep five.
#ifndef PACK2_WAVES
#define PACK2_WAVES (MAXORD <= 8 ? 7 : MAXORD <= 12 ? 6 : 5)
#endif
__global__ __launch_bounds__(NT, NT == 64 ? 4 : PACK2_WAVES) void pack2_kernel(      // (one-wavefront workgroups: the LDS image allows four per SIMD, no more)
                                                    const DevParams P, const int32_t *__restrict__ chan,
                                                    uint32_t nmain, const uint8_t *__restrict__ plan, uint32_t plan_stride,
                                                    const SubDecision *__restrict__ decisions,
                                                    uint8_t *__restrict__ slots, uint32_t *__restrict__ frame_bytes,
                                                    unsigned long long *__restrict__ dbg, const PackOut O,
                                                    uint32_t *__restrict__ hints)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int tid = (int)threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (wave: a scalar register)
	const uint32_t C = P.channels, N = P.blocksize, n = N;
	// LDS: [tables and small state][64 zero bytes][frame image: slot_bytes + 16]
	constexpr uint32_t IMG_OFF = (uint32_t)((sizeof(Pack2Shared) + 15) & ~(size_t)15) + P2_IMG_PAD;
	Pack2Shared *sh = (Pack2Shared *)smem;
	uint32_t *img = (uint32_t *)(smem + IMG_OFF);
	const uint32_t cap_words = P.slot_bytes / 4;
	if(!lds_base_is_zero(smem)) __builtin_trap();             // (frame_crc16_end addresses the CRC tables absolutely)
	const bool fused = O.out != nullptr;
	uint32_t f = blockIdx.x;
	f = (uint32_t)__builtin_amdgcn_readfirstlane((int)f);        // (the frame's addresses are scalar registers)
#define PSTAMP(k) do { if(dbg && tid == 0) dbg[(size_t)blockIdx.x * 16 + (k)] = (unsigned long long)clock64(); } while(0)
#define UNI(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
	PSTAMP(0);
	// the frame's plan (pack_plan_kernel): every field the same for the whole workgroup, i.e. scalar loads
	const PackHead *H = (const PackHead *)(plan + (size_t)f * plan_stride);
	const PackSub *subs = (const PackSub *)(H + 1);
	const uint32_t hwv = H->hw[tid & 3];
	uint32_t pos = 8 * H->hdr_bytes;
	// What the subframe loop will wait for, asked for now: the plan's lines (into the scalar cache: a first touch of a line is a trip
	// to the L2, a thousand cycles and more under load, and the loop below would make one per subframe on its critical path) and the
	// lines of this thread's first run of the first two subframes (from HBM: the planar channels were written a millisecond ago).
	// The values are consumed behind the barrier so that the loads are not dropped.
	uint32_t touch_s = 0, touch_v = 0;
	{
		const uint32_t *pw = (const uint32_t *)H;
		const uint32_t nlines = (plan_stride + 63u) / 64u;
#pragma unroll
		for(uint32_t l = 1; l < 7; l++) touch_s ^= pw[16u * (l < nlines ? l : 0u)];
		const uint32_t c2 = C < 2u ? C : 2u;
#pragma unroll
		for(uint32_t s = 0; s < 2; s++) {
			const PackSub *d = subs + (s < c2 ? s : 0u);
			const uint32_t *src = (const uint32_t *)(chan + ((size_t)f * P.ncand + d->di) * N);
			const uint32_t b = RUN * (uint32_t)tid;
			touch_v ^= src[(b < n ? b : 0u) >> (d->fmt16 ? 1 : 0)];
		}
	}

	// CRC tables: one round of loads; image zeroed meanwhile
	{
		constexpr uint32_t TAB_ROUNDS = (4 * 256 / 2) / NT, XS_ROUNDS = (P2_XSPAN / 2) / NT;
		static_assert((4 * 256 / 2) % NT == 0 && (P2_XSPAN / 2) % NT == 0 && (CRC_SPAN + 2) / 2 <= NT, "whole passes of the workgroup per table");
		const uint32_t *tab32 = (const uint32_t *)g_crc_tables.tab, *xs32 = (const uint32_t *)g_crc_tables.xspan44, *xb32 = (const uint32_t *)g_crc_tables.xbyte;
		uint32_t tv[TAB_ROUNDS], xv[XS_ROUNDS];
#pragma unroll
		for(uint32_t k = 0; k < TAB_ROUNDS; k++) tv[k] = tab32[(uint32_t)tid + k * NT];
#pragma unroll
		for(uint32_t k = 0; k < XS_ROUNDS; k++) xv[k] = xs32[(uint32_t)tid + k * NT];
		const uint32_t bv = xb32[tid < (int)(CRC_SPAN + 2) / 2 ? tid : 0];
		{
			// (16 bytes a store: the pad in front, the image and the 16 bytes behind it; slot_bytes is a multiple of 16)
			uint4 *z4 = (uint4 *)(smem + IMG_OFF - P2_IMG_PAD);
			for(uint32_t q = (uint32_t)tid; q < P2_IMG_PAD / 16 + cap_words / 4 + 1; q += NT) z4[q] = make_uint4(0, 0, 0, 0);
		}
#pragma unroll
		for(uint32_t k = 0; k < TAB_ROUNDS; k++) ((uint32_t *)sh->crc_tab)[(uint32_t)tid + k * NT] = tv[k];
#pragma unroll
		for(uint32_t k = 0; k < XS_ROUNDS; k++) ((uint32_t *)sh->xspan)[(uint32_t)tid + k * NT] = xv[k];
		if(tid < (int)(CRC_SPAN + 2) / 2) ((uint32_t *)sh->xbyte)[tid] = bv;
	}
	__syncthreads();
	asm volatile("" :: "s"(touch_s), "v"(touch_v));
	PSTAMP(1);
	// the frame header: four words (zero behind its end) by four lanes of a wavefront that has no other single-lane duties
	if(tid >= (NT > 64 ? 64 : 0) && tid < (NT > 64 ? 64 : 0) + 4) atomicOr(&img[tid & 3], hwv);
	uint32_t scan_buf = 0;

	// ---- subframes (stream_encoder_framing.c:393-594) -------------------------------------------
	for(uint32_t s = 0; s < C; s++) {
		const PackSub *d = subs + s;
		const uint32_t type = d->type, order = d->order, wasted = d->wasted, sbps = d->sbps, smask = d->smask;
		const bool fmt16 = d->fmt16 != 0;       // 16-bit pairs: sbps <= 16, or a side channel whose samples all fit int16
		const uint32_t *src = (const uint32_t *)(chan + ((size_t)f * P.ncand + d->di) * N);
		const uint8_t *rice_params = decisions[(size_t)f * P.ncand + d->di].params;
		if(tid == 0) {
			or_bits(img, cap_words, pos, d->type_byte, 8);
			if(wasted) or_bits(img, cap_words, pos + 8 + (wasted - 1), 1, 1);
		}
		pos += 8 + wasted;
		if(type == 0) {
			if(tid == 0) or_bits(img, cap_words, pos, d->constant, sbps);
			pos += sbps;
		}
		else if(type == 1) {
			for(uint32_t base = RUN * (uint32_t)tid; base < n; base += RUN * NT) {
				int32_t x[RUN];
				if(RUN != CHUNK) {
					// (runs of 18: word loads, a run starts at a multiple of 36 or 72 bytes)
					if(fmt16) {
#pragma unroll
						for(int k = 0; k < RUN; k++) { const uint32_t w = src[base / 2 + (k >> 1)]; x[k] = (k & 1) ? ((int32_t)w >> 16) : (int32_t)(int16_t)(w & 0xffffu); }
					}
					else {
#pragma unroll
						for(int k = 0; k < RUN; k++) x[k] = (int32_t)src[base + k];
					}
				}
				else if(fmt16) {
					const uint4 a = ((const uint4 *)src)[base / 8], b = ((const uint4 *)src)[base / 8 + 1];
					const uint32_t wv[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
					for(int k = 0; k < CHUNK; k++) x[k] = (k & 1) ? ((int32_t)wv[k >> 1] >> 16) : (int32_t)(int16_t)(wv[k >> 1] & 0xffffu);
				}
				else {
#pragma unroll
					for(int k = 0; k < 4; k++) { const uint4 a = ((const uint4 *)src)[base / 4 + k]; x[4 * k] = (int32_t)a.x; x[4 * k + 1] = (int32_t)a.y; x[4 * k + 2] = (int32_t)a.z; x[4 * k + 3] = (int32_t)a.w; }
				}
#pragma unroll
				for(int k = 0; k < RUN; k++) or_bits(img, cap_words, pos + (base + (uint32_t)k) * sbps, (uint32_t)x[k] & smask, sbps);
			}
			pos += n * sbps;
		}
		else {
			const uint32_t warm_pos = pos;
			pos += order * sbps;
			// LPC precision, shift, coefficients; Rice method and partition order: the plan's bit string, a word per lane
			const uint32_t b_bits = d->b_bits, b_pos = pos;
			const uint32_t b_word = d->B[tid & 7];             // (written behind the residual: the load has the FIR to come back)
			pos += b_bits;
			const int shift = d->shift;
			const uint32_t po = d->po, plen = d->rice2 ? 5u : 4u;
			const int fmode = (int)d->fmode;
			const bool wide = fmode == 2;
			// the taps as int16 pairs in scalar registers for the packed-sample chains (they are the subframe's, not the thread's)
			uint32_t QP[8];
#pragma unroll
			for(int pp = 0; pp < 8; pp++) QP[pp] = 2 * pp < MAXORD ? d->QP[pp] : 0u;
			const int32_t *q = d->q;
			const uint32_t psize = n >> po, inv_psize = d->inv_psize;
			PSTAMP(2 + 4 * s);
			for(uint32_t base0 = 0; base0 < n; base0 += RUN * NT) {
				// (a thread without a run in this pass -- block sizes that are not a multiple of the workgroup's samples -- computes the
				//  block's last run again and writes nothing: no branch around the arithmetic, no registers that are defined on one side only)
				const bool head_wave = base0 == 0 && wave == 0;
				const uint32_t base_own = base0 + RUN * (uint32_t)tid;
				const bool active = base_own < n;
				const uint32_t base = active ? base_own : n - RUN;
				int32_t r[RUN];
				uint32_t mybits = 0, k = 0;
				bool starts = false;
				{
					if(fmt16) {
						// window words: A[0..7] = samples base-16..base-1, A[8..15] = own
						uint32_t A[8 + RUN / 2];
						if(RUN != CHUNK) {
							// nine own words behind eight of history, all at word alignment
#pragma unroll
							for(int m = 0; m < RUN / 2; m++) A[8 + m] = src[base / 2 + m];
#pragma unroll
							for(int m = 0; m < 8; m++) A[m] = base ? src[base / 2 - 8 + m] : 0u;
						}
						else {
							const uint4 c0 = ((const uint4 *)src)[base / 8], c1 = ((const uint4 *)src)[base / 8 + 1];
							A[8] = c0.x; A[9] = c0.y; A[10] = c0.z; A[11] = c0.w; A[12] = c1.x; A[13] = c1.y; A[14] = c1.z; A[15] = c1.w;
							uint4 h0 = make_uint4(0, 0, 0, 0), h1 = make_uint4(0, 0, 0, 0);
							if(base) { h0 = ((const uint4 *)src)[base / 8 - 2]; h1 = ((const uint4 *)src)[base / 8 - 1]; }
							A[0] = h0.x; A[1] = h0.y; A[2] = h0.z; A[3] = h0.w; A[4] = h1.x; A[5] = h1.y; A[6] = h1.z; A[7] = h1.w;
						}
						if(!wide) {
							const uint32_t np = (order + 1) / 2;
							if(MAXORD >= 16 && np > 6) pack_fir_packed<MAXORD >= 16 ? 8 : 2, RUN>(A, QP, (uint32_t)shift, r);
							else if(MAXORD >= 12 && np > 4) pack_fir_packed<MAXORD >= 12 ? 6 : 2, RUN>(A, QP, (uint32_t)shift, r);
							else if(np > 2) pack_fir_packed<4, RUN>(A, QP, (uint32_t)shift, r);
							else pack_fir_packed<2, RUN>(A, QP, (uint32_t)shift, r);
						}
						else {
							int32_t x[16 + RUN];
#pragma unroll
							for(int kk = 0; kk < 16 + RUN; kk++) x[kk] = (kk & 1) ? ((int32_t)A[kk >> 1] >> 16) : (int32_t)(int16_t)(A[kk >> 1] & 0xffffu);
							pack_fir_i32<MAXORD, 2, RUN>(x, q, shift, r);
						}
					}
					else {
						int32_t x[16 + RUN];
						if(RUN != CHUNK) {
#pragma unroll
							for(int kk = 0; kk < 16; kk++) x[kk] = base ? (int32_t)src[base - 16 + kk] : 0;
#pragma unroll
							for(int kk = 0; kk < RUN; kk++) x[16 + kk] = (int32_t)src[base + kk];
						}
						else {
#pragma unroll
						for(int kk = 0; kk < 4; kk++) {
							uint4 a = make_uint4(0, 0, 0, 0);
							if(base) a = ((const uint4 *)src)[base / 4 - 4 + kk];
							x[4 * kk] = (int32_t)a.x; x[4 * kk + 1] = (int32_t)a.y; x[4 * kk + 2] = (int32_t)a.z; x[4 * kk + 3] = (int32_t)a.w;
						}
#pragma unroll
						for(int kk = 0; kk < 4; kk++) {
							const uint4 a = ((const uint4 *)src)[base / 4 + kk];
							x[16 + 4 * kk] = (int32_t)a.x; x[16 + 4 * kk + 1] = (int32_t)a.y; x[16 + 4 * kk + 2] = (int32_t)a.z; x[16 + 4 * kk + 3] = (int32_t)a.w;
						}
						}
						if(fmode == 0) {
							if(MAXORD >= 16 && order > 12) pack_fir_i32<MAXORD >= 16 ? 16 : 4, 0, RUN>(x, q, shift, r);
							else if(MAXORD >= 12 && order > 8) pack_fir_i32<MAXORD >= 12 ? 12 : 4, 0, RUN>(x, q, shift, r);
							else if(order > 4) pack_fir_i32<8, 0, RUN>(x, q, shift, r);
							else pack_fir_i32<4, 0, RUN>(x, q, shift, r);
						}
						else if(fmode == 1) pack_fir_i32<MAXORD, 1, RUN>(x, q, shift, r);
						else pack_fir_i32<MAXORD, 2, RUN>(x, q, shift, r);
					}
					// Rice code sizes: the whole run lies in one partition (partition sizes are multiples of 16)
					const uint32_t part = __umulhi(base, inv_psize);        // = base / psize
					k = rice_params[part];
					starts = base == part * psize;
					if(starts) mybits = plen;
					// (the block's first run: its first `order` samples are the warm-up samples and have no code -- one thread of one
					//  wavefront; that wavefront runs the guarded loops, the others the plain ones)
					const uint32_t kp1 = k + 1u;
					uint32_t zeros = 0;
					if(head_wave) {
						const uint32_t skip = base == 0 ? order : 0u;
#pragma unroll
						for(int t = 0; t < RUN; t++) {
							const uint32_t u = ((uint32_t)r[t] << 1) ^ (uint32_t)(r[t] >> 31);
							r[t] = (int32_t)u;                 // the write phase wants the folded value, not the residual
							zeros += (t < MAXORD && (uint32_t)t < skip) ? 0u : u >> k;
						}
						mybits += zeros + ((uint32_t)RUN - skip) * kp1;
					}
					else {
#pragma unroll
						for(int t = 0; t < RUN; t++) {
							const uint32_t u = ((uint32_t)r[t] << 1) ^ (uint32_t)(r[t] >> 31);
							r[t] = (int32_t)u;
							zeros += u >> k;
						}
						mybits += zeros + (uint32_t)RUN * kp1;
					}
					if(!active) mybits = 0;
				}
				PSTAMP(3 + 4 * s);
				// bit offset of this thread: wavefront scan + wavefront totals through LDS
				const uint32_t incl = wave_scan_incl_dpp(mybits);
				if(lane == 63) sh->wtot[scan_buf][wave] = incl;
				__syncthreads();
				uint32_t woff = 0, total = 0;
#pragma unroll
				for(int w = 0; w < NT / 64; w++) { const uint32_t tw = sh->wtot[scan_buf][w]; if(w < wave) woff += tw; total += tw; }
				woff = UNI(woff); total = UNI(total);          // the same in every lane of a wavefront: the running bit position stays a scalar
				scan_buf ^= 1;
				PSTAMP(4 + 4 * s);
				if(active) {
					uint32_t p = pos + woff + incl - mybits;
					// the verify pass decodes a run per thread (flacgpu_decode_hinted.h): it is told where this one starts
					if(HINTS && base0 == 0) hints[((size_t)f * C + s) * HINT_RUNS + (uint32_t)tid] = p;
					if(starts) { or_bits(img, cap_words, p, k, plen); p += plen; }
					// a run that ends inside the image writes without clamping its word index; one that does not belongs to a frame
					// that overflows its slot and is discarded (frame_bytes = ~0): its codes are not written at all
					if(pos + woff + incl <= cap_words * 32u) {
						// the code left-aligned in a word: stop bit, then the k low bits of u.  u << (31 - k) puts them there; what it leaves in
						// bit 31 (bit k of u) is covered by the stop bit, everything above has left the word
						// at = the bit the stop bit goes to: the one before's, plus that code's 1 + k bits, plus this one's zeros
						const uint32_t lsh = 31u - k, kp1 = k + 1u;
						uint32_t at = p - kp1;
						if(head_wave) {
							const uint32_t skip = base == 0 ? order : 0u;
#pragma unroll
							for(int t = 0; t < RUN; t++) {
								const uint32_t u = (uint32_t)r[t];
								if(t < MAXORD) {
									// (a warm-up position: nothing is added, nothing is written -- by selects, not by branches)
									const bool gone = (uint32_t)t < skip;
									at = at + (gone ? 0u : kp1) + (gone ? 0u : u >> k);
									or_code_abs(IMG_OFF, at, gone ? 0u : (u << lsh) | 0x80000000u);
								}
								else { at = at + kp1 + (u >> k); or_code_abs(IMG_OFF, at, (u << lsh) | 0x80000000u); }
							}
						}
						else {
#pragma unroll
							for(int t = 0; t < RUN; t++) {
								const uint32_t u = (uint32_t)r[t];
								at = at + kp1 + (u >> k);
								or_code_abs(IMG_OFF, at, (u << lsh) | 0x80000000u);
							}
						}
					}
				}
				pos += total;
				PSTAMP(5 + 4 * s);
			}
			// warm-up samples, verbatim: lane i of the first wavefront fetches sample i itself (thread 0 used to write them one after the
			// other out of its window: a serial stretch of `order` LDS round trips in front of the barrier everybody waits at); the lines
			// were read by thread 0 a moment ago
			if((uint32_t)tid < order) {
				const uint32_t warm_v = fmt16 ? (uint32_t)(int32_t)((const int16_t *)src)[tid] : src[tid];
				or_bits(img, cap_words, warm_pos + (uint32_t)tid * sbps, warm_v & smask, sbps);
			}
			if((uint32_t)tid < (b_bits + 31u) >> 5) or_bits(img, cap_words, b_pos + 32u * (uint32_t)tid, b_word, 32);
		}
	}
	__syncthreads();
	PSTAMP(10);

	// ---- zero-pad to a byte, CRC-16 over the whole frame, footer (stream_encoder.c:3720-3734) --------
	const uint32_t body_bytes = (pos + 7) >> 3;
	const uint32_t total_bytes = body_bytes + 2;
	const bool overflow = total_bytes > P.slot_bytes;
	const uint32_t mine = overflow ? 0u : total_bytes;
	// fused output: the frame's length goes out now, ahead of the CRC and the store
	uint64_t before = 0;
	if(fused && tid == NT - 64) before = fo_publish(O, f, mine);
	{
		const uint32_t crc = frame_crc16_end<NT, (int)CRC2_WORDS>(img, overflow ? 0 : body_bytes, sh->crc_parts, tid, sh->xspan, P2_XSPAN, g_crc_tables.xspan44);
		PSTAMP(11);
		if(tid == 0) or_bits(img, cap_words, body_bytes * 8, crc, 16);
		__syncthreads();
	}
	// ---- store: image words are big-endian views, the stream is a byte array ---------------------------
	if(fused) {
		if(tid == NT - 64) fo_close_segment(O, f, nmain, mine, before);          // (the counter has had the CRC to come back)
		// this frame's place in the stream: the sum of the lengths of all frames in front of it
		if(wave == 0) {
			uint64_t excl = 0;
			const bool placed = fo_exclusive(O, f, lane, excl);
			if(lane == 0) { sh->excl_lo = (uint32_t)excl; sh->excl_hi = (uint32_t)(excl >> 32); sh->placed = placed ? 1u : 0u; }
		}
		__syncthreads();
		const uint64_t off = ((uint64_t)sh->excl_hi << 32) | sh->excl_lo;
		const bool placed = sh->placed != 0;
		if(placed) {
			if(mine && off + mine <= O.cap) store_image<NT>(img, O.out + off, mine, tid);
		}
		else {
			// (they did not: the frame goes to its slot and onto fo_fixup_kernel's list)
			uint32_t *dst = (uint32_t *)(slots + (size_t)f * P.slot_bytes);
			const uint32_t words = (umin32(total_bytes, P.slot_bytes) + 3) >> 2;
			for(uint32_t w = (uint32_t)tid; w < words; w += NT) dst[w] = __builtin_bswap32(img[w]);
		}
		if(tid == 0) {
			frame_bytes[f] = overflow ? 0xffffffffu : total_bytes;
			if(placed) {
				O.offsets[f] = off;
				if(f + 1 == nmain) { O.offsets[f + 1] = off + mine; *O.total = off + mine; }      // (a short last block goes behind: append_tail_kernel)
			}
			else O.fall[atomicAdd(&O.nfall[O.epoch & 1u], 1u)] = f;
		}
	}
	else {
		uint32_t *dst = (uint32_t *)(slots + (size_t)f * P.slot_bytes);
		const uint32_t words = (umin32(total_bytes, P.slot_bytes) + 3) >> 2;
		for(uint32_t w = (uint32_t)tid; w < words; w += NT) dst[w] = __builtin_bswap32(img[w]);
		if(tid == 0) {
			frame_bytes[f] = overflow ? 0xffffffffu : total_bytes;
		}
	}
	PSTAMP(12);
#undef PSTAMP
#undef UNI
}

// ---------------------------------------------------------------------------------------------
// ff_kernel: the presets without an LPC search (-0, -1, -2) on 16-bit stereo in their 1152-sample blocks -- ONE kernel, one
// wavefront per frame, everything in registers.
// ---------------------------------------------------------------------------------------------
// For these presets a subframe's only residual candidate is the fixed predictor of the guessed order (prep2_kernel<.,.,DECIDE>,
// flacgpu_prep.hip), and prep, decision and pack of a 1152-sample frame are more per-frame work than per-sample work: three
// kernels, a planar copy and a decision record through HBM, workgroups of two to four wavefronts that meet at barriers.  Here a
// lane owns the 18 samples [18 L, 18 L + 18) of both channels (1152 = 64 x 18) from the load to the last Rice code:
//   coalesced load -> transposed LDS tile (left and right as an int16 pair per word) -> 22 words per lane (4 in front) ->
//   difference sums of every candidate channel (prep2_chunk) -> wasted bits, order guess, CONSTANT / VERBATIM / FIXED with the
//   Rice search on the lane's own sums (rice_search_nodes: a lane's 18 samples are exactly one of the 64 leaves) -> channel
//   assignment -> frame header, subframes (residual = the k-th difference of registers), CRC-16, slot.
// Same decisions and bytes as prep2_kernel + eval_list_kernel + pack2_kernel (stream_encoder.c:3747-4043, 4100-4140, 4701-5075;
// stream_encoder_framing.c:245-594).  A channel whose sums leave the 32-bit node arithmetic of rice_search_nodes (no 16-bit signal
// a fixed predictor was GUESSED for gets there with the presets' partition orders; a hand-picked -r 6 on a pathological burst
// can) takes ff_rice_search_wide below instead: every frame of nominal length is this kernel's, and with the fused output
// (PackOut) a batch of the fast presets is ONE launch.
constexpr int FF_N = 1152, FF_RUN = 18, FF_TS = 66, FF_TILE_BYTES = FF_RUN * FF_TS * 4;
constexpr uint32_t FF_XSPAN = 128;                         // span shifts kept in LDS: frames of up to 5.6 KB (these are 4.7 KB at most)
struct FFShared {
	uint16_t crc_tab[4][256];
	uint16_t xspan[FF_XSPAN];
	uint16_t xbyte[CRC_SPAN + 2];
	uint32_t crc_parts[2];
	uint32_t divtab[7 * (MAX_ORDER + 1)];                   // rows 0..6 (partition orders), columns 0..4 used
	uint8_t kout[4][64];                                   // Rice parameters of the four candidate channels
	uint32_t rec[4][12];                                   // their decision records (FFDec, word by word), between the decision loop and the frame
};
struct FFDec { uint32_t which, type, order, wasted, sbps, bits, po, rice2; int32_t constant; };
// x^(416 m) mod P, m = 0..255: the span shifts of frame_crc16_p2<64, 13> (52-byte spans).  The first FF_XSPAN of them are copied to
// LDS; the frames of 17..24-bit input (up to 7.5 KB: 145 spans) read the rest here
constexpr uint32_t FF_XSPAN_ALL = 256;
struct FFSpan { uint16_t x[FF_XSPAN_ALL]; };
constexpr FFSpan make_ff_span()
{
	FFSpan t{};
	uint32_t xs = 1;
	for(int r = 0; r < 52; r++) xs = crc_mulx8(xs);                          // x^416
	uint32_t c = 1;
	for(uint32_t m = 0; m < FF_XSPAN_ALL; m++) {
		t.x[m] = (uint16_t)c;
		uint32_t r = 0, b = xs;
		for(int i = 0; i < 16; i++) { r = crc_mulx(r); if(b & 0x8000u) r ^= c; b = (b << 1) & 0xffffu; }
		c = r;
	}
	return t;
}
__device__ const FFSpan g_ff_span = make_ff_span();
// 0x40000 / ((1152 >> po) - order), po = 0..6, order = 0..4: the divisors of set_partitioned_rice_'s mean (stream_encoder.c:5018)
struct FFDiv { uint32_t v[35]; };
constexpr FFDiv make_ff_div() { FFDiv t{}; for(uint32_t i = 0; i < 35; i++) t.v[i] = 0x40000u / ((1152u >> (i / 5u)) - (i % 5u)); return t; }
__device__ const FFDiv g_ff_div_tab = make_ff_div();
#define g_ff_div g_ff_div_tab.v
__host__ __device__ inline uint32_t ff_tile_bytes(uint32_t slot_bytes) { const uint32_t img = (slot_bytes + 8 + 15) & ~15u; return img > (uint32_t)FF_TILE_BYTES ? img : (uint32_t)FF_TILE_BYTES; }

// A lane's window: its 18 samples of both channels and the four in front of them.  Up to 16 bits per sample it stays packed, left
// in the low and right in the high half of a word: 22 registers instead of 44 across the kernel.  WIDE (round 6: 17..24-bit input,
// which the reference serves by the same code with its wide sums, stream_encoder.c:4098-4108, fixed.c:301): a register per sample
// and channel.
template <bool WIDE> struct FFWin;
template <> struct FFWin<false> { uint32_t w[FF_RUN + 4]; };
template <> struct FFWin<true> { int32_t a[FF_RUN + 4], b[FF_RUN + 4]; };
template <bool WIDE> __device__ __forceinline__ int32_t ff_left(const FFWin<WIDE> &W, int k) { if constexpr(WIDE) return W.a[k]; else return (int32_t)(int16_t)(W.w[k] & 0xffffu); }
template <bool WIDE> __device__ __forceinline__ int32_t ff_right(const FFWin<WIDE> &W, int k) { if constexpr(WIDE) return W.b[k]; else return (int32_t)W.w[k] >> 16; }
// the candidate channel `which` (0 left, 1 right, 2 mid, 3 side) of this lane's window
template <bool WIDE>
__device__ __forceinline__ void ff_channel(const FFWin<WIDE> &W, uint32_t which, int32_t (&x)[FF_RUN + 4])
{
#pragma unroll
	for(int k = 0; k < FF_RUN + 4; k++) {
		const int32_t a = ff_left<WIDE>(W, k), b = ff_right<WIDE>(W, k);
		x[k] = which == 0 ? a : which == 1 ? b : which == 2 ? ((a + b) >> 1) : (a - b);
	}
}
// find_best_partition_order_ / set_partitioned_rice_ (stream_encoder.c:4701-4795, 4997-5046) on the lanes' leaf sums in 64-bit
// arithmetic, one partition per lane and order: the rare path of ff_decide (nothing here is tuned).  leaf: 64 words of LDS.
// (not inlined: four copies of it in the straight-line code of ff_kernel<1> cost that kernel 30 spilled registers.  Everything goes in
// and out through scalars and LDS -- a pointer to a thread-private object handed to a function that is not inlined read back
// zeros here, DESIGN.md -- and the partition order comes back in the high half of the result)
__device__ __noinline__ uint64_t ff_rice_search_wide(uint32_t v, uint32_t n, uint32_t order, uint32_t max_po, uint32_t min_po, uint32_t rice_limit,
                                                     const uint32_t *divtab, uint32_t *leaf, uint8_t *kout, int lane)
{
	leaf[lane] = v;
	__builtin_amdgcn_wave_barrier();
	uint32_t best_bits = 0, best_po = 0, kbest = 0;
	for(int po = (int)max_po; po >= (int)min_po; po--) {
		const uint32_t nleaf = 64u >> po;
		uint32_t k = 0;
		uint64_t b = 0;
		if((uint32_t)lane < (1u << po)) {
			uint64_t sum = 0;
			for(uint32_t j = 0; j < nleaf; j++) sum += leaf[(uint32_t)lane * nleaf + j];
			const uint32_t o = lane == 0 ? order : 0u, ns = (n >> po) - o, div = divtab[(uint32_t)po * (MAX_ORDER + 1) + o];
			if(sum < 2 || (((sum - 1) * div) >> 18) == 0) k = 0;
			else k = ilog2_u64(((sum - 1) * div) >> 18) + 1;
			if(k >= rice_limit) k = rice_limit - 1;
			b = 4 + (uint64_t)(1 + k) * ns + (k ? (sum >> (k - 1)) : (sum << 1)) - (ns >> 1);
			if(b > 0xffffffffull) b = 0xffffffffull;
		}
		const uint64_t tot = 6 + wave_reduce_add_u64(b);
		const uint32_t bits = tot >= 0xffffffffull ? 0xffffffffu : (uint32_t)tot;
		if(best_bits == 0 || bits < best_bits) { best_bits = bits; best_po = (uint32_t)po; kbest = k; }       // strict <, highest order first (:4735-4763)
	}
	if((uint32_t)lane < (1u << best_po)) kout[lane] = (uint8_t)kbest;
	__builtin_amdgcn_wave_barrier();
	return ((uint64_t)best_po << 32) | best_bits;
}

// statistics and decision of one candidate channel (CPO: the partition order range when it is the presets' 0..3, else -1)
template <int CPO, bool WIDE>
__device__ __forceinline__ void ff_decide(const DevParams &P, uint32_t which, const int32_t (&x)[FF_RUN + 4], bool disable_constant, FFShared *sh, uint32_t *leaf, int lane, FFDec &D, uint32_t &alleq)
{
	constexpr uint32_t n = FF_N;
	Prep2Acc A;
	A.orv = 0; A.diff = 0; A.mag = 0;
#pragma unroll
	for(int k = 0; k < 5; k++) A.e[k] = 0;
	const int32_t first = (int32_t)__builtin_amdgcn_readfirstlane(x[4]);          // sample 0 of the block (lane 0's first own sample)
	uint32_t cs[5], ex[5] = {0, 0, 0, 0, 0};
	if constexpr(WIDE) {
		// |d4| <= 16 max|x|: a lane's eighteen stay below 2^32 for samples of up to 24 bits, not for the 25-bit side channel -- that one
		// adds every |difference| to the 64-bit sums (prep2_kernel<true, ., true> draws the same line)
		if(which == 3) prep2_chunk<true, false, true, FF_RUN>(x, lane == 0, first, A, cs, ex);
		else prep2_chunk<false, false, true, FF_RUN>(x, lane == 0, first, A, cs, ex);
	}
	else prep2_chunk<false, false, true, FF_RUN>(x, lane == 0, first, A, cs, ex);
	const uint32_t orv = wave_or_u32(A.orv), diff = wave_or_u32(A.diff);
	uint64_t e[5];
#pragma unroll
	for(int k = 0; k < 5; k++) e[k] = WIDE ? wave_sum_u50(A.e[k])               // < 2^39: 1152 fourth differences of 25-bit samples
	                                       : wave_sum_u32((uint32_t)A.e[k]);    // < 2^31: 1152 fourth differences of 17-bit samples
	alleq = diff == 0 ? 1u : 0u;
	uint32_t wasted = orv ? (uint32_t)(__ffs((int)orv) - 1) : 0;
	if(wasted > P.bps) wasted = P.bps;
	const uint32_t sbps = P.bps - wasted + (which == 3 ? 1 : 0);
	const uint32_t verbatim_bits = P.disable_verbatim ? 0xffffffffu : 8 + wasted + n * sbps;
	const uint32_t n4 = n - 4;
	const uint64_t e0 = e[0] >> wasted, e1 = e[1] >> wasted, e2 = e[2] >> wasted, e3 = e[3] >> wasted, e4 = e[4] >> wasted;
	uint32_t guess_fixed;
	{
		const uint64_t m34 = e3 < e4 ? e3 : e4, m234 = e2 < m34 ? e2 : m34, m1234 = e1 < m234 ? e1 : m234;
		if(e0 <= m1234) guess_fixed = 0;
		else if(e1 <= m234) guess_fixed = 1;
		else if(e2 <= m34) guess_fixed = 2;
		else if(e3 <= e4) guess_fixed = 3;
		else guess_fixed = 4;
	}
	const bool is_constant = !disable_constant && diff == 0;               // stream_encoder.c:4111-4140
	const int32_t constant = first >> wasted;
	const bool fixed_allowed = !is_constant && (!P.disable_fixed || verbatim_bits == 0xffffffffu);      // (max_lpc_order == 0 here)
	const uint32_t fixed_order = fixed_allowed ? guess_fixed : 0;
	const uint64_t eg = guess_fixed == 0 ? e0 : guess_fixed == 1 ? e1 : guess_fixed == 2 ? e2 : guess_fixed == 3 ? e3 : e4;
	// fixed.c:299 / stream_encoder.c:4169: the order is skipped when log2(ln2 * e / n4), as a float, reaches the sample width.  Below
	// e = n4 * 2^(sbps-1) that logarithm is under sbps - 1.5 whatever the rounding: no need to take it (it is a hundred fp64
	// instructions, flacgpu_log.h); music never gets near
	bool too_wide;
	if(sbps >= 1 && eg < ((uint64_t)n4 << (sbps - 1))) too_wide = false;
	else too_wide = fixed_rbps(eg, n4) >= (float)sbps;
	const bool fixed_valid = fixed_allowed && !too_wide;
	// first minimum in the reference's evaluation order: verbatim -> constant | the fixed order (prep2_kernel<.,.,DECIDE>)
	const uint32_t hdr = 8 + wasted;
	D.which = which; D.wasted = wasted; D.sbps = sbps;
	D.type = 1; D.bits = verbatim_bits; D.po = 0; D.rice2 = 0; D.order = 0; D.constant = 0;
	if(is_constant) {
		const uint32_t bits = hdr + sbps;
		if(bits < D.bits) { D.type = 0; D.constant = constant; D.bits = bits; }
	}
	if(fixed_valid) {
		uint32_t fmax = umin32(7u, P.max_po);                                 // 1152 = 2^7 * 9; <= 6 (prep2_decides)
		const uint32_t fmin = umin32(P.min_po, fmax), ee = 6 - fmax;
		// this lane's 18 samples are one of the 64 leaves of the search: the chunk sum of the chosen order (what the residual has in front
		// of sample 4 added on lane 0), shifted like the signal
		uint32_t po = 0, rbits;
		if constexpr(WIDE) {
			// (A.e[] is this lane's sum in both flavours of the chunk: the lane has one chunk)
			uint64_t v64 = fixed_order == 0 ? A.e[0] : fixed_order == 1 ? A.e[1] : fixed_order == 2 ? A.e[2] : fixed_order == 3 ? A.e[3] : A.e[4];
			if(lane == 0) v64 += fixed_order == 0 ? ex[0] : fixed_order == 1 ? ex[1] : fixed_order == 2 ? ex[2] : fixed_order == 3 ? ex[3] : ex[4];
			v64 >>= wasted;
			// rice_search_nodes is exact while no partition sum reaches 2^31 (its (sum << 1) >> k is the one place that could wrap; the
			// parameter and the bit counts are far from it): every lane below 2^25 guarantees that.  Beyond: the same search on 64-bit
			// sums, the leaf's lanes holding its sum between them (rice_search_owner; `narrow`: stream_encoder.c:4814-4817)
			if(__any((int)(v64 >= (1ull << 25)))) {
				const bool narrow = (sbps + 4) < (32 - ilog2_u32(n >> fmax));
				rbits = rice_search_owner(v64, narrow, ee, n, fixed_order, fmax, fmin, P.rice_limit, sh->divtab, sh->kout[which], &po, lane);
			}
			else rbits = rice_search_nodes((uint32_t)v64, ee, n, fixed_order, fmax, fmin, P.rice_limit, sh->divtab, sh->kout[which], &po, lane);
		}
		else {
			uint32_t v = fixed_order == 0 ? cs[0] : fixed_order == 1 ? cs[1] : fixed_order == 2 ? cs[2] : fixed_order == 3 ? cs[3] : cs[4];
			if(lane == 0) v += fixed_order == 0 ? ex[0] : fixed_order == 1 ? ex[1] : fixed_order == 2 ? ex[2] : fixed_order == 3 ? ex[3] : ex[4];
			v >>= wasted;
			// (a leaf partition is 2^ee lanes: rice_search_nodes wants its sum below 2^23)
			if(__any((int)(v >= ((1u << 23) >> ee)))) {
				const uint64_t r = ff_rice_search_wide(v, n, fixed_order, fmax, fmin, P.rice_limit, sh->divtab, leaf, sh->kout[which], lane);
				rbits = (uint32_t)r; po = (uint32_t)(r >> 32);
			}
			else if(CPO == 3) rbits = rice_search_nodes<3, 3>(v, ee, n, fixed_order, fmax, fmin, P.rice_limit, sh->divtab, sh->kout[which], &po, lane);
			else rbits = rice_search_nodes(v, ee, n, fixed_order, fmax, fmin, P.rice_limit, sh->divtab, sh->kout[which], &po, lane);
		}
		const uint32_t est = sat_add_u32(hdr + fixed_order * sbps, rbits);
		if(est > 0 && est < D.bits) { D.type = 2; D.bits = est; D.po = po; D.order = fixed_order; }
	}
	if(D.bits == 0xffffffffu) { D.type = 1; D.bits = hdr + n * sbps; }       // stream_encoder.c:4281
	if(D.type == 2) {
		uint32_t big = 0;
		if((uint32_t)lane < (1u << D.po)) big = sh->kout[which][lane] >= 15 ? 1u : 0u;
		D.rice2 = __any((int)big) ? 1u : 0u;                                  // stream_encoder.c:4786-4791
	}
	// the record is the same in every lane: scalar registers
#define FFUNI(x) x = (uint32_t)__builtin_amdgcn_readfirstlane((int)(x))
	FFUNI(D.type); FFUNI(D.order); FFUNI(D.wasted); FFUNI(D.sbps); FFUNI(D.bits); FFUNI(D.po); FFUNI(D.rice2);
	D.constant = __builtin_amdgcn_readfirstlane(D.constant);
#undef FFUNI
}

#ifndef FF_WAVES
#define FF_WAVES 4           // (the LDS allows four wavefronts per SIMD)
#endif
// ff_channel with the channel a scalar register: a branch per flavour instead of three selects per word
template <bool WIDE>
__device__ __forceinline__ void ff_channel_u(const FFWin<WIDE> &w, uint32_t which, int32_t (&x)[FF_RUN + 4])
{
	if(which == 0) ff_channel<WIDE>(w, 0, x);
	else if(which == 1) ff_channel<WIDE>(w, 1, x);
	else if(which == 2) ff_channel<WIDE>(w, 2, x);
	else ff_channel<WIDE>(w, 3, x);
}
// one subframe's share of the frame, before a bit of it is written: the folded residuals u (FIXED), the lane's Rice parameter,
// where in the subframe's residual section the lane's codes start, and the subframe's length in bits (everything but u uniform)
struct FFSub { uint32_t k, excl, end, bits; bool starts; };
template <bool WIDE>
__device__ __forceinline__ void ff_subframe_sizes(const FFWin<WIDE> &w, const FFDec &d, const FFShared *sh, int lane, uint32_t (&u)[FF_RUN], FFSub &S)
{
	constexpr uint32_t n = FF_N;
	S.k = 0; S.excl = 0; S.end = 0; S.starts = false;
	if(d.type == 0) { S.bits = 8 + d.wasted + d.sbps; return; }
	if(d.type == 1) { S.bits = 8 + d.wasted + n * d.sbps; return; }
	// the residual of the fixed predictor of this order = the order-th difference (fixed.c:470) of the shifted channel
	int32_t dd[FF_RUN + 4];
	ff_channel_u<WIDE>(w, d.which, dd);
	if(d.wasted) {
#pragma unroll
		for(int t = 0; t < FF_RUN + 4; t++) dd[t] >>= d.wasted;
	}
#pragma unroll
	for(int o = 1; o <= 4; o++) {
		if((uint32_t)o <= d.order) {
#pragma unroll
			for(int t = FF_RUN + 3; t >= o; t--) dd[t] = dd[t] - dd[t - 1];
		}
	}
	const uint32_t psize = n >> d.po, base = (uint32_t)lane * FF_RUN, part = base / psize;
	const uint32_t k = sh->kout[d.which][part];
	S.k = k;
	S.starts = base == part * psize;
	// code sizes: (u >> k) + 1 + k each; the warm-up samples (the first `order` of lane 0) have none
	const uint32_t skip = lane == 0 ? d.order : 0u;
	uint32_t q = 0;
#pragma unroll
	for(int t = 0; t < FF_RUN; t++) {
		const int32_t r = dd[t + 4];
		uint32_t f = ((uint32_t)r << 1) ^ (uint32_t)(r >> 31);
		if(t < 4 && (uint32_t)t < skip) f = 0;
		u[t] = f;
		q += f >> k;
	}
	const uint32_t mybits = q + ((uint32_t)FF_RUN - skip) * (1u + k) + (S.starts ? (d.rice2 ? 5u : 4u) : 0u);
	const uint32_t incl = wave_scan_incl_dpp(mybits);
	S.excl = incl - mybits; S.end = incl;
	S.bits = 8 + d.wasted + d.order * d.sbps + 6 + (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
}
// ... and its bits, at bit `pos` of the zeroed frame image (stream_encoder_framing.c:393-594, bitwriter.c:575-706)
template <bool WIDE>
__device__ __forceinline__ void ff_subframe_write(uint32_t *img, uint32_t img_abs /* LDS byte address of img */, uint32_t cap_words, uint32_t pos, const FFWin<WIDE> &w, const FFDec &d, const uint32_t (&u)[FF_RUN], const FFSub &S, int lane)
{
	const uint32_t type = d.type, order = d.order, wasted = d.wasted, sbps = d.sbps;
	const uint32_t type_bits = type == 0 ? 0x00u : type == 1 ? 0x02u : (0x10u | (order << 1));
	if(lane == 0) {
		or_bits(img, cap_words, pos, type_bits | (wasted ? 1u : 0u), 8);
		if(wasted) or_bits(img, cap_words, pos + 8 + (wasted - 1), 1, 1);
	}
	pos += 8 + wasted;
	const uint32_t smask = sbps >= 32 ? 0xffffffffu : (1u << sbps) - 1u;
	if(type == 0) {
		if(lane == 0) or_bits(img, cap_words, pos, (uint32_t)d.constant & smask, sbps);
		return;
	}
	if(type == 1) {
		int32_t x[FF_RUN + 4];
		ff_channel_u<WIDE>(w, d.which, x);
#pragma unroll
		for(int t = 0; t < FF_RUN; t++) or_bits(img, cap_words, pos + ((uint32_t)lane * FF_RUN + (uint32_t)t) * sbps, (uint32_t)(x[t + 4] >> wasted) & smask, sbps);
		return;
	}
	if(lane == 0) {
		// warm-up samples, verbatim (lane 0 holds them), and the entropy coding header
#pragma unroll
		for(int i = 0; i < 4; i++) {
			if((uint32_t)i < order) {
				const int32_t a = ff_left<WIDE>(w, 4 + i), b = ff_right<WIDE>(w, 4 + i);
				const int32_t xv = d.which == 0 ? a : d.which == 1 ? b : d.which == 2 ? ((a + b) >> 1) : (a - b);
				or_bits(img, cap_words, pos + (uint32_t)i * sbps, (uint32_t)(xv >> wasted) & smask, sbps);
			}
		}
		or_bits(img, cap_words, pos + order * sbps, d.rice2 ? 1u : 0u, 2);
		or_bits(img, cap_words, pos + order * sbps + 2, d.po, 4);
	}
	pos += order * sbps + 6;
	const uint32_t k = S.k, kp1 = k + 1u;
	uint32_t p = pos + S.excl;
	if(S.starts) { const uint32_t plen = d.rice2 ? 5u : 4u; or_bits(img, cap_words, p, k, plen); p += plen; }
	// a run that ends inside the image writes without clamping its word index; one that does not belongs to a frame that overflows
	// its slot and is discarded (frame_bytes = ~0): its codes are not written at all
	if(pos + S.end <= cap_words * 32u) {
		// the code left-aligned in a word: stop bit, then the k low bits of u.  u << (31 - k) puts them there; what it leaves in bit 31
		// (bit k of u) is covered by the stop bit, everything above has left the word.  at = the bit the stop bit goes to: the one
		// before's, plus that code's 1 + k bits, plus this one's zeros -- one v_add3 (a warm-up position has neither)
		const uint32_t lsh = 31u - k, skip = lane == 0 ? order : 0u;
		uint32_t at = p - kp1;
#pragma unroll
		for(int t = 0; t < FF_RUN; t++) {
			const uint32_t ut = u[t];
			if(t < 4) {
				// (lane 0's warm-up positions: no code)
				const bool gone = (uint32_t)t < skip;
				const uint32_t adv = gone ? 0u : kp1;
				at = at + adv + (ut >> k);
				if(!gone) or_code_abs(img_abs, at, (ut << lsh) | 0x80000000u);
			}
			else {
				at = at + kp1 + (ut >> k);
				or_code_abs(img_abs, at, (ut << lsh) | 0x80000000u);
			}
		}
	}
}

template <int MS, int CPO, bool WIDE = false>    // DevParams::ms_mode; partition orders 0..CPO known at compile time (3: what -0 .. -2 set), or -1; WIDE: 17..24-bit input
__global__ __launch_bounds__(64, WIDE ? 3 : FF_WAVES) void ff_kernel(const DevParams P, const int32_t *__restrict__ pcm, uint32_t nmain, uint64_t first_frame_number,
                                                 uint8_t *__restrict__ slots, uint32_t *__restrict__ frame_bytes, FrameInfo *__restrict__ info, const PackOut O)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int lane = (int)threadIdx.x;
	const bool publish = O.fstate != nullptr;
	const uint32_t f = blockIdx.x;
	constexpr uint32_t n = FF_N;
	// LDS: [tables and small state][64 zero bytes][the transposed tile, later the frame image]
	constexpr uint32_t IMG_OFF = (uint32_t)((sizeof(FFShared) + 15) & ~(size_t)15) + P2_IMG_PAD;
	FFShared *sh = (FFShared *)smem;
	uint32_t *tile = (uint32_t *)(smem + IMG_OFF);
	if(!lds_base_is_zero(smem)) __builtin_trap();             // (frame_crc16_end addresses the CRC tables absolutely)
	// ---- every load of the frame at once: samples, CRC tables -------------------------------------------------------------------
	// A lane loads the 18 samples it owns straight from the interleaved block: nine 16-byte loads of 144 consecutive bytes (the
	// wavefront as a whole reads its 9216 bytes once; a 16-byte piece never straddles a cache line).  Round 3 loaded coalesced and
	// transposed through an LDS tile: 18 loads, 18 address computations with a division by 18, 18 LDS writes, 22 LDS reads and
	// two wavefront barriers -- a tenth of the kernel's instructions for a layout change the memory system does as well
	// (profiles/archive/r04_n_ff_direct_loads_ab.txt).  The four samples in front of a lane's run are its left neighbour's last four:
	// a DPP shift by one lane (lane 0 gets zeros: the start of the block).
	FFWin<WIDE> w;
	{
		const fo_u4 *p = (const fo_u4 *)(pcm + (size_t)f * n * 2) + (uint32_t)lane * (FF_RUN / 2);
		fo_u4 v[FF_RUN / 2];
#pragma unroll
		for(int k = 0; k < FF_RUN / 2; k++) v[k] = p[k];
		const uint32_t *tab32 = (const uint32_t *)g_crc_tables.tab, *xs32 = (const uint32_t *)g_ff_span.x;
		uint32_t tv[8];
#pragma unroll
		for(int k = 0; k < 8; k++) tv[k] = tab32[(uint32_t)lane + 64u * (uint32_t)k];
		const uint32_t xv = xs32[lane];
#pragma unroll
		for(int k = 0; k < 8; k++) ((uint32_t *)sh->crc_tab)[(uint32_t)lane + 64u * (uint32_t)k] = tv[k];
		((uint32_t *)sh->xspan)[lane] = xv;
		if(lane < 35) { const uint32_t po = (uint32_t)lane / 5u, o = (uint32_t)lane - po * 5u; sh->divtab[po * (MAX_ORDER + 1) + o] = g_ff_div[lane]; }
		if constexpr(WIDE) {
			// (the loaded words are the window)
#pragma unroll
			for(int k = 0; k < FF_RUN / 2; k++) {
				w.a[4 + 2 * k] = (int32_t)v[k].x; w.b[4 + 2 * k] = (int32_t)v[k].y;
				w.a[4 + 2 * k + 1] = (int32_t)v[k].z; w.b[4 + 2 * k + 1] = (int32_t)v[k].w;
			}
#pragma unroll
			for(int k = 0; k < 4; k++) {
				w.a[k] = __builtin_amdgcn_update_dpp(0, w.a[FF_RUN + k], 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
				w.b[k] = __builtin_amdgcn_update_dpp(0, w.b[FF_RUN + k], 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
			}
		}
		else {
			// left in the low half, right in the high half of a word: one byte permute per sample
#pragma unroll
			for(int k = 0; k < FF_RUN / 2; k++) {
				w.w[4 + 2 * k] = __builtin_amdgcn_perm(v[k].y, v[k].x, 0x05040100u);
				w.w[4 + 2 * k + 1] = __builtin_amdgcn_perm(v[k].w, v[k].z, 0x05040100u);
			}
#pragma unroll
			for(int k = 0; k < 4; k++) w.w[k] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w.w[FF_RUN + k], 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
		}
	}
	__builtin_amdgcn_wave_barrier();

	// ---- candidate channels: one pass of a loop each (the decision records wait in LDS) ------------------------------------------
	uint32_t ca = 0;
	uint32_t *leaf = tile;                                                    // (scratch of the wide Rice search: the tile is free until the image is zeroed)
	uint32_t li = 0, ri = 1;                                                  // MS == 2: the pair the loose search picked
	uint32_t alleq_l = 0;
	if(MS == 2) {
		// loose mid/side (stream_encoder.c:3778-3807): left/right or mid/side for the whole frame, from first differences
		uint32_t lr = 0, ms = 0;
#pragma unroll
		for(int t = 0; t < FF_RUN; t++) {
			if(t > 0 || lane > 0) {
				// (24-bit samples: |pl|, |pr| < 2^25, eighteen times 3 * 2^25 stay below 2^32)
				const int32_t pl = ff_left<WIDE>(w, t + 4) - ff_left<WIDE>(w, t + 3), pr = ff_right<WIDE>(w, t + 4) - ff_right<WIDE>(w, t + 3);
				lr += (uint32_t)(abs(pl) + abs(pr));
				ms += (uint32_t)(abs((pl + pr) >> 1) + abs(pl - pr));
			}
		}
		const uint64_t lrs = wave_sum_u50((uint64_t)lr), mss = wave_sum_u50((uint64_t)ms);
		const bool use_ms = !(lrs < mss);
		if(P.limit_min_bitrate) {
			// (the all-equal flag of the left channel decides whether the second subframe may be CONSTANT, stream_encoder.c:3874-3879)
			uint32_t dl = 0;
			if constexpr(WIDE) {
#pragma unroll
				for(int t = 0; t < FF_RUN; t++) dl |= (uint32_t)(w.a[t + 4] ^ w.a[4]);
				dl |= (uint32_t)(w.a[4] ^ __builtin_amdgcn_readfirstlane(w.a[4]));
			}
			else {
#pragma unroll
				for(int t = 0; t < FF_RUN; t++) dl |= (w.w[t + 4] ^ w.w[4]) & 0xffffu;
				const uint32_t fl = (uint32_t)__builtin_amdgcn_readfirstlane((int)w.w[4]);
				dl |= (w.w[4] ^ fl) & 0xffffu;
			}
			alleq_l = wave_or_u32(dl) == 0 ? 1u : 0u;
		}
		li = use_ms ? 2 : 0; ri = use_ms ? 3 : 1;
		ca = use_ms ? 3 : 0;
	}
	{
		constexpr uint32_t NC = MS == 1 ? 4u : 2u;
		bool dc = P.disable_constant != 0;
#pragma unroll 1
		for(uint32_t ci = 0; ci < NC; ci++) {
			const uint32_t which = MS == 2 ? (ci ? ri : li) : ci;
			if(ci) {
				// every channel but the first: no CONSTANT when all the ones in front are constant (stream_encoder.c:3874-3879); with the
				// loose search only the right channel proper (which == 1) can lose its CONSTANT (prep2_kernel)
				if(MS == 2) { if(P.limit_min_bitrate && which == 1 && alleq_l) dc = true; }
				else if(P.limit_min_bitrate && alleq_l) dc = true;
			}
			// (the window is the same in every pass, and so are its unpacked halves, their mean and their difference: the compiler
			//  would keep all four, 88 registers, across the loop -- the empty asm makes the words look new in every pass)
			if constexpr(WIDE) {
#pragma unroll
				for(int k = 0; k < FF_RUN + 4; k++) { asm volatile("" : "+v"(w.a[k])); asm volatile("" : "+v"(w.b[k])); }
			}
			else {
#pragma unroll
				for(int k = 0; k < FF_RUN + 4; k++) asm volatile("" : "+v"(w.w[k]));
			}
			int32_t x[FF_RUN + 4];
			ff_channel_u<WIDE>(w, which, x);
			FFDec D;
			uint32_t alleq = 0;
			ff_decide<CPO, WIDE>(P, which, x, dc, sh, leaf, lane, D, alleq);
			if(ci == 0 && MS != 2) alleq_l = (uint32_t)__builtin_amdgcn_readfirstlane((int)alleq);
			if(lane == 0) {
				uint32_t *rec = sh->rec[ci];
				rec[0] = D.which; rec[1] = D.type; rec[2] = D.order; rec[3] = D.wasted; rec[4] = D.sbps; rec[5] = D.bits; rec[6] = D.po; rec[7] = D.rice2; rec[8] = (uint32_t)D.constant;
			}
		}
	}
	__builtin_amdgcn_wave_barrier();
	FFDec DL, DR;                                                             // the two subframes of the frame, in stream order
	{
		uint32_t l = 0, r = 1;
		if(MS == 1) {
			// channel assignment (stream_encoder.c:3944-3972)
			const uint32_t b0 = sh->rec[0][5], b1 = sh->rec[1][5], b2 = sh->rec[2][5], b3 = sh->rec[3][5];
			const uint32_t s0 = b0 + b1, s1 = b0 + b3, s2 = b1 + b3, s3 = b2 + b3;
			uint32_t mn = s0;
			if(s1 < mn) { mn = s1; ca = 1; }
			if(s2 < mn) { mn = s2; ca = 2; }
			if(s3 < mn) { mn = s3; ca = 3; }
			ca = (uint32_t)__builtin_amdgcn_readfirstlane((int)ca);
			l = ca == 2 ? 3 : ca == 3 ? 2 : 0;
			r = ca == 0 ? 1 : ca == 2 ? 1 : 3;
		}
#define FFLOAD(D, i) do { const uint32_t *rec = sh->rec[i]; D.which = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec[0]); D.type = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec[1]); \
		D.order = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec[2]); D.wasted = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec[3]); D.sbps = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec[4]); \
		D.bits = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec[5]); D.po = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec[6]); D.rice2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec[7]); \
		D.constant = __builtin_amdgcn_readfirstlane((int)rec[8]); } while(0)
		FFLOAD(DL, l); FFLOAD(DR, r);
#undef FFLOAD
	}

	// ---- the frame's length, before a bit of it is written: with the fused output it goes out now, long before anybody asks ------
	const uint32_t frame_number = (uint32_t)(first_frame_number + f);
	const uint32_t pos0 = 8 * frame_header_len(P, n, frame_number);
	uint32_t uL[FF_RUN], uR[FF_RUN];
	FFSub SL, SR;
	ff_subframe_sizes<WIDE>(w, DL, sh, lane, uL, SL);
	ff_subframe_sizes<WIDE>(w, DR, sh, lane, uR, SR);
	const uint32_t pos = pos0 + SL.bits + SR.bits;
	const uint32_t body_bytes = (pos + 7) >> 3, total_bytes = body_bytes + 2;
	const bool overflow = total_bytes > P.slot_bytes;
	const uint32_t mine = overflow ? 0u : total_bytes;
	uint64_t before = 0;
	if(publish && lane == 0) before = fo_publish(O, f, mine);

	// ---- the frame -----------------------------------------------------------------------------------------------------------
	uint32_t *img = tile;
	const uint32_t cap_words = P.slot_bytes / 4;
	{
		// (slot_bytes is a multiple of 16, and the image has 16 bytes to spare behind it: whole 16-byte stores)
		uint4 *img4 = (uint4 *)(smem + IMG_OFF - P2_IMG_PAD);          // (and the pad in front of it: frame_crc16_end)
		for(uint32_t q = (uint32_t)lane; q < P2_IMG_PAD / 16 + cap_words / 4 + 1; q += 64) img4[q] = make_uint4(0, 0, 0, 0);
	}
	__builtin_amdgcn_wave_barrier();
	{
		// the frame header: scalar code, then lanes 0..3 store a word each (the image is all zeros; the subframes OR into it behind this)
		uint32_t hw[4];
		(void)frame_header_words(P, n, ca, frame_number, hw);
		if(lane < 4) img[lane] = lane == 0 ? hw[0] : lane == 1 ? hw[1] : lane == 2 ? hw[2] : hw[3];
		__builtin_amdgcn_wave_barrier();
	}
	ff_subframe_write<WIDE>(img, IMG_OFF, cap_words, pos0, w, DL, uL, SL, lane);
	ff_subframe_write<WIDE>(img, IMG_OFF, cap_words, pos0 + SL.bits, w, DR, uR, SR, lane);
	if(lane == 0 && info) {
#pragma unroll
		for(int s = 0; s < 2; s++) {
			const FFDec &d = s ? DR : DL;
			flacgpu_subframe_info *si = &info[f].sub[s];
			si->type = (uint8_t)d.type; si->order = (uint8_t)d.order; si->wasted_bits = (uint8_t)d.wasted;
			si->partition_order = (uint8_t)d.po; si->rice2 = (uint8_t)d.rice2; si->precision = 0; si->shift = 0;
			si->pad = 0; si->bits = d.bits;
		}
	}
	__builtin_amdgcn_wave_barrier();
	// ---- zero-pad to a byte, CRC-16, footer (stream_encoder.c:3720-3734) --------------------------------------------------------
	{
		const uint32_t crc = frame_crc16_end<64, 13>(img, overflow ? 0 : body_bytes, sh->crc_parts, lane, sh->xspan, FF_XSPAN, g_ff_span.x);
		if(lane == 0) or_bits(img, cap_words, body_bytes * 8, crc, 16);
		__syncthreads();
	}
	if(publish && lane == 0) fo_close_segment(O, f, nmain, mine, before);    // (the counter has had the write phase and the CRC to come back)
	uint8_t *slot = slots + (size_t)f * P.slot_bytes;
	const uint32_t words = (umin32(total_bytes, P.slot_bytes) + 3) >> 2;
	if(O.lag) {
		// (another wavefront of this launch will read the slot: write-through stores, 16 bytes a lane)
		const fo_u4 *img4 = (const fo_u4 *)img;
		for(uint32_t q = (uint32_t)lane; q < (words + 3) >> 2; q += 64) {
			fo_u4 v = img4[q];
			v.x = __builtin_bswap32(v.x); v.y = __builtin_bswap32(v.y); v.z = __builtin_bswap32(v.z); v.w = __builtin_bswap32(v.w);
			*(fo_gv4)(slot + 16 * q) = v;
		}
		// ... and once they have left (MI355X_MICROARCH.md: write-through payload -> vmcnt(0) -> flag) the frame's word says so
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		if(lane == 0) fo_store(&O.fstate[f], fo_word(O.epoch, (uint64_t)mine | FO_STORED));
	}
	else {
		uint32_t *dst = (uint32_t *)slot;
		for(uint32_t w = (uint32_t)lane; w < words; w += 64) dst[w] = __builtin_bswap32(img[w]);
	}
	if(lane == 0) {
		frame_bytes[f] = overflow ? 0xffffffffu : total_bytes;
		if(info) info[f].channel_assignment = (uint8_t)ca;
	}
	// ---- the compaction rides along: this wavefront places the frame `lag` frames back, whose length, predecessors and bytes have
	// been out for a dozen microseconds -- nothing to wait for (PackOut).  The last `lag` frames of the batch are fo_place_kernel's.
	if(O.lag && f >= O.lag) {
		const uint32_t g = f - O.lag;
		uint64_t off = 0;
		uint32_t nb = 0;
		if(fo_exclusive<true>(O, g, lane, off, &nb)) {
			if(nb && off + nb <= O.cap) fo_copy_slot(slots + (size_t)g * P.slot_bytes, O.out + off, nb, lane);
			if(lane == 0) O.offsets[g] = off;
		}
		else if(lane == 0) O.fall[atomicAdd(&O.nfall[O.epoch & 1u], 1u)] = g;
	}
}

// fo_place_kernel: frames that sit in their slots go to their places in the stream; their offsets are plain sums of published
// words (every length is out and every segment closed once the pack kernel is done).  LISTED: the frames of PackOut::fall (the
// ones that gave up waiting -- in a normal run none: a fixed grid of idle workgroups); else the frames [first, nmain) of the batch,
// a workgroup each (behind ff_kernel, which places all but its last PackOut::lag frames itself: no scan kernel, and a compaction
// kernel for an eighth of the batch).
constexpr int FO_PLACE_GRID = 64;
template <bool LISTED>
__global__ __launch_bounds__(TPB) void fo_place_kernel(const PackOut O, uint32_t first, uint32_t nmain, const uint8_t *__restrict__ slots, uint32_t slot_bytes, const uint32_t *__restrict__ frame_bytes)
{
	__shared__ uint64_t part[TPB / 64];
	const int tid = (int)threadIdx.x;
	uint32_t count = nmain - first;
	if(LISTED) {
		count = __hip_atomic_load(&O.nfall[O.epoch & 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if(blockIdx.x == 0 && tid == 0) {
			__hip_atomic_store(&O.nfall[(O.epoch + 1u) & 1u], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // the next batch's counter
			// what the waits cost is invisible in the bytes: frames that gave up are counted for flacgpu_fused_fallbacks
			if(count) { atomicAdd(&O.nfall[2], count); atomicMax(&O.nfall[3], count); }
		}
	}
	for(uint32_t i = blockIdx.x; i < count; i += gridDim.x) {
		const uint32_t f = LISTED ? O.fall[i] : first + i;
		const uint32_t seg = f >> 6;
		if(LISTED) __syncthreads();                                           // (part[0] of the pass before has been read)
		if(tid < 64) {
			// one wavefront adds up: the totals of the segments in front (all loads in flight at once), the lengths in front in its own
			uint64_t s = (uint32_t)tid < (f & 63u) ? fo_load(&O.fstate[(size_t)seg * 64 + (uint32_t)tid]) & FO_LEN : 0ull;
			for(uint32_t g = (uint32_t)tid; g < seg; g += 64) s += fo_load(&O.sstate[g]) & FO_VAL;
			s = wave_reduce_add_u64(s);
			if(tid == 0) part[0] = s;
		}
		__syncthreads();
		const uint64_t off = part[0];
		const uint32_t nbr = frame_bytes[f], nb = nbr == 0xffffffffu ? 0u : nbr;
		if(nb && off + nb <= O.cap) {
			// head bytes up to 4-byte alignment of the destination, then aligned words assembled from two source words (compact_kernel)
			const uint8_t *src = slots + (size_t)f * slot_bytes;
			uint8_t *dst = O.out + off;
			const uint32_t head = umin32((uint32_t)((4 - ((uintptr_t)dst & 3)) & 3), nb);
			if((uint32_t)tid < head) dst[tid] = src[tid];
			const uint32_t words = (nb - head) >> 2, sh = head * 8;
			const uint32_t *sw = (const uint32_t *)src;
			uint32_t *dw = (uint32_t *)(dst + head);
			for(uint32_t w = (uint32_t)tid; w < words; w += TPB) dw[w] = sh == 0 ? sw[w] : (sw[w] >> sh) | (sw[w + 1] << (32 - sh));
			const uint32_t done = head + words * 4;
			if((uint32_t)tid < nb - done) dst[done + (uint32_t)tid] = src[done + (uint32_t)tid];
		}
		if(tid == 0) {
			O.offsets[f] = off;
			if(f + 1 == nmain) { O.offsets[f + 1] = off + nb; *O.total = off + nb; }
		}
	}
}

// fused output with a short last block: that frame was assembled in its slot by pack_kernel; it goes behind the others
__global__ __launch_bounds__(TPB) void append_tail_kernel(const uint8_t *__restrict__ slot, const uint32_t *__restrict__ frame_bytes, uint32_t f, const PackOut O)
{
	const uint64_t prev = f ? *O.total : 0;                 // (the last frame of nominal length has left the stream's length so far)
	const uint32_t nbr = frame_bytes[f], nb = nbr == 0xffffffffu ? 0u : nbr;
	if(nb && prev + nb <= O.cap) {
		uint8_t *dst = O.out + prev;
		for(uint32_t k = threadIdx.x; k < nb; k += TPB) dst[k] = slot[k];
	}
	if(threadIdx.x == 0) { O.offsets[f] = prev; O.offsets[f + 1] = prev + nb; *O.total = prev + nb; }
}

// ---------------------------------------------------------------------------------------------
// scan + compaction
// ---------------------------------------------------------------------------------------------
// Exclusive prefix sum of the frame lengths.  A workgroup of 1024 threads takes a CHUNK of 1024 frames (one per thread), and the
// chunks do not wait for each other: a chunk first adds up, on its own, everything in front of it (a batch of 16384 frames is 16
// chunks: fifteen loads per thread at most, all in flight at once), then scans its own lengths and writes its offsets.  Round 3's
// version was ONE workgroup walking the batch 16 lengths a thread: 12 us of latency in front of the compaction of a -0 step that
// takes 150; this one is 4-5.  (The redundant sums grow with the square of the batch: 65536 frames are 64 chunks, 63 loads per
// thread in the last one -- still microseconds.)
constexpr int SCAN_T = 1024;
__global__ __launch_bounds__(SCAN_T) void scan_kernel(const uint32_t *__restrict__ frame_bytes, uint32_t nframes,
                                                    uint64_t *__restrict__ offsets, uint64_t *__restrict__ total)
{
	__shared__ uint64_t wave_front[SCAN_T / 64], wave_own[SCAN_T / 64];
	const int tid = (int)threadIdx.x, wave = tid >> 6;
	const uint32_t c0 = blockIdx.x * SCAN_T, i = c0 + (uint32_t)tid;
	// this thread's share of everything in front of the chunk (an overflowing frame's 0xffffffff counts as nothing, as in the compaction)
	uint64_t front = 0;
	for(uint32_t g = (uint32_t)tid; g < c0; g += SCAN_T) { const uint32_t v = frame_bytes[g]; front += v == 0xffffffffu ? 0u : v; }
	uint32_t mine = i < nframes ? frame_bytes[i] : 0u;
	if(mine == 0xffffffffu) mine = 0;
	front = wave_reduce_add_u64(front);
	uint64_t incl = mine;
#pragma unroll
	for(int off = 1; off < 64; off <<= 1) {
		const uint32_t lo = __shfl_up((uint32_t)incl, off), hi = __shfl_up((uint32_t)(incl >> 32), off);
		if((tid & 63) >= off) incl += ((uint64_t)hi << 32) | lo;
	}
	if((tid & 63) == 63) { wave_front[wave] = front; wave_own[wave] = incl; }
	__syncthreads();
	uint64_t o = incl - mine;
	for(int w = 0; w < SCAN_T / 64; w++) { o += wave_front[w]; if(w < wave) o += wave_own[w]; }
	if(i < nframes) offsets[i] = o;
	if(i + 1 == nframes) { offsets[nframes] = o + mine; *total = o + mine; }
}

__global__ __launch_bounds__(TPB) void compact_kernel(const uint8_t *__restrict__ slots, uint32_t slot_bytes,
                                                      const uint32_t *__restrict__ frame_bytes,
                                                      const uint64_t *__restrict__ offsets,
                                                      uint8_t *__restrict__ out, uint64_t out_cap)
{
	const uint32_t f = blockIdx.x;
	const uint32_t nb = frame_bytes[f];
	if(nb == 0xffffffffu) return;
	const uint64_t off = offsets[f];
	if(off + nb > out_cap) return;
	const uint8_t *src = slots + (size_t)f * slot_bytes;
	uint8_t *dst = out + off;
	// head bytes up to 4-byte alignment of dst, then aligned words assembled from two source words
	const uint32_t mis = (uint32_t)((4 - ((uintptr_t)dst & 3)) & 3);
	const uint32_t head = umin32(mis, nb);
	if(threadIdx.x < head) dst[threadIdx.x] = src[threadIdx.x];
	const uint32_t words = (nb - head) >> 2;
	const uint32_t *sw = (const uint32_t *)src;
	uint32_t *dw = (uint32_t *)(dst + head);
	const uint32_t sh = head * 8;
	for(uint32_t w = threadIdx.x; w < words; w += TPB) {
		uint32_t v;
		if(sh == 0) v = sw[w];
		else v = (sw[w] >> sh) | (sw[w + 1] << (32 - sh));
		dw[w] = v;
	}
	const uint32_t done = head + words * 4;
	if(threadIdx.x < nb - done) dst[done + threadIdx.x] = src[done + threadIdx.x];
}

// ---------------------------------------------------------------------------------------------
// crc_check_kernel: the CRC-16 footer of every frame of a batch (crc.c:376), for the self check (flacgpu_verify.hip).
// One workgroup per frame.  Frames sit at arbitrary byte offsets of the output stream, but a CRC that starts at zero does not
// see leading zero bytes: the frame is read as the ALIGNED words it lies in, with the bytes in front of it masked to zero, in
// 44-byte spans (the span algebra of frame_crc16_p2: a word at a time through four 256-entry tables kept in LDS, a lane's
// remainder shifted past the spans behind it, the short last span as words plus at most three bytes).
// (The first version of this kernel read bytes with a table in global memory, a 64-step dependent chain per lane: 0.60 ms per
// 16384 frames, as long as everything else the verify pass does.)
// ---------------------------------------------------------------------------------------------
// x^e mod P for the rare frame that is longer than the span table: square and multiply
__device__ inline uint32_t gf16_xpow(uint64_t e)
{
	uint32_t r = 1, b = 2;
	while(e) { if(e & 1) r = gf16_mul(r, b); b = gf16_mul(b, b); e >>= 1; }
	return r;
}
// per_frame_bad != null (the stream decoder, flacgpu_stream_decode.hip): frames of length 0xffffffff are none of this kernel's business,
// and a footer that does not match sets per_frame_bad[f] instead of the batch's first bad frame
__global__ __launch_bounds__(TPB) void crc_check_kernel(const uint8_t *__restrict__ frames, const uint32_t *__restrict__ frame_bytes,
                                                        const uint64_t *__restrict__ offsets, uint32_t nframes, VerifyState *__restrict__ state,
                                                        uint8_t *__restrict__ per_frame_bad)
{
	__shared__ uint16_t tab[4][256];
	__shared__ uint32_t parts[TPB / 64];
	const uint32_t f = blockIdx.x;
	const int tid = (int)threadIdx.x;
	const uint32_t len = frame_bytes[f];
	if(per_frame_bad && len == 0xffffffffu) return;
	if(len == 0xffffffffu || len < 3) { if(tid == 0) { if(per_frame_bad) per_frame_bad[f] = 1; else atomicMin(&state->first_bad, f); } return; }          // (the same for every thread)
	for(uint32_t w = (uint32_t)tid; w < 4 * 256 / 2; w += TPB) ((uint32_t *)tab)[w] = ((const uint32_t *)g_crc_tables.tab)[w];
	__syncthreads();
	const uint8_t *p = frames + offsets[f];
	const uint32_t mis = (uint32_t)((uintptr_t)p & 3), body = len - 2;
	const uint32_t *w0 = (const uint32_t *)(p - mis);
	const uint32_t total = mis + body;                                   // bytes of the aligned array that count
	const uint32_t nsp = (total + CRC2_SPAN - 1) / CRC2_SPAN;
	const uint32_t last_len = total - (nsp - 1) * CRC2_SPAN;             // 1..44
	uint32_t c = 0;                     // low half: whole spans, shifted among themselves; high half: the last span
	for(uint32_t sp = (uint32_t)tid; sp < nsp; sp += TPB) {
		const uint32_t *wp = w0 + (size_t)sp * CRC2_WORDS;
		const bool last = sp + 1 == nsp;
		const uint32_t nload = last ? (last_len + 3) >> 2 : CRC2_WORDS;   // (nothing is read behind the word that holds the frame's last byte)
		uint32_t w[CRC2_WORDS];
#pragma unroll
		for(int k = 0; k < (int)CRC2_WORDS; k++) w[k] = __builtin_bswap32(wp[(uint32_t)k < nload ? (uint32_t)k : nload - 1]);      // (index clamped, load unconditional:
		                                                                                                                         //  eleven loads in flight, not one after the other)
		if(sp == 0 && mis) w[0] &= 0xffffffffu >> (8 * mis);
		const uint32_t nw = last ? last_len >> 2 : CRC2_WORDS;
		uint32_t cs = 0, tailw = 0;
#pragma unroll
		for(int k = 0; k < (int)CRC2_WORDS; k++) {
			if((uint32_t)k < nw) {
				const uint32_t v = (cs << 16) ^ w[k];
				cs = (uint32_t)tab[3][v >> 24] ^ tab[2][(v >> 16) & 0xffu] ^ tab[1][(v >> 8) & 0xffu] ^ tab[0][v & 0xffu];
			}
			else if((uint32_t)k == nw) tailw = w[k];
		}
		if(last) {
			const uint32_t nb = last_len & 3u;
			for(uint32_t j = 0; j < nb; j++) {
				const uint32_t b = (tailw >> (24 - 8 * j)) & 0xffu;
				cs = ((cs << 8) & 0xffffu) ^ tab[0][(cs >> 8) ^ b];
			}
			c ^= cs << 16;
		}
		else {
			const uint32_t m = nsp - 2 - sp;                              // whole spans behind this one, then the last one
			c ^= m == 0 ? cs : gf16_mul(cs, m < CRC2_MAX_SPANS ? (uint32_t)g_crc_tables.xspan44[m] : gf16_xpow((uint64_t)m * CRC2_SPAN * 8));
		}
	}
#pragma unroll
	for(int off = 32; off >= 1; off >>= 1) c ^= __shfl_xor(c, off);
	if((tid & 63) == 0) parts[tid >> 6] = c;
	__syncthreads();
	if(tid == 0) {
		uint32_t crc = 0;
		for(int w = 0; w < TPB / 64; w++) crc ^= parts[w];
		crc = gf16_mul(crc & 0xffffu, g_crc_tables.xbyte[last_len]) ^ (crc >> 16);
		if(crc != (((uint32_t)p[body] << 8) | p[body + 1])) { if(per_frame_bad) per_frame_bad[f] = 1; else atomicMin(&state->first_bad, f); }
	}
}

} // namespace flacgpu

// ---------------------------------------------------------------------------------------------
// launch wrappers (called from flacgpu_api.cpp)
// ---------------------------------------------------------------------------------------------
using namespace flacgpu;

// pack2_kernel takes the frames of nominal length when every Rice partition is a whole number of 16-sample runs
static bool pack2_applicable(const DevParams &P)
{
	const uint32_t ps = P.blocksize >> P.max_po;
	return P.blocksize % CHUNK == 0 && ps >= (uint32_t)CHUNK && ps % CHUNK == 0 && ((size_t)ps << P.max_po) == P.blocksize && !P.wide_samples && !P.img_global && !P.stream_sig;
}
// the one-wavefront instance whose threads own 18-sample runs (pack2_kernel<., false, 64, 18>): blocks that are a whole number of its
// 1152-sample passes and whose Rice partitions are whole runs -- the presets' 1152 at any partition order down to 18 samples, 2304 at
// its 36 (round 6: at the LPC presets these blocks went to the general pack_kernel, as long as everything else together)
static bool pack2_run18_applicable(const DevParams &P)
{
	const uint32_t ps = P.blocksize >> P.max_po;
	return P.blocksize % 1152u == 0 && ps >= 18u && ps % 18u == 0 && ((size_t)ps << P.max_po) == P.blocksize && !P.wide_samples && !P.img_global && !P.stream_sig;
}
static PackOut make_pack_out(const PackOutArgs *po)
{
	PackOut O;
	memset(&O, 0, sizeof O);
	if(po && po->out) {
		O.out = po->out; O.cap = po->cap; O.offsets = po->offsets; O.total = po->total;
		O.fstate = po->fstate; O.sstate = po->sstate; O.sprefix = po->sprefix; O.scount = po->scount; O.fall = po->fall; O.nfall = po->nfall;
		O.epoch = po->epoch; O.spin_limit = po->spin_limit; O.lag = 0;
	}
	return O;
}
template <int MAXORD>
static hipError_t launch_pack_t(const DevParams &P, const int32_t *chan, uint32_t nframes, uint32_t tail_n, uint64_t first,
                                const SubDecision *dec, uint8_t *plan, uint8_t *slots, uint32_t *fb, FrameInfo *info, unsigned long long *dbg, size_t lds, const PackOutArgs *po, bool *fused_out,
                                uint32_t *hints, uint32_t *hinted_frames, hipStream_t s)
{
	static AttrFlags attr_set;
	if(AttrOnce once{attr_set}) {
		hipError_t e = hipFuncSetAttribute((const void *)pack_kernel<MAXORD>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
		if constexpr(MAXORD <= 16) {
			if(e == hipSuccess) e = hipFuncSetAttribute((const void *)pack2_kernel<MAXORD, false, TPB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
			if(e == hipSuccess) e = hipFuncSetAttribute((const void *)pack2_kernel<MAXORD, true, TPB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
			if(e == hipSuccess) e = hipFuncSetAttribute((const void *)pack2_kernel<MAXORD, false, TPB / 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
			if(e == hipSuccess) e = hipFuncSetAttribute((const void *)pack2_kernel<MAXORD, true, TPB / 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
			if(e == hipSuccess) e = hipFuncSetAttribute((const void *)pack2_kernel<MAXORD, false, 64, 18>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
			if(e == hipSuccess) e = hipFuncSetAttribute((const void *)pack2_kernel<MAXORD, false, 128, 18>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
			if(e == hipSuccess) e = hipFuncSetAttribute((const void *)pack2_kernel<MAXORD, false, 256, 18>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
		}
		if(e != hipSuccess) return e;
		once.ok();
	}
	uint32_t f_lo = 0;
	bool fused = false;
	if(hinted_frames) *hinted_frames = 0;
	PackOut O = make_pack_out(nullptr);
	if constexpr(MAXORD <= 16) {                  // predictors of more than 16 taps (-l 17..32) take the general kernel
		const bool run18_ok = pack2_run18_applicable(P) && !hints && !tune().no_run18;
		if(pack2_applicable(P) || run18_ok) {
			f_lo = tail_n ? nframes - 1 : nframes;
			const size_t lds2 = ((sizeof(Pack2Shared) + 15) & ~(size_t)15) + P2_IMG_PAD + (size_t)P.slot_bytes + 16;
			if(f_lo && po && po->out) {
				// frames written once, at their final place: no slots, no scan / compact kernels
				fused = true;
				O = make_pack_out(po);
			}
			// half the threads for blocks that half of them cover in one pass (the 1152-sample blocks of -0 .. -2: 72 runs)
			const bool half = P.blocksize <= CHUNK * (TPB / 2);
			// the presets' 1152-sample blocks: 64 runs of 18 samples, one wavefront per frame (not with the verify hints: their decoder
			// counts in 16-sample runs; FLACGPU_NO_RUN18=1: the 128-thread instance, for A/B runs)
			const bool no_run18 = tune().no_run18 != 0;
			const uint32_t pstride = (uint32_t)pack_plan_stride(P);
			if(f_lo) hipLaunchKernelGGL(pack_plan_kernel, dim3((f_lo + PLAN_FRAMES - 1) / PLAN_FRAMES), dim3(64), 0, s, P, make_hdr_const(P), f_lo, first, dec, plan, pstride, info);
			const bool run18 = run18_ok && (P.blocksize == 1152 || !pack2_applicable(P));      // (1152: always; larger blocks: where the 16-sample instances do not apply)
			(void)no_run18;
			if(f_lo) note_launch(K_PACK_PLAN | K_PACK2 | (run18 ? K_PACK2_RUN18 : 0u) | (fused ? K_FO_PLACE : 0u));
			// (round 6: blocks of two / four and more 1152-sample passes -- 2304; 4608, 3456 ... -- on two / four wavefronts: the one-wavefront
			//  instance walked them pass by pass, 5.1 at -b 4608 packed at 2.2x the time per sample of -b 4096; FLACGPU_NO_RUN18W=1: as before)
			const uint32_t passes18 = P.blocksize / 1152u;
			if(f_lo && run18 && passes18 >= 4 && !tune().no_run18w) hipLaunchKernelGGL((pack2_kernel<MAXORD, false, 256, 18>), dim3(f_lo), dim3(256), lds2, s, P, chan, f_lo, plan, pstride, dec, slots, fb, dbg, O, hints);
			else if(f_lo && run18 && passes18 >= 2 && !tune().no_run18w) hipLaunchKernelGGL((pack2_kernel<MAXORD, false, 128, 18>), dim3(f_lo), dim3(128), lds2, s, P, chan, f_lo, plan, pstride, dec, slots, fb, dbg, O, hints);
			else if(f_lo && run18) hipLaunchKernelGGL((pack2_kernel<MAXORD, false, 64, 18>), dim3(f_lo), dim3(64), lds2, s, P, chan, f_lo, plan, pstride, dec, slots, fb, dbg, O, hints);
			else if(f_lo && hints && half) hipLaunchKernelGGL((pack2_kernel<MAXORD, true, TPB / 2>), dim3(f_lo), dim3(TPB / 2), lds2, s, P, chan, f_lo, plan, pstride, dec, slots, fb, dbg, O, hints);
			else if(f_lo && hints) hipLaunchKernelGGL((pack2_kernel<MAXORD, true, TPB>), dim3(f_lo), dim3(TPB), lds2, s, P, chan, f_lo, plan, pstride, dec, slots, fb, dbg, O, hints);
			else if(f_lo && half) hipLaunchKernelGGL((pack2_kernel<MAXORD, false, TPB / 2>), dim3(f_lo), dim3(TPB / 2), lds2, s, P, chan, f_lo, plan, pstride, dec, slots, fb, dbg, O, hints);
			else if(f_lo) hipLaunchKernelGGL((pack2_kernel<MAXORD, false, TPB>), dim3(f_lo), dim3(TPB), lds2, s, P, chan, f_lo, plan, pstride, dec, slots, fb, dbg, O, hints);
			if(hinted_frames) *hinted_frames = hints ? f_lo : 0;
			if(fused) hipLaunchKernelGGL(fo_place_kernel<true>, dim3(FO_PLACE_GRID), dim3(TPB), 0, s, O, 0u, f_lo, slots, P.slot_bytes, fb);
		}
	}
	if(f_lo < nframes) note_launch(K_PACK | (fused ? K_APPEND_TAIL : 0u));
	if(f_lo < nframes) hipLaunchKernelGGL(pack_kernel<MAXORD>, dim3(nframes - f_lo), dim3(TPB), lds, s, P, chan, nframes, tail_n, f_lo, first, dec, slots, fb, info);
	if(fused && f_lo < nframes) hipLaunchKernelGGL(append_tail_kernel, dim3(1), dim3(TPB), 0, s, slots + (size_t)f_lo * P.slot_bytes, fb, f_lo, O);
	if(fused_out) *fused_out = fused;
	sync_debug("pack", s);
	return hipGetLastError();
}

namespace flacgpu {
size_t pack_lds_bytes(const DevParams &P)
{
	if(P.img_global) return (size_t)pack_pass_sig_bytes(P) + sizeof(PackShared);
	const size_t a = (size_t)pack_pass_sig_bytes(P) + P.slot_bytes + sizeof(PackShared), b = ((sizeof(Pack2Shared) + 15) & ~(size_t)15) + P2_IMG_PAD + (size_t)P.slot_bytes + 16;
	return a > b ? a : b;
}

hipError_t launch_pack(const DevParams &P, const int32_t *chan, uint32_t nframes, uint32_t tail_n, uint64_t first,
                       const SubDecision *dec, uint8_t *plan, uint8_t *slots, uint32_t *fb, FrameInfo *info, unsigned long long *dbg, const PackOutArgs *po, bool *fused_out,
                       uint32_t *hints, uint32_t *hinted_frames, hipStream_t s)
{
	const size_t lds = pack_lds_bytes(P);
	const uint32_t m = P.max_lpc_order > 4 ? P.max_lpc_order : 4;
	if(P.blocksize > HINT_RUNS * CHUNK) hints = nullptr;          // one run per thread and pass: blocks of up to 4096 samples
	if(m <= 8) return launch_pack_t<8>(P, chan, nframes, tail_n, first, dec, plan, slots, fb, info, dbg, lds, po, fused_out, hints, hinted_frames, s);
	if(m <= 12) return launch_pack_t<12>(P, chan, nframes, tail_n, first, dec, plan, slots, fb, info, dbg, lds, po, fused_out, hints, hinted_frames, s);
	if(m <= 16) return launch_pack_t<16>(P, chan, nframes, tail_n, first, dec, plan, slots, fb, info, dbg, lds, po, fused_out, hints, hinted_frames, s);
	return launch_pack_t<32>(P, chan, nframes, tail_n, first, dec, plan, slots, fb, info, dbg, lds, po, fused_out, hints, hinted_frames, s);
}
// ff_kernel takes 16-bit stereo in 1152-sample blocks when the prep kernel could decide the subframes itself (no LPC search, one
// fixed order) and pack2_kernel could pack them; not with the verify hints (their decoder wants pack2_kernel's run starts)
bool ff_applicable(const DevParams &P)
{
	const bool off = tune().no_ff != 0;
	if(P.bps > 16 && (P.bps > 24 || tune().no_wide_ff)) return false;         // (17..24 bits: ff_kernel<., ., true>, round 6)
	return !off && P.channels == 2 && P.blocksize == FF_N && P.ncand == (P.ms_mode == 1 ? 4u : 2u) && prep2_decides(P) && pack2_applicable(P)
	       && (1152u >> P.max_po) % 18u == 0 && P.slot_bytes <= 52 * (P.bps <= 16 ? FF_XSPAN : FF_XSPAN_ALL) - 64;      // (every span shift of a 16-bit frame in the LDS table, of a wider one in the global table)
}
hipError_t launch_ff(const DevParams &P, const int32_t *pcm, uint32_t nmain, uint64_t first, uint8_t *slots, uint32_t *fb, FrameInfo *info, const PackOutArgs *po, hipStream_t s)
{
	if(nmain == 0) return hipSuccess;
	const size_t lds = ((sizeof(FFShared) + 15) & ~(size_t)15) + P2_IMG_PAD + ff_tile_bytes(P.slot_bytes);
	// How the frames get to their places.  Default: every frame to its slot, scan_kernel + compact_kernel behind this one (the caller,
	// with po == null).  FLACGPU_FF_LAG=n (opt-in, po != null): the kernel publishes its lengths and the wavefront of frame f places
	// frame f - n, whose length, predecessors and bytes have been out for a dozen microseconds; the last n frames, and with n = 0 all
	// of them, are fo_place_kernel's.  Measured (profiles/archive/r04_h_ff_lag_ab.txt, 16384 frames, ms per step, two kernels / lag 2048 /
	// lag 0): -0 0.156 / 0.158 / 0.161, -1 0.164 / 0.165 / 0.169, -2 0.194 / 0.185 / 0.195 -- a kernel of one-wavefront workgroups
	// that live 26 us pays for every extra round trip (the write-through slot stores, the loads of the frame it places) with a
	// wavefront slot that is not computing; only -2, whose wavefronts live longer, comes out ahead.  And a wavefront that places its
	// OWN frame and waits for the lengths in front of it (what pack2_kernel does) was the slowest: 0.167 at -0.
	note_launch(K_FF | (po && po->out ? K_FO_PLACE : 0u));
	const PackOut Oplace = make_pack_out(po);
	PackOut O = Oplace;
	if(po && po->out) { O.lag = po->lag < nmain ? po->lag : 0u; if(!O.lag) O.out = nullptr; }       // (lag 0: publish only)
	const bool preset_po = P.max_po == 3 && P.min_po == 0;               // (stream_encoder.c:117-133: what the presets -0 .. -2 set)
#define FFGO(MS_) do { if(P.bps > 16) hipLaunchKernelGGL((ff_kernel<MS_, -1, true>), dim3(nmain), dim3(64), lds, s, P, pcm, nmain, first, slots, fb, info, O); \
                       else if(preset_po) hipLaunchKernelGGL((ff_kernel<MS_, 3>), dim3(nmain), dim3(64), lds, s, P, pcm, nmain, first, slots, fb, info, O); \
                       else hipLaunchKernelGGL((ff_kernel<MS_, -1>), dim3(nmain), dim3(64), lds, s, P, pcm, nmain, first, slots, fb, info, O); } while(0)
	if(P.ms_mode == 0) FFGO(0);
	else if(P.ms_mode == 1) FFGO(1);
	else FFGO(2);
#undef FFGO
	if(po && po->out) {
		const uint32_t firstp = O.lag ? nmain - O.lag : 0u;
		hipLaunchKernelGGL(fo_place_kernel<false>, dim3(nmain - firstp), dim3(TPB), 0, s, Oplace, firstp, nmain, slots, P.slot_bytes, fb);
		hipLaunchKernelGGL(fo_place_kernel<true>, dim3(FO_PLACE_GRID), dim3(TPB), 0, s, Oplace, 0u, nmain, slots, P.slot_bytes, fb);
	}
	sync_debug("ff", s);
	return hipGetLastError();
}
hipError_t launch_append_tail(const uint8_t *slot, const uint32_t *fb, uint32_t f, const PackOutArgs *po, hipStream_t s)
{
	hipLaunchKernelGGL(append_tail_kernel, dim3(1), dim3(TPB), 0, s, slot, fb, f, make_pack_out(po));
	return hipGetLastError();
}
hipError_t launch_crc_check(const uint8_t *frames, const uint32_t *fb, const uint64_t *offsets, uint32_t nframes, VerifyState *state, hipStream_t s, uint8_t *per_frame_bad)
{
	hipLaunchKernelGGL(crc_check_kernel, dim3(nframes), dim3(TPB), 0, s, frames, fb, offsets, nframes, state, per_frame_bad);
	return hipGetLastError();
}
hipError_t launch_scan(const uint32_t *fb, uint32_t nframes, uint64_t *offsets, uint64_t *total, hipStream_t s)
{
	if(nframes == 0) return hipSuccess;
	note_launch(K_SCAN);
	hipLaunchKernelGGL(scan_kernel, dim3((nframes + SCAN_T - 1) / SCAN_T), dim3(SCAN_T), 0, s, fb, nframes, offsets, total);
	return hipGetLastError();
}
hipError_t launch_compact(const uint8_t *slots, uint32_t slot_bytes, const uint32_t *fb, const uint64_t *offsets,
                          uint8_t *out, uint64_t out_cap, uint32_t nframes, hipStream_t s)
{
	note_launch(K_COMPACT);
	hipLaunchKernelGGL(compact_kernel, dim3(nframes), dim3(TPB), 0, s, slots, slot_bytes, fb, offsets, out, out_cap);
	return hipGetLastError();
}
} // namespace flacgpu
