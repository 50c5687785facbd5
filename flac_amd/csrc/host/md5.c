/* flac_amd/csrc/host/md5.c -- MD5 (RFC 1321) of the unencoded audio for STREAMINFO.
 * The reference hashes the interleaved samples as little-endian integers of ceil(bps/8) bytes
 * (src/libFLAC/md5.c:280 format_input_, :497 FLAC__MD5Accumulate; stream_encoder.c:3448). MD5 is a
 * serial chain per stream, so it stays on a host thread. Written from the RFC. */
#include <string.h>
#include "flacgpu_host.h"

/* the 64 steps unrolled, rounds as RFC 1321 section 3.4 writes them: the chain is serial, so the only speed there is
 * to be had is in not computing table indices and branches per step */
/* b (the x argument) is the value that arrives last in every step: everything that does not depend on it is added to `a`
 * first, and round 2 takes its selection apart -- (z & x) | (~z & y) has disjoint terms, so they can be ADDED separately and
 * only (z & x) waits for x */
#define F1(x, y, z) ((z) ^ ((x) & ((y) ^ (z))))
#define F3(x, y, z) ((x) ^ ((y) ^ (z)))
#define F4(x, y, z) ((y) ^ ((x) | ~(z)))
#define ROL(v, s) (((v) << (s)) | ((v) >> (32 - (s))))
#define STEP(f, a, b, c, d, w, k, s) do { (a) += (w) + (k); (a) += f((b), (c), (d)); (a) = ROL((a), (s)); (a) += (b); } while(0)
#define STEP2(a, b, c, d, w, k, s) do { (a) += (w) + (k) + (~(d) & (c)); (a) += ((d) & (b)); (a) = ROL((a), (s)); (a) += (b); } while(0)

static void md5_block(uint32_t st[4], const uint8_t *p)
{
	uint32_t w[16], a = st[0], b = st[1], c = st[2], d = st[3];
	for(int i = 0; i < 16; i++)
		w[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) | ((uint32_t)p[4 * i + 3] << 24);
	STEP(F1, a, b, c, d, w[0], 0xd76aa478, 7);   STEP(F1, d, a, b, c, w[1], 0xe8c7b756, 12);  STEP(F1, c, d, a, b, w[2], 0x242070db, 17);  STEP(F1, b, c, d, a, w[3], 0xc1bdceee, 22);
	STEP(F1, a, b, c, d, w[4], 0xf57c0faf, 7);   STEP(F1, d, a, b, c, w[5], 0x4787c62a, 12);  STEP(F1, c, d, a, b, w[6], 0xa8304613, 17);  STEP(F1, b, c, d, a, w[7], 0xfd469501, 22);
	STEP(F1, a, b, c, d, w[8], 0x698098d8, 7);   STEP(F1, d, a, b, c, w[9], 0x8b44f7af, 12);  STEP(F1, c, d, a, b, w[10], 0xffff5bb1, 17); STEP(F1, b, c, d, a, w[11], 0x895cd7be, 22);
	STEP(F1, a, b, c, d, w[12], 0x6b901122, 7);  STEP(F1, d, a, b, c, w[13], 0xfd987193, 12); STEP(F1, c, d, a, b, w[14], 0xa679438e, 17); STEP(F1, b, c, d, a, w[15], 0x49b40821, 22);
	STEP2(a, b, c, d, w[1], 0xf61e2562, 5);   STEP2(d, a, b, c, w[6], 0xc040b340, 9);   STEP2(c, d, a, b, w[11], 0x265e5a51, 14); STEP2(b, c, d, a, w[0], 0xe9b6c7aa, 20);
	STEP2(a, b, c, d, w[5], 0xd62f105d, 5);   STEP2(d, a, b, c, w[10], 0x02441453, 9);  STEP2(c, d, a, b, w[15], 0xd8a1e681, 14); STEP2(b, c, d, a, w[4], 0xe7d3fbc8, 20);
	STEP2(a, b, c, d, w[9], 0x21e1cde6, 5);   STEP2(d, a, b, c, w[14], 0xc33707d6, 9);  STEP2(c, d, a, b, w[3], 0xf4d50d87, 14);  STEP2(b, c, d, a, w[8], 0x455a14ed, 20);
	STEP2(a, b, c, d, w[13], 0xa9e3e905, 5);  STEP2(d, a, b, c, w[2], 0xfcefa3f8, 9);   STEP2(c, d, a, b, w[7], 0x676f02d9, 14);  STEP2(b, c, d, a, w[12], 0x8d2a4c8a, 20);
	STEP(F3, a, b, c, d, w[5], 0xfffa3942, 4);   STEP(F3, d, a, b, c, w[8], 0x8771f681, 11);  STEP(F3, c, d, a, b, w[11], 0x6d9d6122, 16); STEP(F3, b, c, d, a, w[14], 0xfde5380c, 23);
	STEP(F3, a, b, c, d, w[1], 0xa4beea44, 4);   STEP(F3, d, a, b, c, w[4], 0x4bdecfa9, 11);  STEP(F3, c, d, a, b, w[7], 0xf6bb4b60, 16);  STEP(F3, b, c, d, a, w[10], 0xbebfbc70, 23);
	STEP(F3, a, b, c, d, w[13], 0x289b7ec6, 4);  STEP(F3, d, a, b, c, w[0], 0xeaa127fa, 11);  STEP(F3, c, d, a, b, w[3], 0xd4ef3085, 16);  STEP(F3, b, c, d, a, w[6], 0x04881d05, 23);
	STEP(F3, a, b, c, d, w[9], 0xd9d4d039, 4);   STEP(F3, d, a, b, c, w[12], 0xe6db99e5, 11); STEP(F3, c, d, a, b, w[15], 0x1fa27cf8, 16); STEP(F3, b, c, d, a, w[2], 0xc4ac5665, 23);
	STEP(F4, a, b, c, d, w[0], 0xf4292244, 6);   STEP(F4, d, a, b, c, w[7], 0x432aff97, 10);  STEP(F4, c, d, a, b, w[14], 0xab9423a7, 15); STEP(F4, b, c, d, a, w[5], 0xfc93a039, 21);
	STEP(F4, a, b, c, d, w[12], 0x655b59c3, 6);  STEP(F4, d, a, b, c, w[3], 0x8f0ccc92, 10);  STEP(F4, c, d, a, b, w[10], 0xffeff47d, 15); STEP(F4, b, c, d, a, w[1], 0x85845dd1, 21);
	STEP(F4, a, b, c, d, w[8], 0x6fa87e4f, 6);   STEP(F4, d, a, b, c, w[15], 0xfe2ce6e0, 10); STEP(F4, c, d, a, b, w[6], 0xa3014314, 15);  STEP(F4, b, c, d, a, w[13], 0x4e0811a1, 21);
	STEP(F4, a, b, c, d, w[4], 0xf7537e82, 6);   STEP(F4, d, a, b, c, w[11], 0xbd3af235, 10); STEP(F4, c, d, a, b, w[2], 0x2ad7d2bb, 15);  STEP(F4, b, c, d, a, w[9], 0xeb86d391, 21);
	st[0] += a; st[1] += b; st[2] += c; st[3] += d;
}

void flacgpu_host_md5_init(flacgpu_host_md5 *m)
{
	m->state[0] = 0x67452301; m->state[1] = 0xefcdab89; m->state[2] = 0x98badcfe; m->state[3] = 0x10325476;
	m->nbytes = 0;
}

void flacgpu_host_md5_update(flacgpu_host_md5 *m, const void *data, size_t len)
{
	const uint8_t *p = (const uint8_t *)data;
	size_t fill = (size_t)(m->nbytes & 63);
	m->nbytes += len;
	if(fill) {
		size_t take = 64 - fill;
		if(take > len) take = len;
		memcpy(m->block + fill, p, take);
		p += take; len -= take; fill += take;
		if(fill < 64) return;
		md5_block(m->state, m->block);
	}
	while(len >= 64) { md5_block(m->state, p); p += 64; len -= 64; }
	if(len) memcpy(m->block, p, len);
}

void flacgpu_host_md5_final(flacgpu_host_md5 *m, uint8_t digest[16])
{
	const uint64_t bits = m->nbytes * 8;
	uint8_t pad[72];
	size_t fill = (size_t)(m->nbytes & 63), padlen = (fill < 56 ? 56 - fill : 120 - fill);
	memset(pad, 0, sizeof pad);
	pad[0] = 0x80;
	for(int i = 0; i < 8; i++) pad[padlen + i] = (uint8_t)(bits >> (8 * i));
	flacgpu_host_md5_update(m, pad, padlen + 8);
	for(int i = 0; i < 4; i++)
		for(int j = 0; j < 4; j++) digest[4 * i + j] = (uint8_t)(m->state[i] >> (8 * j));
}

void flacgpu_host_md5_pcm(flacgpu_host_md5 *m, const int32_t *x, uint32_t channels, size_t samples, uint32_t bytes_per_sample)
{
	uint8_t buf[4096];
	size_t fill = 0;
	const size_t total = samples * channels;
	for(size_t i = 0; i < total; i++) {
		const uint32_t v = (uint32_t)x[i];
		for(uint32_t b = 0; b < bytes_per_sample; b++) buf[fill++] = (uint8_t)(v >> (8 * b));
		if(fill + 4 > sizeof buf) { flacgpu_host_md5_update(m, buf, fill); fill = 0; }
	}
	if(fill) flacgpu_host_md5_update(m, buf, fill);
}

/* ------------------------------------------------------------------------------------------------------------------------
 * Eight chains at once.  One MD5 chain is serial (64 dependent steps per 64 bytes: ~1 GB/s on this host whatever one does), but a
 * corpus is many streams, each with its own STREAMINFO digest, and eight independent chains fit the eight 32-bit lanes of an AVX2
 * register: the same 64 steps, each on eight states.  flacgpu_host_md5_x8_blocks() advances eight contexts by the same number of
 * whole blocks; flacgpu_host_md5_many() hashes any number of buffers, eight at a time over their common length, the rest of each
 * on the single-chain routine.  Same digests as FLAC__MD5Final (tests/test_kat.py, tests/test_md5_multi.py).
 * ---------------------------------------------------------------------------------------------------------------------- */
#if defined(__x86_64__) && defined(__GNUC__)
#include <immintrin.h>
#define MD5_HAVE_X8 1
#define VROL(v, s) _mm256_or_si256(_mm256_slli_epi32((v), (s)), _mm256_srli_epi32((v), 32 - (s)))
#define VADD(x, y) _mm256_add_epi32((x), (y))
#define VF1(x, y, z) _mm256_xor_si256((z), _mm256_and_si256((x), _mm256_xor_si256((y), (z))))
#define VF2(x, y, z) _mm256_xor_si256((y), _mm256_and_si256((z), _mm256_xor_si256((x), (y))))
#define VF3(x, y, z) _mm256_xor_si256((x), _mm256_xor_si256((y), (z)))
#define VF4(x, y, z) _mm256_xor_si256((y), _mm256_or_si256((x), _mm256_xor_si256((z), ones)))
#define VSTEP(f, a, b, c, d, w, k, s) do { (a) = VADD(VADD((a), VADD((w), _mm256_set1_epi32((int)(k)))), f((b), (c), (d))); (a) = VADD(VROL((a), (s)), (b)); } while(0)

/* the 64 steps of RFC 1321 3.4, written once for both vector widths */
#define MD5_64_STEPS(ST, F1, F2, F3, F4) \
	ST(F1, a, b, c, d, w[0], 0xd76aa478, 7);   ST(F1, d, a, b, c, w[1], 0xe8c7b756, 12);  ST(F1, c, d, a, b, w[2], 0x242070db, 17);  ST(F1, b, c, d, a, w[3], 0xc1bdceee, 22); \
	ST(F1, a, b, c, d, w[4], 0xf57c0faf, 7);   ST(F1, d, a, b, c, w[5], 0x4787c62a, 12);  ST(F1, c, d, a, b, w[6], 0xa8304613, 17);  ST(F1, b, c, d, a, w[7], 0xfd469501, 22); \
	ST(F1, a, b, c, d, w[8], 0x698098d8, 7);   ST(F1, d, a, b, c, w[9], 0x8b44f7af, 12);  ST(F1, c, d, a, b, w[10], 0xffff5bb1, 17); ST(F1, b, c, d, a, w[11], 0x895cd7be, 22); \
	ST(F1, a, b, c, d, w[12], 0x6b901122, 7);  ST(F1, d, a, b, c, w[13], 0xfd987193, 12); ST(F1, c, d, a, b, w[14], 0xa679438e, 17); ST(F1, b, c, d, a, w[15], 0x49b40821, 22); \
	ST(F2, a, b, c, d, w[1], 0xf61e2562, 5);   ST(F2, d, a, b, c, w[6], 0xc040b340, 9);   ST(F2, c, d, a, b, w[11], 0x265e5a51, 14); ST(F2, b, c, d, a, w[0], 0xe9b6c7aa, 20); \
	ST(F2, a, b, c, d, w[5], 0xd62f105d, 5);   ST(F2, d, a, b, c, w[10], 0x02441453, 9);  ST(F2, c, d, a, b, w[15], 0xd8a1e681, 14); ST(F2, b, c, d, a, w[4], 0xe7d3fbc8, 20); \
	ST(F2, a, b, c, d, w[9], 0x21e1cde6, 5);   ST(F2, d, a, b, c, w[14], 0xc33707d6, 9);  ST(F2, c, d, a, b, w[3], 0xf4d50d87, 14);  ST(F2, b, c, d, a, w[8], 0x455a14ed, 20); \
	ST(F2, a, b, c, d, w[13], 0xa9e3e905, 5);  ST(F2, d, a, b, c, w[2], 0xfcefa3f8, 9);   ST(F2, c, d, a, b, w[7], 0x676f02d9, 14);  ST(F2, b, c, d, a, w[12], 0x8d2a4c8a, 20); \
	ST(F3, a, b, c, d, w[5], 0xfffa3942, 4);   ST(F3, d, a, b, c, w[8], 0x8771f681, 11);  ST(F3, c, d, a, b, w[11], 0x6d9d6122, 16); ST(F3, b, c, d, a, w[14], 0xfde5380c, 23); \
	ST(F3, a, b, c, d, w[1], 0xa4beea44, 4);   ST(F3, d, a, b, c, w[4], 0x4bdecfa9, 11);  ST(F3, c, d, a, b, w[7], 0xf6bb4b60, 16);  ST(F3, b, c, d, a, w[10], 0xbebfbc70, 23); \
	ST(F3, a, b, c, d, w[13], 0x289b7ec6, 4);  ST(F3, d, a, b, c, w[0], 0xeaa127fa, 11);  ST(F3, c, d, a, b, w[3], 0xd4ef3085, 16);  ST(F3, b, c, d, a, w[6], 0x04881d05, 23); \
	ST(F3, a, b, c, d, w[9], 0xd9d4d039, 4);   ST(F3, d, a, b, c, w[12], 0xe6db99e5, 11); ST(F3, c, d, a, b, w[15], 0x1fa27cf8, 16); ST(F3, b, c, d, a, w[2], 0xc4ac5665, 23); \
	ST(F4, a, b, c, d, w[0], 0xf4292244, 6);   ST(F4, d, a, b, c, w[7], 0x432aff97, 10);  ST(F4, c, d, a, b, w[14], 0xab9423a7, 15); ST(F4, b, c, d, a, w[5], 0xfc93a039, 21); \
	ST(F4, a, b, c, d, w[12], 0x655b59c3, 6);  ST(F4, d, a, b, c, w[3], 0x8f0ccc92, 10);  ST(F4, c, d, a, b, w[10], 0xffeff47d, 15); ST(F4, b, c, d, a, w[1], 0x85845dd1, 21); \
	ST(F4, a, b, c, d, w[8], 0x6fa87e4f, 6);   ST(F4, d, a, b, c, w[15], 0xfe2ce6e0, 10); ST(F4, c, d, a, b, w[6], 0xa3014314, 15);  ST(F4, b, c, d, a, w[13], 0x4e0811a1, 21); \
	ST(F4, a, b, c, d, w[4], 0xf7537e82, 6);   ST(F4, d, a, b, c, w[11], 0xbd3af235, 10); ST(F4, c, d, a, b, w[2], 0x2ad7d2bb, 15);  ST(F4, b, c, d, a, w[9], 0xeb86d391, 21);

__attribute__((target("avx2")))
static void transpose8(__m256i r[8])
{
	const __m256i t0 = _mm256_unpacklo_epi32(r[0], r[1]), t1 = _mm256_unpackhi_epi32(r[0], r[1]), t2 = _mm256_unpacklo_epi32(r[2], r[3]), t3 = _mm256_unpackhi_epi32(r[2], r[3]);
	const __m256i t4 = _mm256_unpacklo_epi32(r[4], r[5]), t5 = _mm256_unpackhi_epi32(r[4], r[5]), t6 = _mm256_unpacklo_epi32(r[6], r[7]), t7 = _mm256_unpackhi_epi32(r[6], r[7]);
	const __m256i u0 = _mm256_unpacklo_epi64(t0, t2), u1 = _mm256_unpackhi_epi64(t0, t2), u2 = _mm256_unpacklo_epi64(t1, t3), u3 = _mm256_unpackhi_epi64(t1, t3);
	const __m256i u4 = _mm256_unpacklo_epi64(t4, t6), u5 = _mm256_unpackhi_epi64(t4, t6), u6 = _mm256_unpacklo_epi64(t5, t7), u7 = _mm256_unpackhi_epi64(t5, t7);
	r[0] = _mm256_permute2x128_si256(u0, u4, 0x20); r[1] = _mm256_permute2x128_si256(u1, u5, 0x20); r[2] = _mm256_permute2x128_si256(u2, u6, 0x20); r[3] = _mm256_permute2x128_si256(u3, u7, 0x20);
	r[4] = _mm256_permute2x128_si256(u0, u4, 0x31); r[5] = _mm256_permute2x128_si256(u1, u5, 0x31); r[6] = _mm256_permute2x128_si256(u2, u6, 0x31); r[7] = _mm256_permute2x128_si256(u3, u7, 0x31);
}
__attribute__((target("avx2")))
static void md5_x8_blocks(uint32_t *st[8], const uint8_t *const p[8], size_t nblocks)
{
	const __m256i ones = _mm256_set1_epi32(-1);
	__m256i S[8];
	for(int k = 0; k < 8; k++) S[k] = _mm256_castsi128_si256(_mm_loadu_si128((const __m128i *)st[k]));
	transpose8(S);                                   /* S[0..3] = a, b, c, d of the eight chains */
	__m256i A = S[0], B = S[1], C = S[2], D = S[3];
	for(size_t blk = 0; blk < nblocks; blk++) {
		__m256i w[16];
		for(int k = 0; k < 8; k++) { w[k] = _mm256_loadu_si256((const __m256i *)(p[k] + 64 * blk)); w[8 + k] = _mm256_loadu_si256((const __m256i *)(p[k] + 64 * blk + 32)); }
		transpose8(w); transpose8(w + 8);             /* w[i] = word i of the eight blocks (x86 is little endian, as MD5 is) */
		__m256i a = A, b = B, c = C, d = D;
		MD5_64_STEPS(VSTEP, VF1, VF2, VF3, VF4)
		A = VADD(A, a); B = VADD(B, b); C = VADD(C, c); D = VADD(D, d);
	}
	S[0] = A; S[1] = B; S[2] = C; S[3] = D;
	S[4] = S[5] = S[6] = S[7] = _mm256_setzero_si256();
	transpose8(S);
	for(int k = 0; k < 8; k++) _mm_storeu_si128((__m128i *)st[k], _mm256_castsi256_si128(S[k]));
}

/* Sixteen chains in the sixteen 32-bit lanes of an AVX-512 register: the round functions are one vpternlogd each, the rotation one
 * vprold -- 3 operations on the dependent path of a step instead of 6 -- for a corpus of sixteen or more streams per thread. */
#define WADD(x, y) _mm512_add_epi32((x), (y))
#define WF1(x, y, z) _mm512_ternarylogic_epi32((x), (y), (z), 0xCA)      /* (x & y) | (~x & z) */
#define WF2(x, y, z) _mm512_ternarylogic_epi32((x), (y), (z), 0xE4)      /* (x & z) | (y & ~z) */
#define WF3(x, y, z) _mm512_ternarylogic_epi32((x), (y), (z), 0x96)      /* x ^ y ^ z */
#define WF4(x, y, z) _mm512_ternarylogic_epi32((x), (y), (z), 0x39)      /* y ^ (x | ~z) */
#define WSTEP(f, a, b, c, d, w, k, s) do { (a) = WADD(WADD((a), WADD((w), _mm512_set1_epi32((int)(k)))), f((b), (c), (d))); (a) = WADD(_mm512_rol_epi32((a), (s)), (b)); } while(0)
__attribute__((target("avx512f")))
static void transpose16(__m512i r[16])
{
	__m512i t[16], u[16];
	for(int i = 0; i < 16; i += 2) { t[i] = _mm512_unpacklo_epi32(r[i], r[i + 1]); t[i + 1] = _mm512_unpackhi_epi32(r[i], r[i + 1]); }
	for(int i = 0; i < 16; i += 4) {
		u[i] = _mm512_unpacklo_epi64(t[i], t[i + 2]); u[i + 1] = _mm512_unpackhi_epi64(t[i], t[i + 2]);
		u[i + 2] = _mm512_unpacklo_epi64(t[i + 1], t[i + 3]); u[i + 3] = _mm512_unpackhi_epi64(t[i + 1], t[i + 3]);
	}
	/* u[4g + j] holds, in 128-bit lane q, the words 4q + j of rows 4g .. 4g + 3: gather lane q of the four row groups */
	for(int j = 0; j < 4; j++) {
		const __m512i a = _mm512_shuffle_i32x4(u[j], u[4 + j], 0x88), b = _mm512_shuffle_i32x4(u[8 + j], u[12 + j], 0x88);      /* lanes 0, 2 */
		const __m512i c = _mm512_shuffle_i32x4(u[j], u[4 + j], 0xDD), d = _mm512_shuffle_i32x4(u[8 + j], u[12 + j], 0xDD);      /* lanes 1, 3 */
		r[j] = _mm512_shuffle_i32x4(a, b, 0x88); r[8 + j] = _mm512_shuffle_i32x4(a, b, 0xDD);
		r[4 + j] = _mm512_shuffle_i32x4(c, d, 0x88); r[12 + j] = _mm512_shuffle_i32x4(c, d, 0xDD);
	}
}
__attribute__((target("avx512f")))
static void md5_x16_blocks(uint32_t *st[16], const uint8_t *const p[16], size_t nblocks)
{
	__m512i S[16];
	for(int k = 0; k < 16; k++) S[k] = _mm512_castsi128_si512(_mm_loadu_si128((const __m128i *)st[k]));
	transpose16(S);                                  /* S[0..3] = a, b, c, d of the sixteen chains */
	__m512i A = S[0], B = S[1], C = S[2], D = S[3];
	for(size_t blk = 0; blk < nblocks; blk++) {
		__m512i w[16];
		for(int k = 0; k < 16; k++) w[k] = _mm512_loadu_si512((const void *)(p[k] + 64 * blk));
		transpose16(w);                              /* w[i] = word i of the sixteen blocks */
		__m512i a = A, b = B, c = C, d = D;
		MD5_64_STEPS(WSTEP, WF1, WF2, WF3, WF4)
		A = WADD(A, a); B = WADD(B, b); C = WADD(C, c); D = WADD(D, d);
	}
	S[0] = A; S[1] = B; S[2] = C; S[3] = D;
	for(int k = 4; k < 16; k++) S[k] = _mm512_setzero_si512();
	transpose16(S);
	for(int k = 0; k < 16; k++) _mm_storeu_si128((__m128i *)st[k], _mm512_castsi512_si128(S[k]));
}
#endif

int flacgpu_host_md5_x8_available(void)
{
#ifdef MD5_HAVE_X8
	return __builtin_cpu_supports("avx2") ? 1 : 0;
#else
	return 0;
#endif
}
/* advance eight contexts, each at a block boundary (nbytes % 64 == 0), by nblocks whole blocks of their data */
void flacgpu_host_md5_x8_blocks(flacgpu_host_md5 *const m[8], const void *const data[8], size_t nblocks)
{
	if(nblocks == 0) return;
#ifdef MD5_HAVE_X8
	int aligned = 1;
	for(int k = 0; k < 8; k++) if(m[k]->nbytes & 63) aligned = 0;
	if(aligned && flacgpu_host_md5_x8_available()) {
		uint32_t *st[8];
		const uint8_t *p[8];
		for(int k = 0; k < 8; k++) { st[k] = m[k]->state; p[k] = (const uint8_t *)data[k]; }
		md5_x8_blocks(st, p, nblocks);
		for(int k = 0; k < 8; k++) m[k]->nbytes += 64 * (uint64_t)nblocks;
		return;
	}
#endif
	for(int k = 0; k < 8; k++) flacgpu_host_md5_update(m[k], data[k], 64 * nblocks);
}
int flacgpu_host_md5_x16_available(void)
{
#ifdef MD5_HAVE_X8
	return __builtin_cpu_supports("avx512f") ? 1 : 0;
#else
	return 0;
#endif
}
/* W (8 or 16) buffers: their common whole blocks W chains at a time, the rest of each on the single-chain routine */
static void md5_group(const void *const *data, const size_t *len, uint32_t w, uint8_t (*digest)[16])
{
	flacgpu_host_md5 ctx[16];
	size_t common = (size_t)-1;
	for(uint32_t k = 0; k < w; k++) { flacgpu_host_md5_init(&ctx[k]); if(len[k] / 64 < common) common = len[k] / 64; }
#ifdef MD5_HAVE_X8
	if(w == 16 && common && flacgpu_host_md5_x16_available()) {
		uint32_t *st[16];
		const uint8_t *p[16];
		for(int k = 0; k < 16; k++) { st[k] = ctx[k].state; p[k] = (const uint8_t *)data[k]; }
		md5_x16_blocks(st, p, common);
		for(int k = 0; k < 16; k++) ctx[k].nbytes += 64 * (uint64_t)common;
	}
	else
#endif
	if(w == 8) {
		flacgpu_host_md5 *mp[8];
		const void *dp[8];
		for(int k = 0; k < 8; k++) { mp[k] = &ctx[k]; dp[k] = data[k]; }
		flacgpu_host_md5_x8_blocks(mp, dp, common);
	}
	else common = 0;
	for(uint32_t k = 0; k < w; k++) {
		flacgpu_host_md5_update(&ctx[k], (const uint8_t *)data[k] + 64 * common, len[k] - 64 * common);
		flacgpu_host_md5_final(&ctx[k], digest[k]);
	}
}
/* MD5 of n buffers: digest[i] = MD5(data[i][0 .. len[i])) */
void flacgpu_host_md5_many(const void *const *data, const size_t *len, uint32_t n, uint8_t (*digest)[16])
{
	uint32_t i = 0;
	if(flacgpu_host_md5_x16_available()) for(; i + 16 <= n; i += 16) md5_group(data + i, len + i, 16, digest + i);
	for(; i + 8 <= n; i += 8) md5_group(data + i, len + i, 8, digest + i);
	for(; i < n; i++) md5_group(data + i, len + i, 1, digest + i);
}

/* The same on `nthreads` host threads (a corpus job hashes its tracks while the GPU encodes them, straight from the buffer the
 * samples were read into).  A chain is serial, so the unit of work is a GROUP of chains: sixteen wide (AVX-512) when that still
 * gives every thread a group, else eight wide (AVX2), single chains for what is left; threads take groups from a shared counter,
 * longest first. */
#include <pthread.h>
#include <stdlib.h>
typedef struct { const void *const *data; const size_t *len; uint8_t (*digest)[16]; const uint32_t *first, *width; uint32_t ngroups; uint32_t next; pthread_mutex_t mu; } md5_mt_job;
static void *md5_mt_worker(void *arg)
{
	md5_mt_job *j = (md5_mt_job *)arg;
	for(;;) {
		pthread_mutex_lock(&j->mu);
		const uint32_t g = j->next < j->ngroups ? j->next++ : (uint32_t)-1;
		pthread_mutex_unlock(&j->mu);
		if(g == (uint32_t)-1) return NULL;
		md5_group(j->data + j->first[g], j->len + j->first[g], j->width[g], j->digest + j->first[g]);
	}
}
void flacgpu_host_md5_many_mt(const void *const *data, const size_t *len, uint32_t n, uint8_t (*digest)[16], uint32_t nthreads)
{
	if(nthreads <= 1 || n < 2) { flacgpu_host_md5_many(data, len, n, digest); return; }
	if(nthreads > 256) nthreads = 256;
	uint32_t w = flacgpu_host_md5_x8_available() ? 8 : 1;
	if(flacgpu_host_md5_x16_available() && n / 16 >= nthreads) w = 16;
	if(w == 8 && n / 8 < nthreads / 2) w = 1;             /* fewer groups than half the threads: a chain per thread is the faster split */
	uint32_t *first = (uint32_t *)malloc(sizeof(uint32_t) * 2 * (size_t)n), *width = first ? first + n : NULL;
	if(!first) { flacgpu_host_md5_many(data, len, n, digest); return; }
	uint32_t ng = 0, i = 0;
	for(; w > 1 && i + w <= n; i += w) { first[ng] = i; width[ng] = w; ng++; }
	for(; i < n; i++) { first[ng] = i; width[ng] = 1; ng++; }
	md5_mt_job job = { data, len, digest, first, width, ng, 0, PTHREAD_MUTEX_INITIALIZER };
	if(nthreads > ng) nthreads = ng;
	pthread_t th[256];
	uint32_t started = 0;
	for(uint32_t t = 0; t + 1 < nthreads; t++) { if(pthread_create(&th[started], NULL, md5_mt_worker, &job) == 0) started++; }
	md5_mt_worker(&job);                                 /* the calling thread works, too */
	for(uint32_t t = 0; t < started; t++) pthread_join(th[t], NULL);
	pthread_mutex_destroy(&job.mu);
	free(first);
}
