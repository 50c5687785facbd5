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
