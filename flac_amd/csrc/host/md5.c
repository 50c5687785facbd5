/* flac_amd/csrc/host/md5.c -- MD5 (RFC 1321) of the unencoded audio for STREAMINFO.
 * The reference hashes the interleaved samples as little-endian integers of ceil(bps/8) bytes
 * (src/libFLAC/md5.c:280 format_input_, :497 FLAC__MD5Accumulate; stream_encoder.c:3448). MD5 is a
 * serial chain per stream, so it stays on a host thread. Written from the RFC. */
#include <string.h>
#include "flacgpu_host.h"

static const uint32_t K[64] = {
	0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501,
	0x698098d8, 0x8b44f7af, 0xffff5bb1, 0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821,
	0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453, 0xd8a1e681, 0xe7d3fbc8,
	0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a,
	0xfffa3942, 0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70,
	0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05, 0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665,
	0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d, 0x85845dd1,
	0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391
};
static const uint8_t S[64] = {
	7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20,
	4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21
};

static void md5_block(uint32_t st[4], const uint8_t *p)
{
	uint32_t w[16], a = st[0], b = st[1], c = st[2], d = st[3];
	for(int i = 0; i < 16; i++)
		w[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) | ((uint32_t)p[4 * i + 3] << 24);
	for(int i = 0; i < 64; i++) {
		uint32_t f, g;
		if(i < 16) { f = (b & c) | (~b & d); g = (uint32_t)i; }
		else if(i < 32) { f = (d & b) | (~d & c); g = (5u * i + 1) & 15; }
		else if(i < 48) { f = b ^ c ^ d; g = (3u * i + 5) & 15; }
		else { f = c ^ (b | ~d); g = (7u * i) & 15; }
		const uint32_t t = a + f + K[i] + w[g];
		a = d; d = c; c = b;
		b = b + ((t << S[i]) | (t >> (32 - S[i])));
	}
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
