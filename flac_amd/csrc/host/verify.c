/* flac_amd/csrc/host/verify.c -- the encoder's self check (FLAC__stream_encoder_set_verify): every frame the GPU produced
 * is decoded again on the host and compared with the samples that went in, as the reference does with its own stream
 * decoder in write_bitbuffer_ (src/libFLAC/stream_encoder.c:3000-3018, verify_write_callback_ :5130-5190).
 * A frame decoder written from the format (the reference's reader is
 * src/libFLAC/stream_decoder.c:2070-2890, the restoration lpc.c:978-1578 / fixed.c:571-667): frame header with UTF-8
 * frame number and CRC-8, the four subframe types, wasted bits, partitioned Rice / Rice2 residual incl. escape
 * partitions, inter-channel decorrelation, zero padding, CRC-16.  Frames are independent, so a batch is verified by a
 * few threads in parallel. */
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include "flacgpu_host.h"

typedef struct { const uint8_t *p; size_t len, pos; uint64_t acc; unsigned nbits; int bad; } bitr;

static void br_init(bitr *b, const uint8_t *p, size_t len) { b->p = p; b->len = len; b->pos = 0; b->acc = 0; b->nbits = 0; b->bad = 0; }
static inline void br_fill(bitr *b)
{
	while(b->nbits <= 56 && b->pos < b->len) { b->acc |= (uint64_t)b->p[b->pos++] << (56 - b->nbits); b->nbits += 8; }
}
static inline uint32_t br_bits(bitr *b, unsigned n)           /* n <= 32 */
{
	if(n == 0) return 0;
	if(b->nbits < n) { br_fill(b); if(b->nbits < n) { b->bad = 1; return 0; } }
	const uint32_t v = (uint32_t)(b->acc >> (64 - n));
	b->acc <<= n; b->nbits -= n;                               /* n <= 32 */
	return v;
}
static inline int32_t br_sbits(bitr *b, unsigned n) { const uint32_t v = br_bits(b, n); return n >= 32 ? (int32_t)v : (int32_t)(v << (32 - n)) >> (32 - n); }
/* a sample of up to 33 bits (the side channel of a 32-bit stream) */
static inline int64_t br_sample(bitr *b, unsigned n)
{
	if(n <= 32) return br_sbits(b, n);
	const uint64_t hi = br_bits(b, n - 32), lo = br_bits(b, 32);
	const uint64_t v = (hi << 32) | lo;
	return (int64_t)(v << (64 - n)) >> (64 - n);
}
static inline uint32_t br_unary(bitr *b)                       /* zeros before the next 1 */
{
	uint32_t z = 0;
	for(;;) {
		if(b->nbits == 0) { br_fill(b); if(b->nbits == 0) { b->bad = 1; return z; } }
		if(b->acc == 0) { z += b->nbits; b->nbits = 0; continue; }
		const unsigned lz = (unsigned)__builtin_clzll(b->acc);
		if(lz >= b->nbits) { z += b->nbits; b->acc = 0; b->nbits = 0; continue; }
		z += lz;
		b->acc = lz == 63 ? 0 : b->acc << (lz + 1);          /* a shift by 64 is not defined */
		b->nbits -= lz + 1;
		return z;
	}
}
static size_t br_bitpos(const bitr *b) { return b->pos * 8 - b->nbits; }

static uint8_t crc8(const uint8_t *p, size_t n)
{
	uint32_t c = 0;
	for(size_t i = 0; i < n; i++) { c ^= p[i]; for(int k = 0; k < 8; k++) c = (c & 0x80) ? ((c << 1) ^ 0x07) & 0xff : (c << 1) & 0xff; }
	return (uint8_t)c;
}
static uint16_t crc16_tab[256];
static pthread_once_t crc16_once = PTHREAD_ONCE_INIT;
static void crc16_init(void)
{
	for(uint32_t v = 0; v < 256; v++) { uint32_t c = v << 8; for(int k = 0; k < 8; k++) c = (c & 0x8000) ? ((c << 1) ^ 0x8005) & 0xffff : (c << 1) & 0xffff; crc16_tab[v] = (uint16_t)c; }
}
static uint16_t crc16(const uint8_t *p, size_t n)
{
	uint32_t c = 0;
	for(size_t i = 0; i < n; i++) c = ((c << 8) & 0xffff) ^ crc16_tab[(c >> 8) ^ p[i]];
	return (uint16_t)c;
}

/* residual of one subframe (stream_encoder_framing.c:522-594 read backwards) */
static int read_residual(bitr *b, int32_t *r, uint32_t n, uint32_t order)
{
	const uint32_t method = br_bits(b, 2);
	if(method > 1) return 0;
	const uint32_t plen = method ? 5 : 4, esc = method ? 31 : 15;
	const uint32_t po = br_bits(b, 4);
	if((n >> po) << po != n || (n >> po) < order) { if(po) return 0; }
	uint32_t i = 0;
	for(uint32_t part = 0; part < (1u << po); part++) {
		uint32_t cnt = (n >> po) - (part == 0 ? order : 0);
		if(po == 0) cnt = n - order;
		const uint32_t k = br_bits(b, plen);
		if(k == esc) {
			const uint32_t raw = br_bits(b, 5);
			for(uint32_t j = 0; j < cnt; j++) r[i++] = raw ? br_sbits(b, raw) : 0;
		}
		else for(uint32_t j = 0; j < cnt; j++) {
			const uint32_t msbs = br_unary(b);
			const uint32_t u = (msbs << k) | br_bits(b, k);
			r[i++] = (int32_t)(u >> 1) ^ -(int32_t)(u & 1);
		}
		if(b->bad) return 0;
	}
	return i == n - order;
}

typedef struct {
	const flacgpu_host_settings *s;
	const uint8_t *frames; const uint32_t *frame_bytes; const uint64_t *offsets;
	const uint8_t *raw; uint32_t width;               /* the samples that went in: little endian, `width` bytes each, interleaved */
	uint32_t nframes, tail, first_frame;
	uint32_t t, nthreads;
	flacgpu_host_verify_result res;                   /* first problem this thread met (stream order) */
} vjob;

static inline int32_t expected_sample(const uint8_t *raw, uint32_t width, size_t idx)
{
	const uint8_t *q = raw + idx * width;
	uint32_t v = 0;
	for(uint32_t k = 0; k < width; k++) v |= (uint32_t)q[k] << (8 * k);
	return width >= 4 ? (int32_t)v : (int32_t)(v << (32 - 8 * width)) >> (32 - 8 * width);
}

/* returns 0 ok, 1 audio mismatch, 2 the frame does not decode */
static int verify_frame(const vjob *J, uint32_t f, int64_t *x /* [C][N] */, int32_t *r, flacgpu_host_verify_result *out)
{
	const flacgpu_host_settings *s = J->s;
	const uint32_t C = s->channels, N = s->blocksize, bps = s->bits_per_sample;
	const uint32_t n = (f + 1 == J->nframes && J->tail) ? J->tail : N;
	const uint8_t *p = J->frames + J->offsets[f];
	const size_t len = J->frame_bytes[f];
	memset(out, 0, sizeof *out);
	out->frame_number = J->first_frame + f;
	out->absolute_sample = (uint64_t)(J->first_frame + f) * N;
	if(len < 6 || crc16(p, len - 2) != (uint16_t)((p[len - 2] << 8) | p[len - 1])) return 2;
	bitr b;
	br_init(&b, p, len - 2);
	if(br_bits(&b, 15) != 0x7ffc || br_bits(&b, 1) != 0) return 2;           /* sync, reserved, fixed-blocksize stream */
	const uint32_t bs_code = br_bits(&b, 4), sr_code = br_bits(&b, 4), ca = br_bits(&b, 4), bps_code = br_bits(&b, 3);
	if(br_bits(&b, 1) != 0) return 2;
	uint64_t fn = 0;
	{
		const uint32_t b0 = br_bits(&b, 8);
		unsigned extra = 0;
		if(b0 < 0x80) fn = b0;
		else if((b0 & 0xe0) == 0xc0) { fn = b0 & 0x1f; extra = 1; }
		else if((b0 & 0xf0) == 0xe0) { fn = b0 & 0x0f; extra = 2; }
		else if((b0 & 0xf8) == 0xf0) { fn = b0 & 0x07; extra = 3; }
		else if((b0 & 0xfc) == 0xf8) { fn = b0 & 0x03; extra = 4; }
		else if((b0 & 0xfe) == 0xfc) { fn = b0 & 0x01; extra = 5; }
		else return 2;
		for(unsigned k = 0; k < extra; k++) { const uint32_t c = br_bits(&b, 8); if((c & 0xc0) != 0x80) return 2; fn = (fn << 6) | (c & 0x3f); }
	}
	uint32_t bs;
	switch(bs_code) {
		case 1: bs = 192; break; case 2: bs = 576; break; case 3: bs = 1152; break; case 4: bs = 2304; break; case 5: bs = 4608; break;
		case 6: bs = br_bits(&b, 8) + 1; break; case 7: bs = br_bits(&b, 16) + 1; break;
		default: if(bs_code >= 8) bs = 256u << (bs_code - 8); else return 2;
	}
	if(sr_code == 12) (void)br_bits(&b, 8); else if(sr_code == 13 || sr_code == 14) (void)br_bits(&b, 16); else if(sr_code == 15) return 2;
	const size_t hdr_bytes = br_bitpos(&b) / 8;
	if(br_bits(&b, 8) != crc8(p, hdr_bytes) || b.bad) return 2;
	static const uint32_t bps_tab[8] = {0, 8, 12, 0, 16, 20, 24, 32};
	if(fn != (uint64_t)(J->first_frame + f) || bs != n || (bps_code && bps_tab[bps_code] != bps)) return 2;
	if((ca < 8 && ca + 1 != C) || ca > 10 || (ca >= 8 && C != 2)) return 2;

	for(uint32_t ch = 0; ch < C; ch++) {
		int64_t *xc = x + (size_t)ch * N;
		if(br_bits(&b, 1) != 0) return 2;
		const uint32_t type = br_bits(&b, 6);
		uint32_t wasted = 0;
		if(br_bits(&b, 1)) wasted = br_unary(&b) + 1;
		const int side = (ca == 8 && ch == 1) || (ca == 9 && ch == 0) || (ca == 10 && ch == 1);
		if(wasted >= bps + (side ? 1u : 0u)) return 2;
		const uint32_t sb = bps - wasted + (side ? 1 : 0);
		if(type == 0) { const int64_t v = br_sample(&b, sb); for(uint32_t i = 0; i < n; i++) xc[i] = v; }
		else if(type == 1) for(uint32_t i = 0; i < n; i++) xc[i] = br_sample(&b, sb);
		else if(type >= 8 && type <= 12) {
			const uint32_t order = type - 8;
			if(order > n) return 2;
			for(uint32_t i = 0; i < order; i++) xc[i] = br_sample(&b, sb);
			if(!read_residual(&b, r, n, order)) return 2;
			for(uint32_t i = order; i < n; i++) {           /* fixed.c:571: the predictors are binomial FIRs */
				int64_t pr;
				switch(order) {
					case 0: pr = 0; break;
					case 1: pr = xc[i - 1]; break;
					case 2: pr = 2 * xc[i - 1] - xc[i - 2]; break;
					case 3: pr = 3 * xc[i - 1] - 3 * xc[i - 2] + xc[i - 3]; break;
					default: pr = 4 * xc[i - 1] - 6 * xc[i - 2] + 4 * xc[i - 3] - xc[i - 4]; break;
				}
				xc[i] = (int64_t)r[i - order] + pr;
			}
		}
		else if(type >= 32) {
			const uint32_t order = type - 31;
			int32_t q[32];
			if(order > n) return 2;
			for(uint32_t i = 0; i < order; i++) xc[i] = br_sample(&b, sb);
			const uint32_t prec = br_bits(&b, 4) + 1;
			if(prec == 16) return 2;
			const int32_t shift = br_sbits(&b, 5);
			if(shift < 0) return 2;
			for(uint32_t j = 0; j < order; j++) q[j] = br_sbits(&b, prec);
			if(!read_residual(&b, r, n, order)) return 2;
			for(uint32_t i = order; i < n; i++) {           /* lpc.c:978 */
				int64_t sum = 0;
				for(uint32_t j = 0; j < order; j++) sum += (int64_t)q[j] * xc[i - 1 - j];
				xc[i] = (int64_t)r[i - order] + (sum >> shift);
			}
		}
		else return 2;
		if(b.bad) return 2;
		if(wasted) for(uint32_t i = 0; i < n; i++) xc[i] = (int64_t)((uint64_t)xc[i] << wasted);
	}
	/* zero padding up to the byte boundary, and nothing but the CRC behind it */
	{
		const size_t bp = br_bitpos(&b);
		if(bp % 8) { if(br_bits(&b, (unsigned)(8 - bp % 8)) != 0) return 2; }
		if(b.bad || br_bitpos(&b) != (len - 2) * 8) return 2;
	}
	/* inter-channel decorrelation (stream_decoder.c:2240-2290) */
	if(ca == 8) for(uint32_t i = 0; i < n; i++) x[N + i] = x[i] - x[N + i];
	else if(ca == 9) for(uint32_t i = 0; i < n; i++) x[i] += x[N + i];
	else if(ca == 10) for(uint32_t i = 0; i < n; i++) {
		const int64_t sd = x[N + i];
		const int64_t mid = (int64_t)(((uint64_t)x[i] << 1) | ((uint64_t)sd & 1));
		x[i] = (mid + sd) >> 1; x[N + i] = (mid - sd) >> 1;
	}
	for(uint32_t i = 0; i < n; i++)
		for(uint32_t ch = 0; ch < C; ch++) {
			const int32_t want = expected_sample(J->raw, J->width, ((size_t)f * N + i) * C + ch);
			if(x[(size_t)ch * N + i] != (int64_t)want) {
				out->absolute_sample += i; out->channel = ch; out->sample = i; out->expected = want; out->got = (int32_t)x[(size_t)ch * N + i];
				return 1;
			}
		}
	return 0;
}

static void *vthread(void *arg)
{
	vjob *J = arg;
	const uint32_t C = J->s->channels, N = J->s->blocksize;
	int64_t *x = malloc(sizeof(int64_t) * ((size_t)C + 1) * N);
	J->res.status = 0;
	if(!x) { J->res.status = 2; J->res.frame_number = J->first_frame; return 0; }
	for(uint32_t f = J->t; f < J->nframes; f += J->nthreads) {
		flacgpu_host_verify_result r;
		const int st = verify_frame(J, f, x, (int32_t *)(x + (size_t)C * N), &r);
		if(st) { r.status = st; J->res = r; break; }           /* frames are visited in increasing order: the first problem of this thread */
	}
	free(x);
	return 0;
}

int flacgpu_host_verify_batch(const flacgpu_host_settings *s, const uint8_t *frames, const uint32_t *frame_bytes, uint32_t nframes, uint32_t tail,
                              uint32_t first_frame, const uint8_t *raw, uint32_t width, uint32_t nthreads, flacgpu_host_verify_result *out)
{
	pthread_once(&crc16_once, crc16_init);
	if(nthreads < 1) nthreads = 1;
	if(nthreads > 64) nthreads = 64;
	if(nthreads > nframes) nthreads = nframes;
	uint64_t *offsets = malloc(sizeof(uint64_t) * ((size_t)nframes + 1));
	vjob *jobs = calloc(nthreads, sizeof *jobs);
	pthread_t *th = calloc(nthreads, sizeof *th);
	memset(out, 0, sizeof *out);
	if(!offsets || !jobs || !th) { free(offsets); free(jobs); free(th); out->status = 2; return 2; }
	offsets[0] = 0;
	for(uint32_t f = 0; f < nframes; f++) offsets[f + 1] = offsets[f] + frame_bytes[f];
	for(uint32_t t = 0; t < nthreads; t++) {
		jobs[t].s = s; jobs[t].frames = frames; jobs[t].frame_bytes = frame_bytes; jobs[t].offsets = offsets; jobs[t].raw = raw; jobs[t].width = width;
		jobs[t].nframes = nframes; jobs[t].tail = tail; jobs[t].first_frame = first_frame; jobs[t].t = t; jobs[t].nthreads = nthreads;
	}
	/* thread t takes frames t, t+nthreads, ...; a thread that cannot be started has its share done here */
	for(uint32_t t = 1; t < nthreads; t++) if(pthread_create(&th[t], 0, vthread, &jobs[t]) != 0) th[t] = 0;
	(void)vthread(&jobs[0]);
	for(uint32_t t = 1; t < nthreads; t++) { if(th[t]) pthread_join(th[t], 0); else (void)vthread(&jobs[t]); }
	int have = 0;
	for(uint32_t t = 0; t < nthreads; t++)
		if(jobs[t].res.status && (!have || jobs[t].res.frame_number < out->frame_number)) { *out = jobs[t].res; have = 1; }
	free(offsets); free(jobs); free(th);
	return out->status;
}

/* CRC-16 recheck of a run of frames laid out back to back (the footer of every frame, crc.c:376 over everything in
 * front of it): the index of the first frame whose footer disagrees, or -1.  A corpus-sized stream is split over
 * `nthreads` threads by frame ranges.  Slicing by four bytes. */
static uint16_t crc16_t4[4][256];
static pthread_once_t crc16_t4_once = PTHREAD_ONCE_INIT;
static void crc16_t4_init(void)
{
	crc16_init();
	for(uint32_t v = 0; v < 256; v++) {
		uint32_t c = crc16_tab[v];
		crc16_t4[0][v] = (uint16_t)c;
		for(int k = 1; k < 4; k++) { c = ((c << 8) & 0xffff) ^ crc16_tab[c >> 8]; crc16_t4[k][v] = (uint16_t)c; }
	}
}
static uint16_t crc16_fast(const uint8_t *p, size_t n)
{
	uint32_t c = 0;
	while(n >= 4) {
		const uint32_t a = (c >> 8) ^ p[0], b = (c & 0xff) ^ p[1];
		c = crc16_t4[3][a] ^ crc16_t4[2][b] ^ crc16_t4[1][p[2]] ^ crc16_t4[0][p[3]];
		p += 4; n -= 4;
	}
	while(n--) c = ((c << 8) & 0xffff) ^ crc16_tab[(c >> 8) ^ *p++];
	return (uint16_t)c;
}
typedef struct { const uint8_t *frames; const uint32_t *fb; uint64_t lo, hi, off; int64_t bad; } crcjob;
static void *crc_thread(void *arg)
{
	crcjob *J = arg;
	const uint8_t *p = J->frames + J->off;
	J->bad = -1;
	for(uint64_t f = J->lo; f < J->hi; f++) {
		const size_t len = J->fb[f];
		if(len < 3 || crc16_fast(p, len - 2) != (uint16_t)((p[len - 2] << 8) | p[len - 1])) { J->bad = (int64_t)f; return 0; }
		p += len;
	}
	return 0;
}
int64_t flacgpu_host_check_frame_crcs(const uint8_t *frames, const uint32_t *frame_bytes, uint64_t nframes, uint32_t nthreads)
{
	pthread_once(&crc16_once, crc16_init);
	pthread_once(&crc16_t4_once, crc16_t4_init);
	if(nthreads < 1) nthreads = 1;
	if(nthreads > 64) nthreads = 64;
	if(nthreads > nframes) nthreads = nframes ? (uint32_t)nframes : 1;
	crcjob jobs[64];
	pthread_t th[64];
	uint64_t off = 0, f = 0;
	for(uint32_t t = 0; t < nthreads; t++) {
		jobs[t].frames = frames; jobs[t].fb = frame_bytes; jobs[t].lo = f; jobs[t].hi = nframes * (t + 1) / nthreads; jobs[t].off = off;
		for(; f < jobs[t].hi; f++) off += frame_bytes[f];
	}
	for(uint32_t t = 1; t < nthreads; t++) if(pthread_create(&th[t], 0, crc_thread, &jobs[t]) != 0) th[t] = 0;
	(void)crc_thread(&jobs[0]);
	int64_t bad = jobs[0].bad;
	for(uint32_t t = 1; t < nthreads; t++) {
		if(th[t]) pthread_join(th[t], 0); else (void)crc_thread(&jobs[t]);
		if(bad < 0 && jobs[t].bad >= 0) bad = jobs[t].bad;
	}
	return bad;
}
