/* flac_amd/csrc/host/ogg.c -- the Ogg FLAC container around the frames (FLAC__stream_encoder_init_ogg_*).
 *
 * Two layers, both host C:
 *   1. the FLAC-to-Ogg mapping of the reference, src/libFLAC/ogg_encoder_aspect.c:95-250 (one packet per write callback:
 *      "fLaC" + STREAMINFO folded into the first packet behind the 0x7F "FLAC" 1.0 <header count> prefix, one packet per
 *      metadata block, one per audio frame; metadata packets are flushed to pages, audio packets are paged out) and the
 *      STREAMINFO fix-up of update_ogg_metadata_ (stream_encoder.c:3303-3440) through the client's read / seek / write
 *      callbacks;
 *   2. the paging itself, which the reference delegates to libogg (ogg_stream_packetin / ogg_stream_pageout /
 *      ogg_stream_flush, libogg 1.3.x src/framing.c).  libogg is NOT part of /root/reference, so this is a restatement of
 *      its published algorithm (RFC 3533 page layout; the page-closing rule of ogg_stream_flush_i: a page is closed by
 *      force, at 255 segments, or once more than 4096 body bytes are queued and at least four packets have ended on it;
 *      the first page carries only the first packet).
 *
 * How it is pinned: the reference here cannot be built with libogg (a build without it reports UNSUPPORTED_CONTAINER), but its
 * tree carries Ogg FLAC streams written by libFLAC + libogg (oss-fuzz/seedcorpus).  tests/test_ogg_cpu.py cuts their logical
 * streams into packets and pages them again with this code the way the mapping does: the same bytes come back, page for page
 * (boundaries incl. the 4096-byte / four-packet rule, flags, granule positions, sequence numbers, CRC-32).  Whole files can only
 * be checked structurally (tests/test_stream_encoder_api.py::test_ogg_flac_container: independent page parser and CRC, packets
 * == the native FLAC stream of the same encoder, the mapping rules, the STREAMINFO patched into the first page).
 */
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include "flacgpu_host.h"
#include "ogg.h"

/* ---- CRC-32 of Ogg pages: polynomial 0x04c11db7, no reflection, initial value and final xor 0 (RFC 3533) ---- */
static uint32_t crc_tab[256];
static pthread_once_t crc_once = PTHREAD_ONCE_INIT;
static void crc_init(void)
{
	for(uint32_t i = 0; i < 256; i++) {
		uint32_t r = i << 24;
		for(int k = 0; k < 8; k++) r = (r & 0x80000000u) ? (r << 1) ^ 0x04c11db7u : r << 1;
		crc_tab[i] = r;
	}
}
static uint32_t crc_update(uint32_t c, const uint8_t *p, size_t n)
{
	for(size_t i = 0; i < n; i++) c = (c << 8) ^ crc_tab[((c >> 24) & 0xff) ^ p[i]];
	return c;
}
void fgh_ogg_page_checksum_set(uint8_t *header, size_t header_len, const uint8_t *body, size_t body_len)
{
	pthread_once(&crc_once, crc_init);         /* encoders on different threads page concurrently */
	header[22] = header[23] = header[24] = header[25] = 0;
	uint32_t c = crc_update(0, header, header_len);
	c = crc_update(c, body, body_len);
	header[22] = (uint8_t)c; header[23] = (uint8_t)(c >> 8); header[24] = (uint8_t)(c >> 16); header[25] = (uint8_t)(c >> 24);
}

/* ---- packet queue -> pages (libogg framing.c, restated) ---- */
int fgh_ogg_stream_init(fgh_ogg_stream *os, long serialno)
{
	memset(os, 0, sizeof *os);
	os->body_storage = 16 * 1024;
	os->lacing_storage = 1024;
	os->body = malloc(os->body_storage);
	os->lacing = malloc(os->lacing_storage * sizeof *os->lacing);
	os->granule = malloc(os->lacing_storage * sizeof *os->granule);
	os->serialno = serialno;
	if(!os->body || !os->lacing || !os->granule) { fgh_ogg_stream_clear(os); return -1; }
	return 0;
}
void fgh_ogg_stream_clear(fgh_ogg_stream *os)
{
	free(os->body); free(os->lacing); free(os->granule);
	memset(os, 0, sizeof *os);
}
int fgh_ogg_stream_packetin(fgh_ogg_stream *os, const uint8_t *data, size_t bytes, int64_t granulepos, int e_o_s)
{
	const size_t nl = bytes / 255 + 1;
	if(!os->body) return -1;
	if(os->body_returned) {
		/* drop what the pages already handed out took */
		os->body_fill -= os->body_returned;
		if(os->body_fill) memmove(os->body, os->body + os->body_returned, os->body_fill);
		os->body_returned = 0;
	}
	if(os->body_storage - os->body_fill <= bytes) {
		const size_t ns = os->body_storage + bytes + 1024;
		uint8_t *nb = realloc(os->body, ns);
		if(!nb) return -1;
		os->body = nb; os->body_storage = ns;
	}
	if(os->lacing_storage - os->lacing_fill <= nl) {
		const size_t ns = os->lacing_storage + nl + 32;
		int *nlv = realloc(os->lacing, ns * sizeof *nlv);
		if(!nlv) return -1;
		os->lacing = nlv;
		int64_t *ng = realloc(os->granule, ns * sizeof *ng);
		if(!ng) return -1;
		os->granule = ng; os->lacing_storage = ns;
	}
	memcpy(os->body + os->body_fill, data, bytes);
	os->body_fill += bytes;
	size_t i;
	for(i = 0; i < nl - 1; i++) { os->lacing[os->lacing_fill + i] = 255; os->granule[os->lacing_fill + i] = os->granulepos; }
	os->lacing[os->lacing_fill + i] = (int)(bytes % 255);
	os->granulepos = os->granule[os->lacing_fill + i] = granulepos;
	os->lacing[os->lacing_fill] |= 0x100;                    /* the first segment of a packet */
	os->lacing_fill += nl;
	os->packetno++;
	if(e_o_s) os->e_o_s = 1;
	return 0;
}
/* 1: a page was produced (header in os->header / header_len, body at *body / *body_len, valid until the next packetin) */
static int flush_i(fgh_ogg_stream *os, int force, size_t nfill, const uint8_t **body, size_t *body_len)
{
	const size_t maxvals = os->lacing_fill > 255 ? 255 : os->lacing_fill;
	size_t vals = 0, bytes = 0, acc = 0;
	int64_t granule_pos = -1;
	if(!os->body || maxvals == 0) return 0;
	if(os->b_o_s == 0) {
		/* the initial header page holds the first packet and nothing else */
		granule_pos = 0;
		for(vals = 0; vals < maxvals; vals++) if((os->lacing[vals] & 0xff) < 255) { vals++; break; }
	}
	else {
		/* do not span pages needlessly, and do not close a page before four packets ended on it unless it has to be */
		int packets_done = 0, packet_just_done = 0;
		for(vals = 0; vals < maxvals; vals++) {
			if(acc > nfill && packet_just_done >= 4) { force = 1; break; }
			acc += (size_t)(os->lacing[vals] & 0xff);
			if((os->lacing[vals] & 0xff) < 255) { granule_pos = os->granule[vals]; packet_just_done = ++packets_done; }
			else packet_just_done = 0;
		}
		if(vals == 255) force = 1;
	}
	if(!force) return 0;
	uint8_t *h = os->header;
	memcpy(h, "OggS", 4);
	h[4] = 0;
	h[5] = 0;
	if((os->lacing[0] & 0x100) == 0) h[5] |= 0x01;                      /* continues a packet */
	if(os->b_o_s == 0) h[5] |= 0x02;                                    /* first page */
	if(os->e_o_s && os->lacing_fill == vals) h[5] |= 0x04;              /* last page */
	os->b_o_s = 1;
	{ uint64_t g = (uint64_t)granule_pos; for(int i = 6; i < 14; i++) { h[i] = (uint8_t)g; g >>= 8; } }
	{ uint32_t s = (uint32_t)os->serialno; for(int i = 14; i < 18; i++) { h[i] = (uint8_t)s; s >>= 8; } }
	{ uint32_t pn = (uint32_t)os->pageno++; for(int i = 18; i < 22; i++) { h[i] = (uint8_t)pn; pn >>= 8; } }
	h[22] = h[23] = h[24] = h[25] = 0;
	h[26] = (uint8_t)vals;
	for(size_t i = 0; i < vals; i++) { h[27 + i] = (uint8_t)(os->lacing[i] & 0xff); bytes += h[27 + i]; }
	os->header_len = 27 + vals;
	*body = os->body + os->body_returned;
	*body_len = bytes;
	os->lacing_fill -= vals;
	memmove(os->lacing, os->lacing + vals, os->lacing_fill * sizeof *os->lacing);
	memmove(os->granule, os->granule + vals, os->lacing_fill * sizeof *os->granule);
	os->body_returned += bytes;
	fgh_ogg_page_checksum_set(h, os->header_len, *body, *body_len);
	return 1;
}
int fgh_ogg_stream_flush(fgh_ogg_stream *os, const uint8_t **body, size_t *body_len) { return flush_i(os, 1, 4096, body, body_len); }
int fgh_ogg_stream_pageout(fgh_ogg_stream *os, const uint8_t **body, size_t *body_len)
{
	int force = 0;
	if((os->e_o_s && os->lacing_fill) || (os->lacing_fill && !os->b_o_s)) force = 1;      /* done: flush; or the initial header page */
	return flush_i(os, force, 4096, body, body_len);
}

/* ---- the mapping (ogg_encoder_aspect.c) ---- */
void fgh_ogg_aspect_set_defaults(fgh_ogg_aspect *a) { a->serial_number = 0; a->num_metadata = 0; }
int fgh_ogg_aspect_init(fgh_ogg_aspect *a)
{
	if(fgh_ogg_stream_init(&a->os, a->serial_number) != 0) return 0;
	a->seen_magic = 0; a->is_first_packet = 1; a->samples_written = 0; a->last_page_granule_pos = 0; a->active = 1;
	return 1;
}
void fgh_ogg_aspect_finish(fgh_ogg_aspect *a) { if(a->active) fgh_ogg_stream_clear(&a->os); a->active = 0; }

/* One write of the encoder -> zero or more (header, body) write pairs to the client.  Returns 0 on failure. */
int fgh_ogg_aspect_write(fgh_ogg_aspect *a, const uint8_t *buf, size_t bytes, uint32_t samples, uint32_t current_frame, int is_last_block,
                         fgh_ogg_write_proxy write, void *encoder, void *client_data)
{
	const int is_metadata = samples == 0;
	if(a->seen_magic) {
		uint8_t first[1 + 4 + 1 + 1 + 2 + 4 + 4 + 34];
		const uint8_t *pk = buf;
		size_t pkn = bytes;
		if(a->is_first_packet) {
			if(bytes != 4 + 34) return 0;                                  /* must be the STREAMINFO block */
			uint8_t *b = first;
			*b++ = 0x7f;
			memcpy(b, "FLAC", 4); b += 4;
			*b++ = 1; *b++ = 0;                                             /* mapping version 1.0 */
			*b++ = (uint8_t)(a->num_metadata >> 8); *b++ = (uint8_t)a->num_metadata;
			memcpy(b, "fLaC", 4); b += 4;
			memcpy(b, buf, bytes);
			pk = first; pkn = sizeof first;
			a->is_first_packet = 0;
		}
		if(fgh_ogg_stream_packetin(&a->os, pk, pkn, (int64_t)(a->samples_written + samples), is_last_block) != 0) return 0;
		for(;;) {
			const uint8_t *body;
			size_t body_len;
			const int got = is_metadata ? fgh_ogg_stream_flush(&a->os, &body, &body_len) : fgh_ogg_stream_pageout(&a->os, &body, &body_len);
			if(!got) break;
			int64_t g = 0;
			for(int i = 13; i >= 6; i--) g = (int64_t)(((uint64_t)g << 8) | a->os.header[i]);
			uint32_t on_page = 0;
			if(g != -1) { on_page = (uint32_t)(g - a->last_page_granule_pos); a->last_page_granule_pos = g; }   /* -1: no packet ends here */
			if(!write(encoder, a->os.header, a->os.header_len, 0, current_frame, client_data)) return 0;
			if(!write(encoder, body, body_len, on_page, current_frame, client_data)) return 0;
		}
	}
	else if(is_metadata && current_frame == 0 && bytes == 4 && memcmp(buf, "fLaC", 4) == 0) a->seen_magic = 1;
	else return 0;
	a->samples_written += samples;
	return 1;
}
