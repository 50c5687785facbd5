/* oracle/ref_shim.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * A thin driver around the *unmodified* reference libFLAC (compiled in place from
 * /root/reference by oracle/Makefile into oracle/_ref/libFLAC_ref.so).  It drives the
 * reference exclusively through its public API (include/FLAC/stream_encoder.h:704-1896)
 * and captures what the write callback receives, so tests and bench.py's cpu_baseline
 * leg can (a) pin the CPU restatement in oracle/flac_oracle.c and the HIP path to the
 * real reference bytes and (b) time the reference on the host cores.
 *
 * Nothing here restates reference logic; it only calls it.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "FLAC/stream_encoder.h"

/* exported-but-unheadered test hooks of the reference (stream_encoder.c:1829,2249) */
extern FLAC__bool FLAC__stream_encoder_set_do_md5(FLAC__StreamEncoder *encoder, FLAC__bool value);
extern FLAC__bool FLAC__stream_encoder_disable_instruction_set(FLAC__StreamEncoder *encoder, int value);
extern FLAC__bool FLAC__stream_encoder_disable_constant_subframes(FLAC__StreamEncoder *encoder, FLAC__bool value);
extern FLAC__bool FLAC__stream_encoder_disable_fixed_subframes(FLAC__StreamEncoder *encoder, FLAC__bool value);
extern FLAC__bool FLAC__stream_encoder_disable_verbatim_subframes(FLAC__StreamEncoder *encoder, FLAC__bool value);

typedef struct {
	uint8_t *out;
	size_t cap, len;
	uint32_t *frame_bytes;
	uint32_t frame_cap, nframes;
	uint64_t header_bytes;
	int overflow;
} capture_t;

static FLAC__StreamEncoderWriteStatus write_cb(const FLAC__StreamEncoder *enc, const FLAC__byte buffer[], size_t bytes, uint32_t samples, uint32_t current_frame, void *client)
{
	capture_t *c = (capture_t *)client;
	(void)enc; (void)current_frame;
	if(c->out) {
		if(c->len + bytes > c->cap) { c->overflow = 1; return FLAC__STREAM_ENCODER_WRITE_STATUS_FATAL_ERROR; }
		memcpy(c->out + c->len, buffer, bytes);
	}
	c->len += bytes;
	if(samples == 0)
		c->header_bytes += bytes;
	else {
		if(c->frame_bytes && c->nframes < c->frame_cap)
			c->frame_bytes[c->nframes] = (uint32_t)bytes;
		c->nframes++;
	}
	return FLAC__STREAM_ENCODER_WRITE_STATUS_OK;
}

/* No seek/tell callbacks: the STREAMINFO block in the captured stream keeps its
 * initial (zeroed MD5 / frame sizes) form; tests compare the audio frames, and the
 * host layer's STREAMINFO fix-up is compared through ref_encode_file() below. */

typedef struct {
	uint32_t channels, bps, sample_rate;
	int32_t level;            /* 0..8, applied first */
	uint32_t blocksize;       /* 0 = keep preset default */
	int32_t do_md5;           /* 0/1 */
	int32_t num_threads;      /* 0/1 = single thread */
	int32_t disable_isa_mask; /* passed to disable_instruction_set when != 0 */
	int32_t limit_min_bitrate;/* 0/1 */
	int32_t streamable_subset;/* 0/1 */
	int32_t max_lpc_order;    /* -1 = keep preset */
	int32_t qlp_precision;    /* -1 = keep preset */
	int32_t min_partition_order, max_partition_order; /* -1 = keep preset */
	int32_t mid_side, loose_mid_side; /* -1 = keep preset */
	const char *apodization;  /* NULL = keep preset */
	int32_t exhaustive, prec_search; /* -e / -p: FLAC__stream_encoder_set_do_exhaustive_model_search / _qlp_coeff_prec_search */
	int32_t disable_constant, disable_fixed, disable_verbatim; /* --disable-*-subframes (stream_encoder.c:2262-2292) */
} ref_cfg_t;

static FLAC__StreamEncoder *make_encoder(const ref_cfg_t *cfg, uint64_t total_samples)
{
	FLAC__StreamEncoder *e = FLAC__stream_encoder_new();
	if(!e) return 0;
	FLAC__stream_encoder_set_channels(e, cfg->channels);
	FLAC__stream_encoder_set_bits_per_sample(e, cfg->bps);
	FLAC__stream_encoder_set_sample_rate(e, cfg->sample_rate);
	FLAC__stream_encoder_set_streamable_subset(e, cfg->streamable_subset ? true : false);
	FLAC__stream_encoder_set_compression_level(e, (uint32_t)cfg->level);
	if(cfg->blocksize) FLAC__stream_encoder_set_blocksize(e, cfg->blocksize);
	if(cfg->max_lpc_order >= 0) FLAC__stream_encoder_set_max_lpc_order(e, (uint32_t)cfg->max_lpc_order);
	if(cfg->qlp_precision >= 0) FLAC__stream_encoder_set_qlp_coeff_precision(e, (uint32_t)cfg->qlp_precision);
	if(cfg->min_partition_order >= 0) FLAC__stream_encoder_set_min_residual_partition_order(e, (uint32_t)cfg->min_partition_order);
	if(cfg->max_partition_order >= 0) FLAC__stream_encoder_set_max_residual_partition_order(e, (uint32_t)cfg->max_partition_order);
	if(cfg->mid_side >= 0) FLAC__stream_encoder_set_do_mid_side_stereo(e, cfg->mid_side ? true : false);
	if(cfg->loose_mid_side >= 0) FLAC__stream_encoder_set_loose_mid_side_stereo(e, cfg->loose_mid_side ? true : false);
	if(cfg->apodization) FLAC__stream_encoder_set_apodization(e, cfg->apodization);
	if(cfg->exhaustive > 0) FLAC__stream_encoder_set_do_exhaustive_model_search(e, true);
	if(cfg->prec_search > 0) FLAC__stream_encoder_set_do_qlp_coeff_prec_search(e, true);
	FLAC__stream_encoder_set_limit_min_bitrate(e, cfg->limit_min_bitrate ? true : false);
	FLAC__stream_encoder_set_do_md5(e, cfg->do_md5 ? true : false);
	FLAC__stream_encoder_set_total_samples_estimate(e, total_samples);
	if(cfg->num_threads > 1) FLAC__stream_encoder_set_num_threads(e, (uint32_t)cfg->num_threads);
	if(cfg->disable_isa_mask) FLAC__stream_encoder_disable_instruction_set(e, cfg->disable_isa_mask);
	if(cfg->disable_constant) FLAC__stream_encoder_disable_constant_subframes(e, true);
	if(cfg->disable_fixed) FLAC__stream_encoder_disable_fixed_subframes(e, true);
	if(cfg->disable_verbatim) FLAC__stream_encoder_disable_verbatim_subframes(e, true);
	return e;
}

/* Encode `nsamples` inter-channel samples of interleaved int32 PCM through the
 * reference encoder (stream callbacks, write only).  Returns total bytes produced,
 * or a negative value on error.  `out` may be NULL (measure only).
 * `elapsed_s` (optional) receives the wall time spent inside process()+finish(). */
int64_t ref_encode_stream(const ref_cfg_t *cfg, const int32_t *pcm_interleaved, uint64_t nsamples,
                          uint8_t *out, uint64_t cap, uint32_t *frame_bytes, uint32_t frame_cap,
                          uint32_t *nframes_out, uint64_t *header_bytes_out, double *elapsed_s)
{
	capture_t c;
	struct timespec t0, t1;
	FLAC__StreamEncoder *e = make_encoder(cfg, nsamples);
	FLAC__bool ok = true;
	uint64_t done = 0;
	memset(&c, 0, sizeof c);
	c.out = out; c.cap = (size_t)cap; c.frame_bytes = frame_bytes; c.frame_cap = frame_cap;
	if(!e) return -1;
	if(FLAC__stream_encoder_init_stream(e, write_cb, 0, 0, 0, &c) != FLAC__STREAM_ENCODER_INIT_STATUS_OK) {
		FLAC__stream_encoder_delete(e);
		return -2;
	}
	clock_gettime(CLOCK_MONOTONIC, &t0);
	while(ok && done < nsamples) {
		/* feed in chunks like a real client would; chunking does not change output */
		uint64_t n = nsamples - done;
		if(n > 1u << 20) n = 1u << 20;
		ok = FLAC__stream_encoder_process_interleaved(e, pcm_interleaved + done * cfg->channels, (uint32_t)n);
		done += n;
	}
	ok = FLAC__stream_encoder_finish(e) && ok;
	clock_gettime(CLOCK_MONOTONIC, &t1);
	FLAC__stream_encoder_delete(e);
	if(elapsed_s) *elapsed_s = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
	if(nframes_out) *nframes_out = c.nframes;
	if(header_bytes_out) *header_bytes_out = c.header_bytes;
	if(!ok || c.overflow) return -3;
	return (int64_t)c.len;
}

/* Same, but through FLAC__stream_encoder_init_file so that the reference performs its
 * STREAMINFO / seektable fix-up (stream_encoder.c:3139-3299); used to pin the host
 * layer's whole-file output. Returns 0 on success. */
int32_t ref_encode_file(const ref_cfg_t *cfg, const int32_t *pcm_interleaved, uint64_t nsamples, const char *path)
{
	FLAC__StreamEncoder *e = make_encoder(cfg, nsamples);
	FLAC__bool ok;
	if(!e) return -1;
	if(FLAC__stream_encoder_init_file(e, path, 0, 0) != FLAC__STREAM_ENCODER_INIT_STATUS_OK) {
		FLAC__stream_encoder_delete(e);
		return -2;
	}
	ok = FLAC__stream_encoder_process_interleaved(e, pcm_interleaved, (uint32_t)nsamples);
	ok = FLAC__stream_encoder_finish(e) && ok;
	FLAC__stream_encoder_delete(e);
	return ok ? 0 : -3;
}

const char *ref_vendor_string(void) { return FLAC__VENDOR_STRING; }

/* ---- the reference's stream DECODER on a stream in memory (FLAC__stream_decoder_init_stream with every callback, so that the
 * decoder can rewind after a damaged frame the way it does on a file, src/libFLAC/stream_decoder.c:2558-2587): what the client
 * receives -- every sample of every write callback, every error callback in order -- for tests/test_stream_decode_*.py to hold the
 * product's GPU stream decoder to.  Nothing here restates reference logic. ---- */
#include "FLAC/stream_decoder.h"

typedef struct {
	const uint8_t *data; uint64_t len, pos;
	int32_t *pcm; uint64_t cap, samples;         /* interleaved, inter-channel samples */
	uint32_t channels, bps, rate;                /* of the first frame written */
	uint32_t *ev; uint32_t max_ev, nev;
	uint64_t frames, format_changes;
	int overflow;
} dec_capture_t;

static FLAC__StreamDecoderReadStatus dec_read_cb(const FLAC__StreamDecoder *d, FLAC__byte buffer[], size_t *bytes, void *client)
{
	dec_capture_t *c = (dec_capture_t *)client; (void)d;
	if(c->pos >= c->len) { *bytes = 0; return FLAC__STREAM_DECODER_READ_STATUS_END_OF_STREAM; }
	if(*bytes > c->len - c->pos) *bytes = (size_t)(c->len - c->pos);
	memcpy(buffer, c->data + c->pos, *bytes);
	c->pos += *bytes;
	return FLAC__STREAM_DECODER_READ_STATUS_CONTINUE;
}
static FLAC__StreamDecoderSeekStatus dec_seek_cb(const FLAC__StreamDecoder *d, FLAC__uint64 off, void *client)
{
	dec_capture_t *c = (dec_capture_t *)client; (void)d;
	if(off > c->len) return FLAC__STREAM_DECODER_SEEK_STATUS_ERROR;
	c->pos = off; return FLAC__STREAM_DECODER_SEEK_STATUS_OK;
}
static FLAC__StreamDecoderTellStatus dec_tell_cb(const FLAC__StreamDecoder *d, FLAC__uint64 *off, void *client)
{ (void)d; *off = ((dec_capture_t *)client)->pos; return FLAC__STREAM_DECODER_TELL_STATUS_OK; }
static FLAC__StreamDecoderLengthStatus dec_length_cb(const FLAC__StreamDecoder *d, FLAC__uint64 *len, void *client)
{ (void)d; *len = ((dec_capture_t *)client)->len; return FLAC__STREAM_DECODER_LENGTH_STATUS_OK; }
static FLAC__bool dec_eof_cb(const FLAC__StreamDecoder *d, void *client)
{ (void)d; return ((dec_capture_t *)client)->pos >= ((dec_capture_t *)client)->len; }
static FLAC__StreamDecoderWriteStatus dec_write_cb(const FLAC__StreamDecoder *d, const FLAC__Frame *f, const FLAC__int32 *const buf[], void *client)
{
	dec_capture_t *c = (dec_capture_t *)client; (void)d;
	if(c->channels == 0) { c->channels = f->header.channels; c->bps = f->header.bits_per_sample; c->rate = f->header.sample_rate; }
	if(f->header.channels != c->channels || f->header.bits_per_sample != c->bps) { c->format_changes++; return FLAC__STREAM_DECODER_WRITE_STATUS_CONTINUE; }
	if(c->samples + f->header.blocksize > c->cap) { c->overflow = 1; c->samples += f->header.blocksize; return FLAC__STREAM_DECODER_WRITE_STATUS_CONTINUE; }
	for(uint32_t i = 0; i < f->header.blocksize; i++)
		for(uint32_t ch = 0; ch < c->channels; ch++) c->pcm[(c->samples + i) * c->channels + ch] = buf[ch][i];
	c->samples += f->header.blocksize;
	c->frames++;
	return FLAC__STREAM_DECODER_WRITE_STATUS_CONTINUE;
}
static void dec_meta_cb(const FLAC__StreamDecoder *d, const FLAC__StreamMetadata *m, void *client)
{
	dec_capture_t *c = (dec_capture_t *)client; (void)d;
	if(m->type == FLAC__METADATA_TYPE_STREAMINFO) { c->channels = m->data.stream_info.channels; c->bps = m->data.stream_info.bits_per_sample; c->rate = m->data.stream_info.sample_rate; }
}
static void dec_error_cb(const FLAC__StreamDecoder *d, FLAC__StreamDecoderErrorStatus st, void *client)
{
	dec_capture_t *c = (dec_capture_t *)client; (void)d;
	if(c->nev < c->max_ev) c->ev[c->nev] = (uint32_t)st + 1;
	c->nev++;
}

/* result: samples, write callbacks taken, error callbacks, process_until_end_of_stream's return, finish's return (MD5), final state
 * before finish, channels, bps, format changes, overflow */
int ref_decode_stream(const uint8_t *stream, uint64_t nbytes, int md5_checking, int32_t *pcm, uint64_t cap_samples, uint32_t *ev_status, uint32_t max_events, uint64_t *result)
{
	dec_capture_t c;
	memset(&c, 0, sizeof c);
	c.data = stream; c.len = nbytes; c.pcm = pcm; c.cap = cap_samples; c.ev = ev_status; c.max_ev = max_events;
	FLAC__StreamDecoder *d = FLAC__stream_decoder_new();
	if(!d) return -1;
	FLAC__stream_decoder_set_md5_checking(d, md5_checking ? true : false);
	if(FLAC__stream_decoder_init_stream(d, dec_read_cb, dec_seek_cb, dec_tell_cb, dec_length_cb, dec_eof_cb, dec_write_cb, dec_meta_cb, dec_error_cb, &c) != FLAC__STREAM_DECODER_INIT_STATUS_OK) {
		FLAC__stream_decoder_delete(d);
		return -2;
	}
	const FLAC__bool ok = FLAC__stream_decoder_process_until_end_of_stream(d);
	const uint32_t state = (uint32_t)FLAC__stream_decoder_get_state(d);
	const FLAC__bool fin = FLAC__stream_decoder_finish(d);
	FLAC__stream_decoder_delete(d);
	result[0] = c.samples; result[1] = c.frames; result[2] = c.nev; result[3] = ok ? 1 : 0; result[4] = fin ? 1 : 0; result[5] = state;
	result[6] = c.channels; result[7] = c.bps; result[8] = c.format_changes; result[9] = (uint64_t)c.overflow;
	return 0;
}
