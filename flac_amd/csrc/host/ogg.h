/* flac_amd/csrc/host/ogg.h -- Ogg FLAC container layer of libFLACgpu.so (see ogg.c for what it restates and how it is pinned) */
#ifndef FGH_OGG_H
#define FGH_OGG_H
#include <stdint.h>
#include <stddef.h>

typedef struct {
	uint8_t *body; size_t body_storage, body_fill, body_returned;
	int *lacing; int64_t *granule; size_t lacing_storage, lacing_fill;
	uint8_t header[282]; size_t header_len;
	int e_o_s, b_o_s;
	long serialno; long pageno;
	int64_t packetno, granulepos;
} fgh_ogg_stream;

int  fgh_ogg_stream_init(fgh_ogg_stream *os, long serialno);
void fgh_ogg_stream_clear(fgh_ogg_stream *os);
int  fgh_ogg_stream_packetin(fgh_ogg_stream *os, const uint8_t *data, size_t bytes, int64_t granulepos, int e_o_s);
int  fgh_ogg_stream_pageout(fgh_ogg_stream *os, const uint8_t **body, size_t *body_len);
int  fgh_ogg_stream_flush(fgh_ogg_stream *os, const uint8_t **body, size_t *body_len);
void fgh_ogg_page_checksum_set(uint8_t *header, size_t header_len, const uint8_t *body, size_t body_len);

typedef struct {
	long serial_number;
	uint32_t num_metadata;
	fgh_ogg_stream os;
	int active, seen_magic, is_first_packet;
	uint64_t samples_written;
	int64_t last_page_granule_pos;
} fgh_ogg_aspect;

/* the client's write callback behind a uniform signature; returns non-zero on success */
typedef int (*fgh_ogg_write_proxy)(void *encoder, const uint8_t *buf, size_t bytes, uint32_t samples, uint32_t current_frame, void *client_data);

void fgh_ogg_aspect_set_defaults(fgh_ogg_aspect *a);
int  fgh_ogg_aspect_init(fgh_ogg_aspect *a);
void fgh_ogg_aspect_finish(fgh_ogg_aspect *a);
int  fgh_ogg_aspect_write(fgh_ogg_aspect *a, const uint8_t *buf, size_t bytes, uint32_t samples, uint32_t current_frame, int is_last_block,
                          fgh_ogg_write_proxy write, void *encoder, void *client_data);
#endif
