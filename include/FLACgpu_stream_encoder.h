/* include/FLACgpu_stream_encoder.h -- the libFLAC stream-encoder API as exported by libFLACgpu.so.
 *
 * ABI mirror of the reference's public encoder interface (include/FLAC/stream_encoder.h:225-1896,
 * metadata structures include/FLAC/format.h:505-895): same symbol names, same argument meaning,
 * same enum values, same structure layouts -- so a program compiled against the reference's
 * <FLAC/stream_encoder.h> can be linked against libFLACgpu.so unchanged.  This header exists for
 * callers and tests that do not have the reference's headers; do not include both.
 *
 * Behind it the per-block hot path runs on the MI355X through include/flacgpu.h; there is no CPU
 * encode path (init fails with ..._INIT_STATUS_ENCODER_ERROR when no GPU engine can be created).
 */
#ifndef FLACGPU_STREAM_ENCODER_H
#define FLACGPU_STREAM_ENCODER_H
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int FLAC__bool;
typedef int32_t FLAC__int32;
typedef uint32_t FLAC__uint32;
typedef uint64_t FLAC__uint64;
typedef uint8_t FLAC__byte;

/* ---- enums (values fixed by the reference ABI) --------------------------------------------------- */
typedef enum {                                   /* stream_encoder.h:225-270 */
	FLAC__STREAM_ENCODER_OK = 0, FLAC__STREAM_ENCODER_UNINITIALIZED, FLAC__STREAM_ENCODER_OGG_ERROR,
	FLAC__STREAM_ENCODER_VERIFY_DECODER_ERROR, FLAC__STREAM_ENCODER_VERIFY_MISMATCH_IN_AUDIO_DATA,
	FLAC__STREAM_ENCODER_CLIENT_ERROR, FLAC__STREAM_ENCODER_IO_ERROR, FLAC__STREAM_ENCODER_FRAMING_ERROR,
	FLAC__STREAM_ENCODER_MEMORY_ALLOCATION_ERROR
} FLAC__StreamEncoderState;

typedef enum {                                   /* stream_encoder.h:298-345 */
	FLAC__STREAM_ENCODER_INIT_STATUS_OK = 0, FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR,
	FLAC__STREAM_ENCODER_INIT_STATUS_UNSUPPORTED_CONTAINER, FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_CALLBACKS,
	FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_NUMBER_OF_CHANNELS, FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_BITS_PER_SAMPLE,
	FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_SAMPLE_RATE, FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_BLOCK_SIZE,
	FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_MAX_LPC_ORDER, FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_QLP_COEFF_PRECISION,
	FLAC__STREAM_ENCODER_INIT_STATUS_BLOCK_SIZE_TOO_SMALL_FOR_LPC_ORDER, FLAC__STREAM_ENCODER_INIT_STATUS_NOT_STREAMABLE,
	FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_METADATA, FLAC__STREAM_ENCODER_INIT_STATUS_ALREADY_INITIALIZED
} FLAC__StreamEncoderInitStatus;

typedef enum { FLAC__STREAM_ENCODER_READ_STATUS_CONTINUE, FLAC__STREAM_ENCODER_READ_STATUS_END_OF_STREAM,
               FLAC__STREAM_ENCODER_READ_STATUS_ABORT, FLAC__STREAM_ENCODER_READ_STATUS_UNSUPPORTED } FLAC__StreamEncoderReadStatus;
typedef enum { FLAC__STREAM_ENCODER_WRITE_STATUS_OK = 0, FLAC__STREAM_ENCODER_WRITE_STATUS_FATAL_ERROR } FLAC__StreamEncoderWriteStatus;
typedef enum { FLAC__STREAM_ENCODER_SEEK_STATUS_OK, FLAC__STREAM_ENCODER_SEEK_STATUS_ERROR,
               FLAC__STREAM_ENCODER_SEEK_STATUS_UNSUPPORTED } FLAC__StreamEncoderSeekStatus;
typedef enum { FLAC__STREAM_ENCODER_TELL_STATUS_OK, FLAC__STREAM_ENCODER_TELL_STATUS_ERROR,
               FLAC__STREAM_ENCODER_TELL_STATUS_UNSUPPORTED } FLAC__StreamEncoderTellStatus;

/* the verify decoder of this library is a plain frame decoder on the host (flac_amd/csrc/host/verify.c): its state reads
 * SEARCH_FOR_FRAME_SYNC (2) while verification is on, else "uninitialized"
 * (FLAC__StreamDecoderState, include/FLAC/stream_decoder.h:205-250: last enumerator) */
typedef int FLAC__StreamDecoderState;
#define FLAC__STREAM_DECODER_UNINITIALIZED 9

#define FLAC__STREAM_ENCODER_SET_NUM_THREADS_OK 0
#define FLAC__STREAM_ENCODER_SET_NUM_THREADS_NOT_COMPILED_WITH_MULTITHREADING_ENABLED 1
#define FLAC__STREAM_ENCODER_SET_NUM_THREADS_ALREADY_INITIALIZED 2
#define FLAC__STREAM_ENCODER_SET_NUM_THREADS_TOO_MANY_THREADS 3

/* ---- metadata blocks the encoder is handed by set_metadata() (format.h:505-895) ------------------- */
typedef enum {
	FLAC__METADATA_TYPE_STREAMINFO = 0, FLAC__METADATA_TYPE_PADDING = 1, FLAC__METADATA_TYPE_APPLICATION = 2,
	FLAC__METADATA_TYPE_SEEKTABLE = 3, FLAC__METADATA_TYPE_VORBIS_COMMENT = 4, FLAC__METADATA_TYPE_CUESHEET = 5,
	FLAC__METADATA_TYPE_PICTURE = 6, FLAC__METADATA_TYPE_UNDEFINED = 7, FLAC__MAX_METADATA_TYPE = 126
} FLAC__MetadataType;

typedef struct {
	uint32_t min_blocksize, max_blocksize, min_framesize, max_framesize, sample_rate, channels, bits_per_sample;
	FLAC__uint64 total_samples;
	FLAC__byte md5sum[16];
} FLAC__StreamMetadata_StreamInfo;
typedef struct { int dummy; } FLAC__StreamMetadata_Padding;
typedef struct { FLAC__byte id[4]; FLAC__byte *data; } FLAC__StreamMetadata_Application;
typedef struct { FLAC__uint64 sample_number, stream_offset; uint32_t frame_samples; } FLAC__StreamMetadata_SeekPoint;
typedef struct { uint32_t num_points; FLAC__StreamMetadata_SeekPoint *points; } FLAC__StreamMetadata_SeekTable;
typedef struct { FLAC__uint32 length; FLAC__byte *entry; } FLAC__StreamMetadata_VorbisComment_Entry;
typedef struct {
	FLAC__StreamMetadata_VorbisComment_Entry vendor_string;
	FLAC__uint32 num_comments;
	FLAC__StreamMetadata_VorbisComment_Entry *comments;
} FLAC__StreamMetadata_VorbisComment;
typedef struct { FLAC__uint64 offset; FLAC__byte number; } FLAC__StreamMetadata_CueSheet_Index;
typedef struct {
	FLAC__uint64 offset;
	FLAC__byte number;
	char isrc[13];
	uint32_t type : 1;
	uint32_t pre_emphasis : 1;
	FLAC__byte num_indices;
	FLAC__StreamMetadata_CueSheet_Index *indices;
} FLAC__StreamMetadata_CueSheet_Track;
typedef struct {
	char media_catalog_number[129];
	FLAC__uint64 lead_in;
	FLAC__bool is_cd;
	uint32_t num_tracks;
	FLAC__StreamMetadata_CueSheet_Track *tracks;
} FLAC__StreamMetadata_CueSheet;
typedef struct {
	int type;                       /* FLAC__StreamMetadata_Picture_Type; 1 = 32x32 PNG file icon, 2 = other file icon */
	char *mime_type;
	FLAC__byte *description;
	FLAC__uint32 width, height, depth, colors, data_length;
	FLAC__byte *data;
} FLAC__StreamMetadata_Picture;
typedef struct { FLAC__byte *data; } FLAC__StreamMetadata_Unknown;

typedef struct FLAC__StreamMetadata {
	FLAC__MetadataType type;
	FLAC__bool is_last;
	uint32_t length;                /* bytes of block data, header excluded */
	union {
		FLAC__StreamMetadata_StreamInfo stream_info;
		FLAC__StreamMetadata_Padding padding;
		FLAC__StreamMetadata_Application application;
		FLAC__StreamMetadata_SeekTable seek_table;
		FLAC__StreamMetadata_VorbisComment vorbis_comment;
		FLAC__StreamMetadata_CueSheet cue_sheet;
		FLAC__StreamMetadata_Picture picture;
		FLAC__StreamMetadata_Unknown unknown;
	} data;
} FLAC__StreamMetadata;

#define FLAC__STREAM_METADATA_SEEKPOINT_PLACEHOLDER 0xffffffffffffffffull

/* ---- the encoder object (stream_encoder.h:462-472: two opaque pointers) --------------------------- */
struct FLAC__StreamEncoderProtected;
struct FLAC__StreamEncoderPrivate;
typedef struct {
	struct FLAC__StreamEncoderProtected *protected_;
	struct FLAC__StreamEncoderPrivate *private_;
} FLAC__StreamEncoder;

/* ---- client callbacks (stream_encoder.h:474-690) -------------------------------------------------- */
typedef FLAC__StreamEncoderReadStatus (*FLAC__StreamEncoderReadCallback)(const FLAC__StreamEncoder *, FLAC__byte buffer[], size_t *bytes, void *client_data);
/* samples == 0: metadata; otherwise one frame holding `samples` inter-channel samples, frame number current_frame */
typedef FLAC__StreamEncoderWriteStatus (*FLAC__StreamEncoderWriteCallback)(const FLAC__StreamEncoder *, const FLAC__byte buffer[], size_t bytes, uint32_t samples, uint32_t current_frame, void *client_data);
typedef FLAC__StreamEncoderSeekStatus (*FLAC__StreamEncoderSeekCallback)(const FLAC__StreamEncoder *, FLAC__uint64 absolute_byte_offset, void *client_data);
typedef FLAC__StreamEncoderTellStatus (*FLAC__StreamEncoderTellCallback)(const FLAC__StreamEncoder *, FLAC__uint64 *absolute_byte_offset, void *client_data);
typedef void (*FLAC__StreamEncoderMetadataCallback)(const FLAC__StreamEncoder *, const FLAC__StreamMetadata *metadata, void *client_data);
typedef void (*FLAC__StreamEncoderProgressCallback)(const FLAC__StreamEncoder *, FLAC__uint64 bytes_written, FLAC__uint64 samples_written, uint32_t frames_written, uint32_t total_frames_estimate, void *client_data);

/* ---- string tables ------------------------------------------------------------------------------- */
/* 1 if the library writes Ogg FLAC (FLAC/export.h:107): it does, see flac_amd/csrc/host/ogg.c */
extern int FLAC_API_SUPPORTS_OGG_FLAC;
extern const char * const FLAC__StreamEncoderStateString[];
extern const char * const FLAC__StreamEncoderInitStatusString[];
extern const char * const FLAC__StreamEncoderReadStatusString[];
extern const char * const FLAC__StreamEncoderWriteStatusString[];
extern const char * const FLAC__StreamEncoderSeekStatusString[];
extern const char * const FLAC__StreamEncoderTellStatusString[];
extern const char *FLAC__VENDOR_STRING;   /* format.c:57: written into the VORBIS_COMMENT block */

/* ---- construction, settings (only legal before init; return false afterwards) ---------------------- */
FLAC__StreamEncoder *FLAC__stream_encoder_new(void);
void FLAC__stream_encoder_delete(FLAC__StreamEncoder *encoder);

FLAC__bool FLAC__stream_encoder_set_ogg_serial_number(FLAC__StreamEncoder *encoder, long serial_number);
FLAC__bool FLAC__stream_encoder_set_verify(FLAC__StreamEncoder *encoder, FLAC__bool value);
FLAC__bool FLAC__stream_encoder_set_streamable_subset(FLAC__StreamEncoder *encoder, FLAC__bool value);
FLAC__bool FLAC__stream_encoder_set_channels(FLAC__StreamEncoder *encoder, uint32_t value);
FLAC__bool FLAC__stream_encoder_set_bits_per_sample(FLAC__StreamEncoder *encoder, uint32_t value);
FLAC__bool FLAC__stream_encoder_set_sample_rate(FLAC__StreamEncoder *encoder, uint32_t value);
FLAC__bool FLAC__stream_encoder_set_compression_level(FLAC__StreamEncoder *encoder, uint32_t value);
FLAC__bool FLAC__stream_encoder_set_blocksize(FLAC__StreamEncoder *encoder, uint32_t value);
FLAC__bool FLAC__stream_encoder_set_do_mid_side_stereo(FLAC__StreamEncoder *encoder, FLAC__bool value);
FLAC__bool FLAC__stream_encoder_set_loose_mid_side_stereo(FLAC__StreamEncoder *encoder, FLAC__bool value);
FLAC__bool FLAC__stream_encoder_set_apodization(FLAC__StreamEncoder *encoder, const char *specification);
FLAC__bool FLAC__stream_encoder_set_max_lpc_order(FLAC__StreamEncoder *encoder, uint32_t value);
FLAC__bool FLAC__stream_encoder_set_qlp_coeff_precision(FLAC__StreamEncoder *encoder, uint32_t value);
FLAC__bool FLAC__stream_encoder_set_do_qlp_coeff_prec_search(FLAC__StreamEncoder *encoder, FLAC__bool value);
FLAC__bool FLAC__stream_encoder_set_do_escape_coding(FLAC__StreamEncoder *encoder, FLAC__bool value);
FLAC__bool FLAC__stream_encoder_set_do_exhaustive_model_search(FLAC__StreamEncoder *encoder, FLAC__bool value);
FLAC__bool FLAC__stream_encoder_set_min_residual_partition_order(FLAC__StreamEncoder *encoder, uint32_t value);
FLAC__bool FLAC__stream_encoder_set_max_residual_partition_order(FLAC__StreamEncoder *encoder, uint32_t value);
uint32_t   FLAC__stream_encoder_set_num_threads(FLAC__StreamEncoder *encoder, uint32_t value);
FLAC__bool FLAC__stream_encoder_set_rice_parameter_search_dist(FLAC__StreamEncoder *encoder, uint32_t value);
FLAC__bool FLAC__stream_encoder_set_total_samples_estimate(FLAC__StreamEncoder *encoder, FLAC__uint64 value);
FLAC__bool FLAC__stream_encoder_set_metadata(FLAC__StreamEncoder *encoder, FLAC__StreamMetadata **metadata, uint32_t num_blocks);
FLAC__bool FLAC__stream_encoder_set_limit_min_bitrate(FLAC__StreamEncoder *encoder, FLAC__bool value);
/* exported by the reference without a header declaration (stream_encoder.c:1829,2249-2297) */
FLAC__bool FLAC__stream_encoder_set_do_md5(FLAC__StreamEncoder *encoder, FLAC__bool value);
FLAC__bool FLAC__stream_encoder_get_do_md5(const FLAC__StreamEncoder *encoder);
FLAC__bool FLAC__stream_encoder_disable_instruction_set(FLAC__StreamEncoder *encoder, uint32_t value);
FLAC__bool FLAC__stream_encoder_disable_constant_subframes(FLAC__StreamEncoder *encoder, FLAC__bool value);
FLAC__bool FLAC__stream_encoder_disable_fixed_subframes(FLAC__StreamEncoder *encoder, FLAC__bool value);
FLAC__bool FLAC__stream_encoder_disable_verbatim_subframes(FLAC__StreamEncoder *encoder, FLAC__bool value);

/* ---- getters ------------------------------------------------------------------------------------- */
FLAC__StreamEncoderState FLAC__stream_encoder_get_state(const FLAC__StreamEncoder *encoder);
FLAC__StreamDecoderState FLAC__stream_encoder_get_verify_decoder_state(const FLAC__StreamEncoder *encoder);
const char *FLAC__stream_encoder_get_resolved_state_string(const FLAC__StreamEncoder *encoder);
void FLAC__stream_encoder_get_verify_decoder_error_stats(const FLAC__StreamEncoder *encoder, FLAC__uint64 *absolute_sample, uint32_t *frame_number, uint32_t *channel, uint32_t *sample, FLAC__int32 *expected, FLAC__int32 *got);
FLAC__bool FLAC__stream_encoder_get_verify(const FLAC__StreamEncoder *encoder);
FLAC__bool FLAC__stream_encoder_get_streamable_subset(const FLAC__StreamEncoder *encoder);
uint32_t   FLAC__stream_encoder_get_channels(const FLAC__StreamEncoder *encoder);
uint32_t   FLAC__stream_encoder_get_bits_per_sample(const FLAC__StreamEncoder *encoder);
uint32_t   FLAC__stream_encoder_get_sample_rate(const FLAC__StreamEncoder *encoder);
uint32_t   FLAC__stream_encoder_get_blocksize(const FLAC__StreamEncoder *encoder);
FLAC__bool FLAC__stream_encoder_get_do_mid_side_stereo(const FLAC__StreamEncoder *encoder);
FLAC__bool FLAC__stream_encoder_get_loose_mid_side_stereo(const FLAC__StreamEncoder *encoder);
uint32_t   FLAC__stream_encoder_get_max_lpc_order(const FLAC__StreamEncoder *encoder);
uint32_t   FLAC__stream_encoder_get_qlp_coeff_precision(const FLAC__StreamEncoder *encoder);
FLAC__bool FLAC__stream_encoder_get_do_qlp_coeff_prec_search(const FLAC__StreamEncoder *encoder);
FLAC__bool FLAC__stream_encoder_get_do_escape_coding(const FLAC__StreamEncoder *encoder);
FLAC__bool FLAC__stream_encoder_get_do_exhaustive_model_search(const FLAC__StreamEncoder *encoder);
uint32_t   FLAC__stream_encoder_get_min_residual_partition_order(const FLAC__StreamEncoder *encoder);
uint32_t   FLAC__stream_encoder_get_max_residual_partition_order(const FLAC__StreamEncoder *encoder);
uint32_t   FLAC__stream_encoder_get_num_threads(const FLAC__StreamEncoder *encoder);
uint32_t   FLAC__stream_encoder_get_rice_parameter_search_dist(const FLAC__StreamEncoder *encoder);
FLAC__uint64 FLAC__stream_encoder_get_total_samples_estimate(const FLAC__StreamEncoder *encoder);
FLAC__bool FLAC__stream_encoder_get_limit_min_bitrate(const FLAC__StreamEncoder *encoder);

/* ---- init / process / finish ---------------------------------------------------------------------- */
FLAC__StreamEncoderInitStatus FLAC__stream_encoder_init_stream(FLAC__StreamEncoder *encoder, FLAC__StreamEncoderWriteCallback write_callback, FLAC__StreamEncoderSeekCallback seek_callback, FLAC__StreamEncoderTellCallback tell_callback, FLAC__StreamEncoderMetadataCallback metadata_callback, void *client_data);
FLAC__StreamEncoderInitStatus FLAC__stream_encoder_init_ogg_stream(FLAC__StreamEncoder *encoder, FLAC__StreamEncoderReadCallback read_callback, FLAC__StreamEncoderWriteCallback write_callback, FLAC__StreamEncoderSeekCallback seek_callback, FLAC__StreamEncoderTellCallback tell_callback, FLAC__StreamEncoderMetadataCallback metadata_callback, void *client_data);
FLAC__StreamEncoderInitStatus FLAC__stream_encoder_init_FILE(FLAC__StreamEncoder *encoder, FILE *file, FLAC__StreamEncoderProgressCallback progress_callback, void *client_data);
FLAC__StreamEncoderInitStatus FLAC__stream_encoder_init_ogg_FILE(FLAC__StreamEncoder *encoder, FILE *file, FLAC__StreamEncoderProgressCallback progress_callback, void *client_data);
FLAC__StreamEncoderInitStatus FLAC__stream_encoder_init_file(FLAC__StreamEncoder *encoder, const char *filename, FLAC__StreamEncoderProgressCallback progress_callback, void *client_data);
FLAC__StreamEncoderInitStatus FLAC__stream_encoder_init_ogg_file(FLAC__StreamEncoder *encoder, const char *filename, FLAC__StreamEncoderProgressCallback progress_callback, void *client_data);
/* planar / interleaved int32 PCM, `samples` per channel, any chunking; blocks are staged and encoded on the
 * GPU in batches, write callbacks arrive in stream order on the calling thread */
FLAC__bool FLAC__stream_encoder_process(FLAC__StreamEncoder *encoder, const FLAC__int32 * const buffer[], uint32_t samples);
FLAC__bool FLAC__stream_encoder_process_interleaved(FLAC__StreamEncoder *encoder, const FLAC__int32 buffer[], uint32_t samples);
FLAC__bool FLAC__stream_encoder_finish(FLAC__StreamEncoder *encoder);

#ifdef __cplusplus
}
#endif
#endif
