/* flac_amd/csrc/host/settings.c -- encoder settings: defaults, presets, apodization specification
 * parser and init-time resolution, stated after the reference's
 *   set_defaults_                    src/libFLAC/stream_encoder.c:2628-2700
 *   compression_levels_[]            :117-140
 *   FLAC__stream_encoder_set_apodization  :1940-2070
 *   init_stream_internal_ checks     :723-829
 * and the mapping of the resolved settings onto the GPU frame engine's configuration.
 */
#include <stdlib.h>
#include <string.h>
#include "flacgpu_host.h"

void flacgpu_host_settings_defaults(flacgpu_host_settings *s)
{
	memset(s, 0, sizeof *s);
	s->streamable_subset = 1;
	s->do_md5 = 1;
	s->channels = 2;
	s->bits_per_sample = 16;
	s->sample_rate = 44100;
	s->num_apodizations = 1;
	s->apodizations[0].type = FGH_APOD_TUKEY;
	s->apodizations[0].p = 0.5f;
	flacgpu_host_settings_level(s, 5);
}

/* {mid_side, loose, max_lpc_order, max_partition_order, apodization} -- the columns that differ
 * between presets; qlp precision 0 (auto), no precision search, no escape coding, no exhaustive
 * search, min partition order 0, rice search distance 0 in every row (stream_encoder.c:129-140) */
static const struct { int ms, loose; uint32_t lpc, max_po; const char *apod; } presets[9] = {
	{0, 0, 0, 3, "tukey(5e-1)"}, {1, 1, 0, 3, "tukey(5e-1)"}, {1, 0, 0, 3, "tukey(5e-1)"},
	{0, 0, 6, 4, "tukey(5e-1)"}, {1, 1, 8, 4, "tukey(5e-1)"}, {1, 0, 8, 5, "tukey(5e-1)"},
	{1, 0, 8, 6, "subdivide_tukey(2)"}, {1, 0, 12, 6, "subdivide_tukey(2)"}, {1, 0, 12, 6, "subdivide_tukey(3)"},
};

void flacgpu_host_settings_level(flacgpu_host_settings *s, uint32_t level)
{
	if(level > 8) level = 8;
	s->do_mid_side_stereo = presets[level].ms;
	s->loose_mid_side_stereo = presets[level].loose;
	flacgpu_host_settings_apodization(s, presets[level].apod);
	s->max_lpc_order = presets[level].lpc;
	s->qlp_coeff_precision = 0;
	s->do_qlp_coeff_prec_search = 0;
	s->do_escape_coding = 0;
	s->do_exhaustive_model_search = 0;
	s->min_residual_partition_order = 0;
	s->max_residual_partition_order = presets[level].max_po;
	s->rice_parameter_search_dist = 0;
}

static int is_name(const char *spec, size_t n, const char *name) { return n == strlen(name) && 0 == strncmp(name, spec, n); }

void flacgpu_host_settings_apodization(flacgpu_host_settings *s, const char *spec)
{
	static const struct { const char *name; flacgpu_host_apod_type t; } plain[] = {
		{"bartlett", FGH_APOD_BARTLETT}, {"bartlett_hann", FGH_APOD_BARTLETT_HANN}, {"blackman", FGH_APOD_BLACKMAN},
		{"blackman_harris_4term_92db", FGH_APOD_BLACKMAN_HARRIS_4TERM_92DB_SIDELOBE}, {"connes", FGH_APOD_CONNES},
		{"flattop", FGH_APOD_FLATTOP}, {"hamming", FGH_APOD_HAMMING}, {"hann", FGH_APOD_HANN},
		{"kaiser_bessel", FGH_APOD_KAISER_BESSEL}, {"nuttall", FGH_APOD_NUTTALL}, {"rectangle", FGH_APOD_RECTANGLE},
		{"triangle", FGH_APOD_TRIANGLE}, {"welch", FGH_APOD_WELCH},
	};
	s->num_apodizations = 0;
	for(;;) {
		const char *semi = strchr(spec, ';');
		const size_t n = semi ? (size_t)(semi - spec) : strlen(spec);
		flacgpu_host_apodization *a = &s->apodizations[s->num_apodizations];
		int matched = 0;
		for(size_t k = 0; k < sizeof plain / sizeof plain[0] && !matched; k++)
			if(is_name(spec, n, plain[k].name)) { a->type = plain[k].t; s->num_apodizations++; matched = 1; }
		if(matched) { /* done */ }
		else if(n > 7 && 0 == strncmp("gauss(", spec, 6)) {
			const float stddev = (float)strtod(spec + 6, 0);
			if(stddev > 0.0 && stddev <= 0.5) { a->p = stddev; a->type = FGH_APOD_GAUSS; s->num_apodizations++; }
		}
		else if(n > 7 && 0 == strncmp("tukey(", spec, 6)) {
			const float p = (float)strtod(spec + 6, 0);
			if(p >= 0.0 && p <= 1.0) { a->p = p; a->type = FGH_APOD_TUKEY; s->num_apodizations++; }
		}
		else if((n > 15 && 0 == strncmp("partial_tukey(", spec, 14)) || (n > 16 && 0 == strncmp("punchout_tukey(", spec, 15))) {
			const int punch = spec[1] == 'u';
			const int32_t parts = (int32_t)strtod(spec + (punch ? 15 : 14), 0);
			const char *s1 = strchr(spec, '/');
			float overlap = s1 ? (float)strtod(s1 + 1, 0) : (punch ? 0.2f : 0.1f);
			if(s1 && overlap > 0.99f) overlap = 0.99f;
			const float overlap_units = 1.0f / (1.0f - overlap) - 1.0f;
			const char *s2 = strchr((s1 ? (s1 + 1) : spec), '/');
			const float tukey_p = s2 ? (float)strtod(s2 + 1, 0) : 0.2f;
			if(parts <= 1) { a->p = tukey_p; a->type = FGH_APOD_TUKEY; s->num_apodizations++; }
			else if(s->num_apodizations + (uint32_t)parts < 32) {
				for(int32_t m = 0; m < parts; m++) {
					flacgpu_host_apodization *b = &s->apodizations[s->num_apodizations++];
					b->p = tukey_p;
					b->start = m / (parts + overlap_units);
					b->end = (m + 1 + overlap_units) / (parts + overlap_units);
					b->type = punch ? FGH_APOD_PUNCHOUT_TUKEY : FGH_APOD_PARTIAL_TUKEY;
				}
			}
		}
		else if(n > 17 && 0 == strncmp("subdivide_tukey(", spec, 16)) {
			const int32_t parts = (int32_t)strtod(spec + 16, 0);
			if(parts > 1) {
				const char *s1 = strchr(spec, '/');
				float p = s1 ? (float)strtod(s1 + 1, 0) : 5e-1;
				if(p > 1) p = 1; else if(p < 0) p = 0;
				a->parts = parts;
				a->p = p / parts;
				a->type = FGH_APOD_SUBDIVIDE_TUKEY;
				s->num_apodizations++;
			}
		}
		if(s->num_apodizations == 32) break;
		if(semi) spec = semi + 1; else break;
	}
	if(s->num_apodizations == 0) {
		s->num_apodizations = 1;
		s->apodizations[0].type = FGH_APOD_TUKEY;
		s->apodizations[0].p = 0.5f;
	}
}

int flacgpu_host_settings_resolve(flacgpu_host_settings *s)
{
	if(s->channels == 0 || s->channels > 8) return FGH_INIT_INVALID_NUMBER_OF_CHANNELS;
	if(s->channels != 2) { s->do_mid_side_stereo = 0; s->loose_mid_side_stereo = 0; }
	else if(!s->do_mid_side_stereo) s->loose_mid_side_stereo = 0;
	if(s->bits_per_sample < 4 || s->bits_per_sample > 32) return FGH_INIT_INVALID_BITS_PER_SAMPLE;
	if(s->sample_rate > 1048575u) return FGH_INIT_INVALID_SAMPLE_RATE;
	if(s->blocksize == 0) s->blocksize = s->max_lpc_order == 0 ? 1152 : 4096;
	if(s->blocksize < 16 || s->blocksize > 65535u) return FGH_INIT_INVALID_BLOCK_SIZE;
	if(s->max_lpc_order > 32) return FGH_INIT_INVALID_MAX_LPC_ORDER;
	if(s->blocksize < s->max_lpc_order) return FGH_INIT_BLOCK_SIZE_TOO_SMALL_FOR_LPC_ORDER;
	if(s->qlp_coeff_precision == 0) {
		const uint32_t bps = s->bits_per_sample, bs = s->blocksize;
		uint32_t p;
		if(bps < 16) { p = 2 + bps / 2; if(p < 5) p = 5; }
		else if(bps == 16) p = bs <= 192 ? 7 : bs <= 384 ? 8 : bs <= 576 ? 9 : bs <= 1152 ? 10 : bs <= 2304 ? 11 : bs <= 4608 ? 12 : 13;
		else p = bs <= 384 ? 13 : bs <= 1152 ? 14 : 15;
		s->qlp_coeff_precision = p;
	}
	else if(s->qlp_coeff_precision < 5 || s->qlp_coeff_precision > 15) return FGH_INIT_INVALID_QLP_COEFF_PRECISION;
	if(s->streamable_subset) {
		const uint32_t bps = s->bits_per_sample, sr = s->sample_rate;
		if(s->blocksize > 16384 || (sr <= 48000 && s->blocksize > 4608)) return FGH_INIT_NOT_STREAMABLE;
		if(sr >= (1u << 16) * 10 || (sr >= (1u << 16) && sr % 10 != 0)) return FGH_INIT_NOT_STREAMABLE;
		if(bps != 8 && bps != 12 && bps != 16 && bps != 20 && bps != 24 && bps != 32) return FGH_INIT_NOT_STREAMABLE;
		if(s->max_residual_partition_order > 8) return FGH_INIT_NOT_STREAMABLE;
		if(sr <= 48000 && (s->blocksize > 4608 || s->max_lpc_order > 12)) return FGH_INIT_NOT_STREAMABLE;
	}
	if(s->max_residual_partition_order >= 16) s->max_residual_partition_order = 15;
	if(s->min_residual_partition_order >= s->max_residual_partition_order) s->min_residual_partition_order = s->max_residual_partition_order;
	return FGH_INIT_OK;
}

int flacgpu_host_engine_config(const flacgpu_host_settings *s, int device, uint32_t max_batch_frames, flacgpu_config *c)
{
	memset(c, 0, sizeof *c);
	/* escape coding is a no-op in production builds of the reference (stream_encoder.c:2107-2114) */
	if(s->max_lpc_order > 0 && s->num_apodizations > FLACGPU_MAX_APODIZATIONS) return FLACGPU_ERR_UNSUPPORTED;
	c->abi_version = FLACGPU_ABI_VERSION;
	c->channels = s->channels; c->bits_per_sample = s->bits_per_sample; c->sample_rate = s->sample_rate;
	c->blocksize = s->blocksize;
	c->do_mid_side_stereo = (uint32_t)s->do_mid_side_stereo; c->loose_mid_side_stereo = (uint32_t)s->loose_mid_side_stereo;
	c->max_lpc_order = s->max_lpc_order; c->qlp_coeff_precision = s->qlp_coeff_precision;
	c->min_residual_partition_order = s->min_residual_partition_order;
	c->max_residual_partition_order = s->max_residual_partition_order;
	c->num_apodizations = s->num_apodizations > FLACGPU_MAX_APODIZATIONS ? FLACGPU_MAX_APODIZATIONS : s->num_apodizations;
	for(uint32_t a = 0; a < c->num_apodizations; a++) {
		const int sub = s->apodizations[a].type == FGH_APOD_SUBDIVIDE_TUKEY;
		c->apodizations[a].kind = sub ? FLACGPU_APOD_SUBDIVIDE_TUKEY : FLACGPU_APOD_WINDOW;
		c->apodizations[a].parts = sub ? (uint32_t)s->apodizations[a].parts : 0;
	}
	c->disable_constant_subframes = (uint32_t)s->disable_constant_subframes;
	c->disable_fixed_subframes = (uint32_t)s->disable_fixed_subframes;
	c->disable_verbatim_subframes = (uint32_t)s->disable_verbatim_subframes;
	c->limit_min_bitrate = (uint32_t)s->limit_min_bitrate;
	c->do_exhaustive_model_search = s->do_exhaustive_model_search ? 1 : 0;
	c->do_qlp_coeff_prec_search = s->do_qlp_coeff_prec_search ? 1 : 0;
	c->device = device;
	c->max_batch_frames = max_batch_frames;
	return FLACGPU_OK;
}

void flacgpu_host_windows(const flacgpu_host_settings *s, uint32_t blocksize, float *out)
{
	for(uint32_t a = 0; a < s->num_apodizations; a++)
		flacgpu_host_window(&s->apodizations[a], out + (size_t)a * blocksize, (int32_t)blocksize);
}
