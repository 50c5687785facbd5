// oracle/cxx_encode.cpp -- TEST INFRASTRUCTURE: a small client of the reference's C++ wrapper (libFLAC++, FLAC::Encoder::File /
// FLAC::Encoder::Stream), linked once against the reference library and once against libFLACgpu.so (oracle/Makefile: cxx).
// usage: cxx_encode <raw s16le stereo file> <out.flac> <level> [ogg]
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#include "FLAC++/encoder.h"

int main(int argc, char **argv)
{
	if(argc < 4) return 2;
	FILE *f = fopen(argv[1], "rb");
	if(!f) return 2;
	std::vector<short> raw;
	short buf[8192];
	size_t n;
	while((n = fread(buf, sizeof(short), 8192, f)) > 0) raw.insert(raw.end(), buf, buf + n);
	fclose(f);
	const bool ogg = argc > 4 && std::string(argv[4]) == "ogg";
	FLAC::Encoder::File enc;
	bool ok = enc.set_channels(2) && enc.set_bits_per_sample(16) && enc.set_sample_rate(44100) && enc.set_compression_level((uint32_t)atoi(argv[3]))
	          && enc.set_verify(true) && enc.set_total_samples_estimate(raw.size() / 2);
	if(ogg) ok = ok && enc.set_ogg_serial_number(4711);
	if(!ok) { fprintf(stderr, "setters failed\n"); return 1; }
	const ::FLAC__StreamEncoderInitStatus st = ogg ? enc.init_ogg(argv[2]) : enc.init(argv[2]);
	if(st != FLAC__STREAM_ENCODER_INIT_STATUS_OK) { fprintf(stderr, "init: %s\n", FLAC__StreamEncoderInitStatusString[st]); return 1; }
	std::vector<FLAC__int32> pcm(raw.begin(), raw.end());
	const size_t total = pcm.size() / 2;
	for(size_t pos = 0; pos < total; pos += 1000) {
		const size_t c = total - pos < 1000 ? total - pos : 1000;
		if(!enc.process_interleaved(&pcm[2 * pos], (uint32_t)c)) { fprintf(stderr, "process: %s\n", enc.get_state().resolved_as_cstring(enc)); return 1; }
	}
	if(!enc.finish()) { fprintf(stderr, "finish failed\n"); return 1; }
	return 0;
}
