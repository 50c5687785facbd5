/* flac_amd/csrc/refrest/config.h -- build configuration for the parts of libFLAC that are NOT the encoder (stream decoder,
 * metadata objects / iterators, format helpers, bit reader / writer, CRC, MD5 ...), compiled UNMODIFIED from the reference
 * sources where they lie ($(REF)/src/libFLAC) into flac_amd/lib/libFLAC.so.14 next to this project's encoder: SURVEY.md 8b,
 * "Decoder/metadata/format symbols must still be present in the same libFLAC (the CLI needs them) -- take them from
 * reference sources unchanged".  This file is ours: it states what the reference's CMake Release build selects on an
 * x86-64 Linux host without libogg (config.cmake.h.in of the reference lists the knobs). */
#ifndef FLACGPU_REFREST_CONFIG_H
#define FLACGPU_REFREST_CONFIG_H
#define CPU_IS_BIG_ENDIAN 0
#define WORDS_BIGENDIAN 0
#define ENABLE_64_BIT_WORDS 1
/* HAS_OGG=1 on the make command line (flac_amd/csrc/Makefile) builds the reference's Ogg decoder half against the system's libogg */
#ifdef FLACGPU_DROPIN_HAS_OGG
#define OGG_FOUND 1
#define FLAC__HAS_OGG 1
#else
#define OGG_FOUND 0
#define FLAC__HAS_OGG 0
#endif
#define FLAC__HAS_X86INTRIN 1
#define FLAC__HAS_NEONINTRIN 0
#define FLAC__HAS_A64NEONINTRIN 0
#define FLAC__SYS_LINUX
#define WITH_AVX
#define FLAC__USE_AVX
#define HAVE_BSWAP16
#define HAVE_BSWAP32
#define HAVE_BYTESWAP_H
#define HAVE_CLOCK_GETTIME
#define HAVE_CPUID_H
#define HAVE_FSEEKO
#define HAVE_INTTYPES_H
#define HAVE_LROUND 1
#define HAVE_PTHREAD 1
#define HAVE_STDINT_H
#define HAVE_STDLIB_H
#define HAVE_STRING_H
#define HAVE_SYS_PARAM_H
#define HAVE_SYS_STAT_H
#define HAVE_SYS_TYPES_H
#define HAVE_UNISTD_H
#define HAVE_X86INTRIN_H
#define PACKAGE_VERSION "1.5.0"
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#ifndef _FILE_OFFSET_BITS
#define _FILE_OFFSET_BITS 64
#endif
#endif
