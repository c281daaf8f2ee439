/* pseld_host.h - host-side (CPU) helpers of the data layer, C ABI, built into pseldnets_amd/libpseld_host.so by gcc (no GPU code).
 *
 * FLAC: the reference reads its synthetic datasets' recordings as FLAC through soundfile (data/components/data.py:81 swaps '.wav' for
 * '.flac'; data/data.py:9-13 `sf.read(path, dtype='float32'[, start, stop])[0].T`). libsndfile / libFLAC are not in this image, so the
 * decoder is own code (csrc/host/flac.cpp). PARITY UNPINNED: no FLAC file or encoder exists here to produce a reference decode; instead
 * every check the format carries is enforced - CRC-8 of each frame header and CRC-16 of each frame in the library, the encoder's MD5
 * signature of the unencoded audio (STREAMINFO) in the Python wrapper pseldnets_amd/data/flac.py - so a misread stream is an error.
 * Binding a maintainer of the reference would add: ctypes.CDLL('libpseld_host.so') and `read_flac(path)` in place of `sf.read(path)`. */
#ifndef PSELD_HOST_H
#define PSELD_HOST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Last error of the calling thread (functions below return -1 and leave a message here). */
const char* pseld_host_last_error(void);

/* STREAMINFO of a FLAC stream held in memory (an ID3v2 tag in front is skipped). md5[16]: the encoder's signature of the unencoded
 * audio (interleaved, little-endian, ceil(bits / 8) bytes per sample); all zero = not stored. Any output pointer may be NULL. */
int pseld_flac_info(const uint8_t* data, long n, int* sample_rate, int* channels, int* bits_per_sample, long* total_samples, uint8_t* md5);

/* Decodes the stream into out[sample][channel] (int32, interleaved, sign-extended sample values; room for `capacity` samples per
 * channel). Returns the number of samples per channel, or -1. */
long pseld_flac_decode(const uint8_t* data, long n, int32_t* out, long capacity);

#ifdef __cplusplus
}
#endif
#endif
