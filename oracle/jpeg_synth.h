/* jpeg_synth.h -- TEST / BENCH INPUT GENERATOR (not part of the product path).
 *
 * The reference ships no sample images and no encoder (SURVEY.md section 4;
 * ExportJpegDo only copies bytes, reference source/JfifDecode.cpp:7632), so the
 * build defines its own reproducible inputs: a seeded synthetic picture
 * (smooth sinusoid colour field + noise) pushed through a
 * small ITU-T T.81 baseline-sequential Huffman encoder (Annex-K tables or
 * per-image optimised tables, IJG-style quality scaling, 4:4:4 / 4:2:2 / 4:2:0 /
 * grayscale, optional DRI), or through a progressive (SOF2) encoder that emits
 * the SAME quantised coefficients as spectral-selection scans.
 */
#ifndef JPEG_SYNTH_H
#define JPEG_SYNTH_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int32_t  width, height;      /* SOF X, Y                                          */
    int32_t  hs, vs;             /* luma sampling factors (1 or 2); chroma is 1x1     */
    int32_t  quality;            /* IJG quality 1..100                                */
    int32_t  restart_interval;   /* MCUs per restart interval, 0 = no DRI             */
    int32_t  gray;               /* 1 = single-component image                        */
    int32_t  optimize_huffman;   /* 1 = per-image optimal tables instead of Annex K   */
    int32_t  progressive;        /* 1 = SOF2 multi-scan (spectral selection only)     */
    int32_t  noise_sigma;        /* std-dev of the per-channel noise (survey: 12)     */
    uint32_t seed;               /* picture seed                                      */
} JsynthParams;

/* Fills rgb[height][width][3] with the seeded synthetic picture. */
void   jsynth_image_rgb(const JsynthParams* p, uint8_t* rgb);

/* Encodes the synthetic picture for `p`.  Returns the number of bytes the file
 * needs; writes it only if that is <= cap (call with cap = 0 to size). */
size_t jsynth_encode(const JsynthParams* p, uint8_t* out, size_t cap);

/* Same, from caller-supplied interleaved RGB pixels. */
size_t jsynth_encode_rgb(const JsynthParams* p, const uint8_t* rgb, uint8_t* out, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
