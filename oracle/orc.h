/*
 * ORACLE (test infrastructure, NOT product code).
 *
 * CPU restatement, in plain C, of the reference's per-frame decode/encode path
 * (tomershay100/mp3-steganography-lib @ mp3stego-lib 1.1.8).  It exists to CHECK
 * the HIP product path: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  The product (mp3-steganography-lib_amd/) never
 * includes, links or calls anything in this directory.
 *
 * Parity status: PINNED.  The reference is Python, so it cannot be compiled into
 * oracle/_ref; instead the restatement is checked against golden vectors that
 * tests/golden/gen_golden.py produced by importing the reference in the build
 * container (tests/test_oracle_*.py): tables, tests/test.mp3 decode (is, side
 * info, stego bits, float64 PCM bit-for-bit, WAV sha256), re-encode at 320 kbps
 * (mdct_freq, ix, GrInfo, scfsi, MP3 sha256; plain / hidden 'ddd' / cleared),
 * per-stage random vectors, and a synthetic 128 kbps stream.
 */
#ifndef ORC_H
#define ORC_H
#include <stdint.h>
#include "orc_tables.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_OK 0
#define ORC_ERR_MALFORMED (-1)  /* input on which the reference raises IndexError/KeyError/ZeroDivisionError */
#define ORC_ERR_UNSUPPORTED (-2)/* mono encode, partial last frame: reference raises IndexError */
#define ORC_ERR_STEP_RANGE (-3) /* quantizer step ran past steptab: reference raises IndexError */

/* per-frame header + side info as the reference holds it (decoder/FrameSideInformation.py:11-37) */
typedef struct {
    int32_t hdr[4];
    int32_t frame_size, nch, sr_idx, main_data_begin;
    int32_t scfsi[2][4];
    int32_t part2_3_length[2][2], big_value[2][2], global_gain[2][2], scale_fac_compress[2][2];
    int32_t window_switching[2][2], block_type[2][2], mixed_block_flag[2][2];
    int32_t region0_count[2][2], region1_count[2][2], pre_flag[2][2], scale_fac_scale[2][2];
    int32_t count1table_select[2][2];
    int32_t table_select[2][2][3], sub_block_gain[2][2][3];
    int32_t scale_fac_l[2][2][22], scale_fac_s[2][2][3][13];
} OrcFrameInfo;

/* ---- decoder ---- */
typedef struct OrcDecoder OrcDecoder;
OrcDecoder *orc_dec_new(void);
void orc_dec_free(OrcDecoder *);
/* decode a whole file image starting at `offset` (after an ID3v2 tag); 0 or ORC_ERR_* */
int orc_dec_run(OrcDecoder *, const uint8_t *file, long flen, long offset);
long orc_dec_n_frames(const OrcDecoder *);
long orc_dec_n_pcm_rows(const OrcDecoder *);
long orc_dec_n_bits(const OrcDecoder *);
int orc_dec_channels(const OrcDecoder *);
int orc_dec_sampling_rate(const OrcDecoder *);
int orc_dec_bit_rate(const OrcDecoder *);
const double *orc_dec_pcm(const OrcDecoder *);        /* [rows][channels] float64 */
const char *orc_dec_bits(const OrcDecoder *);         /* 0/1 per stego bit */
const int16_t *orc_dec_is(const OrcDecoder *);        /* [frames][2 gr][2 ch][576] */
const OrcFrameInfo *orc_dec_frames(const OrcDecoder *);

/* decode stages (decoder/Frame.py), usable stand-alone */
void orc_requantize(double *smp, int global_gain, int scalefac_scale, int block_type, int mixed, int preflag,
                    const int32_t *sub_block_gain, const int32_t *scale_fac_l, const int32_t *scale_fac_s, int sr_idx);
void orc_ms_stereo(double *l, double *r);
void orc_reorder(double *smp, int sr_idx);
void orc_alias_reduction(double *smp);
void orc_imdct(double *smp, int block_type, double *prev);
void orc_frequency_inversion(double *smp);
void orc_synth_filter_bank(double *smp, double *fifo);
int16_t orc_pcm_to_i16(double v);

/* ---- encoder ---- */
typedef struct {
    int32_t part2_3_length, big_values, count1, global_gain, scale_fac_compress, region0_count, region1_count;
    int32_t preflag, scale_fac_scale, count1table_select, part2_length, address1, address2, address3;
    int32_t quantizerStepSize;
    int32_t table_select[3];
} OrcGrInfo;

typedef struct {
    OrcGrInfo gi[2][2];        /* [gr][ch] */
    int32_t scfsi[2][4];
    int32_t written, hide_off, padding;
} OrcEncFrame;

typedef struct OrcEncoder OrcEncoder;
/* pcm: interleaved int16 [n_samples][nch]; hide_bits: 0/1 chars or NULL */
OrcEncoder *orc_enc_new(int samplerate, int nch, int bitrate_kbps, const char *hide_bits, long n_hide);
void orc_enc_free(OrcEncoder *);
int orc_enc_run(OrcEncoder *, const int16_t *pcm, long n_samples_per_ch);
long orc_enc_n_frames(const OrcEncoder *);
long orc_enc_out_len(const OrcEncoder *);
const uint8_t *orc_enc_out(const OrcEncoder *);
long orc_enc_hide_offset(const OrcEncoder *);
const OrcEncFrame *orc_enc_frames(const OrcEncoder *);
const int32_t *orc_enc_mdct_freq(const OrcEncoder *); /* [frames][2 ch][2 gr][576] */
const int32_t *orc_enc_ix(const OrcEncoder *);        /* [frames][2 ch][2 gr][576] (signed, as after format_bitstream) */

/* encode stages usable stand-alone */
/* test hook: the rate loop (MP3_Encoder.py:766-813) of one granule*channel on a given spectrum, fresh GrInfo, budget max_bits */
int orc_enc_rate_unit(OrcEncoder *e, int max_bits, const int32_t *xr576, int32_t *ix576, OrcGrInfo *out);
/* test hook: one probe of the binary search's body (:973-990) at `step` on n spectra, fresh GrInfo; bits -1 = step outside steptab / silence */
void orc_enc_probe_bits(OrcEncoder *e, long n, int step, const int32_t *xr, int32_t *bits, int32_t *big_values, int32_t *count1);
void orc_enc_rate_units(OrcEncoder *e, long n, const int32_t *max_bits, const int32_t *xr, int32_t *ix, OrcGrInfo *out, int32_t *rc);
void orc_enc_window_filter_subband(int32_t *s32, int32_t *x512, int32_t *off);
int32_t orc_enc_quantize(int32_t *ix, int step_size, int32_t xrmax, const int32_t *xr, const int32_t *xrabs);

#ifdef __cplusplus
}
#endif
#endif
