/*
 * ORACLE (test infrastructure, NOT product code).  See oracle/README.md.
 * Constant tables of the reference, rebuilt from ISO data + the reference's own
 * construction formulas.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load anything under oracle/.
 */
#ifndef ORC_TABLES_H
#define ORC_TABLES_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int xlen, ylen, linbits, linmax;
    const uint16_t *hcod;
    const uint8_t *hlen;
} OrcHuff;

typedef struct {
    /* ---- decoder side (reference mp3stego/decoder/tables.py, Frame.py:16-62) ---- */
    double synth_window[512];      /* tables.py:429-514  round(n/65536, 9)              */
    double synth_matrix[64][32];   /* Frame.py:17-29                                    */
    double sine_block[4][36];      /* Frame.py:33-62                                    */
    double imdct_cos36[36][18];    /* Frame.py:130 with n=36 (evaluated per term there) */
    double imdct_cos12[12][6];     /* Frame.py:130 with n=12                            */
    double alias_cs[8], alias_ca[8]; /* Frame.py:609-611 (10-digit literals)            */
    int sfb_long[3][23];           /* band_index_table.long_*   index: 0=44.1k 1=48k 2=32k */
    int sfb_short_width[3][12];    /* band_width_table.short_*                          */
    int pre_tab[21];
    int slen[16][2];
    int dec_linbits[32];           /* tables.py:423 */
    int dec_max[32];               /* tables.py:426 (0 for tables 4 and 14)             */
    /* ---- encoder side (reference mp3stego/encoder/tables.py, MP3_Encoder.py:528-579) ---- */
    int32_t enwindow[512];         /* tables.py:34   int(round(n/2^21,6)*0x7fffffff)    */
    int32_t fl[32][64];            /* MP3_Encoder.py:536-544                            */
    int32_t cos_l[18][36];         /* MP3_Encoder.py:551-556                            */
    double steptab[128];           /* MP3_Encoder.py:566-574                            */
    int32_t steptabi[128];
    int32_t int2idx[10000];        /* MP3_Encoder.py:578-579                            */
    int32_t mdct_cs[8], mdct_ca[8];/* tables.py:308-332                                 */
    int subdv[23][2];              /* tables.py:335-359                                 */
    int transform[32][2];          /* MP3_Encoder.py:419-449 IDX_TO_TRANSFORM_HUF       */
    int in_h0[32];                 /* decoder/util.py:3                                 */
    OrcHuff huff[34];              /* tables.py:271-304                                 */
} OrcTables;

const OrcTables *orc_tables(void);

#ifdef __cplusplus
}
#endif
#endif
