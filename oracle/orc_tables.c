/*
 * ORACLE (test infrastructure, NOT product code).
 * Rebuilds every constant table of the reference hot path with the reference's
 * own construction formulas (same libm calls, same operation order), from ISO
 * data in iso_tables.h.  Pinned bit-for-bit against tests/golden/g1_tables.npz
 * (the reference's evaluated tables) by tests/test_oracle_tables.py.
 */
#include "orc_tables.h"
#include "iso_tables.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static OrcTables T;
static int T_ready = 0;

/* Python round(x, nd): correctly rounded decimal -> nearest double */
static double py_round(double x, int nd)
{
    char buf[64];
    snprintf(buf, sizeof buf, "%.*f", nd, x);
    return strtod(buf, NULL);
}

static const int SFB_LONG[3][23] = {
    /* 44.1 kHz */ {0, 4, 8, 12, 16, 20, 24, 30, 36, 44, 52, 62, 74, 90, 110, 134, 162, 196, 238, 288, 342, 418, 576},
    /* 48 kHz   */ {0, 4, 8, 12, 16, 20, 24, 30, 36, 42, 50, 60, 72, 88, 106, 128, 156, 190, 230, 276, 330, 384, 576},
    /* 32 kHz   */ {0, 4, 8, 12, 16, 20, 24, 30, 36, 44, 54, 66, 82, 102, 126, 156, 194, 240, 296, 364, 448, 550, 576}};
static const int SFB_SHORT_W[3][12] = {
    {4, 4, 4, 4, 6, 8, 10, 12, 14, 18, 22, 30},
    {4, 4, 4, 4, 6, 6, 10, 12, 14, 16, 20, 26},
    {4, 4, 4, 4, 6, 8, 12, 16, 20, 26, 34, 42}};
static const int PRE_TAB[21] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 3, 2};
static const int SLEN[16][2] = {{0, 0}, {0, 1}, {0, 2}, {0, 3}, {3, 0}, {1, 1}, {1, 2}, {1, 3},
                                {2, 1}, {2, 2}, {2, 3}, {3, 1}, {3, 2}, {3, 3}, {4, 2}, {4, 3}};
static const int LINBITS[32] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                1, 2, 3, 4, 6, 8, 10, 13, 4, 5, 6, 7, 8, 9, 11, 13};
static const int DEC_MAX[32] = {1, 2, 3, 3, 0, 4, 4, 6, 6, 6, 8, 8, 8, 16, 0, 16,
                                16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16};
static const int SUBDV[23][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 1}, {1, 1}, {1, 1}, {1, 2}, {2, 2}, {2, 3},
                                 {2, 3}, {3, 4}, {3, 4}, {3, 4}, {4, 5}, {4, 5}, {4, 6}, {5, 6}, {5, 6}, {5, 7}, {6, 7},
                                 {6, 7}};
/* reference MP3_Encoder.py:419-449: (table, bit) -> table.  Row = table, col = bit. */
static const int TRANSFORM[32][2] = {
    {0, 0},   {3, 1},   {3, 2},   {3, 2},   {0, 0},   {6, 5},   {6, 5},   {8, 7},
    {8, 7},   {8, 9},   {11, 10}, {11, 10}, {12, 10}, {15, 13}, {0, 0},   {15, 13},
    {17, 16}, {17, 18}, {19, 18}, {19, 20}, {21, 20}, {21, 22}, {23, 22}, {23, 31},
    {24, 25}, {26, 25}, {26, 27}, {28, 27}, {28, 29}, {30, 29}, {30, 31}, {23, 31}};
static const int H0_SET[14] = {3, 6, 8, 11, 12, 15, 17, 19, 21, 23, 24, 26, 28, 30};
static const double ALIAS_C[8] = {-0.6, -0.535, -0.33, -0.185, -0.095, -0.041, -0.0142, -0.0037};

static void set_huff(int n, int xl, int yl, int lb, int lm, const uint16_t *c, const uint8_t *l)
{
    T.huff[n].xlen = xl; T.huff[n].ylen = yl; T.huff[n].linbits = lb; T.huff[n].linmax = lm;
    T.huff[n].hcod = c; T.huff[n].hlen = l;
}

const OrcTables *orc_tables(void)
{
    if (T_ready) return &T;
    memset(&T, 0, sizeof T);
    const double PI_M = 3.141592653589793; /* math.pi */

    for (int i = 0; i < 512; i++) {
        T.synth_window[i] = py_round((double)ISO_WINDOW_NUM[i] / 65536.0, 9);
        /* encoder/tables.py:34 values == int(round(n/2**21, 6) * 0x7fffffff) */
        T.enwindow[i] = (int32_t)(py_round((double)ISO_WINDOW_NUM[i] / 2097152.0, 6) * 2147483647.0);
    }
    /* Frame.py:24-27 */
    for (int i = 0; i < 64; i++)
        for (int j = 0; j < 32; j++)
            T.synth_matrix[i][j] = cos((16.0 + i) * (2.0 * j + 1.0) * (PI_M / 64.0));
    /* Frame.py:41-60 */
    for (int i = 0; i < 36; i++) T.sine_block[0][i] = sin(PI_M / 36.0 * (i + 0.5));
    for (int i = 0; i < 18; i++) T.sine_block[1][i] = sin(PI_M / 36.0 * (i + 0.5));
    for (int i = 18; i < 24; i++) T.sine_block[1][i] = 1.0;
    for (int i = 24; i < 30; i++) T.sine_block[1][i] = sin(PI_M / 12.0 * (i - 18.0 + 0.5));
    for (int i = 30; i < 36; i++) T.sine_block[1][i] = 1.0;
    for (int i = 0; i < 12; i++) T.sine_block[2][i] = sin(PI_M / 12.0 * (i + 0.5));
    for (int i = 0; i < 6; i++) T.sine_block[3][i] = 0.0;
    for (int i = 6; i < 12; i++) T.sine_block[3][i] = sin(PI_M / 12.0 * (i - 6.0 + 0.5));
    for (int i = 12; i < 18; i++) T.sine_block[3][i] = 1.0;
    for (int i = 18; i < 36; i++) T.sine_block[3][i] = sin(PI_M / 36.0 * (i + 0.5));
    /* Frame.py:130  math.cos(math.pi / (2 * n) * (2 * i + 1 + half_n) * (2 * k + 1)) */
    for (int i = 0; i < 36; i++)
        for (int k = 0; k < 18; k++)
            T.imdct_cos36[i][k] = cos(PI_M / (double)(2 * 36) * (double)(2 * i + 1 + 18) * (double)(2 * k + 1));
    for (int i = 0; i < 12; i++)
        for (int k = 0; k < 6; k++)
            T.imdct_cos12[i][k] = cos(PI_M / (double)(2 * 12) * (double)(2 * i + 1 + 6) * (double)(2 * k + 1));
    /* Frame.py:609-611: 10-decimal literals of ISO Table B.9 derived cs/ca */
    for (int i = 0; i < 8; i++) {
        double c = ALIAS_C[i];
        T.alias_cs[i] = py_round(1.0 / sqrt(1.0 + c * c), 10);
        T.alias_ca[i] = py_round(c / sqrt(1.0 + c * c), 10);
        /* encoder/tables.py:308-313 */
        T.mdct_ca[i] = (int32_t)(c / sqrt(1.0 + (c * c)) * 2147483647.0);
        T.mdct_cs[i] = (int32_t)(1.0 / sqrt(1.0 + (c * c)) * 2147483647.0);
    }
    memcpy(T.sfb_long, SFB_LONG, sizeof SFB_LONG);
    memcpy(T.sfb_short_width, SFB_SHORT_W, sizeof SFB_SHORT_W);
    memcpy(T.pre_tab, PRE_TAB, sizeof PRE_TAB);
    memcpy(T.slen, SLEN, sizeof SLEN);
    memcpy(T.dec_linbits, LINBITS, sizeof LINBITS);
    memcpy(T.dec_max, DEC_MAX, sizeof DEC_MAX);
    memcpy(T.subdv, SUBDV, sizeof SUBDV);
    memcpy(T.transform, TRANSFORM, sizeof TRANSFORM);
    for (int i = 0; i < 14; i++) T.in_h0[H0_SET[i]] = 1;

    /* MP3_Encoder.py:536-544 (util.PI64 = 0.049087385212) */
    for (int i = 0; i < 32; i++)
        for (int j = 0; j < 64; j++) {
            double f = 1e9 * cos((double)((2 * i + 1) * (16 - j)) * 0.049087385212);
            double ip;
            if (f >= 0) modf(f + 0.5, &ip); else modf(f - 0.5, &ip);
            T.fl[i][j] = (int32_t)(ip * 2147483647.0 * 1e-9);
        }
    /* MP3_Encoder.py:551-556 (util.PI36 = 0.087266462599717, util.PI = 3.14159265358979) */
    for (int m = 0; m < 18; m++)
        for (int k = 0; k < 36; k++)
            T.cos_l[m][k] = (int32_t)(sin(0.087266462599717 * (k + 0.5)) *
                                      cos((3.14159265358979 / 72) * (double)(2 * k + 19) * (double)(2 * m + 1)) *
                                      2147483647.0);
    /* MP3_Encoder.py:566-579 */
    for (int i = 0; i < 128; i++) {
        T.steptab[i] = pow(2.0, (double)(127 - i) / 4);
        if (T.steptab[i] * 2 > 2147483647.0) T.steptabi[i] = 0x7fffffff;
        else T.steptabi[i] = (int32_t)(T.steptab[i] * 2 + 0.5);
    }
    for (int i = 0; i < 10000; i++)
        T.int2idx[i] = (int32_t)(sqrt(sqrt((double)i) * (double)i) - 0.0946 + 0.5);

    /* encoder/tables.py:271-304 */
    set_huff(0, 0, 0, 0, 0, NULL, NULL);
    set_huff(1, 2, 2, 0, 0, ISO_HCOD_1, ISO_HLEN_1);
    set_huff(2, 3, 3, 0, 0, ISO_HCOD_2, ISO_HLEN_2);
    set_huff(3, 3, 3, 0, 0, ISO_HCOD_3, ISO_HLEN_3);
    set_huff(4, 0, 0, 0, 0, NULL, NULL);
    set_huff(5, 4, 4, 0, 0, ISO_HCOD_5, ISO_HLEN_5);
    set_huff(6, 4, 4, 0, 0, ISO_HCOD_6, ISO_HLEN_6);
    set_huff(7, 6, 6, 0, 0, ISO_HCOD_7, ISO_HLEN_7);
    set_huff(8, 6, 6, 0, 0, ISO_HCOD_8, ISO_HLEN_8);
    set_huff(9, 6, 6, 0, 0, ISO_HCOD_9, ISO_HLEN_9);
    set_huff(10, 8, 8, 0, 0, ISO_HCOD_10, ISO_HLEN_10);
    set_huff(11, 8, 8, 0, 0, ISO_HCOD_11, ISO_HLEN_11);
    set_huff(12, 8, 8, 0, 0, ISO_HCOD_12, ISO_HLEN_12);
    set_huff(13, 16, 16, 0, 0, ISO_HCOD_13, ISO_HLEN_13);
    set_huff(14, 0, 0, 0, 0, NULL, NULL);
    set_huff(15, 16, 16, 0, 0, ISO_HCOD_15, ISO_HLEN_15);
    static const int lb16[8] = {1, 2, 3, 4, 6, 8, 10, 13}, lb24[8] = {4, 5, 6, 7, 8, 9, 11, 13};
    for (int k = 0; k < 8; k++) {
        set_huff(16 + k, 16, 16, lb16[k], (1 << lb16[k]) - 1, ISO_HCOD_16, ISO_HLEN_16);
        set_huff(24 + k, 16, 16, lb24[k], (1 << lb24[k]) - 1, ISO_HCOD_24, ISO_HLEN_24);
    }
    set_huff(32, 1, 16, 0, 0, ISO_HCOD_32, ISO_HLEN_32);
    set_huff(33, 1, 16, 0, 0, ISO_HCOD_33, ISO_HLEN_33);
    T_ready = 1;
    return &T;
}
