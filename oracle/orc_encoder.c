/*
 * ORACLE (test infrastructure, NOT product code).
 * Plain-C restatement of the reference encoder mp3stego/encoder/MP3_Encoder.py
 * (+ encoder/util.py fixed-point helpers, encoder/encoder.py driver).  Each
 * function cites the reference lines it follows; quirks E1-E16 of SURVEY.md
 * Appendix A are reproduced on purpose.
 */
#include "orc.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* encoder/util.py:123-155 */
static inline int32_t mulsr(int32_t a, int32_t b) { return (int32_t)(((int64_t)a * b + 1073741824LL) >> 31); }
static inline int32_t mulr(int32_t a, int32_t b) { return (int32_t)(((int64_t)a * b + 2147483648LL) >> 32); }
static inline int32_t mul(int32_t a, int32_t b) { return (int32_t)(((int64_t)a * b) >> 32); }
static inline void cmuls(int32_t are, int32_t aim, int32_t bre, int32_t bim, int32_t *dre, int32_t *dim)
{
    int32_t tre = (int32_t)(((int64_t)are * bre - (int64_t)aim * bim) >> 31);
    *dim = (int32_t)(((int64_t)are * bim + (int64_t)aim * bre) >> 31);
    *dre = tre;
}
static inline int32_t labs32(int32_t a) { return (int32_t)(a < 0 ? -(int64_t)a : (int64_t)a); }
static inline int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }

struct OrcEncoder {
    int nch, samplerate, bitrate, samplerate_index, bitrate_index, version, mode;
    /* MPEG struct :41-61 */
    int padding, bits_per_frame, whole_slots_per_frame, mean_bits, side_info_len;
    double frac_slots_per_frame, slot_lag;
    /* Subband :21-29, l3_sb_sample :469 */
    int32_t x[2][512]; int32_t off[2];
    int32_t l3_sb_sample[2][3][18][32];
    int32_t mdct_freq[2][2][576];
    int32_t l3_enc[2][2][576];
    /* L3Loop :145-168 */
    int32_t xrsq[576], xrabs[576]; int32_t xrmax;
    int32_t en_tot[2], en[2][21], xm[2][21], xrmaxl[2];
    const int32_t *xr;
    OrcGrInfo gi[2][2];
    int32_t scfsi[2][4];
    double resv_size; int resv_max;
    /* bitstream :65-77 */
    uint8_t *data; long data_size, data_position; uint32_t cache; int cache_bits;
    /* hide :525-526 */
    char *hide; long n_hide; int hiding; long hide_off;
    /* wav cursor (WAV_Reader.py:109,160-164) */
    const int16_t *buf; long buf_len; long buf_pos[2];
    /* outputs */
    uint8_t *out; long out_len, out_cap;
    OrcEncFrame *frames; int32_t *rec_mdct; int32_t *rec_ix; long n_frames, cap_frames;
    int error;
};

static const int BITRATES_V1[16] = {-1, 32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320, -1};

/* MP3_Encoder.py:462-526 */
OrcEncoder *orc_enc_new(int samplerate, int nch, int bitrate_kbps, const char *hide_bits, long n_hide)
{
    orc_tables();
    OrcEncoder *e = (OrcEncoder *)calloc(1, sizeof *e);
    e->nch = nch; e->samplerate = samplerate; e->bitrate = bitrate_kbps;
    e->mode = nch > 1 ? 0 : 3;                          /* WAV_Reader.py:59-64 */
    e->samplerate_index = samplerate == 44100 ? 0 : samplerate == 48000 ? 1 : samplerate == 32000 ? 2 : -1;
    e->version = 3;                                     /* MPEG-I only (WAV_Reader.py:68) */
    e->bitrate_index = -1;
    for (int i = 0; i < 16; i++) if (BITRATES_V1[i] == bitrate_kbps) { e->bitrate_index = i; break; }
    if (e->samplerate_index < 0 || e->bitrate_index < 0 || nch < 1 || nch > 2) e->error = ORC_ERR_UNSUPPORTED;
    /* :503-513 */
    double avg = ((double)2 * 576 / ((double)samplerate)) * (1000 * (double)bitrate_kbps / (double)8);
    e->whole_slots_per_frame = (int)avg;
    e->frac_slots_per_frame = avg - (double)e->whole_slots_per_frame;
    e->slot_lag = -e->frac_slots_per_frame;
    e->padding = 0;
    e->side_info_len = 8 * (nch == 1 ? 4 + 17 : 4 + 32);
    e->data_size = 4096; e->data = (uint8_t *)calloc(e->data_size, 1);
    e->data_position = 0; e->cache = 0; e->cache_bits = 32;
    e->hiding = (hide_bits && n_hide > 0);              /* hide_str != "" (:1257) */
    e->n_hide = e->hiding ? n_hide : 0;
    e->hide = (char *)malloc(e->n_hide + 1);
    if (e->hiding) memcpy(e->hide, hide_bits, n_hide);
    return e;
}

void orc_enc_free(OrcEncoder *e)
{
    if (!e) return;
    free(e->data); free(e->hide); free(e->out); free(e->frames); free(e->rec_mdct); free(e->rec_ix); free(e);
}

/* MP3_Encoder.py:321-370 window_filter_sub_band (x already holds the 32 new samples) */
void orc_enc_window_filter_subband(int32_t *s, int32_t *x, int32_t *off)
{
    const OrcTables *T = orc_tables();
    int32_t tmp[64];
    for (int i = 63; i >= 0; i--) {
        int32_t v = mul(x[(*off + i + (0 << 6)) & 511], T->enwindow[i + (0 << 6)]);
        for (int k = 1; k < 8; k++) v = wadd(v, mul(x[(*off + i + (k << 6)) & 511], T->enwindow[i + (k << 6)]));
        tmp[i] = v;
    }
    *off = (*off + 480) & 511;
    for (int i = 31; i >= 0; i--) {
        int32_t v = mul(T->fl[i][63], tmp[63]);
        for (int j = 62; j >= 0; j--) v = wadd(v, mul(T->fl[i][j], tmp[j]));
        s[i] = v;
    }
}

/* MP3_Encoder.py:751-758 */
static void replace_samples(OrcEncoder *e, int ch)
{
    for (int i = 31; i >= 0; i--) {
        if (e->buf_pos[ch] >= e->buf_len) { e->error = ORC_ERR_UNSUPPORTED; return; } /* IndexError in the reference */
        e->x[ch][i + e->off[ch]] = (int32_t)((uint32_t)(int32_t)e->buf[e->buf_pos[ch]] << 16);
        e->buf_pos[ch] += 2;
    }
}

/* MP3_Encoder.py:652-749 */
static void mdct_sub(OrcEncoder *e)
{
    const OrcTables *T = orc_tables();
    int32_t mdct_in[36];
    for (int ch = e->nch - 1; ch >= 0; ch--) {
        for (int gr = 0; gr < 2; gr++) {
            for (int k = 0; k < 18; k += 2) {
                replace_samples(e, ch); if (e->error) return;
                orc_enc_window_filter_subband(e->l3_sb_sample[ch][gr + 1][k], e->x[ch], &e->off[ch]);
                replace_samples(e, ch); if (e->error) return;
                orc_enc_window_filter_subband(e->l3_sb_sample[ch][gr + 1][k + 1], e->x[ch], &e->off[ch]);
                for (int band = 1; band < 32; band += 2)
                    e->l3_sb_sample[ch][gr + 1][k + 1][band] = (int32_t)(0u - (uint32_t)e->l3_sb_sample[ch][gr + 1][k + 1][band]);
            }
            int32_t (*mf)[18] = (int32_t (*)[18])e->mdct_freq[ch][gr];
            for (int band = 0; band < 32; band++) {
                for (int k = 17; k >= 0; k--) {
                    mdct_in[k] = e->l3_sb_sample[ch][gr][k][band];
                    mdct_in[k + 18] = e->l3_sb_sample[ch][gr + 1][k][band];
                }
                for (int k = 17; k >= 0; k--) {
                    int32_t vm = mul(mdct_in[35], T->cos_l[k][35]);
                    for (int j = 34; j >= 0; j--) vm = wadd(vm, mul(mdct_in[j], T->cos_l[k][j]));
                    mf[band][k] = vm;
                }
                if (band != 0)
                    for (int i = 0; i < 8; i++)
                        cmuls(mf[band][i], mf[band - 1][17 - i], T->mdct_cs[i], T->mdct_ca[i], &mf[band][i], &mf[band - 1][17 - i]);
            }
        }
        memcpy(e->l3_sb_sample[ch][0], e->l3_sb_sample[ch][2], sizeof e->l3_sb_sample[ch][0]);
    }
}

/* MP3_Encoder.py:373-415 */
int32_t orc_enc_quantize(int32_t *ix, int step_size, int32_t xrmax, const int32_t *xr, const int32_t *xrabs)
{
    const OrcTables *T = orc_tables();
    int32_t ix_max = 0;
    int idx = step_size + 127;
    if (idx < 0) idx += 128;              /* Python negative index */
    if (idx < 0 || idx > 127) return -1;  /* IndexError in the reference */
    int32_t scalei = T->steptabi[idx];
    if (mulr(xrmax, scalei) > 165140) ix_max = 16384;
    else
        for (int i = 0; i < 576; i++) {
            int32_t ln = mulr(labs32(xr[i]), scalei);
            if (ln < 10000) ix[i] = T->int2idx[ln];
            else {
                double scale = T->steptab[idx];
                double dbl = (double)xrabs[i] * scale * 4.656612875e-10;
                ix[i] = (int32_t)sqrt(sqrt(dbl) * dbl);
            }
            if (ix_max < ix[i]) ix_max = ix[i];
        }
    return ix_max;
}

/* MP3_Encoder.py:266-291 */
static void calc_run_len(const int32_t *ix, OrcGrInfo *c)
{
    int i = 576;
    while (i > 1) { if (ix[i - 1] == 0 && ix[i - 2] == 0) i -= 2; else break; }
    c->count1 = 0;
    while (i > 3) {
        if (ix[i - 1] <= 1 && ix[i - 2] <= 1 && ix[i - 3] <= 1 && ix[i - 4] <= 1) { c->count1 += 1; i -= 4; }
        else break;
    }
    c->big_values = i >> 1;
}

/* MP3_Encoder.py:171-211 */
static int count1_bit_count(const int32_t *ix, OrcGrInfo *c)
{
    const OrcTables *T = orc_tables();
    int i = c->big_values << 1, sum0 = 0, sum1 = 0;
    for (int k = 0; k < c->count1; k++) {
        int v = ix[i], w = ix[i + 1], x = ix[i + 2], y = ix[i + 3];
        int p = v + (w << 1) + (x << 2) + (y << 3);
        int sb = (v != 0) + (w != 0) + (x != 0) + (y != 0);
        sum0 += sb; sum1 += sb;
        sum0 += T->huff[32].hlen[p]; sum1 += T->huff[33].hlen[p];
        i += 4;
    }
    if (sum0 < sum1) { c->count1table_select = 0; return sum0; }
    c->count1table_select = 1; return sum1;
}

/* MP3_Encoder.py:214-263 */
static int count_bit(const int32_t *ix, int start, int end, int table)
{
    const OrcTables *T = orc_tables();
    if (table == 0) return 0;
    const OrcHuff *h = &T->huff[table];
    int sum = 0;
    if (table > 15) {
        for (int i = start; i < end; i += 2) {
            int x = ix[i], y = ix[i + 1];
            if (x > 14) { x = 15; sum += h->linbits; }
            if (y > 14) { y = 15; sum += h->linbits; }
            sum += h->hlen[x * h->ylen + y];
            if (x) sum += 1;
            if (y) sum += 1;
        }
    } else {
        for (int i = start; i < end; i += 2) {
            int x = ix[i], y = ix[i + 1];
            sum += h->hlen[x * h->ylen + y];
            if (x != 0) sum += 1;
            if (y != 0) sum += 1;
        }
    }
    return sum;
}

/* MP3_Encoder.py:998-1036 (flattened scale_fact_band_index, E7: nothing touched when big_values == 0) */
static void subdivide(OrcEncoder *e, OrcGrInfo *c)
{
    const OrcTables *T = orc_tables();
    if (c->big_values == 0) { c->region0_count = 0; c->region1_count = 0; return; }
    const int *sfb = T->sfb_long[e->samplerate_index];
    int bvr = 2 * c->big_values;
    int scfb_anz = 0;
    while (sfb[scfb_anz] < bvr) scfb_anz++;
    int tc = T->subdv[scfb_anz][0];
    while (tc > 0) { if (sfb[tc + 1] <= bvr) break; tc--; }
    c->region0_count = tc;
    c->address1 = sfb[tc + 1];
    const int *sfb2 = sfb + tc + 1;
    tc = T->subdv[scfb_anz][1];
    while (tc > 0) { if (sfb2[tc + 1] <= bvr) break; tc--; }
    c->region1_count = tc;
    c->address2 = sfb2[tc + 1];
    c->address3 = bvr;
}

/* MP3_Encoder.py:1170-1264 */
static int new_choose_table(OrcEncoder *e, const int32_t *ix, int begin, int end, long idx)
{
    const OrcTables *T = orc_tables();
    int ix_max = 0;
    for (int i = begin; i < end; i++) if (ix[i] > ix_max) ix_max = ix[i];
    if (ix_max == 0) return 0;
    int choice[2] = {0, 0}, sum[2] = {0, 0};
    if (ix_max < 15) {
        for (int i = 13; i >= 0; i--) if (T->huff[i].xlen > ix_max) { choice[0] = i; break; }
        sum[0] = count_bit(ix, begin, end, choice[0]);
        switch (choice[0]) {
        case 2: sum[1] = count_bit(ix, begin, end, 3); if (sum[1] <= sum[0]) choice[0] = 3; break;
        case 5: sum[1] = count_bit(ix, begin, end, 6); if (sum[1] <= sum[0]) choice[0] = 6; break;
        case 7:
            sum[1] = count_bit(ix, begin, end, 8); if (sum[1] <= sum[0]) choice[0] = 8;
            sum[1] = count_bit(ix, begin, end, 9); if (sum[1] <= sum[0]) choice[0] = 9;
            break;
        case 10:
            sum[1] = count_bit(ix, begin, end, 11); if (sum[1] <= sum[0]) choice[0] = 11;
            sum[1] = count_bit(ix, begin, end, 12); if (sum[1] <= sum[0]) choice[0] = 12;
            break;
        case 13: sum[1] = count_bit(ix, begin, end, 15); if (sum[1] <= sum[0]) choice[0] = 15; break;
        default: break;
        }
    } else {
        ix_max -= 15;
        for (int i = 15; i < 24; i++) if (T->huff[i].linmax >= ix_max) { choice[0] = i; break; }
        for (int i = 24; i < 32; i++) if (T->huff[i].linmax >= ix_max) { choice[1] = i; break; }
        sum[0] = count_bit(ix, begin, end, choice[0]);
        sum[1] = count_bit(ix, begin, end, choice[1]);
        if (sum[1] < sum[0]) choice[0] = choice[1];
    }
    if (e->hiding) {
        if (idx < e->n_hide) return T->transform[choice[0]][e->hide[idx] & 1]; /* accepts 0/1 or '0'/'1' */
        return choice[0];
    }
    return choice[0];
}

/* MP3_Encoder.py:1147-1168 */
static void big_v_tab_select(OrcEncoder *e, const int32_t *ix, OrcGrInfo *c)
{
    long idx = e->hide_off;
    c->table_select[0] = c->address1 <= 0 ? 0 : new_choose_table(e, ix, 0, c->address1, e->hide_off);
    if (c->table_select[0] > 0) idx += 1;
    c->table_select[1] = c->address2 <= c->address1 ? 0 : new_choose_table(e, ix, c->address1, c->address2, idx);
    if (c->table_select[1] > 0) idx += 1;
    c->table_select[2] = (c->big_values << 1) <= c->address2 ? 0 : new_choose_table(e, ix, c->address2, c->big_values << 1, idx);
}

/* MP3_Encoder.py:294-318 */
static int big_v_bit_count(const int32_t *ix, const OrcGrInfo *c)
{
    int bits = 0;
    if (c->table_select[0]) bits += count_bit(ix, 0, c->address1, c->table_select[0]);
    if (c->table_select[1]) bits += count_bit(ix, c->address1, c->address2, c->table_select[1]);
    if (c->table_select[2]) bits += count_bit(ix, c->address2, c->address3, c->table_select[2]);
    return bits;
}

static int rate_body(OrcEncoder *e, int32_t *ix, OrcGrInfo *c)
{
    calc_run_len(ix, c);
    int bits = count1_bit_count(ix, c);
    subdivide(e, c);
    big_v_tab_select(e, ix, c);
    bits += big_v_bit_count(ix, c);
    return bits;
}

/* MP3_Encoder.py:958-996 */
static int bin_search_step_size(OrcEncoder *e, int desired_rate, int32_t *ix, OrcGrInfo *c)
{
    int next = -120, count = 120;
    do {
        int half = count / 2, bit;
        int32_t q = orc_enc_quantize(ix, next + half, e->xrmax, e->xr, e->xrabs);
        if (q < 0) { e->error = ORC_ERR_STEP_RANGE; return next; }
        if (q > 8192) bit = 100000;
        else bit = rate_body(e, ix, c);
        if (bit < desired_rate) count = half;
        else { next += half; count -= half; }
    } while (count > 1);
    return next;
}

/* MP3_Encoder.py:1064-1095 */
static int inner_loop(OrcEncoder *e, int32_t *ix, int max_bits, OrcGrInfo *c)
{
    int bits;
    if (max_bits < 0) c->quantizerStepSize -= 1;
    do {
        for (;;) {
            int32_t q = orc_enc_quantize(ix, c->quantizerStepSize + 1, e->xrmax, e->xr, e->xrabs);
            if (q < 0) { e->error = ORC_ERR_STEP_RANGE; return 0; }
            if (q > 8192) c->quantizerStepSize += 1; else break;
        }
        c->quantizerStepSize += 1;
        bits = rate_body(e, ix, c);
    } while (bits > max_bits);
    return bits;
}

/* MP3_Encoder.py:817-892 (E5: en_tot/en/xm/xrmaxl indexed by granule only) */
static void calc_scfsi(OrcEncoder *e, int ch, int gr)
{
    const OrcTables *T = orc_tables();
    static const int scfsi_band_long[5] = {0, 6, 11, 16, 21};
    const int *sfbl = T->sfb_long[e->samplerate_index];
    int condition = 0;
    e->xrmaxl[gr] = e->xrmax;
    int32_t temp = 0;
    for (int i = 575; i >= 0; i--) temp = wadd(temp, e->xrsq[i] >> 10);
    if (temp) e->en_tot[gr] = (int32_t)(log((double)temp * 4.768371584e-7) / 0.69314718);
    else e->en_tot[gr] = 0;
    for (int sfb = 20; sfb >= 0; sfb--) {
        temp = 0;
        for (int i = sfbl[sfb]; i < sfbl[sfb + 1]; i++) temp = wadd(temp, e->xrsq[i] >> 10);
        if (temp) e->en[gr][sfb] = (int32_t)(log((double)temp * 4.768371584e-7) / 0.69314718);
        else e->en[gr][sfb] = 0;
        e->xm[gr][sfb] = 0;
    }
    if (gr == 1) {
        for (int gr2 = 1; gr2 >= 0; gr2--) { if (e->xrmaxl[gr2]) condition++; condition++; }
        if (abs(e->en_tot[0] - e->en_tot[1]) < 10) condition++;
        int tp = 0;
        for (int sfb = 20; sfb >= 0; sfb--) tp += abs(e->en[0][sfb] - e->en[1][sfb]);
        if (tp < 100) condition++;
        if (condition == 6) {
            for (int b = 0; b < 4; b++) {
                int sum0 = 0, sum1 = 0;
                e->scfsi[ch][b] = 0;
                for (int sfb = scfsi_band_long[b]; sfb < scfsi_band_long[b + 1]; sfb++) {
                    sum0 += abs(e->en[0][sfb] - e->en[1][sfb]);
                    sum1 += abs(e->xm[0][sfb] - e->xm[1][sfb]);
                }
                e->scfsi[ch][b] = (sum0 < 10 && sum1 < 10) ? 1 : 0;
            }
        } else
            for (int b = 0; b < 4; b++) e->scfsi[ch][b] = 0;
    }
}

/* MP3_Encoder.py:1097-1145 */
static void resv_frame_end(OrcEncoder *e)
{
    if (e->nch == 2 && (e->mean_bits & 1)) e->resv_size += 1;
    double over_bits = e->resv_size - e->resv_max;
    if (over_bits < 0) over_bits = 0;
    e->resv_size -= over_bits;
    double stuffing_bits = over_bits + 0;
    over_bits = fmod(e->resv_size, 8); if (over_bits < 0) over_bits += 8; /* Python % */
    if (over_bits) { stuffing_bits += over_bits; e->resv_size -= over_bits; }
    if (stuffing_bits) {
        OrcGrInfo *gi = &e->gi[0][0];
        if (gi->part2_3_length + stuffing_bits < 4095) gi->part2_3_length = (int32_t)(gi->part2_3_length + stuffing_bits);
        else {
            for (int gr = 0; gr < 2; gr++)
                for (int ch = 0; ch < e->nch; ch++) {
                    gi = &e->gi[gr][ch];
                    if (!stuffing_bits) break;
                    double extra = 4095 - gi->part2_3_length;
                    double this_gr = extra < stuffing_bits ? extra : stuffing_bits;
                    gi->part2_3_length = (int32_t)(gi->part2_3_length + this_gr);
                    stuffing_bits -= this_gr;
                }
            /* resv_drain = stuffing_bits: never written out (E6) */
        }
    }
}

/* MP3_Encoder.py:766-813: the body of __iteration_loop for one (ch, gr) */
static void iterate_unit(OrcEncoder *e, int ch, int gr)
{
    int32_t *ix = e->l3_enc[ch][gr];
    e->xr = e->mdct_freq[ch][gr];
    e->xrmax = 0;
    for (int i = 575; i >= 0; i--) {
        e->xrsq[i] = mulsr(e->xr[i], e->xr[i]);
        e->xrabs[i] = labs32(e->xr[i]);
        if (e->xrabs[i] > e->xrmax) e->xrmax = e->xrabs[i];
    }
    OrcGrInfo *c = &e->gi[gr][ch];
    calc_scfsi(e, ch, gr);
    /* :894-931: resv_max == 0 -> mean_bits // nch capped at 4095 */
    int max_bits = e->mean_bits / e->nch;
    if (max_bits > 4095) max_bits = 4095;
    /* :788-803 (address1/2/3 and quantizerStepSize are NOT reset) */
    c->part2_3_length = 0; c->big_values = 0; c->count1 = 0; c->scale_fac_compress = 0;
    c->table_select[0] = c->table_select[1] = c->table_select[2] = 0;
    c->region0_count = 0; c->region1_count = 0; c->part2_length = 0; c->preflag = 0;
    c->scale_fac_scale = 0; c->count1table_select = 0;
    if (e->xrmax) {
        /* :933-956 outer_loop */
        c->quantizerStepSize = bin_search_step_size(e, max_bits, ix, c);
        if (e->error) return;
        c->part2_length = 0; /* :1038-1062 with scale_fac_compress == 0 */
        int huff_bits = max_bits - c->part2_length;
        int bits = inner_loop(e, ix, huff_bits, c);
        if (e->error) return;
        c->part2_3_length = c->part2_length + bits;
        e->hide_off += (c->table_select[0] > 0) + (c->table_select[1] > 0) + (c->table_select[2] > 0);
    }
    e->resv_size += ((double)e->mean_bits / e->nch) - c->part2_3_length;
    c->global_gain = c->quantizerStepSize + 210;
}

/* MP3_Encoder.py:760-815 */
static void iteration_loop(OrcEncoder *e)
{
    for (int ch = 0; ch < e->nch; ch++)
        for (int gr = 0; gr < 2; gr++) {
            iterate_unit(e, ch, gr);
            if (e->error) return;
        }
    resv_frame_end(e);
}

/* Test hook (not a function of the reference): the rate loop of ONE granule*channel on a spectrum handed in, as a stream's first frame
 * sees it -- fresh GrInfo (addresses and quantizerStepSize 0), the message `e` was made with at its start -- for a budget of max_bits.
 * Lets tests/ put spectra in front of the device's rate loop that PCM through the filter bank does not produce (a lone line, empty
 * regions below the last big value).  e: orc_enc_new(samplerate, 2, any bitrate, ...). */
int orc_enc_rate_unit(OrcEncoder *e, int max_bits, const int32_t *xr576, int32_t *ix576, OrcGrInfo *out)
{
    if (e->error) return e->error;
    memcpy(e->mdct_freq[0][0], xr576, 576 * sizeof(int32_t));
    memset(&e->gi[0][0], 0, sizeof e->gi[0][0]);
    memset(e->l3_enc[0][0], 0, sizeof e->l3_enc[0][0]);
    e->hide_off = 0;
    e->mean_bits = max_bits * e->nch;
    iterate_unit(e, 0, 0);
    memcpy(ix576, e->l3_enc[0][0], 576 * sizeof(int32_t));
    *out = e->gi[0][0];
    int rc = e->error;
    e->error = 0;
    return rc;
}

/* Test hook: ONE probe of __bin_search_step_size's body (:973-990) on n spectra with fresh GrInfo: bits[i] = the probe's `bit` at
 * `step` (100000 when quantize refuses), and the run lengths it leaves -- what a bound on a probe's bits is checked against. */
void orc_enc_probe_bits(OrcEncoder *e, long n, int step, const int32_t *xr, int32_t *bits, int32_t *big_values, int32_t *count1)
{
    for (long i = 0; i < n; i++) {
        int32_t *ix = e->l3_enc[0][0];
        OrcGrInfo *c = &e->gi[0][0];
        memset(c, 0, sizeof *c);
        e->xr = xr + i * 576;
        e->xrmax = 0;
        for (int k = 0; k < 576; k++) { e->xrabs[k] = labs32(e->xr[k]); if (e->xrabs[k] > e->xrmax) e->xrmax = e->xrabs[k]; }
        int32_t q = e->xrmax ? orc_enc_quantize(ix, step, e->xrmax, e->xr, e->xrabs) : -1;
        if (q < 0) { bits[i] = -1; big_values[i] = count1[i] = 0; continue; }
        bits[i] = q > 8192 ? 100000 : rate_body(e, ix, c);
        big_values[i] = q > 8192 ? -1 : c->big_values; count1[i] = q > 8192 ? -1 : c->count1;
    }
}

/* the same for n spectra, each with its own budget; rc[i] = 0 or ORC_ERR_STEP_RANGE */
void orc_enc_rate_units(OrcEncoder *e, long n, const int32_t *max_bits, const int32_t *xr, int32_t *ix, OrcGrInfo *out, int32_t *rc)
{
    for (long i = 0; i < n; i++) rc[i] = orc_enc_rate_unit(e, max_bits[i], xr + i * 576, ix + i * 576, out + i);
}

/* MP3_Encoder.py:1362-1392 */
static void put_bits(OrcEncoder *e, uint32_t val, int N)
{
    if (e->cache_bits > N) {
        e->cache_bits -= N;
        e->cache |= (e->cache_bits < 32) ? (val << e->cache_bits) : 0; /* numpy shift by 32 on uint32 gives 0 */
    } else {
        if (e->data_position + 4 >= e->data_size) {
            long ns = e->data_size + e->data_size / 2;
            e->data = (uint8_t *)realloc(e->data, ns);
            memset(e->data + e->data_size, 0, ns - e->data_size);
            e->data_size = ns;
        }
        N -= e->cache_bits;
        e->cache |= (N < 32) ? (val >> N) : 0;
        e->data[e->data_position + 0] = (uint8_t)(e->cache >> 24);
        e->data[e->data_position + 1] = (uint8_t)(e->cache >> 16);
        e->data[e->data_position + 2] = (uint8_t)(e->cache >> 8);
        e->data[e->data_position + 3] = (uint8_t)(e->cache);
        e->data_position += 4;
        e->cache_bits = 32 - N;
        if (N != 0) e->cache = val << e->cache_bits; else e->cache = 0;
    }
}
static long get_bits_count(const OrcEncoder *e) { return e->data_position * 8 + 32 - e->cache_bits; }

/* MP3_Encoder.py:1448-1513 */
static void huffman_code(OrcEncoder *e, int table_select, int x, int y)
{
    const OrcTables *T = orc_tables();
    uint32_t ext = 0; int x_bits = 0;
    int sign_x = x > 0 ? 0 : 1; if (x <= 0) x = -x;     /* util.abs_and_sign :167-172 */
    int sign_y = y > 0 ? 0 : 1; if (y <= 0) y = -y;
    const OrcHuff *h = &T->huff[table_select];
    if (table_select > 15) {
        int lbx = 0, lby = 0, lin_bits = h->linbits;
        if (x > 14) { lbx = x - 15; x = 15; }
        if (y > 14) { lby = y - 15; y = 15; }
        int idx = x * h->ylen + y;
        uint32_t code = h->hcod[idx]; int c_bits = h->hlen[idx];
        if (x > 14) { ext |= lbx; x_bits += lin_bits; }
        if (x != 0) { ext <<= 1; ext |= sign_x; x_bits += 1; }
        if (y > 14) { ext <<= lin_bits; ext |= lby; x_bits += lin_bits; }
        if (y != 0) { ext <<= 1; ext |= sign_y; x_bits += 1; }
        put_bits(e, code, c_bits);
        put_bits(e, ext, x_bits);
    } else {
        int idx = x * h->ylen + y;
        uint32_t code = h->hcod[idx]; int c_bits = h->hlen[idx];
        if (x != 0) { code <<= 1; code |= sign_x; c_bits += 1; }
        if (y != 0) { code <<= 1; code |= sign_y; c_bits += 1; }
        put_bits(e, code, c_bits);
    }
}

/* MP3_Encoder.py:1515-1547 (E13: p = v + 2w + 4x + 8y) */
static void huffman_coder_count1(OrcEncoder *e, const OrcHuff *h, int v, int w, int x, int y)
{
    uint32_t code = 0; int cbits = 0;
    int sv = v > 0 ? 0 : 1; if (v <= 0) v = -v;
    int sw = w > 0 ? 0 : 1; if (w <= 0) w = -w;
    int sx = x > 0 ? 0 : 1; if (x <= 0) x = -x;
    int sy = y > 0 ? 0 : 1; if (y <= 0) y = -y;
    int p = v + (w << 1) + (x << 2) + (y << 3);
    put_bits(e, h->hcod[p], h->hlen[p]);
    if (v) { code = sv; cbits = 1; }
    if (w) { code = (code << 1) | sw; cbits += 1; }
    if (x) { code = (code << 1) | sx; cbits += 1; }
    if (y) { code = (code << 1) | sy; cbits += 1; }
    put_bits(e, code, cbits);
}

/* MP3_Encoder.py:1394-1446 */
static void huffman_code_bits(OrcEncoder *e, int gr, int ch)
{
    const OrcTables *T = orc_tables();
    const int *sf = T->sfb_long[e->samplerate_index];
    const OrcGrInfo *c = &e->gi[gr][ch];
    const int32_t *ix = e->l3_enc[ch][gr];
    long bits = get_bits_count(e);
    int big_values = c->big_values << 1;
    int sfi = c->region0_count + 1;
    int region1_start = sf[sfi];
    sfi += c->region1_count + 1;
    int region2_start = sf[sfi];
    for (int i = 0; i < big_values; i += 2) {
        int idx = (i >= region1_start) + (i >= region2_start);
        int ti = c->table_select[idx];
        if (ti != 0) huffman_code(e, ti, ix[i], ix[i + 1]);
    }
    const OrcHuff *h = &T->huff[c->count1table_select + 32];
    int count1_end = big_values + (c->count1 << 2);
    for (int i = big_values; i < count1_end; i += 4) huffman_coder_count1(e, h, ix[i], ix[i + 1], ix[i + 2], ix[i + 3]);
    bits = get_bits_count(e) - bits;
    bits = c->part2_3_length - c->part2_length - bits;
    if (bits > 0) {
        long words = bits / 32, rem = bits % 32;
        while (words) { put_bits(e, 0xffffffffu, 32); words--; }
        if (rem) put_bits(e, (uint32_t)((1ull << rem) - 1), (int)rem);
    } else if (bits < 0) e->error = ORC_ERR_MALFORMED; /* reference would loop forever */
}

/* MP3_Encoder.py:1266-1360 */
static void format_bitstream(OrcEncoder *e)
{
    for (int ch = 0; ch < e->nch; ch++)
        for (int gr = 0; gr < 2; gr++)
            for (int i = 0; i < 576; i++)
                if (e->mdct_freq[ch][gr][i] < 0 && e->l3_enc[ch][gr][i] > 0) e->l3_enc[ch][gr][i] *= -1;
    /* :1281-1337 */
    put_bits(e, 0x7ff, 11); put_bits(e, e->version, 2); put_bits(e, 1 /* mpeg.layer */, 2); put_bits(e, 1, 1);
    put_bits(e, e->bitrate_index, 4); put_bits(e, e->samplerate_index % 3, 2); put_bits(e, e->padding, 1);
    put_bits(e, 0, 1); put_bits(e, e->mode, 2); put_bits(e, 0, 2); put_bits(e, 0, 1); put_bits(e, 1, 1);
    put_bits(e, 0, 2);
    put_bits(e, 0, 9);
    put_bits(e, 0, e->nch == 2 ? 3 : 5);
    for (int ch = 0; ch < e->nch; ch++)
        for (int b = 0; b < 4; b++) put_bits(e, e->scfsi[ch][b], 1);
    for (int gr = 0; gr < 2; gr++)
        for (int ch = 0; ch < e->nch; ch++) {
            const OrcGrInfo *c = &e->gi[gr][ch];
            put_bits(e, c->part2_3_length, 12); put_bits(e, c->big_values, 9); put_bits(e, c->global_gain, 8);
            put_bits(e, c->scale_fac_compress, 4); put_bits(e, 0, 1);
            for (int r = 0; r < 3; r++) put_bits(e, c->table_select[r], 5);
            put_bits(e, c->region0_count, 4); put_bits(e, c->region1_count, 3);
            put_bits(e, c->preflag, 1); put_bits(e, c->scale_fac_scale, 1); put_bits(e, c->count1table_select, 1);
        }
    /* :1339-1360: slen1 = slen2 = 0 (scale_fac_compress == 0) -> scalefactors take 0 bits */
    for (int gr = 0; gr < 2; gr++)
        for (int ch = 0; ch < e->nch; ch++) {
            int n0 = 0;
            if (gr == 0 || e->scfsi[ch][0] == 0) n0 += 6;
            if (gr == 0 || e->scfsi[ch][1] == 0) n0 += 5;
            if (gr == 0 || e->scfsi[ch][2] == 0) n0 += 5;
            if (gr == 0 || e->scfsi[ch][3] == 0) n0 += 5;
            for (int k = 0; k < n0; k++) put_bits(e, 0, 0);
            huffman_code_bits(e, gr, ch);
        }
}

/* MP3_Encoder.py:623-650 */
static void encode_buffer_internal(OrcEncoder *e)
{
    if (e->frac_slots_per_frame) {
        e->padding = (e->slot_lag <= (e->frac_slots_per_frame - 1.0)) ? 1 : 0;
        e->slot_lag += e->padding - e->frac_slots_per_frame;
    }
    e->bits_per_frame = 8 * (e->whole_slots_per_frame + e->padding);
    e->mean_bits = (int)((double)(e->bits_per_frame - e->side_info_len) / 2);
    mdct_sub(e); if (e->error) return;
    /* record */
    if (e->n_frames + 1 > e->cap_frames) {
        e->cap_frames = e->cap_frames ? e->cap_frames * 2 : 64;
        e->frames = (OrcEncFrame *)realloc(e->frames, e->cap_frames * sizeof(OrcEncFrame));
        e->rec_mdct = (int32_t *)realloc(e->rec_mdct, e->cap_frames * 2304 * sizeof(int32_t));
        e->rec_ix = (int32_t *)realloc(e->rec_ix, e->cap_frames * 2304 * sizeof(int32_t));
    }
    memcpy(e->rec_mdct + e->n_frames * 2304, e->mdct_freq, 2304 * sizeof(int32_t));
    iteration_loop(e); if (e->error) return;
    format_bitstream(e); if (e->error) return;
    long written = e->data_position;
    e->data_position = 0;
    if (e->out_len + written > e->out_cap) {
        e->out_cap = (e->out_len + written) * 2 + 4096;
        e->out = (uint8_t *)realloc(e->out, e->out_cap);
    }
    memcpy(e->out + e->out_len, e->data, written);
    e->out_len += written;
    OrcEncFrame *f = &e->frames[e->n_frames];
    memcpy(f->gi, e->gi, sizeof f->gi);
    memcpy(f->scfsi, e->scfsi, sizeof f->scfsi);
    f->written = (int32_t)written; f->hide_off = (int32_t)e->hide_off; f->padding = e->padding;
    memcpy(e->rec_ix + e->n_frames * 2304, e->l3_enc, 2304 * sizeof(int32_t));
    e->n_frames++;
}

/* MP3_Encoder.py:596-618 encode(); __flush returns nothing new (E14) */
int orc_enc_run(OrcEncoder *e, const int16_t *pcm, long n_samples_per_ch)
{
    if (e->error) return e->error;
    if (e->nch != 2) return ORC_ERR_UNSUPPORTED; /* mono: IndexError in the reference (E3) */
    e->buf = pcm; e->buf_len = n_samples_per_ch * e->nch;
    e->buf_pos[0] = 0; e->buf_pos[1] = 1;
    long samples_per_pass = 1152L * e->nch;
    long total = n_samples_per_ch * e->nch;
    long count = total / samples_per_pass;
    for (long i = 0; i < count; i++) { encode_buffer_internal(e); if (e->error) return e->error; }
    if (total % samples_per_pass) return ORC_ERR_UNSUPPORTED; /* partial frame over-reads the buffer (E3) */
    return 0;
}

long orc_enc_n_frames(const OrcEncoder *e) { return e->n_frames; }
long orc_enc_out_len(const OrcEncoder *e) { return e->out_len; }
const uint8_t *orc_enc_out(const OrcEncoder *e) { return e->out; }
long orc_enc_hide_offset(const OrcEncoder *e) { return e->hide_off; }
const OrcEncFrame *orc_enc_frames(const OrcEncoder *e) { return e->frames; }
const int32_t *orc_enc_mdct_freq(const OrcEncoder *e) { return e->rec_mdct; }
const int32_t *orc_enc_ix(const OrcEncoder *e) { return e->rec_ix; }
