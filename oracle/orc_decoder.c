/*
 * ORACLE (test infrastructure, NOT product code).
 * Plain-C restatement of the reference decoder: mp3stego/decoder/{MP3_Parser,
 * Frame,FrameHeader,FrameSideInformation,util}.py.  Every function cites the
 * reference lines it follows; reference quirks
 * are reproduced on purpose.  Float math is fp64 in the reference's operation
 * order; build with -ffp-contract=off.
 */
#include "orc.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ bits */
/* decoder/util.py:22-64  get_bits: MSB-first, bytes past the end read as 0 */
static uint32_t get_bits(const uint8_t *buf, long len, long start_bit, int n)
{
    uint32_t r = 0;
    for (int k = 0; k < n; k++) {
        long b = start_bit + k;
        long byte = b >> 3;
        int v = (byte >= 0 && byte < len) ? ((buf[byte] >> (7 - (b & 7))) & 1) : 0;
        r = (r << 1) | (uint32_t)v;
    }
    return r;
}

/* Python list slice data[start:stop] appended to dst (negative indices wrap as in Python) */
static long py_slice_append(uint8_t *dst, long dpos, long dcap, const uint8_t *data, long n, long start, long stop)
{
    if (start < 0) { start += n; if (start < 0) start = 0; }
    if (stop < 0) { stop += n; if (stop < 0) stop = 0; }
    if (start > n) start = n;
    if (stop > n) stop = n;
    for (long i = start; i < stop && dpos < dcap; i++) dst[dpos++] = data[i];
    return dpos;
}

/* ------------------------------------------------------------------ state */
struct OrcDecoder {
    /* FrameHeader.py fields (persist across frames like the Python object) */
    double mpeg_version; int layer, crc, bit_rate, sampling_rate, padding, channel_mode, channels;
    int mode_ext0, mode_ext1; int sr_idx; /* sr_idx: 0=44.1k 1=48k 2=32k, -1 = no tables */
    /* Frame.py:226-242 */
    double prev_frame_size[9]; int frame_size;
    double prev_samples[2][32][18]; double fifo[2][1024];
    double samples[2][2][576];
    uint8_t *main_data; long main_len, main_cap;
    OrcFrameInfo si; /* FrameSideInformation arrays persist across frames (D10) */
    /* outputs */
    OrcFrameInfo *frames; int16_t *is; double *pcm; char *bits;
    long n_frames, n_pcm_rows, n_bits, cap_frames, cap_rows, cap_bits;
    int error;
};

OrcDecoder *orc_dec_new(void)
{
    OrcDecoder *d = (OrcDecoder *)calloc(1, sizeof *d);
    d->main_cap = 1 << 16; d->main_data = (uint8_t *)malloc(d->main_cap);
    d->sr_idx = -1;
    orc_tables();
    return d;
}
void orc_dec_free(OrcDecoder *d)
{
    if (!d) return;
    free(d->main_data); free(d->frames); free(d->is); free(d->pcm); free(d->bits); free(d);
}

/* FrameHeader.py:51-192 */
static void init_header(OrcDecoder *d, const uint8_t *b)
{
    int b1 = b[1], b2 = b[2], b3 = b[3];
    if ((b1 & 0x10) && (b1 & 0x08)) d->mpeg_version = 1;            /* :72-80 */
    else if ((b1 & 0x10) && !(b1 & 0x08)) d->mpeg_version = 2;
    else if (!(b1 & 0x10) && (b1 & 0x08)) d->mpeg_version = 0;
    else d->mpeg_version = 2.5;
    d->layer = 4 - (((b1 << 5) & 0xff) >> 6);                        /* :82-91 */
    d->crc = b1 & 1;                                                 /* :95 */
    /* :113-123 sampling rate; rows indexed floor(version)-1 (Python negative index for version 0) */
    static const int rates[3][3] = {{44100, 48000, 32000}, {22050, 24000, 16000}, {11025, 12000, 8000}};
    int row = (int)floor(d->mpeg_version) - 1; if (row < 0) row += 3;
    if (!(b2 & 0x08) && !(b2 & 0x04)) d->sampling_rate = rates[row][0];
    else if (!(b2 & 0x08) && (b2 & 0x04)) d->sampling_rate = rates[row][1];
    else if ((b2 & 0x08) && !(b2 & 0x04)) d->sampling_rate = rates[row][2];
    /* :125-143 tables only for 32/44.1/48 kHz, otherwise the previous ones stay */
    if (d->sampling_rate == 32000) d->sr_idx = 2;
    else if (d->sampling_rate == 44100) d->sr_idx = 0;
    else if (d->sampling_rate == 48000) d->sr_idx = 1;
    d->channel_mode = (b3 >> 6) & 3;                                 /* :145-154 */
    d->channels = d->channel_mode == 3 ? 1 : 2;
    if (d->layer == 3) { d->mode_ext0 = b3 & 0x20; d->mode_ext1 = b3 & 0x10; } /* :156-161 */
    d->padding = (b2 & 2) ? 1 : 0;                                   /* :163-165 */
    /* :167-192 bit rate */
    static const int r13[14] = {32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320};
    static const int r12[14] = {32, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320, 384};
    static const int r2x[14] = {8, 16, 24, 32, 40, 48, 56, 64, 80, 96, 112, 128, 144, 160};
    int bi = (b2 >> 4) - 1; if (bi < 0) bi += 14; /* Python rates[-1] */
    if (d->mpeg_version == 1) {
        if (d->layer == 1) d->bit_rate = b2 * 32;
        else if (d->layer == 2) { if (bi < 14) d->bit_rate = r12[bi] * 1000; else d->error = ORC_ERR_MALFORMED; }
        else if (d->layer == 3) { if (bi < 14) d->bit_rate = r13[bi] * 1000; else d->error = ORC_ERR_MALFORMED; }
    } else {
        if (d->layer == 1) { if (bi < 14) d->bit_rate = r13[bi] * 1000; else d->error = ORC_ERR_MALFORMED; }
        else if (d->layer < 4) { if (bi < 14) d->bit_rate = r2x[bi] * 1000; else d->error = ORC_ERR_MALFORMED; }
    }
}

/* Frame.py:288-316 */
static void set_frame_size(OrcDecoder *d)
{
    int spf = 0;
    if (d->layer == 3) spf = (d->mpeg_version == 1) ? 1152 : 576;
    else if (d->layer == 2) spf = 1152;
    else if (d->layer == 1) spf = 384;
    for (int i = 8; i > 0; i--) d->prev_frame_size[i] = d->prev_frame_size[i - 1];
    d->prev_frame_size[0] = d->frame_size;
    if (d->sampling_rate == 0) { d->error = ORC_ERR_MALFORMED; return; }
    d->frame_size = (int)((((double)spf / 8) * d->bit_rate) / d->sampling_rate);
    if (d->padding == 1) d->frame_size += 1;
}

/* FrameSideInformation.py:39-137 */
static void set_side_info(OrcDecoder *d, const uint8_t *buf, long len)
{
    OrcFrameInfo *s = &d->si;
    long off = 0;
    s->main_data_begin = (int)get_bits(buf, len, 0, 9); off += 9;
    off += d->channel_mode == 3 ? 5 : 3;
    for (int ch = 0; ch < d->channels; ch++)
        for (int b = 0; b < 4; b++) { s->scfsi[ch][b] = get_bits(buf, len, off, 1) != 0; off += 1; }
    for (int gr = 0; gr < 2; gr++)
        for (int ch = 0; ch < d->channels; ch++) {
            s->part2_3_length[gr][ch] = get_bits(buf, len, off, 12); off += 12;
            s->big_value[gr][ch] = get_bits(buf, len, off, 9); off += 9;
            s->global_gain[gr][ch] = get_bits(buf, len, off, 8); off += 8;
            s->scale_fac_compress[gr][ch] = get_bits(buf, len, off, 4); off += 4;
            s->window_switching[gr][ch] = get_bits(buf, len, off, 1) == 1; off += 1;
            if (s->window_switching[gr][ch]) {
                s->block_type[gr][ch] = get_bits(buf, len, off, 2); off += 2;
                s->mixed_block_flag[gr][ch] = get_bits(buf, len, off, 1) == 1; off += 1;
                s->region0_count[gr][ch] = s->block_type[gr][ch] == 2 ? 8 : 7;
                s->region1_count[gr][ch] = 20 - s->region0_count[gr][ch];
                for (int r = 0; r < 2; r++) { s->table_select[gr][ch][r] = get_bits(buf, len, off, 5); off += 5; }
                for (int w = 0; w < 3; w++) { s->sub_block_gain[gr][ch][w] = get_bits(buf, len, off, 3); off += 3; }
            } else {
                s->block_type[gr][ch] = 0; s->mixed_block_flag[gr][ch] = 0;
                for (int r = 0; r < 3; r++) { s->table_select[gr][ch][r] = get_bits(buf, len, off, 5); off += 5; }
                s->region0_count[gr][ch] = get_bits(buf, len, off, 4); off += 4;
                s->region1_count[gr][ch] = get_bits(buf, len, off, 3); off += 3;
            }
            s->pre_flag[gr][ch] = get_bits(buf, len, off, 1); off += 1;
            s->scale_fac_scale[gr][ch] = get_bits(buf, len, off, 1); off += 1;
            s->count1table_select[gr][ch] = get_bits(buf, len, off, 1); off += 1;
        }
}

/* Frame.py:365-441 */
static long unpack_scale_fac(OrcDecoder *d, int gr, int ch, long bit)
{
    const OrcTables *T = orc_tables();
    OrcFrameInfo *s = &d->si;
    const uint8_t *m = d->main_data; long ml = d->main_len;
    int sl0 = T->slen[s->scale_fac_compress[gr][ch]][0], sl1 = T->slen[s->scale_fac_compress[gr][ch]][1];
    if (s->block_type[gr][ch] == 2 && s->window_switching[gr][ch]) {
        if (s->mixed_block_flag[gr][ch] == 1) {
            for (int sfb = 0; sfb < 8; sfb++) { s->scale_fac_l[gr][ch][sfb] = get_bits(m, ml, bit, sl0); bit += sl0; }
            for (int sfb = 3; sfb < 6; sfb++)
                for (int w = 0; w < 3; w++) { s->scale_fac_s[gr][ch][w][sfb] = get_bits(m, ml, bit, sl0); bit += sl0; }
        } else {
            for (int sfb = 0; sfb < 6; sfb++)
                for (int w = 0; w < 3; w++) { s->scale_fac_s[gr][ch][w][sfb] = get_bits(m, ml, bit, sl0); bit += sl0; }
        }
        for (int sfb = 6; sfb < 12; sfb++)
            for (int w = 0; w < 3; w++) { s->scale_fac_s[gr][ch][w][sfb] = get_bits(m, ml, bit, sl1); bit += sl1; }
        for (int w = 0; w < 3; w++) s->scale_fac_s[gr][ch][w][12] = 0;
    } else {
        if (gr == 0) {
            for (int sfb = 0; sfb < 11; sfb++) { s->scale_fac_l[gr][ch][sfb] = get_bits(m, ml, bit, sl0); bit += sl0; }
            for (int sfb = 11; sfb < 21; sfb++) { s->scale_fac_l[gr][ch][sfb] = get_bits(m, ml, bit, sl1); bit += sl1; }
        } else {
            static const int SB[4] = {6, 11, 16, 21}, PSB[4] = {0, 6, 11, 16};
            for (int i = 0; i < 4; i++) {
                int sl = i < 2 ? sl0 : sl1;
                for (int sfb = PSB[i]; sfb < SB[i]; sfb++) {
                    if (s->scfsi[ch][i]) s->scale_fac_l[gr][ch][sfb] = s->scale_fac_l[0][ch][sfb];
                    else { s->scale_fac_l[gr][ch][sfb] = get_bits(m, ml, bit, sl); bit += sl; }
                }
            }
        }
        s->scale_fac_l[gr][ch][21] = 0;
    }
    return bit;
}

/* decoder big-value table entry as the reference stores it: (hcod << (32-len), len) in row-major (x,y) */
static int huff_match(const OrcHuff *h, int max, uint32_t bit_sample, int *x, int *y, int *size)
{
    for (int row = 0; row < max; row++)
        for (int col = 0; col < max; col++) {
            int idx = row * h->ylen + col;
            int sz = h->hlen[idx];
            uint32_t val = (uint32_t)h->hcod[idx] << (32 - sz);
            if ((val >> (32 - sz)) == (bit_sample >> (32 - sz))) { *x = row; *y = col; *size = sz; return 1; }
        }
    return 0;
}

/* Frame.py:443-559 */
static void unpack_samples(OrcDecoder *d, int gr, int ch, long bit, long max_bit)
{
    const OrcTables *T = orc_tables();
    OrcFrameInfo *s = &d->si;
    const uint8_t *m = d->main_data; long ml = d->main_len;
    double *smp = d->samples[gr][ch];
    for (int i = 0; i < 576; i++) smp[i] = 0;
    int region0, region1;
    if (s->window_switching[gr][ch] && s->block_type[gr][ch] == 2) { region0 = 36; region1 = 576; }
    else {
        int i0 = s->region0_count[gr][ch] + 1, i1 = i0 + s->region1_count[gr][ch] + 1;
        if (d->sr_idx < 0 || i0 > 22 || i1 > 22) { d->error = ORC_ERR_MALFORMED; return; } /* Python: IndexError */
        region0 = T->sfb_long[d->sr_idx][i0]; region1 = T->sfb_long[d->sr_idx][i1];
    }
    int sample = 0;
    while (sample < s->big_value[gr][ch] * 2) {
        if (sample + 1 >= 576) { d->error = ORC_ERR_MALFORMED; return; }           /* Python: IndexError */
        int tn = sample < region0 ? s->table_select[gr][ch][0]
               : sample < region1 ? s->table_select[gr][ch][1] : s->table_select[gr][ch][2];
        if (tn == 0) { smp[sample] = 0; sample += 2; continue; }
        uint32_t bs = get_bits(m, ml, bit, 32);
        int max = T->dec_max[tn];                 /* 0 for tables 4/14: nothing decoded, no bits used (D2) */
        int v[2], size;
        /* tables 4/14 map to hft_0 in the reference; max==0 skips the search entirely */
        if (max > 0 && huff_match(&T->huff[tn], max, bs, &v[0], &v[1], &size)) {
            bit += size;
            for (int i = 0; i < 2; i++) {
                int linbit = 0;
                if (T->dec_linbits[tn] != 0 && v[i] == max - 1) {
                    linbit = (int)get_bits(m, ml, bit, T->dec_linbits[tn]); bit += T->dec_linbits[tn];
                }
                int sign = 1;
                if (v[i] > 0) { sign = get_bits(m, ml, bit, 1) > 0 ? -1 : 1; bit += 1; }
                smp[sample + i] = (double)(sign * (v[i] + linbit));
            }
        }
        sample += 2;
    }
    /* quadruples :521-554 (D1: sample + 4 < 576, no overrun discard) */
    while (bit < max_bit && sample + 4 < 576) {
        int val[4] = {0, 0, 0, 0};
        if (s->count1table_select[gr][ch] == 1) {
            uint32_t bs = get_bits(m, ml, bit, 4); bit += 4;
            val[0] = (bs & 8) ? 0 : 1; val[1] = (bs & 4) ? 0 : 1; val[2] = (bs & 2) ? 0 : 1; val[3] = (bs & 1) ? 0 : 1;
        } else {
            uint32_t bs = get_bits(m, ml, bit, 32);
            const OrcHuff *q = &T->huff[32];
            for (int e = 0; e < 16; e++) {
                int sz = q->hlen[e];
                uint32_t cv = (uint32_t)q->hcod[e] << (32 - sz);
                if ((cv >> (32 - sz)) == (bs >> (32 - sz))) {
                    bit += sz;
                    val[0] = (e >> 3) & 1; val[1] = (e >> 2) & 1; val[2] = (e >> 1) & 1; val[3] = e & 1;
                    break;
                }
            }
        }
        for (int i = 0; i < 4; i++)
            if (val[i] > 0) { if (get_bits(m, ml, bit, 1) == 1) val[i] = -val[i]; bit += 1; }
        for (int i = 0; i < 4; i++) smp[sample + i] = val[i];
        sample += 4;
    }
    while (sample < 576) { smp[sample] = 0; sample++; }
}

/* Frame.py:318-363 */
static void set_main_data(OrcDecoder *d, const uint8_t *buffer, long buflen, const uint8_t *file, long flen, long curr_offset)
{
    OrcFrameInfo *s = &d->si;
    int constant = d->channel_mode == 3 ? 21 : 36;
    if (d->crc == 0) constant += 2;
    if (s->main_data_begin == 0) {
        d->main_len = py_slice_append(d->main_data, 0, d->main_cap, buffer, buflen, constant, d->frame_size);
    } else {
        double bound = 0;
        for (int frame = 0; frame < 9; frame++) {
            bound += d->prev_frame_size[frame] - constant;
            if (s->main_data_begin < bound) {
                double ptr_offset = s->main_data_begin + frame * constant;
                double part[9] = {0};
                part[frame] = s->main_data_begin;
                for (int i = 0; i < frame; i++) { part[i] = d->prev_frame_size[i] - constant; part[frame] -= part[i]; }
                long loc = (long)(curr_offset - ptr_offset);
                long pos = py_slice_append(d->main_data, 0, d->main_cap, file, flen, loc, loc + (long)part[frame]);
                ptr_offset -= (part[frame] + constant);
                for (int i = frame - 1; i >= 0; i--) {
                    loc = (long)(curr_offset - ptr_offset);
                    pos = py_slice_append(d->main_data, pos, d->main_cap, file, flen, loc, loc + (long)part[i]);
                    ptr_offset -= (part[i] + constant);
                }
                pos = py_slice_append(d->main_data, pos, d->main_cap, buffer, buflen, constant, d->frame_size);
                d->main_len = pos;
                break;
            }
        }
        /* no break: main_data keeps the previous frame's bytes, as in the reference */
    }
    long bit = 0;
    for (int gr = 0; gr < 2; gr++)
        for (int ch = 0; ch < d->channels; ch++) {
            long max_bit = bit + s->part2_3_length[gr][ch];
            bit = unpack_scale_fac(d, gr, ch, bit);
            unpack_samples(d, gr, ch, bit, max_bit);
            if (d->error) return;
            bit = max_bit;
        }
}

/* ------------------------------------------------------------------ transforms (the hot path) */

/* Frame.py:157-218 re_quantize -- literal restatement incl. the window/sfb tracker (D3, D4) */
void orc_requantize(double *smp, int global_gain, int scalefac_scale, int block_type, int mixed, int preflag,
                    const int32_t *sub_block_gain, const int32_t *scale_fac_l, const int32_t *scale_fac_s /*[3][13]*/,
                    int sr_idx)
{
    const OrcTables *T = orc_tables();
    const int *short_win = T->sfb_short_width[sr_idx];
    const int *long_win = T->sfb_long[sr_idx];
    double exp1, exp2;
    int window = 0, sfb = 0;
    double MULT = scalefac_scale == 0 ? 0.5 : 1;
    int sample = 0, i = 0;
    while (sample < 576) {
        if (block_type == 2 || (mixed && sfb >= 8)) {
            int swv = sfb < 12 ? short_win[sfb] : 0;
            if (i == swv) {
                i = 0;
                if (window == 2) { window = 0; sfb += 1; } else window += 1;
            }
            exp1 = global_gain - 210.0 - 8.0 * sub_block_gain[window];
            exp2 = MULT * scale_fac_s[window * 13 + sfb];
        } else {
            if (sample == long_win[sfb + 1]) sfb += 1;
            exp1 = global_gain - 210.0;
            int ptv = sfb < 21 ? T->pre_tab[sfb] : 0;
            exp2 = MULT * (scale_fac_l[sfb] + preflag * ptv);
        }
        double sign = smp[sample] < 0 ? -1.0 : 1.0;
        double a = pow(fabs(smp[sample]), 4.0 / 3.0);
        double b = pow(2.0, exp1 / 4.0);
        double c = pow(2.0, -exp2);
        smp[sample] = sign * a * b * c;
        sample += 1; i += 1;
    }
}

/* Frame.py:561-572 */
void orc_ms_stereo(double *l, double *r)
{
    const double SQRT2 = sqrt(2.0);
    for (int i = 0; i < 576; i++) {
        double mid = l[i], side = r[i];
        l[i] = (mid + side) / SQRT2;
        r[i] = (mid - side) / SQRT2;
    }
}

/* Frame.py:574-602 (D5: 12 sfbs only, the rest zeroed) */
void orc_reorder(double *smp, int sr_idx)
{
    const OrcTables *T = orc_tables();
    double out[576];
    memset(out, 0, sizeof out);
    int total = 0, start = 0, block = 0;
    for (int sb = 0; sb < 12; sb++) {
        int w = T->sfb_short_width[sr_idx][sb];
        for (int ss = 0; ss < w; ss++) {
            out[start + block + 0] = smp[total + ss + w * 0];
            out[start + block + 6] = smp[total + ss + w * 1];
            out[start + block + 12] = smp[total + ss + w * 2];
            if (block != 0 && block % 5 == 0) { start += 18; block = 0; } else block += 1;
        }
        total += w * 3;
    }
    memcpy(smp, out, sizeof out);
}

/* Frame.py:604-622 (only ever called with mixed == 0, D4) */
void orc_alias_reduction(double *smp)
{
    const OrcTables *T = orc_tables();
    for (int sb = 1; sb < 32; sb++)
        for (int i = 0; i < 8; i++) {
            int o1 = 18 * sb - i - 1, o2 = 18 * sb + i;
            double s1 = smp[o1], s2 = smp[o2];
            smp[o1] = s1 * T->alias_cs[i] - s2 * T->alias_ca[i];
            smp[o2] = s2 * T->alias_cs[i] + s1 * T->alias_ca[i];
        }
}

/* Frame.py:106-154 */
void orc_imdct(double *smp, int block_type, double *prev /*[32][18]*/)
{
    const OrcTables *T = orc_tables();
    double sample_block[36];
    memset(sample_block, 0, sizeof sample_block);
    int n = block_type == 2 ? 12 : 36, half_n = n / 2, sample = 0;
    for (int block = 0; block < 32; block++) {
        for (int win = 0; win < (block_type == 2 ? 3 : 1); win++)
            for (int i = 0; i < n; i++) {
                double xi = 0.0;
                for (int k = 0; k < half_n; k++) {
                    double s = smp[18 * block + half_n * win + k];
                    double c = n == 36 ? T->imdct_cos36[i][k] : T->imdct_cos12[i][k];
                    xi += s * c;
                }
                sample_block[win * n + i] = xi * T->sine_block[block_type][i];
            }
        if (block_type == 2) {
            double t[36];
            memcpy(t, sample_block, sizeof t);
            for (int i = 0; i < 6; i++) sample_block[i] = 0;
            for (int i = 6; i < 12; i++) sample_block[i] = t[i - 6];
            for (int i = 12; i < 18; i++) sample_block[i] = t[i - 6] + t[i];
            for (int i = 18; i < 24; i++) sample_block[i] = t[i] + t[i + 6];
            for (int i = 24; i < 30; i++) sample_block[i] = t[i + 6];
            for (int i = 30; i < 36; i++) sample_block[i] = 0;
        }
        for (int i = 0; i < 18; i++) {
            smp[sample + i] = sample_block[i] + prev[block * 18 + i];
            prev[block * 18 + i] = sample_block[18 + i];
        }
        sample += 18;
    }
}

/* Frame.py:624-631 */
void orc_frequency_inversion(double *smp)
{
    for (int sb = 1; sb < 18; sb += 2)
        for (int i = 1; i < 32; i += 2) smp[i * 18 + sb] *= -1;
}

/* Frame.py:65-103 */
void orc_synth_filter_bank(double *smp, double *fifo /*[1024]*/)
{
    const OrcTables *T = orc_tables();
    double s[32], u[512], w[512], pcm[576];
    for (int sb = 0; sb < 18; sb++) {
        for (int i = 0; i < 32; i++) s[i] = smp[i * 18 + sb];
        for (int i = 1023; i > 63; i--) fifo[i] = fifo[i - 64];
        for (int i = 0; i < 64; i++) {
            fifo[i] = 0.0;
            for (int j = 0; j < 32; j++) fifo[i] += s[j] * T->synth_matrix[i][j];
        }
        for (int i = 0; i < 8; i++)
            for (int j = 0; j < 32; j++) {
                u[i * 64 + j] = fifo[i * 128 + j];
                u[i * 64 + j + 32] = fifo[i * 128 + j + 96];
            }
        for (int i = 0; i < 512; i++) w[i] = u[i] * T->synth_window[i];
        for (int i = 0; i < 32; i++) {
            double sum = 0;
            for (int j = 0; j < 16; j++) sum += w[j * 32 + i];
            pcm[32 * sb + i] = sum;
        }
    }
    memcpy(smp, pcm, sizeof pcm);
}

/* numpy (x*32767).astype(int16) on x86-64: cvttsd2si to int32, keep the low 16 bits (D13) */
int16_t orc_pcm_to_i16(double v)
{
    double x = v * 32767;
    if (!(fabs(x) < 2147483648.0)) return 0; /* integer-indefinite 0x80000000 -> low half 0 */
    return (int16_t)(uint16_t)((uint32_t)(int32_t)x & 0xffffu);
}

/* ------------------------------------------------------------------ frame + stream loop */
static void ensure_out(OrcDecoder *d)
{
    if (d->n_frames + 1 > d->cap_frames) {
        d->cap_frames = d->cap_frames ? d->cap_frames * 2 : 64;
        d->frames = (OrcFrameInfo *)realloc(d->frames, d->cap_frames * sizeof(OrcFrameInfo));
        d->is = (int16_t *)realloc(d->is, d->cap_frames * 2304 * sizeof(int16_t));
    }
    if (d->n_pcm_rows + 1152 > d->cap_rows) {
        d->cap_rows = d->cap_rows ? d->cap_rows * 2 : 1152 * 64;
        d->pcm = (double *)realloc(d->pcm, d->cap_rows * 2 * sizeof(double));
    }
    if (d->n_bits + 16 > d->cap_bits) {
        d->cap_bits = d->cap_bits ? d->cap_bits * 2 : 4096;
        d->bits = (char *)realloc(d->bits, d->cap_bits);
    }
}

/* Frame.py:244-286 init_frame_params; pcm_out = [1152][channels] */
static void init_frame_params(OrcDecoder *d, const uint8_t *buffer, long buflen, const uint8_t *file, long flen,
                              long curr_offset, double *pcm_out)
{
    OrcFrameInfo *s = &d->si;
    set_frame_size(d);
    if (d->error) return;
    int start = d->crc == 0 ? 6 : 4;
    if (start > buflen) start = (int)buflen;
    set_side_info(d, buffer + start, buflen - start);
    /* :262,676-685 stego bits: ch -> gr -> region, skip 0, '0' if in H0 (decoder/util.py:67-81) */
    const OrcTables *T = orc_tables();
    for (int ch = 0; ch < d->channels; ch++)
        for (int gr = 0; gr < 2; gr++)
            for (int r = 0; r < 3; r++) {
                int t = s->table_select[gr][ch][r];
                if (t == 0) continue;
                d->bits[d->n_bits++] = T->in_h0[t] ? 0 : 1;
            }
    set_main_data(d, buffer, buflen, file, flen, curr_offset);
    if (d->error) return;
    /* record is + side info for the tests */
    OrcFrameInfo *fi = &d->frames[d->n_frames];
    *fi = *s;
    fi->hdr[0] = buffer[0]; fi->hdr[1] = buffer[1]; fi->hdr[2] = buffer[2]; fi->hdr[3] = buffer[3];
    fi->frame_size = d->frame_size; fi->nch = d->channels; fi->sr_idx = d->sr_idx;
    int16_t *isf = d->is + d->n_frames * 2304;
    memset(isf, 0, 2304 * sizeof(int16_t));
    for (int gr = 0; gr < 2; gr++)
        for (int ch = 0; ch < d->channels; ch++)
            for (int i = 0; i < 576; i++) isf[(gr * 2 + ch) * 576 + i] = (int16_t)d->samples[gr][ch][i];
    if (d->sr_idx < 0) { d->error = ORC_ERR_MALFORMED; return; }

    for (int gr = 0; gr < 2; gr++) {
        for (int ch = 0; ch < d->channels; ch++)
            orc_requantize(d->samples[gr][ch], s->global_gain[gr][ch], s->scale_fac_scale[gr][ch],
                           s->block_type[gr][ch], s->mixed_block_flag[gr][ch], s->pre_flag[gr][ch],
                           s->sub_block_gain[gr][ch], s->scale_fac_l[gr][ch], &s->scale_fac_s[gr][ch][0][0], d->sr_idx);
        if (d->channel_mode == 1 && d->mode_ext0) orc_ms_stereo(d->samples[gr][0], d->samples[gr][1]);
        for (int ch = 0; ch < d->channels; ch++) {
            if (s->block_type[gr][ch] == 2 || s->mixed_block_flag[gr][ch]) orc_reorder(d->samples[gr][ch], d->sr_idx);
            else orc_alias_reduction(d->samples[gr][ch]);
            orc_imdct(d->samples[gr][ch], s->block_type[gr][ch], &d->prev_samples[ch][0][0]);
            orc_frequency_inversion(d->samples[gr][ch]);
            orc_synth_filter_bank(d->samples[gr][ch], d->fifo[ch]);
        }
    }
    /* :633-640 interleave */
    for (int gr = 0; gr < 2; gr++)
        for (int i = 0; i < 576; i++)
            for (int ch = 0; ch < d->channels; ch++)
                pcm_out[(i + 576 * gr) * d->channels + ch] = d->samples[gr][ch][i];
}

/* MP3_Parser.py:25-85 (+ decoder.py:19-35 for the ID3 offset, passed in by the caller) */
int orc_dec_run(OrcDecoder *d, const uint8_t *file, long flen, long offset)
{
    d->n_frames = d->n_pcm_rows = d->n_bits = 0; d->error = 0;
    if (flen - offset < 2) return ORC_ERR_MALFORMED;
    const uint8_t *buffer = file + offset;
    long buflen = flen - offset;
    int valid = 0;
    if (buffer[0] == 0xFF && buffer[1] >= 0xE0) {
        valid = 1;
        if (buflen < 4) return ORC_ERR_MALFORMED;
        init_header(d, buffer);
        set_frame_size(d);            /* D11 */
    }
    double last_pcm[1152 * 2];
    int last_ch = 0; /* MP3_Parser.py:79 appends curr_frame.pcm even when the header was bad (D12) */
    int have_pcm = 0;
    while (valid && flen > offset + 4) {
        if (buffer[0] == 0xFF && buffer[1] >= 0xE0) init_header(d, buffer); else valid = 0;
        if (d->error) return d->error;
        ensure_out(d);
        if (valid) {
            init_frame_params(d, buffer, buflen, file, flen, offset, last_pcm);
            if (d->error) return d->error;
            last_ch = d->channels; have_pcm = 1;
            d->n_frames++;
            offset += d->frame_size;
            if (d->frame_size <= 0) return ORC_ERR_MALFORMED;
            buffer = file + (offset < flen ? offset : flen);
            buflen = offset < flen ? flen - offset : 0;
        }
        if (have_pcm) {
            /* rows are stored with the channel count of the frame they came from; a mid-stream channel
             * change makes np.array(pcm_data) ragged in the reference -- not supported, flagged */
            if (d->n_pcm_rows && last_ch != d->frames[0].nch) return ORC_ERR_MALFORMED;
            memcpy(d->pcm + d->n_pcm_rows * last_ch, last_pcm, sizeof(double) * 1152 * last_ch);
            d->n_pcm_rows += 1152;
        }
    }
    return 0;
}

long orc_dec_n_frames(const OrcDecoder *d) { return d->n_frames; }
long orc_dec_n_pcm_rows(const OrcDecoder *d) { return d->n_pcm_rows; }
long orc_dec_n_bits(const OrcDecoder *d) { return d->n_bits; }
int orc_dec_channels(const OrcDecoder *d) { return d->channels; }
int orc_dec_sampling_rate(const OrcDecoder *d) { return d->sampling_rate; }
int orc_dec_bit_rate(const OrcDecoder *d) { return d->bit_rate; }
const double *orc_dec_pcm(const OrcDecoder *d) { return d->pcm; }
const char *orc_dec_bits(const OrcDecoder *d) { return d->bits; }
const int16_t *orc_dec_is(const OrcDecoder *d) { return d->is; }
const OrcFrameInfo *orc_dec_frames(const OrcDecoder *d) { return d->frames; }
