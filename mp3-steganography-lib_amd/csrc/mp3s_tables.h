// Constant tables of the hot path, built once on the host (glibc libm -- the reference's own float
// tables come from CPython math.* = the same libm) and uploaded to the device's constant segment.
// Device cos/sin/pow are never used: they differ from glibc in the last ulp and the decoder must be
// int16-exact.
#pragma once
#include <stdint.h>

namespace mp3s {

constexpr int POW43_N = 8207;       // |is| <= 15 + 8191 (linbits 13)
constexpr int POW2Q_MIN = -266;     // exp1 = global_gain - 210 - 8*sub_block_gain  in [-266, 45]
constexpr int POW2Q_N = 312;
constexpr int POW2H_N = 40;         // 2*exp2 in [0, 36]
// First-level index width of the device Huffman tables.  9 bits: 15.4 KB + 4.4 KB of second-level tables.  With 10 bits
// (30.7 + 2.4 KB) the kernel alone is 8 % faster, but its workgroups no longer fit into the LDS the rate loop leaves free
// on a CU, and running under the rate loop is how the pipeline uses it (bench.py: 0.993 -> 0.967 ms per step).
constexpr int HUFF_FAST_BITS = 9;
constexpr int HUFF_L1_N = 1 << HUFF_FAST_BITS;
constexpr int HUFF_L2_N = 2240;     // second-level entries (tables start on even offsets)

// requantisation line map: one byte per spectral line, (is_short << 7) | (window << 5) | sfb
// case 0 = long path, 1 = block_type 2, 2 = mixed flag with block_type != 2 (reference Frame.py:185-208)
struct DevTables {
    // ---- decoder ----
    double synth_matrix[64][32];   // reference Frame.py:17-29
    double synth_window[512];      // reference decoder/tables.py:429-514
    double synth_window_t[32][16]; // [i][j] = synth_window[32 j + i]: the 16 taps of output i in one scalar load
    double imdct_cos36[36][18];    // reference Frame.py:130 (n = 36)
    double imdct_cos12[12][6];     // reference Frame.py:130 (n = 12)
    double sine_block[4][36];      // reference Frame.py:33-62
    double alias_cs[8], alias_ca[8];
    double pow43[POW43_N];         // pow(|is|, 4/3)           Frame.py:211
    double pow2q[POW2Q_N];         // pow(2, exp1/4)           Frame.py:212
    double pow2h[POW2H_N];         // pow(2, -(k*0.5))         Frame.py:213
    double sqrt2;
    // ---- fast synthesis for int16 output (k_dec_synth_fast): X[k] = sum_j S[j] cos((2j+1) k pi/64), k < 32, by splitting
    //      the sum into its even / odd halves level by level (341 multiplications instead of 2048).  Per level n = 32, 16,
    //      8, 4, 2 the factors of the odd outputs, cos((2j+1)(2m+1) pi/(2n)), rows m, columns j: 256 + 64 + 16 + 4 + 1
    double synth_fast[344];
    // the guard of that path: a sample whose value * 32767 lies within  synth_eps_a * (largest sum |S| of a time slot in
    // the tile) + synth_eps_x * |value * 32767|  of an integer is computed again in the reference's order
    double synth_eps_a, synth_eps_x;
    // ... plus synth_eps_g * (largest G of a granule the tile reads): the fast IMDCT (mirrored sums, fused multiply-adds)
    // leaves rows that differ from the reference's by at most imdct_kappa * (sum of |IMDCT input| of the subband in this
    // and the previous granule); G is that sum over the 32 subbands of a granule and channel
    double synth_eps_g, imdct_kappa;
    uint8_t rq_map[3][3][32][20];  // [sr][case][subband][18 lines + 2 pad]: five aligned dwords per lane;
                                   // byte = gain selector << 6 | scalefactor slot (see build_rq_map)
    int16_t reorder_src[3][576];   // [sr][dst line] -> src line or -1 (zero)  Frame.py:574-602
    uint8_t pre_tab[24];
    // ---- encoder ----
    int32_t enwindow[512];         // reference encoder/tables.py:34
    int32_t fl[32][64];            // reference MP3_Encoder.py:536-544
    int32_t cos_l[18][36];         // reference MP3_Encoder.py:551-556
    int32_t mdct_cs[8], mdct_ca[8];
    double steptab[128];
    int32_t steptabi[128];
    alignas(16) uint16_t int2idx[10000];   // values <= 1000 (k_rate_loop stages it 16 bytes at a time)
    int32_t sfb_long[3][23];
    // int32(log(e * 4.768371584e-7) / 0.69314718) of MP3_Encoder.py:841,855 as a step function of the integer e, tabulated
    // per octave with the host's libm: value at e = 2^k, and the e inside [2^k, 2^(k+1)) from which on it is one more
    int32_t en_base[32], en_step[32];
    int32_t subdv[23][2];
    uint32_t subdiv_lut[3][289];   // [sr][big_values] -> r0c | r1c<<4 | address1<<8 | address2<<18 (MP3_Encoder.py:998-1036)
    uint8_t hlen13[256], hlen15[256], hlen16[256], hlen24[256];
    uint8_t hlen_c1a[16];
    uint8_t linbits[32];
    int32_t linmax[32];
    uint8_t transform[32][2];      // reference MP3_Encoder.py:419-449
    // ---- Huffman decode on the device (k_dec_huffman): first-level table (HUFF_FAST_BITS) + second-level tables for longer codes
    uint8_t huff_lut_id[32];       // table_select -> 0..14, 255 = no code book (tables 0, 4, 14)
    uint8_t dec_max[32];           // symbols per axis (reference decoder/tables.py:426)
    uint16_t huff_fast[15][HUFF_L1_N];  // leaf: (len << 8) | (x << 4) | y;  0x8000 | (k << 11) | off / 2: the next k bits
                                        // index huff_l2[off ..];  0: no code
    uint16_t huff_l2[HUFF_L2_N];   // leaf: (total len << 8) | (x << 4) | y;  0: no code
    uint16_t quad_fast[64];        // count1 book A on 6 bits: (len << 4) | value
    // ---- Huffman code words for the device bit packer (k_enc_pack): books 13, 15, 16.., 24.. and count1 A
    uint32_t hcod[4][256];
    uint8_t hcod_c1a[16];
};

struct HostHuff {
    int xlen, ylen, linbits, linmax;
    const uint16_t *hcod;
    const uint8_t *hlen;
};

struct HostTables {
    DevTables dev;
    HostHuff huff[34];
    int sfb_short_width[3][12];
    int slen[16][2];
    uint8_t in_h0[32];
};

const HostTables &host_tables();   // built on first use, thread-safe

}  // namespace mp3s
