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
// Device Huffman decode tables (k_dec_huffman).  Every book has a first-level table indexed by the next w bits of the
// stream, w = min(longest code of the book, HUF_W_MAX): 8 + 64 + 64 + 256 + 128 + 512 entries for books 1, 2, 3, 5, 6, 9 and
// 2^HUF_W_MAX for the other nine, one zero entry in front for the books without code words (0, 4, 14); codes longer than w
// go through a second-level table per first-level prefix.  10 bits: 20.5 KB + 1.6 KB; a look-up is the one thing a decoding
// lane waits for, and a second level doubles it: 6-9 % of the symbols of books 13 / 15 / 16 / 24 with 9 bits, 2-4 % with 10.
constexpr int HUF_W_MAX = 10;
constexpr int HUF_L1_N = 1 + 8 + 64 + 64 + 256 + 128 + 512 + 9 * (1 << HUF_W_MAX) + 7;   // (multiple of 8)
constexpr int HUF_L2_N = 1200;      // second-level entries (tables start on even offsets)
constexpr int HUF_C1_N = 2560;      // count1 book A on 10 bits, book B on 8: code word + sign bits, two entries per index
constexpr int HUF_TAB_N = HUF_L1_N + HUF_L2_N + HUF_C1_N;
static_assert(HUF_L1_N % 8 == 0 && HUF_L2_N % 8 == 0, "the kernel copies the tables 16 bytes at a time");

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
    // the window taps as k_dec_synth_fast multiplies them: outputs i and 32 - i read the same V values (V[32 - i] = -V[i],
    // V[64 - i] = V[32 + i]), so one interval of the kernel sums both from one set of LDS reads and the signs move into the
    // taps: row i < 16 = synth_window_t[i]; rows 17..31 with their even taps negated; row 0 with its odd taps negated (both
    // halves of its V pair are +-X[16]); row 16 with its even taps zero (V[16] = X[32] = 0)
    double synth_window_f[32][16];
    // ... and times 32767 for the int16 path (each tap rounded once: one more rounding per term in the guard's bound), so
    // that a window sum IS the sample's scaled value; synth_xbound * (largest sum |S| of a slot in the tile) bounds every
    // |sample * 32767| of the tile (32767 * Dsum and a little)
    double synth_window_fs[32][16];
    double synth_xbound, synth_reserved;   // (two doubles: what follows keeps its offset modulo 16, and the 16-byte aligned tables their padding)
    // k_dec_synth_fast's constants in the order it uses them, sixteen doubles (one scalar request, two cache lines) per step:
    // per interval t = 0..7  [row of X[17+2t]] [row of X[15-2t]] [rows of X[16+2t], X[16-2t]: 8 + 8, or 4 + 4, or 2 + 2 doubles
    // at 0 and 8, nothing for t = 0] then the taps (synth_window_f; [1]: synth_window_fs) of outputs a = 2t, b = 32 - 2t,
    // c = 2t + 1, d = 31 - 2t as [a 0..7 | b 0..7] [a 8..15 | b 8..15] [c 0..7 | d 0..7] [c 8..15 | d 8..15]
    // (t = 0: outputs 0 and 16 read different slots: [row 0] [row 16] in their places)
    double synth_stream[2][8][112];
    // k_dec_stream (k_decode_stream.hpp), lane = (channel, subband sb):
    //   stream_cx[sb][j]: the lane's 16 cosines as the holder of X[k].  sb < 16: k = 2 sb + 1, the j-th value of its DPP row is the
    //   difference S[j] - S[31-j]: cos((2j+1) k pi/64); sb >= 16: k = 2 (sb - 16), the j-th value of its row is the sum
    //   S[15-j] + S[16+j]: cos((31-2j) k pi/64)
    //   stream_taps[v][i][jj]: the 16 taps of output i = sb, D[i + 32 jj] times the sign of the V value it multiplies -- even jj: V[i]
    //   = +X[16+i] (i <= 15), 0 (i = 16), -X[48-i]; odd jj: V[32+i] = -X[16-i] (i <= 15), -X[i-16] -- and, v = 1, times 32767
    double stream_cx[32][16];
    double stream_taps[2][32][16];
    // the stream kernel's long-block IMDCT as a DCT-IV of 18 points in two halves (mp3s_tables.cpp, k_decode_stream.hpp):
    //   imdct_rot[2k], [2k+1] = cos, sin((2k+1) pi/72): the rotation of the pair (v[k], v[17-k]) into (p[k], q[k]);
    //   imdct_pq[m][k] = cos(m (2k+1) pi/18), imdct_pq[m][9+k] = sin(m (2k+1) pi/18), m = 0..9, k = 0..8: stage m's two rows
    double imdct_rot[1][18];
    double imdct_pq[10][18];
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
    // ---- Huffman decode on the device (k_dec_huffman)
    uint8_t dec_max[32];           // symbols per axis (reference decoder/tables.py:426)
    // table_select -> byte offset of the book's first-level table << 16 | (32 - w) % 32 << 8 | linbits << 4 | w
    uint32_t huf_tinfo[32];
    // [first level, all books | second level | count1 A, count1 B], copied to LDS in one piece.
    // first / second level leaf: esc << 14 | c << 12 | adv << 8 | x << 4 | y -- adv = code length + the sign bits that follow
    //   it, (x != 0) + (y != 0) (second level: minus the book's w); the two bits at offset 30 + c - adv of the 32-bit window
    //   are x's sign above y's (a bit that is no sign lands on a zero); esc = x or y is 15 in a book with linbits;
    //   first-level entry of a longer prefix: 0x8000 | k << 11 | off / 2: the next k bits index second level [off ..]
    // count1: [A (v, w) | A (x, y) | B (v, w) | B (x, y)], 1024 + 1024 + 256 + 256 entries of the same leaf format
    alignas(16) uint16_t huf_tab[HUF_TAB_N];
    // ---- Huffman code words for the device bit packer (k_enc_pack): books 13, 15, 16.., 24.. and count1 A
    uint32_t hcod[4][256];
    uint8_t hcod_c1a[16];
    // ---- k_rate_loop's pair words as it keeps them in LDS (RlTables::hl, ::c1w in k_rate.hpp), built here once instead of by
    //      every workgroup from the four code-length tables
    uint32_t rl_hl[256][2];     // (lands on a 16-byte boundary: huf_tab is aligned, what lies between is 4 096 + 16 bytes)
    uint32_t rl_c1w[16];
    // ---- k_rate_loop's look at a probe of the binary search WITHOUT quantising it (rl_precheck): per quantiser step (index step + 127)
    //      the smallest |xr| whose quantised value is >= 1, >= 2, and -- through quantize's float branch -- > 8192 (0xffffffff: none).
    //      quantize is monotone in |xr|: ix >= v  <=>  |xr| >= threshold (MP3_Encoder.py:396-409)
    uint32_t rl_t1[128], rl_t2[128], rl_t8[128];
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
    bool analysis_plan_ok;   // the filter table built here repeats itself exactly where k_enc_analysis was compiled to share products (analysis_plan.h)
};

const HostTables &host_tables();   // built on first use, thread-safe

}  // namespace mp3s
