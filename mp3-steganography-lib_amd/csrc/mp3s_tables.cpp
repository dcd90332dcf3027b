// Host-side construction of the constant tables (see mp3s_tables.h).  Every float table is evaluated
// with the same libm call and operand order as the reference line cited next to it, so the device
// sees bit-identical constants; tests/test_tables.py checks them against the reference's dump.
#include "mp3s_tables.h"
#include "iso_tables.h"
#include "analysis_plan.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>

namespace mp3s {

namespace {

// Python round(x, nd) (correctly rounded decimal, then nearest double)
double round_decimals(double x, int nd)
{
    char buf[64];
    std::snprintf(buf, sizeof buf, "%.*f", nd, x);
    return std::strtod(buf, nullptr);
}

const int kSfbLong[3][23] = {
    {0, 4, 8, 12, 16, 20, 24, 30, 36, 44, 52, 62, 74, 90, 110, 134, 162, 196, 238, 288, 342, 418, 576},   // 44.1k
    {0, 4, 8, 12, 16, 20, 24, 30, 36, 42, 50, 60, 72, 88, 106, 128, 156, 190, 230, 276, 330, 384, 576},   // 48k
    {0, 4, 8, 12, 16, 20, 24, 30, 36, 44, 54, 66, 82, 102, 126, 156, 194, 240, 296, 364, 448, 550, 576}}; // 32k
const int kSfbShortW[3][12] = {{4, 4, 4, 4, 6, 8, 10, 12, 14, 18, 22, 30},
                               {4, 4, 4, 4, 6, 6, 10, 12, 14, 16, 20, 26},
                               {4, 4, 4, 4, 6, 8, 12, 16, 20, 26, 34, 42}};
const int kPreTab[21] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 3, 2};
const int kSlen[16][2] = {{0, 0}, {0, 1}, {0, 2}, {0, 3}, {3, 0}, {1, 1}, {1, 2}, {1, 3},
                          {2, 1}, {2, 2}, {2, 3}, {3, 1}, {3, 2}, {3, 3}, {4, 2}, {4, 3}};
const int kSubdv[23][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 1}, {1, 1}, {1, 1}, {1, 2}, {2, 2}, {2, 3}, {2, 3},
                           {3, 4}, {3, 4}, {3, 4}, {4, 5}, {4, 5}, {4, 6}, {5, 6}, {5, 6}, {5, 7}, {6, 7}, {6, 7}};
const double kAliasC[8] = {-0.6, -0.535, -0.33, -0.185, -0.095, -0.041, -0.0142, -0.0037};
// table-swap steganography: {table -> table for bit 0, table for bit 1}; bit 0 lands in H0
const uint8_t kTransform[32][2] = {
    {0, 0},   {3, 1},   {3, 2},   {3, 2},   {0, 0},   {6, 5},   {6, 5},   {8, 7},   {8, 7},   {8, 9},   {11, 10},
    {11, 10}, {12, 10}, {15, 13}, {0, 0},   {15, 13}, {17, 16}, {17, 18}, {19, 18}, {19, 20}, {21, 20}, {21, 22},
    {23, 22}, {23, 31}, {24, 25}, {26, 25}, {26, 27}, {28, 27}, {28, 29}, {30, 29}, {30, 31}, {23, 31}};
const int kH0[14] = {3, 6, 8, 11, 12, 15, 17, 19, 21, 23, 24, 26, 28, 30};

HostTables g_tables;
std::once_flag g_once;

void set_huff(int n, int xl, int yl, int lb, int lm, const uint16_t *c, const uint8_t *l)
{
    g_tables.huff[n] = HostHuff{xl, yl, lb, lm, c, l};
}

// literal simulation of the sfb/window tracker of reference Frame.py:179-218 for one (sr, case)
void build_rq_map(int sr, int cse, uint8_t *map)
{
    const int *short_win = kSfbShortW[sr];
    const int *long_win = kSfbLong[sr];
    const bool bt2 = cse == 1, mixed = cse == 2;
    int window = 0, sfb = 0, i = 0;
    for (int sample = 0; sample < 576; sample++, i++) {
        bool sh;
        if (bt2 || (mixed && sfb >= 8)) {
            int swv = sfb < 12 ? short_win[sfb] : 0;
            if (i == swv) {
                i = 0;
                if (window == 2) { window = 0; sfb += 1; } else window += 1;
            }
            sh = true;
        } else {
            if (sample == long_win[sfb + 1]) sfb += 1;
            sh = false;
        }
        // byte = gain selector << 6 | scalefactor slot: the requantiser reads exp1 per selector (0 = global gain, 1 + window =
        // with that window's sub_block_gain) and exp2 per slot (0..21 = long sfb, 22 + 13 window + sfb = short) from two
        // small per-granule tables; the clamps are those of the reference's array reads that can happen (sfb past the table
        // end reads the last entry here, whose pre_tab is 0)
        const int slot = sh ? 22 + 13 * window + (sfb < 13 ? sfb : 12) : (sfb < 22 ? sfb : 21);
        map[sample] = (uint8_t)(((sh ? 1 + window : 0) << 6) | slot);
    }
}

void build()
{
    HostTables &H = g_tables;
    DevTables &T = H.dev;
    std::memset(&H, 0, sizeof H);
    const double kPi = 3.141592653589793;  // math.pi

    for (int i = 0; i < 512; i++) {
        T.synth_window[i] = round_decimals((double)ISO_WINDOW_NUM[i] / 65536.0, 9);
        T.synth_window_t[i & 31][i >> 5] = T.synth_window[i];
        T.enwindow[i] = (int32_t)(round_decimals((double)ISO_WINDOW_NUM[i] / 2097152.0, 6) * 2147483647.0);
    }
    for (int i = 0; i < 64; i++)
        for (int j = 0; j < 32; j++) T.synth_matrix[i][j] = std::cos((16.0 + i) * (2.0 * j + 1.0) * (kPi / 64.0));
    for (int i = 0; i < 36; i++)
        for (int k = 0; k < 18; k++)
            T.imdct_cos36[i][k] = std::cos(kPi / (double)(2 * 36) * (double)(2 * i + 1 + 18) * (double)(2 * k + 1));
    for (int i = 0; i < 12; i++)
        for (int k = 0; k < 6; k++)
            T.imdct_cos12[i][k] = std::cos(kPi / (double)(2 * 12) * (double)(2 * i + 1 + 6) * (double)(2 * k + 1));
    for (int i = 0; i < 36; i++) T.sine_block[0][i] = std::sin(kPi / 36.0 * (i + 0.5));
    for (int i = 0; i < 18; i++) T.sine_block[1][i] = std::sin(kPi / 36.0 * (i + 0.5));
    for (int i = 18; i < 24; i++) T.sine_block[1][i] = 1.0;
    for (int i = 24; i < 30; i++) T.sine_block[1][i] = std::sin(kPi / 12.0 * (i - 18.0 + 0.5));
    for (int i = 30; i < 36; i++) T.sine_block[1][i] = 1.0;   // not 0 as in ISO
    for (int i = 0; i < 12; i++) T.sine_block[2][i] = std::sin(kPi / 12.0 * (i + 0.5));
    for (int i = 6; i < 12; i++) T.sine_block[3][i] = std::sin(kPi / 12.0 * (i - 6.0 + 0.5));
    for (int i = 12; i < 18; i++) T.sine_block[3][i] = 1.0;
    for (int i = 18; i < 36; i++) T.sine_block[3][i] = std::sin(kPi / 36.0 * (i + 0.5));
    for (int i = 0; i < 8; i++) {
        const double c = kAliasC[i];
        T.alias_cs[i] = round_decimals(1.0 / std::sqrt(1.0 + c * c), 10);
        T.alias_ca[i] = round_decimals(c / std::sqrt(1.0 + c * c), 10);
        T.mdct_ca[i] = (int32_t)(c / std::sqrt(1.0 + (c * c)) * 2147483647.0);
        T.mdct_cs[i] = (int32_t)(1.0 / std::sqrt(1.0 + (c * c)) * 2147483647.0);
    }
    for (int i = 0; i < POW43_N; i++) T.pow43[i] = std::pow((double)i, 4.0 / 3.0);
    for (int i = 0; i < POW2Q_N; i++) T.pow2q[i] = std::pow(2.0, (double)(i + POW2Q_MIN) / 4.0);
    for (int i = 0; i < POW2H_N; i++) T.pow2h[i] = std::pow(2.0, -((double)i * 0.5));
    T.sqrt2 = std::sqrt(2.0);
    {
        // fast synthesis: factors of the odd outputs per level (true cosines, rounded once by libm)
        int o = 0;
        for (int n = 32; n >= 2; n >>= 1)
            for (int m = 0; m < n / 2; m++)
                for (int j = 0; j < n / 2; j++)
                    T.synth_fast[o++] = std::cos((double)((2 * j + 1) * (2 * m + 1)) * (kPi / (2.0 * n)));
        // The guard.  u = 2^-53.  With A = sum_j |S[j]| of a slot:
        //   reference:  |V_ref - V_true| <= (dN + g33) A,   dN = max |N[i][j] - cos((16+i)(2j+1) pi/64)| of the reference's
        //               own table (its arguments are rounded before the cosine is taken: measured below in long double),
        //               g33 = 33u/(1-33u): 32 products summed one after the other
        //   fast:       |V_fast - V_true| <= g24 A:  at most 5 additions in front of a product, a factor that is off by at
        //               most 2u, 16 products summed one after the other
        //   window:     both paths run the same 16-tap sum on their V; the results differ by at most
        //               Dsum (dV + 2 g18 (1 + dV)) Amax,  Dsum = max_i sum_taps |D|,  dV = dN + g33 + g24
        //               (g18, not g17: the fast path's taps carry the factor 32767 and are rounded once more for it)
        //   * 32767:    two roundings of one multiplication: 2u |x|  (the fast path has none since round 4: kept)
        // and twice that, against slips in the algebra above.
        const double u = 1.1102230246251565e-16;
        long double dN = 0;
        const long double pil = 3.14159265358979323846264338327950288L;
        for (int i = 0; i < 64; i++)
            for (int j = 0; j < 32; j++) {
                const long double c = cosl((long double)((16 + i) * (2 * j + 1)) * pil / 64.0L);
                const long double d = fabsl((long double)T.synth_matrix[i][j] - c);
                if (d > dN) dN = d;
            }
        double dsum = 0;
        for (int i = 0; i < 32; i++) {
            double a = 0;
            for (int k = 0; k < 16; k++) a += std::fabs(T.synth_window[32 * k + i]);
            if (a > dsum) dsum = a;
        }
        const double g33 = 33 * u / (1 - 33 * u), g24 = 24 * u / (1 - 24 * u), g18s = 18 * u / (1 - 18 * u);
        const double dV = (double)dN + 1e-19 + g33 + g24;
        T.synth_eps_a = 2.0 * 32767.0 * dsum * (dV + 2 * g18s * (1 + dV));
        // The fast IMDCT (imdct_run<true>).  Inputs v[k] of a subband are the same in both paths (requantisation .. alias
        // reduction are not touched); B = sum_k |v[k]|, |C| <= 1, |window| <= 1.
        //   reference:  X[i] = the 18 products summed one after the other:            |X[i] - sum v C[i]| <= g18 B
        //   fast:       Y[i], i = 0..8 and 18..26, fused multiply-adds:              |Y[i] - sum v C[i]| <= g18 B
        //               Y[17-i] = -Y[i], Y[53-i] = Y[i]; the reference's table is only symmetric up to dT =
        //               max |C[i][k] + C[17-i][k]|, |C[i][k] - C[53-i][k]| (arguments rounded before the cosine: measured
        //               below in long double against the table itself):               |Y[i'] - sum v C[i']| <= (g18 + dT) B
        //   so |Y - X| <= (2 g18 + dT) B for every row; the window factor adds one rounding on each side (2.1 u B), the
        //   overlap-add another on each side and the previous granule's share with its own B':
        //               |S_fast - S_ref| <= kappa (B + B'),  kappa = 2 g18 + dT + 4.2 u
        //   (short blocks keep the reference's order in both paths.)
        // Through the synthesis: the matrixing is a sum over the 32 subbands with factors <= 1, so V moves by at most
        // kappa * sum_subbands (B + B') <= 2 kappa Gmax (G = sum of B over the subbands of a granule and channel, Gmax
        // over the granules whose rows a window sum reads and the granule in front of them); the sum |S| of a slot, which
        // scales the other terms, moves by the same amount (second order, covered by the factor 2 below); the 16-tap
        // window sum multiplies by Dsum as above.  G itself is a float sum of 576 non-negative terms (relative error
        // < 600 u): nothing against the factor 2.
        long double dT = 0;
        for (int i = 0; i < 9; i++)
            for (int k = 0; k < 18; k++) {
                const long double a = fabsl((long double)T.imdct_cos36[i][k] + (long double)T.imdct_cos36[17 - i][k]);
                const long double b = fabsl((long double)T.imdct_cos36[18 + i][k] - (long double)T.imdct_cos36[35 - i][k]);
                if (a > dT) dT = a;
                if (b > dT) dT = b;
            }
        const double g18 = 18 * u / (1 - 18 * u);
        // The stream kernel's rows (k_dec_stream) are the same 18 independent values by another route: y[n] = sum_k v[k] cos((2n+1)(2k+1) pi/72),
        // a DCT-IV of 18 points, rows 0..8 = y[9..17], rows 18..26 = -y[8..0] (the other rows by the mirror signs, as above).  With
        // a = (2k+1) pi/72 and the pairs x = v[k], x' = v[17-k], k < 9:
        //     p[k] = x cos a + x' sin a,   q[k] = -x sin a + x' cos a                                   (9 rotations)
        //     P[m] = sum_k p[k] cos(m (2k+1) pi/18),   Q[m] = sum_k q[k] sin(m (2k+1) pi/18),  m = 0..9  (P[9] = Q[0] = 0)
        //     y[2m] = P[m] + Q[m],   y[2m-1] = P[m] - Q[m]
        // (the angle-addition formulas for (4m +- 1) a; checked against the direct sums in tests/test_tables.py): 236 instead of 324
        // multiply-adds per subband.  Its distance from the TRUE sums, factors rounded once (<= u each), a product and a fused
        // multiply-add per rotation, nine fused multiply-adds per sum, one addition:
        //     |dp| <= 3.01 u (|x| + |x'|),  sum_k (|p| + |q|) <= sqrt2 B (1 + 4u),
        //     |dP| <= (g9 + u) sum |p| + sum |dp|,  |dy| <= |dP| + |dQ| + u (sum |p| + sum |q|)  <=  ((g9 + 2u) sqrt2 + 6.02 u) B  <  22 u B
        // and the reference's X from the true sums: g18 B for its own additions + dC B, dC = max |C[i][k] - cos((2i+19)(2k+1) pi/72)| of
        // its table (arguments rounded before the cosine: measured in long double).  So |Y - X| <= (g18 + dC + 22u) B; window factor and
        // overlap-add as above.  kappa covers both routes (k_dec_imdct<true>, k_dec_fixup's callers keep the mirrored sums).
        long double dC = 0;
        for (int i = 0; i < 36; i++)
            for (int k = 0; k < 18; k++) {
                const long double c = cosl((long double)((2 * i + 19) * (2 * k + 1)) * pil / 72.0L);
                const long double d = fabsl((long double)T.imdct_cos36[i][k] - c);
                if (d > dC) dC = d;
            }
        const double kappa_mirror = 2 * g18 + (double)dT + 4.2 * u, kappa_dct4 = g18 + (double)dC + 1e-19 + 22 * u + 4.2 * u;
        T.imdct_kappa = kappa_mirror > kappa_dct4 ? kappa_mirror : kappa_dct4;
        for (int k = 0; k < 9; k++) {
            T.imdct_rot[0][2 * k] = (double)cosl((long double)(2 * k + 1) * pil / 72.0L);
            T.imdct_rot[0][2 * k + 1] = (double)sinl((long double)(2 * k + 1) * pil / 72.0L);
        }
        for (int m = 0; m < 10; m++)
            for (int k = 0; k < 9; k++) {
                // (the argument reduced mod 36 in integers first: m (2k+1) pi/18 has the period 36)
                const int r = (m * (2 * k + 1)) % 36;
                T.imdct_pq[m][k] = m == 9 ? 0.0 : (double)cosl((long double)r * pil / 18.0L);       // (cos((2k+1) pi/2) = 0)
                T.imdct_pq[m][9 + k] = m == 0 ? 0.0 : (double)sinl((long double)r * pil / 18.0L);
            }
        T.synth_eps_g = 2.0 * 32767.0 * dsum * 2.0002 * T.imdct_kappa;
        T.synth_eps_x = 2.0 * 2 * u;
        for (int i = 0; i < 32; i++)
            for (int k = 0; k < 16; k++) {
                double w = T.synth_window_t[i][k];
                if (i == 0 && (k & 1)) w = -w;
                if (i == 16 && !(k & 1)) w = 0.0;
                if (i > 16 && !(k & 1)) w = -w;
                T.synth_window_f[i][k] = w;
                T.synth_window_fs[i][k] = w * 32767.0;
            }
        T.synth_xbound = 32767.0 * dsum * (1.0 + 1e-6);
        for (int v = 0; v < 2; v++)
            for (int t = 0; t < 8; t++) {
                double *q = T.synth_stream[v][t];
                const double (*W)[16] = v ? T.synth_window_fs : T.synth_window_f;
                const double *C32 = T.synth_fast, *C16 = T.synth_fast + 256, *C8 = T.synth_fast + 320, *C4 = T.synth_fast + 336;
                const int ka1 = 17 + 2 * t, kb1 = 15 - 2 * t, ka0 = 16 + 2 * t, kb0 = 16 - 2 * t;
                for (int j = 0; j < 16; j++) { q[j] = C32[(ka1 >> 1) * 16 + j]; q[16 + j] = C32[(kb1 >> 1) * 16 + j]; }
                if (t & 1) for (int j = 0; j < 8; j++) { q[32 + j] = C16[((ka0 - 2) >> 2) * 8 + j]; q[40 + j] = C16[((kb0 - 2) >> 2) * 8 + j]; }
                else if (t & 2) for (int j = 0; j < 4; j++) { q[32 + j] = C8[((ka0 - 4) >> 3) * 4 + j]; q[40 + j] = C8[((kb0 - 4) >> 3) * 4 + j]; }
                else if (t) for (int j = 0; j < 2; j++) { q[32 + j] = C4[((ka0 - 8) >> 4) * 2 + j]; q[40 + j] = C4[((kb0 - 8) >> 4) * 2 + j]; }
                const int oa = 2 * t, ob = t ? 32 - 2 * t : 16, oc = 2 * t + 1, od = 31 - 2 * t;
                if (t) for (int j = 0; j < 8; j++) { q[48 + j] = W[oa][j]; q[56 + j] = W[ob][j]; q[64 + j] = W[oa][8 + j]; q[72 + j] = W[ob][8 + j]; }
                else for (int j = 0; j < 16; j++) { q[48 + j] = W[oa][j]; q[64 + j] = W[ob][j]; }
                for (int j = 0; j < 8; j++) { q[80 + j] = W[oc][j]; q[88 + j] = W[od][j]; q[96 + j] = W[oc][8 + j]; q[104 + j] = W[od][8 + j]; }
            }
    }
    for (int sb = 0; sb < 32; sb++)
        for (int j = 0; j < 16; j++) {
            const int k = sb < 16 ? 2 * sb + 1 : 2 * (sb - 16);
            // (true cosines rounded once: the multiple of pi/64 reduced modulo its period in integers, the cosine taken in long double)
            const int n = ((sb < 16 ? 2 * j + 1 : 31 - 2 * j) * k) % 128;
            T.stream_cx[sb][j] = (double)cosl((long double)n * 3.14159265358979323846264338327950288L / 64.0L);
            const double d = T.synth_window_t[sb][j];
            const double w = (j & 1) ? -d : (sb <= 15 ? d : (sb == 16 ? 0.0 : -d));
            T.stream_taps[0][sb][j] = w;
            T.stream_taps[1][sb][j] = w * 32767.0;
        }
    for (int sr = 0; sr < 3; sr++) {
        for (int c = 0; c < 3; c++) {
            uint8_t flat[576];
            build_rq_map(sr, c, flat);
            for (int sb = 0; sb < 32; sb++)
                for (int k = 0; k < 18; k++) T.rq_map[sr][c][sb][k] = flat[sb * 18 + k];
        }
        // reference Frame.py:581-602 as a gather map
        for (int i = 0; i < 576; i++) T.reorder_src[sr][i] = -1;
        int total = 0, start = 0, block = 0;
        for (int sb = 0; sb < 12; sb++) {
            const int w = kSfbShortW[sr][sb];
            for (int ss = 0; ss < w; ss++) {
                T.reorder_src[sr][start + block + 0] = (int16_t)(total + ss + w * 0);
                T.reorder_src[sr][start + block + 6] = (int16_t)(total + ss + w * 1);
                T.reorder_src[sr][start + block + 12] = (int16_t)(total + ss + w * 2);
                if (block != 0 && block % 5 == 0) { start += 18; block = 0; } else block += 1;
            }
            total += w * 3;
        }
        for (int i = 0; i < 23; i++) T.sfb_long[sr][i] = kSfbLong[sr][i];
        for (int i = 0; i < 12; i++) H.sfb_short_width[sr][i] = kSfbShortW[sr][i];
    }
    // __subdivide as a function of big_values (reference MP3_Encoder.py:1008-1036, flattened index walk included)
    for (int sr = 0; sr < 3; sr++)
        for (int bv = 1; bv <= 288; bv++) {
            const int *sfb = kSfbLong[sr];
            const int bvr = 2 * bv;
            int anz = 0;
            while (sfb[anz] < bvr) anz++;
            int tc = kSubdv[anz][0];
            while (tc > 0) { if (sfb[tc + 1] <= bvr) break; tc--; }
            const int r0c = tc, a1 = sfb[tc + 1], base = tc + 1;
            tc = kSubdv[anz][1];
            while (tc > 0) { if (sfb[base + tc + 1] <= bvr) break; tc--; }
            const int r1c = tc, a2 = sfb[base + tc + 1];
            T.subdiv_lut[sr][bv] = (uint32_t)r0c | ((uint32_t)r1c << 4) | ((uint32_t)a1 << 8) | ((uint32_t)a2 << 18);
        }
    for (int i = 0; i < 21; i++) T.pre_tab[i] = (uint8_t)kPreTab[i];
    std::memcpy(H.slen, kSlen, sizeof kSlen);
    std::memcpy(T.subdv, kSubdv, sizeof kSubdv);
    std::memcpy(T.transform, kTransform, sizeof kTransform);
    for (int t : kH0) H.in_h0[t] = 1;

    // encoder fixed-point tables (util.PI64 = 0.049087385212, PI36 = 0.087266462599717, PI = 3.14159265358979)
    for (int i = 0; i < 32; i++)
        for (int j = 0; j < 64; j++) {
            double f = 1e9 * std::cos((double)((2 * i + 1) * (16 - j)) * 0.049087385212), ip;
            if (f >= 0) std::modf(f + 0.5, &ip); else std::modf(f - 0.5, &ip);
            T.fl[i][j] = (int32_t)(ip * 2147483647.0 * 1e-9);
        }
    // k_enc_analysis computes a product once where two outputs of a pass hold the same coefficient (analysis_plan.h, generated from this
    // formula by tools/gen_analysis_plan.py): the table built HERE, with this machine's libm, must repeat itself in exactly those places
    H.analysis_plan_ok = true;
    for (int p = 0; p < 8; p++)
        for (int k = 0; k < 64; k++) {
            const int o[4] = {p, 15 - p, 16 + p, 31 - p};
            unsigned word = 0;
            for (int j = 0; j < 4; j++) {
                int rep = j;
                for (int q = j - 1; q >= 0; q--) if (T.fl[o[q]][k] == T.fl[o[j]][k]) rep = q;
                word |= (unsigned)rep << (2 * j);
                if (T.fl[o[j]][k] == 0) word |= 1u << (8 + j);
            }
            if (word != ANALYSIS_PLAN[p][k]) H.analysis_plan_ok = false;
        }
    for (int m = 0; m < 18; m++)
        for (int k = 0; k < 36; k++)
            T.cos_l[m][k] = (int32_t)(std::sin(0.087266462599717 * (k + 0.5)) *
                                      std::cos((3.14159265358979 / 72) * (double)(2 * k + 19) * (double)(2 * m + 1)) *
                                      2147483647.0);
    for (int i = 0; i < 128; i++) {
        T.steptab[i] = std::pow(2.0, (double)(127 - i) / 4);
        T.steptabi[i] = (T.steptab[i] * 2 > 2147483647.0) ? 0x7fffffff : (int32_t)(T.steptab[i] * 2 + 0.5);
    }
    for (int i = 0; i < 10000; i++)
        T.int2idx[i] = (uint16_t)(int32_t)(std::sqrt(std::sqrt((double)i) * (double)i) - 0.0946 + 0.5);
    {
        // thresholds of rl_precheck.  ln = mulr(|xr|, steptabi) = (|xr| * scalei + 2^31) >> 32 is monotone in |xr|, int2idx in ln, the float
        // branch's int(sqrt(sqrt(d) * d)), d = |xr| * steptab * 4.656612875e-10, in |xr|: "quantised value >= v" is "|xr| >= T".
        int l1 = 0, l2 = 0;                              // smallest ln with int2idx[ln] >= 1, >= 2
        while (l1 < 9999 && T.int2idx[l1] < 1) l1++;
        while (l2 < 9999 && T.int2idx[l2] < 2) l2++;
        auto at_least = [](int ln, uint32_t scalei) -> uint32_t {   // smallest a with (a * scalei + 2^31) >> 32 >= ln
            if (scalei == 0) return 0xffffffffu;
            const unsigned __int128 need = ((unsigned __int128)(uint32_t)ln << 32) - (1ull << 31);
            const unsigned __int128 a = (need + scalei - 1) / scalei;
            return a > 0x80000000ull ? 0xffffffffu : (uint32_t)a;     // (|xr| <= 2^31)
        };
        for (int i = 0; i < 128; i++) {
            const uint32_t sc = (uint32_t)T.steptabi[i];
            T.rl_t1[i] = at_least(l1, sc);
            T.rl_t2[i] = at_least(l2, sc);
            // float branch (values whose ln >= 10000): the kernel's own expression, correctly rounded square roots on both sides
            auto big = [&](uint64_t a) {
                const double dbl = (double)(uint32_t)a * T.steptab[i] * 4.656612875e-10;
                return (int32_t)std::sqrt(std::sqrt(dbl) * dbl) > 8192;
            };
            const uint64_t lo_ln = at_least(10000, sc);    // below it a value takes the table branch (<= 1000)
            uint64_t lo = lo_ln, hi = 0x80000000ull;
            if (lo_ln == 0xffffffffu || !big(hi)) T.rl_t8[i] = 0xffffffffu;
            else {
                while (lo < hi) { const uint64_t mid = (lo + hi) / 2; if (big(mid)) hi = mid; else lo = mid + 1; }
                T.rl_t8[i] = (uint32_t)lo;
            }
        }
    }

    set_huff(0, 0, 0, 0, 0, nullptr, nullptr);
    set_huff(1, 2, 2, 0, 0, ISO_HCOD_1, ISO_HLEN_1);
    set_huff(2, 3, 3, 0, 0, ISO_HCOD_2, ISO_HLEN_2);
    set_huff(3, 3, 3, 0, 0, ISO_HCOD_3, ISO_HLEN_3);
    set_huff(4, 0, 0, 0, 0, nullptr, nullptr);
    set_huff(5, 4, 4, 0, 0, ISO_HCOD_5, ISO_HLEN_5);
    set_huff(6, 4, 4, 0, 0, ISO_HCOD_6, ISO_HLEN_6);
    set_huff(7, 6, 6, 0, 0, ISO_HCOD_7, ISO_HLEN_7);
    set_huff(8, 6, 6, 0, 0, ISO_HCOD_8, ISO_HLEN_8);
    set_huff(9, 6, 6, 0, 0, ISO_HCOD_9, ISO_HLEN_9);
    set_huff(10, 8, 8, 0, 0, ISO_HCOD_10, ISO_HLEN_10);
    set_huff(11, 8, 8, 0, 0, ISO_HCOD_11, ISO_HLEN_11);
    set_huff(12, 8, 8, 0, 0, ISO_HCOD_12, ISO_HLEN_12);
    set_huff(13, 16, 16, 0, 0, ISO_HCOD_13, ISO_HLEN_13);
    set_huff(14, 0, 0, 0, 0, nullptr, nullptr);
    set_huff(15, 16, 16, 0, 0, ISO_HCOD_15, ISO_HLEN_15);
    static const int lb16[8] = {1, 2, 3, 4, 6, 8, 10, 13}, lb24[8] = {4, 5, 6, 7, 8, 9, 11, 13};
    for (int k = 0; k < 8; k++) {
        set_huff(16 + k, 16, 16, lb16[k], (1 << lb16[k]) - 1, ISO_HCOD_16, ISO_HLEN_16);
        set_huff(24 + k, 16, 16, lb24[k], (1 << lb24[k]) - 1, ISO_HCOD_24, ISO_HLEN_24);
    }
    {
        // the scfsi band energies take a log of an integer sum: one value per octave and at most one step inside it
        auto en_of = [](int64_t e) { return (int32_t)(std::log((double)e * 4.768371584e-7) / 0.69314718); };
        for (int k = 0; k < 32; k++) { T.en_base[k] = 0; T.en_step[k] = 0x7fffffff; }
        for (int k = 0; k < 31; k++) {
            const int64_t lo = (int64_t)1 << k, hi = ((int64_t)1 << (k + 1)) - 1;
            T.en_base[k] = en_of(lo);
            if (en_of(hi) == T.en_base[k]) continue;
            if (en_of(hi) != T.en_base[k] + 1) abort();          // an octave moves the quotient by one
            int64_t a = lo, b = hi;                                // en_of(a) == base, en_of(b) == base + 1
            while (b - a > 1) { const int64_t m = (a + b) / 2; (en_of(m) == T.en_base[k] ? a : b) = m; }
            T.en_step[k] = (int32_t)b;
        }
    }

    set_huff(32, 1, 16, 0, 0, ISO_HCOD_32, ISO_HLEN_32);
    set_huff(33, 1, 16, 0, 0, ISO_HCOD_33, ISO_HLEN_33);
    for (int i = 0; i < 256; i++) {
        T.hlen13[i] = ISO_HLEN_13[i]; T.hlen15[i] = ISO_HLEN_15[i];
        T.hlen16[i] = ISO_HLEN_16[i]; T.hlen24[i] = ISO_HLEN_24[i];
    }
    for (int i = 0; i < 16; i++) T.hlen_c1a[i] = ISO_HLEN_32[i];
    for (int i = 0; i < 256; i++) {
        // (k_rate.hpp, RlTables::hl) .x: code length in books 13 | 15 << 5 | 16.. << 10 | 24.. << 15 | non-zero values << 20 | values > 14 << 22
        // | (shortest of the four + non-zero values; 0 for the pair (0,0)) << 25;  .y: that shortest field | (longest + non-zero + 13 per escape) << 16
        const uint32_t x = (uint32_t)i >> 4, y = (uint32_t)i & 15u, nz = (x != 0) + (y != 0), esc = (x == 15) + (y == 15);
        const uint32_t l13 = T.hlen13[i], l15 = T.hlen15[i], l16 = T.hlen16[i], l24 = T.hlen24[i];
        const uint32_t shortest = std::min(std::min(l13, l15), std::min(l16, l24)), longest = std::max(std::max(l13, l15), std::max(l16, l24));
        T.rl_hl[i][0] = l13 | (l15 << 5) | (l16 << 10) | (l24 << 15) | (nz << 20) | (esc << 22) | (i ? (shortest + nz) << 25 : 0u);
        T.rl_hl[i][1] = (i ? shortest + nz : 0u) | ((longest + nz + 13u * esc) << 16);
    }
    for (int i = 0; i < 16; i++) T.rl_c1w[i] = (uint32_t)T.hlen_c1a[i] | ((uint32_t)__builtin_popcount(i & 3) << 16);
    for (int i = 0; i < 32; i++) { T.linbits[i] = (uint8_t)H.huff[i].linbits; T.linmax[i] = H.huff[i].linmax; }
    // ---- device Huffman decode tables
    static const int kDecMax[32] = {1, 2, 3, 3, 0, 4, 4, 6, 6, 6, 8, 8, 8, 16, 0, 16,
                                    16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16};
    static const int kBooks[15] = {1, 2, 3, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15, 16, 24};
    for (int t = 0; t < 32; t++) T.dec_max[t] = (uint8_t)kDecMax[t];
    uint16_t *L1 = T.huf_tab, *L2 = T.huf_tab + HUF_L1_N, *C1 = T.huf_tab + HUF_L1_N + HUF_L2_N;
    int l1_used = 1, l2_used = 0;                     // entry 0: the books without code words (zeros, no bits: SURVEY D2)
    for (int t = 0; t < 32; t++) T.huf_tinfo[t] = (uint32_t)H.huff[t].linbits << 4;   // w = 0: entry 0
    for (int b = 0; b < 15; b++) {
        const int t = kBooks[b];
        const HostHuff &h = H.huff[t];
        // binary trie of the book: child per bit, 0x8000 | (x << 4) | y = leaf, else node index, 0 = no code
        static uint16_t tree[512][2];
        std::memset(tree, 0, sizeof tree);
        int n_nodes = 1, longest = 0;   // node 0 = root
        for (int x = 0; x < kDecMax[t]; x++)
            for (int y = 0; y < kDecMax[t]; y++) {
                const int len = h.hlen[x * h.ylen + y];
                const uint32_t code = h.hcod[x * h.ylen + y];
                longest = std::max(longest, len);
                int node = 0;
                for (int d = 0; d < len; d++) {
                    const int bit = (code >> (len - 1 - d)) & 1;
                    if (d == len - 1) { if (!tree[node][bit]) tree[node][bit] = (uint16_t)(0x8000 | (x << 4) | y); }
                    else {
                        if (!tree[node][bit]) tree[node][bit] = (uint16_t)n_nodes++;
                        node = tree[node][bit];
                        if (node & 0x8000) break;   // cannot happen for a prefix code
                    }
                }
            }
        // depth of the subtree below a node = bits its second-level table is indexed with
        std::function<int(int)> depth = [&](int node) {
            int m = 0;
            for (int bit = 0; bit < 2; bit++) {
                const uint16_t nx = tree[node][bit];
                if (!nx) continue;
                m = std::max(m, (nx & 0x8000) ? 1 : 1 + depth(nx));
            }
            return m;
        };
        const int w = std::min(longest, HUF_W_MAX), base = l1_used;
        const bool has_linbits = t >= 16;
        auto leaf = [&](int len_total, int rel, uint16_t sym) {   // rel: what adv is counted from (0, or w for the second level)
            const int x = (sym >> 4) & 15, y = sym & 15, nsig = (x != 0) + (y != 0), adv = len_total + nsig - rel;
            const int c = nsig + (x == 0 && y != 0);   // the two bits at (30 + c - adv) of the window: x's sign above y's
            if (adv < 0 || adv > 15) abort();
            return (uint16_t)(((has_linbits && (x == 15 || y == 15)) ? 0x4000 : 0) | (c << 12) | (adv << 8) | (x << 4) | y);
        };
        if (l1_used + (1 << w) > HUF_L1_N) abort();
        l1_used += 1 << w;
        const uint32_t info = ((uint32_t)(base * 2) << 16) | ((uint32_t)((32 - w) & 31) << 8) | (uint32_t)w;
        for (int k = (t == 16 || t == 24 ? t : t), k_end = (t == 16 || t == 24 ? t + 8 : t + 1); k < k_end; k++)
            T.huf_tinfo[k] = info | ((uint32_t)H.huff[k].linbits << 4);
        for (uint32_t v = 0; v < (1u << w); v++) {
            int node = 0;
            uint16_t e = 0;
            for (int d = 0; d < w; d++) {
                const uint16_t nxt = tree[node][(v >> (w - 1 - d)) & 1];
                if (!nxt) { e = 0; node = -1; break; }
                if (nxt & 0x8000) { e = leaf(d + 1, 0, nxt); node = -1; break; }
                node = nxt;
            }
            if (node > 0) {   // codes longer than w bits share this prefix: one second-level table for the subtree
                const int k = depth(node), off = l2_used;
                if (k > 15 || (off & 1) || off + (1 << k) > HUF_L2_N || (off >> 1) >= 2048) abort();
                l2_used += (1 << k) + ((1 << k) & 1);          // keep the next table on an even offset
                for (uint32_t u = 0; u < (1u << k); u++) {
                    int nd = node;
                    uint16_t le = 0;
                    for (int d = 0; d < k; d++) {
                        const uint16_t nxt = tree[nd][(u >> (k - 1 - d)) & 1];
                        if (!nxt) break;
                        if (nxt & 0x8000) { le = leaf(w + d + 1, w, nxt); break; }
                        nd = nxt;
                    }
                    L2[off + u] = le;
                }
                e = (uint16_t)(0x8000 | (k << 11) | (off >> 1));
            }
            L1[base + v] = e;
        }
    }
    {
        // count1: book A (ISO table 32, code words of 1..6 bits) and book B (four bits, inverted), each followed by one sign
        // bit per non-zero value in the order v, w, x, y (reference decoder/Frame.py:521-554)
        const HostHuff &q = H.huff[32];
        uint16_t qa[64] = {0};
        for (int e = 0; e < 16; e++) {
            const int len = q.hlen[e];
            const uint32_t base = (uint32_t)q.hcod[e] << (6 - len);
            for (uint32_t f = 0; f < (1u << (6 - len)); f++)
                if (!qa[base + f]) qa[base + f] = (uint16_t)((len << 4) | e);
            T.hcod_c1a[e] = (uint8_t)q.hcod[e];
        }
        // two entries per quadruple, in the format of the big-values leaves: (v, w) with adv = code word + their sign bits, then
        // (x, y) with adv = code word + all sign bits -- the decoder looks the same bits up twice and moves on after the second
        for (int book = 0; book < 2; book++) {
            const int w = book ? 8 : 10;
            uint16_t *half0 = C1 + (book ? 2048 : 0), *half1 = half0 + (1 << w);
            for (uint32_t v = 0; v < (1u << w); v++) {
                int val, len;
                if (book) { val = (int)((v >> 4) ^ 15u); len = 4; }
                else { const uint16_t s = qa[v >> 4]; val = s ? (s & 15) : 0; len = s >> 4; }
                const int q0 = (val >> 3) & 1, q1 = (val >> 2) & 1, q2 = (val >> 1) & 1, q3 = val & 1;
                const int n01 = q0 + q1, n23 = q2 + q3;
                half0[v] = (uint16_t)(((n01 + (!q0 && q1)) << 12) | ((len + n01) << 8) | (q0 << 4) | q1);
                half1[v] = (uint16_t)(((n23 + (!q2 && q3)) << 12) | ((len + n01 + n23) << 8) | (q2 << 4) | q3);
            }
        }
        for (int i = 0; i < 256; i++) {
            T.hcod[0][i] = ISO_HCOD_13[i]; T.hcod[1][i] = ISO_HCOD_15[i];
            T.hcod[2][i] = ISO_HCOD_16[i]; T.hcod[3][i] = ISO_HCOD_24[i];
        }
    }
}

}  // namespace

const HostTables &host_tables()
{
    std::call_once(g_once, build);
    return g_tables;
}

}  // namespace mp3s
