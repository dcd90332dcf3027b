// Decode transform kernels (gfx950).  Included by mp3s_device.hip only.
//
//   k_dec_imdct : requantise -> MS stereo -> reorder | alias reduction -> IMDCT + window + overlap-add
//                 (reference decoder/Frame.py:157-218, 561-622, 106-154; frequency inversion :624-631
//                 is folded into the stores).  One wavefront walks `run` consecutive granules, lane =
//                 (channel, subband): the 18 lines of a subband and the 18-sample overlap tail live in
//                 one lane's registers, the IMDCT twiddle of a given (output, term) is the same for every
//                 lane, so it is a scalar (SGPR) operand fetched through the scalar cache -- no LDS
//                 traffic in the inner loop.  The run is primed with the tail of the granule before it.
//   k_dec_synth : polyphase matrixing, windowing and PCM conversion
//                 (reference Frame.py:65-103, 633-640, MP3_Parser.py:91).  lane = time slot:
//                 the 32 subband samples of a slot live in registers, the 64x32 matrix and D[] are
//                 scalar operands; the 16-slot V history is exchanged between lanes through LDS.
//
// All float math is fp64 with the reference's operation order (separate multiply and add, sums
// ascending from +0.0): the results are bit-identical to the reference's float64 samples.
#pragma once

namespace mp3s {

// Workgroup b runs on XCD b % 8 and every XCD has its own L2.  Neighbouring tiles of the transform kernels share
// halo rows (previous granule, 15 earlier time slots), so each XCD gets a CONTIGUOUS range of tiles: the halo a
// workgroup needs was just read by the workgroup before it on the same XCD.
__device__ __forceinline__ int xcd_tile()
{
    const int b = blockIdx.x, G = gridDim.x;
    const int x = b & 7, q = G >> 3, r = G & 7;
    return x * q + (x < r ? x : r) + (b >> 3);
}


// a lane permutation of the ALU (DPP) on both halves of a double
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// A bound on the largest of a wave's non-negative doubles (NaN counts as the largest): the high words compared as integers
// (six DPP steps on 32 bits instead of six 64-bit permutes, compares and selects), one unit in the 20th mantissa bit added
// on the way back.  Only for scales of guards.
__device__ __forceinline__ double wave_max_bound_f64(double a)
{
    uint32_t v = (uint32_t)__double2hiint(a);
#define MP3S_DPP_MAX(ctrl, rm) v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rm, 0xf, false))
    MP3S_DPP_MAX(0x111, 0xf); MP3S_DPP_MAX(0x112, 0xf); MP3S_DPP_MAX(0x114, 0xf); MP3S_DPP_MAX(0x118, 0xf);   // row_shr:1, 2, 4, 8
    MP3S_DPP_MAX(0x142, 0xa); MP3S_DPP_MAX(0x143, 0xc);                                                       // row_bcast:15, 31
#undef MP3S_DPP_MAX
    return __hiloint2double(__builtin_amdgcn_readlane((int)v, 63) + 1, 0);
}

__device__ __forceinline__ double shfl_xor_f64(double v, int mask)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, mask, 64);
    hi = __shfl_xor(hi, mask, 64);
    return __hiloint2double(hi, lo);
}

// ---------------------------------------------------------------------------------------------
// Kernel A.  S layout: float64 [ch][slot][32 subbands] (slot = granule*18 + i): the time-domain subband
// samples after overlap-add and frequency inversion -- what synth_filter_bank reads (Frame.py:78-79).
// A wave stores one 256-byte row per channel per slot: fully coalesced.
// ---------------------------------------------------------------------------------------------
constexpr int DEC_A_WAVES = 4;

typedef double dvec8 __attribute__((ext_vector_type(8)));
typedef double dvec2 __attribute__((ext_vector_type(2)));
struct DecShared {
    double pow2q[POW2Q_N];                 // copies of the small exponent tables: random per-lane reads go to LDS
    double pow2h[POW2H_N];
    double buf[DEC_A_WAVES][2][32 * 19];   // per wave: spectrum exchange for reorder / alias reduction; a subband's 18 lines at a stride of 19
                                           // doubles (38 dwords: the 32 lanes of a channel hit 32 different bank pairs; at 18 they hit 16, twice)
    uint32_t side[DEC_A_WAVES][2][18];     // per wave: the two 72-byte side records of the current granule
    double exp2f[DEC_A_WAVES][2][64];      // per wave and channel: 2^(-exp2) per scalefactor slot of the current granule
    double exp1f[DEC_A_WAVES][2][4];       // ... and 2^(exp1/4) per gain selector
    double win[4][36];                     // sine_block: the fast IMDCT reads its window factors per lane (the channels may differ
                                           // in block type) instead of carrying both channels' factors in scalar registers
};
__device__ __forceinline__ void dec_stage_tables(DecShared &sh)
{
    for (int i = threadIdx.x; i < POW2Q_N; i += blockDim.x) sh.pow2q[i] = c_tab.pow2q[i];
    if (threadIdx.x < POW2H_N) sh.pow2h[threadIdx.x] = c_tab.pow2h[threadIdx.x];
    if (threadIdx.x < 144) (&sh.win[0][0])[threadIdx.x] = (&c_tab.sine_block[0][0])[threadIdx.x];
    __syncthreads();
}

// what a lane reads from memory for a granule: its dword of the two 72-byte side records (lanes 0..35) and the 18 int16 lines
// of its subband (9 dwords).  Asked for one granule ahead (imdct_run): under the 18 rows of the granule in front.
struct GranIn { uint32_t side; uint32_t xw[9]; };
__device__ __forceinline__ GranIn dec_fetch(const int16_t *__restrict__ is, const mp3s_granule_si *__restrict__ si, int g, int nch, int lane)
{
    const int ch = lane >> 5, sb = lane & 31;
    const bool live = ch < nch;
    GranIn in;
    in.side = lane < 36 ? reinterpret_cast<const uint32_t *>(si + (long)g * 2)[lane] : 0u;
    const uint32_t *xp = reinterpret_cast<const uint32_t *>(is + ((long)g * 2 + (live ? ch : 0)) * 576 + sb * 18);
#pragma unroll
    for (int k = 0; k < 9; k++) in.xw[k] = live ? xp[k] : 0u;
    return in;
}

// The side records of a granule's channels into the wave's LDS slice and, from them, the two exponent factors of requantisation per
// scalefactor slot / gain selector (Frame.py:185-213): what every line's requantisation reads.  SH: the kernel's shared block (side, exp2f,
// exp1f, pow2q, pow2h: DecShared here, StShared in k_decode_stream.hpp).  bt / cse: this lane's channel's block type and case (0 long,
// 1 block_type 2, 2 mixed flag with another block type).
template <class SH>
__device__ __forceinline__ void dec_requant_tables(SH &sh, int wave, const GranIn &in, int nch, int lane, int &bt_out, int &cse_out)
{
    const int ch = lane >> 5, sb = lane & 31;
    const bool live = ch < nch;
    // ---- side records of both channels -> LDS (36 dwords), then byte reads from there
    __builtin_amdgcn_wave_barrier();
    if (lane < 36) (&sh.side[wave][0][0])[lane] = in.side;
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    const uint8_t *gb = reinterpret_cast<const uint8_t *>(sh.side[wave][live ? ch : 0]);
    const int gg = gb[0], bt = gb[2] & 3, mixed = gb[3] ? 1 : 0;
    const int mult2 = gb[1] ? 2 : 1, preflag = gb[4] ? 1 : 0;
    bt_out = bt; cse_out = bt == 2 ? 1 : (mixed ? 2 : 0);
    // ---- the two exponent factors depend on the line only through its scalefactor band / window: 61 + 4 values per
    //      granule and channel, worked out once by the wave (two slots per lane) instead of once per line
    //      long slot s:  exp2 = mult * (sf_l[s] + preflag * pretab[s]);  short slot 22 + 13 w + s:  exp2 = mult * sf_s[w][s]
    //      selector 0:   exp1 = gg - 210;  selector 1 + w:  exp1 = gg - 210 - 8 * sub_block_gain[w]
    double *e2 = sh.exp2f[wave][ch], *e1 = sh.exp1f[wave][ch];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int slot = sb + 32 * h;
        if (slot < 61) {
            int k2;
            if (slot < 22) {
                const int pt = slot < 11 || slot > 20 ? 0 : (int)((0x2333221111ull >> ((slot - 11) * 4)) & 15);   // pre_tab[11..20]
                k2 = mult2 * ((gb[8 + slot] & 15) + preflag * pt);
            } else k2 = mult2 * (gb[30 + (slot - 22)] & 15);
            e2[slot] = sh.pow2h[k2 < POW2H_N ? k2 : POW2H_N - 1];
        }
    }
    if (sb < 4) e1[sb] = sh.pow2q[gg - 210 - (sb ? 8 * (gb[5 + sb - 1] & 7) : 0) - POW2Q_MIN];
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
}

// requantise (Frame.py:157-218) and MS stereo (:561-572) of granule `in` for this lane's subband, lines in spectrum order.
template <class SH>
__device__ __forceinline__ void dec_requant_ms(const DevTables &tab, SH &sh, int wave, double (&v)[18], const GranIn &in,
                                               int sr, bool ms, int nch, int lane, int &bt_out, int &cse_out)
{
    const int ch = lane >> 5, sb = lane & 31;
    dec_requant_tables(sh, wave, in, nch, lane, bt_out, cse_out);
    const int cse = cse_out;
    // ---- 18 int16 spectrum values (9 dwords) and 18 line-map bytes (5 dwords) of this subband
    uint32_t mw[5];
    const uint32_t (&xw)[9] = in.xw;
    {
        const uint32_t *mp = reinterpret_cast<const uint32_t *>(tab.rq_map[sr][cse][sb]);
#pragma unroll
        for (int k = 0; k < 5; k++) mw[k] = mp[k];
    }
    {
        const double *e2 = sh.exp2f[wave][ch], *e1 = sh.exp1f[wave][ch];
        // ---- requantise (Frame.py:210-215): ((sign * |is|^(4/3)) * 2^(exp1/4)) * 2^(-exp2)
        // (|is|^(4/3) from a scalar base + a 32-bit lane offset: behind the table's 29 KB offset in DevTables the compiler built a 64-bit
        // address per value, three vector instructions more; the map byte's two fields by bit-field extracts)
        typedef const double __attribute__((address_space(1))) *gtab_ptr;
        gtab_ptr p43 = (gtab_ptr)tab.pow43;
        asm volatile("" : "+s"(p43));
#pragma unroll
        for (int k = 0; k < 18; k++) {
            const int x = (int)(int16_t)(xw[k >> 1] >> ((k & 1) * 16));
            const uint32_t mwk = mw[k >> 2];
            // (spelled out: the compiler turns a bit-field extract + scaled add back into shift, mask and add)
            uint32_t i1, i2;
            asm("v_bfe_u32 %0, %1, %2, 2" : "=v"(i1) : "v"(mwk), "n"((k & 3) * 8 + 6));
            asm("v_bfe_u32 %0, %1, %2, 6" : "=v"(i2) : "v"(mwk), "n"((k & 3) * 8));
            uint32_t ax = (uint32_t)(x < 0 ? -x : x);
            ax = ax < (uint32_t)POW43_N ? ax : (uint32_t)(POW43_N - 1);
            const double a = p43[ax];
            // sign * a is exact: the sign bit of the 16-bit value goes straight into the high word (a >= 0)
            const uint32_t sgn = (k & 1) ? xw[k >> 1] : xw[k >> 1] << 16;
            const double sa = __hiloint2double((int)(((uint32_t)__double2hiint(a) & 0x7fffffffu) | (sgn & 0x80000000u)), __double2loint(a));
            v[k] = (sa * e1[i1]) * e2[i2];
        }
    }
    // ---- MS stereo (Frame.py:568-572): L = (M + S) / sqrt2, R = (M - S) / sqrt2
    if (ms && nch == 2) {
#pragma unroll
        for (int k = 0; k < 18; k++) {
            const double o = shfl_xor_f64(v[k], 32);
            v[k] = ch == 0 ? (v[k] + o) / tab.sqrt2 : (o - v[k]) / tab.sqrt2;
        }
    }
}

// requantise .. alias/reorder of granule g for this lane's subband; v[18] = IMDCT input
__device__ __forceinline__ void dec_prepare(const DevTables &tab, DecShared &sh, int wave, double (&v)[18], const GranIn &in,
                                            int sr, bool ms, int nch, int lane, int &bt_out)
{
    const int ch = lane >> 5, sb = lane & 31;
    int cse;
    dec_requant_ms(tab, sh, wave, v, in, sr, ms, nch, lane, bt_out, cse);
    // ---- reorder (short / mixed) or alias reduction (long) through the wave's LDS slice
    double *buf = sh.buf[wave][ch];
#pragma unroll
    for (int k = 0; k < 18; k++) buf[sb * 19 + k] = v[k];
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's own LDS writes have landed
    __builtin_amdgcn_wave_barrier();
    if (cse != 0) {
        // (the subband's 18 source lines as 9 dwords, all requested before the first is used: loaded one by one the compiler
        // waited for each in turn -- 18 memory latencies per short granule)
        const uint32_t *src = reinterpret_cast<const uint32_t *>(tab.reorder_src[sr] + sb * 18);
        uint32_t sw[9];
#pragma unroll
        for (int k = 0; k < 9; k++) sw[k] = src[k];
#pragma unroll
        for (int k = 0; k < 18; k++) {
            const int s = (int)(int16_t)(sw[k >> 1] >> ((k & 1) * 16));   // a line of the spectrum, or -1
            const double x = buf[s >= 0 ? s + ((s * 3641) >> 16) : 0];    // line s sits at s + s / 18 (s < 576)
            v[k] = s >= 0 ? x : 0.0;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            // butterfly with subband sb-1 (lower line, Frame.py:622: s2*cs + s1*ca) and with subband sb+1 (upper line,
            // :621: s1*cs - s2*ca); branch-free: every lane evaluates both and the edge subbands keep their value
            const int ilo = 19 * sb - 2 - i, ihi = 19 * (sb + 1) + i;      // line 17 - i of subband sb - 1, line i of subband sb + 1
            const double nlo = buf[ilo < 0 ? 0 : ilo], nhi = buf[ihi > 607 ? 607 : ihi];
            const double cs = tab.alias_cs[i], ca = tab.alias_ca[i];
            const double lo = v[i] * cs + nlo * ca;
            const double hi = v[17 - i] * cs - nhi * ca;
            v[i] = sb >= 1 ? lo : v[i];
            v[17 - i] = sb <= 30 ? hi : v[17 - i];
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
}

// twiddle row i alone (the fast IMDCT takes its window factors from LDS)
struct Row18s { dvec8 a, b; dvec2 c; };
__device__ __forceinline__ Row18s load_row18s(const double (*C36)[18], int i)
{
    Row18s r;
    const double *p = C36[i];
    r.a = *reinterpret_cast<const dvec8 *>(p);
    r.b = *reinterpret_cast<const dvec8 *>(p + 8);
    r.c = *reinterpret_cast<const dvec2 *>(p + 16);
    return r;
}

// IMDCT + window (Frame.py:124-148), overlap (:151-153) and frequency inversion (:629-631) of ONE granule of this lane's subband in the
// REFERENCE's order: every output a sum of separately rounded products ascending from +0.0.  v[18] = the IMDCT's input, tail = the second half of
// the block in front (in) / of this block (out).  `rows`: the granule's 18 time samples are wanted -- out(i, x) gets sample i, in slot order --,
// else only the tail (a granule that primes a run).  Used by the exact kernels' S rows (imdct_run), the fix-up kernel and the exact stream kernel
// (k_decode_exact.hpp): one source for the order of the sums.
template <class OUT>
__device__ __forceinline__ void imdct_rows_exact(const DevTables &tab, const DecShared &sh, const double (&v)[18], int bt, double (&tail)[18], bool rows,
                                                 uint32_t sgn_odd, OUT out)
{
    auto flip = [&](double x) { return __hiloint2double(__double2hiint(x) ^ (int)sgn_odd, __double2loint(x)); };
    const double(*C36)[18] = tab.imdct_cos36;
    const double(*C12)[6] = tab.imdct_cos12;
    if (bt != 2) {
        // One row of twiddles (18 doubles) per scalar batch, the row after the one being multiplied requested first: its latency
        // passes under 18 multiply-adds per lane.  The window factor is the lane's own read of the staged copy in LDS (as in
        // the fast rows): with the two channel halves' factors in the scalar batch two rows in flight were 80 scalar registers --
        // most of this kernel's 400 scalar spills, a read-lane per five products in the rows.  Same doubles, same products.
        const double *wl = sh.win[bt];
        Row18s cur = load_row18s(C36, rows ? 0 : 18);
        double wc = wl[rows ? 0 : 18];
        if (rows) {
#pragma unroll
            for (int i = 0; i < 18; i++) {
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_sched_barrier(0);
                const Row18s nxt = load_row18s(C36, i + 1);
                const double wn = wl[i + 1];
                __builtin_amdgcn_sched_barrier(0);
                double x = 0.0;
#pragma unroll
                for (int k = 0; k < 18; k++) x += v[k] * (k < 8 ? cur.a[k & 7] : (k < 16 ? cur.b[k & 7] : cur.c[k & 1]));
                asm volatile("" : "+v"(x));      // (as in the fast path: the sum stays in front of the store's branch)
                x = x * wc + tail[i];
                if ((i) & 1) x = flip(x);
                out(i, x);
                __builtin_amdgcn_sched_barrier(0);
                cur = nxt; wc = wn;
            }
        }
#pragma unroll
        for (int i = 18; i < 36; i++) {
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_sched_barrier(0);
            const int in = i < 35 ? i + 1 : 35;
            const Row18s nxt = load_row18s(C36, in);
            const double wn = wl[in];
            __builtin_amdgcn_sched_barrier(0);
            double x = 0.0;
#pragma unroll
            for (int k = 0; k < 18; k++) x += v[k] * (k < 8 ? cur.a[k & 7] : (k < 16 ? cur.b[k & 7] : cur.c[k & 1]));
            tail[i - 18] = x * wc;
            __builtin_amdgcn_sched_barrier(0);
            cur = nxt; wc = wn;
        }
    } else {
        // three 12-point windows placed at 6/12/18 (Frame.py:135-148).  The three windows share their coefficients, so the
        // walk is over the 12 ROWS: row j gives t[j] (window 0, to sample_block[6 + j]), t[12 + j] (window 1, to [12 + j]) and
        // t[24 + j] (window 2, to [18 + j]), each used exactly once -- six sums are kept (window 1's first half, until rows
        // 6..11 bring window 0's second half), the rest goes where it belongs at once.  Same sums, same order, same roundings
        // as the reference's; 14 scalar words per row instead of the whole 12 x 6 table in flight (which was most of the
        // kernel's scalar spills) and 12 instead of 48 registers for the windows.
        typedef double dvec4 __attribute__((ext_vector_type(4)));
        struct Row6 { dvec4 a; dvec2 b; double s; };
        auto load_row6 = [&](int j) { Row6 r; r.a = *reinterpret_cast<const dvec4 *>(C12[j]); r.b = *reinterpret_cast<const dvec2 *>(C12[j] + 4); r.s = tab.sine_block[2][j]; return r; };
        if (rows) {
#pragma unroll
            for (int i = 0; i < 6; i++) {                       // sample_block[0..5] = 0
                double x = 0.0 + tail[i];
                if ((i) & 1) x = flip(x);
                out(i, x);
            }
        }
        double hold[6];
        Row6 cur = load_row6(0);
#pragma unroll
        for (int j = 0; j < 12; j++) {
            const Row6 nxt = load_row6(j < 11 ? j + 1 : 11);
            __builtin_amdgcn_sched_barrier(0);
            double a0 = 0.0, a1 = 0.0, a2 = 0.0;
#pragma unroll
            for (int k = 0; k < 6; k++) {
                const double c = k < 4 ? cur.a[k] : cur.b[k - 4];
                a0 += v[k] * c; a1 += v[6 + k] * c; a2 += v[12 + k] * c;
            }
            a0 = a0 * cur.s; a1 = a1 * cur.s; a2 = a2 * cur.s;
            asm volatile("" : "+v"(a0));
            if (j < 6) {
                if (rows) {                                     // sample_block[6..11] = t[0..5]
                    double x = a0 + tail[6 + j];
                    if ((6 + j) & 1) x = flip(x);
                    out(6 + j, x);
                }
                hold[j] = a1;                                   // t[12..17], for sample_block[12..17]
                tail[j] = a2;                                   // t[24..29], half of sample_block[18..23]
            } else {
                if (rows) {                                     // sample_block[12..17] = t[6..11] + t[12..17]
                    double x = (a0 + hold[j - 6]) + tail[6 + j];
                    if ((6 + j) & 1) x = flip(x);
                    out(6 + j, x);
                }
                tail[j - 6] = a1 + tail[j - 6];                 // sample_block[18..23] = t[18..23] + t[24..29]
                tail[j] = a2;                                   // sample_block[24..29] = t[30..35]
                tail[6 + j] = 0.0;                              // sample_block[30..35]
            }
            __builtin_amdgcn_sched_barrier(0);
            cur = nxt;
        }
    }
}

// One wave walks granules g0 .. g0 + run - 1 (lane = channel, subband), primed with the second half of granule g0 - 1.
// Rows go to S + ((ch * T) + g * 18 + i - slot0) * 32 (slot0 = 0 for the batch's scratch; the fix-up kernel writes a
// private window of slots).
//
// FAST (int16 output only, see k_dec_synth_fast): the 36 outputs of a long-block IMDCT are 18 sums and their mirror
// images -- x[17-i] = -x[i], x[53-i] = x[i] -- and the sums are fused multiply-adds: 324 instead of 1296 instructions per
// subband.  The result is NOT the reference's bit pattern: it differs from it by at most kappa * (B + B') with B, B' the
// sums of |input| of this subband in this and the previous granule (DevTables::imdct_kappa, derivation in
// mp3s_tables.cpp); the wave leaves G = sum of B over the subbands of each channel behind, the synthesis guard adds
// kappa-scaled G to its bound, and a sample the guard cannot vouch for is recomputed from `is` in the reference's order
// (k_dec_fixup, which runs this function with FAST = false).  Short blocks keep the reference's order in both modes.
template <bool FAST, bool ALLWR = false /* stereo, every lane stores its rows: no exec mask around the stores */>
__device__ __forceinline__ void imdct_run(DecShared &sh, int wave, int lane, const int16_t *__restrict__ is,
                                          const mp3s_granule_si *__restrict__ si, const mp3s_frame_hdr *__restrict__ hdr,
                                          int n_granules, int nch, int g0, int run, double *S, long T, long slot0,
                                          int sf_base, double *__restrict__ G, int only_ch = -1)
{
    const int ch = lane >> 5, sb = lane & 31;
    const bool live = ch < nch;
    const bool wr = live && (only_ch < 0 || ch == only_ch);   // (the fix-up kernel keeps one channel's rows)
    const bool neg_odd = (sb & 1) != 0;
    constexpr bool all_wr = ALLWR;
    // frequency inversion (Frame.py:629-631): odd lines of odd subbands change sign -- a per-lane sign word for the high dword
    // (a select per row would keep a lane mask in two scalar registers this loop does not have)
    const uint32_t sgn_odd = neg_odd ? 0x80000000u : 0u;
    auto flip = [&](double x) { return __hiloint2double(__double2hiint(x) ^ (int)sgn_odd, __double2loint(x)); };
    double tail[18];
#pragma unroll
    for (int i = 0; i < 18; i++) tail[i] = 0.0;

    GranIn next_in = {};
    bool have_next = false;
    // gi = -1 primes the overlap with the second half of granule g0-1 (when it belongs to the same stream)
    for (int gi = -1; gi < run; gi++) {
        const int g = g0 + gi;
        if (g >= n_granules) break;
        if (g < 0) continue;
        const mp3s_frame_hdr fh = hdr[g >> 1];
        // stream_first counts from frame sf_base of the batch; this launch starts there (a chunk of a longer batch)
        const int first_gran = fh.stream_first > (uint32_t)sf_base ? (int)(fh.stream_first - (uint32_t)sf_base) * 2 : 0;
        if (gi < 0 && g0 <= first_gran) continue;          // the run starts a stream: nothing before it
        if (gi >= 0 && g == first_gran) {
#pragma unroll
            for (int i = 0; i < 18; i++) tail[i] = 0.0;    // Frame.py:234 prev_samples starts as zeros
        }
        const int sr = fh.sr_idx < 3 ? fh.sr_idx : 0;
        // The twiddle tables are invariant over this loop; left alone, LICM hoists all 648 scalar loads out of it
        // and spills the SGPRs.  An opaque zero offset per iteration keeps them streaming through the scalar cache.
        int zoff = 0;
        asm volatile("" : "+s"(zoff));
        const DevTables &tab = *reinterpret_cast<const DevTables *>(reinterpret_cast<const char *>(&c_tab) + zoff);
        const double(*C36)[18] = tab.imdct_cos36;
        double v[18];
        int bt;
        // this granule's lines and side records: asked for under the rows of the granule in front (below), or here for the
        // first one; the next granule's set off behind this granule's own loads (table look-ups of the requantisation), so that
        // a whole granule of arithmetic lies between the request and the first use
        const GranIn in = have_next ? next_in : dec_fetch(is, si, g, nch, lane);
        dec_prepare(tab, sh, wave, v, in, sr, fh.ms_stereo != 0, nch, lane, bt);
        have_next = gi + 1 < run && g + 1 < n_granules;
        if (have_next) next_in = dec_fetch(is, si, g + 1, nch, lane);
        if (FAST && gi >= 0 && G) {
            // G[g][ch] = sum over the channel's 32 subbands of sum_k |v[k]| (what the guard of the synthesis scales kappa with):
            // inside a row of 16 lanes by lane permutations of the ALU (quad, half row, row), the two rows of a channel through
            // one scalar read -- five round trips through the LDS crossbar (ds_bpermute) were a twentieth of a granule
            double b = 0.0;
#pragma unroll
            for (int k = 0; k < 18; k++) b += fabs(v[k]);
            b += dpp_f64<0xB1>(b);        // quad_perm [1, 0, 3, 2]
            b += dpp_f64<0x4E>(b);        // quad_perm [2, 3, 0, 1]
            b += dpp_f64<0x141>(b);       // row_half_mirror
            b += dpp_f64<0x140>(b);       // row_mirror
            const double r1 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(b), 16), __builtin_amdgcn_readlane(__double2loint(b), 16));
            const double r3 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(b), 48), __builtin_amdgcn_readlane(__double2loint(b), 48));
            if (sb == 0 && live) G[(long)g * 2 + ch] = b + (ch ? r3 : r1);
        }
        // ---- IMDCT + window (Frame.py:124-148), overlap (:151-153), frequency inversion (:629-631) in the sign
        double *row = S + ((long)(live ? ch : 0) * T + ((long)g * 18 - slot0)) * 32 + sb;
        if (bt != 2) {
            if (FAST) {
                // rows 0..8 and their mirrors 17..9, then rows 18..26 and their mirrors 35..27; window factors from LDS.
                // A row's twiddles (scalar cache) and its two window factors (LDS) are asked for one row ahead, and the wait for
                // them is spelled out at the TOP of a row, in front of the next request: scalar loads and LDS reads share one
                // counter that can only be waited to zero while both kinds are in flight, and wherever the compiler places that
                // wait itself -- at the first use, or (round 3) in front of the next LDS read, whose destination registers the
                // skipped branch of a store may have left pending -- it lands behind the request and exposes its whole latency.
                const double *wl = sh.win[bt];
                Row18s cur = load_row18s(C36, gi >= 0 ? 0 : 18);
                double wa = wl[gi >= 0 ? 0 : 18], wb = wl[gi >= 0 ? 17 : 35];
                if (gi >= 0) {
#pragma unroll
                    for (int i = 0; i < 9; i++) {
                        __builtin_amdgcn_s_waitcnt(0xc07f);
                        __builtin_amdgcn_sched_barrier(0);
                        const int in = i < 8 ? i + 1 : 18;
                        const Row18s nxt = load_row18s(C36, in);
                        const double nwa = wl[in], nwb = wl[in < 18 ? 17 - in : 53 - in];
                        __builtin_amdgcn_sched_barrier(0);
                        double y = 0.0;
#pragma unroll
                        for (int k = 0; k < 18; k++) y = __builtin_fma(v[k], k < 8 ? cur.a[k & 7] : (k < 16 ? cur.b[k & 7] : cur.c[k & 1]), y);
                        // (the sum is used by the stores alone: left to itself the compiler sinks the whole chain into their branch and asks
                        // for the row again THERE -- request, wait a full scalar-cache latency, 18 multiply-adds, row after row)
                        asm volatile("" : "+v"(y));
                        double xa = y * wa + tail[i];
                        double xb = -y * wb + tail[17 - i];
                        if (i & 1) xa = flip(xa);
                        if ((17 - i) & 1) xb = flip(xb);
                        if (all_wr) { row[(long)i * 32] = xa; row[(long)(17 - i) * 32] = xb; }
                        else if (wr) { row[(long)i * 32] = xa; row[(long)(17 - i) * 32] = xb; }
                        __builtin_amdgcn_sched_barrier(0);
                        cur = nxt; wa = nwa; wb = nwb;
                    }
                }
#pragma unroll
                for (int i = 18; i < 27; i++) {
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                    __builtin_amdgcn_sched_barrier(0);
                    const int in = i < 26 ? i + 1 : 26;
                    const Row18s nxt = load_row18s(C36, in);
                    const double nwa = wl[in], nwb = wl[53 - in];
                    __builtin_amdgcn_sched_barrier(0);
                    double y = 0.0;
#pragma unroll
                    for (int k = 0; k < 18; k++) y = __builtin_fma(v[k], k < 8 ? cur.a[k & 7] : (k < 16 ? cur.b[k & 7] : cur.c[k & 1]), y);
                    tail[i - 18] = y * wa;
                    tail[35 - i] = y * wb;
                    __builtin_amdgcn_sched_barrier(0);
                    cur = nxt; wa = nwa; wb = nwb;
                }
            } else {
                imdct_rows_exact(tab, sh, v, bt, tail, gi >= 0, sgn_odd, [&](int i, double x) { if (all_wr) row[(long)i * 32] = x; else if (wr) row[(long)i * 32] = x; });
            }
        } else {
            imdct_rows_exact(tab, sh, v, bt, tail, gi >= 0, sgn_odd, [&](int i, double x) { if (all_wr) row[(long)i * 32] = x; else if (wr) row[(long)i * 32] = x; });
        }
    }
}

template <bool FAST>
__global__ __launch_bounds__(DEC_A_WAVES * 64, 3) void k_dec_imdct(
    const int16_t *__restrict__ is, const mp3s_granule_si *__restrict__ si, const mp3s_frame_hdr *__restrict__ hdr,
    int n_granules, int nch, int run, double *__restrict__ S, long T, int sf_base, double *__restrict__ G)
{
    __shared__ DecShared sh;
    dec_stage_tables(sh);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int g0 = (xcd_tile() * DEC_A_WAVES + wave) * run;
    if (g0 >= n_granules) return;  // whole wave exits together
    if (nch == 2) imdct_run<FAST, true>(sh, wave, lane, is, si, hdr, n_granules, 2, g0, run, S, T, 0, sf_base, G);
    else imdct_run<FAST, false>(sh, wave, lane, is, si, hdr, n_granules, nch, g0, run, S, T, 0, sf_base, G);
}

// ---------------------------------------------------------------------------------------------
// Kernel BC.  One workgroup = a tile of TW*64 consecutive slots for every channel; the first 15
// lanes of a tile only rebuild V history for the others (halo), so a tile emits TW*64-15 slots.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int16_t pcm_to_i16(double v)
{
    const double x = v * 32767;
    if (!(fabs(x) < 2147483648.0)) return 0;          // x86 cvttsd2si "indefinite" -> low half 0
    return (int16_t)(uint16_t)((uint32_t)(int32_t)x & 0xffffu);
}

// The 64 rows of S a synthesis wave needs (256 bytes each, one per lane) are one contiguous 16 KB: fetched in four rounds of four requests in
// which FOUR LANES SHARE A CACHE LINE (lane l: row l / 4 + 16 k, 16-byte column 4 r + l % 4), through 5 KB of LDS to the lane that owns the
// row.  (Every lane reading its own row, 16 bytes per request, touched 64 lines per request and used a quarter of each: beside other kernels the
// synthesis was paying for the traffic between L2 and the vector caches -- the step 0.663-0.670 -> 0.652-0.659 ms.)  `stage`: the wave's own
// SYNTH_STAGE_WAVE bytes; they may be reused once every wave of the workgroup is behind a barrier.
constexpr int SYNTH_STAGE_ROW = 80, SYNTH_STAGE_WAVE = 64 * SYNTH_STAGE_ROW;
__device__ __forceinline__ void synth_fetch_rows(unsigned char *stage, const double *__restrict__ S, long T, int ch, long t, int lane, double (&Sv)[32])
{
    const long tw0 = __builtin_amdgcn_readfirstlane((int)((t - lane) >> 32)) * 0x100000000l + (uint32_t)__builtin_amdgcn_readfirstlane((int)(t - lane));   // slot of the wave's lane 0 (a scalar)
    // the wave's block behind a scalar base + a 32-bit lane offset (global address space spelled out: the loads take the base as it is)
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef const u32x4 __attribute__((address_space(1))) *gq_ptr;
    gq_ptr blk = (gq_ptr)(S + ((long)ch * T + tw0) * 32);                 // (rows outside the batch are not read)
    const uint32_t voff = (uint32_t)((lane >> 2) * 16 + (lane & 3));      // in 16-byte units: row lane / 4, column lane % 4
    const bool inside = tw0 >= 0 && tw0 + 64 <= T;                        // (wave-uniform: all but a batch's first and last tiles)
    const int rlo = tw0 < 0 ? (int)-tw0 : 0, rhi = T - tw0 < 64 ? (int)(T - tw0) : 64;
    uint4 q[2][4];
    auto fetch = [&](int r, uint4 (&qq)[4]) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int row = (lane >> 2) + 16 * k;
            if (inside) { const u32x4 v = blk[voff + (uint32_t)(k * 256 + r * 4)]; qq[k] = make_uint4(v.x, v.y, v.z, v.w); }
            else {
                qq[k] = make_uint4(0, 0, 0, 0);
                if (row >= rlo && row < rhi) { const u32x4 v = blk[voff + (uint32_t)(k * 256 + r * 4)]; qq[k] = make_uint4(v.x, v.y, v.z, v.w); }
            }
        }
    };
    const uint32_t wr = (uint32_t)((lane >> 2) * SYNTH_STAGE_ROW + (lane & 3) * 16), rd = (uint32_t)(lane * SYNTH_STAGE_ROW);
    fetch(0, q[0]);
#pragma unroll
    for (int r = 0; r < 4; r++) {
        if (r < 3) fetch(r + 1, q[(r + 1) & 1]);
        __builtin_amdgcn_wave_barrier();                           // (the round before has been read back)
#pragma unroll
        for (int k = 0; k < 4; k++) *reinterpret_cast<uint4 *>(stage + wr + k * 16 * SYNTH_STAGE_ROW) = q[r & 1][k];
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const double2 v = *reinterpret_cast<const double2 *>(stage + rd + c * 16);
            Sv[8 * r + 2 * c] = v.x; Sv[8 * r + 2 * c + 1] = v.y;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
    }
}

template <int TW>
__global__ __launch_bounds__(TW * 64 * 2, 4) void k_dec_synth(
    const double *__restrict__ S, long T, const mp3s_frame_hdr *__restrict__ hdr, int nch, int n_halo, int out_format,
    void *__restrict__ pcm_out, int sf_base)
{
    constexpr int TL_LANES = TW * 64, OUT = TL_LANES - 15;
    constexpr int OROW = 33;                                   // dwords per staged slot (32 + 1 pad: no bank conflicts)
    // one block of LDS: ex[2][2][4][TL_LANES] doubles ([parity][ch][index of the pair * 2 + V half][lane]) | otile[OUT * OROW] dwords; in front
    // of the loop the same bytes stage the waves' rows of S (synth_fetch_rows)
    constexpr int EX_BYTES = 2 * 2 * 4 * TL_LANES * 8, OT_BYTES = OUT * OROW * 4;
    static_assert(TW * 2 * SYNTH_STAGE_WAVE <= EX_BYTES + OT_BYTES, "the staging of S fits the loop's LDS");
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[EX_BYTES + OT_BYTES];
    double (*const ex)[2][4][TL_LANES] = reinterpret_cast<double (*)[2][4][TL_LANES]>(lds_raw);
    uint32_t *const otile = reinterpret_cast<uint32_t *>(lds_raw + EX_BYTES);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int ch = wave / TW, tl = (wave % TW) * 64 + lane;
    const long tile0 = (long)xcd_tile() * OUT;
    const long t = tile0 - 15 + tl;
    const bool valid = t >= 0 && t < T;
    int lim = -1;            // number of earlier in-stream slots (V history available), -1: slot not valid
    if (valid) {
        const uint32_t sf = hdr[t / 36].stream_first;
        const long s0 = sf > (uint32_t)sf_base ? (long)(sf - (uint32_t)sf_base) * 36 : 0;
        lim = (int)((t - s0) < 64 ? (t - s0) : 64);
    }
    double Sv[32];
    synth_fetch_rows(lds_raw + wave * SYNTH_STAGE_WAVE, S, T, ch, t, lane, Sv);
    __syncthreads();                                           // (the staging bytes become the exchange buffers)
    const long halo_slots = (long)n_halo * 36;
    const bool emit = valid && tl >= 15 && t >= halo_slots;
    const bool full_hist = __ballot(tl >= 15 && lim < 15) == 0;
    uint16_t *ot16 = reinterpret_cast<uint16_t *>(otile);
    // The matrix rows and window taps are scalar operands fetched in 8-double batches.  All scalar loads share one
    // counter that can only be waited to zero, so the loop is software-pipelined by hand: the batch after the one being
    // multiplied is requested first, and its latency passes under 16 multiply-adds per lane.
    typedef double d8 __attribute__((ext_vector_type(8)));
    const d8 *M = reinterpret_cast<const d8 *>(&c_tab.synth_matrix[0][0]);      // row r, batch b: M[r * 4 + b]
    const d8 *Wt = reinterpret_cast<const d8 *>(&c_tab.synth_window_t[0][0]);   // output i, taps 8h..8h+7: Wt[i * 2 + h]
    d8 c0 = M[0], c1 = M[32 * 4];
    int p = 0;
    // Two output indices per barrier interval: matrixing of i and i+1, one barrier, then their two window sums.
#pragma unroll 1
    for (int i0 = 0; i0 < 32; i0 += 2) {
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const int i = i0 + s;
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int b = 0; b < 4; b++) {            // Frame.py:84-87
                d8 n0, n1;
                if (b < 3) { n0 = M[i * 4 + b + 1]; n1 = M[(32 + i) * 4 + b + 1]; }
                else if (s == 0) { n0 = M[(i + 1) * 4]; n1 = M[(32 + i + 1) * 4]; }   // first batch of the second index
                else { n0 = Wt[i0 * 2]; n1 = Wt[i0 * 2 + 1]; }                         // window taps of the first index
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    a0 += Sv[8 * b + j] * c0[j];
                    a1 += Sv[8 * b + j] * c1[j];
                }
                __builtin_amdgcn_sched_barrier(0);
                c0 = n0; c1 = n1;
            }
            ex[p][ch][2 * s][tl] = a0;
            ex[p][ch][2 * s + 1][tl] = a1;
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const int i = i0 + s;
            d8 n0, n1;
            if (s == 0) { n0 = Wt[(i0 + 1) * 2]; n1 = Wt[(i0 + 1) * 2 + 1]; }          // taps of the second index
            else { const int inext = i0 < 30 ? i0 + 2 : 31; n0 = M[inext * 4]; n1 = M[(32 + inext) * 4]; }   // next interval
            __builtin_amdgcn_sched_barrier(0);
            if (tl >= 15) {
                double sum = 0.0;
                if (full_hist) {                    // wave-uniform: every lane has 15 in-stream predecessors (the common case)
#pragma unroll
                    for (int jj = 0; jj < 16; jj++)   // Frame.py:89-101 (u, w, sum over 16 windowed taps)
                        sum += ex[p][ch][2 * s + (jj & 1)][tl - jj] * (jj < 8 ? c0[jj & 7] : c1[jj & 7]);
                } else {
#pragma unroll
                    for (int jj = 0; jj < 16; jj++) {
                        double u = ex[p][ch][2 * s + (jj & 1)][tl - jj];
                        if (jj > lim) u = 0.0;       // before the stream started the fifo holds zeros
                        sum += u * (jj < 8 ? c0[jj & 7] : c1[jj & 7]);
                    }
                }
                if (emit) {
                    const long to = t - halo_slots;
                    if (out_format == MP3S_PCM_I16) ot16[(tl - 15) * OROW * 2 + i * nch + ch] = (uint16_t)pcm_to_i16(sum);
                    else if (out_format == MP3S_PCM_F64) ((double *)pcm_out)[(to * 32 + i) * nch + ch] = sum;
                    else ((float *)pcm_out)[(to * 32 + i) * nch + ch] = (float)sum;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            c0 = n0; c1 = n1;
        }
        p ^= 1;
    }
    if (out_format == MP3S_PCM_I16) {
        __syncthreads();
        const int dw_per_slot = 16 * nch;                      // 32 samples * nch * 2 bytes / 4
        const int n_dw = OUT * dw_per_slot;
        uint32_t *outp = (uint32_t *)pcm_out;
        for (int c = threadIdx.x; c < n_dw; c += blockDim.x) {
            const int sl = c / dw_per_slot, w = c - sl * dw_per_slot;
            const long slot = tile0 + sl;
            if (slot < halo_slots || slot >= T) continue;
            outp[(slot - halo_slots) * dw_per_slot + w] = otile[sl * OROW + w];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Kernel BC, fast variant for int16 output.  Same tiling and V exchange as k_dec_synth; the 64 x 32 matrixing is replaced
// by X[k] = sum_j S[j] cos((2j+1) k pi/64), k < 32, evaluated by even / odd splitting (341 multiplications instead of
// 2048; every factor is a cosine, nothing is amplified), from which V[i] = X[16+i] (i <= 16, X[32] = 0), -X[48-i]
// (17 <= i <= 47), -X[i-48] (i >= 48).  The result is NOT the reference's float64 bit pattern: it differs from it by a few
// 1e-14 of the slot's sum |S| (mostly because the reference's own matrix is that far from the true cosines).  What the
// int16 format promises -- (pcm * 32767) truncated exactly as the reference truncates it -- is kept by a guard: a sample
// whose scaled value lies closer to a non-zero integer than the proven bound on that difference (DevTables::synth_eps_a / _x / _g,
// derivation in mp3s_tables.cpp and DESIGN.md; one width per tile since round 4) puts its slot and channel on the fix-up list,
// whose 32 samples are computed again in the reference's order from `is` (k_dec_fixup; the rows in S are the fast IMDCT's, not
// the reference's).  About two samples in ten thousand take that path.  eps_scale (tests): inflates the guard so that the exact
// path is exercised.
// ---------------------------------------------------------------------------------------------
__device__ __noinline__ double synth_exact_sample(const double *S, long T, int ch, long t, int i, int lim)
{
    // Frame.py:84-101 for one output sample: 16 taps, each a 32-term matrixing sum of an earlier slot
    double sum = 0.0;
    for (int jj = 0; jj < 16; jj++) {
        double u = 0.0;
        if (jj <= lim) {
            const double *row = S + ((long)ch * T + (t - jj)) * 32;
            const double *nrow = c_tab.synth_matrix[(jj & 1) ? 32 + i : i];
            double a = 0.0;
            for (int j = 0; j < 32; j++) a += row[j] * nrow[j];
            u = a;
        }
        sum += u * c_tab.synth_window_t[i][jj];
    }
    return sum;
}

// One output of the even / odd splitting: sum_j d[j] * row[j].  The row is wave-uniform and arrives through the scalar
// cache; its address is a run-time value (so the unrolled code around it cannot pull every row of the table to the front),
// and it is REQUESTED one stage before it is used: all scalar loads of a wave share one counter that can only be waited
// to zero, so a stage first asks for the operands of the stage after it, then multiplies with its own.
template <int N> struct SynthRow { typedef double vec __attribute__((ext_vector_type(N >= 8 ? 8 : N))); vec c[N >= 8 ? N / 8 : 1]; };
template <int N>
__device__ __forceinline__ SynthRow<N> synth_row(const double *__restrict__ row)
{
    SynthRow<N> r;
    const typename SynthRow<N>::vec *q = reinterpret_cast<const typename SynthRow<N>::vec *>(row);
#pragma unroll
    for (int k = 0; k < (N >= 8 ? N / 8 : 1); k++) r.c[k] = q[k];
    return r;
}
template <int N>
__device__ __forceinline__ double synth_dot(const double (&d)[N], const SynthRow<N> &r)
{
    constexpr int V = N >= 8 ? 8 : N;
    double a = 0.0;
#pragma unroll
    for (int j = 0; j < N; j++) a = __builtin_fma(d[j], r.c[j / V][j % V], a);   // fused: one rounding per term instead of two
    return a;                                                                      // (the guard's bound covers either; half the instructions)
}

// F32 (MP3S_OPT_FLOAT_FAST): float32 output of the same sums, no guard and no fix-up -- the fast sums differ from the reference's
// by a few 1e-14 of the slot's magnitude, the format's own rounding is 6e-8 of the sample and the contract's tolerance 1e-5;
// a sample goes straight to its place (no tile: 64 floats per slot would double the kernel's LDS).
//
// synth_fast_tile: a workgroup's tile of TW * 64 slot lanes per channel (the first 15 rebuild V history, OUT of the others emit) from the
// lanes' rows Sv[32] on -- shared by k_dec_synth_fast (rows from S in HBM) and k_dec_fused (rows from the workgroup's own IMDCT in LDS).
//   lds_raw : SynthFastLds<TW, OUT>::BYTES of LDS that nothing else uses from this call's first barrier on (ex | otile)
//   amax_w  : TW * 2 doubles of LDS;  gmax_s: the largest G of the granules the tile reads, written in front of this call
template <int TW, int OUT>
struct SynthFastLds {
    static constexpr int TL_LANES = TW * 64, OROW = 33;
    // ex[2][2][4][TL_LANES] doubles | otile[(OUT + 1) * OROW] dwords (+ one row nobody reads: where the lanes that do not emit put their samples)
    static constexpr int EX_BYTES = 2 * 2 * 4 * TL_LANES * 8, OT_BYTES = (OUT + 1) * OROW * 4, BYTES = EX_BYTES + OT_BYTES;
};
template <int TW, bool F32, int OUT, bool DEEP = false>
__device__ __forceinline__ void synth_fast_tile(unsigned char *lds_raw, double *amax_w, const double *gmax_p, double (&Sv)[32], long T, long tile0,
                                                long t, int tl, int ch, int wave, int lane, bool valid, int lim, int nch, int n_halo,
                                                int16_t *__restrict__ pcm_out, double eps_scale, uint2 *__restrict__ fix_list,
                                                int32_t *__restrict__ fix_count)
{
    constexpr int TL_LANES = TW * 64, OROW = SynthFastLds<TW, OUT>::OROW, EX_BYTES = SynthFastLds<TW, OUT>::EX_BYTES;
    static_assert(OUT <= TL_LANES - 15, "the first 15 lanes of a tile only rebuild V history");
    double (*const ex)[2][4][TL_LANES] = reinterpret_cast<double (*)[2][4][TL_LANES]>(lds_raw);
    uint32_t *const otile = reinterpret_cast<uint32_t *>(lds_raw + EX_BYTES);
    // ---- the differences of the splitting, level by level: X[k] of odd k is a 16-term sum over d16, of k = 2 mod 4 an
    //      8-term sum over d8, k = 4 mod 8: d4, k = 8 mod 16: d2, and X[16] = (u2[0] - u2[1]) cos(pi/4), X[0] = u2[0] + u2[1]
    double d16[16], d8v[8], d4[4], d2[2], x0, x16;
    double asum = 0.0;
    {
#pragma unroll
        for (int j = 0; j < 32; j++) asum += fabs(Sv[j]);
        double u16[16], u8[8], u4[4], u2[2];
#pragma unroll
        for (int j = 0; j < 16; j++) { u16[j] = Sv[j] + Sv[31 - j]; d16[j] = Sv[j] - Sv[31 - j]; }
#pragma unroll
        for (int j = 0; j < 8; j++) { u8[j] = u16[j] + u16[15 - j]; d8v[j] = u16[j] - u16[15 - j]; }
#pragma unroll
        for (int j = 0; j < 4; j++) { u4[j] = u8[j] + u8[7 - j]; d4[j] = u8[j] - u8[7 - j]; }
#pragma unroll
        for (int j = 0; j < 2; j++) { u2[j] = u4[j] + u4[3 - j]; d2[j] = u4[j] - u4[3 - j]; }
        x0 = u2[0] + u2[1];
        x16 = (u2[0] - u2[1]) * c_tab.synth_fast[340];
    }
    // ---- the largest sum |S| of a slot in this tile (both channels): the scale of the guard
    {
        const double a = wave_max_bound_f64(asum);
        if (lane == 0) amax_w[wave] = a;
    }
    __syncthreads();
    double amax = amax_w[0];
#pragma unroll
    for (int w = 1; w < TW * 2; w++) if (w < TW * nch) amax = amax > amax_w[w] ? amax : amax_w[w];
    // The guard's width for the whole tile, a scalar: synth_xbound * amax bounds every |sample * 32767| of the tile, so the term
    // that grows with the sample (eps_x |x|: 3 % of the other at most) is taken at that bound instead of per sample; a tile
    // whose bound leaves int32 (noise from a damaged file; NaN) goes to the exact path as a whole.
    const double xb_v = c_tab.synth_xbound * amax;
    const double eg_v = (c_tab.synth_eps_a * amax + c_tab.synth_eps_g * *gmax_p + c_tab.synth_eps_x * xb_v) * eps_scale;
    const double eps_t = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(eg_v)), __builtin_amdgcn_readfirstlane(__double2loint(eg_v)));
    const bool safe = __builtin_amdgcn_readfirstlane((int)(xb_v < 2147483000.0)) != 0;
    const long halo_slots = (long)n_halo * 36;
    const bool emit = valid && tl >= 15 && tl < 15 + OUT && t >= halo_slots;
    const bool full_hist = __ballot(tl >= 15 && tl < 15 + OUT && lim < 15) == 0;
    uint16_t *ot16 = reinterpret_cast<uint16_t *>(otile);
    unsigned long long dmask = 0;                               // lanes with a sample the guard cannot vouch for
    int p = 0;
    // Outputs 2t, 32 - 2t, 2t + 1 and 31 - 2t per barrier interval (t = 0: outputs 0, 16, 1 and 31).  Outputs i and 32 - i read
    // the same two X of every slot: V[i] = X[16+i], V[32+i] = -X[16-i] for i <= 16, and V[32-i] = -X[16+i], V[64-i] = -X[16-i]
    // -- so an interval computes A0 = X[16+2t], B0 = X[16-2t] (8 / 4 / 2 terms, nothing to multiply for t = 0), A1 = X[17+2t],
    // B1 = X[15-2t] (16 terms each) ONCE, exchanges (A0, -B0, A1, -B1), and every set of sixteen V values read back from the
    // other lanes feeds two window sums; the signs of the mirrored output live in its taps (DevTables::synth_window_f:
    // (-a) * w and a * (-w) are the same double, so the sums are what round 3's kernel, which computed every X twice, got).
    // 853 instead of 1 192 multiply-adds and 256 instead of 512 LDS reads per slot, 8 instead of 16 barriers.
    // Stages of an interval: odd-k sums A1, B1, even-k sums A0, B0, barrier, four window sums; each stage requests the scalar
    // operands of the next one before it computes.
    // ---- the window sums.  Two rules shape this loop:
    //  * Scalar loads (matrix rows, window taps) and LDS reads (the V values of 16 slots) share ONE counter, and while both
    //    kinds are in flight it can only be waited to zero.  So no scalar load is in flight while a window's LDS reads are
    //    waited for: a stage waits for everything asked for so far, THEN asks for the operands of the stage behind it, then
    //    computes (round 3 asked first; every wait of the compiler's then landed behind the request and took its whole latency).
    //  * No branch per output: every lane sums (lanes 0..14 of the tile, which only rebuild history, from the slots of lane 15),
    //    the guard is one comparison into a wave mask, and a lane that does not emit writes its sample to a row of the tile
    //    nobody reads -- round 3's four exec-mask branches per output cost more instructions than its 16 multiply-adds.
    const int tlc = tl >= 15 ? tl : 15;
    const int orow = emit ? tl - 15 : OUT;
    uint16_t *const oslot = ot16 + orow * OROW * 2 + ch;       // a slot's row: [output][channel], two channels wide also for mono
    // the V values of taps 8 half .. 8 half + 7: even taps from exchange slot se, odd taps from slot so; eight at a time (16
    // registers; all sixteen at once put the kernel at 123 VGPRs, one SIMD's whole register file for its four waves, with no
    // room for a wave of the neighbouring batches' kernels)
    const double *const exb = &ex[0][ch][0][tlc - 15];
    auto window_read = [&](int se, int so, int half, double (&u)[8]) {
        // (one base register, fifteen slots back: every read is base + a non-negative immediate)
        const double *e0 = exb + (p * 8 + se) * TL_LANES, *e1 = exb + (p * 8 + so) * TL_LANES;
#pragma unroll
        for (int j = 0; j < 8; j++) { const int jj = 8 * half + j; u[j] = (jj & 1 ? e1 : e0)[15 - jj]; }
        if (!full_hist) {                                   // (wave-uniform, rare: a stream starts inside the tile)
            int lim_here = lim;
            asm volatile("" : "+v"(lim_here));              // (compared HERE: sixteen lane masks kept across the loop were scalar spills)
#pragma unroll
            for (int j = 0; j < 8; j++) if (8 * half + j > lim_here) u[j] = 0.0;      // before the stream started the fifo holds zeros
        }
    };
    float *const fslot = reinterpret_cast<float *>(pcm_out) + ((valid ? t - (long)n_halo * 36 : 0) * 32) * nch + ch;
    auto window_emit = [&](int io, double sum) {
        if (F32) {
            if (emit) fslot[io * nch] = (float)sum;
            return;
        }
        // the guard: is the truncation of x = sample * 32767 (the taps carry the factor) beyond doubt?  Truncation is toward
        // zero: every x in (-1, 1) gives 0, so the integer 0 is not a boundary -- the distance is taken to the nearest integer
        // that is not zero.  One comparison into the wave's mask (round 3: per sample its own bit, thirteen instructions; the
        // fix-up kernel recomputes the slot's 32 samples, which costs it nothing: they are its lanes)
        const double x = sum, xi = rint(x);
        const double dist = fabs(x) - fmax(fabs(xi), 1.0);
        dmask |= __ballot(!(fabs(dist) > eps_t));
        asm volatile("" : "+s"(dmask));                   // (merged HERE: the compiler kept the 32 masks for a tree behind the loop, as scalar spills)
        oslot[io * 2] = (uint16_t)(int)x;                   // (beyond int32 only in a tile that is not `safe`: recomputed as a whole)
    };
    // The loop's constants come as ONE stream in the order of their use (DevTables::synth_stream), sixteen doubles per step behind
    // one base address the compiler cannot see through: a scalar load is base + immediate (it rebuilt a pc-relative address, three
    // scalar instructions, in front of each), and a step is: wait for the piece asked for a step earlier (and for the step's LDS
    // reads), ask for the next piece, sixteen multiply-adds -- two pieces live at any time, 64 scalar registers.  (Round 3 kept
    // whole rows, 112 scalar registers at the peak, 65 of them spilled into vector lanes; pieces of eight doubles, until r04c,
    // waited twice as often: the kernel issued a third of the vector instructions its SIMDs had slots for.)
    typedef const double __attribute__((address_space(4))) *ctab_ptr;   // (the constant segment: what makes the loads scalar ones)
    ctab_ptr ct = (ctab_ptr)&c_tab.synth_stream[F32 ? 0 : 1][0][0];
    asm volatile("" : "+s"(ct));
    typedef double d8 __attribute__((ext_vector_type(8)));
    struct Piece { d8 lo, hi; };
    auto ld16 = [](ctab_ptr q) { Piece r; r.lo = *(const d8 __attribute__((address_space(4))) *)q; r.hi = *(const d8 __attribute__((address_space(4))) *)(q + 8); return r; };
    auto dot8 = [](const double *d, const d8 &c, double a) {
#pragma unroll
        for (int j = 0; j < 8; j++) a = __builtin_fma(d[j], c[j], a);
        return a;
    };
#define MP3S_ARRIVED() do { __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_sched_barrier(0); } while (0)
#define MP3S_GO() __builtin_amdgcn_sched_barrier(0)
    Piece cur = ld16(ct);                                           // interval t = 0: the row of X[17]
#pragma unroll
    for (int tt = 0; tt < 8; tt++) {
        // odd k: A1 = X[17+2t], B1 = X[15-2t].  even k: A0 = X[16+2t], B0 = X[16-2t]
        ctab_ptr q = ct + tt * 112;
        const int oa = 2 * tt, ob = tt ? 32 - 2 * tt : 16, oc = 2 * tt + 1, od = 31 - 2 * tt;   // outputs of the interval
        Piece nxt;
        // ---- the odd-k sums
        MP3S_ARRIVED(); nxt = ld16(q + 16); MP3S_GO();
        double va1 = dot8(d16 + 8, cur.hi, dot8(d16, cur.lo, 0.0)); MP3S_GO();
        cur = nxt;
        MP3S_ARRIVED(); nxt = ld16(q + (tt ? 32 : 48)); MP3S_GO();
        double vb1 = -dot8(d16 + 8, cur.hi, dot8(d16, cur.lo, 0.0)); MP3S_GO();
        cur = nxt;
        // ---- the even-k sums (8 / 4 / 2 terms; t = 0: A0 = B0 = X[16] for output 0, X[32] = 0 and X[0] for output 16)
        double va0, vb0;
        if (tt == 0) { va0 = x16; vb0 = -x0; }
        else {
            MP3S_ARRIVED(); nxt = ld16(q + 48); MP3S_GO();
            if (tt & 1) { va0 = dot8(d8v, cur.lo, 0.0); vb0 = -dot8(d8v, cur.hi, 0.0); }
            else {
                va0 = 0.0; vb0 = 0.0;
#pragma unroll
                for (int j = 0; j < ((tt & 2) ? 4 : 2); j++) {
                    va0 = __builtin_fma((tt & 2) ? d4[j] : d2[j], cur.lo[j], va0);
                    vb0 = __builtin_fma((tt & 2) ? d4[j] : d2[j], cur.hi[j], vb0);
                }
                vb0 = -vb0;
            }
            MP3S_GO();
            cur = nxt;
        }
        ex[p][ch][0][tl] = va0;
        ex[p][ch][1][tl] = vb0;
        ex[p][ch][2][tl] = va1;
        ex[p][ch][3][tl] = vb1;
        __syncthreads();                                        // (its wait covers the taps asked for above)
        if constexpr (DEEP) {
            // (k_dec_fused: two waves per SIMD, nobody to run under a wait.)  The V values of a step are asked for a step AHEAD, with
            // the next piece of constants: what a step's wait at its top is for has had the step in front of it, sixteen
            // multiply-adds, to arrive -- only the first read of an interval, which cannot go out in front of the barrier, is waited
            // for in full.  Two sets of eight V values, 32 registers (this kernel has them).
            double ua[8], ub[8];
            if (tt == 0) {
                window_read(0, 0, 0, ua);
                MP3S_ARRIVED(); nxt = ld16(q + 64); window_read(0, 0, 1, ub); MP3S_GO();
                double sum = dot8(ua, cur.lo, 0.0); MP3S_GO();
                MP3S_ARRIVED(); window_read(1, 1, 0, ua); MP3S_GO();
                sum = dot8(ub, cur.hi, sum);
                window_emit(oa, sum); MP3S_GO();
                cur = nxt;
                MP3S_ARRIVED(); nxt = ld16(q + 80); window_read(1, 1, 1, ub); MP3S_GO();
                sum = dot8(ua, cur.lo, 0.0); MP3S_GO();
                MP3S_ARRIVED(); window_read(2, 3, 0, ua); MP3S_GO();
                sum = dot8(ub, cur.hi, sum);
                window_emit(ob, sum); MP3S_GO();
                cur = nxt;
            } else {
                window_read(0, 1, 0, ua);
                MP3S_ARRIVED(); nxt = ld16(q + 64); window_read(0, 1, 1, ub); MP3S_GO();
                double sa = dot8(ua, cur.lo, 0.0), sb = dot8(ua, cur.hi, 0.0); MP3S_GO();
                cur = nxt;
                MP3S_ARRIVED(); nxt = ld16(q + 80); window_read(2, 3, 0, ua); MP3S_GO();
                sa = dot8(ub, cur.lo, sa);
                window_emit(oa, sa);
                sb = dot8(ub, cur.hi, sb);
                window_emit(ob, sb); MP3S_GO();
                cur = nxt;
            }
            MP3S_ARRIVED(); nxt = ld16(q + 96); window_read(2, 3, 1, ub); MP3S_GO();
            double sc = dot8(ua, cur.lo, 0.0), sd = dot8(ua, cur.hi, 0.0); MP3S_GO();
            cur = nxt;
            MP3S_ARRIVED(); nxt = ld16(ct + (tt == 7 ? 0 : tt + 1) * 112); MP3S_GO();
            sc = dot8(ub, cur.lo, sc);
            window_emit(oc, sc);
            sd = dot8(ub, cur.hi, sd);
            window_emit(od, sd); MP3S_GO();
            cur = nxt;
        } else {
        double u[8];
        // ---- eight taps of two outputs at a time: V values from LDS, wait (for them and for the piece asked for a step earlier), ask
        //      for the next piece, multiply
        if (tt == 0) {
            // output 0: every tap reads X[16] (slot 0); output 16: X[0] (slot 1) under taps that are zero where V[16] stands
            window_read(0, 0, 0, u);
            MP3S_ARRIVED(); nxt = ld16(q + 64); MP3S_GO();
            double sum = dot8(u, cur.lo, 0.0); MP3S_GO();
            window_read(0, 0, 1, u);
            MP3S_ARRIVED(); MP3S_GO();                          // (the next piece is in flight under this wait: one interval in eight)
            sum = dot8(u, cur.hi, sum);
            window_emit(oa, sum); MP3S_GO();
            cur = nxt;
            window_read(1, 1, 0, u);
            MP3S_ARRIVED(); nxt = ld16(q + 80); MP3S_GO();
            sum = dot8(u, cur.lo, 0.0); MP3S_GO();
            window_read(1, 1, 1, u);
            MP3S_ARRIVED(); MP3S_GO();
            sum = dot8(u, cur.hi, sum);
            window_emit(ob, sum); MP3S_GO();
            cur = nxt;
        } else {
            window_read(0, 1, 0, u);
            MP3S_ARRIVED(); nxt = ld16(q + 64); MP3S_GO();
            double sa = dot8(u, cur.lo, 0.0), sb = dot8(u, cur.hi, 0.0); MP3S_GO();
            cur = nxt;
            window_read(0, 1, 1, u);
            MP3S_ARRIVED(); nxt = ld16(q + 80); MP3S_GO();
            sa = dot8(u, cur.lo, sa);
            window_emit(oa, sa);
            sb = dot8(u, cur.hi, sb);
            window_emit(ob, sb); MP3S_GO();
            cur = nxt;
        }
        // ---- outputs 2t + 1 and 31 - 2t; under the last piece the first piece of the next interval (behind the last one: any row)
        {
            window_read(2, 3, 0, u);
            MP3S_ARRIVED(); nxt = ld16(q + 96); MP3S_GO();
            double sc = dot8(u, cur.lo, 0.0), sd = dot8(u, cur.hi, 0.0); MP3S_GO();
            cur = nxt;
            window_read(2, 3, 1, u);
            MP3S_ARRIVED(); nxt = ld16(ct + (tt == 7 ? 0 : tt + 1) * 112); MP3S_GO();
            sc = dot8(u, cur.lo, sc);
            window_emit(oc, sc);
            sd = dot8(u, cur.hi, sd);
            window_emit(od, sd); MP3S_GO();
            cur = nxt;
        }
        }
        p ^= 1;
    }
#undef MP3S_ARRIVED
#undef MP3S_GO
    // ---- the samples the guard could not vouch for go on the fix-up list: slot | channel << 31, mask of output indices
    //      (k_dec_fixup recomputes them from `is` in the reference's order and overwrites what is stored below)
    if (F32) return;
    {
        const unsigned long long em = __ballot(emit);
        const unsigned long long mine = safe ? dmask & em : em;
        if ((mine >> lane) & 1ull) {
            const int at = atomicAdd(fix_count, 1);
            fix_list[at] = make_uint2((uint32_t)t | ((uint32_t)ch << 31), 0xffffffffu);
        }
    }
    __syncthreads();
    {
        // the tile's slots are one contiguous piece of the output: dword c of the tile goes to dword c of the piece (round 3
        // divided by the run-time row length and compared 64-bit slot numbers per dword: a third of the kernel's vector instructions)
        const int sh = nch == 2 ? 5 : 4;                                     // log2 of a slot's dwords
        const long lo_l = halo_slots - tile0, hi_l = T - tile0;             // (hi_l > 0: the tile starts inside the batch)
        const int sl_lo = lo_l > 0 ? (lo_l < OUT ? (int)lo_l : OUT) : 0, sl_hi = hi_l < OUT ? (int)hi_l : OUT;
        uint32_t *const outp = (uint32_t *)pcm_out;
        const long obase = (tile0 - halo_slots) << sh;
        for (int c = (sl_lo << sh) + (int)threadIdx.x; c < (sl_hi << sh); c += (int)blockDim.x) {
            const int sl = c >> sh, w = c & ((1 << sh) - 1);
            uint32_t v;
            if (nch == 2) v = otile[sl * OROW + w];
            else v = (otile[sl * OROW + 2 * w] & 0xffffu) | (otile[sl * OROW + 2 * w + 1] << 16);   // mono: every other half word of the row
            outp[obase + c] = v;
        }
    }
}

template <int TW, bool F32 = false>
__global__ __launch_bounds__(TW * 64 * 2, TW == 4 ? 2 : 4) void k_dec_synth_fast(
    const double *__restrict__ S, long T, const mp3s_frame_hdr *__restrict__ hdr, int nch, int n_halo,
    int16_t *__restrict__ pcm_out, int sf_base, double eps_scale, const double *__restrict__ G, int n_granules,
    uint2 *__restrict__ fix_list, int32_t *__restrict__ fix_count)
{
    constexpr int TL_LANES = TW * 64, OUT = TL_LANES - 15;
    // in front of the tile's loop its LDS stages the waves' rows of S (synth_fetch_rows)
    static_assert(TW * 2 * SYNTH_STAGE_WAVE <= SynthFastLds<TW, OUT>::BYTES, "the staging of S fits the loop's LDS");
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[SynthFastLds<TW, OUT>::BYTES];
    __shared__ double amax_w[TW * 2];
    __shared__ double gmax_s;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int ch = wave / TW, tl = (wave % TW) * 64 + lane;
    const long tile0 = (long)xcd_tile() * OUT;
    if (wave == 0) {
        // the largest G (sum of |IMDCT input| of a granule and channel, left behind by the fast IMDCT) among the granules
        // whose rows this tile reads and the granule in front of them (its overlap tail is part of their rows)
        const long tlo = tile0 - 15 > 0 ? tile0 - 15 : 0, thi = tile0 + OUT - 1 < T - 1 ? tile0 + OUT - 1 : T - 1;
        const int ga = (int)(tlo / 18) > 0 ? (int)(tlo / 18) - 1 : 0, gb = (int)(thi / 18);
        static_assert((TW * 64 + 17) / 18 + 2 <= 32, "one lane pair per granule of the tile");
        const int g = ga + (lane >> 1), c = lane & 1;
        double gm = 0.0;
        if (G && g <= gb && g < n_granules && c < nch) gm = G[(long)g * 2 + c];
        gm = wave_max_bound_f64(gm);
        if (lane == 0) gmax_s = gm;
    }
    const long t = tile0 - 15 + tl;
    const bool valid = t >= 0 && t < T;
    int lim = -1;
    if (valid) {
        const uint32_t sf = hdr[t / 36].stream_first;
        const long s0 = sf > (uint32_t)sf_base ? (long)(sf - (uint32_t)sf_base) * 36 : 0;
        lim = (int)((t - s0) < 64 ? (t - s0) : 64);
    }
    double Sv[32];
    synth_fetch_rows(lds_raw + wave * SYNTH_STAGE_WAVE, S, T, ch, t, lane, Sv);
    synth_fast_tile<TW, F32, OUT>(lds_raw, amax_w, &gmax_s, Sv, T, tile0, t, tl, ch, wave, lane, valid, lim, nch, n_halo, pcm_out, eps_scale,
                                  fix_list, fix_count);
}


// ---------------------------------------------------------------------------------------------
// Fix-up of the fast int16 path.  Entry = a time slot of one channel + the mask of its output samples whose truncation
// the guard of k_dec_synth_fast could not vouch for.  One wave per entry: the rows of the 16 slots the samples read are
// computed again from `is` in the reference's order (imdct_run<false>, one or two granules primed with the granule in
// front of them) into a window in LDS, then lane j computes the j-th flagged sample exactly as k_dec_synth would and
// overwrites it in the PCM.  About one sample in a million comes here; with the guard inflated (tests) all of them do.
// counters: [0] entries (written by k_dec_synth_fast), [1] workgroups through; the last one clears both.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(DEC_A_WAVES * 64, 1) void k_dec_fixup(
    const int16_t *__restrict__ is, const mp3s_granule_si *__restrict__ si, const mp3s_frame_hdr *__restrict__ hdr,
    int n_granules, int nch, long T, int n_halo, int sf_base, int16_t *__restrict__ pcm_out,
    const uint2 *__restrict__ fix_list, int32_t *counters, int32_t *__restrict__ n_exact)
{
    __shared__ DecShared sh;
    __shared__ double win[DEC_A_WAVES][36 * 32];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int n = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&counters[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    if ((int)(blockIdx.x * DEC_A_WAVES) < n) dec_stage_tables(sh);     // (uniform per workgroup; most launches find an empty or short list)
    const long halo_slots = (long)n_halo * 36;
    double *Sp = win[wave];
    for (int e = blockIdx.x * DEC_A_WAVES + wave; e < n; e += gridDim.x * DEC_A_WAVES) {
        const uint2 en = fix_list[e];
        const long t = (long)(en.x & 0x7fffffffu);
        const int ch = (int)(en.x >> 31);
        const uint32_t mask = en.y;
        if (t >= T || ch >= nch) continue;                                  // (never: the list is the synthesis kernel's)
        const uint32_t sf = hdr[t / 36].stream_first;
        const long s0 = sf > (uint32_t)sf_base ? (long)(sf - (uint32_t)sf_base) * 36 : 0;
        const int lim = (int)((t - s0) < 64 ? (t - s0) : 64);
        const long tlo = t - 15 > s0 ? t - 15 : s0;
        const int ga = (int)(tlo / 18), gb = (int)(t / 18);
        __builtin_amdgcn_wave_barrier();
        imdct_run<false>(sh, wave, lane, is, si, hdr, n_granules, nch, ga, gb - ga + 1, Sp, 0, (long)ga * 18, sf_base, nullptr, ch);
        __builtin_amdgcn_s_waitcnt(0);                                      // the wave's rows are in LDS
        __builtin_amdgcn_wave_barrier();
        // lane j takes the j-th flagged sample
        int i = -1;
        {
            uint32_t m = mask;
            for (int j = 0; j < 32; j++) {
                if (!m) break;
                const int b = __builtin_ctz(m);
                if (j == lane) i = b;
                m &= m - 1;
            }
        }
        if (i >= 0 && t >= halo_slots) {
            const double v = synth_exact_sample(Sp, 0, 0, t - (long)ga * 18, i, lim);
            pcm_out[((t - halo_slots) * 32 + i) * nch + ch] = pcm_to_i16(v);
        }
        if (lane == 0 && n_exact) atomicAdd(n_exact, __popc(mask));
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(&counters[1], 1) == (int)gridDim.x - 1) {
        atomicExch(&counters[0], 0);
        atomicExch(&counters[1], 0);
    }
}

}  // namespace mp3s
