// Rate-loop kernel (gfx950).  Included by mp3s_device.hip only.
//
// One wavefront per granule*channel ("unit").  The 288 value pairs of the granule are spread over the
// lanes (lane l holds the consecutive pairs 5l .. 5l+4); every step of the reference's loop body is a
// lane-parallel map plus DPP wave reductions (independent ones written side by side), and all control
// flow (binary search, inner loop, region split, table choice, hide swap) is wave-uniform scalar code.
//   reference encoder/MP3_Encoder.py: __iteration_loop :760-815, __calc_scfsi energies :835-859,
//   __bin_search_step_size :958-996, __inner_loop :1064-1095, quantize :373-415, calc_run_len :266-291,
//   count1_bit_count :171-211, __subdivide :998-1036, __big_v_tab_select :1147-1168,
//   __new_choose_table :1170-1264, count_bit :214-263, big_v_bit_count :294-318.
// Integer work only (bit exact); the one float spot is quantize's ln >= 10000 path (correctly rounded fp64 sqrt).  The
// scfsi log is a table built with the host's libm (DevTables::en_base / en_step).
#pragma once
#ifndef MP3S_RL_STATS
#define MP3S_RL_STATS 0   // 1: what every probe of k_rate_loop ends on and the shader clocks of a wave's phases, summed into g_rl_stats and read by
                          // mp3s_debug_rl_stats (a probe: tools/rl_stats.py; such a build is not the product's)
#endif
#ifndef MP3S_RL_OCC
#define MP3S_RL_OCC 6     // waves per SIMD k_rate_loop is compiled for (r06: 80 VGPRs, no scratch; 0.214 -> 0.204 ms in the step against five waves and 96 VGPRs)
#endif
#if MP3S_RL_STATS
__device__ unsigned long long g_rl_stats[128];
extern "C" __attribute__((visibility("default"))) int mp3s_debug_rl_stats(unsigned long long *out, int clear)
{
    (void)hipDeviceSynchronize();
    int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rl_stats), sizeof(unsigned long long) * 128);
    if (clear) { unsigned long long z[128] = {}; rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_rl_stats), z, sizeof z); }
    return rc;
}
#define RL_STAT(i, v) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_rl_stats[(i)], (unsigned long long)(v)); } while (0)
#define RL_CLK() __builtin_readcyclecounter()
#else
#define RL_STAT(i, v) do { } while (0)
#endif

namespace mp3s {

// one DPP step of a wave reduction: lanes without a source read 0 (ctrl and row mask must be literals)
#define RL_DPP(v, ctrl, rm) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), ctrl, rm, 0xf, false))

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
    v = max(v, RL_DPP(v, 0x111, 0xf));  // row_shr:1
    v = max(v, RL_DPP(v, 0x112, 0xf));  // row_shr:2
    v = max(v, RL_DPP(v, 0x114, 0xf));  // row_shr:4
    v = max(v, RL_DPP(v, 0x118, 0xf));  // row_shr:8
    v = max(v, RL_DPP(v, 0x142, 0xa));  // row_bcast:15
    v = max(v, RL_DPP(v, 0x143, 0xc));  // row_bcast:31
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t wave_add_u32(uint32_t v)
{
    v += RL_DPP(v, 0x111, 0xf);
    v += RL_DPP(v, 0x112, 0xf);
    v += RL_DPP(v, 0x114, 0xf);
    v += RL_DPP(v, 0x118, 0xf);
    v += RL_DPP(v, 0x142, 0xa);
    v += RL_DPP(v, 0x143, 0xc);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// inclusive prefix sums over the lanes, the same six DPP steps: within the rows of 16 by shifts of 1, 2, 4, 8, then the last lane of
// rows 0 and 2 into rows 1 and 3, then lane 31 into rows 2 and 3 (six additions; through __shfl_up a step was an LDS permute, a compare,
// a select and an addition with the address arithmetic of the permute on top)
__device__ __forceinline__ uint32_t wave_scan_u32(uint32_t v)
{
    v += RL_DPP(v, 0x111, 0xf);
    v += RL_DPP(v, 0x112, 0xf);
    v += RL_DPP(v, 0x114, 0xf);
    v += RL_DPP(v, 0x118, 0xf);
    v += RL_DPP(v, 0x142, 0xa);
    v += RL_DPP(v, 0x143, 0xc);
    return v;
}
// Several reductions at once: a DPP step has to wait two issue slots for the VALU write in front of it, so independent
// chains are written step by step side by side -- the second and third fill the slots the first would idle in.
__device__ __forceinline__ void wave_add2(uint32_t &a, uint32_t &b)
{
#define RL_STEP(ctrl, rm) { const uint32_t ta = RL_DPP(a, ctrl, rm), tb = RL_DPP(b, ctrl, rm); a += ta; b += tb; }
    RL_STEP(0x111, 0xf) RL_STEP(0x112, 0xf) RL_STEP(0x114, 0xf) RL_STEP(0x118, 0xf) RL_STEP(0x142, 0xa) RL_STEP(0x143, 0xc)
#undef RL_STEP
    a = (uint32_t)__builtin_amdgcn_readlane((int)a, 63);
    b = (uint32_t)__builtin_amdgcn_readlane((int)b, 63);
}
__device__ __forceinline__ void wave_add3(uint32_t &a, uint32_t &b, uint32_t &c)
{
#define RL_STEP(ctrl, rm) { const uint32_t ta = RL_DPP(a, ctrl, rm), tb = RL_DPP(b, ctrl, rm), tc = RL_DPP(c, ctrl, rm); a += ta; b += tb; c += tc; }
    RL_STEP(0x111, 0xf) RL_STEP(0x112, 0xf) RL_STEP(0x114, 0xf) RL_STEP(0x118, 0xf) RL_STEP(0x142, 0xa) RL_STEP(0x143, 0xc)
#undef RL_STEP
    a = (uint32_t)__builtin_amdgcn_readlane((int)a, 63);
    b = (uint32_t)__builtin_amdgcn_readlane((int)b, 63);
    c = (uint32_t)__builtin_amdgcn_readlane((int)c, 63);
}
__device__ __forceinline__ void wave_max3(uint32_t &a, uint32_t &b, uint32_t &c)
{
#define RL_STEP(ctrl, rm) { const uint32_t ta = RL_DPP(a, ctrl, rm), tb = RL_DPP(b, ctrl, rm), tc = RL_DPP(c, ctrl, rm); a = max(a, ta); b = max(b, tb); c = max(c, tc); }
    RL_STEP(0x111, 0xf) RL_STEP(0x112, 0xf) RL_STEP(0x114, 0xf) RL_STEP(0x118, 0xf) RL_STEP(0x142, 0xa) RL_STEP(0x143, 0xc)
#undef RL_STEP
    a = (uint32_t)__builtin_amdgcn_readlane((int)a, 63);
    b = (uint32_t)__builtin_amdgcn_readlane((int)b, 63);
    c = (uint32_t)__builtin_amdgcn_readlane((int)c, 63);
}

// encoder/util.py:130-133 mulr for non-negative a: (a*b + 2^31) >> 32
__device__ __forceinline__ uint32_t mulr_u(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b + 0x80000000ull) >> 32); }   // one v_mad_u64_u32

// the device's list of units to run again (written by k_chain_apply, read by k_rate_redo), int32: [0] = entries wanted,
// then from REDO_HEAD: unit[REDO_CAP] (-1 behind the end), cursor[REDO_CAP], inherited state[REDO_CAP][4]
constexpr int REDO_CAP = 4096, REDO_HEAD = 4, REDO_WORDS = REDO_HEAD + 6 * REDO_CAP;
constexpr int REDO_GROUPS = 256;          // workgroups of k_rate_redo (persistent: a group takes entries g, g + REDO_GROUPS * RL_WAVES, ...)
constexpr int REDO_CHAIN_MAX = 64;        // units a re-run may follow its chain through (the units behind it that inherit what it leaves)
constexpr int RL_WAVES = 4;   // 4 waves share one copy of the lookup tables: 33 KB LDS -> 4 workgroups = 16 waves per CU (a fifth wave
                              // per SIMD at 96 VGPRs was measured: 0.351 instead of 0.305 ms, and the Huffman kernel no longer fits beside it)
constexpr int RL_NP = 5;      // pairs per lane: lane l holds the CONSECUTIVE pairs 5l .. 5l+4 (lines 10l .. 10l+9); lanes 0..57 hold
                              // the granule's 288 pairs (lane 57 three of them), what lies beyond is zero throughout.  (Until r02c
                              // pair p sat in lane p % 64: the highest non-zero pair then was a chain of ten ballots and ~90 scalar
                              // selects per evaluation, and the neighbour pair of a count1 quad came through LDS, one round trip per
                              // slot.  Now the run lengths are two ballots and two readlanes, the neighbour is the next register.)

struct RlTables {
    uint16_t int2idx[10000];
    uint2 hl[256];          // .x per pair (min(x,15), min(y,15)): code length in books 13 | 15 << 5 | 16.. << 10 | 24.. << 15,
                            // | number of non-zero values << 20 | number of values > 14 << 22
                            // | (shortest of the four lengths + non-zero values, 0 for the pair (0,0)) << 25
                            // .y bounds of the pair's bits under any candidate book: the shortest field once more
                            // | (longest of the four lengths + non-zero values + 13 bits per value > 14) << 16
    uint32_t c1w[16];       // count1 book A: code length of the quad | number of ones in its first pair << 16
    uint8_t transform[64];  // [table][bit]
    uint32_t subdiv[292];   // __subdivide result per big_values (0..288) for this workgroup's sample rate (see DevTables); padded to 16-byte pieces
};

// linbits of table t (encoder/tables.py:287-302) without a memory lookup: nibble k of the packed constants
__device__ __forceinline__ int lin_bits_of(int t)
{
    return t < 16 ? 0 : (int)(((t < 24 ? 0xDA864321u : 0xDB987654u) >> (4 * (t & 7))) & 15u);
}

struct RlState {           // wave-uniform GrInfo fields that live across rate_body calls
    int big_values, count1, c1sel, r0c, r1c, a1, a2, a3, ts0, ts1, ts2;
    bool addr_fresh, used_addr_in;
};

__device__ __forceinline__ int family_of(int t) { return t == 13 ? 0 : (t == 15 ? 1 : (t < 24 ? 2 : 3)); }

// bits of one pair (count_bit, MP3_Encoder.py:234-261) under a table given by its Huffman-length family (0..3 =
// 13, 15, 16.., 24..) and its linbits; both are wave-uniform per region and selected per lane without branches
__device__ __forceinline__ uint32_t pair_bits(const RlTables &tb, int fam, int lb, int x, int y)
{
    const int xx = x > 14 ? 15 : x, yy = y > 14 ? 15 : y;
    const uint32_t h = tb.hl[xx * 16 + yy].x;
    return ((h >> (5 * fam)) & 31u) + ((h >> 20) & 3u) + (uint32_t)lb * ((h >> 22) & 3u);
}
__device__ __forceinline__ int sel3(int r, int a, int b, int c) { return r == 0 ? a : (r == 1 ? b : c); }

// steptabi[step + 127]: the binary search asks for the scale of both steps it may probe next while it still works on
// the current one (a scalar load takes longer than a wave has other work between probes).  The index is wrapped into the
// table; rl_quantize rejects a step that lies outside it before the scale is used.
__device__ __forceinline__ uint32_t rl_scale_of(int step)
{
    return (uint32_t)c_tab.steptabi[(step + 127) & 127];
}

// quantize (MP3_Encoder.py:389-415); returns 0, or 8193 when some quantised value exceeds 8192, 16384 for the early out,
// -1 when the step leaves steptab (IndexError in the reference).  (The callers only ask "more than 8192?".)
// mulr is monotone in its first argument, so the largest ln of the granule is the one of xrmax: whether the float path
// is needed at all is a scalar question, and the table path cannot exceed 1000 -- no wave reduction in the common case.
__device__ __forceinline__ int rl_quantize(const RlTables &tb, const uint32_t (&xa)[2 * RL_NP], int32_t (&ix)[2 * RL_NP],
                                           int step, uint32_t scalei, uint32_t xrmax)
{
    const int idx = step + 127;
    if (idx < 0 || idx > 127) return -1;
    const uint32_t lnmax = mulr_u(xrmax, scalei);
    if (lnmax > 165140u) return 16384;
    if (lnmax < 10000u) {
#pragma unroll
        for (int e = 0; e < 2 * RL_NP; e++) ix[e] = tb.int2idx[mulr_u(xa[e], scalei)];   // quick lookup (:403-404)
        return 0;
    }
    // some value is outside the table range: those go through floats (:405-409)
    const double scale = c_tab.steptab[idx];
    bool big = false;
#pragma unroll
    for (int e = 0; e < 2 * RL_NP; e++) {
        uint32_t a = xa[e];
        asm volatile("" : "+v"(a));        // (keeps the ten conversions to double out of the loop's live registers: this path is rare)
        const uint32_t ln = mulr_u(a, scalei);
        const double dbl = (double)a * scale * 4.656612875e-10;
        const int32_t v = (int32_t)__dsqrt_rn(__dsqrt_rn(dbl) * dbl);
        ix[e] = ln >= 10000u ? v : (int32_t)tb.int2idx[ln < 10000u ? ln : 9999u];
        big |= ix[e] > 8192;
    }
    return __ballot(big) ? 8193 : 0;
}

// calc_run_len (MP3_Encoder.py:266-291) and __subdivide (:998-1036) from what every lane knows about its own pairs: k0 = its highest
// slot (+ 1) holding a non-zero value, k1 = its highest slot (+ 1) holding a value > 1.  Everything with a side effect that the reference's
// loop body has in front of its bit counts: count1, big_values, region counts and addresses (left untouched when big_values == 0: E7).
__device__ __forceinline__ void rl_run_lengths(const RlTables &tb, int k0, int k1, RlState &st)
{
    // per lane the highest slot (+1), then the highest lane: two ballots and two readlanes
    const unsigned long long nzm = __ballot(k0 != 0), bgm = __ballot(k1 != 0);
    const int L0 = 63 - (nzm ? __builtin_clzll(nzm) : 0), L1 = 63 - (bgm ? __builtin_clzll(bgm) : 0);
    const int r0 = __builtin_amdgcn_readlane(k0, L0), r1 = __builtin_amdgcn_readlane(k1, L1);
    const int P0 = nzm ? 5 * L0 + r0 - 1 : -1, P1 = bgm ? 5 * L1 + r1 - 1 : -1;
    const int count1 = (P0 - P1) >> 1;
    const int bv = (P0 + 1) - 2 * count1;
    st.count1 = count1;
    st.big_values = bv;
    // ---- __subdivide (addresses are left untouched when big_values == 0: E7)
    if (bv == 0) {
        st.r0c = 0; st.r1c = 0;
        if (!st.addr_fresh) st.used_addr_in = true;
    } else {
        // __subdivide (:1008-1036) depends on big_values only: looked up in the per-rate table built on the host
        const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)tb.subdiv[bv < 289 ? bv : 288]);   // (wave-uniform: its fields are scalar arithmetic,
                                                                                                                 //  not four vector registers per kept evaluation)
        st.r0c = (int)(e & 15); st.r1c = (int)((e >> 4) & 7);
        st.a1 = (int)((e >> 8) & 1023); st.a2 = (int)((e >> 18) & 1023);
        st.a3 = 2 * bv;
        st.addr_fresh = true;
    }
}

// A probe of the binary search decided WITHOUT quantising it (MP3_Encoder.py:958-996 with :373-415, :266-291, :998-1036 inside).
// The early probes sit at steps far too fine for the budget: all the search wants from them is "the bits reach max_bits" and
// the side effects of the loop body.  quantize is monotone in |xr| (DevTables::rl_t1 / _t2 / _t8: the smallest |xr| of the step
// whose quantised value is >= 1, >= 2, > 8192), so which of its three ways out the probe takes and, if the body runs, its run
// lengths -- hence count1, big_values and the region addresses, exactly -- come from ten comparisons per lane.  The lower bound of
// its bits: a sign bit per non-zero value, at least one bit of code per count1 quadruple (both count1 books: :171-211) and per
// big-value pair THAT HOLDS A NON-ZERO VALUE -- the pair (0, 0) is free in a region whose maximum is 0 (__new_choose_table gives
// such a region table 0 and count_bit(0) is 0: :1182-1184, :228-229), exactly what rl_hl's `shortest` field says, so a pair of
// zeros below big_values counts nothing here.  When that sum alone reaches the limit the decision stands.  Returns the probe's
// `bit` (100000 as the caller sets it for quantize's two refusals, else that lower bound) or -1: undecided, the caller runs the
// probe in full (the side effects applied here are the ones it applies again).  q_early: quantize's early out (:394-395), which
// leaves the kept quantisation valid.
__device__ __forceinline__ int rl_precheck(const RlTables &tb, const uint32_t (&xa)[2 * RL_NP], int p0, RlState &st, int step, uint32_t scalei,
                                           uint32_t xrmax, int limit, bool &q_early)
{
    q_early = false;
    const int idx = step + 127;
    if (idx < 0 || idx > 127) return -1;
    if (mulr_u(xrmax, scalei) > 165140u) { q_early = true; return 100000; }
    const uint32_t t1 = (uint32_t)c_tab.rl_t1[idx], t2 = (uint32_t)c_tab.rl_t2[idx], t8 = (uint32_t)c_tab.rl_t8[idx];
    if (xrmax >= t8) return 100000;
    int k0 = 0, k1 = 0;
    uint32_t nnz = 0, nzp = 0;                 // non-zero values; bit m: the lane's pair m holds one
#pragma unroll
    for (int m = 0; m < RL_NP; m++) {
        const uint32_t pm = max(xa[2 * m], xa[2 * m + 1]);
        k0 = pm >= t1 ? m + 1 : k0;
        k1 = pm >= t2 ? m + 1 : k1;
        nzp |= pm >= t1 ? 1u << m : 0u;
        nnz += (xa[2 * m] >= t1 ? 1u : 0u) + (xa[2 * m + 1] >= t1 ? 1u : 0u);
    }
    rl_run_lengths(tb, k0, k1, st);
    // the lane's non-zero pairs below big_values (pairs p0 .. p0 + 4)
    const int below = min(max(st.big_values - p0, 0), RL_NP);
    nnz += (uint32_t)__builtin_popcount(nzp & ((1u << below) - 1u));
    const int lb = (int)wave_add_u32(nnz) + st.count1;
    return lb >= limit ? lb : -1;
}

// calc_run_len + count1_bit_count + __subdivide + __big_v_tab_select + big_v_bit_count
// `limit`: the caller only wants to know whether the bit count reaches it (binary search: max_bits; inner loop:
// max_bits + 1).  A lower bound -- exact count1 bits + per big-value pair the shortest code any candidate book has for it
// plus its sign bits -- and an upper bound (longest code, 13 linbits per escape) come out of one table word per pair and
// ONE pair of wave reductions; when the lower bound reaches the limit (most probes that do), or -- with `ub_ok`, for
// binary-search probes whose result is not reused -- the upper bound stays below it, the region maxima, candidate books
// and their bit sums are skipped and the bound is returned.  Everything with a side effect the reference's body has (run
// lengths, count1 table, __subdivide and the stale-address rule) happens before that point.
__device__ __forceinline__ int rl_body(const RlTables &tb, const int32_t (&ix)[2 * RL_NP], int p0,
                                       RlState &st, const uint8_t *__restrict__ hide, int n_hide, int cursor,
                                       int limit, bool ub_ok, bool &full)
{
    // ---- calc_run_len: highest non-zero pair, highest pair holding a value > 1.  Values are >= 0 here, so "some value
    //      of the pair > 1" is (x | y) > 1.
    int k0 = 0, k1 = 0;
#pragma unroll
    for (int m = 0; m < RL_NP; m++) {
        const uint32_t o = (uint32_t)(ix[2 * m] | ix[2 * m + 1]);
        k0 = o != 0 ? m + 1 : k0;
        k1 = o > 1 ? m + 1 : k1;
    }
    rl_run_lengths(tb, k0, k1, st);
    const int count1 = st.count1, bv = st.big_values, bvr = 2 * bv;

    // ---- one pass over the lane's pairs: table word of the pair (lower / upper bound of the big-value bits: pairs below
    //      big_values at their shortest code; every pair the regions can reach -- with big_values == 0 the stale address2
    //      still delimits regions 0 and 1 (E7) -- at its longest code with 13 linbits per escape) and count1_bit_count:
    //      quad k = pairs (bv+2k, bv+2k+1), p = v + 2w + 4x + 8y; the second pair of a quad is the lane's next slot, or
    //      the first slot of the lane above
    uint32_t h[RL_NP], c2[RL_NP + 1];
    uint32_t bnd = 0, acc = 0;
    {
        const int reach = st.a2 > bvr ? st.a2 : bvr;
        // pairs of this lane below big_values / below `reach`: slot m is inside when m < the distance -- one subtraction per lane and bound, then
        // compares with the slot's number as an inline constant (no per-slot register: ten of them were hoisted out of the search loop until r06)
        const int d_bv = bv - p0, d_reach = ((reach + 1) >> 1) - p0;
        // inside the count1 region every value is 0 or 1, so x + 2y is the pair's code there; elsewhere the sum is only
        // kept inside the table (& 15) and its result dropped
#pragma unroll
        for (int m = 0; m < RL_NP; m++) {
            const int x = ix[2 * m], y = ix[2 * m + 1];
            const uint2 hh = tb.hl[(x > 14 ? 15 : x) * 16 + (y > 14 ? 15 : y)];
            h[m] = hh.x;
            bnd += m < d_bv ? hh.y : (m < d_reach ? hh.y & 0xffff0000u : 0u);
        }
        if (count1 > 0) {                                   // (wave-uniform: a probe whose values are all above 1 up to the last non-zero pair has no quads)
#pragma unroll
            for (int m = 0; m < RL_NP; m++) c2[m] = (uint32_t)(ix[2 * m] + 2 * ix[2 * m + 1]);
            c2[RL_NP] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c2[0], 0x130, 0xf, 0xf, false);   // wave_shl:1 = lane + 1's
            const uint32_t c1n = (uint32_t)(2 * count1);
            const uint32_t rel0 = (uint32_t)(p0 - bv);
            // a pair at an even distance from big_values starts a quad (code length + its ones), the others add their ones only
            const uint32_t keep_even = (rel0 & 1u) ? 0xffff0000u : 0xffffffffu, keep_odd = (rel0 & 1u) ? 0xffffffffu : 0xffff0000u;
#pragma unroll
            for (int m = 0; m < RL_NP; m++) {
                const uint32_t quad = tb.c1w[(c2[m] | (c2[m + 1] << 2)) & 15u];
                acc += rel0 + (uint32_t)m < c1n ? quad & ((m & 1) ? keep_odd : keep_even) : 0u;
            }
            wave_add2(bnd, acc);
        } else bnd = wave_add_u32(bnd);
    }
    const int signs = (int)(acc >> 16), sum0 = signs + (int)(acc & 0xffff), sum1 = signs + 4 * count1;
    int bits;
    if (sum0 < sum1) { st.c1sel = 0; bits = sum0; } else { st.c1sel = 1; bits = sum1; }   // ties -> table B (E10)
    full = false;
    {
        const int lb = bits + (int)(bnd & 0xffffu), ub = bits + (int)(bnd >> 16);
        if (lb >= limit) return lb;
        if (ub_ok && ub < limit) return ub;
        full = true;
    }

    // ---- region maxima; a pair starting at line s belongs to r0 if s < a1, r1 if s < a2, r2 if s < 2*bv
    const int a1 = st.a1, a2 = st.a2;
    uint32_t mx0 = 0, mx1 = 0, mx2 = 0;
    int rid[RL_NP];
    const int d_a1 = ((a1 + 1) >> 1) - p0, d_a2 = ((a2 + 1) >> 1) - p0, d_bv3 = bv - p0;   // (line 2 (p0 + m) below a bound <=> m below the bound's pair count - p0)
#pragma unroll
    for (int m = 0; m < RL_NP; m++) {
        const uint32_t pm = (uint32_t)max(ix[2 * m], ix[2 * m + 1]);
        const bool in0 = m < d_a1, in1 = !in0 && m < d_a2, in2 = !in0 && !in1 && m < d_bv3;
        mx0 = max(mx0, in0 ? pm : 0u);
        mx1 = max(mx1, in1 ? pm : 0u);
        mx2 = max(mx2, in2 ? pm : 0u);
        rid[m] = in0 ? 0 : (in1 ? 1 : (in2 ? 2 : -1));
    }
    wave_max3(mx0, mx1, mx2);
    const int rmax[3] = {(int)mx0, (int)mx1, (int)mx2};

    // ---- candidates per region (__new_choose_table)
    int tA[3], tB[3];
    // both candidates of a pair come out of ONE table word; what differs per region is wave-uniform and travels in one
    // scalar: bit offset of candidate A's length | B's << 8 | A's linbits << 16 | B's << 20 (offset 25 = empty field)
    uint32_t Kr[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const int mxr = rmax[r];
        // mxr < 15: reference :1190-1193 scans tables 13..0 for x_len > ix_max; table 13 (x_len 16) always hits first
        // else: first table of each family whose linmax = 2^linbits - 1 covers ix_max - 15 (:1237-1246);
        // linbits are {1,2,3,4,6,8,10,13} for 16..23 and {4,5,6,7,8,9,11,13} for 24..31, table 15 has none
        const int need = mxr - 15;
        const int nb = need > 0 ? 32 - __builtin_clz((unsigned)need) : 0;   // bits of need
        // (nb > 1) + (nb > 2) + (nb > 3) + (nb > 4) + (nb > 6) + (nb > 8) + (nb > 10) and (nb > 4) + ... + (nb > 9) + (nb > 11)
        // as nibbles of two constants (nb <= 13): one scalar shift each
        const int kA = (int)((0x77766554432100ull >> (4 * nb)) & 15ull);
        const int kB = (int)((0x77665432100000ull >> (4 * nb)) & 15ull);
        const bool none = mxr == 0, small = mxr < 15, esc0 = need == 0;
        tA[r] = none ? 0 : (small ? 13 : (esc0 ? 15 : 16 + kA));
        tB[r] = none ? 0 : (small ? 15 : 24 + kB);
        const uint32_t lbA = (0xDA864321u >> (4 * kA)) & 15u, lbB = (0xDB987654u >> (4 * kB)) & 15u;
        const uint32_t big = (esc0 ? 5u : (10u | (lbA << 16))) | (15u << 8) | (lbB << 20);
        Kr[r] = none ? (25u | (25u << 8)) : (small ? (0u | (5u << 8)) : big);
    }
#if MP3S_RL_STATS
    RL_STAT(63, 1);                                                         // evaluations in full ...
    if (rmax[0] < 15 && rmax[1] < 15 && rmax[2] < 15) RL_STAT(64, 1);       // ... whose three regions all stay below 15 (no escapes, books 13 / 15 only)
    if (rmax[1] < 15 && rmax[2] < 15) RL_STAT(65, 1);                       // ... whose regions 1 and 2 do
#endif
    uint32_t w0 = 0, w1 = 0, w2 = 0;
#pragma unroll
    for (int m = 0; m < RL_NP; m++) {
        const int r = rid[m];
        const uint32_t kp = (uint32_t)sel3(r, (int)Kr[0], (int)Kr[1], (int)Kr[2]);
        const uint32_t nz = __builtin_amdgcn_ubfe(h[m], 20, 2), esc = __builtin_amdgcn_ubfe(h[m], 22, 2);
        const uint32_t a = __builtin_amdgcn_ubfe(h[m], kp & 31u, 5) + nz + __builtin_amdgcn_ubfe(kp, 16, 4) * esc;
        const uint32_t b = __builtin_amdgcn_ubfe(h[m], (kp >> 8) & 31u, 5) + nz + __builtin_amdgcn_ubfe(kp, 20, 4) * esc;
        const uint32_t v = a | (b << 16);
        w0 += r == 0 ? v : 0u;   // pairs past big_values (r < 0) match no region
        w1 += r == 1 ? v : 0u;
        w2 += r == 2 ? v : 0u;
    }
    wave_add3(w0, w1, w2);
    const uint32_t wsum[3] = {w0, w1, w2};
    int ts[3], rbits[3];
    bool redo[3];
    int idx = cursor;
#pragma unroll
    for (int r = 0; r < 3; r++) {
        ts[r] = 0; rbits[r] = 0; redo[r] = false;
        if (tA[r]) {                                   // uniform
            const uint32_t s = wsum[r];
            const int bA = (int)(s & 0xffff), bB = (int)(s >> 16);
            int choice;
            if (tA[r] == 13) choice = (bB <= bA) ? 15 : 13;     // :1227-1231 ties -> 15
            else choice = (bB < bA) ? tB[r] : tA[r];            // :1254
            if (n_hide > 0 && idx < n_hide) choice = tb.transform[choice * 2 + (hide[idx] & 1)];   // :1257-1263
            ts[r] = choice;
            if (choice == tA[r]) rbits[r] = bA;
            else if (choice == tB[r]) rbits[r] = bB;
            else redo[r] = true;
            if (choice > 0) idx += 1;
        }
    }
    if (redo[0] || redo[1] || redo[2]) {               // a hide swap picked a table outside the two candidates
        uint32_t extra = 0;
        const int f0 = family_of(ts[0]), f1 = family_of(ts[1]), f2 = family_of(ts[2]);
        const int l0 = lin_bits_of(ts[0]), l1 = lin_bits_of(ts[1]), l2 = lin_bits_of(ts[2]);
#pragma unroll
        for (int m = 0; m < RL_NP; m++) {
            const int r = rid[m];
            const bool rd = r >= 0 && sel3(r, redo[0], redo[1], redo[2]) != 0;
            const uint32_t v = pair_bits(tb, sel3(r, f0, f1, f2), sel3(r, l0, l1, l2), ix[2 * m], ix[2 * m + 1]);
            extra += rd ? v : 0u;
        }
        bits += (int)wave_add_u32(extra);
    }
    st.ts0 = ts[0]; st.ts1 = ts[1]; st.ts2 = ts[2];
    return bits + rbits[0] + rbits[1] + rbits[2];
}

// Entries behind the launch's own units (message variants decided on the device, k_chain_select): entry e re-runs unit
// unit[e] with the message cursor cursor[e] and no inherited state; its results go to element e of ix / out / en.
struct RateVariants {
    const int32_t *unit, *cursor;
    int n;
    int16_t *ix; mp3s_gr_out *out; int32_t *en;
    uint8_t *tables;   // the entries' table counts once more, one byte each (what the selection's walk reads)
};

// CHAIN (k_rate_redo): the workgroup is persistent over the list (`n_list` entries are there, read from the device), and a wave
// that has run its entry again FOLLOWS THE CHAIN behind it: the unit of the same (gr, ch) in the next frame inherits the
// addresses and the quantiser step this one leaves (encoder/MP3_Encoder.py:1004-1006, 788-803) -- if it is silent it
// only passes them on, if it read them (MP3S_RF_USED_ADDR_IN) and was given others it is run again on the spot, and so on until
// a unit stands on its own.  One re-run used to settle only the listed unit itself; what it changed for the units behind
// it went to the host (2.3 % of the long-stream jobs of tools/soak_select_long.py).  `cursor_all`: the cursors by unit, for
// the units a chain reaches (k_chain_apply keeps them current).
template <bool CHAIN = false>
__device__ __forceinline__ void rate_units(
    const int32_t *__restrict__ mdct, const mp3s_rate_frame *__restrict__ frames, int n_units,
    const uint8_t *__restrict__ hide, int n_hide_all, const int32_t *__restrict__ cursor_in,
    const int32_t *__restrict__ state_in, const int32_t *__restrict__ unit_list, int n_list,
    int16_t *__restrict__ ix_out, mp3s_gr_out *__restrict__ out, int32_t *__restrict__ en_out, int out_base, int compact,
    RateVariants var, const int32_t *__restrict__ cursor_all = nullptr)
{
    const int n_total = n_list + var.n;
    // compact (re-runs of a unit list): cursor_in / state_in / out are indexed by the position in the list, so that a
    // pass moves a few bytes per listed unit over PCIe instead of whole-batch arrays; ix / en still land in place.
    // compact == 2 (message variants: the list names a unit once per 3-bit pattern): ix / en go by list position too
    // compact == 3 (the device's own re-runs, k_rate_redo): only cursor_in / state_in by list position, results in place
    // results of unit u go to element u - out_base of the three output arrays (0 except for the message variants,
    // whose arrays hold one chunk of units)
    __shared__ __attribute__((aligned(16))) RlTables tb;
#if MP3S_RL_STATS
    const unsigned long long clk0 = RL_CLK();
#endif
    const int sr0 = frames[0].sr_idx;             // one sample rate per launch (the host splits batches otherwise)
    const int sr_wg = sr0 >= 0 && sr0 < 3 ? sr0 : 0;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int p0 = 5 * lane;                      // the lane's first pair
    // The wave's own lines are asked for BEFORE the tables are staged (plain launches: one unit per wave): their latency passes
    // under the staging and its barrier instead of behind it (staging measured at 7 % of the kernel, this wait at half as much again).
    int2 xpre[RL_NP];
    int u_pre = -1;
    if (!CHAIN) {
        const int li = blockIdx.x * RL_WAVES + wave;
        if (li < n_list + var.n) {
            const int32_t *ul = li >= n_list ? var.unit : unit_list;
            const int lj = li >= n_list ? li - n_list : li;
            u_pre = __builtin_amdgcn_readfirstlane(ul ? ul[lj] : lj);
            if (u_pre >= 0 && u_pre < n_units) {
                const int32_t *xr = mdct + (long)u_pre * 576;
#pragma unroll
                for (int m = 0; m < RL_NP; m++) {
                    xpre[m] = make_int2(0, 0);
                    if (p0 + m < 288) xpre[m] = *reinterpret_cast<const int2 *>(xr + 2 * (p0 + m));
                }
            }
        }
    }
    // The tables: asked for HERE and waited for (the workgroup's one barrier) where a wave first needs them -- in a plain launch behind its
    // unit's own front (lines, energies, scfsi logs: nothing of it reads a table), so that the tables' way through the caches passes under that
    // arithmetic; a re-run launch (CHAIN) waits at once.  The three large tables go STRAIGHT into LDS
    // (global_load_lds_dwordx4: each lane's 16 bytes land at the wave's LDS base + 16 * lane; no register holds them -- through registers the
    // staging kept 24 VGPRs alive across the unit's front, the kernel's peak: r06), the two of 64 bytes through a register each.
    static_assert(sizeof(tb.int2idx) % 16 == 0 && sizeof(c_tab.int2idx) == sizeof(tb.int2idx), "int2idx is staged 16 bytes at a time");
    static_assert(sizeof(tb.hl) == sizeof(c_tab.rl_hl) && sizeof(tb.hl) % 1024 == 0, "the pair words are staged one wave-load (1 KB) at a time");
    static_assert(offsetof(RlTables, hl) % 16 == 0 && offsetof(RlTables, subdiv) % 16 == 0 && sizeof(tb.subdiv) % 16 == 0, "16-byte pieces in LDS");
    constexpr int I2I_CHUNKS = (int)(sizeof(tb.int2idx) / 16), I2I_ROUNDS = (I2I_CHUNKS + RL_WAVES * 64 - 1) / (RL_WAVES * 64);
    constexpr int HL_CHUNKS = (int)(sizeof(tb.hl) / 16), SD_CHUNKS = (int)(sizeof(tb.subdiv) / 16);
    static_assert(HL_CHUNKS + SD_CHUNKS <= RL_WAVES * 64 && HL_CHUNKS % 64 == 0, "pair words and region table: one piece per thread");
    uint32_t t_c1w = 0, t_tr = 0;
    {
        typedef const __attribute__((address_space(1))) void *gsrc;
        typedef __attribute__((address_space(3))) void *ldst;
        char *const lds0 = reinterpret_cast<char *>(&tb);
        const char *const g_i2i = reinterpret_cast<const char *>(c_tab.int2idx);
#pragma unroll
        for (int r = 0; r < I2I_ROUNDS; r++) {
            const int base = (r * RL_WAVES + wave) * 64;                      // the wave's first piece of this round
            if (base + lane < I2I_CHUNKS)
                __builtin_amdgcn_global_load_lds((gsrc)(g_i2i + (size_t)(base + lane) * 16), (ldst)(lds0 + offsetof(RlTables, int2idx) + base * 16), 16, 0, 0);
        }
        {
            // pieces 0 .. HL_CHUNKS-1: the pair words; behind them (a wave boundary) the region table of this sample rate, whose last piece reads a few
            // bytes past its 289 words (still inside DevTables) into the padding of tb.subdiv
            const int base = wave * 64;
            const bool is_hl = base < HL_CHUNKS;
            const char *src = is_hl ? reinterpret_cast<const char *>(c_tab.rl_hl) + (size_t)(base + lane) * 16
                                    : reinterpret_cast<const char *>(c_tab.subdiv_lut[sr_wg]) + (size_t)(base - HL_CHUNKS + lane) * 16;
            char *dst = is_hl ? lds0 + offsetof(RlTables, hl) + base * 16 : lds0 + offsetof(RlTables, subdiv) + (base - HL_CHUNKS) * 16;
            if (is_hl || base - HL_CHUNKS + lane < SD_CHUNKS) __builtin_amdgcn_global_load_lds((gsrc)src, (ldst)dst, 16, 0, 0);
        }
        if (threadIdx.x < 16) t_c1w = c_tab.rl_c1w[threadIdx.x];
        if (threadIdx.x < 64) t_tr = c_tab.transform[threadIdx.x >> 1][threadIdx.x & 1];
    }
    auto tables_in = [&]() {
        if (threadIdx.x < 16) tb.c1w[threadIdx.x] = t_c1w;
        if (threadIdx.x < 64) tb.transform[threadIdx.x] = (uint8_t)t_tr;
        __builtin_amdgcn_s_waitcnt(0);            // (this wave's pieces have landed in LDS)
        __syncthreads();
    };
    if (CHAIN) tables_in();

    int li0 = blockIdx.x * RL_WAVES + wave;
    if (li0 >= n_total) { if (!CHAIN) tables_in(); return; }
    do {                                          // (one trip unless CHAIN: the plain rate loop keeps its straight-line shape)
    int li = li0;
    uint8_t *tables_out = nullptr;
    if (li >= n_list) {                           // a variant entry: by entry position throughout (as compact == 2)
        li -= n_list;
        tables_out = var.tables;
        unit_list = var.unit; cursor_in = var.cursor; state_in = nullptr;
        ix_out = var.ix; out = var.out; en_out = var.en;
        compact = 2;
    }
    int u = CHAIN ? __builtin_amdgcn_readfirstlane(unit_list ? unit_list[li] : li) : u_pre;
    if (u < 0 || u >= n_units) { if (CHAIN) continue; tables_in(); return; }
    // what the unit runs on: from the arrays (by unit, or by list position), or -- a unit reached through a chain -- from the run in front
    bool chained = false;
    int ch_cursor = 0, ch_state[4] = {0, 0, 0, 0};
    for (int steps = 0; ; steps++) {
    if (CHAIN) __builtin_amdgcn_wave_barrier();   // (the wave's LDS scratch of the unit before is no longer read)
    const mp3s_rate_frame fr = frames[u >> 2];
    const int sr = sr_wg;
    const int max_bits = fr.max_bits;
    const int ci = compact ? li : u;
    const int cursor = chained ? ch_cursor : ((n_hide_all > 0 && cursor_in) ? cursor_in[ci] : 0);
    const int n_hide = min(n_hide_all, fr.hide_end);   // streams of a batch keep their messages back to back in `hide`

    // ---- load xr, |xr|, xrsq >> 10 (:770-776, :837-838)
    const int32_t *xr = mdct + (long)u * 576;
    uint32_t xa[2 * RL_NP];
    uint32_t negmask = 0, lmax = 0, esum = 0;
    uint32_t pk[2 * RL_NP - 1];                   // the lane's own running sums: pk[k] = its lines 0..k
#pragma unroll
    for (int m = 0; m < RL_NP; m++) {
        const int p = p0 + m;
        int2 v = make_int2(0, 0);
        if (!CHAIN) v = xpre[m];
        else if (p < 288) v = *reinterpret_cast<const int2 *>(xr + 2 * p);
        const int32_t vv[2] = {v.x, v.y};
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const uint32_t a = vv[e] < 0 ? (0u - (uint32_t)vv[e]) : (uint32_t)vv[e];
            xa[2 * m + e] = a;
            if (vv[e] < 0) negmask |= 1u << (2 * m + e);
            lmax = max(lmax, a);
            const int32_t sq = (int32_t)(((uint64_t)a * a + (1ull << 30)) >> 31);   // util.mulsr(xr, xr)
            const int32_t e10 = sq >> 10;
            esum += (uint32_t)e10;
            if (2 * m + e < 2 * RL_NP - 1) pk[2 * m + e] = esum;
        }
    }
    uint32_t xrmax = lmax, etot = 0;
    // ---- scalefactor-band energies for __calc_scfsi (:840-857); lane b < 21 sums band b, lane 21 = total
    {
        // Band sums through prefix sums: lane l holds the sum of lines 10l .. 10l+9 already; a wave scan turns the lane sums
        // into the sums of everything in front of each lane.  Lane j < 22 then builds S(j), the sum of all lines in front of
        // band bound j = 10 q + r: what lies in front of lane q, plus lane q's own running sum over r lines -- both fetched
        // from lane q by ds_bpermute (nine running sums are offered, the one with k + 1 == r is kept), and a band is
        // S(j + 1) - S(j) with the neighbour's S through DPP.  Sums stay below 2^31 (576 x 2^21).  (Round 3 had the band
        // lanes walk their up to nine lines through a copy of all squares in LDS -- 9 KB per workgroup and the fifth wave
        // per SIMD --, round 4 first through the unit's lines in memory: two divergent loops of dependent loads, 8 % of the
        // kernel's vector instructions.)
        const uint32_t incl = wave_scan_u32(esum);
        etot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        xrmax = wave_max_u32(lmax);
        const int bj = c_tab.sfb_long[sr][lane < 22 ? lane : 22];
        const int qj = (bj * 6554) >> 16, rj = bj - 10 * qj;                 // x / 10 for x <= 576; q <= 57
        uint32_t S = (uint32_t)__builtin_amdgcn_ds_bpermute(qj << 2, (int)(incl - esum)), part = 0;
#pragma unroll
        for (int k = 0; k < 2 * RL_NP - 1; k++) {
            const uint32_t v = (uint32_t)__builtin_amdgcn_ds_bpermute(qj << 2, (int)pk[k]);
            part = rj == k + 1 ? v : part;
        }
        S += part;
        const uint32_t Sn = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)S, 0x130, 0xf, 0xf, false);   // wave_shl:1 = lane + 1's
        const int32_t temp = lane < 21 ? (int32_t)(Sn - S) : (lane == 21 ? (int32_t)etot : 0);
        // en = int32(log(temp * 4.768371584e-7) / 0.69314718): tabulated with the host's libm per octave of temp (value at
        // 2^k, and the argument from which on it is one more), so no device log and nothing to guard
        int32_t en = 0;
        if (lane < 22 && temp > 0) {
            const int k = 31 - __builtin_clz((unsigned)temp);
            en = c_tab.en_base[k] + (temp >= c_tab.en_step[k] ? 1 : 0);
        }
        if (lane < 22) en_out[(long)(compact == 2 ? li : u - out_base) * 22 + lane] = en;
        RlState st;
        st.big_values = st.count1 = st.c1sel = st.r0c = st.r1c = 0;
        // (wave-uniform: read into scalar registers -- as vector loads they lived in three VGPRs across the whole search for the one word that
        //  reports them at the end, the kernel's last scratch spills at six waves per SIMD)
        const int in_a1 = __builtin_amdgcn_readfirstlane(chained ? ch_state[0] : (state_in ? state_in[(long)ci * 4 + 0] : 0));
        const int in_a2 = __builtin_amdgcn_readfirstlane(chained ? ch_state[1] : (state_in ? state_in[(long)ci * 4 + 1] : 0));
        const int in_a3 = __builtin_amdgcn_readfirstlane(chained ? ch_state[2] : (state_in ? state_in[(long)ci * 4 + 2] : 0));
        const int in_given = in_a1 | (in_a2 << 10) | (in_a3 << 20);          // what k_chain_apply compares with the true chain (each < 1024)
        st.a1 = in_a1; st.a2 = in_a2; st.a3 = in_a3;
        int qstep = chained ? ch_state[3] : (state_in ? state_in[(long)ci * 4 + 3] : 0);
        st.ts0 = st.ts1 = st.ts2 = 0;
        st.addr_fresh = false; st.used_addr_in = false;
        int32_t ix[2 * RL_NP];
#pragma unroll
        for (int e = 0; e < 2 * RL_NP; e++) ix[e] = 0;
        int bits = 0, flags = 0;
        bool err = false;

#if MP3S_RL_STATS
        const unsigned long long clk1 = RL_CLK();
#endif
        if (!CHAIN) tables_in();               // (every wave of the workgroup comes by here, or by one of the two returns above, exactly once)
#if MP3S_RL_STATS
        const unsigned long long clk2 = RL_CLK();
        unsigned long long clk3 = clk2, clk4 = clk2, clk5 = clk2;
        RL_STAT(54, clk1 - clk0); RL_STAT(55, clk2 - clk1); RL_STAT(60, 1);
        int probe_j = 0;
#endif
        if (xrmax) {
            flags |= MP3S_RF_ACTIVE;
            RL_STAT(35, 1);
            // ---- __bin_search_step_size (:958-996)
            int next = -120, count = 120;
            int body_step = 1 << 20, body_bits = 0;   // step whose quantisation + rl_body results are still in ix / st
            uint32_t sc = rl_scale_of(next + count / 2);
            asm volatile("s_mov_b32 %0, %0" : "+s"(sc));   // arrived before the loop: inside it no use of `sc` has to wait for the loads issued at its top
            {
                // the first probe (step -60) is decided from thresholds where that can be done (rl_precheck): in front of the loop, whose
                // registers it then does not share
                const uint32_t sc_lo = rl_scale_of(next + 30), sc_hi = rl_scale_of(next + 90);
                bool q_early;
                const int bit = rl_precheck(tb, xa, p0, st, next + 60, sc, xrmax, max_bits, q_early);
                if (bit >= 0) {
                    if (bit < max_bits) { count = 60; sc = sc_lo; }
                    else { next += 60; count -= 60; sc = sc_hi; }
                    asm volatile("s_mov_b32 %0, %0" : "+s"(sc));
#if MP3S_RL_STATS
                    RL_STAT(0 * 5 + 0, 1); RL_STAT(40, 1); RL_STAT(47, 1); probe_j = 1;
#endif
                }
            }
#if MP3S_RL_STATS
            clk3 = RL_CLK();
#endif
            do {
                const int half = count / 2;
                // the scales of the two steps the search may probe next, asked for now
                const uint32_t sc_lo = rl_scale_of(next + half / 2), sc_hi = rl_scale_of(next + half + (count - half) / 2);
                const int q = rl_quantize(tb, xa, ix, next + half, sc, xrmax);
                int bit;
                if (q < 0) { err = true; break; }
                if (q > 8192) { bit = 100000; if (q != 16384) body_step = 1 << 20; RL_STAT(probe_j * 5 + 1, 1); }   // 16384 = early out, ix untouched
                else {
                    bool full;
                    // the last probe (half == 1) is the one the inner loop may reuse: it is evaluated in full
                    bit = rl_body(tb, ix, p0, st, hide, n_hide, cursor, max_bits, half > 1, full);
                    if (full) { body_step = next + half; body_bits = bit; } else body_step = 1 << 20;
#if MP3S_RL_STATS
                    RL_STAT(probe_j * 5 + (full ? 4 : (bit >= max_bits ? 2 : 3)), 1);
                    RL_STAT(62, 1); if (st.count1 > 0) RL_STAT(61, 1);
#endif
                }
#if MP3S_RL_STATS
                if (bit >= max_bits) {           // what the threshold pre-check would have said about this probe
                    RlState st2 = st; bool qe;
                    const int pb = rl_precheck(tb, xa, p0, st2, next + half, sc, xrmax, max_bits, qe);
                    RL_STAT(40 + probe_j, 1); if (pb >= max_bits) RL_STAT(47 + probe_j, 1);
                }
                probe_j++;
#endif
                if (bit < max_bits) { count = half; sc = sc_lo; }
                else { next += half; count -= half; sc = sc_hi; }
            } while (count > 1);
            qstep = next;
#if MP3S_RL_STATS
            clk4 = RL_CLK();
#endif
            // ---- __inner_loop (:1064-1095), part2_length == 0
            if (!err) {
                if (max_bits < 0) qstep -= 1;
                do {
                    if (qstep + 1 == body_step) {
                        // the reference re-quantises the step the binary search probed last and gets the same ix, GrInfo
                        // and bit count again (rl_body is a pure function of ix, the cursor and -- only when
                        // big_values == 0 -- the addresses it leaves untouched): reuse them
                        qstep += 1;
                        bits = body_bits;
                        body_step = 1 << 20;
                        RL_STAT(37, 1);
                        continue;
                    }
                    int q;
                    while ((q = rl_quantize(tb, xa, ix, qstep + 1, rl_scale_of(qstep + 1), xrmax)) > 8192) qstep += 1;
                    if (q < 0) { err = true; break; }
                    qstep += 1;
                    bool full;
                    bits = rl_body(tb, ix, p0, st, hide, n_hide, cursor, max_bits + 1, false, full);
#if MP3S_RL_STATS
                    RL_STAT(36, 1); RL_STAT(full ? 39 : 38, 1); RL_STAT(62, 1); if (st.count1 > 0) RL_STAT(61, 1);
#endif
                } while (bits > max_bits);
            }
            if (err) flags |= MP3S_RF_STEP_RANGE;
#if MP3S_RL_STATS
            clk5 = RL_CLK();
            RL_STAT(56, clk3 - clk2); RL_STAT(57, clk4 - clk3); RL_STAT(58, clk5 - clk4);
#endif
        }
        if (st.used_addr_in) flags |= MP3S_RF_USED_ADDR_IN;
        if (CHAIN && !chained) flags |= MP3S_RF_LISTED;    // (a list entry's unit stays marked: see the chain walk below)

        // ---- signed ix (format_bitstream :1272-1276) as int16 pairs
        int16_t *ixo = ix_out + (long)(compact == 2 ? li : u - out_base) * 576;
#pragma unroll
        for (int m = 0; m < RL_NP; m++) {
            const int p = p0 + m;
            if (p < 288) {
                int a = xrmax ? ix[2 * m] : 0, b = xrmax ? ix[2 * m + 1] : 0;
                if ((negmask >> (2 * m)) & 1) a = -a;
                if ((negmask >> (2 * m + 1)) & 1) b = -b;
                *reinterpret_cast<uint32_t *>(ixo + 2 * p) = ((uint32_t)(uint16_t)(int16_t)a) | ((uint32_t)(uint16_t)(int16_t)b << 16);
            }
        }
        if (lane == 0) {
            mp3s_gr_out o;
            const bool act = xrmax != 0;
            o.part2_3_length = act ? bits : 0;
            o.big_values = act ? st.big_values : 0;
            o.count1 = act ? st.count1 : 0;
            o.quantizer_step = qstep;
            o.region0_count = act ? st.r0c : 0;
            o.region1_count = act ? st.r1c : 0;
            o.count1table_select = act ? st.c1sel : 0;
            o.table_select[0] = act ? st.ts0 : 0;
            o.table_select[1] = act ? st.ts1 : 0;
            o.table_select[2] = act ? st.ts2 : 0;
            o.address[0] = st.a1; o.address[1] = st.a2; o.address[2] = st.a3;
            o.n_tables = act ? (st.ts0 > 0) + (st.ts1 > 0) + (st.ts2 > 0) : 0;
            o.flags = flags;
            // the inherited addresses the unit was given (each < 1024): what k_chain_apply compares with the true chain
            o.reserved0 = (state_in || chained) ? in_given : 0;
            o.xrmax = (int32_t)xrmax;
            o.reserved = 0;
            out[compact == 1 || compact == 2 ? li : u - out_base] = o;
            if (tables_out) tables_out[li] = (uint8_t)o.n_tables;
        }
#if MP3S_RL_STATS
        RL_STAT(59, RL_CLK() - clk5);
#endif
        if (!CHAIN) break;
        // ---- the chain behind this unit: what it leaves (an active unit: its own addresses and step; a silent one passes on what it got)
        ch_state[0] = __builtin_amdgcn_readfirstlane(st.a1); ch_state[1] = __builtin_amdgcn_readfirstlane(st.a2);
        ch_state[2] = __builtin_amdgcn_readfirstlane(st.a3); ch_state[3] = __builtin_amdgcn_readfirstlane(qstep);
        bool go = false;
        int un = u;
        for (;;) {
            un += 4;                                               // the same (gr, ch) of the next frame
            if (++steps >= REDO_CHAIN_MAX || un >= n_units || frames[un >> 2].stream != fr.stream) break;
            const int f2 = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&out[un - out_base].flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            // A unit that is itself an entry of the list has its own wave in this launch -- possibly at work right now, and it may
            // have written its record already or not: the mark k_chain_apply gave it stays set either way.  The chain stops here;
            // the entry runs on what the check derived from the first pass, and the check behind this launch judges the result.
            if (f2 & MP3S_RF_LISTED) break;
            if (!(f2 & MP3S_RF_ACTIVE)) continue;                  // silent: it inherits and passes on (k_chain_apply writes its record)
            const int given = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&out[un - out_base].reserved0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            go = (f2 & MP3S_RF_USED_ADDR_IN) && given != (ch_state[0] | (ch_state[1] << 10) | (ch_state[2] << 20));
            break;                                                 // (an active unit that did not look at what it inherited: the chain ends)
        }
        if (!go) break;
        u = un; chained = true;
        ch_cursor = (n_hide_all > 0 && cursor_all) ? cursor_all[u] : 0;
    }
    }   // the unit and its chain
    } while (CHAIN && (li0 += (int)gridDim.x * RL_WAVES) < n_total);   // list entries
}

__global__ __launch_bounds__(RL_WAVES * 64, MP3S_RL_OCC) void k_rate_loop(
    const int32_t *__restrict__ mdct, const mp3s_rate_frame *__restrict__ frames, int n_units,
    const uint8_t *__restrict__ hide, int n_hide, const int32_t *__restrict__ cursor_in,
    const int32_t *__restrict__ state_in, const int32_t *__restrict__ unit_list, int n_list,
    int16_t *__restrict__ ix_out, mp3s_gr_out *__restrict__ out, int32_t *__restrict__ en_out, int out_base, int compact,
    RateVariants var)
{
    rate_units(mdct, frames, n_units, hide, n_hide, cursor_in, state_in, unit_list, n_list, ix_out, out, en_out, out_base, compact, var);
}

// The units k_chain_apply listed (they ran on inherited addresses or a cursor that turned out different: k_chain.hpp) once
// more, on what it found, results in place.  The list lives on the device and is short or empty: its length is not a
// launch parameter: redo[0] entries are there (at most REDO_CAP are kept; what the check finds beyond them stays with the
// verdict), REDO_GROUPS persistent workgroups share them, and each re-run follows its chain (rate_units<true>).
__global__ __launch_bounds__(RL_WAVES * 64, 4) void k_rate_redo(
    const int32_t *__restrict__ mdct, const mp3s_rate_frame *__restrict__ frames, int n_units,
    const uint8_t *__restrict__ hide, int n_hide, const int32_t *__restrict__ redo, int16_t *__restrict__ ix_out,
    mp3s_gr_out *__restrict__ out, int32_t *__restrict__ en_out, const int32_t *__restrict__ cursor_all)
{
    const int32_t *unit_list = redo + REDO_HEAD;
    const int n_list = min(__builtin_amdgcn_readfirstlane(__hip_atomic_load(&redo[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)), REDO_CAP);
    if ((int)blockIdx.x * RL_WAVES >= n_list) return;              // (most launches find an empty list: nothing staged)
    const RateVariants none = {nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr};
    rate_units<true>(mdct, frames, n_units, hide, n_hide, unit_list + REDO_CAP, unit_list + 2 * REDO_CAP, unit_list, n_list, ix_out, out, en_out,
                  0, 3, none, cursor_all);
}

// Message variants (enc_resolve): a unit's result depends on the message only through the <= 3 bits at its cursor, so
// the units a message reaches are run once per 3-bit pattern (compact == 2 above) and the host, walking the cursor
// chain, names the entry each unit really sees.  The walk needs one number per entry -- the tables it took -- so those
// come down as a byte array (k_gather_tables), and one wave per (entry, unit) pair then copies that entry's ix, energies
// and GrInfo into the unit's place in the final arrays.
__global__ __launch_bounds__(256) void k_gather_tables(const mp3s_gr_out *__restrict__ outv, int n, uint8_t *__restrict__ tables)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) tables[i] = (uint8_t)outv[i].n_tables;
}

__global__ __launch_bounds__(256) void k_scatter_entries(const int2 *__restrict__ pairs, int n_pairs, const int16_t *__restrict__ ixv,
                                                         const int32_t *__restrict__ env, const mp3s_gr_out *__restrict__ outv,
                                                         int16_t *__restrict__ ix, int32_t *__restrict__ en, mp3s_gr_out *__restrict__ out)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + wave;
    if (j >= n_pairs) return;
    const int2 pr = pairs[j];
    if (pr.x < 0) return;                                     // (k_chain_select: the unit's own run stands)
    const long src = pr.x, dst = pr.y;
    const uint32_t *a = reinterpret_cast<const uint32_t *>(ixv + src * 576);
    uint32_t *b = reinterpret_cast<uint32_t *>(ix + dst * 576);
    for (int i = lane; i < 288; i += 64) b[i] = a[i];
    if (lane < 22) en[dst * 22 + lane] = env[src * 22 + lane];
    static_assert(sizeof(mp3s_gr_out) % 4 == 0, "GrInfo is copied by words");
    if (lane < (int)(sizeof(mp3s_gr_out) / 4))
        reinterpret_cast<uint32_t *>(out + dst)[lane] = reinterpret_cast<const uint32_t *>(outv + src)[lane];
}

}  // namespace mp3s
