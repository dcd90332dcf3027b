// The fast decode transform as ONE wave-local stream (int16 output behind the guard; float32 with MP3S_OPT_FLOAT_FAST):
// requantise -> MS stereo -> reorder | alias reduction -> IMDCT + window + overlap-add -> frequency inversion -> polyphase synthesis ->
// PCM, reference decoder/Frame.py:157-218, 561-631, 106-154, 65-103, 633-640 and MP3_Parser.py:91.  Included by mp3s_device.hip behind
// k_decode.hpp (dec_fetch, dec_requant_ms, load_row18s, dpp_f64, xcd_tile).
//
// The reference runs imdct -> __frequency_inversion -> synth_filter_bank on one array (Frame.py:282-284).  Here that array never exists:
// a wave walks `run` consecutive granules with lane = (channel, subband) from the first stage to the last, and what the stages hand to
// each other stays in the wave's registers.
//   * IMDCT as in k_dec_imdct<true>: 18 lines per lane, twiddle rows as scalar operands, the mirrored rows by sign, fused multiply-adds;
//     the overlap tail of a granule waits for the next one in the wave's own 9 KB of LDS.  Its 18 outputs are the subband's samples of
//     the granule's 18 time slots: S[p], one register pair per slot.
//   * Matrixing of slot p ACROSS the lanes: X[k] = sum_j S[j] cos((2j+1) k pi/64) = sum_{j<16} (S[j] -+ S[31-j]) cos(..) (k odd: the
//     differences, k even: the sums).  A lane gets its mirror subband's sample by ds_swizzle (xor 31), forms its difference (subbands
//     0..15: DPP row 0 / 2 of the wave) or sum (16..31: row 1 / 3), and every lane of a row sums 16 products of ITS row's 16 values with
//     its own 16 cosines: v_fmac_f64 with DPP row_newbcast:j takes the value of lane j of the row as the multiplicand -- sixteen
//     instructions give the 32 X of both channels.  (tools/ubench/dpp64.hip: the DPP form issues at the plain v_fma_f64's rate.)
//   * V[i] = X[16+i] (i <= 16, X[32] = 0), -X[48-i] (17..47), -X[i-48] (48..63): lane (ch, i) fetches A = V[i] and B = V[32+i] from the
//     lanes that hold them (ds_bpermute), signs and the zero live in its taps.
//   * Window: output i of slot t is sum_jj D[i + 32 jj] * (jj even ? A : B)(slot t - jj) (Frame.py:89-101).  Instead of keeping 16 slots of
//     V history, a slot's A and B are added into the sums of the 16 outputs they belong to as they arrive: 18 running sums per lane, indexed
//     by the slot's place in its granule (p + jj) mod 18 -- compile-time indices, since a granule is 18 slots; the sum of slot p is complete
//     when slot p adds its own term and leaves as the PCM sample of (slot, channel, i): a wave stores 128 contiguous bytes per slot.
// Nothing is exchanged between waves: no workgroup barrier behind the staging of the small tables, no LDS for data beyond the tail, no
// device-memory scratch.  A run is primed with the granule in front of it (IMDCT + synthesis without output: the 15 slots of history)
// and the tail of the one before that.
//
// The sums are not the reference's bit patterns (mirrored, fused, other order): the int16 format keeps its promise -- (pcm * 32767)
// truncated exactly as the reference truncates it -- through the guard of DESIGN.md section 2, whose bound covers these sums (at most one
// addition in front of a product, a cosine off by <= 2u, 16 products summed in turn: below the gamma_24 the bound grants the matrixing;
// the 16-tap sum in any order).  The guard's scale A_max (largest sum |S| of a slot) is taken from above here: a slot's sum |S| is at
// most G(its granule) + G(the granule in front), G = sum of |IMDCT input| per granule and channel, so 2 max(G of the three granules a
// window sum can reach) stands for A_max and the same maximum for G_max.  A sample the guard cannot vouch for puts (slot, channel, bit i)
// on the fix-up list; k_dec_fixup computes it again from `is` in the reference's order.
#pragma once
#ifndef MP3S_ST_CLOCKS
#define MP3S_ST_CLOCKS 0   // 1: shader-clock sums of a wave's phases into g_st_clocks, read by mp3s_debug_st_clocks (a probe: tools/st_clocks.py)
#endif
#if MP3S_ST_CLOCKS
__device__ uint32_t g_st_clocks[4096 * 8];
extern "C" __attribute__((visibility("default"))) int mp3s_debug_st_clocks(uint32_t *out)
{
    (void)hipDeviceSynchronize();
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_st_clocks), sizeof(uint32_t) * 4096 * 8);
}
#endif

namespace mp3s {

constexpr int ST_WAVES = 4;                 // waves per workgroup; they share the staged tables and nothing else
constexpr int ST_P43_N = 512;

struct StShared {
    double pow2q[POW2Q_N];                  // (as DecShared: random per-lane reads go to LDS)
    double pow2h[POW2H_N];
    uint32_t side[ST_WAVES][2][18];         // per wave: the two 72-byte side records of the current granule
    double exp2f[ST_WAVES][2][64];          // per wave and channel: 2^(-exp2) per scalefactor slot of the current granule
    double exp1f[ST_WAVES][2][4];           // ... and 2^(exp1/4) per gain selector
    double win[4][36];                      // sine_block
    double tail[ST_WAVES][18][64];          // per wave: the overlap tail of the granule before, [row][lane]
    double p43[2 * ST_P43_N];               // sign(is) |is|^(4/3) of the small values (most of a stream), entry is + 512 for -512 <= is < 512: an LDS look-up where
                                            // the whole table is a trip to L2 -- signed, so that a line needs neither its absolute value nor its sign put back
    double cx[16][32], wt[16][32];          // the synthesis' per-lane constants, [term][subband]: read into registers in front of a granule's 18 slots and
                                            // dead again behind them (64 registers that requantisation and IMDCT have better use for)
};

// One line of channel c of granule g, requantised as dec_requant_ms does it (Frame.py:210-215), for a lane that needs it out of order
// (the reordered lines of a short / mixed granule): is and the line map from memory, the exponent factors of channel c from the wave's tables.
__device__ __forceinline__ double st_requant_line(const DevTables &tab, StShared &sh, int wave, const int16_t *__restrict__ is, int g, int c, int sr, int s)
{
    const uint8_t *gb = reinterpret_cast<const uint8_t *>(sh.side[wave][c]);
    const int bt = gb[2] & 3, mixed = gb[3] ? 1 : 0;
    const int cse = bt == 2 ? 1 : (mixed ? 2 : 0);
    const int sbs = (s * 3641) >> 16;                                   // s / 18 (s < 576)
    const uint32_t m = tab.rq_map[sr][cse][sbs][s - 18 * sbs];
    const int x = is[((long)g * 2 + c) * 576 + s];
    uint32_t ax = (uint32_t)(x < 0 ? -x : x);
    ax = ax < (uint32_t)POW43_N ? ax : (uint32_t)(POW43_N - 1);
    const double a = tab.pow43[ax];
    const double sa = x < 0 ? -a : a;
    return (sa * sh.exp1f[wave][c][m >> 6]) * sh.exp2f[wave][c][m & 63];
}

// X of this lane: sum_j (value of lane j of this lane's DPP row) * c[j], fused multiply-adds in turn.  One block: the compiler does not
// know the DPP hazard (a vector write of `t` needs two wait states before a DPP read, a write of EXEC five) behind inline assembly, so the s_nop is in it.
__device__ __forceinline__ double st_row_dot16(double t, const double (&c)[16])
{
    double x = 0.0;
    asm("s_nop 4\n\t"
        "v_fmac_f64_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %3 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %4 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %6 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %7 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %8 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %10 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %11 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %12 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %13 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %14 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %15 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %16 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %17 row_newbcast:15 row_mask:0xf bank_mask:0xf"
        : "+v"(x)
        : "v"(t), "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(c[4]), "v"(c[5]), "v"(c[6]), "v"(c[7]), "v"(c[8]), "v"(c[9]), "v"(c[10]),
          "v"(c[11]), "v"(c[12]), "v"(c[13]), "v"(c[14]), "v"(c[15]));
    return x;
}

// MARGIN (a probe, mp3s_debug_guard_margin: tests/test_guard_margin.py, tools/soak_fast_synth.py): the same kernel also leaves, per sample it
// stores, the fast value x (fp64, before truncation) in mg_x and the width eps_t its guard compared with in mg_eps (infinite where the
// granule is not vouched for at all), at the sample's index in the launch's output + mg_base -- so that the distance of x from the
// reference-order value can be measured against the bound on the device, not only counted as equal / unequal after truncation.
template <int NCH, bool F32, bool MARGIN = false>
__global__ __launch_bounds__(ST_WAVES * 64, 2) void k_dec_stream(
    const int16_t *__restrict__ is, const mp3s_granule_si *__restrict__ si, const mp3s_frame_hdr *__restrict__ hdr,
    int n_granules, int run, int n_halo, void *__restrict__ pcm_out, int sf_base, double eps_scale,
    uint2 *__restrict__ fix_list, int32_t *__restrict__ fix_count,
    double *__restrict__ mg_x = nullptr, double *__restrict__ mg_eps = nullptr, long mg_base = 0, long mg_cap = 0)
{
    __shared__ StShared sh;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
#if MP3S_ST_CLOCKS
    const unsigned long long ck_start = __builtin_readcyclecounter();
    unsigned long long ck_req = 0, ck_rows = 0, ck_syn = 0, ck_setup = 0;
#endif
    for (int i = threadIdx.x; i < POW2Q_N; i += blockDim.x) sh.pow2q[i] = c_tab.pow2q[i];
    if (threadIdx.x < POW2H_N) sh.pow2h[threadIdx.x] = c_tab.pow2h[threadIdx.x];
    if (threadIdx.x < 144) (&sh.win[0][0])[threadIdx.x] = (&c_tab.sine_block[0][0])[threadIdx.x];
    for (int i = threadIdx.x; i < 2 * ST_P43_N; i += blockDim.x) {
        const int x = i - ST_P43_N;
        sh.p43[i] = x < 0 ? -c_tab.pow43[-x] : c_tab.pow43[x];
    }
    // the lanes' constants of the synthesis: 16 cosines (as the holder of X[k]: k = 2 (sb & 15) + 1 in the rows of the differences, 2 (sb & 15)
    // in the rows of the sums) and 16 taps (as output i = sb)
    for (int i = threadIdx.x; i < 512; i += blockDim.x) {
        sh.cx[i >> 5][i & 31] = c_tab.stream_cx[i & 31][i >> 5];
        sh.wt[i >> 5][i & 31] = c_tab.stream_taps[F32 ? 0 : 1][i & 31][i >> 5];
    }
    __syncthreads();
    const int ga = (xcd_tile() * ST_WAVES + wave) * run;       // this wave's granules: ga .. ga + run - 1
    if (ga >= n_granules) return;                                // (whole waves; no barrier behind this line)

    const int ch = lane >> 5, sb = lane & 31;
    const bool live = ch < NCH;
    const uint32_t sgn_odd = (sb & 1) ? 0x80000000u : 0u;       // frequency inversion (Frame.py:629-631): odd slots of odd subbands
    auto flip = [&](double x) { return __hiloint2double(__double2hiint(x) ^ (int)sgn_odd, __double2loint(x)); };
    // ---- where the lane's A and B come from (as output i = sb)
    int addr_a, addr_b;
    {
        const int ka = sb <= 15 ? 16 + sb : (sb == 16 ? 0 : 48 - sb);        // A = V[i]: X[16+i] | 0 (tap) | -X[48-i]
        const int kb = sb <= 15 ? 16 - sb : sb - 16;                         // B = V[32+i]: -X[16-i] | -X[i-16]
        auto holder = [&](int k) { return (ch << 5) + ((k & 1) ? (k >> 1) : 16 + (k >> 1)); };
        addr_a = holder(ka) * 4; addr_b = holder(kb) * 4;
    }
    const uint32_t sgn_lo = sb < 16 ? 0x80000000u : 0u;         // subbands 0..15 form S - mirror, 16..31 S + mirror
    // the guard's width per unit of the largest G in reach (header comment); F32 has no guard
    const double eps_k = (2.0 * c_tab.synth_eps_a + c_tab.synth_eps_g + 2.0 * c_tab.synth_eps_x * c_tab.synth_xbound) * eps_scale;
    const double xb_k = 2.0 * c_tab.synth_xbound;
    const long halo_slots = (long)n_halo * 36;
    typedef double __attribute__((address_space(3))) lds_f64;
    lds_f64 *const tl = (lds_f64 *)reinterpret_cast<double *>(&sh.tail[wave][0][lane]);   // row r of the tail at tl[r * 64]

    double acc[18];
#pragma unroll
    for (int i = 0; i < 18; i++) acc[i] = 0.0;
    double gm1 = 0.0, gm2 = 0.0;                                // G of the two granules in front (this lane's channel)
    // the stream the run starts in begins at granule s_a of the launch: nothing in front of it primes the run
    const int s_a = [&] { const uint32_t sf = hdr[ga >> 1].stream_first; return sf > (uint32_t)sf_base ? (int)(sf - (uint32_t)sf_base) * 2 : 0; }();
    bool fresh = true;                                          // the tail in LDS and the running sums are to be cleared before the next granule
    GranIn next_in = {};
    bool have_next = false;

    // priming: the granule in front of the run (history of the synthesis) and the tail of the one before, as far as they belong to the stream
    const int gi0 = ga - 2 >= s_a ? -2 : (ga - 1 >= s_a ? -1 : 0);
    mp3s_frame_hdr fh = hdr[(ga + gi0) >> 1];
    // the line map of long blocks (scalefactor band of each of the lane's 18 lines) depends on the sample rate alone: kept across granules
    int sr_map = fh.sr_idx < 3 ? fh.sr_idx : 0;
    uint32_t mwl[5];
#pragma unroll
    for (int k = 0; k < 5; k++) mwl[k] = reinterpret_cast<const uint32_t *>(c_tab.rq_map[sr_map][0][sb])[k];
#if MP3S_ST_CLOCKS
    ck_setup = __builtin_readcyclecounter() - ck_start;
#endif
#pragma unroll 1
    for (int gi = gi0; gi < run; gi++) {
        const int g = ga + gi;
        if (g >= n_granules) break;
#if MP3S_ST_CLOCKS
        const unsigned long long ck_a = __builtin_readcyclecounter();
#endif
        const int first_gran = fh.stream_first > (uint32_t)sf_base ? (int)(fh.stream_first - (uint32_t)sf_base) * 2 : 0;
        const int sr = fh.sr_idx < 3 ? fh.sr_idx : 0;
        const bool ms = fh.ms_stereo != 0;
        // (the next granule's frame header: a scalar load whose latency passes under this granule)
        fh = hdr[(g + 1 < n_granules ? g + 1 : g) >> 1];
        if (g == first_gran) fresh = true;                      // Frame.py:234-235: prev_samples and the fifo start as zeros
        if (fresh) {
#pragma unroll
            for (int i = 0; i < 18; i++) { tl[i * 64] = 0.0; acc[i] = 0.0; }
            gm1 = gm2 = 0.0;
            fresh = false;
        }
        const bool tail_only = gi == -2;                        // two granules in front of the run: only its overlap tail is needed
        const bool emit = gi >= 0;
        // the twiddle tables are invariant over this loop: an opaque zero offset per granule keeps their scalar loads inside it (k_dec_imdct)
        int zoff = 0;
        asm volatile("" : "+s"(zoff));
        const DevTables &tab = *reinterpret_cast<const DevTables *>(reinterpret_cast<const char *>(&c_tab) + zoff);
        const double(*C12)[6] = tab.imdct_cos12;

        // ---- requantise, MS stereo (dec_requant_ms), then reorder | alias reduction without an exchange buffer
        double v[18];
        int bt, cse;
        const GranIn in = have_next ? next_in : dec_fetch(is, si, g, NCH, lane);
        // The common granule -- long blocks in every channel, no value beyond the small |is|^(4/3) table -- without a trip to memory: the
        // side records' fields from the registers they arrived in (lane d of the first 36 holds dword d: v_readlane for the two
        // head dwords of a channel, one ds_bpermute for a lane's scalefactor byte), the line map kept from the granule before,
        // |is|^(4/3) from LDS.  Same products in the same order as dec_requant_ms, which takes everything else.
        bool quick;
        uint32_t w0, w1;
        {
            const uint32_t d0a = (uint32_t)__builtin_amdgcn_readlane((int)in.side, 0), d1a = (uint32_t)__builtin_amdgcn_readlane((int)in.side, 1);
            const uint32_t d0b = (uint32_t)__builtin_amdgcn_readlane((int)in.side, 18), d1b = (uint32_t)__builtin_amdgcn_readlane((int)in.side, 19);
            auto is_long = [](uint32_t d0) { return ((d0 >> 16) & 3u) != 2u && (d0 >> 24) == 0u; };     // block_type != 2, mixed_block_flag == 0
            quick = is_long(d0a) && (NCH == 1 || is_long(d0b));
            w0 = ch ? d0b : d0a; w1 = ch ? d1b : d1a;
            // every line inside the table: -512 <= is < 512  <=>  (is + 512) has no bit above the tenth, for both halves of a dword at once
            uint32_t hi_bits = 0;
#pragma unroll
            for (int k = 0; k < 9; k++) {
                uint32_t t;
                asm("v_pk_add_u16 %0, %1, %2" : "=v"(t) : "v"(in.xw[k]), "v"(0x02000200u));
                hi_bits |= t;
            }
            quick = quick && __ballot((hi_bits & 0xfc00fc00u) != 0) == 0;
        }
        if (quick && sr != sr_map) {
            sr_map = sr;
#pragma unroll
            for (int k = 0; k < 5; k++) mwl[k] = reinterpret_cast<const uint32_t *>(tab.rq_map[sr][0][sb])[k];
        }
        if (quick) {
            bt = (int)((w0 >> 16) & 3u); cse = 0;
            const int gg = (int)(w0 & 0xffu), mult2 = ((w0 >> 8) & 0xffu) ? 2 : 1, preflag = (w1 & 0xffu) ? 1 : 0;
            // 2^(-exp2) per scalefactor band: lane sb < 22 of a channel takes band sb (Frame.py:201-208; byte 8 + sb of the record)
            double *e2 = sh.exp2f[wave][ch];
            const int bsrc = ((ch ? 18 : 0) + 2 + (sb >> 2)) * 4;
            const uint32_t sfw = (uint32_t)__builtin_amdgcn_ds_bpermute(bsrc, (int)in.side);
            if (sb < 22) {
                const int pt = sb < 11 || sb > 20 ? 0 : (int)((0x2333221111ull >> ((sb - 11) * 4)) & 15);   // pre_tab[11..20]
                const int k2 = mult2 * ((int)((sfw >> ((sb & 3) * 8)) & 15u) + preflag * pt);
                e2[sb] = sh.pow2h[k2 < POW2H_N ? k2 : POW2H_N - 1];
            }
            const double e1 = sh.pow2q[gg - 210 - POW2Q_MIN];
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 18; k++) {
                const uint32_t xwk = in.xw[k >> 1];
                const int x = (k & 1) ? (int)xwk >> 16 : (int)(int16_t)xwk;
                const double sa = sh.p43[x + ST_P43_N];
                uint32_t i2;
                asm("v_bfe_u32 %0, %1, %2, 6" : "=v"(i2) : "v"(mwl[k >> 2]), "n"((k & 3) * 8));
                v[k] = (sa * e1) * e2[i2];
            }
            if (ms && NCH == 2) {
#pragma unroll
                for (int k = 0; k < 18; k++) {
                    const double o = shfl_xor_f64(v[k], 32);
                    v[k] = ch == 0 ? (v[k] + o) / tab.sqrt2 : (o - v[k]) / tab.sqrt2;
                }
            }
        } else {
            // Everything else.  Both channels in the same short / mixed case (the usual shape of such a granule, and every mono one): a lane
            // requantises the 18 lines its subband holds AFTER the reorder (Frame.py:574-602: line k of subband sb is line src of the
            // spectrum, or nothing) straight from `is` -- the other channel's lane of the same subband does the same line, so MS stereo is
            // the ordinary exchange.  Channels in different cases (rare): in spectrum order first (dec_requant_ms, MS included), then a short /
            // mixed lane computes its reordered lines again, both channels' values of each (below).
            const uint32_t sd0a = (uint32_t)__builtin_amdgcn_readlane((int)in.side, 0), sd0b = (uint32_t)__builtin_amdgcn_readlane((int)in.side, 18);
            auto case_of = [](uint32_t d0) { return ((d0 >> 16) & 3u) == 2u ? 1 : ((d0 >> 24) ? 2 : 0); };
            const int cse_a = case_of(sd0a), cse_b = NCH == 2 ? case_of(sd0b) : cse_a;
            if (cse_a == cse_b && cse_a != 0) {
                dec_requant_tables(sh, wave, in, NCH, lane, bt, cse);
                const uint32_t *srcw = reinterpret_cast<const uint32_t *>(tab.reorder_src[sr] + sb * 18);
                uint32_t sw[9];
#pragma unroll
                for (int k = 0; k < 9; k++) sw[k] = srcw[k];
#pragma unroll
                for (int k = 0; k < 18; k++) {
                    const int s = (int)(int16_t)(sw[k >> 1] >> ((k & 1) * 16));
                    v[k] = s >= 0 && live ? st_requant_line(tab, sh, wave, is, g, ch, sr, s) : 0.0;
                }
                if (ms && NCH == 2) {
#pragma unroll
                    for (int k = 0; k < 18; k++) {
                        const double o = shfl_xor_f64(v[k], 32);
                        v[k] = ch == 0 ? (v[k] + o) / tab.sqrt2 : (o - v[k]) / tab.sqrt2;
                    }
                }
                cse = -1;                                           // (its lines are in place)
            } else
                dec_requant_ms(tab, sh, wave, v, in, sr, ms, NCH, lane, bt, cse);
        }
        if (cse > 0) {
            // reorder of a short / mixed lane whose other channel is in another case: line k of this subband is line src of the spectrum
            // (or nothing), computed again from `is` -- requantised and, under MS stereo, combined with the other channel's line src
            const uint32_t *srcw = reinterpret_cast<const uint32_t *>(tab.reorder_src[sr] + sb * 18);
            uint32_t sw[9];
#pragma unroll
            for (int k = 0; k < 9; k++) sw[k] = srcw[k];
#pragma unroll
            for (int k = 0; k < 18; k++) {
                const int s = (int)(int16_t)(sw[k >> 1] >> ((k & 1) * 16));
                double x = 0.0;
                if (s >= 0 && live) {
                    x = st_requant_line(tab, sh, wave, is, g, ch, sr, s);
                    if (ms && NCH == 2) {
                        const double o = st_requant_line(tab, sh, wave, is, g, ch ^ 1, sr, s);
                        x = ch == 0 ? (x + o) / tab.sqrt2 : (o - x) / tab.sqrt2;
                    }
                }
                v[k] = x;
            }
        } else if (cse == 0) {
            // alias reduction (Frame.py:604-622): butterflies with line 17 - i of subband sb - 1 and line i of subband sb + 1, the
            // neighbours' lines by DPP wave shifts (both read before either is changed); the edge subbands keep their value
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const double nlo = dpp_f64<0x138>(v[17 - i]);   // wave_shr:1 = lane - 1's
                const double nhi = dpp_f64<0x130>(v[i]);        // wave_shl:1 = lane + 1's
                const double cs = tab.alias_cs[i], ca = tab.alias_ca[i];
                const double lo = v[i] * cs + nlo * ca;
                const double hi = v[17 - i] * cs - nhi * ca;
                v[i] = sb >= 1 ? lo : v[i];
                v[17 - i] = sb <= 30 ? hi : v[17 - i];
            }
        }
        // ---- G of this granule and channel: sum over the channel's subbands of sum_k |v[k]|; the guard's width for its slots
        double eps_t = 0.0;
        bool safe = true;
        if (!F32) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 18; k++) s += fabs(v[k]);
            s += dpp_f64<0xB1>(s);                              // quad_perm [1, 0, 3, 2]
            s += dpp_f64<0x4E>(s);                              // quad_perm [2, 3, 0, 1]
            s += dpp_f64<0x141>(s);                             // row_half_mirror
            s += dpp_f64<0x140>(s);                             // row_mirror: the sum of the lane's row of 16
            const double o = __hiloint2double(__builtin_amdgcn_ds_swizzle(__double2hiint(s), 0x401f), __builtin_amdgcn_ds_swizzle(__double2loint(s), 0x401f));   // xor 16: the channel's other row
            const double gc = s + o;
            double gx = gc > gm1 ? gc : gm1;
            gx = gx > gm2 ? gx : gm2;
            if (!(gx >= 0.0)) gx = __builtin_inf();             // NaN: nothing is safe
            eps_t = eps_k * gx;
            safe = __ballot(!(xb_k * gx < 2147483000.0)) == 0;  // (beyond int32 the kernel's conversion and the reference's wrapping one differ)
            gm2 = gm1; gm1 = gc;
        }

#if MP3S_ST_CLOCKS
        const unsigned long long ck_b = __builtin_readcyclecounter();
        ck_req += ck_b - ck_a;
#endif
        double S[18];
        if (bt != 2) {
            // ---- long windows.  The 36 outputs are 18 values and their mirror images (x[17-i] = -x[i], x[53-i] = x[i]), and the 18 are a
            //      DCT-IV of the 18 lines: y[n] = sum_k v[k] cos((2n+1)(2k+1) pi/72), rows 0..8 = y[9..17], rows 18..26 = -y[8..0].  In two
            //      halves (DevTables::imdct_rot / imdct_pq, derivation and error bound in mp3s_tables.cpp): nine rotations of the pairs
            //      (v[k], v[17-k]) into (p, q), then ten stages m of two nine-term sums P[m], Q[m] that give y[2m] = P + Q and
            //      y[2m-1] = P - Q -- 236 multiply-adds where the mirrored rows took 324.  Stages 9..5 give the rows that read the old tail,
            //      stages 4..0 the rows that write the new one.  A stage's 18 factors (scalar cache), its rows' window factors and tail
            //      values (LDS) are asked for one stage ahead, the wait for them at the TOP of a stage (k_dec_imdct).
            const double *wl = sh.win[bt];
            double p[9], q[9];
            {
                const Row18s r = load_row18s(tab.imdct_rot, 0);
#pragma unroll
                for (int k = 0; k < 9; k++) {
                    const double c = 2 * k < 8 ? r.a[2 * k] : (2 * k < 16 ? r.b[2 * k - 8] : r.c[2 * k - 16]);
                    const double sn = 2 * k + 1 < 8 ? r.a[2 * k + 1] : (2 * k + 1 < 16 ? r.b[2 * k + 1 - 8] : r.c[2 * k + 1 - 16]);
                    p[k] = __builtin_fma(v[17 - k], sn, v[k] * c);
                    q[k] = __builtin_fma(v[17 - k], c, -(v[k] * sn));
                }
            }
            struct StageIn { Row18s c; double w[4], t[4]; };
            // rows of stage m: the one of y[2m] first (none for m = 9), then the one of y[2m-1] (none for m = 0)
            auto stage_in = [&](auto mc) {
                constexpr int m = decltype(mc)::value;
                StageIn in;
                in.c = load_row18s(tab.imdct_pq, m);
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int n = h == 0 ? 2 * m : 2 * m - 1;
                    in.w[2 * h] = in.w[2 * h + 1] = in.t[2 * h] = in.t[2 * h + 1] = 0.0;
                    if (n < 0 || n > 17) continue;
                    if (n >= 9) {                                               // row i = n - 9 and its mirror 17 - i
                        const int i = n - 9;
                        in.w[2 * h] = wl[i]; in.w[2 * h + 1] = wl[17 - i];
                        in.t[2 * h] = tl[i * 64]; in.t[2 * h + 1] = tl[(17 - i) * 64];
                    } else {                                                    // row i = 26 - n and its mirror 53 - i
                        const int i = 26 - n;
                        in.w[2 * h] = wl[i]; in.w[2 * h + 1] = wl[53 - i];
                    }
                }
                return in;
            };
            auto stage = [&](auto mc, const StageIn &in) {
                constexpr int m = decltype(mc)::value;
                double P = 0.0, Q = 0.0;
#pragma unroll
                for (int k = 0; k < 9; k++) {
                    const double cp = k < 8 ? in.c.a[k] : in.c.b[0];
                    const double cq = k < 7 ? in.c.b[1 + k] : in.c.c[k - 7];
                    if (m != 9) P = __builtin_fma(p[k], cp, P);          // (P[9] = 0: its factors are zeros)
                    if (m != 0) Q = __builtin_fma(q[k], cq, Q);          // (Q[0] = 0)
                }
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int n = h == 0 ? 2 * m : 2 * m - 1;
                    if (n < 0 || n > 17) continue;
                    const double y = h == 0 ? P + Q : P - Q;
                    if (n >= 9) {
                        const int i = n - 9;
                        double xa = y * in.w[2 * h] + in.t[2 * h];
                        double xb = -y * in.w[2 * h + 1] + in.t[2 * h + 1];
                        if (i & 1) xa = flip(xa);
                        if ((17 - i) & 1) xb = flip(xb);
                        S[i] = xa; S[17 - i] = xb;
                    } else {
                        const int i = 26 - n;
                        tl[(i - 18) * 64] = -y * in.w[2 * h];
                        tl[(35 - i) * 64] = -y * in.w[2 * h + 1];
                    }
                }
            };
            // (stage by stage, written out: the stage number is a compile-time constant of every one)
#define MP3S_ST_STAGE(M, MNEXT)                                                                 \
            {                                                                                   \
                __builtin_amdgcn_s_waitcnt(0xc07f);                                             \
                __builtin_amdgcn_sched_barrier(0);                                              \
                const StageIn nxt = stage_in(std::integral_constant<int, MNEXT>{});             \
                __builtin_amdgcn_sched_barrier(0);                                              \
                stage(std::integral_constant<int, M>{}, cur);                                   \
                __builtin_amdgcn_sched_barrier(0);                                              \
                cur = nxt;                                                                      \
            }
            StageIn cur;
            if (!tail_only) {
                cur = stage_in(std::integral_constant<int, 9>{});
                MP3S_ST_STAGE(9, 8) MP3S_ST_STAGE(8, 7) MP3S_ST_STAGE(7, 6) MP3S_ST_STAGE(6, 5) MP3S_ST_STAGE(5, 4)
            } else
                cur = stage_in(std::integral_constant<int, 4>{});
            MP3S_ST_STAGE(4, 3) MP3S_ST_STAGE(3, 2) MP3S_ST_STAGE(2, 1) MP3S_ST_STAGE(1, 0) MP3S_ST_STAGE(0, 0)
#undef MP3S_ST_STAGE
        } else {
            // ---- three 12-point windows placed at 6 / 12 / 18 (Frame.py:135-148), the reference's order, walked over the 12 rows as
            //      k_dec_imdct does
            typedef double dvec4 __attribute__((ext_vector_type(4)));
            struct Row6 { dvec4 a; dvec2 b; double s; };
            auto load_row6 = [&](int j) { Row6 r; r.a = *reinterpret_cast<const dvec4 *>(C12[j]); r.b = *reinterpret_cast<const dvec2 *>(C12[j] + 4); r.s = tab.sine_block[2][j]; return r; };
            // (the tail of the granule before is read row by row where it is used and the new one written where it is complete: a row of
            // the new tail never lands on a row of the old one that is still to be read -- rows 0..5 are read first, row 6 + j at step j,
            // and step j writes rows j >= 6 and 6 + j; six sums wait in registers.  Eighteen registers instead of eighty.)
            double hold[6], tn[6];
#pragma unroll
            for (int i = 0; i < 6; i++) { double x = 0.0 + (double)tl[i * 64]; if (i & 1) x = flip(x); S[i] = x; }    // sample_block[0..5] = 0
            Row6 cur = load_row6(0);
#pragma unroll
            for (int j = 0; j < 12; j++) {
                const Row6 nxt = load_row6(j < 11 ? j + 1 : 11);
                const double pj = tl[(6 + j) * 64];
                __builtin_amdgcn_sched_barrier(0);
                double a0 = 0.0, a1 = 0.0, a2 = 0.0;
#pragma unroll
                for (int k = 0; k < 6; k++) {
                    const double c = k < 4 ? cur.a[k] : cur.b[k - 4];
                    a0 += v[k] * c; a1 += v[6 + k] * c; a2 += v[12 + k] * c;
                }
                a0 = a0 * cur.s; a1 = a1 * cur.s; a2 = a2 * cur.s;
                if (j < 6) {
                    double x = a0 + pj;                                 // sample_block[6..11] = t[0..5]
                    if ((6 + j) & 1) x = flip(x);
                    S[6 + j] = x;
                    hold[j] = a1;                                       // t[12..17]
                    tn[j] = a2;                                         // t[24..29], half of sample_block[18..23]
                } else {
                    double x = (a0 + hold[j - 6]) + pj;                 // sample_block[12..17] = t[6..11] + t[12..17]
                    if ((6 + j) & 1) x = flip(x);
                    S[6 + j] = x;
                    tn[j - 6] = a1 + tn[j - 6];                         // sample_block[18..23] = t[18..23] + t[24..29]
                    tl[j * 64] = a2;                                    // sample_block[24..29] = t[30..35]
                    tl[(6 + j) * 64] = 0.0;                             // sample_block[30..35]
                }
                __builtin_amdgcn_sched_barrier(0);
                cur = nxt;
            }
#pragma unroll
            for (int i = 0; i < 6; i++) tl[i * 64] = tn[i];
        }
        // the next granule's lines and side records: asked for here, behind the rows (their registers are wanted there), under the synthesis
        have_next = gi + 1 < run && g + 1 < n_granules;
        if (have_next) next_in = dec_fetch(is, si, g + 1, NCH, lane);
#if MP3S_ST_CLOCKS
        const unsigned long long ck_c = __builtin_readcyclecounter();
        ck_rows += ck_c - ck_b;
#endif
        if (tail_only) continue;

        // ---- synthesis of the granule's 18 slots, in time order
        const long t0 = (long)g * 18;                                           // the granule's first slot
        const bool store = emit && t0 >= halo_slots;                            // (a halo is whole frames)
        int16_t *const o16 = reinterpret_cast<int16_t *>(pcm_out) + ((t0 - halo_slots) * 32 + sb) * NCH + ch;
        float *const o32 = reinterpret_cast<float *>(pcm_out) + ((t0 - halo_slots) * 32 + sb) * NCH + ch;
        double cx[16], wt[16];
#pragma unroll
        for (int j = 0; j < 16; j++) { cx[j] = sh.cx[j][sb]; wt[j] = sh.wt[j][sb]; }
        // slot p's X: mirror subband by swizzle, difference | sum, the row's 16 x 16 products
        auto mirror = [&](double s) { return __hiloint2double(__builtin_amdgcn_ds_swizzle(__double2hiint(s), 0x7c1f), __builtin_amdgcn_ds_swizzle(__double2loint(s), 0x7c1f)); };
        auto xsum = [&](double s, double m) {
            const double ms2 = __hiloint2double(__double2hiint(m) ^ (int)sgn_lo, __double2loint(m));
            return st_row_dot16(s + ms2, cx);
        };
        // Slot p's A and B and slot p + 1's mirror are asked for one slot AHEAD: under the sixteen multiply-adds of the slot in front
        // (LDS operations of a wave complete in order: when the mirror a slot's X needs is here, so are the A and B asked for before it)
        double mir = mirror(S[0]);
        double X = xsum(S[0], mir);
        int a_lo = __builtin_amdgcn_ds_bpermute(addr_a, __double2loint(X)), a_hi = __builtin_amdgcn_ds_bpermute(addr_a, __double2hiint(X));
        int b_lo = __builtin_amdgcn_ds_bpermute(addr_b, __double2loint(X)), b_hi = __builtin_amdgcn_ds_bpermute(addr_b, __double2hiint(X));
        mir = mirror(S[1]);
#pragma unroll
        for (int p = 0; p < 18; p++) {
            const double A = __hiloint2double(a_hi, a_lo), B = __hiloint2double(b_hi, b_lo);
            if (p < 17) {
                X = xsum(S[p + 1], mir);
                a_lo = __builtin_amdgcn_ds_bpermute(addr_a, __double2loint(X)); a_hi = __builtin_amdgcn_ds_bpermute(addr_a, __double2hiint(X));
                b_lo = __builtin_amdgcn_ds_bpermute(addr_b, __double2loint(X)); b_hi = __builtin_amdgcn_ds_bpermute(addr_b, __double2hiint(X));
                if (p < 16) mir = mirror(S[p + 2]);
            }
            // the 16 sums this slot belongs to: lag jj ahead, its even lags take A, the odd ones B; lag 15 opens a sum
#pragma unroll
            for (int jj = 0; jj < 16; jj++) {
                const int q = (p + jj) % 18;
                const double val = (jj & 1) ? B : A;
                acc[q] = jj == 15 ? wt[15] * val : __builtin_fma(wt[jj], val, acc[q]);
            }
            const double x = acc[p];                                            // complete: this slot's own term was its last
            if (F32) {
                if (store && live) o32[p * 32 * NCH] = (float)x;
            } else {
                // the guard (k_dec_synth_fast): is the truncation of x = sample * 32767 beyond doubt?  distance to the nearest non-zero integer
                const double xi = rint(x);
                const double dist = fabs(x) - fmax(fabs(xi), 1.0);
                if (store) {
                    if (live) o16[p * 32 * NCH] = (int16_t)(int)x;
                    if (MARGIN && live) {
                        const long at = mg_base + ((t0 - halo_slots + p) * 32 + sb) * NCH + ch;
                        if (at >= 0 && at < mg_cap) { mg_x[at] = x; mg_eps[at] = safe ? eps_t : __builtin_inf(); }
                    }
                    unsigned long long m = safe ? __ballot(!(fabs(dist) > eps_t)) : ~0ull;
                    if (NCH == 1) m &= 0xffffffffull;
                    if (__builtin_expect(m != 0, 0)) {                          // (wave-uniform, rare: out of line)
                        if (lane == 0) {
                            const uint32_t m0 = (uint32_t)m, m1 = (uint32_t)(m >> 32);
                            const int n = (m0 != 0) + (m1 != 0);
                            int at = atomicAdd(fix_count, n);
                            if (m0) fix_list[at++] = make_uint2((uint32_t)(t0 + p), m0);
                            if (m1) fix_list[at] = make_uint2((uint32_t)(t0 + p) | 0x80000000u, m1);
                        }
                    }
                }
            }
        }
#if MP3S_ST_CLOCKS
        ck_syn += __builtin_readcyclecounter() - ck_c;
#endif
    }
#if MP3S_ST_CLOCKS
    if (lane == 0) {
        uint32_t *d = g_st_clocks + ((size_t)(xcd_tile() * ST_WAVES + wave) & 4095) * 8;
        d[0] = (uint32_t)ck_setup; d[1] = (uint32_t)ck_req; d[2] = (uint32_t)ck_rows; d[3] = (uint32_t)ck_syn;
        d[4] = (uint32_t)(__builtin_readcyclecounter() - ck_start);
    }
#endif
}

}  // namespace mp3s
