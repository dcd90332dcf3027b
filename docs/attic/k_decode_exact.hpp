// NOT BUILT -- kept as the record of a round-6 experiment (docs/LOG.md): the decode transform of the FLOAT formats as one
// wave-local stream in the REFERENCE's order (gfx950).  Bit-identical float64 / float32 PCM to the two-kernel path in every form tried (sha256
// over the 10 000-frame batch, tests/test_gpu_parity.py + test_decode_corpus.py + test_streaming.py green through it), and slower:
//   slot loop unrolled over a granule's 18 slots, V history as a ring of 18 register pairs   0.680 ms  (80 KB of code per kernel, 62-142 spilled registers)
//   slot loop rolled over pairs of slots, history as four chains of eight (this file's form)  0.418 ms  (30 spilled registers, all outside the slot loop)
//   both slots of a pair through one pass over the coefficient columns                        1.56  ms  (180 spilled registers)
//   k_dec_imdct<false> + k_dec_synth (the product)                                            0.357 ms
// Why: with lane = (channel, subband) every lane needs its own 64 matrix coefficients per slot -- 16 KB of LDS reads per slot and wave where
// k_dec_synth (lane = slot) takes them as scalar operands --, doubles have no DPP multiply (a v_mov_b64 row_newbcast per column on top of the 128
// multiplies and adds), and 36 + 64 registers of overlap and history leave two waves per SIMD nothing to hide the chains behind.  To build it again:
// #include it behind k_decode_stream.hpp and launch k_dec_exact<NCH, F32> from launch_decode for the float formats (tools/dec_only.py times both).
//
//
// Rounds 1-5 ran requantise .. IMDCT (k_dec_imdct<false>) and the synthesis filter bank (k_dec_synth) as two kernels with the time-domain
// subband samples S -- float64, 18 KB per frame -- written to and read from device memory between them: 368 MB per 10 000-frame batch, two
// lane mappings (subband / time slot), 103 + 207 us alone.  This kernel keeps lane = (channel, subband) from the spectrum to the PCM, as the
// int16 stream kernel does (k_decode_stream.hpp), but forms every sum exactly as the reference does (decoder/Frame.py:65-154, 561-640):
//
//   requantise, MS stereo, reorder | alias reduction          dec_prepare (shared with the two-kernel path)
//   IMDCT, window, overlap, frequency inversion               imdct_rows_exact (shared): 36 sums of 18 separately rounded products per block
//   matrixing  V[i] = sum_j S[j] N[i][j], j ascending          Frame.py:82-85.  Lane (ch, sb) forms V[sb] and V[32 + sb]: S[j] of its own row of
//                                                             16 lanes comes by DPP (v_mov_b64 row_newbcast:j -- multiply and add have no DPP form
//                                                             for doubles), the other row's through one ds_swizzle of the slot's sample; the
//                                                             lane's two coefficients of column j are one 16-byte LDS read; product and sum are
//                                                             separate instructions (-ffp-contract=off), the sum starts at +0.0
//   window     pcm[i] = sum_k V(t - k)[i | 32 + i] D[32 k + i]  Frame.py:87-101: the V pairs of the last 18 slots stay in registers (a ring indexed
//                                                             by the slot's place in its granule: compile-time, a granule is 18 slots > 15 lags)
//
// No S array, no workgroup barrier behind the table staging.  A run of granules is primed with the granule in front (its V history) and the
// tail of the one before.  float64 / float32 PCM bit-identical to the two-kernel path (tests/test_gpu_parity.py, test_decode_corpus.py run
// through this kernel by default; MP3S_OPT_FUSED_DECODE = 0 restores the two kernels).
#pragma once

namespace mp3s {

constexpr int EX_WAVES = DEC_A_WAVES;

struct ExShared {
    DecShared d;                          // exponent tables, windows, the waves' exchange buffers (dec_prepare)
    double nm[32][32][2];                 // [j][sb] = {synth_matrix[sb][j], synth_matrix[32 + sb][j]}: a lane's two coefficients of column j
    double dw[16][32];                    // [k][i] = synth_window[32 k + i]
};

template <int J>
__device__ __forceinline__ double ex_bcast(double v)          // lane J of the lane's row of 16, for every lane of the row
{
    double r;
    asm("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(J));
    return r;
}

template <int NCH, bool F32>
__global__ __launch_bounds__(EX_WAVES * 64, 2) void k_dec_exact(
    const int16_t *__restrict__ is, const mp3s_granule_si *__restrict__ si, const mp3s_frame_hdr *__restrict__ hdr,
    int n_granules, int run, int n_halo, void *__restrict__ pcm_out, int sf_base)
{
    __shared__ ExShared sh;
    dec_stage_tables(sh.d);
    for (int i = threadIdx.x; i < 32 * 32; i += blockDim.x) {
        const int j = i >> 5, sb = i & 31;
        sh.nm[j][sb][0] = c_tab.synth_matrix[sb][j];
        sh.nm[j][sb][1] = c_tab.synth_matrix[32 + sb][j];
    }
    for (int i = threadIdx.x; i < 512; i += blockDim.x) sh.dw[i >> 5][i & 31] = c_tab.synth_window[i];
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int ga = (xcd_tile() * EX_WAVES + wave) * run;      // this wave's granules: ga .. ga + run - 1
    if (ga >= n_granules) return;                              // (whole waves; no barrier behind this line)
    const int ch = lane >> 5, sb = lane & 31;
    const bool live = ch < NCH;
    const bool low_row = sb < 16;
    const uint32_t sgn_odd = (sb & 1) ? 0x80000000u : 0u;     // frequency inversion (Frame.py:629-631): odd slots of odd subbands
    const long halo_slots = (long)n_halo * 36;

    // IMDCT overlap; V[sb] (x0) and V[32 + sb] (x1) of the last sixteen slots as four chains of eight: a slot's window takes V[sb] of the slots an even
    // number of slots back and V[32 + sb] of those an odd number back (below), so the values of even and of odd slots never meet in one sum --
    // e0 / e1 = x0 / x1 of the even slots (newest first), o0 / o1 of the odd ones; a slot shifts the two chains of its own parity (14 moves) and the
    // slot loop is rolled over PAIRS of slots (the first form kept a ring indexed by the slot and was unrolled over a granule's 18: 80 KB of code per
    // kernel, three times the two kernels' time)
    double tail[18], e0[8], e1[8], o0[8], o1[8];
#pragma unroll
    for (int i = 0; i < 18; i++) tail[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 8; i++) { e0[i] = 0.0; e1[i] = 0.0; o0[i] = 0.0; o1[i] = 0.0; }
    // the granule's 18 samples of this lane wait in the wave's exchange buffer (free between two granules' requantisation): [slot][lane]
    typedef double __attribute__((address_space(3))) lds_f64;
    lds_f64 *const sl = (lds_f64 *)reinterpret_cast<double *>(&sh.d.buf[wave][0][0]) + lane;
    static_assert(sizeof(sh.d.buf[0]) >= 18 * 64 * sizeof(double), "a granule's samples fit the wave's exchange buffer");
    // the stream the run starts in begins at granule s_a of the launch: nothing in front of it primes the run
    const int s_a = [&] { const uint32_t sf = hdr[ga >> 1].stream_first; return sf > (uint32_t)sf_base ? (int)(sf - (uint32_t)sf_base) * 2 : 0; }();
    const int gi0 = ga - 2 >= s_a ? -2 : (ga - 1 >= s_a ? -1 : 0);
    GranIn next_in = {};
    bool have_next = false;
#pragma unroll 1
    for (int gi = gi0; gi < run; gi++) {
        const int g = ga + gi;
        if (g >= n_granules) break;
        const mp3s_frame_hdr fh = hdr[g >> 1];
        const int first_gran = fh.stream_first > (uint32_t)sf_base ? (int)(fh.stream_first - (uint32_t)sf_base) * 2 : 0;
        if (g == first_gran) {                                 // Frame.py:234-235: prev_samples and the fifo start as zeros
#pragma unroll
            for (int i = 0; i < 18; i++) tail[i] = 0.0;
#pragma unroll
            for (int i = 0; i < 8; i++) { e0[i] = 0.0; e1[i] = 0.0; o0[i] = 0.0; o1[i] = 0.0; }
        }
        const int sr = fh.sr_idx < 3 ? fh.sr_idx : 0;
        const bool tail_only = gi == -2;                       // two granules in front of the run: only its overlap tail is needed
        // the twiddle tables are invariant over this loop: an opaque zero offset per granule keeps their scalar loads inside it (k_dec_imdct)
        int zoff = 0;
        asm volatile("" : "+s"(zoff));
        const DevTables &tab = *reinterpret_cast<const DevTables *>(reinterpret_cast<const char *>(&c_tab) + zoff);
        {
            double v[18];
            int bt;
            const GranIn in = have_next ? next_in : dec_fetch(is, si, g, NCH, lane);
            dec_prepare(tab, sh.d, wave, v, in, sr, fh.ms_stereo != 0, NCH, lane, bt);
            have_next = gi + 1 < run && g + 1 < n_granules;
            if (have_next) next_in = dec_fetch(is, si, g + 1, NCH, lane);
            __builtin_amdgcn_wave_barrier();                   // (every lane has read its neighbours' lines: the buffer takes the samples now)
            imdct_rows_exact(tab, sh.d, v, bt, tail, !tail_only, sgn_odd, [&](int i, double x) { sl[i * 64] = x; });
        }
        if (tail_only) continue;
        __builtin_amdgcn_s_waitcnt(0xc07f);

        // ---- synthesis filter bank (Frame.py:65-101) of the granule's 18 slots, in time order, two slots per trip
        const long t0 = (long)g * 18;
        const bool store = gi >= 0 && t0 >= halo_slots && live;    // (a halo is whole frames)
        const bool window = gi >= 0;                               // (the granule in front of the run: its V history only)
        double *o64 = reinterpret_cast<double *>(pcm_out) + ((t0 - halo_slots) * 32 + sb) * NCH + ch;
        float *o32 = reinterpret_cast<float *>(pcm_out) + ((t0 - halo_slots) * 32 + sb) * NCH + ch;
        typedef double dvec2e __attribute__((ext_vector_type(2)));
        const double *const nms = &sh.nm[0][sb][0], *const dws = &sh.dw[0][sb];
        // two slots per trip (an even one and the odd one behind it): their samples -> V[sb], V[32 + sb] of both (one pass over the 32 columns: a
        // column's two coefficients are read once for both slots) -> the chains of each slot's parity -> each slot's window sum
        auto bring = [&](int p, double &lo, double &hi) {
            // the channel's 32 samples of the slot as two registers every lane of a row can broadcast from: lo = S[lane & 15], hi = S[16 + (lane & 15)]
            const double s = sl[p * 64];
            const double so = __hiloint2double(__builtin_amdgcn_ds_swizzle(__double2hiint(s), 0x401f), __builtin_amdgcn_ds_swizzle(__double2loint(s), 0x401f));   // xor 16
            lo = low_row ? s : so; hi = low_row ? so : s;
        };
        auto finish = [&](int p, double x0, double x1, double (&mine0)[8], double (&mine1)[8], const double (&theirs1)[8]) {
#pragma unroll
            for (int k = 7; k > 0; k--) { mine0[k] = mine0[k - 1]; mine1[k] = mine1[k - 1]; }
            mine0[0] = x0; mine1[0] = x1;
            if (!window) return;
            // lag k takes V[i] of slot t - k when k is even, V[32 + i] when it is odd (u[64 a + i] = fifo[128 a + i], u[64 a + 32 + i] =
            // fifo[128 a + 96 + i]: Frame.py:87-93); the sum ascends in k from +0.0
            double sum = 0.0;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const double val = (k & 1) ? theirs1[k >> 1] : mine0[k >> 1];
                const double w = val * dws[k * 32];
                sum = sum + w;
            }
            if (store) {
                if (F32) o32[(long)p * 32 * NCH] = (float)sum;
                else o64[(long)p * 32 * NCH] = sum;
            }
        };
#pragma unroll 1
        for (int pp = 0; pp < 9; pp++) {
            double lo_a, hi_a, lo_b, hi_b;
            bring(2 * pp, lo_a, hi_a);
            bring(2 * pp + 1, lo_b, hi_b);
            asm volatile("s_nop 1" : "+v"(lo_a), "+v"(hi_a), "+v"(lo_b), "+v"(hi_b));   // (a vector write needs two wait states before a DPP read: once per trip)
            double xa0 = 0.0, xa1 = 0.0, xb0 = 0.0, xb1 = 0.0;
            auto col = [&](double ba, double bb, int j) {
                const dvec2e n = *reinterpret_cast<const dvec2e *>(nms + j * 64);
                const double pa0 = ba * n.x, pa1 = ba * n.y, pb0 = bb * n.x, pb1 = bb * n.y;
                xa0 = xa0 + pa0; xa1 = xa1 + pa1; xb0 = xb0 + pb0; xb1 = xb1 + pb1;
            };
            // (columns in groups of eight: the scheduler may not gather all 32 coefficient reads -- 128 registers -- in front of their use)
#define EX_BAR __builtin_amdgcn_sched_barrier(0);
#define EX_COL(J) col(ex_bcast<J>(lo_a), ex_bcast<J>(lo_b), J);
            EX_BAR EX_COL(0) EX_COL(1) EX_COL(2) EX_COL(3) EX_COL(4) EX_COL(5) EX_COL(6) EX_COL(7)
            EX_BAR EX_COL(8) EX_COL(9) EX_COL(10) EX_COL(11) EX_COL(12) EX_COL(13) EX_COL(14) EX_COL(15)
#undef EX_COL
#define EX_COL(J) col(ex_bcast<J>(hi_a), ex_bcast<J>(hi_b), 16 + J);
            EX_BAR EX_COL(0) EX_COL(1) EX_COL(2) EX_COL(3) EX_COL(4) EX_COL(5) EX_COL(6) EX_COL(7)
            EX_BAR EX_COL(8) EX_COL(9) EX_COL(10) EX_COL(11) EX_COL(12) EX_COL(13) EX_COL(14) EX_COL(15)
            EX_BAR
#undef EX_COL
#undef EX_BAR
            finish(2 * pp, xa0, xa1, e0, e1, o1);              // (18 slots per granule: a slot's parity in its granule is its parity in the stream)
            finish(2 * pp + 1, xb0, xb1, o0, o1, e1);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();                       // (the samples have been read: the buffer is the next granule's exchange buffer again)
    }
}

}  // namespace mp3s
