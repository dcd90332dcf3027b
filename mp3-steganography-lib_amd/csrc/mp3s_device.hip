// Single device translation unit: the constant-table symbol, the three kernel groups and their
// launchers.  Compiled for gfx950 only (hipcc --offload-arch=gfx950 -ffp-contract=off).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <atomic>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/mp3s.h"
#include "mp3s_device.h"
#include "mp3s_host.h"
#include "mp3s_tables.h"

namespace mp3s {
__constant__ DevTables c_tab;
}

#include "k_sync.hpp"
#include "k_decode.hpp"
#include "k_decode_stream.hpp"
#include "k_encode.hpp"
#include "k_rate.hpp"
#include "k_huffman.hpp"
#include "k_parse.hpp"
#include "k_pack.hpp"
#include "k_chain.hpp"

namespace mp3s {

int Profiler::begin(hipStream_t s, int k)
{
    if (!enabled || !((mask >> k) & 1u)) return -1;
    if (n_pairs >= MAX_PAIRS) { dropped++; return -1; }
    while (n_created < 2 * (n_pairs + 1)) {
        if (hipEventCreate(&ev[n_created]) != hipSuccess) { dropped++; return -1; }
        n_created++;
    }
    kid[n_pairs] = k;
    (void)hipEventRecord(ev[2 * n_pairs], s);
    return n_pairs++;
}
void Profiler::end(hipStream_t s, int pair)
{
    if (pair >= 0) (void)hipEventRecord(ev[2 * pair + 1], s);
}
bool Profiler::attach(int k, hipEvent_t *start, hipEvent_t *stop)
{
    if (!enabled || !((mask >> k) & 1u)) return false;
    if (n_pairs >= MAX_PAIRS) { dropped++; return false; }
    while (n_created < 2 * (n_pairs + 1)) {
        if (hipEventCreate(&ev[n_created]) != hipSuccess) { dropped++; return false; }
        n_created++;
    }
    kid[n_pairs] = k;
    *start = ev[2 * n_pairs]; *stop = ev[2 * n_pairs + 1];
    n_pairs++;
    return true;
}

int dev_upload_tables(hipStream_t stream)
{
    const HostTables &h = host_tables();
    hipError_t e = hipMemcpyToSymbolAsync(HIP_SYMBOL(c_tab), &h.dev, sizeof(DevTables), 0, hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return (int)e;
    return (int)hipStreamSynchronize(stream);
}

// decode scratch: S (float64 [ch][slot][32]) | G (float64 [granule][2], left by the fast IMDCT for the guard of the
// fast synthesis) | fix-up list (one entry per slot and channel at most)
static size_t dec_S_bytes(int n_frames, int nch) { return (size_t)nch * (size_t)n_frames * 36 * 32 * sizeof(double); }
static size_t dec_G_bytes(int n_frames) { return (size_t)n_frames * 2 * 2 * sizeof(double); }
size_t dec_scratch_bytes(int n_frames, int nch)
{
    return dec_S_bytes(n_frames, nch) + dec_G_bytes(n_frames) + (size_t)nch * (size_t)n_frames * 36 * sizeof(uint2);
}

int launch_decode(hipStream_t stream, const int16_t *d_is, const mp3s_granule_si *d_si, const mp3s_frame_hdr *d_hdr,
                  int n_frames, int nch, int n_halo, int out_format, void *d_pcm, void *d_scratch, Profiler *prof, int sf_base,
                  double synth_eps_scale, int32_t *d_sync, bool fast_imdct, bool float_fast, bool fused, hipEvent_t done, const GuardProbe *probe)
{
    const long T = (long)n_frames * 36;
    double *S = (double *)d_scratch;
    double *G = (double *)((char *)d_scratch + dec_S_bytes(n_frames, nch));
    uint2 *fix_list = (uint2 *)((char *)G + dec_G_bytes(n_frames));
    const int n_gran = n_frames * 2;
    // int16 output through the fast kernels (guarded; needs the context's counters), everything else in the reference's order
    const bool fast = out_format == MP3S_PCM_I16 && synth_eps_scale > 0 && d_sync;
    // float32 output through the same fast sums, unguarded (MP3S_OPT_FLOAT_FAST: within 1e-5 of the reference, not bit-identical)
    const bool fast32 = out_format == MP3S_PCM_F32 && float_fast;
    // the fast paths as ONE kernel, a wave-local stream with no intermediate array (k_decode_stream.hpp); timed as the synthesis
    if (fused && ((fast && fast_imdct) || fast32)) {
        // granules per wave: a run is primed with a granule and a half of work in front of it, so longer runs waste less; short enough that
        // the waves fill the chip's slots (two waves per SIMD: 2 048) in whole rounds
        int run = 16;
        {
            double best = 1e30;
            for (int r = 4; r <= 16; r++) {
                const long waves = (n_gran + r - 1) / r;
                const long rounds = (waves + 2047) / 2048;
                const double cost = (double)rounds * (r + 1.6);
                if (cost < best) { best = cost; run = r; }
            }
        }
        const int runs = (n_gran + run - 1) / run;
        const dim3 grid((runs + ST_WAVES - 1) / ST_WAVES), block(ST_WAVES * 64);
        const int pf = prof ? prof->begin(stream, K_DEC_SYNTH) : -1;
        if (fast32) {
            if (nch == 2) hipLaunchKernelGGL((k_dec_stream<2, true>), grid, block, 0, stream, d_is, d_si, d_hdr, n_gran, run, n_halo, d_pcm, sf_base, 1.0, (uint2 *)nullptr, (int32_t *)nullptr, (double *)nullptr, (double *)nullptr, 0L, 0L);
            else hipLaunchKernelGGL((k_dec_stream<1, true>), grid, block, 0, stream, d_is, d_si, d_hdr, n_gran, run, n_halo, d_pcm, sf_base, 1.0, (uint2 *)nullptr, (int32_t *)nullptr, (double *)nullptr, (double *)nullptr, 0L, 0L);
        } else {
            if (probe && probe->x) {     // (mp3s_debug_guard_margin: the probe's instantiation of the same kernel)
                if (nch == 2) hipLaunchKernelGGL((k_dec_stream<2, false, true>), grid, block, 0, stream, d_is, d_si, d_hdr, n_gran, run, n_halo, d_pcm, sf_base, synth_eps_scale, fix_list, d_sync + 6, probe->x, probe->eps, (long)probe->base, (long)probe->cap);
                else hipLaunchKernelGGL((k_dec_stream<1, false, true>), grid, block, 0, stream, d_is, d_si, d_hdr, n_gran, run, n_halo, d_pcm, sf_base, synth_eps_scale, fix_list, d_sync + 6, probe->x, probe->eps, (long)probe->base, (long)probe->cap);
            } else if (nch == 2) hipLaunchKernelGGL((k_dec_stream<2, false>), grid, block, 0, stream, d_is, d_si, d_hdr, n_gran, run, n_halo, d_pcm, sf_base, synth_eps_scale, fix_list, d_sync + 6, (double *)nullptr, (double *)nullptr, 0L, 0L);
            else hipLaunchKernelGGL((k_dec_stream<1, false>), grid, block, 0, stream, d_is, d_si, d_hdr, n_gran, run, n_halo, d_pcm, sf_base, synth_eps_scale, fix_list, d_sync + 6, (double *)nullptr, (double *)nullptr, 0L, 0L);
            const int fix_groups = (int)((T * nch + DEC_A_WAVES - 1) / DEC_A_WAVES) < 128 ? (int)((T * nch + DEC_A_WAVES - 1) / DEC_A_WAVES) : 128;
            if (done)
                hipExtLaunchKernelGGL(k_dec_fixup, dim3(fix_groups), dim3(DEC_A_WAVES * 64), 0, stream, nullptr, done, 0, d_is, d_si, d_hdr, n_gran, nch, T, n_halo,
                                      sf_base, (int16_t *)d_pcm, (const uint2 *)fix_list, d_sync + 6, d_sync + 2);
            else
                hipLaunchKernelGGL(k_dec_fixup, dim3(fix_groups), dim3(DEC_A_WAVES * 64), 0, stream, d_is, d_si, d_hdr, n_gran, nch, T, n_halo,
                                   sf_base, (int16_t *)d_pcm, (const uint2 *)fix_list, d_sync + 6, d_sync + 2);
            done = nullptr;                                  // (signalled by the dispatch itself)
        }
        if (prof) prof->end(stream, pf);
        if (done && hipEventRecord(done, stream) != hipSuccess) return (int)hipErrorUnknown;
        return (int)hipGetLastError();
    }
    // granules per wave: each wave also primes itself with half an IMDCT of the granule before its run, so longer
    // runs waste less; pick the run length (2..8) that fills whole rounds of the chip's wave slots best (168 VGPRs ->
    // 3 waves per SIMD -> 256 CUs x 12 = 3072 slots)
    int run = 8;
    {
        double best = 1e30;
        for (int r = 2; r <= 8; r++) {
            const long waves = (n_gran + r - 1) / r;
            const long rounds = (waves + 3071) / 3072;
            const double cost = (double)rounds * (r + 0.55);
            if (cost < best) { best = cost; run = r; }
        }
    }
    const int runs = (n_gran + run - 1) / run;
    int pp = prof ? prof->begin(stream, K_DEC_IMDCT) : -1;
    if ((fast && fast_imdct) || fast32)
        hipLaunchKernelGGL(k_dec_imdct<true>, dim3((runs + DEC_A_WAVES - 1) / DEC_A_WAVES), dim3(DEC_A_WAVES * 64), 0, stream,
                           d_is, d_si, d_hdr, n_gran, nch, run, S, T, sf_base, fast32 ? (double *)nullptr : G);
    else
        hipLaunchKernelGGL(k_dec_imdct<false>, dim3((runs + DEC_A_WAVES - 1) / DEC_A_WAVES), dim3(DEC_A_WAVES * 64), 0, stream,
                           d_is, d_si, d_hdr, n_gran, nch, run, S, T, sf_base, (double *)nullptr);
    if (prof) prof->end(stream, pp);
    constexpr int TW = DEC_SYNTH_TW;
    pp = prof ? prof->begin(stream, K_DEC_SYNTH) : -1;
    if (fast32) {
        constexpr int W = DEC_SYNTH_FAST_TW;
        hipLaunchKernelGGL((k_dec_synth_fast<W, true>), dim3((unsigned)((T + (W * 64 - 15) - 1) / (W * 64 - 15))), dim3(W * 64 * nch), 0, stream,
                           (const double *)S, T, d_hdr, nch, n_halo, (int16_t *)d_pcm, sf_base, 1.0, (const double *)nullptr, n_gran, (uint2 *)nullptr, (int32_t *)nullptr);
    } else if (fast) {
        constexpr int ftw = DEC_SYNTH_FAST_TW;
        const double *Gk = fast_imdct ? G : nullptr;
#define MP3S_FAST_LAUNCH(W)                                                                                                        \
        hipLaunchKernelGGL(k_dec_synth_fast<W>, dim3((unsigned)((T + (W * 64 - 15) - 1) / (W * 64 - 15))), dim3(W * 64 * nch), 0, stream, \
                           (const double *)S, T, d_hdr, nch, n_halo, (int16_t *)d_pcm, sf_base, synth_eps_scale, Gk, n_gran, fix_list, d_sync + 6)
        MP3S_FAST_LAUNCH(ftw);
#undef MP3S_FAST_LAUNCH
        // the samples the guard could not vouch for, again from `is` in the reference's order (a handful per batch)
        const int fix_groups = (int)((T * nch + DEC_A_WAVES - 1) / DEC_A_WAVES) < 128 ? (int)((T * nch + DEC_A_WAVES - 1) / DEC_A_WAVES) : 128;
        hipLaunchKernelGGL(k_dec_fixup, dim3(fix_groups), dim3(DEC_A_WAVES * 64), 0, stream, d_is, d_si, d_hdr, n_gran, nch, T, n_halo,
                           sf_base, (int16_t *)d_pcm, (const uint2 *)fix_list, d_sync + 6, d_sync + 2);
    } else {
        const int out_per_tile = TW * 64 - 15;
        const int tiles = (int)((T + out_per_tile - 1) / out_per_tile);
        hipLaunchKernelGGL(k_dec_synth<TW>, dim3(tiles), dim3(TW * 64 * nch), 0, stream, (const double *)S, T, d_hdr, nch,
                           n_halo, out_format, d_pcm, sf_base);
    }
    if (prof) prof->end(stream, pp);
    if (done && hipEventRecord(done, stream) != hipSuccess) return (int)hipErrorUnknown;
    return (int)hipGetLastError();
}

size_t enc_scratch_bytes(int n_frames)
{
    return (size_t)2 * 32 * (size_t)n_frames * 36 * sizeof(int32_t);
}

int launch_encode(hipStream_t stream, const int16_t *d_pcm, const mp3s_frame_hdr *d_hdr, int n_frames, int32_t *d_mdct,
                  void *d_scratch, Profiler *prof, bool fused)
{
    if (fused) {
        // analysis and MDCT as one kernel, the subband samples in LDS (k_enc_fused); timed as the analysis
        const long n_gran = (long)n_frames * 2;
        const int pf = prof ? prof->begin(stream, K_ENC_ANALYSIS) : -1;
        hipLaunchKernelGGL(k_enc_fused, dim3((unsigned)((n_gran + EF_GR - 1) / EF_GR)), dim3(EF_WAVES * 64), 0, stream, d_pcm, d_hdr, n_frames, d_mdct);
        if (prof) prof->end(stream, pf);
        return (int)hipGetLastError();
    }
    const long Ts = (long)n_frames * 36;
    int32_t *SB = (int32_t *)d_scratch;
    const long waves = 2 * ((Ts + 63) / 64);
    int pp = prof ? prof->begin(stream, K_ENC_ANALYSIS) : -1;
    hipLaunchKernelGGL(k_enc_analysis, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, stream, d_pcm, d_hdr, n_frames, SB,
                       Ts);
    if (prof) prof->end(stream, pp);
    pp = prof ? prof->begin(stream, K_ENC_MDCT) : -1;
    hipLaunchKernelGGL(k_enc_mdct, dim3((n_frames + 3) / 4), dim3(256), 0, stream, (const int32_t *)SB, Ts, d_hdr, n_frames,
                       d_mdct);
    if (prof) prof->end(stream, pp);
    return (int)hipGetLastError();
}

int launch_rate(hipStream_t stream, const int32_t *d_mdct, const mp3s_rate_frame *d_frames, int n_frames,
                const uint8_t *d_hide, int n_hide, const int32_t *d_cursor, const int32_t *d_state,
                const int32_t *d_list, int n_list, int16_t *d_ix, mp3s_gr_out *d_out, int32_t *d_en, Profiler *prof, int out_base, int compact,
                const RateVariantArgs *variants, hipEvent_t done)
{
    const int n_units = n_frames * 4;
    if (compact && !d_list) return (int)hipErrorInvalidValue;
    const int n = d_list ? n_list : n_units;
    RateVariants var = {nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr};
    if (variants && variants->n > 0) var = RateVariants{variants->d_unit, variants->d_cursor, variants->n, variants->d_ix, variants->d_out, variants->d_en, variants->d_tables};
    if (n + var.n <= 0) return done ? (int)hipEventRecord(done, stream) : 0;
    // a timed launch carries its event pair as the dispatch's own start and stop (the kernel's time stamps, no record packets around it);
    // `done` is then recorded behind it -- one packet on the timed steps instead of two
    hipEvent_t t_start = nullptr, t_stop = nullptr;
    const bool timed = prof && prof->attach(K_RATE_LOOP, &t_start, &t_stop);
    if (timed || done)
        hipExtLaunchKernelGGL(k_rate_loop, dim3((n + var.n + RL_WAVES - 1) / RL_WAVES), dim3(RL_WAVES * 64), 0, stream, timed ? t_start : nullptr, timed ? t_stop : done, 0,
                              d_mdct, d_frames, n_units, d_hide, n_hide, d_cursor, d_state, d_list, n, d_ix, d_out, d_en, out_base, compact, var);
    else
        hipLaunchKernelGGL(k_rate_loop, dim3((n + var.n + RL_WAVES - 1) / RL_WAVES), dim3(RL_WAVES * 64), 0, stream, d_mdct, d_frames,
                           n_units, d_hide, n_hide, d_cursor, d_state, d_list, n, d_ix, d_out, d_en, out_base, compact, var);
    if (timed && done && hipEventRecord(done, stream) != hipSuccess) return (int)hipErrorUnknown;
    return (int)hipGetLastError();
}

static size_t chain_scan_bytes(int n_frames) { return ((size_t)((n_frames + CH_THREADS - 1) / CH_THREADS) * sizeof(ChainEl) + 31) & ~(size_t)15; }
// rounds of re-runs behind the first check.  Measured with 3 (round 4): the jobs of tools/soak_select_long.py that one round leaves to
// the host -- raw PCM with silences inside a long message's reach, 20 - 160 units behind the plan whose cursors move with every
// re-run that takes another number of tables -- went from 2.7 % to 2.0 %, and the six more launches, empty on every other job,
// cost the resident step 0.03 ms.  One round it stays; what a re-run changes for the units that INHERIT from it is followed
// inside the round (k_rate_redo).
constexpr int kRedoRounds = 1;
size_t chain_agg_bytes(int n_frames) { return chain_scan_bytes(n_frames) + (size_t)2 * REDO_WORDS * 4; }   // (+ the two lists of units to run again, taking turns)

int launch_chain(hipStream_t stream, mp3s_gr_out *d_gr, const mp3s_rate_frame *d_frames, int n_frames, const mp3s_chain_seg *d_segs,
                 const int32_t *d_cursor, const int32_t *d_state, void *d_agg, int32_t *d_verdict, mp3s_chain_seg_out *d_seg_out,
                 Profiler *prof, const ChainRedoArgs *redo)
{
    if (n_frames <= 0) return 0;
    const int blocks = (n_frames + CH_THREADS - 1) / CH_THREADS;
    int32_t *d_redo = redo ? (int32_t *)((uint8_t *)d_agg + chain_scan_bytes(n_frames)) : nullptr;
    int32_t *cursor = const_cast<int32_t *>(d_cursor);   // (written only with `redo`)
    const int pp = prof ? prof->begin(stream, K_CHAIN) : -1;
    hipLaunchKernelGGL(k_chain_sum, dim3(blocks), dim3(CH_THREADS), 0, stream, (const mp3s_gr_out *)d_gr, d_frames, d_segs, n_frames,
                       (ChainEl *)d_agg, d_verdict, d_seg_out, d_redo, (const int32_t *)nullptr);
    hipLaunchKernelGGL(k_chain_apply, dim3(blocks), dim3(CH_THREADS), 0, stream, d_gr, d_frames, d_segs, n_frames,
                       (const ChainEl *)d_agg, cursor, d_state, d_verdict, d_seg_out, d_redo, (const int32_t *)nullptr);
    if (redo) {
        // The units the check listed run again (each following its chain of inheriting units), and the check is made again -- up to
        // kRedoRounds times, the lists taking turns: a re-run that takes another number of tables than the unit's first run moved
        // the cursor of every unit behind it as long as the message is live, so one round settles the units up to the first
        // such change, the next round the stretch behind it (tools/soak_select_long.py: the jobs one round left to the host had
        // 20 - 160 units still to redo, a few such changes apart).  A round whose list is empty returns at once, kernel by kernel.
        int32_t *lists[2] = {d_redo, d_redo + REDO_WORDS};
        for (int round = 0; round < kRedoRounds; round++) {
            const int32_t *cur = lists[round & 1];
            int32_t *nxt = round + 1 < kRedoRounds ? lists[(round + 1) & 1] : nullptr;
            hipLaunchKernelGGL(k_rate_redo, dim3(REDO_GROUPS), dim3(RL_WAVES * 64), 0, stream, redo->d_mdct, d_frames, n_frames * 4,
                               redo->d_hide, redo->n_hide, cur, redo->d_ix, d_gr, redo->d_en, (const int32_t *)d_cursor);
            hipLaunchKernelGGL(k_chain_sum, dim3(blocks), dim3(CH_THREADS), 0, stream, (const mp3s_gr_out *)d_gr, d_frames, d_segs, n_frames,
                               (ChainEl *)d_agg, d_verdict, d_seg_out, nxt, cur);
            hipLaunchKernelGGL(k_chain_apply, dim3(blocks), dim3(CH_THREADS), 0, stream, d_gr, d_frames, d_segs, n_frames,
                               (const ChainEl *)d_agg, cursor, d_state, d_verdict, d_seg_out, nxt, cur);
        }
    }
    if (prof) prof->end(stream, pp);
    return (int)hipGetLastError();
}

int launch_select(hipStream_t stream, const mp3s_chain_seg *d_segs, const mp3s_select_span *d_spans, int n_segs, int max_reach,
                  const uint8_t *d_hide, const RateVariantArgs &v, int16_t *d_ix, int32_t *d_en, mp3s_gr_out *d_out, int32_t *d_cursor,
                  void *d_pairs, Profiler *prof)
{
    if (n_segs <= 0 || max_reach <= 0 || v.n <= 0) return 0;
    if (max_reach > MP3S_SELECT_MAX_REACH || (int64_t)n_segs * max_reach > 0x3fffffff) return (int)hipErrorInvalidValue;
    const int pp = prof ? prof->begin(stream, K_CHAIN) : -1;
    hipLaunchKernelGGL(k_chain_select, dim3(n_segs), dim3(SEL_THREADS), select_lds_bytes(max_reach), stream, d_segs, d_spans, max_reach,
                       d_hide, (const uint8_t *)v.d_tables, (const mp3s_gr_out *)d_out, d_cursor, (int2 *)d_pairs);
    const int n_pairs = n_segs * max_reach;
    hipLaunchKernelGGL(k_scatter_entries, dim3((n_pairs + 3) / 4), dim3(256), 0, stream, (const int2 *)d_pairs, n_pairs,
                       (const int16_t *)v.d_ix, (const int32_t *)v.d_en, (const mp3s_gr_out *)v.d_out, d_ix, d_en, d_out);
    if (prof) prof->end(stream, pp);
    return (int)hipGetLastError();
}

int launch_gather_tables(hipStream_t stream, const mp3s_gr_out *d_outv, int n, uint8_t *d_tables)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_gather_tables, dim3((n + 255) / 256), dim3(256), 0, stream, d_outv, n, d_tables);
    return (int)hipGetLastError();
}

int launch_scatter(hipStream_t stream, const int32_t *d_pairs, int n_pairs, const int16_t *d_ixv, const int32_t *d_env,
                   const mp3s_gr_out *d_outv, int16_t *d_ix, int32_t *d_en, mp3s_gr_out *d_out)
{
    if (n_pairs <= 0) return 0;
    hipLaunchKernelGGL(k_scatter_entries, dim3((n_pairs + 3) / 4), dim3(256), 0, stream, (const int2 *)d_pairs, n_pairs, d_ixv, d_env,
                       d_outv, d_ix, d_en, d_out);
    return (int)hipGetLastError();
}

int launch_huffman(hipStream_t stream, const uint8_t *d_blob, const mp3s_frame_side *d_side, int n_frames, int nch, int max_bits,
                   int16_t *d_is, mp3s_granule_si *d_si, int32_t *d_status, int32_t *d_sync, Profiler *prof, bool per_frame, int lanes_opt)
{
    // (no fills in front of the kernel: it writes every sample pair, every side record and every status word itself)
    // LDS per workgroup = 27 KB of tables + W words per decoding lane, at most 64 KB.  Widest waves that still give
    // every SIMD about three waves to interleave; narrower ones otherwise (and for long granules, whose staging is big).
    const long units = (long)n_frames * 4;
    const int W = huf_words_for(max_bits);
    // (static 27 KB + dynamic: a workgroup of this chip may take more than the 64 KB a launch gets by default -- all 160 KB of a CU;
    // asked for once per device and kernel instance)
    int dev = 0;
    (void)hipGetDevice(&dev);
    const int dslot = dev >= 0 && dev < 64 ? dev : 0;
    static std::atomic<int> lds_of_device[64];
    int lds_max = lds_of_device[dslot].load(std::memory_order_relaxed);
    if (lds_max == 0) {
        if (hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || lds_max < 64 * 1024) lds_max = 64 * 1024;
        lds_of_device[dslot].store(lds_max, std::memory_order_relaxed);
    }
    const size_t kLdsMax = (size_t)lds_max;
    auto dyn = [&](int cols, int wv) { return ((size_t)(W + 17) * cols + 17 * wv) * 4; };
    auto fits = [&](int cols) { return dyn(cols, 8) + HUF_TAB_N * 2 + 256 <= kLdsMax; };
    int lanes = 16, waves = 4;
    // one wave per SIMD runs as fast as a wave can (1 024 SIMDs): 32 lanes per wave up to 8 192 frames, 64 beyond
    if (fits(128)) { lanes = 32; waves = 4; }
    if (fits(256) && units > 32 * 1024) { lanes = 64; waves = 4; }
    if (lanes_opt > 0) {          // MP3S_OPT_HUF_LANES
        const int a = lanes_opt;
        if (a == 64 && fits(256)) { lanes = 64; waves = 4; }
        if (a == 62 && fits(128)) { lanes = 64; waves = 2; }
        if (a == 32 && fits(128)) { lanes = 32; waves = 4; }
        if (a == 16) { lanes = 16; waves = 4; }
        if (a == 8) { lanes = 8; waves = 8; }
    }
    static const bool trace = getenv("MP3S_TRACE") != nullptr;
    if (trace) fprintf(stderr, "[mp3s] huffman launch: %ld units, W %d, %d waves x %d lanes, %zu bytes of dynamic LDS\n", units, W, waves, lanes, dyn(waves * lanes, waves));
    const int pp = prof ? prof->begin(stream, K_DEC_HUFFMAN) : -1;
#define MP3S_HUF_LAUNCH(WV, LN)                                                                                         \
    do {                                                                                                                \
        static std::atomic<unsigned long long> asked{0};                                                              \
        if (!(asked.load(std::memory_order_relaxed) & (1ull << dslot))) {                                               \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dec_huffman<WV, LN>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      (int)(kLdsMax - HUF_TAB_N * 2 - 256));                                            \
            asked.fetch_or(1ull << dslot, std::memory_order_relaxed);                                                   \
        }                                                                                                               \
        hipLaunchKernelGGL((k_dec_huffman<WV, LN>), dim3((unsigned)((units + WV * LN - 1) / (WV * LN))), dim3(WV * 64), \
                           dyn(WV * LN, WV), stream, d_blob, d_side, n_frames, nch, W, max_bits, d_is, d_si, d_status, per_frame ? 1 : 0, d_sync); \
    } while (0)
    if (lanes == 64 && waves == 4) MP3S_HUF_LAUNCH(4, 64);
    else if (lanes == 64) MP3S_HUF_LAUNCH(2, 64);
    else if (lanes == 32) MP3S_HUF_LAUNCH(4, 32);
    else if (lanes == 8) MP3S_HUF_LAUNCH(8, 8);
    else MP3S_HUF_LAUNCH(4, 16);
#undef MP3S_HUF_LAUNCH
    if (prof) prof->end(stream, pp);
    return (int)hipGetLastError();
}

static_assert(sizeof(ParseFrameRef) == sizeof(FrameRef) && sizeof(ParseStreamRef) == sizeof(StreamRef), "frame / stream references out of sync with mp3s_host.h");

int launch_parse(hipStream_t stream, const uint8_t *d_image, uint32_t image_base, const FrameRef *d_refs, const StreamRef *d_streams, int n_frames,
                 uint32_t md_base, mp3s_frame_side *d_side, mp3s_frame_hdr *d_hdr, uint8_t *d_blob, uint64_t *d_tsel, int32_t *d_status)
{
    if (n_frames <= 0) return 0;
    hipLaunchKernelGGL(k_dec_parse, dim3((n_frames + PARSE_WAVES - 1) / PARSE_WAVES), dim3(PARSE_WAVES * 64), 0, stream, d_image, image_base,
                       reinterpret_cast<const ParseFrameRef *>(d_refs), reinterpret_cast<const ParseStreamRef *>(d_streams), n_frames, md_base, d_side,
                       d_hdr, d_blob, d_tsel, d_status);
    return (int)hipGetLastError();
}

// host-decoded frames (the asynchronous pipe decodes the last frame of every stream on the host: E14) into their places:
// entry = [int32 frame, 12 bytes pad | int16 is[2304] | mp3s_granule_si si[4]] = 4912 bytes, one workgroup per entry
__global__ __launch_bounds__(256) void k_place_frames(const uint8_t *__restrict__ entries, int n_entries, int16_t *__restrict__ is,
                                                      mp3s_granule_si *__restrict__ si)
{
    const uint8_t *e = entries + (size_t)blockIdx.x * 4912;
    const int f = *reinterpret_cast<const int32_t *>(e);
    const uint32_t *src = reinterpret_cast<const uint32_t *>(e + 16);
    uint32_t *d_is = reinterpret_cast<uint32_t *>(is + (size_t)f * 2304), *d_si = reinterpret_cast<uint32_t *>(si + (size_t)f * 4);
    for (int i = threadIdx.x; i < 1152; i += 256) d_is[i] = src[i];
    for (int i = threadIdx.x; i < 72; i += 256) d_si[i] = src[1152 + i];
}
int launch_place_frames(hipStream_t stream, const uint8_t *d_entries, int n_entries, int16_t *d_is, mp3s_granule_si *d_si)
{
    if (n_entries <= 0) return 0;
    hipLaunchKernelGGL(k_place_frames, dim3(n_entries), dim3(256), 0, stream, d_entries, n_entries, d_is, d_si);
    return (int)hipGetLastError();
}

// One wave that keeps its hardware queue busy for `ticks` of the 100 MHz wall clock (bounded), and a kernel that does nothing:
// the pipe finds out with them which of its candidate streams run beside a context's compute stream (pipe_lanes.cpp, pick_lanes)
__global__ void k_spin(long long ticks, int *sink)
{
    const long long t0 = wall_clock64();
    int it = 0;
    while (wall_clock64() - t0 < ticks && it < (1 << 22)) it++;
    if (sink && it < 0) *sink = it;
}
__global__ void k_noop() {}
int launch_spin(hipStream_t stream, int microseconds)
{
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, stream, (long long)microseconds * 100, (int *)nullptr);
    return (int)hipGetLastError();
}
int launch_noop(hipStream_t stream)
{
    hipLaunchKernelGGL(k_noop, dim3(1), dim3(64), 0, stream);
    return (int)hipGetLastError();
}

// plain device copy, 16 bytes per lane and step: what HBM gives a streaming kernel on this box (BASELINE.md section 4 asks
// for the achievable figure beside the 8 TB/s of the data sheet)
__global__ __launch_bounds__(256) void k_copy16(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
}
int launch_copy(hipStream_t stream, const void *d_src, void *d_dst, size_t bytes)
{
    hipLaunchKernelGGL(k_copy16, dim3(256 * 16), dim3(256), 0, stream, (const uint4 *)d_src, (uint4 *)d_dst, bytes / 16);
    return (int)hipGetLastError();
}

int launch_pack(hipStream_t stream, const int16_t *d_ix, const mp3s_gr_out *d_gr, const int32_t *d_en, int n_frames, int sri,
                int bri, int whole_slots, const uint32_t *d_frame_off, const uint8_t *d_padding, uint8_t *d_mp3,
                int32_t *d_scfsi, int32_t *d_status, int32_t *d_sync, Profiler *prof, int f_begin, int f_end)
{
    const int pp = prof ? prof->begin(stream, K_ENC_PACK) : -1;
    if (f_end < 0 || f_end > n_frames) f_end = n_frames;
    const int count = f_end - f_begin;
    // persistent: 8 groups per CU; a short batch in fewer groups of at least four frames each (the tables are staged per group)
    const int groups = count >= 8192 ? 2048 : (count + 3) / 4;
    if (count > 0)
        hipLaunchKernelGGL(k_enc_pack, dim3(groups), dim3(256), 0, stream,
                           d_ix, d_gr, d_en, f_end, sri, bri, whole_slots,
                           d_frame_off, d_padding, d_mp3, d_scfsi, d_status, d_sync, f_begin);
    if (prof) prof->end(stream, pp);
    return (int)hipGetLastError();
}

}  // namespace mp3s
