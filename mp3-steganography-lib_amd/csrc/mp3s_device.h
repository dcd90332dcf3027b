// Host-visible launchers of the device translation unit (mp3s_device.hip).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/mp3s.h"

namespace mp3s {

// waves per channel in a synthesis tile (tile = TW*64 - 15 output slots).  2: 31 KB of LDS per workgroup, five groups =
// five waves per SIMD on a CU; the kernel waits for scalar-cache misses (the 16 KB matrix is streamed once per tile), so
// the fifth wave buys more (0.217 -> 0.207 ms) than the larger halo share (15/128 instead of 15/256) costs
#ifndef MP3S_DEC_SYNTH_TW
#define MP3S_DEC_SYNTH_TW 2
#endif
constexpr int DEC_SYNTH_TW = MP3S_DEC_SYNTH_TW;
constexpr int DEC_SYNTH_FAST_TW = 2;   // the fast int16 variant (k_dec_synth_fast)

// optional per-kernel HIP-event timing: when non-null, every kernel launch is bracketed by two events
// recorded on the launch stream; mp3s_profile_collect() turns the pairs into per-kernel totals.
enum KernelId { K_DEC_IMDCT = 0, K_DEC_SYNTH, K_ENC_ANALYSIS, K_ENC_MDCT, K_RATE_LOOP, K_DEC_HUFFMAN, K_ENC_PACK, K_CHAIN, K_COUNT };
struct Profiler {
    static constexpr int MAX_PAIRS = 8192;
    hipEvent_t ev[2 * MAX_PAIRS];
    int kid[MAX_PAIRS];
    int n_pairs = 0, n_created = 0;
    long dropped = 0;                         // launches that asked for an event pair and got none (MAX_PAIRS reached, event creation failed): mp3s_profile_collect refuses a table that misses them
    bool enabled = false;
    unsigned mask = ~0u;                      // bit k: kernel k gets an event pair
    double total_ms[K_COUNT] = {0};
    long count[K_COUNT] = {0};
    int begin(hipStream_t s, int k);          // returns pair index or -1
    void end(hipStream_t s, int pair);
    // the pair as the START and STOP event of one dispatch (hipExtLaunchKernelGGL): the kernel's own time stamps, no record packets around it
    bool attach(int k, hipEvent_t *start, hipEvent_t *stop);
};

int dev_upload_tables(hipStream_t stream);

// mp3s_debug_guard_margin: where the fused int16 decode leaves its fast values and guard widths (device arrays of `cap` doubles; a launch's
// sample i goes to element base + i)
struct GuardProbe { double *x, *eps; int64_t base, cap; };
// scratch: time-domain subband samples, float64 [nch][36 n][32], + what the fast int16 path keeps beside them
size_t dec_scratch_bytes(int n_frames, int nch);
int launch_decode(hipStream_t stream, const int16_t *d_is, const mp3s_granule_si *d_si, const mp3s_frame_hdr *d_hdr,
                  int n_frames, int nch, int n_halo, int out_format, void *d_pcm, void *d_scratch, Profiler *prof,
                  int sf_base = 0 /* hdr[].stream_first counts from this frame of the batch; d_is / d_si / d_hdr start there */,
                  double synth_eps_scale = 1.0 /* int16 output: guard of the fast kernels (imdct_run<true>, k_dec_synth_fast);
                                                  0 = the exact kernels */,
                  int32_t *d_sync = nullptr /* the context's self-clearing words (8): [2] counts the samples the guard sent
                                               through the exact order, [6..7] belong to the fix-up list; without them the
                                               exact kernels run */,
                  bool fast_imdct = true /* MP3S_OPT_FAST_IMDCT: the mirrored, fused IMDCT under the fast synthesis */,
                  bool float_fast = false /* MP3S_OPT_FLOAT_FAST: float32 output through the fast sums, unguarded (within 1e-5, not bit-identical) */,
                  bool fused = true /* MP3S_OPT_FUSED_DECODE: the fast paths as one kernel, S in LDS (k_decode_fused.hpp) */,
                  hipEvent_t done = nullptr /* recorded behind the transforms: as the last dispatch's own completion signal where the path has one
                                               (the fused int16 path: no record packet in the queue), by a record otherwise */,
                  const GuardProbe *probe = nullptr /* the fused int16 path also writes its fast values and guard widths there (a probe) */);

// scratch: subband samples int32 [2][32][36 n]
size_t enc_scratch_bytes(int n_frames);
int launch_encode(hipStream_t stream, const int16_t *d_pcm, const mp3s_frame_hdr *d_hdr, int n_frames, int32_t *d_mdct,
                  void *d_scratch, Profiler *prof, bool fused = false /* MP3S_OPT_FUSED_ENCODE: one kernel, no scratch (d_scratch may be null) */);

// entries behind the launch's own units: unit d_unit[e] once more with cursor d_cursor[e] and no inherited state, results
// to element e of d_ix / d_out / d_en (the message variants the device chooses from, launch_select)
struct RateVariantArgs {
    const int32_t *d_unit, *d_cursor;
    int n;
    int16_t *d_ix; mp3s_gr_out *d_out; int32_t *d_en;
    uint8_t *d_tables;   // n bytes: the entries' table counts (kept behind the GrInfo records: variant_out_bytes)
};
// the GrInfo array of n variant entries is followed by their table counts, one byte each
inline size_t variant_out_bytes(int n) { return (size_t)n * sizeof(mp3s_gr_out) + (((size_t)n + 15) & ~(size_t)15); }
// state: int32 [units][4] = address1, address2, address3, quantizerStepSize inherited from the previous frame
int launch_rate(hipStream_t stream, const int32_t *d_mdct, const mp3s_rate_frame *d_frames, int n_frames,
                const uint8_t *d_hide, int n_hide, const int32_t *d_cursor, const int32_t *d_state,
                const int32_t *d_list, int n_list, int16_t *d_ix, mp3s_gr_out *d_out, int32_t *d_en, Profiler *prof,
                int out_base = 0 /* unit whose results land on element 0 of d_ix / d_out / d_en */,
                int compact = 0 /* 1: d_cursor / d_state / d_out are indexed by the position in d_list; 2: d_ix / d_en too */,
                const RateVariantArgs *variants = nullptr,
                hipEvent_t done = nullptr /* the dispatch's own completion signal: no record packet behind it (MP3S_OPT_RATE_SIGNALS) */);

// the serial chains of the rate loop (k_chain.hpp): two small launches; d_agg: chain_agg_bytes(n_frames) of scratch.
// With `redo` (the rate loop's other operands) the units the check finds wrong -- they read inherited addresses that were
// not what the chain holds, or ran on another cursor -- are listed on the device, run again on what the check found
// (k_rate_redo, results in place, d_cursor updated for them: it is written then) and everything is checked once more:
// five launches, the verdict is the second check's.
struct ChainRedoArgs { const int32_t *d_mdct; const uint8_t *d_hide; int n_hide; int16_t *d_ix; int32_t *d_en; };
size_t chain_agg_bytes(int n_frames);
int launch_chain(hipStream_t stream, mp3s_gr_out *d_gr, const mp3s_rate_frame *d_frames, int n_frames, const mp3s_chain_seg *d_segs,
                 const int32_t *d_cursor, const int32_t *d_state, void *d_agg, int32_t *d_verdict, mp3s_chain_seg_out *d_seg_out,
                 Profiler *prof, const ChainRedoArgs *redo = nullptr);

// the message cursor decided on the device (k_chain_select): one workgroup per stream, after a launch_rate with variants
int launch_select(hipStream_t stream, const mp3s_chain_seg *d_segs, const mp3s_select_span *d_spans, int n_segs, int max_reach,
                  const uint8_t *d_hide, const RateVariantArgs &v, int16_t *d_ix, int32_t *d_en, mp3s_gr_out *d_out, int32_t *d_cursor,
                  void *d_pairs /* n_segs * max_reach * 8 bytes of scratch */, Profiler *prof);

// side-info parse + main-data gather on the device (k_parse.hpp) from the file image and the host's frame walk (FrameRef /
// StreamRef of mp3s_host.h, 16 / 40 bytes): side records, frame headers and the blob as the host scan would have written
// them.  image_base / md_base: what d_image[0] and d_blob[0] are in the offsets the references hold (a chunk of a long
// file brings its own piece of both).  d_tsel optional (decode jobs: the stego bits are put together from it on the host);
// d_status: one word the caller has zeroed, PARSE_* bits are OR-ed in.
using FrameRef = mp3s_frame_ref;
using StreamRef = mp3s_stream_ref;
constexpr int kParseInherits = 1, kParseMismatch = 2;
int launch_parse(hipStream_t stream, const uint8_t *d_image, uint32_t image_base, const FrameRef *d_refs, const StreamRef *d_streams, int n_frames,
                 uint32_t md_base, mp3s_frame_side *d_side, mp3s_frame_hdr *d_hdr, uint8_t *d_blob, uint64_t *d_tsel, int32_t *d_status);

constexpr size_t kPlaceEntry = 4912;   // [int32 frame, pad to 16 | int16 is[2304] | mp3s_granule_si si[4]]
int launch_place_frames(hipStream_t stream, const uint8_t *d_entries, int n_entries, int16_t *d_is, mp3s_granule_si *d_si);
int launch_copy(hipStream_t stream, const void *d_src, void *d_dst, size_t bytes);
// one wave that occupies its hardware queue for about `microseconds`; a kernel that does nothing
int launch_spin(hipStream_t stream, int microseconds);
int launch_noop(hipStream_t stream);

// bit-level stages on the device
int launch_huffman(hipStream_t stream, const uint8_t *d_blob, const mp3s_frame_side *d_side, int n_frames, int nch, int max_bits,
                   int16_t *d_is, mp3s_granule_si *d_si, int32_t *d_status, int32_t *d_sync /* one zeroed 64-bit word owned by the context (k_sync.hpp) */,
                   Profiler *prof,
                   bool per_frame = false /* d_status holds 1 + n_frames words: [0] = all, [1 + f] = frame f */,
                   int lanes_opt = 0 /* MP3S_OPT_HUF_LANES: 0 = from the launch's size */);
// the tables every entry of a compact == 2 launch took, one byte each
int launch_gather_tables(hipStream_t stream, const mp3s_gr_out *d_outv, int n, uint8_t *d_tables);
// pairs: int32 [n][2] = (entry, unit): entry's ix / energies / GrInfo (compact == 2 arrays) -> the unit's place
int launch_scatter(hipStream_t stream, const int32_t *d_pairs, int n_pairs, const int16_t *d_ixv, const int32_t *d_env,
                   const mp3s_gr_out *d_outv, int16_t *d_ix, int32_t *d_en, mp3s_gr_out *d_out);
int launch_pack(hipStream_t stream, const int16_t *d_ix, const mp3s_gr_out *d_gr, const int32_t *d_en, int n_frames, int sri,
                int bri, int whole_slots, const uint32_t *d_frame_off, const uint8_t *d_padding, uint8_t *d_mp3,
                int32_t *d_scfsi, int32_t *d_status, int32_t *d_sync /* one zeroed 64-bit word owned by the context (k_sync.hpp) */, Profiler *prof,
                int f_begin = 0, int f_end = -1 /* frames [f_begin, f_end) only (f_end < 0: to the end); with d_sync == null only:
                                                   the arrival counter hands out ONE launch's bits */);

}  // namespace mp3s
