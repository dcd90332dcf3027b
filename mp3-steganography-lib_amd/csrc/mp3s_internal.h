// Shared by the translation units behind include/mp3s.h (mp3s_api.cpp, mp3s_decode_pipeline.cpp,
// mp3s_encode_pipeline.cpp): the context, result owners, page-locked blocks and the pipeline helpers.  Nothing here is
// part of the C-ABI.
#pragma once
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mp3s.h"
#include "mp3s_device.h"
#include "mp3s_host.h"

namespace mp3s {
// records the text mp3s_last_error() returns on this thread; returns `code`
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
}  // namespace mp3s

using namespace mp3s;

#define HIPCHK(call)                                                                                 \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) return fail(MP3S_E_HIP, "%s: %s", #call, hipGetErrorString(e_));       \
    } while (0)

struct mp3s_pipe;
struct mp3s_ctx {
    int device = 0;
    int64_t opt[MP3S_OPT_COUNT] = {0};   // MP3S_OPT_*: defaults from the environment at creation, then mp3s_ctx_set_option
    mp3s_run_stats run_stats = {0, 0, 0, 0, 0};
    mp3s_pipe *own_pipe = nullptr;       // the overlapped stages the one-file calls run their chunks through (made on first use)
    std::vector<mp3s_pipe *> parked_pipes;   // own pipes that became too small (a file with larger frames came): quiet, kept until the context goes (run_file.cpp)
    int sink_fd = -1; size_t sink_done = 0, sink_base = 0; bool sink_early = false;   // (sink_early: the file was empty when the call began -- only then do chunks go to it before the call has succeeded)
   // mp3s_*_fd: the file the result goes to, how many of its bytes (behind sink_base: the WAV header's place) run_file has written already
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_order = nullptr;
    hipEvent_t ev_sel = nullptr; bool sel_pending = false;   // a selection on a tail stream has read the variant buffers (enc_issue)
    hipEvent_t rate_done = nullptr;   // set by enc_issue around its call of a rate entry point: the rate loop's dispatch signals it (no record packet)
    int32_t *d_sync = nullptr;        // 64-bit word {finished workgroups | error bits} of the pack kernel in flight (k_sync.hpp): self-clearing;
                                      // [2] counts the samples the fast synthesis computed again in the exact order
    double synth_eps_scale = 1.0;     // int16 decode: scale of the fast synthesis guard (0 = always the exact kernel)
    GuardProbe guard_probe = {nullptr, nullptr, 0, 0};   // mp3s_debug_guard_margin
    void *scratch = nullptr; size_t scratch_bytes = 0;
    Profiler prof;
    // device buffers of the stream pipelines, kept between calls (hipMalloc/hipFree cost more than a small file's work)
    static constexpr int kPoolSlots = 32;
    void *pool[kPoolSlots] = {nullptr};
    size_t pool_bytes[kPoolSlots] = {0};
    void *grab(int slot, size_t bytes)
    {
        if (bytes < 16) bytes = 16;
        if (pool_bytes[slot] >= bytes) return pool[slot];
        if (pool[slot]) { hipStreamSynchronize(stream); hipFree(pool[slot]); pool[slot] = nullptr; pool_bytes[slot] = 0; }
        const size_t want = bytes + bytes / 4;   // head room: similar-sized files reuse the buffer
        if (hipMalloc(&pool[slot], want) != hipSuccess) { pool[slot] = nullptr; return nullptr; }
        pool_bytes[slot] = want;
        return pool[slot];
    }
    // host-side work arrays of the encoder, kept between calls: beyond a few MB a fresh vector means fresh pages from
    // the kernel on every call (page faults cost more than the work done in them)
    std::vector<int32_t> h_cursor, h_state, h_want, h_state_want;
    std::vector<uint8_t> h_in;
    // ... and of the decoder: the concatenated batch
    std::vector<mp3s_frame_hdr> h_hdr;
    std::vector<mp3s_frame_side> h_side;
    std::vector<uint8_t> h_blob;
    // ... and the scan result of the last single-file call, lent to the next one for its capacity
    ScannedStream spare_scan;
    // ... and the frame table / table counts of the file run_file is working on
    std::vector<FrameRef> h_refs;
    std::vector<uint8_t> h_tables;
    int ensure_scratch(size_t bytes)
    {
        if (bytes <= scratch_bytes) return 0;
        if (scratch) { hipFree(scratch); scratch = nullptr; scratch_bytes = 0; }
        hipError_t e = hipMalloc(&scratch, bytes);
        if (e != hipSuccess) return fail(MP3S_E_NOMEM, "hipMalloc(%zu) for scratch: %s", bytes, hipGetErrorString(e));
        scratch_bytes = bytes;
        return 0;
    }
    // the encode transforms' scratch (subband samples between analysis and MDCT) is a buffer of its own: in the pipe the decode
    // transforms of the next job run on another stream beside them
    void *scratch_enc = nullptr; size_t scratch_enc_bytes = 0;
    int ensure_scratch_enc(size_t bytes)
    {
        if (bytes <= scratch_enc_bytes) return 0;
        if (scratch_enc) { hipFree(scratch_enc); scratch_enc = nullptr; scratch_enc_bytes = 0; }
        hipError_t e = hipMalloc(&scratch_enc, bytes);
        if (e != hipSuccess) return fail(MP3S_E_NOMEM, "hipMalloc(%zu) for scratch: %s", bytes, hipGetErrorString(e));
        scratch_enc_bytes = bytes;
        return 0;
    }
};

// Page-locked host memory for large results (decoded PCM): the device writes it at PCIe speed, no bounce buffer, no
// page faults.  Pinning costs more than the copy it saves, so blocks are kept and reused: process-wide, because a
// result may outlive the context that produced it.  (Blocks still cached at exit are left to the OS.)
// mp3s_debug_guard_margin's probe is an instantiation of the stream kernel: an int16 decode that would take another route while the probe is set
// (MP3S_OPT_FUSED_DECODE or MP3S_OPT_FAST_IMDCT off, the guard switched off) is refused -- it would leave the probe's arrays as they were.
// (Float formats never fill them and pass.)
inline int guard_probe_usable(const mp3s_ctx *c, int out_format)
{
    if (!c->guard_probe.x || out_format != MP3S_PCM_I16) return MP3S_OK;
    if (c->opt[MP3S_OPT_FUSED_DECODE] && c->opt[MP3S_OPT_FAST_IMDCT] && c->synth_eps_scale > 0 && c->d_sync) return MP3S_OK;
    return fail(MP3S_E_ARG, "the guard probe is set (mp3s_debug_guard_margin) and this int16 decode would not run the stream kernel that fills it "
                            "(MP3S_OPT_FUSED_DECODE / MP3S_OPT_FAST_IMDCT off, or the guard's scale is 0)");
}

int local_world_size();   // mp3s_hostinfo.cpp: LOCAL_WORLD_SIZE of the launcher (1 without one)

class PinnedBlock {
public:
    PinnedBlock() = default;
    PinnedBlock(const PinnedBlock &) = delete;
    PinnedBlock &operator=(const PinnedBlock &) = delete;
    ~PinnedBlock() { release(); }
    bool reserve(size_t bytes)
    {
        if (bytes <= cap_) return true;
        release();
        if (bytes > kMaxPinned) {   // hours of audio in one call: pinning gigabytes costs seconds, ordinary memory then
            p_ = (uint8_t *)std::malloc(bytes);
            if (!p_) return false;
            cap_ = bytes; pinned_ = false;
            return true;
        }
        pinned_ = true;
        {
            std::lock_guard<std::mutex> g(mu());
            auto &fl = free_list();
            size_t best = fl.size();
            // best fit, but a small request does not take a large block away from the next large request (which would
            // then have to pin fresh pages: about a millisecond per 4 MB)
            const size_t too_big = std::max(4 * bytes, bytes + ((size_t)1 << 20));
            for (size_t i = 0; i < fl.size(); i++)
                if (fl[i].second >= bytes && fl[i].second <= too_big && (best == fl.size() || fl[i].second < fl[best].second)) best = i;
            if (best < fl.size()) { p_ = fl[best].first; cap_ = fl[best].second; fl.erase(fl.begin() + best); return true; }
        }
        const size_t want = bytes + bytes / 8 + (1 << 16);
        void *q = nullptr;
        if (hipHostMalloc(&q, want, hipHostMallocDefault) != hipSuccess) return false;
        p_ = (uint8_t *)q; cap_ = want;
        return true;
    }
    uint8_t *data() const { return p_; }
    // what the process keeps pooled right now, and the cap (mp3s_ctx_host_share)
    static void pool_state(size_t *held, size_t *cap)
    {
        std::lock_guard<std::mutex> g(mu());
        size_t h = 0;
        for (auto &e : free_list()) h += e.second;
        *held = h; *cap = pool_cap();
    }

private:
    void release()
    {
        if (!p_) return;
        if (!pinned_) { std::free(p_); p_ = nullptr; cap_ = 0; return; }
        std::lock_guard<std::mutex> g(mu());
        auto &fl = free_list();
        size_t held = 0;
        for (auto &e : fl) held += e.second;
        if (fl.size() < 48 && held + cap_ <= pool_cap()) fl.emplace_back(p_, cap_);
        else hipHostFree(p_);
        p_ = nullptr; cap_ = 0;
    }
    static std::mutex &mu() { static std::mutex *m = new std::mutex(); return *m; }
    static std::vector<std::pair<uint8_t *, size_t>> &free_list()
    {
        static auto *v = new std::vector<std::pair<uint8_t *, size_t>>();
        return *v;
    }
    static constexpr size_t kMaxPinned = (size_t)4 << 30;   // (a 100 000-frame file decodes to 460 MB of PCM: two such results and the MP3 results beside them stay pooled)
    // what stays pooled between calls: the ranks of one host share its page-locked memory (memlock / cgroup limits), so eight ranks
    // keep 1 GB each where a lone process keeps 4; a block that does not fit goes back to the system when it is released
    static size_t pool_cap()
    {
        static const size_t cap = std::max<size_t>((size_t)1 << 30, kMaxPinned / (size_t)local_world_size());
        return cap;
    }
    uint8_t *p_ = nullptr;
    size_t cap_ = 0;
    bool pinned_ = true;
};

struct mp3s_multi {     // owner payload of mp3s_decode_streams
    std::vector<std::pair<const uint8_t *, size_t>> files;   // borrowed for the duration of the call
    std::vector<ParsedStream> parsed;
    std::vector<ScannedStream> scanned;
    PinnedBlock arena[3];                 // PCM of all mono / all stereo streams, index = channel count
    size_t head_room = 0;                 // bytes kept free in front of the PCM (mp3s_decode_file puts the WAV header there)
    std::vector<const uint8_t *> pcm;     // per stream, into its arena
    // blocks of a stream: only frames [first, first + count) of stream i are kept after parsing (absent: all of them);
    // keep_dup = the frame the reference repeats after a bad header (D12) still belongs to a window that ends the stream
    struct Window { long first, count; bool keep_dup; };
    std::vector<Window> window;
    std::vector<std::vector<uint8_t>> all_bits;   // ... and the stego bits of the whole stream
    // an index of the (single) stream: its window is scanned on its own, from the resume point in front of it, instead of
    // the whole file (then all_bits holds the bits of the window's frames only)
    const StreamIndex *index = nullptr;
};

struct mp3s_buf {
    std::shared_ptr<mp3s_multi> multi;
    ParsedStream parsed;
    ScannedStream scanned;
    std::vector<uint8_t> bytes;      // generic payload (pcm / mp3)
    std::vector<uint8_t> bits;
    std::vector<int32_t> scfsi;
    std::vector<std::unique_ptr<mp3s_buf>> parts;   // results of the batches of a multi-file call
    // encoder results: MP3 bytes and GrInfo records land in page-locked blocks and are handed out from there
    PinnedBlock big[3];              // MP3 bytes, GrInfo records, the small results (verdict, per-stream chain ends)
    uint8_t *mp3 = nullptr;
    mp3s_gr_out *gr_out = nullptr;
};

// MP3S_TRACE=1: phase timings of the file pipelines on stderr
inline bool trace_on() { static const bool on = getenv("MP3S_TRACE") != nullptr; return on; }
inline double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// f(i) for i in [0, n) on a few host threads: the front ends of the files of a batch are independent.  Small batches
// (by bytes) stay on the calling thread -- starting a thread costs about what scanning 100 KB does.
template <class F>
void parallel_files(int n, size_t total_bytes, F f)
{
    const unsigned hw = std::thread::hardware_concurrency();
    const int workers = (int)std::min<size_t>({(size_t)n, (size_t)std::min(hw ? hw : 1u, 16u), total_bytes / (256u << 10) + 1});
    if (workers <= 1) {
        for (int i = 0; i < n; i++) f(i);
        return;
    }
    std::atomic<int> next{0};
    auto run = [&]() { for (int i; (i = next.fetch_add(1)) < n;) f(i); };
    std::vector<std::thread> pool;
    for (int w = 1; w < workers; w++) pool.emplace_back(run);
    run();
    for (auto &t : pool) t.join();
}

// ---------------------------------------------------------------- decode pipeline (mp3s_decode_pipeline.cpp)
constexpr int kDecodeChunk = 16384;   // frames per decode launch group (scratch ~0.6 GB); chunks overlap by a 1-frame halo
inline size_t pcm_elem(int fmt) { return fmt == MP3S_PCM_I16 ? 2 : (fmt == MP3S_PCM_F32 ? 4 : 8); }
int max_part2_3(const mp3s_frame_side *side, long n);
// keep frames [first, first + count) of a parsed stream (its main data, side records, samples)
void cut_window(ParsedStream &p, ScannedStream &sc, long first, long count);
// host front end of stream i of m (scan; full host parse where the device cannot decode), cut to the stream's window
int front_end(mp3s_multi &m, int i);
int decode_transform_chunk(mp3s_ctx *c, const int16_t *d_is, const mp3s_granule_si *d_si, const mp3s_frame_hdr *d_hdr, long first, int cnt,
                           int nch, int halo, int out_format, void *d_pcm, hipStream_t stream = nullptr /* null: the context's */,
                           hipEvent_t done = nullptr /* recorded behind the chunk's transforms (launch_decode) */);
// decode the streams `idx` of m (one channel count) as one batch; d_keep: int16 PCM stays on the device there
int decode_group(mp3s_ctx *c, mp3s_multi &m, const std::vector<int> &idx, int nch, int out_format, void *d_keep = nullptr);

// ---------------------------------------------------------------- encode pipeline (mp3s_encode_pipeline.cpp)
constexpr int kLongMessageBits = 1024;    // above: the first pass does not guess cursors at all
constexpr int32_t kNoCursor = MP3S_NO_CURSOR; // "behind every message": such a unit hides nothing
constexpr int kPatternBytes = 32;         // the eight 3-bit patterns, 4 bytes apart, in front of the messages
constexpr int kVariantEntries = 65536;    // (unit, pattern) entries per variant launch
constexpr size_t kFewUnits = 8;           // that few wrong cursors after the first pass: re-run them directly

struct EncSeg {             // one stream of an encode batch: frames back to back in the batch's PCM
    int n_frames = 0;
    const uint8_t *hide = nullptr;   // 0/1 bytes
    int n_hide = 0;
    // a block of a longer stream (mp3s_encode_block; only as the single stream of a batch)
    int lead = 0;                    // frames of PCM in front of the block: transformed for their state, then dropped
    int64_t first_frame = 0;         // index of the block's first frame in its stream (padding recurrence)
    bool last = true;                // the stream ends with this block (the reference drops the cached tail there: E14)
    const mp3s_carry *carry_in = nullptr;
    // The first pass of the rate loop runs every unit on a GUESSED message cursor: the tables the units in front of it will
    // take.  Without better knowledge that is three per unit.  A stream that is being re-encoded brings better knowledge:
    // the non-zero table indices of the SAME audio in the stream it was decoded from (side info of the input, unit order
    // frame / channel / granule; silence has none).  n_frames * 4 entries, or null.
    const uint8_t *tables_guess = nullptr;
    int n_guess = -1;                // units tables_guess covers (-1: all n_frames * 4); behind them the guess is 3 per unit
    int any_silent = -1;             // 1 / 0: some / no unit of the stream is without tables; -1: look into tables_guess
    // filled by encode_batch
    int first = 0, hide_base = 0;
    int reach = 0, first_entry = 0;  // the message cursor is decided on the device over the stream's first `reach` units (0: guessed)
    int64_t hide_offset = 0;         // message bits consumed (from the start of the stream)
    size_t mp3_off = 0, mp3_len = 0; // the stream's bytes inside the batch's output
    mp3s_carry carry_out = {};
    bool carry_used = false;         // the block's bytes depend on carry_in
};
// The host-made inputs of an encode batch travel as ONE block (one copy):
//   [frame headers (n_all) | rate frames (n) | assumed cursors (units) | message bits | chain segments | frame offsets (n+1) | padding (n)
//    | selection spans (n_segs) | variant entries: units (n_entries), cursors (n_entries)]
struct EncLayout {
    int n = 0, n_all = 0, lead = 0, units = 0, n_hide = 0, n_segs = 0;
    int n_entries = 0, max_reach = 0;   // message variants run inside the first rate-loop launch (mp3s_rate_select_dev)
    bool redo = true;              // the chain check is issued with the device's re-runs behind it (launch_chain's `redo`)
    int sri = 0, bri = 0, whole = 0, samplerate = 0, kbps = 0;
    size_t o_rf = 0, o_cur = 0, o_hide = 0, o_segs = 0, o_off = 0, o_pad = 0, o_spans = 0, o_ent = 0, bytes = 0;   // (headers at offset 0)
    size_t mp3_bytes = 0;          // all frames of the batch
    int64_t bytes_before = 0;      // size of the frames in front of a block (E14: the tail cut depends on it)
};
// small results of a batch, in one block: verdict[2] (mp3s_chain_resolve_dev), pack status, Huffman status, parse status, then seg_out[n_segs]
constexpr size_t kSmallHead = 32;          // (word 4: status of the device-side parse, MP3S_PS_*; 5..7 spare)
inline size_t small_bytes(int n_segs) { return kSmallHead + (size_t)n_segs * sizeof(mp3s_chain_seg_out); }
// checks the streams and fills first / hide_base of each
int enc_layout(std::vector<EncSeg> &segs, int samplerate, int bitrate_kbps, EncLayout &L, bool select = true /* MP3S_OPT_SELECT */);
// writes the block (L.bytes at dst); *mp3_off_len: per stream {offset, length} of its bytes in the batch's output
int enc_fill(std::vector<EncSeg> &segs, EncLayout &L, uint8_t *dst);
struct EncDev {
    const int16_t *d_pcm = nullptr;      // [n_all][1152][2]
    const uint8_t *d_in = nullptr;       // the block above
    int32_t *d_mdct_all = nullptr;       // [n_all][2][2][576]
    int16_t *d_ix = nullptr; mp3s_gr_out *d_out = nullptr; int32_t *d_en = nullptr;
    void *d_agg = nullptr;               // chain_agg_bytes(n)
    uint8_t *d_mp3 = nullptr; int32_t *d_sc = nullptr;
    int32_t *d_small = nullptr;          // small_bytes(n_segs)
    bool direct_status = false;          // d_small's status words were zeroed with the job's inputs: the kernels OR into them directly
    // the bit packer in two launches, frames [0, pack_split) and the rest, pack_half recorded between them (the last chunk of a
    // one-file call: its first half comes down while the second is packed); 0: one launch
    int pack_split = 0; hipEvent_t pack_half = nullptr;
    // results of the variant entries (L.n_entries of them; read by the selection right behind the rate loop)
    int16_t *d_ixv = nullptr; mp3s_gr_out *d_outv = nullptr; int32_t *d_env = nullptr;
};
// the entries of a stream's first `reach` units (variant-major, the two "bits left" rows from MP3S_SELECT_TAIL_FIRST on: mp3s.h)
void select_entries(int first_unit, int reach, int first_entry, int hide_end, int64_t bits_left, int32_t *ent_unit, int32_t *ent_cursor);
// device buffers for L.n_entries variant entries from the context's pool (slots of the host's variants, free at that point)
bool enc_variant_buffers(mp3s_ctx *c, const EncLayout &L, EncDev &d);
// transforms -> rate loop on the guessed cursors -> chain check -> bit packing, all on c->stream, nothing waited for.
// The packed bytes are final iff verdict[0] == 0 and verdict[1] == 0 (d_small[0], d_small[1]).
int enc_issue(mp3s_ctx *c, const EncLayout &L, const EncDev &d, hipStream_t tail = nullptr /* chain check + packing on this stream, ordered behind the rate loop through tail_from */,
              hipEvent_t tail_from = nullptr, hipEvent_t rate_after = nullptr /* the rate loop waits for this event (the previous job's tail) */,
              hipEvent_t pcm_read = nullptr /* recorded behind the encode transforms: the PCM they read may be overwritten */);
// verdict != 0: the host resolves the chains on the first pass's device buffers (walk, message variants, exact re-runs,
// packing again); `in` = the host copy of the block.  Pool slots of its own: the entries of the exact re-runs, the variants.
constexpr int kSlotRedo = 11, kSlotVariants = 19;
int enc_resolve(mp3s_ctx *c, const EncLayout &L, std::vector<EncSeg> &segs, const uint8_t *in, const EncDev &d, mp3s_buf *b, bool have_gr,
                int *passes_out);
int encode_batch(mp3s_ctx *c, const int16_t *pcm, const int16_t *pcm_dev, std::vector<EncSeg> &segs, int samplerate, int bitrate_kbps,
                 mp3s_buf *b, int *passes_out, bool want_gr = true);
// non-zero table indices per unit of a scanned stream, in the encoder's unit order (frame, channel, granule): see
// EncSeg::tables_guess; `extra` more frames (the repeated last frame of a stream that ends in a bad header) repeat the last
// mp3s_select_plan with a lower limit for each stream's reach (nullptr: none)
int select_plan(const mp3s_chain_seg *segs, int n_segs, mp3s_select_span *spans, int32_t *ent_unit, int32_t *ent_cursor, int cap,
                const int32_t *min_reach);
void tables_guess_of(const mp3s_frame_side *side, long n_frames, int extra, std::vector<uint8_t> &out);

// ---------------------------------------------------------------- one file as chunks through the overlapped stages (run_file.cpp)
constexpr int kRunFallback = 1;          // run_file: not for this path -- the caller takes the synchronous one (same bytes)
constexpr int kRunHide = 0, kRunClear = 1, kRunDecode = 2;
struct RunResult {
    int64_t n_frames = 0, n_rows = 0;
    int nch = 0, sampling_rate = 0, bit_rate = 0, kbps = 0;
    const uint8_t *mp3 = nullptr; size_t mp3_len = 0;        // hide / clear
    int64_t hide_offset = 0; int too_long = 0;
    const uint8_t *pcm = nullptr;                            // decode: [n_rows][nch] in the format asked for, 64 bytes into its block
    const uint8_t *bits = nullptr; size_t n_bits = 0;
};
// mode: kRunHide (utf8 / n_msg = the message) / kRunClear / kRunDecode (out_format).  MP3S_OK, kRunFallback, or an error.
int run_file(mp3s_ctx *c, const uint8_t *mp3, size_t len, int mode, const uint8_t *utf8, size_t n_msg, int out_format, mp3s_buf **owner, RunResult *out);
void destroy_own_pipe(mp3s_ctx *c);
void pipe_quiesce(mp3s_pipe *P);     // pipe_jobs.cpp: threads ended, streams drained, nothing freed
void own_pipe_lanes(const mp3s_ctx *c, mp3s_run_stats *out);

// ---------------------------------------------------------------- what this process may use of the host (mp3s_hostinfo.cpp)
// CPUs the process may run on: sched_getaffinity, cut down to the cgroup's CPU quota
int host_cpus_allowed();
// ranks sharing this host: LOCAL_WORLD_SIZE of the launcher, 1 without one
int local_world_size();
// host threads a rank should scan / walk with: MP3S_OPT_SCAN_THREADS, or its share of the allowed CPUs (one is kept for the
// thread that issues and collects), between 1 and 3
int default_scan_threads(const mp3s_ctx *c);
// the CPUs of the GPU's NUMA node that this process may run on (sysfs numa_node of the PCI device); empty: unknown, no binding
std::vector<int> gpu_node_cpus(int device);
