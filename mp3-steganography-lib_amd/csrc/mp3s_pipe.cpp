// C-ABI of the library (include/mp3s.h), part 4: the overlapped stages -- the asynchronous host-fed pipeline (mp3s_pipe_*)
// and, on the same machinery, ONE file as a sequence of chunks (run_file: what mp3s_hide_message, mp3s_clear_file,
// mp3s_decode_file and mp3s_decode_stream are made of since round 3).
//
// The reference runs one file at a time through two serial frame loops (decoder/MP3_Parser.py:68-80,
// encoder/MP3_Encoder.py:607-609, glued by steganography.py:153-159).  Here a job (one file, a list of files that form one
// device batch, or a chunk of a long file) passes through stages that overlap with those of the jobs around it:
//
//   walk      a host thread steps from frame header to frame header (FrameWalker: frame sizes, reservoir pointers -> where
//             each frame's main data goes; 16 bytes per frame) -- or, for a stream the walk does not take (false syncs,
//             inherited header fields ...), the byte-level scan of round 2 (parse_stream_sink straight into the staging)
//   upload    hipMemcpyAsync on the copy-up stream: the file bytes as they are, from the caller's memory, + the frame table
//   front end the Huffman stream: side-info parse + main-data gather (k_dec_parse) -> Huffman decode (k_dec_huffman)
//   compute   the context's stream: decode transforms -> encode transforms -> rate loop -> chain check -> bit packing; the
//             scratch between the kernels is shared by all slots (one stream = one job at a time)
//   download  hipMemcpyAsync on the copy-down stream: MP3 bytes + the small verdict block (or PCM) -> page-locked results
//
// Events order the streams; the host waits for nothing until a result is collected.  A job the device cannot take alone
// (mono re-encode, a repeated last frame, unsupported rates, staging too small), or whose verdict says the Huffman data is
// damaged, is redone by the synchronous path -- same bytes, by construction of that path; the fast path is an
// optimisation, never a different answer.
#include <sched.h>
#include <time.h>

#include <condition_variable>
#include <deque>

#include "mp3s_internal.h"

namespace {

constexpr int kMaxFastFiles = 1024;
constexpr size_t kDirectUpload = (size_t)256 << 10;   // a file at least this long goes up from the caller's memory in a copy of its own
constexpr uint32_t kImageLead = 1024;                 // bytes in front of a chunk's first frame its reservoir pointers can name (511 + 8 x 38)
constexpr int kRunDepth = 3;                          // chunks of one file in flight

struct Slot {
    uint8_t *h_stage = nullptr;          // page-locked: [blob | side records | packed inputs]; the walk uses the last part only
    uint8_t *d_stage = nullptr;          // the same layout on the device, + [decoder frame headers | table-index words]
    size_t blob_cap = 0, side_cap = 0 /* frames */, in_cap = 0, fix_cap = 0 /* entries */, pack_cap = 0, o_side = 0, o_in = 0, o_dechdr = 0, o_tsel = 0,
           stage_bytes = 0;
    uint8_t *d_image = nullptr; size_t image_cap = 0;   // the file bytes of a walked job
    uint8_t *h_image = nullptr;                          // page-locked, made on first need: short files are laid end to end here first
    uint8_t *d_mp3 = nullptr; size_t mp3_cap = 0;
    int32_t *d_small = nullptr;
    // the encoder's intermediates of the slot's job (mdct, quantised lines, GrInfo, energies, scfsi): the slot's own, so
    // that a job whose cursor guess failed is resolved on them at collect time while later jobs have long been issued
    uint8_t *d_enc = nullptr; size_t enc_cap = 0;
    hipEvent_t e_start = nullptr, e_up = nullptr, e_in = nullptr, e_huff = nullptr, e_rate = nullptr, e_comp = nullptr, e_down = nullptr;
    bool busy = false;
};

struct Upload { size_t dst; const uint8_t *src; size_t bytes; };   // into the slot's d_image

// a chunk of one file (run_file): frames [w0, w0 + n_win) of the stream go to the device, of which the first `halo` only
// rebuild decoder state (IMDCT overlap, synthesis fifo: < 1 frame, Frame.py:151-153, 81-92) and the next `lead` only
// encoder state (filter bank + MDCT history: 1 056 samples, MP3_Encoder.py:356, 685, 747)
struct Chunk {
    bool on = false;
    const FrameRef *refs = nullptr;      // the stream's frames as walked (file / blob offsets of the whole stream)
    long w0 = 0, n_win = 0, first = 0, count = 0;
    int halo = 0, lead = 0;
    bool last = false;
    bool has_carry = false; mp3s_carry carry_in = {};
    int out_format = MP3S_PCM_I16;
    uint8_t *dst = nullptr;              // where the chunk's bytes go on the host (MP3 frames; PCM of a decode)
    const uint8_t *file = nullptr; size_t file_len = 0;
    uint32_t image_lo = 0, image_hi = 0; // the piece of the file that goes up
    const uint8_t *fix = nullptr;        // kPlaceEntry bytes: the stream's last frame decoded on the host (index in the window filled in here), or null
    const uint8_t *tables = nullptr; int n_tables = 0; int any_silent = -1;   // the walk's table counts for the chunk's own units
    const uint8_t *hide = nullptr; int n_hide = 0;
    int rate = 0, kbps = 0, nch = 2;
    bool decode = false;
};

struct Job {
    int64_t ticket = 0;
    int slot = -1;
    std::vector<std::pair<const uint8_t *, size_t>> files, msgs;   // borrowed until the job is collected
    bool clear_all = false;
    bool decode = false;                 // MP3 -> WAV (int16) instead of hide / clear
    enum State { QUEUED, ISSUED, SLOW_DONE } state = QUEUED;
    // fast path
    bool walked = false;                 // side info and main data are taken apart on the device (k_dec_parse)
    std::vector<Upload> ups;
    size_t o_small = 0, o_encblk = 0, o_fix = 0, o_refs = 0, o_streams = 0, pack_end = 0;   // packed inputs inside the slot's stage (from its start)
    size_t front_end = 0;                // a chunk of a one-file call: [o_small, front_end) is what the front end needs, the encoder's inputs lie behind
    int set = 0;                         // which of the two sets of Huffman outputs / PCM buffers the job has
    bool down_pending = false;           // the copies of its results are not queued yet (issue_down)
    uint32_t image_base = 0, md_base = 0;
    const uint8_t *d_file = nullptr; size_t file_need = 0;   // the whole file on the device (FileUp) instead of a piece in the slot's d_image: bytes [0, file_need) are read
    std::vector<uint32_t> stream_first;  // first frame of every stream of the batch
    std::vector<std::vector<uint8_t>> bits, guess;
    std::vector<EncSeg> segs;
    EncLayout L;
    EncDev dev;
    int rate = 0, kbps = 0;
    std::unique_ptr<mp3s_buf> res;
    // synchronous path
    mp3s_buf *slow_owner = nullptr;
    std::vector<mp3s_file> slow_out;
    std::vector<int32_t> slow_st;
    int slow_rc = 0;
    std::string slow_err;
    double scan_ms = 0, issue_ms = 0;
    int n_fix = 0;
    // decode jobs: per file the frames, rows, header fields, where its WAV starts in the result block, its stego bits
    struct DecFile { int n_frames, nch, rate, bit_rate; size_t wav_off, bits_off, n_bits; long first; };
    std::vector<DecFile> dec;
    std::vector<uint8_t> res_bits;
    int nch = 2, n_total = 0, max_p23 = 0;
    Chunk ck;
    // block jobs (mp3s_pipe_submit_block): one rank's share of a stream
    bool block = false;
    int rank = 0, world = 1;
    bool has_carry = false; mp3s_carry carry = {};
    std::vector<FrameRef> refs;          // the stream's frames as walked (ck.refs points here)
    std::vector<uint8_t> fix;            // its last frame decoded on the host (kPlaceEntry bytes), if that is needed
    mp3s_block blk = {};                 // what the caller gets
    mp3s_buf *blk_owner = nullptr;       // ... from the synchronous path
};

}  // namespace

// One-file calls (the context's own pipe): the whole file goes to the device in pieces that a helper thread queues while the
// caller walks the frame headers -- a copy from ordinary memory occupies the thread that queues it for as long as the copy
// takes (0.09 ms per 4 MB), and the chunk whose bytes are on the way is exactly the one the caller is busy preparing.
struct FileUp {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    bool stop = false, busy = false, started = false;
    const uint8_t *src = nullptr;
    std::vector<size_t> ends;            // piece i = bytes [ends[i - 1], ends[i])
    std::vector<hipEvent_t> ev;          // ... is on the device when ev[i] has passed (recorded on s_img)
    std::atomic<long> recorded{0};       // pieces whose copy and event are queued
    std::atomic<int> err{0};
    uint8_t *d_file = nullptr; size_t cap = 0;
    bool active = false;                 // the call in progress reads its file from d_file
};

struct WalkOut {                         // the walker's state behind a chunk of a one-file call
    long got = 0;
    bool ended = false, irregular = false, dup_last = false, any_silent = false, have_fix = false;
    int nch = 0, sampling_rate = 0, bit_rate = 0, max_p23 = 0;
    long tables_frames = 0;
};

struct mp3s_pipe {
    mp3s_ctx *c = nullptr;
    int depth = 0;
    hipStream_t s_img = nullptr;         // the file pieces' own copy stream (the packed inputs of a chunk must not queue behind them)
    FileUp up;
    bool internal = false;               // the context's own (run_file): no worker threads, jobs issued by the caller
    size_t max_job_bytes = 0;
    std::vector<Slot> slots;
    hipStream_t s_up = nullptr, s_down = nullptr;
    hipStream_t s_dec = nullptr;         // the decode transforms of job k+1 under the encode transforms and the rate loop of job k (null: on the compute stream)
    hipEvent_t e_enc[2] = {nullptr, nullptr}; bool enc_used[2] = {false, false};   // the encode transforms that read PCM buffer x last are done
    hipStream_t s_comp = nullptr, s_ctx = nullptr;   // a compute stream of the pipe's own (pick_lanes), and the context's while the pipe has put its own in its place
    // The Huffman kernel is a latency chain that leaves the vector units mostly idle; the rate loop is bound by them.  The
    // front end of job k+1 therefore runs on a stream of its own, under the encode half of job k, with two sets of
    // Huffman outputs (is / side records) taken in turn; e_dec[x] = the decode transforms that read set x last are done.
    hipStream_t s_huff = nullptr;
    // ... and, optionally (MP3S_OPT_PIPE_TAIL; measured slower here, see mp3s_pipe_create), the tail of a job (chain check +
    // bit packing) on another one, under the decode transforms of the next job; e_rate orders it behind the job's rate loop
    hipStream_t s_tail = nullptr;
    int last_tail = -1;                  // slot of the job whose tail was issued last
    bool tail_throttle = false;
    size_t direct_upload = kDirectUpload;
    hipEvent_t e_dec[2] = {nullptr, nullptr};
    bool dec_used[2] = {false, false};
    unsigned issued = 0;
    // the PCM of a batch lives in one of two device buffers taken in turn; a decode job downloads from it while the
    // next job computes: keep_slot[x] = slot of the job whose download reads buffer x last (-1: none)
    int keep_slot[2] = {-1, -1};
    std::mutex mu;                       // queue, job states, slots, statistics
    std::condition_variable cv_work, cv_done, cv_turn;
    int64_t next_issue = 0;              // ticket of the job whose turn it is to be issued
    std::mutex mu_issue;                 // everything that touches the context (its stream, pool, profiler)
    std::deque<std::unique_ptr<Job>> inflight;   // ticket order; front = next to collect
    // One queue per worker, and a slot always goes to the same worker (slot % workers): the staging of a slot stays in
    // the cache hierarchy of the core that wrote it last, and a scan from another core complex would fetch every line it
    // overwrites from there (measured: 0.75 ms per 10 000 frames on the slot's own worker, 2.5 ms on changing ones).
    std::vector<std::deque<Job *>> todo;
    std::vector<std::thread> workers;
    std::vector<int> node_cpus;          // CPUs of the GPU's NUMA node this process may run on (empty: unknown / no binding)
    bool stop = false;
    int64_t next_ticket = 0;
    mp3s_pipe_stats st = {};
};

namespace {

double thread_cpu_ms()
{
    timespec ts;
    clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

int reencode_params(int sampling_rate, int bit_rate, int nch, long n_frames, int dup_last, int *kbps_out)
{
    const int kbps = bit_rate / 1000;
    int sri, bri, whole;
    if (sampling_rate != 32000 && sampling_rate != 44100 && sampling_rate != 48000) return 1;
    if (kbps <= 0 || stream_params(sampling_rate, kbps, &sri, &bri, &whole)) return 1;
    if (nch != 2 || n_frames <= 0 || dup_last) return 1;
    *kbps_out = kbps;
    return 0;
}

// the result block of a decode job: per file a WAV image, its PCM 64 bytes into an aligned region, the 44-byte header right
// in front of it (what mp3s_decode_file hands out)
bool decode_result(Job &j)
{
    size_t off = 0;
    for (auto &d : j.dec) {
        d.wav_off = off + 64 - 44;
        off += 64 + (((size_t)d.n_frames * 1152 * d.nch * 2 + 63) & ~(size_t)63);
    }
    j.res.reset(new mp3s_buf());
    if (!j.res->big[0].reserve(off) || !j.res->big[2].reserve(small_bytes(1))) return false;
    if (j.walked && !j.res->big[1].reserve((size_t)j.n_total * 8 + 16)) return false;   // the table-index words the stego bits are made of
    j.res->mp3 = j.res->big[0].data();
    for (const auto &d : j.dec) wav_header((int64_t)d.n_frames * 1152, d.nch, d.rate, j.res->mp3 + d.wav_off);
    return true;
}

bool encode_inputs(Job &j, Slot &s, uint8_t *encblk, size_t room, bool select)
{
    if (enc_layout(j.segs, j.rate, j.kbps, j.L, select)) return false;
    if (j.L.bytes > room) return false;
    if (enc_fill(j.segs, j.L, encblk)) return false;
    if (j.L.mp3_bytes + 16 > s.mp3_cap) return false;
    return true;
}

// ---- the walk: frame table + what the encoder needs of every stream, packed [encoder inputs | host-decoded frames | refs |
//      streams] behind o_in; false = not for this path (the byte-level scan, or the synchronous path, takes the job)
bool prepare_walk(mp3s_pipe *P, Job &j, Slot &s)
{
    const int nf = (int)j.files.size();
    if (nf > kMaxFastFiles) return false;
    j.walked = true;
    j.segs.assign((size_t)nf, EncSeg());
    j.bits.assign((size_t)nf, {});
    j.guess.assign((size_t)nf, {});
    j.dec.clear(); j.res_bits.clear(); j.ups.clear(); j.stream_first.assign((size_t)nf, 0);
    j.n_fix = 0; j.image_base = 0; j.md_base = 0;
    // refs and streams are written at the END of the packed area first (their place depends on the encoder block's size) and
    // moved down once that is known; room for them is what the packed area has behind in_cap
    FrameRef *refs = reinterpret_cast<FrameRef *>(s.h_stage + s.o_in + s.in_cap + s.fix_cap * kPlaceEntry);
    StreamRef *streams = reinterpret_cast<StreamRef *>(reinterpret_cast<uint8_t *>(refs) + s.side_cap * sizeof(FrameRef));
    uint8_t *fix = s.h_stage + s.o_in + s.in_cap;
    long n = 0;
    size_t img = 0;                  // bytes of d_image in use
    size_t run_lo = 0, run_hi = 0;   // short files collected in h_image since the last flush
    uint32_t md = 0;
    int max_p23 = 0;
    for (int i = 0; i < nf; i++) {
        const uint8_t *file = j.files[i].first;
        const size_t len = j.files[i].second;
        if (!file) return false;
        img = (img + 15) & ~(size_t)15;
        if (img + len + 64 > s.image_cap || img + len > 0xfffffff0ull) return false;
        FrameWalker w;
        if (w.open(file, len)) return false;
        w.md_cursor = md;
        const bool hiding = !j.decode && !j.clear_all && j.msgs[i].first;
        if (hiding) {
            message_frame(j.msgs[i].first, j.msgs[i].second, j.bits[i]);
            if (j.bits[i].size() > 0x7fffffff) return false;
            w.tables_wanted = (long)j.bits[i].size() + (long)j.bits[i].size() / 16 + 64;
            j.guess[i].resize((len / 96 + 16) * 4);     // (a silence offers no tables: the counting may go on for the whole file)
        }
        long got = 0;
        while (!w.ended && !w.irregular) {
            const long room = (long)s.side_cap - n - got;
            if (room <= 0) return false;
            uint8_t *tb = hiding && (size_t)(got + 1) * 4 <= j.guess[i].size() && w.tables_frames == got ? j.guess[i].data() + (size_t)got * 4 : nullptr;
            const long cap = tb ? std::min<long>(room, (long)(j.guess[i].size() / 4) - got) : room;
            got += w.next(refs + n + got, cap, tb, (uint32_t)img, (uint16_t)i);
        }
        if (w.irregular || got <= 0) return false;
        if (j.decode) {
            if (w.dup_last || w.nch < 1 || w.nch > 2) return false;
            if (i == 0) j.nch = w.nch;
            else if (w.nch != j.nch) return false;        // one device batch per channel count
            j.dec.push_back({(int)got, w.nch, w.sampling_rate, w.bit_rate, 0, 0, 0, n});
        } else {
            int kbps = 0;
            if (reencode_params(w.sampling_rate, w.bit_rate, w.nch, got, w.dup_last ? 1 : 0, &kbps)) return false;
            if (i == 0) { j.rate = w.sampling_rate; j.kbps = kbps; }
            else if (w.sampling_rate != j.rate || kbps != j.kbps) return false;   // more than one device batch
        }
        StreamRef &sr = streams[i];
        std::memset(&sr, 0, sizeof sr);
        sr.base = (uint32_t)img; sr.end = (uint32_t)(img + len); sr.first_frame = (uint32_t)n; sr.n_frames = (uint32_t)got;
        FrameWalker::history(refs + n, 0, sr.prev_size);
        j.stream_first[(size_t)i] = (uint32_t)n;
        max_p23 = std::max(max_p23, w.max_p23);
        // The last frame of a stream the reference's encoder wrote lacks the 0-3 bytes its writer drops (E14), and in one
        // file out of eight the Huffman data reaches into them: the device kernel would flag it.  One frame per stream is
        // cheap on the host, so it is decoded here, the kernel skips it, and its samples are placed behind the kernel --
        // unless it inherits scalefactors from earlier frames (then the kernel's walk back through the stream has the answer).
        if ((size_t)j.n_fix >= s.fix_cap) return false;
        {
            uint8_t *e = fix + (size_t)j.n_fix * kPlaceEntry;
            bool alone = false;
            std::memset(e, 0, 16);
            *reinterpret_cast<int32_t *>(e) = (int32_t)(n + got - 1);
            if (w.decode_last(reinterpret_cast<int16_t *>(e + 16), reinterpret_cast<mp3s_granule_si *>(e + 16 + 4608), &alone) == 0 && alone) {
                refs[n + got - 1].flags |= MP3S_FS_HOST_DECODED;
                j.n_fix++;
            }
        }
        j.segs[i].n_frames = (int)got;
        if (hiding) {
            j.segs[i].hide = j.bits[i].data(); j.segs[i].n_hide = (int)j.bits[i].size();
            j.segs[i].tables_guess = j.guess[i].data(); j.segs[i].n_guess = (int)std::min<long>(w.tables_frames, got) * 4;
            j.segs[i].any_silent = w.any_silent ? 1 : 0;
        }
        // the file's bytes: long files go up from where they lie, short ones are laid end to end in page-locked staging first
        if (len >= P->direct_upload) {
            if (run_hi > run_lo) { j.ups.push_back({run_lo, s.h_image + run_lo, run_hi - run_lo}); run_lo = run_hi = 0; }
            j.ups.push_back({img, file, len});
        } else {
            if (!s.h_image && hipHostMalloc((void **)&s.h_image, s.image_cap, hipHostMallocDefault) != hipSuccess) { s.h_image = nullptr; return false; }
            if (run_hi == run_lo) run_lo = img;
            std::memcpy(s.h_image + img, file, len);
            run_hi = img + len;
        }
        img += len;
        n += got;
        md = w.md_cursor;
        if ((size_t)md + 64 > s.blob_cap) return false;
    }
    if (run_hi > run_lo) j.ups.push_back({run_lo, s.h_image + run_lo, run_hi - run_lo});
    j.n_total = (int)n;
    j.L = EncLayout();
    j.max_p23 = max_p23;
    size_t enc_bytes = 0;
    // pack: [small results, zeroed | encoder inputs | host-decoded frames | refs | streams], one copy up
    j.o_small = s.o_in;
    const size_t small_room = (small_bytes(j.decode ? 1 : nf) + 15) & ~(size_t)15;
    std::memset(s.h_stage + j.o_small, 0, kSmallHead);
    j.o_encblk = j.o_small + small_room;
    if (j.decode) {
        if (!decode_result(j)) return false;
    } else {
        if (small_room >= s.in_cap || !encode_inputs(j, s, s.h_stage + j.o_encblk, s.in_cap - small_room, P->c->opt[MP3S_OPT_SELECT] != 0)) return false;
        enc_bytes = j.L.bytes;
        j.res.reset(new mp3s_buf());
        if (!j.res->big[0].reserve(j.L.mp3_bytes) || !j.res->big[2].reserve(small_bytes(j.L.n_segs))) return false;
        j.res->mp3 = j.res->big[0].data();
    }
    if (trace_on() && !j.decode) fprintf(stderr, "mp3s:   walk job %lld: %d stream(s), redo launches %d, variant entries %d, first stream: any_silent %d, tables known for %d units, reach %d\n",
                                         (long long)j.ticket, nf, (int)j.L.redo, j.L.n_entries, j.segs[0].any_silent, j.segs[0].n_guess, j.segs[0].reach);
    j.o_fix = (j.o_encblk + enc_bytes + 15) & ~(size_t)15;
    if (j.o_fix != s.o_in + s.in_cap) std::memmove(s.h_stage + j.o_fix, fix, (size_t)j.n_fix * kPlaceEntry);
    j.o_refs = (j.o_fix + (size_t)j.n_fix * kPlaceEntry + 15) & ~(size_t)15;
    std::memmove(s.h_stage + j.o_refs, refs, (size_t)n * sizeof(FrameRef));
    j.o_streams = (j.o_refs + (size_t)n * sizeof(FrameRef) + 15) & ~(size_t)15;
    std::memmove(s.h_stage + j.o_streams, streams, (size_t)nf * sizeof(StreamRef));
    j.pack_end = j.o_streams + (size_t)nf * sizeof(StreamRef);
    j.ck.on = false;
    return true;
}

// ---- a chunk of one file: the same packed inputs for frames [w0, w0 + n_win) of a stream that run_file has walked
void bind_to(const std::vector<int> &cpus);

// ---- the whole file of a one-file call on the device (FileUp)
constexpr size_t kFileOnDevice = (size_t)1 << 30;       // longer files: chunk by chunk through the slots' own image buffers
constexpr size_t kFilePiece = (size_t)4 << 20;

void file_up_thread(mp3s_pipe *P)
{
    FileUp &u = P->up;
    (void)hipSetDevice(P->c->device);
    if (!P->node_cpus.empty()) bind_to(P->node_cpus);
    std::unique_lock<std::mutex> lk(u.mu);
    for (;;) {
        u.cv.wait(lk, [&] { return u.stop || (u.busy && !u.started); });
        if (u.stop) return;
        u.started = true;
        lk.unlock();
        // (piece 0 is the caller's own: it needs it first, and this thread takes longer to wake up than the copy takes)
        while (u.recorded.load(std::memory_order_acquire) < 1 && !u.err.load()) std::this_thread::yield();
        size_t from = u.ends[0];
        for (size_t i = 1; i < u.ends.size() && !u.err.load(); i++) {
            if (hipMemcpyAsync(u.d_file + from, u.src + from, u.ends[i] - from, hipMemcpyHostToDevice, P->s_img) != hipSuccess ||
                hipEventRecord(u.ev[i], P->s_img) != hipSuccess) { u.err.store(1); break; }
            from = u.ends[i];
            u.recorded.store((long)i + 1, std::memory_order_release);
        }
        lk.lock();
        u.busy = false;
        u.cv.notify_all();
    }
}

void file_up_end(mp3s_pipe *P);

// queue the upload of file[0, len): a first piece of first_bytes (the first chunk's), then kFilePiece at a time
bool file_up_begin(mp3s_pipe *P, const uint8_t *file, size_t len, size_t first_bytes)
{
    FileUp &u = P->up;
    u.active = false;
    if (!P->internal || !P->s_img || len > kFileOnDevice || getenv("MP3S_NO_FILE_UP")) return false;
    if (len + 256 > u.cap) {
        if (u.d_file) (void)hipFree(u.d_file);
        u.d_file = nullptr; u.cap = 0;
        const size_t want = std::max<size_t>(len + len / 4 + 4096, (size_t)8 << 20);
        if (hipMalloc((void **)&u.d_file, want) != hipSuccess) { (void)hipGetLastError(); return false; }
        u.cap = want;
    }
    u.ends.clear();
    size_t at = std::min(len, std::max<size_t>(first_bytes, 4096));
    u.ends.push_back(at);
    while (at < len) { at = std::min(len, at + kFilePiece); u.ends.push_back(at); }
    while (u.ev.size() < u.ends.size()) {
        hipEvent_t e = nullptr;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return false;
        u.ev.push_back(e);
    }
    u.recorded.store(0); u.err.store(0);
    {
        std::lock_guard<std::mutex> g(u.mu);
        if (!u.th.joinable()) {
            try { u.th = std::thread(file_up_thread, P); }
            catch (const std::exception &) { return false; }     // (no thread to be had: every chunk's piece from the caller, as for a file above 1 GB)
        }
        u.src = file; u.busy = true; u.started = false;
    }
    u.cv.notify_all();
    u.active = true;
    if (hipMemcpyAsync(u.d_file, file, u.ends[0], hipMemcpyHostToDevice, P->s_img) != hipSuccess || hipEventRecord(u.ev[0], P->s_img) != hipSuccess) {
        (void)hipGetLastError();
        u.err.store(1);
        file_up_end(P);
        return false;
    }
    u.recorded.store(1, std::memory_order_release);
    return true;
}

// `stream` waits until file[0, need) is on the device
int file_up_wait(mp3s_pipe *P, size_t need, hipStream_t stream)
{
    FileUp &u = P->up;
    size_t i = 0;
    while (i + 1 < u.ends.size() && u.ends[i] < need) i++;
    while (u.recorded.load(std::memory_order_acquire) <= (long)i) {
        if (u.err.load()) return fail(MP3S_E_HIP, "uploading the file failed");
        std::this_thread::yield();
    }
    HIPCHK(hipStreamWaitEvent(stream, u.ev[i], 0));
    return MP3S_OK;
}

// the call is over: the helper is done with the caller's memory
void file_up_end(mp3s_pipe *P)
{
    FileUp &u = P->up;
    if (!u.active) return;
    std::unique_lock<std::mutex> lk(u.mu);
    u.cv.wait(lk, [&] { return !u.busy; });
    u.active = false;
}

bool prepare_chunk(mp3s_pipe *P, Job &j, Slot &s, int max_p23)
{
    Chunk &k = j.ck;
    j.walked = true; j.decode = k.decode; j.clear_all = k.n_hide == 0;
    j.segs.assign(1, EncSeg());
    j.dec.clear(); j.ups.clear(); j.stream_first.assign(1, 0);
    j.n_fix = 0;
    j.n_total = (int)k.n_win; j.nch = k.nch; j.rate = k.rate; j.kbps = k.kbps;
    const bool on_device = P->up.active;
    if ((size_t)k.n_win > s.side_cap || (!on_device && k.image_hi - k.image_lo + 64 > s.image_cap)) return false;
    j.image_base = on_device ? 0 : k.image_lo; j.md_base = k.refs[k.w0].md_off;
    j.d_file = on_device ? P->up.d_file : nullptr; j.file_need = k.image_hi;
    const FrameRef &lastr = k.refs[k.w0 + k.n_win - 1];
    if ((size_t)lastr.md_off - j.md_base + lastr.md_len + 64 > s.blob_cap) return false;
    if (!on_device) j.ups.push_back({0, k.file + k.image_lo, (size_t)(k.image_hi - k.image_lo)});
    j.L = EncLayout();
    // packed [small results | host-decoded frame | refs | stream | encoder inputs]: what the front end reads comes first and goes
    // up first -- the encoder's inputs are laid out (prepare_chunk_encode) while parse and Huffman kernels already run
    j.o_small = s.o_in;
    std::memset(s.h_stage + j.o_small, 0, kSmallHead);
    j.res.reset(new mp3s_buf());
    if (!j.res->big[2].reserve(small_bytes(1))) return false;
    if (k.decode && !j.res->big[1].reserve((size_t)k.n_win * 8 + 16)) return false;
    j.o_fix = j.o_small + ((small_bytes(1) + 15) & ~(size_t)15);
    if (k.fix) {
        if (s.fix_cap < 1) return false;
        std::memcpy(s.h_stage + j.o_fix, k.fix, kPlaceEntry);
        *reinterpret_cast<int32_t *>(s.h_stage + j.o_fix) = (int32_t)(k.n_win - 1);
        j.n_fix = 1;
    }
    j.o_refs = (j.o_fix + (size_t)j.n_fix * kPlaceEntry + 15) & ~(size_t)15;
    FrameRef *refs = reinterpret_cast<FrameRef *>(s.h_stage + j.o_refs);
    std::memcpy(refs, k.refs + k.w0, (size_t)k.n_win * sizeof(FrameRef));
    for (long f = 0; f < k.n_win; f++) { refs[f].stream = 0; refs[f].flags = 0; }
    if (k.fix) refs[k.n_win - 1].flags = MP3S_FS_HOST_DECODED;
    j.o_streams = (j.o_refs + (size_t)k.n_win * sizeof(FrameRef) + 15) & ~(size_t)15;
    StreamRef *sr = reinterpret_cast<StreamRef *>(s.h_stage + j.o_streams);
    std::memset(sr, 0, sizeof *sr);
    sr->base = 0; sr->end = (uint32_t)k.file_len; sr->first_frame = 0; sr->n_frames = (uint32_t)k.n_win;
    FrameWalker::history(k.refs, k.w0, sr->prev_size);
    j.o_encblk = (j.o_streams + sizeof(StreamRef) + 15) & ~(size_t)15;
    j.front_end = j.pack_end = j.o_encblk;
    if (j.pack_end > s.o_in + s.pack_cap) return false;
    j.max_p23 = max_p23;
    return true;
}

// ... and the encoder's inputs of the chunk behind them
bool prepare_chunk_encode(mp3s_pipe *P, Job &j, Slot &s)
{
    Chunk &k = j.ck;
    if (k.decode) return true;
    EncSeg &sg = j.segs[0];
    sg.n_frames = (int)k.count; sg.hide = k.hide; sg.n_hide = k.n_hide;
    sg.lead = k.lead; sg.first_frame = k.first; sg.last = k.last; sg.carry_in = k.has_carry ? &k.carry_in : nullptr;
    sg.tables_guess = k.tables; sg.n_guess = k.tables ? k.n_tables : -1; sg.any_silent = k.any_silent;
    if (!encode_inputs(j, s, s.h_stage + j.o_encblk, s.o_in + s.pack_cap - j.o_encblk, P->c->opt[MP3S_OPT_SELECT] != 0)) return false;
    j.pack_end = j.o_encblk + j.L.bytes;
    return j.pack_end <= s.o_in + s.pack_cap;
}

// ---- a block job: walk the stream, cut out the rank's share (as mp3s_reencode_block cuts it), and queue it as a chunk
bool prepare_block(mp3s_pipe *P, Job &j, Slot &s)
{
    const uint8_t *file = j.files[0].first;
    const size_t len = j.files[0].second;
    if (!file || len > 0xffff0000ull) return false;
    FrameWalker w;
    if (w.open(file, len) || w.ended) return false;
    j.refs.resize(len / 24 + 16);
    j.bits.assign(1, {});
    if (j.msgs[0].first) {
        message_frame(j.msgs[0].first, j.msgs[0].second, j.bits[0]);
        if (j.bits[0].size() > 0x3fffff00) return false;
    }
    long n = 0;
    while (!w.ended && !w.irregular && (size_t)n < j.refs.size()) n += w.next(j.refs.data() + n, (long)j.refs.size() - n, nullptr, 0, 0);
    if (w.irregular || !w.ended || n <= 0 || w.dup_last) return false;
    int kbps = 0;
    if (reencode_params(w.sampling_rate, w.bit_rate, w.nch, n, 0, &kbps)) return false;
    const long base = n / j.world, rem = n % j.world;
    const long first = j.rank * base + std::min<long>(j.rank, rem), count = base + (j.rank < rem ? 1 : 0);
    std::memset(&j.blk, 0, sizeof j.blk);
    j.blk.total_frames = n; j.blk.first_frame = first; j.blk.n_frames = count; j.blk.is_last = first + count == n;
    j.blk.file.kbps = kbps; j.blk.file.sampling_rate = w.sampling_rate; j.blk.file.channels = 2;
    if (count <= 0 || count > kDecodeChunk - 2) return false;    // (an empty share, or one longer than a transform group: the synchronous path)
    if ((first == 0) != !j.has_carry) return false;
    Chunk &k = j.ck;
    k = Chunk();
    k.on = true; k.decode = false; k.refs = j.refs.data(); k.first = first; k.count = count; k.last = j.blk.is_last != 0;
    k.lead = first > 0 ? 1 : 0; k.halo = first - k.lead > 0 ? 1 : 0;
    k.w0 = first - k.lead - k.halo; k.n_win = count + k.lead + k.halo;
    k.file = file; k.file_len = len; k.rate = w.sampling_rate; k.kbps = kbps; k.nch = 2;
    const uint32_t lo = j.refs[(size_t)k.w0].file_off;
    k.image_lo = first == 0 ? 0 : (lo > kImageLead ? lo - kImageLead : 0);
    const FrameRef &lr = j.refs[(size_t)(first + count - 1)];
    k.image_hi = (uint32_t)std::min<uint64_t>(len, (uint64_t)lr.file_off + lr.frame_size + 64);
    if (k.last) {
        j.fix.assign(kPlaceEntry, 0);
        bool alone = false;
        if (w.decode_last(reinterpret_cast<int16_t *>(j.fix.data() + 16), reinterpret_cast<mp3s_granule_si *>(j.fix.data() + 16 + 4608), &alone) == 0 && alone) k.fix = j.fix.data();
    }
    k.hide = j.bits[0].data(); k.n_hide = (int)j.bits[0].size();
    k.has_carry = j.has_carry; k.carry_in = j.carry;
    k.any_silent = w.any_silent ? 1 : 0;
    if (!prepare_chunk(P, j, s, w.max_p23) || !prepare_chunk_encode(P, j, s)) return false;
    if (!j.res->big[0].reserve(j.L.mp3_bytes + 16)) return false;
    j.res->mp3 = j.res->big[0].data();
    j.ck.dst = j.res->mp3;
    return true;
}

// scan the job's files into the slot's staging (round 2's byte-level scan) and lay out the encoder's inputs; false = this
// job takes the synchronous path
bool prepare_fast(mp3s_pipe *P, Job &j, Slot &s, ParsedStream &p, size_t *blob_len, int *max_p23)
{
    const int nf = (int)j.files.size();
    if (nf > kMaxFastFiles) return false;
    j.walked = false; j.ck.on = false;
    mp3s_frame_side *side = (mp3s_frame_side *)(s.h_stage + s.o_side);
    uint8_t *in = s.h_stage + s.o_in;
    size_t base = 0;
    long n = 0;
    j.segs.assign((size_t)nf, EncSeg());
    j.bits.assign((size_t)nf, {});
    j.guess.assign((size_t)nf, {});
    j.stream_first.assign((size_t)nf, 0);
    j.n_fix = 0;
    j.dec.clear(); j.res_bits.clear(); j.ups.clear();
    mp3s_frame_hdr *dechdr = (mp3s_frame_hdr *)in;      // the input block starts with the decoder's frame headers
    for (int i = 0; i < nf; i++) {
        if (!j.files[i].first) return false;
        ScanSink k;
        k.blob = s.h_stage + base; k.blob_cap = s.blob_cap - base;
        k.side = side + n; k.side_cap = s.side_cap - (size_t)n;
        k.hdr = dechdr + n;
        k.lean = !j.decode;                              // a decode job hands out the stego bits too
        if (parse_stream_sink(j.files[i].first, j.files[i].second, p, &k)) return false;
        if (j.decode) {
            if (p.n_frames <= 0 || p.dup_last_frame || p.nch < 1 || p.nch > 2) return false;
            if (i == 0) j.nch = p.nch;
            else if (p.nch != j.nch) return false;        // one device batch per channel count
            j.dec.push_back({p.n_frames, p.nch, p.sampling_rate, p.bit_rate, 0, j.res_bits.size(), p.bits.size(), n});
            j.res_bits.insert(j.res_bits.end(), p.bits.begin(), p.bits.end());
        } else {
            int kbps = 0;
            if (reencode_params(p.sampling_rate, p.bit_rate, p.nch, p.n_frames, p.dup_last_frame, &kbps)) return false;
            if (i == 0) { j.rate = p.sampling_rate; j.kbps = kbps; }
            else if (p.sampling_rate != j.rate || kbps != j.kbps) return false;   // more than one device batch
        }
        j.stream_first[(size_t)i] = (uint32_t)n;
        for (int f = 0; f < p.n_frames; f++) {
            side[n + f].md_off += (uint32_t)base;
            side[n + f].reserved = (uint32_t)n;           // where the stream starts in the batch (scalefactor inheritance walks back to it)
            dechdr[n + f].stream_first = (uint32_t)n;
        }
        *max_p23 = std::max(*max_p23, max_part2_3(side + n, p.n_frames));
        if (k.gpu_ok) {   // (a frame of a stream with inherited scalefactors cannot be decoded on its own; E14: see prepare_walk)
            if ((size_t)j.n_fix >= s.fix_cap) return false;
            uint8_t *e = s.h_stage + s.o_in + s.in_cap + (size_t)j.n_fix * kPlaceEntry;
            mp3s_frame_side &last = side[n + p.n_frames - 1];
            std::memset(e, 0, 16);
            *reinterpret_cast<int32_t *>(e) = (int32_t)(n + p.n_frames - 1);
            if (parse_scanned_frame(last, s.h_stage, reinterpret_cast<int16_t *>(e + 16), reinterpret_cast<mp3s_granule_si *>(e + 16 + 4608))) return false;
            last.flags |= MP3S_FS_HOST_DECODED;
            j.n_fix++;
        }
        j.segs[i].n_frames = p.n_frames;
        if (!j.decode && !j.clear_all && j.msgs[i].first) {
            message_frame(j.msgs[i].first, j.msgs[i].second, j.bits[i]);
            if (j.bits[i].size() > 0x7fffffff) return false;
            j.segs[i].hide = j.bits[i].data(); j.segs[i].n_hide = (int)j.bits[i].size();
            tables_guess_of(side + n, p.n_frames, 0, j.guess[i]);        // the cursor guess of the first pass: the input's own tables
            j.segs[i].tables_guess = j.guess[i].data();
        }
        n += p.n_frames;
        base = (base + k.blob_len + 3) & ~(size_t)3;
        if (base + 16 > s.blob_cap) return false;
    }
    *blob_len = base;
    j.n_total = (int)n;
    j.o_fix = s.o_in + s.in_cap;
    if (j.decode) {
        if (!decode_result(j)) return false;
        j.res->bits = std::move(j.res_bits);
        return true;
    }
    const size_t o_enc = ((size_t)n * sizeof(mp3s_frame_hdr) + 15) & ~(size_t)15;
    if (o_enc > s.in_cap || !encode_inputs(j, s, in + o_enc, s.in_cap - o_enc, P->c->opt[MP3S_OPT_SELECT] != 0)) return false;
    j.o_encblk = s.o_in + o_enc;
    j.res.reset(new mp3s_buf());
    if (!j.res->big[0].reserve(j.L.mp3_bytes) || !j.res->big[2].reserve(small_bytes(j.L.n_segs))) return false;
    j.res->mp3 = j.res->big[0].data();
    return true;
}

// everything a fast job does on the device, queued on the streams; nothing is waited for
// the front end of a job: uploads, side-info parse, Huffman decode (on the copy-up and front-end streams).  inputs_later: only
// [o_small, front_end) of the packed inputs goes up here, the rest with issue_back
int issue_front(mp3s_pipe *P, Job &j, Slot &s, size_t blob_len, int max_p23, bool inputs_later)
{
    mp3s_ctx *c = P->c;
    const double t_issue0 = trace_on() ? now_ms() : 0;
    const EncLayout &L = j.L;
    const Chunk &ck = j.ck;
    const int n = j.n_total, nch = j.decode ? j.nch : 2;
    const int out_format = ck.on && j.decode ? ck.out_format : MP3S_PCM_I16;
    const size_t esz = pcm_elem(out_format), frame_elems = (size_t)1152 * nch;
    const int set = j.set = (int)(P->issued++ & 1u);
    void *d_is = c->grab(set ? 24 : 0, (size_t)n * 2304 * 2), *d_si = c->grab(set ? 25 : 1, (size_t)n * 4 * sizeof(mp3s_granule_si)),
         *d_keep = c->grab(set ? 26 : 7, (size_t)n * frame_elems * esz);
    if (!d_is || !d_si || !d_keep) return fail(MP3S_E_NOMEM, "hipMalloc failed for a %d-frame job", n);
    uint8_t *d_blob = s.d_stage, *d_side = s.d_stage + s.o_side;
    const mp3s_frame_hdr *d_dechdr;
    HIPCHK(hipEventRecord(s.e_start, P->s_up));
    if (j.walked) {
        for (const Upload &u : j.ups) HIPCHK(hipMemcpyAsync(s.d_image + u.dst, u.src, u.bytes, hipMemcpyHostToDevice, P->s_up));
        HIPCHK(hipMemcpyAsync(s.d_stage + j.o_small, s.h_stage + j.o_small, (inputs_later ? j.front_end : j.pack_end) - j.o_small, hipMemcpyHostToDevice, P->s_up));
        d_dechdr = (const mp3s_frame_hdr *)(s.d_stage + s.o_dechdr);
    } else {
        const size_t o_enc = j.o_encblk - s.o_in, in_bytes = j.decode ? ((size_t)n * sizeof(mp3s_frame_hdr) + 15) & ~(size_t)15 : o_enc + L.bytes;
        HIPCHK(hipMemcpyAsync(d_blob, s.h_stage, blob_len, hipMemcpyHostToDevice, P->s_up));
        HIPCHK(hipMemcpyAsync(d_side, s.h_stage + s.o_side, (size_t)n * sizeof(mp3s_frame_side), hipMemcpyHostToDevice, P->s_up));
        HIPCHK(hipMemcpyAsync(s.d_stage + s.o_in, s.h_stage + s.o_in, in_bytes, hipMemcpyHostToDevice, P->s_up));
        HIPCHK(hipMemcpyAsync(s.d_stage + j.o_fix, s.h_stage + j.o_fix, (size_t)j.n_fix * kPlaceEntry, hipMemcpyHostToDevice, P->s_up));
        d_dechdr = (const mp3s_frame_hdr *)(s.d_stage + s.o_in);
    }
    HIPCHK(hipEventRecord(s.e_up, P->s_up));
    if (trace_on()) fprintf(stderr, "mp3s:   uploads queued %.3f ms after the job's start\n", now_ms() - t_issue0);
    HIPCHK(hipStreamWaitEvent(P->s_huff, s.e_up, 0));
    if (j.d_file) { const int rc = file_up_wait(P, j.file_need, P->s_huff); if (rc) return rc; }
    if (P->dec_used[set]) HIPCHK(hipStreamWaitEvent(P->s_huff, P->e_dec[set], 0));
    // the small results: a walked job brings its block, status words zeroed, with its inputs (the kernels OR into them directly);
    // a scanned one uses the slot's, written by the last workgroup of each kernel
    int32_t *const d_small = j.walked ? (int32_t *)(s.d_stage + j.o_small) : s.d_small;
    if (j.walked) {
        uint64_t *d_tsel = j.decode ? (uint64_t *)(s.d_stage + s.o_tsel) : nullptr;
        const int e = launch_parse(P->s_huff, j.d_file ? j.d_file : s.d_image, j.image_base, (const FrameRef *)(s.d_stage + j.o_refs), (const StreamRef *)(s.d_stage + j.o_streams), n,
                                   j.md_base, (mp3s_frame_side *)d_side, (mp3s_frame_hdr *)(s.d_stage + s.o_dechdr), d_blob, d_tsel, d_small + 4);
        if (e) return fail(MP3S_E_HIP, "parse launch: %s", hipGetErrorString((hipError_t)e));
    }
    const int e = launch_huffman(P->s_huff, d_blob, (const mp3s_frame_side *)d_side, n, nch, max_p23, (int16_t *)d_is, (mp3s_granule_si *)d_si,
                                 d_small + 3, j.walked ? nullptr : c->d_sync + 4, &c->prof, false);
    if (e) return fail(MP3S_E_HIP, "huffman launch: %s", hipGetErrorString((hipError_t)e));
    if (launch_place_frames(P->s_huff, s.d_stage + j.o_fix, j.n_fix, (int16_t *)d_is, (mp3s_granule_si *)d_si))
        return fail(MP3S_E_HIP, "placing the host-decoded frames failed");
    HIPCHK(hipEventRecord(s.e_huff, P->s_huff));
    if (trace_on()) fprintf(stderr, "mp3s:   front end queued %.3f ms after the job's start\n", now_ms() - t_issue0);
    return MP3S_OK;
}

int issue_down(mp3s_pipe *P, Job &j, Slot &s);

// ... and everything behind it: decode transforms, encode side, tail, download (defer_down: the download is queued by issue_down
// later -- a copy that waits for its job's kernels holds up every copy queued behind it, in either direction, on engines the
// runtime shares between the copy streams: the inputs of the next chunk of a one-file call must not queue behind it)
int issue_back(mp3s_pipe *P, Job &j, Slot &s, bool inputs_later, bool defer_down = false)
{
    mp3s_ctx *c = P->c;
    const double t_issue0 = trace_on() ? now_ms() : 0;
    const EncLayout &L = j.L;
    const Chunk &ck = j.ck;
    const int n = j.n_total, nch = j.decode ? j.nch : 2;
    const int out_format = ck.on && j.decode ? ck.out_format : MP3S_PCM_I16;
    const size_t esz = pcm_elem(out_format), frame_elems = (size_t)1152 * nch;
    const int set = j.set;
    void *d_is = c->grab(set ? 24 : 0, (size_t)n * 2304 * 2), *d_si = c->grab(set ? 25 : 1, (size_t)n * 4 * sizeof(mp3s_granule_si)),
         *d_keep = c->grab(set ? 26 : 7, (size_t)n * frame_elems * esz);
    if (!d_is || !d_si || !d_keep) return fail(MP3S_E_NOMEM, "hipMalloc failed for a %d-frame job", n);
    const mp3s_frame_hdr *d_dechdr = j.walked ? (const mp3s_frame_hdr *)(s.d_stage + s.o_dechdr) : (const mp3s_frame_hdr *)(s.d_stage + s.o_in);
    int32_t *const d_small = j.walked ? (int32_t *)(s.d_stage + j.o_small) : s.d_small;
    if (inputs_later && j.pack_end > j.front_end) {
        HIPCHK(hipMemcpyAsync(s.d_stage + j.front_end, s.h_stage + j.front_end, j.pack_end - j.front_end, hipMemcpyHostToDevice, P->s_up));
        HIPCHK(hipEventRecord(s.e_in, P->s_up));
    }
    // the decode transforms: on a stream of their own where the pipe has one (they wait for scalar operands half of the time, the
    // encode transforms and the rate loop of the job in front are bound by the vector units: side by side a batch takes 4 % less)
    hipStream_t ds = P->s_dec ? P->s_dec : c->stream;
    HIPCHK(hipStreamWaitEvent(ds, s.e_huff, 0));
    // the PCM buffer of this set may still be read by the download of the decode job that used it last ...
    if (P->keep_slot[set] >= 0) HIPCHK(hipStreamWaitEvent(ds, P->slots[(size_t)P->keep_slot[set]].e_down, 0));
    P->keep_slot[set] = j.decode ? j.slot : -1;
    // ... or by the encode transforms of the job before last (on the compute stream)
    if (P->s_dec && P->enc_used[set]) HIPCHK(hipStreamWaitEvent(ds, P->e_enc[set], 0));
    // a transform group that starts inside a stream re-runs one frame of it for the state (the chunk's own halo comes first)
    auto inside_stream = [&](long f) {
        size_t k = std::upper_bound(j.stream_first.begin(), j.stream_first.end(), (uint32_t)f) - j.stream_first.begin();
        return k > 0 && j.stream_first[k - 1] < (uint32_t)f;
    };
    const int halo0 = ck.on ? ck.halo : 0;
    for (long start = halo0; start < n; start += kDecodeChunk) {
        const int halo = start == halo0 ? halo0 : (inside_stream(start) ? 1 : 0);
        const int cnt = (int)std::min<long>(kDecodeChunk, n - start) + halo;
        const int rc = decode_transform_chunk(c, (const int16_t *)d_is, (const mp3s_granule_si *)d_si, d_dechdr, start - halo, cnt, nch, halo,
                                              out_format, (uint8_t *)d_keep + (size_t)(start - halo0) * frame_elems * esz, ds);
        if (rc) return rc;
    }
    HIPCHK(hipEventRecord(P->e_dec[set], ds));
    if (trace_on()) fprintf(stderr, "mp3s:   decode transforms queued %.3f ms after the job's start\n", now_ms() - t_issue0);
    P->dec_used[set] = true;
    if (P->s_dec && !j.decode) HIPCHK(hipStreamWaitEvent(c->stream, P->e_dec[set], 0));   // the encode side starts when the PCM is there
    if (inputs_later && j.pack_end > j.front_end && !j.decode) HIPCHK(hipStreamWaitEvent(c->stream, s.e_in, 0));   // ... and its own inputs
    if (j.decode) {
        HIPCHK(hipEventRecord(s.e_comp, ds));
        j.down_pending = true;
        return defer_down ? MP3S_OK : issue_down(P, j, s);
    }
    const int units = L.units;
    const size_t b_mdct = (size_t)L.n_all * 2304 * 4, b_ix = (size_t)L.n * 2304 * 2, b_out = (size_t)units * sizeof(mp3s_gr_out),
                 b_en = ((size_t)units * 22 * 4 + 255) & ~(size_t)255, b_sc = (size_t)L.n * 8 * 4;
    const size_t need = b_mdct + b_ix + b_out + b_en + b_sc;
    if (need > s.enc_cap) {   // (hipFree waits for the device; only while the slot is growing to its job size)
        if (s.d_enc) (void)hipFree(s.d_enc);
        s.d_enc = nullptr; s.enc_cap = 0;
        if (hipMalloc((void **)&s.d_enc, need + need / 8) != hipSuccess) return fail(MP3S_E_NOMEM, "hipMalloc failed for a %d-frame job", n);
        s.enc_cap = need + need / 8;
    }
    void *d_agg = c->grab(15, chain_agg_bytes(L.n));
    if (!d_agg) return fail(MP3S_E_NOMEM, "hipMalloc failed for a %d-frame job", n);
    EncDev &dev = j.dev;
    dev.d_pcm = (const int16_t *)d_keep; dev.d_in = s.d_stage + j.o_encblk; dev.d_mdct_all = (int32_t *)s.d_enc; dev.d_ix = (int16_t *)(s.d_enc + b_mdct);
    dev.d_out = (mp3s_gr_out *)(s.d_enc + b_mdct + b_ix); dev.d_en = (int32_t *)(s.d_enc + b_mdct + b_ix + b_out); dev.d_agg = d_agg;
    dev.d_mp3 = s.d_mp3; dev.d_sc = (int32_t *)(s.d_enc + b_mdct + b_ix + b_out + b_en);
    dev.d_small = d_small; dev.direct_status = j.walked;
    if (!enc_variant_buffers(c, L, dev)) return fail(MP3S_E_NOMEM, "hipMalloc failed for %d variant entries", L.n_entries);
    const int rc = enc_issue(c, L, dev, P->s_tail, s.e_rate, P->s_tail && P->last_tail >= 0 && P->tail_throttle ? P->slots[(size_t)P->last_tail].e_comp : nullptr,
                             P->s_dec ? P->e_enc[set] : nullptr);
    if (rc) return rc;
    if (trace_on()) fprintf(stderr, "mp3s:   encode side queued %.3f ms after the job's start\n", now_ms() - t_issue0);
    if (P->s_dec) P->enc_used[set] = true;
    HIPCHK(hipEventRecord(s.e_comp, P->s_tail ? P->s_tail : c->stream));
    P->last_tail = (int)(&s - P->slots.data());
    j.down_pending = true;
    return defer_down ? MP3S_OK : issue_down(P, j, s);
}

// the copies of a job's results to the host, behind its last kernel (e_comp)
int issue_down(mp3s_pipe *P, Job &j, Slot &s)
{
    if (!j.down_pending) return MP3S_OK;
    j.down_pending = false;
    mp3s_ctx *c = P->c;
    const Chunk &ck = j.ck;
    const int n = j.n_total, nch = j.decode ? j.nch : 2;
    const int out_format = ck.on && j.decode ? ck.out_format : MP3S_PCM_I16;
    const size_t esz = pcm_elem(out_format), frame_elems = (size_t)1152 * nch;
    int32_t *const d_small = j.walked ? (int32_t *)(s.d_stage + j.o_small) : s.d_small;
    HIPCHK(hipStreamWaitEvent(P->s_down, s.e_comp, 0));
    if (j.decode) {
        void *d_keep = c->grab(j.set ? 26 : 7, (size_t)n * frame_elems * esz);
        if (!d_keep) return fail(MP3S_E_NOMEM, "hipMalloc failed for a %d-frame job", n);
        HIPCHK(hipMemcpyAsync(j.res->big[2].data(), d_small, kSmallHead, hipMemcpyDeviceToHost, P->s_down));
        if (j.walked) HIPCHK(hipMemcpyAsync(j.res->big[1].data(), s.d_stage + s.o_tsel, (size_t)n * 8, hipMemcpyDeviceToHost, P->s_down));
        if (ck.on) {
            HIPCHK(hipMemcpyAsync(ck.dst, d_keep, (size_t)ck.count * frame_elems * esz, hipMemcpyDeviceToHost, P->s_down));
        } else {
            size_t first = 0;
            for (const auto &d : j.dec) {
                const size_t bytes = (size_t)d.n_frames * frame_elems * 2;
                HIPCHK(hipMemcpyAsync(j.res->mp3 + d.wav_off + 44, (const int16_t *)d_keep + first * frame_elems, bytes, hipMemcpyDeviceToHost, P->s_down));
                first += (size_t)d.n_frames;
            }
        }
    } else {
        const size_t total = j.segs.back().mp3_off + j.segs.back().mp3_len;
        HIPCHK(hipMemcpyAsync(j.res->big[2].data(), d_small, small_bytes(j.L.n_segs), hipMemcpyDeviceToHost, P->s_down));
        if (total) HIPCHK(hipMemcpyAsync(ck.on ? ck.dst : j.res->mp3, s.d_mp3, total, hipMemcpyDeviceToHost, P->s_down));
    }
    HIPCHK(hipEventRecord(s.e_down, P->s_down));
    return MP3S_OK;
}


int issue_fast(mp3s_pipe *P, Job &j, Slot &s, size_t blob_len, int max_p23)
{
    const int rc = issue_front(P, j, s, blob_len, max_p23, false);
    return rc ? rc : issue_back(P, j, s, false);
}

void sync_all(mp3s_pipe *P)
{
    (void)hipStreamSynchronize(P->s_up); (void)hipStreamSynchronize(P->s_huff);
    if (P->s_dec) (void)hipStreamSynchronize(P->s_dec);
    (void)hipStreamSynchronize(P->c->stream);
    if (P->s_tail) (void)hipStreamSynchronize(P->s_tail);
    (void)hipStreamSynchronize(P->s_down);
}

void run_slow(mp3s_pipe *P, Job &j)   // mu_issue held
{
    const int nf = (int)j.files.size();
    std::vector<const uint8_t *> fp(nf), mp(nf);
    std::vector<size_t> fl(nf), ml(nf);
    for (int i = 0; i < nf; i++) {
        fp[i] = j.files[i].first; fl[i] = j.files[i].second;
        mp[i] = j.clear_all ? nullptr : j.msgs[i].first; ml[i] = j.clear_all ? 0 : j.msgs[i].second;
    }
    (void)hipStreamSynchronize(P->s_huff);   // the synchronous path uses the same Huffman output buffers
    if (P->s_dec) (void)hipStreamSynchronize(P->s_dec);   // ... and the decode scratch
    (void)hipStreamSynchronize(P->s_down);   // ... and the PCM buffer a download may still be reading
    if (P->s_tail) (void)hipStreamSynchronize(P->s_tail);   // ... and the context's chain scratch and packer words
    j.slow_out.assign((size_t)nf, mp3s_file());
    j.slow_st.assign((size_t)nf, 0);
    j.res.reset();
    // (the context's own pipe is not entered from here: this pipe owns the context)
    const int64_t keep = P->c->opt[MP3S_OPT_FILE_PIPELINE];
    P->c->opt[MP3S_OPT_FILE_PIPELINE] = 0;
    if (j.block) {
        j.slow_rc = mp3s_reencode_block(P->c, fp[0], fl[0], mp[0], ml[0], j.rank, j.world, j.has_carry ? &j.carry : nullptr, &j.blk_owner, &j.blk);
        j.slow_err = mp3s_last_error();
    } else if (j.decode) {
        // file by file through mp3s_decode_file; the owners travel in one
        std::unique_ptr<mp3s_buf> top(new mp3s_buf());
        for (int i = 0; i < nf; i++) {
            mp3s_buf *part = nullptr;
            j.slow_st[(size_t)i] = j.files[i].first ? mp3s_decode_file(P->c, j.files[i].first, j.files[i].second, &part, &j.slow_out[(size_t)i]) : MP3S_E_ARG;
            if (j.slow_st[(size_t)i]) { j.slow_err = mp3s_last_error(); std::memset(&j.slow_out[(size_t)i], 0, sizeof(mp3s_file)); }
            else top->parts.emplace_back(part);
        }
        j.slow_rc = MP3S_OK;
        j.slow_owner = top.release();
    } else {
        j.slow_rc = mp3s_hide_messages(P->c, fp.data(), fl.data(), nf, j.clear_all ? nullptr : mp.data(), ml.data(), &j.slow_owner, j.slow_out.data(),
                                       j.slow_st.data());
        j.slow_err = mp3s_last_error();
    }
    P->c->opt[MP3S_OPT_FILE_PIPELINE] = keep;
}

void bind_to(const std::vector<int> &cpus)
{
    if (cpus.empty()) return;
    cpu_set_t set;
    CPU_ZERO(&set);
    for (int cpu : cpus) if (cpu >= 0 && cpu < CPU_SETSIZE) CPU_SET(cpu, &set);
    (void)sched_setaffinity(0, sizeof set, &set);
}

void worker(mp3s_pipe *P, int me)
{
    (void)hipSetDevice(P->c->device);
    bind_to(P->node_cpus);   // the slot's page-locked staging lives on the GPU's NUMA node: so does the thread that fills it
    ParsedStream scratch;   // per worker, capacity kept from job to job
    for (;;) {
        Job *j = nullptr;
        {
            std::unique_lock<std::mutex> g(P->mu);
            std::deque<Job *> &q = P->todo[(size_t)me];
            P->cv_work.wait(g, [&] { return P->stop || !q.empty(); });
            if (q.empty()) return;       // stop
            j = q.front(); q.pop_front();
        }
        Slot &s = P->slots[(size_t)j->slot];
        const double t0 = now_ms(), c0 = thread_cpu_ms();
        size_t blob_len = 0;
        int max_p23 = 0;
        bool fast;
        if (j->block) {
            fast = P->c->opt[MP3S_OPT_DEVICE_PARSE] && prepare_block(P, *j, s);
            max_p23 = j->max_p23;
        } else {
            fast = P->c->opt[MP3S_OPT_DEVICE_PARSE] && prepare_walk(P, *j, s);
            if (fast) max_p23 = j->max_p23;
            else fast = prepare_fast(P, *j, s, scratch, &blob_len, &max_p23);
        }
        const double t1 = now_ms(), c1 = thread_cpu_ms();
        if (trace_on()) fprintf(stderr, "mp3s: pipe job %lld slot %d on cpu %d: %s + layout %.3f ms (cpu %.3f)%s\n", (long long)j->ticket, j->slot, sched_getcpu(), j->walked ? "walk" : "scan", t1 - t0, c1 - c0, fast ? "" : " -> synchronous path");
        Job::State st;
        {
            // Jobs go to the device in the order they were submitted: the streams are queues and results are collected in
            // ticket order, so a job that overtakes the one in front of it makes that one's collector wait a whole job longer
            // (two workers on four slots: 1.36 instead of 0.82 ms per batch).  Walks and scans still run side by side.
            std::unique_lock<std::mutex> g(P->mu);
            P->cv_turn.wait(g, [&] { return P->next_issue == j->ticket; });
        }
        {
            std::lock_guard<std::mutex> gi(P->mu_issue);
            if (fast && issue_fast(P, *j, s, blob_len, max_p23) != MP3S_OK) {
                sync_all(P);
                fast = false;
            }
            if (!fast) run_slow(P, *j);
            st = fast ? Job::ISSUED : Job::SLOW_DONE;
        }
        const double t2 = now_ms();
        {
            std::lock_guard<std::mutex> g(P->mu);
            j->scan_ms = t1 - t0; j->issue_ms = t2 - t1;   // (with the wait for its turn)
            j->state = st;
            P->next_issue = j->ticket + 1;
            P->st.scan_ms += t1 - t0; P->st.issue_ms += t2 - t1; P->st.scan_cpu_ms += c1 - c0;
        }
        P->cv_turn.notify_all();
        P->cv_done.notify_all();
    }
}

void free_slot(Slot &s)
{
    if (s.h_stage) (void)hipHostFree(s.h_stage);
    if (s.h_image) (void)hipHostFree(s.h_image);
    if (s.d_stage) (void)hipFree(s.d_stage);
    if (s.d_image) (void)hipFree(s.d_image);
    if (s.d_mp3) (void)hipFree(s.d_mp3);
    if (s.d_small) (void)hipFree(s.d_small);
    if (s.d_enc) (void)hipFree(s.d_enc);
    for (hipEvent_t e : {s.e_start, s.e_up, s.e_in, s.e_huff, s.e_rate, s.e_comp, s.e_down}) if (e) (void)hipEventDestroy(e);
    s = Slot();
}

// Candidate streams of a device, made once per process: four of the highest priority (copies) and four of the lowest (the
// front end: its workgroups fill in beside the compute stream's instead of competing with them).  The runtime multiplexes
// streams onto a few hardware queues; which queue a stream gets depends on everything the process has created before it
// (round 2 believed priorities chose the set of queues; a context's own pipe in front of a user's pipe showed otherwise:
// 0.97 - 1.25 instead of 0.81 ms per batch, tools/pipe_queue_probe.py; the first of three contexts ran its one-file calls at
// half speed, tools/bench_queue_probe4.py), and a pipeline whose streams share queues with its compute stream loses its
// overlap -- every kernel of it takes longer, not only the ones that wait.  The runtime does not tell which stream sits
// where, and a spin kernel beside an empty one or beside a small copy does not show it either (all eight candidates passed
// that test on a context that then ran at half speed).  So the pipe REHEARSES: four miniature jobs -- a copy up, a spin on
// the front-end stream, two spins on the compute stream, a copy down, chained by events exactly as issue_fast chains a
// job's stages -- through each rotation of the candidates, and keeps the rotation that got them through fastest
// (about 1 ms per rotation when a pipe is made).
struct LaneChoice { hipStream_t ctx_stream; int want_tail; hipStream_t up, down, huff, comp, tail, dec, img; };
struct LanePool {
    std::vector<LaneChoice> chosen;       // what the rehearsal decided for a context's stream (asked again only by another context)
    hipStream_t hi[4] = {nullptr, nullptr, nullptr, nullptr}, lo[4] = {nullptr, nullptr, nullptr, nullptr}, cs[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev[4][6] = {};
    hipEvent_t t0 = nullptr, t1 = nullptr;
    uint8_t *d_buf = nullptr, *h_buf = nullptr;
    bool ok = false;
};
constexpr size_t kRehearseBytes = (size_t)256 << 10;

// tail: the stream a job's last stage (three short spins: selection, chain check, packing) runs on, behind the job's compute
// stage and beside the next job's; null: on the compute stream itself
// dec: the stream a job's decode stage (the first of the two compute spins) runs on, ahead of the compute stage of the job in front
float rehearse(LanePool &lp, hipStream_t comp, hipStream_t up, hipStream_t down, hipStream_t huff, hipStream_t tail = nullptr, hipStream_t dec = nullptr)
{
    float ms = 1e9f;
    hipStream_t ts = tail ? tail : comp;
    hipStream_t dst = dec ? dec : comp;
    bool ok = hipEventRecord(lp.t0, up) == hipSuccess;
    for (int k = 0; k < 4 && ok; k++) {
        ok = hipMemcpyAsync(lp.d_buf + k * kRehearseBytes, lp.h_buf + k * kRehearseBytes, kRehearseBytes, hipMemcpyHostToDevice, up) == hipSuccess &&
             hipEventRecord(lp.ev[k][0], up) == hipSuccess && hipStreamWaitEvent(huff, lp.ev[k][0], 0) == hipSuccess && launch_spin(huff, 60) == 0 &&
             hipEventRecord(lp.ev[k][1], huff) == hipSuccess && hipStreamWaitEvent(dst, lp.ev[k][1], 0) == hipSuccess && launch_spin(dst, 50) == 0;
        if (ok && dec) ok = hipEventRecord(lp.ev[k][5], dec) == hipSuccess && hipStreamWaitEvent(comp, lp.ev[k][5], 0) == hipSuccess;
        ok = ok && launch_spin(comp, 50) == 0 && launch_noop(comp) == 0 && hipEventRecord(lp.ev[k][2], comp) == hipSuccess;
        if (ok && tail) ok = hipStreamWaitEvent(tail, lp.ev[k][2], 0) == hipSuccess;
        ok = ok && launch_spin(ts, 10) == 0 && launch_spin(ts, 10) == 0 && launch_spin(ts, 20) == 0 && hipEventRecord(lp.ev[k][4], ts) == hipSuccess &&
             hipStreamWaitEvent(down, lp.ev[k][4], 0) == hipSuccess &&
             hipMemcpyAsync(lp.h_buf + (4 + k) * kRehearseBytes, lp.d_buf + k * kRehearseBytes, kRehearseBytes, hipMemcpyDeviceToHost, down) == hipSuccess &&
             hipEventRecord(lp.ev[k][3], down) == hipSuccess;
    }
    ok = ok && hipEventRecord(lp.t1, down) == hipSuccess;
    (void)hipStreamSynchronize(up); (void)hipStreamSynchronize(huff); (void)hipStreamSynchronize(comp); (void)hipStreamSynchronize(down);
    if (tail) (void)hipStreamSynchronize(tail);
    if (dec) (void)hipStreamSynchronize(dec);
    if (!ok || hipEventElapsedTime(&ms, lp.t0, lp.t1) != hipSuccess) return 1e9f;
    return ms;
}

std::mutex &lane_mu() { static std::mutex *m = new std::mutex(); return *m; }
std::vector<LanePool> &lane_pools() { static auto *v = new std::vector<LanePool>(); return *v; }

int pick_lanes(mp3s_ctx *c, hipStream_t *up, hipStream_t *down, hipStream_t *huff, hipStream_t *comp /* a stream to compute on instead of the context's, or null */,
               hipStream_t *tail /* a stream for the tail of a job, or null */, int want_tail /* 0: none, 1: always, 2: if the rehearsal is faster with it */,
               hipStream_t *dec = nullptr /* a stream for the decode transforms, or null */, int want_dec = 1,
               hipStream_t *img = nullptr /* a second copy-up stream (the file pieces of a one-file call), or null */)
{
    std::lock_guard<std::mutex> g(lane_mu());
    auto &pools = lane_pools();
    if ((size_t)c->device >= pools.size()) pools.resize((size_t)c->device + 1);
    LanePool &lp = pools[(size_t)c->device];
    if (!lp.ok) {
        int prio_low = 0, prio_high = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
        const char *hp = getenv("MP3S_PIPE_HUFF_PRIO");
        const int huff_prio = hp ? atoi(hp) : prio_low;
        bool ok = hipEventCreate(&lp.t0) == hipSuccess && hipEventCreate(&lp.t1) == hipSuccess && hipMalloc((void **)&lp.d_buf, 4 * kRehearseBytes) == hipSuccess &&
                  hipHostMalloc((void **)&lp.h_buf, 8 * kRehearseBytes, hipHostMallocDefault) == hipSuccess;
        for (int k = 0; k < 4 && ok; k++)
            for (int q = 0; q < 6 && ok; q++) ok = hipEventCreateWithFlags(&lp.ev[k][q], hipEventDisableTiming) == hipSuccess;
        for (int i = 0; i < 4 && ok; i++) ok = hipStreamCreateWithPriority(&lp.hi[i], hipStreamNonBlocking, prio_high) == hipSuccess;
        for (int i = 0; i < 4 && ok; i++) ok = hipStreamCreateWithPriority(&lp.lo[i], hipStreamNonBlocking, huff_prio) == hipSuccess;
        for (int i = 0; i < 4 && ok; i++) ok = hipStreamCreateWithFlags(&lp.cs[i], hipStreamNonBlocking) == hipSuccess;
        if (!ok) return 1;   // (what was created stays with the process)
        std::memset(lp.h_buf, 0, 8 * kRehearseBytes);
        lp.ok = true;
    }
    for (const LaneChoice &k : lp.chosen)
        if (k.ctx_stream == c->stream && k.want_tail == want_tail) {
            *up = k.up; *down = k.down; *huff = k.huff;
            if (comp) *comp = k.comp;
            if (tail) *tail = k.tail;
            if (dec) *dec = k.dec;
            if (img) *img = k.img;
            return 0;
        }
    (void)hipStreamSynchronize(c->stream);
    (void)rehearse(lp, c->stream, lp.hi[0], lp.hi[1], lp.lo[0]);      // (first launches: not a measurement)
    // the context's own stream with every rotation of the lanes; if none of them gets the rehearsal through as fast as a
    // pipeline without shared queues does, the compute candidates too (the pipe then computes on one of those)
    int best = 0, best_cs = -1;
    float best_ms = 1e9f;
    std::string seen;
    auto tryout = [&](int ci, int r) {
        hipStream_t comp_s = ci < 0 ? c->stream : lp.cs[ci];
        const float ms = std::min(rehearse(lp, comp_s, lp.hi[r], lp.hi[(r + 1) & 3], lp.lo[r]), rehearse(lp, comp_s, lp.hi[r], lp.hi[(r + 1) & 3], lp.lo[r]));
        if (trace_on()) { char b[32]; snprintf(b, sizeof b, " %.3f", ms); seen += b; }
        if (ms < best_ms * 0.97f) { best_ms = ms; best = r; best_cs = ci; }
    };
    for (int r = 0; r < 4; r++) tryout(-1, r);
    const float own_ms = best_ms;
    const int own_best = best;
    if (comp && best_ms > 0.80f)          // (4 x (2 x 50 + 40) us on the compute stream behind one front-end spin never take less than 0.72 ms: this one lost its overlap somewhere)
        for (int ci = 0; ci < 4 && best_ms > 0.78f; ci++) {
            if (trace_on()) seen += " |";
            for (int r = 0; r < 4; r++) tryout(ci, r);
        }
    if (best_cs >= 0 && best_ms > own_ms * 0.93f) { best_cs = -1; best_ms = own_ms; best = own_best; }   // (not worth leaving the context's stream for)
    // a stream for the tail of a job (MP3S_OPT_PIPE_TAIL): the candidate, other than the compute stream, that gets the
    // rehearsal through fastest -- kept if that is faster than the tail on the compute stream itself
    int best_tail = -1;
    if (tail) {
        *tail = nullptr;
        hipStream_t comp_s = best_cs < 0 ? c->stream : lp.cs[best_cs];
        float tail_ms = best_ms * (want_tail == 1 ? 1.05f : 0.97f);   // (1: unless it clearly loses -- the real kernels gain more from it than spins do)
        if (trace_on()) seen += " | tail:";
        for (int ti = 0; ti < 4 && want_tail; ti++) {
            if (ti == best_cs) continue;
            const float ms = std::min(rehearse(lp, comp_s, lp.hi[best], lp.hi[(best + 1) & 3], lp.lo[best], lp.cs[ti]), rehearse(lp, comp_s, lp.hi[best], lp.hi[(best + 1) & 3], lp.lo[best], lp.cs[ti]));
            if (trace_on()) { char b[32]; snprintf(b, sizeof b, " %.3f", ms); seen += b; }
            if (ms < tail_ms) { tail_ms = ms; best_tail = ti; }
        }
        if (best_tail >= 0) *tail = lp.cs[best_tail];
    }
    // a stream for the decode transforms: the candidate (other than the compute and tail streams) that does best, if it gains
    int best_dec = -1;
    if (dec) {
        *dec = nullptr;
        hipStream_t comp_s = best_cs < 0 ? c->stream : lp.cs[best_cs];
        hipStream_t tail_s = best_tail >= 0 ? lp.cs[best_tail] : nullptr;
        float ref_ms = std::min(rehearse(lp, comp_s, lp.hi[best], lp.hi[(best + 1) & 3], lp.lo[best], tail_s), rehearse(lp, comp_s, lp.hi[best], lp.hi[(best + 1) & 3], lp.lo[best], tail_s));
        float dec_ms = ref_ms * (want_dec == 2 ? 10.f : 0.95f);
        if (trace_on()) { char b[48]; snprintf(b, sizeof b, " | dec (%.3f):", ref_ms); seen += b; }
        for (int di = 0; di < 4 && want_dec; di++) {
            if (di == best_cs || di == best_tail) continue;
            const float ms = std::min(rehearse(lp, comp_s, lp.hi[best], lp.hi[(best + 1) & 3], lp.lo[best], tail_s, lp.cs[di]), rehearse(lp, comp_s, lp.hi[best], lp.hi[(best + 1) & 3], lp.lo[best], tail_s, lp.cs[di]));
            if (trace_on()) { char b[32]; snprintf(b, sizeof b, " %.3f", ms); seen += b; }
            if (ms < dec_ms) { dec_ms = ms; best_dec = di; }
        }
        if (best_dec >= 0) *dec = lp.cs[best_dec];
    }
    if (trace_on()) fprintf(stderr, "mp3s: pipe lanes rehearsed:%s ms -> compute stream %d, rotation %d (%.3f ms), tail stream %d, decode stream %d\n", seen.c_str(), best_cs, best, best_ms, best_tail, best_dec);
    if (comp) *comp = best_cs >= 0 ? lp.cs[best_cs] : nullptr;
    *up = lp.hi[best]; *down = lp.hi[(best + 1) & 3]; *huff = lp.lo[best];
    if (img) *img = lp.hi[(best + 2) & 3];
    lp.chosen.push_back({c->stream, want_tail, *up, *down, *huff, comp ? *comp : nullptr, tail ? *tail : nullptr, dec ? *dec : nullptr, lp.hi[(best + 2) & 3]});
    return 0;
}

// a context is going away: its stream's address may come back as another context's
void forget_lanes(mp3s_ctx *c)
{
    std::lock_guard<std::mutex> g(lane_mu());
    auto &pools = lane_pools();
    if ((size_t)c->device >= pools.size()) return;
    auto &v = pools[(size_t)c->device].chosen;
    v.erase(std::remove_if(v.begin(), v.end(), [&](const LaneChoice &k) { return k.ctx_stream == c->stream; }), v.end());
}

int pipe_create(mp3s_ctx *c, int depth, size_t max_job_bytes, int scan_threads, bool internal, mp3s_pipe **out)
{
    *out = nullptr;
    HIPCHK(hipSetDevice(c->device));
    std::unique_ptr<mp3s_pipe> P(new mp3s_pipe());
    P->c = c; P->depth = depth; P->internal = internal; P->max_job_bytes = max_job_bytes;
    auto destroy = [&](int code, const char *what) {
        for (auto &s : P->slots) free_slot(s);                // (the streams belong to the device: pick_lanes)
        for (hipEvent_t e : P->e_dec) if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : P->e_enc) if (e) (void)hipEventDestroy(e);
        return fail(code, "%s", what);
    };
    // copy-up, copy-down and front-end streams that run beside this context's compute stream (pick_lanes above)
    if (pick_lanes(c, &P->s_up, &P->s_down, &P->s_huff, &P->s_comp, &P->s_tail, (int)c->opt[MP3S_OPT_PIPE_TAIL], &P->s_dec, getenv("MP3S_PIPE_DEC") ? atoi(getenv("MP3S_PIPE_DEC")) : 0, &P->s_img))   // (a stream of their own for the decode transforms: +4 % on a resident batch
                                                                                   // fed through four contexts (bench.py --decode-stream on), nothing in this pipe: off)
        return destroy(MP3S_E_HIP, "stream creation failed");
    if (hipEventCreateWithFlags(&P->e_dec[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&P->e_dec[1], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&P->e_enc[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&P->e_enc[1], hipEventDisableTiming) != hipSuccess)
        return destroy(MP3S_E_HIP, "stream creation failed");
    // The tail of a job (selection, chain check, bit packing) on a stream of its own, under the decode transforms of the next job:
    // worth 5 % on a resident batch fed through three contexts (bench.py, region (i)); in this pipe round 2 measured it slower
    // at every priority -- that was the lanes sharing queues, not the arrangement: with a tail stream the rehearsal vouches for
    // a batch takes 0.79 instead of 0.82 ms (tools/pipe_tail_probe.py).  MP3S_OPT_PIPE_TAIL: 0 off, 1 on, 2 if the rehearsal gains.
    P->tail_throttle = getenv("MP3S_PIPE_TAIL_THROTTLE") != nullptr;
    if (const char *du = getenv("MP3S_PIPE_DIRECT_UPLOAD")) P->direct_upload = (size_t)atoll(du);
    // the page-locked staging of the slots is allocated by a thread that runs on the GPU's NUMA node (first touch), and the
    // workers that fill it stay there
    P->node_cpus = gpu_node_cpus(c->device);
    cpu_set_t before;
    const bool rebind = !P->node_cpus.empty() && sched_getaffinity(0, sizeof before, &before) == 0;
    if (rebind) bind_to(P->node_cpus);
    P->slots.resize((size_t)depth);
    bool ok = true;
    const unsigned ord_flags = trace_on() ? hipEventDefault : hipEventDisableTiming;   // (a traced run prints when each stage of a chunk ended)
    for (auto &s : P->slots) {
        // main data: the file minus headers plus alignment and 8 zero bytes per frame; frames: 96 bytes is the smallest
        // Layer III frame (32 kbit/s at 48 kHz); anything denser (false syncs) overflows the sink and takes the other path
        s.blob_cap = (max_job_bytes + max_job_bytes / 8 + 4096 + 15) & ~(size_t)15;
        s.side_cap = max_job_bytes / 96 + 16;
        s.in_cap = (s.side_cap * (72 + 16) + max_job_bytes / 4 + (size_t)kMaxFastFiles * (sizeof(mp3s_chain_seg) + sizeof(mp3s_select_span)) + s.side_cap * 4 * MP3S_SELECT_VARIANTS * 8 + 4096 + 15) & ~(size_t)15;   // (+ variant entries: at most 10 per unit, 8 bytes each)
        s.o_side = s.blob_cap;
        s.o_in = (s.o_side + s.side_cap * sizeof(mp3s_frame_side) + 15) & ~(size_t)15;
        s.fix_cap = std::min<size_t>(kMaxFastFiles, s.side_cap);
        s.pack_cap = (s.in_cap + s.fix_cap * kPlaceEntry + s.side_cap * sizeof(FrameRef) + (size_t)kMaxFastFiles * sizeof(StreamRef) + 256 + 15) & ~(size_t)15;
        s.stage_bytes = s.o_in + s.pack_cap;
        s.o_dechdr = s.stage_bytes;
        s.o_tsel = (s.o_dechdr + s.side_cap * sizeof(mp3s_frame_hdr) + 15) & ~(size_t)15;
        s.image_cap = max_job_bytes + 2 * kImageLead + (size_t)kMaxFastFiles * 16 + 4096;
        s.mp3_cap = max_job_bytes + s.side_cap + 4096;
        if (hipHostMalloc((void **)&s.h_stage, s.stage_bytes, hipHostMallocDefault) != hipSuccess ||
            hipMalloc((void **)&s.d_stage, s.o_tsel + s.side_cap * 8 + 64) != hipSuccess || hipMalloc((void **)&s.d_image, s.image_cap) != hipSuccess ||
            hipMalloc((void **)&s.d_mp3, s.mp3_cap) != hipSuccess || hipMalloc((void **)&s.d_small, small_bytes(kMaxFastFiles)) != hipSuccess ||
            // (only e_start and e_down are read as times; the ordering events carry no time stamps: 1 % per job)
            hipEventCreate(&s.e_start) != hipSuccess || hipEventCreateWithFlags(&s.e_up, ord_flags) != hipSuccess || hipEventCreateWithFlags(&s.e_in, ord_flags) != hipSuccess ||
            hipEventCreateWithFlags(&s.e_huff, ord_flags) != hipSuccess || hipEventCreateWithFlags(&s.e_comp, ord_flags) != hipSuccess || hipEventCreateWithFlags(&s.e_rate, ord_flags) != hipSuccess ||
            // the collecting thread sleeps on this one instead of spinning: with one process per GPU on a shared host the
            // cores are needed by the scan workers (the wake-up latency disappears behind the jobs in flight); the
            // context's own pipe has no job behind the one it waits for and spins
            hipEventCreateWithFlags(&s.e_down, internal || getenv("MP3S_PIPE_SPIN") ? hipEventDefault : hipEventBlockingSync) != hipSuccess) {
            ok = false;
            break;
        }
    }
    if (rebind) (void)sched_setaffinity(0, sizeof before, &before);
    if (!ok) return destroy(MP3S_E_NOMEM, "slot allocation failed");
    if (P->s_comp && !internal) {   // a user's pipe owns its context: the context computes on the pipe's stream until the pipe is gone
        (void)hipStreamSynchronize(c->stream);
        P->s_ctx = c->stream; c->stream = P->s_comp;
    }
    P->todo.resize((size_t)std::max(scan_threads, 1));
    for (int t = 0; t < scan_threads; t++) P->workers.emplace_back(worker, P.get(), t);
    *out = P.release();
    return MP3S_OK;
}

}  // namespace

extern "C" {

int mp3s_pipe_create(mp3s_ctx *c, int depth, size_t max_job_bytes, int scan_threads, mp3s_pipe **out)
{
    if (!c || !out || depth < 1 || depth > 64 || scan_threads < 0 || scan_threads > 64 || max_job_bytes < 4096)
        return fail(MP3S_E_ARG, "bad argument (1 <= depth <= 64; 0 <= scan_threads <= 64, 0 = as many as this rank's share of the host's cores allows; max_job_bytes >= 4096)");
    if (scan_threads == 0) scan_threads = default_scan_threads(c);
    return pipe_create(c, depth, max_job_bytes, scan_threads, false, out);
}

void mp3s_pipe_destroy(mp3s_pipe *P)
{
    if (!P) return;
    {
        std::lock_guard<std::mutex> g(P->mu);
        P->stop = true;
        for (auto &q : P->todo) q.clear();
    }
    P->cv_work.notify_all();
    for (auto &t : P->workers) t.join();
    {
        std::lock_guard<std::mutex> g(P->up.mu);
        P->up.stop = true;
    }
    P->up.cv.notify_all();
    if (P->up.th.joinable()) P->up.th.join();
    (void)hipSetDevice(P->c->device);
    sync_all(P);
    if (P->s_img) (void)hipStreamSynchronize(P->s_img);
    for (hipEvent_t e : P->up.ev) (void)hipEventDestroy(e);
    if (P->up.d_file) (void)hipFree(P->up.d_file);
    if (P->s_ctx) { P->c->stream = P->s_ctx; P->s_ctx = nullptr; }
    for (auto &j : P->inflight) { if (j->slow_owner) mp3s_buf_free(j->slow_owner); if (j->blk_owner) mp3s_buf_free(j->blk_owner); }
    for (auto &s : P->slots) free_slot(s);
    for (hipEvent_t e : P->e_dec) (void)hipEventDestroy(e);
    for (hipEvent_t e : P->e_enc) (void)hipEventDestroy(e);
    delete P;
}

static int submit_job(mp3s_pipe *P, std::unique_ptr<Job> j, int64_t *ticket)
{
    {
        std::lock_guard<std::mutex> g(P->mu);
        int slot = -1;
        for (int k = 0; k < P->depth; k++) if (!P->slots[(size_t)k].busy) { slot = k; break; }
        if (slot < 0) return fail(MP3S_E_BUSY, "all %d slots are taken: collect a result first", P->depth);
        P->slots[(size_t)slot].busy = true;
        j->slot = slot; j->ticket = P->next_ticket++;
        if (ticket) *ticket = j->ticket;
        P->todo[(size_t)slot % P->todo.size()].push_back(j.get());
        P->inflight.push_back(std::move(j));
        P->st.submitted++;
    }
    P->cv_work.notify_all();
    return MP3S_OK;
}

int mp3s_pipe_submit(mp3s_pipe *P, const uint8_t *const *mp3s, const size_t *lens, int n_files, const uint8_t *const *msgs,
                     const size_t *msg_lens, int64_t *ticket)
{
    if (!P || P->internal || !mp3s || !lens || n_files <= 0 || (msgs && !msg_lens)) return fail(MP3S_E_ARG, "bad argument");
    std::unique_ptr<Job> j(new Job());
    j->files.resize((size_t)n_files); j->msgs.assign((size_t)n_files, {nullptr, 0});
    j->clear_all = msgs == nullptr;
    for (int i = 0; i < n_files; i++) {
        j->files[i] = {mp3s[i], lens[i]};
        if (msgs) {
            if (!msgs[i] && msg_lens[i]) return fail(MP3S_E_ARG, "file %d: null message of non-zero length", i);
            j->msgs[i] = {msgs[i], msg_lens[i]};
        }
    }
    return submit_job(P, std::move(j), ticket);
}

int mp3s_pipe_submit_decode(mp3s_pipe *P, const uint8_t *const *mp3s, const size_t *lens, int n_files, int64_t *ticket)
{
    if (!P || P->internal || !mp3s || !lens || n_files <= 0) return fail(MP3S_E_ARG, "bad argument");
    std::unique_ptr<Job> j(new Job());
    j->decode = true; j->clear_all = true;
    j->files.resize((size_t)n_files); j->msgs.assign((size_t)n_files, {nullptr, 0});
    for (int i = 0; i < n_files; i++) j->files[i] = {mp3s[i], lens[i]};
    return submit_job(P, std::move(j), ticket);
}

// the job's results are on the host: are they final?  *resolved: the host had to resolve the chains (on the job's own buffers)
static bool finish_fast(mp3s_pipe *P, Job *j, Slot &s, bool *resolved)
{
    const int32_t *small = (const int32_t *)j->res->big[2].data();
    const bool parse_ok = !j->walked || (small[4] & kParseMismatch) == 0;
    bool fast_ok = parse_ok && (j->decode ? small[3] == 0 : (small[0] == 0 && small[1] == 0 && small[2] == 0 && small[3] == 0));
    *resolved = false;
    if (!fast_ok) {
        if (trace_on()) fprintf(stderr, "mp3s: pipe job %lld: verdict %d units to redo, step range %d, packer %d, Huffman status 0x%x, parse status 0x%x\n",
                                (long long)j->ticket, small[0], small[1], small[2], small[3], small[4]);
        // only the cursor / address guesses failed (a long message, a start the input's tables did not predict):
        // the host resolves the chains on the job's own device buffers -- scan, decode and transforms stand
        if (P->s_tail) (void)hipStreamSynchronize(P->s_tail);   // (later jobs' tails: the resolve below packs through the same context)
        if (parse_ok && !j->decode && small[0] != 0 && small[1] == 0 && small[3] == 0) {
            int passes = 0;
            *resolved = enc_resolve(P->c, j->L, j->segs, s.h_stage + j->o_encblk, j->dev, j->res.get(), false, &passes) == MP3S_OK;
            if (hipStreamSynchronize(P->c->stream) != hipSuccess) *resolved = false;
        }
    }
    return fast_ok;
}

int mp3s_pipe_collect(mp3s_pipe *P, int64_t *ticket, mp3s_buf **owner, mp3s_file *out, int32_t *status, int max_files, int *n_files)
{
    if (!P || P->internal || !owner || !out || !status) return fail(MP3S_E_ARG, "null pointer");
    Job *j = nullptr;
    {
        std::unique_lock<std::mutex> g(P->mu);
        if (P->inflight.empty()) return fail(MP3S_E_BUSY, "nothing in flight");
        j = P->inflight.front().get();
        if (ticket) *ticket = j->ticket;   // (also when the call fails: the caller learns which job it is stuck on)
        if (n_files) *n_files = (int)j->files.size();
        if ((int)j->files.size() > max_files) return fail(MP3S_E_ARG, "the next job has %zu files, room for %d", j->files.size(), max_files);
        if (j->block) return fail(MP3S_E_ARG, "the next job is a block job: mp3s_pipe_collect_block");
        P->cv_done.wait(g, [&] { return j->state != Job::QUEUED; });
    }
    Slot &s = P->slots[(size_t)j->slot];
    const int nf = (int)j->files.size();
    int rc = MP3S_OK;
    bool fast_ok = false, resolved = false;
    double span_ms = -1;
    if (j->state == Job::ISSUED) {
        (void)hipSetDevice(P->c->device);
        if (hipEventSynchronize(s.e_down) != hipSuccess) rc = fail(MP3S_E_HIP, "waiting for the job's results failed");
        else {
            float ms = 0;
            if (hipEventElapsedTime(&ms, s.e_start, s.e_down) == hipSuccess) span_ms = ms;
            std::unique_lock<std::mutex> gi(P->mu_issue, std::defer_lock);
            const int32_t *small = (const int32_t *)j->res->big[2].data();
            if (!(small[0] == 0 && small[1] == 0 && small[2] == 0 && small[3] == 0 && (small[4] & kParseMismatch) == 0)) gi.lock();
            fast_ok = finish_fast(P, j, s, &resolved);
            // damaged Huffman data, a quantiser step out of range, a failed resolve: the synchronous path decides, file by file
            if (!fast_ok && !resolved) { if (!gi.owns_lock()) gi.lock(); run_slow(P, *j); }
        }
    }
    if (!rc) {
        if (fast_ok && j->decode) {
            if (j->walked) {
                // the stego bits: a serial pass over the table-index words the device left
                const uint64_t *tsel = (const uint64_t *)j->res->big[1].data();
                std::vector<uint8_t> &bits = j->res->bits;
                bits.clear();
                for (auto &d : j->dec) {
                    uint8_t carry[4] = {0, 0, 0, 0};
                    d.bits_off = bits.size();
                    stego_bits_from_tsel(tsel + d.first, d.n_frames, d.nch, carry, bits);
                    d.n_bits = bits.size() - d.bits_off;
                }
            }
            for (int i = 0; i < nf; i++) {
                const Job::DecFile &d = j->dec[(size_t)i];
                std::memset(&out[i], 0, sizeof out[i]);
                out[i].data = j->res->mp3 + d.wav_off; out[i].len = 44 + (size_t)d.n_frames * 1152 * d.nch * 2;
                out[i].kbps = d.bit_rate / 1000; out[i].sampling_rate = d.rate; out[i].channels = d.nch; out[i].n_frames = d.n_frames;
                out[i].n_bits = (int32_t)d.n_bits; out[i].bits = j->res->bits.data() + d.bits_off;
                status[i] = MP3S_OK;
            }
            *owner = j->res.release();
        } else if (fast_ok || resolved) {
            const mp3s_chain_seg_out *so = (const mp3s_chain_seg_out *)(j->res->big[2].data() + kSmallHead);
            for (int i = 0; i < nf; i++) {
                const EncSeg &sg = j->segs[(size_t)i];
                std::memset(&out[i], 0, sizeof out[i]);
                out[i].data = j->res->mp3 + sg.mp3_off; out[i].len = sg.mp3_len;
                out[i].kbps = j->kbps; out[i].sampling_rate = j->rate; out[i].channels = 2; out[i].n_frames = sg.n_frames;
                out[i].hide_offset = resolved ? sg.hide_offset : so[i].cursor - sg.hide_base;
                out[i].too_long = out[i].hide_offset < (int64_t)sg.n_hide - 1 ? 1 : 0;
                status[i] = MP3S_OK;
            }
            *owner = j->res.release();
        } else {
            rc = j->slow_rc;
            if (rc) fail(rc, "%s", j->slow_err.c_str());
            for (int i = 0; i < nf; i++) { out[i] = j->slow_out[(size_t)i]; status[i] = j->slow_st[(size_t)i]; }
            if (!rc) {
                for (int i = 0; i < nf; i++) if (status[i]) { fail(status[i], "%s", j->slow_err.c_str()); break; }
                *owner = j->slow_owner; j->slow_owner = nullptr;
            }
        }
    }
    {
        std::lock_guard<std::mutex> g(P->mu);
        P->st.collected++;
        if (span_ms >= 0) P->st.last_device_span_ms = span_ms;
        if (fast_ok) P->st.fast++; else if (resolved) P->st.resolved++; else P->st.slow++;
        s.busy = false;
        P->inflight.pop_front();
    }
    return rc;
}

int mp3s_pipe_submit_block(mp3s_pipe *P, const uint8_t *mp3, size_t len, const uint8_t *utf8, size_t n_msg, int rank, int world,
                           const mp3s_carry *carry_in, int64_t *ticket)
{
    if (!P || P->internal || !mp3 || world <= 0 || rank < 0 || rank >= world || (rank == 0 && carry_in) || (rank > 0 && !carry_in) || (!utf8 && n_msg))
        return fail(MP3S_E_ARG, "bad argument (rank 0 has no carry, every other rank has one)");
    std::unique_ptr<Job> j(new Job());
    j->block = true; j->rank = rank; j->world = world;
    j->files.assign(1, {mp3, len}); j->msgs.assign(1, {utf8, n_msg});
    j->clear_all = utf8 == nullptr;
    if (carry_in) { j->has_carry = true; j->carry = *carry_in; }
    return submit_job(P, std::move(j), ticket);
}

int mp3s_pipe_next_is_block(mp3s_pipe *P)
{
    if (!P) return fail(MP3S_E_ARG, "null pointer");
    std::lock_guard<std::mutex> g(P->mu);
    if (P->inflight.empty()) return fail(MP3S_E_BUSY, "nothing in flight");
    return P->inflight.front()->block ? 1 : 0;
}

int mp3s_pipe_collect_block(mp3s_pipe *P, int64_t *ticket, mp3s_buf **owner, mp3s_block *out)
{
    if (!P || P->internal || !owner || !out) return fail(MP3S_E_ARG, "null pointer");
    Job *j = nullptr;
    {
        std::unique_lock<std::mutex> g(P->mu);
        if (P->inflight.empty()) return fail(MP3S_E_BUSY, "nothing in flight");
        j = P->inflight.front().get();
        if (ticket) *ticket = j->ticket;
        if (!j->block) return fail(MP3S_E_ARG, "the next job is not a block job: mp3s_pipe_collect");
        P->cv_done.wait(g, [&] { return j->state != Job::QUEUED; });
    }
    Slot &s = P->slots[(size_t)j->slot];
    int rc = MP3S_OK;
    bool fast_ok = false, resolved = false;
    if (j->state == Job::ISSUED) {
        (void)hipSetDevice(P->c->device);
        if (hipEventSynchronize(s.e_down) != hipSuccess) rc = fail(MP3S_E_HIP, "waiting for the job's results failed");
        else {
            std::lock_guard<std::mutex> gi(P->mu_issue);
            fast_ok = finish_fast(P, j, s, &resolved);
            const int32_t *small = (const int32_t *)j->res->big[2].data();
            // (a share of a stream that inherits scalefactors across frames cannot look back on the device)
            if ((fast_ok || resolved) && (small[4] & kParseInherits) && j->world > 1) fast_ok = resolved = false;
            if (!fast_ok && !resolved) run_slow(P, *j);
        }
    }
    if (!rc) {
        if (fast_ok || resolved) {
            const EncSeg &sg = j->segs[0];
            const mp3s_chain_seg_out *so = (const mp3s_chain_seg_out *)(j->res->big[2].data() + kSmallHead);
            if (resolved) { j->blk.carry_out = sg.carry_out; j->blk.carry_used = sg.carry_used ? 1 : 0; }
            else {
                j->blk.carry_out.cursor = so[0].cursor - sg.hide_base;
                std::memcpy(j->blk.carry_out.chain, so[0].chain, sizeof j->blk.carry_out.chain);
                j->blk.carry_used = so[0].carry_used != 0;
            }
            j->blk.file.data = j->res->mp3 + sg.mp3_off; j->blk.file.len = sg.mp3_len; j->blk.file.n_frames = (int32_t)j->blk.n_frames;
            j->blk.file.hide_offset = j->blk.carry_out.cursor;
            j->blk.file.too_long = j->blk.file.hide_offset < (int64_t)sg.n_hide - 1 ? 1 : 0;
            *out = j->blk;
            *owner = j->res.release();
        } else {
            rc = j->slow_rc;
            if (rc) fail(rc, "%s", j->slow_err.c_str());
            else { *out = j->blk; *owner = j->blk_owner; j->blk_owner = nullptr; }
        }
    }
    {
        std::lock_guard<std::mutex> g(P->mu);
        P->st.collected++;
        if (fast_ok) P->st.fast++; else if (resolved) P->st.resolved++; else P->st.slow++;
        s.busy = false;
        P->inflight.pop_front();
    }
    return rc;
}

int mp3s_pipe_get_stats(mp3s_pipe *P, mp3s_pipe_stats *out)
{
    if (!P || !out) return fail(MP3S_E_ARG, "null pointer");
    std::lock_guard<std::mutex> g(P->mu);
    *out = P->st;
    return MP3S_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------ one file as chunks
// The reference's call shape is one file per call (steganography.py:137-162: decode loop MP3_Parser.py:68-80, then encode
// loop MP3_Encoder.py:607-609).  run_file gives that call the overlap the pipe gives a stream of jobs: the calling thread
// walks the frame headers a chunk at a time and queues each chunk on the stages above -- walk(k+1) || upload || front end ||
// kernels(k) || download(k-1) -- and the chunks' bytes land side by side in ONE result block.  What crosses a chunk
// boundary is what crosses a block boundary of a sharded stream: a frame of decoder state and a frame of
// PCM in front of the chunk are recomputed and dropped; the padding recurrence is replayed from the frame index; the message
// cursor and the inherited addresses (17 integers) are GUESSED -- "the message is hidden, nothing is inherited" -- and every
// chunk reports whether it looked at them: only a chunk that did, on a guess that was wrong, is run again on the real carry.
namespace {

bool same_effect(const mp3s_carry &a, const mp3s_carry &b, int64_t n_hide)
{
    return std::memcmp(a.chain, b.chain, sizeof a.chain) == 0 && std::min<int64_t>(a.cursor, n_hide) == std::min<int64_t>(b.cursor, n_hide);
}

struct RunChunk {
    std::unique_ptr<Job> job;
    long first = 0, count = 0;
    bool last = false;
    mp3s_carry guess = {}, out = {};
    bool carry_used = false;
    int64_t out_off = 0, out_len = 0;
    bool done = false;
};

}  // namespace

int ensure_own_pipe(mp3s_ctx *c, size_t chunk_bytes)
{
    if (c->own_pipe && c->own_pipe->max_job_bytes >= chunk_bytes) return MP3S_OK;
    if (c->own_pipe) { mp3s_pipe_destroy(c->own_pipe); c->own_pipe = nullptr; }
    const size_t want = std::max<size_t>(chunk_bytes + chunk_bytes / 4, (size_t)1 << 20);
    return pipe_create(c, kRunDepth, want, 0, true, &c->own_pipe);
}

void destroy_own_pipe(mp3s_ctx *c)   // (the context is being destroyed)
{
    if (c->own_pipe) { mp3s_pipe_destroy(c->own_pipe); c->own_pipe = nullptr; }
    forget_lanes(c);
}

constexpr int kRunWhole = 2;             // run_file_impl: the stream inherits scalefactors across frames -- once more, as one piece
constexpr long kWholeFrames = 4 * kDecodeChunk;   // ... if it is not longer than this (the slots are sized for the piece)

static int run_file_impl(mp3s_ctx *c, const uint8_t *mp3, size_t len, int mode, const uint8_t *utf8, size_t n_msg, int out_format, mp3s_buf **owner, RunResult *out, bool whole);

int run_file(mp3s_ctx *c, const uint8_t *mp3, size_t len, int mode, const uint8_t *utf8, size_t n_msg, int out_format, mp3s_buf **owner, RunResult *out)
{
    int rc = run_file_impl(c, mp3, len, mode, utf8, n_msg, out_format, owner, out, false);
    // A granule of a mixed block (or one behind a short granule 0 with scfsi set) reads scalefactors written many frames earlier
    // (SURVEY D10); the Huffman kernel finds them by walking back through the stream's side records, which a chunk that
    // starts in the middle of the stream cannot.  Such a file goes through the same stages in ONE piece (the transforms still
    // in groups): 1.7 instead of 4.4 ms for 10 000 frames, which the synchronous path spends in the host parser.
    if (rc == kRunWhole) rc = run_file_impl(c, mp3, len, mode, utf8, n_msg, out_format, owner, out, true);
    return rc == kRunWhole ? kRunFallback : rc;
}

static int run_file_impl(mp3s_ctx *c, const uint8_t *mp3, size_t len, int mode, const uint8_t *utf8, size_t n_msg, int out_format, mp3s_buf **owner, RunResult *out, bool whole)
{
    if (!c->opt[MP3S_OPT_FILE_PIPELINE] || !c->opt[MP3S_OPT_DEVICE_PARSE]) return kRunFallback;
    const bool decode = mode == kRunDecode;
    const double t_call0 = trace_on() ? now_ms() : 0;
    FrameWalker w;
    // (errors and empty streams: the synchronous path words them)
    if (len < 8 || len > 0xffff0000ull || w.open(mp3, len) || w.ended || w.hd.version != 1 || w.hd.layer != 3 || w.frame_size <= 0) {
        c->run_stats.fallbacks++;
        return kRunFallback;
    }
    const long fs0 = w.frame_size;
    const long n_est = (long)((len - (size_t)w.offset) / (size_t)std::max<long>(fs0 - 1, 24)) + 8;
    std::vector<uint8_t> bits;
    if (mode == kRunHide) {
        message_frame(utf8, n_msg, bits);
        if (bits.size() > 0x3fffff00) { c->run_stats.fallbacks++; return kRunFallback; }
    }
    const int64_t n_hide = (int64_t)bits.size();
    // ---- the chunk plan.  Every chunk costs a dozen launches and their gaps, so few chunks; the first one small, so that
    //      the device starts early; a message's reach inside the first chunk, where the cursor is decided (not guessed)
    const long reach_frames = n_hide ? (long)((n_hide * 5 / 14 + 32) / 4 * 9 / 8 + 64) : 0;
    long chunk = (long)c->opt[MP3S_OPT_CHUNK_FRAMES];
    long first_chunk;
    const long kMaxChunk = kDecodeChunk - 2;
    if (chunk > 0) { chunk = std::min(chunk, kMaxChunk); first_chunk = chunk; }
    else if (n_est <= 3000) { chunk = first_chunk = std::min(kMaxChunk, n_est + 16); }                    // one chunk: nothing to overlap with
    else {
        // A short first chunk, so that the device starts early, then chunks as long as a transform group takes: every chunk
        // costs the host 0.15 ms of walking, laying out and queueing (two dozen runtime calls), which four chunks of a
        // 10 000-frame file do not win back (tools/chunk_plan_probe.py: 2 048 + the rest 1.41 ms, four chunks 1.49, one 1.57;
        // a 100 000-frame file: 8 192 + 16 000s 8.1 ms)
        first_chunk = std::min<long>(8192, std::max<long>(2048, n_est / 12));
        chunk = kMaxChunk;
    }
    if (c->opt[MP3S_OPT_FIRST_CHUNK_FRAMES] > 0) first_chunk = (long)std::min<int64_t>(c->opt[MP3S_OPT_FIRST_CHUNK_FRAMES], kMaxChunk);
    first_chunk = std::min(kMaxChunk, std::max(first_chunk, reach_frames));
    if (whole) {
        if (n_est > kWholeFrames) { c->run_stats.fallbacks++; return kRunFallback; }
        first_chunk = chunk = n_est + 64;          // one piece
    }
    const long cap_frames = std::max(chunk, first_chunk) + 2;
    // (a message that reaches further than a chunk: the synchronous path's plan over the whole file)
    if (reach_frames > kMaxChunk || ensure_own_pipe(c, (size_t)cap_frames * (size_t)(fs0 + 2) + 4096)) { c->run_stats.fallbacks++; return kRunFallback; }
    mp3s_pipe *P = c->own_pipe;
    HIPCHK(hipSetDevice(c->device));
    // (for the duration of the call the context computes on the stream its pipe rehearsed best with, if that is not its own)
    struct StreamSwap {
        mp3s_ctx *c; hipStream_t keep;
        StreamSwap(mp3s_ctx *c_, hipStream_t s) : c(c_), keep(c_->stream) { if (s) c->stream = s; }
        ~StreamSwap() { if (c->stream != keep) { (void)hipStreamSynchronize(c->stream); c->stream = keep; } }
    } swap(c, P->s_comp);
    // the file's bytes set off for the device now (the first chunk's first), the walk follows
    struct FileUpGuard { mp3s_pipe *P; ~FileUpGuard() { file_up_end(P); } } up_guard{P};
    file_up_begin(P, mp3, len, (size_t)w.offset + (size_t)std::min<long>(first_chunk, n_est) * (size_t)(fs0 + 1) + 2048);
    // ---- the stream's frame table, grown as the walk proceeds
    std::vector<FrameRef> &refs = c->h_refs;
    if ((long)refs.size() < n_est + 64) refs.resize((size_t)n_est + 64);
    std::vector<uint8_t> &tables = c->h_tables;
    if (n_hide) { w.tables_wanted = (long)n_hide + (long)n_hide / 16 + 64; tables.resize((size_t)first_chunk * 4 + 16); }
    long n_walked = 0;
    WalkOut wv;                                  // the walker's state behind the chunk in hand
    std::unique_ptr<mp3s_buf> res(new mp3s_buf());
    std::vector<RunChunk> chunks;
    uint8_t fix[kPlaceEntry];
    bool have_fix = false;
    int rate = 0, kbps = 0, nch = 0;
    const size_t esz = pcm_elem(out_format);
    size_t res_cap = 0;
    auto fallback = [&](const char *why, int code = kRunFallback) {
        if (trace_on()) fprintf(stderr, "mp3s: run_file: %s -> %s\n", why, code == kRunWhole ? "once more, in one piece" : "synchronous path");
        if (code == kRunFallback) c->run_stats.fallbacks++;
        sync_all(P);
        for (auto &s : P->slots) s.busy = false;
        P->keep_slot[0] = P->keep_slot[1] = -1;
        return code;
    };
    // retire chunk k: wait for its results, settle its verdict and its carry
    int64_t hide_offset = 0;
    auto retire = [&](size_t k) -> int {
        RunChunk &rc = chunks[k];
        if (rc.done) return MP3S_OK;
        Job *j = rc.job.get();
        Slot &s = P->slots[(size_t)j->slot];
        if (issue_down(P, *j, s)) return kRunFallback;
        if (hipEventSynchronize(s.e_down) != hipSuccess) return fail(MP3S_E_HIP, "waiting for a chunk's results failed");
        if (trace_on()) {
            float up = 0, huff = 0, rate = 0, comp = 0, down = 0;
            hipEvent_t e0 = P->slots[(size_t)chunks[0].job->slot].e_start;
            if (k >= (size_t)P->depth) e0 = s.e_start;
            (void)hipEventElapsedTime(&up, e0, s.e_up); (void)hipEventElapsedTime(&huff, e0, s.e_huff); (void)hipEventElapsedTime(&rate, e0, s.e_rate);
            (void)hipEventElapsedTime(&comp, e0, s.e_comp); (void)hipEventElapsedTime(&down, e0, s.e_down);
            (void)hipGetLastError();
            fprintf(stderr, "mp3s: run_file: chunk %zu's results are here %.3f ms after the call's start; on the device, from the first chunk's start: inputs up %.3f, "
                            "front end done %.3f, rate loop done %.3f, tail done %.3f, results down %.3f ms\n", k, now_ms() - t_call0, up, huff, rate, comp, down);
        }
        bool resolved = false;
        const bool ok = finish_fast(P, j, s, &resolved);
        if (!ok && !resolved) return kRunFallback;
        if (resolved) c->run_stats.resolved++;
        const int32_t *small = (const int32_t *)j->res->big[2].data();
        if (j->walked && (small[4] & kParseInherits) && chunks.size() + (wv.ended ? 0 : 1) > 1) return kRunWhole;   // scalefactors inherited across frames: the stream in one piece
        if (!decode) {
            EncSeg &sg = j->segs[0];
            if (resolved) {
                std::memcpy(j->ck.dst, j->res->mp3 + sg.mp3_off, sg.mp3_len);
                rc.out = sg.carry_out; rc.carry_used = sg.carry_used;
            } else {
                const mp3s_chain_seg_out *so = (const mp3s_chain_seg_out *)(j->res->big[2].data() + kSmallHead);
                rc.out.cursor = so[0].cursor - sg.hide_base;
                std::memcpy(rc.out.chain, so[0].chain, sizeof rc.out.chain);
                rc.carry_used = so[0].carry_used != 0;
            }
            rc.out_len = (int64_t)sg.mp3_len;
        }
        rc.done = true;
        return MP3S_OK;
    };
    // issue frames [first, first + count) of the stream as a chunk on slot `slot`
    auto issue = [&](size_t k, const mp3s_carry *carry) -> int {
        RunChunk &rc = chunks[k];
        rc.job.reset(new Job());
        Job &j = *rc.job;
        j.slot = (int)(k % (size_t)P->depth);
        j.ticket = (int64_t)k;
        Slot &s = P->slots[(size_t)j.slot];
        Chunk &ck = j.ck;
        ck.on = true; ck.decode = decode; ck.refs = refs.data(); ck.first = rc.first; ck.count = rc.count; ck.last = rc.last;
        ck.lead = !decode && rc.first > 0 ? 1 : 0;
        ck.halo = rc.first - ck.lead > 0 ? 1 : 0;
        ck.w0 = rc.first - ck.lead - ck.halo; ck.n_win = rc.count + ck.lead + ck.halo;
        ck.out_format = out_format; ck.file = mp3; ck.file_len = len; ck.rate = rate; ck.kbps = kbps; ck.nch = nch;
        const uint32_t lo = refs[(size_t)ck.w0].file_off;
        ck.image_lo = rc.first == 0 ? 0 : (lo > kImageLead ? lo - kImageLead : 0);
        const FrameRef &lr = refs[(size_t)(rc.first + rc.count - 1)];
        ck.image_hi = (uint32_t)std::min<uint64_t>(len, (uint64_t)lr.file_off + lr.frame_size + 64);
        ck.fix = rc.last && have_fix ? fix : nullptr;
        ck.hide = bits.data(); ck.n_hide = (int)n_hide;
        ck.has_carry = rc.first > 0;
        if (ck.has_carry) ck.carry_in = carry ? *carry : rc.guess;
        ck.tables = rc.first == 0 && n_hide ? tables.data() : nullptr;
        ck.n_tables = (int)std::min<long>(wv.tables_frames, rc.count) * 4;
        ck.any_silent = wv.any_silent ? 1 : 0;   // (of the frames walked so far: at worst the re-run launches are issued without need)
        if (decode) ck.dst = res->big[0].data() + 64 + (size_t)rc.first * 1152 * (size_t)nch * esz;
        // the front end first (parse and Huffman kernels are a latency chain of 0.1 ms whatever the chunk's size), the encoder's
        // inputs are laid out while it runs
        if (!prepare_chunk(P, j, s, wv.max_p23)) return kRunFallback;
        const double t_i = trace_on() ? now_ms() : 0;
        if (issue_front(P, j, s, 0, wv.max_p23, true)) return kRunFallback;
        // (the results of the chunk in front come down behind this chunk's inputs, not in front of them)
        if (k > 0 && chunks[k - 1].job && issue_down(P, *chunks[k - 1].job, P->slots[(size_t)chunks[k - 1].job->slot])) return kRunFallback;
        if (!prepare_chunk_encode(P, j, s)) { sync_all(P); return kRunFallback; }
        if (!decode) {
            rc.out_off = j.L.bytes_before;
            if ((size_t)rc.out_off + j.L.mp3_bytes > res_cap) { sync_all(P); return kRunFallback; }
            j.ck.dst = res->big[0].data() + rc.out_off;
        }
        const int e = issue_back(P, j, s, true, true);
        if (trace_on()) fprintf(stderr, "mp3s:   front + inputs + back %.3f ms\n", now_ms() - t_i);
        return e ? kRunFallback : MP3S_OK;
    };
    // ---- walk and issue, chunk after chunk
    long want = first_chunk;
    // (a helper thread that walks the chunks behind the first while this one queues was tried: it wakes up later than the walk takes --
    // 1.56 instead of 1.38 ms per 10 000-frame file)
    for (size_t k = 0; !wv.ended; k++) {
        const double t_walk0 = trace_on() ? now_ms() : 0;
        long got = 0;
        const long room = (long)refs.size() - n_walked - 8;
        if (room <= 0) return fallback("more frames than the file's first frame size promised");
        want = std::min(want, room);
        uint8_t *tb = k == 0 && n_hide ? tables.data() : nullptr;
        while (got < want && !w.ended && !w.irregular) got += w.next(refs.data() + n_walked + got, want - got, tb ? tb + (size_t)got * 4 : nullptr, 0, 0);
        wv.got = got; wv.ended = w.ended; wv.irregular = w.irregular || got <= 0; wv.dup_last = w.dup_last; wv.any_silent = w.any_silent;
        wv.nch = w.nch; wv.sampling_rate = w.sampling_rate; wv.bit_rate = w.bit_rate; wv.max_p23 = w.max_p23; wv.tables_frames = w.tables_frames;
        if (wv.ended && !wv.irregular && !wv.dup_last) {
            bool alone = false;
            wv.have_fix = w.decode_last(reinterpret_cast<int16_t *>(fix + 16), reinterpret_cast<mp3s_granule_si *>(fix + 16 + 4608), &alone) == 0 && alone;
            std::memset(fix, 0, 16);
        }
        if (wv.irregular || got <= 0) return fallback("the walk does not take this stream");
        if (k == 0) {
            nch = wv.nch; rate = wv.sampling_rate;
            if (nch < 1 || nch > 2) return fallback("channel count");
            if (!decode && reencode_params(wv.sampling_rate, wv.bit_rate, wv.nch, got, 0, &kbps)) return fallback("not a stream the encoder takes");
            // the result block: the frames the file can hold at its first frame's size
            res_cap = decode ? 64 + (size_t)(n_est + 64) * 1152 * (size_t)nch * esz : (size_t)(n_est + 64) * (size_t)(fs0 + 2);
            if (!res->big[0].reserve(res_cap)) return fallback("no memory for the result");
        } else if (wv.nch != nch) return fallback("channel count changes");
        if (wv.ended) {
            if (wv.dup_last) return fallback("a repeated last frame");
            have_fix = wv.have_fix;
        }
        const double t_walk1 = trace_on() ? now_ms() : 0;
        if (k >= (size_t)P->depth) {               // the slot's previous chunk first
            const int r = retire(k - (size_t)P->depth);
            if (r) return r == kRunFallback || r == kRunWhole ? fallback("a chunk needs another path", r) : r;
        }
        const double t_ret = trace_on() ? now_ms() : 0;
        chunks.emplace_back();
        RunChunk &rc = chunks.back();
        rc.first = n_walked; rc.count = got; rc.last = wv.ended;
        rc.guess.cursor = MP3S_NO_CURSOR;          // "the message is hidden, nothing is inherited"
        n_walked += got;
        if (decode && 64 + (size_t)n_walked * 1152 * (size_t)nch * esz > res_cap) return fallback("more frames than the result block holds");
        const int r = issue(k, nullptr);
        if (trace_on()) fprintf(stderr, "mp3s: run_file chunk %zu (%ld frames): walk %.3f ms, wait for the slot %.3f ms, prepare + issue %.3f ms\n", k, got, t_walk1 - t_walk0, t_ret - t_walk1, now_ms() - t_ret);
        if (r) return r == kRunFallback ? fallback("a chunk does not fit the stages") : r;
        want = chunk;
    }
    if (trace_on()) fprintf(stderr, "mp3s: run_file: all chunks queued %.3f ms after the call's start\n", now_ms() - t_call0);
    // ---- settle the chunks in order: the carries
    if (!decode && (wv.sampling_rate != rate || wv.bit_rate / 1000 != kbps)) return fallback("the last header names another rate");
    // (first everything that needs a chunk's device buffers -- its verdict, a resolve -- then the carries: a chunk that is run
    // again takes a slot, and with it the buffers of the chunk that had it last)
    for (size_t k = 0; k < chunks.size(); k++) {
        const int r = retire(k);
        if (r) return r == kRunFallback || r == kRunWhole ? fallback("a chunk needs another path", r) : r;
    }
    mp3s_carry real = {};
    for (size_t k = 0; k < chunks.size(); k++) {
        int r = MP3S_OK;
        RunChunk &rc = chunks[k];
        if (decode) continue;
        if (k > 0) {
            const bool live = std::min<int64_t>(real.cursor, n_hide) < n_hide;     // the message is still being hidden at this boundary
            if (!same_effect(real, rc.guess, n_hide) && (rc.carry_used || live)) {
                // the chunk looked at its carry and the guess was wrong: once more, on the real one (everything behind it has been issued
                // and stays as it is unless its own carry turns out wrong in turn)
                if (trace_on()) fprintf(stderr, "mp3s: run_file: chunk %zu depends on its carry: again\n", k);
                sync_all(P);
                c->run_stats.reruns++;
                rc.done = false;
                r = issue(k, &real);
                if (!r) r = retire(k);
                if (r) return r == kRunFallback || r == kRunWhole ? fallback("a chunk needs another path", r) : r;
                sync_all(P);
            } else {
                // nothing in the chunk looked at the carry: every chain entry it hands on is its own; only the count of tables
                // seen so far moves with the real cursor
                rc.out.cursor = real.cursor + (rc.out.cursor - rc.guess.cursor);
            }
        }
        real = rc.out;
        hide_offset = real.cursor;
    }
    // ---- the result
    std::memset(out, 0, sizeof *out);
    out->n_frames = n_walked; out->nch = nch; out->sampling_rate = wv.sampling_rate; out->bit_rate = wv.bit_rate;
    if (decode) {
        // stego bits: the serial pass over the table-index words of all chunks (their halo frames left out)
        uint8_t carry[4] = {0, 0, 0, 0};
        for (auto &rc : chunks) {
            const Job &j = *rc.job;
            stego_bits_from_tsel((const uint64_t *)j.res->big[1].data() + j.ck.halo, rc.count, nch, carry, res->bits);
        }
        out->pcm = res->big[0].data() + 64; out->n_rows = (int64_t)n_walked * 1152;
        out->bits = res->bits.data(); out->n_bits = res->bits.size();
    } else {
        const RunChunk &lc = chunks.back();
        out->mp3 = res->big[0].data(); out->mp3_len = (size_t)(lc.out_off + lc.out_len);
        out->kbps = kbps;
        out->hide_offset = hide_offset;
        out->too_long = hide_offset < n_hide - 1 ? 1 : 0;
    }
    for (auto &s : P->slots) s.busy = false;
    c->run_stats.files++; c->run_stats.chunks += (int64_t)chunks.size();
    *owner = res.release();
    if (trace_on()) fprintf(stderr, "mp3s: run_file: done %.3f ms after the call's start\n", now_ms() - t_call0);
    return MP3S_OK;
}
