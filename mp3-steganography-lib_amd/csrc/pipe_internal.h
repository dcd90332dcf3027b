// Internal to the overlapped stages (pipe_jobs.cpp, pipe_lanes.cpp, file_up.cpp, run_file.cpp): the records of a slot, a
// chunk and a job, the pipe itself, and what the four files call of one another.  See pipe_jobs.cpp for the stages.
#pragma once
#include <sched.h>
#include <time.h>

#include <condition_variable>
#include <deque>

#include "mp3s_internal.h"

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
    hipEvent_t e_half = nullptr;         // the first of the bit packer's two launches is through (the last chunk of a one-file call)
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
    size_t down_split = 0;               // bytes of the result the first of the packer's two launches wrote (0: one launch, one copy)
    bool down_pending = false;           // the copies of its results are not queued yet (issue_down)
    uint32_t image_base = 0, md_base = 0;
    int grab_frames = 0;                 // the context's per-set buffers are asked for at this many frames at least (the chunks of one call: all alike)
    bool file_wide = false;              // side records and main data go to the file-wide arrays of FileUp (a chunk of a one-file call)
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
    // ... and what the chunks' parse kernels make of it, file-wide: side records (frame index) and gathered main data (md_off from the
    // file's start), so that a granule of chunk k that inherits scalefactors from a frame of chunk k - 3 finds them
    mp3s_frame_side *d_side = nullptr; size_t side_cap = 0 /* frames */;
    uint8_t *d_blob = nullptr; size_t blob_cap = 0;
    bool active = false;                 // the call in progress reads its file from d_file
};

struct WalkOut {                         // the walker's state behind a chunk of a one-file call
    long got = 0;
    bool ended = false, irregular = false, dup_last = false, any_silent = false, have_fix = false;
    int nch = 0, sampling_rate = 0, bit_rate = 0, max_p23 = 0;
    long tables_frames = 0;
};

// what choosing a pipe's streams took and found (mp3s_pipe_stats / mp3s_run_stats hand it out)
struct LaneReport { double rehearsal_ms = 0; int64_t rehearsals = 0, lanes = 0, queue_shared = 0; float serial_ms = 0, best_ms = 0; };

struct mp3s_pipe {
    mp3s_ctx *c = nullptr;
    int depth = 0;
    hipStream_t s_img = nullptr;         // the file pieces' own copy stream (the packed inputs of a chunk must not queue behind them)
    FileUp up;
    bool internal = false;               // the context's own (run_file): no worker threads, jobs issued by the caller
    size_t max_job_bytes = 0;
    size_t max_frames = 0;               // frames a job can have at most (the context's own pipe cuts its chunks by frames); 0: from the bytes, 96 per frame
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
    Job *pending_down = nullptr;         // (depth 1 - 2) the job issued last, whose copy down is queued behind the next job's issue or by its collector
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
    LaneReport lanes;                    // how its streams were chosen
};

// ---- pipe_jobs.cpp
int reencode_params(int sampling_rate, int bit_rate, int nch, long n_frames, int dup_last, int *kbps_out);
bool prepare_chunk(mp3s_pipe *P, Job &j, Slot &s, int max_p23);
bool prepare_chunk_encode(mp3s_pipe *P, Job &j, Slot &s);
int issue_front(mp3s_pipe *P, Job &j, Slot &s, size_t blob_len, int max_p23, bool inputs_later);
int issue_back(mp3s_pipe *P, Job &j, Slot &s, bool inputs_later, bool defer_down = false);
int issue_down(mp3s_pipe *P, Job &j, Slot &s);
void sync_all(mp3s_pipe *P);
void bind_to(const std::vector<int> &cpus);
bool finish_fast(mp3s_pipe *P, Job *j, Slot &s, bool *resolved);
int pipe_create(mp3s_ctx *c, int depth, size_t max_job_bytes, int scan_threads, bool internal, mp3s_pipe **out, size_t max_frames = 0);
// the slot's buffers exist (made on first need in the context's own pipe); false: out of memory
bool pipe_slot_ready(mp3s_pipe *P, Slot &s);
// ---- pipe_lanes.cpp
int pick_lanes(mp3s_ctx *c, hipStream_t *up, hipStream_t *down, hipStream_t *huff, hipStream_t *comp /* a stream to compute on instead of the context's, or null */,
               hipStream_t *tail /* a stream for the tail of a job, or null */, int want_tail /* 0: none, 1: always, 2: if the rehearsal is faster with it */,
               hipStream_t *dec = nullptr /* a stream for the decode transforms, or null */, int want_dec = 1,
               hipStream_t *img = nullptr /* a second copy-up stream (the file pieces of a one-file call), or null */, LaneReport *report = nullptr);
void forget_lanes(mp3s_ctx *c);
// ---- file_up.cpp
constexpr size_t kFileOnDevice = (size_t)1 << 30;       // longer files: chunk by chunk through the slots' own image buffers
constexpr size_t kFilePiece = (size_t)4 << 20;
bool file_up_begin(mp3s_pipe *P, const uint8_t *file, size_t len, size_t first_bytes, long n_est);
int file_up_wait(mp3s_pipe *P, size_t need, hipStream_t stream);
void file_up_end(mp3s_pipe *P);
