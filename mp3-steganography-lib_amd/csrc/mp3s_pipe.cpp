// C-ABI of the library (include/mp3s.h), part 4: the asynchronous host-fed pipeline.
//
// The reference runs one file at a time through two serial frame loops (decoder/MP3_Parser.py:68-80,
// encoder/MP3_Encoder.py:607-609, glued by steganography.py:153-159).  Here a job (one file, or a list of files that
// form one device batch) passes through four stages that overlap with those of the jobs around it:
//
//   scan      a host worker thread walks the frames (sync, header, side info, reservoir) and writes main data and side
//             records straight into the job slot's page-locked staging -- no intermediate vectors, no second copy
//   upload    hipMemcpyAsync on the copy-up stream: staging -> the slot's device inputs
//   compute   the context's stream: Huffman decode -> decode transforms -> encode transforms -> rate loop -> chain check ->
//             bit packing; the scratch between the kernels is shared by all slots (one stream = one job at a time)
//   download  hipMemcpyAsync on the copy-down stream: MP3 bytes + the 16-byte verdict -> a page-locked result block
//
// Events order the three streams; the host waits for nothing until the caller collects a result.  A job whose streams
// the device cannot take alone (mono, mixed blocks, a repeated last frame, unsupported rates, staging too small), or
// whose verdict says the cursor guess failed or the Huffman data is damaged, is redone by the synchronous path
// (mp3s_hide_messages) -- same bytes, by construction of that path; the fast path is an optimisation, never a
// different answer.
#include <sched.h>
#include <time.h>

#include <condition_variable>
#include <deque>

#include "mp3s_internal.h"

namespace {

constexpr int kMaxFastFiles = 1024;

struct Slot {
    uint8_t *h_stage = nullptr;          // page-locked: [blob | side records | input block]
    uint8_t *d_stage = nullptr;          // the same layout on the device
    size_t blob_cap = 0, side_cap = 0 /* frames */, in_cap = 0, fix_cap = 0 /* entries */, o_side = 0, o_in = 0, o_fix = 0, stage_bytes = 0;
    uint8_t *d_mp3 = nullptr; size_t mp3_cap = 0;
    int32_t *d_small = nullptr;
    // the encoder's intermediates of the slot's job (mdct, quantised lines, GrInfo, energies, scfsi): the slot's own, so
    // that a job whose cursor guess failed is resolved on them at collect time while later jobs have long been issued
    uint8_t *d_enc = nullptr; size_t enc_cap = 0;
    hipEvent_t e_start = nullptr, e_up = nullptr, e_huff = nullptr, e_rate = nullptr, e_comp = nullptr, e_down = nullptr;
    bool busy = false;
};

struct Job {
    int64_t ticket = 0;
    int slot = -1;
    std::vector<std::pair<const uint8_t *, size_t>> files, msgs;   // borrowed until the job is collected
    bool clear_all = false;
    bool decode = false;                 // MP3 -> WAV (int16) instead of hide / clear
    enum State { QUEUED, ISSUED, SLOW_DONE } state = QUEUED;
    // fast path
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
    struct DecFile { int n_frames, nch, rate, bit_rate; size_t wav_off, bits_off, n_bits; };
    std::vector<DecFile> dec;
    std::vector<uint8_t> res_bits;
    int nch = 2, n_total = 0, keep_set = 0;
};

}  // namespace

struct mp3s_pipe {
    mp3s_ctx *c = nullptr;
    int depth = 0;
    std::vector<Slot> slots;
    hipStream_t s_up = nullptr, s_down = nullptr;
    // The Huffman kernel is a latency chain that leaves the vector units mostly idle; the rate loop is bound by them.  The
    // front end of job k+1 therefore runs on a stream of its own, under the encode half of job k, with two sets of
    // Huffman outputs (is / side records) taken in turn; e_dec[x] = the decode transforms that read set x last are done.
    hipStream_t s_huff = nullptr;
    // ... and, optionally (MP3S_PIPE_TAIL=1; measured slower here, see mp3s_pipe_create), the tail of a job (chain check +
    // bit packing) on another one, under the decode transforms of the next job; e_rate orders it behind the job's rate loop
    hipStream_t s_tail = nullptr;
    int last_tail = -1;                  // slot of the job whose tail was issued last
    hipEvent_t e_dec[2] = {nullptr, nullptr};
    bool dec_used[2] = {false, false};
    unsigned issued = 0;
    // the int16 PCM of a batch lives in one of two device buffers taken in turn; a decode job downloads from it while the
    // next job computes: keep_slot[x] = slot of the job whose download reads buffer x last (-1: none)
    int keep_slot[2] = {-1, -1};
    std::mutex mu;                       // queue, job states, slots
    std::condition_variable cv_work, cv_done;
    std::mutex mu_issue;                 // everything that touches the context (its stream, pool, profiler)
    std::deque<std::unique_ptr<Job>> inflight;   // ticket order; front = next to collect
    // One queue per worker, and a slot always goes to the same worker (slot % workers): the staging of a slot stays in
    // the cache hierarchy of the core that wrote it last, and a scan from another core complex would fetch every line it
    // overwrites from there (measured: 0.75 ms per 10 000 frames on the slot's own worker, 2.5 ms on changing ones).
    std::vector<std::deque<Job *>> todo;
    std::vector<std::thread> workers;
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

int reencode_params(const ParsedStream &p, int *kbps_out)
{
    const int kbps = p.bit_rate / 1000;
    int sri, bri, whole;
    if (p.sampling_rate != 32000 && p.sampling_rate != 44100 && p.sampling_rate != 48000) return 1;
    if (kbps <= 0 || stream_params(p.sampling_rate, kbps, &sri, &bri, &whole)) return 1;
    if (p.nch != 2 || p.n_frames <= 0 || p.dup_last_frame) return 1;
    *kbps_out = kbps;
    return 0;
}

// scan the job's files into the slot's staging and lay out the encoder's inputs; false = this job takes the synchronous path
bool prepare_fast(mp3s_pipe *P, Job &j, Slot &s, ParsedStream &p, size_t *blob_len, int *max_p23)
{
    const int nf = (int)j.files.size();
    if (nf > kMaxFastFiles) return false;
    mp3s_frame_side *side = (mp3s_frame_side *)(s.h_stage + s.o_side);
    uint8_t *in = s.h_stage + s.o_in;
    size_t base = 0;
    long n = 0;
    j.segs.assign((size_t)nf, EncSeg());
    j.bits.assign((size_t)nf, {});
    j.guess.assign((size_t)nf, {});
    j.n_fix = 0;
    j.dec.clear(); j.res_bits.clear();
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
            j.dec.push_back({p.n_frames, p.nch, p.sampling_rate, p.bit_rate, 0, j.res_bits.size(), p.bits.size()});
            j.res_bits.insert(j.res_bits.end(), p.bits.begin(), p.bits.end());
        } else {
            int kbps = 0;
            if (reencode_params(p, &kbps)) return false;
            if (i == 0) { j.rate = p.sampling_rate; j.kbps = kbps; }
            else if (p.sampling_rate != j.rate || kbps != j.kbps) return false;   // more than one device batch
        }
        for (int f = 0; f < p.n_frames; f++) {
            side[n + f].md_off += (uint32_t)base;
            side[n + f].reserved = (uint32_t)n;           // where the stream starts in the batch (scalefactor inheritance walks back to it)
            dechdr[n + f].stream_first = (uint32_t)n;
        }
        *max_p23 = std::max(*max_p23, max_part2_3(side + n, p.n_frames));
        if (k.gpu_ok) {   // (a frame of a stream with inherited scalefactors cannot be decoded on its own)
            // The last frame of a stream the reference's encoder wrote lacks the 0-3 bytes its writer drops (E14), and in one
            // file out of eight the Huffman data reaches into them: the device kernel would flag it.  One frame per stream is
            // cheap on the host, so it is decoded here, the kernel skips it, and its samples are placed behind the kernel.
            if ((size_t)j.n_fix >= s.fix_cap) return false;
            uint8_t *e = s.h_stage + s.o_fix + (size_t)j.n_fix * kPlaceEntry;
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
    if (j.decode) {
        // the result block: per file a WAV image, its PCM 64 bytes into an aligned region, the 44-byte header right in
        // front of it (what mp3s_decode_file hands out)
        size_t off = 0;
        for (auto &d : j.dec) {
            d.wav_off = off + 64 - 44;
            off += 64 + (((size_t)d.n_frames * 1152 * d.nch * 2 + 63) & ~(size_t)63);
        }
        j.res.reset(new mp3s_buf());
        if (!j.res->big[0].reserve(off) || !j.res->big[2].reserve(small_bytes(1))) return false;
        j.res->mp3 = j.res->big[0].data();
        for (const auto &d : j.dec) wav_header((int64_t)d.n_frames * 1152, d.nch, d.rate, j.res->mp3 + d.wav_off);
        j.res->bits = std::move(j.res_bits);
        return true;
    }
    const double tA = trace_on() ? now_ms() : 0;
    if (enc_layout(j.segs, j.rate, j.kbps, j.L)) return false;
    const size_t o_enc = ((size_t)n * sizeof(mp3s_frame_hdr) + 15) & ~(size_t)15;
    if (o_enc + j.L.bytes > s.in_cap) return false;
    if (enc_fill(j.segs, j.L, in + o_enc)) return false;
    if (j.L.mp3_bytes + 16 > s.mp3_cap) return false;
    const double tB = trace_on() ? now_ms() : 0;
    j.res.reset(new mp3s_buf());
    if (!j.res->big[0].reserve(j.L.mp3_bytes) || !j.res->big[2].reserve(small_bytes(j.L.n_segs))) return false;
    j.res->mp3 = j.res->big[0].data();
    if (trace_on()) fprintf(stderr, "mp3s:   job %lld: input layout %.3f ms, result blocks %.3f ms\n", (long long)j.ticket, tB - tA, now_ms() - tB);
    return true;
}

// everything a fast job does on the device, queued on the streams; nothing is waited for
int issue_fast(mp3s_pipe *P, Job &j, Slot &s, size_t blob_len, int max_p23)
{
    mp3s_ctx *c = P->c;
    const EncLayout &L = j.L;
    const int n = j.n_total, units = n * 4, nch = j.decode ? j.nch : 2;
    const size_t o_enc = ((size_t)n * sizeof(mp3s_frame_hdr) + 15) & ~(size_t)15, in_bytes = j.decode ? o_enc : o_enc + L.bytes;
    const int set = (int)(P->issued++ & 1u);
    void *d_is = c->grab(set ? 24 : 0, (size_t)n * 2304 * 2), *d_si = c->grab(set ? 25 : 1, (size_t)n * 4 * sizeof(mp3s_granule_si)),
         *d_keep = c->grab(set ? 26 : 7, (size_t)n * 2304 * 2);
    if (!d_is || !d_si || !d_keep) return fail(MP3S_E_NOMEM, "hipMalloc failed for a %d-frame job", n);
    uint8_t *d_blob = s.d_stage, *d_side = s.d_stage + s.o_side, *d_in = s.d_stage + s.o_in;
    HIPCHK(hipEventRecord(s.e_start, P->s_up));
    HIPCHK(hipMemcpyAsync(d_blob, s.h_stage, blob_len, hipMemcpyHostToDevice, P->s_up));
    HIPCHK(hipMemcpyAsync(d_side, s.h_stage + s.o_side, (size_t)n * sizeof(mp3s_frame_side), hipMemcpyHostToDevice, P->s_up));
    HIPCHK(hipMemcpyAsync(d_in, s.h_stage + s.o_in, in_bytes, hipMemcpyHostToDevice, P->s_up));
    HIPCHK(hipMemcpyAsync(s.d_stage + s.o_fix, s.h_stage + s.o_fix, (size_t)j.n_fix * kPlaceEntry, hipMemcpyHostToDevice, P->s_up));
    HIPCHK(hipEventRecord(s.e_up, P->s_up));
    HIPCHK(hipStreamWaitEvent(P->s_huff, s.e_up, 0));
    if (P->dec_used[set]) HIPCHK(hipStreamWaitEvent(P->s_huff, P->e_dec[set], 0));
    const int e = launch_huffman(P->s_huff, d_blob, (const mp3s_frame_side *)d_side, n, nch, max_p23, (int16_t *)d_is, (mp3s_granule_si *)d_si,
                                 s.d_small + 3, c->d_sync + 4, &c->prof, false);
    if (e) return fail(MP3S_E_HIP, "huffman launch: %s", hipGetErrorString((hipError_t)e));
    if (launch_place_frames(P->s_huff, s.d_stage + s.o_fix, j.n_fix, (int16_t *)d_is, (mp3s_granule_si *)d_si))
        return fail(MP3S_E_HIP, "placing the host-decoded frames failed");
    HIPCHK(hipEventRecord(s.e_huff, P->s_huff));
    HIPCHK(hipStreamWaitEvent(c->stream, s.e_huff, 0));
    // the PCM buffer of this set may still be read by the download of the decode job that used it last
    if (P->keep_slot[set] >= 0) HIPCHK(hipStreamWaitEvent(c->stream, P->slots[(size_t)P->keep_slot[set]].e_down, 0));
    P->keep_slot[set] = j.decode ? j.slot : -1;
    const mp3s_frame_hdr *d_dechdr = (const mp3s_frame_hdr *)d_in;
    const mp3s_frame_hdr *h_dechdr = (const mp3s_frame_hdr *)(s.h_stage + s.o_in);
    const size_t frame_elems = (size_t)1152 * nch;
    for (long start = 0; start < n; start += kDecodeChunk) {
        const int halo = (start && h_dechdr[start].stream_first < (uint32_t)start) ? 1 : 0;
        const int cnt = (int)std::min<long>(kDecodeChunk, n - start) + halo;
        const int rc = decode_transform_chunk(c, (const int16_t *)d_is, (const mp3s_granule_si *)d_si, d_dechdr, start - halo, cnt, nch, halo,
                                              MP3S_PCM_I16, (int16_t *)d_keep + (size_t)start * frame_elems);
        if (rc) return rc;
    }
    HIPCHK(hipEventRecord(P->e_dec[set], c->stream));
    P->dec_used[set] = true;
    if (j.decode) {
        HIPCHK(hipEventRecord(s.e_comp, c->stream));
        HIPCHK(hipStreamWaitEvent(P->s_down, s.e_comp, 0));
        HIPCHK(hipMemcpyAsync(j.res->big[2].data(), s.d_small, kSmallHead, hipMemcpyDeviceToHost, P->s_down));
        size_t first = 0;
        for (const auto &d : j.dec) {
            const size_t bytes = (size_t)d.n_frames * frame_elems * 2;
            HIPCHK(hipMemcpyAsync(j.res->mp3 + d.wav_off + 44, (const int16_t *)d_keep + first * frame_elems, bytes, hipMemcpyDeviceToHost, P->s_down));
            first += (size_t)d.n_frames;
        }
        HIPCHK(hipEventRecord(s.e_down, P->s_down));
        return MP3S_OK;
    }
    const size_t b_mdct = (size_t)n * 2304 * 4, b_ix = (size_t)n * 2304 * 2, b_out = (size_t)units * sizeof(mp3s_gr_out),
                 b_en = ((size_t)units * 22 * 4 + 255) & ~(size_t)255, b_sc = (size_t)n * 8 * 4;
    const size_t need = b_mdct + b_ix + b_out + b_en + b_sc;
    if (need > s.enc_cap) {   // (hipFree waits for the device; only while the slot is growing to its job size)
        if (s.d_enc) (void)hipFree(s.d_enc);
        s.d_enc = nullptr; s.enc_cap = 0;
        if (hipMalloc((void **)&s.d_enc, need + need / 8) != hipSuccess) return fail(MP3S_E_NOMEM, "hipMalloc failed for a %d-frame job", n);
        s.enc_cap = need + need / 8;
    }
    void *d_agg = c->grab(15, chain_agg_bytes(n));
    if (!d_agg) return fail(MP3S_E_NOMEM, "hipMalloc failed for a %d-frame job", n);
    EncDev &dev = j.dev;
    dev.d_pcm = (const int16_t *)d_keep; dev.d_in = d_in + o_enc; dev.d_mdct_all = (int32_t *)s.d_enc; dev.d_ix = (int16_t *)(s.d_enc + b_mdct);
    dev.d_out = (mp3s_gr_out *)(s.d_enc + b_mdct + b_ix); dev.d_en = (int32_t *)(s.d_enc + b_mdct + b_ix + b_out); dev.d_agg = d_agg;
    dev.d_mp3 = s.d_mp3; dev.d_sc = (int32_t *)(s.d_enc + b_mdct + b_ix + b_out + b_en);
    dev.d_small = s.d_small;
    if (!enc_variant_buffers(c, L, dev)) return fail(MP3S_E_NOMEM, "hipMalloc failed for %d variant entries", L.n_entries);
    const int rc = enc_issue(c, L, dev, P->s_tail, s.e_rate, P->s_tail && P->last_tail >= 0 && getenv("MP3S_PIPE_TAIL_THROTTLE") ? P->slots[(size_t)P->last_tail].e_comp : nullptr);
    if (rc) return rc;
    HIPCHK(hipEventRecord(s.e_comp, P->s_tail ? P->s_tail : c->stream));
    P->last_tail = (int)(&s - P->slots.data());
    HIPCHK(hipStreamWaitEvent(P->s_down, s.e_comp, 0));
    const size_t total = j.segs.back().mp3_off + j.segs.back().mp3_len;
    HIPCHK(hipMemcpyAsync(j.res->big[2].data(), s.d_small, small_bytes(L.n_segs), hipMemcpyDeviceToHost, P->s_down));
    if (total) HIPCHK(hipMemcpyAsync(j.res->mp3, s.d_mp3, total, hipMemcpyDeviceToHost, P->s_down));
    HIPCHK(hipEventRecord(s.e_down, P->s_down));
    return MP3S_OK;
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
    (void)hipStreamSynchronize(P->s_down);   // ... and the PCM buffer a download may still be reading
    if (P->s_tail) (void)hipStreamSynchronize(P->s_tail);   // ... and the context's chain scratch and packer words
    j.slow_out.assign((size_t)nf, mp3s_file());
    j.slow_st.assign((size_t)nf, 0);
    j.res.reset();
    if (j.decode) {
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
        return;
    }
    j.slow_rc = mp3s_hide_messages(P->c, fp.data(), fl.data(), nf, j.clear_all ? nullptr : mp.data(), ml.data(), &j.slow_owner, j.slow_out.data(),
                                   j.slow_st.data());
    j.slow_err = mp3s_last_error();
}

void worker(mp3s_pipe *P, int me)
{
    (void)hipSetDevice(P->c->device);
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
        bool fast = prepare_fast(P, *j, s, scratch, &blob_len, &max_p23);
        const double t1 = now_ms(), c1 = thread_cpu_ms();
        if (trace_on()) fprintf(stderr, "mp3s: pipe job %lld slot %d on cpu %d: scan + layout %.3f ms (cpu %.3f)%s\n", (long long)j->ticket, j->slot, sched_getcpu(), t1 - t0, c1 - c0, fast ? "" : " -> synchronous path");
        Job::State st;
        {
            std::lock_guard<std::mutex> gi(P->mu_issue);
            if (fast && issue_fast(P, *j, s, blob_len, max_p23) != MP3S_OK) {
                (void)hipStreamSynchronize(P->s_up); (void)hipStreamSynchronize(P->s_huff); (void)hipStreamSynchronize(P->c->stream);
                if (P->s_tail) (void)hipStreamSynchronize(P->s_tail);
                (void)hipStreamSynchronize(P->s_down);
                fast = false;
            }
            if (!fast) run_slow(P, *j);
            st = fast ? Job::ISSUED : Job::SLOW_DONE;
        }
        const double t2 = now_ms();
        {
            std::lock_guard<std::mutex> g(P->mu);
            j->scan_ms = t1 - t0; j->issue_ms = t2 - t1;
            j->state = st;
            P->st.scan_ms += t1 - t0; P->st.issue_ms += t2 - t1; P->st.scan_cpu_ms += c1 - c0;
        }
        P->cv_done.notify_all();
    }
}

void free_slot(Slot &s)
{
    if (s.h_stage) (void)hipHostFree(s.h_stage);
    if (s.d_stage) (void)hipFree(s.d_stage);
    if (s.d_mp3) (void)hipFree(s.d_mp3);
    if (s.d_small) (void)hipFree(s.d_small);
    if (s.d_enc) (void)hipFree(s.d_enc);
    for (hipEvent_t e : {s.e_start, s.e_up, s.e_huff, s.e_rate, s.e_comp, s.e_down}) if (e) (void)hipEventDestroy(e);
    s = Slot();
}

}  // namespace

extern "C" {

int mp3s_pipe_create(mp3s_ctx *c, int depth, size_t max_job_bytes, int scan_threads, mp3s_pipe **out)
{
    if (!c || !out || depth < 1 || depth > 64 || scan_threads < 1 || scan_threads > 64 || max_job_bytes < 4096)
        return fail(MP3S_E_ARG, "bad argument (1 <= depth, scan_threads <= 64; max_job_bytes >= 4096)");
    *out = nullptr;
    HIPCHK(hipSetDevice(c->device));
    std::unique_ptr<mp3s_pipe> P(new mp3s_pipe());
    P->c = c; P->depth = depth;
    auto destroy = [&](int code, const char *what) {
        for (auto &s : P->slots) free_slot(s);
        if (P->s_up) (void)hipStreamDestroy(P->s_up);
        if (P->s_down) (void)hipStreamDestroy(P->s_down);
        if (P->s_huff) (void)hipStreamDestroy(P->s_huff);
        if (P->s_tail) (void)hipStreamDestroy(P->s_tail);
        for (hipEvent_t e : P->e_dec) if (e) (void)hipEventDestroy(e);
        return fail(code, "%s", what);
    };
    // The runtime multiplexes streams onto a few hardware queues (round robin, four by default), and work of two streams
    // that share a queue runs in order: a copy stream that lands on the compute stream's queue stops overlapping with the
    // kernels (measured: every second pipe of a process, 1.35 instead of 1.10 ms per 10 000-frame batch).  Streams of
    // another priority come from another set of queues, so the copy streams are created with the highest one.
    int prio_low = 0, prio_high = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
    if (hipStreamCreateWithPriority(&P->s_up, hipStreamNonBlocking, prio_high) != hipSuccess ||
        hipStreamCreateWithPriority(&P->s_down, hipStreamNonBlocking, prio_high) != hipSuccess)
        return destroy(MP3S_E_HIP, "stream creation failed");
    // the front-end stream takes the lowest priority: another set of queues again, and its workgroups fill in beside the
    // compute stream's instead of competing with them
    const char *hp = getenv("MP3S_PIPE_HUFF_PRIO");
    const int huff_prio = hp ? atoi(hp) : prio_low;
    if (hipStreamCreateWithPriority(&P->s_huff, hipStreamNonBlocking, huff_prio) != hipSuccess ||
        hipEventCreateWithFlags(&P->e_dec[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&P->e_dec[1], hipEventDisableTiming) != hipSuccess)
        return destroy(MP3S_E_HIP, "stream creation failed");
    // The tail stream is OFF unless MP3S_PIPE_TAIL=1.  On a resident batch fed through three contexts the arrangement is
    // worth 5 % (bench.py, region (i): 0.829 -> 0.787 ms per step); in this pipe it measured slower at every priority, with
    // and without making the next rate loop wait for the tail in front of it (0.89 - 1.25 against 0.83 ms per batch: with
    // the copy streams and the front end there are then five streams at work, and every kernel of the two overlapping jobs
    // stretches by half; tools/two_pipes_probe.py with MP3S_PIPE_TAIL / MP3S_PIPE_TAIL_PRIO / MP3S_PIPE_TAIL_THROTTLE).
    const char *tp = getenv("MP3S_PIPE_TAIL_PRIO");
    const int tail_prio = tp ? atoi(tp) : 0;
    if (getenv("MP3S_PIPE_TAIL") && atoi(getenv("MP3S_PIPE_TAIL")) == 1 &&
        hipStreamCreateWithPriority(&P->s_tail, hipStreamNonBlocking, tail_prio) != hipSuccess)
        return destroy(MP3S_E_HIP, "stream creation failed");
    P->slots.resize((size_t)depth);
    for (auto &s : P->slots) {
        // main data: the file minus headers plus alignment and 8 zero bytes per frame; frames: 96 bytes is the smallest
        // Layer III frame (32 kbit/s at 48 kHz); anything denser (false syncs) overflows the sink and takes the other path
        s.blob_cap = (max_job_bytes + max_job_bytes / 8 + 4096 + 15) & ~(size_t)15;
        s.side_cap = max_job_bytes / 96 + 16;
        s.in_cap = (s.side_cap * (72 + 16) + max_job_bytes / 4 + (size_t)kMaxFastFiles * (sizeof(mp3s_chain_seg) + sizeof(mp3s_select_span)) + s.side_cap * 4 * MP3S_SELECT_VARIANTS * 8 + 4096 + 15) & ~(size_t)15;   // (+ variant entries: at most 10 per unit, 8 bytes each)
        s.o_side = s.blob_cap;
        s.o_in = (s.o_side + s.side_cap * sizeof(mp3s_frame_side) + 15) & ~(size_t)15;
        s.fix_cap = std::min<size_t>(kMaxFastFiles, s.side_cap);
        s.o_fix = s.o_in + s.in_cap;
        s.stage_bytes = s.o_fix + s.fix_cap * kPlaceEntry;
        s.mp3_cap = max_job_bytes + s.side_cap + 4096;
        if (hipHostMalloc((void **)&s.h_stage, s.stage_bytes, hipHostMallocDefault) != hipSuccess || hipMalloc((void **)&s.d_stage, s.stage_bytes) != hipSuccess ||
            hipMalloc((void **)&s.d_mp3, s.mp3_cap) != hipSuccess || hipMalloc((void **)&s.d_small, small_bytes(kMaxFastFiles)) != hipSuccess ||
            // (only e_start and e_down are read as times; the ordering events carry no time stamps: 1 % per job)
            hipEventCreate(&s.e_start) != hipSuccess || hipEventCreateWithFlags(&s.e_up, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s.e_huff, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&s.e_comp, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&s.e_rate, hipEventDisableTiming) != hipSuccess ||
            // the collecting thread sleeps on this one instead of spinning: with one process per GPU on a shared host the
            // cores are needed by the scan workers (the wake-up latency disappears behind the jobs in flight)
            hipEventCreateWithFlags(&s.e_down, getenv("MP3S_PIPE_SPIN") ? hipEventDefault : hipEventBlockingSync) != hipSuccess)
            return destroy(MP3S_E_NOMEM, "slot allocation failed");
    }
    P->todo.resize((size_t)scan_threads);
    for (int t = 0; t < scan_threads; t++) P->workers.emplace_back(worker, P.get(), t);
    *out = P.release();
    return MP3S_OK;
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
    (void)hipSetDevice(P->c->device);
    (void)hipStreamSynchronize(P->s_up); (void)hipStreamSynchronize(P->s_huff); (void)hipStreamSynchronize(P->c->stream);
    if (P->s_tail) (void)hipStreamSynchronize(P->s_tail);
    (void)hipStreamSynchronize(P->s_down);
    for (auto &j : P->inflight) if (j->slow_owner) mp3s_buf_free(j->slow_owner);
    for (auto &s : P->slots) free_slot(s);
    (void)hipStreamDestroy(P->s_up); (void)hipStreamDestroy(P->s_down); (void)hipStreamDestroy(P->s_huff);
    if (P->s_tail) (void)hipStreamDestroy(P->s_tail);
    for (hipEvent_t e : P->e_dec) (void)hipEventDestroy(e);
    delete P;
}

int mp3s_pipe_submit(mp3s_pipe *P, const uint8_t *const *mp3s, const size_t *lens, int n_files, const uint8_t *const *msgs,
                     const size_t *msg_lens, int64_t *ticket)
{
    if (!P || !mp3s || !lens || n_files <= 0 || (msgs && !msg_lens)) return fail(MP3S_E_ARG, "bad argument");
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

int mp3s_pipe_submit_decode(mp3s_pipe *P, const uint8_t *const *mp3s, const size_t *lens, int n_files, int64_t *ticket)
{
    if (!P || !mp3s || !lens || n_files <= 0) return fail(MP3S_E_ARG, "bad argument");
    std::unique_ptr<Job> j(new Job());
    j->decode = true; j->clear_all = true;
    j->files.resize((size_t)n_files); j->msgs.assign((size_t)n_files, {nullptr, 0});
    for (int i = 0; i < n_files; i++) j->files[i] = {mp3s[i], lens[i]};
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

int mp3s_pipe_collect(mp3s_pipe *P, int64_t *ticket, mp3s_buf **owner, mp3s_file *out, int32_t *status, int max_files, int *n_files)
{
    if (!P || !owner || !out || !status) return fail(MP3S_E_ARG, "null pointer");
    Job *j = nullptr;
    {
        std::unique_lock<std::mutex> g(P->mu);
        if (P->inflight.empty()) return fail(MP3S_E_BUSY, "nothing in flight");
        j = P->inflight.front().get();
        if ((int)j->files.size() > max_files) return fail(MP3S_E_ARG, "the next job has %zu files, room for %d", j->files.size(), max_files);
        P->cv_done.wait(g, [&] { return j->state != Job::QUEUED; });
    }
    Slot &s = P->slots[(size_t)j->slot];
    const int nf = (int)j->files.size();
    int rc = MP3S_OK;
    bool fast_ok = false, resolved = false;
    if (j->state == Job::ISSUED) {
        (void)hipSetDevice(P->c->device);
        if (hipEventSynchronize(s.e_down) != hipSuccess) rc = fail(MP3S_E_HIP, "waiting for the job's results failed");
        else {
            const int32_t *small = (const int32_t *)j->res->big[2].data();
            fast_ok = j->decode ? small[3] == 0 : (small[0] == 0 && small[1] == 0 && small[2] == 0 && small[3] == 0);
            float ms = 0;
            if (hipEventElapsedTime(&ms, s.e_start, s.e_down) == hipSuccess) P->st.last_device_span_ms = ms;
            if (!fast_ok) {
                if (trace_on()) fprintf(stderr, "mp3s: pipe job %lld: verdict %d units to redo, step range %d, packer %d, Huffman status 0x%x\n",
                                        (long long)j->ticket, small[0], small[1], small[2], small[3]);
                std::lock_guard<std::mutex> gi(P->mu_issue);
                // only the cursor / address guesses failed (a long message, a start the input's tables did not predict):
                // the host resolves the chains on the job's own device buffers -- scan, decode and transforms stand
                if (P->s_tail) (void)hipStreamSynchronize(P->s_tail);   // (later jobs' tails: the resolve below packs through the same context)
                if (!j->decode && small[0] != 0 && small[1] == 0 && small[3] == 0) {
                    int passes = 0;
                    resolved = enc_resolve(P->c, j->L, j->segs, s.h_stage + s.o_in + (((size_t)j->n_total * sizeof(mp3s_frame_hdr) + 15) & ~(size_t)15),
                                           j->dev, j->res.get(), false, &passes) == MP3S_OK;
                    if (hipStreamSynchronize(P->c->stream) != hipSuccess) resolved = false;
                }
                // damaged Huffman data, a quantiser step out of range, a failed resolve: the synchronous path decides, file by file
                if (!resolved) run_slow(P, *j);
            }
        }
    }
    if (!rc) {
        if (fast_ok && j->decode) {
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
    if (ticket) *ticket = j->ticket;
    if (n_files) *n_files = nf;
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
