// ------------------------------------------------------------------------------------------------ one file as chunks
// The reference's call shape is one file per call (steganography.py:137-162: decode loop MP3_Parser.py:68-80, then encode
// loop MP3_Encoder.py:607-609).  run_file gives that call the overlap the pipe gives a stream of jobs: the calling thread
// walks the frame headers a chunk at a time and queues each chunk on the stages above -- walk(k+1) || upload || front end ||
// kernels(k) || download(k-1) -- and the chunks' bytes land side by side in ONE result block.  What crosses a chunk
// boundary is what crosses a block boundary of a sharded stream: a frame of decoder state and a frame of
// PCM in front of the chunk are recomputed and dropped; the padding recurrence is replayed from the frame index; the message
// cursor and the inherited addresses (17 integers) are GUESSED -- "the message is hidden, nothing is inherited" -- and every
// chunk reports whether it looked at them: only a chunk that did, on a guess that was wrong, is run again on the real carry.
#include <unistd.h>
#include "pipe_internal.h"

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

int ensure_own_pipe(mp3s_ctx *c, size_t chunk_bytes, size_t chunk_frames)
{
    if (c->own_pipe && c->own_pipe->max_job_bytes >= chunk_bytes && c->own_pipe->max_frames >= chunk_frames) return MP3S_OK;
    if (c->own_pipe) {
        // A file with larger frames than any before it (a higher bit rate): the slots' byte buffers are too small.  The old pipe is PARKED, not
        // destroyed: handing its page-locked and device buffers back to the driver at this point made every copy of the next ten calls run at
        // half rate (r06, tools/repro_config5.py + rocprofv3 --memory-copy-trace: 28 instead of 56 GB/s down, the same sizes, agents and stream;
        // kept instead: the first call 9 instead of 20 ms, none slow behind it -- config 5's joint_ms_short_48k_192 at 1.6 or 2.5 ms by where the
        // seven timed calls fell).  Frame sizes are bounded (1 441 bytes), so at most a few pipes are ever parked; they go with the context.
        pipe_quiesce(c->own_pipe);
        c->parked_pipes.push_back(c->own_pipe);
        c->own_pipe = nullptr;
    }
    const size_t want = std::max<size_t>(chunk_bytes + chunk_bytes / 4, (size_t)1 << 20);
    // (chunks are cut by frames: the slots' per-frame arrays are sized by what a chunk can hold, not by the 96-byte frames its bytes could be)
    return pipe_create(c, kRunDepth, want, 0, true, &c->own_pipe, std::max<size_t>(chunk_frames + chunk_frames / 4, 4096));
}

// how the streams of the context's own pipe were chosen (mp3s_ctx_run_stats)
void own_pipe_lanes(const mp3s_ctx *c, mp3s_run_stats *out)
{
    if (!c->own_pipe) return;
    const LaneReport &r = c->own_pipe->lanes;
    out->rehearsal_us = (int64_t)(r.rehearsal_ms * 1e3); out->rehearsals = r.rehearsals; out->lanes = r.lanes; out->queue_shared = r.queue_shared;
}

void destroy_own_pipe(mp3s_ctx *c)   // (the context is being destroyed)
{
    if (c->own_pipe) { mp3s_pipe_destroy(c->own_pipe); c->own_pipe = nullptr; }
    for (mp3s_pipe *p : c->parked_pipes) mp3s_pipe_destroy(p);
    c->parked_pipes.clear();
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
    c->sink_done = 0;                                                // (mp3s_*_fd: nothing of THIS attempt's result is in the file yet)
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
        // 10 000-frame file do not win back (round 3, profiles/r03a_chunk_plans.json: 2 048 + the rest 1.41 ms, four chunks 1.49, one 1.57;
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
    if (reach_frames > kMaxChunk || ensure_own_pipe(c, (size_t)cap_frames * (size_t)(fs0 + 2) + 4096, (size_t)cap_frames)) { c->run_stats.fallbacks++; return kRunFallback; }
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
    file_up_begin(P, mp3, len, (size_t)w.offset + (size_t)std::min<long>(first_chunk, n_est) * (size_t)(fs0 + 1) + 2048, n_est);
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
    // EVERY way out of the call after the first chunk has been queued goes through here (another path takes the file, or a hard
    // error such as MP3S_E_HIP): nothing of this call may still be in flight when its result block, its jobs and the slots are
    // let go -- a queued download would otherwise write into page-locked memory the pool hands to the next call
    auto fallback = [&](const char *why, int code = kRunFallback) {
        if (trace_on()) fprintf(stderr, "mp3s: run_file: %s -> %s\n", why, code == kRunWhole ? "once more, in one piece" : code == kRunFallback ? "synchronous path" : "error");
        if (code == kRunFallback) c->run_stats.fallbacks++;
        sync_all(P);
        if (P->s_img) (void)hipStreamSynchronize(P->s_img);
        (void)hipGetLastError();
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
        // scalefactors inherited across frames: the Huffman kernel walks back to the granule that wrote them.  With the
        // file's side records and main data in file-wide arrays it finds them across chunk boundaries (round 4); without them
        // (a file above 1 GB, no memory) the stream goes through the stages once more, in one piece
        if (j->walked && !j->file_wide && (small[4] & kParseInherits) && chunks.size() + (wv.ended ? 0 : 1) > 1) return kRunWhole;
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
    // (mp3s_*_fd: a chunk's bytes go to the file as soon as they are final -- while the chunks behind it are still on the device.  Final =
    // the chunk will not be run again on its real carry: the test of the settling loop at the end, on a copy of the carries, nothing
    // settled here.  Chunks 0 .. k have been retired.)
    // Only into a file that was EMPTY when the call began (sink_early): a stream that is refused late -- a damaged frame in the last chunk, found
    // on the device -- ends in an error, and the reference has not touched its output file at that point; a file that held something is written
    // at the end of a call that succeeded (the entry point does that), a fresh one is removed by the caller that created it.
    bool sink_on = c->sink_fd >= 0 && c->sink_early && (!decode || out_format == MP3S_PCM_I16);   // (a decode's chunks have no carries: final when they are down)
    size_t sink_next = 0;
    mp3s_carry sink_real = {};
    auto sink_upto = [&](size_t k) {
        for (; sink_on && sink_next <= k && sink_next < chunks.size(); sink_next++) {
            const RunChunk &rc = chunks[sink_next];
            mp3s_carry mine = rc.out;
            if (!decode && sink_next > 0) {
                const bool live = std::min<int64_t>(sink_real.cursor, n_hide) < n_hide;
                if (!same_effect(sink_real, rc.guess, n_hide) && (rc.carry_used || live)) { sink_on = false; break; }   // (it will be run again: its bytes change)
                mine.cursor = sink_real.cursor + (rc.out.cursor - rc.guess.cursor);
            }
            const size_t off = decode ? (size_t)rc.first * 1152 * (size_t)nch * esz : (size_t)rc.out_off;
            size_t left = decode ? (size_t)rc.count * 1152 * (size_t)nch * esz : (size_t)rc.out_len, at = off;
            if (off != c->sink_done) { sink_on = false; break; }
            const uint8_t *src = res->big[0].data() + (decode ? 64 : 0) + off;
            while (left) {
                const ssize_t w = pwrite(c->sink_fd, src, left, (off_t)(c->sink_base + at));
                if (w <= 0) { sink_on = false; break; }                     // (the caller writes what is missing, and reports what fails there)
                src += w; at += (size_t)w; left -= (size_t)w;
            }
            if (!sink_on) break;
            c->sink_done = at;
            sink_real = mine;
        }
    };
    // issue frames [first, first + count) of the stream as a chunk on slot `slot`
    // A chunk is queued in two steps.  issue_a: its front end (file piece, parse, Huffman -- a latency chain of 0.1 ms whatever the
    // chunk's size).  issue_b: the results of the chunk in front, this chunk's encoder inputs (laid out here, 0.1 ms of host time
    // for 8 000 frames) and everything behind the front end.  The main loop runs issue_a(k + 1) BEFORE issue_b(k) (round 4): the next
    // chunk's front end is then on the device while this chunk's kernels still run -- queued behind issue_b(k) it started 0.1 ms
    // later, ended behind the chunk's rate loop, and the compute stream sat idle for 0.05 ms between two chunks.
    auto issue_a = [&](size_t k, const mp3s_carry *carry) -> int {
        RunChunk &rc = chunks[k];
        rc.job.reset(new Job());
        Job &j = *rc.job;
        j.slot = (int)(k % (size_t)P->depth);
        j.ticket = (int64_t)k;
        // (the context's buffers the chunks take turns with -- Huffman outputs, PCM -- are asked for at the size of the call's largest
        // chunk: a chunk that asked for more than the one before it in the same set would have the buffer replaced while that
        // chunk's results still wait in it to be copied out, now that their copy is queued behind the next front end)
        j.grab_frames = (int)cap_frames;
        Slot &s = P->slots[(size_t)j.slot];
        if (!pipe_slot_ready(P, s)) return kRunFallback;
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
        if (!prepare_chunk(P, j, s, wv.max_p23)) return kRunFallback;
        if (issue_front(P, j, s, 0, wv.max_p23, true)) return kRunFallback;
        if (c->opt[MP3S_OPT_FAIL_CHUNK] == (int64_t)k + 1) {      // (test aid: a hard error with this chunk's front end and the chunks in front of it in flight)
            c->opt[MP3S_OPT_FAIL_CHUNK] = 0;
            return fail(MP3S_E_HIP, "chunk %zu: failure injected by MP3S_OPT_FAIL_CHUNK", k);
        }
        return MP3S_OK;
    };
    auto issue_b = [&](size_t k) -> int {
        RunChunk &rc = chunks[k];
        Job &j = *rc.job;
        Slot &s = P->slots[(size_t)j.slot];
        const double t_i = trace_on() ? now_ms() : 0;
        // (the results of the chunk in front come down behind this chunk's inputs, not in front of them)
        if (k > 0 && chunks[k - 1].job && issue_down(P, *chunks[k - 1].job, P->slots[(size_t)chunks[k - 1].job->slot])) return kRunFallback;
        if (!prepare_chunk_encode(P, j, s)) { sync_all(P); return kRunFallback; }
        if (!decode) {
            rc.out_off = j.L.bytes_before;
            if ((size_t)rc.out_off + j.L.mp3_bytes > res_cap) { sync_all(P); return kRunFallback; }
            j.ck.dst = res->big[0].data() + rc.out_off;
        }
        const int e = issue_back(P, j, s, true, true);
        if (trace_on()) fprintf(stderr, "mp3s:   inputs + back of chunk %zu %.3f ms\n", k, now_ms() - t_i);
        return e ? kRunFallback : MP3S_OK;
    };
    auto issue = [&](size_t k, const mp3s_carry *carry) -> int {
        const int r = issue_a(k, carry);
        return r ? r : issue_b(k);
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
            if (r) return fallback("a chunk needs another path", r);
            sink_upto(k - (size_t)P->depth);
        }
        const double t_ret = trace_on() ? now_ms() : 0;
        chunks.emplace_back();
        RunChunk &rc = chunks.back();
        rc.first = n_walked; rc.count = got; rc.last = wv.ended;
        rc.guess.cursor = MP3S_NO_CURSOR;          // "the message is hidden, nothing is inherited"
        n_walked += got;
        if (decode && 64 + (size_t)n_walked * 1152 * (size_t)nch * esz > res_cap) return fallback("more frames than the result block holds");
        int r = issue_a(k, nullptr);
        if (!r && k > 0) r = issue_b(k - 1);
        if (trace_on()) fprintf(stderr, "mp3s: run_file chunk %zu (%ld frames): walk %.3f ms, wait for the slot %.3f ms, prepare + issue %.3f ms\n", k, got, t_walk1 - t_walk0, t_ret - t_walk1, now_ms() - t_ret);
        if (r) return fallback("a chunk does not fit the stages", r);
        want = chunk;
    }
    {
        const int r = issue_b(chunks.size() - 1);
        if (r) return fallback("a chunk does not fit the stages", r);
    }
    if (trace_on()) fprintf(stderr, "mp3s: run_file: all chunks queued %.3f ms after the call's start\n", now_ms() - t_call0);
    // ---- settle the chunks in order: the carries
    if (!decode && (wv.sampling_rate != rate || wv.bit_rate / 1000 != kbps)) return fallback("the last header names another rate");
    // (first everything that needs a chunk's device buffers -- its verdict, a resolve -- then the carries: a chunk that is run
    // again takes a slot, and with it the buffers of the chunk that had it last)
    for (size_t k = 0; k < chunks.size(); k++) {
        const int r = retire(k);
        if (r) return fallback("a chunk needs another path", r);
        sink_upto(k);
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
                if (r) return fallback("a chunk needs another path", r);
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
