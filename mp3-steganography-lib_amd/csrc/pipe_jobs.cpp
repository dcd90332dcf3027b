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
#include "pipe_internal.h"

namespace {

double thread_cpu_ms()
{
    timespec ts;
    clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

}  // namespace

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

namespace {

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
        if (len >= kDirectUpload) {
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

}  // namespace

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
    const FrameRef &lastr = k.refs[k.w0 + k.n_win - 1];
    // the whole file is on the device: the chunk's side records and main data go where the file's frames have their places (FileUp),
    // and what a granule inherits from a frame of an earlier chunk is there for the Huffman kernel's walk back
    j.file_wide = on_device && P->up.d_side && (size_t)(k.w0 + k.n_win) <= P->up.side_cap && (size_t)lastr.md_off + lastr.md_len + 64 <= P->up.blob_cap;
    j.image_base = on_device ? 0 : k.image_lo; j.md_base = j.file_wide ? 0 : k.refs[k.w0].md_off;
    j.d_file = on_device ? P->up.d_file : nullptr; j.file_need = k.image_hi;
    if (!j.file_wide && (size_t)lastr.md_off - j.md_base + lastr.md_len + 64 > s.blob_cap) return false;
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
    if (j.file_wide) { sr->side_back[0] = (uint16_t)((uint32_t)k.w0 & 0xffffu); sr->side_back[1] = (uint16_t)((uint32_t)k.w0 >> 16); }
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

namespace {

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

}  // namespace

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
    const size_t ng = (size_t)std::max(n, j.grab_frames);
    void *d_is = c->grab(set ? 24 : 0, ng * 2304 * 2), *d_si = c->grab(set ? 25 : 1, ng * 4 * sizeof(mp3s_granule_si)),
         *d_keep = c->grab(set ? 26 : 7, ng * frame_elems * esz);
    if (!d_is || !d_si || !d_keep) return fail(MP3S_E_NOMEM, "hipMalloc failed for a %d-frame job", n);
    uint8_t *d_blob = s.d_stage, *d_side = s.d_stage + s.o_side;
    if (j.file_wide) { d_blob = P->up.d_blob; d_side = reinterpret_cast<uint8_t *>(P->up.d_side + ck.w0); }
    HIPCHK(hipEventRecord(s.e_start, P->s_up));
    if (j.walked) {
        for (const Upload &u : j.ups) HIPCHK(hipMemcpyAsync(s.d_image + u.dst, u.src, u.bytes, hipMemcpyHostToDevice, P->s_up));
        HIPCHK(hipMemcpyAsync(s.d_stage + j.o_small, s.h_stage + j.o_small, (inputs_later ? j.front_end : j.pack_end) - j.o_small, hipMemcpyHostToDevice, P->s_up));
    } else {
        const size_t o_enc = j.o_encblk - s.o_in, in_bytes = j.decode ? ((size_t)n * sizeof(mp3s_frame_hdr) + 15) & ~(size_t)15 : o_enc + L.bytes;
        HIPCHK(hipMemcpyAsync(d_blob, s.h_stage, blob_len, hipMemcpyHostToDevice, P->s_up));
        HIPCHK(hipMemcpyAsync(d_side, s.h_stage + s.o_side, (size_t)n * sizeof(mp3s_frame_side), hipMemcpyHostToDevice, P->s_up));
        HIPCHK(hipMemcpyAsync(s.d_stage + s.o_in, s.h_stage + s.o_in, in_bytes, hipMemcpyHostToDevice, P->s_up));
        HIPCHK(hipMemcpyAsync(s.d_stage + j.o_fix, s.h_stage + j.o_fix, (size_t)j.n_fix * kPlaceEntry, hipMemcpyHostToDevice, P->s_up));
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
                                 d_small + 3, j.walked ? nullptr : c->d_sync + 4, &c->prof, false, (int)c->opt[MP3S_OPT_HUF_LANES]);
    if (e) return fail(MP3S_E_HIP, "huffman launch: %s", hipGetErrorString((hipError_t)e));
    if (launch_place_frames(P->s_huff, s.d_stage + j.o_fix, j.n_fix, (int16_t *)d_is, (mp3s_granule_si *)d_si))
        return fail(MP3S_E_HIP, "placing the host-decoded frames failed");
    HIPCHK(hipEventRecord(s.e_huff, P->s_huff));
    if (trace_on()) fprintf(stderr, "mp3s:   front end queued %.3f ms after the job's start\n", now_ms() - t_issue0);
    return MP3S_OK;
}

// ... and everything behind it: decode transforms, encode side, tail, download (defer_down: the download is queued by issue_down
// later -- a copy that waits for its job's kernels holds up every copy queued behind it, in either direction, on engines the
// runtime shares between the copy streams: the inputs of the next chunk of a one-file call must not queue behind it)
int issue_back(mp3s_pipe *P, Job &j, Slot &s, bool inputs_later, bool defer_down)
{
    mp3s_ctx *c = P->c;
    const double t_issue0 = trace_on() ? now_ms() : 0;
    const EncLayout &L = j.L;
    const Chunk &ck = j.ck;
    const int n = j.n_total, nch = j.decode ? j.nch : 2;
    const int out_format = ck.on && j.decode ? ck.out_format : MP3S_PCM_I16;
    const size_t esz = pcm_elem(out_format), frame_elems = (size_t)1152 * nch;
    const int set = j.set;
    const size_t ng = (size_t)std::max(n, j.grab_frames);
    void *d_is = c->grab(set ? 24 : 0, ng * 2304 * 2), *d_si = c->grab(set ? 25 : 1, ng * 4 * sizeof(mp3s_granule_si)),
         *d_keep = c->grab(set ? 26 : 7, ng * frame_elems * esz);
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
        // (the last group's last dispatch signals e_dec itself where it can: a record is a packet of its own in the queue, 7-8 us of nothing
        //  between two kernels: tools/timeline.sh)
        const bool last_group = start + kDecodeChunk >= n;
        const int rc = decode_transform_chunk(c, (const int16_t *)d_is, (const mp3s_granule_si *)d_si, d_dechdr, start - halo, cnt, nch, halo,
                                              out_format, (uint8_t *)d_keep + (size_t)(start - halo0) * frame_elems * esz, ds,
                                              last_group && (c->opt[MP3S_OPT_PIPE_SIGNALS] & 1) ? P->e_dec[set] : nullptr);
        if (rc) return rc;
    }
    if (n <= halo0 || !(c->opt[MP3S_OPT_PIPE_SIGNALS] & 1)) HIPCHK(hipEventRecord(P->e_dec[set], ds));     // (no group ran: nothing has signalled it)
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
    // the last chunk of a one-file call: nothing behind it hides its tail and its copy down, so the packer runs as two launches and
    // the first half of the bytes comes down under the second (the frame offsets are the host's own: enc_fill wrote them)
    dev.pack_split = 0; dev.pack_half = nullptr; j.down_split = 0;
    if (ck.on && ck.last && j.walked && L.n >= 1024 && s.e_half && P->internal) {
        dev.pack_split = L.n / 2; dev.pack_half = s.e_half;
        j.down_split = reinterpret_cast<const uint32_t *>(s.h_stage + j.o_encblk + L.o_off)[dev.pack_split];
    }
    if (!enc_variant_buffers(c, L, dev)) return fail(MP3S_E_NOMEM, "hipMalloc failed for %d variant entries", L.n_entries);
    const int rc = enc_issue(c, L, dev, P->s_tail, s.e_rate, nullptr,
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
    size_t done_bytes = 0;
    if (!j.decode && j.down_split > 0) {      // the bytes of the packer's first launch: behind ITS event, not the job's last kernel
        const size_t total = j.segs.back().mp3_off + j.segs.back().mp3_len;
        done_bytes = std::min(j.down_split & ~(size_t)3, total);
        HIPCHK(hipStreamWaitEvent(P->s_down, s.e_half, 0));
        if (done_bytes) HIPCHK(hipMemcpyAsync((ck.on ? ck.dst : j.res->mp3), s.d_mp3, done_bytes, hipMemcpyDeviceToHost, P->s_down));
    }
    HIPCHK(hipStreamWaitEvent(P->s_down, s.e_comp, 0));
    if (j.decode) {
        void *d_keep = c->grab(j.set ? 26 : 7, (size_t)std::max(n, j.grab_frames) * frame_elems * esz);
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
        uint8_t *dst = ck.on ? ck.dst : j.res->mp3;
        if (total) HIPCHK(hipMemcpyAsync(dst + done_bytes, s.d_mp3 + done_bytes, total - done_bytes, hipMemcpyDeviceToHost, P->s_down));
    }
    HIPCHK(hipEventRecord(s.e_down, P->s_down));
    return MP3S_OK;
}

namespace {

// A pipe of depth 1 or 2 has no job queued far enough ahead to hide this: the copy of a job's results waits for the job's last
// kernel, and queued at once it holds up the copy of the NEXT job's inputs (queued behind it on another stream, on engines the
// runtime shares between the copy streams) until then -- the device sits idle for that job's front end.  There the copy down is
// queued behind the next job's issue instead, or by the collector if that comes first (as the one-file calls do since round 3;
// deeper pipes have their jobs' inputs up long before the job in front asks for its results).
int issue_fast(mp3s_pipe *P, Job &j, Slot &s, size_t blob_len, int max_p23)
{
    const bool defer = P->depth <= 2 && !P->internal;
    int rc = issue_front(P, j, s, blob_len, max_p23, false);
    if (!rc) rc = issue_back(P, j, s, false, defer);
    if (P->pending_down && P->pending_down != &j) {           // the job in front: its results may come down now
        Job *q = P->pending_down;
        P->pending_down = nullptr;
        const int rd = issue_down(P, *q, P->slots[(size_t)q->slot]);
        if (!rc) rc = rd;
    }
    if (!rc && defer && j.down_pending) P->pending_down = &j;
    return rc;
}

// the collector's half of the above: the job's copy down, if nobody has queued it yet (mu_issue taken here)
void down_now(mp3s_pipe *P, Job &j, Slot &s)
{
    if (P->depth > 2 || P->internal) return;
    std::lock_guard<std::mutex> gi(P->mu_issue);
    if (P->pending_down == &j) P->pending_down = nullptr;
    if (j.down_pending) (void)issue_down(P, j, s);
}

}  // namespace

void sync_all(mp3s_pipe *P)
{
    (void)hipStreamSynchronize(P->s_up); (void)hipStreamSynchronize(P->s_huff);
    if (P->s_dec) (void)hipStreamSynchronize(P->s_dec);
    (void)hipStreamSynchronize(P->c->stream);
    if (P->s_tail) (void)hipStreamSynchronize(P->s_tail);
    (void)hipStreamSynchronize(P->s_down);
}

namespace {

void run_slow(mp3s_pipe *P, Job &j)   // mu_issue held
{
    const int nf = (int)j.files.size();
    std::vector<const uint8_t *> fp(nf), mp(nf);
    std::vector<size_t> fl(nf), ml(nf);
    for (int i = 0; i < nf; i++) {
        fp[i] = j.files[i].first; fl[i] = j.files[i].second;
        mp[i] = j.clear_all ? nullptr : j.msgs[i].first; ml[i] = j.clear_all ? 0 : j.msgs[i].second;
    }
    // (a shallow pipe's last issued job may still wait for its copy down to be queued: the synchronous path below writes the PCM and
    // Huffman buffers it would read)
    if (P->pending_down && P->pending_down != &j) {
        Job *q = P->pending_down;
        P->pending_down = nullptr;
        (void)issue_down(P, *q, P->slots[(size_t)q->slot]);
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

}  // namespace

void bind_to(const std::vector<int> &cpus)
{
    if (cpus.empty()) return;
    cpu_set_t set;
    CPU_ZERO(&set);
    for (int cpu : cpus) if (cpu >= 0 && cpu < CPU_SETSIZE) CPU_SET(cpu, &set);
    (void)sched_setaffinity(0, sizeof set, &set);
}

namespace {

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
        bool cancelled = false;
        {
            // Jobs go to the device in the order they were submitted: the streams are queues and results are collected in
            // ticket order, so a job that overtakes the one in front of it makes that one's collector wait a whole job longer
            // (two workers on four slots: 1.36 instead of 0.82 ms per batch).  Walks and scans still run side by side.
            // (a pipe that is being destroyed drops its queued jobs: the ticket in front of this one may never be issued)
            std::unique_lock<std::mutex> g(P->mu);
            P->cv_turn.wait(g, [&] { return P->stop || P->next_issue == j->ticket; });
            cancelled = P->stop;
        }
        if (cancelled) {
            j->slow_rc = MP3S_E_BUSY; j->slow_err = "the pipe was destroyed with this job in flight";
            st = Job::SLOW_DONE;
        } else {
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
    for (hipEvent_t e : {s.e_start, s.e_up, s.e_in, s.e_huff, s.e_rate, s.e_comp, s.e_down, s.e_half}) if (e) (void)hipEventDestroy(e);
    s = Slot();
}

}  // namespace

// a slot's staging and device buffers (sizes from the pipe's): 20 - 60 MB of page-locked memory and as much on the device, milliseconds
// to pin -- a user's pipe takes all its slots when it is made, the context's own pipe takes a slot when a chunk first needs it
// (a two-chunk file never pays for the third)
bool pipe_slot_ready(mp3s_pipe *P, Slot &s)
{
    if (s.h_stage) return true;
    const size_t max_job_bytes = P->max_job_bytes;
    const bool internal = P->internal;
    cpu_set_t before;
    const bool rebind = !P->node_cpus.empty() && sched_getaffinity(0, sizeof before, &before) == 0;
    if (rebind) bind_to(P->node_cpus);           // (first touch of the page-locked staging on the GPU's NUMA node)
    const unsigned ord_flags = trace_on() ? hipEventDefault : hipEventDisableTiming;   // (a traced run prints when each stage of a chunk ended)
    bool ok = true;
    {
        // main data: the file minus headers plus alignment and 8 zero bytes per frame; frames: 96 bytes is the smallest
        // Layer III frame (32 kbit/s at 48 kHz); anything denser (false syncs) overflows the sink and takes the other path.
        // (The context's own pipe cuts its chunks by FRAMES and says how many a chunk can have: max_frames.)
        s.blob_cap = (max_job_bytes + max_job_bytes / 8 + 4096 + 15) & ~(size_t)15;
        s.side_cap = P->max_frames > 0 ? P->max_frames + 16 : max_job_bytes / 96 + 16;
        s.in_cap = (s.side_cap * (72 + 16) + max_job_bytes / 4 + (size_t)kMaxFastFiles * (sizeof(mp3s_chain_seg) + sizeof(mp3s_select_span)) + s.side_cap * 4 * MP3S_SELECT_VARIANTS * 8 + 4096 + 15) & ~(size_t)15;   // (+ variant entries: at most 10 per unit, 8 bytes each)
        s.o_side = s.blob_cap;
        s.o_in = (s.o_side + s.side_cap * sizeof(mp3s_frame_side) + 15) & ~(size_t)15;
        s.fix_cap = internal ? 8 : std::min<size_t>(kMaxFastFiles, s.side_cap);     // (host-decoded last frames: one per file of a job; the context's own pipe runs ONE file's chunks)
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
            hipEventCreateWithFlags(&s.e_huff, ord_flags) != hipSuccess || hipEventCreateWithFlags(&s.e_half, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&s.e_comp, ord_flags) != hipSuccess || hipEventCreateWithFlags(&s.e_rate, ord_flags) != hipSuccess ||
            // the collecting thread sleeps on this one instead of spinning: with one process per GPU on a shared host the
            // cores are needed by the scan workers (the wake-up latency disappears behind the jobs in flight); the
            // context's own pipe has no job behind the one it waits for and spins
            hipEventCreateWithFlags(&s.e_down, internal ? hipEventDefault : hipEventBlockingSync) != hipSuccess)
            ok = false;
    }
    if (rebind) (void)sched_setaffinity(0, sizeof before, &before);
    if (!ok) { (void)hipGetLastError(); free_slot(s); }
    return ok;
}

int pipe_create(mp3s_ctx *c, int depth, size_t max_job_bytes, int scan_threads, bool internal, mp3s_pipe **out, size_t max_frames)
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
    // (the context's OWN pipe -- the chunks of one file -- takes no tail stream: with one its calls run at 1.06-1.12 ms per 10 000 frames or, where
    //  the rehearsal's choice lands the tail on a hardware queue that is in somebody's way, at 1.4-1.6; without, at 1.04-1.08 every time, and the
    //  four tail miniatures are spared its first call: round 5, docs/LOG.md)
    if (pick_lanes(c, &P->s_up, &P->s_down, &P->s_huff, &P->s_comp, &P->s_tail, internal ? 0 : (int)c->opt[MP3S_OPT_PIPE_TAIL], &P->s_dec, c->opt[MP3S_OPT_PIPE_DEC] != 0, &P->s_img, &P->lanes))   // (a stream of their own for the decode transforms: +4 % on a resident batch
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
    // a batch takes 0.79 instead of 0.82 ms (round 3).  MP3S_OPT_PIPE_TAIL: 0 off, 1 on, 2 if the rehearsal gains.
    // the page-locked staging of the slots is allocated by a thread that runs on the GPU's NUMA node (first touch), and the
    // workers that fill it stay there
    if (c->opt[MP3S_OPT_NUMA]) P->node_cpus = gpu_node_cpus(c->device);
    P->max_frames = max_frames;
    P->slots.resize((size_t)depth);
    // (the context's own pipe: the first slot now, the others when a chunk first lands on them -- pipe_slot_ready)
    for (size_t k = 0; k < P->slots.size(); k++)
        if ((!internal || k == 0) && !pipe_slot_ready(P.get(), P->slots[k])) return destroy(MP3S_E_NOMEM, "slot allocation failed");
    if (P->s_comp && !internal) {   // a user's pipe owns its context: the context computes on the pipe's stream until the pipe is gone
        (void)hipStreamSynchronize(c->stream);
        P->s_ctx = c->stream; c->stream = P->s_comp;
    }
    P->todo.resize((size_t)std::max(scan_threads, 1));
    for (int t = 0; t < scan_threads; t++) P->workers.emplace_back(worker, P.get(), t);
    *out = P.release();
    return MP3S_OK;
}

// the job's results are on the host: are they final?  *resolved: the host had to resolve the chains (on the job's own buffers)
bool finish_fast(mp3s_pipe *P, Job *j, Slot &s, bool *resolved)
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

// The pipe's threads ended and its streams drained; its buffers, events and results stay as they are (mp3s_pipe_destroy is what is left to do).
// A context's own pipe that has become too small is parked like this until the context goes (ensure_own_pipe: giving its buffers back at that
// point halves the rate of the copies behind it for the next ten calls).
void pipe_quiesce(mp3s_pipe *P)
{
    if (!P) return;
    {
        std::lock_guard<std::mutex> g(P->mu);
        P->stop = true;
        for (auto &q : P->todo) q.clear();
    }
    P->cv_work.notify_all();
    P->cv_turn.notify_all();      // (a worker that waits for its turn behind a dropped ticket)
    for (auto &t : P->workers) t.join();
    P->workers.clear();
    {
        std::lock_guard<std::mutex> g(P->up.mu);
        P->up.stop = true;
    }
    P->up.cv.notify_all();
    if (P->up.th.joinable()) P->up.th.join();
    (void)hipSetDevice(P->c->device);
    sync_all(P);
    if (P->s_img) (void)hipStreamSynchronize(P->s_img);
    if (P->s_ctx) { P->c->stream = P->s_ctx; P->s_ctx = nullptr; }
}

extern "C" {

int mp3s_pipe_create(mp3s_ctx *c, int depth, size_t max_job_bytes, int scan_threads, mp3s_pipe **out)
{
    if (!c || !out || depth < 1 || depth > 64 || scan_threads < 0 || scan_threads > 64 || max_job_bytes < 4096)
        return fail(MP3S_E_ARG, "bad argument (1 <= depth <= 64; 0 <= scan_threads <= 64, 0 = as many as this rank's share of the host's cores allows; max_job_bytes >= 4096)");
    if (scan_threads == 0) scan_threads = default_scan_threads(c);
    return pipe_create(c, depth, max_job_bytes, scan_threads, false, out, 0);
}

void mp3s_pipe_destroy(mp3s_pipe *P)
{
    if (!P) return;
    pipe_quiesce(P);
    for (hipEvent_t e : P->up.ev) (void)hipEventDestroy(e);
    if (P->up.d_file) (void)hipFree(P->up.d_file);
    if (P->up.d_side) (void)hipFree(P->up.d_side);
    if (P->up.d_blob) (void)hipFree(P->up.d_blob);
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
        down_now(P, *j, s);
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
        down_now(P, *j, s);
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
    out->rehearsal_ms = P->lanes.rehearsal_ms; out->rehearsals = P->lanes.rehearsals; out->lanes = P->lanes.lanes; out->queue_shared = P->lanes.queue_shared;
    return MP3S_OK;
}

}  // extern "C"
