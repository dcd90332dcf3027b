// C-ABI of the library (include/mp3s.h), part 3: PCM in (from the host, or left in HBM by the decode pipeline), MP3 out
// -- the encoder's device batch with its serial chains, and the file / message / block entry points built on it.
#include <sys/stat.h>
#include <unistd.h>
#include <cerrno>
#include "mp3s_internal.h"

static inline size_t up16(size_t x) { return (x + 15) & ~(size_t)15; }

// message bits the frames in front of a block have taken
static int64_t cursor0(const EncSeg &s)
{
    return s.carry_in ? std::min<int64_t>(std::max<int64_t>(s.carry_in->cursor, 0), kNoCursor) : 0;
}

int enc_layout(std::vector<EncSeg> &segs, int samplerate, int bitrate_kbps, EncLayout &L, bool select)
{
    // -1 sits in the reference's bitrate table (encoder/util.py:27,42), so its header check lets it through and the
    // encoder then runs on negative slot counts; nothing meaningful to reproduce
    if (bitrate_kbps <= 0) return fail(MP3S_E_UNSUPPORTED, "bitrate %d", bitrate_kbps);
    if (stream_params(samplerate, bitrate_kbps, &L.sri, &L.bri, &L.whole))
        return fail(MP3S_E_UNSUPPORTED, "unsupported samplerate/bitrate %d/%d", samplerate, bitrate_kbps);
    L.samplerate = samplerate; L.kbps = bitrate_kbps;
    int64_t n64 = 0, hide64 = kPatternBytes;
    for (auto &s : segs) {
        if (s.n_frames <= 0 || s.n_hide < 0 || (s.n_hide > 0 && !s.hide)) return fail(MP3S_E_ARG, "bad stream in the encode batch");
        s.first = (int)n64; s.hide_base = (int)hide64;
        n64 += s.n_frames; hide64 += s.n_hide;
        if (n64 > 0x7fffffff / 8 || hide64 >= kNoCursor - 8) return fail(MP3S_E_ARG, "encode batch too large");
    }
    if (n64 <= 0) return fail(MP3S_E_ARG, "empty encode batch");
    L.lead = segs[0].lead;   // frames that only the transforms see
    if (L.lead < 0 || L.lead > 2 || (segs.size() > 1 && (L.lead || segs[0].first_frame || segs[0].carry_in)))
        return fail(MP3S_E_ARG, "a block of a longer stream is encoded on its own");
    L.n = (int)n64; L.units = L.n * 4; L.n_hide = (int)hide64; L.n_all = L.n + L.lead; L.n_segs = (int)segs.size();
    L.o_rf = up16((size_t)L.n_all * sizeof(mp3s_frame_hdr));
    L.o_cur = L.o_rf + up16((size_t)L.n * sizeof(mp3s_rate_frame));
    L.o_hide = L.o_cur + up16((size_t)L.units * 4);
    L.o_segs = L.o_hide + up16((size_t)L.n_hide);
    L.o_off = L.o_segs + up16((size_t)L.n_segs * sizeof(mp3s_chain_seg));
    L.o_pad = L.o_off + up16(((size_t)L.n + 1) * 4);
    // ---- the streams whose message cursor the device decides (mp3s_select_plan's rule)
    std::vector<mp3s_chain_seg> cs(segs.size());
    std::vector<mp3s_select_span> spans(segs.size());
    for (size_t si = 0; si < segs.size(); si++) {
        const EncSeg &s = segs[si];
        cs[si] = mp3s_chain_seg();
        cs[si].first_frame = s.first; cs[si].n_frames = s.n_frames; cs[si].hide_base = s.hide_base;
        cs[si].hide_begin = (int32_t)((int64_t)s.hide_base + cursor0(s)); cs[si].hide_end = s.hide_base + s.n_hide;
    }
    // every unit a message can reach costs 8 to MP3S_SELECT_VARIANTS entries (1.3 KB each in HBM, one more wave of the
    // rate loop); the alternative for a stream left out is the host resolving its chains, which runs 8 variants per unit
    // as well and needs the host in the middle of the job (a 14 KB message in 10 000 frames: 5.7 ms here, 5.6 ms and more
    // there for the synchronous call; through the pipe the host's turn is what stalls: DESIGN 4a), so the only limit is
    // memory: 8 GB of entries
    const int budget = !select ? 0 : (int)std::min<int64_t>((int64_t)L.units * MP3S_SELECT_VARIANTS, 6000000);
    // how far the message gets is a question of the tables the units in front offer: 2.8 per unit on music, none in silence
    // (the first seconds of many a file).  Where the stream being re-encoded is known, its own table counts say how many
    // units that takes (hiding takes a table away here and there: 1/16 more, and some)
    // The device's re-runs (launch_chain's `redo`: three more launches, which return at once when nothing is listed but
    // still take their 15 us of a stream's time) are issued where they can be needed: what they put right are units that
    // read inherited addresses -- quiet granules, behind a silence -- or ran behind the end of a plan, behind a silence
    // again.  A stream whose own units all carry tables has neither; raw PCM (no stream to ask) gets them always.  A wrong
    // guess costs time only: the check's verdict then sends the job to the host as before.
    L.redo = false;
    for (const EncSeg &s : segs) {
        const size_t known = s.n_guess < 0 ? (size_t)s.n_frames * 4 : (size_t)std::min(s.n_guess, s.n_frames * 4);
        if (s.any_silent >= 0 ? s.any_silent != 0 : (!s.tables_guess || known < (size_t)s.n_frames * 4 || std::memchr(s.tables_guess, 0, known))) L.redo = true;
    }
    std::vector<int32_t> min_reach(segs.size(), 0);
    for (size_t si = 0; si < segs.size(); si++) {
        const EncSeg &s = segs[si];
        const int64_t left = (int64_t)cs[si].hide_end - cs[si].hide_begin;
        if (!s.tables_guess || left <= 0) continue;
        const int64_t need = left + left / 16 + 48;
        int64_t offered = 0, j = 0;
        const int64_t known = s.n_guess < 0 ? (int64_t)s.n_frames * 4 : std::min<int64_t>(s.n_guess, (int64_t)s.n_frames * 4);
        while (j < known && offered < need) offered += s.tables_guess[j++];
        if (offered < need) j += (need - offered) * 5 / 14;   // behind what is known of the stream: 2.8 tables per unit
        min_reach[si] = (int32_t)std::min<int64_t>((int64_t)s.n_frames * 4, j + 32);
    }
    L.n_entries = select_plan(cs.data(), L.n_segs, spans.data(), nullptr, nullptr, budget, min_reach.data());
    L.max_reach = 0;
    for (size_t si = 0; si < segs.size(); si++) {
        segs[si].reach = spans[si].reach; segs[si].first_entry = spans[si].first_entry;
        L.max_reach = std::max(L.max_reach, spans[si].reach);
    }
    L.o_spans = L.o_pad + up16((size_t)L.n);
    L.o_ent = L.o_spans + up16((size_t)L.n_segs * sizeof(mp3s_select_span));
    L.bytes = L.o_ent + up16((size_t)L.n_entries * 8);
    return MP3S_OK;
}

bool enc_variant_buffers(mp3s_ctx *c, const EncLayout &L, EncDev &d)
{
    if (L.n_entries <= 0) { d.d_ixv = nullptr; d.d_outv = nullptr; d.d_env = nullptr; return true; }
    d.d_ixv = (int16_t *)c->grab(kSlotVariants + 1, (size_t)L.n_entries * 1152);
    d.d_outv = (mp3s_gr_out *)c->grab(kSlotVariants + 2, variant_out_bytes(L.n_entries));
    d.d_env = (int32_t *)c->grab(kSlotVariants + 3, (size_t)L.n_entries * 88);
    return d.d_ixv && d.d_outv && d.d_env;
}

int enc_fill(std::vector<EncSeg> &segs, EncLayout &L, uint8_t *dst)
{
    std::memset(dst, 0, L.bytes);
    mp3s_frame_hdr *hdr = (mp3s_frame_hdr *)dst;
    mp3s_rate_frame *rf = (mp3s_rate_frame *)(dst + L.o_rf);
    int32_t *cursor = (int32_t *)(dst + L.o_cur);
    uint8_t *hide_all = dst + L.o_hide;          // [patterns | message of stream 0 | message of stream 1 ...]
    mp3s_chain_seg *cs = (mp3s_chain_seg *)(dst + L.o_segs);
    uint32_t *off = (uint32_t *)(dst + L.o_off);
    uint8_t *pad8 = dst + L.o_pad;
    mp3s_select_span *spans = (mp3s_select_span *)(dst + L.o_spans);
    mp3s_select_patterns(hide_all);
    std::vector<int32_t> padding;
    L.bytes_before = 0;
    for (size_t si = 0; si < segs.size(); si++) {
        EncSeg &s = segs[si];
        // padding / slot lag restart with every stream (MP3_Encoder.py:623-636)
        padding.resize((size_t)s.n_frames);
        const int rc = rate_frames(L.samplerate, L.kbps, 2, s.n_frames, rf + s.first, padding.data(), s.first_frame, &L.bytes_before);
        if (rc) return fail(rc, "unsupported samplerate/bitrate %d/%d", L.samplerate, L.kbps);
        for (int f = s.first; f < s.first + s.n_frames; f++) {
            rf[f].hide_end = s.hide_base + s.n_hide;
            rf[f].stream = (int32_t)si;
            pad8[f] = (uint8_t)padding[(size_t)(f - s.first)];
        }
        for (int f = s.first; f < s.first + s.n_frames + L.lead; f++) {
            hdr[f].sr_idx = (uint8_t)L.sri; hdr[f].nch = 2; hdr[f].ms_stereo = 0; hdr[f].flags = 0; hdr[f].stream_first = (uint32_t)s.first;
        }
        if (s.n_hide) std::memcpy(hide_all + s.hide_base, s.hide, (size_t)s.n_hide);
        // A unit sees the message only through the <= 3 bits at its cursor.  First pass: guess three tables per unit (for
        // a short message in a long stream that is almost always right); a long message is left to the variants.
        // A stream with a plan (reach > 0) needs no guess: its units run "behind every message" and the variants, in the
        // same launch, cover what the message can reach; the selection writes the cursors the units really saw.
        const int64_t c0 = cursor0(s);
        const bool long_msg = s.reach > 0 || s.n_hide - c0 > kLongMessageBits;
        int64_t ahead = 0;               // tables the units in front are expected to take
        for (int j = 0; j < s.n_frames * 4; j++) {
            cursor[(size_t)s.first * 4 + j] = long_msg ? kNoCursor : (int32_t)std::min<int64_t>((int64_t)s.hide_base + c0 + ahead, kNoCursor);
            ahead += s.tables_guess && (s.n_guess < 0 || j < s.n_guess) ? s.tables_guess[j] : 3;
        }
        spans[si].first_entry = s.first_entry; spans[si].reach = s.reach;
        cs[si].first_frame = s.first; cs[si].n_frames = s.n_frames;
        cs[si].hide_base = s.hide_base; cs[si].hide_begin = (int32_t)((int64_t)s.hide_base + c0);   // (both below 2^30)
        cs[si].hide_end = s.hide_base + s.n_hide;
        if (s.carry_in) std::memcpy(cs[si].chain_in, s.carry_in->chain, sizeof cs[si].chain_in);
    }
    if (L.n_entries > 0) {
        int32_t *ent = (int32_t *)(dst + L.o_ent);
        for (const auto &s : segs)
            if (s.reach > 0) select_entries(s.first * 4, s.reach, s.first_entry, s.hide_base + s.n_hide, (int64_t)s.n_hide - cursor0(s), ent, ent + L.n_entries);
    }
    off[0] = 0;
    for (int f = 0; f < L.n; f++) off[f + 1] = off[f] + (uint32_t)(L.whole + pad8[f]);
    L.mp3_bytes = off[L.n];
    for (auto &s : segs) {
        s.mp3_off = off[s.first];
        s.mp3_len = off[s.first + s.n_frames] - off[s.first];
        // the reference drops the cached tail (< 32 bits) at the end of the stream: E14
        if (s.last) s.mp3_len -= std::min<size_t>(s.mp3_len, (size_t)((L.bytes_before + (int64_t)s.mp3_len) % 4));
    }
    return MP3S_OK;
}

void tables_guess_of(const mp3s_frame_side *side, long n_frames, int extra, std::vector<uint8_t> &out)
{
    out.resize((size_t)(n_frames + extra) * 4);
    for (long f = 0; f < n_frames + extra; f++) {
        const mp3s_frame_side &fs = side[f < n_frames ? f : n_frames - 1];
        for (int ch = 0; ch < 2; ch++)
            for (int gr = 0; gr < 2; gr++) {
                const mp3s_unit_side &u = fs.unit[gr][ch];
                // (a granule without big values has no table of its own in this encoder's output; a window-switching one
                // of a foreign stream carries two)
                int n = 0;
                if (u.big_values) for (int r = 0; r < (u.window_switching ? 2 : 3); r++) n += u.table_select[r] != 0;
                out[(size_t)f * 4 + ch * 2 + gr] = (uint8_t)n;
            }
    }
}

int enc_issue(mp3s_ctx *c, const EncLayout &L, const EncDev &d, hipStream_t tail, hipEvent_t tail_from, hipEvent_t rate_after, hipEvent_t pcm_read)
{
    const mp3s_frame_hdr *d_hdr = (const mp3s_frame_hdr *)d.d_in;
    const mp3s_rate_frame *d_rf = (const mp3s_rate_frame *)(d.d_in + L.o_rf);
    const int32_t *d_cur = (const int32_t *)(d.d_in + L.o_cur);
    const uint8_t *d_hide = d.d_in + L.o_hide;
    const mp3s_chain_seg *d_segs = (const mp3s_chain_seg *)(d.d_in + L.o_segs);
    const int32_t *d_mdct = d.d_mdct_all + (size_t)L.lead * 2304;   // the block's own frames
    int rc = mp3s_encode_transform_dev(c, d.d_pcm, d_hdr, L.n_all, d.d_mdct_all);
    if (!rc && pcm_read && hipEventRecord(pcm_read, c->stream) != hipSuccess) rc = fail(MP3S_E_HIP, "event record failed");   // the PCM buffer may be written again
    // (the tail of the job in front is through before this job's rate loop starts: tails do not queue up behind one another)
    if (!rc && rate_after && hipStreamWaitEvent(c->stream, rate_after, 0) != hipSuccess) rc = fail(MP3S_E_HIP, "ordering behind the previous tail failed");
    const bool select_on_tail = tail && L.n_entries > 0;
    // with a tail stream the rate loop is the compute stream's last launch of the job and the tail waits for IT: its dispatch carries
    // tail_from as its own completion signal (a record would be a packet of its own in the queue: 7-8 us of nothing in front of the
    // next job's decode, tools/timeline.sh)
    const bool rate_signals = tail && tail_from && (select_on_tail || L.n_entries <= 0) && (c->opt[MP3S_OPT_PIPE_SIGNALS] & 2);
    if (rate_signals) c->rate_done = tail_from;
    if (!rc && L.n_entries > 0) {
        // short messages: their variants run in the same launch and the device decides the cursor chain (no guess).  With a tail
        // stream the selection (two small launches) belongs to the tail: the compute stream is free for the next job's decode
        // transforms 25 us earlier.  The variant buffers are the context's: a rate loop that writes them waits for the selection in front of it (mp3s_rate_variants_dev).
        const int32_t *d_ent = (const int32_t *)(d.d_in + L.o_ent);
        if (!rc && !select_on_tail)
            rc = mp3s_rate_select_dev(c, d_mdct, d_rf, L.n, d_hide, L.n_hide, const_cast<int32_t *>(d_cur), d_segs,
                                      (const mp3s_select_span *)(d.d_in + L.o_spans), L.n_segs, L.max_reach, d_ent, d_ent + L.n_entries,
                                      L.n_entries, d.d_ix, d.d_out, d.d_en, d.d_ixv, d.d_outv, d.d_env);
        else if (!rc)
            rc = mp3s_rate_variants_dev(c, d_mdct, d_rf, L.n, d_hide, L.n_hide, d_cur, d_ent, d_ent + L.n_entries, L.n_entries, d.d_ix, d.d_out, d.d_en,
                                        d.d_ixv, d.d_outv, d.d_env);
    } else if (!rc)
        rc = mp3s_rate_loop_dev(c, d_mdct, d_rf, L.n, d_hide, L.n_hide, d_cur, nullptr, nullptr, 0, d.d_ix, d.d_out, d.d_en);
    // The tail of a job -- selection, chain check (two small launches each) and bit packing -- reads only the job's own buffers: the pipe
    // gives it a stream of its own, so that the small launches and their gaps lie under the decode transforms of the next
    // job instead of in front of them (bench.py --pack-overlap: 0.829 -> 0.785 ms per step).  Every tail goes through the
    // same stream, so the context's chain scratch and the packer's sync words are still used by one launch at a time.
    c->rate_done = nullptr;
    hipStream_t ts = c->stream;
    if (!rc && tail) {
        if ((!rate_signals && hipEventRecord(tail_from, c->stream) != hipSuccess) || hipStreamWaitEvent(tail, tail_from, 0) != hipSuccess)
            rc = fail(MP3S_E_HIP, "ordering the tail stream failed");
        ts = tail;
    }
    if (!rc && select_on_tail) {
        const int32_t *d_ent = (const int32_t *)(d.d_in + L.o_ent);
        hipStream_t keep = c->stream;
        c->stream = ts;                       // (the entry point launches on the context's stream)
        rc = mp3s_select_dev(c, d_hide, const_cast<int32_t *>(d_cur), d_segs, (const mp3s_select_span *)(d.d_in + L.o_spans), L.n_segs, L.max_reach,
                             d_ent, d_ent + L.n_entries, L.n_entries, d.d_ix, d.d_out, d.d_en, d.d_ixv, d.d_outv, d.d_env);
        c->stream = keep;
        if (!rc && !c->ev_sel && hipEventCreateWithFlags(&c->ev_sel, hipEventDisableTiming) != hipSuccess) rc = fail(MP3S_E_HIP, "event creation failed");
        if (!rc && hipEventRecord(c->ev_sel, ts) != hipSuccess) rc = fail(MP3S_E_HIP, "event record failed");
        if (!rc) c->sel_pending = true;
    }
    if (!rc) {
        // (with the device's own re-runs of the units that inherited other addresses than the zeros they were given)
        const ChainRedoArgs redo = {d_mdct, d_hide, L.n_hide, d.d_ix, d.d_en};
        const int e = launch_chain(ts, d.d_out, d_rf, L.n, d_segs, d_cur, nullptr, d.d_agg, d.d_small,
                                   (mp3s_chain_seg_out *)((uint8_t *)d.d_small + kSmallHead), &c->prof, !c->opt[MP3S_OPT_REDO] || !L.redo ? nullptr : &redo);
        if (e) rc = fail(MP3S_E_HIP, "chain launch: %s", hipGetErrorString((hipError_t)e));
    }
    // packed on the assumption that the verdict is "nothing to redo" (the common case); the caller discards it otherwise
    if (!rc) {
        const bool halves = d.direct_status && d.pack_half && d.pack_split > 0 && d.pack_split < L.n;
        int e = launch_pack(ts, d.d_ix, d.d_out, d.d_en, L.n, L.sri, L.bri, L.whole, (const uint32_t *)(d.d_in + L.o_off), d.d_in + L.o_pad, d.d_mp3,
                            d.d_sc, d.d_small + 2, d.direct_status ? nullptr : c->d_sync, &c->prof, 0, halves ? d.pack_split : -1);
        if (!e && halves) {
            if (hipEventRecord(d.pack_half, ts) != hipSuccess) e = (int)hipErrorUnknown;
            else e = launch_pack(ts, d.d_ix, d.d_out, d.d_en, L.n, L.sri, L.bri, L.whole, (const uint32_t *)(d.d_in + L.o_off), d.d_in + L.o_pad, d.d_mp3,
                                 d.d_sc, d.d_small + 2, nullptr, &c->prof, d.pack_split, -1);
        }
        if (e) rc = fail(MP3S_E_HIP, "pack launch: %s", hipGetErrorString((hipError_t)e));
    }
    return rc;
}

// The guesses of the first pass did not hold (verdict != 0): resolve the serial chains on the host -- the GrInfo of the
// first pass comes down, walk() finds the units that ran on wrong inputs, message variants and exact re-runs replace them,
// and the frames are packed again.  `in` = the host copy of the input block enc_fill wrote, `d` = the device buffers of
// the first pass (its mdct / ix / GrInfo / energies must still be there).  Results as encode_batch's.
int enc_resolve(mp3s_ctx *c, const EncLayout &L, std::vector<EncSeg> &segs, const uint8_t *in, const EncDev &d, mp3s_buf *b, bool have_gr,
                int *passes_out)
{
    const int n = L.n, units = L.units, n_hide = L.n_hide, samplerate = L.samplerate, bitrate_kbps = L.kbps;
    const uint8_t *hide_all = in + L.o_hide;
    const uint8_t *const d_in = d.d_in;
    int16_t *const d_ix = d.d_ix; mp3s_gr_out *const d_out = d.d_out; int32_t *const d_en = d.d_en; uint8_t *const d_mp3 = d.d_mp3;
    int32_t *const d_sc = d.d_sc, *const d_small = d.d_small;
    const int32_t *d_mdct = d.d_mdct_all + (size_t)L.lead * 2304;   // the block's own frames
    const mp3s_rate_frame *d_rf = (const mp3s_rate_frame *)(d_in + L.o_rf);
    const uint8_t *d_hide = d_in + L.o_hide;
    const size_t total = segs.back().mp3_off + segs.back().mp3_len;
    void *d_redo = c->grab(kSlotRedo, (size_t)units * 24);
    if (!d_redo || !b->big[0].reserve(L.mp3_bytes)) return fail(MP3S_E_NOMEM, "memory for resolving %d units", units);
    b->mp3 = b->big[0].data();
    int rc = MP3S_OK;
    double t_mark = trace_on() ? now_ms() : 0;
    auto mark = [&](const char *what) {
        if (!trace_on()) return;
        const double t = now_ms();
        fprintf(stderr, "mp3s:   resolve: %-28s %.3f ms\n", what, t - t_mark);
        t_mark = t;
    };
    if (!b->big[1].reserve((size_t)units * sizeof(mp3s_gr_out))) return fail(MP3S_E_NOMEM, "host memory for %d units", units);
    mp3s_gr_out *const gr = b->gr_out = (mp3s_gr_out *)b->big[1].data();
    if (!have_gr) rc = mp3s_dev_download(c, gr, d_out, (size_t)units * sizeof(mp3s_gr_out));
    b->scfsi.assign((size_t)n * 8, 0);
    std::vector<int32_t> &cursor = c->h_cursor, &state = c->h_state;
    cursor.assign((const int32_t *)(in + L.o_cur), (const int32_t *)(in + L.o_cur) + units);
    // (the selection on the device has written the cursors of the units it replaced, and the chain check's re-runs those
    // of the units they ran again -- also in a batch without a plan)
    if (!rc && (L.n_entries > 0 || (L.redo && c->opt[MP3S_OPT_REDO]))) rc = mp3s_dev_download(c, cursor.data(), d_in + L.o_cur, (size_t)units * 4);
    state.assign((size_t)units * 4, 0);
    int passes = 1;
    // ---- resolve the serial chains, stream by stream: hide cursor (MP3_Encoder.py:808-809) and the per-(gr,ch) inherited
    //      address1/2/3 + quantizerStepSize (E7).  walk() lists the units whose assumed inputs were wrong and, per stream,
    //      where the cursor first went wrong.
    std::vector<int32_t> list, redo_in;
    std::vector<mp3s_gr_out> tmp;
    struct Pending { int unit; int64_t cur; };   // first unit of a stream that ran on a wrong cursor, the right cursor there
    std::vector<Pending> pend(segs.size());
    // cursor / state: what each unit's current result was computed with; want / state_want: what the walk says it should be
    std::vector<int32_t> &want = c->h_want, &state_want = c->h_state_want;
    want.assign(units, 0); state_want.assign((size_t)units * 4, 0);
    auto walk = [&]() {
        list.clear();
        for (size_t si = 0; si < segs.size(); si++) {
            EncSeg &s = segs[si];
            pend[si] = {-1, 0};
            int64_t cur = s.hide_base + cursor0(s);
            const int64_t end = (int64_t)s.hide_base + s.n_hide;
            int32_t chain[4][4] = {};   // [(ch*2+gr)][a1,a2,a3,step]
            if (s.carry_in) std::memcpy(chain, s.carry_in->chain, sizeof chain);
            bool own[4] = {false, false, false, false};   // the block has set chain[k] itself
            s.carry_used = cur < end;                     // the message is still being hidden when the block starts
            for (int u = s.first * 4; u < (s.first + s.n_frames) * 4; u++) {
                const int k = u & 3;
                mp3s_gr_out &g = gr[u];
                bool redo = false;
                const bool active = g.flags & MP3S_RF_ACTIVE;
                if (!own[k] && (!active || (g.flags & MP3S_RF_USED_ADDR_IN))) s.carry_used = true;
                if (active) own[k] = true;
                if (s.n_hide > 0 && active) {
                    const int64_t used = cursor[u];
                    if (used != cur && std::min<int64_t>(used, cur) < end) {
                        redo = true;
                        if (pend[si].unit < 0) pend[si] = {u, cur};
                    }
                }
                if ((g.flags & MP3S_RF_USED_ADDR_IN) &&
                    (state[(size_t)u * 4] != chain[k][0] || state[(size_t)u * 4 + 1] != chain[k][1] ||
                     state[(size_t)u * 4 + 2] != chain[k][2]))
                    redo = true;
                if (redo) list.push_back(u);
                want[u] = (int32_t)std::min<int64_t>(cur, kNoCursor);
                for (int j = 0; j < 4; j++) state_want[(size_t)u * 4 + j] = chain[k][j];
                if (active) {
                    cur += g.n_tables;
                    chain[k][0] = g.address[0]; chain[k][1] = g.address[1]; chain[k][2] = g.address[2];
                    chain[k][3] = g.quantizer_step;
                } else {   // silent unit: everything is inherited (quantizerStepSize and addresses pass through)
                    g.address[0] = chain[k][0]; g.address[1] = chain[k][1]; g.address[2] = chain[k][2];
                    g.quantizer_step = chain[k][3];
                }
                if (g.flags & MP3S_RF_STEP_RANGE) return fail(MP3S_E_STEP_RANGE, "quantizer step left the table in unit %d", u);
            }
            s.hide_offset = cur - s.hide_base;
            s.carry_out.cursor = s.hide_offset;
            std::memcpy(s.carry_out.chain, chain, sizeof chain);
        }
        return (int)MP3S_OK;
    };
    // one launch over a list of (unit, cursor, inherited state) entries: 24 bytes per entry up -- [units | cursors | states]
    // -- and the entries' GrInfo down (into tmp); compact = 1: ix / energies in place, 2: by entry into d_ixv / d_env, and only
    // the table count of every entry comes down (one byte each, into tables_of, through d_tables)
    std::vector<uint8_t> tables_of;
    auto run_entries = [&](const std::vector<int32_t> &units_of, const std::vector<int32_t> &cursor_of, void *d_entries, int compact,
                           int16_t *ix_to, mp3s_gr_out *out_to, int32_t *en_to, uint8_t *d_tables) {
        const size_t nl = units_of.size();
        redo_in.resize(nl * 6);
        for (size_t i = 0; i < nl; i++) {
            const int u = units_of[i];
            redo_in[i] = u;
            redo_in[nl + i] = cursor_of[i];
            for (int j = 0; j < 4; j++) redo_in[2 * nl + 4 * i + j] = state_want[(size_t)u * 4 + j];
        }
        int r = mp3s_dev_upload(c, d_entries, redo_in.data(), nl * 24);
        if (!r) {
            const int32_t *dr = (const int32_t *)d_entries;
            const int e = launch_rate(c->stream, d_mdct, d_rf, n, d_hide, n_hide, dr + nl, dr + 2 * nl, dr, (int)nl, ix_to, out_to, en_to,
                                      &c->prof, 0, compact);
            if (e) r = fail(MP3S_E_HIP, "rate launch: %s", hipGetErrorString((hipError_t)e));
        }
        if (d_tables) {
            if (!r && launch_gather_tables(c->stream, out_to, (int)nl, d_tables)) r = fail(MP3S_E_HIP, "gathering the table counts failed");
            tables_of.resize(nl);
            if (!r) r = mp3s_dev_download(c, tables_of.data(), d_tables, nl);
        } else {
            tmp.resize(nl);
            if (!r) r = mp3s_dev_download(c, tmp.data(), out_to, nl * sizeof(mp3s_gr_out));
        }
        passes++;
        return r;
    };
    mark("GrInfo down");
    if (!rc) rc = walk();
    mark("walk");
    if (!rc && list.size() > kFewUnits) {
        // ---- message variants
        struct Span { size_t seg; int unit, count; int64_t cur; size_t entry; };
        std::vector<Span> spans;
        std::vector<int32_t> ent_unit, ent_cursor, pairs;
        void *d_ent = nullptr, *d_ixv = nullptr, *d_outv = nullptr, *d_env = nullptr, *d_pairs = nullptr;
        for (bool more = true; more && !rc;) {
            spans.clear(); ent_unit.clear(); ent_cursor.clear(); pairs.clear();
            for (size_t si = 0; si < segs.size(); si++) {
                if (pend[si].unit < 0) continue;
                const EncSeg &s = segs[si];
                const int64_t end = (int64_t)s.hide_base + s.n_hide;
                if (pend[si].cur + 3 > end) { pend[si].unit = -1; continue; }   // only the message's last unit is left
                // as many units as the rest of the message can reach at 2.8 tables per unit (measured on music: 2.96; a stream
                // that takes fewer -- silent units take none -- simply comes round again), as many as still fit into this launch
                const int room = (kVariantEntries - (int)ent_unit.size()) / 8;
                const int count = (int)std::min<int64_t>({(int64_t)(s.first + s.n_frames) * 4 - pend[si].unit, (end - pend[si].cur) * 5 / 14 + 32, (int64_t)room});
                if (count <= 0) continue;                                       // next launch
                spans.push_back({si, pend[si].unit, count, pend[si].cur, ent_unit.size()});
                for (int v = 0; v < 8; v++)
                    for (int j = 0; j < count; j++) { ent_unit.push_back(pend[si].unit + j); ent_cursor.push_back(4 * v); }
            }
            if (spans.empty()) break;
            const size_t ne = ent_unit.size();
            int slot = kSlotVariants;
            auto alloc = [&](void **p, size_t bytes) { *p = c->grab(slot++, bytes); return *p != nullptr; };
            if (!alloc(&d_ent, ne * 24) || !alloc(&d_ixv, ne * 1152) || !alloc(&d_outv, ne * sizeof(mp3s_gr_out)) ||
                !alloc(&d_env, ne * 88) || !alloc(&d_pairs, ne))   // at most one pair per unit = ne / 8 pairs of 8 bytes
                rc = fail(MP3S_E_NOMEM, "hipMalloc failed for the message variants");
            mark("variant entries listed");
            // (d_pairs holds the table counts first, the chosen pairs afterwards: ne bytes either way)
            if (!rc) rc = run_entries(ent_unit, ent_cursor, d_ent, 2, (int16_t *)d_ixv, (mp3s_gr_out *)d_outv, (int32_t *)d_env, (uint8_t *)d_pairs);
            mark("variants: up, launch, down");
            more = false;
            for (const Span &sp : spans) {
                if (rc) break;
                const EncSeg &s = segs[sp.seg];
                const int64_t end = (int64_t)s.hide_base + s.n_hide;
                int64_t cur = sp.cur;
                int j = 0;
                for (; j < sp.count && cur + 3 <= end; j++) {     // all three bits the unit can ask for exist
                    const int u = sp.unit + j;
                    const int v = (hide_all[cur] & 1) * 4 + (hide_all[cur + 1] & 1) * 2 + (hide_all[cur + 2] & 1);
                    const size_t e = sp.entry + (size_t)v * sp.count + j;
                    cursor[u] = (int32_t)cur;
                    for (int q = 0; q < 4; q++) state[(size_t)u * 4 + q] = state_want[(size_t)u * 4 + q];
                    pairs.push_back((int32_t)e); pairs.push_back(u);
                    cur += tables_of[e];                           // (the entry's GrInfo goes to the unit's place on the device)
                }
                // span used up with message left: the stream goes on in the next launch; otherwise the message's last unit
                // (fewer than three bits left) and whatever lies behind it are the exact re-run's
                if (j == sp.count && cur < end && sp.unit + sp.count < (s.first + s.n_frames) * 4) { pend[sp.seg] = {sp.unit + sp.count, cur}; more = true; }
                else pend[sp.seg].unit = -1;
            }
            for (size_t si = 0; si < segs.size(); si++) more |= pend[si].unit >= 0;
            mark("variants: select");
            if (!rc && !pairs.empty()) {
                rc = mp3s_dev_upload(c, d_pairs, pairs.data(), pairs.size() * 4);
                mark("variants: pairs up");
                if (!rc) {
                    const int e = launch_scatter(c->stream, (const int32_t *)d_pairs, (int)(pairs.size() / 2), (const int16_t *)d_ixv,
                                                 (const int32_t *)d_env, (const mp3s_gr_out *)d_outv, (int16_t *)d_ix, (int32_t *)d_en, d_out);
                    if (e) rc = fail(MP3S_E_HIP, "scatter: %s", hipGetErrorString((hipError_t)e));
                }
                if (!rc && hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(MP3S_E_HIP, "sync failed");   // the entry arrays are reused
            }
            mark("variants: select, scatter");
        }
        // the GrInfo with the chosen entries in place
        if (!rc) rc = mp3s_dev_download(c, gr, d_out, (size_t)units * sizeof(mp3s_gr_out));
        mark("GrInfo down");
        if (!rc) rc = walk();
        mark("walk");
    }
    // ---- exact re-runs of what is left, until nothing changes
    std::vector<int32_t> cur_of;
    while (!rc && !list.empty()) {
        if (passes > units + 16) { rc = fail(MP3S_E_HIP, "rate-loop chain did not converge"); break; }
        cur_of.resize(list.size());
        for (size_t i = 0; i < list.size(); i++) cur_of[i] = want[list[i]];
        const std::vector<int32_t> units_of = list;
        rc = run_entries(units_of, cur_of, d_redo, 1, (int16_t *)d_ix, (mp3s_gr_out *)d_out, (int32_t *)d_en, nullptr);
        if (!rc)
            for (size_t i = 0; i < units_of.size(); i++) {
                const int u = units_of[i];
                gr[u] = tmp[i];
                cursor[u] = cur_of[i];
                for (int q = 0; q < 4; q++) state[(size_t)u * 4 + q] = state_want[(size_t)u * 4 + q];
            }
        if (!rc) rc = walk();
    }
    mark("exact re-runs");
    if (!rc) {
        // ---- bit packing again, on the final GrInfo
        rc = mp3s_dev_upload(c, d_out, gr, (size_t)units * sizeof(mp3s_gr_out));
        if (!rc) rc = mp3s_pack_frames_dev(c, (const int16_t *)d_ix, (const mp3s_gr_out *)d_out, (const int32_t *)d_en, n, samplerate,
                                           bitrate_kbps, (const uint32_t *)((uint8_t *)d_in + L.o_off), (const uint8_t *)d_in + L.o_pad,
                                           (uint8_t *)d_mp3, (int32_t *)d_sc, (int32_t *)d_small + 2);
        int32_t st = 0;
        if (!rc) rc = mp3s_dev_download(c, &st, (int32_t *)d_small + 2, sizeof st);
        if (!rc && st) rc = fail(MP3S_E_HIP, "bit packer reported status %d", st);
        if (!rc) rc = mp3s_dev_download(c, b->mp3, d_mp3, total);
        if (!rc) rc = mp3s_dev_download(c, b->scfsi.data(), d_sc, (size_t)n * 8 * 4);
    }
    mark("GrInfo up, pack, bytes down");
    if (passes_out) *passes_out = passes;
    return rc;
}


// Encode the streams of `segs` (stereo, one sampling rate and bitrate) as ONE batch: transforms, rate loop, bit packing.
// pcm_dev != nullptr: the int16 PCM is already in HBM (re-encode after a device decode) and pcm is ignored.
// Results: b->mp3 = the streams' MP3 bytes (segs[i].mp3_off / mp3_len); with want_gr also b->gr_out and b->scfsi in
// batch frame order.
int encode_batch(mp3s_ctx *c, const int16_t *pcm, const int16_t *pcm_dev, std::vector<EncSeg> &segs, int samplerate, int bitrate_kbps,
                 mp3s_buf *b, int *passes_out, bool want_gr)
{
    EncLayout L;
    int rc = enc_layout(segs, samplerate, bitrate_kbps, L, c->opt[MP3S_OPT_SELECT] != 0);
    if (rc) return rc;
    const int n = L.n, units = L.units, n_all = L.n_all;
    HIPCHK(hipSetDevice(c->device));
    std::vector<uint8_t> &in = c->h_in;
    in.resize(L.bytes);
    rc = enc_fill(segs, L, in.data());
    if (rc) return rc;
    void *d_pcm = nullptr, *d_in = nullptr, *d_mdct_all = nullptr, *d_ix = nullptr, *d_out = nullptr, *d_en = nullptr,
         *d_agg = nullptr, *d_mp3 = nullptr, *d_sc = nullptr, *d_small = nullptr;
    auto cleanup = [&]() { hipStreamSynchronize(c->stream); };   // the buffers stay in the context's pool
    int slot = 8;
    auto alloc = [&](void **p, size_t bytes) { *p = c->grab(slot++, bytes); return *p != nullptr; };
    if (pcm_dev) { d_pcm = const_cast<int16_t *>(pcm_dev); slot++; }
    else if (!alloc(&d_pcm, (size_t)n_all * 2304 * 2)) d_pcm = nullptr;
    if (!d_pcm || !alloc(&d_in, L.bytes) || !alloc(&d_mdct_all, (size_t)n_all * 2304 * 4) || (slot++ /* kSlotRedo */, false) ||
        !alloc(&d_ix, (size_t)n * 2304 * 2) || !alloc(&d_out, (size_t)units * sizeof(mp3s_gr_out)) || !alloc(&d_en, (size_t)units * 22 * 4) ||
        !alloc(&d_agg, chain_agg_bytes(n)) || !alloc(&d_mp3, L.mp3_bytes + 16) || !alloc(&d_sc, (size_t)n * 8 * 4) ||
        !alloc(&d_small, small_bytes(L.n_segs))) {
        cleanup();
        return fail(MP3S_E_NOMEM, "hipMalloc failed for a %d-frame encode", n);
    }
    const size_t total = segs.back().mp3_off + segs.back().mp3_len;
    if (!b->big[0].reserve(L.mp3_bytes) || !b->big[2].reserve(small_bytes(L.n_segs)) ||
        (want_gr && !b->big[1].reserve((size_t)units * sizeof(mp3s_gr_out)))) {
        cleanup();
        return fail(MP3S_E_NOMEM, "host memory for a %d-frame encode", n);
    }
    b->mp3 = b->big[0].data();
    int32_t *const small = (int32_t *)b->big[2].data();
    const mp3s_chain_seg_out *const seg_out = (const mp3s_chain_seg_out *)(b->big[2].data() + kSmallHead);
    if (want_gr) b->scfsi.assign((size_t)n * 8, 0);
    // ---- one pass, nothing waited for in between: inputs up, transforms, rate loop on the guessed cursors, the chain
    //      check on the device, bit packing, results down.  The verdict says whether the guesses held.
    if (!pcm_dev && hipMemcpyAsync(d_pcm, pcm, (size_t)n_all * 2304 * 2, hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = fail(MP3S_E_HIP, "PCM upload failed");
    if (!rc && hipMemcpyAsync(d_in, in.data(), L.bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = fail(MP3S_E_HIP, "input upload failed");
    EncDev dev;
    dev.d_pcm = (const int16_t *)d_pcm; dev.d_in = (const uint8_t *)d_in; dev.d_mdct_all = (int32_t *)d_mdct_all; dev.d_ix = (int16_t *)d_ix;
    dev.d_out = (mp3s_gr_out *)d_out; dev.d_en = (int32_t *)d_en; dev.d_agg = d_agg; dev.d_mp3 = (uint8_t *)d_mp3; dev.d_sc = (int32_t *)d_sc;
    dev.d_small = (int32_t *)d_small;
    if (!rc && !enc_variant_buffers(c, L, dev)) rc = fail(MP3S_E_NOMEM, "hipMalloc failed for %d variant entries", L.n_entries);
    if (!rc) rc = enc_issue(c, L, dev);
    auto down = [&](void *dst, const void *src, size_t bytes) {
        if (!rc && bytes && hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = fail(MP3S_E_HIP, "download failed");
    };
    down(small, d_small, small_bytes(L.n_segs));
    down(b->mp3, d_mp3, total);
    if (want_gr) {
        down(b->big[1].data(), d_out, (size_t)units * sizeof(mp3s_gr_out));
        down(b->scfsi.data(), d_sc, (size_t)n * 8 * 4);
    }
    if (!rc && hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(MP3S_E_HIP, "sync failed");
    if (rc) { cleanup(); return rc; }
    if (want_gr) b->gr_out = (mp3s_gr_out *)b->big[1].data();
    if (small[0] == 0 && small[1] == 0) {
        if (small[2]) { cleanup(); return fail(MP3S_E_HIP, "bit packer reported status %d", small[2]); }
        for (size_t si = 0; si < segs.size(); si++) {
            EncSeg &s = segs[si];
            s.hide_offset = seg_out[si].cursor - s.hide_base;
            s.carry_out.cursor = s.hide_offset;
            std::memcpy(s.carry_out.chain, seg_out[si].chain, sizeof s.carry_out.chain);
            s.carry_used = seg_out[si].carry_used != 0;
        }
        if (passes_out) *passes_out = 1;
        return MP3S_OK;
    }
    // ---- the guesses did not hold: resolve the chains on the host, as round 1 did
    if (trace_on()) fprintf(stderr, "mp3s: encode of %d frames: %d units still to redo behind the device's re-runs (step range %d), %d variant entries, redo launches %d -> the host resolves\n", n, small[0], small[1], L.n_entries, (int)L.redo);
    EncDev devr = dev;
    rc = enc_resolve(c, L, segs, in.data(), devr, b, want_gr, passes_out);
    cleanup();
    return rc;
}

extern "C" {

static int encode_core(mp3s_ctx *c, const int16_t *pcm, const int16_t *pcm_dev, int64_t n_samples_per_ch, int nch, int samplerate,
                       int bitrate_kbps, const uint8_t *hide_bits, int n_hide, mp3s_buf **owner, mp3s_encoded *out)
{
    if (nch != 2) return fail(MP3S_E_UNSUPPORTED, "mono encode raises IndexError in the reference (SURVEY E3)");
    if (n_samples_per_ch <= 0 || n_samples_per_ch % 1152)
        return fail(MP3S_E_UNSUPPORTED, "sample count %lld is not a multiple of 1152 (reference over-reads, E3)",
                    (long long)n_samples_per_ch);
    if (n_hide < 0 || (n_hide > 0 && !hide_bits)) return fail(MP3S_E_ARG, "bad hide arguments");
    if (n_samples_per_ch / 1152 > 0x7fffffff / 8) return fail(MP3S_E_ARG, "too many frames");
    std::vector<EncSeg> segs(1);
    segs[0].n_frames = (int)(n_samples_per_ch / 1152); segs[0].hide = hide_bits; segs[0].n_hide = n_hide;
    std::unique_ptr<mp3s_buf> b(new mp3s_buf());
    int passes = 0;
    const int rc = encode_batch(c, pcm, pcm_dev, segs, samplerate, bitrate_kbps, b.get(), &passes);
    if (rc) return rc;
    out->n_frames = segs[0].n_frames;
    out->hide_offset = segs[0].hide_offset;
    out->too_long = out->hide_offset < (int64_t)n_hide - 1 ? 1 : 0;
    out->mp3 = b->mp3; out->mp3_len = segs[0].mp3_len;
    out->gr = b->gr_out; out->scfsi = b->scfsi.data();
    out->rate_passes = passes;
    *owner = b.release();
    return MP3S_OK;
}

int mp3s_encode_block(mp3s_ctx *c, const int16_t *pcm, int64_t n_samples_per_ch, int lead_frames, int64_t first_frame, int last_block,
                      int samplerate, int bitrate_kbps, const uint8_t *hide_bits, int n_hide, const mp3s_carry *carry_in,
                      mp3s_carry *carry_out, int32_t *carry_used, mp3s_buf **owner, mp3s_encoded *out)
{
    if (!c || !pcm || !owner || !out) return fail(MP3S_E_ARG, "null pointer");
    if (lead_frames < 0 || lead_frames > 2 || first_frame < 0 || (first_frame == 0 && (lead_frames || carry_in)) ||
        (first_frame > 0 && !carry_in))
        return fail(MP3S_E_ARG, "block arguments: the first block has no lead and no carry, later blocks have a carry");
    if (n_samples_per_ch % 1152 || n_samples_per_ch / 1152 <= lead_frames)
        return fail(MP3S_E_UNSUPPORTED, "sample count %lld is not a multiple of 1152 / holds no frame of its own", (long long)n_samples_per_ch);
    if (n_hide < 0 || (n_hide > 0 && !hide_bits)) return fail(MP3S_E_ARG, "bad hide arguments");
    if (n_samples_per_ch / 1152 > 0x7fffffff / 8) return fail(MP3S_E_ARG, "too many frames");
    std::vector<EncSeg> segs(1);
    EncSeg &s = segs[0];
    s.n_frames = (int)(n_samples_per_ch / 1152) - lead_frames; s.hide = hide_bits; s.n_hide = n_hide;
    s.lead = lead_frames; s.first_frame = first_frame; s.last = last_block != 0; s.carry_in = carry_in;
    std::unique_ptr<mp3s_buf> b(new mp3s_buf());
    int passes = 0;
    const int rc = encode_batch(c, pcm, nullptr, segs, samplerate, bitrate_kbps, b.get(), &passes);
    if (rc) return rc;
    if (carry_out) *carry_out = s.carry_out;
    if (carry_used) *carry_used = s.carry_used ? 1 : 0;
    out->n_frames = s.n_frames;
    out->hide_offset = s.hide_offset;
    out->too_long = out->hide_offset < (int64_t)n_hide - 1 ? 1 : 0;
    out->mp3 = b->mp3; out->mp3_len = s.mp3_len;
    out->gr = b->gr_out; out->scfsi = b->scfsi.data();
    out->rate_passes = passes;
    *owner = b.release();
    return MP3S_OK;
}

int mp3s_encode_pcm(mp3s_ctx *c, const int16_t *pcm, int64_t n_samples_per_ch, int nch, int samplerate, int bitrate_kbps,
                    const uint8_t *hide_bits, int n_hide, mp3s_buf **owner, mp3s_encoded *out)
{
    if (!c || !pcm || !owner || !out) return fail(MP3S_E_ARG, "null pointer");
    return encode_core(c, pcm, nullptr, n_samples_per_ch, nch, samplerate, bitrate_kbps, hide_bits, n_hide, owner, out);
}

static void file_from_encoded(const mp3s_encoded &e, int kbps, int rate, mp3s_file *out)
{
    std::memset(out, 0, sizeof *out);
    out->data = e.mp3; out->len = e.mp3_len; out->kbps = kbps; out->sampling_rate = rate; out->channels = 2;
    out->n_frames = e.n_frames; out->too_long = e.too_long; out->hide_offset = e.hide_offset;
}

int mp3s_encode_file(mp3s_ctx *c, const uint8_t *wav, size_t len, int bitrate_kbps, const uint8_t *hide_bits, int n_hide,
                     mp3s_buf **owner, mp3s_file *out)
{
    if (!c || !wav || !owner || !out) return fail(MP3S_E_ARG, "null pointer");
    mp3s_wav_info w;
    int rc = mp3s_wav_parse(wav, len, bitrate_kbps, &w);
    if (rc) return rc;
    // MP3_Encoder.py:596-618 walks num_of_samples * channels values in steps of 1152 * channels and indexes the buffer
    // as if it were stereo: mono input and a partial last frame both end in IndexError there
    if (w.channels != 2) return fail(MP3S_E_UNSUPPORTED, "mono input: the reference encoder indexes the sample buffer out of bounds");
    // count whole frames, plus one more when samples are left over (:611-614): that frame is read from whatever follows in
    // the buffer -- np.fromfile was asked for twice the declared count (WAV_Reader.py:108), so a chunk behind the data
    // chunk, or the second half of samples that are not 16 bits wide, is taken for audio; only a buffer that ends inside
    // the frame raises (IndexError)
    const int64_t total = w.num_of_samples * 2;
    const int64_t count = total / 2304 + (total % 2304 ? 1 : 0);
    if (count <= 0 || w.n_values < count * 2304)
        return fail(MP3S_E_UNSUPPORTED, "sample count is not a multiple of 1152 per channel and the file ends inside the last frame: the reference encoder reads past the end of the sample buffer");
    std::vector<int16_t> pcm((size_t)count * 2304);   // the data chunk may sit at an odd offset
    std::memcpy(pcm.data(), wav + w.data_offset, pcm.size() * 2);
    mp3s_encoded e;
    rc = encode_core(c, pcm.data(), nullptr, count * 1152, 2, w.samplerate, bitrate_kbps, hide_bits, n_hide, owner, &e);
    if (!rc) file_from_encoded(e, bitrate_kbps, w.samplerate, out);
    return rc;
}

// what the reference's WAV reader / encoder would say to the WAV its decoder writes for this stream
static int reencode_check(const ParsedStream &p, int *kbps_out)
{
    const int kbps = p.bit_rate / 1000;
    int sri, bri, whole;
    // the WAV the reference writes carries the last header's sampling rate; its reader checks the rate, then the bitrate
    if (p.sampling_rate != 32000 && p.sampling_rate != 44100 && p.sampling_rate != 48000)
        return fail(MP3S_E_EXIT, "Unsupported sampling frequency.");
    if (stream_params(p.sampling_rate, kbps, &sri, &bri, &whole)) return fail(MP3S_E_EXIT, "Unsupported bitrate configuration.");
    if (p.nch != 2) return fail(MP3S_E_UNSUPPORTED, "mono input: the reference encoder indexes the sample buffer out of bounds");
    if (p.n_frames <= 0) return fail(MP3S_E_UNSUPPORTED, "no frame in the stream");
    *kbps_out = kbps;
    return MP3S_OK;
}

// Decode the streams `idx` of m (stereo, one sampling rate and bitrate) on the device into HBM and encode them from
// there as one batch: steganography.py:133-182 without the temporary WAV.  bits[i] = framed message of file i (empty:
// nothing hidden).  The batch's bytes are kept in a new part of `top`; out[i] points into it.
static int reencode_group(mp3s_ctx *c, mp3s_multi &m, const std::vector<int> &idx, const std::vector<std::vector<uint8_t>> &bits,
                          int samplerate, int kbps, mp3s_buf *top, mp3s_file *out)
{
    std::vector<EncSeg> segs(idx.size());
    std::vector<std::vector<uint8_t>> guess(idx.size());
    int64_t rows_frames = 0;
    for (size_t k = 0; k < idx.size(); k++) {
        const ParsedStream &p = m.parsed[idx[k]];
        segs[k].n_frames = p.n_frames + (p.dup_last_frame ? 1 : 0);
        const ScannedStream &sc = m.scanned[idx[k]];
        if (!sc.host_parsed && (long)sc.side.size() == p.n_frames && p.n_frames > 0 && !bits[idx[k]].empty()) {
            tables_guess_of(sc.side.data(), p.n_frames, p.dup_last_frame ? 1 : 0, guess[k]);
            segs[k].tables_guess = guess[k].data();
        }
        if (bits[idx[k]].size() > 0x7fffffff) return fail(MP3S_E_ARG, "message too long");
        segs[k].hide = bits[idx[k]].data(); segs[k].n_hide = (int)bits[idx[k]].size();
        rows_frames += segs[k].n_frames;
    }
    if (hipSetDevice(c->device) != hipSuccess) return fail(MP3S_E_HIP, "hipSetDevice failed");
    void *d_keep = c->grab(7, (size_t)rows_frames * 2304 * 2);
    if (!d_keep) return fail(MP3S_E_NOMEM, "hipMalloc failed for %lld frames of PCM", (long long)rows_frames);
    const double t0 = trace_on() ? now_ms() : 0;
    int rc = decode_group(c, m, idx, 2, MP3S_PCM_I16, d_keep);
    if (rc) return rc;
    if (trace_on()) hipStreamSynchronize(c->stream);
    const double t1 = trace_on() ? now_ms() : 0;
    std::unique_ptr<mp3s_buf> part(new mp3s_buf());
    int passes = 0;
    rc = encode_batch(c, nullptr, (const int16_t *)d_keep, segs, samplerate, kbps, part.get(), &passes, false);
    if (rc) return rc;
    if (trace_on())
        fprintf(stderr, "mp3s:   %zu stream(s), %lld frames: decode %.3f ms, encode %.3f ms (%d rate passes)\n", idx.size(),
                (long long)rows_frames, t1 - t0, now_ms() - t1, passes);
    for (size_t k = 0; k < idx.size(); k++) {
        mp3s_file &o = out[idx[k]];
        std::memset(&o, 0, sizeof o);
        o.data = part->mp3 + segs[k].mp3_off; o.len = segs[k].mp3_len;
        o.kbps = kbps; o.sampling_rate = samplerate; o.channels = 2; o.n_frames = segs[k].n_frames;
        o.hide_offset = segs[k].hide_offset;
        o.too_long = segs[k].hide_offset < (int64_t)segs[k].n_hide - 1 ? 1 : 0;
    }
    top->parts.push_back(std::move(part));
    return MP3S_OK;
}

int mp3s_hide_messages(mp3s_ctx *c, const uint8_t *const *mp3s, const size_t *lens, int n_files, const uint8_t *const *msgs,
                       const size_t *msg_lens, mp3s_buf **owner, mp3s_file *out, int32_t *status)
{
    if (!c || !mp3s || !lens || !owner || !out || n_files <= 0 || (msgs && !msg_lens)) return fail(MP3S_E_ARG, "bad argument");
    std::unique_ptr<mp3s_buf> top(new mp3s_buf());
    top->multi.reset(new mp3s_multi());
    mp3s_multi &m = *top->multi;
    m.parsed.resize(n_files); m.scanned.resize(n_files); m.pcm.assign(n_files, nullptr); m.files.resize(n_files);
    std::vector<std::vector<uint8_t>> bits(n_files);
    std::vector<int32_t> st(n_files, MP3S_OK);
    struct Group { int rate, kbps; std::vector<int> idx; };
    std::vector<Group> groups;
    size_t total = 0;
    const double t0 = trace_on() ? now_ms() : 0;
    for (int i = 0; i < n_files; i++) {
        std::memset(&out[i], 0, sizeof out[i]);
        if (!mp3s[i] || (msgs && msgs[i] == nullptr && msg_lens[i])) { st[i] = MP3S_E_ARG; continue; }
        m.files[i] = {mp3s[i], lens[i]};
        total += lens[i];
    }
    if (n_files == 1) m.scanned[0] = std::move(c->spare_scan);   // its capacity: no fresh pages for the blob of a long file
    parallel_files(n_files, total, [&](int i) { if (!st[i]) st[i] = front_end(m, i); });
    const double t1 = trace_on() ? now_ms() : 0;
    for (int i = 0; i < n_files; i++) {
        int kbps = 0;
        if (st[i]) { fail(st[i], st[i] == MP3S_E_ARG ? "file %d: null pointer" : "file %d: malformed or unsupported MP3 stream", i); continue; }
        st[i] = reencode_check(m.parsed[i], &kbps);
        if (st[i]) continue;
        if (msgs && msgs[i]) message_frame(msgs[i], msg_lens[i], bits[i]);
        const int rate = m.parsed[i].sampling_rate;
        size_t g = 0;
        while (g < groups.size() && (groups[g].rate != rate || groups[g].kbps != kbps)) g++;
        if (g == groups.size()) groups.push_back({rate, kbps, {}});
        groups[g].idx.push_back(i);
    }
    const double t2 = trace_on() ? now_ms() : 0;
    for (const Group &g : groups) {
        const int rc = reencode_group(c, m, g.idx, bits, g.rate, g.kbps, top.get(), out);
        if (!rc) continue;
        // one stream spoils its batch (main data the host parser rejects ...): each file on its own, to name it
        for (int i : g.idx) st[i] = g.idx.size() == 1 ? rc : reencode_group(c, m, std::vector<int>{i}, bits, g.rate, g.kbps, top.get(), out);
    }
    if (trace_on())
        fprintf(stderr, "mp3s: hide_messages, %d file(s): scan %.3f ms, messages + grouping %.3f ms, device batches %.3f ms\n", n_files,
                t1 - t0, t2 - t1, now_ms() - t2);
    m.files.clear();   // borrowed pointers
    if (n_files == 1) c->spare_scan = std::move(m.scanned[0]);
    int first_bad = MP3S_OK;
    for (int i = 0; i < n_files; i++) {
        if (status) status[i] = st[i];
        if (st[i] && !first_bad) first_bad = st[i];
    }
    if (!status && first_bad) return first_bad;
    *owner = top.release();
    return MP3S_OK;
}

int mp3s_reencode_block(mp3s_ctx *c, const uint8_t *mp3, size_t len, const uint8_t *utf8, size_t n_msg, int rank, int world,
                        const mp3s_carry *carry_in, mp3s_buf **owner, mp3s_block *out)
{
    return mp3s_reencode_block_indexed(c, mp3, len, nullptr, utf8, n_msg, rank, world, carry_in, owner, out);
}

int mp3s_reencode_block_indexed(mp3s_ctx *c, const uint8_t *mp3, size_t len, const mp3s_index *index, const uint8_t *utf8, size_t n_msg,
                                int rank, int world, const mp3s_carry *carry_in, mp3s_buf **owner, mp3s_block *out)
{
    if (!c || !mp3 || !owner || !out || world <= 0 || rank < 0 || rank >= world || (rank == 0 && carry_in) || (rank > 0 && !carry_in))
        return fail(MP3S_E_ARG, "bad argument (rank 0 has no carry, every other rank has one)");
    std::unique_ptr<mp3s_buf> top(new mp3s_buf());
    top->multi.reset(new mp3s_multi());
    mp3s_multi &m = *top->multi;
    m.parsed.resize(1); m.scanned.resize(1); m.pcm.assign(1, nullptr); m.files.assign(1, {mp3, len});
    const StreamIndex *ix = reinterpret_cast<const StreamIndex *>(index);
    const bool ranged = ix && index_info(ix).gpu_ok;      // scan the block alone; otherwise the whole file, then cut
    int rc = MP3S_OK;
    ParsedStream &p = m.parsed[0];
    if (ranged) {
        const StreamIndexInfo &info = index_info(ix);
        p.n_frames = (int)info.n_frames; p.nch = info.nch; p.sampling_rate = info.sampling_rate; p.bit_rate = info.bit_rate;
        p.dup_last_frame = info.dup_last_frame;
    } else {
        rc = front_end(m, 0);
        if (!rc && !m.scanned[0].gpu_ok && !m.scanned[0].host_parsed && world > 1) {
            // scalefactors inherited across frames: a block of such a stream cannot look back on the device
            rc = parse_stream(mp3, len, p, nullptr);
            m.scanned[0].side.clear(); m.scanned[0].blob.clear();
            m.scanned[0].host_parsed = true;
        }
    }
    if (rc) return fail(rc, "malformed or unsupported MP3 stream");
    int kbps = 0;
    rc = reencode_check(p, &kbps);
    if (rc) return rc;
    const int samplerate = p.sampling_rate;
    std::vector<uint8_t> bits;
    if (utf8) message_frame(utf8, n_msg, bits);
    if (bits.size() > 0x7fffffff) return fail(MP3S_E_ARG, "message too long");
    // blocks of PCM frames; the frame the decoder repeats after a bad header (D12) is the stream's last PCM frame
    const long n = p.n_frames, total = n + (p.dup_last_frame ? 1 : 0);
    const long base = total / world, rem = total % world;
    const long first = rank * base + std::min<long>(rank, rem), count = base + (rank < rem ? 1 : 0);
    std::memset(out, 0, sizeof *out);
    out->total_frames = total; out->first_frame = first; out->n_frames = count; out->is_last = first + count == total;
    out->file.kbps = kbps; out->file.sampling_rate = samplerate; out->file.channels = 2;
    if (count == 0) { m.files.clear(); *owner = top.release(); return MP3S_OK; }
    const int lead = first > 0 ? 1 : 0;                   // PCM in front of the block, for the encoder's filter state
    const int halo = first - lead > 0 ? 1 : 0;            // a frame in front of that, for the decoder's
    const long w0 = first - lead - halo, w1 = std::min(first + count, n);
    const bool with_dup = first + count == total && p.dup_last_frame;
    m.window.assign(1, {w0, w1 - w0, with_dup});
    if (ranged) {
        m.index = ix;
        rc = front_end(m, 0);
        m.index = nullptr;
        if (rc) return fail(rc, "malformed or unsupported MP3 stream");
    } else cut_window(p, m.scanned[0], w0, w1 - w0);
    if (!with_dup) p.dup_last_frame = 0;
    const int64_t rows_frames = (w1 - w0) + (with_dup ? 1 : 0);
    if (hipSetDevice(c->device) != hipSuccess) return fail(MP3S_E_HIP, "hipSetDevice failed");
    void *d_keep = c->grab(7, (size_t)rows_frames * 2304 * 2);
    if (!d_keep) return fail(MP3S_E_NOMEM, "hipMalloc failed for %lld frames of PCM", (long long)rows_frames);
    rc = decode_group(c, m, std::vector<int>{0}, 2, MP3S_PCM_I16, d_keep);
    if (rc) return rc;
    std::vector<EncSeg> segs(1);
    EncSeg &s = segs[0];
    s.n_frames = (int)count; s.hide = bits.data(); s.n_hide = (int)bits.size();
    s.lead = lead; s.first_frame = first; s.last = out->is_last != 0; s.carry_in = carry_in;
    std::unique_ptr<mp3s_buf> part(new mp3s_buf());
    rc = encode_batch(c, nullptr, (const int16_t *)d_keep + (size_t)halo * 2304, segs, samplerate, kbps, part.get(), nullptr, false);
    if (rc) return rc;
    out->carry_out = s.carry_out; out->carry_used = s.carry_used ? 1 : 0;
    out->file.data = part->mp3 + s.mp3_off; out->file.len = s.mp3_len; out->file.n_frames = (int32_t)count;
    out->file.hide_offset = s.hide_offset;
    out->file.too_long = s.hide_offset < (int64_t)s.n_hide - 1 ? 1 : 0;
    top->parts.push_back(std::move(part));
    m.files.clear();
    *owner = top.release();
    return MP3S_OK;
}

int mp3s_hide_message_chunked(mp3s_ctx *c, const uint8_t *mp3, size_t len, const uint8_t *utf8, size_t n_msg, int64_t chunk_frames,
                              mp3s_buf **owner, mp3s_file *out)
{
    if (!c || !mp3 || !owner || !out || chunk_frames < 2) return fail(MP3S_E_ARG, "bad argument (chunks of at least 2 frames)");
    if (chunk_frames >= 4) {
        // the chunks through the overlapped stages, several in flight; a stream that path does not take: chunk after chunk below
        const int64_t keep = c->opt[MP3S_OPT_CHUNK_FRAMES];
        c->opt[MP3S_OPT_CHUNK_FRAMES] = chunk_frames;
        RunResult r;
        static const uint8_t empty = 0;
        const int rr = run_file(c, mp3, len, utf8 ? kRunHide : kRunClear, utf8 ? utf8 : &empty, utf8 ? n_msg : 0, MP3S_PCM_I16, owner, &r);
        c->opt[MP3S_OPT_CHUNK_FRAMES] = keep;
        if (rr != kRunFallback) {
            if (rr) return rr;
            std::memset(out, 0, sizeof *out);
            out->data = r.mp3; out->len = r.mp3_len; out->kbps = r.kbps; out->sampling_rate = r.sampling_rate; out->channels = 2;
            out->n_frames = (int32_t)r.n_frames; out->too_long = r.too_long; out->hide_offset = r.hide_offset;
            return MP3S_OK;
        }
    }
    mp3s_index *index = nullptr;
    mp3s_index_info info;
    int rc = mp3s_index_stream(mp3, len, &index, &info);
    if (rc) return rc;
    const int64_t total = info.n_frames + (info.dup_last_frame ? 1 : 0);
    const int64_t world64 = std::max<int64_t>(1, (total + chunk_frames - 1) / chunk_frames);
    if (world64 > 0x7fffffff) { mp3s_index_free(index); return fail(MP3S_E_ARG, "too many chunks"); }
    const int world = (int)world64;
    std::unique_ptr<mp3s_buf> top(new mp3s_buf());
    mp3s_carry carry;
    std::memset(out, 0, sizeof *out);
    for (int r = 0; r < world && !rc; r++) {
        mp3s_buf *part = nullptr;
        mp3s_block blk;
        rc = mp3s_reencode_block_indexed(c, mp3, len, index, utf8, n_msg, r, world, r ? &carry : nullptr, &part, &blk);
        if (rc) break;
        // every chunk runs on the real carry of the one in front of it: nothing is guessed, nothing re-run
        carry = blk.carry_out;
        if (blk.file.len) top->bytes.insert(top->bytes.end(), blk.file.data, blk.file.data + blk.file.len);
        out->kbps = blk.file.kbps; out->sampling_rate = blk.file.sampling_rate; out->channels = blk.file.channels;
        out->n_frames += (int32_t)blk.n_frames;
        if (blk.n_frames) { out->too_long = blk.file.too_long; out->hide_offset = blk.file.hide_offset; }
        mp3s_buf_free(part);
    }
    mp3s_index_free(index);
    if (rc) return rc;
    out->data = top->bytes.data(); out->len = top->bytes.size();
    *owner = top.release();
    return MP3S_OK;
}

static int reencode(mp3s_ctx *c, const uint8_t *mp3, size_t len, const uint8_t *msg, size_t n_msg, bool hide, mp3s_buf **owner, mp3s_file *out)
{
    // the file as chunks through the overlapped stages (walk || upload || kernels || download); what that path does not take
    // (irregular streams, mono, a repeated last frame, a message that reaches further than a chunk ...) comes back as
    // kRunFallback and goes through the stages one after the other below -- the same bytes either way
    RunResult r;
    const int rc = run_file(c, mp3, len, hide ? kRunHide : kRunClear, msg, n_msg, MP3S_PCM_I16, owner, &r);
    if (rc != kRunFallback) {
        if (rc) return rc;
        std::memset(out, 0, sizeof *out);
        out->data = r.mp3; out->len = r.mp3_len; out->kbps = r.kbps; out->sampling_rate = r.sampling_rate; out->channels = 2;
        out->n_frames = (int32_t)r.n_frames; out->too_long = r.too_long; out->hide_offset = r.hide_offset;
        return MP3S_OK;
    }
    c->sink_done = 0;                                  // (mp3s_*_fd: what the chunks wrote is written again from the result of the other path)
    const uint8_t *const no_msg = nullptr;
    return mp3s_hide_messages(c, &mp3, &len, 1, hide ? &msg : &no_msg, &n_msg, owner, out, nullptr);
}

// mp3s_hide_message_fd / mp3s_clear_file_fd: reencode() with the context's sink set -- run_file writes the chunks it can as they become
// final --, then whatever is missing, the cut to length, and the result block goes back
static int reencode_fd(mp3s_ctx *c, const uint8_t *mp3, size_t len, const uint8_t *msg, size_t n_msg, bool hide, int fd, mp3s_file *out)
{
    mp3s_buf *owner = nullptr;
    struct stat st;
    c->sink_fd = fd; c->sink_done = 0; c->sink_base = 0; c->sink_early = fstat(fd, &st) == 0 && st.st_size == 0;
    int rc = reencode(c, mp3, len, msg, n_msg, hide, &owner, out);
    c->sink_fd = -1;
    if (rc == MP3S_OK) {
        size_t at = std::min(c->sink_done, out->len);
        while (at < out->len) {
            const ssize_t w = pwrite(fd, out->data + at, out->len - at, (off_t)at);
            if (w <= 0) { rc = fail(MP3S_E_ARG, "writing the result to the file descriptor failed (errno %d)", errno); break; }
            at += (size_t)w;
        }
        if (rc == MP3S_OK && ftruncate(fd, (off_t)out->len) != 0) rc = fail(MP3S_E_ARG, "cutting the output file to length failed (errno %d)", errno);
    }
    if (owner) mp3s_buf_free(owner);
    out->data = nullptr;
    return rc;
}

int mp3s_hide_message(mp3s_ctx *c, const uint8_t *mp3, size_t len, const uint8_t *utf8, size_t n_msg, mp3s_buf **owner, mp3s_file *out)
{
    if (!c || !mp3 || (!utf8 && n_msg) || !owner || !out) return fail(MP3S_E_ARG, "null pointer");
    static const uint8_t empty = 0;
    return reencode(c, mp3, len, utf8 ? utf8 : &empty, n_msg, true, owner, out);
}

int mp3s_clear_file(mp3s_ctx *c, const uint8_t *mp3, size_t len, mp3s_buf **owner, mp3s_file *out)
{
    if (!c || !mp3 || !owner || !out) return fail(MP3S_E_ARG, "null pointer");
    return reencode(c, mp3, len, nullptr, 0, false, owner, out);
}

int mp3s_hide_message_fd(mp3s_ctx *c, const uint8_t *mp3, size_t len, const uint8_t *utf8, size_t n_msg, int fd, mp3s_file *out)
{
    if (!c || !mp3 || (!utf8 && n_msg) || !out || fd < 0) return fail(MP3S_E_ARG, "bad argument");
    static const uint8_t empty = 0;
    return reencode_fd(c, mp3, len, utf8 ? utf8 : &empty, n_msg, true, fd, out);
}

int mp3s_clear_file_fd(mp3s_ctx *c, const uint8_t *mp3, size_t len, int fd, mp3s_file *out)
{
    if (!c || !mp3 || !out || fd < 0) return fail(MP3S_E_ARG, "bad argument");
    return reencode_fd(c, mp3, len, nullptr, 0, false, fd, out);
}

}  // extern "C"
