// C-ABI of the library (include/mp3s.h), part 2: MP3 streams in, PCM out -- the host front end, the device batch
// (Huffman decode + transforms, chunked with a one-frame halo) and the entry points built on them.
#include <sys/stat.h>
#include <unistd.h>
#include <cerrno>
#include "mp3s_internal.h"

// keep frames [first, first + count) of a parsed stream (its main data, side records, samples)
void cut_window(ParsedStream &p, ScannedStream &sc, long first, long count)
{
    const long n = p.n_frames;
    first = std::min(std::max(first, 0L), n);
    count = std::min(std::max(count, 0L), n - first);
    auto cut = [&](auto &v, size_t per) {
        if (v.size() >= (size_t)n * per) v.assign(v.begin() + (size_t)first * per, v.begin() + (size_t)(first + count) * per);
    };
    if (!sc.side.empty()) {
        const size_t b0 = first < n ? sc.side[first].md_off : sc.blob.size();
        const size_t b1 = first + count < n ? sc.side[first + count].md_off : sc.blob.size();
        sc.blob.assign(sc.blob.begin() + b0, sc.blob.begin() + std::max(b0, b1));
        cut(sc.side, 1);
        for (auto &fs : sc.side) fs.md_off -= (uint32_t)b0;
    }
    cut(p.is, 2304); cut(p.si, 4); cut(p.hdr, 1); cut(p.table_select, 12); cut(p.frame_size, 1);
    if (first + count < n) p.dup_last_frame = 0;   // the repeated last frame belongs to the block that ends the stream
    p.n_frames = (int)count;
}

// host front end of stream i of m: byte-level scan; scalefactors + Huffman run on the device unless the stream inherits
// scalefactors across frames (mixed blocks ...), in which case the host parser produces its frames
int front_end(mp3s_multi &m, int i)
{
    ParsedStream &p = m.parsed[i];
    ScannedStream &sc = m.scanned[i];
    if (m.index && index_info(m.index).gpu_ok && (size_t)i < m.window.size()) {
        // the window alone, from the resume point in front of it
        const StreamIndexInfo &info = index_info(m.index);
        const long first = std::min(std::max(m.window[i].first, 0L), info.n_frames);
        const long count = std::min(std::max(m.window[i].count, 0L), info.n_frames - first);
        const int rc = parse_stream_range(m.files[i].first, m.files[i].second, m.index, first, count, p, sc);
        if (rc) return rc;
        p.nch = info.nch; p.sampling_rate = info.sampling_rate; p.bit_rate = info.bit_rate;
        p.dup_last_frame = (first + count >= info.n_frames && m.window[i].keep_dup) ? info.dup_last_frame : 0;
        if ((size_t)i < m.all_bits.size()) m.all_bits[i] = p.bits;
        return MP3S_OK;
    }
    int rc = parse_stream(m.files[i].first, m.files[i].second, p, &sc);
    sc.host_parsed = false;
    // scalefactors inherited across frames (mixed blocks, scfsi behind a short granule 0): the device kernel walks back
    // through the stream for them, so it needs the stream from its first frame on; a window of such a stream is parsed here
    if (!rc && !sc.gpu_ok && (size_t)i < m.window.size()) {
        rc = parse_stream(m.files[i].first, m.files[i].second, p, nullptr);
        sc.side.clear(); sc.blob.clear();   // not used for host-parsed streams
        sc.host_parsed = true;
    }
    // no sync where the stream should start: the reference parses nothing and writes an empty WAV (MP3_Parser.py:37-46)
    if (!rc && (size_t)i < m.window.size()) {
        if ((size_t)i < m.all_bits.size()) m.all_bits[i] = p.bits;
        cut_window(p, sc, m.window[i].first, m.window[i].count);
        if (!m.window[i].keep_dup) p.dup_last_frame = 0;
    }
    return rc;
}

// transforms of frames [first, first + cnt) of a resident batch (arrays indexed by batch frame; stream_first counts from
// frame 0 of the batch); the PCM of all but the first `halo` of them goes to d_pcm
int decode_transform_chunk(mp3s_ctx *c, const int16_t *d_is, const mp3s_granule_si *d_si, const mp3s_frame_hdr *d_hdr, long first, int cnt,
                           int nch, int halo, int out_format, void *d_pcm, hipStream_t stream, hipEvent_t done)
{
    int rc = c->ensure_scratch(dec_scratch_bytes(cnt, nch));
    if (rc) return rc;
    if (int pe = guard_probe_usable(c, out_format)) return pe;
    const int e = launch_decode(stream ? stream : c->stream, d_is + (size_t)first * 2304, d_si + (size_t)first * 4, d_hdr + first, cnt, nch, halo, out_format,
                                d_pcm, c->scratch, &c->prof, (int)first, c->synth_eps_scale, c->d_sync, c->opt[MP3S_OPT_FAST_IMDCT] != 0, c->opt[MP3S_OPT_FLOAT_FAST] != 0,
                                c->opt[MP3S_OPT_FUSED_DECODE] != 0, done, [&]() -> const GuardProbe * {
                                    if (!c->guard_probe.x) return nullptr;
                                    c->guard_probe.base = (int64_t)(first + halo) * 1152 * nch;   // (a batch's chunks fill the probe's arrays side by side)
                                    return &c->guard_probe; }());
    if (e) return fail(MP3S_E_HIP, "decode launch: %s", hipGetErrorString((hipError_t)e));
    return MP3S_OK;
}

// Decode the streams listed in `idx` (all with the same channel count) as ONE batch.
// d_keep != nullptr: the PCM of the group stays on the device there (frames back to back, a duplicated last frame
// included) and nothing is downloaded -- the re-encode path of mp3s_hide_message / mp3s_clear_file.
int decode_group(mp3s_ctx *c, mp3s_multi &m, const std::vector<int> &idx, int nch, int out_format, void *d_keep)
{
    const size_t esz = pcm_elem(out_format), frame_bytes = (size_t)1152 * nch * esz;
    long n = 0;
    for (int i : idx) n += m.parsed[i].n_frames;
    if (n <= 0) return MP3S_OK;
    if (n > 0x7fffffff / 8) return fail(MP3S_E_ARG, "batch of %ld frames is too large", n);
    const double t_cat0 = trace_on() ? now_ms() : 0;
    // ---- one batch: frame headers (stream_first = first frame of the file), side records, main-data blobs.  A single
    //      device-decoded stream is used where it lies; several are laid end to end in work arrays the context keeps
    //      (tens of MB of fresh vectors per call cost more in page faults than the copy itself).
    std::vector<mp3s_frame_hdr> &hdr = c->h_hdr;
    hdr.resize((size_t)n);
    std::vector<long> first_of(idx.size());
    bool any_dev = false, any_host = false;
    const bool in_place = idx.size() == 1 && !m.scanned[idx[0]].host_parsed;
    if (!in_place) { c->h_side.resize((size_t)n); c->h_blob.clear(); }
    long f0 = 0;
    for (size_t k = 0; k < idx.size(); k++) {
        const ParsedStream &p = m.parsed[idx[k]];
        const ScannedStream &sc = m.scanned[idx[k]];
        first_of[k] = f0;
        const bool dev = !sc.host_parsed;
        (dev ? any_dev : any_host) = true;
        if (dev && !in_place) c->h_blob.resize((c->h_blob.size() + 3) & ~(size_t)3, 0);   // md_off stays a multiple of 4 (mp3s.h)
        const uint32_t base = (uint32_t)c->h_blob.size();
        if (dev && !in_place) c->h_blob.insert(c->h_blob.end(), sc.blob.begin(), sc.blob.end());
        for (int f = 0; f < p.n_frames; f++) {
            hdr[(size_t)f0 + f] = p.hdr[f];
            hdr[(size_t)f0 + f].stream_first = (uint32_t)f0;
            if (in_place) continue;
            if (dev) { c->h_side[(size_t)f0 + f] = sc.side[f]; c->h_side[(size_t)f0 + f].md_off += base; c->h_side[(size_t)f0 + f].reserved = (uint32_t)f0; }
            else std::memset(&c->h_side[(size_t)f0 + f], 0, sizeof(mp3s_frame_side));   // filled from the host parse below
        }
        f0 += p.n_frames;
    }
    if (!in_place && c->h_blob.empty()) c->h_blob.resize(16, 0);
    const mp3s_frame_side *side = in_place ? m.scanned[idx[0]].side.data() : c->h_side.data();
    const uint8_t *blob = in_place ? m.scanned[idx[0]].blob.data() : c->h_blob.data();
    const size_t blob_bytes = in_place ? m.scanned[idx[0]].blob.size() : c->h_blob.size();
    if (hipSetDevice(c->device) != hipSuccess) return fail(MP3S_E_HIP, "hipSetDevice failed");
    const int chunk = (int)std::min<long>(n, kDecodeChunk) + 1;
    int slot = 0;
    auto grab = [&](size_t bytes) { return c->grab(slot++, bytes); };
    void *d_is = grab((size_t)n * 2304 * 2), *d_si = grab((size_t)n * 4 * sizeof(mp3s_granule_si)),
         *d_hdr = grab((size_t)n * sizeof(mp3s_frame_hdr)), *d_pcm = grab((size_t)chunk * frame_bytes), *d_st = grab(((size_t)n + 1) * 4),
         *d_blob = grab(blob_bytes), *d_side = grab((size_t)n * sizeof(mp3s_frame_side));
    if (!d_is || !d_si || !d_hdr || !d_pcm || !d_st || !d_blob || !d_side)
        return fail(MP3S_E_NOMEM, "hipMalloc failed for a %ld-frame decode", n);
    int rc = MP3S_OK;
    const double t_up0 = trace_on() ? now_ms() : 0;
    if (any_dev) {
        rc = mp3s_dev_upload(c, d_blob, blob, blob_bytes);
        if (!rc) rc = mp3s_dev_upload(c, d_side, side, (size_t)n * sizeof(mp3s_frame_side));
        if (!rc) {
            const int e = launch_huffman(c->stream, (const uint8_t *)d_blob, (const mp3s_frame_side *)d_side, (int)n, nch,
                                         max_part2_3(side, n), (int16_t *)d_is, (mp3s_granule_si *)d_si, (int32_t *)d_st, c->d_sync + 4, &c->prof, true, (int)c->opt[MP3S_OPT_HUF_LANES]);
            if (e) rc = fail(MP3S_E_HIP, "huffman launch: %s", hipGetErrorString((hipError_t)e));
        }
        const double t_up1 = trace_on() ? now_ms() : 0;
        int32_t st = 0;
        if (!rc) rc = mp3s_dev_download(c, &st, d_st, sizeof st);
        if (trace_on()) fprintf(stderr, "mp3s:   decode batch of %ld frames: concatenate %.3f ms, upload %.3f ms, Huffman %.3f ms\n", n, t_up0 - t_cat0, t_up1 - t_up0, now_ms() - t_up1);
        if (!rc && st) {
            // Something in the Huffman data of some frames is off (region counts, big_values past 576 lines, big values
            // running past part2_3_length -- which is what the cut tail of this library's own output looks like in one file
            // out of eight).  The host parser decides those frames -- it walks a frame with the reference's single bit
            // cursor -- and its samples replace the device's.  Frames of a device-decoded stream inherit nothing from one
            // another, so only the flagged ones are redone.
            std::vector<int32_t> fst((size_t)n);
            rc = mp3s_dev_download(c, fst.data(), (const int32_t *)d_st + 1, (size_t)n * 4);
            std::vector<long> flagged;
            for (long f = 0; f < n; f++)
                if (fst[(size_t)f]) flagged.push_back(f);
            std::vector<int16_t> his(flagged.size() * 2304);          // staged here until the copies have been issued and waited for
            std::vector<mp3s_granule_si> hsi(flagged.size() * 4);
            int redone = 0;
            std::vector<char> whole(idx.size(), 0);                   // streams that go to the host parser as a whole
            for (size_t q = 0; q < flagged.size() && !rc; q++) {
                const long f = flagged[q];
                const size_t k = (size_t)(std::upper_bound(first_of.begin(), first_of.end(), f) - first_of.begin()) - 1;
                const int i = idx[k];
                if (m.scanned[i].host_parsed) continue;
                if (!m.scanned[i].gpu_ok) { whole[k] = 1; continue; }   // its frames inherit from one another: one frame cannot be redone alone
                const int prc = parse_scanned_frame(m.scanned[i].side[(size_t)(f - first_of[k])], m.scanned[i].blob.data(), &his[q * 2304], &hsi[q * 4]);
                if (prc) { rc = fail(prc, "file %d: malformed main data", i); break; }
                if (hipMemcpyAsync((int16_t *)d_is + (size_t)f * 2304, &his[q * 2304], 2304 * sizeof(int16_t), hipMemcpyHostToDevice, c->stream) != hipSuccess ||
                    hipMemcpyAsync((mp3s_granule_si *)d_si + (size_t)f * 4, &hsi[q * 4], 4 * sizeof(mp3s_granule_si), hipMemcpyHostToDevice, c->stream) != hipSuccess)
                    rc = fail(MP3S_E_HIP, "upload of a host-decoded frame failed");
                redone++;
            }
            if (hipStreamSynchronize(c->stream) != hipSuccess && !rc) rc = fail(MP3S_E_HIP, "sync failed");
            for (size_t k = 0; k < idx.size() && !rc; k++) {
                if (!whole[k]) continue;
                ParsedStream &p = m.parsed[idx[k]];
                // into a stream of its own: a parse that fails half way must leave the scan's record as it was (the batch
                // entry points run a failing group's files again one by one: a record cut down to the frames in front of the
                // damage let such a file through as a shorter one)
                ParsedStream whole_parse;
                const int prc = parse_stream(m.files[idx[k]].first, m.files[idx[k]].second, whole_parse, nullptr);
                if (prc) { rc = fail(prc, "file %d: malformed main data", idx[k]); break; }
                whole_parse.bits = std::move(p.bits);
                p = std::move(whole_parse);
                m.scanned[idx[k]].host_parsed = true;
                any_host = true;
                redone += p.n_frames;
            }
            if (trace_on()) fprintf(stderr, "mp3s:   device Huffman status 0x%x: %d frame(s) decoded on the host instead\n", st, redone);
        }
    }
    if (any_host)   // streams that inherit scalefactors across frames were parsed on the host: place their frames
        for (size_t k = 0; k < idx.size() && !rc; k++) {
            const ParsedStream &p = m.parsed[idx[k]];
            if (!m.scanned[idx[k]].host_parsed || !p.n_frames) continue;
            rc = mp3s_dev_upload(c, (int16_t *)d_is + (size_t)first_of[k] * 2304, p.is.data(), (size_t)p.n_frames * 2304 * 2);
            if (!rc) rc = mp3s_dev_upload(c, (mp3s_granule_si *)d_si + (size_t)first_of[k] * 4, p.si.data(),
                                          (size_t)p.n_frames * 4 * sizeof(mp3s_granule_si));
        }
    // ---- transforms in chunks of kDecodeChunk frames; a chunk that starts inside a stream re-runs one halo frame.
    //      Host layout = device layout plus one extra frame after every stream that ends in a bad header (D12): the
    //      reference appends that stream's last PCM frame once more.
    std::vector<long> out_first(idx.size());
    long extra = 0;
    for (size_t k = 0; k < idx.size(); k++) { out_first[k] = first_of[k] + extra; extra += m.parsed[idx[k]].dup_last_frame ? 1 : 0; }
    uint8_t *arena = nullptr;
    if (!d_keep) {
        if (!m.arena[nch].reserve(m.head_room + (size_t)(n + extra) * frame_bytes))
            return fail(MP3S_E_NOMEM, "hipHostMalloc failed for %ld frames of PCM", n + extra);
        arena = m.arena[nch].data() + m.head_room;
    }
    if (!rc) rc = mp3s_dev_upload(c, d_hdr, hdr.data(), (size_t)n * sizeof(mp3s_frame_hdr));
    for (long start = 0; start < n && !rc; start += kDecodeChunk) {
        const int halo = (start && hdr[(size_t)start].stream_first < (uint32_t)start) ? 1 : 0;
        const long first = start - halo;
        const int cnt = (int)std::min<long>(kDecodeChunk, n - start) + halo;
        rc = decode_transform_chunk(c, (const int16_t *)d_is, (const mp3s_granule_si *)d_si, (const mp3s_frame_hdr *)d_hdr, first, cnt, nch,
                                    halo, out_format, d_pcm);
        // copy out in runs that are contiguous on both sides (a run ends where a duplicated frame is inserted)
        const long end = start + (cnt - halo);
        for (long a = start; a < end && !rc;) {
            size_t k = (size_t)(std::upper_bound(first_of.begin(), first_of.end(), a) - first_of.begin()) - 1;
            long b = end;
            for (size_t j = k; j < idx.size() && first_of[j] < end; j++)
                if (m.parsed[idx[j]].dup_last_frame) { b = std::min<long>(end, first_of[j] + m.parsed[idx[j]].n_frames); break; }
            if (b <= a) b = std::min<long>(end, a + 1);
            const size_t dst = (size_t)(out_first[k] + (a - first_of[k])) * frame_bytes, bytes = (size_t)(b - a) * frame_bytes;
            const uint8_t *src = (const uint8_t *)d_pcm + (size_t)(a - start) * frame_bytes;
            if (d_keep) {
                if (hipMemcpyAsync((uint8_t *)d_keep + dst, src, bytes, hipMemcpyDeviceToDevice, c->stream) != hipSuccess)
                    rc = fail(MP3S_E_HIP, "device copy failed");
            } else rc = mp3s_dev_download(c, arena + dst, src, bytes);
            a = b;
        }
    }
    for (size_t k = 0; k < idx.size() && !rc; k++) {
        const ParsedStream &p = m.parsed[idx[k]];
        if (p.dup_last_frame && p.n_frames > 0) {
            const size_t last = (size_t)(out_first[k] + p.n_frames - 1) * frame_bytes;
            if (d_keep) {
                if (hipMemcpyAsync((uint8_t *)d_keep + last + frame_bytes, (uint8_t *)d_keep + last, frame_bytes, hipMemcpyDeviceToDevice,
                                   c->stream) != hipSuccess)
                    rc = fail(MP3S_E_HIP, "device copy failed");
            } else {
                hipStreamSynchronize(c->stream);
                std::memcpy(arena + last + frame_bytes, arena + last, frame_bytes);
            }
        }
        if (!d_keep) m.pcm[idx[k]] = arena + (size_t)out_first[k] * frame_bytes;
    }
    if (!d_keep) hipStreamSynchronize(c->stream);
    return rc;
}

extern "C" {

static int decode_streams_impl(mp3s_ctx *c, const uint8_t *const *files, const size_t *lens, int n_files, int out_format,
                               size_t head_room, mp3s_buf **owner, mp3s_decoded *out, int32_t *status)
{
    if (!c || !files || !lens || !owner || !out || n_files <= 0) return fail(MP3S_E_ARG, "bad argument");
    if (out_format < 0 || out_format > 2) return fail(MP3S_E_ARG, "out_format=%d", out_format);
    mp3s_buf *b = new mp3s_buf();
    b->multi.reset(new mp3s_multi());
    mp3s_multi &m = *b->multi;
    m.head_room = head_room;
    m.parsed.resize(n_files); m.scanned.resize(n_files); m.pcm.assign(n_files, nullptr); m.files.resize(n_files);
    std::vector<int> group[3];
    size_t total = 0;
    std::vector<int> frc(n_files, MP3S_OK);
    for (int i = 0; i < n_files; i++) {
        if (!files[i]) {
            if (!status) { delete b; return fail(MP3S_E_ARG, "file %d is null", i); }
            frc[i] = MP3S_E_ARG;
            continue;
        }
        m.files[i] = {files[i], lens[i]};
        total += lens[i];
    }
    if (n_files == 1) m.scanned[0] = std::move(c->spare_scan);   // its capacity: no fresh pages for the blob of a long file
    parallel_files(n_files, total, [&](int i) { if (!frc[i]) frc[i] = front_end(m, i); });
    for (int i = 0; i < n_files; i++) {
        if (frc[i]) {
            if (!status) { delete b; return fail(frc[i], "file %d: malformed or unsupported MP3 stream", i); }
            m.parsed[i] = ParsedStream();        // nothing of it goes into a batch
            continue;
        }
        if (m.parsed[i].n_frames > 0) group[m.parsed[i].nch].push_back(i);
    }
    for (int nch = 1; nch <= 2; nch++)
        if (!group[nch].empty()) {
            int rc = decode_group(c, m, group[nch], nch, out_format);
            if (rc && status && group[nch].size() > 1) {
                // one stream spoils its batch (main data the host parser rejects): each file on its own, to name it.
                // (The arena of this channel count is re-used per file, so the good files are decoded a second time as a
                // batch once the bad ones are known.)
                std::vector<int> good;
                for (int i : group[nch]) {
                    const int r1 = decode_group(c, m, std::vector<int>{i}, nch, out_format);
                    if (r1) { frc[i] = r1; m.parsed[i] = ParsedStream(); m.pcm[i] = nullptr; }
                    else good.push_back(i);
                }
                rc = good.empty() ? MP3S_OK : decode_group(c, m, good, nch, out_format);
            } else if (rc && status) {
                frc[group[nch][0]] = rc; m.parsed[group[nch][0]] = ParsedStream(); m.pcm[group[nch][0]] = nullptr;
                rc = MP3S_OK;
            }
            if (rc) { delete b; return rc; }
        }
    for (int i = 0; i < n_files; i++) {
        const ParsedStream &p = m.parsed[i];
        std::memset(&out[i], 0, sizeof out[i]);
        if (status) status[i] = frc[i];
        if (frc[i]) continue;
        out[i].n_frames = p.n_frames; out[i].nch = p.nch; out[i].sampling_rate = p.sampling_rate; out[i].bit_rate = p.bit_rate;
        out[i].n_bits = (int32_t)p.bits.size(); out[i].n_rows = (int64_t)1152 * (p.n_frames + p.dup_last_frame);
        out[i].pcm = m.pcm[i]; out[i].bits = p.bits.data();
    }
    if (n_files == 1) c->spare_scan = std::move(m.scanned[0]);   // (the result refers to the parse and the PCM only)
    m.files.clear();   // borrowed
    *owner = b;
    return MP3S_OK;
}

int mp3s_decode_streams(mp3s_ctx *c, const uint8_t *const *files, const size_t *lens, int n_files, int out_format,
                        mp3s_buf **owner, mp3s_decoded *out, int32_t *status)
{
    return decode_streams_impl(c, files, lens, n_files, out_format, 0, owner, out, status);
}

int mp3s_decode_block(mp3s_ctx *c, const uint8_t *file, size_t len, int64_t first_frame, int64_t n_frames, int out_format,
                      mp3s_buf **owner, mp3s_decoded *out)
{
    return mp3s_decode_block_indexed(c, file, len, nullptr, first_frame, n_frames, out_format, owner, out);
}

int mp3s_index_stream(const uint8_t *file, size_t len, mp3s_index **index, mp3s_index_info *info)
{
    if (!file || !index) return fail(MP3S_E_ARG, "null pointer");
    int rc = 0;
    StreamIndexInfo si;
    StreamIndex *ix = index_stream(file, len, &si, &rc);
    if (!ix) return fail(rc ? rc : MP3S_E_NOMEM, "malformed or unsupported MP3 stream");
    *index = reinterpret_cast<mp3s_index *>(ix);
    if (info) {
        info->n_frames = si.n_frames; info->nch = si.nch; info->sampling_rate = si.sampling_rate; info->bit_rate = si.bit_rate;
        info->dup_last_frame = si.dup_last_frame; info->gpu_ok = si.gpu_ok ? 1 : 0;
    }
    return MP3S_OK;
}

void mp3s_index_free(mp3s_index *index) { index_free(reinterpret_cast<StreamIndex *>(index)); }

int mp3s_scan_range(const uint8_t *file, size_t len, const mp3s_index *index, int64_t first_frame, int64_t n_frames, mp3s_buf **owner,
                    mp3s_scanned *out)
{
    if (!file || !index || !owner || !out || first_frame < 0 || n_frames < 0) return fail(MP3S_E_ARG, "bad argument");
    mp3s_buf *b = new mp3s_buf();
    const int rc = parse_stream_range(file, len, reinterpret_cast<const StreamIndex *>(index), (long)first_frame, (long)n_frames, b->parsed, b->scanned);
    if (rc) { delete b; return fail(rc, "malformed or unsupported MP3 stream"); }
    const ParsedStream &p = b->parsed;
    const StreamIndexInfo &info = index_info(reinterpret_cast<const StreamIndex *>(index));
    out->n_frames = p.n_frames; out->nch = info.nch; out->sampling_rate = info.sampling_rate; out->bit_rate = info.bit_rate;
    out->n_bits = (int32_t)p.bits.size(); out->dup_last_frame = first_frame + p.n_frames >= info.n_frames ? info.dup_last_frame : 0;
    out->gpu_ok = b->scanned.gpu_ok ? 1 : 0;
    out->max_part2_3_length = max_part2_3(b->scanned.side.data(), p.n_frames);
    out->side = b->scanned.side.data(); out->hdr = p.hdr.data();
    out->blob = b->scanned.blob.data(); out->blob_len = b->scanned.blob.size();
    out->bits = p.bits.data(); out->frame_size = p.frame_size.data();
    *owner = b;
    return MP3S_OK;
}

int mp3s_decode_block_indexed(mp3s_ctx *c, const uint8_t *file, size_t len, const mp3s_index *index, int64_t first_frame, int64_t n_frames,
                              int out_format, mp3s_buf **owner, mp3s_decoded *out)
{
    if (!c || !file || !owner || !out || first_frame < 0 || n_frames <= 0 || first_frame > 0x7fffffff || n_frames > 0x7fffffff)
        return fail(MP3S_E_ARG, "bad argument");
    if (out_format < 0 || out_format > 2) return fail(MP3S_E_ARG, "out_format=%d", out_format);
    // one frame in front of the block is decoded for its state and dropped: the IMDCT overlap and the synthesis fifo
    // reach back less than a frame (Frame.py:151-153, 81-92)
    const int halo = first_frame > 0 ? 1 : 0;
    std::unique_ptr<mp3s_buf> b(new mp3s_buf());
    b->multi.reset(new mp3s_multi());
    mp3s_multi &m = *b->multi;
    m.parsed.resize(1); m.scanned.resize(1); m.pcm.assign(1, nullptr); m.files.assign(1, {file, len});
    m.window.assign(1, {(long)first_frame - halo, (long)n_frames + halo, true});
    m.all_bits.resize(1);
    m.index = reinterpret_cast<const StreamIndex *>(index);
    int rc = front_end(m, 0);
    m.index = nullptr;
    if (rc) return fail(rc, "malformed or unsupported MP3 stream");
    const ParsedStream &p = m.parsed[0];
    if (p.n_frames <= halo) return fail(MP3S_E_ARG, "the block starts behind the last frame of the stream");
    if (p.nch < 1 || p.nch > 2) return fail(MP3S_E_MALFORMED, "channel count");
    rc = decode_group(c, m, std::vector<int>{0}, p.nch, out_format);
    if (rc) return rc;
    m.files.clear();   // borrowed
    const size_t frame_bytes = (size_t)1152 * p.nch * pcm_elem(out_format);
    out->n_frames = p.n_frames - halo; out->nch = p.nch; out->sampling_rate = p.sampling_rate; out->bit_rate = p.bit_rate;
    out->n_rows = (int64_t)1152 * (p.n_frames - halo + p.dup_last_frame);
    out->pcm = m.pcm[0] + (size_t)halo * frame_bytes;
    out->n_bits = (int32_t)m.all_bits[0].size(); out->bits = m.all_bits[0].data();
    *owner = b.release();
    return MP3S_OK;
}

int mp3s_decode_stream(mp3s_ctx *c, const uint8_t *file, size_t len, int out_format, mp3s_buf **owner, mp3s_decoded *out)
{
    if (!file) return fail(MP3S_E_ARG, "null pointer");
    if (c && owner && out && out_format >= 0 && out_format <= 2) {
        // the file as chunks through the overlapped stages; kRunFallback: one stage after the other below (same samples)
        RunResult r;
        const int rc = run_file(c, file, len, kRunDecode, nullptr, 0, out_format, owner, &r);
        if (rc != kRunFallback) {
            if (rc) return rc;
            out->n_frames = (int32_t)r.n_frames; out->nch = r.nch; out->sampling_rate = r.sampling_rate; out->bit_rate = r.bit_rate;
            out->n_bits = (int32_t)r.n_bits; out->n_rows = r.n_rows; out->pcm = r.pcm; out->bits = r.bits;
            return MP3S_OK;
        }
    }
    return mp3s_decode_streams(c, &file, &len, 1, out_format, owner, out, nullptr);
}

static int decode_file_impl(mp3s_ctx *c, const uint8_t *mp3, size_t len, mp3s_buf **owner, mp3s_file *out);

int mp3s_decode_file(mp3s_ctx *c, const uint8_t *mp3, size_t len, mp3s_buf **owner, mp3s_file *out)
{
    if (!c || !mp3 || !owner || !out) return fail(MP3S_E_ARG, "null pointer");
    return decode_file_impl(c, mp3, len, owner, out);
}

// the WAV into an open file: the PCM of a file that goes through the stages as chunks is written chunk by chunk (run_file, the context's
// sink), then the header, whatever is missing, and the cut to length; *owner keeps out->bits alive
int mp3s_decode_file_fd(mp3s_ctx *c, const uint8_t *mp3, size_t len, int fd, mp3s_buf **owner, mp3s_file *out)
{
    if (!c || !mp3 || !owner || !out || fd < 0) return fail(MP3S_E_ARG, "bad argument");
    struct stat st;
    c->sink_fd = fd; c->sink_done = 0; c->sink_base = 44; c->sink_early = fstat(fd, &st) == 0 && st.st_size == 0;
    int rc = decode_file_impl(c, mp3, len, owner, out);
    c->sink_fd = -1;
    if (rc) return rc;
    const size_t done = out->len > 44 ? std::min(c->sink_done, out->len - 44) : 0;
    auto put = [&](const uint8_t *src, size_t n, size_t at) {
        while (n) {
            const ssize_t w = pwrite(fd, src, n, (off_t)at);
            if (w <= 0) return false;
            src += w; at += (size_t)w; n -= (size_t)w;
        }
        return true;
    };
    if (!put(out->data, 44, 0) || !put(out->data + 44 + done, out->len - 44 - done, 44 + done) || ftruncate(fd, (off_t)out->len) != 0) {
        mp3s_buf_free(*owner); *owner = nullptr;
        return fail(MP3S_E_ARG, "writing the WAV to the file descriptor failed (errno %d)", errno);
    }
    out->data = nullptr;
    return MP3S_OK;
}

static int decode_file_impl(mp3s_ctx *c, const uint8_t *mp3, size_t len, mp3s_buf **owner, mp3s_file *out)
{
    // the PCM lands 64 bytes into its buffer; the 44-byte WAV header goes right in front of it: no second copy
    mp3s_buf *b = nullptr;
    mp3s_decoded d;
    RunResult r;
    int rc = run_file(c, mp3, len, kRunDecode, nullptr, 0, MP3S_PCM_I16, &b, &r);   // chunks through the overlapped stages ...
    if (rc == MP3S_OK) {
        d.n_frames = (int32_t)r.n_frames; d.nch = r.nch; d.sampling_rate = r.sampling_rate; d.bit_rate = r.bit_rate;
        d.n_bits = (int32_t)r.n_bits; d.n_rows = r.n_rows; d.pcm = r.pcm; d.bits = r.bits;
    } else if (rc == kRunFallback) {                                                  // ... or one stage after the other
        c->sink_done = 0;
        rc = decode_streams_impl(c, &mp3, &len, 1, MP3S_PCM_I16, 64, &b, &d, nullptr);
    }
    if (rc) return rc;
    uint8_t *wav;
    if (d.n_rows == 0) {   // nothing decoded: what scipy writes for an empty 1-d array at the header object's initial rate 0
        b->bytes.assign(44, 0);
        wav = b->bytes.data();
        wav_header(0, 1, d.sampling_rate, wav);
    } else {
        wav = const_cast<uint8_t *>(static_cast<const uint8_t *>(d.pcm)) - 44;
        wav_header(d.n_rows, d.nch, d.sampling_rate, wav);
    }
    std::memset(out, 0, sizeof *out);
    out->data = wav; out->len = 44 + (size_t)d.n_rows * (size_t)d.nch * 2;
    out->kbps = d.bit_rate / 1000; out->sampling_rate = d.sampling_rate; out->channels = d.nch; out->n_frames = d.n_frames;
    out->n_bits = d.n_bits; out->bits = d.bits;
    *owner = b;
    return MP3S_OK;
}

}  // extern "C"
