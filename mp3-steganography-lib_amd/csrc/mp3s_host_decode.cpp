// Host front end of the decoder: the serial bit-parsing stage that stays on the CPU by design
// (SURVEY.md section 8 rows a9/a10).  Turns an MP3 file image into the batch the decode-transform
// kernels consume: Huffman-decoded spectra `is`, per granule*channel side records and per-frame
// headers, plus the stego bit string.
//
// Behaviour follows the reference, quirks included:
//   stream loop          decoder/MP3_Parser.py:25-85          (D11, D12)
//   header               decoder/FrameHeader.py:51-192         (D16)
//   side info            decoder/FrameSideInformation.py:39-137 (D10: arrays persist across frames)
//   frame size/reservoir decoder/Frame.py:288-363              (D15, Python slice semantics)
//   scalefactors         decoder/Frame.py:365-441
//   Huffman              decoder/Frame.py:443-559              (D1, D2) -- table driven here, the
//                        reference searches the code book linearly; prefix codes make both agree
//   stego bits           decoder/Frame.py:676-685, decoder/util.py:67-81 (D14)
#include "mp3s_host.h"

#include <cmath>
#include <cstring>
#include <mutex>

namespace mp3s {

namespace {

// ---------------------------------------------------------------- bit reader (decoder/util.py:22-64)
struct Bits {
    const uint8_t *p; long len;
    uint32_t get(long pos, int n) const   // MSB first, zero padded past the end, n <= 32
    {
        uint64_t acc = 0;
        const long b0 = pos >> 3;
        for (int k = 0; k < 5; k++) {
            const long b = b0 + k;
            acc = (acc << 8) | ((b >= 0 && b < len) ? p[b] : 0);
        }
        const int sh = 40 - (int)(pos & 7) - n;
        return n == 0 ? 0u : (uint32_t)((acc >> sh) & ((n == 32) ? 0xffffffffull : ((1ull << n) - 1)));
    }
};

// the same on a buffer that is known to be long enough (zero padded by the caller): 1 <= n <= 32
struct FastBits {
    const uint8_t *p;
    uint32_t get(long pos, int n) const
    {
        uint64_t w;
        std::memcpy(&w, p + (pos >> 3), 8);
        w = __builtin_bswap64(w);
        return (uint32_t)((w << (pos & 7)) >> (64 - n));
    }
};

// ---------------------------------------------------------------- Huffman lookup tables
constexpr int LUT_BITS = 10;
struct HuffLut {
    int max = 0;                       // symbols per axis the reference searches (big_value_max)
    int linbits = 0;
    std::vector<uint16_t> fast;        // [1<<LUT_BITS]: (len<<8)|(x<<4)|y, 0 = go to the long list
    std::vector<uint32_t> long_code;   // left-aligned 32-bit codes longer than LUT_BITS
    std::vector<uint16_t> long_sym;    // (len<<8)|(x<<4)|y
};
HuffLut g_lut[32];
uint16_t g_quad_fast[64];              // count1 table A on 6 bits: (len<<4)|value
std::once_flag g_lut_once;

void build_luts()
{
    const HostTables &H = host_tables();
    static const int dec_max[32] = {1, 2, 3, 3, 0, 4, 4, 6, 6, 6, 8, 8, 8, 16, 0, 16,
                                    16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16};
    for (int t = 0; t < 32; t++) {
        HuffLut &L = g_lut[t];
        L.max = dec_max[t];
        L.linbits = H.huff[t].linbits;
        if (t == 0 || L.max == 0) continue;
        const HostHuff &h = H.huff[t];
        L.fast.assign(1 << LUT_BITS, 0);
        // row-major order = the reference's search order; the first match wins there, so keep the first
        for (int x = 0; x < L.max; x++)
            for (int y = 0; y < L.max; y++) {
                const int len = h.hlen[x * h.ylen + y];
                const uint32_t code = h.hcod[x * h.ylen + y];
                const uint16_t sym = (uint16_t)((len << 8) | (x << 4) | y);
                if (len <= LUT_BITS) {
                    const uint32_t base = code << (LUT_BITS - len);
                    for (uint32_t f = 0; f < (1u << (LUT_BITS - len)); f++)
                        if (!L.fast[base + f]) L.fast[base + f] = sym;
                } else {
                    L.long_code.push_back(code << (32 - len));
                    L.long_sym.push_back(sym);
                }
            }
    }
    const HostHuff &q = H.huff[32];
    for (int e = 0; e < 16; e++) {
        const int len = q.hlen[e];
        const uint32_t base = (uint32_t)q.hcod[e] << (6 - len);
        for (uint32_t f = 0; f < (1u << (6 - len)); f++)
            if (!g_quad_fast[base + f]) g_quad_fast[base + f] = (uint16_t)((len << 4) | e);
    }
}

inline bool huff_decode(const HuffLut &L, uint32_t window, int &x, int &y, int &len)
{
    uint16_t s = L.fast[window >> (32 - LUT_BITS)];
    if (!s) {
        for (size_t i = 0; i < L.long_code.size(); i++) {
            const int l = L.long_sym[i] >> 8;
            if ((L.long_code[i] >> (32 - l)) == (window >> (32 - l))) { s = L.long_sym[i]; break; }
        }
        if (!s) return false;
    }
    len = s >> 8; x = (s >> 4) & 15; y = s & 15;
    return true;
}

// Python list slicing data[start:stop] appended to dst
void py_slice_append(std::vector<uint8_t> &dst, const uint8_t *data, long n, long start, long stop)
{
    if (start < 0) { start += n; if (start < 0) start = 0; }
    if (stop < 0) { stop += n; if (stop < 0) stop = 0; }
    if (start > n) start = n;
    if (stop > n) stop = n;
    if (stop > start) dst.insert(dst.end(), data + start, data + stop);
}

using Header = WalkHeader;   // FrameHeader fields; persist from frame to frame like the Python object (mp3s_host.h)

int parse_header(Header &h, const uint8_t *b)
{
    const int b1 = b[1], b2 = b[2], b3 = b[3];
    const bool v10 = b1 & 0x10, v08 = b1 & 0x08;
    h.version = v10 && v08 ? 1 : (v10 ? 2 : (v08 ? 0 : 2.5));
    h.layer = 4 - (((b1 << 5) & 0xff) >> 6);
    h.crc = b1 & 1;
    static const int rates[3][3] = {{44100, 48000, 32000}, {22050, 24000, 16000}, {11025, 12000, 8000}};
    int row = (int)std::floor(h.version) - 1; if (row < 0) row += 3;   // Python rates[-1]
    const int sc = (b2 >> 2) & 3;
    if (sc != 3) h.sampling_rate = rates[row][sc];
    if (h.sampling_rate == 44100) h.sr_idx = 0;
    else if (h.sampling_rate == 48000) h.sr_idx = 1;
    else if (h.sampling_rate == 32000) h.sr_idx = 2;
    h.mode = (b3 >> 6) & 3;
    h.channels = h.mode == 3 ? 1 : 2;
    if (h.layer == 3) h.mode_ext0 = b3 & 0x20;
    h.padding = (b2 & 2) ? 1 : 0;
    static const int r13[14] = {32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320};
    static const int r12[14] = {32, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320, 384};
    static const int r2x[14] = {8, 16, 24, 32, 40, 48, 56, 64, 80, 96, 112, 128, 144, 160};
    int bi = (b2 >> 4) - 1; if (bi < 0) bi += 14;
    const int *tab = nullptr;
    if (h.version == 1) {
        if (h.layer == 1) { h.bit_rate = b2 * 32; return 0; }
        tab = h.layer == 2 ? r12 : (h.layer == 3 ? r13 : nullptr);
    } else {
        tab = h.layer == 1 ? r13 : (h.layer < 4 ? r2x : nullptr);
    }
    if (tab) { if (bi >= 14) return MP3S_E_MALFORMED; h.bit_rate = tab[bi] * 1000; }
    return 0;
}

struct SideInfo {          // FrameSideInformation arrays (persist across frames, D10)
    int main_data_begin = 0;
    int scfsi[2][4] = {};
    int part2_3_length[2][2] = {}, big_value[2][2] = {}, global_gain[2][2] = {}, scale_fac_compress[2][2] = {};
    int window_switching[2][2] = {}, block_type[2][2] = {}, mixed[2][2] = {}, region0[2][2] = {}, region1[2][2] = {};
    int preflag[2][2] = {}, scalefac_scale[2][2] = {}, count1table[2][2] = {};
    int table_select[2][2][3] = {}, sub_block_gain[2][2][3] = {};
    int sf_l[2][2][22] = {}, sf_s[2][2][3][13] = {};
};

// Scalefactors + Huffman data of one frame (Frame.py:365-559).  `si` brings the frame's side info and keeps the
// scalefactors, which persist from frame to frame in the reference (D10); is / gsi / table_select receive the frame's
// 2304 samples, 4 side records and 12 table indices (is and gsi zeroed by the caller).
int decode_main_data(const HostTables &HT, SideInfo &si, int nch, int sr_idx, const Bits &mb, int16_t *is, mp3s_granule_si *gsi,
                     int32_t *table_select)
{
    long bit = 0;
    for (int gr = 0; gr < 2; gr++)
        for (int ch = 0; ch < nch; ch++) {
            const long max_bit = bit + si.part2_3_length[gr][ch];
            const int sl0 = HT.slen[si.scale_fac_compress[gr][ch]][0], sl1 = HT.slen[si.scale_fac_compress[gr][ch]][1];
            if (si.block_type[gr][ch] == 2 && si.window_switching[gr][ch]) {
                if (si.mixed[gr][ch] == 1) {
                    for (int s = 0; s < 8; s++) { si.sf_l[gr][ch][s] = mb.get(bit, sl0); bit += sl0; }
                    for (int s = 3; s < 6; s++)
                        for (int w = 0; w < 3; w++) { si.sf_s[gr][ch][w][s] = mb.get(bit, sl0); bit += sl0; }
                } else {
                    for (int s = 0; s < 6; s++)
                        for (int w = 0; w < 3; w++) { si.sf_s[gr][ch][w][s] = mb.get(bit, sl0); bit += sl0; }
                }
                for (int s = 6; s < 12; s++)
                    for (int w = 0; w < 3; w++) { si.sf_s[gr][ch][w][s] = mb.get(bit, sl1); bit += sl1; }
                for (int w = 0; w < 3; w++) si.sf_s[gr][ch][w][12] = 0;
            } else {
                if (gr == 0) {
                    for (int s = 0; s < 11; s++) { si.sf_l[gr][ch][s] = mb.get(bit, sl0); bit += sl0; }
                    for (int s = 11; s < 21; s++) { si.sf_l[gr][ch][s] = mb.get(bit, sl1); bit += sl1; }
                } else {
                    static const int SB[5] = {0, 6, 11, 16, 21};
                    for (int i = 0; i < 4; i++) {
                        const int sl = i < 2 ? sl0 : sl1;
                        for (int s = SB[i]; s < SB[i + 1]; s++) {
                            if (si.scfsi[ch][i]) si.sf_l[gr][ch][s] = si.sf_l[0][ch][s];
                            else { si.sf_l[gr][ch][s] = mb.get(bit, sl); bit += sl; }
                        }
                    }
                }
                si.sf_l[gr][ch][21] = 0;
            }
            // ---- Huffman (Frame.py:443-559)
            int16_t *smp = &is[(gr * 2 + ch) * 576];
            int region0, region1;
            if (si.window_switching[gr][ch] && si.block_type[gr][ch] == 2) { region0 = 36; region1 = 576; }
            else {
                const int i0 = si.region0[gr][ch] + 1, i1 = i0 + si.region1[gr][ch] + 1;
                if (i0 > 22 || i1 > 22) return MP3S_E_MALFORMED;
                region0 = HT.dev.sfb_long[sr_idx][i0]; region1 = HT.dev.sfb_long[sr_idx][i1];
            }
            int sample = 0;
            const int bv2 = si.big_value[gr][ch] * 2;
            while (sample < bv2) {
                if (sample + 1 >= 576) return MP3S_E_MALFORMED;
                const int tn = si.table_select[gr][ch][sample < region0 ? 0 : (sample < region1 ? 1 : 2)];
                const HuffLut &L = g_lut[tn];
                if (tn == 0 || L.max == 0) { sample += 2; continue; }   // tables 0, 4, 14: zeros, no bits (D2)
                int v[2], len;
                if (huff_decode(L, mb.get(bit, 32), v[0], v[1], len)) {
                    bit += len;
                    for (int i = 0; i < 2; i++) {
                        int lin = 0;
                        if (L.linbits && v[i] == L.max - 1) { lin = (int)mb.get(bit, L.linbits); bit += L.linbits; }
                        int sign = 1;
                        if (v[i] > 0) { sign = mb.get(bit, 1) ? -1 : 1; bit += 1; }
                        smp[sample + i] = (int16_t)(sign * (v[i] + lin));
                    }
                }
                sample += 2;
            }
            while (bit < max_bit && sample + 4 < 576) {   // D1
                int val[4] = {0, 0, 0, 0};
                if (si.count1table[gr][ch] == 1) {
                    const uint32_t bs = mb.get(bit, 4); bit += 4;
                    val[0] = (bs & 8) ? 0 : 1; val[1] = (bs & 4) ? 0 : 1; val[2] = (bs & 2) ? 0 : 1; val[3] = (bs & 1) ? 0 : 1;
                } else {
                    const uint16_t s = g_quad_fast[mb.get(bit, 6)];
                    if (s) {
                        bit += s >> 4;
                        const int e = s & 15;
                        val[0] = (e >> 3) & 1; val[1] = (e >> 2) & 1; val[2] = (e >> 1) & 1; val[3] = e & 1;
                    }
                }
                for (int i = 0; i < 4; i++)
                    if (val[i] > 0) { if (mb.get(bit, 1)) val[i] = -val[i]; bit += 1; }
                for (int i = 0; i < 4; i++) smp[sample + i] = (int16_t)val[i];
                sample += 4;
            }
            bit = max_bit;
            // ---- side record for the kernels
            mp3s_granule_si &g = gsi[gr * 2 + ch];
            g.global_gain = (uint8_t)si.global_gain[gr][ch];
            g.scalefac_scale = (uint8_t)si.scalefac_scale[gr][ch];
            g.block_type = (uint8_t)si.block_type[gr][ch];
            g.mixed_block_flag = (uint8_t)si.mixed[gr][ch];
            g.preflag = (uint8_t)si.preflag[gr][ch];
            for (int w = 0; w < 3; w++) g.sub_block_gain[w] = (uint8_t)si.sub_block_gain[gr][ch][w];
            for (int s = 0; s < 22; s++) g.scale_fac_l[s] = (uint8_t)si.sf_l[gr][ch][s];
            for (int w = 0; w < 3; w++)
                for (int s = 0; s < 13; s++) g.scale_fac_s[w][s] = (uint8_t)si.sf_s[gr][ch][w][s];
            for (int r = 0; r < 3; r++) table_select[(gr * 2 + ch) * 3 + r] = si.table_select[gr][ch][r];
        }
    return 0;
}

}  // namespace

static bool grow_vectors(ScanSink *k, size_t blob_need, size_t side_need)
{
    ScannedStream *sc = static_cast<ScannedStream *>(k->user);
    if (blob_need > k->blob_cap) { sc->blob.resize(std::max(blob_need, 2 * k->blob_cap)); k->blob = sc->blob.data(); k->blob_cap = sc->blob.size(); }
    if (side_need > k->side_cap) { sc->side.resize(std::max(side_need, 2 * k->side_cap)); k->side = sc->side.data(); k->side_cap = sc->side.size(); }
    return true;
}

int parse_stream(const uint8_t *file, size_t flen, ParsedStream &out, ScannedStream *scan)
{
    if (!scan) return parse_stream_sink(file, flen, out, nullptr);
    // main data is the file minus headers and side info, plus padding: one allocation instead of doubling through 40 MB
    // (capacity a caller lends is kept)
    scan->blob.resize(std::max(scan->blob.capacity(), flen + flen / 32 + 64));
    scan->side.resize(std::max(scan->side.capacity(), flen / 96 + 1));
    ScanSink k;
    k.blob = scan->blob.data(); k.blob_cap = scan->blob.size();
    k.side = scan->side.data(); k.side_cap = scan->side.size();
    k.user = scan; k.grow = grow_vectors;
    const int rc = parse_stream_sink(file, flen, out, &k);
    scan->blob.resize(k.blob_len); scan->side.resize(k.n_side);
    scan->gpu_ok = k.gpu_ok;
    return rc;
}

namespace {
// where the frame loop stands in front of a frame: everything the reference's parser objects carry from frame to frame
struct ScanState {
    Header hd;
    SideInfo si;
    double prev_frame_size[9] = {0};
    int frame_size = 0, first_nch = 0;
    long offset = 0;
    std::vector<uint8_t> main_data;     // of the frame before (reused when the reservoir pointer finds nothing: Frame.py:336-356)
};
}  // namespace

// resume points of a stream, one every kIndexStep frames (the state in FRONT of frame k * kIndexStep)
constexpr long kIndexStep = 256;
struct StreamIndex {
    std::vector<ScanState> at;
    StreamIndexInfo info;
};

static int scan_core(const uint8_t *file, size_t flen_, ParsedStream &out, ScanSink *scan, const ScanState *resume, long skip,
                     long max_frames, StreamIndex *record);

int parse_stream_sink(const uint8_t *file, size_t flen, ParsedStream &out, ScanSink *scan)
{
    return scan_core(file, flen, out, scan, nullptr, 0, -1, nullptr);
}

StreamIndex *index_stream(const uint8_t *file, size_t len, StreamIndexInfo *info, int *rc_out)
{
    std::unique_ptr<StreamIndex> ix(new StreamIndex());
    ParsedStream p;
    ScanSink dry;                        // nothing is stored: every frame is "skipped", the state is recorded on the way
    dry.lean = true;
    const int rc = scan_core(file, len, p, &dry, nullptr, -1, -1, ix.get());
    if (rc_out) *rc_out = rc;
    if (rc) return nullptr;
    ix->info.gpu_ok = dry.gpu_ok;
    if (info) *info = ix->info;
    return ix.release();
}
void index_free(StreamIndex *ix) { delete ix; }
const StreamIndexInfo &index_info(const StreamIndex *ix) { return ix->info; }

int parse_stream_range(const uint8_t *file, size_t len, const StreamIndex *ix, long first, long count, ParsedStream &out, ScanSink *sink)
{
    if (!ix || !sink || first < 0 || count < 0) return MP3S_E_ARG;
    first = std::min(first, ix->info.n_frames);
    count = std::min(count, ix->info.n_frames - first);
    const size_t k = std::min<size_t>((size_t)(first / kIndexStep), ix->at.empty() ? 0 : ix->at.size() - 1);
    if (ix->at.empty()) {               // a stream without a frame
        out = ParsedStream(); sink->blob_len = 0; sink->n_side = 0;
        return 0;
    }
    return scan_core(file, len, out, sink, &ix->at[k], first - (long)k * kIndexStep, count, nullptr);
}

int parse_stream_range(const uint8_t *file, size_t len, const StreamIndex *ix, long first, long count, ParsedStream &out, ScannedStream &scan)
{
    const size_t guess = (size_t)std::max<long>(count, 1) * 1500 + 4096;   // grows if the frames are larger
    scan.blob.resize(std::max(scan.blob.capacity(), std::min(guess, len + len / 32 + 64)));
    scan.side.resize(std::max<size_t>(scan.side.capacity(), (size_t)count + 1));
    ScanSink k;
    k.blob = scan.blob.data(); k.blob_cap = scan.blob.size();
    k.side = scan.side.data(); k.side_cap = scan.side.size();
    k.user = &scan; k.grow = grow_vectors;
    const int rc = parse_stream_range(file, len, ix, first, count, out, &k);
    scan.blob.resize(k.blob_len); scan.side.resize(k.n_side);
    scan.gpu_ok = k.gpu_ok;
    return rc;
}

// skip: frames in front of the first one to emit (state only, nothing stored; -1 = all of them); max_frames: frames to emit
// (-1 = to the end of the stream); record: resume points + the stream's summary go there
static int scan_core(const uint8_t *file, size_t flen_, ParsedStream &out, ScanSink *scan, const ScanState *resume, long skip,
                     long max_frames, StreamIndex *record)
{
    std::call_once(g_lut_once, build_luts);
    const HostTables &HT = host_tables();
    const long flen = (long)flen_;
    // (the vectors keep their capacity: a caller that scans file after file with one ParsedStream allocates nothing here --
    // fresh blocks of this size come from mmap, and page faults / unmaps of several scanning threads queue up on the
    // process's memory-map lock)
    out.n_frames = out.nch = out.sampling_rate = out.bit_rate = out.dup_last_frame = 0;
    out.is.clear(); out.si.clear(); out.hdr.clear(); out.bits.clear(); out.table_select.clear(); out.frame_size.clear();
    if (scan) {
        scan->blob_len = 0; scan->n_side = 0; scan->gpu_ok = true;
        if (!scan->hdr) out.hdr.reserve(flen_ / 96 + 1);
        if (!scan->lean) { out.frame_size.reserve(flen_ / 96 + 1); out.bits.reserve(flen_ / 8 + 16); out.table_select.reserve((flen_ / 96 + 1) * 12); }
    }
    // blob bytes of the sink: room for `extra` more, or (fixed capacity) give up
    auto blob_room = [&](size_t extra) -> bool {
        if (scan->blob_len + extra <= scan->blob_cap) return true;
        return scan->grow && scan->grow(scan, scan->blob_len + extra, 0);
    };
    ScanState st;
    if (resume) st = *resume;
    Header &hd = st.hd;
    SideInfo &si = st.si;
    double (&prev_frame_size)[9] = st.prev_frame_size;
    int &frame_size = st.frame_size, &first_nch = st.first_nch;
    long &offset = st.offset;
    std::vector<uint8_t> &main_data = st.main_data;
    long seen = 0;                       // frames passed since the start of this call (skipped ones included)
    bool all_gpu_ok = true;
    // ID3v2 skip (decoder/ID3_Parser.py:95-131): only `offset` and `is_valid` matter to decoding
    if (!resume) {
    if (flen >= 10 && file[0] == 'I' && file[1] == 'D' && file[2] == '3' && !(file[5] & 0x0f)) {
        long size = 0;
        for (int i = 0; i < 4; i++) size = (size << 7) + file[6 + i];
        offset = size + ((file[5] >> 4) & 1 ? 20 : 10);
    }
    if (flen - offset < 2) return MP3S_E_MALFORMED;   // the reference indexes buffer[0] and buffer[1] (MP3_Parser.py:37)
    }
    auto set_frame_size = [&]() -> int {   // Frame.py:288-316
        int spf = 0;
        if (hd.layer == 3) spf = hd.version == 1 ? 1152 : 576;
        else if (hd.layer == 2) spf = 1152;
        else if (hd.layer == 1) spf = 384;
        for (int i = 8; i > 0; i--) prev_frame_size[i] = prev_frame_size[i - 1];
        prev_frame_size[0] = frame_size;
        if (hd.sampling_rate == 0) return MP3S_E_MALFORMED;
        frame_size = (int)((((double)spf / 8) * hd.bit_rate) / hd.sampling_rate);
        if (hd.padding == 1) frame_size += 1;
        return 0;
    };
    const uint8_t *buffer = file + offset;
    bool valid = resume != nullptr;
    if (!resume && buffer[0] == 0xFF && buffer[1] >= 0xE0) {
        valid = true;
        if (flen - offset < 4) return MP3S_E_MALFORMED;   // a sync with no header behind it: IndexError in the reference
        int rc = parse_header(hd, buffer); if (rc) return rc;
        rc = set_frame_size(); if (rc) return rc;   // D11
    }
    bool dup_last = false;
    while (valid && flen > offset + 4) {
        if (max_frames >= 0 && out.n_frames >= max_frames) break;        // the caller's range ends here
        const bool dry = skip < 0 || seen < skip;                        // state only: nothing of this frame is stored
        if (record && seen % kIndexStep == 0) record->at.push_back(st);
        buffer = file + offset;
        const long buflen = flen - offset;
        if (buffer[0] == 0xFF && buffer[1] >= 0xE0) { int rc = parse_header(hd, buffer); if (rc) return rc; }
        else { valid = false; dup_last = seen > 0 || resume; out.dup_last_frame = out.n_frames > 0 ? 1 : 0; break; }   // D12
        int rc = set_frame_size(); if (rc) return rc;
        // D16: band tables exist for 32 / 44.1 / 48 kHz only and are chosen by the sampling rate alone (FrameHeader.py:125-143);
        // a header with another rate keeps the tables of the frame before it, a stream that starts with one has none
        // (IndexError).  Neither the version nor the layer field is checked by the reference: such frames -- reached
        // through false syncs -- are taken apart as Layer III with that header's frame size and bit-rate rule.
        if (frame_size <= 0 || hd.sr_idx < 0) return MP3S_E_MALFORMED;
        if (first_nch == 0) first_nch = hd.channels;
        else if (hd.channels != first_nch) return MP3S_E_UNSUPPORTED;   // ragged pcm_data in the reference
        const int nch = hd.channels;

        // ---- side info: at most 32 bytes, read field by field from a zero-padded copy (bits past the end of the file
        //      read as 0, as in the reference's reader)
        const long sstart = hd.crc == 0 ? 6 : 4;
        uint8_t sbuf[48] = {0};
        if (buflen > sstart) std::memcpy(sbuf, buffer + sstart, (size_t)std::min<long>(buflen - sstart, 40));
        const FastBits sb{sbuf};
        long off = 0;
        si.main_data_begin = (int)sb.get(0, 9); off += 9;
        off += hd.mode == 3 ? 5 : 3;
        for (int ch = 0; ch < nch; ch++)
            for (int b = 0; b < 4; b++) { si.scfsi[ch][b] = sb.get(off, 1) != 0; off += 1; }
        for (int gr = 0; gr < 2; gr++)
            for (int ch = 0; ch < nch; ch++) {
                si.part2_3_length[gr][ch] = sb.get(off, 12); off += 12;
                si.big_value[gr][ch] = sb.get(off, 9); off += 9;
                si.global_gain[gr][ch] = sb.get(off, 8); off += 8;
                si.scale_fac_compress[gr][ch] = sb.get(off, 4); off += 4;
                si.window_switching[gr][ch] = sb.get(off, 1) == 1; off += 1;
                if (si.window_switching[gr][ch]) {
                    si.block_type[gr][ch] = sb.get(off, 2); off += 2;
                    si.mixed[gr][ch] = sb.get(off, 1) == 1; off += 1;
                    si.region0[gr][ch] = si.block_type[gr][ch] == 2 ? 8 : 7;
                    si.region1[gr][ch] = 20 - si.region0[gr][ch];
                    for (int r = 0; r < 2; r++) { si.table_select[gr][ch][r] = sb.get(off, 5); off += 5; }   // [2] stays stale
                    for (int w = 0; w < 3; w++) { si.sub_block_gain[gr][ch][w] = sb.get(off, 3); off += 3; }
                } else {
                    si.block_type[gr][ch] = 0; si.mixed[gr][ch] = 0;
                    for (int r = 0; r < 3; r++) { si.table_select[gr][ch][r] = sb.get(off, 5); off += 5; }
                    si.region0[gr][ch] = sb.get(off, 4); off += 4;
                    si.region1[gr][ch] = sb.get(off, 3); off += 3;
                }
                si.preflag[gr][ch] = sb.get(off, 1); off += 1;
                si.scalefac_scale[gr][ch] = sb.get(off, 1); off += 1;
                si.count1table[gr][ch] = sb.get(off, 1); off += 1;
            }
        // scalefactors that requantisation would read without this frame having written them (D10): mixed blocks, and
        // scfsi reuse when granule 0 carried short-block scalefactors -- such a stream is parsed on the host
        for (int ch = 0; ch < nch; ch++) {
            const bool g0_short = si.window_switching[0][ch] && si.block_type[0][ch] == 2;
            const bool g1_short = si.window_switching[1][ch] && si.block_type[1][ch] == 2;
            if ((si.window_switching[0][ch] && si.mixed[0][ch]) || (si.window_switching[1][ch] && si.mixed[1][ch]) ||
                (g0_short && !g1_short && (si.scfsi[ch][0] | si.scfsi[ch][1] | si.scfsi[ch][2] | si.scfsi[ch][3])))
                all_gpu_ok = false;
        }
        // ---- stego bits: ch -> gr -> region, zeros skipped, H0 -> 0
        if (!dry && !(scan && scan->lean))
        for (int ch = 0; ch < nch; ch++)
            for (int gr = 0; gr < 2; gr++)
                for (int r = 0; r < 3; r++) {
                    const int t = si.table_select[gr][ch][r];
                    if (t) out.bits.push_back(HT.in_h0[t] ? 0 : 1);
                }
        // ---- main data (bit reservoir); in scan mode it is assembled in the blob directly
        int constant = hd.mode == 3 ? 21 : 36;
        if (hd.crc == 0) constant += 2;
        size_t md_start = 0;
        bool full = false;                       // a sink of fixed capacity ran out of room
        const bool to_blob = scan && !dry;
        if (to_blob) {
            if (!blob_room(4)) return MP3S_E_NOMEM;
            while (scan->blob_len & 3) scan->blob[scan->blob_len++] = 0;
            md_start = scan->blob_len;
        }
        // Python list slicing data[start:stop] appended to the frame's main data (the blob in scan mode)
        auto md_append = [&](const uint8_t *data, long n, long start, long stop) {
            if (!to_blob) { py_slice_append(main_data, data, n, start, stop); return; }
            if (start < 0) { start += n; if (start < 0) start = 0; }
            if (stop < 0) { stop += n; if (stop < 0) stop = 0; }
            if (start > n) start = n;
            if (stop > n) stop = n;
            if (stop <= start || full) return;
            if (!blob_room((size_t)(stop - start))) { full = true; return; }
            std::memcpy(scan->blob + scan->blob_len, data + start, (size_t)(stop - start));
            scan->blob_len += (size_t)(stop - start);
        };
        bool rebuilt = false;
        if (si.main_data_begin == 0) {
            if (!to_blob) main_data.clear();
            md_append(buffer, buflen, constant, frame_size);
            rebuilt = true;
        } else {
            double bound = 0;
            for (int fr = 0; fr < 9; fr++) {
                bound += prev_frame_size[fr] - constant;
                if (si.main_data_begin < bound) {
                    double ptr_offset = si.main_data_begin + fr * constant;
                    double part[9] = {0};
                    part[fr] = si.main_data_begin;
                    for (int i = 0; i < fr; i++) { part[i] = prev_frame_size[i] - constant; part[fr] -= part[i]; }
                    if (!to_blob) main_data.clear();
                    long loc = (long)(offset - ptr_offset);
                    md_append(file, flen, loc, loc + (long)part[fr]);
                    ptr_offset -= (part[fr] + constant);
                    for (int i = fr - 1; i >= 0; i--) {
                        loc = (long)(offset - ptr_offset);
                        md_append(file, flen, loc, loc + (long)part[i]);
                        ptr_offset -= (part[i] + constant);
                    }
                    md_append(buffer, buflen, constant, frame_size);
                    rebuilt = true;
                    break;
                }
            }   // not found: the previous frame's main_data is reused, as in the reference
        }
        Bits mb{main_data.data(), (long)main_data.size()};

        if (dry) { seen++; offset += frame_size; continue; }
        if (scan) {
            // ---- scan mode: record the side info; the main data sits in the blob already
            mp3s_frame_side fs;
            std::memset(&fs, 0, sizeof fs);
            if (!scan->lean) out.table_select.resize(((size_t)out.n_frames + 1) * 12, 0);
            if (!rebuilt && scan->n_side) {
                const mp3s_frame_side pv = scan->side[scan->n_side - 1];
                if (!blob_room(pv.md_len)) return MP3S_E_NOMEM;
                std::memmove(scan->blob + scan->blob_len, scan->blob + pv.md_off, pv.md_len);   // (the room may have moved the blob)
                scan->blob_len += pv.md_len;
            } else if (!rebuilt && !main_data.empty()) {   // the frame before was skipped: its main data is in the state
                if (!blob_room(main_data.size())) return MP3S_E_NOMEM;
                std::memcpy(scan->blob + scan->blob_len, main_data.data(), main_data.size());
                scan->blob_len += main_data.size();
            }
            // (a skipped frame that follows an emitted one does not occur: the skipped frames come first)
            if (full || !blob_room(8)) return MP3S_E_NOMEM;
            fs.md_off = (uint32_t)md_start;
            fs.md_len = (uint32_t)(scan->blob_len - md_start);
            std::memset(scan->blob + scan->blob_len, 0, 8);
            scan->blob_len += 8;
            fs.nch = (uint8_t)nch; fs.sr_idx = (uint8_t)hd.sr_idx;
            fs.ms_stereo = (hd.mode == 1 && hd.mode_ext0) ? 1 : 0;
            for (int ch = 0; ch < nch; ch++)
                for (int b = 0; b < 4; b++) fs.scfsi[ch][b] = (uint8_t)si.scfsi[ch][b];
            for (int gr = 0; gr < 2; gr++)
                for (int ch = 0; ch < nch; ch++) {
                    mp3s_unit_side &u = fs.unit[gr][ch];
                    u.part2_3_length = (uint16_t)si.part2_3_length[gr][ch]; u.big_values = (uint16_t)si.big_value[gr][ch];
                    u.global_gain = (uint8_t)si.global_gain[gr][ch]; u.scalefac_compress = (uint8_t)si.scale_fac_compress[gr][ch];
                    u.window_switching = (uint8_t)si.window_switching[gr][ch]; u.block_type = (uint8_t)si.block_type[gr][ch];
                    u.mixed_block_flag = (uint8_t)si.mixed[gr][ch];
                    for (int r = 0; r < 3; r++) u.table_select[r] = (uint8_t)si.table_select[gr][ch][r];
                    u.region0_count = (uint8_t)si.region0[gr][ch]; u.region1_count = (uint8_t)si.region1[gr][ch];
                    u.preflag = (uint8_t)si.preflag[gr][ch]; u.scalefac_scale = (uint8_t)si.scalefac_scale[gr][ch];
                    u.count1table_select = (uint8_t)si.count1table[gr][ch];
                    for (int w = 0; w < 3; w++) u.sub_block_gain[w] = (uint8_t)si.sub_block_gain[gr][ch][w];
                    if (!scan->lean)
                        for (int r = 0; r < 3; r++)
                            out.table_select[((size_t)out.n_frames * 4 + gr * 2 + ch) * 3 + r] = si.table_select[gr][ch][r];
                }
            if (scan->n_side >= scan->side_cap && !(scan->grow && scan->grow(scan, 0, scan->n_side + 1))) return MP3S_E_NOMEM;
            scan->side[scan->n_side++] = fs;
        } else {
        // ---- per granule*channel: scalefactors + Huffman
        const size_t f = (size_t)out.n_frames;
        out.is.resize((f + 1) * 2304, 0);
        out.si.resize((f + 1) * 4);
        out.table_select.resize((f + 1) * 12, 0);
        std::memset(&out.si[f * 4], 0, 4 * sizeof(mp3s_granule_si));
        const int rc2 = decode_main_data(HT, si, nch, hd.sr_idx, mb, &out.is[f * 2304], &out.si[f * 4], &out.table_select[f * 12]);
        if (rc2) return rc2;
        }
        mp3s_frame_hdr fh;
        fh.sr_idx = (uint8_t)hd.sr_idx; fh.nch = (uint8_t)nch;
        fh.ms_stereo = (hd.mode == 1 && hd.mode_ext0) ? 1 : 0;
        fh.flags = 0; fh.stream_first = 0;
        if (scan && scan->hdr) scan->hdr[out.n_frames] = fh;   // (room for as many headers as side records)
        else out.hdr.push_back(fh);
        if (!(scan && scan->lean)) out.frame_size.push_back(frame_size);
        out.n_frames++;
        seen++;
        offset += frame_size;
    }
    out.nch = first_nch ? first_nch : hd.channels;
    out.sampling_rate = hd.sampling_rate;
    out.bit_rate = hd.bit_rate;
    if (scan) scan->gpu_ok = all_gpu_ok;
    if (record) {
        record->info.n_frames = seen; record->info.nch = out.nch; record->info.sampling_rate = hd.sampling_rate;
        record->info.bit_rate = hd.bit_rate; record->info.dup_last_frame = dup_last && seen > 0 ? 1 : 0;
    }
    return 0;
}

// One frame of a scanned stream on the host: what decode_group does with the frames the device Huffman kernel flags.
// Valid for streams the scan marked gpu_ok -- nothing in such a frame is inherited from another one.
int parse_scanned_frame(const mp3s_frame_side &fs, const uint8_t *blob, int16_t *is2304, mp3s_granule_si *si4)
{
    std::call_once(g_lut_once, build_luts);
    const HostTables &HT = host_tables();
    if (fs.nch < 1 || fs.nch > 2 || fs.sr_idx > 2) return MP3S_E_MALFORMED;
    SideInfo si;
    for (int ch = 0; ch < fs.nch; ch++)
        for (int b = 0; b < 4; b++) si.scfsi[ch][b] = fs.scfsi[ch][b];
    for (int gr = 0; gr < 2; gr++)
        for (int ch = 0; ch < fs.nch; ch++) {
            const mp3s_unit_side &u = fs.unit[gr][ch];
            si.part2_3_length[gr][ch] = u.part2_3_length; si.big_value[gr][ch] = u.big_values;
            si.global_gain[gr][ch] = u.global_gain; si.scale_fac_compress[gr][ch] = u.scalefac_compress;
            si.window_switching[gr][ch] = u.window_switching; si.block_type[gr][ch] = u.block_type;
            si.mixed[gr][ch] = u.mixed_block_flag;
            for (int r = 0; r < 3; r++) si.table_select[gr][ch][r] = u.table_select[r];
            si.region0[gr][ch] = u.region0_count; si.region1[gr][ch] = u.region1_count;
            si.preflag[gr][ch] = u.preflag; si.scalefac_scale[gr][ch] = u.scalefac_scale;
            si.count1table[gr][ch] = u.count1table_select;
            for (int w = 0; w < 3; w++) si.sub_block_gain[gr][ch][w] = u.sub_block_gain[w];
        }
    std::memset(is2304, 0, 2304 * sizeof(int16_t));
    std::memset(si4, 0, 4 * sizeof(mp3s_granule_si));
    int32_t table_select[12];
    const Bits mb{blob + fs.md_off, (long)fs.md_len};
    return decode_main_data(HT, si, fs.nch, fs.sr_idx, mb, is2304, si4, table_select);
}


// ---------------------------------------------------------------- the frame walk (device-side parse: k_parse.hpp)
namespace {
// n <= 25 bits at bit `pos` of a byte string that is readable for 8 bytes from (pos >> 3)
inline uint32_t bits_at(const uint8_t *p, int pos, int n)
{
    uint64_t w;
    std::memcpy(&w, p + (pos >> 3), 8);
    w = __builtin_bswap64(w);
    return (uint32_t)((w << (pos & 7)) >> (64 - n));
}
}  // namespace

int FrameWalker::open(const uint8_t *f, size_t len)
{
    *this = FrameWalker();
    file = f; flen = (long)len;
    // ID3v2 skip (decoder/ID3_Parser.py:95-131), as scan_core
    if (flen >= 10 && file[0] == 'I' && file[1] == 'D' && file[2] == '3' && !(file[5] & 0x0f)) {
        long size = 0;
        for (int i = 0; i < 4; i++) size = (size << 7) + file[6 + i];
        offset = size + ((file[5] >> 4) & 1 ? 20 : 10);
    }
    if (flen - offset < 2) return MP3S_E_MALFORMED;
    const uint8_t *buffer = file + offset;
    if (!(buffer[0] == 0xFF && buffer[1] >= 0xE0)) { ended = true; return 0; }   // nothing is parsed: an empty WAV at rate 0
    if (flen - offset < 4) return MP3S_E_MALFORMED;
    int rc = parse_header(hd, buffer);
    if (rc) return rc;
    // D11: set_frame_size runs once before the loop
    if (hd.sampling_rate == 0) return MP3S_E_MALFORMED;
    {
        int spf = 0;
        if (hd.layer == 3) spf = hd.version == 1 ? 1152 : 576;
        else if (hd.layer == 2) spf = 1152;
        else if (hd.layer == 1) spf = 384;
        frame_size = (int)((((double)spf / 8) * hd.bit_rate) / hd.sampling_rate);
        if (hd.padding == 1) frame_size += 1;
    }
    if (!(flen > offset + 4)) ended = true;
    return 0;
}

// eight bytes from p as one big-endian word: the fields of a side-info unit lie inside 56 bits of it
static inline uint64_t be64_at(const uint8_t *p) { uint64_t v; std::memcpy(&v, p, 8); return __builtin_bswap64(v); }

long FrameWalker::next(FrameRef *refs, long cap, uint8_t *tables4, uint32_t image_base, uint16_t stream)
{
    long n = 0;
    // (the loop's state in locals: the stores into refs[] may alias any member as far as the compiler can tell)
    long off = offset, nf = n_frames;
    int fsz = frame_size, maxp = max_p23, pv[9];
    uint32_t mdc = md_cursor;
    bool silent = any_silent;
    std::memcpy(pv, prev, sizeof pv);
    // what a parsed header leaves for the frames that repeat it
    int nch_f = hd.channels, sstart = hd.crc == 0 ? 6 : 4, constant = (hd.mode == 3 ? 21 : 36) + (hd.crc == 0 ? 2 : 0), ubase = nch_f == 2 ? 20 : 18;
    while (n < cap && !ended && !irregular) {
        const uint8_t *buffer = file + off;
        const long buflen = flen - off;
        // the state in front of this frame (decode_last resumes the byte-level scan from it): kept for the frames a stream can end
        // with regularly -- the ones inside its last bytes; a stream that ends at a bad header repeats its last frame (D12),
        // which no caller decodes here
        last_valid = buflen <= 4096;
        if (last_valid) { last_hd = hd; last_frame_size = fsz; std::memcpy(last_prev, pv, sizeof pv); last_offset = off; last_first_nch = first_nch; }
        else { __builtin_prefetch(buffer + 8 * fsz); __builtin_prefetch(buffer + 8 * fsz + 40); }   // (headers a few frames on: 38 bytes every few hundred)
        // a header that names what the one before named (everything but the padding and private bits and the second mode-extension
        // bit) leaves every field where it is: nothing to take apart
        const uint32_t key = ((uint32_t)buffer[1] << 16) | ((uint32_t)(buffer[2] & 0xfc) << 8) | (uint32_t)(buffer[3] & 0xe0);
        int padding = (buffer[2] >> 1) & 1;
        if (key != hd_key || !hd_key_ok) {
            hd_key_ok = false;
            int rc = parse_header(hd, buffer);   // (the sync was checked when the frame before was passed)
            if (rc) { irregular = true; error = rc; break; }
            // anything but MPEG-1 Layer III with a sampling rate of its own: header fields would carry over from earlier
            // frames (FrameHeader.py:125-143), false syncs are taken apart as whatever they claim -- the byte-level scan's business
            if (hd.version != 1 || hd.layer != 3 || ((buffer[2] >> 2) & 3) == 3 || hd.sr_idx < 0) { irregular = true; break; }
            fs_base = (int)(((1152.0 / 8) * hd.bit_rate) / hd.sampling_rate);
            if (fs_base + padding <= 0) { irregular = true; error = MP3S_E_MALFORMED; break; }
            if (first_nch == 0) first_nch = hd.channels;
            else if (hd.channels != first_nch) { irregular = true; error = MP3S_E_UNSUPPORTED; break; }
            hd_key = key; hd_key_ok = fs_base > 0;
            nch_f = hd.channels; sstart = hd.crc == 0 ? 6 : 4; constant = (hd.mode == 3 ? 21 : 36) + (hd.crc == 0 ? 2 : 0); ubase = nch_f == 2 ? 20 : 18;
            padding = hd.padding;
        }
        std::memmove(pv + 1, pv, 8 * sizeof pv[0]);
        pv[0] = fsz;
        fsz = fs_base + padding;
        // ---- side info: the fields the host needs, read where they lie (fixed positions: both layouts of a granule take 59 bits)
        uint8_t sbuf[48];
        const uint8_t *sb = buffer + sstart;
        if (buflen < sstart + 40) {          // the stream ends inside the side info: bits past the end read as 0
            std::memset(sbuf, 0, sizeof sbuf);
            if (buflen > sstart) std::memcpy(sbuf, buffer + sstart, (size_t)std::min<long>(buflen - sstart, 40));
            sb = sbuf;
        }
        const int mdb = (int)(((uint32_t)sb[0] << 1) | (sb[1] >> 7));
        for (int u = 0; u < 2 * nch_f; u++) {
            const int b = ubase + 59 * u;
            const uint64_t x = be64_at(sb + (b >> 3)) << (b & 7);   // part2_3_length (12) | big_values (9) | 12 bits | window_switching + 15
            maxp = std::max(maxp, (int)(x >> 52));
            // a granule without a code book in use (no big values, or book 0 in every region it has): what tables_guess_of calls 0
            const uint32_t wt = (uint32_t)(x >> 15) & 0xffffu;      // window_switching | the 15 bits behind it
            const uint32_t books = (wt & 0x8000u) ? (wt >> 2) & 0x3ffu : wt & 0x7fffu;
            if (((x >> 43) & 0x1ff) == 0 || books == 0) silent = true;
        }
        if (tables4 && nch_f == 2 && tables_frames == nf && tables_seen < tables_wanted) {
            for (int gr = 0; gr < 2; gr++)
                for (int ch = 0; ch < 2; ch++) {
                    const int b = ubase + 59 * (gr * 2 + ch);
                    int t = 0;
                    if (bits_at(sb, b + 12, 9)) {
                        if (bits_at(sb, b + 33, 1)) t = (bits_at(sb, b + 37, 5) != 0) + (bits_at(sb, b + 42, 5) != 0);
                        else t = (bits_at(sb, b + 34, 5) != 0) + (bits_at(sb, b + 39, 5) != 0) + (bits_at(sb, b + 44, 5) != 0);
                    }
                    tables4[(size_t)n * 4 + ch * 2 + gr] = (uint8_t)t;
                    tables_seen += t;
                }
            tables_frames++;
        }
        // ---- main data: how long it is (Frame.py:318-363); every part must lie inside the file where the pointers say
        long md_len = std::max<long>(0, std::min<long>(fsz, buflen) - constant);
        if (mdb != 0) {
            long bound = 0;
            int fr = 0;
            for (; fr < 9; fr++) {
                const long part = (long)pv[fr] - constant;
                if (mdb < bound + part) break;
                if (part < 0) { fr = 9; break; }
                bound += part;
            }
            // not found: the reference goes on with the main data of the frame before; a pointer in front of the file wraps
            // around (Python slices) -- both are the byte-level scan's business
            if (fr >= 9 || off - mdb - (long)fr * constant < 0) { irregular = true; break; }
            md_len += mdb;
        }
        if (md_len > 0xffff || fsz > 0xffff || (uint64_t)off + image_base > 0xffffffffull || (uint64_t)mdc + (uint64_t)md_len + 16 > 0xffffffffull) {
            irregular = true; break;
        }
        FrameRef &r = refs[n];
        r.file_off = (uint32_t)off + image_base; r.md_off = mdc; r.md_len = (uint16_t)md_len; r.frame_size = (uint16_t)fsz;
        r.stream = stream; r.flags = 0;
        mdc = (uint32_t)((mdc + md_len + 8 + 3) & ~(uint32_t)3);
        n++; nf++;
        off += fsz;
        if (!(flen > off + 4)) ended = true;
        else if (!(file[off] == 0xFF && file[off + 1] >= 0xE0)) { ended = true; dup_last = true; }   // D12
    }
    // (a frame that made the walk give up has moved nothing: its predecessor's state stands, as before)
    offset = off; n_frames = nf; md_cursor = mdc; max_p23 = maxp; any_silent = silent;
    if (!irregular) { frame_size = fsz; std::memcpy(prev, pv, sizeof pv); hd.padding = fsz - fs_base; }
    nch = first_nch ? first_nch : hd.channels;
    sampling_rate = hd.sampling_rate; bit_rate = hd.bit_rate;
    return n;
}

void FrameWalker::history(const FrameRef *stream_refs, long f, uint16_t out[9])
{
    // prev[i] at frame f = the size of frame f - 1 - i; in front of frame 0 stands frame 0's own size once (D11), zeros before it
    for (int i = 0; i < 9; i++) {
        const long g = f - 1 - i;
        out[i] = g >= 0 ? stream_refs[g].frame_size : (g == -1 && f >= 0 ? stream_refs[0].frame_size : 0);
    }
}

int FrameWalker::decode_last(int16_t *is2304, mp3s_granule_si *si4, bool *alone)
{
    if (n_frames <= 0 || !last_valid) return MP3S_E_ARG;
    ScanState st;
    st.hd = last_hd; st.frame_size = last_frame_size; st.first_nch = last_first_nch; st.offset = last_offset;
    for (int i = 0; i < 9; i++) st.prev_frame_size[i] = last_prev[i];
    uint8_t blob[4096];
    mp3s_frame_side side[1];
    ScanSink k;
    k.blob = blob; k.blob_cap = sizeof blob; k.side = side; k.side_cap = 1; k.lean = true;
    ParsedStream tmp;
    const int rc = scan_core(file, (size_t)flen, tmp, &k, &st, 0, 1, nullptr);
    if (rc) return rc;
    if (k.n_side != 1) return MP3S_E_MALFORMED;
    if (alone) *alone = k.gpu_ok;
    return parse_scanned_frame(side[0], blob, is2304, si4);
}

void stego_bits_from_tsel(const uint64_t *tsel, long n_frames, int nch, uint8_t carry[4], std::vector<uint8_t> &bits)
{
    const HostTables &HT = host_tables();
    for (long f = 0; f < n_frames; f++) {
        const uint64_t w = tsel[f];
        for (int ch = 0; ch < nch; ch++)
            for (int gr = 0; gr < 2; gr++) {
                const int cls = ch * 2 + gr;
                const bool ws = (w >> (60 + cls)) & 1;
                for (int r = 0; r < 3; r++) {
                    int t = (int)((w >> (5 * (cls * 3 + r))) & 31);
                    if (r == 2) { if (ws) t = carry[cls]; else carry[cls] = (uint8_t)t; }
                    if (t) bits.push_back(HT.in_h0[t] ? 0 : 1);
                }
            }
    }
}

}  // namespace mp3s
