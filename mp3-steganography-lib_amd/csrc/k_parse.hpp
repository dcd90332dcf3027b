// Side-info parse and main-data gather on the device (gfx950).  Included by mp3s_device.hip only.
//
//   k_dec_parse : the byte-level half of the reference's frame loop for every frame of the batch -- the 17 / 32 bytes of
//                 side info taken apart field by field (decoder/FrameSideInformation.py:39-137) and the frame's main data
//                 collected from where the bit reservoir left it (decoder/Frame.py:318-363: up to nine frames back, the
//                 header and side-info bytes in between skipped).  The host keeps the walk from header to header (frame
//                 sizes are a recurrence on the headers: FrameWalker, mp3s_host.h) and hands over 16 bytes per frame; the
//                 file image itself goes up as it is.  Output = exactly what the byte-level scan on the host produces
//                 (mp3s_scan_stream): mp3s_frame_side records, the main-data blob (4-byte aligned, eight zero bytes behind
//                 every frame), mp3s_frame_hdr -- except for the two fields a granule does not parse itself and keeps from
//                 the frame before (table_select[2] of a window-switching granule, sub_block_gain of any other):
//                 nothing on the device reads them (region 2 of such a granule is empty, the gains belong to short
//                 windows), they are written as zero here, and the one place they show -- the stego bits -- is served by
//                 `tsel` and a serial pass on the host (stego_bits_from_tsel).
//                 One wavefront per frame: lanes 0..3 parse one granule*channel each (both layouts of a granule take 59
//                 bits, so every field sits at a fixed position), all 64 lanes copy the main data, a dword of output each.
#pragma once

namespace mp3s {

constexpr int PARSE_WAVES = 4;
constexpr int PARSE_INHERITS = 1;   // some frame inherits scalefactors across frames (mp3s_scanned.gpu_ok == 0)
constexpr int PARSE_MISMATCH = 2;   // the gather came out another length than the walk said: the host's scan decides

struct ParseFrameRef { uint32_t file_off, md_off; uint16_t md_len, frame_size, stream, flags; };
struct ParseStreamRef { uint32_t base, end, first_frame, n_frames; uint16_t prev_size[9]; uint16_t side_back[2]; uint16_t reserved; };

__global__ __launch_bounds__(PARSE_WAVES * 64) void k_dec_parse(
    const uint8_t *__restrict__ image, uint32_t image_base /* image[0] is byte `image_base` of what file_off / base / end count in */,
    const ParseFrameRef *__restrict__ refs, const ParseStreamRef *__restrict__ streams, int n_frames, uint32_t md_base,
    mp3s_frame_side *__restrict__ side, mp3s_frame_hdr *__restrict__ hdr, uint8_t *__restrict__ blob, uint64_t *__restrict__ tsel,
    int32_t *__restrict__ status)
{
    __shared__ uint8_t hb[PARSE_WAVES][64];
    __shared__ uint32_t sg_src[PARSE_WAVES][10], sg_len[PARSE_WAVES][10];   // the parts of the frame's main data (lane-indexed below)
    const int wave = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
    const int f = (int)blockIdx.x * PARSE_WAVES + wave;
    if (f >= n_frames) return;
    const ParseFrameRef ref = refs[f];
    const ParseStreamRef *st = streams + ref.stream;
    const uint32_t s_end = st->end, s_first = st->first_frame;
    const uint8_t *img = image - image_base;                      // indexed by the offsets the records hold
    const uint32_t off = ref.file_off;
    const uint32_t avail = s_end > off ? s_end - off : 0u;        // bytes of the stream from the frame's header on (buflen)
    // ---- header, CRC, side info: 40 bytes at most, bytes past the end of the stream read as zero
    hb[wave][lane] = lane < 48 && (uint32_t)lane < avail ? img[off + lane] : (uint8_t)0;
    __builtin_amdgcn_wave_barrier();
    const uint8_t *h = hb[wave];
    const uint32_t b1 = h[1], b2 = h[2], b3 = h[3];
    const uint32_t crc = b1 & 1u, mode = b3 >> 6, nch = mode == 3 ? 1u : 2u, sr_idx = (b2 >> 2) & 3u;
    const uint32_t ms = (mode == 1 && (b3 & 0x20u)) ? 1u : 0u;
    const uint8_t *sb = h + (crc == 0 ? 6 : 4);
    auto get = [&](int pos, int n) -> uint32_t {                  // n <= 12 bits at bit `pos` of the side info
        const int b = pos >> 3;
        const uint32_t w = ((uint32_t)sb[b] << 16) | ((uint32_t)sb[b + 1] << 8) | sb[b + 2];
        return (w >> (24 - (pos & 7) - n)) & ((1u << n) - 1u);
    };
    const uint32_t mdb = get(0, 9);
    const int priv = nch == 2 ? 3 : 5;
    // ---- one granule*channel per lane (lanes 0..3: gr = lane >> 1, ch = lane & 1)
    const int gr = (lane >> 1) & 1, ch = lane & 1;
    const bool unit = lane < 4 && (uint32_t)ch < nch;
    uint32_t u0 = 0, u1 = 0, u2 = 0, u3 = 0, u4 = 0, ws = 0, bt = 0, mixed = 0, ts0 = 0, ts1 = 0, ts2 = 0;
    if (unit) {
        const int base = 9 + priv + 4 * (int)nch + 59 * (gr * (int)nch + ch);
        const uint32_t p23 = get(base, 12), bv = get(base + 12, 9), gg = get(base + 21, 8), sfc = get(base + 29, 4);
        ws = get(base + 33, 1);
        uint32_t r0, r1, g0 = 0, g1 = 0, g2 = 0;
        if (ws) {
            bt = get(base + 34, 2); mixed = get(base + 36, 1);
            ts0 = get(base + 37, 5); ts1 = get(base + 42, 5);
            g0 = get(base + 47, 3); g1 = get(base + 50, 3); g2 = get(base + 53, 3);
            r0 = bt == 2 ? 8u : 7u; r1 = 20u - r0;
        } else {
            ts0 = get(base + 34, 5); ts1 = get(base + 39, 5); ts2 = get(base + 44, 5);
            r0 = get(base + 49, 4); r1 = get(base + 53, 3);
        }
        const uint32_t pre = get(base + 56, 1), sfs = get(base + 57, 1), c1 = get(base + 58, 1);
        u0 = p23 | (bv << 16);
        u1 = gg | (sfc << 8) | (ws << 16) | (bt << 24);
        u2 = mixed | (ts0 << 8) | (ts1 << 16) | (ts2 << 24);
        u3 = r0 | (r1 << 8) | (pre << 16) | (sfs << 24);
        u4 = c1 | (g0 << 8) | (g1 << 16) | (g2 << 24);
    }
    // scfsi, one byte per band: lane 0 -> channel 0, lane 1 -> channel 1
    uint32_t scf = 0;
    if (lane < 2 && (uint32_t)lane < nch)
        for (int b = 0; b < 4; b++) scf |= get(9 + priv + 4 * lane + b, 1) << (8 * b);
    // scalefactors that requantisation would read without this frame having written them (mixed blocks, scfsi behind a
    // short granule 0: SURVEY D10) -- the same test as the host scan's gpu_ok
    const bool short_win = ws && bt == 2;
    const unsigned long long m_mixed = __ballot(unit && ws && mixed), m_short = __ballot(unit && short_win), m_scf = __ballot(scf != 0);
    bool inherits = (m_mixed & 15ull) != 0;
    for (int c = 0; c < 2; c++)
        if (((m_short >> c) & 1ull) && !((m_short >> (2 + c)) & 1ull) && ((m_scf >> c) & 1ull)) inherits = true;
    // ---- the records
    uint32_t *d = reinterpret_cast<uint32_t *>(side + f);
    if (lane < 4) {
        uint32_t *ud = d + 5 + 5 * lane;
        ud[0] = u0; ud[1] = u1; ud[2] = u2; ud[3] = u3; ud[4] = u4;
    }
    if (lane < 2) d[3 + lane] = scf;
    if (lane == 0) {
        d[0] = ref.md_off - md_base; d[1] = ref.md_len;
        d[2] = nch | (sr_idx << 8) | (ms << 16) | ((uint32_t)(ref.flags & 0xff) << 24);
        d[25] = s_first - ((uint32_t)st->side_back[0] | ((uint32_t)st->side_back[1] << 16));   // (int32: negative when records of the stream lie in front of side[0])
        uint32_t *hd = reinterpret_cast<uint32_t *>(hdr + f);
        hd[0] = sr_idx | (nch << 8) | (ms << 16);
        hd[1] = s_first;
        if (inherits) atomicOr(status, PARSE_INHERITS);
    }
    if (tsel) {
        // table indices in the order the stego bits walk them (channel, granule, region), window-switching flags on top
        const int cls = ch * 2 + gr;
        uint64_t w = unit ? ((uint64_t)(ts0 | (ts1 << 5) | (ts2 << 10)) << (15 * cls)) | ((uint64_t)ws << (60 + cls)) : 0ull;
        w |= __shfl_xor(w, 1, 64);
        w |= __shfl_xor(w, 2, 64);
        if (lane == 0) tsel[f] = w;
    }
    // ---- main data (Frame.py:318-363): the parts the reservoir pointer names, oldest first, then the frame's own
    const uint32_t constant = (mode == 3 ? 21u : 36u) + (crc == 0 ? 2u : 0u);
    const uint32_t fsz = ref.frame_size;
    const uint32_t own_end = fsz < avail ? fsz : avail;
    const uint32_t own = own_end > constant ? own_end - constant : 0u;
    uint32_t *seg_src = sg_src[wave], *seg_len = sg_len[wave];   // (every lane computes and writes the same values)
    int n_seg = 0;
    uint32_t total = 0;
    if (mdb != 0) {
        // Frame.__prev_frame_size in front of this frame: the frames before it in the batch, then what the stream record holds
        uint32_t prev[9];
        const uint32_t k = (uint32_t)f - s_first;              // frames of the stream in front of this one in the batch
#pragma unroll
        for (int i = 0; i < 9; i++) prev[i] = (uint32_t)i < k ? refs[f - 1 - i].frame_size : st->prev_size[(uint32_t)i - k];
        uint32_t bound = 0;
        int fr = 0;
        for (; fr < 9; fr++) {
            const uint32_t part = prev[fr] > constant ? prev[fr] - constant : 0u;
            if (mdb < bound + part) break;
            bound += part;
        }
        if (fr < 9) {
            uint32_t ptr = mdb + (uint32_t)fr * constant;       // distance of the oldest part from this frame's header
            const uint32_t first_len = mdb - bound;
            seg_src[0] = off - ptr; seg_len[0] = first_len; n_seg = 1;
            ptr -= first_len + constant;
            for (int i = fr - 1; i >= 0; i--) {
                const uint32_t part = prev[i] - constant;
                seg_src[n_seg] = off - ptr; seg_len[n_seg] = part; n_seg++;
                ptr -= part + constant;
            }
            total = mdb;
        }
    }
    seg_src[n_seg] = off + constant; seg_len[n_seg] = own; n_seg++;
    total += own;
    __builtin_amdgcn_wave_barrier();
    const uint32_t md_len = ref.md_len;
    if (total != md_len && lane == 0) atomicOr(status, PARSE_MISMATCH);
    uint32_t *out = reinterpret_cast<uint32_t *>(blob + (ref.md_off - md_base));
    const uint32_t n_dw = (md_len + 8 + 3) >> 2;               // the zero bytes up to the next frame's aligned start included
    for (uint32_t dw = (uint32_t)lane; dw < n_dw; dw += 64) {
        uint32_t v = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            uint32_t p = dw * 4 + (uint32_t)j;
            if (p < md_len && p < total) {
                int s = 0;
                while (s < n_seg - 1 && p >= seg_len[s]) { p -= seg_len[s]; s++; }
                v |= (uint32_t)img[seg_src[s] + p] << (8 * j);
            }
        }
        out[dw] = v;
    }
}

}  // namespace mp3s
