// Host back end of the encoder: the serial bit-packing stage that stays on the CPU by design
// (SURVEY.md section 8 row a18).  Consumes what the rate-loop kernel produced (signed ix, GrInfo per
// granule*channel, scfsi) and emits the MP3 byte stream exactly as the reference does:
//   padding / slot lag replay   encoder/MP3_Encoder.py:503-513, 630-636
//   __resv_frame_end            :1097-1145 (all slack becomes stuffing, E6)
//   __encode_side_info          :1281-1337 (E12)
//   __encode_main_data          :1339-1360, __huffman_code_bits :1394-1446, __huffman_code :1448-1513,
//   __huffman_coder_count1      :1515-1547 (E13)
//   __put_bits / __flush        :1362-1392, :1549-1552 (32-bit cache, tail dropped: E14)
#include "mp3s_host.h"

#include <cmath>
#include <cstring>

namespace mp3s {

namespace {

const int kBitratesV1[16] = {-1, 32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320, -1};

int samplerate_index(int sr) { return sr == 44100 ? 0 : sr == 48000 ? 1 : sr == 32000 ? 2 : -1; }
int bitrate_index(int kbps)
{
    for (int i = 0; i < 16; i++) if (kBitratesV1[i] == kbps) return i;
    return -1;
}

struct SlotLag {           // reference MPEG fields whole_slots_per_frame / frac_slots_per_frame / slot_lag
    int whole; double frac, lag; int padding;
    SlotLag(int samplerate, int kbps)
    {
        const double avg = ((double)2 * 576 / ((double)samplerate)) * (1000 * (double)kbps / (double)8);
        whole = (int)avg; frac = avg - (double)whole; lag = -frac; padding = 0;
    }
    int next()             // returns bits_per_frame
    {
        if (frac) { padding = lag <= (frac - 1.0) ? 1 : 0; lag += padding - frac; }
        return 8 * (whole + padding);
    }
};

struct BitWriter {         // reference BitstreamStruct + __put_bits
    std::vector<uint8_t> &out;
    uint32_t cache = 0; int cache_bits = 32;
    explicit BitWriter(std::vector<uint8_t> &o) : out(o) {}
    void put(uint32_t val, int n)
    {
        if (cache_bits > n) {
            cache_bits -= n;
            if (cache_bits < 32) cache |= val << cache_bits;
        } else {
            n -= cache_bits;
            cache |= n < 32 ? (val >> n) : 0;
            out.push_back((uint8_t)(cache >> 24)); out.push_back((uint8_t)(cache >> 16));
            out.push_back((uint8_t)(cache >> 8)); out.push_back((uint8_t)cache);
            cache_bits = 32 - n;
            cache = n ? (val << cache_bits) : 0;
        }
    }
    long count() const { return (long)out.size() * 8 + 32 - cache_bits; }
};

}  // namespace

int stream_params(int samplerate, int bitrate_kbps, int *sri, int *bri, int *whole_slots)
{
    const int s = samplerate_index(samplerate), b = bitrate_index(bitrate_kbps);
    if (s < 0 || b < 0) return 1;
    SlotLag sl(samplerate, bitrate_kbps);
    *sri = s; *bri = b; *whole_slots = sl.whole;
    return 0;
}

int rate_frames(int samplerate, int bitrate_kbps, int nch, int n_frames, mp3s_rate_frame *out, int32_t *padding,
                int64_t first_frame, int64_t *bytes_before)
{
    const int sri = samplerate_index(samplerate);
    if (sri < 0 || bitrate_index(bitrate_kbps) < 0 || nch < 1 || nch > 2 || first_frame < 0) return MP3S_E_UNSUPPORTED;
    SlotLag sl(samplerate, bitrate_kbps);
    const int side_info_len = 8 * (nch == 1 ? 4 + 17 : 4 + 32);
    int64_t before = 0;
    for (int64_t f = 0; f < first_frame; f++) before += sl.next() / 8;   // the padding recurrence has no closed form in fp64
    if (bytes_before) *bytes_before = before;
    for (int f = 0; f < n_frames; f++) {
        const int bits_per_frame = sl.next();
        const int mean_bits = (int)((double)(bits_per_frame - side_info_len) / 2);
        int mb = mean_bits / nch;                     // resv_max == 0: MP3_Encoder.py:904-912
        if (mb > 4095) mb = 4095;
        out[f].max_bits = mb; out[f].sr_idx = sri; out[f].hide_end = 0x7fffffff; out[f].stream = 0;
        if (padding) padding[f] = sl.padding;
    }
    return 0;
}

void host_scfsi_energies(const int32_t *xr, int sr_idx, int32_t *en)
{
    const int32_t *sfb = host_tables().dev.sfb_long[sr_idx];
    uint32_t e10[576], tot = 0;
    for (int i = 0; i < 576; i++) {
        const int64_t a = xr[i];
        const int32_t sq = (int32_t)((a * a + (1ll << 30)) >> 31);   // util.mulsr(xr, xr)
        e10[i] = (uint32_t)(sq >> 10);
        tot += e10[i];
    }
    auto lg = [](uint32_t t) -> int32_t {
        const int32_t v = (int32_t)t;
        return v ? (int32_t)(std::log((double)v * 4.768371584e-7) / 0.69314718) : 0;
    };
    for (int b = 0; b < 21; b++) {
        uint32_t t = 0;
        for (int i = sfb[b]; i < sfb[b + 1]; i++) t += e10[i];
        en[b] = lg(t);
    }
    en[21] = lg(tot);
}

void decide_scfsi(int n_frames, const int32_t *en, const mp3s_gr_out *gr, int32_t *scfsi)
{
    static const int band[5] = {0, 6, 11, 16, 21};
    for (int f = 0; f < n_frames; f++)
        for (int ch = 0; ch < 2; ch++) {
            const int u0 = (f * 2 + ch) * 2, u1 = u0 + 1;
            const int32_t *e0 = en + (size_t)u0 * 22, *e1 = en + (size_t)u1 * 22;
            int cond = 2;
            if (gr[u0].xrmax) cond++;
            if (gr[u1].xrmax) cond++;
            if (std::abs(e0[21] - e1[21]) < 10) cond++;
            int tp = 0;
            for (int s = 0; s < 21; s++) tp += std::abs(e0[s] - e1[s]);
            if (tp < 100) cond++;
            for (int b = 0; b < 4; b++) {
                int v = 0;
                if (cond == 6) {
                    int sum0 = 0;
                    for (int s = band[b]; s < band[b + 1]; s++) sum0 += std::abs(e0[s] - e1[s]);
                    v = sum0 < 10 ? 1 : 0;            // xm is all zero: sum1 == 0 < 10
                }
                scfsi[(f * 2 + ch) * 4 + b] = v;
            }
        }
}

int format_stream(int samplerate, int bitrate_kbps, int n_frames, const int16_t *ix, const mp3s_gr_out *gr_in,
                  const int32_t *scfsi, std::vector<uint8_t> &mp3)
{
    const HostTables &HT = host_tables();
    const int sri = samplerate_index(samplerate), bri = bitrate_index(bitrate_kbps);
    if (sri < 0 || bri < 0) return MP3S_E_UNSUPPORTED;
    const int nch = 2;
    SlotLag sl(samplerate, bitrate_kbps);
    const int side_info_len = 8 * (4 + 32);
    mp3.clear();
    mp3.reserve((size_t)n_frames * (size_t)(sl.whole + 1) + 16);
    BitWriter bw(mp3);
    const int32_t *sfb = HT.dev.sfb_long[sri];
    for (int f = 0; f < n_frames; f++) {
        const int bits_per_frame = sl.next();
        const int mean_bits = (int)((double)(bits_per_frame - side_info_len) / 2);
        // unit u = (f*2 + ch)*2 + gr
        int p23[2][2];
        for (int grn = 0; grn < 2; grn++)
            for (int ch = 0; ch < nch; ch++) p23[grn][ch] = gr_in[((size_t)f * 2 + ch) * 2 + grn].part2_3_length;
        // ---- __resv_frame_end: resv_size accumulates mean_bits/nch - part2_3_length over the four units
        double resv = 0;
        for (int ch = 0; ch < nch; ch++)
            for (int grn = 0; grn < 2; grn++) resv += ((double)mean_bits / nch) - p23[grn][ch];
        if (nch == 2 && (mean_bits & 1)) resv += 1;
        double over = resv; if (over < 0) over = 0;
        resv -= over;
        double stuffing = over;
        double rem = std::fmod(resv, 8); if (rem < 0) rem += 8;
        if (rem) { stuffing += rem; resv -= rem; }
        if (stuffing) {
            if (p23[0][0] + stuffing < 4095) p23[0][0] = (int)(p23[0][0] + stuffing);
            else
                for (int grn = 0; grn < 2; grn++)
                    for (int ch = 0; ch < nch; ch++) {
                        if (!stuffing) break;
                        const double extra = 4095 - p23[grn][ch];
                        const double now = extra < stuffing ? extra : stuffing;
                        p23[grn][ch] = (int)(p23[grn][ch] + now);
                        stuffing -= now;
                    }
        }
        // ---- header + side info
        bw.put(0x7ff, 11); bw.put(3, 2); bw.put(1, 2); bw.put(1, 1); bw.put(bri, 4); bw.put(sri % 3, 2);
        bw.put(sl.padding, 1); bw.put(0, 1); bw.put(0 /* stereo */, 2); bw.put(0, 2); bw.put(0, 1); bw.put(1, 1);
        bw.put(0, 2);
        bw.put(0, 9); bw.put(0, 3);
        for (int ch = 0; ch < nch; ch++)
            for (int b = 0; b < 4; b++) bw.put(scfsi[((size_t)f * 2 + ch) * 4 + b], 1);
        for (int grn = 0; grn < 2; grn++)
            for (int ch = 0; ch < nch; ch++) {
                const mp3s_gr_out &g = gr_in[((size_t)f * 2 + ch) * 2 + grn];
                bw.put(p23[grn][ch], 12); bw.put(g.big_values, 9); bw.put(g.quantizer_step + 210, 8);
                bw.put(0, 4); bw.put(0, 1);
                for (int r = 0; r < 3; r++) bw.put(g.table_select[r], 5);
                bw.put(g.region0_count, 4); bw.put(g.region1_count, 3);
                bw.put(0, 1); bw.put(0, 1); bw.put(g.count1table_select, 1);
            }
        // ---- main data (scalefactors are all zero-width)
        for (int grn = 0; grn < 2; grn++)
            for (int ch = 0; ch < nch; ch++) {
                const size_t u = ((size_t)f * 2 + ch) * 2 + grn;
                const mp3s_gr_out &g = gr_in[u];
                const int16_t *x = ix + u * 576;
                const long start = bw.count();
                const int big_values = g.big_values << 1;
                int sfi = g.region0_count + 1;
                const int region1_start = sfb[sfi];
                sfi += g.region1_count + 1;
                const int region2_start = sfb[sfi];
                for (int i = 0; i < big_values; i += 2) {
                    const int ti = g.table_select[(i >= region1_start) + (i >= region2_start)];
                    if (!ti) continue;
                    int xv = x[i], yv = x[i + 1];
                    const int sx = xv > 0 ? 0 : 1, sy = yv > 0 ? 0 : 1;   // util.abs_and_sign: 0 has "sign" 1
                    if (xv < 0) xv = -xv;
                    if (yv < 0) yv = -yv;
                    const HostHuff &h = HT.huff[ti];
                    if (ti > 15) {
                        uint32_t ext = 0; int xbits = 0, lbx = 0, lby = 0;
                        if (xv > 14) { lbx = xv - 15; xv = 15; }
                        if (yv > 14) { lby = yv - 15; yv = 15; }
                        const int idx = xv * h.ylen + yv;
                        if (xv > 14) { ext |= lbx; xbits += h.linbits; }
                        if (xv != 0) { ext <<= 1; ext |= sx; xbits += 1; }
                        if (yv > 14) { ext <<= h.linbits; ext |= lby; xbits += h.linbits; }
                        if (yv != 0) { ext <<= 1; ext |= sy; xbits += 1; }
                        bw.put(h.hcod[idx], h.hlen[idx]);
                        bw.put(ext, xbits);
                    } else {
                        const int idx = xv * h.ylen + yv;
                        uint32_t code = h.hcod[idx]; int cbits = h.hlen[idx];
                        if (xv != 0) { code = (code << 1) | sx; cbits += 1; }
                        if (yv != 0) { code = (code << 1) | sy; cbits += 1; }
                        bw.put(code, cbits);
                    }
                }
                const HostHuff &q = HT.huff[32 + g.count1table_select];
                const int count1_end = big_values + (g.count1 << 2);
                for (int i = big_values; i < count1_end; i += 4) {
                    int v[4], s[4];
                    for (int k = 0; k < 4; k++) { v[k] = x[i + k]; s[k] = v[k] > 0 ? 0 : 1; if (v[k] < 0) v[k] = -v[k]; }
                    const int p = v[0] + (v[1] << 1) + (v[2] << 2) + (v[3] << 3);
                    bw.put(q.hcod[p], q.hlen[p]);
                    uint32_t code = 0; int cbits = 0;
                    for (int k = 0; k < 4; k++)
                        if (v[k]) { code = (code << 1) | s[k]; cbits += 1; }
                    bw.put(code, cbits);
                }
                long bits = p23[grn][ch] - (bw.count() - start);
                if (bits < 0) return MP3S_E_MALFORMED;
                for (long wds = bits / 32; wds; wds--) bw.put(0xffffffffu, 32);
                if (bits % 32) bw.put((uint32_t)((1ull << (bits % 32)) - 1), (int)(bits % 32));
            }
    }
    return 0;   // the cached tail (< 32 bits) is dropped, as __flush does
}

}  // namespace mp3s
