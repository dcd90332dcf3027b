// Huffman / scalefactor decode on the device (gfx950).  Included by mp3s_device.hip only.
//
//   k_dec_huffman : __unpack_scale_fac + __unpack_samples (reference decoder/Frame.py:365-559) for every
//                   granule*channel of the batch.  The bit stream of a granule is serial, but granule boundaries
//                   come from the side info (part2_3_length), so one THREAD decodes one granule*channel (granule 1
//                   re-reads the scalefactors scfsi lets it share with granule 0 from granule 0's own bits): a
//                   10 000-frame batch gives 40 000 independent threads.  Code books: 10-bit first-level table in LDS, binary
//                   trie in global memory for the rare longer codes (same prefix codes the reference searches
//                   linearly, so the same symbol and length come out).  Quirks kept: D1 (count1 stops at line 572,
//                   no overrun discard), D2 (books 4/14 read no bits), bits past the buffer read as 0.
#pragma once

namespace mp3s {

constexpr int HUF_THREADS = 64;
constexpr int HUF_WORDS = 132;   // 4095 bits of part2_3_length + alignment slack, per thread

// global-memory word fetch: big-endian, bytes past `len` read as zero (decoder/util.py:41-43)
__device__ __forceinline__ uint32_t md_word(const uint32_t *w, uint32_t len, uint32_t i)
{
    const uint32_t byte0 = i * 4;
    if (byte0 >= len) return 0;
    uint32_t v = __builtin_bswap32(w[i]);
    const uint32_t valid = len - byte0;          // 1..3: mask the bytes past the end
    if (valid < 4) v &= 0xffffffffu << (8 * (4 - valid));
    return v;
}
__device__ __forceinline__ uint32_t md_get(const uint32_t *w, uint32_t len, uint32_t pos, int n)
{
    if (!n) return 0;
    const uint32_t i = pos >> 5, sh = pos & 31;
    const uint32_t w0 = md_word(w, len, i), w1 = md_word(w, len, i + 1);
    return (sh ? (w0 << sh) | (w1 >> (32 - sh)) : w0) >> (32 - n);
}

// Bit reader over the lane's column of LDS: the words covering this granule's bits are copied there up front, so the
// decode loop touches no global memory except its (never waited for) stores.
struct BitReader {
    const uint32_t *lds;   // word j of this lane at lds[j * HUF_THREADS]
    uint32_t base;         // index of the first staged word
    __device__ __forceinline__ uint64_t peek64(uint32_t pos) const
    {
        uint32_t j = (pos >> 5) - base;
        const uint32_t sh = pos & 31;
        j = j < HUF_WORDS - 3 ? j : HUF_WORDS - 3;   // malformed streams may run past part2_3_length: stay inside the column
        const uint32_t w0 = lds[j * HUF_THREADS], w1 = lds[(j + 1) * HUF_THREADS], w2 = lds[(j + 2) * HUF_THREADS];
        const uint32_t hi = sh ? (w0 << sh) | (w1 >> (32 - sh)) : w0;
        const uint32_t lo = sh ? (w1 << sh) | (w2 >> (32 - sh)) : w1;
        return ((uint64_t)hi << 32) | lo;
    }
    __device__ __forceinline__ uint32_t peek32(uint32_t pos) const { return (uint32_t)(peek64(pos) >> 32); }
    __device__ __forceinline__ uint32_t get(uint32_t pos, int n) const { return n ? peek32(pos) >> (32 - n) : 0u; }
};


// one thread per granule*channel; unit index = (frame*2 + gr)*2 + ch (same order as the si / is arrays)
__global__ __launch_bounds__(HUF_THREADS) void k_dec_huffman(
    const uint8_t *__restrict__ blob, const mp3s_frame_side *__restrict__ side, int n_frames, int nch, int active,
    int16_t *__restrict__ is, mp3s_granule_si *__restrict__ si_out, int32_t *__restrict__ status)
{
    __shared__ uint16_t fast[15][1024];
    __shared__ uint16_t quad[64];
    __shared__ uint32_t words[HUF_WORDS * HUF_THREADS];
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(&c_tab.huff_fast[0][0]);
        uint4 *dst = reinterpret_cast<uint4 *>(&fast[0][0]);
#pragma unroll 6
        for (int i = threadIdx.x; i < 15 * 1024 * 2 / 16; i += HUF_THREADS) dst[i] = src[i];
        quad[threadIdx.x & 63] = c_tab.quad_fast[threadIdx.x & 63];
    }
    __syncthreads();
    // only `active` lanes of the wave decode: every lane walks its own bit stream, so a wave runs as long as its
    // slowest lane and pays every lane's branches; with few waves to spare, narrower waves finish sooner
    const long tid0 = (long)blockIdx.x * active;
    const long tid = tid0 + threadIdx.x;
    const bool worker = (int)threadIdx.x < active && tid < (long)n_frames * 4 && (int)(tid & 1) < nch;
    if (worker) {
    const int f = (int)(tid >> 2), gr = (int)((tid >> 1) & 1), ch = (int)(tid & 1);
    const mp3s_frame_side *fs = side + f;
    static const uint8_t kSlen[16][2] = {{0, 0}, {0, 1}, {0, 2}, {0, 3}, {3, 0}, {1, 1}, {1, 2}, {1, 3},
                                         {2, 1}, {2, 2}, {2, 3}, {3, 1}, {3, 2}, {3, 3}, {4, 2}, {4, 3}};
    const int sr = fs->sr_idx < 3 ? fs->sr_idx : 0;
    // bit offset of this unit and (for scfsi) of granule 0 of the same channel: units are laid out gr-major
    uint32_t bit = 0, bit_g0 = 0;
    for (int g2 = 0; g2 < 2; g2++)
        for (int c = 0; c < nch; c++) {
            if (g2 == 0 && c < ch) bit_g0 += fs->unit[0][c].part2_3_length;
            if (g2 * 2 + c < gr * 2 + ch) bit += fs->unit[g2][c].part2_3_length;
        }
    const mp3s_unit_side &u = fs->unit[gr][ch];
    const uint32_t max_bit = bit + u.part2_3_length;
    int err = 0;
    const uint32_t *mdw = reinterpret_cast<const uint32_t *>(blob + fs->md_off);
    const uint32_t md_len = fs->md_len;
    BitReader br;
    br.lds = words + threadIdx.x;
    br.base = bit >> 5;
    {
        const uint32_t nw = ((max_bit + 63) >> 5) - br.base + 3;
        for (uint32_t j = 0; j < nw && j < HUF_WORDS; j++) words[j * HUF_THREADS + threadIdx.x] = md_word(mdw, md_len, br.base + j);
    }
    uint8_t *g = reinterpret_cast<uint8_t *>(si_out + tid);
    g[0] = u.global_gain; g[1] = u.scalefac_scale; g[2] = u.block_type; g[3] = u.mixed_block_flag; g[4] = u.preflag;
    g[5] = u.sub_block_gain[0]; g[6] = u.sub_block_gain[1]; g[7] = u.sub_block_gain[2];
    uint8_t *sf_l = g + 8, *sf_s = g + 30;   // scale_fac_l[22], scale_fac_s[3][13]; the record was zeroed by the launcher
    const int sl0 = kSlen[u.scalefac_compress & 15][0], sl1 = kSlen[u.scalefac_compress & 15][1];
    // ---- scalefactors (Frame.py:365-441)
    if (gr == 1 && !(u.block_type == 2 && u.window_switching) &&
        (fs->scfsi[ch][0] | fs->scfsi[ch][1] | fs->scfsi[ch][2] | fs->scfsi[ch][3])) {
        // bands flagged by scfsi are copied from granule 0 (:423-437): decode them from granule 0's own bits
        const mp3s_unit_side u0 = fs->unit[0][ch];
        const int z0 = kSlen[u0.scalefac_compress & 15][0], z1 = kSlen[u0.scalefac_compress & 15][1];
        uint32_t b0 = bit_g0;
        for (int s = 0; s < 21; s++) {
            const int sl = s < 11 ? z0 : z1;
            const int band = s < 6 ? 0 : (s < 11 ? 1 : (s < 16 ? 2 : 3));
            const uint32_t v = md_get(mdw, md_len, b0, sl); b0 += sl;
            if (fs->scfsi[ch][band]) sf_l[s] = (uint8_t)v;
        }
    }
    if (u.block_type == 2 && u.window_switching) {
        if (u.mixed_block_flag) {
            for (int s = 0; s < 8; s++) { sf_l[s] = (uint8_t)br.get(bit, sl0); bit += sl0; }
            for (int s = 3; s < 6; s++)
                for (int w = 0; w < 3; w++) { sf_s[w * 13 + s] = (uint8_t)br.get(bit, sl0); bit += sl0; }
        } else {
            for (int s = 0; s < 6; s++)
                for (int w = 0; w < 3; w++) { sf_s[w * 13 + s] = (uint8_t)br.get(bit, sl0); bit += sl0; }
        }
        for (int s = 6; s < 12; s++)
            for (int w = 0; w < 3; w++) { sf_s[w * 13 + s] = (uint8_t)br.get(bit, sl1); bit += sl1; }
    } else if (gr == 0) {
        for (int s = 0; s < 11; s++) { sf_l[s] = (uint8_t)br.get(bit, sl0); bit += sl0; }
        for (int s = 11; s < 21; s++) { sf_l[s] = (uint8_t)br.get(bit, sl1); bit += sl1; }
    } else {
        for (int s = 0; s < 21; s++) {
            const int band = s < 6 ? 0 : (s < 11 ? 1 : (s < 16 ? 2 : 3)), sl = s < 11 ? sl0 : sl1;
            if (!fs->scfsi[ch][band]) { sf_l[s] = (uint8_t)br.get(bit, sl); bit += sl; }
        }
    }
    // ---- big values (Frame.py:458-518)
    uint32_t *smp = reinterpret_cast<uint32_t *>(is) + tid * 288;   // pair j; the buffer was zeroed by the launcher
    int region0, region1;
    bool ok = true;
    if (u.window_switching && u.block_type == 2) { region0 = 36; region1 = 576; }
    else {
        const int i0 = u.region0_count + 1, i1 = i0 + u.region1_count + 1;
        if (i0 > 22 || i1 > 22) { err |= MP3S_HS_BAD_REGION; ok = false; region0 = region1 = 0; }
        else { region0 = c_tab.sfb_long[sr][i0]; region1 = c_tab.sfb_long[sr][i1]; }
    }
    if (ok) {
        int sample = 0;
        const int bv2 = (int)u.big_values * 2;
        for (int r = 0; r < 3 && sample < bv2; r++) {
            const int rend = r == 0 ? region0 : (r == 1 ? region1 : 1 << 30);
            const int tn = (r == 0 ? u.table_select[0] : (r == 1 ? u.table_select[1] : u.table_select[2])) & 31;
            const int lut = c_tab.huff_lut_id[tn], lb = c_tab.linbits[tn];
            while (sample < bv2 && sample < rend) {
                if (sample + 1 >= 576) { err |= MP3S_HS_BIG_VALUES; break; }
                if (lut == 255) { sample += 2; continue; }        // books 0, 4, 14: zeros, no bits (D2)
                const uint64_t win64 = br.peek64(bit);
                const uint32_t window = (uint32_t)(win64 >> 32);
                const uint32_t e = fast[lut][window >> 22];
                int len = 0, sym = -1;
                if (e & 0x8000u) {                                // continue in the trie below the 10-bit prefix
                    uint32_t node = e & 0x7fffu;
                    for (int d = HUFF_FAST_BITS; d < 24; d++) {
                        const uint32_t nxt = c_tab.huff_tree[lut][node][(window >> (31 - d)) & 1];
                        if (!nxt) break;
                        if (nxt & 0x8000u) { sym = nxt & 0xff; len = d + 1; break; }
                        node = nxt;
                    }
                } else if (e) { sym = e & 0xff; len = e >> 8; }
                if (sym >= 0) {
                    // linbits and sign bits follow the code word: x linbits, x sign, y linbits, y sign (:499-513)
                    uint64_t rest = win64 << len;
                    int used = len;
                    int v0 = sym >> 4, v1 = sym & 15;
                    if (lb && v0 == 15) { v0 += (int)(rest >> (64 - lb)); rest <<= lb; used += lb; }
                    if ((sym >> 4) > 0) { if (rest >> 63) v0 = -v0; rest <<= 1; used += 1; }
                    if (lb && v1 == 15) { v1 += (int)(rest >> (64 - lb)); rest <<= lb; used += lb; }
                    if ((sym & 15) > 0) { if (rest >> 63) v1 = -v1; used += 1; }
                    bit += used;
                    if (v0 | v1) smp[sample >> 1] = (uint32_t)(uint16_t)v0 | ((uint32_t)(uint16_t)v1 << 16);
                }
                sample += 2;
            }
            if (err) break;
        }
        // ---- count1 quadruples (Frame.py:521-554, D1)
        while (!err && bit < max_bit && sample + 4 < 576) {
            uint32_t window = br.peek32(bit);
            int val, used;
            if (u.count1table_select) { val = (int)((window >> 28) ^ 15u); used = 4; }
            else {
                const uint32_t s = quad[window >> 26];
                val = s ? (int)(s & 15) : 0;
                used = (int)(s >> 4);
            }
            window <<= used;
            int q[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                q[i] = (val >> (3 - i)) & 1;
                if (q[i]) { if (window >> 31) q[i] = -1; window <<= 1; used += 1; }
            }
            bit += used;
            if (val) {
                smp[sample >> 1] = (uint32_t)(uint16_t)q[0] | ((uint32_t)(uint16_t)q[1] << 16);
                smp[(sample >> 1) + 1] = (uint32_t)(uint16_t)q[2] | ((uint32_t)(uint16_t)q[3] << 16);
            }
            sample += 4;
        }
    }
    if (err) atomicOr(status, err);
    }   // worker
}

}  // namespace mp3s
