// Huffman / scalefactor decode on the device (gfx950).  Included by mp3s_device.hip only.
//
//   k_dec_huffman : __unpack_scale_fac + __unpack_samples (reference decoder/Frame.py:365-559) for every
//                   granule*channel of the batch.  The bit stream of a granule is serial, but granule boundaries
//                   come from the side info (part2_3_length), so one THREAD decodes one granule*channel (granule 1
//                   re-reads the scalefactors scfsi lets it share with granule 0 from granule 0's own bits): a
//                   10 000-frame batch gives 40 000 independent threads.  Code books: 9-bit first-level table and second-level
//                   tables for the rare longer codes, both in LDS (same prefix codes the reference searches
//                   linearly, so the same symbol and length come out).  Quirks kept: D1 (count1 stops at line 572,
//                   no overrun discard), D2 (books 4/14 read no bits), bits past the buffer read as 0.
#pragma once

namespace mp3s {

constexpr int HUF_WORDS_MAX = 132;   // 4095 bits of part2_3_length + alignment slack, per thread
// words of LDS staging a thread needs for granules of at most `bits` bits (any alignment, +64 bits of look-ahead)
constexpr int huf_words_for(int bits) { return (bits + 94) / 32 + 4 < HUF_WORDS_MAX ? (bits + 94) / 32 + 4 : HUF_WORDS_MAX; }

// Bit stream over the lane's column of LDS: the words covering this granule's bits are copied there up front, so the
// decode loop touches no global memory except its (never waited for) stores.  The next 33..64 bits live in a register
// window; the word that will be appended next is loaded one refill ahead, so the only LDS access a symbol waits for is
// its code-book look-up.
template <int T>
struct BitStream {
    const uint32_t *lds;   // word j of this lane at lds[j * T]
    uint32_t last;         // W - 1: malformed streams may run past part2_3_length, the reader stays inside the column
    uint32_t idx;          // staged word that `nxt` holds
    uint32_t nxt;
    uint64_t win;          // next bit = MSB; bits below `valid` are zero
    int valid;
    __device__ __forceinline__ void open(const uint32_t *col, uint32_t W, uint32_t bit)
    {
        lds = col; last = W - 1;
        const uint32_t sh = bit & 31;
        win = (((uint64_t)lds[0] << 32) | lds[T]) << sh;
        valid = 64 - (int)sh;
        idx = 2; nxt = lds[2 * T];
    }
    // afterwards more than 32 bits are valid: enough for a code word and two sign bits, or for linbits + sign twice
    __device__ __forceinline__ void refill()
    {
        if (valid <= 32) {
            win |= (uint64_t)nxt << (32 - valid);
            valid += 32;
            idx = idx < last ? idx + 1 : last;
            nxt = lds[idx * T];
        }
    }
    __device__ __forceinline__ uint32_t top(int n) const { return (uint32_t)(win >> (64 - n)); }   // 1 <= n <= 32
    __device__ __forceinline__ void skip(int n) { win <<= n; valid -= n; }
    __device__ __forceinline__ uint32_t get(int n)   // n <= 4 (scalefactors)
    {
        if (!n) return 0;
        refill();
        const uint32_t v = top(n);
        skip(n);
        return v;
    }
};

// Scalefactors a granule reads without having written them: the reference keeps every scalefactor array from frame to
// frame (decoder/FrameSideInformation.py:11-37), and two kinds of granules look at entries their own bits do
// not set -- mixed short blocks (the three lowest short bands: Frame.py:392-397 writes sf_s[w][3..11] only) and granule 1
// behind a short granule 0 with scfsi set (Frame.py:423-437 copies sf_l[0][ch][s], which a short granule 0 did not
// write).  What they see is what the last granule of the same (gr, ch) that DID write the entry left there, possibly many
// frames back.  These helpers walk back through the side records of the stream (frame_side.reserved = the stream's first
// frame in the batch) and read the value straight from that granule's bits in the blob.  Rare: global loads, no staging.
__device__ __forceinline__ uint32_t huf_global_bits(const uint8_t *__restrict__ blob, uint32_t md_off, uint32_t md_len, uint32_t bit, int n)
{
    if (!n) return 0;
    uint32_t acc = 0;                                   // three bytes cover n <= 4 bits at any alignment
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const uint32_t b = (bit >> 3) + k;
        acc = (acc << 8) | (b < md_len ? blob[md_off + b] : 0u);   // bits past the main data read as 0 (decoder/util.py:41-43)
    }
    return (acc >> (24 - (bit & 7) - n)) & ((1u << n) - 1);
}
struct HufUnitAt {               // what the walk needs of unit (gr, ch) of frame f
    uint32_t md_off, md_len, bit, slen0, slen1;
    bool short_win, mixed;
};
__device__ __forceinline__ HufUnitAt huf_unit_at(const mp3s_frame_side *__restrict__ side, long f, int gr, int ch, int nch)
{
    const uint32_t *d = reinterpret_cast<const uint32_t *>(side + f);
    const int k = gr * 2 + ch;
    HufUnitAt u;
    u.md_off = d[0]; u.md_len = d[1];
    uint32_t bit = 0;
    for (int q = 0; q < k; q++)
        if ((q & 1) < nch) bit += d[5 + 5 * q] & 0xffffu;          // part2_3_length of the units in front (gr-major, ch inner)
    u.bit = bit;
    const uint32_t u1 = d[5 + 5 * k + 1], u2 = d[5 + 5 * k + 2];
    const uint32_t sc = (u1 >> 8) & 15;
    const uint8_t sl[16][2] = {{0, 0}, {0, 1}, {0, 2}, {0, 3}, {3, 0}, {1, 1}, {1, 2}, {1, 3},
                               {2, 1}, {2, 2}, {2, 3}, {3, 1}, {3, 2}, {3, 3}, {4, 2}, {4, 3}};
    u.slen0 = sl[sc][0]; u.slen1 = sl[sc][1];
    u.short_win = ((u1 >> 16) & 0xff) && (u1 >> 24) == 2;
    u.mixed = u.short_win && (u2 & 0xff);
    return u;
}
// sf_s[w][s] as a granule (gr, ch) of frame f finds it when its own bits do not set it: written last by a short granule
// of the same (gr, ch) -- a mixed one only writes bands 3..11 (Frame.py:392-405)
__device__ __noinline__ uint32_t huf_inherit_short(const uint8_t *__restrict__ blob, const mp3s_frame_side *__restrict__ side, long f, long first,
                                                   int gr, int ch, int nch, int w, int s)
{
    for (long fb = f - 1; fb >= first; fb--) {
        const HufUnitAt u = huf_unit_at(side, fb, gr, ch, nch);
        if (!u.short_win || (u.mixed && s < 3)) continue;
        uint32_t off;
        if (s >= 6) off = (u.mixed ? 17u : 18u) * u.slen0 + (uint32_t)((s - 6) * 3 + w) * u.slen1;
        else off = u.mixed ? (8u + (uint32_t)((s - 3) * 3 + w)) * u.slen0 : (uint32_t)(s * 3 + w) * u.slen0;
        return huf_global_bits(blob, u.md_off, u.md_len, u.bit + off, (int)(s >= 6 ? u.slen1 : u.slen0));
    }
    return 0;
}
// sf_l[0][ch][s] as granule 1 finds it when granule 0 of frame f did not write it (granule 0 short; mixed: s >= 8)
__device__ __noinline__ uint32_t huf_inherit_long(const uint8_t *__restrict__ blob, const mp3s_frame_side *__restrict__ side, long f, long first,
                                                  int ch, int nch, int s)
{
    for (long fb = f - 1; fb >= first; fb--) {
        const HufUnitAt u = huf_unit_at(side, fb, 0, ch, nch);
        if (!u.short_win)
            return huf_global_bits(blob, u.md_off, u.md_len, u.bit + (s < 11 ? (uint32_t)s * u.slen0 : 11 * u.slen0 + (uint32_t)(s - 11) * u.slen1),
                                   (int)(s < 11 ? u.slen0 : u.slen1));
        if (u.mixed && s < 8) return huf_global_bits(blob, u.md_off, u.md_len, u.bit + (uint32_t)s * u.slen0, (int)u.slen0);
    }
    return 0;
}

// mp3s_frame_side (104 bytes) in 26 registers: 13 independent 8-byte loads, one wait; fields by constant offsets, the
// four unit records by select (a struct copy with a run-time index would go through scratch)
struct SideRegs {
    uint32_t d[26];
    __device__ __forceinline__ void load(const mp3s_frame_side *p)
    {
        const uint2 *q = reinterpret_cast<const uint2 *>(p);
#pragma unroll
        for (int i = 0; i < 13; i++) { const uint2 v = q[i]; d[2 * i] = v.x; d[2 * i + 1] = v.y; }
    }
    __device__ __forceinline__ uint32_t unit_dw(int k, int j) const   // dword j of unit[k >> 1][k & 1]
    {
        const uint32_t a = k & 1 ? d[10 + j] : d[5 + j], b = k & 1 ? d[20 + j] : d[15 + j];
        return k & 2 ? b : a;
    }
    __device__ __forceinline__ uint32_t p23(int k) const { return d[5 + 5 * k] & 0xffffu; }   // k is a constant here
};

// One thread per granule*channel; unit index = (frame*2 + gr)*2 + ch (same order as the si / is arrays).
// LDS = the shared first-level tables (30.8 KB) + W words of staged bits per thread.  A wave is a chain of dependent LDS
// look-ups, so throughput comes from resident waves: W is sized by the launcher from the longest granule of the batch
// (30 words at 128 kbps instead of the worst-case 132), which lets 8 waves share a CU instead of 2.
// WAVES x 64 threads per workgroup of which LANES per wave decode: every lane walks its own bit stream, so a wave is a
// latency chain that runs as long as its slowest lane.  Small batches use narrow waves (more waves per SIMD to
// interleave, less waiting for the slowest lane), large batches full ones (more granules in flight per CU).
template <int WAVES, int LANES>
__global__ __launch_bounds__(WAVES * 64) void k_dec_huffman(
    const uint8_t *__restrict__ blob, const mp3s_frame_side *__restrict__ side, int n_frames, int nch, int W, int max_bits,
    int16_t *__restrict__ is, mp3s_granule_si *__restrict__ si_out, int32_t *__restrict__ status, int per_frame,
    int32_t *__restrict__ sync /* {finished workgroups, error bits}: zero between launches, owned by the context; null: status[0] was
                                  zeroed by the caller and takes the error bits directly */)
{
    __shared__ uint16_t fast[15][HUFF_L1_N];
    __shared__ uint16_t lut2[HUFF_L2_N];    // second-level tables for the codes longer than the first-level index
    __shared__ uint16_t quad[64];
    __shared__ uint16_t tinfo[32];        // table_select -> first-level table id | linbits << 8
    __shared__ int wg_err;                // the group's error bits
    extern __shared__ uint32_t words[];   // [W][COLS]
    constexpr int T = WAVES * 64, COLS = WAVES * LANES;
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(&c_tab.huff_fast[0][0]);
        uint4 *dst = reinterpret_cast<uint4 *>(&fast[0][0]);
        constexpr int N16 = 15 * HUFF_L1_N * 2 / 16, ROUNDS = (N16 + T - 1) / T;
        uint4 v[ROUNDS <= 8 ? ROUNDS : 8];
        if constexpr (ROUNDS <= 8) {      // every load in flight before the first LDS write
#pragma unroll
            for (int r = 0; r < ROUNDS; r++) { const int i = threadIdx.x + r * T; v[r] = src[i < N16 ? i : N16 - 1]; }
#pragma unroll
            for (int r = 0; r < ROUNDS; r++) { const int i = threadIdx.x + r * T; if (i < N16) dst[i] = v[r]; }
        } else {
#pragma unroll 6
            for (int i = threadIdx.x; i < N16; i += T) dst[i] = src[i];
        }
        for (int i = threadIdx.x; i < HUFF_L2_N / 2; i += T)
            reinterpret_cast<uint32_t *>(lut2)[i] = reinterpret_cast<const uint32_t *>(c_tab.huff_l2)[i];
        if (threadIdx.x < 64) quad[threadIdx.x] = c_tab.quad_fast[threadIdx.x];
        if (threadIdx.x < 32) tinfo[threadIdx.x] = (uint16_t)(c_tab.huff_lut_id[threadIdx.x] | (c_tab.linbits[threadIdx.x] << 8));
        if (threadIdx.x == 0) wg_err = 0;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, col = (int)(threadIdx.x >> 6) * LANES + lane;
    const long tid = (long)blockIdx.x * COLS + col;
    bool worker = lane < LANES && tid < (long)n_frames * 4 && (int)(tid & 1) < nch;
    // a frame the host has decoded itself (MP3S_FS_HOST_DECODED: its samples are placed behind this kernel) is left alone
    const bool host_frame = tid < (long)n_frames * 4 && ((reinterpret_cast<const uint32_t *>(side + (tid >> 2))[2] >> 24) & MP3S_FS_HOST_DECODED);
    if (host_frame) worker = false;
    int err = 0;
    // Nothing is cleared in front of this kernel: every unit writes all of its 288 sample pairs and its whole side record
    // (the second channel's unit of a mono stream: zeros), and the status words are plain stores.
    if (!worker && !host_frame && lane < LANES && tid < (long)n_frames * 4) {
        uint32_t *z = reinterpret_cast<uint32_t *>(is) + tid * 288;
        for (int j = 0; j < 288; j++) z[j] = 0;
        uint32_t *zr = reinterpret_cast<uint32_t *>(si_out + tid);
        for (int j = 0; j < 18; j++) zr[j] = 0;
    }
    if (worker) {
    const int f = (int)(tid >> 2), k = (int)(tid & 3), gr = k >> 1, ch = k & 1;
    SideRegs fs;
    fs.load(side + f);
    static const uint8_t kSlen[16][2] = {{0, 0}, {0, 1}, {0, 2}, {0, 3}, {3, 0}, {1, 1}, {1, 2}, {1, 3},
                                         {2, 1}, {2, 2}, {2, 3}, {3, 1}, {3, 2}, {3, 3}, {4, 2}, {4, 3}};
    const uint32_t md_off = fs.d[0], md_len = fs.d[1];
    const int sr_raw = (int)((fs.d[2] >> 8) & 0xff), sr = sr_raw < 3 ? sr_raw : 0;
    // bit offset of this unit and (for scfsi) of granule 0 of the same channel: units are laid out gr-major
    const uint32_t l0 = fs.p23(0), l1 = nch > 1 ? fs.p23(1) : 0u, l2 = fs.p23(2);
    const uint32_t bit_g0 = ch ? l0 : 0u;
    uint32_t bit = (k >= 1 ? l0 : 0u) + (k >= 2 ? l1 : 0u) + (k >= 3 ? l2 : 0u);
    if (nch == 1) bit = gr ? l0 : 0u;
    const uint32_t u0 = fs.unit_dw(k, 0), u1 = fs.unit_dw(k, 1), u2 = fs.unit_dw(k, 2), u3 = fs.unit_dw(k, 3), u4 = fs.unit_dw(k, 4);
    const uint32_t part2_3_length = u0 & 0xffffu, big_values = u0 >> 16;
    const uint32_t scalefac_compress = (u1 >> 8) & 0xff, window_switching = (u1 >> 16) & 0xff, block_type = u1 >> 24;
    const uint32_t mixed_block_flag = u2 & 0xff;
    const uint32_t region0_count = u3 & 0xff, region1_count = (u3 >> 8) & 0xff;
    const uint32_t count1table_select = u4 & 0xff;
    const uint32_t scfsi = ch ? fs.d[4] : fs.d[3];   // one byte per band
    const uint32_t max_bit = bit + part2_3_length;
    const uint32_t *mdw = reinterpret_cast<const uint32_t *>(blob + md_off);
    if ((int)part2_3_length > max_bits) err |= MP3S_HS_HINT;   // the caller's bound on part2_3_length does not hold
    BitStream<COLS> br;
    {
        // stage the words covering this granule: loads first (clamped to a word inside the zero bytes that follow the
        // frame), then the big-endian swap, the masking of bytes past md_len (decoder/util.py:41-43) and the LDS writes
        const uint32_t base = bit >> 5, nw0 = ((max_bit + 63) >> 5) - base + 3, nw = nw0 < (uint32_t)W ? nw0 : (uint32_t)W;
        const uint32_t imax = (md_len + 3) >> 2;
        for (uint32_t j0 = 0; j0 < nw; j0 += 8) {
            uint32_t v[8];
#pragma unroll
            for (int q = 0; q < 8; q++) { const uint32_t i = base + j0 + q; v[q] = mdw[i < imax ? i : imax]; }
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const uint32_t i = base + j0 + q, byte0 = i * 4;
                uint32_t x = __builtin_bswap32(v[q]);
                const uint32_t valid = md_len - byte0;                 // meaningful when byte0 < md_len
                if (byte0 >= md_len) x = 0;
                else if (valid < 4) x &= 0xffffffffu << (8 * (4 - valid));
                if (j0 + q < nw) words[(j0 + q) * COLS + col] = x;
            }
        }
        br.open(words + col, (uint32_t)W, bit);
    }
    uint32_t *g32 = reinterpret_cast<uint32_t *>(si_out + tid);
#pragma unroll
    for (int q = 2; q < 18; q++) g32[q] = 0;   // the scalefactor bytes are stored one by one below, the rest stays zero
    // global_gain, scalefac_scale, block_type, mixed_block_flag | preflag, sub_block_gain[3]
    g32[0] = (u1 & 0xff) | ((u3 >> 24) << 8) | (block_type << 16) | (mixed_block_flag << 24);
    g32[1] = ((u3 >> 16) & 0xff) | ((u4 >> 8) << 8);
    uint8_t *g = reinterpret_cast<uint8_t *>(g32);
    uint8_t *sf_l = g + 8, *sf_s = g + 30;   // scale_fac_l[22], scale_fac_s[3][13]
    const int sl0 = kSlen[scalefac_compress & 15][0], sl1 = kSlen[scalefac_compress & 15][1];
    const bool short_win = block_type == 2 && window_switching;
    // ---- scalefactors (Frame.py:365-441)
    const long stream_first = (long)fs.d[25];           // frame_side.reserved: first frame of this stream in the batch
    if (gr == 1 && !short_win && scfsi) {
        // bands flagged by scfsi are copied from granule 0 (:423-437): decode them from granule 0's own bits, which the
        // lane two columns to the left (same frame, same wave) has staged
        const uint32_t g0u1 = ch ? fs.d[11] : fs.d[6], g0u2 = ch ? fs.d[12] : fs.d[7];
        const uint32_t c0 = (g0u1 >> 8) & 15;
        const int z0 = kSlen[c0][0], z1 = kSlen[c0][1];
        const bool g0_short = ((g0u1 >> 16) & 0xff) && (g0u1 >> 24) == 2, g0_mixed = g0_short && (g0u2 & 0xff);
        __builtin_amdgcn_wave_barrier();
        if (!g0_short) {
            BitStream<COLS> b0;
            b0.open(words + col - 2, (uint32_t)W, bit_g0);
            for (int s = 0; s < 21; s++) {
                const int sl = s < 11 ? z0 : z1;
                const int band = s < 6 ? 0 : (s < 11 ? 1 : (s < 16 ? 2 : 3));
                const uint32_t v = b0.get(sl);
                if ((scfsi >> (8 * band)) & 0xff) sf_l[s] = (uint8_t)v;
            }
        } else {
            // granule 0 is short: it wrote sf_l[0..7] if mixed (first in its bits), nothing of sf_l otherwise; the rest is
            // what earlier frames left in the array
            for (int s = 0; s < 21; s++) {
                const int band = s < 6 ? 0 : (s < 11 ? 1 : (s < 16 ? 2 : 3));
                if (!((scfsi >> (8 * band)) & 0xff)) continue;
                if (g0_mixed && s < 8) sf_l[s] = (uint8_t)huf_global_bits(blob, md_off, md_len, bit_g0 + (uint32_t)(s * z0), z0);
                else sf_l[s] = (uint8_t)huf_inherit_long(blob, side, f, stream_first, ch, nch, s);
            }
        }
    }
    if (short_win) {
        if (mixed_block_flag) {
            for (int s = 0; s < 8; s++) { sf_l[s] = (uint8_t)br.get(sl0); bit += sl0; }
            for (int s = 3; s < 6; s++)
                for (int w = 0; w < 3; w++) { sf_s[w * 13 + s] = (uint8_t)br.get(sl0); bit += sl0; }
            for (int s = 0; s < 3; s++)                 // not in this granule's bits: what the array still holds
                for (int w = 0; w < 3; w++) sf_s[w * 13 + s] = (uint8_t)huf_inherit_short(blob, side, f, stream_first, gr, ch, nch, w, s);
        } else {
            for (int s = 0; s < 6; s++)
                for (int w = 0; w < 3; w++) { sf_s[w * 13 + s] = (uint8_t)br.get(sl0); bit += sl0; }
        }
        for (int s = 6; s < 12; s++)
            for (int w = 0; w < 3; w++) { sf_s[w * 13 + s] = (uint8_t)br.get(sl1); bit += sl1; }
    } else if (gr == 0) {
        for (int s = 0; s < 11; s++) { sf_l[s] = (uint8_t)br.get(sl0); bit += sl0; }
        for (int s = 11; s < 21; s++) { sf_l[s] = (uint8_t)br.get(sl1); bit += sl1; }
    } else {
        for (int s = 0; s < 21; s++) {
            const int band = s < 6 ? 0 : (s < 11 ? 1 : (s < 16 ? 2 : 3)), sl = s < 11 ? sl0 : sl1;
            if (!((scfsi >> (8 * band)) & 0xff)) { sf_l[s] = (uint8_t)br.get(sl); bit += sl; }
        }
    }
    if (!short_win && window_switching && mixed_block_flag) {
        // a start / stop block with the mixed flag: its scalefactors are the long ones (:406-441), but re_quantize switches
        // to the short bands from sfb 8 on (Frame.py:186) and reads sf_s[w][8..], which only short granules write
        for (int s = 8; s < 12; s++)
            for (int w = 0; w < 3; w++) sf_s[w * 13 + s] = (uint8_t)huf_inherit_short(blob, side, f, stream_first, gr, ch, nch, w, s);
    }
    // ---- big values (Frame.py:458-518).  One flat loop over the pairs: the region (and with it the code book) is
    //      looked up per pair, so a wave runs for its longest granule, not for the longest region 0 + region 1 + region 2.
    uint32_t *smp = reinterpret_cast<uint32_t *>(is) + tid * 288;   // pair j
    int region0, region1;
    bool ok = true;
    if (short_win) { region0 = 36; region1 = 576; }
    else {
        const int i0 = (int)region0_count + 1, i1 = i0 + (int)region1_count + 1;
        if (i0 > 22 || i1 > 22) { err |= MP3S_HS_BAD_REGION; ok = false; region0 = region1 = 0; }
        else { region0 = c_tab.sfb_long[sr][i0]; region1 = c_tab.sfb_long[sr][i1]; }
    }
    const int bv2 = (int)big_values * 2;
    if (bv2 > 576) err |= MP3S_HS_BIG_VALUES;                     // the reference runs off its sample array (IndexError)
    int sample = 0;
    if (ok && !err) {
        const uint32_t ti0 = tinfo[(u2 >> 8) & 31], ti1 = tinfo[(u2 >> 16) & 31], ti2 = tinfo[(u2 >> 24) & 31];
        while (sample < bv2) {
            // big values that run past part2_3_length read on into the data that follows (the reference has one bit
            // cursor per frame); the staged window covers one code word of that, a second one is reported
            if (bit > max_bit) { err |= MP3S_HS_OVERRUN; break; }
            const uint32_t ti = sample < region0 ? ti0 : (sample < region1 ? ti1 : ti2);
            const int lut = (int)(ti & 0xff), lb = (int)(ti >> 8);
            if (lut == 255) {                                     // books 0, 4, 14: zeros, no bits (D2): skip the region
                const int rend = sample < region0 ? region0 : (sample < region1 ? region1 : bv2);
                const int to = rend < bv2 ? rend : bv2;           // region bounds are even
                for (int j = sample >> 1; j < (to >> 1); j++) smp[j] = 0;
                sample = to;
                continue;
            }
            br.refill();
            const uint32_t window = br.top(32);
            const uint32_t e = fast[lut][window >> (32 - HUFF_FAST_BITS)];
            int len = 0, sym = -1;
            uint32_t leaf = e;
            if (e & 0x8000u) {                                    // longer than the index: the next k bits pick the leaf
                const uint32_t k = (e >> 11) & 15;
                leaf = lut2[2 * (e & 0x7ffu) + ((window << HUFF_FAST_BITS) >> (32 - k))];
            }
            if (leaf) { sym = (int)(leaf & 0xff); len = (int)(leaf >> 8); }
            if (sym >= 0) {
                // linbits and sign bits follow the code word: x linbits, x sign, y linbits, y sign (:499-513)
                int v0 = sym >> 4, v1 = sym & 15;
                br.skip(len);
                int used = len;
                if (lb && (v0 == 15 || v1 == 15)) {               // escape values: up to 2 x (13 + 1) more bits
                    br.refill();
                    if (v0 == 15) { v0 += (int)br.top(lb); br.skip(lb); used += lb; }
                    if (v0) { if (br.top(1)) v0 = -v0; br.skip(1); used += 1; }
                    if (v1 == 15) { v1 += (int)br.top(lb); br.skip(lb); used += lb; }
                    if (v1) { if (br.top(1)) v1 = -v1; br.skip(1); used += 1; }
                } else {
                    const uint32_t sg = br.top(2);
                    const int n0 = v0 != 0, n1 = v1 != 0;
                    if (n0 && (sg >> 1)) v0 = -v0;
                    if (n1 && ((n0 ? sg : sg >> 1) & 1)) v1 = -v1;
                    br.skip(n0 + n1); used += n0 + n1;
                }
                bit += used;
                smp[sample >> 1] = (uint32_t)(uint16_t)v0 | ((uint32_t)(uint16_t)v1 << 16);
            } else smp[sample >> 1] = 0;                          // no code word matches: the reference leaves the pair at zero
            sample += 2;
        }
        // ---- count1 quadruples (Frame.py:521-554, D1)
        while (!err && bit < max_bit && sample + 4 < 576) {
            br.refill();
            uint32_t window = br.top(32);
            int val, used;
            if (count1table_select) { val = (int)((window >> 28) ^ 15u); used = 4; }
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
            br.skip(used);
            bit += used;
            smp[sample >> 1] = (uint32_t)(uint16_t)q[0] | ((uint32_t)(uint16_t)q[1] << 16);
            smp[(sample >> 1) + 1] = (uint32_t)(uint16_t)q[2] | ((uint32_t)(uint16_t)q[3] << 16);
            sample += 4;
        }
    }
    for (int j = sample >> 1; j < 288; j++) smp[j] = 0;   // what no code word reached (everything, when the unit was rejected)
    }   // worker
    // ---- status: the four units of a frame sit in four neighbouring lanes; the first of them stores the frame's word
    //      (which frame: the caller re-parses only the streams that hold one), errors of the whole launch are collected
    //      in the context's pair and handed out by the workgroup that finishes last (no fill launch in front of the kernel)
    int e4 = err | __shfl_xor(err, 1, 64);
    e4 |= __shfl_xor(e4, 2, 64);
    if (lane < LANES && tid < (long)n_frames * 4 && (tid & 3) == 0) {
        if (per_frame) status[1 + (tid >> 2)] = e4;
        if (e4) atomicOr(&wg_err, e4);                   // (LDS)
    }
    __syncthreads();
    // One thread speaks for the group, and what it has to say goes out in program order: the group's error bits first, its
    // arrival after them -- two device-scope atomics of one lane on neighbouring words, so the last group to arrive finds
    // every group's bits (no fence: an agent-scope release writes the L2 back, which two thousand groups cannot afford).
    if (threadIdx.x != 0) return;
    const int g = wg_err;
    if (!sync) {          // the caller has zeroed status[0] itself (the overlapped stages: the word travels with the job's inputs)
        if (g) atomicOr(&status[0], g);
        return;
    }
    if (g) atomicOr(&sync[1], g);
    if (atomicAdd(&sync[0], 1) == (int)gridDim.x - 1) {
        status[0] = atomicExch(&sync[1], 0);
        atomicExch(&sync[0], 0);
    }
}

}  // namespace mp3s
