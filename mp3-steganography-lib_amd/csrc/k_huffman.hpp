// Huffman / scalefactor decode on the device (gfx950).  Included by mp3s_device.hip only.
//
//   k_dec_huffman : __unpack_scale_fac + __unpack_samples (reference decoder/Frame.py:365-559) for every
//                   granule*channel of the batch.  The bit stream of a granule is serial, but granule boundaries
//                   come from the side info (part2_3_length), so one THREAD decodes one granule*channel (granule 1
//                   re-reads the scalefactors scfsi lets it share with granule 0 from granule 0's own bits): a
//                   10 000-frame batch gives 40 000 independent threads.  Code books: a first-level table per book indexed
//                   by up to 10 bits and second-level tables for the longer codes, both in LDS (same prefix codes the
//                   reference searches linearly, so the same symbol and length come out).  A lane is a chain of dependent
//                   look-ups: the loop is written so that between two of them lie the shift of the bit window and the index
//                   arithmetic only -- signs, stores and the refill of the window run while the next look-up is in flight.
//                   Quirks kept: D1 (count1 stops at line 572,
//                   no overrun discard), D2 (books 4/14 read no bits), bits past the buffer read as 0.
#pragma once
#ifndef MP3S_HUF_CLOCKS
#define MP3S_HUF_CLOCKS 0   // 1: shader-clock deltas of the phases into sample pairs 280..285 of every row (a round-3 probe)
#endif

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
    // the same without a branch, for the symbol loops (every lane refills at its own pace: as a branch it is taken by some
    // lane almost every time, and the look-up behind it would wait for the branch); re-reads the staged word when nothing
    // was appended
    __device__ __forceinline__ void refill_always()
    {
        const bool need = valid <= 32;
        const uint64_t add = (uint64_t)(need ? nxt : 0u) << ((32 - valid) & 31);
        win |= add;
        valid += need ? 32 : 0;
        const uint32_t up = idx + (need ? 1u : 0u);
        idx = up < last ? up : last;
        // (the staged word is read AFTER the old one was used -- the address is made to depend on the window: one register, no copy and so
        //  no wait for the word at the loop's back edge)
        asm volatile("" : "+v"(idx) : "v"((uint32_t)win));
        nxt = lds[idx * T];
    }
    __device__ __forceinline__ uint32_t hi() const { return (uint32_t)(win >> 32); }
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
// LDS = the shared first-level tables + W words of staged bits per thread (W is sized by the launcher from the longest
// granule of the batch: 30 words at 128 kbps instead of the worst-case 132).
// WAVES x 64 threads per workgroup of which LANES per wave decode.  The lanes of a wave walk their 288 pairs in step, so
// the kernel lasts as long as ONE wave's chain of dependent look-ups as long as every SIMD has at most one wave (a second
// wave on a SIMD costs +20 %: DESIGN.md section 4): the launcher picks LANES so that the batch spreads over as many SIMDs
// as there are (32 lanes per wave up to 8 192 frames, 64 beyond) -- the kernel's time is a latency, not a throughput.
template <int WAVES, int LANES>
__global__ __launch_bounds__(WAVES * 64) void k_dec_huffman(
    const uint8_t *__restrict__ blob, const mp3s_frame_side *__restrict__ side, int n_frames, int nch, int W, int max_bits,
    int16_t *__restrict__ is, mp3s_granule_si *__restrict__ si_out, int32_t *__restrict__ status, int per_frame,
    int32_t *__restrict__ sync /* {finished workgroups | error bits}, one 8-byte aligned 64-bit word (k_sync.hpp): zero between launches, owned by the context; null: status[0] was
                                  zeroed by the caller and takes the error bits directly */)
{
#if MP3S_HUF_CLOCKS
    const unsigned long long clk0 = __builtin_readcyclecounter();
#endif
    __shared__ __attribute__((aligned(16))) uint16_t tab[HUF_TAB_N];   // first level | second level | count1 (mp3s_tables.h)
    __shared__ uint32_t tinfo[32];        // table_select -> byte offset of the first level << 16 | (32 - w) % 32 << 8 | linbits << 4 | w
    __shared__ int wg_err;                // the group's error bits
    extern __shared__ uint32_t words[];   // [W][COLS] staged bits, [WAVES][LANES][17] output tiles
    constexpr int T = WAVES * 64, COLS = WAVES * LANES;
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(c_tab.huf_tab);
        uint4 *dst = reinterpret_cast<uint4 *>(tab);
        constexpr int N16 = HUF_TAB_N * 2 / 16, ROUNDS = (N16 + T - 1) / T;
        static_assert(HUF_TAB_N % 8 == 0, "16-byte copies");
        uint4 v[ROUNDS];
#pragma unroll
        for (int r = 0; r < ROUNDS; r++) { const int i = threadIdx.x + r * T; v[r] = src[i < N16 ? i : N16 - 1]; }   // every load in flight before the first LDS write
#pragma unroll
        for (int r = 0; r < ROUNDS; r++) { const int i = threadIdx.x + r * T; if (i < N16) dst[i] = v[r]; }
        if (threadIdx.x < 32) tinfo[threadIdx.x] = c_tab.huf_tinfo[threadIdx.x];
        if (threadIdx.x == 0) wg_err = 0;
    }
    __syncthreads();
#if MP3S_HUF_CLOCKS
    const unsigned long long clk1 = __builtin_readcyclecounter();
#endif
    const int lane = threadIdx.x & 63, col = (int)(threadIdx.x >> 6) * LANES + lane;
    const long tid = (long)blockIdx.x * COLS + col;
    bool worker = lane < LANES && tid < (long)n_frames * 4 && (int)(tid & 1) < nch;
    // a frame the host has decoded itself (MP3S_FS_HOST_DECODED: its samples are placed behind this kernel) is left alone
    const bool host_frame = tid < (long)n_frames * 4 && ((reinterpret_cast<const uint32_t *>(side + (tid >> 2))[2] >> 24) & MP3S_FS_HOST_DECODED);
    if (host_frame) worker = false;
    int err = 0;
    // Nothing is cleared in front of this kernel: every unit writes all of its 288 sample pairs and its whole side record
    // (the second channel's unit of a mono stream: zeros), and the status words are plain stores.
    const bool has_row = lane < LANES && tid < (long)n_frames * 4 && !host_frame;   // this lane's 288 sample pairs are written here
    if (!worker && has_row) {
        uint32_t *zr = reinterpret_cast<uint32_t *>(si_out + tid);
        for (int j = 0; j < 18; j++) zr[j] = 0;
    }
    // what the symbol loop below needs of a lane (a lane without a unit: nothing to decode, zeros if it has a row)
    BitStream<COLS> br;
    br.lds = words + (lane < LANES ? col : (int)(threadIdx.x >> 6) * LANES); br.last = 0; br.idx = 0; br.nxt = 0; br.win = 0; br.valid = 64;
    uint32_t bit = 0, max_bit = 0, ti0 = 0, ti1 = 0, ti2 = 0, tic1a = 0, tic1b = 0;
    int r0p = 0, r1p = 0, bvp = 0;         // region bounds and big values in pairs
    bool c1_open = false;
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
    bit = (k >= 1 ? l0 : 0u) + (k >= 2 ? l1 : 0u) + (k >= 3 ? l2 : 0u);
    if (nch == 1) bit = gr ? l0 : 0u;
    const uint32_t u0 = fs.unit_dw(k, 0), u1 = fs.unit_dw(k, 1), u2 = fs.unit_dw(k, 2), u3 = fs.unit_dw(k, 3), u4 = fs.unit_dw(k, 4);
    const uint32_t part2_3_length = u0 & 0xffffu, big_values = u0 >> 16;
    const uint32_t scalefac_compress = (u1 >> 8) & 0xff, window_switching = (u1 >> 16) & 0xff, block_type = u1 >> 24;
    const uint32_t mixed_block_flag = u2 & 0xff;
    const uint32_t region0_count = u3 & 0xff, region1_count = (u3 >> 8) & 0xff;
    const uint32_t count1table_select = u4 & 0xff;
    const uint32_t scfsi = ch ? fs.d[4] : fs.d[3];   // one byte per band
    max_bit = bit + part2_3_length;
    const uint32_t *mdw = reinterpret_cast<const uint32_t *>(blob + md_off);
    if ((int)part2_3_length > max_bits) err |= MP3S_HS_HINT;   // the caller's bound on part2_3_length does not hold
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
    const long stream_first = (long)(int32_t)fs.d[25];  // frame_side.reserved: first frame of this stream, relative to side[0] (negative: earlier chunks' records in front)
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
    // ---- big values and count1: set up here, decoded below by all lanes of the wave in step
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
    if (ok && !err) {
        ti0 = tinfo[(u2 >> 8) & 31]; ti1 = tinfo[(u2 >> 16) & 31]; ti2 = tinfo[(u2 >> 24) & 31];
        // count1: book A on ten bits, book B on eight; (v, w) and (x, y) of a quadruple are two look-ups of the same bits
        constexpr uint32_t C1 = (uint32_t)(HUF_L1_N + HUF_L2_N) * 2;
        tic1b = count1table_select ? ((C1 + 4096 + 512) << 16) | (24u << 8) | 8u : ((C1 + 2048) << 16) | (22u << 8) | 10u;
        tic1a = (tic1b - ((count1table_select ? 512u : 2048u) << 16)) | (1u << 13);   // bit 13: the window stays where it is
        r0p = region0 >> 1; r1p = region1 >> 1; bvp = (int)big_values;
        c1_open = true;
        br.refill();
    }
    }   // worker
    // ---- big values (Frame.py:458-518) and count1 quadruples (:521-554, D1).  The lanes of a wave go through the 288 pairs of
    //      their units in step: in iteration p a lane emits pair p -- a big-values symbol, half of a count1 quadruple, or zero --
    //      into the wave's tile in LDS, and every 16 iterations the wave writes the tile out as whole 64-byte pieces of the
    //      rows.  A wave is a chain of dependent look-ups and runs as fast as it issues instructions, so an iteration is
    //      one look-up and as little around it as the formats allow: the three kinds of pairs share ONE entry format (count1:
    //      two entries per quadruple, the second moves the window; zeros: entry 0), the window shift and the index of the
    //      NEXT look-up come first, signs, values and the refill of the window follow while it is in flight.
    {
#if MP3S_HUF_CLOCKS
        const unsigned long long clk2 = __builtin_readcyclecounter();
#endif
        typedef short s16x2 __attribute__((ext_vector_type(2)));
        uint32_t *tile = words + (size_t)W * COLS + (threadIdx.x >> 6) * ((LANES + 1) * 17);
        uint32_t *const my_tile = tile + (lane < LANES ? lane : LANES) * 17;   // (a lane without a row: the spare one)
        const unsigned long long rows = __ballot(has_row);
        const long tid0 = (long)blockIdx.x * COLS + (long)(threadIdx.x >> 6) * LANES;   // the wave's first unit
        uint32_t *const is32 = reinterpret_cast<uint32_t *>(is);
        const char *tabb = reinterpret_cast<const char *>(tab);
        // The books of a lane, pair by pair: region 0 / 1 / 2 up to `bve`, then the two count1 entries in turn (c1n: the next
        // one, tog: what turns one into the other), zeros (book 0) once count1 has ended.  What ends a phase early is an EVENT:
        // big values or a quadruple that start at or behind the unit's last bit (mb), looked for with one compare.
        int bve = bvp;
        uint32_t c1n = c1_open ? tic1a : 0u, tog = c1_open ? tic1a ^ tic1b : 0u;
        uint32_t mb = c1_open ? max_bit : 0xffffffffu;
        uint32_t ti = 0 < r0p ? ti0 : (0 < r1p ? ti1 : ti2);
        if (!(0 < bve)) { ti = c1n; c1n ^= tog; }
        uint32_t e = *reinterpret_cast<const uint16_t *>(tabb + (ti >> 16) + 2 * __builtin_amdgcn_ubfe(br.hi(), (ti >> 8) & 31, ti & 15));
        // Eighteen tiles of sixteen pairs: the inner loop is counted and has no way out (as one loop with the write-out and the `break` inside
        // an `if` the compiler kept two flags and seven scalar instructions per pair alive for them).
        int p = 0;
        for (int tile_no = 0; tile_no < 18; tile_no++) {
#pragma unroll 4
        for (int q16 = 0; q16 < 16; q16++, p++) {
            // -- what does not depend on the look-up in flight
            const uint32_t chk = (ti >> 13) & 1;                  // 1: (v, w) of a quadruple -- the entry does not move the window
            const uint32_t am = (chk - 1) & 31u;                  // ... (0; 31 otherwise: code words of up to 19 bits + 2 signs)
            // big values that run past part2_3_length read on into the data that follows (the reference has one bit cursor per
            // frame; the staged window covers one code word of that, a second one is reported); a quadruple is only started
            // in front of the last bit and of line 572 (D1: from pair 286 on the compare is lost for every (v, w))
            const uint32_t lim = bit + (chk << (p >= 286 ? 30 : 0));
            if (__builtin_expect(__any(lim > mb), 0)) {   // (unlikely: a wave alone on its SIMD pays every TAKEN branch with an empty instruction buffer)
                if (lim > mb) {
                    if (!chk) err |= MP3S_HS_OVERRUN;
                    bve = 0; c1n = 0; tog = 0; mb = 0xffffffffu;
                    e = 0;                                        // entry 0: nothing
                }
            }
            const int pn = p + 1;
            const bool more_bv = pn < bve;
            const uint32_t tn = more_bv ? (pn < r0p ? ti0 : (pn < r1p ? ti1 : ti2)) : c1n;
            c1n ^= more_bv ? 0u : tog;
            const uint32_t hi_old = br.hi();
            // -- the entry
            uint32_t adv = (e >> 8) & 15;          // code word + sign bits (books 0, 4, 14 -- entry 0: nothing, D2)
            uint32_t out_esc = 0;
            bool escaped = false;
            if (__builtin_expect((e & 0xc000u) != 0, 0)) {   // (big-values books only)
                if (e & 0x8000u) {                 // longer than the index: the next k bits pick the leaf
                    const uint32_t k = (e >> 11) & 15, w = ti & 15;
                    e = tab[HUF_L1_N + 2 * (e & 0x7ffu) + ((hi_old << w) >> (32 - k))];
                    adv = w + ((e >> 8) & 15);
                }
                if (e & 0x4000u) {                 // escape values: x linbits, x sign, y linbits, y sign (:499-513), up to 2 x (13 + 1) more bits
                    int v0 = (int)((e >> 4) & 15), v1 = (int)(e & 15);
                    const int lb = (int)((ti >> 4) & 15), len = (int)adv - (v0 != 0) - (v1 != 0);
                    int used = len;
                    br.skip(len);
                    br.refill();
                    if (v0 == 15) { v0 += (int)br.top(lb); br.skip(lb); used += lb; }
                    if (v0) { if (br.top(1)) v0 = -v0; br.skip(1); used += 1; }
                    if (v1 == 15) { v1 += (int)br.top(lb); br.skip(lb); used += lb; }
                    if (v1) { if (br.top(1)) v1 = -v1; br.skip(1); used += 1; }
                    br.refill();
                    bit += (uint32_t)used;
                    out_esc = (uint32_t)(uint16_t)v0 | ((uint32_t)(uint16_t)v1 << 16);
                    escaped = true;
                }
            }
            const uint32_t step = escaped ? 0u : adv & am;
            br.win <<= step;
            // the next pair's look-up goes out before this pair's values are put together
            const uint32_t en = *reinterpret_cast<const uint16_t *>(tabb + (tn >> 16) + 2 * __builtin_amdgcn_ubfe(br.hi(), (tn >> 8) & 31, tn & 15));
            __builtin_amdgcn_sched_barrier(0);
            br.valid -= (int)step; bit += step;
            // the window's refill comes FIRST behind the look-up: the staged word it asks for is then old by the time the top of the loop waits
            // for everything outstanding (it stood last, a dozen instructions in front of that wait)
            br.refill_always();
            __builtin_amdgcn_sched_barrier(0);
            // signs: two bits of the window, x's above y's (where they are: the entry); negating a zero changes nothing, so a
            // bit that is no sign may land on a zero
            const uint32_t sg = __builtin_amdgcn_ubfe(hi_old, 30 + ((e >> 12) & 3) - adv, 2);
            const uint32_t mag = ((e >> 4) & 15) | ((e & 15) << 16);
            const uint32_t neg = ((uint32_t)__builtin_amdgcn_sbfe(sg, 1, 1) & 0xffffu) | ((uint32_t)__builtin_amdgcn_sbfe(sg, 0, 1) & 0xffff0000u);
            const uint32_t val = __builtin_bit_cast(uint32_t, __builtin_bit_cast(s16x2, mag ^ neg) - __builtin_bit_cast(s16x2, neg));
            my_tile[q16] = escaped ? out_esc : val;
            ti = tn; e = en;
        }
            {
                // the tile: 16 pairs of every row, 64 bytes each, four lanes per row
                __builtin_amdgcn_wave_barrier();
                // (the pair counter as the write-out sees it is made opaque: the compiler otherwise keeps the three row addresses as induction
                //  variables and steps them in EVERY iteration -- five instructions of a chain that issues one every five clocks)
                const int p0 = __builtin_amdgcn_readfirstlane(p - 16);
#pragma unroll
                for (int item = lane; item < LANES * 4; item += 64) {
                    const int r = item >> 2, q = item & 3;
                    if ((rows >> r) & 1) {
                        const uint32_t *t4 = tile + r * 17 + 4 * q;
                        const uint4 v = make_uint4(t4[0], t4[1], t4[2], t4[3]);
                        *reinterpret_cast<uint4 *>(is32 + (tid0 + r) * 288 + p0 + 4 * q) = v;
                    }
                }
                __builtin_amdgcn_wave_barrier();
                if (!__any(p < bve || tog != 0)) break;      // (p: the pairs done, a multiple of 16)
            }
        }
#if MP3S_HUF_CLOCKS
        const unsigned long long clk3 = __builtin_readcyclecounter();
#endif
        // what no code word reached: zeros, row by row
        const int rest16 = (288 - p) >> 2;          // 16-byte pieces per row
        if (rest16 > 0)
            for (int r = 0; r < LANES; r++) {
                if (!((rows >> r) & 1)) continue;
                uint4 *z = reinterpret_cast<uint4 *>(is32 + (tid0 + r) * 288 + p);
                for (int i = lane; i < rest16; i += 64) z[i] = make_uint4(0, 0, 0, 0);
            }
#if MP3S_HUF_CLOCKS
        const unsigned long long clk4 = __builtin_readcyclecounter();
        if (has_row) {
            uint32_t *d = is32 + tid * 288 + 280;
            d[0] = (uint32_t)(clk1 - clk0); d[1] = (uint32_t)(clk2 - clk1); d[2] = (uint32_t)(clk3 - clk2); d[3] = (uint32_t)(clk4 - clk3); d[4] = (uint32_t)p;
            d[5] = (uint32_t)__builtin_amdgcn_s_memrealtime();
        }
#endif
    }
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
    // One thread speaks for the group, once: its arrival and its error bits in ONE atomic on the context's 64-bit word
    // (k_sync.hpp); the group that arrives last hands out every group's bits.
    if (threadIdx.x != 0) return;
    const int g = wg_err;
    if (!sync) {          // the caller has zeroed status[0] itself (the overlapped stages: the word travels with the job's inputs)
        if (g) atomicOr(&status[0], g);
        return;
    }
    unsigned all;
    if (arrive_with_bits(sync, (unsigned)g, gridDim.x, &all)) status[0] = (int32_t)all;
}

}  // namespace mp3s
