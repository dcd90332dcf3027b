// Bit packing on the device (gfx950).  Included by mp3s_device.hip only.
//
//   k_enc_pack : __resv_frame_end + __format_bitstream (reference encoder/MP3_Encoder.py:1097-1145, 1266-1547) for
//                every frame of the batch.  The reference writes bits serially through a 32-bit cache; here the
//                position of every code word is known up front: a frame is exactly 8*(slots+padding) bits (all
//                slack becomes stuffing, E6), a granule*channel starts after the part2_3_lengths before it, and inside
//                it the offset of a pair is the prefix sum of the code lengths.  One workgroup per frame, one
//                wavefront per granule*channel, lane = 5 consecutive pairs, bits OR-ed into an LDS image of the frame.
#pragma once

namespace mp3s {

constexpr int PACK_DW = 372;          // LDS image of one frame: 1441 bytes max (320 kbps @ 32 kHz) + slack
constexpr int PACK_FAM = 260;         // a book family's 256 words + 4: small values of different families (what a wave mostly reads at once) in different banks
constexpr int PACK_PT_QUAD = 4 * PACK_FAM, PACK_PT_NONE = PACK_PT_QUAD + 32, PACK_PT = PACK_PT_NONE + 1;   // the code-word table of k_enc_pack
#define MP3S_PS_OVERFLOW 1            // Huffman bits exceed part2_3_length (cannot happen for rate-loop output)
#define MP3S_PS_BAD_TABLE 2           // a code book this encoder never selects

__device__ __forceinline__ void lds_put(uint32_t *fb, uint32_t pos, uint32_t val, int n)
{
    if (n <= 0) return;
    const uint32_t d = pos >> 5, o = pos & 31;
    const uint64_t w = (uint64_t)val << (64 - o - n);
    atomicOr(&fb[d], (uint32_t)(w >> 32));
    if ((uint32_t)w) atomicOr(&fb[d + 1], (uint32_t)w);
}

// wave sum of a small non-negative value over the lanes selected by `on`
__device__ __forceinline__ int pack_wave_sum(int v, bool on) { return (int)wave_add_u32(on ? (uint32_t)v : 0u); }

// Persistent workgroups: the code tables are staged in LDS once, then the group walks frames f, f + gridDim.x, ...
// Everything the four waves need about the frame (part2_3_length after stuffing, start offsets) is wave-uniform
// scalar work done redundantly by each wave; the header / side info bits are written by the wave they describe.
__global__ __launch_bounds__(256) void k_enc_pack(
    const int16_t *__restrict__ ix, const mp3s_gr_out *__restrict__ gr, const int32_t *__restrict__ en, int n_frames,
    int sri, int bri, int whole_slots, const uint32_t *__restrict__ frame_off, const uint8_t *__restrict__ padding,
    uint8_t *__restrict__ mp3, int32_t *__restrict__ scfsi_out, int32_t *__restrict__ status, int32_t *__restrict__ sync,
    int f_begin /* frames [f_begin, n_frames) of the batch: a one-file call packs its last chunk in two launches, the first half on its way
                   down while the second is packed */)
{
    __shared__ uint32_t fb3[3][PACK_DW];   // frame images, rotating: written / being copied out / being cleared
    // code words of every kind in ONE table, code | length << 24 (a code has at most 19 bits): [book family][min(x,15)][min(y,15)] for the four
    // families of big-value books (13, 15, 16.., 24..), then the quadruples of count1 table A and B, then an empty word -- a pair picks its INDEX
    // (big value / first pair of a quadruple / nothing) and reads once; code and length used to come from two to four reads and as many selects
    __shared__ uint32_t pt[PACK_PT];
    __shared__ int wg_err;                 // the group's error bits
    const int wave0 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane0 = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 1024; i += 256) {
        const uint32_t len = i < 256 ? c_tab.hlen13[i] : (i < 512 ? c_tab.hlen15[i - 256] : (i < 768 ? c_tab.hlen16[i - 512] : c_tab.hlen24[i - 768]));
        pt[(i >> 8) * PACK_FAM + (i & 255)] = (&c_tab.hcod[0][0])[i] | (len << 24);
    }
    if (threadIdx.x < 16) {
        pt[PACK_PT_QUAD + threadIdx.x] = (uint32_t)c_tab.hcod_c1a[threadIdx.x] | ((uint32_t)c_tab.hlen_c1a[threadIdx.x] << 24);
        pt[PACK_PT_QUAD + 16 + threadIdx.x] = (15u - threadIdx.x) | (4u << 24);      // table B: four bits, the complement
    }
    if (threadIdx.x == 0) pt[PACK_PT_NONE] = 0;
    for (int i = threadIdx.x; i < 3 * PACK_DW; i += 256) (&fb3[0][0])[i] = 0;
    if (threadIdx.x == 0) wg_err = 0;
    __syncthreads();
    int rot = 0;
    for (int f = f_begin + blockIdx.x; f < n_frames; f += gridDim.x) {
    // The wave number and a copy of the lane number used ONLY in comparisons are made opaque once per frame: what follows from them is
    // invariant over the frame loop, and kept across it every lane predicate (lane < 22, lane == 0 ...) is a mask in two scalar
    // registers and every wave-derived scalar one more -- 71 scalar spills, read back lane by lane (v_readlane) in front of each use, a
    // fifth of the kernel's vector instructions.  A predicate recomputed is one comparison, a scalar recomputed is scalar arithmetic;
    // the lane number's ARITHMETIC (addresses, pair numbers) stays invariant and hoisted (made opaque too, round 4's first try, it cost
    // more than the spills).
    const int lane = lane0, tid = (int)threadIdx.x;
    int lane_p = lane0, wave = wave0;
    asm volatile("" : "+v"(lane_p), "+s"(wave));
    // One barrier per frame.  This frame's image is fb3[rot]; the image of the frame before last -- every thread
    // finished copying it out before the previous barrier -- is cleared now and is ready after this frame's barrier.
    uint32_t *fb = fb3[rot];
    {
        uint32_t *cl = fb3[rot == 2 ? 0 : rot + 1];
        for (int i = tid; i < PACK_DW; i += 256) cl[i] = 0;
    }
    const int pad = padding[f];
    // ---- __resv_frame_end (:1097-1145): all slack of the frame becomes stuffing; emission order e = gr*2 + ch
    int p23v[4];
    {
        const int bits_per_frame = 8 * (whole_slots + pad);
        const int mean_bits = (bits_per_frame - 288) / 2;
        int sum = 0;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            p23v[e] = __builtin_amdgcn_readfirstlane(gr[((long)f * 2 + (e & 1)) * 2 + (e >> 1)].part2_3_length);
            sum += p23v[e];
        }
        int stuffing = 2 * mean_bits - sum + ((mean_bits & 1) ? 1 : 0);
        if (stuffing < 0) stuffing = 0;
        if (stuffing) {
            if (p23v[0] + stuffing < 4095) p23v[0] += stuffing;
            else
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int extra = 4095 - p23v[e], now = extra < stuffing ? extra : stuffing;
                    p23v[e] += now; stuffing -= now;
                }
        }
    }
    const int e = wave, grn = e >> 1, ch = e & 1;
    const long u = ((long)f * 2 + ch) * 2 + grn;
    const mp3s_gr_out &g = gr[u];
    if (wave < 2) {
        // ---- scfsi of channel `wave` (:861-892) from the band energies of its two granules: lane s holds band s
        const long u0 = ((long)f * 2 + wave) * 2, u1 = u0 + 1;
        int a = 0, tot21 = 0;
        if (lane_p < 22) {
            a = en[u0 * 22 + lane] - en[u1 * 22 + lane];
            a = a < 0 ? -a : a;
        }
        tot21 = __builtin_amdgcn_readlane(a, 21);
        // the five sums over lanes (all 21 bands; bands 0-5, 6-10, 11-15, 16-20) from ONE inclusive scan of the rows' 16 lanes + the row in
        // front: five wave reductions before
        int sc = lane_p < 21 ? a : 0;
        sc += __builtin_amdgcn_update_dpp(0, sc, 0x111, 0xf, 0xf, false);   // row_shr:1
        sc += __builtin_amdgcn_update_dpp(0, sc, 0x112, 0xf, 0xf, false);   // row_shr:2
        sc += __builtin_amdgcn_update_dpp(0, sc, 0x114, 0xf, 0xf, false);   // row_shr:4
        sc += __builtin_amdgcn_update_dpp(0, sc, 0x118, 0xf, 0xf, false);   // row_shr:8  (lane l: sum of its row's lanes up to l)
        const int row0 = __builtin_amdgcn_readlane(sc, 15);                 // bands 0..15
        const int p5 = __builtin_amdgcn_readlane(sc, 5), p10 = __builtin_amdgcn_readlane(sc, 10), q20 = __builtin_amdgcn_readlane(sc, 20);   // (q20: bands 16..20)
        const int tp = row0 + q20;
        int cond = 2 + (gr[u0].xrmax != 0) + (gr[u1].xrmax != 0);
        if (tot21 < 10) cond++;
        if (tp < 100) cond++;
        const int band_sum[4] = {p5, p10 - p5, row0 - p10, q20};
        uint32_t scbits = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int s0 = band_sum[b];
            const int v = (cond == 6 && s0 < 10) ? 1 : 0;
            scbits = (scbits << 1) | (uint32_t)v;
            if (lane_p == 0) scfsi_out[((long)f * 2 + wave) * 4 + b] = v;
        }
        if (lane_p == 0) {
            lds_put(fb, 44 + 4 * wave, scbits, 4);
            if (wave == 0) {
                // ---- frame header (:1281-1300): sync 11, version 2, layer 2, no CRC 1, bitrate 4, rate 2, padding 1, ext 1,
                //      mode 2, mode ext 2, copyright 1, original 1, emphasis 2;  main_data_begin 9 + private 3 stay zero
                const uint32_t h = (0x7ffu << 21) | (3u << 19) | (1u << 17) | (1u << 16) | ((uint32_t)bri << 12) |
                                   ((uint32_t)(sri % 3) << 10) | ((uint32_t)pad << 9) | (1u << 2);
                fb[0] = h;     // nothing else writes the first dword
            }
        }
    }
    if (lane_p < 4) {
        // ---- side info of this wave's granule*channel (:1305-1337), 59 bits at 52 + 59 e: its four fields by four lanes, one put each
        //      (one lane putting them one after the other had the whole wave step through four times the instructions)
        const uint32_t f0 = ((uint32_t)p23v[e] << 9) | (uint32_t)g.big_values;                                            // 21 bits
        const uint32_t f1 = (((uint32_t)(g.quantizer_step + 210) & 0xff) << 5);                                           // 13: global_gain 8, scalefac_compress 4, window_switching 1
        const uint32_t f2 = ((uint32_t)g.table_select[0] << 10) | ((uint32_t)g.table_select[1] << 5) | (uint32_t)g.table_select[2];   // 15
        const uint32_t f3 = ((uint32_t)g.region0_count << 6) | ((uint32_t)g.region1_count << 3) | (uint32_t)g.count1table_select;     // 10: + preflag, scalefac_scale = 0
        const uint32_t v = lane_p == 0 ? f0 : (lane_p == 1 ? f1 : (lane_p == 2 ? f2 : f3));
        const int n = lane_p == 0 ? 21 : (lane_p == 1 ? 13 : (lane_p == 2 ? 15 : 10));
        const uint32_t at = lane_p == 0 ? 0u : (lane_p == 1 ? 21u : (lane_p == 2 ? 34u : 49u));
        lds_put(fb, 52 + 59 * (uint32_t)e + at, v, n);
    }

    // ---- main data of this wave's granule*channel (:1394-1446)
    const int ts0 = g.table_select[0], ts1 = g.table_select[1], ts2 = g.table_select[2], c1sel = g.count1table_select;
    uint32_t ustart = 288;
#pragma unroll
    for (int k = 0; k < 3; k++) ustart += k < e ? (uint32_t)p23v[k] : 0u;
    const int bv = g.big_values, c1 = g.count1;
    const int32_t *sfb = c_tab.sfb_long[sri];
    const int r1s = sfb[g.region0_count + 1], r2s = sfb[g.region0_count + 1 + g.region1_count + 1];
    int xv[10];
    {
        const uint32_t *xp = reinterpret_cast<const uint32_t *>(ix + u * 576 + lane * 10);
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const uint32_t w = (lane_p * 5 + k) < 288 ? xp[k] : 0u;
            xv[2 * k] = (int)(int16_t)(w & 0xffff); xv[2 * k + 1] = (int)(int16_t)(w >> 16);
        }
    }
    // the low bits of the lane's pairs and of the pair behind them (the first pair of the lane above): a count1 quadruple
    // is this pair + the next one
    uint32_t c2[6];
#pragma unroll
    for (int k = 0; k < 5; k++) c2[k] = (uint32_t)((xv[2 * k] & 1) | ((xv[2 * k + 1] & 1) << 1));   // (two's complement: the low bit of |v| is the low bit of v)
    c2[5] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c2[0], 0x130, 0xf, 0xf, false);              // wave_shl:1
    uint32_t code0[5], code1[5]; int n0[5], n1[5];
    int tot = 0;
    // What differs between the regions is wave-uniform: one scalar word per region -- Huffman-length family (books 13, 15,
    // 16.., 24..) | linbits << 2 | "book in use" << 6 -- picked per lane; everything else is the same arithmetic for every
    // pair, without a branch and (round 4) nearly without selects: the escape and sign bits (MP3_Encoder.py:1452-1500;
    // books 13 and 15 have no linbits, so their value 15 emits none) are field extractions of a width that is zero where
    // nothing is emitted, and the pair's own code word -- its book's, or the quadruple's with the first pair of a quad (E13;
    // :1502-1547), or none -- is one read of `pt`.
    uint32_t Kr[3];
    bool any_bad = false;
    {
        const int tsr[3] = {ts0, ts1, ts2};
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const int ti = tsr[r];
            const int fam = ti == 13 ? 0 : (ti == 15 ? 1 : (ti < 24 ? 2 : 3));
            Kr[r] = ti ? (uint32_t)fam | ((uint32_t)lin_bits_of(ti) << 2) | (1u << 6) : 0u;
            // a book this encoder never selects, in a region that holds pairs
            const int lo = r == 0 ? 0 : (r == 1 ? r1s : r2s), hi = r == 0 ? r1s : (r == 1 ? r2s : 576);
            any_bad |= ti != 0 && ti != 13 && ti != 15 && ti < 16 && lo < 2 * bv && lo < hi;
        }
    }
    const uint32_t quad_base = (uint32_t)(PACK_PT_QUAD + (c1sel ? 16 : 0));
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const int p = lane * 5 + k;
        const int vx = xv[2 * k], vy = xv[2 * k + 1];
        const uint32_t x = (uint32_t)(vx < 0 ? -vx : vx), y = (uint32_t)(vy < 0 ? -vy : vy);
        const int i = 2 * p;
        const uint32_t K = i >= r2s ? Kr[2] : (i >= r1s ? Kr[1] : Kr[0]);
        const uint32_t Ke = p < bv ? K : 0u;                              // != 0: a big-value pair of a book in use
        const bool in_c1 = (uint32_t)(p - bv) < (uint32_t)(2 * c1);
        const uint32_t lb = __builtin_amdgcn_ubfe(Ke, 2, 4);
        // escape and sign bits: [x - 15 in lb bits] [sign of x] [y - 15 in lb bits] [sign of y]; a value of 0 has no sign bit,
        // a value below 15 (or a book without linbits) no escape bits: fields of width 0
        const uint32_t lbx = x > 14 ? lb : 0u, lby = y > 14 ? lb : 0u;
        const uint32_t nzx = x < 1 ? x : 1u, nzy = y < 1 ? y : 1u;
        uint32_t ext = __builtin_amdgcn_ubfe(x - 15u, 0, lbx);
        ext = (ext << nzx) | ((uint32_t)vx >> 31);
        ext = (ext << lby) | __builtin_amdgcn_ubfe(y - 15u, 0, lby);
        ext = (ext << nzy) | ((uint32_t)vy >> 31);
        const uint32_t xb = lbx + nzx + lby + nzy;
        const bool inside = Ke != 0 || in_c1;
        // the pair's own code word
        const uint32_t xx = x > 14 ? 15u : x, yy = y > 14 ? 15u : y;
        const uint32_t q = (c2[k] | (c2[k + 1] << 2)) & 15u;
        const bool quad = in_c1 && !((p - bv) & 1);
        const uint32_t tix = Ke != 0 ? (Ke & 3u) * (uint32_t)PACK_FAM + xx * 16u + yy : (quad ? quad_base + q : (uint32_t)PACK_PT_NONE);
        const uint32_t word = pt[tix];
        code0[k] = word & 0xffffffu;
        n0[k] = (int)(word >> 24);
        code1[k] = inside ? ext : 0u;
        n1[k] = inside ? (int)xb : 0;
        tot += n0[k] + n1[k];
    }
    // exclusive prefix of the lanes' bit counts
    const int incl = (int)wave_scan_u32((uint32_t)tot);
    const int huff_bits = __builtin_amdgcn_readlane(incl, 63);
    // The lane's code words are consecutive in the stream: they are gathered in a 64-bit register (in front of them as many
    // zero bits as the lane's start lies behind a dword boundary) and leave as dwords of the frame image.  A word has at most
    // 28 bits and fewer than 32 are pending in front of it.  Round 4: after EVERY word the dword the pending bits begin in is
    // OR-ed into the image, finished or not (an unfinished one is OR-ed again, with more bits, by the next word: the same
    // bits twice do no harm), and position and count move on by arithmetic -- the flush used to be a per-lane branch, ten
    // of them per frame.
    if (tot) {   // (the lanes behind the last value hold nothing and all stand on ONE dword: 64 ORs of zero into one address, eleven times)
        const uint32_t pos0 = ustart + (uint32_t)(incl - tot);
        uint32_t d = pos0 >> 5;
        uint32_t cnt = pos0 & 31u;
        uint64_t top = 0;                                                       // the pending bits, from bit 63 down: cnt of them behind the dword boundary
        auto append = [&](uint32_t v, int n) {
            top |= (uint64_t)v << ((64u - cnt - (uint32_t)n) & 63u);             // cnt + n <= 59; (n == 0: v == 0)
            cnt += (uint32_t)n;
            atomicOr(&fb[d], (uint32_t)(top >> 32));
            const uint32_t adv = cnt >> 5;                                      // 0 or 1
            top <<= 32u * adv; d += adv; cnt &= 31u;
        };
#pragma unroll
        for (int k = 0; k < 5; k++) { append(code0[k], n0[k]); append(code1[k], n1[k]); }
        atomicOr(&fb[d], (uint32_t)(top >> 32));                                // what the last word left behind a dword boundary
    }
    // stuffing with ones up to part2_3_length (:1433-1446)
    const int p23 = e == 0 ? p23v[0] : (e == 1 ? p23v[1] : (e == 2 ? p23v[2] : p23v[3]));
    if (huff_bits > p23 || any_bad) { if (lane_p == 0) atomicOr(&wg_err, any_bad ? MP3S_PS_BAD_TABLE : MP3S_PS_OVERFLOW); }   // (LDS)
    else
        for (uint32_t s = ustart + huff_bits + 32u * lane; s < ustart + (uint32_t)p23; s += 64u * 32u) {
            const uint32_t n = (ustart + p23 - s) < 32u ? (ustart + p23 - s) : 32u;
            lds_put(fb, s, n == 32 ? 0xffffffffu : ((1u << n) - 1), (int)n);
        }
    __syncthreads();
    // ---- LDS image -> global bytes: unaligned head / tail as bytes, the aligned interior as dwords
    const int nbytes = whole_slots + pad;
    const uint32_t off = frame_off[f];
    auto byte_at = [&](int k) -> uint32_t { return (fb[k >> 2] >> (24 - 8 * (k & 3))) & 0xffu; };
    const int head = (int)((4u - (off & 3u)) & 3u) < nbytes ? (int)((4u - (off & 3u)) & 3u) : nbytes;
    const int n_dw = (nbytes - head) >> 2, tail0 = head + 4 * n_dw;
    if (tid < head) mp3[off + tid] = (uint8_t)byte_at(tid);
    uint32_t *outw = reinterpret_cast<uint32_t *>(mp3 + off + head);
    {
        // bytes head + 4j .. + 3 of the image (big-endian in its dwords) as one little-endian dword: a funnel shift over two image
        // dwords and a byte swap (byte by byte it was two dozen instructions per dword)
        const int hq = head >> 2, hs = 8 * (head & 3);               // (wave-uniform)
        for (int j = tid; j < n_dw; j += 256) {
            const uint32_t hi = fb[hq + j], lo = fb[hq + j + 1];     // (PACK_DW leaves slack behind the last byte)
            const uint32_t be = hs ? __builtin_amdgcn_alignbit(hi, lo, 32 - hs) : hi;
            outw[j] = __builtin_bswap32(be);
        }
    }
    if (tid < nbytes - tail0) mp3[off + tail0 + tid] = (uint8_t)byte_at(tail0 + tid);
    rot = rot == 2 ? 0 : rot + 1;
    }   // frames
    // The caller's status word is written once, by the workgroup that finishes last, from the context's 64-bit word {finished
    // groups | error bits} (k_sync.hpp: one atomic per group, no fill kernel in front of every launch).
    // sync == null: the caller has zeroed *status itself (the overlapped stages, where the word travels with the job's inputs): two
    // thousand arrivals on one counter are 0.1 ms of serialised atomics when the groups finish together, as they do on a short batch.
    __syncthreads();
    if (threadIdx.x != 0) return;
    const int g = wg_err;
    if (!sync) {
        if (g) atomicOr(status, g);
        return;
    }
    unsigned all;
    if (arrive_with_bits(sync, (unsigned)g, gridDim.x, &all)) *status = (int32_t)all;
}

}  // namespace mp3s
