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

__global__ __launch_bounds__(256) void k_enc_pack(
    const int16_t *__restrict__ ix, const mp3s_gr_out *__restrict__ gr, const int32_t *__restrict__ en, int n_frames,
    int sri, int bri, int whole_slots, const uint32_t *__restrict__ frame_off, const uint8_t *__restrict__ padding,
    uint8_t *__restrict__ mp3, int32_t *__restrict__ scfsi_out, int32_t *__restrict__ status)
{
    __shared__ uint32_t fb[PACK_DW];
    __shared__ uint32_t hc[4][256];
    __shared__ uint8_t hl[4][256];
    __shared__ uint8_t pc[4][324];
    __shared__ int p23f[4];
    const int f = blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < PACK_DW; i += 256) fb[i] = 0;
    for (int i = threadIdx.x; i < 1024; i += 256) {
        (&hc[0][0])[i] = (&c_tab.hcod[0][0])[i];
        (&hl[0][0])[i] = i < 256 ? c_tab.hlen13[i] : (i < 512 ? c_tab.hlen15[i - 256] : (i < 768 ? c_tab.hlen16[i - 512] : c_tab.hlen24[i - 768]));
    }
    __syncthreads();
    const int pad = padding[f];
    if (threadIdx.x == 0) {
        // ---- __resv_frame_end (:1097-1145): all slack of the frame becomes stuffing
        const int bits_per_frame = 8 * (whole_slots + pad);
        const int mean_bits = (bits_per_frame - 288) / 2;
        int p[4], sum = 0;   // emission order e = gr*2 + ch
#pragma unroll
        for (int e = 0; e < 4; e++) { p[e] = gr[((long)f * 2 + (e & 1)) * 2 + (e >> 1)].part2_3_length; sum += p[e]; }
        int stuffing = 2 * mean_bits - sum + ((mean_bits & 1) ? 1 : 0);
        if (stuffing < 0) stuffing = 0;
        if (stuffing) {
            if (p[0] + stuffing < 4095) p[0] += stuffing;
            else
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    if (!stuffing) break;
                    const int extra = 4095 - p[e], now = extra < stuffing ? extra : stuffing;
                    p[e] += now; stuffing -= now;
                }
        }
#pragma unroll
        for (int e = 0; e < 4; e++) p23f[e] = p[e];
        // ---- scfsi (:861-892) from the band energies of the two granules of each channel
        int sc[2][4];
#pragma unroll
        for (int ch = 0; ch < 2; ch++) {
            const long u0 = ((long)f * 2 + ch) * 2, u1 = u0 + 1;
            const int32_t *e0 = en + u0 * 22, *e1 = en + u1 * 22;
            int cond = 2 + (gr[u0].xrmax != 0) + (gr[u1].xrmax != 0);
            int d = e0[21] - e1[21]; if (d < 0) d = -d;
            if (d < 10) cond++;
            int tp = 0;
            for (int s = 0; s < 21; s++) { int a = e0[s] - e1[s]; tp += a < 0 ? -a : a; }
            if (tp < 100) cond++;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                int v = 0;
                if (cond == 6) {
                    const int lo = b == 0 ? 0 : 1 + 5 * b, hi = 6 + 5 * b;
                    int s0 = 0;
                    for (int s = lo; s < hi; s++) { int a = e0[s] - e1[s]; s0 += a < 0 ? -a : a; }
                    v = s0 < 10;
                }
                sc[ch][b] = v;
                scfsi_out[((long)f * 2 + ch) * 4 + b] = v;
            }
        }
        // ---- header + side info (:1281-1337), 288 bits
        uint32_t pos = 0;
        auto put = [&](uint32_t v, int n) { lds_put(fb, pos, v, n); pos += n; };
        put(0x7ff, 11); put(3, 2); put(1, 2); put(1, 1); put(bri, 4); put(sri % 3, 2); put(pad, 1); put(0, 1);
        put(0, 2); put(0, 2); put(0, 1); put(1, 1); put(0, 2);
        put(0, 9); put(0, 3);
#pragma unroll
        for (int ch = 0; ch < 2; ch++)
#pragma unroll
            for (int b = 0; b < 4; b++) put(sc[ch][b], 1);
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const mp3s_gr_out &g = gr[((long)f * 2 + (e & 1)) * 2 + (e >> 1)];
            put(p[e], 12); put(g.big_values, 9); put((uint32_t)(g.quantizer_step + 210) & 0xff, 8); put(0, 4); put(0, 1);
            for (int r = 0; r < 3; r++) put(g.table_select[r], 5);
            put(g.region0_count, 4); put(g.region1_count, 3); put(0, 1); put(0, 1); put(g.count1table_select, 1);
        }
    }
    __syncthreads();

    // ---- main data of this wave's granule*channel (:1394-1446)
    const int e = wave, grn = e >> 1, ch = e & 1;
    const long u = ((long)f * 2 + ch) * 2 + grn;
    const mp3s_gr_out &g = gr[u];
    const int ts0 = g.table_select[0], ts1 = g.table_select[1], ts2 = g.table_select[2], c1sel = g.count1table_select;
    uint32_t ustart = 288;
#pragma unroll
    for (int k = 0; k < 3; k++) ustart += k < e ? (uint32_t)p23f[k] : 0u;
    const int bv = g.big_values, c1 = g.count1;
    const int32_t *sfb = c_tab.sfb_long[sri];
    const int r1s = sfb[g.region0_count + 1], r2s = sfb[g.region0_count + 1 + g.region1_count + 1];
    int xv[10];
    {
        const uint32_t *xp = reinterpret_cast<const uint32_t *>(ix + u * 576 + lane * 10);
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const uint32_t w = (lane * 5 + k) < 288 ? xp[k] : 0u;
            xv[2 * k] = (int)(int16_t)(w & 0xffff); xv[2 * k + 1] = (int)(int16_t)(w >> 16);
        }
    }
    uint8_t *pcw = pc[wave];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const int a = xv[2 * k] < 0 ? -xv[2 * k] : xv[2 * k], b = xv[2 * k + 1] < 0 ? -xv[2 * k + 1] : xv[2 * k + 1];
        pcw[lane * 5 + k] = (uint8_t)((a & 1) | ((b & 1) << 1));
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    uint32_t code0[5], code1[5]; int n0[5], n1[5];
    int tot = 0, bad = 0;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const int p = lane * 5 + k;
        int x = xv[2 * k], y = xv[2 * k + 1];
        const uint32_t sx = x > 0 ? 0 : 1, sy = y > 0 ? 0 : 1;     // util.abs_and_sign: "sign" of 0 is 1, never emitted
        x = x < 0 ? -x : x; y = y < 0 ? -y : y;
        code0[k] = code1[k] = 0; n0[k] = n1[k] = 0;
        if (p < bv) {
            const int i = 2 * p;
            const int ti = i >= r2s ? ts2 : (i >= r1s ? ts1 : ts0);
            if (ti) {
                const int fam = ti == 13 ? 0 : (ti == 15 ? 1 : (ti < 16 ? -1 : (ti < 24 ? 2 : 3)));
                if (fam < 0) bad = 1;
                else if (ti > 15) {
                    const int lb = lin_bits_of(ti);
                    const int lbx = x > 14 ? x - 15 : 0, lby = y > 14 ? y - 15 : 0;
                    const int xx = x > 14 ? 15 : x, yy = y > 14 ? 15 : y;
                    code0[k] = hc[fam][xx * 16 + yy]; n0[k] = hl[fam][xx * 16 + yy];
                    uint32_t ext = 0; int xb = 0;
                    if (xx > 14) { ext |= (uint32_t)lbx; xb += lb; }
                    if (xx != 0) { ext = (ext << 1) | sx; xb += 1; }
                    if (yy > 14) { ext = (ext << lb) | (uint32_t)lby; xb += lb; }
                    if (yy != 0) { ext = (ext << 1) | sy; xb += 1; }
                    code1[k] = ext; n1[k] = xb;
                } else {
                    uint32_t c = hc[fam][x * 16 + y]; int nb = hl[fam][x * 16 + y];
                    if (x != 0) { c = (c << 1) | sx; nb += 1; }
                    if (y != 0) { c = (c << 1) | sy; nb += 1; }
                    code0[k] = c; n0[k] = nb;
                }
            }
        } else if (p < bv + 2 * c1) {
            // count1 quadruple = this pair (v, w) + the next one (x, y); code word with the first pair (E13)
            if (!((p - bv) & 1)) {
                const int q = (x & 1) | ((y & 1) << 1) | (((int)pcw[p + 1] & 3) << 2);
                if (c1sel) { code0[k] = 15 - q; n0[k] = 4; }
                else { code0[k] = c_tab.hcod_c1a[q]; n0[k] = c_tab.hlen_c1a[q]; }
            }
            uint32_t s = 0; int nb = 0;
            if (x) { s = sx; nb = 1; }
            if (y) { s = (s << 1) | sy; nb += 1; }
            code1[k] = s; n1[k] = nb;
        }
        tot += n0[k] + n1[k];
    }
    // exclusive prefix of the lanes' bit counts
    int incl = tot;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    const int huff_bits = __shfl(incl, 63, 64);
    uint32_t pos = ustart + (uint32_t)(incl - tot);
#pragma unroll
    for (int k = 0; k < 5; k++) {
        lds_put(fb, pos, code0[k], n0[k]); pos += n0[k];
        lds_put(fb, pos, code1[k], n1[k]); pos += n1[k];
    }
    // stuffing with ones up to part2_3_length (:1433-1446)
    const int p23 = p23f[e];
    if (huff_bits > p23 || bad) { if (lane == 0) atomicOr(status, bad ? MP3S_PS_BAD_TABLE : MP3S_PS_OVERFLOW); }
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
    if ((int)threadIdx.x < head) mp3[off + threadIdx.x] = (uint8_t)byte_at(threadIdx.x);
    uint32_t *outw = reinterpret_cast<uint32_t *>(mp3 + off + head);
    for (int j = threadIdx.x; j < n_dw; j += 256) {
        const int k = head + 4 * j;
        outw[j] = byte_at(k) | (byte_at(k + 1) << 8) | (byte_at(k + 2) << 16) | (byte_at(k + 3) << 24);
    }
    if ((int)threadIdx.x < nbytes - tail0) mp3[off + tail0 + threadIdx.x] = (uint8_t)byte_at(tail0 + threadIdx.x);
}

}  // namespace mp3s
