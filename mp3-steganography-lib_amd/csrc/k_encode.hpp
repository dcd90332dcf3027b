// Encode transform kernels (gfx950).  Included by mp3s_device.hip only.
//
//   k_enc_analysis : __replace_samples + window_filter_sub_band (reference encoder/MP3_Encoder.py:
//                    321-370, 751-758) + the odd-slot/odd-band sign flip (:676-679).
//                    lane = time slot (32 new PCM samples); its 512-sample window is read from an
//                    LDS tile (rows padded to 80 B so 16-byte reads are conflict free), enwindow and
//                    the 32x64 matrix fl are scalar operands.
//   k_enc_mdct     : 36->18 MDCT with the fused sine window + alias butterflies (:681-744).
//                    lane = (channel, band); cos_l is a scalar operand, the butterfly partner comes
//                    from the neighbouring lane.
//
// Everything is the reference's int32 fixed point: mul = (a*b)>>32 per product (v_mul_hi_i32), then
// wrapping int32 adds -- integer sums are order independent, each product's floor is not, so every
// product is floored on its own exactly as the reference does.
#pragma once
#include <type_traits>
#include "analysis_plan.h"

namespace mp3s {

// (v * s) >> 32 with the wave-uniform factor in an SGPR (v_mul_hi_i32 and v_mad_i64_i32 both issue at full rate on
// gfx950; measured in tools/ubench/valu_rates.hip).  Spelled out where the compiler would otherwise widen the vector
// operands to 64 bits once and keep their sign words alive (twice the registers).  Feed it from vector-typed scalar
// loads: single dwords named in an asm operand are each fetched by their own s_load_dword.
__device__ __forceinline__ int32_t mulhi_vs(int32_t v, int32_t s)
{
    int32_t r;
    asm("v_mul_hi_i32 %0, %1, %2" : "=v"(r) : "v"(v), "s"(s));
    return r;
}
typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
typedef int32_t i32x16 __attribute__((ext_vector_type(16)));
typedef int32_t i32x8 __attribute__((ext_vector_type(8)));
// f(integral_constant<int, N>) for N = B .. E - 1, unrolled at compile time
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (B < E) { f(std::integral_constant<int, B>{}); static_for<B + 1, E>(f); }
}

constexpr int ENC_ROW = 34;   // int16 per LDS row: 32 samples + 2 pad: 17 dwords, so the 64 lanes' rows start in 32 different banks twice over
                              // (the window sums read one sample per lane and instruction)
constexpr int ENC_LDS_DW = 79 * 17 + 1;   // per-wave LDS in dwords: the PCM tile (79 rows of 17 dwords), then -- twice -- a 64 x 16 half of the output tile (rows of 17):
                                      // 5.4 KB per wave, 21.5 per workgroup: five workgroups = five waves per SIMD fit a CU's LDS (8.4 KB per wave and four until round 4)

// The analysis of 64 consecutive slots t0 .. t0 + 63 of channel ch by one wave (lane = slot).  `lds`: ENC_LDS_DW dwords of the wave's own.
// TILE == false (k_enc_analysis): the subband samples go to SB in device memory, int32 [ch][Ts][32] (a 128-byte row per slot, bands at sb_pos), staged
// through `lds` so that they leave as 16-byte pieces of the rows.  TILE == true (k_enc_fused): they go to `rows` in LDS, row r of the
// workgroup's tile at rows + r * ENC_TROW with r = the lane's slot - tile_t0, for the slots [lo, hi) only; `between`() runs once when the
// wave has read the last sample of its PCM staging (the fused kernel's workgroup barrier: the staging lies inside the tile).
constexpr int ENC_TROW = 33;  // dwords per row of the fused kernel's tile (32 bands + 1: the slot lanes' writes and the band lanes' reads are conflict free)
template <bool TILE, class Between>
__device__ __forceinline__ void enc_analysis_wave(const int16_t *__restrict__ pcm, const mp3s_frame_hdr *__restrict__ hdr, long Ts, int ch, long t0, int lane,
                                                  uint32_t *lds, int32_t *__restrict__ SB, uint32_t *rows, long tile_t0, long lo, long hi, Between between)
{
    int16_t *tw = reinterpret_cast<int16_t *>(lds);
    // stage rows t0-15 .. t0+63 of channel ch (zeros outside the batch): the tile is one contiguous piece of the interleaved PCM,
    // read 16 bytes (four samples of both channels) per lane and trip, the wave's channel picked out by two byte permutes
    // (sample by sample it was forty trips of a dozen instructions: a tenth of the kernel's vector instructions)
    if ((reinterpret_cast<uintptr_t>(pcm) & 15u) == 0) {
        const int c0 = (int)(t0 - 15) * 8, c1 = (int)Ts * 8;     // the tile's first chunk and the batch's end, in 16-byte chunks (Ts < 2^28)
        const uint4 *gp = reinterpret_cast<const uint4 *>(pcm);
        const uint32_t sel = ch ? 0x07060302u : 0x05040100u;
        for (int c = lane; c < 79 * 8; c += 64) {
            const int gc = c0 + c;
            uint4 q = make_uint4(0, 0, 0, 0);
            if (gc >= 0 && gc < c1) q = gp[gc];
            uint2 o;
            o.x = __builtin_amdgcn_perm(q.y, q.x, sel);
            o.y = __builtin_amdgcn_perm(q.w, q.z, sel);
            uint32_t *dst = reinterpret_cast<uint32_t *>(tw + (c >> 3) * ENC_ROW + (c & 7) * 4);   // (rows are 4-byte aligned)
            dst[0] = o.x; dst[1] = o.y;
        }
    } else {                                                    // (a caller's PCM that does not start on a 16-byte boundary)
        for (int e = lane; e < 79 * 32; e += 64) {
            const int r = e >> 5, s = e & 31;
            const long row = t0 - 15 + r;
            int16_t v = 0;
            if (row >= 0 && row < Ts) v = pcm[(row * 32 + s) * 2 + ch];
            tw[r * ENC_ROW + s] = v;
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    const long t = t0 + lane;
    const bool valid = t >= 0 && t < Ts;
    long s0 = 0;
    if (valid) s0 = (long)hdr[t / 36].stream_first * 36;

    // ---- y[i] = sum_k mul(x[(off + i + 64k) & 511], enwindow[i + 64k])   (MP3_Encoder.py:336-354)
    // sample 32t+31-m sits `m>>5` rows back at column 31-(m&31); one k step = rows 2k and 2k+1 back
    int32_t y[64];
    // A stream that starts inside the wave's window (its first slots see zeros where the ring x was still empty) is rare:
    // the usual case reads the tile without the per-row masks.
    const bool starts_inside = __ballot(valid && (t - 15) < s0) != 0;
    // A sample reaches its multiplier as (sample << 16) (MP3_Encoder.py:341: the ring holds the PCM in the high half).  It is READ that way:
    // ds_read_u16_d16_hi puts the 16 bits into the high half of a register and leaves the low half (zero throughout) alone -- one LDS
    // instruction per sample instead of a 16-byte read per eight samples plus a shift or a mask for each: a ninth of the kernel's vector
    // instructions moved to the LDS port, which the kernel hardly used.  The loads are inline assembly (the compiler has no pattern for
    // them that keeps the low half): they, the wait behind a group of sixteen and the products are chained through the registers.
    const uint32_t tw_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const int16_t *)tw;
    uint32_t xs[16];
#pragma unroll
    for (int i = 0; i < 16; i++) xs[i] = 0;
#define MP3S_LDS_HI16(reg, addr, off) asm volatile("ds_read_u16_d16_hi %0, %1 offset:%2" : "+v"(reg) : "v"(addr), "n"(off))
    auto window_sums = [&](auto masked) {
    // two k steps per trip: every y gets two products at a time, added by ONE v_add3_u32 (a product per trip cost an add each)
    auto trip = [&](int k, auto first) {                   // (the first trip sets the sums: no 64 registers to clear in front of the loop)
        const i32x16 *ewa = reinterpret_cast<const i32x16 *>(c_tab.enwindow + 64 * k), *ewb = ewa + 4;   // enwindow[64 (k + 1) ..]
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const i32x16 a0 = ewa[2 * h], a1 = ewa[2 * h + 1];   // enwindow[64k + 32h .. +31]
            const i32x16 b0 = ewb[2 * h], b1 = ewb[2 * h + 1];   // enwindow[64(k+1) + 32h .. +31]
            const int back_a = 2 * k + h, back_b = back_a + 2;  // rows back from the lane's own row
            const bool in_a = (t - back_a) >= s0, in_b = (t - back_b) >= s0;     // ring x starts zeroed (MP3_Encoder.py:532-534)
            // sample 32t+31-m sits `m>>5` rows back at column 31-(m&31); one k step = rows 2k and 2k+1 back
            const uint32_t ra = tw_lds + (uint32_t)((lane + 15 - back_a) * ENC_ROW * 2), rb = ra - 2 * ENC_ROW * 2;
#pragma unroll
            for (int cb = 0; cb < 4; cb++) {
#pragma unroll
                for (int e = 0; e < 8; e++) {                   // columns 8 cb .. 8 cb + 7 of both rows
                    MP3S_LDS_HI16(xs[e], ra, (cb * 8 + e) * 2);
                    MP3S_LDS_HI16(xs[8 + e], rb, (cb * 8 + e) * 2);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xs[0]), "+v"(xs[1]), "+v"(xs[2]), "+v"(xs[3]), "+v"(xs[4]), "+v"(xs[5]), "+v"(xs[6]), "+v"(xs[7]),
                             "+v"(xs[8]), "+v"(xs[9]), "+v"(xs[10]), "+v"(xs[11]), "+v"(xs[12]), "+v"(xs[13]), "+v"(xs[14]), "+v"(xs[15]));
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const int col = cb * 8 + e, c = 31 - col;                 // coefficient within this half
                    int32_t xa = (int32_t)xs[e], xb = (int32_t)xs[8 + e];
                    if (decltype(masked)::value && !in_a) xa = 0;
                    if (decltype(masked)::value && !in_b) xb = 0;
                    const int32_t two = mulhi_vs(xa, c < 16 ? a0[c & 15] : a1[c & 15]) + mulhi_vs(xb, c < 16 ? b0[c & 15] : b1[c & 15]);
                    if (decltype(first)::value) y[h * 32 + c] = two; else y[h * 32 + c] += two;
                }
                __builtin_amdgcn_sched_barrier(0);   // sixteen products at a time: bounds the scheduler's hoisting
            }
        }
    };
    trip(0, std::true_type{});
#pragma unroll 1
    for (int k = 2; k < 8; k += 2) trip(k, std::false_type{});
    };
    if (starts_inside) window_sums(std::true_type{}); else window_sums(std::false_type{});
#undef MP3S_LDS_HI16
    // ---- s[sb] = sum_j mul(fl[sb][j], y[j])   (MP3_Encoder.py:358-368); four bands per pass: {p, 15 - p, 16 + p, 31 - p}.  The filter is
    // cos((2 sb + 1)(16 - j) pi / 64): within such a set the coefficients of a column repeat themselves (16 - j = 2 mod 4: in two pairs; a multiple
    // of 4: all four; j = 48: zeros) wherever the table's rounding has not pulled them two units apart -- and a product with the same coefficient
    // is the same product: computed once (analysis_plan.h: 1 516 products per slot instead of 2 048; each product still floored on its own, the
    // sums wrap -- bit for bit the reference's).  The plan is compiled in, so the passes are unrolled; the host checks its table against it.
    // Results are staged in the wave's LDS region as [slot][16 bands] (17-dword rows) and written out as 16-byte pieces of the rows: passes
    // 0..3 complete bands 0-3, 12-15, 16-19, 28-31, passes 4..7 the other four pieces.
    const bool odd_slot = (t & 1) != 0;      // slot-in-granule parity == slot parity (18 is even)
    __builtin_amdgcn_wave_barrier();         // every lane is done reading the PCM tile
    between();
    uint32_t *ot = lds;
    uint32_t *out = TILE ? nullptr : reinterpret_cast<uint32_t *>(SB) + ((long)ch * Ts + t0) * 32;
    const int n_rows = (Ts - t0) < 64 ? (int)(Ts - t0) : 64;
    // (TILE) the lane's row of the tile; a lane outside [lo, hi) writes a row nobody reads (the tile's spare row)
    uint32_t *trow = TILE ? rows + (valid && t >= lo && t < hi ? (t - tile_t0) : -1) * ENC_TROW : nullptr;
    // The coefficients reach the multiplier as scalar operands, eight columns of the pass's four rows at a time (4 x s_load_dwordx8), and the NEXT
    // block's are asked for before this block's products start: every block begins by waiting for its own (asked for a block ago; scalar loads
    // return out of order, so the wait is for everything outstanding and comes BEFORE the next request goes out).  Until round 4's last change a
    // block asked for its coefficients and waited for them on the spot, 32 times per wave: 45 % of the kernel's wave cycles stood at a wait.
    int32_t a[4] = {0, 0, 0, 0};
    i32x8 cur[4], nxt[4];
    // (the table's address once, in a scalar pair the compiler cannot re-derive: every request is then base + immediate offset -- it built the
    // address of each of the 256 requests from the program counter, three scalar instructions apiece)
    typedef const int32_t __attribute__((address_space(4))) *fl_ptr;
    fl_ptr flp = (fl_ptr)&c_tab.fl[0][0];
    asm volatile("" : "+s"(flp));
    auto load_block = [&](auto nc, i32x8 (&c)[4]) {
        constexpr int N = decltype(nc)::value, P = N >> 3, B = N & 7;
        constexpr int O[4] = {P, 15 - P, 16 + P, 31 - P};
#pragma unroll
        for (int o = 0; o < 4; o++) c[o] = *(const i32x8 __attribute__((address_space(4))) *)(flp + O[o] * 64 + 8 * B);
    };
    load_block(std::integral_constant<int, 0>{}, cur);
    static_for<0, 64>([&](auto nc) {
        constexpr int N = decltype(nc)::value, P = N >> 3, B = N & 7;
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xc07f);      // this block's coefficients are here; nothing else is outstanding
        __builtin_amdgcn_sched_barrier(0);       // (the next request does not move in front of the wait)
        if constexpr (N < 63) load_block(std::integral_constant<int, N + 1>{}, nxt);
        __builtin_amdgcn_sched_barrier(0);       // (the request stays in front of the products; nothing of the next block is started under this one)
#pragma unroll
        for (int j = 0; j < 8; j += 2) {                        // two columns at a time: an output's two products leave in one v_add3_u32
            int32_t m[2][4];
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const int k = B * 8 + j + e;
                const unsigned plan = ANALYSIS_PLAN[P][k];
                const int32_t cj[4] = {cur[0][j + e], cur[1][j + e], cur[2][j + e], cur[3][j + e]};
#pragma unroll
                for (int o = 0; o < 4; o++) {
                    const unsigned rep = (plan >> (2 * o)) & 3u;
                    if ((plan >> (8 + o)) & 1u) m[e][o] = 0;
                    else if (rep == (unsigned)o) m[e][o] = mulhi_vs(y[k], cj[o]);
                    else m[e][o] = m[e][rep];
                }
            }
#pragma unroll
            for (int o = 0; o < 4; o++) {
                a[o] += m[0][o] + m[1][o];
                asm("" : "+v"(a[o]));                           // (sums are not re-associated across outputs: shared partial sums lived long enough to spill)
            }
            if (j == 2) __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (B == 7) {
            // ---- the pass's four bands are complete.  Odd bands of odd slots change sign (:365-368): bands P and 16 + P are odd with P, 15 - P and 31 - P with P even
            if (odd_slot) {
                if (P & 1) { a[0] = (int32_t)(0u - (uint32_t)a[0]); a[2] = (int32_t)(0u - (uint32_t)a[2]); }
                else       { a[1] = (int32_t)(0u - (uint32_t)a[1]); a[3] = (int32_t)(0u - (uint32_t)a[3]); }
            }
            constexpr int q = P & 3;
            if constexpr (TILE) {
                trow[P] = (uint32_t)a[0]; trow[15 - P] = (uint32_t)a[1]; trow[16 + P] = (uint32_t)a[2]; trow[31 - P] = (uint32_t)a[3];
            } else {
            ot[lane * 17 + q] = (uint32_t)a[0];               // piece 0: bands 4g .. 4g + 3 (g = P >> 2)
            ot[lane * 17 + 4 + 3 - q] = (uint32_t)a[1];       // piece 1: bands 12 - 4g .. 15 - 4g
            ot[lane * 17 + 8 + q] = (uint32_t)a[2];           // piece 2: bands 16 + 4g ..
            ot[lane * 17 + 12 + 3 - q] = (uint32_t)a[3];      // piece 3: bands 28 - 4g ..
            }
            a[0] = a[1] = a[2] = a[3] = 0;
            if constexpr (q == 3 && !TILE) {
                // ---- sixteen bands of the 64 slots are complete: out 16 bytes per lane, four lanes (= four pieces) per row, sixteen rows per trip
                constexpr int g = P >> 2;
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
                // (the trip's sixteen bands -- 4g.., 12-4g.., 16+4g.., 28-4g.. -- lie SIDE BY SIDE in the row: sb_pos, below; 64 contiguous bytes per row
                //  and trip instead of four 16-byte pieces, each half of a 32-byte sector whose other half came a quarter of the wave's life later)
                const int piece = lane & 3;
#pragma unroll
                for (int r = lane >> 2; r < 64; r += 16) {
                    const uint32_t *src = ot + r * 17 + piece * 4;
                    const uint4 v = make_uint4(src[0], src[1], src[2], src[3]);
                    if (r < n_rows) *reinterpret_cast<uint4 *>(out + r * 32 + g * 16 + piece * 4) = v;
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();         // (the tile is free for the other sixteen bands)
            }
        }
        if constexpr (N < 63) {
#pragma unroll
            for (int o = 0; o < 4; o++) cur[o] = nxt[o];
        }
    });
}

// SB layout: int32 [ch][Ts][32] with Ts = n_frames * 36 slots (a 128-byte row per slot); band b of a slot sits at sb_pos(b) of its row -- the order in
// which the analysis completes its bands (four at a time: b >> 2 = 0, 3, 4, 7 in its first trip, 1, 2, 5, 6 in its second), so that a trip's sixteen
// bands are one half of the row.  (Round 6: in band order a trip wrote four 16-byte pieces per row, half a sector each; the counters showed 123 MB
// written for the array's 92.)
__device__ __forceinline__ int sb_pos(int band) { return (int)((0x37621540u >> (4 * (band >> 2))) & 7u) * 4 + (band & 3); }
__global__ __launch_bounds__(256, 5) void k_enc_analysis(
    const int16_t *__restrict__ pcm, const mp3s_frame_hdr *__restrict__ hdr, int n_frames,
    int32_t *__restrict__ SB, long Ts)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds_all[4][ENC_LDS_DW];
    static_assert(79 * ENC_ROW * 2 <= ENC_LDS_DW * 4, "PCM tile must fit");
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const long wid = (long)xcd_tile() * 4 + wave;
    const int ch = (int)(wid & 1);
    const long t0 = (wid >> 1) * 64;          // first slot of this wave
    if (t0 >= Ts) return;
    enc_analysis_wave<false>(pcm, hdr, Ts, ch, t0, lane, lds_all[wave], SB, nullptr, 0, 0, 0, [] {});
}

// mdct layout: int32 [frame][ch][gr][576]  (reference __mdct_freq)
// One wave per FRAME: lane = (channel, band) transforms both granules, which share 18 of their 36 input rows and every
// cos_l coefficient (a scalar operand loaded once for two multiply-adds).  Results go to an LDS tile [granule][channel]
// [line k][band] (rows padded to 33), where the alias butterflies between neighbouring bands are applied
// (MP3_Encoder.py:704-744, util.cmuls :143-155) and from where the four 576-line blocks leave as full rows.
constexpr int MD_ROW = 33, MD_BLK = 18 * MD_ROW;

// Two consecutive granules ga, ga + 1 of both channels from their 54 rows of subband samples in[] (18 of the granule in front, 18 + 18 of
// their own), lane = (channel, band): the MDCT sums, the alias butterflies through the wave's tile xs (4 * MD_BLK dwords, block = granule
// in the pair * 2 + channel), and the blocks of the granules g_lo <= g < g_hi out as full rows.  (MP3_Encoder.py:681-744)
__device__ __forceinline__ void enc_mdct_pair(const int32_t (&in)[54], int32_t *xs, int lane, int32_t *__restrict__ mdct, long ga, long g_lo, long g_hi)
{
    const int ch = lane >> 5, band = lane & 31;
    int32_t *x0 = xs + ch * MD_BLK + band, *x1 = x0 + 2 * MD_BLK;   // block index = granule * 2 + channel
    // The 36 coefficients of output k are one scalar batch (9 x s_load_dwordx4).  All scalar loads of a wave share one counter
    // that can only be waited to zero, and the compiler waits where a value is first used: the batch of output k + 1 is
    // therefore requested AFTER the first coefficient of batch k has been named (the wait: that batch was requested a whole
    // output ago) and before its 72 multiply-adds, which then run with the next batch in flight.
    struct Coef { i32x4 q[9]; };
    auto load_coef = [&](int k) {
        Coef c;
        const i32x4 *crow = reinterpret_cast<const i32x4 *>(c_tab.cos_l[k]);
#pragma unroll
        for (int q = 0; q < 9; q++) c.q[q] = crow[q];
        return c;
    };
    Coef cur = load_coef(0);
#pragma unroll 2
    for (int k = 0; k < 18; k++) {
        asm volatile("" ::"s"(cur.q[0].x));
        __builtin_amdgcn_sched_barrier(0);
        const Coef nxt = load_coef(k < 17 ? k + 1 : 17);
        __builtin_amdgcn_sched_barrier(0);
        int32_t a0 = 0, a1 = 0;
#pragma unroll
        for (int q = 0; q < 9; q++) {
            const i32x4 c = cur.q[q];
            a0 += mulhi_vs(in[4 * q], c.x);     a1 += mulhi_vs(in[18 + 4 * q], c.x);
            a0 += mulhi_vs(in[4 * q + 1], c.y); a1 += mulhi_vs(in[19 + 4 * q], c.y);
            a0 += mulhi_vs(in[4 * q + 2], c.z); a1 += mulhi_vs(in[20 + 4 * q], c.z);
            a0 += mulhi_vs(in[4 * q + 3], c.w); a1 += mulhi_vs(in[21 + 4 * q], c.w);
        }
        x0[k * MD_ROW] = a0;
        x1[k * MD_ROW] = a1;
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    if (band >= 1) {
#pragma unroll
        for (int g = 0; g < 2; g++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                int32_t *pa = x0 + g * 2 * MD_BLK + i * MD_ROW, *pb = x0 + g * 2 * MD_BLK + (17 - i) * MD_ROW - 1;
                const int64_t cs = c_tab.mdct_cs[i], ca = c_tab.mdct_ca[i];
                const int64_t a = *pa, b = *pb;          // X[band][i], X[band-1][17-i]
                *pa = (int32_t)((a * cs - b * ca) >> 31);
                *pb = (int32_t)((a * ca + b * cs) >> 31);
            }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 4; q++) {            // tile block q = (granule in the pair, channel) -> the reference's [frame][ch][gr][576]
        const long g = ga + (q >> 1);
        if (g < g_lo || g >= g_hi) continue;
        const int32_t *src = xs + q * MD_BLK;
        int32_t *o = mdct + (((g >> 1) * 2 + (q & 1)) * 2 + (g & 1)) * 576;
#pragma unroll
        for (int t = 0; t < 9; t++) {
            const int e = lane + 64 * t, b = (e * 3641) >> 16, k = e - 18 * b;   // e = band * 18 + k
            o[e] = src[k * MD_ROW + b];
        }
    }
}

__global__ __launch_bounds__(256, 4) void k_enc_mdct(
    const int32_t *__restrict__ SB, long Ts, const mp3s_frame_hdr *__restrict__ hdr, int n_frames,
    int32_t *__restrict__ mdct)
{
    __shared__ int32_t xs_all[4][4 * MD_BLK];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int f = xcd_tile() * 4 + wave;
    if (f >= n_frames) return;
    const int ch = lane >> 5, band = lane & 31;
    const bool has_prev = f > (int)hdr[f].stream_first;   // l3_sb_sample[ch][0] starts zeroed
    const int32_t *row = SB + ((long)ch * Ts + (long)f * 36) * 32 + sb_pos(band);
    // (the rows of the frame in front through an address that exists either way: compiled as an unconditional load + select -- as a
    // fully unrolled variant of this kernel was in round 4 -- `row - 18 * 32` of a batch's first frame lies in front of the buffer)
    const int32_t *prow = has_prev ? row - 18 * 32 : row;
    int32_t in[54];
#pragma unroll
    for (int j = 0; j < 18; j++) {
        const int32_t pv = prow[j * 32];
        in[j] = has_prev ? pv : 0;
        in[18 + j] = row[j * 32];
        in[36 + j] = row[(18 + j) * 32];
    }
    enc_mdct_pair(in, xs_all[wave], lane, mdct, 2l * f, 2l * f, 2l * f + 2);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Analysis and MDCT in ONE kernel: the subband samples between them (the reference's l3_sb_sample, MP3_Encoder.py:652-749: the filter
// bank is called from inside __mdct_sub) stay in LDS.
// A workgroup of eight waves owns EF_GR consecutive granules of both channels.
//   phase 1, lane = slot: its waves analyse the granules' 18 EF_GR slots and the 18 of the granule in front (what the first MDCT
//     overlaps with: computed again rather than carried) -- 4 waves x 64 lanes per channel, 252 of 256 lanes at work -- exactly as
//     k_enc_analysis does (enc_analysis_wave), into a tile of rows [channel][slot][32 bands + 1] in LDS.  The waves' PCM staging
//     lies INSIDE the tile: a barrier separates the last read of a sample from the first row written.
//   phase 2, lane = (channel, band): seven of the waves take two consecutive granules each (the seventh one), read their 54 rows
//     from the tile, and -- behind a barrier, after which the tile is dead -- run enc_mdct_pair with their butterfly tile where
//     the rows were (7 x 4 blocks are the tile's size to the byte).
// No SB array, one launch; the analysis of one granule in thirteen is done twice.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int EF_GR = 13, EF_SLOTS = 18 * (EF_GR + 1), EF_WAVES = 8;
constexpr int EF_TILE_DW = 2 * (EF_SLOTS + 1) * ENC_TROW;          // per channel one spare row in front (what lanes outside the tile write)
static_assert(EF_SLOTS <= 4 * 64, "four analysis waves per channel");
static_assert(EF_WAVES * ENC_LDS_DW <= EF_TILE_DW, "the waves' PCM staging fits the tile");
static_assert(((EF_GR + 1) / 2) * 4 * MD_BLK <= EF_TILE_DW, "the MDCT waves' butterfly tiles fit the tile");

__global__ __launch_bounds__(EF_WAVES * 64, 4) void k_enc_fused(
    const int16_t *__restrict__ pcm, const mp3s_frame_hdr *__restrict__ hdr, int n_frames, int32_t *__restrict__ mdct)
{
    __shared__ __attribute__((aligned(16))) uint32_t tile[EF_TILE_DW];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const long Ts = (long)n_frames * 36, n_gran = (long)n_frames * 2;
    const long g0 = (long)xcd_tile() * EF_GR;                      // the workgroup's granules g0 .. g0 + EF_GR - 1
    const long s0 = g0 * 18, t_first = s0 - 18;                    // ... their slots, and the tile's first row
    // ---- phase 1
    {
        const int ch = wave >> 2;
        const long t0 = t_first + 64 * (wave & 3);
        uint32_t *rows = tile + ((long)ch * (EF_SLOTS + 1) + 1) * ENC_TROW;
        enc_analysis_wave<true>(pcm, hdr, Ts, ch, t0, lane, tile + wave * ENC_LDS_DW, nullptr, rows, t_first, t_first, t_first + EF_SLOTS,
                                [] { __syncthreads(); });
    }
    __syncthreads();                                               // every row of the tile is written
    // ---- phase 2: the granules in pairs (even, odd) = the two of a frame, so that the second one never starts a stream; a pair that reaches
    //      out of the workgroup's range on either side computes that granule too and drops it
    const long ga = g0 - (g0 & 1) + 2 * wave;                      // this wave's granules ga, ga + 1
    const long g_hi = g0 + EF_GR < n_gran ? g0 + EF_GR : n_gran;
    const bool unit = ga < g_hi;                                   // (seven of the eight waves; fewer at the end of the batch)
    int32_t in[54];
    if (unit) {
        const int ch = lane >> 5, band = lane & 31;
        // l3_sb_sample[ch][0] starts zeroed (MP3_Encoder.py:528-534): a stream's first granule overlaps with zeros
        const bool has_prev = ga >= g0 && ga > (long)hdr[ga >> 1].stream_first * 2;      // (ga == g0 - 1: its result is dropped)
        // row 0 of the tile is slot 18 (g0 - 1); rows of the granule in front of ga from r_prev on (not there for ga == g0 - 1)
        const long r_prev = (ga - g0) * 18;
        const uint32_t *row = tile + ((long)ch * (EF_SLOTS + 1) + 1 + (r_prev >= 0 ? r_prev : 0)) * ENC_TROW + band;
        const long r_own = r_prev >= 0 ? 18 : 0;                   // the granule's own rows from here
#pragma unroll
        for (int j = 0; j < 18; j++) {
            const int32_t pv = (int32_t)row[j * ENC_TROW];
            in[j] = has_prev ? pv : 0;
            in[18 + j] = (int32_t)row[(r_own + j) * ENC_TROW];
            in[36 + j] = ga + 1 < g_hi ? (int32_t)row[(r_own + 18 + j) * ENC_TROW] : 0;
        }
    }
    __syncthreads();                                               // every wave has its rows: the tile is free
    if (unit) enc_mdct_pair(in, reinterpret_cast<int32_t *>(tile) + wave * 4 * MD_BLK, lane, mdct, ga, g0, g_hi);
}

}  // namespace mp3s
