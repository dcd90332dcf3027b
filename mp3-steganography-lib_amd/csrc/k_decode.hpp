// Decode transform kernels (gfx950).  Included by mp3s_device.hip only.
//
//   k_dec_imdct : requantise -> MS stereo -> reorder | alias reduction -> IMDCT + window + overlap-add
//                 (reference decoder/Frame.py:157-218, 561-622, 106-154; frequency inversion :624-631
//                 is folded into the stores).  One wavefront walks DEC_RUN consecutive granules, lane =
//                 (channel, subband): the 18 lines of a subband and the 18-sample overlap tail live in
//                 one lane's registers, the IMDCT twiddle of a given (output, term) is the same for every
//                 lane, so it is a scalar (SGPR) operand fetched through the scalar cache -- no LDS
//                 traffic in the inner loop.  The run is primed with the tail of the granule before it.
//   k_dec_synth : polyphase matrixing, windowing and PCM conversion
//                 (reference Frame.py:65-103, 633-640, MP3_Parser.py:91).  lane = time slot:
//                 the 32 subband samples of a slot live in registers, the 64x32 matrix and D[] are
//                 scalar operands; the 16-slot V history is exchanged between lanes through LDS.
//
// All float math is fp64 with the reference's operation order (separate multiply and add, sums
// ascending from +0.0): the results are bit-identical to the reference's float64 samples.
#pragma once

namespace mp3s {

__device__ __forceinline__ double shfl_xor_f64(double v, int mask)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, mask, 64);
    hi = __shfl_xor(hi, mask, 64);
    return __hiloint2double(hi, lo);
}

// ---------------------------------------------------------------------------------------------
// Kernel A.  S layout: float64 [ch][slot][32 subbands] (slot = granule*18 + i): the time-domain subband
// samples after overlap-add and frequency inversion -- what synth_filter_bank reads (Frame.py:78-79).
// A wave stores one 256-byte row per channel per slot: fully coalesced.
// ---------------------------------------------------------------------------------------------
constexpr int DEC_A_WAVES = 4;
constexpr int DEC_RUN = 4;     // granules per wave (+1 priming granule whose second half only is computed)

// requantise .. alias/reorder of granule g for this lane's subband; v[18] = IMDCT input
__device__ __forceinline__ void dec_prepare(double (&v)[18], double *buf, const int16_t *__restrict__ is,
                                            const mp3s_granule_si *__restrict__ si, int g, int sr, bool ms, int nch,
                                            int ch, int sb, bool live, int &bt_out)
{
    const mp3s_granule_si *gs = &si[(long)g * 2 + (live ? ch : 0)];
    const int gg = gs->global_gain, bt = gs->block_type & 3, mixed = gs->mixed_block_flag ? 1 : 0;
    const int mult2 = gs->scalefac_scale ? 2 : 1, preflag = gs->preflag ? 1 : 0;
    const int cse = bt == 2 ? 1 : (mixed ? 2 : 0);
    const uint8_t *map = c_tab.rq_map[sr][cse];
    bt_out = bt;
    // ---- requantise (Frame.py:210-215): ((sign * |is|^(4/3)) * 2^(exp1/4)) * 2^(-exp2)
    const int16_t *isp = is + ((long)g * 2 + (live ? ch : 0)) * 576 + sb * 18;
#pragma unroll
    for (int k = 0; k < 18; k++) {
        const int x = live ? (int)isp[k] : 0;
        const int m = map[sb * 18 + k];
        const int sfb = m & 31, win = (m >> 5) & 3;
        int e1, k2;
        if (m & 0x80) {
            e1 = gg - 210 - 8 * (gs->sub_block_gain[win] & 7);
            k2 = mult2 * (gs->scale_fac_s[win][sfb < 13 ? sfb : 12] & 15);
        } else {
            e1 = gg - 210;
            k2 = mult2 * ((gs->scale_fac_l[sfb < 22 ? sfb : 21] & 15) + preflag * c_tab.pre_tab[sfb]);
        }
        int ax = x < 0 ? -x : x;
        ax = ax < POW43_N ? ax : POW43_N - 1;
        const double a = c_tab.pow43[ax];
        const double sa = x < 0 ? -a : a;   // sign * a is exact
        v[k] = (sa * c_tab.pow2q[e1 - POW2Q_MIN]) * c_tab.pow2h[k2 < POW2H_N ? k2 : POW2H_N - 1];
    }
    // ---- MS stereo (Frame.py:568-572): L = (M + S) / sqrt2, R = (M - S) / sqrt2
    if (ms && nch == 2) {
#pragma unroll
        for (int k = 0; k < 18; k++) {
            const double o = shfl_xor_f64(v[k], 32);
            v[k] = ch == 0 ? (v[k] + o) / c_tab.sqrt2 : (o - v[k]) / c_tab.sqrt2;
        }
    }
    // ---- reorder (short / mixed) or alias reduction (long) through the wave's LDS slice
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 18; k++) buf[sb * 18 + k] = v[k];
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's own LDS writes have landed
    __builtin_amdgcn_wave_barrier();
    if (cse != 0) {
        const int16_t *src = c_tab.reorder_src[sr];
#pragma unroll
        for (int k = 0; k < 18; k++) {
            const int s = src[sb * 18 + k];
            v[k] = s >= 0 ? buf[s] : 0.0;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            // lower line of the butterfly with subband sb-1:  s2*cs + s1*ca   (Frame.py:622)
            if (sb >= 1) {
                const double s1 = buf[18 * sb - 1 - i], s2 = v[i];
                v[i] = s2 * c_tab.alias_cs[i] + s1 * c_tab.alias_ca[i];
            }
            // upper line of the butterfly with subband sb+1:  s1*cs - s2*ca   (Frame.py:621)
            if (sb <= 30) {
                const double s1 = v[17 - i], s2 = buf[18 * (sb + 1) + i];
                v[17 - i] = s1 * c_tab.alias_cs[i] - s2 * c_tab.alias_ca[i];
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
}

__global__ __launch_bounds__(DEC_A_WAVES * 64) void k_dec_imdct(
    const int16_t *__restrict__ is, const mp3s_granule_si *__restrict__ si, const mp3s_frame_hdr *__restrict__ hdr,
    int n_granules, int nch, double *__restrict__ S, long T)
{
    __shared__ double lds[DEC_A_WAVES][2][576];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int g0 = (blockIdx.x * DEC_A_WAVES + wave) * DEC_RUN;
    if (g0 >= n_granules) return;  // whole wave exits together
    const int ch = lane >> 5, sb = lane & 31;
    const bool live = ch < nch;
    double *buf = lds[wave][ch];
    const bool neg_odd = (sb & 1) != 0;
    double tail[18];
#pragma unroll
    for (int i = 0; i < 18; i++) tail[i] = 0.0;

    // gi = -1 primes the overlap with the second half of granule g0-1 (when it belongs to the same stream)
    for (int gi = -1; gi < DEC_RUN; gi++) {
        const int g = g0 + gi;
        if (g >= n_granules) break;
        if (g < 0) continue;
        const mp3s_frame_hdr fh = hdr[g >> 1];
        const int first_gran = (int)fh.stream_first * 2;
        if (gi < 0 && g0 <= first_gran) continue;          // the run starts a stream: nothing before it
        if (gi >= 0 && g == first_gran) {
#pragma unroll
            for (int i = 0; i < 18; i++) tail[i] = 0.0;    // Frame.py:234 prev_samples starts as zeros
        }
        const int sr = fh.sr_idx < 3 ? fh.sr_idx : 0;
        double v[18];
        int bt;
        dec_prepare(v, buf, is, si, g, sr, fh.ms_stereo != 0, nch, ch, sb, live, bt);

        // ---- IMDCT + window (Frame.py:124-148), overlap (:151-153), frequency inversion (:629-631) in the sign
        double *row = S + ((long)(live ? ch : 0) * T + (long)g * 18) * 32 + sb;
        if (bt != 2) {
            const double *win = c_tab.sine_block[bt];
            if (gi >= 0) {
#pragma unroll 2
                for (int i = 0; i < 18; i++) {
                    double x = 0.0;
#pragma unroll
                    for (int k = 0; k < 18; k++) x += v[k] * c_tab.imdct_cos36[i][k];
                    x = x * win[i] + tail[i];
                    if (neg_odd && (i & 1)) x = -x;
                    if (live) row[(long)i * 32] = x;
                }
            }
#pragma unroll 2
            for (int i = 18; i < 36; i++) {
                double x = 0.0;
#pragma unroll
                for (int k = 0; k < 18; k++) x += v[k] * c_tab.imdct_cos36[i][k];
                tail[i - 18] = x * win[i];
            }
        } else {
            double t[36];
#pragma unroll
            for (int w = 0; w < 3; w++)
#pragma unroll
                for (int i = 0; i < 12; i++) {
                    double x = 0.0;
#pragma unroll
                    for (int k = 0; k < 6; k++) x += v[6 * w + k] * c_tab.imdct_cos12[i][k];
                    t[w * 12 + i] = x * c_tab.sine_block[2][i];
                }
            if (gi >= 0) {
#pragma unroll
                for (int i = 0; i < 18; i++) {
                    // sample_block[0..5] = 0, [6..11] = t[0..5], [12..17] = t[6..11] + t[12..17]   (:136-142)
                    const double blk = i < 6 ? 0.0 : (i < 12 ? t[i - 6] : t[i - 6] + t[i]);
                    double x = blk + tail[i];
                    if (neg_odd && (i & 1)) x = -x;
                    if (live) row[(long)i * 32] = x;
                }
            }
#pragma unroll
            for (int i = 0; i < 6; i++) {
                tail[i] = t[18 + i] + t[24 + i];           // sample_block[18..23]
                tail[6 + i] = t[30 + i];                   // sample_block[24..29]
                tail[12 + i] = 0.0;                        // sample_block[30..35]
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Kernel BC.  One workgroup = a tile of TW*64 consecutive slots for every channel; the first 15
// lanes of a tile only rebuild V history for the others (halo), so a tile emits TW*64-15 slots.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int16_t pcm_to_i16(double v)
{
    const double x = v * 32767;
    if (!(fabs(x) < 2147483648.0)) return 0;          // x86 cvttsd2si "indefinite" -> low half 0
    return (int16_t)(uint16_t)((uint32_t)(int32_t)x & 0xffffu);
}

template <int TW>
__global__ __launch_bounds__(TW * 64 * 2, (TW * 2 * 2) / 4) void k_dec_synth(
    const double *__restrict__ S, long T, const mp3s_frame_hdr *__restrict__ hdr, int nch, int n_halo, int out_format,
    void *__restrict__ pcm_out)
{
    constexpr int TL_LANES = TW * 64, OUT = TL_LANES - 15;
    constexpr int OROW = 33;                                   // dwords per staged slot (32 + 1 pad: no bank conflicts)
    __shared__ double ex[2][2][2][TL_LANES];                   // [parity][ch][V half][lane]
    __shared__ uint32_t otile[OUT * OROW];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int ch = wave / TW, tl = (wave % TW) * 64 + lane;
    const long tile0 = (long)blockIdx.x * OUT;
    const long t = tile0 - 15 + tl;
    const bool valid = t >= 0 && t < T;
    int lim = -1;            // number of earlier in-stream slots (V history available), -1: slot not valid
    if (valid) {
        const long s0 = (long)hdr[t / 36].stream_first * 36;
        lim = (int)((t - s0) < 64 ? (t - s0) : 64);
    }
    double Sv[32];
    {
        const double2 *sp = reinterpret_cast<const double2 *>(S + ((long)ch * T + (valid ? t : 0)) * 32);
#pragma unroll
        for (int j = 0; j < 16; j++) {
            double2 q = make_double2(0.0, 0.0);
            if (valid) q = sp[j];
            Sv[2 * j] = q.x; Sv[2 * j + 1] = q.y;
        }
    }
    const long halo_slots = (long)n_halo * 36;
    const bool emit = valid && tl >= 15 && t >= halo_slots;
    uint16_t *ot16 = reinterpret_cast<uint16_t *>(otile);
    int p = 0;
    for (int i = 0; i < 32; i++) {
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int j = 0; j < 32; j++) {           // Frame.py:84-87
            a0 += Sv[j] * c_tab.synth_matrix[i][j];
            a1 += Sv[j] * c_tab.synth_matrix[32 + i][j];
        }
        ex[p][ch][0][tl] = a0;
        ex[p][ch][1][tl] = a1;
        __syncthreads();
        if (tl >= 15) {
            double sum = 0.0;
#pragma unroll
            for (int jj = 0; jj < 16; jj++) {   // Frame.py:89-101 (u, w, sum over 16 windowed taps)
                double u = ex[p][ch][jj & 1][tl - jj];
                if (jj > lim) u = 0.0;           // before the stream started the fifo holds zeros
                sum += u * c_tab.synth_window[32 * jj + i];
            }
            if (emit) {
                const long to = t - halo_slots;
                if (out_format == MP3S_PCM_I16) ot16[(tl - 15) * OROW * 2 + i * nch + ch] = (uint16_t)pcm_to_i16(sum);
                else if (out_format == MP3S_PCM_F64) ((double *)pcm_out)[(to * 32 + i) * nch + ch] = sum;
                else ((float *)pcm_out)[(to * 32 + i) * nch + ch] = (float)sum;
            }
        }
        p ^= 1;
    }
    if (out_format == MP3S_PCM_I16) {
        __syncthreads();
        const int dw_per_slot = 16 * nch;                      // 32 samples * nch * 2 bytes / 4
        const int n_dw = OUT * dw_per_slot;
        uint32_t *outp = (uint32_t *)pcm_out;
        for (int c = threadIdx.x; c < n_dw; c += blockDim.x) {
            const int sl = c / dw_per_slot, w = c - sl * dw_per_slot;
            const long slot = tile0 + sl;
            if (slot < halo_slots || slot >= T) continue;
            outp[(slot - halo_slots) * dw_per_slot + w] = otile[sl * OROW + w];
        }
    }
}

}  // namespace mp3s
