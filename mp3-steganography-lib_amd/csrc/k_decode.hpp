// Decode transform kernels (gfx950).  Included by mp3s_device.hip only.
//
//   k_dec_imdct : requantise -> MS stereo -> reorder | alias reduction -> IMDCT + window
//                 (reference decoder/Frame.py:157-218, 561-622, 106-154; frequency inversion :624-631
//                 is folded into the stores).  One wavefront per granule, lane = (channel, subband):
//                 the 18 lines of a subband live in one lane's registers, the IMDCT twiddle of a
//                 given (output, term) is the same for every lane, so it is a scalar (SGPR) operand
//                 fetched through the scalar cache -- no LDS traffic in the inner loop.
//   k_dec_synth : overlap-add, polyphase matrixing, windowing and PCM conversion
//                 (reference Frame.py:150-153, 65-103, 633-640, MP3_Parser.py:91).  lane = time slot:
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
// Kernel A.  H / TL layout: [ch][sb][slot] doubles, row stride Tp = n_frames*36 + 18.
//   H [slot]  (slot = g*18 + i, i < 18)      first half of granule g's windowed IMDCT
//   TL[slot]  (slot = g*18 + i, i in 18..35) second half, i.e. it lands on granule g+1's slots
// so that the time-domain subband sample of a slot is simply H[slot] + TL[slot].
// ---------------------------------------------------------------------------------------------
constexpr int DEC_A_WAVES = 4;

__global__ __launch_bounds__(DEC_A_WAVES * 64) void k_dec_imdct(
    const int16_t *__restrict__ is, const mp3s_granule_si *__restrict__ si, const mp3s_frame_hdr *__restrict__ hdr,
    int n_granules, int nch, double *__restrict__ H, double *__restrict__ TL, long Tp)
{
    __shared__ double lds[DEC_A_WAVES][2][576];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = blockIdx.x * DEC_A_WAVES + wave;
    if (g >= n_granules) return;  // whole wave exits together
    const int ch = lane >> 5, sb = lane & 31;
    const bool live = ch < nch;
    const mp3s_frame_hdr fh = hdr[g >> 1];
    const int sr = fh.sr_idx < 3 ? fh.sr_idx : 0;
    const mp3s_granule_si *gs = &si[(long)g * 2 + (live ? ch : 0)];
    const int gg = gs->global_gain, bt = gs->block_type & 3, mixed = gs->mixed_block_flag ? 1 : 0;
    const int mult2 = gs->scalefac_scale ? 2 : 1, preflag = gs->preflag ? 1 : 0;
    const int cse = bt == 2 ? 1 : (mixed ? 2 : 0);
    const uint8_t *map = c_tab.rq_map[sr][cse];

    // ---- requantise (Frame.py:210-215): ((sign * |is|^(4/3)) * 2^(exp1/4)) * 2^(-exp2)
    double v[18];
    const int16_t *isp = is + ((long)g * 2 + (live ? ch : 0)) * 576 + sb * 18;
#pragma unroll
    for (int k = 0; k < 18; k++) {
        const int line = sb * 18 + k;
        const int x = live ? (int)isp[k] : 0;
        const int m = map[line];
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
    if (fh.ms_stereo && nch == 2) {
#pragma unroll
        for (int k = 0; k < 18; k++) {
            const double o = shfl_xor_f64(v[k], 32);
            v[k] = ch == 0 ? (v[k] + o) / c_tab.sqrt2 : (o - v[k]) / c_tab.sqrt2;
        }
    }

    // ---- reorder (short / mixed) or alias reduction (long) through the wave's LDS slice
    double *buf = lds[wave][ch];
#pragma unroll
    for (int k = 0; k < 18; k++) buf[sb * 18 + k] = v[k];
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): own wave's LDS writes landed (single-wave slice)
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

    // ---- IMDCT + window (Frame.py:124-148); frequency inversion (:629-631) folded into the sign
    const long row = ((long)ch * 32 + sb) * Tp + (long)g * 18;
    double *Hp = H + row, *Tp_ = TL + row;
    const bool neg_odd = (sb & 1) != 0;
    if (bt != 2) {
        const double *win = c_tab.sine_block[bt];
#pragma unroll 2
        for (int i = 0; i < 36; i += 2) {
            double x0 = 0.0, x1 = 0.0;
#pragma unroll
            for (int k = 0; k < 18; k++) {
                x0 += v[k] * c_tab.imdct_cos36[i][k];
                x1 += v[k] * c_tab.imdct_cos36[i + 1][k];
            }
            x0 = x0 * win[i];
            x1 = x1 * win[i + 1];
            if (neg_odd) x1 = -x1;   // odd subband, odd slot
            if (live) {
                double2 o = make_double2(x0, x1);
                if (i < 18) *reinterpret_cast<double2 *>(Hp + i) = o;
                else *reinterpret_cast<double2 *>(Tp_ + i) = o;
            }
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
        double o[36];
#pragma unroll
        for (int i = 0; i < 6; i++) {
            o[i] = 0.0;
            o[6 + i] = t[i];
            o[12 + i] = t[6 + i] + t[12 + i];
            o[18 + i] = t[18 + i] + t[24 + i];
            o[24 + i] = t[30 + i];
            o[30 + i] = 0.0;
        }
        if (live) {
#pragma unroll
            for (int i = 0; i < 36; i += 2) {
                double2 w2 = make_double2(o[i], neg_odd ? -o[i + 1] : o[i + 1]);
                if (i < 18) *reinterpret_cast<double2 *>(Hp + i) = w2;
                else *reinterpret_cast<double2 *>(Tp_ + i) = w2;
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
__global__ __launch_bounds__(TW * 64 * 2) void k_dec_synth(
    const double *__restrict__ H, const double *__restrict__ TL, long Tp, const mp3s_frame_hdr *__restrict__ hdr,
    int n_frames, int nch, int n_halo, int out_format, void *__restrict__ pcm_out)
{
    constexpr int TL_LANES = TW * 64, OUT = TL_LANES - 15;
    __shared__ double ex[2][2][2][TL_LANES];                   // [parity][ch][V half][lane]
    __shared__ __attribute__((aligned(16))) int16_t otile[OUT * 32 * 2];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ch = wave / TW, tl = (wave % TW) * 64 + lane;
    const long T = (long)n_frames * 36;
    const long tile0 = (long)blockIdx.x * OUT;
    const long t = tile0 - 15 + tl;
    const bool valid = t >= 0 && t < T;
    int lim = -1;            // number of earlier in-stream slots (V history available), -1: slot not valid
    bool has_tail = false;
    if (valid) {
        const long s0 = (long)hdr[t / 36].stream_first * 36;
        lim = (int)((t - s0) < 64 ? (t - s0) : 64);
        has_tail = (t / 18) > (s0 / 18);
    }
    double S[32];
    {
        const double *hp = H + (long)ch * 32 * Tp + t, *tp = TL + (long)ch * 32 * Tp + t;
#pragma unroll
        for (int j = 0; j < 32; j++) {
            double h = 0.0, tv = 0.0;
            if (valid) { h = hp[(long)j * Tp]; if (has_tail) tv = tp[(long)j * Tp]; }
            S[j] = h + tv;   // Frame.py:152  sample_block[i] + prev_samples[ch][block][i]
        }
    }
    const long halo_slots = (long)n_halo * 36;
    const bool emit = valid && tl >= 15 && t >= halo_slots;
    int p = 0;
    for (int i = 0; i < 32; i++) {
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int j = 0; j < 32; j++) {           // Frame.py:84-87
            a0 += S[j] * c_tab.synth_matrix[i][j];
            a1 += S[j] * c_tab.synth_matrix[32 + i][j];
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
                if (out_format == MP3S_PCM_I16) otile[((tl - 15) * 32 + i) * nch + ch] = pcm_to_i16(sum);
                else if (out_format == MP3S_PCM_F64) ((double *)pcm_out)[(to * 32 + i) * nch + ch] = sum;
                else ((float *)pcm_out)[(to * 32 + i) * nch + ch] = (float)sum;
            }
        }
        p ^= 1;
    }
    if (out_format == MP3S_PCM_I16) {
        __syncthreads();
        const int chunks_per_slot = (32 * nch * 2) / 16;      // 16-byte chunks per slot
        const int n_chunks = OUT * chunks_per_slot;
        int16_t *outp = (int16_t *)pcm_out;
        for (int c = threadIdx.x; c < n_chunks; c += blockDim.x) {
            const long slot = tile0 + c / chunks_per_slot;
            if (slot < halo_slots || slot >= T) continue;
            const uint4 val = reinterpret_cast<const uint4 *>(otile)[c];
            reinterpret_cast<uint4 *>(outp + (slot - halo_slots) * 32 * nch)[c % chunks_per_slot] = val;
        }
    }
}

}  // namespace mp3s
