// Serial chains of the rate loop, checked on the device (gfx950).  Included by mp3s_device.hip only.
//
// The reference encoder carries two things from one granule*channel ("unit") to the next:
//   * the message cursor __hide_str_offset: every unit adds the number of its non-zero table_select
//     (reference encoder/MP3_Encoder.py:808-809), and __new_choose_table reads the message at that cursor (:1154-1168);
//   * address1/2/3 and quantizerStepSize of the same (gr, ch) of the frame before, which a unit without big values
//     (or without any signal) does not overwrite (:1004-1006, :788-803).
// k_rate_loop runs every unit on assumed inputs (cursor_in, state_in).  These two kernels do what the host walk of
// round 1 did after downloading 72 bytes per unit: a segmented scan over the frames of the batch (segments = streams)
// gives every frame the true cursor and the last values its four (gr, ch) classes were left with, every unit's
// assumptions are compared with them, silent units get the values they inherit (the bit packer writes them into the
// side info), and the host reads back ONE verdict: the number of units that ran on wrong assumptions that mattered.
// Zero is the common case (a short message in a long stream, guess = 3 tables per unit); otherwise the host resolves
// the chain as before (csrc/mp3s_encode_pipeline.cpp).
//
// Scan element per frame: (reset, tables, last[4]) -- reset: the frame starts a stream; tables: non-zero tables of its
// active units; last[k]: index of its active unit of class k = ch*2+gr, or -1.  The operator takes the right operand's
// reset, adds tables unless the right operand resets, and keeps the right-most last[k].
#pragma once

namespace mp3s {

constexpr int CH_THREADS = 256;   // frames per workgroup

struct ChainEl { int reset, tables, last[4]; };

__device__ __forceinline__ ChainEl chain_identity() { return ChainEl{0, 0, {-1, -1, -1, -1}}; }
__device__ __forceinline__ ChainEl chain_combine(const ChainEl &l, const ChainEl &r)
{
    if (r.reset) return r;
    ChainEl o;
    o.reset = l.reset;
    o.tables = l.tables + r.tables;
#pragma unroll
    for (int k = 0; k < 4; k++) o.last[k] = r.last[k] >= 0 ? r.last[k] : l.last[k];
    return o;
}
__device__ __forceinline__ ChainEl chain_shfl_up(const ChainEl &x, int d)
{
    ChainEl o;
    o.reset = __shfl_up(x.reset, d, 64);
    o.tables = __shfl_up(x.tables, d, 64);
#pragma unroll
    for (int k = 0; k < 4; k++) o.last[k] = __shfl_up(x.last[k], d, 64);
    return o;
}

// inclusive scan over the 256 threads of the workgroup; *total = the combination of all 256 elements
__device__ __forceinline__ ChainEl chain_block_scan(ChainEl x, ChainEl (&wave_tot)[CH_THREADS / 64], ChainEl *total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const ChainEl o = chain_shfl_up(x, d);
        if (lane >= d) x = chain_combine(o, x);
    }
    __syncthreads();                       // the previous use of wave_tot is over
    if (lane == 63) wave_tot[wave] = x;
    __syncthreads();
    ChainEl pre = chain_identity();
#pragma unroll
    for (int w = 0; w < CH_THREADS / 64; w++) {
        if (w < wave) pre = chain_combine(pre, wave_tot[w]);
    }
    x = chain_combine(pre, x);
    if (total) {
        ChainEl t = wave_tot[0];
#pragma unroll
        for (int w = 1; w < CH_THREADS / 64; w++) t = chain_combine(t, wave_tot[w]);
        *total = t;
    }
    return x;
}

__device__ __forceinline__ ChainEl chain_element(const mp3s_gr_out *__restrict__ gr, const mp3s_rate_frame *__restrict__ rf,
                                                 const mp3s_chain_seg *__restrict__ segs, int f, int n_frames)
{
    ChainEl e = chain_identity();
    if (f >= n_frames) return e;
    e.reset = segs[rf[f].stream].first_frame == f ? 1 : 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const mp3s_gr_out &g = gr[(long)f * 4 + k];
        if (g.flags & MP3S_RF_ACTIVE) { e.tables += g.n_tables; e.last[k] = f * 4 + k; }
    }
    return e;
}

// pass 1: one aggregate per workgroup of 256 frames
__global__ __launch_bounds__(CH_THREADS) void k_chain_sum(const mp3s_gr_out *__restrict__ gr, const mp3s_rate_frame *__restrict__ rf,
                                                          const mp3s_chain_seg *__restrict__ segs, int n_frames,
                                                          ChainEl *__restrict__ agg, int32_t *__restrict__ verdict,
                                                          mp3s_chain_seg_out *__restrict__ seg_out, int32_t *__restrict__ redo,
                                                          const int32_t *__restrict__ only_after)
{
    __shared__ ChainEl wave_tot[CH_THREADS / 64];
    if (only_after && only_after[0] == 0) {          // the check behind the re-runs: nothing was run again, the verdict in front stands
        if (redo && blockIdx.x == 0 && threadIdx.x == 0) redo[0] = 0;   // (... and the round after this one finds an empty list, not the one before last's)
        return;
    }
    const int f = blockIdx.x * CH_THREADS + threadIdx.x;
    // what pass 2 accumulates into starts from zero (no fill launches in front of the pair)
    if (f == 0) { verdict[0] = 0; verdict[1] = 0; }
    if (redo) {
        if (f == 0) redo[0] = 0;
        // (the first check of a launch sequence also empties the second list: a later round that is skipped -- nothing was run
        // again in front of it -- must leave an empty list behind for the round after it)
        if (f == 0 && !only_after) redo[REDO_WORDS] = 0;
    }
    if (f < n_frames && segs[rf[f].stream].first_frame == f) seg_out[rf[f].stream].carry_used = 0;
    ChainEl total;
    chain_block_scan(chain_element(gr, rf, segs, f, n_frames), wave_tot, &total);
    if (threadIdx.x == 0) agg[blockIdx.x] = total;
}

// pass 2: every frame walks its four units with the true inputs in hand
//   verdict[0] += units whose assumed cursor / inherited addresses were wrong and mattered; verdict[1] |= 1: a
//   quantizer step left the table;  seg_out[s]: cursor behind the stream's last unit (index into the batch's message
//   array), the four chains as the stream leaves them, carry_used (see mp3s_encode_block)
__global__ __launch_bounds__(CH_THREADS) void k_chain_apply(mp3s_gr_out *__restrict__ gr, const mp3s_rate_frame *__restrict__ rf,
                                                            const mp3s_chain_seg *__restrict__ segs, int n_frames,
                                                            const ChainEl *__restrict__ agg, int32_t *__restrict__ cursor_in,
                                                            const int32_t *__restrict__ state_in, int32_t *__restrict__ verdict,
                                                            mp3s_chain_seg_out *__restrict__ seg_out, int32_t *__restrict__ redo_list,
                                                            const int32_t *__restrict__ only_after)
{
    if (only_after && only_after[0] == 0) return;
    __shared__ ChainEl wave_tot[CH_THREADS / 64];
    const int f = blockIdx.x * CH_THREADS + threadIdx.x;
    // what the workgroups in front of this one add up to
    ChainEl before = chain_identity();
    for (int c0 = 0; c0 < (int)blockIdx.x; c0 += CH_THREADS) {
        const int j = c0 + (int)threadIdx.x;
        ChainEl total;
        chain_block_scan(j < (int)blockIdx.x ? agg[j] : chain_identity(), wave_tot, &total);
        before = chain_combine(before, total);
    }
    const ChainEl own_el = chain_element(gr, rf, segs, f, n_frames);
    const ChainEl incl = chain_block_scan(own_el, wave_tot, nullptr);
    // exclusive prefix: the element of the thread before (wave_tot of the wave before for lane 0)
    ChainEl excl = chain_shfl_up(incl, 1);
    if ((threadIdx.x & 63) == 0) {
        excl = chain_identity();
        for (int w = 0; w < (int)(threadIdx.x >> 6); w++) excl = chain_combine(excl, wave_tot[w]);
    }
    excl = chain_combine(before, excl);
    int redo_here = 0, err_here = 0;
    if (f < n_frames) {
        const int s = rf[f].stream;
        const mp3s_chain_seg sg = segs[s];
        const bool first = sg.first_frame == f;
        if (first) excl = chain_identity();          // whatever lies in front belongs to another stream
        long cur = (long)sg.hide_begin + excl.tables;
        const long end = sg.hide_end;
        const bool hiding = sg.hide_end > sg.hide_base;
        int carry_used = first && cur < end ? 1 : 0; // the message is still being hidden when the stream (block) starts
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const long u = (long)f * 4 + k;
            mp3s_gr_out &g = gr[u];
            int32_t ch[4];
            const bool own = excl.last[k] >= 0;      // the stream has set chain k itself by now
            if (own) {
                const mp3s_gr_out &p = gr[excl.last[k]];
                ch[0] = p.address[0]; ch[1] = p.address[1]; ch[2] = p.address[2]; ch[3] = p.quantizer_step;
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) ch[j] = sg.chain_in[k][j];
            }
            const int flags = g.flags;
            const bool active = flags & MP3S_RF_ACTIVE;
            if (!own && (!active || (flags & MP3S_RF_USED_ADDR_IN))) carry_used = 1;
            bool redo = false;
            if (hiding && active) {
                const long used = cursor_in[u];
                if (used != cur && (used < cur ? used : cur) < end) redo = true;
            }
            if (flags & MP3S_RF_USED_ADDR_IN) {
                // (what the unit was given: the caller's array, else the unit's own record of it)
                const int32_t given = state_in ? state_in[u * 4] | (state_in[u * 4 + 1] << 10) | (state_in[u * 4 + 2] << 20) : g.reserved0;
                if (given != (ch[0] | (ch[1] << 10) | (ch[2] << 20))) redo = true;
            }
            redo_here += redo ? 1 : 0;
            if (redo && redo_list) {
                // for k_rate_redo: the unit once more, on the cursor and the addresses found here (which hold if the units
                // in front keep their results; the check after the re-runs says whether they did)
                const int at = atomicAdd(&redo_list[0], 1);
                if (at < REDO_CAP) {
                    const int32_t c = (int32_t)(cur < (long)MP3S_NO_CURSOR ? cur : (long)MP3S_NO_CURSOR);
                    redo_list[REDO_HEAD + at] = (int32_t)u;
                    g.flags = flags | MP3S_RF_LISTED;     // one writer per unit in k_rate_redo: a chain that reaches this unit leaves it to its entry
                    redo_list[REDO_HEAD + REDO_CAP + at] = hiding ? c : (cursor_in ? cursor_in[u] : 0);
                    if (hiding) cursor_in[u] = c;
#pragma unroll
                    for (int j = 0; j < 4; j++) redo_list[REDO_HEAD + 2 * REDO_CAP + 4 * at + j] = ch[j];
                }
            } else if (flags & MP3S_RF_LISTED) {
                g.flags = flags & ~MP3S_RF_LISTED;        // not on THIS check's list: the mark of the list before goes (a chain of the next round may
            }                                             // run through the unit again; and the records the caller reads carry no internal mark)
            if (flags & MP3S_RF_STEP_RANGE) err_here = 1;
            if (active) cur += g.n_tables;
            else {   // silent unit: everything is inherited (quantizerStepSize and addresses pass through)
                g.address[0] = ch[0]; g.address[1] = ch[1]; g.address[2] = ch[2];
                g.quantizer_step = ch[3];
            }
        }
        if (carry_used) atomicOr(&seg_out[s].carry_used, 1);
        if (f == sg.first_frame + sg.n_frames - 1) {
            // the chains behind the stream's last frame: this frame's active units, else what it inherited
            seg_out[s].cursor = cur;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const mp3s_gr_out &g = gr[(long)f * 4 + k];   // (silent units carry their inherited values by now)
                seg_out[s].chain[k][0] = g.address[0]; seg_out[s].chain[k][1] = g.address[1];
                seg_out[s].chain[k][2] = g.address[2]; seg_out[s].chain[k][3] = g.quantizer_step;
            }
        }
    }
    const int n_redo = (int)wave_add_u32((uint32_t)redo_here);
    const bool any_err = __ballot(err_here != 0) != 0;
    if ((threadIdx.x & 63) == 0) {
        if (n_redo) atomicAdd(&verdict[0], n_redo);
        if (any_err) atomicOr(&verdict[1], 1);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The message cursor, decided on the device.  A unit sees the message only through the <= 3 bits at its cursor, so the
// units a message can reach are run once per possibility inside the rate-loop launch itself (RateVariants:
// the 8 three-bit patterns, "two bits left", "one bit left"; the launch's own run of the unit, with the cursor behind
// every message, is the eleventh: "nothing left").  What remains is the serial part -- which possibility does unit j
// see? -- and that is a composition of small maps: with d = 3j - (cursor - start) (the tables the units so far fell
// short of three each), unit j maps d to d + 3 - tables(j, possibility at that cursor).  One workgroup
// per stream works through the planned units in rounds of SEL_ROUND: it keeps two byte arrays of the round in LDS -- the
// possibility at every cursor position the round can see, the table count of every (possibility, unit) -- and its 16
// waves compose a chunk of units each (a lane carries the states lane and lane + 64 through the chunk: two LDS reads per
// unit and state); the chunk totals give every chunk its starting state, a second walk gives every unit its own, and the
// last chunk's end is the exact d the next round starts from: the states of a round are d RELATIVE to that (0 <= d - d0 <
// 128), so the reach of a plan is not limited by LDS or by the total the chain falls short of (until r02e: one round of
// at most 2 048 units, absolute d).  A round whose chain leaves that range all the same -- a silence: such units take no
// table and d grows by three with each -- is done again in rounds of SEL_SAFE units, which cannot leave it (3 x 32 < 128),
// up to the next multiple of SEL_ROUND.  The entry each unit takes goes into
// a list that k_scatter_entries (many workgroups: one compute unit alone copies too slowly) works off.  cursor[] receives
// what each unit really saw, for the check in k_chain_apply, which stays the judge: a stream the plan did not cover (the
// message reached further than the planned units) fails that check.
constexpr int SEL_THREADS = 1024, SEL_WAVES = SEL_THREADS / 64, SEL_D = 128;
constexpr int SEL_ROUND = SEL_THREADS;           // units per round
constexpr int SEL_SAFE = 32;                     // ... of a stretch where d grows faster than a round of SEL_ROUND can follow
constexpr int SEL_NONE = MP3S_SELECT_VARIANTS;   // index of "the unit's own run" in the per-unit table counts

__host__ __device__ inline size_t select_lds_bytes(int reach)
{
    const int rs = reach < SEL_ROUND ? reach : SEL_ROUND;
    return (size_t)rs * (SEL_NONE + 1) + (size_t)(3 * rs + SEL_D + 8) + (size_t)rs + SEL_WAVES * SEL_D + 64;
}

// pairs: int2 [n_segs][max_reach] = (entry or -1, unit) for k_scatter_entries
__global__ __launch_bounds__(SEL_THREADS) void k_chain_select(const mp3s_chain_seg *__restrict__ segs, const mp3s_select_span *__restrict__ spans,
                                                              int max_reach, const uint8_t *__restrict__ hide,
                                                              const uint8_t *__restrict__ tabv, const mp3s_gr_out *__restrict__ out,
                                                              int32_t *__restrict__ cursor, int2 *__restrict__ pairs)
{
    extern __shared__ uint8_t sel_lds[];
    __shared__ int overflow, d_next, d_try, sat_round;
    const int s = blockIdx.x;
    const mp3s_select_span sp = spans[s];
    const int R = sp.reach;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int2 *my_pairs = pairs + (long)s * max_reach;
    if (R <= 0) {
        for (int j = tid; j < max_reach; j += SEL_THREADS) my_pairs[j] = make_int2(-1, 0);
        return;
    }
    const mp3s_chain_seg sg = segs[s];
    const long c0 = sg.hide_begin, end = sg.hide_end;
    const long u0 = (long)sg.first_frame * 4;
    const int RS = min(R, SEL_ROUND);                         // row length of the LDS arrays
    // the two "bits left" rows of the entries start at unit `tail` (MP3S_SELECT_TAIL_FIRST: no unit in front can meet them)
    const long left = end - c0;
    const int tail = (int)MP3S_SELECT_TAIL_FIRST(left, (long)R), R2 = R - tail;
    const long row8 = (long)sp.first_entry + 8L * R - tail;    // entry (v >= 8, j) = row8 + (v - 8) * R2 + j
    uint8_t *tabs = sel_lds;                                  // [11][RS]: table counts of the entries, [10] = the unit's own run
    uint8_t *code = tabs + (size_t)RS * (SEL_NONE + 1);       // [3RS + 128 + 8]: the possibility a unit sees with its cursor at start + rel0 + p
    uint8_t *delta = code + 3 * RS + SEL_D + 8;               // [RS]: d - d0 in front of every unit
    uint8_t *tot = delta + RS;                                // [16][128]
    if (tid == 0) { overflow = 0; d_next = 0; }
    int size = SEL_ROUND;
    for (int J = 0; J < R;) {
        __syncthreads();                                      // (the round before is through with the arrays; d_next is there)
        if (tid == 0) sat_round = 0;
        const int Rs = min(size, R - J);
        const int d0 = d_next;
        const long rel0 = 3L * J - d0 - (SEL_D - 1);          // the lowest cursor position (from the start) a state of this round can be at
        if (c0 + rel0 >= end) {                               // the chain is behind the message: nothing to replace from here on
            for (int jj = tid; jj < Rs; jj += SEL_THREADS) my_pairs[J + jj] = make_int2(-1, 0);
            J += Rs;
            continue;
        }
        for (int v = 0; v < 8; v++)
            for (int jj = tid; jj < Rs; jj += SEL_THREADS) tabs[v * RS + jj] = tabv[sp.first_entry + (long)v * R + J + jj];
        for (int v = 8; v < SEL_NONE; v++)
            for (int jj = tid; jj < Rs; jj += SEL_THREADS) tabs[v * RS + jj] = J + jj >= tail ? tabv[row8 + (long)(v - 8) * R2 + J + jj] : (uint8_t)0;
        for (int jj = tid; jj < Rs; jj += SEL_THREADS) tabs[SEL_NONE * RS + jj] = (uint8_t)out[u0 + J + jj].n_tables;
        for (int p = tid; p < 3 * Rs + SEL_D + 8; p += SEL_THREADS) {
            const long rel = rel0 + p, cur = c0 + rel;
            int v = SEL_NONE;                                 // (also: in front of the start, where the chain cannot be)
            if (rel >= 0) {
                if (cur + 3 <= end) v = (hide[cur] & 1) * 4 + (hide[cur + 1] & 1) * 2 + (hide[cur + 2] & 1);
                else if (cur == end - 2) v = 8;
                else if (cur == end - 1) v = 9;
            }
            code[p] = (uint8_t)v;
        }
        __syncthreads();
        // unit J + jj, state d (relative) -> the next state (d stays once the cursor is behind the message: it no longer
        // matters there, and the cursor only moves on); states the chain cannot be in stay too
        auto step = [&](int jj, int d) -> int {
            const int v = code[3 * jj + (SEL_D - 1) - d];
            return v == SEL_NONE ? d : min(d + 3 - (int)tabs[v * RS + jj], SEL_D - 1);
        };
        const int chunk = (Rs + SEL_WAVES - 1) / SEL_WAVES, j0 = wave * chunk, j1 = min(Rs, j0 + chunk);
        {
            int m0 = lane, m1 = lane + 64;
            for (int jj = j0; jj < j1; jj++) { m0 = step(jj, m0); m1 = step(jj, m1); }
            tot[wave * SEL_D + lane] = (uint8_t)m0;
            tot[wave * SEL_D + 64 + lane] = (uint8_t)m1;
        }
        __syncthreads();
        {
            int d = 0;
            for (int w = 0; w < wave; w++) d = tot[w * SEL_D + d];
            bool sat = false;
            for (int jj = j0; jj < j1; jj++) {
                if (lane == 0) delta[jj] = (uint8_t)d;
                sat |= d >= SEL_D - 1;
                d = step(jj, d);
            }
            if (sat && lane == 0) sat_round = 1;
            if (wave == SEL_WAVES - 1 && lane == 0) d_try = d0 + d;   // (a wave without units of its own walks nothing)
        }
        __syncthreads();
        if (sat_round) {
            if (size > SEL_SAFE) { size = SEL_SAFE; continue; }   // the same units again, a few at a time
            if (tid == 0) overflow = 1;                           // (cannot happen: 3 x SEL_SAFE < SEL_D)
        }
        for (int jj = tid; jj < Rs; jj += SEL_THREADS) {
            const int p = 3 * jj + (SEL_D - 1) - (int)delta[jj];
            const int v = code[p];
            // (entry, cursor position from the start) for now: the unit number goes in once the whole plan held
            const int j = J + jj;
            if (v >= 8 && v != SEL_NONE && j < tail) overflow = 1;        // (cannot happen: see MP3S_SELECT_TAIL_FIRST)
            my_pairs[j] = v == SEL_NONE ? make_int2(-1, 0)
                                        : make_int2(v < 8 ? sp.first_entry + v * R + j : (int)(row8 + (long)(v - 8) * R2 + j), (int)(rel0 + p));
        }
        if (tid == 0) d_next = d_try;
        J += Rs;
        if (size == SEL_SAFE && J % SEL_ROUND == 0) size = SEL_ROUND;
    }
    __syncthreads();
    const bool bad = overflow != 0;
    for (int j = tid; j < max_reach; j += SEL_THREADS) {
        int src = -1;
        if (j < R && !bad) {
            const int2 pr = my_pairs[j];
            src = pr.x;
            if (src >= 0) cursor[u0 + j] = (int32_t)(c0 + pr.y);
        }
        my_pairs[j] = make_int2(src, (int)(u0 + j));
    }
}

}  // namespace mp3s
