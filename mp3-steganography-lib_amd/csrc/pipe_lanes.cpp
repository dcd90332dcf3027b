// The streams of a pipe, chosen by rehearsal: the runtime multiplexes streams onto a few hardware queues,
// and a pipeline whose streams share queues loses its overlap.
#include "pipe_internal.h"

namespace {

// Candidate streams of a device, made once per process: four of the highest priority (copies) and four of the lowest (the
// front end: its workgroups fill in beside the compute stream's instead of competing with them).  The runtime multiplexes
// streams onto a few hardware queues; which queue a stream gets depends on everything the process has created before it
// (round 2 believed priorities chose the set of queues; a context's own pipe in front of a user's pipe showed otherwise:
// 0.97 - 1.25 instead of 0.81 ms per batch, tools/pipe_queue_probe.py; the first of three contexts ran its one-file calls at
// half speed, tools/bench_queue_probe4.py), and a pipeline whose streams share queues with its compute stream loses its
// overlap -- every kernel of it takes longer, not only the ones that wait.  The runtime does not tell which stream sits
// where, and a spin kernel beside an empty one or beside a small copy does not show it either (all eight candidates passed
// that test on a context that then ran at half speed).  So the pipe REHEARSES: four miniature jobs -- a copy up, a spin on
// the front-end stream, two spins on the compute stream, a copy down, chained by events exactly as issue_fast chains a
// job's stages -- through each rotation of the candidates, and keeps the rotation that got them through fastest
// (about 1 ms per rotation when a pipe is made).
struct LaneChoice { hipStream_t ctx_stream; int want_tail; hipStream_t up, down, huff, comp, tail, dec, img; };
struct LanePool {
    std::vector<LaneChoice> chosen;       // what the rehearsal decided for a context's stream (asked again only by another context)
    hipStream_t hi[4] = {nullptr, nullptr, nullptr, nullptr}, lo[4] = {nullptr, nullptr, nullptr, nullptr}, cs[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev[4][6] = {};
    hipEvent_t t0 = nullptr, t1 = nullptr;
    uint8_t *d_buf = nullptr, *h_buf = nullptr;
    bool ok = false;
};
constexpr size_t kRehearseBytes = (size_t)256 << 10;

// tail: the stream a job's last stage (three short spins: selection, chain check, packing) runs on, behind the job's compute
// stage and beside the next job's; null: on the compute stream itself
// dec: the stream a job's decode stage (the first of the two compute spins) runs on, ahead of the compute stage of the job in front
float rehearse(LanePool &lp, hipStream_t comp, hipStream_t up, hipStream_t down, hipStream_t huff, hipStream_t tail = nullptr, hipStream_t dec = nullptr)
{
    float ms = 1e9f;
    hipStream_t ts = tail ? tail : comp;
    hipStream_t dst = dec ? dec : comp;
    bool ok = hipEventRecord(lp.t0, up) == hipSuccess;
    for (int k = 0; k < 4 && ok; k++) {
        ok = hipMemcpyAsync(lp.d_buf + k * kRehearseBytes, lp.h_buf + k * kRehearseBytes, kRehearseBytes, hipMemcpyHostToDevice, up) == hipSuccess &&
             hipEventRecord(lp.ev[k][0], up) == hipSuccess && hipStreamWaitEvent(huff, lp.ev[k][0], 0) == hipSuccess && launch_spin(huff, 60) == 0 &&
             hipEventRecord(lp.ev[k][1], huff) == hipSuccess && hipStreamWaitEvent(dst, lp.ev[k][1], 0) == hipSuccess && launch_spin(dst, 50) == 0;
        if (ok && dec) ok = hipEventRecord(lp.ev[k][5], dec) == hipSuccess && hipStreamWaitEvent(comp, lp.ev[k][5], 0) == hipSuccess;
        ok = ok && launch_spin(comp, 50) == 0 && launch_noop(comp) == 0 && hipEventRecord(lp.ev[k][2], comp) == hipSuccess;
        if (ok && tail) ok = hipStreamWaitEvent(tail, lp.ev[k][2], 0) == hipSuccess;
        ok = ok && launch_spin(ts, 10) == 0 && launch_spin(ts, 10) == 0 && launch_spin(ts, 20) == 0 && hipEventRecord(lp.ev[k][4], ts) == hipSuccess &&
             hipStreamWaitEvent(down, lp.ev[k][4], 0) == hipSuccess &&
             hipMemcpyAsync(lp.h_buf + (4 + k) * kRehearseBytes, lp.d_buf + k * kRehearseBytes, kRehearseBytes, hipMemcpyDeviceToHost, down) == hipSuccess &&
             hipEventRecord(lp.ev[k][3], down) == hipSuccess;
    }
    ok = ok && hipEventRecord(lp.t1, down) == hipSuccess;
    (void)hipStreamSynchronize(up); (void)hipStreamSynchronize(huff); (void)hipStreamSynchronize(comp); (void)hipStreamSynchronize(down);
    if (tail) (void)hipStreamSynchronize(tail);
    if (dec) (void)hipStreamSynchronize(dec);
    if (!ok || hipEventElapsedTime(&ms, lp.t0, lp.t1) != hipSuccess) return 1e9f;
    return ms;
}

std::mutex &lane_mu() { static std::mutex *m = new std::mutex(); return *m; }
std::vector<LanePool> &lane_pools() { static auto *v = new std::vector<LanePool>(); return *v; }

}  // namespace

int pick_lanes(mp3s_ctx *c, hipStream_t *up, hipStream_t *down, hipStream_t *huff, hipStream_t *comp /* a stream to compute on instead of the context's, or null */,
               hipStream_t *tail /* a stream for the tail of a job, or null */, int want_tail /* 0: none, 1: always, 2: if the rehearsal is faster with it */,
               hipStream_t *dec /* a stream for the decode transforms, or null */, int want_dec,
               hipStream_t *img /* a second copy-up stream (the file pieces of a one-file call), or null */)
{
    std::lock_guard<std::mutex> g(lane_mu());
    auto &pools = lane_pools();
    if ((size_t)c->device >= pools.size()) pools.resize((size_t)c->device + 1);
    LanePool &lp = pools[(size_t)c->device];
    if (!lp.ok) {
        int prio_low = 0, prio_high = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
        const int huff_prio = prio_low;
        bool ok = hipEventCreate(&lp.t0) == hipSuccess && hipEventCreate(&lp.t1) == hipSuccess && hipMalloc((void **)&lp.d_buf, 4 * kRehearseBytes) == hipSuccess &&
                  hipHostMalloc((void **)&lp.h_buf, 8 * kRehearseBytes, hipHostMallocDefault) == hipSuccess;
        for (int k = 0; k < 4 && ok; k++)
            for (int q = 0; q < 6 && ok; q++) ok = hipEventCreateWithFlags(&lp.ev[k][q], hipEventDisableTiming) == hipSuccess;
        for (int i = 0; i < 4 && ok; i++) ok = hipStreamCreateWithPriority(&lp.hi[i], hipStreamNonBlocking, prio_high) == hipSuccess;
        for (int i = 0; i < 4 && ok; i++) ok = hipStreamCreateWithPriority(&lp.lo[i], hipStreamNonBlocking, huff_prio) == hipSuccess;
        for (int i = 0; i < 4 && ok; i++) ok = hipStreamCreateWithFlags(&lp.cs[i], hipStreamNonBlocking) == hipSuccess;
        if (!ok) return 1;   // (what was created stays with the process)
        std::memset(lp.h_buf, 0, 8 * kRehearseBytes);
        lp.ok = true;
    }
    for (const LaneChoice &k : lp.chosen)
        if (k.ctx_stream == c->stream && k.want_tail == want_tail) {
            *up = k.up; *down = k.down; *huff = k.huff;
            if (comp) *comp = k.comp;
            if (tail) *tail = k.tail;
            if (dec) *dec = k.dec;
            if (img) *img = k.img;
            return 0;
        }
    (void)hipStreamSynchronize(c->stream);
    (void)rehearse(lp, c->stream, lp.hi[0], lp.hi[1], lp.lo[0]);      // (first launches: not a measurement)
    // the context's own stream with every rotation of the lanes; if none of them gets the rehearsal through as fast as a
    // pipeline without shared queues does, the compute candidates too (the pipe then computes on one of those)
    int best = 0, best_cs = -1;
    float best_ms = 1e9f;
    std::string seen;
    auto tryout = [&](int ci, int r) {
        hipStream_t comp_s = ci < 0 ? c->stream : lp.cs[ci];
        const float ms = std::min(rehearse(lp, comp_s, lp.hi[r], lp.hi[(r + 1) & 3], lp.lo[r]), rehearse(lp, comp_s, lp.hi[r], lp.hi[(r + 1) & 3], lp.lo[r]));
        if (trace_on()) { char b[32]; snprintf(b, sizeof b, " %.3f", ms); seen += b; }
        if (ms < best_ms * 0.97f) { best_ms = ms; best = r; best_cs = ci; }
    };
    for (int r = 0; r < 4; r++) tryout(-1, r);
    const float own_ms = best_ms;
    const int own_best = best;
    if (comp && best_ms > 0.80f)          // (4 x (2 x 50 + 40) us on the compute stream behind one front-end spin never take less than 0.72 ms: this one lost its overlap somewhere)
        for (int ci = 0; ci < 4 && best_ms > 0.78f; ci++) {
            if (trace_on()) seen += " |";
            for (int r = 0; r < 4; r++) tryout(ci, r);
        }
    if (best_cs >= 0 && best_ms > own_ms * 0.93f) { best_cs = -1; best_ms = own_ms; best = own_best; }   // (not worth leaving the context's stream for)
    // a stream for the tail of a job (MP3S_OPT_PIPE_TAIL): the candidate, other than the compute stream, that gets the
    // rehearsal through fastest -- kept if that is faster than the tail on the compute stream itself
    int best_tail = -1;
    if (tail) {
        *tail = nullptr;
        hipStream_t comp_s = best_cs < 0 ? c->stream : lp.cs[best_cs];
        float tail_ms = best_ms * (want_tail == 1 ? 1.05f : 0.97f);   // (1: unless it clearly loses -- the real kernels gain more from it than spins do)
        if (trace_on()) seen += " | tail:";
        for (int ti = 0; ti < 4 && want_tail; ti++) {
            if (ti == best_cs) continue;
            const float ms = std::min(rehearse(lp, comp_s, lp.hi[best], lp.hi[(best + 1) & 3], lp.lo[best], lp.cs[ti]), rehearse(lp, comp_s, lp.hi[best], lp.hi[(best + 1) & 3], lp.lo[best], lp.cs[ti]));
            if (trace_on()) { char b[32]; snprintf(b, sizeof b, " %.3f", ms); seen += b; }
            if (ms < tail_ms) { tail_ms = ms; best_tail = ti; }
        }
        if (best_tail >= 0) *tail = lp.cs[best_tail];
    }
    // a stream for the decode transforms: the candidate (other than the compute and tail streams) that does best, if it gains
    int best_dec = -1;
    if (dec) {
        *dec = nullptr;
        hipStream_t comp_s = best_cs < 0 ? c->stream : lp.cs[best_cs];
        hipStream_t tail_s = best_tail >= 0 ? lp.cs[best_tail] : nullptr;
        float ref_ms = std::min(rehearse(lp, comp_s, lp.hi[best], lp.hi[(best + 1) & 3], lp.lo[best], tail_s), rehearse(lp, comp_s, lp.hi[best], lp.hi[(best + 1) & 3], lp.lo[best], tail_s));
        float dec_ms = ref_ms * (want_dec == 2 ? 10.f : 0.95f);
        if (trace_on()) { char b[48]; snprintf(b, sizeof b, " | dec (%.3f):", ref_ms); seen += b; }
        for (int di = 0; di < 4 && want_dec; di++) {
            if (di == best_cs || di == best_tail) continue;
            const float ms = std::min(rehearse(lp, comp_s, lp.hi[best], lp.hi[(best + 1) & 3], lp.lo[best], tail_s, lp.cs[di]), rehearse(lp, comp_s, lp.hi[best], lp.hi[(best + 1) & 3], lp.lo[best], tail_s, lp.cs[di]));
            if (trace_on()) { char b[32]; snprintf(b, sizeof b, " %.3f", ms); seen += b; }
            if (ms < dec_ms) { dec_ms = ms; best_dec = di; }
        }
        if (best_dec >= 0) *dec = lp.cs[best_dec];
    }
    if (trace_on()) fprintf(stderr, "mp3s: pipe lanes rehearsed:%s ms -> compute stream %d, rotation %d (%.3f ms), tail stream %d, decode stream %d\n", seen.c_str(), best_cs, best, best_ms, best_tail, best_dec);
    if (comp) *comp = best_cs >= 0 ? lp.cs[best_cs] : nullptr;
    *up = lp.hi[best]; *down = lp.hi[(best + 1) & 3]; *huff = lp.lo[best];
    if (img) *img = lp.hi[(best + 2) & 3];
    lp.chosen.push_back({c->stream, want_tail, *up, *down, *huff, comp ? *comp : nullptr, tail ? *tail : nullptr, dec ? *dec : nullptr, lp.hi[(best + 2) & 3]});
    return 0;
}

// a context is going away: its stream's address may come back as another context's
void forget_lanes(mp3s_ctx *c)
{
    std::lock_guard<std::mutex> g(lane_mu());
    auto &pools = lane_pools();
    if ((size_t)c->device >= pools.size()) return;
    auto &v = pools[(size_t)c->device].chosen;
    v.erase(std::remove_if(v.begin(), v.end(), [&](const LaneChoice &k) { return k.ctx_stream == c->stream; }), v.end());
}
