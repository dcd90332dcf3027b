// C-ABI of the library (include/mp3s.h): context, device memory, the batch entry points and the
// whole-stream pipelines that glue the host stages to the HIP kernels.  No CPU fallback exists for
// the transforms: every path that needs them goes through launch_* in mp3s_device.hip.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mp3s.h"
#include "mp3s_device.h"
#include "mp3s_host.h"

using namespace mp3s;

namespace {

thread_local std::string g_err;
int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIPCHK(call)                                                                                 \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) return fail(MP3S_E_HIP, "%s: %s", #call, hipGetErrorString(e_));       \
    } while (0)

}  // namespace

struct mp3s_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_order = nullptr;
    int32_t *d_sync = nullptr;        // {finished workgroups, error bits} of the pack kernel in flight: self-clearing
    void *scratch = nullptr; size_t scratch_bytes = 0;
    Profiler prof;
    // device buffers of the stream pipelines, kept between calls (hipMalloc/hipFree cost more than a small file's work)
    static constexpr int kPoolSlots = 32;
    void *pool[kPoolSlots] = {nullptr};
    size_t pool_bytes[kPoolSlots] = {0};
    void *grab(int slot, size_t bytes)
    {
        if (bytes < 16) bytes = 16;
        if (pool_bytes[slot] >= bytes) return pool[slot];
        if (pool[slot]) { hipStreamSynchronize(stream); hipFree(pool[slot]); pool[slot] = nullptr; pool_bytes[slot] = 0; }
        const size_t want = bytes + bytes / 4;   // head room: similar-sized files reuse the buffer
        if (hipMalloc(&pool[slot], want) != hipSuccess) { pool[slot] = nullptr; return nullptr; }
        pool_bytes[slot] = want;
        return pool[slot];
    }
    // host-side work arrays of the encoder, kept between calls: beyond a few MB a fresh vector means fresh pages from
    // the kernel on every call (page faults cost more than the work done in them)
    std::vector<int32_t> h_cursor, h_state, h_want, h_state_want;
    std::vector<uint8_t> h_in;
    int ensure_scratch(size_t bytes)
    {
        if (bytes <= scratch_bytes) return 0;
        if (scratch) { hipFree(scratch); scratch = nullptr; scratch_bytes = 0; }
        hipError_t e = hipMalloc(&scratch, bytes);
        if (e != hipSuccess) return fail(MP3S_E_NOMEM, "hipMalloc(%zu) for scratch: %s", bytes, hipGetErrorString(e));
        scratch_bytes = bytes;
        return 0;
    }
};

// Page-locked host memory for large results (decoded PCM): the device writes it at PCIe speed, no bounce buffer, no
// page faults.  Pinning costs more than the copy it saves, so blocks are kept and reused: process-wide, because a
// result may outlive the context that produced it.  (Blocks still cached at exit are left to the OS.)
class PinnedBlock {
public:
    PinnedBlock() = default;
    PinnedBlock(const PinnedBlock &) = delete;
    PinnedBlock &operator=(const PinnedBlock &) = delete;
    ~PinnedBlock() { release(); }
    bool reserve(size_t bytes)
    {
        if (bytes <= cap_) return true;
        release();
        if (bytes > kMaxPinned) {   // hours of audio in one call: pinning gigabytes costs seconds, ordinary memory then
            p_ = (uint8_t *)std::malloc(bytes);
            if (!p_) return false;
            cap_ = bytes; pinned_ = false;
            return true;
        }
        pinned_ = true;
        {
            std::lock_guard<std::mutex> g(mu());
            auto &fl = free_list();
            size_t best = fl.size();
            for (size_t i = 0; i < fl.size(); i++)
                if (fl[i].second >= bytes && (best == fl.size() || fl[i].second < fl[best].second)) best = i;
            if (best < fl.size()) { p_ = fl[best].first; cap_ = fl[best].second; fl.erase(fl.begin() + best); return true; }
        }
        const size_t want = bytes + bytes / 8 + (1 << 16);
        void *q = nullptr;
        if (hipHostMalloc(&q, want, hipHostMallocDefault) != hipSuccess) return false;
        p_ = (uint8_t *)q; cap_ = want;
        return true;
    }
    uint8_t *data() const { return p_; }

private:
    void release()
    {
        if (!p_) return;
        if (!pinned_) { std::free(p_); p_ = nullptr; cap_ = 0; return; }
        std::lock_guard<std::mutex> g(mu());
        auto &fl = free_list();
        size_t held = 0;
        for (auto &e : fl) held += e.second;
        if (fl.size() < 8 && held + cap_ <= kMaxPinned) fl.emplace_back(p_, cap_);
        else hipHostFree(p_);
        p_ = nullptr; cap_ = 0;
    }
    static std::mutex &mu() { static std::mutex *m = new std::mutex(); return *m; }
    static std::vector<std::pair<uint8_t *, size_t>> &free_list()
    {
        static auto *v = new std::vector<std::pair<uint8_t *, size_t>>();
        return *v;
    }
    static constexpr size_t kMaxPinned = (size_t)1 << 30;
    uint8_t *p_ = nullptr;
    size_t cap_ = 0;
    bool pinned_ = true;
};

struct mp3s_multi;
struct mp3s_buf {
    std::shared_ptr<mp3s_multi> multi;
    ParsedStream parsed;
    ScannedStream scanned;
    std::vector<uint8_t> bytes;      // generic payload (pcm / mp3)
    std::vector<uint8_t> bits;
    std::vector<mp3s_gr_out> gr;
    std::vector<int32_t> scfsi;
    std::vector<std::unique_ptr<mp3s_buf>> parts;   // results of the batches of a multi-file call
    // encoder results: MP3 bytes and GrInfo records land in page-locked blocks and are handed out from there
    PinnedBlock big[2];
    uint8_t *mp3 = nullptr;
    mp3s_gr_out *gr_out = nullptr;
};

// MP3S_TRACE=1: phase timings of the file pipelines on stderr
static bool trace_on() { static const bool on = getenv("MP3S_TRACE") != nullptr; return on; }
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// f(i) for i in [0, n) on a few host threads: the front ends of the files of a batch are independent.  Small batches
// (by bytes) stay on the calling thread -- starting a thread costs about what scanning 100 KB does.
template <class F>
static void parallel_files(int n, size_t total_bytes, F f)
{
    const unsigned hw = std::thread::hardware_concurrency();
    const int workers = (int)std::min<size_t>({(size_t)n, (size_t)std::min(hw ? hw : 1u, 16u), total_bytes / (256u << 10) + 1});
    if (workers <= 1) {
        for (int i = 0; i < n; i++) f(i);
        return;
    }
    std::atomic<int> next{0};
    auto run = [&]() { for (int i; (i = next.fetch_add(1)) < n;) f(i); };
    std::vector<std::thread> pool;
    for (int w = 1; w < workers; w++) pool.emplace_back(run);
    run();
    for (auto &t : pool) t.join();
}

extern "C" {

const char *mp3s_last_error(void) { return g_err.c_str(); }
const char *mp3s_version(void) { return "mp3s-hip 0.1 (gfx950)"; }
void mp3s_buf_free(mp3s_buf *b) { delete b; }
const void *mp3s_debug_tables(size_t *bytes)
{
    if (bytes) *bytes = sizeof(DevTables);
    return &host_tables().dev;
}

int mp3s_debug_scfsi_energies(const int32_t *xr576, int sr_idx, int32_t *en22)
{
    if (!xr576 || !en22 || sr_idx < 0 || sr_idx > 2) return fail(MP3S_E_ARG, "bad argument");
    host_scfsi_energies(xr576, sr_idx, en22);
    return MP3S_OK;
}

int mp3s_ctx_create(int device, mp3s_ctx **out)
{
    if (!out) return fail(MP3S_E_ARG, "out is null");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(MP3S_E_NO_DEVICE, "no HIP device available (%s); the transforms have no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(MP3S_E_ARG, "device %d out of range (0..%d)", device, n - 1);
    e = hipSetDevice(device);
    if (e != hipSuccess) return fail(MP3S_E_NO_DEVICE, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
    mp3s_ctx *c = new mp3s_ctx();
    c->device = device;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_order, hipEventDisableTiming) != hipSuccess) {
        mp3s_ctx_destroy(c);   // releases whatever was created
        return fail(MP3S_E_NO_DEVICE, "stream/event creation failed");
    }
    if (hipMalloc((void **)&c->d_sync, 16) != hipSuccess || hipMemsetAsync(c->d_sync, 0, 16, c->stream) != hipSuccess) {
        mp3s_ctx_destroy(c);
        return fail(MP3S_E_NO_DEVICE, "device scratch allocation failed");
    }
    const int rc = dev_upload_tables(c->stream);
    if (rc) {
        mp3s_ctx_destroy(c);
        return fail(MP3S_E_NO_DEVICE, "constant table upload failed: %s (is this a gfx950 device?)",
                    hipGetErrorString((hipError_t)rc));
    }
    *out = c;
    return MP3S_OK;
}

void mp3s_ctx_destroy(mp3s_ctx *c)
{
    if (!c) return;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->scratch) hipFree(c->scratch);
    if (c->d_sync) hipFree(c->d_sync);
    for (void *q : c->pool) if (q) hipFree(q);
    if (c->ev0) hipEventDestroy(c->ev0);
    if (c->ev1) hipEventDestroy(c->ev1);
    if (c->ev_order) hipEventDestroy(c->ev_order);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

int mp3s_device_name(mp3s_ctx *c, char *buf, size_t n)
{
    if (!c || !buf || !n) return fail(MP3S_E_ARG, "bad argument");
    hipDeviceProp_t p;
    HIPCHK(hipGetDeviceProperties(&p, c->device));
    snprintf(buf, n, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
    return MP3S_OK;
}

int mp3s_sync(mp3s_ctx *c)
{
    if (!c) return fail(MP3S_E_ARG, "ctx is null");
    HIPCHK(hipStreamSynchronize(c->stream));
    return MP3S_OK;
}

int mp3s_ctx_wait(mp3s_ctx *c, mp3s_ctx *other)
{
    if (!c || !other) return fail(MP3S_E_ARG, "ctx is null");
    if (c == other) return MP3S_OK;
    if (c->device != other->device) return fail(MP3S_E_ARG, "contexts on different devices");
    HIPCHK(hipEventRecord(other->ev_order, other->stream));
    HIPCHK(hipStreamWaitEvent(c->stream, other->ev_order, 0));
    return MP3S_OK;
}

int mp3s_dev_alloc(mp3s_ctx *c, size_t bytes, void **dptr)
{
    if (!c || !dptr) return fail(MP3S_E_ARG, "bad argument");
    HIPCHK(hipSetDevice(c->device));
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 16);
    if (e != hipSuccess) return fail(MP3S_E_NOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
    return MP3S_OK;
}
int mp3s_dev_free(mp3s_ctx *c, void *dptr)
{
    if (!c) return fail(MP3S_E_ARG, "ctx is null");
    if (dptr) { HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipFree(dptr)); }
    return MP3S_OK;
}
int mp3s_dev_upload(mp3s_ctx *c, void *dptr, const void *host, size_t bytes)
{
    if (!c || (!dptr && bytes) || (!host && bytes)) return fail(MP3S_E_ARG, "bad argument");
    if (!bytes) return MP3S_OK;
    HIPCHK(hipMemcpyAsync(dptr, host, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return MP3S_OK;
}
int mp3s_dev_download(mp3s_ctx *c, void *host, const void *dptr, size_t bytes)
{
    if (!c || (!dptr && bytes) || (!host && bytes)) return fail(MP3S_E_ARG, "bad argument");
    if (!bytes) return MP3S_OK;
    HIPCHK(hipMemcpyAsync(host, dptr, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return MP3S_OK;
}
int mp3s_dev_memset(mp3s_ctx *c, void *dptr, int value, size_t bytes)
{
    if (!c || !dptr) return fail(MP3S_E_ARG, "bad argument");
    HIPCHK(hipMemsetAsync(dptr, value, bytes, c->stream));
    return MP3S_OK;
}
int mp3s_timer_start(mp3s_ctx *c)
{
    if (!c) return fail(MP3S_E_ARG, "ctx is null");
    HIPCHK(hipEventRecord(c->ev0, c->stream));
    return MP3S_OK;
}
int mp3s_timer_stop(mp3s_ctx *c, float *ms)
{
    if (!c || !ms) return fail(MP3S_E_ARG, "bad argument");
    HIPCHK(hipEventRecord(c->ev1, c->stream));
    HIPCHK(hipEventSynchronize(c->ev1));
    HIPCHK(hipEventElapsedTime(ms, c->ev0, c->ev1));
    return MP3S_OK;
}

int mp3s_profile_enable(mp3s_ctx *c, int on)
{
    if (!c) return fail(MP3S_E_ARG, "ctx is null");
    HIPCHK(hipStreamSynchronize(c->stream));
    c->prof.enabled = on != 0;
    c->prof.n_pairs = 0;
    for (int k = 0; k < K_COUNT; k++) { c->prof.total_ms[k] = 0; c->prof.count[k] = 0; }
    return MP3S_OK;
}

int mp3s_profile_select(mp3s_ctx *c, unsigned mask)
{
    if (!c) return fail(MP3S_E_ARG, "ctx is null");
    c->prof.mask = mask;
    return MP3S_OK;
}

int mp3s_profile_collect(mp3s_ctx *c, double *total_ms, int64_t *launches, int n)
{
    if (!c || !total_ms || !launches || n < K_COUNT) return fail(MP3S_E_ARG, "bad argument (need %d slots)", (int)K_COUNT);
    HIPCHK(hipStreamSynchronize(c->stream));
    Profiler &p = c->prof;
    for (int i = 0; i < p.n_pairs; i++) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, p.ev[2 * i], p.ev[2 * i + 1]));
        p.total_ms[p.kid[i]] += ms;
        p.count[p.kid[i]] += 1;
    }
    p.n_pairs = 0;
    for (int k = 0; k < K_COUNT; k++) { total_ms[k] = p.total_ms[k]; launches[k] = p.count[k]; }
    static_assert(K_COUNT == MP3S_N_KERNELS, "kernel list out of sync with mp3s.h");
    return MP3S_OK;
}

// ------------------------------------------------------------------------------------------------ batches
int mp3s_decode_transform_dev(mp3s_ctx *c, const int16_t *d_is, const mp3s_granule_si *d_si, const mp3s_frame_hdr *d_hdr,
                              int n_frames, int nch, int n_halo, int out_format, void *d_pcm)
{
    if (!c || !d_is || !d_si || !d_hdr || !d_pcm) return fail(MP3S_E_ARG, "null pointer");
    if (n_frames <= 0 || nch < 1 || nch > 2 || n_halo < 0 || n_halo >= n_frames || out_format < 0 || out_format > 2)
        return fail(MP3S_E_ARG, "bad sizes: n_frames=%d nch=%d n_halo=%d fmt=%d", n_frames, nch, n_halo, out_format);
    int rc = c->ensure_scratch(dec_scratch_bytes(n_frames, nch));
    if (rc) return rc;
    const int e = launch_decode(c->stream, d_is, d_si, d_hdr, n_frames, nch, n_halo, out_format, d_pcm, c->scratch, &c->prof);
    if (e) return fail(MP3S_E_HIP, "decode launch: %s", hipGetErrorString((hipError_t)e));
    return MP3S_OK;
}

constexpr int kDecodeChunk = 16384;   // frames per decode launch group (scratch ~0.6 GB); chunks overlap by a 1-frame halo

static size_t pcm_elem(int fmt) { return fmt == MP3S_PCM_I16 ? 2 : (fmt == MP3S_PCM_F32 ? 4 : 8); }

// frame 0 of a batch always starts from zero state; later frames must not point forward
static int check_hdr(const mp3s_frame_hdr *hdr, int n)
{
    for (int f = 0; f < n; f++) {
        if (hdr[f].stream_first > (uint32_t)f) return fail(MP3S_E_ARG, "hdr[%d].stream_first=%u points forward", f, hdr[f].stream_first);
        if (f && hdr[f].stream_first != (uint32_t)f && hdr[f].stream_first != hdr[f - 1].stream_first)
            return fail(MP3S_E_ARG, "hdr[%d].stream_first is not monotone", f);
        if (hdr[f].sr_idx > 2) return fail(MP3S_E_ARG, "hdr[%d].sr_idx=%d", f, hdr[f].sr_idx);
    }
    if (n && hdr[0].stream_first != 0) return fail(MP3S_E_ARG, "hdr[0].stream_first must be 0");
    return 0;
}

int mp3s_decode_transform(mp3s_ctx *c, const int16_t *is, const mp3s_granule_si *si, const mp3s_frame_hdr *hdr,
                          int n_frames, int nch, int n_halo, int out_format, void *pcm)
{
    if (!c || !is || !si || !hdr || !pcm) return fail(MP3S_E_ARG, "null pointer");
    if (n_frames <= 0 || n_halo < 0 || n_halo >= n_frames) return fail(MP3S_E_ARG, "bad sizes");
    int rc = check_hdr(hdr, n_frames);
    if (rc) return rc;
    for (size_t i = 0; i < (size_t)n_frames * 2304; i++)
        if (is[i] > 8206 || is[i] < -8206) return fail(MP3S_E_ARG, "is[%zu]=%d outside +-8206", i, is[i]);
    HIPCHK(hipSetDevice(c->device));
    void *d_is = nullptr, *d_si = nullptr, *d_hdr = nullptr, *d_pcm = nullptr;
    const size_t b_is = (size_t)n_frames * 2304 * 2, b_si = (size_t)n_frames * 4 * sizeof(mp3s_granule_si),
                 b_hdr = (size_t)n_frames * sizeof(mp3s_frame_hdr),
                 b_pcm = (size_t)(n_frames - n_halo) * 1152 * nch * pcm_elem(out_format);
    rc = MP3S_OK;
    if (hipMalloc(&d_is, b_is) != hipSuccess || hipMalloc(&d_si, b_si) != hipSuccess ||
        hipMalloc(&d_hdr, b_hdr) != hipSuccess || hipMalloc(&d_pcm, b_pcm) != hipSuccess)
        rc = fail(MP3S_E_NOMEM, "hipMalloc failed for a %d-frame batch", n_frames);
    if (!rc) rc = mp3s_dev_upload(c, d_is, is, b_is);
    if (!rc) rc = mp3s_dev_upload(c, d_si, si, b_si);
    if (!rc) rc = mp3s_dev_upload(c, d_hdr, hdr, b_hdr);
    if (!rc) rc = mp3s_decode_transform_dev(c, (const int16_t *)d_is, (const mp3s_granule_si *)d_si,
                                            (const mp3s_frame_hdr *)d_hdr, n_frames, nch, n_halo, out_format, d_pcm);
    if (!rc) rc = mp3s_dev_download(c, pcm, d_pcm, b_pcm);
    hipStreamSynchronize(c->stream);
    hipFree(d_is); hipFree(d_si); hipFree(d_hdr); hipFree(d_pcm);
    return rc;
}

int mp3s_encode_transform_dev(mp3s_ctx *c, const int16_t *d_pcm, const mp3s_frame_hdr *d_hdr, int n_frames, int32_t *d_mdct)
{
    if (!c || !d_pcm || !d_hdr || !d_mdct) return fail(MP3S_E_ARG, "null pointer");
    if (n_frames <= 0) return fail(MP3S_E_ARG, "n_frames=%d", n_frames);
    int rc = c->ensure_scratch(enc_scratch_bytes(n_frames));
    if (rc) return rc;
    const int e = launch_encode(c->stream, d_pcm, d_hdr, n_frames, d_mdct, c->scratch, &c->prof);
    if (e) return fail(MP3S_E_HIP, "encode launch: %s", hipGetErrorString((hipError_t)e));
    return MP3S_OK;
}

int mp3s_encode_transform(mp3s_ctx *c, const int16_t *pcm, const mp3s_frame_hdr *hdr, int n_frames, int32_t *mdct)
{
    if (!c || !pcm || !hdr || !mdct) return fail(MP3S_E_ARG, "null pointer");
    if (n_frames <= 0) return fail(MP3S_E_ARG, "n_frames=%d", n_frames);
    int rc = check_hdr(hdr, n_frames);
    if (rc) return rc;
    HIPCHK(hipSetDevice(c->device));
    void *d_pcm = nullptr, *d_hdr = nullptr, *d_mdct = nullptr;
    const size_t b_pcm = (size_t)n_frames * 1152 * 2 * 2, b_hdr = (size_t)n_frames * sizeof(mp3s_frame_hdr),
                 b_mdct = (size_t)n_frames * 2304 * 4;
    if (hipMalloc(&d_pcm, b_pcm) != hipSuccess || hipMalloc(&d_hdr, b_hdr) != hipSuccess ||
        hipMalloc(&d_mdct, b_mdct) != hipSuccess)
        rc = fail(MP3S_E_NOMEM, "hipMalloc failed for a %d-frame batch", n_frames);
    if (!rc) rc = mp3s_dev_upload(c, d_pcm, pcm, b_pcm);
    if (!rc) rc = mp3s_dev_upload(c, d_hdr, hdr, b_hdr);
    if (!rc) rc = mp3s_encode_transform_dev(c, (const int16_t *)d_pcm, (const mp3s_frame_hdr *)d_hdr, n_frames, (int32_t *)d_mdct);
    if (!rc) rc = mp3s_dev_download(c, mdct, d_mdct, b_mdct);
    hipStreamSynchronize(c->stream);
    hipFree(d_pcm); hipFree(d_hdr); hipFree(d_mdct);
    return rc;
}

int mp3s_rate_loop_dev(mp3s_ctx *c, const int32_t *d_mdct, const mp3s_rate_frame *d_frames, int n_frames,
                       const uint8_t *d_hide_bits, int n_hide, const int32_t *d_cursor_in, const int32_t *d_state_in,
                       const int32_t *d_unit_list, int n_list, int16_t *d_ix, mp3s_gr_out *d_out, int32_t *d_en)
{
    if (!c || !d_mdct || !d_frames || !d_ix || !d_out || !d_en) return fail(MP3S_E_ARG, "null pointer");
    if (n_frames <= 0 || n_hide < 0 || (n_hide > 0 && (!d_hide_bits || !d_cursor_in)))
        return fail(MP3S_E_ARG, "bad sizes / missing hide inputs");
    const int e = launch_rate(c->stream, d_mdct, d_frames, n_frames, d_hide_bits, n_hide, d_cursor_in, d_state_in,
                              d_unit_list, n_list, d_ix, d_out, d_en, &c->prof);
    if (e) return fail(MP3S_E_HIP, "rate launch: %s", hipGetErrorString((hipError_t)e));
    return MP3S_OK;
}

static int max_part2_3(const mp3s_frame_side *side, long n)
{
    int m = 0;
    for (long f = 0; f < n; f++)
        for (int k = 0; k < 4; k++) m = std::max<int>(m, side[f].unit[k >> 1][k & 1].part2_3_length);
    return m;
}

int mp3s_huffman_decode_dev(mp3s_ctx *c, const uint8_t *d_blob, const mp3s_frame_side *d_side, int n_frames, int nch,
                            int max_part2_3_length, int16_t *d_is, mp3s_granule_si *d_si, int32_t *d_status)
{
    if (!c || !d_blob || !d_side || !d_is || !d_si || !d_status) return fail(MP3S_E_ARG, "null pointer");
    if (n_frames <= 0 || nch < 1 || nch > 2 || max_part2_3_length < 0) return fail(MP3S_E_ARG, "bad sizes");
    if (max_part2_3_length == 0 || max_part2_3_length > 4095) max_part2_3_length = 4095;
    const int e = launch_huffman(c->stream, d_blob, d_side, n_frames, nch, max_part2_3_length, d_is, d_si, d_status, &c->prof);
    if (e) return fail(MP3S_E_HIP, "huffman launch: %s", hipGetErrorString((hipError_t)e));
    return MP3S_OK;
}

int mp3s_pack_frames_dev(mp3s_ctx *c, const int16_t *d_ix, const mp3s_gr_out *d_gr, const int32_t *d_en, int n_frames,
                         int samplerate, int bitrate_kbps, const uint32_t *d_frame_off, const uint8_t *d_padding,
                         uint8_t *d_mp3, int32_t *d_scfsi, int32_t *d_status)
{
    if (!c || !d_ix || !d_gr || !d_en || !d_frame_off || !d_padding || !d_mp3 || !d_scfsi || !d_status)
        return fail(MP3S_E_ARG, "null pointer");
    int sri, bri, whole;
    if (n_frames <= 0 || stream_params(samplerate, bitrate_kbps, &sri, &bri, &whole))
        return fail(MP3S_E_UNSUPPORTED, "unsupported samplerate/bitrate %d/%d", samplerate, bitrate_kbps);
    const int e = launch_pack(c->stream, d_ix, d_gr, d_en, n_frames, sri, bri, whole, d_frame_off, d_padding, d_mp3, d_scfsi,
                              d_status, c->d_sync, &c->prof);
    if (e) return fail(MP3S_E_HIP, "pack launch: %s", hipGetErrorString((hipError_t)e));
    return MP3S_OK;
}

// ------------------------------------------------------------------------------------------------ host stages
int mp3s_scan_stream(const uint8_t *file, size_t len, mp3s_buf **owner, mp3s_scanned *out)
{
    if (!file || !owner || !out) return fail(MP3S_E_ARG, "null pointer");
    mp3s_buf *b = new mp3s_buf();
    const int rc = parse_stream(file, len, b->parsed, &b->scanned);
    if (rc) { delete b; return fail(rc, "malformed or unsupported MP3 stream"); }
    const ParsedStream &p = b->parsed;
    out->n_frames = p.n_frames; out->nch = p.nch; out->sampling_rate = p.sampling_rate; out->bit_rate = p.bit_rate;
    out->n_bits = (int32_t)p.bits.size(); out->dup_last_frame = p.dup_last_frame;
    out->gpu_ok = b->scanned.gpu_ok ? 1 : 0;
    out->max_part2_3_length = max_part2_3(b->scanned.side.data(), p.n_frames);
    out->side = b->scanned.side.data(); out->hdr = p.hdr.data();
    out->blob = b->scanned.blob.data(); out->blob_len = b->scanned.blob.size();
    out->bits = p.bits.data(); out->frame_size = p.frame_size.data();
    *owner = b;
    return MP3S_OK;
}

int mp3s_parse_stream(const uint8_t *file, size_t len, mp3s_buf **owner, mp3s_parsed *out)
{
    if (!file || !owner || !out) return fail(MP3S_E_ARG, "null pointer");
    mp3s_buf *b = new mp3s_buf();
    const int rc = parse_stream(file, len, b->parsed);
    if (rc) { delete b; return fail(rc, "malformed or unsupported MP3 stream"); }
    const ParsedStream &p = b->parsed;
    out->n_frames = p.n_frames; out->nch = p.nch; out->sampling_rate = p.sampling_rate; out->bit_rate = p.bit_rate;
    out->n_bits = (int32_t)p.bits.size(); out->dup_last_frame = p.dup_last_frame;
    out->is = p.is.data(); out->si = p.si.data(); out->hdr = p.hdr.data(); out->bits = p.bits.data();
    out->table_select = p.table_select.data(); out->frame_size = p.frame_size.data();
    *owner = b;
    return MP3S_OK;
}

int mp3s_rate_frames(int samplerate, int bitrate_kbps, int nch, int n_frames, mp3s_rate_frame *out, int32_t *padding)
{
    if (!out || n_frames < 0) return fail(MP3S_E_ARG, "bad argument");
    const int rc = rate_frames(samplerate, bitrate_kbps, nch, n_frames, out, padding);
    if (rc) return fail(rc, "unsupported samplerate/bitrate %d/%d", samplerate, bitrate_kbps);
    return MP3S_OK;
}

int mp3s_format_stream(int samplerate, int bitrate_kbps, int n_frames, const int16_t *ix, const mp3s_gr_out *gr,
                       const int32_t *scfsi, mp3s_buf **owner, const uint8_t **mp3, size_t *mp3_len)
{
    if (!ix || !gr || !scfsi || !owner || !mp3 || !mp3_len) return fail(MP3S_E_ARG, "null pointer");
    mp3s_buf *b = new mp3s_buf();
    const int rc = format_stream(samplerate, bitrate_kbps, n_frames, ix, gr, scfsi, b->bytes);
    if (rc) { delete b; return fail(rc, "format_stream failed"); }
    *owner = b; *mp3 = b->bytes.data(); *mp3_len = b->bytes.size();
    return MP3S_OK;
}

// ------------------------------------------------------------------------------------------------ pipelines
struct mp3s_multi {     // owner payload of mp3s_decode_streams
    std::vector<std::pair<const uint8_t *, size_t>> files;   // borrowed for the duration of the call
    std::vector<ParsedStream> parsed;
    std::vector<ScannedStream> scanned;
    PinnedBlock arena[3];                 // PCM of all mono / all stereo streams, index = channel count
    size_t head_room = 0;                 // bytes kept free in front of the PCM (mp3s_decode_file puts the WAV header there)
    std::vector<const uint8_t *> pcm;     // per stream, into its arena
    // mp3s_decode_block: only frames [first, first + count) of stream i are kept after parsing (absent: all of them)
    std::vector<std::pair<long, long>> window;
    std::vector<std::vector<uint8_t>> all_bits;   // ... and the stego bits of the whole stream
};

// keep frames [first, first + count) of a parsed stream (its main data, side records, samples)
static void cut_window(ParsedStream &p, ScannedStream &sc, long first, long count)
{
    const long n = p.n_frames;
    first = std::min(std::max(first, 0L), n);
    count = std::min(std::max(count, 0L), n - first);
    auto cut = [&](auto &v, size_t per) {
        if (v.size() >= (size_t)n * per) v.assign(v.begin() + (size_t)first * per, v.begin() + (size_t)(first + count) * per);
    };
    if (!sc.side.empty()) {
        const size_t b0 = first < n ? sc.side[first].md_off : sc.blob.size();
        const size_t b1 = first + count < n ? sc.side[first + count].md_off : sc.blob.size();
        sc.blob.assign(sc.blob.begin() + b0, sc.blob.begin() + std::max(b0, b1));
        cut(sc.side, 1);
        for (auto &fs : sc.side) fs.md_off -= (uint32_t)b0;
    }
    cut(p.is, 2304); cut(p.si, 4); cut(p.hdr, 1); cut(p.table_select, 12); cut(p.frame_size, 1);
    if (first + count < n) p.dup_last_frame = 0;   // the repeated last frame belongs to the block that ends the stream
    p.n_frames = (int)count;
}

// host front end of stream i of m: byte-level scan; scalefactors + Huffman run on the device unless the stream inherits
// scalefactors across frames (mixed blocks ...) or `full` asks for it, in which case the host parser produces its frames
static int front_end(mp3s_multi &m, int i, bool full = false)
{
    ParsedStream &p = m.parsed[i];
    ScannedStream &sc = m.scanned[i];
    int rc = full ? MP3S_OK : parse_stream(m.files[i].first, m.files[i].second, p, &sc);
    if (!rc && (full || !sc.gpu_ok)) {
        rc = parse_stream(m.files[i].first, m.files[i].second, p, nullptr);
        sc.gpu_ok = false;
        sc.side.clear(); sc.blob.clear();   // (possibly cut to a window already; not used for host-parsed streams)
    }
    // no sync where the stream should start: the reference parses nothing and writes an empty WAV (MP3_Parser.py:37-46)
    if (!rc && (size_t)i < m.window.size()) {
        if ((size_t)i < m.all_bits.size()) m.all_bits[i] = p.bits;
        cut_window(p, sc, m.window[i].first, m.window[i].second);
    }
    return rc;
}

// Decode the streams listed in `idx` (all with the same channel count) as ONE batch.
// d_keep != nullptr: the PCM of the group stays on the device there (frames back to back, a duplicated last frame
// included) and nothing is downloaded -- the re-encode path of mp3s_hide_message / mp3s_clear_file.
static int decode_group(mp3s_ctx *c, mp3s_multi &m, const std::vector<int> &idx, int nch, int out_format, void *d_keep = nullptr)
{
    const size_t esz = pcm_elem(out_format), frame_bytes = (size_t)1152 * nch * esz;
    long n = 0;
    for (int i : idx) n += m.parsed[i].n_frames;
    if (n <= 0) return MP3S_OK;
    if (n > 0x7fffffff / 8) return fail(MP3S_E_ARG, "batch of %ld frames is too large", n);
    // ---- concatenate: frame headers (stream_first = first frame of the file), side records, main-data blobs
    std::vector<mp3s_frame_hdr> hdr((size_t)n);
    std::vector<mp3s_frame_side> side((size_t)n);
    std::vector<uint8_t> blob;
    std::vector<long> first_of(idx.size());
    bool any_dev = false, any_host = false;
    long f0 = 0;
    for (size_t k = 0; k < idx.size(); k++) {
        const ParsedStream &p = m.parsed[idx[k]];
        const ScannedStream &sc = m.scanned[idx[k]];
        first_of[k] = f0;
        const bool dev = sc.gpu_ok;
        (dev ? any_dev : any_host) = true;
        const uint32_t base = (uint32_t)blob.size();
        if (dev) blob.insert(blob.end(), sc.blob.begin(), sc.blob.end());
        for (int f = 0; f < p.n_frames; f++) {
            hdr[(size_t)f0 + f] = p.hdr[f];
            hdr[(size_t)f0 + f].stream_first = (uint32_t)f0;
            if (dev) { side[(size_t)f0 + f] = sc.side[f]; side[(size_t)f0 + f].md_off += base; }
            else std::memset(&side[(size_t)f0 + f], 0, sizeof(mp3s_frame_side));   // filled from the host parse below
        }
        f0 += p.n_frames;
    }
    if (blob.empty()) blob.resize(16, 0);
    if (hipSetDevice(c->device) != hipSuccess) return fail(MP3S_E_HIP, "hipSetDevice failed");
    const int chunk = (int)std::min<long>(n, kDecodeChunk) + 1;
    int slot = 0;
    auto grab = [&](size_t bytes) { return c->grab(slot++, bytes); };
    void *d_is = grab((size_t)n * 2304 * 2), *d_si = grab((size_t)n * 4 * sizeof(mp3s_granule_si)),
         *d_hdr = grab((size_t)chunk * sizeof(mp3s_frame_hdr)), *d_pcm = grab((size_t)chunk * frame_bytes), *d_st = grab(16),
         *d_blob = grab(blob.size()), *d_side = grab((size_t)n * sizeof(mp3s_frame_side));
    if (!d_is || !d_si || !d_hdr || !d_pcm || !d_st || !d_blob || !d_side)
        return fail(MP3S_E_NOMEM, "hipMalloc failed for a %ld-frame decode", n);
    int rc = MP3S_OK;
    if (any_dev) {
        rc = mp3s_dev_upload(c, d_blob, blob.data(), blob.size());
        if (!rc) rc = mp3s_dev_upload(c, d_side, side.data(), (size_t)n * sizeof(mp3s_frame_side));
        if (!rc) rc = mp3s_huffman_decode_dev(c, (const uint8_t *)d_blob, (const mp3s_frame_side *)d_side, (int)n, nch, max_part2_3(side.data(), n),
                                              (int16_t *)d_is, (mp3s_granule_si *)d_si, (int32_t *)d_st);
        int32_t st = 0;
        if (!rc) rc = mp3s_dev_download(c, &st, d_st, sizeof st);
        if (!rc && st) {
            // Something in the Huffman data is off (region counts, big_values past 576 lines, big values running past
            // part2_3_length).  The full host parser decides -- it walks the frame with the reference's single bit
            // cursor -- and its frames replace the device's.
            for (size_t k = 0; k < idx.size() && !rc; k++) {
                const int i = idx[k];
                if (!m.scanned[i].gpu_ok) continue;
                const int n_before = m.parsed[i].n_frames;
                rc = front_end(m, i, true);
                if (rc) { rc = fail(rc, "file %d: malformed main data", i); break; }
                if (m.parsed[i].n_frames != n_before) { rc = fail(MP3S_E_MALFORMED, "file %d: inconsistent parse", i); break; }
                any_host = true;
            }
        }
    }
    if (any_host)   // streams that inherit scalefactors across frames were parsed on the host: place their frames
        for (size_t k = 0; k < idx.size() && !rc; k++) {
            const ParsedStream &p = m.parsed[idx[k]];
            if (m.scanned[idx[k]].gpu_ok || !p.n_frames) continue;
            rc = mp3s_dev_upload(c, (int16_t *)d_is + (size_t)first_of[k] * 2304, p.is.data(), (size_t)p.n_frames * 2304 * 2);
            if (!rc) rc = mp3s_dev_upload(c, (mp3s_granule_si *)d_si + (size_t)first_of[k] * 4, p.si.data(),
                                          (size_t)p.n_frames * 4 * sizeof(mp3s_granule_si));
        }
    // ---- transforms in chunks of kDecodeChunk frames; a chunk that starts inside a stream re-runs one halo frame.
    //      Host layout = device layout plus one extra frame after every stream that ends in a bad header (D12): the
    //      reference appends that stream's last PCM frame once more.
    std::vector<long> out_first(idx.size());
    long extra = 0;
    for (size_t k = 0; k < idx.size(); k++) { out_first[k] = first_of[k] + extra; extra += m.parsed[idx[k]].dup_last_frame ? 1 : 0; }
    uint8_t *arena = nullptr;
    if (!d_keep) {
        if (!m.arena[nch].reserve(m.head_room + (size_t)(n + extra) * frame_bytes))
            return fail(MP3S_E_NOMEM, "hipHostMalloc failed for %ld frames of PCM", n + extra);
        arena = m.arena[nch].data() + m.head_room;
    }
    std::vector<mp3s_frame_hdr> hc;
    for (long start = 0; start < n && !rc; start += kDecodeChunk) {
        const int halo = (start && hdr[(size_t)start].stream_first < (uint32_t)start) ? 1 : 0;
        const long first = start - halo;
        const int cnt = (int)std::min<long>(kDecodeChunk, n - start) + halo;
        hc.assign(hdr.begin() + first, hdr.begin() + first + cnt);
        for (auto &h : hc) h.stream_first = h.stream_first > (uint32_t)first ? h.stream_first - (uint32_t)first : 0;
        rc = mp3s_dev_upload(c, d_hdr, hc.data(), (size_t)cnt * sizeof(mp3s_frame_hdr));
        if (!rc) rc = mp3s_decode_transform_dev(c, (const int16_t *)d_is + (size_t)first * 2304, (const mp3s_granule_si *)d_si + (size_t)first * 4,
                                                (const mp3s_frame_hdr *)d_hdr, cnt, nch, halo, out_format, d_pcm);
        // copy out in runs that are contiguous on both sides (a run ends where a duplicated frame is inserted)
        const long end = start + (cnt - halo);
        for (long a = start; a < end && !rc;) {
            size_t k = (size_t)(std::upper_bound(first_of.begin(), first_of.end(), a) - first_of.begin()) - 1;
            long b = end;
            for (size_t j = k; j < idx.size() && first_of[j] < end; j++)
                if (m.parsed[idx[j]].dup_last_frame) { b = std::min<long>(end, first_of[j] + m.parsed[idx[j]].n_frames); break; }
            if (b <= a) b = std::min<long>(end, a + 1);
            const size_t dst = (size_t)(out_first[k] + (a - first_of[k])) * frame_bytes, bytes = (size_t)(b - a) * frame_bytes;
            const uint8_t *src = (const uint8_t *)d_pcm + (size_t)(a - start) * frame_bytes;
            if (d_keep) {
                if (hipMemcpyAsync((uint8_t *)d_keep + dst, src, bytes, hipMemcpyDeviceToDevice, c->stream) != hipSuccess)
                    rc = fail(MP3S_E_HIP, "device copy failed");
            } else rc = mp3s_dev_download(c, arena + dst, src, bytes);
            a = b;
        }
    }
    for (size_t k = 0; k < idx.size() && !rc; k++) {
        const ParsedStream &p = m.parsed[idx[k]];
        if (p.dup_last_frame && p.n_frames > 0) {
            const size_t last = (size_t)(out_first[k] + p.n_frames - 1) * frame_bytes;
            if (d_keep) {
                if (hipMemcpyAsync((uint8_t *)d_keep + last + frame_bytes, (uint8_t *)d_keep + last, frame_bytes, hipMemcpyDeviceToDevice,
                                   c->stream) != hipSuccess)
                    rc = fail(MP3S_E_HIP, "device copy failed");
            } else {
                hipStreamSynchronize(c->stream);
                std::memcpy(arena + last + frame_bytes, arena + last, frame_bytes);
            }
        }
        if (!d_keep) m.pcm[idx[k]] = arena + (size_t)out_first[k] * frame_bytes;
    }
    if (!d_keep) hipStreamSynchronize(c->stream);
    return rc;
}

static int decode_streams_impl(mp3s_ctx *c, const uint8_t *const *files, const size_t *lens, int n_files, int out_format,
                               size_t head_room, mp3s_buf **owner, mp3s_decoded *out)
{
    if (!c || !files || !lens || !owner || !out || n_files <= 0) return fail(MP3S_E_ARG, "bad argument");
    if (out_format < 0 || out_format > 2) return fail(MP3S_E_ARG, "out_format=%d", out_format);
    mp3s_buf *b = new mp3s_buf();
    b->multi.reset(new mp3s_multi());
    mp3s_multi &m = *b->multi;
    m.head_room = head_room;
    m.parsed.resize(n_files); m.scanned.resize(n_files); m.pcm.assign(n_files, nullptr); m.files.resize(n_files);
    std::vector<int> group[3];
    size_t total = 0;
    for (int i = 0; i < n_files; i++) {
        if (!files[i]) { delete b; return fail(MP3S_E_ARG, "file %d is null", i); }
        m.files[i] = {files[i], lens[i]};
        total += lens[i];
    }
    std::vector<int> frc(n_files, MP3S_OK);
    parallel_files(n_files, total, [&](int i) { frc[i] = front_end(m, i); });
    for (int i = 0; i < n_files; i++) {
        if (frc[i]) { delete b; return fail(frc[i], "file %d: malformed or unsupported MP3 stream", i); }
        if (m.parsed[i].n_frames > 0) group[m.parsed[i].nch].push_back(i);
    }
    for (int nch = 1; nch <= 2; nch++)
        if (!group[nch].empty()) {
            const int rc = decode_group(c, m, group[nch], nch, out_format);
            if (rc) { delete b; return rc; }
        }
    for (int i = 0; i < n_files; i++) {
        const ParsedStream &p = m.parsed[i];
        out[i].n_frames = p.n_frames; out[i].nch = p.nch; out[i].sampling_rate = p.sampling_rate; out[i].bit_rate = p.bit_rate;
        out[i].n_bits = (int32_t)p.bits.size(); out[i].n_rows = (int64_t)1152 * (p.n_frames + p.dup_last_frame);
        out[i].pcm = m.pcm[i]; out[i].bits = p.bits.data();
    }
    *owner = b;
    return MP3S_OK;
}

int mp3s_decode_streams(mp3s_ctx *c, const uint8_t *const *files, const size_t *lens, int n_files, int out_format,
                        mp3s_buf **owner, mp3s_decoded *out)
{
    return decode_streams_impl(c, files, lens, n_files, out_format, 0, owner, out);
}

int mp3s_decode_block(mp3s_ctx *c, const uint8_t *file, size_t len, int64_t first_frame, int64_t n_frames, int out_format,
                      mp3s_buf **owner, mp3s_decoded *out)
{
    if (!c || !file || !owner || !out || first_frame < 0 || n_frames <= 0 || first_frame > 0x7fffffff || n_frames > 0x7fffffff)
        return fail(MP3S_E_ARG, "bad argument");
    if (out_format < 0 || out_format > 2) return fail(MP3S_E_ARG, "out_format=%d", out_format);
    // one frame in front of the block is decoded for its state and dropped: the IMDCT overlap and the synthesis fifo
    // reach back less than a frame (Frame.py:151-153, 81-92)
    const int halo = first_frame > 0 ? 1 : 0;
    std::unique_ptr<mp3s_buf> b(new mp3s_buf());
    b->multi.reset(new mp3s_multi());
    mp3s_multi &m = *b->multi;
    m.parsed.resize(1); m.scanned.resize(1); m.pcm.assign(1, nullptr); m.files.assign(1, {file, len});
    m.window.assign(1, {(long)first_frame - halo, (long)n_frames + halo});
    m.all_bits.resize(1);
    int rc = front_end(m, 0);
    if (rc) return fail(rc, "malformed or unsupported MP3 stream");
    const ParsedStream &p = m.parsed[0];
    if (p.n_frames <= halo) return fail(MP3S_E_ARG, "the block starts behind the last frame of the stream");
    if (p.nch < 1 || p.nch > 2) return fail(MP3S_E_MALFORMED, "channel count");
    rc = decode_group(c, m, std::vector<int>{0}, p.nch, out_format);
    if (rc) return rc;
    m.files.clear();   // borrowed
    const size_t frame_bytes = (size_t)1152 * p.nch * pcm_elem(out_format);
    out->n_frames = p.n_frames - halo; out->nch = p.nch; out->sampling_rate = p.sampling_rate; out->bit_rate = p.bit_rate;
    out->n_rows = (int64_t)1152 * (p.n_frames - halo + p.dup_last_frame);
    out->pcm = m.pcm[0] + (size_t)halo * frame_bytes;
    out->n_bits = (int32_t)m.all_bits[0].size(); out->bits = m.all_bits[0].data();
    *owner = b.release();
    return MP3S_OK;
}

int mp3s_decode_stream(mp3s_ctx *c, const uint8_t *file, size_t len, int out_format, mp3s_buf **owner, mp3s_decoded *out)
{
    if (!file) return fail(MP3S_E_ARG, "null pointer");
    return mp3s_decode_streams(c, &file, &len, 1, out_format, owner, out);
}

constexpr int kLongMessageBits = 1024;    // above: the first pass does not guess cursors at all
constexpr int32_t kNoCursor = 0x3fffffff; // "behind every message": such a unit hides nothing
constexpr int kPatternBytes = 32;         // the eight 3-bit patterns, 4 bytes apart, in front of the messages
constexpr int kVariantEntries = 65536;    // (unit, pattern) entries per variant launch
constexpr size_t kFewUnits = 8;           // that few wrong cursors after the first pass: re-run them directly

struct EncSeg {             // one stream of an encode batch: frames back to back in the batch's PCM
    int n_frames = 0;
    const uint8_t *hide = nullptr;   // 0/1 bytes
    int n_hide = 0;
    // a block of a longer stream (mp3s_encode_block; only as the single stream of a batch)
    int lead = 0;                    // frames of PCM in front of the block: transformed for their state, then dropped
    int64_t first_frame = 0;         // index of the block's first frame in its stream (padding recurrence)
    bool last = true;                // the stream ends with this block (the reference drops the cached tail there: E14)
    const mp3s_carry *carry_in = nullptr;
    // filled by encode_batch
    int first = 0, hide_base = 0;
    int64_t hide_offset = 0;         // message bits consumed (from the start of the stream)
    size_t mp3_off = 0, mp3_len = 0; // the stream's bytes inside the batch's output
    mp3s_carry carry_out = {};
    bool carry_used = false;         // the block's bytes depend on carry_in
};

// Encode the streams of `segs` (stereo, one sampling rate and bitrate) as ONE batch: transforms, rate loop, bit packing.
// pcm_dev != nullptr: the int16 PCM is already in HBM (re-encode after a device decode) and pcm is ignored.
// Results: b->mp3 = the streams' MP3 bytes (segs[i].mp3_off / mp3_len), b->gr_out and b->scfsi in batch frame order.
static int encode_batch(mp3s_ctx *c, const int16_t *pcm, const int16_t *pcm_dev, std::vector<EncSeg> &segs, int samplerate,
                        int bitrate_kbps, mp3s_buf *b, int *passes_out)
{
    // -1 sits in the reference's bitrate table (encoder/util.py:27,42), so its header check lets it through and the
    // encoder then runs on negative slot counts; nothing meaningful to reproduce
    if (bitrate_kbps <= 0) return fail(MP3S_E_UNSUPPORTED, "bitrate %d", bitrate_kbps);
    int sri = 0, bri = 0, whole = 0;
    if (stream_params(samplerate, bitrate_kbps, &sri, &bri, &whole))
        return fail(MP3S_E_UNSUPPORTED, "unsupported samplerate/bitrate %d/%d", samplerate, bitrate_kbps);
    int64_t n64 = 0, hide64 = kPatternBytes;
    for (auto &s : segs) {
        if (s.n_frames <= 0 || s.n_hide < 0 || (s.n_hide > 0 && !s.hide)) return fail(MP3S_E_ARG, "bad stream in the encode batch");
        s.first = (int)n64; s.hide_base = (int)hide64;
        n64 += s.n_frames; hide64 += s.n_hide;
        if (n64 > 0x7fffffff / 8 || hide64 >= kNoCursor - 8) return fail(MP3S_E_ARG, "encode batch too large");
    }
    if (n64 <= 0) return fail(MP3S_E_ARG, "empty encode batch");
    const int lead = segs[0].lead;   // frames that only the transforms see
    if (lead < 0 || lead > 2 || (segs.size() > 1 && (lead || segs[0].first_frame || segs[0].carry_in)))
        return fail(MP3S_E_ARG, "a block of a longer stream is encoded on its own");
    const int n = (int)n64, units = n * 4, n_hide = (int)hide64, n_all = n + lead;
    std::vector<mp3s_rate_frame> rf(n);
    std::vector<mp3s_frame_hdr> hdr(n_all);
    std::vector<int32_t> padding(n);
    std::vector<uint8_t> hide_all((size_t)n_hide, 0);   // [patterns | message of stream 0 | message of stream 1 ...]
    for (int v = 0; v < 8; v++) { hide_all[4 * v] = (v >> 2) & 1; hide_all[4 * v + 1] = (v >> 1) & 1; hide_all[4 * v + 2] = v & 1; }
    int64_t bytes_before = 0;
    for (const auto &s : segs) {
        // padding / slot lag restart with every stream (MP3_Encoder.py:623-636)
        const int rc = rate_frames(samplerate, bitrate_kbps, 2, s.n_frames, rf.data() + s.first, padding.data() + s.first, s.first_frame,
                                   &bytes_before);
        if (rc) return fail(rc, "unsupported samplerate/bitrate %d/%d", samplerate, bitrate_kbps);
        for (int f = s.first; f < s.first + s.n_frames; f++) rf[f].hide_end = s.hide_base + s.n_hide;
        for (int f = s.first; f < s.first + s.n_frames + lead; f++) {
            hdr[f].sr_idx = (uint8_t)sri; hdr[f].nch = 2; hdr[f].ms_stereo = 0; hdr[f].flags = 0; hdr[f].stream_first = (uint32_t)s.first;
        }
        if (s.n_hide) std::memcpy(hide_all.data() + s.hide_base, s.hide, (size_t)s.n_hide);
    }
    HIPCHK(hipSetDevice(c->device));

    // every host-made input in one block, one copy: [frame headers | rate frames | cursors | message bits]
    auto up16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    const size_t o_rf = up16((size_t)n_all * sizeof(mp3s_frame_hdr)), o_cur = o_rf + up16((size_t)n * sizeof(mp3s_rate_frame)),
                 o_hide = o_cur + up16((size_t)units * 4), in_bytes = o_hide + up16((size_t)n_hide);
    void *d_pcm = nullptr, *d_in = nullptr, *d_mdct_all = nullptr, *d_state = nullptr, *d_list = nullptr, *d_redo = nullptr, *d_ix = nullptr,
         *d_out = nullptr, *d_en = nullptr;
    auto cleanup = [&]() { hipStreamSynchronize(c->stream); };   // the buffers stay in the context's pool
    int slot = 8;
    auto alloc = [&](void **p, size_t bytes) { *p = c->grab(slot++, bytes); return *p != nullptr; };
    if (pcm_dev) { d_pcm = const_cast<int16_t *>(pcm_dev); slot++; }
    else if (!alloc(&d_pcm, (size_t)n_all * 2304 * 2)) d_pcm = nullptr;
    if (!d_pcm || !alloc(&d_in, in_bytes) || !alloc(&d_mdct_all, (size_t)n_all * 2304 * 4) || !alloc(&d_state, (size_t)units * 16) ||
        !alloc(&d_list, (size_t)units * 4) || !alloc(&d_redo, (size_t)units * 24) || !alloc(&d_ix, (size_t)n * 2304 * 2) ||
        !alloc(&d_out, (size_t)units * sizeof(mp3s_gr_out)) || !alloc(&d_en, (size_t)units * 22 * 4)) {
        cleanup();
        return fail(MP3S_E_NOMEM, "hipMalloc failed for a %d-frame encode", n);
    }
    const mp3s_frame_hdr *d_hdr = (const mp3s_frame_hdr *)d_in;
    const int32_t *d_mdct = (const int32_t *)d_mdct_all + (size_t)lead * 2304;   // the block's own frames
    const mp3s_rate_frame *d_rf = (const mp3s_rate_frame *)((uint8_t *)d_in + o_rf);
    const int32_t *d_cur = (const int32_t *)((uint8_t *)d_in + o_cur);
    const uint8_t *d_hide = (const uint8_t *)d_in + o_hide;
    std::vector<int32_t> &cursor = c->h_cursor, &state = c->h_state;
    cursor.assign(units, 0); state.assign((size_t)units * 4, 0);
    if (!b->big[1].reserve((size_t)units * sizeof(mp3s_gr_out))) { cleanup(); return fail(MP3S_E_NOMEM, "host memory for %d units", units); }
    mp3s_gr_out *const gr = b->gr_out = (mp3s_gr_out *)b->big[1].data();   // filled by the first pass's download
    // A unit sees the message only through the <= 3 bits at its cursor.  First pass: guess three tables per unit (for a
    // short message in a long stream that is almost always right: one launch).  Where the guess fails, everything behind
    // the first wrong unit shifts, and re-running shifts it again (hidden bits change bit counts, these the quantiser step,
    // that the number of tables): pass by pass this converges one unit in fifteen at a time.  Instead the units the rest
    // of the message can reach are run once per 3-bit pattern (entries = unit x pattern, one launch), the cursor walk
    // names the entry each unit really sees, and only the message's last unit and the granules that inherit state are
    // left for exact re-runs.
    auto cursor0 = [](const EncSeg &s) {   // message bits the frames in front of a block have taken
        return s.carry_in ? std::min<int64_t>(std::max<int64_t>(s.carry_in->cursor, 0), kNoCursor) : 0;
    };
    for (const auto &s : segs) {
        const int64_t c0 = cursor0(s);
        const bool long_msg = s.n_hide - c0 > kLongMessageBits;
        for (int j = 0; j < s.n_frames * 4; j++)
            cursor[(size_t)s.first * 4 + j] = long_msg ? kNoCursor : (int32_t)std::min<int64_t>((int64_t)s.hide_base + c0 + 3 * (int64_t)j, kNoCursor);
    }
    std::vector<uint8_t> &in = c->h_in;
    in.assign(in_bytes, 0);
    std::memcpy(in.data(), hdr.data(), (size_t)n_all * sizeof(mp3s_frame_hdr));
    std::memcpy(in.data() + o_rf, rf.data(), (size_t)n * sizeof(mp3s_rate_frame));
    std::memcpy(in.data() + o_cur, cursor.data(), (size_t)units * 4);
    if (n_hide) std::memcpy(in.data() + o_hide, hide_all.data(), (size_t)n_hide);
    int rc = MP3S_OK;
    if (!pcm_dev && hipMemcpyAsync(d_pcm, pcm, (size_t)n_all * 2304 * 2, hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = fail(MP3S_E_HIP, "PCM upload failed");
    if (!rc && hipMemsetAsync(d_state, 0, (size_t)units * 16, c->stream) != hipSuccess) rc = fail(MP3S_E_HIP, "memset failed");
    if (!rc) rc = mp3s_dev_upload(c, d_in, in.data(), in_bytes);
    if (!rc) rc = mp3s_encode_transform_dev(c, (const int16_t *)d_pcm, d_hdr, n_all, (int32_t *)d_mdct_all);
    if (!rc) rc = mp3s_rate_loop_dev(c, d_mdct, d_rf, n, d_hide, n_hide, d_cur, (const int32_t *)d_state, nullptr, 0,
                                     (int16_t *)d_ix, (mp3s_gr_out *)d_out, (int32_t *)d_en);
    if (!rc) rc = mp3s_dev_download(c, gr, d_out, (size_t)units * sizeof(mp3s_gr_out));
    int passes = 1;
    // ---- resolve the serial chains, stream by stream: hide cursor (MP3_Encoder.py:808-809) and the per-(gr,ch) inherited
    //      address1/2/3 + quantizerStepSize (E7).  walk() lists the units whose assumed inputs were wrong and, per stream,
    //      where the cursor first went wrong.
    std::vector<int32_t> list, redo_in;
    std::vector<mp3s_gr_out> tmp;
    struct Pending { int unit; int64_t cur; };   // first unit of a stream that ran on a wrong cursor, the right cursor there
    std::vector<Pending> pend(segs.size());
    // cursor / state: what each unit's current result was computed with; want / state_want: what the walk says it should be
    std::vector<int32_t> &want = c->h_want, &state_want = c->h_state_want;
    want.assign(units, 0); state_want.assign((size_t)units * 4, 0);
    auto walk = [&]() {
        list.clear();
        for (size_t si = 0; si < segs.size(); si++) {
            EncSeg &s = segs[si];
            pend[si] = {-1, 0};
            int64_t cur = s.hide_base + cursor0(s);
            const int64_t end = (int64_t)s.hide_base + s.n_hide;
            int32_t chain[4][4] = {};   // [(ch*2+gr)][a1,a2,a3,step]
            if (s.carry_in) std::memcpy(chain, s.carry_in->chain, sizeof chain);
            bool own[4] = {false, false, false, false};   // the block has set chain[k] itself
            s.carry_used = cur < end;                     // the message is still being hidden when the block starts
            for (int u = s.first * 4; u < (s.first + s.n_frames) * 4; u++) {
                const int k = u & 3;
                mp3s_gr_out &g = gr[u];
                bool redo = false;
                const bool active = g.flags & MP3S_RF_ACTIVE;
                if (!own[k] && (!active || (g.flags & MP3S_RF_USED_ADDR_IN))) s.carry_used = true;
                if (active) own[k] = true;
                if (s.n_hide > 0 && active) {
                    const int64_t used = cursor[u];
                    if (used != cur && std::min<int64_t>(used, cur) < end) {
                        redo = true;
                        if (pend[si].unit < 0) pend[si] = {u, cur};
                    }
                }
                if ((g.flags & MP3S_RF_USED_ADDR_IN) &&
                    (state[(size_t)u * 4] != chain[k][0] || state[(size_t)u * 4 + 1] != chain[k][1] ||
                     state[(size_t)u * 4 + 2] != chain[k][2]))
                    redo = true;
                if (redo) list.push_back(u);
                want[u] = (int32_t)std::min<int64_t>(cur, kNoCursor);
                for (int j = 0; j < 4; j++) state_want[(size_t)u * 4 + j] = chain[k][j];
                if (active) {
                    cur += g.n_tables;
                    chain[k][0] = g.address[0]; chain[k][1] = g.address[1]; chain[k][2] = g.address[2];
                    chain[k][3] = g.quantizer_step;
                } else {   // silent unit: everything is inherited (quantizerStepSize and addresses pass through)
                    g.address[0] = chain[k][0]; g.address[1] = chain[k][1]; g.address[2] = chain[k][2];
                    g.quantizer_step = chain[k][3];
                }
                if (g.flags & MP3S_RF_STEP_RANGE) return fail(MP3S_E_STEP_RANGE, "quantizer step left the table in unit %d", u);
            }
            s.hide_offset = cur - s.hide_base;
            s.carry_out.cursor = s.hide_offset;
            std::memcpy(s.carry_out.chain, chain, sizeof chain);
        }
        return (int)MP3S_OK;
    };
    // one launch over a list of (unit, cursor, inherited state) entries: 24 bytes per entry up -- [units | cursors | states]
    // -- and the entries' GrInfo down (into tmp); compact = 1: ix / energies in place, 2: by entry into d_ixv / d_env
    auto run_entries = [&](const std::vector<int32_t> &units_of, const std::vector<int32_t> &cursor_of, void *d_entries, int compact,
                           int16_t *ix_to, mp3s_gr_out *out_to, int32_t *en_to) {
        const size_t nl = units_of.size();
        redo_in.resize(nl * 6);
        for (size_t i = 0; i < nl; i++) {
            const int u = units_of[i];
            redo_in[i] = u;
            redo_in[nl + i] = cursor_of[i];
            for (int j = 0; j < 4; j++) redo_in[2 * nl + 4 * i + j] = state_want[(size_t)u * 4 + j];
        }
        int r = mp3s_dev_upload(c, d_entries, redo_in.data(), nl * 24);
        if (!r) {
            const int32_t *dr = (const int32_t *)d_entries;
            const int e = launch_rate(c->stream, d_mdct, d_rf, n, d_hide, n_hide, dr + nl, dr + 2 * nl, dr, (int)nl, ix_to, out_to, en_to,
                                      &c->prof, 0, compact);
            if (e) r = fail(MP3S_E_HIP, "rate launch: %s", hipGetErrorString((hipError_t)e));
        }
        tmp.resize(nl);
        if (!r) r = mp3s_dev_download(c, tmp.data(), out_to, nl * sizeof(mp3s_gr_out));
        passes++;
        return r;
    };
    if (!rc) rc = walk();
    if (!rc && list.size() > kFewUnits) {
        // ---- message variants
        struct Span { size_t seg; int unit, count; int64_t cur; size_t entry; };
        std::vector<Span> spans;
        std::vector<int32_t> ent_unit, ent_cursor, pairs;
        void *d_ent = nullptr, *d_ixv = nullptr, *d_outv = nullptr, *d_env = nullptr, *d_pairs = nullptr;
        const int slot_var = slot;
        for (bool more = true; more && !rc;) {
            spans.clear(); ent_unit.clear(); ent_cursor.clear(); pairs.clear();
            for (size_t si = 0; si < segs.size(); si++) {
                if (pend[si].unit < 0) continue;
                const EncSeg &s = segs[si];
                const int64_t end = (int64_t)s.hide_base + s.n_hide;
                if (pend[si].cur + 3 > end) { pend[si].unit = -1; continue; }   // only the message's last unit is left
                // as many units as the rest of the message can reach at two tables per unit (silent units take none: the
                // stream simply comes round again), as many as still fit into this launch
                const int room = (kVariantEntries - (int)ent_unit.size()) / 8;
                const int count = (int)std::min<int64_t>({(int64_t)(s.first + s.n_frames) * 4 - pend[si].unit, (end - pend[si].cur) / 2 + 16, (int64_t)room});
                if (count <= 0) continue;                                       // next launch
                spans.push_back({si, pend[si].unit, count, pend[si].cur, ent_unit.size()});
                for (int v = 0; v < 8; v++)
                    for (int j = 0; j < count; j++) { ent_unit.push_back(pend[si].unit + j); ent_cursor.push_back(4 * v); }
            }
            if (spans.empty()) break;
            const size_t ne = ent_unit.size();
            slot = slot_var;
            if (!alloc(&d_ent, ne * 24) || !alloc(&d_ixv, ne * 1152) || !alloc(&d_outv, ne * sizeof(mp3s_gr_out)) ||
                !alloc(&d_env, ne * 88) || !alloc(&d_pairs, ne))   // at most one pair per unit = ne / 8 pairs of 8 bytes
                rc = fail(MP3S_E_NOMEM, "hipMalloc failed for the message variants");
            if (!rc) rc = run_entries(ent_unit, ent_cursor, d_ent, 2, (int16_t *)d_ixv, (mp3s_gr_out *)d_outv, (int32_t *)d_env);
            more = false;
            for (const Span &sp : spans) {
                if (rc) break;
                const EncSeg &s = segs[sp.seg];
                const int64_t end = (int64_t)s.hide_base + s.n_hide;
                int64_t cur = sp.cur;
                int j = 0;
                for (; j < sp.count && cur + 3 <= end; j++) {     // all three bits the unit can ask for exist
                    const int u = sp.unit + j;
                    const int v = (hide_all[cur] & 1) * 4 + (hide_all[cur + 1] & 1) * 2 + (hide_all[cur + 2] & 1);
                    const size_t e = sp.entry + (size_t)v * sp.count + j;
                    gr[u] = tmp[e];
                    cursor[u] = (int32_t)cur;
                    for (int q = 0; q < 4; q++) state[(size_t)u * 4 + q] = state_want[(size_t)u * 4 + q];
                    pairs.push_back((int32_t)e); pairs.push_back(u);
                    cur += gr[u].n_tables;
                }
                // span used up with message left: the stream goes on in the next launch; otherwise the message's last unit
                // (fewer than three bits left) and whatever lies behind it are the exact re-run's
                if (j == sp.count && cur < end && sp.unit + sp.count < (s.first + s.n_frames) * 4) { pend[sp.seg] = {sp.unit + sp.count, cur}; more = true; }
                else pend[sp.seg].unit = -1;
            }
            for (size_t si = 0; si < segs.size(); si++) more |= pend[si].unit >= 0;
            if (!rc && !pairs.empty()) {
                rc = mp3s_dev_upload(c, d_pairs, pairs.data(), pairs.size() * 4);
                if (!rc) {
                    const int e = launch_scatter(c->stream, (const int32_t *)d_pairs, (int)(pairs.size() / 2), (const int16_t *)d_ixv,
                                                 (const int32_t *)d_env, (int16_t *)d_ix, (int32_t *)d_en);
                    if (e) rc = fail(MP3S_E_HIP, "scatter: %s", hipGetErrorString((hipError_t)e));
                }
                if (!rc && hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(MP3S_E_HIP, "sync failed");   // the entry arrays are reused
            }
        }
        if (!rc) rc = walk();
    }
    // ---- exact re-runs of what is left, until nothing changes
    std::vector<int32_t> cur_of;
    while (!rc && !list.empty()) {
        if (passes > units + 16) { rc = fail(MP3S_E_HIP, "rate-loop chain did not converge"); break; }
        cur_of.resize(list.size());
        for (size_t i = 0; i < list.size(); i++) cur_of[i] = want[list[i]];
        const std::vector<int32_t> units_of = list;
        rc = run_entries(units_of, cur_of, d_redo, 1, (int16_t *)d_ix, (mp3s_gr_out *)d_out, (int32_t *)d_en);
        if (!rc)
            for (size_t i = 0; i < units_of.size(); i++) {
                const int u = units_of[i];
                gr[u] = tmp[i];
                cursor[u] = cur_of[i];
                for (int q = 0; q < 4; q++) state[(size_t)u * 4 + q] = state_want[(size_t)u * 4 + q];
            }
        if (!rc) rc = walk();
    }
    if (!rc) {
        // ---- bit packing on the device: final GrInfo + frame offsets up, MP3 bytes + scfsi down
        std::vector<uint32_t> off((size_t)n + 1, 0);
        std::vector<uint8_t> pad8(n);
        for (int f = 0; f < n; f++) { pad8[f] = (uint8_t)padding[f]; off[f + 1] = off[f] + (uint32_t)(whole + padding[f]); }
        void *d_off = nullptr, *d_pad = nullptr, *d_mp3 = nullptr, *d_sc = nullptr, *d_st = nullptr;
        if (!alloc(&d_off, ((size_t)n + 1) * 4) || !alloc(&d_pad, (size_t)n) || !alloc(&d_mp3, (size_t)off[n] + 16) ||
            !alloc(&d_sc, (size_t)n * 8 * 4) || !alloc(&d_st, 16))
            rc = fail(MP3S_E_NOMEM, "hipMalloc failed for the packer");
        if (!rc) rc = mp3s_dev_upload(c, d_off, off.data(), ((size_t)n + 1) * 4);
        if (!rc) rc = mp3s_dev_upload(c, d_pad, pad8.data(), (size_t)n);
        if (!rc) rc = mp3s_dev_upload(c, d_out, gr, (size_t)units * sizeof(mp3s_gr_out));
        if (!rc) rc = mp3s_pack_frames_dev(c, (const int16_t *)d_ix, (const mp3s_gr_out *)d_out, (const int32_t *)d_en, n, samplerate,
                                           bitrate_kbps, (const uint32_t *)d_off, (const uint8_t *)d_pad, (uint8_t *)d_mp3,
                                           (int32_t *)d_sc, (int32_t *)d_st);
        int32_t st = 0;
        if (!rc) rc = mp3s_dev_download(c, &st, d_st, sizeof st);
        if (!rc && st) rc = fail(MP3S_E_HIP, "bit packer reported status %d", st);
        for (auto &s : segs) {
            s.mp3_off = off[s.first];
            s.mp3_len = off[s.first + s.n_frames] - off[s.first];
            // the reference drops the cached tail (< 32 bits) at the end of the stream: E14
            if (s.last) s.mp3_len -= std::min<size_t>(s.mp3_len, (size_t)((bytes_before + (int64_t)s.mp3_len) % 4));
        }
        const size_t total = segs.back().mp3_off + segs.back().mp3_len;
        if (!rc && !b->big[0].reserve(total)) rc = fail(MP3S_E_NOMEM, "host memory for %zu bytes of MP3", total);
        b->mp3 = b->big[0].data();
        b->scfsi.assign((size_t)n * 8, 0);
        if (!rc) rc = mp3s_dev_download(c, b->mp3, d_mp3, total);
        if (!rc) rc = mp3s_dev_download(c, b->scfsi.data(), d_sc, (size_t)n * 8 * 4);
    }
    cleanup();
    if (passes_out) *passes_out = passes;
    return rc;
}

static int encode_core(mp3s_ctx *c, const int16_t *pcm, const int16_t *pcm_dev, int64_t n_samples_per_ch, int nch, int samplerate,
                       int bitrate_kbps, const uint8_t *hide_bits, int n_hide, mp3s_buf **owner, mp3s_encoded *out)
{
    if (nch != 2) return fail(MP3S_E_UNSUPPORTED, "mono encode raises IndexError in the reference (SURVEY E3)");
    if (n_samples_per_ch <= 0 || n_samples_per_ch % 1152)
        return fail(MP3S_E_UNSUPPORTED, "sample count %lld is not a multiple of 1152 (reference over-reads, E3)",
                    (long long)n_samples_per_ch);
    if (n_hide < 0 || (n_hide > 0 && !hide_bits)) return fail(MP3S_E_ARG, "bad hide arguments");
    if (n_samples_per_ch / 1152 > 0x7fffffff / 8) return fail(MP3S_E_ARG, "too many frames");
    std::vector<EncSeg> segs(1);
    segs[0].n_frames = (int)(n_samples_per_ch / 1152); segs[0].hide = hide_bits; segs[0].n_hide = n_hide;
    std::unique_ptr<mp3s_buf> b(new mp3s_buf());
    int passes = 0;
    const int rc = encode_batch(c, pcm, pcm_dev, segs, samplerate, bitrate_kbps, b.get(), &passes);
    if (rc) return rc;
    out->n_frames = segs[0].n_frames;
    out->hide_offset = segs[0].hide_offset;
    out->too_long = out->hide_offset < (int64_t)n_hide - 1 ? 1 : 0;
    out->mp3 = b->mp3; out->mp3_len = segs[0].mp3_len;
    out->gr = b->gr_out; out->scfsi = b->scfsi.data();
    out->rate_passes = passes;
    *owner = b.release();
    return MP3S_OK;
}

int mp3s_encode_block(mp3s_ctx *c, const int16_t *pcm, int64_t n_samples_per_ch, int lead_frames, int64_t first_frame, int last_block,
                      int samplerate, int bitrate_kbps, const uint8_t *hide_bits, int n_hide, const mp3s_carry *carry_in,
                      mp3s_carry *carry_out, int32_t *carry_used, mp3s_buf **owner, mp3s_encoded *out)
{
    if (!c || !pcm || !owner || !out) return fail(MP3S_E_ARG, "null pointer");
    if (lead_frames < 0 || lead_frames > 2 || first_frame < 0 || (first_frame == 0 && (lead_frames || carry_in)) ||
        (first_frame > 0 && !carry_in))
        return fail(MP3S_E_ARG, "block arguments: the first block has no lead and no carry, later blocks have a carry");
    if (n_samples_per_ch % 1152 || n_samples_per_ch / 1152 <= lead_frames)
        return fail(MP3S_E_UNSUPPORTED, "sample count %lld is not a multiple of 1152 / holds no frame of its own", (long long)n_samples_per_ch);
    if (n_hide < 0 || (n_hide > 0 && !hide_bits)) return fail(MP3S_E_ARG, "bad hide arguments");
    if (n_samples_per_ch / 1152 > 0x7fffffff / 8) return fail(MP3S_E_ARG, "too many frames");
    std::vector<EncSeg> segs(1);
    EncSeg &s = segs[0];
    s.n_frames = (int)(n_samples_per_ch / 1152) - lead_frames; s.hide = hide_bits; s.n_hide = n_hide;
    s.lead = lead_frames; s.first_frame = first_frame; s.last = last_block != 0; s.carry_in = carry_in;
    std::unique_ptr<mp3s_buf> b(new mp3s_buf());
    int passes = 0;
    const int rc = encode_batch(c, pcm, nullptr, segs, samplerate, bitrate_kbps, b.get(), &passes);
    if (rc) return rc;
    if (carry_out) *carry_out = s.carry_out;
    if (carry_used) *carry_used = s.carry_used ? 1 : 0;
    out->n_frames = s.n_frames;
    out->hide_offset = s.hide_offset;
    out->too_long = out->hide_offset < (int64_t)n_hide - 1 ? 1 : 0;
    out->mp3 = b->mp3; out->mp3_len = s.mp3_len;
    out->gr = b->gr_out; out->scfsi = b->scfsi.data();
    out->rate_passes = passes;
    *owner = b.release();
    return MP3S_OK;
}

int mp3s_encode_pcm(mp3s_ctx *c, const int16_t *pcm, int64_t n_samples_per_ch, int nch, int samplerate, int bitrate_kbps,
                    const uint8_t *hide_bits, int n_hide, mp3s_buf **owner, mp3s_encoded *out)
{
    if (!c || !pcm || !owner || !out) return fail(MP3S_E_ARG, "null pointer");
    return encode_core(c, pcm, nullptr, n_samples_per_ch, nch, samplerate, bitrate_kbps, hide_bits, n_hide, owner, out);
}

/* ---------------------------------------------------------------- (vi) files and messages */
int mp3s_wav_parse(const uint8_t *file, size_t len, int bitrate_kbps, mp3s_wav_info *out)
{
    if ((!file && len) || !out) return fail(MP3S_E_ARG, "null pointer");
    const char *msg = "";
    const int rc = wav_parse(file, len, bitrate_kbps, out, &msg);
    return rc ? fail(rc, "%s", msg) : MP3S_OK;
}

int mp3s_wav_header(int64_t n_rows, int nch, int rate, uint8_t *out44)
{
    if (!out44 || n_rows < 0 || nch < 1 || nch > 2 || rate < 0) return fail(MP3S_E_ARG, "bad argument");
    wav_header(n_rows, nch, rate, out44);
    return MP3S_OK;
}

int mp3s_message_frame(const uint8_t *utf8, size_t n, mp3s_buf **owner, const uint8_t **bits, size_t *n_bits)
{
    if ((!utf8 && n) || !owner || !bits || !n_bits) return fail(MP3S_E_ARG, "null pointer");
    mp3s_buf *b = new mp3s_buf();
    message_frame(utf8, n, b->bits);
    *bits = b->bits.data(); *n_bits = b->bits.size(); *owner = b;
    return MP3S_OK;
}

int mp3s_message_reveal(const uint8_t *bits, size_t n_bits, mp3s_buf **owner, const uint8_t **text, size_t *n_text)
{
    if ((!bits && n_bits) || !owner || !text || !n_text) return fail(MP3S_E_ARG, "null pointer");
    mp3s_buf *b = new mp3s_buf();
    message_reveal(bits, n_bits, b->bytes);
    *text = b->bytes.data(); *n_text = b->bytes.size(); *owner = b;
    return MP3S_OK;
}

int mp3s_decode_file(mp3s_ctx *c, const uint8_t *mp3, size_t len, mp3s_buf **owner, mp3s_file *out)
{
    if (!c || !mp3 || !owner || !out) return fail(MP3S_E_ARG, "null pointer");
    // the PCM lands 64 bytes into its buffer; the 44-byte WAV header goes right in front of it: no second copy
    mp3s_buf *b = nullptr;
    mp3s_decoded d;
    const int rc = decode_streams_impl(c, &mp3, &len, 1, MP3S_PCM_I16, 64, &b, &d);
    if (rc) return rc;
    uint8_t *wav;
    if (d.n_rows == 0) {   // nothing decoded: what scipy writes for an empty 1-d array at the header object's initial rate 0
        b->bytes.assign(44, 0);
        wav = b->bytes.data();
        wav_header(0, 1, d.sampling_rate, wav);
    } else {
        wav = const_cast<uint8_t *>(static_cast<const uint8_t *>(d.pcm)) - 44;
        wav_header(d.n_rows, d.nch, d.sampling_rate, wav);
    }
    std::memset(out, 0, sizeof *out);
    out->data = wav; out->len = 44 + (size_t)d.n_rows * (size_t)d.nch * 2;
    out->kbps = d.bit_rate / 1000; out->sampling_rate = d.sampling_rate; out->channels = d.nch; out->n_frames = d.n_frames;
    out->n_bits = d.n_bits; out->bits = d.bits;
    *owner = b;
    return MP3S_OK;
}

static void file_from_encoded(const mp3s_encoded &e, int kbps, int rate, mp3s_file *out)
{
    std::memset(out, 0, sizeof *out);
    out->data = e.mp3; out->len = e.mp3_len; out->kbps = kbps; out->sampling_rate = rate; out->channels = 2;
    out->n_frames = e.n_frames; out->too_long = e.too_long; out->hide_offset = e.hide_offset;
}

int mp3s_encode_file(mp3s_ctx *c, const uint8_t *wav, size_t len, int bitrate_kbps, const uint8_t *hide_bits, int n_hide,
                     mp3s_buf **owner, mp3s_file *out)
{
    if (!c || !wav || !owner || !out) return fail(MP3S_E_ARG, "null pointer");
    mp3s_wav_info w;
    int rc = mp3s_wav_parse(wav, len, bitrate_kbps, &w);
    if (rc) return rc;
    // MP3_Encoder.py:596-618 walks num_of_samples * channels values in steps of 1152 * channels and indexes the buffer
    // as if it were stereo: mono input and a partial last frame both end in IndexError there
    if (w.channels != 2) return fail(MP3S_E_UNSUPPORTED, "mono input: the reference encoder indexes the sample buffer out of bounds");
    const int64_t total = w.num_of_samples * 2, count = total / 2304;
    if (total % 2304 || w.n_values < count * 2304)
        return fail(MP3S_E_UNSUPPORTED, "sample count is not a multiple of 1152 per channel: the reference encoder reads past the end of the sample buffer");
    std::vector<int16_t> pcm((size_t)count * 2304);   // the data chunk may sit at an odd offset
    std::memcpy(pcm.data(), wav + w.data_offset, pcm.size() * 2);
    mp3s_encoded e;
    rc = encode_core(c, pcm.data(), nullptr, count * 1152, 2, w.samplerate, bitrate_kbps, hide_bits, n_hide, owner, &e);
    if (!rc) file_from_encoded(e, bitrate_kbps, w.samplerate, out);
    return rc;
}

// what the reference's WAV reader / encoder would say to the WAV its decoder writes for this stream
static int reencode_check(const ParsedStream &p, int *kbps_out)
{
    const int kbps = p.bit_rate / 1000;
    int sri, bri, whole;
    // the WAV the reference writes carries the last header's sampling rate; its reader checks the rate, then the bitrate
    if (p.sampling_rate != 32000 && p.sampling_rate != 44100 && p.sampling_rate != 48000)
        return fail(MP3S_E_EXIT, "Unsupported sampling frequency.");
    if (stream_params(p.sampling_rate, kbps, &sri, &bri, &whole)) return fail(MP3S_E_EXIT, "Unsupported bitrate configuration.");
    if (p.nch != 2) return fail(MP3S_E_UNSUPPORTED, "mono input: the reference encoder indexes the sample buffer out of bounds");
    if (p.n_frames <= 0) return fail(MP3S_E_UNSUPPORTED, "no frame in the stream");
    *kbps_out = kbps;
    return MP3S_OK;
}

// Decode the streams `idx` of m (stereo, one sampling rate and bitrate) on the device into HBM and encode them from
// there as one batch: steganography.py:133-182 without the temporary WAV.  bits[i] = framed message of file i (empty:
// nothing hidden).  The batch's bytes are kept in a new part of `top`; out[i] points into it.
static int reencode_group(mp3s_ctx *c, mp3s_multi &m, const std::vector<int> &idx, const std::vector<std::vector<uint8_t>> &bits,
                          int samplerate, int kbps, mp3s_buf *top, mp3s_file *out)
{
    std::vector<EncSeg> segs(idx.size());
    int64_t rows_frames = 0;
    for (size_t k = 0; k < idx.size(); k++) {
        const ParsedStream &p = m.parsed[idx[k]];
        segs[k].n_frames = p.n_frames + (p.dup_last_frame ? 1 : 0);
        if (bits[idx[k]].size() > 0x7fffffff) return fail(MP3S_E_ARG, "message too long");
        segs[k].hide = bits[idx[k]].data(); segs[k].n_hide = (int)bits[idx[k]].size();
        rows_frames += segs[k].n_frames;
    }
    if (hipSetDevice(c->device) != hipSuccess) return fail(MP3S_E_HIP, "hipSetDevice failed");
    void *d_keep = c->grab(7, (size_t)rows_frames * 2304 * 2);
    if (!d_keep) return fail(MP3S_E_NOMEM, "hipMalloc failed for %lld frames of PCM", (long long)rows_frames);
    const double t0 = trace_on() ? now_ms() : 0;
    int rc = decode_group(c, m, idx, 2, MP3S_PCM_I16, d_keep);
    if (rc) return rc;
    if (trace_on()) hipStreamSynchronize(c->stream);
    const double t1 = trace_on() ? now_ms() : 0;
    std::unique_ptr<mp3s_buf> part(new mp3s_buf());
    int passes = 0;
    rc = encode_batch(c, nullptr, (const int16_t *)d_keep, segs, samplerate, kbps, part.get(), &passes);
    if (rc) return rc;
    if (trace_on())
        fprintf(stderr, "mp3s:   %zu stream(s), %lld frames: decode %.3f ms, encode %.3f ms (%d rate passes)\n", idx.size(),
                (long long)rows_frames, t1 - t0, now_ms() - t1, passes);
    for (size_t k = 0; k < idx.size(); k++) {
        mp3s_file &o = out[idx[k]];
        std::memset(&o, 0, sizeof o);
        o.data = part->mp3 + segs[k].mp3_off; o.len = segs[k].mp3_len;
        o.kbps = kbps; o.sampling_rate = samplerate; o.channels = 2; o.n_frames = segs[k].n_frames;
        o.hide_offset = segs[k].hide_offset;
        o.too_long = segs[k].hide_offset < (int64_t)segs[k].n_hide - 1 ? 1 : 0;
    }
    top->parts.push_back(std::move(part));
    return MP3S_OK;
}

int mp3s_hide_messages(mp3s_ctx *c, const uint8_t *const *mp3s, const size_t *lens, int n_files, const uint8_t *const *msgs,
                       const size_t *msg_lens, mp3s_buf **owner, mp3s_file *out, int32_t *status)
{
    if (!c || !mp3s || !lens || !owner || !out || n_files <= 0 || (msgs && !msg_lens)) return fail(MP3S_E_ARG, "bad argument");
    std::unique_ptr<mp3s_buf> top(new mp3s_buf());
    top->multi.reset(new mp3s_multi());
    mp3s_multi &m = *top->multi;
    m.parsed.resize(n_files); m.scanned.resize(n_files); m.pcm.assign(n_files, nullptr); m.files.resize(n_files);
    std::vector<std::vector<uint8_t>> bits(n_files);
    std::vector<int32_t> st(n_files, MP3S_OK);
    struct Group { int rate, kbps; std::vector<int> idx; };
    std::vector<Group> groups;
    size_t total = 0;
    const double t0 = trace_on() ? now_ms() : 0;
    for (int i = 0; i < n_files; i++) {
        std::memset(&out[i], 0, sizeof out[i]);
        if (!mp3s[i] || (msgs && msgs[i] == nullptr && msg_lens[i])) { st[i] = MP3S_E_ARG; continue; }
        m.files[i] = {mp3s[i], lens[i]};
        total += lens[i];
    }
    parallel_files(n_files, total, [&](int i) { if (!st[i]) st[i] = front_end(m, i); });
    const double t1 = trace_on() ? now_ms() : 0;
    for (int i = 0; i < n_files; i++) {
        int kbps = 0;
        if (st[i]) { fail(st[i], st[i] == MP3S_E_ARG ? "file %d: null pointer" : "file %d: malformed or unsupported MP3 stream", i); continue; }
        st[i] = reencode_check(m.parsed[i], &kbps);
        if (st[i]) continue;
        if (msgs && msgs[i]) message_frame(msgs[i], msg_lens[i], bits[i]);
        const int rate = m.parsed[i].sampling_rate;
        size_t g = 0;
        while (g < groups.size() && (groups[g].rate != rate || groups[g].kbps != kbps)) g++;
        if (g == groups.size()) groups.push_back({rate, kbps, {}});
        groups[g].idx.push_back(i);
    }
    const double t2 = trace_on() ? now_ms() : 0;
    for (const Group &g : groups) {
        const int rc = reencode_group(c, m, g.idx, bits, g.rate, g.kbps, top.get(), out);
        if (!rc) continue;
        // one stream spoils its batch (main data the host parser rejects ...): each file on its own, to name it
        for (int i : g.idx) st[i] = g.idx.size() == 1 ? rc : reencode_group(c, m, std::vector<int>{i}, bits, g.rate, g.kbps, top.get(), out);
    }
    if (trace_on())
        fprintf(stderr, "mp3s: hide_messages, %d file(s): scan %.3f ms, messages + grouping %.3f ms, device batches %.3f ms\n", n_files,
                t1 - t0, t2 - t1, now_ms() - t2);
    m.files.clear();   // borrowed pointers
    int first_bad = MP3S_OK;
    for (int i = 0; i < n_files; i++) {
        if (status) status[i] = st[i];
        if (st[i] && !first_bad) first_bad = st[i];
    }
    if (!status && first_bad) return first_bad;
    *owner = top.release();
    return MP3S_OK;
}

int mp3s_reencode_block(mp3s_ctx *c, const uint8_t *mp3, size_t len, const uint8_t *utf8, size_t n_msg, int rank, int world,
                        const mp3s_carry *carry_in, mp3s_buf **owner, mp3s_block *out)
{
    if (!c || !mp3 || !owner || !out || world <= 0 || rank < 0 || rank >= world || (rank == 0 && carry_in) || (rank > 0 && !carry_in))
        return fail(MP3S_E_ARG, "bad argument (rank 0 has no carry, every other rank has one)");
    std::unique_ptr<mp3s_buf> top(new mp3s_buf());
    top->multi.reset(new mp3s_multi());
    mp3s_multi &m = *top->multi;
    m.parsed.resize(1); m.scanned.resize(1); m.pcm.assign(1, nullptr); m.files.assign(1, {mp3, len});
    int rc = front_end(m, 0);
    if (rc) return fail(rc, "malformed or unsupported MP3 stream");
    ParsedStream &p = m.parsed[0];
    int kbps = 0;
    rc = reencode_check(p, &kbps);
    if (rc) return rc;
    const int samplerate = p.sampling_rate;
    std::vector<uint8_t> bits;
    if (utf8) message_frame(utf8, n_msg, bits);
    if (bits.size() > 0x7fffffff) return fail(MP3S_E_ARG, "message too long");
    // blocks of PCM frames; the frame the decoder repeats after a bad header (D12) is the stream's last PCM frame
    const long n = p.n_frames, total = n + (p.dup_last_frame ? 1 : 0);
    const long base = total / world, rem = total % world;
    const long first = rank * base + std::min<long>(rank, rem), count = base + (rank < rem ? 1 : 0);
    std::memset(out, 0, sizeof *out);
    out->total_frames = total; out->first_frame = first; out->n_frames = count; out->is_last = first + count == total;
    out->file.kbps = kbps; out->file.sampling_rate = samplerate; out->file.channels = 2;
    if (count == 0) { m.files.clear(); *owner = top.release(); return MP3S_OK; }
    const int lead = first > 0 ? 1 : 0;                   // PCM in front of the block, for the encoder's filter state
    const int halo = first - lead > 0 ? 1 : 0;            // a frame in front of that, for the decoder's
    const long w0 = first - lead - halo, w1 = std::min(first + count, n);
    const bool with_dup = first + count == total && p.dup_last_frame;
    cut_window(p, m.scanned[0], w0, w1 - w0);
    p.dup_last_frame = with_dup ? 1 : 0;
    const int64_t rows_frames = (w1 - w0) + (with_dup ? 1 : 0);
    if (hipSetDevice(c->device) != hipSuccess) return fail(MP3S_E_HIP, "hipSetDevice failed");
    void *d_keep = c->grab(7, (size_t)rows_frames * 2304 * 2);
    if (!d_keep) return fail(MP3S_E_NOMEM, "hipMalloc failed for %lld frames of PCM", (long long)rows_frames);
    rc = decode_group(c, m, std::vector<int>{0}, 2, MP3S_PCM_I16, d_keep);
    if (rc) return rc;
    std::vector<EncSeg> segs(1);
    EncSeg &s = segs[0];
    s.n_frames = (int)count; s.hide = bits.data(); s.n_hide = (int)bits.size();
    s.lead = lead; s.first_frame = first; s.last = out->is_last != 0; s.carry_in = carry_in;
    std::unique_ptr<mp3s_buf> part(new mp3s_buf());
    rc = encode_batch(c, nullptr, (const int16_t *)d_keep + (size_t)halo * 2304, segs, samplerate, kbps, part.get(), nullptr);
    if (rc) return rc;
    out->carry_out = s.carry_out; out->carry_used = s.carry_used ? 1 : 0;
    out->file.data = part->mp3 + s.mp3_off; out->file.len = s.mp3_len; out->file.n_frames = (int32_t)count;
    out->file.hide_offset = s.hide_offset;
    out->file.too_long = s.hide_offset < (int64_t)s.n_hide - 1 ? 1 : 0;
    top->parts.push_back(std::move(part));
    m.files.clear();
    *owner = top.release();
    return MP3S_OK;
}

static int reencode(mp3s_ctx *c, const uint8_t *mp3, size_t len, const uint8_t *msg, size_t n_msg, bool hide, mp3s_buf **owner, mp3s_file *out)
{
    const uint8_t *const no_msg = nullptr;
    return mp3s_hide_messages(c, &mp3, &len, 1, hide ? &msg : &no_msg, &n_msg, owner, out, nullptr);
}

int mp3s_hide_message(mp3s_ctx *c, const uint8_t *mp3, size_t len, const uint8_t *utf8, size_t n_msg, mp3s_buf **owner, mp3s_file *out)
{
    if (!c || !mp3 || (!utf8 && n_msg) || !owner || !out) return fail(MP3S_E_ARG, "null pointer");
    static const uint8_t empty = 0;
    return reencode(c, mp3, len, utf8 ? utf8 : &empty, n_msg, true, owner, out);
}

int mp3s_clear_file(mp3s_ctx *c, const uint8_t *mp3, size_t len, mp3s_buf **owner, mp3s_file *out)
{
    if (!c || !mp3 || !owner || !out) return fail(MP3S_E_ARG, "null pointer");
    return reencode(c, mp3, len, nullptr, 0, false, owner, out);
}

int mp3s_reveal_message(const uint8_t *mp3, size_t len, mp3s_buf **owner, mp3s_file *out)
{
    if (!mp3 || !owner || !out) return fail(MP3S_E_ARG, "null pointer");
    mp3s_buf *b = new mp3s_buf();
    const int rc = parse_stream(mp3, len, b->parsed, &b->scanned);
    if (rc) { delete b; return fail(rc, "malformed or unsupported MP3 stream"); }
    message_reveal(b->parsed.bits.data(), b->parsed.bits.size(), b->bytes);
    std::memset(out, 0, sizeof *out);
    out->data = b->bytes.data(); out->len = b->bytes.size();
    out->kbps = b->parsed.bit_rate / 1000; out->sampling_rate = b->parsed.sampling_rate; out->channels = b->parsed.nch;
    out->n_frames = b->parsed.n_frames; out->n_bits = (int32_t)b->parsed.bits.size(); out->bits = b->parsed.bits.data();
    *owner = b;
    return MP3S_OK;
}

}  // extern "C"
