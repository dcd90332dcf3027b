// C-ABI of the library (include/mp3s.h), part 1: context, device memory, timers, the resident-batch entry points and the
// host stages.  The stream / file pipelines live in mp3s_decode_pipeline.cpp and mp3s_encode_pipeline.cpp.  No CPU
// fallback exists for the transforms: every path that needs them goes through launch_* in mp3s_device.hip.
#include "mp3s_internal.h"

namespace {
thread_local std::string g_err;
}

int mp3s::fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

int max_part2_3(const mp3s_frame_side *side, long n)
{
    int m = 0;
    for (long f = 0; f < n; f++)
        for (int k = 0; k < 4; k++) m = std::max<int>(m, side[f].unit[k >> 1][k & 1].part2_3_length);
    return m;
}

void select_entries(int first_unit, int reach, int first_entry, int hide_end, int64_t bits_left, int32_t *ent_unit, int32_t *ent_cursor)
{
    const int tail = (int)MP3S_SELECT_TAIL_FIRST(bits_left, (int64_t)reach);
    int at = first_entry;
    for (int v = 0; v < MP3S_SELECT_VARIANTS; v++) {
        // 0..7: the three-bit patterns in front of the messages; 8 / 9: the message's own last two bits / last bit, which
        // only the units from `tail` on can meet
        const int32_t cur = v < 8 ? 4 * v : hide_end - (10 - v);
        for (int j = v < 8 ? 0 : tail; j < reach; j++) {
            ent_unit[at] = first_unit + j;
            ent_cursor[at++] = cur;
        }
    }
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

int mp3s_debug_parse_scanned_frame(const void *frame_side, const uint8_t *blob, int16_t *is2304, mp3s_granule_si *si4)
{
    if (!frame_side || !blob || !is2304 || !si4) return fail(MP3S_E_ARG, "null pointer");
    const int rc = parse_scanned_frame(*static_cast<const mp3s_frame_side *>(frame_side), blob, is2304, si4);
    return rc ? fail(rc, "malformed main data") : MP3S_OK;
}

int mp3s_debug_walk_rate(const uint8_t *file, size_t len, double seconds, double *frames_per_s, int64_t *frames_per_pass)
{
    if (!file || !frames_per_s || seconds <= 0) return fail(MP3S_E_ARG, "bad argument");
    std::vector<FrameRef> refs(len / 24 + 16);
    std::vector<uint8_t> tables(refs.size() * 4);
    const double t0 = now_ms();
    int64_t frames = 0, per_pass = 0;
    double t = t0;
    do {
        FrameWalker w;
        if (w.open(file, len)) return fail(MP3S_E_MALFORMED, "malformed or unsupported MP3 stream");
        w.tables_wanted = 1000;
        long n = 0;
        while (!w.ended && !w.irregular && (size_t)n < refs.size()) n += w.next(refs.data() + n, (long)refs.size() - n, tables.data() + (size_t)n * 4, 0, 0);
        if (w.irregular) return fail(MP3S_E_UNSUPPORTED, "not a stream the walk takes");
        frames += n; per_pass = n;
        t = now_ms();
    } while (t - t0 < seconds * 1e3);
    *frames_per_s = (double)frames / ((t - t0) * 1e-3);
    if (frames_per_pass) *frames_per_pass = per_pass;
    return MP3S_OK;
}

int mp3s_device_count(int *n)
{
    if (!n) return fail(MP3S_E_ARG, "n is null");
    int k = 0;
    *n = hipGetDeviceCount(&k) == hipSuccess && k > 0 ? k : 0;
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
    {
        // the environment provides the defaults, once
        auto env = [](const char *name, int64_t dflt) { const char *v = getenv(name); return v && *v ? (int64_t)atoll(v) : dflt; };
        c->opt[MP3S_OPT_SELECT] = getenv("MP3S_NO_SELECT") ? 0 : 1;
        c->opt[MP3S_OPT_REDO] = getenv("MP3S_NO_REDO") ? 0 : 1;
        c->opt[MP3S_OPT_FAST_IMDCT] = env("MP3S_FAST_IMDCT", 1) != 0;
        c->opt[MP3S_OPT_PIPE_TAIL] = std::min<int64_t>(2, std::max<int64_t>(0, env("MP3S_PIPE_TAIL", 1)));
        c->opt[MP3S_OPT_CHUNK_FRAMES] = std::max<int64_t>(0, env("MP3S_CHUNK_FRAMES", 0));
        c->opt[MP3S_OPT_DEVICE_PARSE] = env("MP3S_DEVICE_PARSE", 1) != 0;
        c->opt[MP3S_OPT_FILE_PIPELINE] = env("MP3S_FILE_PIPELINE", 1) != 0;
        c->opt[MP3S_OPT_SCAN_THREADS] = std::max<int64_t>(0, env("MP3S_SCAN_THREADS", 0));
        c->opt[MP3S_OPT_FIRST_CHUNK_FRAMES] = std::max<int64_t>(0, env("MP3S_FIRST_CHUNK_FRAMES", 0));
        c->opt[MP3S_OPT_FILE_UP] = getenv("MP3S_NO_FILE_UP") ? 0 : 1;
        c->opt[MP3S_OPT_HUF_LANES] = std::max<int64_t>(0, env("MP3S_HUF_LANES", 0));
        c->opt[MP3S_OPT_NUMA] = getenv("MP3S_NO_NUMA") ? 0 : 1;
        c->opt[MP3S_OPT_FLOAT_FAST] = env("MP3S_FLOAT_FAST", 0) != 0;
        c->opt[MP3S_OPT_FAIL_CHUNK] = 0;
        c->opt[MP3S_OPT_FUSED_DECODE] = env("MP3S_FUSED_DECODE", 1) != 0;
        c->opt[MP3S_OPT_FUSED_ENCODE] = env("MP3S_FUSED_ENCODE", 0) != 0;
        c->opt[MP3S_OPT_PIPE_DEC] = env("MP3S_PIPE_DEC", 0) != 0;
        c->opt[MP3S_OPT_RATE_SIGNALS] = env("MP3S_RATE_SIGNALS", 0) != 0;
        c->opt[MP3S_OPT_PIPE_SIGNALS] = env("MP3S_PIPE_SIGNALS", 0) & 3;
    }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_order, hipEventDisableTiming) != hipSuccess) {
        mp3s_ctx_destroy(c);   // releases whatever was created
        return fail(MP3S_E_NO_DEVICE, "stream/event creation failed");
    }
    // self-clearing words of kernels in flight: [0..1] bit packer, [2] fast synthesis counter, [4..5] Huffman kernel, [6..7] fix-up list of the fast int16 decode
    if (hipMalloc((void **)&c->d_sync, 32) != hipSuccess || hipMemsetAsync(c->d_sync, 0, 32, c->stream) != hipSuccess) {
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
    destroy_own_pipe(c);
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->scratch) hipFree(c->scratch);
    if (c->scratch_enc) hipFree(c->scratch_enc);
    if (c->d_sync) hipFree(c->d_sync);
    for (void *q : c->pool) if (q) hipFree(q);
    if (c->ev0) hipEventDestroy(c->ev0);
    if (c->ev1) hipEventDestroy(c->ev1);
    if (c->ev_order) hipEventDestroy(c->ev_order);
    if (c->ev_sel) hipEventDestroy(c->ev_sel);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

int mp3s_ctx_host_share(mp3s_ctx *c, mp3s_host_share *out)
{
    if (!out) return fail(MP3S_E_ARG, "null pointer");
    size_t held = 0, cap = 0;
    PinnedBlock::pool_state(&held, &cap);
    out->pinned_pooled_bytes = held; out->pinned_pool_cap_bytes = cap;
    out->local_world_size = local_world_size();
    out->cpus_allowed = host_cpus_allowed();
    out->gpu_node_cpus = c ? (int32_t)gpu_node_cpus(c->device).size() : 0;   // (no context: the host's side alone -- what a rank would get, asked without a GPU)
    out->reserved = 0;
    return MP3S_OK;
}

int mp3s_ctx_set_option(mp3s_ctx *c, int option, int64_t value)
{
    if (!c || option <= 0 || option >= MP3S_OPT_COUNT) return fail(MP3S_E_ARG, "unknown option %d", option);
    if (value < 0 || ((option == MP3S_OPT_CHUNK_FRAMES || option == MP3S_OPT_FIRST_CHUNK_FRAMES) && value != 0 && value < 4) || (option == MP3S_OPT_SCAN_THREADS && value > 64))
        return fail(MP3S_E_ARG, "option %d: value %lld out of range", option, (long long)value);
    const bool number = option == MP3S_OPT_CHUNK_FRAMES || option == MP3S_OPT_SCAN_THREADS || option == MP3S_OPT_FIRST_CHUNK_FRAMES ||
                        option == MP3S_OPT_HUF_LANES || option == MP3S_OPT_FAIL_CHUNK || option == MP3S_OPT_PIPE_SIGNALS;
    c->opt[option] = number ? value : (option == MP3S_OPT_PIPE_TAIL ? std::min<int64_t>(value, 2) : (value != 0));
    return MP3S_OK;
}

int mp3s_ctx_run_stats(mp3s_ctx *c, mp3s_run_stats *out)
{
    if (!c || !out) return fail(MP3S_E_ARG, "null pointer");
    *out = c->run_stats;
    own_pipe_lanes(c, out);
    return MP3S_OK;
}

int mp3s_ctx_get_option(mp3s_ctx *c, int option, int64_t *value)
{
    if (!c || !value || option <= 0 || option >= MP3S_OPT_COUNT) return fail(MP3S_E_ARG, "unknown option %d", option);
    *value = c->opt[option];
    return MP3S_OK;
}

int mp3s_device_name(mp3s_ctx *c, char *buf, size_t n)
{
    if (!c || !buf || !n) return fail(MP3S_E_ARG, "bad argument");
    hipDeviceProp_t p;
    HIPCHK(hipGetDeviceProperties(&p, c->device));
    snprintf(buf, n, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
    return MP3S_OK;
}

int mp3s_device_pci(mp3s_ctx *c, char *buf, size_t n)
{
    if (!c || !buf || n < 16) return fail(MP3S_E_ARG, "bad argument");
    HIPCHK(hipDeviceGetPCIBusId(buf, (int)n, c->device));
    for (char *p = buf; *p; p++) *p = (char)tolower(*p);
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

int mp3s_ctx_wait_last(mp3s_ctx *c, mp3s_ctx *other)
{
    if (!c || !other) return fail(MP3S_E_ARG, "ctx is null");
    if (c == other) return MP3S_OK;
    if (c->device != other->device) return fail(MP3S_E_ARG, "contexts on different devices");
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
int mp3s_dev_copy(mp3s_ctx *c, void *d_dst, const void *d_src, size_t bytes)
{
    if (!c || !d_dst || !d_src) return fail(MP3S_E_ARG, "bad argument");
    if (bytes) HIPCHK(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, c->stream));
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

int mp3s_synth_mode(mp3s_ctx *c, double eps_scale, int64_t *exact_samples)
{
    if (!c || eps_scale < 0) return fail(MP3S_E_ARG, "bad argument");
    HIPCHK(hipStreamSynchronize(c->stream));
    int32_t n = 0;
    HIPCHK(hipMemcpy(&n, c->d_sync + 2, 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemset(c->d_sync + 2, 0, 4));
    if (exact_samples) *exact_samples = n;
    c->synth_eps_scale = eps_scale;
    return MP3S_OK;
}

int mp3s_bench_copy(mp3s_ctx *c, size_t bytes, int iters, double *gb_per_s)
{
    if (!c || !gb_per_s || bytes < 4096 || iters < 1) return fail(MP3S_E_ARG, "bad argument");
    HIPCHK(hipSetDevice(c->device));
    bytes &= ~(size_t)15;
    void *a = nullptr, *b = nullptr;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) { if (a) hipFree(a); return fail(MP3S_E_NOMEM, "hipMalloc(%zu) x 2", bytes); }
    int rc = MP3S_OK;
    float ms = 0;
    if (hipMemsetAsync(a, 1, bytes, c->stream) != hipSuccess || launch_copy(c->stream, a, b, bytes) != 0) rc = fail(MP3S_E_HIP, "copy launch failed");   // warm
    if (!rc && hipEventRecord(c->ev0, c->stream) != hipSuccess) rc = fail(MP3S_E_HIP, "event");
    for (int i = 0; i < iters && !rc; i++) if (launch_copy(c->stream, i & 1 ? b : a, i & 1 ? a : b, bytes) != 0) rc = fail(MP3S_E_HIP, "copy launch failed");
    if (!rc && (hipEventRecord(c->ev1, c->stream) != hipSuccess || hipEventSynchronize(c->ev1) != hipSuccess ||
                hipEventElapsedTime(&ms, c->ev0, c->ev1) != hipSuccess)) rc = fail(MP3S_E_HIP, "event");
    hipStreamSynchronize(c->stream);
    hipFree(a); hipFree(b);
    if (!rc) *gb_per_s = 2.0 * (double)bytes * iters / (ms * 1e-3) / 1e9;   // read + write
    return rc;
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
    if (p.dropped) {
        // (a table that silently under-counts a kernel is worse than none: collect more often -- MAX_PAIRS launches fit between two calls)
        const long lost = p.dropped;
        p.dropped = 0;
        return fail(MP3S_E_ARG, "%ld timed launches had no event pair left (collect at least every %d launches)", lost, Profiler::MAX_PAIRS);
    }
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
    if (int pe = guard_probe_usable(c, out_format)) return pe;
    const int e = launch_decode(c->stream, d_is, d_si, d_hdr, n_frames, nch, n_halo, out_format, d_pcm, c->scratch, &c->prof, 0,
                                c->synth_eps_scale, c->d_sync, c->opt[MP3S_OPT_FAST_IMDCT] != 0, c->opt[MP3S_OPT_FLOAT_FAST] != 0,
                                c->opt[MP3S_OPT_FUSED_DECODE] != 0, nullptr, [&]() -> const GuardProbe * {
                                    if (!c->guard_probe.x) return nullptr;
                                    c->guard_probe.base = 0;
                                    return &c->guard_probe; }());
    if (e) return fail(MP3S_E_HIP, "decode launch: %s", hipGetErrorString((hipError_t)e));
    return MP3S_OK;
}

int mp3s_debug_guard_margin(mp3s_ctx *c, double *d_x, double *d_eps, int64_t capacity)
{
    if (!c) return fail(MP3S_E_ARG, "null context");
    if ((d_x == nullptr) != (d_eps == nullptr) || (d_x && capacity <= 0)) return fail(MP3S_E_ARG, "both arrays and their capacity, or neither");
    HIPCHK(hipStreamSynchronize(c->stream));
    c->guard_probe = GuardProbe{d_x, d_eps, 0, d_x ? capacity : 0};
    return MP3S_OK;
}



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
    // k_enc_analysis computes a product once where the filter table repeats itself (analysis_plan.h, generated from the table's values): a
    // host whose libm rounds the table differently gets no wrong subband samples -- the encode entry points refuse, decoding is not affected
    if (!host_tables().analysis_plan_ok)
        return fail(MP3S_E_TABLES, "the analysis filter table built on this host does not repeat itself where the kernel shares products "
                                   "(csrc/analysis_plan.h: run tools/gen_analysis_plan.py here and rebuild)");
    const bool fused = c->opt[MP3S_OPT_FUSED_ENCODE] != 0;
    int rc = fused ? MP3S_OK : c->ensure_scratch_enc(enc_scratch_bytes(n_frames));   // (the fused kernel keeps the subband samples in LDS)
    if (rc) return rc;
    const int e = launch_encode(c->stream, d_pcm, d_hdr, n_frames, d_mdct, c->scratch_enc, &c->prof, fused);
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
                              d_unit_list, n_list, d_ix, d_out, d_en, &c->prof, 0, 0, nullptr,
                              c->rate_done ? c->rate_done : (c->opt[MP3S_OPT_RATE_SIGNALS] ? c->ev_order : nullptr));
    if (e) return fail(MP3S_E_HIP, "rate launch: %s", hipGetErrorString((hipError_t)e));
    return MP3S_OK;
}



void mp3s_select_patterns(uint8_t out[32])
{
    for (int v = 0; v < 8; v++) { out[4 * v] = (v >> 2) & 1; out[4 * v + 1] = (v >> 1) & 1; out[4 * v + 2] = v & 1; out[4 * v + 3] = 0; }
}

int mp3s_select_plan(const mp3s_chain_seg *segs, int n_segs, mp3s_select_span *spans, int32_t *ent_unit, int32_t *ent_cursor, int cap)
{
    return select_plan(segs, n_segs, spans, ent_unit, ent_cursor, cap, nullptr);
}

}  // extern "C"

int select_plan(const mp3s_chain_seg *segs, int n_segs, mp3s_select_span *spans, int32_t *ent_unit, int32_t *ent_cursor, int cap,
                const int32_t *min_reach)
{
    if (!segs || !spans || n_segs <= 0 || cap < 0 || ((ent_unit == nullptr) != (ent_cursor == nullptr))) return fail(MP3S_E_ARG, "bad argument");
    int used = 0;
    for (int i = 0; i < n_segs; i++) {
        const mp3s_chain_seg &g = segs[i];
        spans[i].first_entry = used; spans[i].reach = 0;
        const int64_t left = (int64_t)g.hide_end - g.hide_begin;
        if (left <= 0 || g.n_frames <= 0 || g.hide_base < 32) continue;
        // units the rest of the message can reach at 2.8 tables per unit (measured on music: 2.96), and some
        // ... or as far as the caller expects it to get (a stream that starts in silence offers no tables for a while)
        const int64_t reach = std::min<int64_t>((int64_t)g.n_frames * 4, std::max<int64_t>(left * 5 / 14 + 32, min_reach ? min_reach[i] : 0));
        if (reach > MP3S_SELECT_MAX_REACH || (int64_t)used + MP3S_SELECT_ENTRIES(left, reach) > cap) continue;
        spans[i].reach = (int32_t)reach;
        if (ent_unit) select_entries(g.first_frame * 4, (int)reach, used, g.hide_end, left, ent_unit, ent_cursor);
        used += (int)MP3S_SELECT_ENTRIES(left, reach);
    }
    return used;
}

extern "C" {

int mp3s_rate_variants_dev(mp3s_ctx *c, const int32_t *d_mdct, const mp3s_rate_frame *d_frames, int n_frames, const uint8_t *d_hide_bits,
                           int n_hide, const int32_t *d_cursor, const int32_t *d_ent_unit, const int32_t *d_ent_cursor, int n_entries,
                           int16_t *d_ix, mp3s_gr_out *d_out, int32_t *d_en, int16_t *d_ixv, mp3s_gr_out *d_outv, int32_t *d_env)
{
    if (!c || !d_mdct || !d_frames || !d_ix || !d_out || !d_en || !d_hide_bits || !d_cursor || !d_ent_unit || !d_ent_cursor || !d_ixv ||
        !d_outv || !d_env)
        return fail(MP3S_E_ARG, "null pointer");
    if (n_frames <= 0 || n_hide < 32 || n_entries <= 0) return fail(MP3S_E_ARG, "bad sizes");
    // (a pipe runs the selection of the job in front on its tail stream: that one reads the variant buffers this launch writes)
    if (c->sel_pending && hipStreamWaitEvent(c->stream, c->ev_sel, 0) != hipSuccess) return fail(MP3S_E_HIP, "ordering behind the previous selection failed");
    c->sel_pending = false;
    const RateVariantArgs va = {d_ent_unit, d_ent_cursor, n_entries, d_ixv, d_outv, d_env, (uint8_t *)(d_outv + n_entries)};
    const int e = launch_rate(c->stream, d_mdct, d_frames, n_frames, d_hide_bits, n_hide, d_cursor, nullptr, nullptr, 0, d_ix, d_out, d_en,
                              &c->prof, 0, 0, &va, c->rate_done ? c->rate_done : (c->opt[MP3S_OPT_RATE_SIGNALS] ? c->ev_order : nullptr));
    if (e) return fail(MP3S_E_HIP, "rate launch: %s", hipGetErrorString((hipError_t)e));
    return MP3S_OK;
}

int mp3s_select_dev(mp3s_ctx *c, const uint8_t *d_hide_bits, int32_t *d_cursor, const mp3s_chain_seg *d_segs, const mp3s_select_span *d_spans,
                    int n_segs, int max_reach, const int32_t *d_ent_unit, const int32_t *d_ent_cursor, int n_entries, int16_t *d_ix,
                    mp3s_gr_out *d_out, int32_t *d_en, int16_t *d_ixv, mp3s_gr_out *d_outv, int32_t *d_env)
{
    if (!c || !d_ix || !d_out || !d_en || !d_hide_bits || !d_cursor || !d_segs || !d_spans || !d_ent_unit || !d_ent_cursor || !d_ixv ||
        !d_outv || !d_env)
        return fail(MP3S_E_ARG, "null pointer");
    if (n_segs <= 0 || n_entries <= 0 || max_reach <= 0 || max_reach > MP3S_SELECT_MAX_REACH) return fail(MP3S_E_ARG, "bad sizes");
    const RateVariantArgs va = {d_ent_unit, d_ent_cursor, n_entries, d_ixv, d_outv, d_env, (uint8_t *)(d_outv + n_entries)};
    void *d_pairs = c->grab(29, (size_t)n_segs * max_reach * 8);
    if (!d_pairs) return fail(MP3S_E_NOMEM, "hipMalloc failed for the selection scratch");
    const int e = launch_select(c->stream, d_segs, d_spans, n_segs, max_reach, d_hide_bits, va, d_ix, d_en, d_out, d_cursor, d_pairs, &c->prof);
    if (e) return fail(MP3S_E_HIP, "select launch: %s", hipGetErrorString((hipError_t)e));
    return MP3S_OK;
}

int mp3s_rate_select_dev(mp3s_ctx *c, const int32_t *d_mdct, const mp3s_rate_frame *d_frames, int n_frames, const uint8_t *d_hide_bits,
                         int n_hide, int32_t *d_cursor, const mp3s_chain_seg *d_segs, const mp3s_select_span *d_spans, int n_segs,
                         int max_reach, const int32_t *d_ent_unit, const int32_t *d_ent_cursor, int n_entries, int16_t *d_ix,
                         mp3s_gr_out *d_out, int32_t *d_en, int16_t *d_ixv, mp3s_gr_out *d_outv, int32_t *d_env)
{
    if (!d_segs || !d_spans) return fail(MP3S_E_ARG, "null pointer");
    if (n_segs <= 0 || max_reach <= 0 || max_reach > MP3S_SELECT_MAX_REACH) return fail(MP3S_E_ARG, "bad sizes");
    const int rc = mp3s_rate_variants_dev(c, d_mdct, d_frames, n_frames, d_hide_bits, n_hide, d_cursor, d_ent_unit, d_ent_cursor, n_entries, d_ix,
                                          d_out, d_en, d_ixv, d_outv, d_env);
    if (rc) return rc;
    return mp3s_select_dev(c, d_hide_bits, d_cursor, d_segs, d_spans, n_segs, max_reach, d_ent_unit, d_ent_cursor, n_entries, d_ix, d_out, d_en,
                           d_ixv, d_outv, d_env);
}

int mp3s_chain_resolve_dev(mp3s_ctx *c, mp3s_gr_out *d_gr, const mp3s_rate_frame *d_frames, int n_frames, const mp3s_chain_seg *d_segs,
                           int n_segs, const int32_t *d_cursor_in, const int32_t *d_state_in, int32_t *d_verdict,
                           mp3s_chain_seg_out *d_seg_out)
{
    if (!c || !d_gr || !d_frames || !d_segs || !d_verdict || !d_seg_out) return fail(MP3S_E_ARG, "null pointer");
    if (n_frames <= 0 || n_segs <= 0) return fail(MP3S_E_ARG, "bad sizes");
    void *d_agg = c->grab(30, chain_agg_bytes(n_frames));
    if (!d_agg) return fail(MP3S_E_NOMEM, "hipMalloc failed for the chain scratch");
    const int e = launch_chain(c->stream, d_gr, d_frames, n_frames, d_segs, d_cursor_in, d_state_in, d_agg, d_verdict, d_seg_out, &c->prof);
    if (e) return fail(MP3S_E_HIP, "chain launch: %s", hipGetErrorString((hipError_t)e));
    return MP3S_OK;
}

int mp3s_chain_redo_dev(mp3s_ctx *c, const int32_t *d_mdct, const mp3s_rate_frame *d_frames, int n_frames, const uint8_t *d_hide_bits, int n_hide,
                        int32_t *d_cursor, const mp3s_chain_seg *d_segs, int n_segs, int16_t *d_ix, mp3s_gr_out *d_gr, int32_t *d_en,
                        int32_t *d_verdict, mp3s_chain_seg_out *d_seg_out)
{
    if (!c || !d_mdct || !d_gr || !d_frames || !d_segs || !d_verdict || !d_seg_out || !d_ix || !d_en) return fail(MP3S_E_ARG, "null pointer");
    if (n_frames <= 0 || n_segs <= 0 || n_hide < 0 || (n_hide > 0 && (!d_hide_bits || !d_cursor))) return fail(MP3S_E_ARG, "bad sizes");
    void *d_agg = c->grab(30, chain_agg_bytes(n_frames));
    if (!d_agg) return fail(MP3S_E_NOMEM, "hipMalloc failed for the chain scratch");
    const ChainRedoArgs redo = {d_mdct, d_hide_bits, n_hide, d_ix, d_en};
    const int e = launch_chain(c->stream, d_gr, d_frames, n_frames, d_segs, d_cursor, nullptr, d_agg, d_verdict, d_seg_out, &c->prof, &redo);
    if (e) return fail(MP3S_E_HIP, "chain launch: %s", hipGetErrorString((hipError_t)e));
    return MP3S_OK;
}

int mp3s_huffman_decode_dev(mp3s_ctx *c, const uint8_t *d_blob, const mp3s_frame_side *d_side, int n_frames, int nch,
                            int max_part2_3_length, int16_t *d_is, mp3s_granule_si *d_si, int32_t *d_status)
{
    if (!c || !d_blob || !d_side || !d_is || !d_si || !d_status) return fail(MP3S_E_ARG, "null pointer");
    if (n_frames <= 0 || nch < 1 || nch > 2 || max_part2_3_length < 0) return fail(MP3S_E_ARG, "bad sizes");
    if (max_part2_3_length == 0 || max_part2_3_length > 4095) max_part2_3_length = 4095;
    const int e = launch_huffman(c->stream, d_blob, d_side, n_frames, nch, max_part2_3_length, d_is, d_si, d_status, c->d_sync + 4, &c->prof, false, (int)c->opt[MP3S_OPT_HUF_LANES]);
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

int mp3s_walk_stream(const uint8_t *file, size_t len, mp3s_buf **owner, mp3s_walked *out)
{
    if (!file || !owner || !out) return fail(MP3S_E_ARG, "null pointer");
    std::memset(out, 0, sizeof *out);
    std::unique_ptr<mp3s_buf> b(new mp3s_buf());
    FrameWalker w;
    const int rc = w.open(file, len);
    if (rc) return fail(rc, "malformed or unsupported MP3 stream");
    // refs and table counts live in the generic payload: [refs (16 bytes each) | tables (4 bytes each)]
    const size_t cap = len / 24 + 16;          // no Layer III frame is shorter than 24 bytes
    b->bytes.resize(cap * 20);
    FrameRef *refs = reinterpret_cast<FrameRef *>(b->bytes.data());
    uint8_t *tables = b->bytes.data() + cap * 16;
    std::memset(tables, 0, cap * 4);
    w.tables_wanted = 0x7fffffffffffffffL;
    long n = 0;
    while (!w.ended && !w.irregular && (size_t)n < cap) n += w.next(refs + n, (long)(cap - (size_t)n), tables + (size_t)n * 4, 0, 0);
    *owner = b.release();
    if (w.irregular || !w.ended) return MP3S_OK;   // regular = 0
    out->regular = 1;
    out->n_frames = (int32_t)n; out->nch = w.nch; out->sampling_rate = w.sampling_rate; out->bit_rate = w.bit_rate;
    out->dup_last_frame = w.dup_last ? 1 : 0; out->max_part2_3_length = w.max_p23; out->any_silent = w.any_silent ? 1 : 0;
    out->blob_len = n > 0 ? (size_t)refs[n - 1].md_off + refs[n - 1].md_len + 8 : 0; out->refs = refs; out->tables = tables;
    out->stream.base = 0; out->stream.end = (uint32_t)len; out->stream.first_frame = 0; out->stream.n_frames = (uint32_t)n;
    if (n > 0) FrameWalker::history(refs, 0, out->stream.prev_size);
    return MP3S_OK;
}

int mp3s_parse_frames_dev(mp3s_ctx *c, const uint8_t *d_image, uint32_t image_base, const mp3s_frame_ref *d_refs, const mp3s_stream_ref *d_streams,
                          int n_frames, uint32_t md_base, mp3s_frame_side *d_side, mp3s_frame_hdr *d_hdr, uint8_t *d_blob, uint64_t *d_tsel,
                          int32_t *d_status)
{
    if (!c || !d_image || !d_refs || !d_streams || !d_side || !d_hdr || !d_blob || !d_status) return fail(MP3S_E_ARG, "null pointer");
    if (n_frames <= 0) return fail(MP3S_E_ARG, "n_frames=%d", n_frames);
    const int e = launch_parse(c->stream, d_image, image_base, d_refs, d_streams, n_frames, md_base, d_side, d_hdr, d_blob, d_tsel, d_status);
    if (e) return fail(MP3S_E_HIP, "parse launch: %s", hipGetErrorString((hipError_t)e));
    return MP3S_OK;
}

int mp3s_stego_bits(const uint64_t *tsel, int64_t n_frames, int nch, uint8_t carry[4], mp3s_buf **owner, const uint8_t **bits, size_t *n_bits)
{
    if ((!tsel && n_frames) || n_frames < 0 || nch < 1 || nch > 2 || !carry || !owner || !bits || !n_bits) return fail(MP3S_E_ARG, "bad argument");
    mp3s_buf *b = new mp3s_buf();
    stego_bits_from_tsel(tsel, (long)n_frames, nch, carry, b->bits);
    *bits = b->bits.data(); *n_bits = b->bits.size(); *owner = b;
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
