// One-file calls: the whole file goes to the device in pieces that a helper thread queues while the caller walks the frame
// headers.
#include "pipe_internal.h"

namespace {

void file_up_thread(mp3s_pipe *P)
{
    FileUp &u = P->up;
    (void)hipSetDevice(P->c->device);
    if (!P->node_cpus.empty()) bind_to(P->node_cpus);
    std::unique_lock<std::mutex> lk(u.mu);
    for (;;) {
        u.cv.wait(lk, [&] { return u.stop || (u.busy && !u.started); });
        if (u.stop) return;
        u.started = true;
        lk.unlock();
        // (piece 0 is the caller's own: it needs it first, and this thread takes longer to wake up than the copy takes)
        while (u.recorded.load(std::memory_order_acquire) < 1 && !u.err.load()) std::this_thread::yield();
        size_t from = u.ends[0];
        for (size_t i = 1; i < u.ends.size() && !u.err.load(); i++) {
            if (hipMemcpyAsync(u.d_file + from, u.src + from, u.ends[i] - from, hipMemcpyHostToDevice, P->s_img) != hipSuccess ||
                hipEventRecord(u.ev[i], P->s_img) != hipSuccess) { u.err.store(1); break; }
            from = u.ends[i];
            u.recorded.store((long)i + 1, std::memory_order_release);
        }
        lk.lock();
        u.busy = false;
        u.cv.notify_all();
    }
}

}  // namespace

// queue the upload of file[0, len): a first piece of first_bytes (the first chunk's), then kFilePiece at a time
bool file_up_begin(mp3s_pipe *P, const uint8_t *file, size_t len, size_t first_bytes, long n_est)
{
    FileUp &u = P->up;
    u.active = false;
    if (!P->internal || !P->s_img || len > kFileOnDevice || !P->c->opt[MP3S_OPT_FILE_UP]) return false;
    if (len + 256 > u.cap) {
        if (u.d_file) (void)hipFree(u.d_file);
        u.d_file = nullptr; u.cap = 0;
        const size_t want = std::max<size_t>(len + len / 4 + 4096, (size_t)8 << 20);
        if (hipMalloc((void **)&u.d_file, want) != hipSuccess) { (void)hipGetLastError(); return false; }
        u.cap = want;
    }
    // the file-wide side records and main data (a frame's main data: at most its own bytes + 511 of reservoir; 8 zero bytes and
    // up to 3 of alignment behind each)
    // (every frame's share counted in full: 511 bytes of reservoir + 8 zero bytes + 3 of alignment -- with 12 per frame, as until round 5, a stream
    // that uses its reservoir to the full did not fit, its chunks were not file-wide and a mixed-block file went through the stages twice)
    const size_t want_frames = (size_t)n_est + 64, want_blob = len + (size_t)(n_est + 64) * (511 + 12) + 4096;
    if (want_frames > u.side_cap || want_blob > u.blob_cap) {
        if (u.d_side) (void)hipFree(u.d_side);
        if (u.d_blob) (void)hipFree(u.d_blob);
        u.d_side = nullptr; u.d_blob = nullptr; u.side_cap = u.blob_cap = 0;
        const size_t nf = want_frames + want_frames / 4, nb = want_blob + want_blob / 4;
        if (hipMalloc((void **)&u.d_side, nf * sizeof(mp3s_frame_side)) != hipSuccess || hipMalloc((void **)&u.d_blob, nb) != hipSuccess) {
            (void)hipGetLastError();
            if (u.d_side) (void)hipFree(u.d_side);
            u.d_side = nullptr; u.d_blob = nullptr;
        } else { u.side_cap = nf; u.blob_cap = nb; }
    }
    u.ends.clear();
    size_t at = std::min(len, std::max<size_t>(first_bytes, 4096));
    u.ends.push_back(at);
    while (at < len) { at = std::min(len, at + kFilePiece); u.ends.push_back(at); }
    while (u.ev.size() < u.ends.size()) {
        hipEvent_t e = nullptr;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return false;
        u.ev.push_back(e);
    }
    u.recorded.store(0); u.err.store(0);
    {
        std::lock_guard<std::mutex> g(u.mu);
        if (!u.th.joinable()) {
            try { u.th = std::thread(file_up_thread, P); }
            catch (const std::exception &) { return false; }     // (no thread to be had: every chunk's piece from the caller, as for a file above 1 GB)
        }
        u.src = file; u.busy = true; u.started = false;
    }
    u.cv.notify_all();
    u.active = true;
    if (hipMemcpyAsync(u.d_file, file, u.ends[0], hipMemcpyHostToDevice, P->s_img) != hipSuccess || hipEventRecord(u.ev[0], P->s_img) != hipSuccess) {
        (void)hipGetLastError();
        u.err.store(1);
        file_up_end(P);
        return false;
    }
    u.recorded.store(1, std::memory_order_release);
    return true;
}

// `stream` waits until file[0, need) is on the device
int file_up_wait(mp3s_pipe *P, size_t need, hipStream_t stream)
{
    FileUp &u = P->up;
    size_t i = 0;
    while (i + 1 < u.ends.size() && u.ends[i] < need) i++;
    while (u.recorded.load(std::memory_order_acquire) <= (long)i) {
        if (u.err.load()) return fail(MP3S_E_HIP, "uploading the file failed");
        std::this_thread::yield();
    }
    HIPCHK(hipStreamWaitEvent(stream, u.ev[i], 0));
    return MP3S_OK;
}

// the call is over: the helper is done with the caller's memory
void file_up_end(mp3s_pipe *P)
{
    FileUp &u = P->up;
    if (!u.active) return;
    std::unique_lock<std::mutex> lk(u.mu);
    u.cv.wait(lk, [&] { return !u.busy; });
    u.active = false;
}
