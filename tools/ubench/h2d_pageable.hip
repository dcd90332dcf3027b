// How a file image in ordinary (pageable) memory gets to the device: one hipMemcpyAsync straight from the caller's
// buffer (the runtime stages or pins on its own) against a memcpy into page-locked staging followed by an asynchronous copy.
// Prints, per size: the time the call keeps the host (returns) and the time until the bytes are on the device (done).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const size_t MAXN = 64u << 20;
    void *d; hipMalloc(&d, MAXN);
    void *pin; hipHostMalloc(&pin, MAXN, hipHostMallocDefault);
    std::vector<uint8_t> a(MAXN, 1), b(MAXN, 2);
    memset(pin, 3, MAXN);
    for (size_t n : {(size_t)64 << 10, (size_t)256 << 10, (size_t)1 << 20, (size_t)4 << 20, (size_t)16 << 20, (size_t)64 << 20}) {
        for (int it = 0; it < 4; it++) {
            const uint8_t *src = (it & 1) ? b.data() : a.data();
            double t0 = now(); hipMemcpyAsync(d, src, n, hipMemcpyHostToDevice, s); double t1 = now(); hipStreamSynchronize(s); double t2 = now();
            if (it >= 2) printf("%8zu KB pageable direct : returns %.3f ms, done %.3f ms (%.1f GB/s)\n", n >> 10, t1 - t0, t2 - t0, n / (t2 - t0) / 1e6);
        }
        for (int it = 0; it < 3; it++) {
            double t0 = now(); memcpy(pin, a.data(), n); double t1 = now(); hipMemcpyAsync(d, pin, n, hipMemcpyHostToDevice, s); double t2 = now(); hipStreamSynchronize(s); double t3 = now();
            if (it >= 1) printf("%8zu KB memcpy to pinned : memcpy %.3f ms, call %.3f ms, done %.3f ms\n", n >> 10, t1 - t0, t2 - t1, t3 - t0);
        }
        for (int it = 0; it < 3; it++) {
            double t0 = now(); hipMemcpyAsync(d, pin, n, hipMemcpyHostToDevice, s); double t1 = now(); hipStreamSynchronize(s); double t2 = now();
            if (it >= 1) printf("%8zu KB pinned           : call %.3f ms, done %.3f ms (%.1f GB/s)\n", n >> 10, t1 - t0, t2 - t0, n / (t2 - t0) / 1e6);
        }
    }
    // registering the caller's buffer for the duration of a call
    for (size_t n : {(size_t)4 << 20, (size_t)42 << 20}) for (int it = 0; it < 3; it++) {
        double t0 = now(); hipError_t e = hipHostRegister(a.data(), n, hipHostRegisterDefault); double t1 = now();
        hipMemcpyAsync(d, a.data(), n, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); double t2 = now();
        hipHostUnregister(a.data()); double t3 = now();
        printf("%8zu KB hipHostRegister(%d): register %.3f ms, copy %.3f ms, unregister %.3f ms\n", n >> 10, (int)e, t1 - t0, t2 - t1, t3 - t2);
    }
    return 0;
}
