#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t N = 52u << 20;
    std::vector<uint8_t> src(N, 1);
    void *d; hipMalloc(&d, N);
    hipStream_t s; hipStreamCreate(&s);
    for (int it = 0; it < 3; it++) { double t0 = now(); hipMemcpyAsync(d, src.data(), N, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); printf("pageable direct   %.3f ms\n", now() - t0); }
    const size_t C = 4u << 20; void *p[2]; hipEvent_t ev[2];
    for (int i = 0; i < 2; i++) { hipHostMalloc(&p[i], C, hipHostMallocDefault); hipEventCreateWithFlags(&ev[i], hipEventDisableTiming); }
    for (size_t chunk : {1u << 20, 2u << 20, 4u << 20}) for (int it = 0; it < 2; it++) {
        double t0 = now(); int k = 0; bool used[2] = {false, false};
        for (size_t off = 0; off < N; off += chunk, k ^= 1) {
            const size_t n = std::min(chunk, N - off);
            if (used[k]) hipEventSynchronize(ev[k]);
            memcpy(p[k], src.data() + off, n);
            hipMemcpyAsync((uint8_t *)d + off, p[k], n, hipMemcpyHostToDevice, s);
            hipEventRecord(ev[k], s); used[k] = true;
        }
        hipStreamSynchronize(s);
        printf("staged %zu MB chunks %.3f ms\n", chunk >> 20, now() - t0);
    }
    std::vector<uint8_t> fresh(N);  // fresh pages as a scan result would be
    { double t0 = now(); hipMemcpyAsync(d, fresh.data(), N, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); printf("pageable (zero pages) %.3f ms\n", now() - t0); }
    return 0;
}
