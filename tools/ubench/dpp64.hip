// Does v_fmac_f64 with DPP row_newbcast do what k_dec_stream needs on gfx950, and at what rate?
//   1. semantics: acc += lane (16 r + j)'s src * coef, for every lane of row r
//   2. issue rate against a plain v_fma_f64 (wave64, independent accumulators), 1 and 2 waves per SIMD
//   3. ds_swizzle xor 31 (mirror within 32 lanes)
// build: hipcc --offload-arch=gfx950 -O3 -o dpp64 dpp64.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define FMAC_BCAST(acc, src, coef, J) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #J " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(coef))

__global__ void k_sem(double *out)
{
    const int lane = threadIdx.x & 63;
    double src = (double)lane, coef = 1.0 + lane * 0.001, a3 = 0.0, a12 = 100.0;
    asm volatile("s_nop 1");
    FMAC_BCAST(a3, src, coef, 3);
    FMAC_BCAST(a12, src, coef, 12);
    out[lane] = a3; out[64 + lane] = a12;
    // mirror within 32 lanes: ds_swizzle bit mode, xor 0x1f
    int lo = __double2loint(src), hi = __double2hiint(src);
    lo = __builtin_amdgcn_ds_swizzle(lo, 0x7c1f);   // and 0x1f, or 0, xor 0x1f: offset = xor << 10 | or << 5 | and
    hi = __builtin_amdgcn_ds_swizzle(hi, 0x7c1f);
    out[128 + lane] = __hiloint2double(hi, lo);
}

template <int MODE>
__global__ void k_rate(double *out, int iters)
{
    const int lane = threadIdx.x & 63;
    double src = 1.0 + lane * 1e-9, coef = 1.0 - lane * 1e-9;
    double a[8];
    for (int i = 0; i < 8; i++) a[i] = i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            if (MODE == 0) {
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "v"(src), "v"(coef));
            } else if (MODE == 1) {
                FMAC_BCAST(a[0], src, coef, 0); FMAC_BCAST(a[1], src, coef, 1); FMAC_BCAST(a[2], src, coef, 2); FMAC_BCAST(a[3], src, coef, 3);
                FMAC_BCAST(a[4], src, coef, 4); FMAC_BCAST(a[5], src, coef, 5); FMAC_BCAST(a[6], src, coef, 6); FMAC_BCAST(a[7], src, coef, 7);
            } else if (MODE == 2) {   // one dependent chain, plain
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[0]) : "v"(src), "v"(coef));
            } else {                  // one dependent chain, dpp
                FMAC_BCAST(a[0], src, coef, 0); FMAC_BCAST(a[0], src, coef, 1); FMAC_BCAST(a[0], src, coef, 2); FMAC_BCAST(a[0], src, coef, 3);
                FMAC_BCAST(a[0], src, coef, 4); FMAC_BCAST(a[0], src, coef, 5); FMAC_BCAST(a[0], src, coef, 6); FMAC_BCAST(a[0], src, coef, 7);
            }
        }
    }
    double s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main()
{
    double *d;
    hipMalloc(&d, 1 << 24);
    k_sem<<<1, 64>>>(d);
    std::vector<double> h(192);
    hipMemcpy(h.data(), d, 192 * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++) {
        const int r = l / 16;
        const double e3 = (16 * r + 3) * (1.0 + l * 0.001), e12 = 100.0 + (16 * r + 12) * (1.0 + l * 0.001);
        if (h[l] != e3 || h[64 + l] != e12 || h[128 + l] != (double)((l & 32) | (31 - (l & 31)))) { if (bad < 8) printf("lane %d: %g (%g) %g (%g) %g\n", l, h[l], e3, h[64 + l], e12, h[128 + l]); bad++; }
    }
    printf("semantics: %s (%d lanes off)\n", bad ? "DIFFERENT" : "ok: row_newbcast:j multiplies lane 16 r + j's src by the lane's own coef; swizzle mirrors 32 lanes", bad);
    const int iters = 4096;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[4] = {"v_fma_f64, 8 independent", "v_fmac_f64_dpp row_newbcast, 8 independent", "v_fma_f64, one chain", "v_fmac_f64_dpp, one chain"};
    for (int wpsimd = 1; wpsimd <= 2; wpsimd++)
        for (int m = 0; m < 4; m++) {
            const int blocks = 256, threads = 256 * wpsimd;   // one block per CU: 4 or 8 waves
            float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if (m == 0) k_rate<0><<<blocks, threads>>>(d, iters);
                if (m == 1) k_rate<1><<<blocks, threads>>>(d, iters);
                if (m == 2) k_rate<2><<<blocks, threads>>>(d, iters);
                if (m == 3) k_rate<3><<<blocks, threads>>>(d, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            const double n = (double)iters * 32;                       // instructions per wave
            printf("%d wave(s) per SIMD, %-44s: %.2f ns per instruction and wave (%.2f clk at 2.4 GHz)\n", wpsimd, names[m], ms * 1e6 / n, ms * 1e6 / n * 2.4);
        }
    return 0;
}
