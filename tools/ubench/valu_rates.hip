// Micro-benchmark: issue rate of the VALU instructions the hot path is made of (gfx950).
// hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 4096
template <int OP>
__global__ void k(uint32_t *out, uint32_t seed, double dseed)
{
    uint32_t a[8]; double d[8];
    for (int i = 0; i < 8; i++) { a[i] = seed * (threadIdx.x + i + 1); d[i] = dseed * (threadIdx.x + i + 1); }
    uint32_t b = seed | 1; double db = dseed + 1.0;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) a[i] = __mulhi((int)a[i], (int)b) + 1;                       // v_mul_hi_i32 (+add)
            if (OP == 1) a[i] = a[i] * b;                                              // v_mul_lo_u32
            if (OP == 2) a[i] = a[i] + b;                                              // v_add_u32
            if (OP == 3) d[i] = d[i] * db;                                             // v_mul_f64
            if (OP == 4) d[i] = d[i] + db;                                             // v_add_f64
            if (OP == 5) d[i] = __fma_rn(d[i], db, db);                                // v_fma_f64
            if (OP == 6) a[i] = (uint32_t)__mul24((int)a[i], (int)b);                  // v_mul_i32_i24
            if (OP == 7) a[i] = __umulhi(a[i], b);                                     // v_mul_hi_u32 alone
            if (OP == 8) d[i] = d[i] * db + 1.0;                                       // mul + add (no contraction)
            if (OP == 9) { int t; asm("v_mul_hi_i32 %0, %1, %2" : "=v"(t) : "v"(a[i]), "v"(b)); a[i] = (uint32_t)t + a[(i + 1) & 7]; }   // forced mul_hi + add
            if (OP == 10) { a[i] = a[i] + a[(i + 3) & 7]; a[i] ^= it; }                 // add + xor (2 full-rate ops)
            if (OP == 12) { long long t; asm("v_mad_i64_i32 %0, vcc, %1, %2, 0" : "=v"(t) : "v"(a[i]), "s"(b) : "vcc"); a[i] = (uint32_t)(t >> 32) + a[(i + 1) & 7]; }   // mad_i64_i32 (hi word) + add
            if (OP == 13) { int t; asm("v_mul_hi_i32 %0, %1, %2" : "=v"(t) : "v"(a[i]), "s"(b)); a[i] = (uint32_t)t; }   // mul_hi alone (sgpr operand)
            if (OP == 14) { long long t; asm("v_mad_i64_i32 %0, vcc, %1, %2, 0" : "=v"(t) : "v"(a[i]), "s"(b) : "vcc"); a[i] = (uint32_t)(t >> 32); }   // mad_i64_i32 alone
            if (OP == 15) { unsigned long long t; asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(t) : "v"(a[i]), "s"(b), "v"((unsigned long long)a[(i + 1) & 7]) : "vcc"); a[i] = (uint32_t)(t >> 32); }   // mad_u64_u32 with 64-bit addend
            if (OP == 11) { int t; asm("v_mul_hi_i32 %0, %1, %2" : "=v"(t) : "v"(a[i]), "s"(b)); a[i] = (uint32_t)t + a[(i + 1) & 7]; } // scalar operand
        }
    }
    uint32_t s = 0; double ds = 0;
    for (int i = 0; i < 8; i++) { s += a[i]; ds += d[i]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (uint32_t)ds;
}
template <int OP> void run(const char *name, int waves_per_simd)
{
    uint32_t *out; hipMalloc(&out, 256 * 1024 * 64 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * waves_per_simd;   // 256 threads per block = 4 waves = 1 per SIMD
    k<OP><<<blocks, 256>>>(out, 12345, 1.0000001);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(out, 12345, 1.0000001);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)ITER * 8 * waves_per_simd;      // wave-instructions per SIMD
    printf("%-28s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instr per SIMD  (%.2f clk @2.4GHz)\n", name, waves_per_simd, ms,
           ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
    hipFree(out);
}
int main()
{
    for (int w : {1, 2, 4}) {
        run<0>("mul_hi_i32 + add", w); run<7>("mul_hi_u32", w); run<1>("mul_lo_u32", w); run<2>("add_u32", w);
        run<6>("mul_i32_i24", w); run<3>("mul_f64", w); run<4>("add_f64", w); run<5>("fma_f64", w); run<8>("mul_f64 + add_f64", w); run<9>("asm mul_hi_i32 + add", w); run<10>("add + xor", w); run<11>("asm mul_hi_i32(sgpr) + add", w); run<12>("asm mad_i64_i32(sgpr) + add", w); run<13>("asm mul_hi_i32(sgpr) alone", w); run<14>("asm mad_i64_i32 alone", w); run<15>("asm mad_u64_u32 + 64b addend", w);
    }
    return 0;
}
