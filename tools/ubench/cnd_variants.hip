// v_cndmask_b32 in the forms the compiler emits: condition from VCC written by a v_cmp in front of it, from an SGPR pair, from a VCC nobody wrote.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 2048
template <int OP>
__global__ void k(uint32_t *out, uint32_t seed)
{
    uint32_t a[8], b = seed | 1u;
    for (int i = 0; i < 8; i++) a[i] = seed * (threadIdx.x + i + 1);
    unsigned long long m = 0x5555aaaa5555aaaaull + seed;
    asm volatile("s_mov_b64 vcc, %0" : : "s"(m) : "vcc");
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
            if (OP == 1) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(m));
            if (OP == 2) asm volatile("v_cmp_gt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
            if (OP == 3) asm volatile("v_cmp_gt_u32_e64 %2, %0, %1\n\tv_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]), "+v"(b), "+s"(m));
            if (OP == 4) asm volatile("v_cmp_gt_u32 vcc, %0, %1\n\tv_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b) : "vcc");
            if (OP == 5) asm volatile("v_max_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> void run(const char *name, uint32_t *out, int per)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w : {1, 4, 5}) {
        k<OP><<<256 * w, 256>>>(out, 12345); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0); k<OP><<<256 * w, 256>>>(out, 12345); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %d waves/SIMD: %.2f ns per wave instruction and SIMD\n", name, w, ms * 1e6 / ((double)ITER * 8 * per * w));
    }
}
int main()
{
    uint32_t *out; (void)hipMalloc(&out, (size_t)256 * 5 * 256 * 4);
    run<0>("v_cndmask_b32 vcc (set by s_mov)", out, 1);
    run<1>("v_cndmask_b32_e64 sgpr pair", out, 1);
    run<2>("v_cmp vcc + v_cndmask vcc (per instr)", out, 2);
    run<3>("v_cmp_e64 sgpr + v_cndmask_e64 (per instr)", out, 2);
    run<4>("v_cmp vcc + v_add_u32 (per instr)", out, 2);
    run<5>("v_max_u32 + v_add_u32 (per instr)", out, 2);
    return 0;
}
