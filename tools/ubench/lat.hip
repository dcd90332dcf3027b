// Micro-benchmark: what ONE wave pays per instruction in a dependent chain (gfx950) -- the decode loops of k_dec_huffman are such
// chains.  hipcc --offload-arch=gfx950 -O3 -o lat lat.hip && ./lat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N 2048
template <int OP>
__global__ void k(uint32_t *out, uint32_t seed)
{
    __shared__ uint32_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (i * 37 + 11) & 4095;
    __syncthreads();
    uint32_t a = seed + threadIdx.x, b = seed * 3 + 1, c = threadIdx.x;
    uint64_t w = ((uint64_t)seed << 32) | threadIdx.x;
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 16
    for (int it = 0; it < N; it++) {
        if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));
        if (OP == 1) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(w));
        if (OP == 2) { a = lds[a & 4095]; asm volatile("" : "+v"(a)); }
        if (OP == 3) { a = ((const uint16_t *)lds)[a & 8191]; asm volatile("" : "+v"(a)); }
        if (OP == 4) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_add_u32 %0, %0, %2" : "+v"(a) : "v"(b), "v"(c) : "vcc");
        if (OP == 5) asm volatile("v_cmp_lt_u32 s[10:11], %0, %1\n\ts_and_b64 s[10:11], s[10:11], exec\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %1, s[10:11]\n\tv_add_u32 %0, %0, %2" : "+v"(a) : "v"(b), "v"(c) : "s10", "s11");
        if (OP == 6) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\ts_and_saveexec_b64 s[10:11], vcc\n\ts_cbranch_execz 1f\n\tv_add_u32 %0, %0, %2\n1:\n\ts_or_b64 exec, exec, s[10:11]\n\tv_add_u32 %0, %0, 1" : "+v"(a) : "v"(b), "v"(c) : "s10", "s11", "vcc");
        if (OP == 7) asm volatile("v_bfe_u32 %0, %0, 3, 9\n\tv_lshl_add_u32 %0, %0, 1, %1" : "+v"(a) : "v"(b));
        if (OP == 8) asm volatile("v_add_u32 %0, %0, %2\n\tv_add_u32 %1, %1, %2" : "+v"(a), "+v"(c) : "v"(b));   // two independent chains
        if (OP == 9) { lds[(threadIdx.x * 17 + (it & 15)) & 4095] = a; asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b)); }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = a + (uint32_t)w + c;
    if (threadIdx.x == 0) out[64] = (uint32_t)(t1 - t0);
}
template <int OP> void run(const char *name, int per_iter)
{
    uint32_t *out; hipMalloc(&out, 4096);
    k<OP><<<1, 64>>>(out, 12345); hipDeviceSynchronize();
    k<OP><<<1, 64>>>(out, 12345); hipDeviceSynchronize();
    uint32_t h[65]; hipMemcpy(h, out, 65 * 4, hipMemcpyDeviceToHost);
    printf("%-52s %.1f clk per iteration (%d instr: %.1f clk each)\n", name, (double)h[64] / N, per_iter, (double)h[64] / N / per_iter);
    hipFree(out);
}
int main()
{
    run<0>("dependent v_add_u32", 1);
    run<8>("two independent v_add_u32 chains", 2);
    run<1>("dependent v_lshlrev_b64", 1);
    run<7>("dependent v_bfe_u32 + v_lshl_add_u32", 2);
    run<2>("LDS pointer chase ds_read_b32 (+and)", 2);
    run<3>("LDS pointer chase ds_read_u16 (+and)", 2);
    run<4>("v_cmp vcc + v_cndmask vcc + v_add", 3);
    run<5>("v_cmp sgpr + s_and + v_cndmask sgpr + v_add", 4);
    run<6>("v_cmp + saveexec + cbranch(not taken: all lanes) + add + or exec + add", 6);
    run<9>("ds_write_b32 + v_add", 2);
    return 0;
}
