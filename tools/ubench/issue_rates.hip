// Issue cost of the vector instructions the step is made of, per opcode class (gfx950): wave64 instructions per SIMD and time with
// 1, 2, 4 and 5 waves on every SIMD, eight independent chains per wave (so that neither a dependency nor an empty issue slot is
// measured, only the pipe).  Every form is inline assembly: what is timed is the opcode named, not what the compiler makes of it.
// The table is what bench.py's roofline_alu weights the dominant kernel's instruction mix with (profiles/r05_issue_rates.json).
// build: hipcc --offload-arch=gfx950 -O3 -o issue_rates issue_rates.hip ; run: ./issue_rates [json]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#define ITER 2048
enum { ADD_U32, CNDMASK, BFE, AND, LSHL_ADD, ADD3, MAX_U32, DPP_ADD, DPP_MOV, READLANE, MAD_U64, MUL_HI, MUL_LO, FMA_F64, MUL_F64, ADD_F64, FMAC_DPP64, CMP_U32, MOV_B32, N_OPS };
static const char *NAMES[N_OPS] = {"v_add_u32", "v_cndmask_b32", "v_bfe_u32", "v_and_b32", "v_lshl_add_u32", "v_add3_u32", "v_max_u32", "v_add_u32 dpp row_shr:1",
                                   "v_mov_b32 dpp row_shr:1", "v_readlane_b32", "v_mad_u64_u32", "v_mul_hi_i32", "v_mul_lo_u32", "v_fma_f64", "v_mul_f64", "v_add_f64",
                                   "v_fmac_f64 dpp row_newbcast", "v_cmp_gt_u32 (vcc)", "v_mov_b32"};
template <int OP>
__global__ void k(uint32_t *out, uint32_t seed)
{
    uint32_t a[8], b = seed | 1u, c = seed ^ 0x5555u;
    double d[8], db = 1.0000001, dc = 0.9999999;
    unsigned long long w[8];
    for (int i = 0; i < 8; i++) { a[i] = seed * (threadIdx.x + i + 1); d[i] = 1.0 + 1e-9 * (threadIdx.x + i); w[i] = a[i]; }
    int sacc = 0;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : );
            if (OP == BFE) asm volatile("v_bfe_u32 %0, %0, 3, 17" : "+v"(a[i]));
            if (OP == AND) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == ADD3) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == MAX_U32) asm volatile("v_max_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == DPP_ADD) asm volatile("v_add_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b));
            if (OP == DPP_MOV) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b));
            if (OP == READLANE) { int s; asm volatile("v_readlane_b32 %0, %1, 7" : "=s"(s) : "v"(a[i])); sacc += s; }
            if (OP == MAD_U64) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(b), "v"(c) : "vcc");
            if (OP == MUL_HI) asm volatile("v_mul_hi_i32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == MUL_LO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == FMA_F64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(db), "v"(dc));
            if (OP == MUL_F64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(db));
            if (OP == ADD_F64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dc));
            if (OP == FMAC_DPP64) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(d[i]) : "v"(db), "v"(dc));
            if (OP == CMP_U32) asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
            if (OP == MOV_B32) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b));
        }
    }
    uint32_t s = (uint32_t)sacc; double ds = 0;
    for (int i = 0; i < 8; i++) { s += a[i] + (uint32_t)w[i]; ds += d[i]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (uint32_t)ds;
}
template <int OP> double run(int waves_per_simd, uint32_t *out)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 256 * waves_per_simd;   // 256 threads = 4 waves = one per SIMD of a CU; 256 CUs
    k<OP><<<blocks, 256>>>(out, 12345);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0);
        k<OP><<<blocks, 256>>>(out, 12345);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best * 1e6 / ((double)ITER * 8 * waves_per_simd);      // ns per wave instruction and SIMD
}
template <int OP> void row(uint32_t *out, bool json, bool last)
{
    const int ws[4] = {1, 2, 4, 5};
    double ns[4];
    for (int i = 0; i < 4; i++) ns[i] = run<OP>(ws[i], out);
    if (json) printf("  \"%s\": {\"ns_1\": %.3f, \"ns_2\": %.3f, \"ns_4\": %.3f, \"ns_5\": %.3f}%s\n", NAMES[OP], ns[0], ns[1], ns[2], ns[3], last ? "" : ",");
    else printf("%-30s %7.2f %7.2f %7.2f %7.2f   ns per wave instruction and SIMD at 1 / 2 / 4 / 5 waves per SIMD   (%.2f clk at 2.4 GHz with 5)\n", NAMES[OP], ns[0], ns[1], ns[2], ns[3], ns[3] * 2.4);
}
template <int OP> void all(uint32_t *out, bool json)
{
    row<OP>(out, json, OP == N_OPS - 1);
    if constexpr (OP + 1 < N_OPS) all<OP + 1>(out, json);
}
int main(int argc, char **argv)
{
    const bool json = argc > 1 && !strcmp(argv[1], "json");
    uint32_t *out; (void)hipMalloc(&out, (size_t)256 * 5 * 256 * 4);
    if (json) printf("{\n");
    all<0>(out, json);
    if (json) printf("}\n");
    return 0;
}
