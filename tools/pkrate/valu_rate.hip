// Issue cost of vector instruction classes on gfx950 (one wave64 instruction = ? cycles of its SIMD), 4 waves per SIMD resident,
// independent chains:  hipcc --offload-arch=gfx950 -O3 -o tools/pkrate/valu_rate tools/pkrate/valu_rate.hip && tools/pkrate/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X X X X X X X X
#define BODY(ASM)                                                                                  \
    for (int it = 0; it < iters; ++it) {                                                          \
        REP8(asm volatile(ASM "\n\t" ASM##1 "\n\t" ASM##2 "\n\t" ASM##3 : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));) \
    }
template <int OP>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
    const unsigned long long msk = 0x5555555555555555ull * (1 + (blockIdx.x & 1));
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f pa = {1.f, 2.f}, pb = {1.0001f, 0.9999f}, pc = {0.5f, 0.25f}, pd = {3.f, 4.f};
    float a = threadIdx.x * 1e-3f, b = a + 1.f, c = a + 2.f, d = a + 3.f, e = 1.0001f, f = 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
            if (OP == 1) asm volatile("v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
            if (OP == 2) asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n\tv_cndmask_b32 %1, %1, %4, vcc\n\tv_cndmask_b32 %2, %2, %4, vcc\n\tv_cndmask_b32 %3, %3, %4, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f) : "vcc");
            if (OP == 3) asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
            if (OP == 4) asm volatile("v_rcp_f32 %0, %0\n\tv_rcp_f32 %1, %1\n\tv_rcp_f32 %2, %2\n\tv_rcp_f32 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
            if (OP == 5) asm volatile("v_mul_f32 %0, %0, %4\n\tv_mul_f32 %1, %1, %4\n\tv_mul_f32 %2, %2, %4\n\tv_mul_f32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
            if (OP == 6) asm volatile("v_add_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %4\n\tv_add_f32 %2, %2, %4\n\tv_add_f32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
            if (OP == 7) asm volatile("v_cvt_f16_f32 %0, %0\n\tv_cvt_f16_f32 %1, %1\n\tv_cvt_f16_f32 %2, %2\n\tv_cvt_f16_f32 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
            if (OP == 8) asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %4\n\tv_mov_b32 %2, %4\n\tv_mov_b32 %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
            if (OP == 9) asm volatile("v_lshlrev_b32 %0, 1, %0\n\tv_lshlrev_b32 %1, 1, %1\n\tv_lshlrev_b32 %2, 1, %2\n\tv_lshlrev_b32 %3, 1, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
            if (OP == 10) asm volatile("v_mov_b32_dpp %0, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %2, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
            if (OP == 11) asm volatile("v_pk_mul_f16 %0, %0, %4\n\tv_pk_mul_f16 %1, %1, %4\n\tv_pk_mul_f16 %2, %2, %4\n\tv_pk_mul_f16 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
            if (OP == 12) asm volatile("v_cmp_gt_f32 vcc, %0, %4\n\tv_cmp_gt_f32 vcc, %1, %4\n\tv_cmp_gt_f32 vcc, %2, %4\n\tv_cmp_gt_f32 vcc, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f) : "vcc");
            if (OP == 13) asm volatile("v_max_f32 %0, %0, %4\n\tv_max_f32 %1, %1, %4\n\tv_max_f32 %2, %2, %4\n\tv_max_f32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
            if (OP == 14) asm volatile("v_mad_u32_u24 %0, %0, %4, %5\n\tv_mad_u32_u24 %1, %1, %4, %5\n\tv_mad_u32_u24 %2, %2, %4, %5\n\tv_mad_u32_u24 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
            if (OP == 16) asm volatile("v_cndmask_b32_e64 %0, %0, %4, %6\n\tv_cndmask_b32_e64 %1, %1, %4, %6\n\tv_cndmask_b32_e64 %2, %2, %4, %6\n\tv_cndmask_b32_e64 %3, %3, %4, %6" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "s"(msk));
            if (OP == 17) asm volatile("v_cmp_gt_f32 vcc, %0, %4\n\tv_cndmask_b32 %1, %1, %4, vcc\n\tv_cmp_gt_f32 vcc, %2, %4\n\tv_cndmask_b32 %3, %3, %4, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f) : "vcc");
            if (OP == 18) asm volatile("v_cndmask_b32 %0, %5, %4, vcc\n\tv_cndmask_b32 %1, %5, %4, vcc\n\tv_cndmask_b32 %2, %5, %4, vcc\n\tv_cndmask_b32 %3, %5, %4, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f) : "vcc");
            if (OP == 19) asm volatile("v_sub_f32 %0, %0, %4\n\tv_sub_f32 %1, %1, %4\n\tv_sub_f32 %2, %2, %4\n\tv_sub_f32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
            if (OP == 20) asm volatile("v_xor_b32 %0, %0, %4\n\tv_xor_b32 %1, %1, %4\n\tv_xor_b32 %2, %2, %4\n\tv_xor_b32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
            if (OP == 21) asm volatile("v_fmac_f32 %0, %4, %5\n\tv_fmac_f32 %1, %4, %5\n\tv_fmac_f32 %2, %4, %5\n\tv_fmac_f32 %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
            if (OP == 22) asm volatile("v_pk_fma_f32 %0, %1, %2, %0\n\tv_pk_fma_f32 %3, %1, %2, %3\n\tv_pk_fma_f32 %0, %1, %2, %0\n\tv_pk_fma_f32 %3, %1, %2, %3" : "+v"(pa), "+v"(pb), "+v"(pc), "+v"(pd));
            if (OP == 23) asm volatile("v_cvt_f32_f16 %0, %0\n\tv_cvt_f32_f16 %1, %1\n\tv_cvt_f32_f16 %2, %2\n\tv_cvt_f32_f16 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
            if (OP == 24) asm volatile("v_and_b32 %0, %0, %4\n\tv_and_b32 %1, %1, %4\n\tv_and_b32 %2, %2, %4\n\tv_and_b32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
            if (OP == 25) asm volatile("v_log_f32 %0, %0\n\tv_log_f32 %1, %1\n\tv_log_f32 %2, %2\n\tv_log_f32 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
            if (OP == 15) asm volatile("v_mul_lo_u32 %0, %0, %4\n\tv_mul_lo_u32 %1, %1, %4\n\tv_mul_lo_u32 %2, %2, %4\n\tv_mul_lo_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + pa.x + pd.y;
}
template <int OP> void run(const char *name, float *out) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 4000, blocks = 1024 * 4;   // 4 workgroups of 4 waves per CU x 4: 4 waves per SIMD resident
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double instr = (double)blocks * 4 * iters * 32;   // wave-instructions
    const double per_simd_per_s = instr / (ms * 1e-3) / (256.0 * 4);
    printf("%-16s %7.2f ms  %6.3f G instr/s per SIMD  (= %.2f cycles per instruction at 2.4 GHz)\n", name, ms, per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s);
}
int main() {
    float *out; (void)hipMalloc(&out, 4096 * 4 * 256 * 4);
    run<0>("v_fma_f32", out); run<5>("v_mul_f32", out); run<6>("v_add_f32", out); run<13>("v_max_f32", out);
    run<1>("v_add_u32", out); run<9>("v_lshlrev_b32", out); run<8>("v_mov_b32", out); run<2>("v_cndmask_b32", out); run<12>("v_cmp_gt_f32", out);
    run<10>("v_mov_b32_dpp", out); run<7>("v_cvt_f16_f32", out); run<11>("v_pk_mul_f16", out); run<14>("v_mad_u32_u24", out); run<15>("v_mul_lo_u32", out);
    run<16>("v_cndmask_e64 sgpr", out); run<17>("cmp+cndmask vcc", out); run<18>("v_cndmask vcc (indep)", out); run<19>("v_sub_f32", out); run<20>("v_xor_b32", out); run<24>("v_and_b32", out); run<21>("v_fmac_f32", out); run<22>("v_pk_fma_f32", out); run<23>("v_cvt_f32_f16", out); run<25>("v_log_f32", out);
    run<3>("v_exp_f32", out); run<4>("v_rcp_f32", out);
    return 0;
}
