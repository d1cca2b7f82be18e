// Operand layout and timing of v_mfma_scale_f32_16x16x128_f8f6f4 with fp8 (E4M3) / bf8 (E5M2) operands -- the instruction DESIGN.md 5b-2
// would run the exactly-representable P2 x a_hi product on (one 32-cycle MFMA over K = 128 instead of four 16-cycle f16 MFMAs over K = 32).
//   hipcc --offload-arch=gfx950 -O3 -o tools/mx_probe tools/mx_probe.hip && tools/mx_probe
// 1. Layout check with exact small integers: A (16 x 128) and B (128 x 16) random in {-3 .. 3} (exact in both 8-bit formats), packed under the
//    hypothesis  lane l = (r = l & 15, kq = l >> 4), byte j of the lane's 32 bytes <-> k = 32 kq + j  (A: row r; B: column r), scales 2^0;
//    D (col = l & 15, row = 4 (l >> 4) + reg) must equal A.B exactly.  (Any k order shared by A and B gives the same sums: what the check pins is
//    the row / column <-> lane map and that a lane group's 32 bytes of A meet the same lane group's 32 bytes of B.)
// 2. The scale operand: A scaled by 2^3 through scale_a = 130 must give 8 x the result.
// 3. Cycles per instruction (one wave per SIMD, four independent accumulators), beside v_mfma_f32_16x16x32_f16.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
using i32x8 = __attribute__((ext_vector_type(8))) int;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

// small integers as E4M3 (bias 7, 3 mantissa bits) and E5M2 (bias 15, 2 mantissa bits); |v| <= 3 is exact in both
static uint8_t enc(int v, bool e5m2) {
    if (v == 0) return 0;
    const uint8_t s = v < 0 ? 0x80 : 0;
    const int a = abs(v);
    int e = 0, m = 0;   // a = 2^e (1 + m / 2^mbits)
    if (a == 1) { e = 0; m = 0; }
    if (a == 2) { e = 1; m = 0; }
    if (a == 3) { e = 1; m = 1; }   // 1.5 x 2
    if (e5m2) return s | (uint8_t)((e + 15) << 2) | (uint8_t)(m ? 2 : 0);
    return s | (uint8_t)((e + 7) << 3) | (uint8_t)(m ? 4 : 0);
}

template <int FMT>   // 0: fp8 E4M3, 1: bf8 E5M2
__global__ void one(const i32x8 *a, const i32x8 *b, float *d, int scale_a) {
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[threadIdx.x], b[threadIdx.x], c, FMT, FMT, 0, scale_a, 0, 127);
    for (int r = 0; r < 4; ++r) d[(4 * (threadIdx.x >> 4) + r) * 16 + (threadIdx.x & 15)] = c[r];
}

template <int KIND>
__global__ __launch_bounds__(256) void rate(float *out, unsigned long long *cyc, int iters) {
    f32x4 acc[4] = {};
    i32x8 a8, b8;
    f16x8 a16, b16;
    for (int i = 0; i < 8; ++i) { a8[i] = 0x38383838; b8[i] = 0x38383838; a16[i] = (_Float16)1.0f; b16[i] = (_Float16)1.0f; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 64; ++g) {
            if (KIND == 0) acc[g & 3] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, acc[g & 3], 1, 0, 0, 127, 0, 127);
            else acc[g & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a16, b16, acc[g & 3], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    srand(7);
    std::vector<int> A(16 * 128), B(128 * 16);
    for (auto &v : A) v = rand() % 7 - 3;
    for (auto &v : B) v = rand() % 7 - 3;
    std::vector<float> want(256, 0.f);
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            int s = 0;
            for (int k = 0; k < 128; ++k) s += A[i * 128 + k] * B[k * 16 + j];
            want[i * 16 + j] = (float)s;
        }
    i32x8 *da, *db;
    float *dd;
    (void)hipMalloc(&da, 64 * sizeof(i32x8));
    (void)hipMalloc(&db, 64 * sizeof(i32x8));
    (void)hipMalloc(&dd, 256 * sizeof(float));
    for (int fmt = 0; fmt < 2; ++fmt) {
        std::vector<uint8_t> pa(64 * 32), pb(64 * 32);
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 32; ++j) {
                const int r = l & 15, k = 32 * (l >> 4) + j;
                pa[l * 32 + j] = enc(A[r * 128 + k], fmt == 1);
                pb[l * 32 + j] = enc(B[k * 16 + r], fmt == 1);
            }
        (void)hipMemcpy(da, pa.data(), pa.size(), hipMemcpyHostToDevice);
        (void)hipMemcpy(db, pb.data(), pb.size(), hipMemcpyHostToDevice);
        for (int sc : {127, 130}) {
            if (fmt == 0) hipLaunchKernelGGL(one<0>, dim3(1), dim3(64), 0, 0, da, db, dd, sc);
            else hipLaunchKernelGGL(one<1>, dim3(1), dim3(64), 0, 0, da, db, dd, sc);
            std::vector<float> got(256);
            (void)hipMemcpy(got.data(), dd, 256 * sizeof(float), hipMemcpyDeviceToHost);
            int bad = 0;
            const float f = sc == 127 ? 1.f : 8.f;
            for (int i = 0; i < 256; ++i) bad += got[i] != f * want[i];
            printf("{\"format\": \"%s\", \"scale_a\": %d, \"hypothesis\": \"lane (r = l & 15, kq = l >> 4), byte j <-> k = 32 kq + j\", \"wrong_of_256\": %d, \"d[0][0..3]\": [%g, %g, %g, %g], \"want\": [%g, %g, %g, %g]}\n",
                   fmt ? "bf8 E5M2" : "fp8 E4M3", sc, bad, got[0], got[1], got[2], got[3], f * want[0], f * want[1], f * want[2], f * want[3]);
        }
    }
    float *out;
    unsigned long long *c;
    (void)hipMalloc(&out, 256 * 256 * sizeof(float));
    (void)hipMalloc(&c, 256 * sizeof(unsigned long long));
    for (int kind = 0; kind < 2; ++kind) {
        for (int r = 0; r < 2; ++r) {
            if (kind == 0) hipLaunchKernelGGL(rate<0>, dim3(256), dim3(256), 0, 0, out, c, 500);
            else hipLaunchKernelGGL(rate<1>, dim3(256), dim3(256), 0, 0, out, c, 500);
        }
        (void)hipDeviceSynchronize();
        unsigned long long h[256];
        (void)hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
        double m = 0;
        for (int i = 0; i < 256; ++i) m += (double)h[i];
        printf("{\"instruction\": \"%s\", \"cycles_per_instruction\": %.2f}\n", kind == 0 ? "v_mfma_scale_f32_16x16x128_f8f6f4 (bf8 x fp8)" : "v_mfma_f32_16x16x32_f16",
               m / 256.0 / (64.0 * 500));
    }
    return 0;
}
