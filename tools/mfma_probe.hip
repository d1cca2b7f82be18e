// mfma_probe.hip -- prints/validates the lane<->element maps of the f32 MFMA forms libuvad relies on.
// Build: hipcc --offload-arch=gfx950 -O2 tools/mfma_probe.hip -o tools/mfma_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

__global__ void k4(const float *A, const float *B, float *D) {  // 4x4x1, 16 blocks
    const int l = threadIdx.x;
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(A[l], B[l], c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[l * 4 + r] = c[r];
}
__global__ void k32(const float *A, const float *B, float *D) {  // 32x32x2
    const int l = threadIdx.x;
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0;
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(A[l], B[l], c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[l * 16 + r] = c[r];
}
int main() {
    float *dA, *dB, *dD;
    hipMalloc(&dA, 64 * 4); hipMalloc(&dB, 64 * 4); hipMalloc(&dD, 64 * 16 * 4);
    std::vector<float> A(64), B(64), D(64 * 16);
    // 4x4x1: assumed A lane l = A_blk[i=l%4], B lane l = B_blk[j=l%4], D reg r lane l = D_blk[i=r][j=l%4]
    for (int l = 0; l < 64; ++l) { A[l] = 1 + l; B[l] = 100 + 3 * l; }
    hipMemcpy(dA, A.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k4, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(D.data(), dD, 64 * 4 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const int blk = l / 4, j = l % 4;
            const float want = A[blk * 4 + r] * B[blk * 4 + j];
            if (D[l * 4 + r] != want) { if (bad < 8) printf("4x4x1 mismatch lane %d reg %d got %g want %g\n", l, r, D[l * 4 + r], want); ++bad; }
        }
    printf("mfma_f32_4x4x1f32 assumed map (A: lane=4b+i, B: lane=4b+j, D: reg=i lane=4b+j): %s\n", bad ? "FAIL" : "PASS");
    // 32x32x2: A lane l = A[i=l&31][k=l>>5], B lane l = B[k=l>>5][j=l&31], D reg r lane l: row=(r&3)+8*(r>>2)+4*(l>>5), col=l&31
    hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(D.data(), dD, 64 * 16 * 4, hipMemcpyDeviceToHost);
    int bad2 = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
            const float want = A[row] * B[col] + A[32 + row] * B[32 + col];
            if (D[l * 16 + r] != want) { if (bad2 < 8) printf("32x32x2 mismatch lane %d reg %d got %g want %g\n", l, r, D[l * 16 + r], want); ++bad2; }
        }
    printf("mfma_f32_32x32x2f32 assumed map: %s\n", bad2 ? "FAIL" : "PASS");
    return (bad || bad2) ? 1 : 0;
}
