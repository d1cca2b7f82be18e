// gemm_f16x3_ablate.hip -- diagnostic timing / phase stamps of gemm_f16x3_kernel on the layer-1..3 projection shape.
#include "../universal-voice-activity-detection_amd/csrc/gemm_f16x3.hip"
#include <cstdio>
namespace uvad { int gemm_padded_k(int K) { return (K + 31) / 32 * 32; } }
int main() {
    const int M = 256000, N = 1024, K = 256;
    float *A, *b, *C; unsigned short *Ws;
    hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&Ws, (size_t)2 * N * K * 2); hipMalloc(&b, N * 4); hipMalloc(&C, (size_t)M * N * 4);
    hipMemset(A, 0, (size_t)M * K * 4); hipMemset(Ws, 0, (size_t)2 * N * K * 2); hipMemset(b, 0, N * 4);
    uvad::GemmArgs a{};
    a.A = A; a.W = nullptr; a.Wsplit16 = Ws; a.ldw = K; a.bias = b; a.C = C; a.M = M; a.N = N; a.K = K; a.lda = K; a.ldc = N; a.a_mode = 0; a.B = 256; a.T = 1000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    uvad::launch_gemm_f16x3(a, 0); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) uvad::launch_gemm_f16x3(a, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("gemm_f16x3 %s: %.3f ms per launch = %.1f f32-equivalent TFLOP/s (zero-filled operands read high)\n", ABL_NAME, ms / 5, 2.0 * M * N * K / (ms / 5 * 1e-3) / 1e12);
#ifdef UVAD_GS_STAMP
    unsigned long long h[32]; hipMemcpy(h, C, sizeof(h), hipMemcpyDeviceToHost);
    const char *nm[8] = {"prologue", "issue loads", "reads+MFMA", "barrier1", "wait+split+store", "barrier2", "-", "epilogue"};
    for (int wv = 0; wv < 4; ++wv) { printf("  wave %d:", wv); unsigned long long tot = 0; for (int i = 0; i < 8; ++i) tot += h[wv * 8 + i];
        for (int i = 0; i < 8; ++i) if (i != 6) printf(" %s %llu", nm[i], h[wv * 8 + i]); printf(" | total %llu cycles (8 K-steps)\n", tot); }
#endif
    return 0;
}
