// gemm_ablate.hip -- diagnostic timing of gemm_f32_kernel on the layer-1..3 projection shape.
#include "../universal-voice-activity-detection_amd/csrc/gemm.hip"
#include <cstdio>
int main() {
    const int M = 256000, N = 1024, K = 256;
    float *A, *W, *b, *C;
    hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&W, (size_t)N * K * 4); hipMalloc(&b, N * 4); hipMalloc(&C, (size_t)M * N * 4);
    hipMemset(A, 0, (size_t)M * K * 4); hipMemset(W, 0, (size_t)N * K * 4); hipMemset(b, 0, N * 4);
    uvad::GemmArgs a{};
    a.A = A; a.W = W; a.ldw = K; a.bias = b; a.C = C; a.M = M; a.N = N; a.K = K; a.lda = K; a.ldc = N; a.a_mode = 0; a.B = 256; a.T = 1000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    uvad::launch_gemm(a, 0); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) uvad::launch_gemm(a, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s: %.3f ms per launch = %.1f TFLOP/s (zero-filled operands read high)\n", ABL_NAME, ms / 5, 2.0 * M * N * K / (ms / 5 * 1e-3) / 1e12);
    return 0;
}
