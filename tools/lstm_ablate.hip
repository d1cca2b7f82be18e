// lstm_ablate.hip -- diagnostic timing of the recurrent kernel with parts removed (-DUVAD_ABL_*).
// Includes the product kernel source unchanged; outputs of ablated builds are meaningless.
#include "../universal-voice-activity-detection_amd/csrc/lstm.hip"
#include <cstdio>
#include <vector>
int main() {
    const int tiles = 64, T = 1000, H = 128, D = 2;
    const size_t M = (size_t)tiles * 4 * T;
    float *G, *Y, *W;
    hipMalloc(&G, M * 4 * H * D * 4); hipMalloc(&Y, M * H * D * 4); hipMalloc(&W, (size_t)D * 4 * H * H * 4);
    hipMemset(G, 0, M * 4 * H * D * 4);
    std::vector<float> w((size_t)D * 4 * H * H, 0.01f);
    hipMemcpy(W, w.data(), w.size() * 4, hipMemcpyHostToDevice);
    uvad::LstmArgs a{};
    a.G = G; a.ldg = 4 * H * D; a.Whh_packed = W; a.Y = Y; a.ldy = H * D; a.tiles = tiles; a.T = T; a.H = H; a.dirs = D;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    uvad::launch_lstm(a, 0); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) uvad::launch_lstm(a, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s: %.3f ms per launch = %.3f us per step\n", ABL_NAME, ms / 5, ms / 5 / T * 1e3);
    return 0;
}
