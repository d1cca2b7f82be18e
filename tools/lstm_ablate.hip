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
#ifdef UVAD_STAMP
    { void *p; hipMalloc(&p, (size_t)tiles * D * 8 * 4 * 8); a.hN = (float *)p; a.cN = (float *)p; }
#endif
    a.G = G; a.ldg = 4 * H * D; a.Whh_packed = W; a.Y = Y; a.ldy = H * D; a.tiles = tiles; a.T = T; a.H = H; a.dirs = D;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    uvad::launch_lstm(a, 0); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) uvad::launch_lstm(a, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s: %.3f ms per launch = %.3f us per step\n", ABL_NAME, ms / 5, ms / 5 / T * 1e3);
#ifdef UVAD_STAMP
    {
        const int waves = uvad::lstm_waves(H);
        std::vector<unsigned long long> st((size_t)tiles * D * waves * 4);
        hipMemcpy(st.data(), a.hN, st.size() * 8, hipMemcpyDeviceToHost);
        for (int blk : {0, 77}) for (int wv = 0; wv < waves; ++wv) {
            const unsigned long long *o = &st[((size_t)blk * waves + wv) * 4];
            printf("  block %3d wave %d: cycles/step  reads+mfma %.0f  gates+write %.0f  barrier+top %.0f  (sum %.0f)\n", blk, wv,
                   o[0] / (double)T, o[1] / (double)T, o[3] / (double)T, (o[0] + o[1] + o[3]) / (double)T);
        }
    }
#endif
    return 0;
}
