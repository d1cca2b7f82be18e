// fbank_ablate.hip -- diagnostic: per-stage cycle shares of fbank_kernel (build with -DUVAD_FB_STAMP).
#include "../universal-voice-activity-detection_amd/csrc/fbank.hip"
#include <cmath>
#include <cstdio>
#include <vector>
int main() {
    const int B = 256, F = 64; const long long S = 160000, T = 1000;
    float *pcm, *feats, *win, *melw, *tw; int *st, *ln;
    hipMalloc(&pcm, B * S * 4); hipMalloc(&feats, B * T * F * 4); hipMalloc(&win, 400 * 4);
    std::vector<float> hp(B * S); for (size_t i = 0; i < hp.size(); ++i) hp[i] = 0.1f * std::sin(0.001f * i) + 0.05f * ((i * 2654435761u >> 16) & 0xff) / 255.f;
    hipMemcpy(pcm, hp.data(), hp.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> w(400, 0.5f); hipMemcpy(win, w.data(), 1600, hipMemcpyHostToDevice);
    const int stride = 20;
    std::vector<int> hst(F), hln(F); for (int m = 0; m < F; ++m) { hst[m] = 1 + m * 3; hln[m] = 2 + m * 18 / 63; }
    std::vector<float> hw(F * stride, 0.5f), htw(1024);
    for (int j = 0; j < 512; ++j) { htw[2 * j] = std::cos(2 * M_PI * j / 512); htw[2 * j + 1] = -std::sin(2 * M_PI * j / 512); }
    hipMalloc(&st, F * 4); hipMalloc(&ln, F * 4); hipMalloc(&melw, hw.size() * 4); hipMalloc(&tw, 4096);
    hipMemcpy(st, hst.data(), F * 4, hipMemcpyHostToDevice); hipMemcpy(ln, hln.data(), F * 4, hipMemcpyHostToDevice);
    hipMemcpy(melw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice); hipMemcpy(tw, htw.data(), 4096, hipMemcpyHostToDevice);
    uvad::FbankArgs a{};
    a.pcm = pcm; a.B = B; a.S = S; a.T = T; a.frame_len = 400; a.frame_shift = 160; a.n_mels = F; a.preemph = 0.97f; a.log_floor = 1e-7f;
    a.remove_dc = 1; a.snip_edges = 0; a.feats = feats;
    a.tab.window = win; a.tab.mel_start = st; a.tab.mel_len = ln; a.tab.mel_w = melw; a.tab.mel_stride = stride; a.tab.tw512 = tw;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    uvad::launch_fbank(a, 0); hipDeviceSynchronize();
    hipEventRecord(e0); for (int i = 0; i < 10; ++i) uvad::launch_fbank(a, 0); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s: %.3f ms per launch (B=256 x 10 s) = %.0f M frames/s\n", ABL_NAME, ms / 10, B * T / (ms / 10 * 1e-3) / 1e6);
#ifdef UVAD_FB_STAMP
    unsigned long long h[32]; hipMemcpy(h, feats, sizeof(h), hipMemcpyDeviceToHost);
    const char *nm[7] = {"stage tile+consts", "frame/DC/preemph/win", "pass1+transpose", "pass2+transpose", "pass3+Z->LDS", "split+power", "mel+log+store"};
    for (int wv = 0; wv < 4; ++wv) { printf("  wave %d (5 pairs):", wv); unsigned long long tot = 0; for (int i = 0; i < 7; ++i) tot += h[wv * 8 + i];
        for (int i = 0; i < 7; ++i) printf(" %s %llu", nm[i], h[wv * 8 + i]); printf(" | total %llu cycles\n", tot); }
#endif
    return 0;
}
