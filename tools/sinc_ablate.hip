// sinc_ablate.hip -- diagnostic timing of conv_pool_kernel on the three SincNet stages at B = 256 x 5 s.
// Build variants with -DUVAD_SN_ABL_NOK / -DUVAD_SN_ABL_NOEPI to see what the K loop and the epilogue cost.
#include "../universal-voice-activity-detection_amd/csrc/sincnet.hip"
#include <cstdio>
#include <vector>
int main() {
    const int B = 256;
    const int cin[3] = {1, 80, 60}, cout[3] = {80, 60, 60}, kw[3] = {251, 5, 5}, stride[3] = {10, 1, 1};
    int L = 80000;
    for (int i = 0; i < 3; ++i) {
        const int Lconv = (L - kw[i]) / stride[i] + 1, Lpool = Lconv / 3, NW = (cout[i] + 31) / 32 * 32;
        const int Ktot = cin[i] * kw[i], Kp = (Ktot + 7) / 8 * 8;
        uvad::SincConvArgs pa{};
        pa.Cin = cin[i]; pa.Cout = cout[i]; pa.Kw = kw[i]; pa.stride = stride[i]; pa.Ktot = Ktot; pa.Kp = Kp;
        const uvad::SincConvPlan plan = uvad::sinc_conv_plan(pa);
        const int ntiles = (Lpool + plan.pt - 1) / plan.pt;
        float *in, *sc, *sh, *wt, *bias, *out, *part;
        hipMalloc(&in, (size_t)B * cin[i] * L * 4); hipMemset(in, 0, (size_t)B * cin[i] * L * 4);
        hipMalloc(&sc, (size_t)B * cin[i] * 4); hipMemset(sc, 0, (size_t)B * cin[i] * 4);
        hipMalloc(&sh, (size_t)B * cin[i] * 4); hipMemset(sh, 0, (size_t)B * cin[i] * 4);
        hipMalloc(&wt, (size_t)Kp * NW * 4); hipMemset(wt, 0, (size_t)Kp * NW * 4);
        hipMalloc(&bias, NW * 4); hipMemset(bias, 0, NW * 4);
        hipMalloc(&out, (size_t)B * cout[i] * Lpool * 4);
        hipMalloc(&part, (size_t)B * ntiles * plan.phases * NW * 2 * 4);
        uvad::SincConvArgs a{};
        a.in = in; a.in_bstride = (long long)cin[i] * L; a.Cin = cin[i]; a.Lin = L; a.in_scale = sc; a.in_shift = sh; a.in_lrelu = i > 0; a.slope = 0.01f;
        a.Wt2 = wt; a.bias = bias; a.Kw = kw[i]; a.stride = stride[i]; a.Ktot = Ktot; a.Kp = Kp; a.Cout = cout[i]; a.do_abs = i == 0;
        a.Lconv = Lconv; a.Lpool = Lpool; a.ntiles = ntiles; a.out = out; a.partials = part; a.B = B;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipError_t e = uvad::launch_sinc_conv(a, 0); hipDeviceSynchronize();
        if (e != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(e)); return 1; }
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) uvad::launch_sinc_conv(a, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        ms /= 5;
        const double flop = 2.0 * B * (double)Lconv * cout[i] * Ktot;
        const double tiles_per_cu = (double)B * ntiles / 256.0;
        printf("%s stage %d (%d waves): %.3f ms = %.1f TFLOP/s (%.2f us per tile-slot; MFMA-only bound %.2f us at 2.4 GHz)\n", ABL_NAME, i, plan.waves, ms,
               flop / ms / 1e9, ms * 1e3 / tiles_per_cu, (Kp / 2) * ((cout[i] + 31) / 32) * 64 / 2400.0 * (plan.waves == 8 ? 2 : 1));
        hipFree(in); hipFree(sc); hipFree(sh); hipFree(wt); hipFree(bias); hipFree(out); hipFree(part);
        L = Lpool;
    }
    return 0;
}
