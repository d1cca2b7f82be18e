#!/usr/bin/env python3
"""PyanNet (SincNet front end) on the reference's 5 s cuts (80000 samples -> 293 frames,
src/datasets/custom_vad.py:47): times uvad_sincnet alone and the whole uvad_forward_wav, and checks a slice of the
batch against the torch-CPU restatement.  FLOP accounting of the front end (multiply-add = 2 FLOP), per utterance:
  conv1 7975 x 80 x 251, conv2 2654 x 60 x 400, conv3 880 x 60 x 300  ->  0.479 GFLOP / 5 s cut."""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import uvad_amd
from uvad_amd.synth import synth_pcm_device, seed_weights

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--seconds", type=float, default=5.0)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--check", type=int, default=4, help="utterances compared with the CPU oracle (0 = skip)")
ap.add_argument("--lib", default=None, help="another build of libuvad.so (A/B on one box)")
ap.add_argument("--mode", default="f16p", help="GEMM mode: f16p (split-f16 SincNet stages) or f32 (the exact-f32 stages of sincnet.hip)")
ap.add_argument("--stages", action="store_true", help="per-kernel durations of one uvad_sincnet call (torch profiler, device time)")
args = ap.parse_args()
if args.lib:
    from uvad_amd import _lib
    _lib.LIB_PATH = os.path.abspath(args.lib)
dev = torch.device("cuda:0")
B, S = args.batch, int(args.seconds * 16000)
torch.manual_seed(1234)   # the SincNet convolutions keep torch's default initialisation: seeded, so that two runs time and check the same network
m = uvad_amd.PyanNet()
m.build()
seed_weights(m, 1234, 4.0)   # classifier only; the SincNet front end keeps its mel-spaced initialisation
m = m.to(dev).eval()
rt = m.runtime(dev)
rt.set_gemm_mode(args.mode)
wav = synth_pcm_device(B, S, 1000, dev)
T = rt.sincnet_num_frames(S)


def timed(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / args.reps


ms_front = timed(lambda: rt.sincnet(wav))
ms_all = timed(lambda: rt.forward_wav(wav))
L1 = (S - 251) // 10 + 1; P1 = L1 // 3; L2 = P1 - 4; P2 = L2 // 3; L3 = P2 - 4
flop = 2.0 * (L1 * 80 * 251 + L2 * 60 * 400 + L3 * 60 * 300) * B
out = {"config": f"PyanNet, B={B} x {args.seconds:g} s ({T} frames each), synthetic PCM, default-init SincNet + seeded x4 classifier",
       "sincnet_ms": ms_front, "sincnet_TFLOPs_f32": flop / ms_front / 1e9, "frac_of_157.3_TFLOPs": flop / ms_front / 1e9 / 157.3,
       "forward_wav_ms": ms_all, "frames_per_s": B * T / ms_all * 1e3, "audio_seconds_per_s": B * args.seconds / ms_all * 1e3}
out["gemm_mode"] = args.mode
if args.stages:
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(5):
            rt.sincnet(wav)
        torch.cuda.synchronize()
    st = {}
    for e in prof.key_averages():
        if "uvad" in e.key:
            st[e.key.split("(")[0].replace("void uvad::(anonymous namespace)::", "").replace("uvad::(anonymous namespace)::", "")] = round(e.device_time_total / e.count / 1e3, 4)
    out["stage_ms"] = st
if args.check:
    from oracle import torch_ref as tr
    front = tr.TorchSincNet().eval()
    sd = m.state_dict()
    fsd = {"wav_norm1d.weight": sd["sincnet.wav_norm1d.weight"], "wav_norm1d.bias": sd["sincnet.wav_norm1d.bias"],
           "low_hz_": sd["sincnet.conv1d.0.filterbank.low_hz_"], "band_hz_": sd["sincnet.conv1d.0.filterbank.band_hz_"]}
    for i in range(3):
        for p in ("weight", "bias"):
            fsd[f"norm1d.{i}.{p}"] = sd[f"sincnet.norm1d.{i}.{p}"]
    for i in range(2):
        for p in ("weight", "bias"):
            fsd[f"conv1d.{i}.{p}"] = sd[f"sincnet.conv1d.{i + 1}.{p}"]
    front.load_state_dict({k: v.cpu() for k, v in fsd.items()})
    cls = tr.TorchPyanNet2(60)
    cls.load_state_dict({k: v.cpu() for k, v in sd.items() if not k.startswith("sincnet.")})
    n = args.check
    feats = front(wav[:n].cpu().unsqueeze(1)).transpose(1, 2).contiguous()
    want, _ = cls(feats)
    got_f = rt.sincnet(wav)[:n].cpu()
    got, _ = rt.forward_wav(wav)
    out["max_abs_feature_err"] = float((got_f - feats).abs().max())
    out["max_abs_logit_err"] = float((got[:n].cpu() - want).abs().max())
    out["logit_range"] = [float(want.min()), float(want.max())]
print(json.dumps(out))
