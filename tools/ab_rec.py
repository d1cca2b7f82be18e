#!/usr/bin/env python3
"""A/B of the 16-sequence recurrence forms in one experiment library (UVAD_W16 = 8 | 4 read at launch): bit-compare + per-layer times."""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
ap = argparse.ArgumentParser()
ap.add_argument("--lib", required=True)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--reps", type=int, default=8)
ap.add_argument("--forms", default="8,4")
ap.add_argument("--mode", default="f16p")
args = ap.parse_args()
import uvad_amd
from uvad_amd import _lib
_lib.LIB_PATH = os.path.abspath(args.lib)
from uvad_amd.synth import seed_weights, synth_pcm_device
dev = torch.device("cuda:0")
m = uvad_amd.PyanNet2(encoding_dim=64); m.build(); seed_weights(m, 1234, 4.0)
m.attach_fbank(uvad_amd.FbankConfig(num_filters=64, window_type="hamming")); m = m.to(dev).eval()
rt = m.runtime(dev)
rt.set_recurrent_tile(16)
rt.set_gemm_mode(args.mode)
pcm = synth_pcm_device(args.batch, 160000, seed=42, device=dev)
ref = None
for form in args.forms.split(","):
    os.environ["UVAD_W16"] = form
    for _ in range(2):
        out = rt.forward(pcm, want_probs=False)[0]
    torch.cuda.synchronize()
    if ref is None:
        ref = out.clone()
    same = bool(torch.equal(out, ref))
    rt.set_timing(True)
    rec = []
    for _ in range(args.reps):
        rt.forward(pcm, want_probs=False)
        rec.append([b for a, b in rt.layer_timing_ms()])
    rt.set_timing(False)
    import statistics
    per = [round(statistics.median(r[k] for r in rec), 4) for k in range(4)]
    print(json.dumps({"form": form, "identical_to_first": same, "max_diff": float((out - ref).abs().max()), "rec_ms_per_layer": per, "finite": bool(torch.isfinite(out).all())}), flush=True)
