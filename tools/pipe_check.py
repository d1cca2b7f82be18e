#!/usr/bin/env python3
"""The bench regime checked for correctness: N steps of the SAME batch through ForwardPipeline (depth steps in flight, the
throughput recurrence) -- every step's logits must equal the ones of a single call bit for bit.
    GPU_MAX_HW_QUEUES=16 python tools/pipe_check.py [--lib build.so] [--depth 12] [--steps 96] [--batch 256] [--tile 16]"""
import argparse, json, os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--depth", type=int, default=12)
ap.add_argument("--steps", type=int, default=96)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--seconds", type=float, default=10.0)
ap.add_argument("--tile", type=int, default=16)
ap.add_argument("--no-dc", action="store_true")
ap.add_argument("--stage", default="forward", choices=["forward", "classify", "fbank"])
ap.add_argument("--mode", default="f16p", choices=["f16p", "f16p3", "f32", "f16p_stream"], help="GEMM mode of every context")
args = ap.parse_args()
import uvad_amd
from uvad_amd import _lib
if args.lib:
    _lib.LIB_PATH = os.path.abspath(args.lib)
from uvad_amd.synth import seed_weights, synth_pcm_device
dev = torch.device("cuda:0")
m = uvad_amd.PyanNet2(encoding_dim=64); m.build(); seed_weights(m, 1234, 4.0)
m.attach_fbank(uvad_amd.FbankConfig(num_filters=64, window_type="hamming", remove_dc_offset=not args.no_dc)); m = m.to(dev).eval()
pcm = synth_pcm_device(args.batch, int(args.seconds * 16000), seed=42, device=dev)
rt = m.runtime(dev)
rt.set_recurrent_tile(args.tile)
rt.set_gemm_mode(args.mode)
feats = rt.fbank(pcm)
def run_one(r):
    if args.stage == "forward": return r.forward(pcm, want_probs=False)[0]
    if args.stage == "classify": return r.classify(feats, want_probs=False)[0]
    return r.fbank(pcm)
want = run_one(rt).clone()
assert torch.equal(want, run_one(rt))
pipe = uvad_amd.ForwardPipeline(m, dev, depth=args.depth, recurrent_tile=args.tile)
for r in pipe.runtimes: r.set_gemm_mode(args.mode)
bad, worst, nseq = 0, 0.0, 0
for base in range(0, args.steps, args.depth):
    n = min(args.depth, args.steps - base)
    outs = []
    for k in range(n):
        i = k % args.depth
        sidx = pipe.streams[i]
        sidx.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(sidx):
            outs.append(run_one(pipe.runtimes[i]))
    torch.cuda.synchronize(dev)
    for got in outs:
        if not torch.equal(got, want):
            bad += 1
            d = (got - want).abs()
            worst = max(worst, float(d.max()))
            nseq += int((d.reshape(d.shape[0], -1).max(dim=1).values > 0).sum())
pipe.close()
print(json.dumps({"lib": args.lib or "default", "depth": args.depth, "tile": args.tile, "batch": args.batch, "steps": args.steps, "stage": args.stage, "mode": args.mode,
                  "steps_with_wrong_logits": bad, "sequences_affected": nseq, "worst_abs_diff": worst}))
sys.exit(1 if bad else 0)
