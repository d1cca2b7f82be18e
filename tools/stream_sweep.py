#!/usr/bin/env python3
"""A longer version of test_streaming_random_chunk_sweep_equals_offline: N random (chunk, B, hidden, layers, F, head) set-ups of a causal
model, streaming == offline frame for frame.  python tools/stream_sweep.py [--cases 60] [--seed 7]"""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import uvad_amd
from uvad_amd.synth import synth_pcm, seed_weights
ap = argparse.ArgumentParser(); ap.add_argument("--cases", type=int, default=60); ap.add_argument("--seed", type=int, default=7)
args = ap.parse_args()
dev = torch.device("cuda:0")
rng = np.random.default_rng(args.seed)
worst = 0.0
for case in range(args.cases):
    chunk = int(rng.choice([160, 200, 320, 333, 480, 640, 700, 1024, 4800]))
    B = int(rng.choice([1, 2, 3, 4, 5, 8, 13, 64]))
    H = int(rng.choice([64, 128, 128, 128]))
    Lyr = int(rng.integers(1, 5))
    F = int(rng.choice([40, 64, 64, 80]))
    lin = [None, {"hidden_size": 128, "num_layers": 1}, {"hidden_size": 128, "num_layers": 3}, {"hidden_size": 64, "num_layers": 2}, {"num_layers": 0}][int(rng.integers(0, 5))]
    steps = max(3, int(24000 // chunk))
    S = steps * chunk
    pcm = synth_pcm(B, S, seed=900 + case)
    kw = {"lstm": {"bidirectional": False, "hidden_size": H, "num_layers": Lyr}, "encoding_dim": F}
    if lin is not None: kw["linear"] = lin
    m = uvad_amd.PyanNet2(**kw); m.build(); seed_weights(m, 70 + case, 3.0)
    m.attach_fbank(uvad_amd.FbankConfig(num_filters=F, window_type=["povey", "hamming"][case & 1]))
    m = m.to(dev).eval(); rt = m.runtime(dev)
    x = torch.from_numpy(pcm).to(dev)
    offline, _ = rt.forward(x)
    st = rt.stream_open(B, chunk, graphs=bool(case % 3 == 0))
    outs = [rt.stream_step(st, x[:, i * chunk:(i + 1) * chunk].contiguous()).clone() for i in range(steps)]
    got = torch.cat(outs, dim=1); n = got.shape[1]
    assert n == max(0, (S + 120 - 400) // 160 + 1), (case, chunk, n)
    err = float((got - offline[:, :n]).abs().max()) if n else 0.0
    worst = max(worst, err)
    print(f"case {case}: chunk {chunk} B {B} H {H} L {Lyr} F {F} lin {lin}: {n} frames, max diff {err:.2e}", flush=True)
    assert err < 1e-4, (case, err)
    rt.close()
print(f"STREAM_SWEEP_OK worst {worst:.2e}")
