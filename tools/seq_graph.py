#!/usr/bin/env python3
"""One cfg-2 batch alone on the GPU (latency form: 4-sequence recurrence, time-chunked layers): the step enqueued launch by launch
against the same step replayed from a hipGraph (uvad_forward is capturable: no allocation, no synchronisation).
    python tools/seq_graph.py [--batch 256] [--reps 20] [--chunks 0]"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--chunks", type=int, default=0)
args = ap.parse_args()
import uvad_amd
from uvad_amd.synth import seed_weights, synth_pcm_device
dev = torch.device("cuda:0")
m = uvad_amd.PyanNet2(encoding_dim=64); m.build(); seed_weights(m, 1234, 4.0)
m.attach_fbank(uvad_amd.FbankConfig(num_filters=64, window_type="hamming")); m = m.to(dev).eval()
rt = m.runtime(dev)
rt.set_time_chunks(args.chunks)
pcm = synth_pcm_device(args.batch, 160000, seed=42, device=dev)
side = torch.cuda.Stream(device=dev)


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); side.synchronize()
    t0 = time.perf_counter()
    e0.record(side)
    for _ in range(args.reps):
        fn()
    e1.record(side)
    side.synchronize()
    return e0.elapsed_time(e1) / args.reps, (time.perf_counter() - t0) / args.reps * 1e3


with torch.cuda.stream(side):
    for _ in range(3):
        want = rt.forward(pcm, want_probs=False)[0].clone()
    side.synchronize()
    used = rt.time_chunks()
    eager_dev, eager_wall = timed(lambda: rt.forward(pcm, want_probs=False))
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        out = rt.forward(pcm, want_probs=False)[0]
    used_graph = rt.time_chunks()
    graph_dev, graph_wall = timed(graph.replay)
    same = bool(torch.equal(out, want))
print(json.dumps({"batch": args.batch, "chunks_used": used, "chunks_in_graph": used_graph, "tile": rt.recurrent_tile(),
                  "eager_ms_device": round(eager_dev, 4), "eager_ms_wall": round(eager_wall, 4),
                  "graph_ms_device": round(graph_dev, 4), "graph_ms_wall": round(graph_wall, 4), "graph_output_identical": same}))
