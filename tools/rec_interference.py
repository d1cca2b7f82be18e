#!/usr/bin/env python3
"""Does a recurrence launch (32 workgroups, one per CU) slow down because of what the OTHER CUs do?  Per-layer recurrence times of one
cfg-2 step alone, beside a matrix-pipe-only neighbour on 192 CUs, beside HBM writers, beside HBM readers (tools/mfma_burner.hip).
    python tools/rec_interference.py"""
import ctypes, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import uvad_amd
from uvad_amd.synth import seed_weights, synth_pcm_device
dev = torch.device("cuda:0")
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libburner.so"))
lib.burner_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
m = uvad_amd.PyanNet2(encoding_dim=64); m.build(); seed_weights(m, 1234, 4.0)
m.attach_fbank(uvad_amd.FbankConfig(num_filters=64, window_type="hamming")); m = m.to(dev).eval()
rt = m.runtime(dev)
rt.set_recurrent_tile(16)
pcm = synth_pcm_device(256, 160000, seed=42, device=dev)
feats = rt.fbank(pcm).clone()
sink = torch.empty(192 * (4 << 20) // 4, device=dev)
sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
for _ in range(3): rt.classify(feats, want_probs=False)
torch.cuda.synchronize()
rt.set_timing(True)
NB = 192
cases = [("alone", None, 0), ("mfma_only x192", 5, 60000), ("hbm_writers x192", 10, 40), ("hbm_readers x192", 11, 40), ("valu_only x192", 4, 60000)]
for name, kind, iters in cases:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if kind is not None:
        with torch.cuda.stream(sb):
            e0.record(sb)
            lib.burner_launch(kind, sink.data_ptr(), NB, iters, sb.cuda_stream)
            e1.record(sb)
    with torch.cuda.stream(sa):
        rt.classify(feats, want_probs=False)
    torch.cuda.synchronize()
    lt = rt.layer_timing_ms()
    print(json.dumps({"case": name, "neighbour_ms": round(e0.elapsed_time(e1), 2) if kind is not None else None,
                      "proj_ms": [round(a, 3) for a, _ in lt], "rec_ms": [round(b, 3) for _, b in lt]}), flush=True)
print("REC_INTERFERENCE_DONE")
