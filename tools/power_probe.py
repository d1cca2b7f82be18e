#!/usr/bin/env python3
"""What the GPU sustains at its power limit: shader clock and socket power (sysfs hwmon, sampled every 20 ms by a thread) while
  (a) a matrix-pipe-only kernel (tools/mfma_burner.hip kind 5: v_mfma_f32_32x32x16_f16 back to back, no memory) fills every CU,
  (b) cfg-2 steps run 12 in flight (the bench's timed region).
    python tools/power_probe.py"""
import ctypes, glob, json, os, sys, threading, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from gpu_power import PowerSampler, hwmon_of, bdf_of_torch_device
dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
paths = hwmon_of(bdf_of_torch_device(0))
print("sensors of this GPU:", json.dumps(paths), flush=True)

class Sampler(PowerSampler):
    def __init__(self, paths): super().__init__(paths)
    @property
    def stop(self): return self._stop_flag
    @stop.setter
    def stop(self, v): self._stop_flag = v
    def summary(self, skip=0.3):
        t0, t1 = self.rows[0][0], self.rows[-1][0]
        r = PowerSampler.summary(self, t0 + skip * (t1 - t0), None)
        return {"samples": r["samples"], "sclk_mhz": r["sclk_mhz"], "power_w": r["socket_power_w"], "cap_w": r["power_cap_w"]}

lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libburner.so"))
lib.burner_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
sink = torch.empty(4096 * 256, device=dev)
st = torch.cuda.current_stream(dev).cuda_stream

def burn(kind, blocks, iters, secs, flop_per_iter_per_block):
    lib.burner_launch(kind, sink.data_ptr(), blocks, iters, st); torch.cuda.synchronize()
    s = Sampler(paths); s.start()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 0; t0 = time.time(); e0.record()
    while time.time() - t0 < secs:
        for _ in range(8): lib.burner_launch(kind, sink.data_ptr(), blocks, iters, st); n += 1
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize(); s.stop = True; s.join()
    ms = e0.elapsed_time(e1)
    r = s.summary(); r["TFLOPs"] = round(n * blocks * iters * flop_per_iter_per_block / (ms * 1e-3) / 1e12, 1) if flop_per_iter_per_block else None
    return r

# kind 5: 4 waves per block, 8 MFMAs of 32x32x16 (32768 FLOP) per iteration and wave
for blocks in (256, 512, 1024):
    print(json.dumps({"case": f"mfma_only blocks={blocks}", **burn(5, blocks, 20000, 2.0, 4 * 8 * 32768)}), flush=True)
print(json.dumps({"case": "valu_only blocks=1024", **burn(4, 1024, 4000, 1.5, 0)}), flush=True)

import uvad_amd
from uvad_amd.synth import seed_weights, synth_pcm_device
from uvad_amd.pipeline import ForwardPipeline
m = uvad_amd.PyanNet2(encoding_dim=64); m.build(); seed_weights(m, 1234, 4.0)
m.attach_fbank(uvad_amd.FbankConfig(num_filters=64, window_type="hamming")); m = m.to(dev).eval()
pcm = synth_pcm_device(256, 160000, seed=42, device=dev)
for mode in ("f16p", "f16p3", "f32"):
    pipe = ForwardPipeline(m, dev, depth=12, recurrent_tile=16)
    for r in pipe.runtimes: r.set_gemm_mode(mode)
    def run(nsteps):
        pend = []
        for k in range(nsteps):
            if len(pend) >= 12: pend.pop(0).result()
            pend.append(pipe.submit(pcm, want_probs=False))
        for p in pend: p.result()
    run(24); torch.cuda.synchronize()
    s = Sampler(paths); s.start(); t0 = time.time(); run(500); torch.cuda.synchronize(); dt = time.time() - t0; s.stop = True; s.join()
    print(json.dumps({"case": f"cfg-2 steps, 12 in flight, mode {mode}", **s.summary(), "ms_per_step": round(dt / 500 * 1e3, 3)}), flush=True)
    pipe.close()
print("POWER_PROBE_DONE")
