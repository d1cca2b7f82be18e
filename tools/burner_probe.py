#!/usr/bin/env python3
"""fbank_kernel beside a synthetic MFMA burner (tools/mfma_burner.hip): python tools/burner_probe.py"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import uvad_amd
from uvad_amd import _lib
if len(sys.argv) > 1 and sys.argv[1].endswith('.so'): _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from uvad_amd.synth import seed_weights, synth_pcm_device
dev = torch.device("cuda:0")
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libburner.so"))
lib.burner_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
m = uvad_amd.PyanNet2(encoding_dim=64); m.build(); seed_weights(m, 1234, 4.0)
m.attach_fbank(uvad_amd.FbankConfig(num_filters=64, window_type="hamming")); m = m.to(dev).eval()
rt = m.runtime(dev)
pcm = synth_pcm_device(256, 160000, seed=42, device=dev)
want = rt.fbank(pcm).clone()
sink = torch.empty(8192 * 256, device=dev)
sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
assert rt.streams_overlap(sa, sb), "streams share a queue: rerun"
feats = want
x = torch.randn(64 * 1024 * 1024 // 4, device=dev)
victims = {
    "fbank": lambda: rt.fbank(pcm),
    "classify": lambda: rt.classify(feats, want_probs=False)[0],
    "torch.sin": lambda: torch.sin(x),
    "torch.fft.rfft": lambda: torch.view_as_real(torch.fft.rfft(x.view(-1, 512))),
}
mw = uvad_amd.PyanNet(); mw.build(); seed_weights(mw, 1234, 4.0); mw = mw.to(dev).eval()
rtw = mw.runtime(dev)
wav = synth_pcm_device(256, 80000, seed=7, device=dev)
victims["sincnet"] = lambda: rtw.sincnet(wav)
victims["forward_wav"] = lambda: rtw.forward_wav(wav, want_probs=False)[0]   # SincNet (64-bit LDS reads) + the classifier, as one pipeline step
# a streaming step of a causal model (fbank on virtual rows + lstm_stack_kernel: the whole stack and the head in one launch), state reset every time
mc = uvad_amd.PyanNet2(lstm={"bidirectional": False}, encoding_dim=64); mc.build(); seed_weights(mc, 1234, 4.0)
mc.attach_fbank(uvad_amd.FbankConfig(num_filters=64, window_type="hamming")); mc = mc.to(dev).eval()
rtc = mc.runtime(dev)
chunks = synth_pcm_device(512, 3 * 320, seed=9, device=dev)
def stream_victim():
    st = rtc.stream_open(512, 320, graphs=False)
    return torch.cat([rtc.stream_step(st, chunks[:, i * 320:(i + 1) * 320].contiguous()).clone() for i in range(3)], dim=1)
victims["stream_step"] = stream_victim
rt2 = uvad_amd.VadRuntime(device=dev, fbank=m._fbank_cfg, model={"encoding_dim": 64, "lstm": m.hparams.lstm, "linear": m.hparams.linear})
rt2.load_state_dict(m.state_dict())
rt2.set_recurrent_tile(16)
A16 = torch.randn(8192, 8192, device=dev, dtype=torch.float16)
B16 = torch.randn(8192, 8192, device=dev, dtype=torch.float16)
def aggressor(kind):
    if kind == "burner": lib.burner_launch(0, sink.data_ptr(), 8192, 400, sb.cuda_stream)
    elif kind == "matmul": torch.matmul(A16, B16)       # the stock f16 GEMM of the installed BLAS (MFMA + LDS + barriers)
    elif kind == "step":                                # a whole cfg-2 step of ANOTHER context (its split-f16 GEMMs and recurrences) in flight
        rt2.forward(pcm, want_probs=False)
only = [a for a in sys.argv[1:] if not a.endswith('.so')]
total_bad = {}
for vname, fn in victims.items():
    if only and vname not in only: continue
    ref = fn().clone()
    torch.cuda.synchronize(dev)
    for kind in ("none", "burner", "matmul", "step"):
        bad = 0
        for rep in range(30):
            with torch.cuda.stream(sb):
                aggressor(kind)
            with torch.cuda.stream(sa):
                outs = [fn() for _ in range(3)]
            torch.cuda.synchronize(dev)
            bad += sum(int(not torch.equal(o, ref)) for o in outs)
        print(f"victim {vname} beside '{kind}': {bad} wrong of 90", flush=True)
        total_bad[vname] = total_bad.get(vname, 0) + bad
print("SUMMARY", __import__("json").dumps(total_bad))
