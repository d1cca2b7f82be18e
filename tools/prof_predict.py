import sys, os, time, wave, tempfile, cProfile, pstats, io, contextlib
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from config.config import load_config
from uvad_amd.scripts import predict_vad
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(9)
hour = (torch.randn(3600 * 16000, generator=g, device=dev) * 0.1).clamp_(-1, 1)
q = (hour * 32767.0).round().to(torch.int16).cpu().numpy()
td = tempfile.mkdtemp()
path = os.path.join(td, "hour.wav")
with wave.open(path, "wb") as w:
    w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(q.tobytes())
cfg = load_config(); cfg.input.kind, cfg.input.paths = "wav", [path]
with contextlib.redirect_stdout(sys.stderr):
    predict_vad(**cfg)
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
with contextlib.redirect_stdout(sys.stderr):
    predict_vad(**cfg)
torch.cuda.synchronize()
pr.disable()
print("wall", time.perf_counter() - t0)
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:5000])
