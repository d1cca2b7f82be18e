"""Shared error statistics for the parity checks (CPU ORACLE side -- test infrastructure only; imported by
tests/, bench.py's cpu_baseline leg, __graft_entry__.smoke() and tools/err_probe.py, never by the product).

``truth_logits``  -- the classifier (PyanNet2.forward, src/models/segmentation/PyanNet2.py:154-187) evaluated in
                     float64 THROUGHOUT on the f32 weights and features with stock torch CPU operators
                     (nn.LSTM / nn.Linear in double).  Pinned to oracle/uvad_oracle.c: orc_classify_f64 (an independent
                     plain-C double evaluation) by tests/test_oracle.py to ~1e-9.
``error_stats``   -- max / mean / rms / p99.9 of |a - truth| and the number of frames above a bound.
"""
import numpy as np
import torch

from . import torch_ref as tr


def truth_logits(state_dict, feats, encoding_dim, hidden=128, num_layers=4, bidirectional=True, lin_hidden=128,
                 lin_layers=2, leaky_slope=0.01, threads=None):
    """state_dict: torch-keyed f32 tensors; feats (B, T, F) f32 tensor or array.  Returns (B, T) float64 numpy."""
    if threads:
        torch.set_num_threads(int(threads))
    m = tr.TorchPyanNet2(encoding_dim, hidden, num_layers, bidirectional, lin_hidden, lin_layers, leaky_slope).double()
    m.load_state_dict({k: torch.as_tensor(v).detach().cpu().double() for k, v in state_dict.items()})
    x = torch.as_tensor(feats).detach().cpu().double()
    return m(x)[0].numpy()


def error_stats(a, truth, bound=1e-4):
    e = np.abs(np.asarray(a, np.float64) - np.asarray(truth, np.float64)).ravel()
    return {"max": float(e.max()), "mean": float(e.mean()), "rms": float(np.sqrt((e * e).mean())),
            "p99.9": float(np.quantile(e, 0.999)), "frames_over_bound": int((e > bound).sum()), "bound": bound,
            "frames": int(e.size)}


def fmt(name, st):
    return (f"{name}: max {st['max']:.2e} mean {st['mean']:.2e} rms {st['rms']:.2e} p99.9 {st['p99.9']:.2e} "
            f"frames>{st['bound']:.0e}: {st['frames_over_bound']}/{st['frames']}")
