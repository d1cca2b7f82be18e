"""Shared error statistics for the parity checks (CPU ORACLE side -- test infrastructure only; imported by
tests/, bench.py's cpu_baseline leg, __graft_entry__.smoke() and tools/err_probe.py, never by the product).

``truth_logits``  -- the classifier (PyanNet2.forward, src/models/segmentation/PyanNet2.py:154-187) evaluated in
                     float64 THROUGHOUT on the f32 weights and features with stock torch CPU operators
                     (nn.LSTM / nn.Linear in double).  Pinned to oracle/uvad_oracle.c: orc_classify_f64 (an independent
                     plain-C double evaluation) by tests/test_oracle.py to ~1e-9.
``truth_sincnet``  -- the SincNet front end (src/models/blocks/sincnet.py:72-103) in float64 THROUGHOUT on the f32 filter bank /
                     conv weights / waveform as given (the filter bank is materialised in f32 exactly as the fp32 path
                     materialises it -- it is a weight -- and then cast): the truth of the PyanNet at-size parity test.
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


@torch.no_grad()
def truth_sincnet(front, wav, chunk=32):
    """front: a torch_ref.TorchSincNet (f32 parameters); wav (B, S) f32.  Returns (B, frames, 60) float64 tensor: every
    operator of TorchSincNet.forward on double tensors (conv1d / max_pool1d / instance_norm / leaky_relu in float64)."""
    import torch.nn.functional as F
    filt = tr.sinc_filters(front.low_hz_, front.band_hz_).double().unsqueeze(1)        # the f32 filter bank, cast
    p = {k: v.detach().double() for k, v in front.state_dict().items()}
    out = []
    wav = torch.as_tensor(wav)
    for i in range(0, wav.shape[0], chunk):
        x = wav[i:i + chunk].double().unsqueeze(1)
        x = F.instance_norm(x, weight=p["wav_norm1d.weight"], bias=p["wav_norm1d.bias"], eps=front.wav_norm1d.eps)
        x = torch.abs(F.conv1d(x, filt, stride=front.stride))
        x = F.leaky_relu(F.instance_norm(F.max_pool1d(x, 3, 3), weight=p["norm1d.0.weight"], bias=p["norm1d.0.bias"], eps=front.norm1d[0].eps))
        for j in range(2):
            x = F.conv1d(x, p[f"conv1d.{j}.weight"], p[f"conv1d.{j}.bias"])
            x = F.leaky_relu(F.instance_norm(F.max_pool1d(x, 3, 3), weight=p[f"norm1d.{j + 1}.weight"], bias=p[f"norm1d.{j + 1}.bias"],
                                             eps=front.norm1d[j + 1].eps))
        out.append(x.transpose(1, 2).contiguous())
    return torch.cat(out)


def error_stats(a, truth, bound=1e-4):
    e = np.abs(np.asarray(a, np.float64) - np.asarray(truth, np.float64)).ravel()
    return {"max": float(e.max()), "mean": float(e.mean()), "rms": float(np.sqrt((e * e).mean())),
            "p99.9": float(np.quantile(e, 0.999)), "frames_over_bound": int((e > bound).sum()), "bound": bound,
            "frames": int(e.size)}


def fmt(name, st):
    return (f"{name}: max {st['max']:.2e} mean {st['mean']:.2e} rms {st['rms']:.2e} p99.9 {st['p99.9']:.2e} "
            f"frames>{st['bound']:.0e}: {st['frames_over_bound']}/{st['frames']}")
