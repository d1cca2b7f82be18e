"""Synthetic inputs and seeded weights (there are no corpora or checkpoints in the reference).

``synth_pcm``   -- SURVEY.md 8(d): 0.1*N(0,1) noise plus five AM-modulated 100-300 Hz harmonic
                   bursts per utterance, clipped to [-1, 1]; utterance i uses seed + i, so any
                   rank can regenerate exactly its shard without communication.
``seed_weights``-- deterministic weights for a PyanNet2: uniform(-1/sqrt(fan), 1/sqrt(fan)) like
                   torch's default init, times ``scale`` (x4 makes the outputs span 0.03..0.99
                   instead of the ~0.506 flat line of the default init).
"""
import math

import numpy as np
import torch


def synth_pcm(B: int, S: int, seed: int = 1000, sample_rate: int = 16000, first: int = 0) -> np.ndarray:
    out = np.empty((B, S), np.float32)
    t = np.arange(S, dtype=np.float64) / sample_rate
    dur = S / sample_rate
    for i in range(B):
        rng = np.random.default_rng(seed + first + i)
        x = 0.1 * rng.standard_normal(S)
        for _ in range(5):
            f0 = rng.uniform(100.0, 300.0)
            st = rng.uniform(0.0, max(dur - 0.5, 0.1))
            ln = rng.uniform(0.5, 3.0)
            env = ((t >= st) & (t < st + ln)) * (0.5 + 0.5 * np.sin(2 * np.pi * 4.0 * (t - st)))
            sig = sum(np.sin(2 * np.pi * f0 * h * t + rng.uniform(0, 6.28)) / h for h in range(1, 6))
            x = x + 0.15 * env * sig
        out[i] = np.clip(x, -1.0, 1.0).astype(np.float32)
    return out


def synth_pcm_device(B: int, S: int, seed: int, device, first: int = 0) -> torch.Tensor:
    """Cheap on-device generator for throughput runs (BASELINE cfg 3/4: corpora far larger than host
    memory): per-utterance Philox stream keyed by (seed, utterance id), 0.1*N(0,1) noise plus one
    200 Hz tone burst.  Reproducible for a given utterance id regardless of how the batch is sharded."""
    out = torch.empty((B, S), dtype=torch.float32, device=device)
    t = torch.arange(S, device=device, dtype=torch.float32) / 16000.0
    for i in range(B):
        g = torch.Generator(device=device)
        g.manual_seed(seed * 1_000_003 + first + i)
        x = 0.1 * torch.randn(S, generator=g, device=device)
        st = float(torch.rand(1, generator=g, device=device)) * max(S / 16000.0 - 1.0, 0.1)
        env = ((t >= st) & (t < st + 1.0)).float()
        out[i] = torch.clamp(x + 0.2 * env * torch.sin(2 * math.pi * 200.0 * t), -1.0, 1.0)
    return out


@torch.no_grad()
def seed_weights(model: torch.nn.Module, seed: int = 1234, scale: float = 4.0) -> torch.nn.Module:
    """In-place; iterates ``state_dict()`` in order with one CPU generator."""
    g = torch.Generator().manual_seed(seed)
    hidden = model.hparams.lstm["hidden_size"] if hasattr(model, "hparams") else 128
    for k, v in model.state_dict().items():
        if k.startswith("sincnet."):
            continue   # the SincNet front end of a PyanNet keeps its own (mel-spaced / affine-identity) initialisation
        if k.startswith("lstm"):
            bound = 1.0 / math.sqrt(hidden)
        else:
            bound = 1.0 / math.sqrt(v.shape[-1] if v.dim() > 1 else hidden)
        v.copy_(((torch.rand(v.shape, generator=g) * 2 - 1) * bound * scale).to(v.dtype))
    return model
