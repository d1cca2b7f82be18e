"""torch-CPU restatement of the hot path (CPU ORACLE -- test infrastructure only).

Same operator sequence as the reference, written with stock torch CPU operators so it is
multi-threaded (this is the ``cpu_baseline`` ``bench.py`` times, kind "port"):

* ``TorchPyanNet2``  -- src/models/segmentation/PyanNet2.py:95 (nn.LSTM construction),
                        :122-135 (linear stack), :149-152 (classifier), :154-187 (forward).
                        Pinned against the reference's own class by tests/golden/pyannet2_*.npz.
* ``torch_fbank``    -- lhotse Fbank (un-vendored third party; call sites
                        src/datasets/ami/utils.py:153, src/utils/helper.py:120): strided frames
                        of the reflect-padded waveform, mean removal, pre-emphasis with a
                        replicate-padded shift, window, zero-pad to n_fft, ``torch.fft.rfft``,
                        ``|.|^2``, matmul with the mel matrix, ``max(eps).log()``.
                        PARITY UNPINNED (no fixture in the reference); cross-checked against
                        the independent float64 DFT in uvad_oracle.c.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def num_frames(S, frame_len=400, shift=160, snip_edges=False):
    if snip_edges:
        return 0 if S < frame_len else 1 + (S - frame_len) // shift
    return (S + shift // 2) // shift


def make_window(kind, n):
    if kind == "povey":
        return torch.hann_window(n, periodic=False, dtype=torch.float64).pow(0.85).float()
    if kind == "hamming":
        return torch.hamming_window(n, periodic=False, alpha=0.54, beta=0.46, dtype=torch.float64).float()
    if kind == "hanning":
        return torch.hann_window(n, periodic=False, dtype=torch.float64).float()
    if kind == "rectangular":
        return torch.ones(n)
    raise ValueError(kind)


def make_mel(n_mels=80, n_fft=512, sample_rate=16000.0, low_hz=20.0, high_hz=-400.0):
    """[n_mels][n_fft/2+1] triangular filters, SURVEY.md Appendix A step 4."""
    if high_hz <= 0:
        high_hz = sample_rate / 2 + high_hz
    mel = lambda hz: 1127.0 * np.log(1.0 + np.asarray(hz, np.float64) / 700.0)
    ml, mh = mel(low_hz), mel(high_hz)
    d = (mh - ml) / (n_mels + 1)
    nb = n_fft // 2 + 1
    out = np.zeros((n_mels, nb), np.float32)
    mk = mel(np.arange(n_fft // 2) * sample_rate / n_fft)
    for m in range(n_mels):
        l, c, r = ml + m * d, ml + (m + 1) * d, ml + (m + 2) * d
        up = (mk - l) / (c - l)
        dn = (r - mk) / (r - c)
        w = np.where(mk <= c, up, dn)
        w = np.where((mk > l) & (mk < r), w, 0.0)
        out[m, : n_fft // 2] = w.astype(np.float32)
    return torch.from_numpy(out)


def torch_fbank(pcm, window, mel, frame_shift=160, n_fft=512, preemph=0.97, remove_dc=True,
                snip_edges=False, log_floor=float(np.finfo(np.float32).eps)):
    """pcm (B,S) f32 -> (B,T,n_mels) f32 on CPU."""
    pcm = torch.as_tensor(pcm, dtype=torch.float32)
    B, S = pcm.shape
    L = window.numel()
    T = num_frames(S, L, frame_shift, snip_edges)
    if not snip_edges:
        n_left = (L - frame_shift) // 2
        n_right = (T - 1) * frame_shift + L - S - n_left
        left = torch.flip(pcm[:, :n_left], (1,))
        right = torch.flip(pcm[:, S - n_right:], (1,)) if n_right > 0 else pcm.new_zeros(B, 0)
        x = torch.cat((left, pcm, right), dim=1).contiguous()
    else:
        x = pcm.contiguous()
    fr = x.as_strided((B, T, L), (x.stride(0), frame_shift, 1))
    if remove_dc:
        fr = fr - fr.mean(dim=2, keepdim=True)
    if preemph != 0.0:
        sh = F.pad(fr, (1, 0), mode="replicate")[:, :, :-1]
        fr = fr - preemph * sh
    fr = fr * window
    fr = F.pad(fr, (0, n_fft - L))
    spec = torch.fft.rfft(fr)
    pw = spec.real.square() + spec.imag.square()
    out = torch.matmul(pw, mel.t())
    return torch.clamp_min(out, log_floor).log()


class TorchPyanNet2(nn.Module):
    """Operator-for-operator restatement of the reference classifier on stock torch."""

    def __init__(self, encoding_dim=80, hidden=128, num_layers=4, bidirectional=True,
                 lin_hidden=128, lin_layers=2, leaky_slope=0.01):
        super().__init__()
        self.lstm = nn.LSTM(encoding_dim, hidden, num_layers=num_layers, bidirectional=bidirectional,
                            batch_first=True, dropout=0.0)
        dims = [hidden * (2 if bidirectional else 1)] + [lin_hidden] * lin_layers
        self.linear = nn.ModuleList([nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:])])
        self.classifier = nn.Linear(dims[-1], 1)
        self.leaky_slope = leaky_slope

    @torch.no_grad()
    def forward(self, feats, taps=False):
        y, _ = self.lstm(feats)
        z = y
        for lin in self.linear:
            z = F.leaky_relu(lin(z), self.leaky_slope)
        logits = self.classifier(z).squeeze(-1)
        probs = torch.sigmoid(logits)
        if taps:
            return logits, probs, y, z
        return logits, probs


def seeded_state_dict(encoding_dim=80, hidden=128, num_layers=4, bidirectional=True, lin_hidden=128,
                      lin_layers=2, seed=1234, scale=4.0):
    """Deterministic weights.  Default torch init gives outputs ~0.506 +- 0.001 (SURVEY.md
    App.B), which makes parity tests vacuous, so weights are scaled (x4 spans 0.05..0.99)."""
    g = torch.Generator().manual_seed(seed)
    m = TorchPyanNet2(encoding_dim, hidden, num_layers, bidirectional, lin_hidden, lin_layers)
    sd = {}
    for k, v in m.state_dict().items():
        bound = 1.0 / math.sqrt(hidden if k.startswith("lstm") else v.shape[-1] if v.dim() > 1 else hidden)
        sd[k] = ((torch.rand(v.shape, generator=g) * 2 - 1) * bound * scale).float()
    return sd


def synth_pcm(B, S, seed=1000, sample_rate=16000):
    """0.1*N(0,1) noise + AM-modulated 100-300 Hz harmonic bursts, per-utterance
    seed = seed + i, clipped to [-1, 1].  Pure numpy so every rank/shard can regenerate it."""
    out = np.empty((B, S), np.float32)
    t = np.arange(S, dtype=np.float64) / sample_rate
    for i in range(B):
        rng = np.random.default_rng(seed + i)
        x = 0.1 * rng.standard_normal(S)
        dur = S / sample_rate
        for _ in range(5):
            f0 = rng.uniform(100.0, 300.0)
            st = rng.uniform(0.0, max(dur - 0.5, 0.1))
            ln = rng.uniform(0.5, 3.0)
            env = ((t >= st) & (t < st + ln)) * (0.5 + 0.5 * np.sin(2 * np.pi * 4.0 * (t - st)))
            sig = sum(np.sin(2 * np.pi * f0 * h * t + rng.uniform(0, 6.28)) / h for h in range(1, 6))
            x = x + 0.15 * env * sig
        out[i] = np.clip(x, -1.0, 1.0).astype(np.float32)
    return out


# ---------------------------------------------------------------------------------------------------------------
# SincNet front end: reference src/models/blocks/sincnet.py:33-103, used by PyanNet.forward
# (src/models/segmentation/PyanNet.py:162-195).  The first layer is asteroid_filterbanks.ParamSincFB
# (requirements.txt:1, asteroid-filterbanks==0.4), which is NOT in this image: its filter construction is
# restated below from the published algorithm (cos / sin band-pass pairs from learnable low / band edges, half
# Hamming window, mel-spaced initialisation) and is PARITY UNPINNED; the layers after the filter bank are stock
# torch modules (Conv1d / MaxPool1d / InstanceNorm1d / leaky_relu) and follow sincnet.py line by line.
# Known answer pinned by the reference: 80000 samples -> 293 frames (src/datasets/custom_vad.py:47,
# src/utils/receptive_field.py:165-193).

def sinc_init_params(n_filters=80, sample_rate=16000.0, min_low_hz=50.0, min_band_hz=50.0):
    """Mel-spaced initial (low_hz_, band_hz_) of ParamSincFB, each (n_filters/2, 1)."""
    to_mel = lambda hz: 2595.0 * np.log10(1.0 + hz / 700.0)
    to_hz = lambda mel: 700.0 * (10.0 ** (mel / 2595.0) - 1.0)
    mel = np.linspace(to_mel(30.0), to_mel(sample_rate / 2 - (min_low_hz + min_band_hz)), n_filters // 2 + 1, dtype="float32")
    hz = to_hz(mel).astype(np.float32)
    return torch.from_numpy(hz[:-1]).view(-1, 1), torch.from_numpy(np.diff(hz)).view(-1, 1)


def sinc_filters(low_hz_, band_hz_, kernel_size=251, sample_rate=16000.0, min_low_hz=50.0, min_band_hz=50.0):
    """(n_filters, kernel_size) f32: [cos filters ; sin filters]."""
    half = kernel_size // 2
    window = torch.from_numpy(np.hamming(kernel_size)[:half].astype(np.float32))
    n_ = 2 * np.pi * (torch.arange(-half, 0.0).view(1, -1) / sample_rate)
    low = min_low_hz + torch.abs(low_hz_.float())
    high = torch.clamp(low + min_band_hz + torch.abs(band_hz_.float()), min_low_hz, sample_rate / 2)
    band = (high - low)[:, 0]
    ft_low, ft_high = torch.matmul(low, n_), torch.matmul(high, n_)
    cos_left = ((torch.sin(ft_high) - torch.sin(ft_low)) / (n_ / 2)) * window
    cos = torch.cat([cos_left, 2 * band.view(-1, 1), torch.flip(cos_left, dims=[1])], dim=1) / (2 * band[:, None])
    sin_left = ((torch.cos(ft_low) - torch.cos(ft_high)) / (n_ / 2)) * window
    sin = torch.cat([sin_left, torch.zeros_like(band.view(-1, 1)), -torch.flip(sin_left, dims=[1])], dim=1) / (2 * band[:, None])
    return torch.cat([cos, sin], dim=0).float()


def sincnet_num_frames(S, stride=10):
    n = (S - 251) // stride + 1
    n = n // 3
    n = (n - 4) // 3
    n = (n - 4) // 3
    return n


class TorchSincNet(nn.Module):
    """sincnet.py:33-103 on stock torch modules; the sinc layer is a conv1d with the materialised filter bank."""

    def __init__(self, stride=10):
        super().__init__()
        self.stride = stride
        self.wav_norm1d = nn.InstanceNorm1d(1, affine=True)
        low, band = sinc_init_params()
        self.low_hz_ = nn.Parameter(low)
        self.band_hz_ = nn.Parameter(band)
        self.norm1d = nn.ModuleList([nn.InstanceNorm1d(80, affine=True), nn.InstanceNorm1d(60, affine=True), nn.InstanceNorm1d(60, affine=True)])
        self.conv1d = nn.ModuleList([nn.Conv1d(80, 60, 5), nn.Conv1d(60, 60, 5)])   # conv1d.1 / conv1d.2 of the reference

    @torch.no_grad()
    def forward(self, wav):   # (B, 1, S) -> (B, 60, frames)
        x = self.wav_norm1d(wav)
        x = F.conv1d(x, sinc_filters(self.low_hz_, self.band_hz_).unsqueeze(1), stride=self.stride)
        x = torch.abs(x)
        x = F.leaky_relu(self.norm1d[0](F.max_pool1d(x, 3, 3)))
        x = F.leaky_relu(self.norm1d[1](F.max_pool1d(self.conv1d[0](x), 3, 3)))
        x = F.leaky_relu(self.norm1d[2](F.max_pool1d(self.conv1d[1](x), 3, 3)))
        return x


def seeded_sincnet(seed=99):
    g = torch.Generator().manual_seed(seed)
    m = TorchSincNet()
    with torch.no_grad():
        for name, p in m.named_parameters():
            if name in ("low_hz_", "band_hz_"):
                p.mul_(1.0 + 0.05 * (torch.rand(p.shape, generator=g) - 0.5))       # perturb the mel initialisation
            elif "norm" in name and name.endswith("weight"):
                p.copy_(1.0 + 0.2 * (torch.rand(p.shape, generator=g) - 0.5))
            elif "norm" in name:
                p.copy_(0.2 * (torch.rand(p.shape, generator=g) - 0.5))
            else:
                p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) / math.sqrt(p.shape[1] * p.shape[2] if p.dim() == 3 else 300.0))
    return m.eval()
