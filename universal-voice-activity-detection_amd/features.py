"""Host-side mirror of the feature-extractor interface the reference uses.

The reference builds its extractor as ``Fbank(FbankConfig(sampling_rate=16000, device="cuda"))``
(src/datasets/ami/utils.py:153 and 17 more recipes; src/utils/helper.py:120) and hands it to
lhotse, which calls ``extractor.extract_batch(samples, sampling_rate)``.  ``FbankConfig`` /
``Fbank`` below keep those names, argument meanings and defaults (lhotse's: 25 ms / 10 ms frames,
povey window, 80 filters, 20 Hz .. nyquist-400 Hz, snip_edges=False, dither 0); the arithmetic
runs in the fused HIP kernel behind ``uvad_fbank`` (csrc/fbank.hip).  The window and the mel
matrix are built here on the host (float64 numpy) and uploaded once with ``uvad_set_tables``.
"""
from dataclasses import dataclass, asdict
from typing import List, Optional

import numpy as np

EPSILON = float(np.finfo(np.float32).eps)


@dataclass
class FbankConfig:
    sampling_rate: int = 16000
    frame_length: float = 0.025
    frame_shift: float = 0.01
    round_to_power_of_two: bool = True
    remove_dc_offset: bool = True
    preemph_coeff: float = 0.97
    window_type: str = "povey"      # "povey" | "hamming" | "hanning" | "rectangular"
    dither: float = 0.0
    snip_edges: bool = False
    energy_floor: float = EPSILON
    raw_energy: bool = True
    use_energy: bool = False
    use_fft_mag: bool = False
    low_freq: float = 20.0
    high_freq: float = -400.0
    num_filters: int = 80
    num_mel_bins: Optional[int] = None   # alias of num_filters
    norm_filters: bool = False
    device: str = "cuda"

    def __post_init__(self):
        if self.num_mel_bins is not None:
            self.num_filters = self.num_mel_bins
        self.num_mel_bins = self.num_filters

    def to_dict(self):
        return asdict(self)

    @property
    def frame_len_samples(self) -> int:
        return int(np.floor(self.frame_length * self.sampling_rate))

    @property
    def frame_shift_samples(self) -> int:
        return int(np.floor(self.frame_shift * self.sampling_rate))

    @property
    def n_fft(self) -> int:
        n = self.frame_len_samples
        return 1 << (n - 1).bit_length() if self.round_to_power_of_two else n


def make_window(window_type: str, n: int) -> np.ndarray:
    a = 2.0 * np.pi * np.arange(n, dtype=np.float64) / (n - 1)
    if window_type == "povey":
        w = (0.5 - 0.5 * np.cos(a)) ** 0.85
    elif window_type == "hamming":
        w = 0.54 - 0.46 * np.cos(a)
    elif window_type == "hanning":
        w = 0.5 - 0.5 * np.cos(a)
    elif window_type == "rectangular":
        w = np.ones(n)
    else:
        raise ValueError(f"unknown window_type {window_type!r}")
    return w.astype(np.float32)


def make_mel_matrix(num_filters: int, n_fft: int, sampling_rate: float, low_freq: float, high_freq: float,
                    norm_filters: bool = False) -> np.ndarray:
    """(num_filters, n_fft/2+1) triangular filters, equally spaced on the 1127*ln(1+f/700) scale;
    the Nyquist bin carries no weight (Kaldi convention)."""
    if high_freq <= 0:
        high_freq = sampling_rate / 2.0 + high_freq
    mel = lambda hz: 1127.0 * np.log(1.0 + np.asarray(hz, np.float64) / 700.0)
    lo, hi = float(mel(low_freq)), float(mel(high_freq))
    step = (hi - lo) / (num_filters + 1)
    nb = n_fft // 2 + 1
    bins = mel(np.arange(n_fft // 2, dtype=np.float64) * sampling_rate / n_fft)
    out = np.zeros((num_filters, nb), np.float32)
    for m in range(num_filters):
        left, center, right = lo + m * step, lo + (m + 1) * step, lo + (m + 2) * step
        rise = (bins - left) / (center - left)
        fall = (right - bins) / (right - center)
        tri = np.where(bins <= center, rise, fall)
        tri = np.where((bins > left) & (bins < right), tri, 0.0)
        if norm_filters and tri.sum() > 0:
            tri = tri / tri.sum()
        out[m, : n_fft // 2] = tri.astype(np.float32)
    return out


class Fbank:
    """Drop-in for lhotse's ``Fbank`` as the reference uses it.  ``extract_batch`` accepts what
    lhotse passes (a list of 1-D/2-D arrays or tensors, or one 2-D batch) and returns per-cut
    ``(T, F)`` arrays / tensors; the batch is padded to the longest cut and runs as one launch."""

    name = "kaldi-fbank"

    def __init__(self, config: Optional[FbankConfig] = None):
        self.config = config or FbankConfig()
        c = self.config
        if c.dither != 0.0 or c.use_energy or c.use_fft_mag:
            raise NotImplementedError("dither / use_energy / use_fft_mag are not on the reference's path")
        if c.n_fft != 512:
            raise NotImplementedError("only 512-point frames (25 ms @ 16 kHz) are implemented")
        self._rt = None

    @property
    def frame_shift(self) -> float:
        return self.config.frame_shift

    def feature_dim(self, sampling_rate: int) -> int:
        return self.config.num_filters

    @property
    def device(self):
        return self.config.device

    def _runtime(self, device):
        from .runtime import VadRuntime
        if self._rt is None or self._rt.device != device:
            self._rt = VadRuntime(device=device, fbank=self.config, model=None)
        return self._rt

    def extract_batch(self, samples, sampling_rate: int, lengths=None):
        import torch
        if sampling_rate != self.config.sampling_rate:
            raise ValueError(f"extractor built for {self.config.sampling_rate} Hz, got {sampling_rate}")
        dev = torch.device(self.config.device if self.config.device != "gpu" else "cuda")
        if dev.type != "cuda":
            raise RuntimeError("Fbank runs on the GPU only (device='cuda'); there is no CPU path in this package")
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        as_list = isinstance(samples, (list, tuple))
        items = list(samples) if as_list else [samples]
        was_numpy = isinstance(items[0], np.ndarray)
        rows: List = []
        for it in items:
            t = torch.as_tensor(it, dtype=torch.float32)
            t = t.reshape(-1, t.shape[-1]) if t.dim() > 1 else t.reshape(1, -1)
            if not as_list and t.shape[0] > 1:
                rows.extend(list(t))
            else:
                if t.shape[0] != 1:
                    raise ValueError("each cut must be mono")
                rows.append(t[0])
        lens = [int(r.numel()) for r in rows]
        S = max(lens)
        rt = self._runtime(dev)
        if all(l == S for l in lens):
            batch = torch.stack([r.to(dev, non_blocking=True) for r in rows]).contiguous()
            feats = rt.fbank(batch)
            outs = [feats[i] for i in range(len(rows))]
        else:  # ragged: every cut framed on its own length (reflection happens at ITS end)
            outs = [rt.fbank(r.to(dev).reshape(1, -1).contiguous())[0] for r in rows]
        if was_numpy:
            outs = [o.cpu().numpy() for o in outs]
        if not as_list and len(outs) == 1 and torch.as_tensor(items[0]).dim() <= 1:
            return outs[0]
        return outs if as_list else (torch.stack(outs) if not was_numpy else np.stack(outs))

    def extract(self, samples, sampling_rate: int):
        out = self.extract_batch([samples], sampling_rate)
        return out[0]
