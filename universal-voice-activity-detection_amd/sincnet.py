"""Host-side mirror of the reference's SincNet block (src/models/blocks/sincnet.py:33-103).

The modules below are PARAMETER CONTAINERS with the reference's ``state_dict`` names and layouts
(``wav_norm1d.*``, ``conv1d.0.filterbank.{low_hz_,band_hz_}``, ``conv1d.{1,2}.*``, ``norm1d.{0,1,2}.*``), so a
PyanNet checkpoint loads unchanged; nothing here computes the forward pass -- that is ``uvad_sincnet`` in
libuvad.so (csrc/sincnet.hip).  The only arithmetic on the host is ``ParamSincFB.filters()``: the 80 x 251
band-pass bank materialised once per weight update from the 2 x 40 learnable band edges, which the reference
gets from asteroid_filterbanks.ParamSincFB (requirements.txt, not vendored and not installed here: restated from
the published algorithm, PARITY UNPINNED -- see DESIGN.md)."""
import math

import numpy as np
import torch
import torch.nn as nn


class ParamSincFB(nn.Module):
    """Parameterised sinc filter bank: n_filters/2 (cos, sin) band-pass pairs of odd length kernel_size."""

    def __init__(self, n_filters=80, kernel_size=251, stride=10, sample_rate=16000.0, min_low_hz=50, min_band_hz=50):
        super().__init__()
        if n_filters % 2 != 0:
            raise ValueError("n_filters must be even (cos / sin pairs)")
        if kernel_size % 2 == 0:
            kernel_size += 1
        self.n_filters, self.kernel_size, self.stride = n_filters, kernel_size, stride
        self.sample_rate, self.min_low_hz, self.min_band_hz = float(sample_rate), min_low_hz, min_band_hz
        self.half_kernel = kernel_size // 2
        self.cutoff = n_filters // 2
        to_mel = lambda hz: 2595.0 * np.log10(1.0 + hz / 700.0)
        to_hz = lambda mel: 700.0 * (10.0 ** (mel / 2595.0) - 1.0)
        mel = np.linspace(to_mel(30.0), to_mel(self.sample_rate / 2 - (min_low_hz + min_band_hz)), self.cutoff + 1, dtype="float32")
        hz = to_hz(mel).astype(np.float32)
        self.low_hz_ = nn.Parameter(torch.from_numpy(hz[:-1]).view(-1, 1))
        self.band_hz_ = nn.Parameter(torch.from_numpy(np.diff(hz)).view(-1, 1))
        self.register_buffer("window_", torch.from_numpy(np.hamming(kernel_size)[: self.half_kernel]).float())
        self.register_buffer("n_", 2 * math.pi * (torch.arange(-self.half_kernel, 0.0).view(1, -1) / self.sample_rate))

    @torch.no_grad()
    def filters(self) -> torch.Tensor:
        """(n_filters, 1, kernel_size) f32 on the CPU: [cos filters ; sin filters]."""
        low_p, band_p = self.low_hz_.detach().float().cpu(), self.band_hz_.detach().float().cpu()
        window, n_ = self.window_.detach().float().cpu(), self.n_.detach().float().cpu()
        low = self.min_low_hz + torch.abs(low_p)
        high = torch.clamp(low + self.min_band_hz + torch.abs(band_p), self.min_low_hz, self.sample_rate / 2)
        band = (high - low)[:, 0]
        ft_low, ft_high = torch.matmul(low, n_), torch.matmul(high, n_)
        cos_left = ((torch.sin(ft_high) - torch.sin(ft_low)) / (n_ / 2)) * window
        cos = torch.cat([cos_left, 2 * band.view(-1, 1), torch.flip(cos_left, dims=[1])], dim=1) / (2 * band[:, None])
        sin_left = ((torch.cos(ft_low) - torch.cos(ft_high)) / (n_ / 2)) * window
        sin = torch.cat([sin_left, torch.zeros_like(band.view(-1, 1)), -torch.flip(sin_left, dims=[1])], dim=1) / (2 * band[:, None])
        return torch.cat([cos, sin], dim=0).view(self.n_filters, 1, self.kernel_size)


class Encoder(nn.Module):
    """asteroid's Encoder(filterbank): conv1d of the waveform with filterbank.filters() at filterbank.stride."""

    def __init__(self, filterbank: ParamSincFB):
        super().__init__()
        self.filterbank = filterbank


class SincNet(nn.Module):
    def __init__(self, sample_rate: int = 16000, stride: int = 1):
        super().__init__()
        if sample_rate != 16000:
            raise NotImplementedError("Only 16kHz audio supported for now.")   # as the reference, sincnet.py:37
        self.stride = stride
        self.wav_norm1d = nn.InstanceNorm1d(1, affine=True)
        self.conv1d = nn.ModuleList([Encoder(ParamSincFB(80, 251, stride=stride, sample_rate=sample_rate, min_low_hz=50, min_band_hz=50)),
                                     nn.Conv1d(80, 60, 5, stride=1), nn.Conv1d(60, 60, 5, stride=1)])
        self.pool1d = nn.ModuleList([nn.MaxPool1d(3, stride=3, padding=0, dilation=1) for _ in range(3)])
        self.norm1d = nn.ModuleList([nn.InstanceNorm1d(80, affine=True), nn.InstanceNorm1d(60, affine=True), nn.InstanceNorm1d(60, affine=True)])
        self._run = None   # set by the owning PyanNet: waveform (B, S) on the GPU -> (B, frames, 60)

    def config(self) -> dict:
        fb = self.conv1d[0].filterbank
        return {"stride": self.stride, "n_filters": fb.n_filters, "kernel_size": fb.kernel_size,
                "c2": self.conv1d[1].out_channels, "k2": self.conv1d[1].kernel_size[0],
                "c3": self.conv1d[2].out_channels, "k3": self.conv1d[2].kernel_size[0],
                "leaky_slope": 0.01, "eps": float(self.norm1d[0].eps)}

    @staticmethod
    def num_frames(num_samples, stride: int = 10):
        """Frames of `num_samples` samples (src/utils/receptive_field.py:165-193).  An int gives an int; a float is floor-divided
        as a float, which is what the reference's predict script does with 16000 * duration (predict_sincnet.py:333)."""
        n = (num_samples - 251) // stride + 1
        for _ in range(2):
            n = n // 3 - 4
        return n // 3

    @torch.no_grad()
    def forward(self, waveforms: torch.Tensor) -> torch.Tensor:
        """(batch, channel = 1, sample) -> (batch, feature, frames), as sincnet.py:72-103."""
        assert waveforms.shape[1] == 1, f"Only single channel is supported. You have {waveforms.shape[1]}"
        if self._run is None:
            raise RuntimeError("SincNet runs inside its PyanNet (HIP kernels only, no CPU path): call PyanNet.forward")
        return self._run(waveforms[:, 0, :]).transpose(1, 2)
