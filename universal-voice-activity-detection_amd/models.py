"""Host-side mirror of the reference's ``src.models`` API for the log-mel path.

``PyanNet2`` keeps the reference's constructor, ``build()``, ``encoding_dim`` / ``hparams``
attributes, ``forward`` contract and -- because its parameters live in the same
``nn.LSTM`` / ``nn.Linear`` containers -- the exact ``state_dict`` key names and layouts of
src/models/segmentation/PyanNet2.py:69-152, so checkpoints load unchanged.  Those modules are
parameter containers only: ``forward`` never calls them.  It hands the raw device pointer of the
feature tensor to ``uvad_classify`` (hand-written HIP: MFMA input projections, register-resident
recurrence, MFMA feed-forward, shuffle-reduced classifier) and returns the probabilities as
``(B, T, 1)`` exactly like PyanNet2.py:154-187.  A CPU tensor is an error, not a fallback.
"""
from typing import Optional

import torch
import torch.nn as nn


class HParams(dict):
    """Attribute-style dict standing in for Lightning's ``self.hparams``."""
    __getattr__ = dict.__getitem__

    def __setattr__(self, k, v):
        self[k] = v


def _merged(defaults: dict, custom: Optional[dict]) -> dict:
    out = dict(defaults)
    out.update(custom or {})
    return out


class PyanNet2(nn.Module):
    LSTM_DEFAULTS = {"hidden_size": 128, "num_layers": 4, "bidirectional": True, "monolithic": True, "dropout": 0.5}
    LINEAR_DEFAULTS = {"hidden_size": 128, "num_layers": 2}

    def __init__(self, lstm: dict = None, linear: dict = None, encoding_dim: int = 768,
                 sample_rate: int = 16000, num_channels: int = 1):
        super().__init__()
        lstm_hp = _merged(self.LSTM_DEFAULTS, lstm)
        lstm_hp["batch_first"] = True
        linear_hp = _merged(self.LINEAR_DEFAULTS, linear)
        self.hparams = HParams(lstm=lstm_hp, linear=linear_hp)
        self.encoding_dim = encoding_dim
        self.sample_rate, self.num_channels = sample_rate, num_channels

        width = lstm_hp["hidden_size"] * (2 if lstm_hp["bidirectional"] else 1)
        kw = {k: v for k, v in lstm_hp.items() if k != "monolithic"}
        if lstm_hp["monolithic"]:
            if kw["num_layers"] == 1:
                kw["dropout"] = 0.0  # (torch only warns; eval-mode inference never applies it)
            self.lstm = nn.LSTM(encoding_dim, **kw)
        else:
            kw.update(num_layers=1, dropout=0.0)
            self.lstm = nn.ModuleList(
                [nn.LSTM(encoding_dim if i == 0 else width, **kw) for i in range(lstm_hp["num_layers"])])
        if linear_hp["num_layers"] >= 1:
            dims = [width] + [linear_hp["hidden_size"]] * linear_hp["num_layers"]
            self.linear = nn.ModuleList([nn.Linear(i, o) for i, o in zip(dims[:-1], dims[1:])])
        self._rt = None
        self._rt_stamp = None

    def build(self):
        lin = self.hparams.linear
        lstm = self.hparams.lstm
        in_features = lin["hidden_size"] if lin["num_layers"] > 0 else lstm["hidden_size"] * (2 if lstm["bidirectional"] else 1)
        self.classifier = nn.Linear(in_features, 1)
        self.activation = nn.Sigmoid()

    # ---------------------------------------------------------------- HIP plumbing
    def _stamp(self, device):
        return (str(device),) + tuple((p.data_ptr(), p._version) for p in self.parameters())

    def _flat_state_dict(self):
        """What goes to uvad_set_weight (the C side maps the ModuleList names ``lstm.{k}.weight_ih_l0`` -> ``lstm.weight_ih_l{k}``)."""
        return dict(self.state_dict())

    def runtime(self, device):
        """The VadRuntime bound to this module's current weights on ``device`` (rebuilt if they changed)."""
        from .runtime import VadRuntime
        if not hasattr(self, "classifier"):
            raise RuntimeError("call .build() before using the model (as the reference does, vad_engine.py:42)")
        stamp = self._stamp(device)
        if self._rt is None or self._rt_stamp != stamp:
            same_ctx = self._rt is not None and self._rt_stamp is not None and self._rt_stamp[0] == stamp[0]
            if not same_ctx:   # first use, another device, or another feature front end: new context
                if self._rt is not None:
                    self._rt.close()
                model = {"encoding_dim": self.encoding_dim, "lstm": self.hparams.lstm, "linear": self.hparams.linear}
                self._rt = VadRuntime(device=device, fbank=self._fbank_cfg, model=model, **self._runtime_extra())
            self._rt.load_state_dict(self._flat_state_dict())   # changed parameters: weight hot-swap (uvad_finalize replaces the upload)
            self._rt_stamp = stamp
        return self._rt

    _fbank_cfg = None

    def _runtime_extra(self) -> dict:
        return {}

    def attach_fbank(self, config):
        """Optional: give the model a FbankConfig so ``forward_waveform`` can run the fused
        PCM -> logits path (``uvad_forward``) with features staying in the workspace."""
        if config.num_filters != self.encoding_dim:
            raise ValueError("FbankConfig.num_filters must equal encoding_dim")
        self._fbank_cfg = config
        self._rt_stamp = None
        return self

    @staticmethod
    def _require_gpu(x, what):
        if not (torch.is_tensor(x) and x.is_cuda):
            raise RuntimeError(f"{what} must be a GPU tensor: this package runs the VAD path in HIP kernels only "
                               "and has no CPU fallback (move the batch to 'cuda')")

    @torch.no_grad()
    def forward(self, audio_feats: torch.Tensor) -> torch.Tensor:
        """(batch, frames, features) -> (batch, frames, 1) speech probabilities."""
        self._require_gpu(audio_feats, "audio_feats")
        _, probs = self.runtime(audio_feats.device).classify(audio_feats, want_logits=False)
        return probs.unsqueeze(-1)

    @torch.no_grad()
    def forward_logits(self, audio_feats: torch.Tensor):
        """(logits, probs), both (batch, frames): the pre-sigmoid value BASELINE.json's metric is judged on."""
        self._require_gpu(audio_feats, "audio_feats")
        return self.runtime(audio_feats.device).classify(audio_feats)

    @torch.no_grad()
    def forward_waveform(self, pcm: torch.Tensor):
        """(batch, samples) PCM -> (logits, probs); needs ``attach_fbank``."""
        self._require_gpu(pcm, "pcm")
        if self._fbank_cfg is None:
            raise RuntimeError("attach_fbank(FbankConfig(...)) first")
        return self.runtime(pcm.device).forward(pcm)


class PyanNet(PyanNet2):
    """Mirror of src/models/segmentation/PyanNet.py:37-195: SincNet front end on the raw waveform, then the same
    LSTM / feed-forward / classifier stack as PyanNet2 with ``encoding_dim`` = 60.  ``forward`` takes the
    reference's (batch, channel = 1, samples) tensor ON THE GPU and returns (batch, frames, 1) probabilities;
    the whole pass is ``uvad_forward_wav`` (csrc/sincnet.hip + the classifier kernels)."""

    SINCNET_DEFAULTS = {"stride": 10}

    def __init__(self, sincnet: dict = None, lstm: dict = None, linear: dict = None, encoding_dim: int = 60,
                 sample_rate: int = 16000, num_channels: int = 1):
        super().__init__(lstm=lstm, linear=linear, encoding_dim=encoding_dim, sample_rate=sample_rate, num_channels=num_channels)
        from .sincnet import SincNet
        sn = _merged(self.SINCNET_DEFAULTS, sincnet)
        sn["sample_rate"] = sample_rate
        self.hparams["sincnet"] = sn
        self.sincnet = SincNet(**sn)
        self.sincnet._run = self._sincnet_features

    def _sincnet_features(self, wav2d):
        self._require_gpu(wav2d, "waveforms")
        return self.runtime(wav2d.device).sincnet(wav2d)

    def _runtime_extra(self) -> dict:
        return {"sincnet": self.sincnet.config()}

    def _flat_state_dict(self):
        sd = super()._flat_state_dict()
        sd["sincnet.conv1d.0.filters"] = self.sincnet.conv1d[0].filterbank.filters()[:, 0, :]
        return sd

    def num_frames(self, num_samples: int) -> int:
        return self.sincnet.num_frames(num_samples, self.sincnet.stride)

    @torch.no_grad()
    def forward(self, waveforms: torch.Tensor) -> torch.Tensor:
        """(batch, channel, samples) -> (batch, frames, 1) speech probabilities."""
        _, probs = self.forward_logits(waveforms, want_logits=False)
        return probs.unsqueeze(-1)

    @torch.no_grad()
    def forward_logits(self, waveforms: torch.Tensor, want_logits=True):
        self._require_gpu(waveforms, "waveforms")
        if waveforms.dim() == 3:
            assert waveforms.shape[1] == 1, f"Only single channel is supported. You have {waveforms.shape[1]}"
            waveforms = waveforms[:, 0, :]
        return self.runtime(waveforms.device).forward_wav(waveforms, want_logits=want_logits)

    def forward_waveform(self, pcm: torch.Tensor):
        return self.forward_logits(pcm)

    def attach_fbank(self, config):
        raise RuntimeError("PyanNet consumes raw waveforms (SincNet); log-mel features belong to PyanNet2")
