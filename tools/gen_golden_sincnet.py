#!/usr/bin/env python3
"""Generate tests/golden/pyannet_sincnet_*.npz by running the REFERENCE's own SincNet / PyanNet classes on CPU
(src/models/blocks/sincnet.py, src/models/segmentation/PyanNet.py).  Build container only (needs /root/reference).

What is and is not the reference here.  The import chain needs pytorch_lightning, lhotse (name-only placeholders,
as in tools/gen_golden.py) and asteroid_filterbanks.  asteroid's ``Encoder(ParamSincFB(...))`` is the FIRST LAYER's
arithmetic and the package is not in this image, so its placeholder is the oracle's restatement
(oracle/torch_ref.py: sinc_init_params / sinc_filters; conv1d with the materialised bank at the bank's stride):
that layer is PARITY UNPINNED.  Everything after it -- wav_norm1d, |.|, the three MaxPool1d / InstanceNorm1d /
leaky_relu stages, the two Conv1d, the rearrange, LSTM stack, linear layers, classifier, sigmoid, and the order
they are applied in -- is the reference's code executing on stock torch.nn, and that is what these fixtures pin.

Fixture contents: wav (B, S); the SincNet part of the reference model's state_dict (torch default init under
manual_seed(77), band edges perturbed so the bank is not the initial one); the filter bank that was used;
sincnet_out (B, 60, frames); probs (B, frames); classifier weights are seeded_state_dict(60, seed 1234, x4) and
only their sha256 is stored."""
import hashlib
import os
import sys
import tempfile

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REPO)

from tools.gen_golden import _install_placeholders   # noqa: E402


def _install_filterbank_restatement():
    d = tempfile.mkdtemp(prefix="uvad_stub_fb_")
    os.makedirs(os.path.join(d, "asteroid_filterbanks"))
    with open(os.path.join(d, "asteroid_filterbanks", "__init__.py"), "w") as f:
        f.write(
            "import torch, torch.nn as nn, torch.nn.functional as F\n"
            "from oracle.torch_ref import sinc_init_params, sinc_filters\n"
            "class ParamSincFB(nn.Module):\n"
            "    def __init__(self, n_filters, kernel_size, stride=None, sample_rate=16000.0, min_low_hz=50, min_band_hz=50):\n"
            "        super().__init__()\n"
            "        self.kernel_size, self.stride, self.sample_rate = kernel_size, stride, float(sample_rate)\n"
            "        self.min_low_hz, self.min_band_hz = min_low_hz, min_band_hz\n"
            "        low, band = sinc_init_params(n_filters, self.sample_rate, min_low_hz, min_band_hz)\n"
            "        self.low_hz_, self.band_hz_ = nn.Parameter(low), nn.Parameter(band)\n"
            "    def filters(self):\n"
            "        return sinc_filters(self.low_hz_, self.band_hz_, self.kernel_size, self.sample_rate, self.min_low_hz, self.min_band_hz).unsqueeze(1)\n"
            "class Encoder(nn.Module):\n"
            "    def __init__(self, filterbank):\n"
            "        super().__init__()\n"
            "        self.filterbank = filterbank\n"
            "    def forward(self, waveform):\n"
            "        return F.conv1d(waveform, self.filterbank.filters(), stride=self.filterbank.stride)\n")
    sys.path.insert(0, d)   # ahead of the name-only placeholder of tools/gen_golden.py


def main():
    from oracle.torch_ref import seeded_state_dict, synth_pcm
    _install_placeholders()
    _install_filterbank_restatement()
    sys.path.insert(0, REF)
    import importlib
    PyanNet = importlib.import_module("src.models.segmentation.PyanNet").PyanNet
    out_dir = os.path.join(REPO, "tests", "golden")
    torch.set_num_threads(4)
    for name, B, S in (("pyannet_sincnet_S24000", 2, 24000), ("pyannet_sincnet_S80000", 1, 80000)):
        torch.manual_seed(77)
        model = PyanNet()
        model.build()
        model.eval()
        csd = seeded_state_dict(60, seed=1234, scale=4.0)
        with torch.no_grad():
            g = torch.Generator().manual_seed(78)
            fb = model.sincnet.conv1d[0].filterbank
            fb.low_hz_.mul_(1.0 + 0.05 * (torch.rand(fb.low_hz_.shape, generator=g) - 0.5))
            fb.band_hz_.mul_(1.0 + 0.05 * (torch.rand(fb.band_hz_.shape, generator=g) - 0.5))
            for i in range(3):   # non-trivial affine norms (default init is the identity)
                model.sincnet.norm1d[i].weight.copy_(1.0 + 0.2 * (torch.rand(model.sincnet.norm1d[i].weight.shape, generator=g) - 0.5))
                model.sincnet.norm1d[i].bias.copy_(0.2 * (torch.rand(model.sincnet.norm1d[i].bias.shape, generator=g) - 0.5))
            model.sincnet.wav_norm1d.weight.fill_(1.1)
            model.sincnet.wav_norm1d.bias.fill_(-0.05)
        missing, unexpected = model.load_state_dict(csd, strict=False)
        assert not unexpected and all(k.startswith("sincnet.") for k in missing), (missing, unexpected)
        wav = torch.from_numpy(synth_pcm(B, S, seed=500))
        taps = {}

        def tap(mod, inp, out):
            taps["sincnet_out"] = out

        h = model.sincnet.register_forward_hook(tap)
        with torch.no_grad():
            probs = model(wav.unsqueeze(1))
        h.remove()
        blob = {"wav": wav.numpy(), "sincnet_out": taps["sincnet_out"].numpy(), "probs": probs.squeeze(-1).numpy(),
                "filters": fb.filters()[:, 0].detach().numpy(),
                "classifier_sha256": np.array(hashlib.sha256(b"".join(csd[k].numpy().tobytes() for k in sorted(csd))).hexdigest())}
        for k, v in model.state_dict().items():
            if k.startswith("sincnet."):
                blob["sd:" + k] = v.numpy()
        path = os.path.join(out_dir, name + ".npz")
        np.savez_compressed(path, **blob)
        p = blob["probs"]
        print(f"{name}: sincnet_out {blob['sincnet_out'].shape} probs [{p.min():.3f}, {p.max():.3f}] -> {os.path.getsize(path) / 1e6:.2f} MB")


if __name__ == "__main__":
    main()


def geometry_fixture():
    """tests/golden/sincnet_geometry.json: frame counts and receptive-field numbers from the reference's own
    src/utils/receptive_field.py (pure Python, imported standalone from /root/reference)."""
    import importlib.util, json
    spec = importlib.util.spec_from_file_location("ref_receptive_field", os.path.join(REF, "src", "utils", "receptive_field.py"))
    rf = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rf)
    samples = [991, 1000, 1260, 1261, 1300, 4000, 16000, 24000, 32037, 48000, 80000, 160000, 480000]
    out = {"num_frames": {str(s): rf.get_num_frames(s) for s in samples},
           "receptive_field_size": {str(n): rf.receptive_field_size(num_frames=n) for n in (1, 2, 293)}}
    with open(os.path.join(REPO, "tests", "golden", "sincnet_geometry.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("sincnet_geometry.json:", out["num_frames"]["80000"], out["receptive_field_size"])


if __name__ == "__main__":
    geometry_fixture()
