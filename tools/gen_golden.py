#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own PyanNet2 class on CPU.

Runs only in the build container (it needs /root/reference); the fixtures it writes are
data (inputs, weights, expected outputs) and travel to the GPU box, the reference does not.

How the reference is imported: its import chain needs three packages that are not in this
image (pytorch_lightning, lhotse, asteroid_filterbanks).  None of them
contributes arithmetic to PyanNet2.forward, so NAME-ONLY placeholders are put on sys.path in
a temp dir: ``LightningModule`` = ``nn.Module`` + ``save_hyperparameters`` (captures the
caller's locals into ``self.hparams``); lhotse / asteroid_filterbanks export empty classes.
Every floating-point operation executed is the reference's code + stock torch.nn.

Fixtures (weights scaled x4, see oracle/torch_ref.py:seeded_state_dict):
  pyannet2_f80_T500.npz    B=2  T=500  F=80  default model (reference dims, 5 s cut)
  pyannet2_f64_T1000.npz   B=2  T=1000 F=64  default model (BASELINE cfg 2 frame shape)
  pyannet2_f64_T3000.npz   B=1  T=3000 F=64  error-growth case (BASELINE cfg 1: 30 s)
  pyannet2_f64_T7.npz      B=3  T=7    F=64  tiny / ragged-batch edge case
  pyannet2_uni_f64_T200.npz   bidirectional=False (streaming oracle, cfg 5)
  pyannet2_l1_f64_T100.npz    num_layers=1 bring-up case
  pyannet2_nonmono_f64_T50.npz monolithic=False variant (PyanNet2.py:174-181; same numbers)
Each holds: the sha256 of the seeded state_dict (weights are regenerated from seed 1234), feats, lstm_out, lin_out, logits, probs,
and the medfilt'ed labels of predict_step (scipy.signal.medfilt, kernel 49) as "labels49".
"""
import os
import sys
import tempfile

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REPO)


def _install_placeholders():
    d = tempfile.mkdtemp(prefix="uvad_stub_")
    os.makedirs(os.path.join(d, "pytorch_lightning"))
    with open(os.path.join(d, "pytorch_lightning", "__init__.py"), "w") as f:
        f.write(
            "import inspect, torch.nn as nn\n"
            "class _HP(dict):\n"
            "    __getattr__ = dict.__getitem__\n"
            "class LightningModule(nn.Module):\n"
            "    def save_hyperparameters(self, *names):\n"
            "        fr = inspect.currentframe().f_back\n"
            "        if not hasattr(self, 'hparams'): object.__setattr__(self, 'hparams', _HP())\n"
            "        for n in names: self.hparams[n] = fr.f_locals[n]\n"
            "def seed_everything(s): pass\n")
    os.makedirs(os.path.join(d, "lhotse"))
    with open(os.path.join(d, "lhotse", "__init__.py"), "w") as f:
        f.write("class _N: pass\nload_manifest_lazy = CutSet = FbankConfig = Fbank = LilcomChunkyWriter = _N\n")
    os.makedirs(os.path.join(d, "asteroid_filterbanks"))
    with open(os.path.join(d, "asteroid_filterbanks", "__init__.py"), "w") as f:
        f.write("class Encoder: pass\nclass ParamSincFB: pass\n")
    sys.path.insert(0, d)


def _load_reference_pyannet2():
    _install_placeholders()
    sys.path.insert(0, REF)
    # import the module file directly; src/models/__init__.py would also pull PyanNet (SincNet)
    import importlib
    mod = importlib.import_module("src.models.segmentation.PyanNet2")
    return mod.PyanNet2


def main():
    from oracle.torch_ref import seeded_state_dict
    from scipy.signal import medfilt

    PyanNet2 = _load_reference_pyannet2()
    out_dir = os.path.join(REPO, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    torch.set_num_threads(4)

    cases = [
        ("pyannet2_f80_T500", dict(F=80, B=2, T=500, lstm=None)),
        ("pyannet2_f64_T1000", dict(F=64, B=2, T=1000, lstm=None)),
        ("pyannet2_f64_T3000", dict(F=64, B=1, T=3000, lstm=None)),
        ("pyannet2_f64_T7", dict(F=64, B=3, T=7, lstm=None)),
        ("pyannet2_uni_f64_T200", dict(F=64, B=2, T=200, lstm={"bidirectional": False})),
        ("pyannet2_l1_f64_T100", dict(F=64, B=2, T=100, lstm={"num_layers": 1})),
        ("pyannet2_nonmono_f64_T50", dict(F=64, B=2, T=50, lstm={"monolithic": False})),
    ]
    for name, c in cases:
        model = PyanNet2(lstm=c["lstm"], encoding_dim=c["F"])
        model.build()
        model.eval()
        hp = model.hparams.lstm
        sd = seeded_state_dict(c["F"], hp["hidden_size"], hp["num_layers"], hp["bidirectional"], seed=1234)
        if hp["monolithic"]:
            model.load_state_dict(sd)
        else:  # ModuleList of 1-layer LSTMs: lstm.{k}.weight_ih_l0...
            remap = {}
            for k, v in sd.items():
                if k.startswith("lstm."):
                    nm = k[len("lstm."):]
                    base, layer = nm.rsplit("_l", 1)
                    rev = layer.endswith("_reverse")
                    li = int(layer.replace("_reverse", ""))
                    remap[f"lstm.{li}.{base}_l0" + ("_reverse" if rev else "")] = v
                else:
                    remap[k] = v
            model.load_state_dict(remap)
        g = torch.Generator().manual_seed(4321)
        # log-mel-like inputs: mean -8, std 4 (range of real fbank of [-1,1] audio)
        feats = torch.randn(c["B"], c["T"], c["F"], generator=g) * 4.0 - 8.0
        taps = {}
        hooks = []

        def tap_lstm(mod, inp, out):
            taps["lstm_out"] = out[0]

        def tap_classifier(mod, inp, out):
            taps["lin_out"] = inp[0]
            taps["logits"] = out

        last_lstm = model.lstm if hp["monolithic"] else model.lstm[-1]
        hooks.append(last_lstm.register_forward_hook(tap_lstm))
        hooks.append(model.classifier.register_forward_hook(tap_classifier))
        with torch.no_grad():
            probs = model(feats)
        for h in hooks:
            h.remove()
        p = probs.squeeze(-1).numpy()
        hard = np.where(p < 0.5, 0, 1).astype(np.float64)
        labels = np.stack([medfilt(r, kernel_size=49) for r in hard]).astype(np.uint8)
        # weights are NOT stored (1.4 M floats per case): they are regenerated from the seed by
        # oracle.torch_ref.seeded_state_dict and verified against this digest by the tests.
        import hashlib
        dig = hashlib.sha256(b"".join(sd[k].numpy().tobytes() for k in sorted(sd))).hexdigest()
        blob = {"weights_sha256": np.array(dig), "weights_seed": np.int32(1234)}
        if c["T"] > 1000:
            taps["lstm_out"] = taps["lstm_out"][:, :64]   # keep the fixture small: first 64 frames only
            taps["lin_out"] = taps["lin_out"][:, :64]
        blob.update(feats=feats.numpy(), lstm_out=taps["lstm_out"].numpy(), lin_out=taps["lin_out"].numpy(),
                    logits=taps["logits"].squeeze(-1).numpy(), probs=p, labels49=labels,
                    bidirectional=np.int32(hp["bidirectional"]), num_layers=np.int32(hp["num_layers"]))
        path = os.path.join(out_dir, name + ".npz")
        np.savez_compressed(path, **blob)
        print(f"{name}: probs [{p.min():.3f}, {p.max():.3f}] speech frac {labels.mean():.2f} "
              f"-> {os.path.getsize(path) / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
