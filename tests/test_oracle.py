"""CPU tests of the ORACLE itself: it must reproduce every golden vector produced by the reference's
own PyanNet2 class (tools/gen_golden.py) before it is trusted as the parity checker."""
import numpy as np
import pytest
import torch

from conftest import GOLDEN_CASES, load_golden
from oracle import c_oracle as co, torch_ref as tr

ORACLE_TOL = 5e-5   # C oracle (double accumulation) vs the reference's fp32 torch run


@pytest.mark.parametrize("name", list(GOLDEN_CASES))
def test_c_oracle_matches_reference_golden(name):
    g, sd, case = load_golden(name)
    if g["feats"].shape[1] > 1000:
        pytest.skip("T=3000 case is covered by the torch restatement below (C oracle is scalar)")
    sdn = {k: v.numpy() for k, v in sd.items()}
    mc = co.ModelCfg(case["F"], 128, case["num_layers"], int(case["bidirectional"]), 128, 2, 0.01)
    logits, probs, y, z = co.classify(sdn, mc, g["feats"], taps=True)
    assert np.abs(logits - g["logits"]).max() < ORACLE_TOL
    assert np.abs(probs - g["probs"]).max() < ORACLE_TOL
    assert np.abs(y - g["lstm_out"]).max() < ORACLE_TOL
    assert np.abs(z - g["lin_out"]).max() < ORACLE_TOL


@pytest.mark.parametrize("name", list(GOLDEN_CASES))
def test_torch_restatement_matches_reference_golden(name):
    g, sd, case = load_golden(name)
    m = tr.TorchPyanNet2(case["F"], 128, case["num_layers"], case["bidirectional"])
    m.load_state_dict(sd)
    logits, probs, y, z = m(torch.from_numpy(g["feats"]), taps=True)
    nt = g["lstm_out"].shape[1]
    assert np.abs(logits.numpy() - g["logits"]).max() < 5e-6
    assert np.abs(probs.numpy() - g["probs"]).max() < 5e-6
    assert np.abs(y.numpy()[:, :nt] - g["lstm_out"]).max() < 5e-6
    assert np.abs(z.numpy()[:, :nt] - g["lin_out"]).max() < 5e-6


def test_golden_known_answers():
    # parameter counts of the reference class
    assert sum(v.numel() for v in tr.seeded_state_dict(80).values()) == 1450369
    assert sum(v.numel() for v in tr.seeded_state_dict(64).values()) == 1433985
    g, _, _ = load_golden("pyannet2_f64_T1000")
    assert g["probs"].min() < 0.1 and g["probs"].max() > 0.9       # fixtures are not the 0.506 flat line


def test_num_frames_pins():
    cfg = co.default_fbank_cfg(80)
    assert co.num_frames(80000, cfg) == 500      # data/test_data.py:23: 5 s cut -> (B, 500, 80)
    assert co.num_frames(160000, cfg) == 1000
    assert co.num_frames(480000, cfg) == 3000
    assert co.num_frames(80079, cfg) == 500 and co.num_frames(80080, cfg) == 501
    snip = co.default_fbank_cfg(80, snip_edges=1)
    assert co.num_frames(80000, snip) == 498     # Kaldi snip_edges=True would give 498
    assert tr.num_frames(80000) == 500


def _numpy_fbank_f64(pcm, win, mel, shift=160, nfft=512, preemph=0.97, floor=np.finfo(np.float32).eps):
    """Independent float64 restatement (np.fft) of SURVEY.md Appendix A."""
    B, S = pcm.shape
    L = len(win)
    T = (S + shift // 2) // shift
    out = np.empty((B, T, mel.shape[0]))
    for b in range(B):
        x = pcm[b].astype(np.float64)
        n_right = (T - 1) * shift + L - S - 120
        xp = np.concatenate([x[:120][::-1], x, x[S - n_right:][::-1] if n_right > 0 else x[:0]])
        for t in range(T):
            f = xp[t * shift: t * shift + L].copy()
            f -= f.mean()
            f = f - preemph * np.concatenate([[f[0]], f[:-1]])
            p = np.abs(np.fft.rfft(f * win, nfft)) ** 2
            out[b, t] = np.log(np.maximum(mel.astype(np.float64) @ p, floor))
    return out


@pytest.mark.parametrize("window", ["povey", "hamming"])
def test_fbank_restatements_agree(window):
    from uvad_amd.synth import synth_pcm
    pcm = synth_pcm(2, 16000 + 57, seed=11)
    cfg = co.default_fbank_cfg(80)
    win, mel = co.window(window, 400), co.mel_banks(cfg)
    a = co.fbank(pcm, cfg, win, mel)
    b = tr.torch_fbank(pcm, torch.from_numpy(win), torch.from_numpy(mel)).numpy()
    c = _numpy_fbank_f64(pcm, win.astype(np.float64), mel)
    assert a.shape == b.shape == c.shape == (2, 100, 80)
    assert np.abs(a - c).max() < 2e-4      # C oracle (float frames, double DFT) vs pure float64
    assert np.abs(b - c).max() < 1e-3      # torch fp32 rfft vs float64
    # the float64-throughout C evaluation (the truth of the full-size feature / end-to-end tests) vs the independent numpy one
    # (on the configuration's f32 pre-emphasis coefficient, 0.97f, which is what every fp32 implementation is handed)
    d = co.fbank_f64(pcm, cfg, win, mel, threads=2)
    c32 = _numpy_fbank_f64(pcm, win.astype(np.float64), mel, preemph=float(np.float32(0.97)))
    assert d.dtype == np.float64 and np.abs(d - c32).max() < 1e-9


def test_oracle_tables_match_product_tables():
    from uvad_amd.features import make_mel_matrix, make_window
    for kind in ("povey", "hamming", "hanning", "rectangular"):
        assert np.array_equal(co.window(kind, 400), make_window(kind, 400))
        assert np.allclose(tr.make_window(kind, 400).numpy(), make_window(kind, 400), atol=1e-7)
    for n in (64, 80):
        cfg = co.default_fbank_cfg(n)
        m = make_mel_matrix(n, 512, 16000, 20.0, -400.0)
        assert np.abs(co.mel_banks(cfg) - m).max() < 1e-6
        assert np.abs(tr.make_mel(n).numpy() - m).max() < 1e-6
        assert (m[:, 256] == 0).all() and ((m > 0).sum(axis=0) <= 2).all()   # <= 2 filters per bin


def test_median_filter_and_intervals_vs_scipy():
    from scipy.signal import medfilt
    rng = np.random.default_rng(0)
    probs = rng.random((3, 400)).astype(np.float32)
    probs[0, :60] = 0.9
    probs[1, 100:300] = 0.1
    for k in (49, 25, 3, 1):
        want = np.stack([medfilt(np.where(r < 0.5, 0, 1).astype(np.float64), k) for r in probs]).astype(np.uint8)
        assert np.array_equal(co.median_filter(probs, k), want)
    g, _, _ = load_golden("pyannet2_f64_T1000")
    assert np.array_equal(co.median_filter(g["probs"], 49), g["labels49"])
    lab = np.array([0, 1, 1, 1, 0, 0, 1, 0, 1, 1], np.uint8)
    # predict.py:472-490: (k*shift, (k2-1)*shift), dropped when end-start <= 0, open run closed at len-1
    assert co.intervals(lab, 0.01) == [(0.01, 0.03), (0.08, 0.09)]


# ---- SincNet front end -------------------------------------------------------------------------------

SINC_GOLDENS = ("pyannet_sincnet_S24000", "pyannet_sincnet_S80000")


def _load_sinc_golden(name):
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"), allow_pickle=False)
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd:")}
    return g, sd


@pytest.mark.parametrize("name", SINC_GOLDENS)
def test_sincnet_restatement_matches_reference_class_golden(name):
    """oracle.torch_ref.TorchSincNet + TorchPyanNet2(60) reproduce what the reference's own SincNet / PyanNet classes
    produced (tools/gen_golden_sincnet.py): pins the order of norm / pool / activation / rearrange and the frame count.
    The filter bank is the restated ParamSincFB on both sides (asteroid is absent: PARITY UNPINNED for that layer)."""
    import hashlib
    from oracle import torch_ref as tr
    g, sd = _load_sinc_golden(name)
    front = tr.TorchSincNet().eval()
    fsd = {"wav_norm1d.weight": sd["sincnet.wav_norm1d.weight"], "wav_norm1d.bias": sd["sincnet.wav_norm1d.bias"],
           "low_hz_": sd["sincnet.conv1d.0.filterbank.low_hz_"], "band_hz_": sd["sincnet.conv1d.0.filterbank.band_hz_"]}
    for i in range(3):
        for p in ("weight", "bias"):
            fsd[f"norm1d.{i}.{p}"] = sd[f"sincnet.norm1d.{i}.{p}"]
    for i in range(2):
        for p in ("weight", "bias"):
            fsd[f"conv1d.{i}.{p}"] = sd[f"sincnet.conv1d.{i + 1}.{p}"]
    front.load_state_dict(fsd)
    assert torch.equal(tr.sinc_filters(front.low_hz_, front.band_hz_), torch.from_numpy(g["filters"]))
    wav = torch.from_numpy(g["wav"])
    feats = front(wav.unsqueeze(1))
    assert feats.shape == g["sincnet_out"].shape and feats.shape[2] == tr.sincnet_num_frames(wav.shape[1])
    assert np.abs(feats.numpy() - g["sincnet_out"]).max() < 1e-5
    csd = tr.seeded_state_dict(60, seed=1234, scale=4.0)
    assert hashlib.sha256(b"".join(csd[k].numpy().tobytes() for k in sorted(csd))).hexdigest() == str(g["classifier_sha256"])
    cls = tr.TorchPyanNet2(60)
    cls.load_state_dict(csd)
    _, probs = cls(feats.transpose(1, 2).contiguous())
    assert np.abs(probs.numpy() - g["probs"]).max() < 1e-5


def test_sincnet_frame_count_matches_reference_receptive_field_module():
    """tests/golden/sincnet_geometry.json holds get_num_frames / receptive_field_size of the reference's own
    src/utils/receptive_field.py (tools/gen_golden_sincnet.py): 991 samples -> 1 frame, 1261 -> 2, 80000 -> 293."""
    import json, os
    import uvad_amd
    from oracle import torch_ref as tr
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "sincnet_geometry.json")))
    for s, n in g["num_frames"].items():
        assert tr.sincnet_num_frames(int(s)) == n, s
        assert uvad_amd.SincNet.num_frames(int(s), 10) == n, s
    assert g["receptive_field_size"]["1"] == 991 and g["receptive_field_size"]["2"] - g["receptive_field_size"]["1"] == 270


def test_c_oracle_handles_wide_feed_forward_and_input():
    """Regression: the C oracle's scratch buffers must hold the widest activation (lin_hidden > hidden*dirs once
    corrupted the heap).  Cross-checked against the torch restatement on awkward constructor shapes."""
    from oracle import c_oracle as co, torch_ref as tr
    for (F, H, L, bi, lin_h, lin_l, B, T) in ((60, 64, 3, False, 100, 2, 8, 2), (132, 64, 2, False, 128, 0, 2, 1),
                                              (4, 64, 1, False, 200, 3, 3, 5), (300, 128, 1, True, 4, 1, 2, 3)):
        sd = tr.seeded_state_dict(F, H, L, bi, lin_h, lin_l, seed=9, scale=2.0)
        g = torch.Generator().manual_seed(1)
        feats = torch.randn(B, T, F, generator=g)
        want, _ = co.classify({k: v.numpy() for k, v in sd.items()}, co.ModelCfg(F, H, L, int(bi), lin_h, lin_l, 0.01), feats.numpy())
        ref = tr.TorchPyanNet2(F, H, L, bi, lin_h, lin_l)
        ref.load_state_dict(sd)
        assert np.abs(ref(feats)[0].numpy() - want).max() < 1e-5
