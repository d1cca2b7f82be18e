"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against (a) the golden
vectors produced by the reference's own PyanNet2 class and (b) the CPU oracle on seeded inputs.

Tolerances (BASELINE.json north_star: per-frame logits within 1e-4 max-abs of the CPU reference):
  LOGIT_TOL = 1e-4 on logits; probabilities 1e-4; LSTM / feed-forward taps 1e-4;
  log-mel features 5e-4 absolute in the log domain vs the float64-DFT oracle (the torch-CPU
  rfft restatement itself differs from that oracle by ~1e-4; parity vs lhotse is UNPINNED).
"""
import numpy as np
import pytest
import torch

from conftest import GOLDEN_CASES, load_golden

pytestmark = pytest.mark.gpu

LOGIT_TOL = 1e-4
FEAT_TOL = 5e-4


def _model(case, sd, dev):
    import uvad_amd
    lstm = {"num_layers": case["num_layers"], "bidirectional": case["bidirectional"]}
    m = uvad_amd.PyanNet2(lstm=lstm, encoding_dim=case["F"])
    m.build()
    m.load_state_dict(sd)
    return m.to(dev).eval()


@pytest.mark.parametrize("name", [n for n in GOLDEN_CASES if "nonmono" not in n])
def test_classifier_matches_reference_golden(name):
    g, sd, case = load_golden(name)
    dev = torch.device("cuda:0")
    m = _model(case, sd, dev)
    feats = torch.from_numpy(g["feats"]).to(dev)
    logits, probs = m.forward_logits(feats)
    torch.cuda.synchronize()
    err = np.abs(logits.cpu().numpy() - g["logits"]).max()
    perr = np.abs(probs.cpu().numpy() - g["probs"]).max()
    y, z = m.runtime(dev).taps()
    nt = g["lstm_out"].shape[1]
    yerr = np.abs(y.cpu().numpy()[:, :nt] - g["lstm_out"]).max()
    zerr = np.abs(z.cpu().numpy()[:, :nt] - g["lin_out"]).max()
    print(f"{name}: logit err {err:.2e} prob err {perr:.2e} lstm err {yerr:.2e} lin err {zerr:.2e}")
    assert yerr < LOGIT_TOL and zerr < LOGIT_TOL
    assert err < LOGIT_TOL and perr < LOGIT_TOL
    # forward() contract: (B, T, 1) probabilities
    out = m(feats)
    assert out.shape == (feats.shape[0], feats.shape[1], 1)
    assert torch.equal(out.squeeze(-1), probs)


def test_nonmonolithic_variant_same_numbers():
    import uvad_amd
    g, sd, case = load_golden("pyannet2_nonmono_f64_T50")
    dev = torch.device("cuda:0")
    m = uvad_amd.PyanNet2(lstm={"monolithic": False}, encoding_dim=64)
    m.build()
    remap = {}
    for k, v in sd.items():
        if k.startswith("lstm."):
            base, layer = k[5:].rsplit("_l", 1)
            rev = layer.endswith("_reverse")
            remap[f"lstm.{int(layer.replace('_reverse', ''))}.{base}_l0" + ("_reverse" if rev else "")] = v
        else:
            remap[k] = v
    m.load_state_dict(remap)
    m = m.to(dev).eval()
    logits, _ = m.forward_logits(torch.from_numpy(g["feats"]).to(dev))
    assert np.abs(logits.cpu().numpy() - g["logits"]).max() < LOGIT_TOL


def test_predict_step_labels_match_scipy_medfilt_golden():
    import uvad_amd
    g, sd, case = load_golden("pyannet2_f64_T1000")
    dev = torch.device("cuda:0")
    vm = uvad_amd.VadModel(model_name="PyanNet2", model_dict={"encoding_dim": 64})
    vm.model.load_state_dict(sd)
    vm = vm.to(dev).eval()
    batch = {"inputs": torch.from_numpy(g["feats"]).to(dev), "is_voice": torch.zeros(2, 1000)}
    labels = vm.predict_step(batch, 0)
    assert labels.shape == (2, 1000, 1) and labels.dtype == torch.int64
    got = labels.squeeze(-1).cpu().numpy()
    # frames whose probability sits within tolerance of the 0.5 threshold may legitimately flip
    near = np.abs(g["probs"] - 0.5) < 2e-4
    if not near.any():
        assert np.array_equal(got, g["labels49"])
    # the filter itself, on the golden probabilities: exact
    lab2 = uvad_amd.median_filter(torch.from_numpy(g["probs"]).to(dev), window=0.01).cpu().numpy()
    assert np.array_equal(lab2, g["labels49"])


@pytest.mark.parametrize("n_mels,window", [(80, "povey"), (64, "hamming")])
@pytest.mark.parametrize("S", [16000 * 3, 16000 * 2 + 77, 1000])
def test_fbank_matches_oracle(n_mels, window, S):
    import uvad_amd
    from uvad_amd.synth import synth_pcm
    from oracle import c_oracle as co
    dev = torch.device("cuda:0")
    pcm = synth_pcm(3, S, seed=7)
    cfg = co.default_fbank_cfg(n_mels)
    want = co.fbank(pcm, cfg, co.window(window, 400), co.mel_banks(cfg))
    ext = uvad_amd.Fbank(uvad_amd.FbankConfig(sampling_rate=16000, num_filters=n_mels, window_type=window, device="cuda"))
    got = torch.stack(ext.extract_batch(list(torch.from_numpy(pcm).to(dev)), sampling_rate=16000)).cpu().numpy()
    assert got.shape == want.shape == (3, (S + 80) // 160, n_mels)
    err = np.abs(got - want).max()
    print(f"fbank n_mels={n_mels} {window} S={S}: max abs err {err:.2e}")
    assert err < FEAT_TOL


@pytest.mark.parametrize("F", [40, 64, 80])
def test_fbank_arbitrary_mel_matrix_with_weight_on_the_last_bin(F):
    """uvad_set_tables takes ANY (n_mels, 257) matrix.  The kaldi-style tables of the reference carry a zero column at bin 256 and the
    kernel then skips that bin's power (FbankTables::nyquist); a matrix that does weigh it -- here random non-negative bands of random
    width, some reaching bin 256, one filter made of bins 255 and 256 only -- must take the other path and still match the float64
    oracle on the same table.  Also covers band lengths / starts the bank-spreading shift of the mel stage has not seen in the other tests."""
    import uvad_amd
    from oracle import c_oracle as co, torch_ref as tr
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(1000 + F)
    mel = np.zeros((F, 257), np.float32)
    for m in range(F):
        width = int(rng.integers(1, 30))
        lo = int(rng.integers(0, 257 - width + 1)) if m % 5 else 257 - width     # every fifth band ends at bin 256
        mel[m, lo:lo + width] = rng.uniform(0.05, 1.0, width).astype(np.float32)
    mel[F - 1] = 0.0
    mel[F - 1, 255:257] = (0.5, 1.0)
    oc = co.default_fbank_cfg(F)
    win = co.window("povey", 400)
    pcm = tr.synth_pcm(3, 16000 + 77, seed=31)
    rt = uvad_amd.Fbank(uvad_amd.FbankConfig(num_filters=F, window_type="povey", device="cuda"))._runtime(dev)
    rt.set_tables(win, mel)
    got = rt.fbank(torch.from_numpy(pcm).to(dev)).cpu().numpy()
    truth = co.fbank_f64(pcm, oc, win, mel)
    cpu = tr.torch_fbank(pcm, torch.from_numpy(win), torch.from_numpy(mel), frame_shift=160, n_fft=512, preemph=0.97, remove_dc=True, snip_edges=False).numpy()
    e_gpu, e_cpu = float(np.abs(got - truth).max()), float(np.abs(cpu - truth).max())
    print(f"F={F}: arbitrary mel matrix, bin 256 weighted: max err vs f64 GPU {e_gpu:.2e}, torch-CPU {e_cpu:.2e}")
    assert e_gpu < max(FEAT_TOL, 2.0 * e_cpu)
    # the same matrix with the last column cleared takes the skipping path: the filters that never touched bin 256 keep their bits
    mel0 = mel.copy()
    mel0[:, 256] = 0.0
    rt.set_tables(win, mel0)
    got0 = rt.fbank(torch.from_numpy(pcm).to(dev)).cpu().numpy()
    untouched = np.nonzero(mel[:, 256] == 0.0)[0]
    assert len(untouched) > F // 2
    assert np.abs(got0[..., untouched] - truth[..., untouched]).max() < max(FEAT_TOL, 2.0 * e_cpu)
    rt.set_tables(win, co.mel_banks(oc))


def test_mel_matrix_too_wide_for_the_lds_image_is_refused_by_name():
    """A (n_mels, 257) matrix whose longest band does not fit the feature kernel's LDS image (with 80 filters the image's rows are 128
    floats: a dense matrix needs 260 x 128 x 4 = 133 KiB beside the PCM tile and the scratch) is refused by uvad_set_tables with a
    message that names the band length -- not by a failing launch later."""
    import uvad_amd
    from oracle import c_oracle as co
    dev = torch.device("cuda:0")
    rt = uvad_amd.Fbank(uvad_amd.FbankConfig(num_filters=80, window_type="povey", device="cuda"))._runtime(dev)
    dense = np.full((80, 257), 0.01, np.float32)
    with pytest.raises(Exception, match="longest band"):
        rt.set_tables(co.window("povey", 400), dense)
    # a dense matrix of 40 filters (64-float rows: 65 KiB) is taken, and matches the float64 oracle
    from oracle import torch_ref as tr
    rt40 = uvad_amd.Fbank(uvad_amd.FbankConfig(num_filters=40, window_type="povey", device="cuda"))._runtime(dev)
    d40 = np.random.default_rng(5).uniform(0.001, 0.02, (40, 257)).astype(np.float32)
    rt40.set_tables(co.window("povey", 400), d40)
    pcm = tr.synth_pcm(2, 8000, seed=3)
    got = rt40.fbank(torch.from_numpy(pcm).to(dev)).cpu().numpy()
    truth = co.fbank_f64(pcm, co.default_fbank_cfg(40), co.window("povey", 400), d40)
    assert np.abs(got - truth).max() < FEAT_TOL


def test_fbank_log_of_a_normal_floor_equals_ocml_logf_bit_for_bit():
    """fbank_pair.h log_floored(): with an energy floor that is a normal float (the reference's is FLT_EPSILON) the kernel takes
    v_log_f32 + the double-float product with ln 2 without ocml logf's subnormal-argument handling; with a floor below FLT_MIN it
    calls ocml's logf.  Same PCM through both (floors 1.2e-38 and 1e-39; quiet and loud rows, so that energies span 1e-12 .. 1e+3):
    every feature must have the same BITS in both -- both floors lie far below every band energy here, so neither clamps."""
    import uvad_amd
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(11)
    amp = torch.logspace(-5, 0, 12, device=dev).unsqueeze(1)
    pcm = amp * torch.randn(12, 16000 + 123, generator=g, device=dev)
    outs = []
    for floor in (1.2e-38, 1.0e-39):
        rt = uvad_amd.Fbank(uvad_amd.FbankConfig(num_filters=80, window_type="povey", energy_floor=floor, device="cuda"))._runtime(dev)
        outs.append(rt.fbank(pcm))
        assert torch.isfinite(outs[-1]).all()
    lo, hi = float(outs[0].min()), float(outs[0].max())
    print(f"features span [{lo:.1f}, {hi:.1f}] (log of 1.2e-38 = -87.3)")
    assert lo > -60.0 and hi > 0.0                      # nothing was clamped by either floor
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("S", [1000, 3999, 4000, 16000 + 57, 24 * 160 * 3 + 1])
def test_fbank_reads_nothing_outside_its_rows(S):
    """VERDICT r3 #6 (the two GPU memory-access faults of round 3 came from an ABLATION build whose framing loads had their bounds
    guard forced on, profiles/README.md): the shipped kernel's 16-byte framing loads are guarded (fbank.hip: `fast = !virt && g >= 0
    && g + 3 < a.S`, everything else goes through the clamped scalar path).  Checked by value, not by fault: the rows handed to
    uvad_fbank sit between guard rows inside one allocation, and the output may not depend on what the guard rows hold (NaN / huge
    values for f32, +-32767 for int16) -- an out-of-row read at a first or last tile would pull them into the edge frames.  Lengths
    around the 24-frame tile edges and off the 16-byte alignment."""
    import uvad_amd
    dev = torch.device("cuda:0")
    rt = uvad_amd.Fbank(uvad_amd.FbankConfig(num_filters=64, window_type="hamming", device="cuda"))._runtime(dev)
    g = torch.Generator(device=dev).manual_seed(S)
    B = 5
    body = 0.2 * torch.randn(B, S, generator=g, device=dev)
    outs = []
    for poison in (float("nan"), 3.0e38, 0.0):
        buf = torch.full((B + 2, S), poison, device=dev)
        buf[1:B + 1] = body
        outs.append(rt.fbank(buf[1:B + 1]))                    # a contiguous view: rows 1..B of the allocation
        assert torch.isfinite(outs[-1]).all()
    assert torch.equal(outs[0], outs[2]) and torch.equal(outs[1], outs[2])
    body16 = (body * 32767).clamp(-32767, 32767).to(torch.int16)
    o16 = []
    for poison in (32767, -32767, 0):
        buf = torch.full((B + 2, S), poison, dtype=torch.int16, device=dev)
        buf[1:B + 1] = body16
        o16.append(rt.fbank(buf[1:B + 1]))
    assert torch.equal(o16[0], o16[2]) and torch.equal(o16[1], o16[2])


def test_fbank_int16_path_and_edge_cases():
    import uvad_amd
    from oracle import c_oracle as co
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(3)
    i16 = rng.integers(-20000, 20000, size=(2, 16000), dtype=np.int16)
    cfg = co.default_fbank_cfg(80)
    want = co.fbank(i16.astype(np.float32) / 32768.0, cfg, co.window("povey", 400), co.mel_banks(cfg))
    rt = uvad_amd.Fbank(uvad_amd.FbankConfig(device="cuda"))._runtime(dev)
    got = rt.fbank(torch.from_numpy(i16).to(dev)).cpu().numpy()
    assert np.abs(got - want).max() < FEAT_TOL
    # all-zero input: every bin floors at log(eps)
    z = rt.fbank(torch.zeros(1, 4000, device=dev)).cpu().numpy()
    assert np.allclose(z, np.log(np.finfo(np.float32).eps), atol=1e-5)
    # full-scale DC: removed by the per-frame mean -> floor as well
    d = rt.fbank(torch.ones(1, 4000, device=dev)).cpu().numpy()
    assert np.allclose(d, np.log(np.finfo(np.float32).eps), atol=1e-5)


@pytest.mark.parametrize("scale", [4.0, 2.0])
def test_forward_end_to_end_vs_oracle(scale):
    """PCM -> logits through uvad_forward vs oracle fbank + oracle classifier (cfg-2 frame shape, small B), then the
    classifier alone on identical features against the fp32 CPU path and the float64 truth.
    scale 2: contractive network -> the north-star bound, 1e-4 max-abs, against both.
    scale 4: near-chaotic network (see tests/test_gpu_scale.py) -> the GPU path must be as close to the float64 truth as
             the reference's fp32 CPU path is (factor 1.5 on the rms over the sample; the max of 2000 frames of a chaotic
             system is a lottery for ANY fp32 implementation, it is held to 3x the CPU path's, floor 1e-4)."""
    import uvad_amd
    from uvad_amd.synth import synth_pcm, seed_weights
    from oracle import c_oracle as co, torch_ref as tr, parity_stats as ps
    dev = torch.device("cuda:0")
    B, S, F = 5, 16000 * 4, 64      # B not a multiple of the 4-sequence tile
    pcm = synth_pcm(B, S, seed=1000)
    m = uvad_amd.PyanNet2(encoding_dim=F)
    m.build()
    seed_weights(m, 1234, scale)
    m.attach_fbank(uvad_amd.FbankConfig(num_filters=F, window_type="hamming"))
    m = m.to(dev).eval()
    logits, probs = m.forward_waveform(torch.from_numpy(pcm).to(dev))
    cfg = co.default_fbank_cfg(F)
    feats = co.fbank(pcm, cfg, co.window("hamming", 400), co.mel_banks(cfg))
    sd = {k: v.cpu().numpy() for k, v in m.state_dict().items()}
    mc = co.ModelCfg(F, 128, 4, 1, 128, 2, 0.01)
    want_logits, want_probs = co.classify(sd, mc, feats)
    err = np.abs(logits.cpu().numpy() - want_logits).max()
    print(f"x{scale:g}: end-to-end logit err {err:.2e} (range {want_logits.min():.2f}..{want_logits.max():.2f})")
    # features differ by ~1e-4 (fp32 FFT vs float64 DFT oracle) before the classifier amplifies them
    assert err < (5e-3 if scale == 4.0 else 1e-4)
    # same features through the classifiers
    cpu = tr.TorchPyanNet2(F)
    cpu.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    ref32 = cpu(torch.from_numpy(feats))[0].numpy()
    truth = co.classify_f64(sd, mc, feats)
    lg2 = m.forward_logits(torch.from_numpy(feats).to(dev))[0].cpu().numpy()
    st_gpu, st_cpu, st_ref = ps.error_stats(lg2, truth), ps.error_stats(ref32, truth), ps.error_stats(lg2, ref32)
    print("  " + ps.fmt("GPU vs f64 truth     ", st_gpu))
    print("  " + ps.fmt("CPU fp32 vs f64 truth", st_cpu))
    print("  " + ps.fmt("GPU vs CPU fp32      ", st_ref))
    print(f"  f32-state C oracle vs f64 truth: {np.abs(want_logits - truth).max():.2e}")
    if scale == 2.0:
        assert st_ref["max"] < LOGIT_TOL and st_gpu["max"] < LOGIT_TOL
    else:
        assert st_gpu["rms"] <= 1.5 * st_cpu["rms"], (st_gpu, st_cpu)
        assert st_gpu["max"] <= max(3.0 * st_cpu["max"], LOGIT_TOL), (st_gpu, st_cpu)


def test_batch_invariance_full_size_property():
    """Size-independent property at BASELINE cfg-2 scale (B=256 x 10 s): the result for utterance i
    does not depend on which other utterances share the batch (sequences never interact), so a
    full batch, a 4-utterance slice and a permuted batch agree BIT FOR BIT."""
    import uvad_amd
    from uvad_amd.synth import synth_pcm_device, seed_weights
    dev = torch.device("cuda:0")
    B, S, F = 256, 160000, 64
    m = uvad_amd.PyanNet2(encoding_dim=F)
    m.build()
    seed_weights(m, 1234, 4.0)
    m.attach_fbank(uvad_amd.FbankConfig(num_filters=F, window_type="hamming"))
    m = m.to(dev).eval()
    pcm = synth_pcm_device(B, S, seed=42, device=dev)
    full, _ = m.forward_waveform(pcm)
    assert full.shape == (B, 1000) and torch.isfinite(full).all()
    sub, _ = m.forward_waveform(pcm[100:104].contiguous())
    assert torch.equal(full[100:104], sub)
    perm = torch.randperm(B, device=dev)
    pfull, _ = m.forward_waveform(pcm[perm].contiguous())
    assert torch.equal(pfull, full[perm])
    # logits must actually vary (not the 0.506 flat line of default-initialised weights)
    assert full.std().item() > 0.5


def _oracle_windowed(pcm, F, window, sd, scale_cfg, W=80000, keep=48000, frame_shift=0.01):
    """The reference's cut geometry on the CPU oracle: 5 s windows, tails <= 3 s dropped, features of a kept tail padded with
    log(1e-10) frames, classifier + 49-tap median per window row, rows laid end to end and cut to ceil(D / shift) + 1."""
    import math
    from oracle import c_oracle as co
    fc = co.default_fbank_cfg(F)
    win, mel = co.window(window, 400), co.mel_banks(fc)
    mc = co.ModelCfg(F, 128, 4, 1, 128, 2, 0.01)
    probs_rows, label_rows = [], []
    for st in range(0, len(pcm), W):
        x = pcm[st:st + W]
        if len(x) <= keep:
            continue
        feats = co.fbank(x[None], fc, win, mel)
        if feats.shape[1] < W // 160:
            pad = np.full((1, W // 160 - feats.shape[1], F), math.log(1e-10), np.float32)
            feats = np.concatenate([feats, pad], axis=1)
        _, p = co.classify(sd, mc, feats)
        probs_rows.append(p[0])
        label_rows.append(co.median_filter(p, 49)[0])
    n = min(int(math.ceil(len(pcm) / 16000 / frame_shift)) + 1, sum(len(r) for r in label_rows))
    return np.concatenate(probs_rows)[:n], np.concatenate(label_rows)[:n]


def test_main_config1_plumbing_30s():
    """BASELINE cfg 1: one 30 s utterance through main.main(load_config()) -> frame labels + intervals, with the reference's
    cut geometry (six 5 s windows, zero LSTM state in each), checked against the oracle run window by window."""
    import main as entry
    from config.config import load_config
    from uvad_amd.synth import synth_pcm
    from oracle import c_oracle as co, torch_ref as tr
    cfg = load_config()
    cfg.model_dict.encoding_dim = 64
    cfg.input.seconds = 30.0
    cfg.window_type = "hamming"
    res = entry.main(cfg)
    assert len(res) == 1 and res[0]["num_frames"] == 3000
    pcm = synth_pcm(1, 480000, seed=cfg.input.seed)
    sd = {k: v.numpy() for k, v in tr.seeded_state_dict(64, seed=cfg.weights_seed, scale=cfg.weights_scale).items()}
    probs, want = _oracle_windowed(pcm[0], 64, "hamming", sd, None)
    assert np.abs(res[0]["probs"] - probs).max() < 2e-3          # end to end from PCM on the x4 network (fp32 FFT vs f64 DFT features)
    agree = (want == res[0]["labels"]).mean()
    print("label agreement with oracle:", agree, "intervals:", res[0]["intervals"][:3])
    assert agree > 0.995
    assert res[0]["intervals"] == co.intervals(res[0]["labels"], 0.01)
    # whole-recording mode is a different computation (one BiLSTM pass over 3000 frames): available, not the default
    cfg.window_seconds = None
    whole = entry.main(cfg)
    assert whole[0]["num_frames"] == 3000 and np.abs(whole[0]["probs"] - res[0]["probs"]).max() > 1e-2


def test_predict_vad_reference_window_geometry_from_wav(tmp_path):
    """predict_vad on int16 wav files of awkward lengths (23.7 s: four full windows + a kept 3.7 s tail; 12 s: two windows,
    the 2 s tail dropped; 4.2 s: one kept tail only; 2 s: nothing) == the oracle run window by window with lhotse's feature
    padding; several batches in flight (max_duration forces > 1 batch); weights x2 so that the comparison is tight."""
    import wave
    from config.config import load_config
    from src.scripts import predict_vad
    from uvad_amd.synth import synth_pcm
    from oracle import torch_ref as tr
    lens = {"a.wav": int(23.7 * 16000), "b.wav": 12 * 16000, "c.wav": int(4.2 * 16000), "d.wav": 2 * 16000}
    pcm = {}
    for k, (name, n) in enumerate(lens.items()):
        x = synth_pcm(1, n, seed=300 + k)[0]
        q = np.round(x * 32767.0).astype("<i2")
        with wave.open(str(tmp_path / name), "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(q.tobytes())
        pcm[name] = q.astype(np.float32) / 32768.0
    cfg = load_config()
    cfg.model_dict.encoding_dim = 80                      # the reference's fbank width, povey window
    cfg.weights_scale = 2.0
    cfg.max_duration = 12                                  # two 5 s windows per batch -> 3 full-window batches + tails
    cfg.input.kind = "wav"
    cfg.input.paths = [str(tmp_path / n) for n in lens]
    res = {r["recording_id"]: r for r in predict_vad(**cfg)}
    sd = {k: v.numpy() for k, v in tr.seeded_state_dict(80, seed=cfg.weights_seed, scale=2.0).items()}
    assert res["d.wav"]["num_frames"] == 0 and res["d.wav"]["intervals"] == []
    for name, want_frames in (("a.wav", 2371), ("b.wav", 1000), ("c.wav", 421)):
        probs, labels = _oracle_windowed(pcm[name], 80, "povey", sd, None)
        r = res[name]
        assert r["num_frames"] == want_frames == len(labels), (name, r["num_frames"], len(labels))
        err = np.abs(r["probs"] - probs).max()
        near = np.abs(probs - 0.5) < 1e-3
        print(f"{name}: {r['num_frames']} frames, prob err {err:.2e}, {int((r['labels'] != labels).sum())} label differences")
        assert err < 1e-4
        if not near.any():
            assert np.array_equal(r["labels"], labels)


def test_predict_vad_without_concurrent_streams_gives_the_same_predictions(tmp_path, monkeypatch):
    """with streams that never overlap (as with GPU_MAX_HW_QUEUES = 1) predict_vad runs batch after batch and returns
    the predictions of the pipelined run."""
    from config.config import load_config
    from src.scripts import predict_vad
    from uvad_amd.runtime import VadRuntime
    cfg = load_config()
    cfg.model_dict.encoding_dim = 64
    cfg.weights_scale = 2.0
    cfg.max_duration = 12
    cfg.input.kind = "synthetic"
    cfg.input.num_utterances, cfg.input.seconds, cfg.input.seed = 3, 16.0, 77
    want = predict_vad(**cfg)
    monkeypatch.setattr(VadRuntime, "streams_overlap", lambda self, a, b: False)
    got = predict_vad(**cfg)
    assert len(got) == len(want) == 3
    for g, w in zip(got, want):
        assert g["num_frames"] == w["num_frames"] > 0 and np.array_equal(g["labels"], w["labels"]) and g["intervals"] == w["intervals"]
        assert np.array_equal(g["probs"], w["probs"])


@pytest.mark.parametrize("F,B,S,window", [(64, 5, 16000 * 3 + 123, "hamming"), (80, 3, 8000, "povey"), (40, 6, 4800, "povey"), (64, 64, 160000, "hamming")])
def test_forward_writes_the_operand_planes_directly_with_the_bits_of_the_split_path(F, B, S, window):
    """VERDICT r2 #5 / SURVEY 7 step 6: in uvad_forward the feature kernel writes the two K-blocked f16 planes the first projection
    reads (tile-major rows, padding columns and padding sequences zeroed) -- no f32 feature tensor, no split pass.  The logits must
    equal, bit for bit, those of uvad_classify on the f32 features of uvad_fbank (which go through split_features_kernel): batch sizes
    that are not a multiple of the sequence tile, feature widths that need padding columns (80 -> 96, 40 -> 64), ragged sample
    counts."""
    import uvad_amd
    from uvad_amd.synth import seed_weights, synth_pcm_device
    dev = torch.device("cuda:0")
    m = uvad_amd.PyanNet2(encoding_dim=F)
    m.build()
    seed_weights(m, 1234, 4.0)
    m.attach_fbank(uvad_amd.FbankConfig(num_filters=F, window_type=window))
    m = m.to(dev).eval()
    rt = m.runtime(dev)
    pcm = synth_pcm_device(B, S, seed=91, device=dev)
    fused, fprobs = rt.forward(pcm)
    fused, fprobs = fused.clone(), fprobs.clone()
    y_f, z_f = (t.clone() for t in rt.taps())
    split, sprobs = rt.classify(rt.fbank(pcm))
    y_s, z_s = rt.taps()
    assert torch.isfinite(fused).all()
    assert torch.equal(fused, split) and torch.equal(fprobs, sprobs) and torch.equal(y_f, y_s) and torch.equal(z_f, z_s)
    q = torch.round(pcm * 32767.0).to(torch.int16)
    assert torch.equal(rt.forward(q)[0], rt.classify(rt.fbank(q))[0])       # the int16 ingest path as well


def test_library_is_loaded_and_errors_are_loud():
    import uvad_amd
    from uvad_amd import _lib
    lib = _lib.load()
    maps = open("/proc/self/maps").read()
    assert "libuvad.so" in maps
    dev = torch.device("cuda:0")
    m = uvad_amd.PyanNet2(encoding_dim=64)
    m.build()
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 10, 64))              # CPU tensor: error, not fallback
    m = m.to(dev)
    with pytest.raises(ValueError):
        m(torch.zeros(1, 10, 80, device=dev))  # wrong feature width
    rt = m.runtime(dev)
    with pytest.raises(_lib.UvadError) as ei:
        _lib.check(lib, rt.ctx, lib.uvad_classify(rt.ctx, torch.zeros(4, device=dev).data_ptr(), 1, 10, None, None,
                                                  torch.zeros(4, device=dev).data_ptr(), 16, None))
    assert ei.value.code == -4                 # workspace too small


@pytest.mark.parametrize("chunk", [320, 160, 1600])
def test_streaming_equals_offline_causal_model(chunk):
    """BASELINE cfg 5 semantics: B lock-step streams fed `chunk` samples per step through uvad_stream_step give,
    frame for frame, the logits of the offline path on the whole signal (causal model => identical by causality),
    and match the reference's unidirectional PyanNet2 (golden pyannet2_uni) operator sequence in the oracle."""
    import uvad_amd
    from uvad_amd.synth import synth_pcm, seed_weights
    from oracle import c_oracle as co
    dev = torch.device("cuda:0")
    B, S, F = 6, 16000 * 2, 64
    pcm = synth_pcm(B, S, seed=77)
    m = uvad_amd.PyanNet2(lstm={"bidirectional": False}, encoding_dim=F)
    m.build()
    seed_weights(m, 1234, 4.0)
    m.attach_fbank(uvad_amd.FbankConfig(num_filters=F, window_type="povey"))
    m = m.to(dev).eval()
    rt = m.runtime(dev)
    x = torch.from_numpy(pcm).to(dev)
    offline, _ = rt.forward(x)
    st = rt.stream_open(B, chunk)
    outs = []
    for i in range(S // chunk):
        outs.append(rt.stream_step(st, x[:, i * chunk:(i + 1) * chunk].contiguous()).clone())
    got = torch.cat(outs, dim=1)
    n = got.shape[1]
    T = S // 160
    assert n == (S + 120 - 400) // 160 + 1      # every frame whose last sample has arrived
    assert n >= T - 2
    # Every emitted frame equals the offline logit up to fp32 rounding: the chunked feature kernel pairs
    # frames differently inside its two-frames-per-FFT trick (1e-7 feature differences, amplified by the net).
    sdiff = float((got - offline[:, :n]).abs().max())
    print(f"streaming chunk={chunk}: {n} frames, max |stream - offline| = {sdiff:.2e}")
    assert sdiff < LOGIT_TOL
    # and the offline causal path itself against the oracle
    cfg = co.default_fbank_cfg(F)
    feats = co.fbank(pcm, cfg, co.window("povey", 400), co.mel_banks(cfg))
    sd = {k: v.cpu().numpy() for k, v in m.state_dict().items()}
    want, _ = co.classify(sd, co.ModelCfg(F, 128, 4, 0, 128, 2, 0.01), feats)
    assert np.abs(offline.cpu().numpy() - want).max() < 5e-3
    lg2, _ = m.forward_logits(torch.from_numpy(feats).to(dev))
    assert np.abs(lg2.cpu().numpy() - want).max() < LOGIT_TOL


@pytest.mark.parametrize("chunk", [320, 250, 1600])
def test_streaming_graph_replay_equals_kernel_by_kernel_enqueue(chunk):
    """stream_step replays a hipGraph per distinct (k, offset, parity, first) of uvad_stream_peek and moves the counters with
    uvad_stream_advance: bit-identical to enqueuing every step kernel by kernel (graphs=False), for a chunk that is a multiple
    of the shift (two graphs after the first step) and one that is not (250 samples: the step shapes cycle)."""
    import uvad_amd
    from uvad_amd.synth import synth_pcm, seed_weights
    dev = torch.device("cuda:0")
    B, S, F = 5, 16000, 64
    x = torch.from_numpy(synth_pcm(B, S, seed=78)).to(dev)
    m = uvad_amd.PyanNet2(lstm={"bidirectional": False}, encoding_dim=F)
    m.build()
    seed_weights(m, 1234, 4.0)
    m.attach_fbank(uvad_amd.FbankConfig(num_filters=F, window_type="povey"))
    rt = m.to(dev).eval().runtime(dev)
    runs = {}
    for graphs in (False, True):
        st = rt.stream_open(B, chunk, graphs=graphs)
        outs = [rt.stream_step(st, x[:, i * chunk:(i + 1) * chunk].contiguous()).clone() for i in range(S // chunk)]
        runs[graphs] = torch.cat(outs, dim=1)
        if graphs:
            print(f"chunk {chunk}: {len(st['graphs'])} graphs for {S // chunk} steps")
            assert len(st["graphs"]) < S // chunk
    assert runs[True].shape == runs[False].shape and torch.equal(runs[True], runs[False])


def test_streaming_rejects_bidirectional_and_unreset_state():
    import uvad_amd
    from uvad_amd import _lib
    dev = torch.device("cuda:0")
    m = uvad_amd.PyanNet2(encoding_dim=64)
    m.build()
    m.attach_fbank(uvad_amd.FbankConfig(num_filters=64))
    rt = m.to(dev).runtime(dev)
    with pytest.raises(_lib.UvadError) as ei:
        rt.stream_open(4, 320)
    assert ei.value.code == -5


@pytest.mark.parametrize("mode", ["f32", "f16p"])
@pytest.mark.parametrize("name", ["pyannet2_f64_T1000", "pyannet2_f80_T500", "pyannet2_f64_T3000"])
def test_all_gemm_modes_meet_the_logit_bound(mode, name):
    g, sd, case = load_golden(name)
    dev = torch.device("cuda:0")
    m = _model(case, sd, dev)
    rt = m.runtime(dev)
    rt.set_gemm_mode(mode)
    logits, _ = m.forward_logits(torch.from_numpy(g["feats"]).to(dev))
    err = np.abs(logits.cpu().numpy() - g["logits"]).max()
    print(f"gemm mode {mode}, {name}: logit err vs reference golden {err:.2e}")
    assert err < LOGIT_TOL


def test_f16p_gemm_is_f32_accurate_and_handles_awkward_operands():
    """The 2-way f16 split against the exact f32-MFMA kernel on one projection-shaped product with operands spanning
    the magnitudes of the path (tiny activations, log-mel-sized features, weights x4): relative error of the same order
    as f32 accumulation noise.  Operands OUTSIDE the f16 range never reach the split: a weight >= 65504 makes the context
    run the exact kernel (host check in uvad_finalize), a feature >= 65504 or Inf handed to uvad_classify makes that
    call's first projection run the exact kernel (device-side flag), with the answer of the exact mode in both cases."""
    import uvad_amd
    from oracle import torch_ref as tr
    dev = torch.device("cuda:0")
    sd = tr.seeded_state_dict(64, 128, 1, False, lin_layers=0, seed=5, scale=4.0)
    m = uvad_amd.PyanNet2(lstm={"num_layers": 1, "bidirectional": False}, linear={"num_layers": 0}, encoding_dim=64)
    m.build()
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    g = torch.Generator().manual_seed(3)
    feats = torch.randn(8, 300, 64, generator=g) * 4.0 - 8.0
    feats[:, :, :8] *= 1e-4                 # tiny columns
    feats[:, :, 8:16] = feats[:, :, 8:16] * 3.0   # up to ~ +-40
    out = {}
    for mode in ("f32", "f16p"):
        m.runtime(dev).set_gemm_mode(mode)
        out[mode], _ = m.forward_logits(feats.to(dev))
    torch.cuda.synchronize()
    e16 = (out["f16p"] - out["f32"]).abs().max().item()
    print(f"single layer: f16p vs exact f32 MFMA {e16:.2e}")
    assert e16 < 2e-5
    # (a) features outside the f16 range (the reference accepts any float; e.g. unnormalised 768-dim SSL features)
    big = feats.clone()
    big[1, 7, 3] = 1.0e5
    big[5, 100, 60] = -7.0e4
    for mode in ("f32", "f16p"):
        m.runtime(dev).set_gemm_mode(mode)
        out[mode], _ = m.forward_logits(big.to(dev))
    assert torch.isfinite(out["f16p"]).all()
    assert torch.equal(out["f16p"], out["f32"])          # single layer, no feed-forward: the whole model ran the exact kernel
    assert (out["f32"][1] - m.forward_logits(feats.to(dev))[0][1]).abs().max() > 1e-2     # the big value did matter
    inf = feats.clone()
    inf[2, 5, 0] = float("inf")
    m.runtime(dev).set_gemm_mode("f16p")
    li, _ = m.forward_logits(inf.to(dev))
    ref = tr.TorchPyanNet2(64, 128, 1, False, lin_layers=0)
    ref.load_state_dict(sd)
    want = ref(inf)[0]
    assert torch.equal(torch.isfinite(li.cpu()), torch.isfinite(want))     # non-finite exactly where torch's are
    ok = torch.isfinite(want)
    assert (li.cpu()[ok] - want[ok]).abs().max() < LOGIT_TOL
    # (b) a huge weight is fine for the split (each matrix carries its own power-of-two scale) ...
    sd2 = {k: v.clone() for k, v in sd.items()}
    sd2["lstm.weight_ih_l0"][0, 0] = 1.0e5
    m.load_state_dict(sd2)
    m.runtime(dev).set_gemm_mode("f16p")
    l16, _ = m.forward_logits(feats.to(dev))
    m.runtime(dev).set_gemm_mode("f32")
    l32, _ = m.forward_logits(feats.to(dev))
    assert torch.isfinite(l16).all() and (l16 - l32).abs().max() < 2e-5
    # ... while feed-forward weights so large that the activations BETWEEN the feed-forward layers can leave the f16 range
    # (|z| <= sum_k |w_jk| + |b_j| is checked by uvad_finalize) make the context run the exact kernel everywhere
    sd3 = tr.seeded_state_dict(64, 128, 1, False, lin_layers=2, seed=6, scale=2.0)
    sd3["linear.0.weight"] *= 2.0e4
    m3 = uvad_amd.PyanNet2(lstm={"num_layers": 1, "bidirectional": False}, encoding_dim=64)
    m3.build()
    m3.load_state_dict(sd3)
    m3 = m3.to(dev).eval()
    m3.runtime(dev).set_gemm_mode("f16p")
    z16, _ = m3.forward_logits(feats.to(dev))
    m3.runtime(dev).set_gemm_mode("f32")
    z32, _ = m3.forward_logits(feats.to(dev))
    assert torch.isfinite(z16).all() and torch.equal(z16, z32)


def test_finalize_twice_hot_swaps_weights():
    """uvad_finalize is idempotent: loading a second state_dict into the same context (weight hot-swap through the C ABI)
    replaces the device copy -- results follow the new weights, and the first set can be restored bit for bit."""
    import uvad_amd
    from oracle import torch_ref as tr
    dev = torch.device("cuda:0")
    sd_a = tr.seeded_state_dict(64, seed=11, scale=2.0)
    sd_b = tr.seeded_state_dict(64, seed=12, scale=2.0)
    m = uvad_amd.PyanNet2(encoding_dim=64)
    m.build()
    m.load_state_dict(sd_a)
    m = m.to(dev).eval()
    x = torch.randn(3, 50, 64, generator=torch.Generator().manual_seed(1)).to(dev) * 2 - 3
    la = m.forward_logits(x)[0].clone()
    rt = m.runtime(dev)
    cpu = tr.TorchPyanNet2(64); cpu.load_state_dict(sd_b)

    def cycle():
        rt.load_state_dict(sd_b)
        lb = m.forward_logits(x)[0].clone()
        rt.load_state_dict(sd_a)
        la2 = m.forward_logits(x)[0].clone()
        return lb, la2

    def check(lb, la2):
        torch.cuda.synchronize()
        assert not torch.equal(la, lb) and torch.equal(la, la2)
        assert (lb.cpu() - cpu(x.cpu())[0]).abs().max() < LOGIT_TOL

    # one cycle and its checks BEFORE the baseline: torch's first torch.equal / .cpu() of a process take device memory of their own (16 MiB when
    # this test runs first or alone), which is not this library's
    check(*cycle())
    free0 = torch.cuda.mem_get_info(dev)[0]
    for _ in range(20):
        lb, la2 = cycle()
    check(lb, la2)
    leaked = free0 - torch.cuda.mem_get_info(dev)[0]
    assert leaked < 8 << 20, f"40 reloads leaked {leaked} bytes of device memory"


def test_contexts_with_identical_weights_share_one_upload_and_stay_independent():
    """uvad_finalize: contexts of the process that hold identical tensors / configuration on one device share the packed weights (one host-side
    packing and upload instead of one per context: uvad_weights_shared_by).  Sharing must be invisible: same logits bit for bit, a hot-swap in ONE
    context leaves the others on the old weights and on their own block, a block outlives the context that created it, and a context that swaps
    back joins the shared block again."""
    import time
    import uvad_amd
    from oracle import torch_ref as tr
    dev = torch.device("cuda:0")
    sd_a = tr.seeded_state_dict(64, seed=21, scale=2.0)
    sd_b = tr.seeded_state_dict(64, seed=22, scale=2.0)
    cfg = {"encoding_dim": 64, "lstm": None, "linear": None}
    m = uvad_amd.PyanNet2(encoding_dim=64)
    m.build()
    cfg = {"encoding_dim": 64, "lstm": m.hparams.lstm, "linear": m.hparams.linear}
    x = torch.randn(5, 70, 64, generator=torch.Generator().manual_seed(2)).to(dev) * 2 - 3
    rts = []
    t_first = t_next = 0.0
    for k in range(4):
        rt = uvad_amd.VadRuntime(device=dev, fbank=None, model=cfg)
        t0 = time.perf_counter()
        rt.load_state_dict(sd_a)
        dt = time.perf_counter() - t0
        t_first, t_next = (dt, t_next) if k == 0 else (t_first, t_next + dt / 3)
        rts.append(rt)
    assert [r.weights_shared_by() for r in rts] == [4, 4, 4, 4]
    print(f"load_state_dict + uvad_finalize: first context {t_first * 1e3:.1f} ms, the ones that share its upload {t_next * 1e3:.1f} ms each")
    la = [r.classify(x, want_probs=False)[0].clone() for r in rts]
    assert all(torch.equal(la[0], l) for l in la[1:])
    cpu = tr.TorchPyanNet2(64); cpu.load_state_dict(sd_a)
    assert (la[0].cpu() - cpu(x.cpu())[0]).abs().max() < LOGIT_TOL
    # a hot-swap in one context: its own block, the others untouched
    rts[1].load_state_dict(sd_b)
    assert [r.weights_shared_by() for r in rts] == [3, 1, 3, 3]
    lb = rts[1].classify(x, want_probs=False)[0].clone()
    cpu_b = tr.TorchPyanNet2(64); cpu_b.load_state_dict(sd_b)
    assert (lb.cpu() - cpu_b(x.cpu())[0]).abs().max() < LOGIT_TOL and not torch.equal(lb, la[0])
    assert torch.equal(rts[0].classify(x, want_probs=False)[0], la[0]) and torch.equal(rts[3].classify(x, want_probs=False)[0], la[0])
    # the context that created the block goes away: the block stays for the others
    torch.cuda.synchronize()
    rts[0].close()
    assert [r.weights_shared_by() for r in rts[1:]] == [1, 2, 2]
    assert torch.equal(rts[2].classify(x, want_probs=False)[0], la[0])
    # swapping back joins the shared block again
    rts[1].load_state_dict(sd_a)
    assert [r.weights_shared_by() for r in rts[1:]] == [3, 3, 3]
    assert torch.equal(rts[1].classify(x, want_probs=False)[0], la[0])
    for r in rts[1:]:
        r.close()


def test_label_runs_on_device_equal_the_host_walk_and_the_oracle():
    """uvad_label_runs + labels_to_intervals_batch == labels_to_intervals (numpy restatement of predict.py:472-490)
    == the C oracle's orc_intervals, on random rows and the edge cases (empty, all speech, run open at the end,
    single-frame runs, alternating frames = the (T+1)/2 maximum, T not a multiple of 64, T = 1)."""
    import uvad_amd
    from oracle import c_oracle
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(11)
    for T in (1, 7, 64, 65, 1000, 3001):
        rows = [np.zeros(T, np.uint8), np.ones(T, np.uint8), (np.arange(T) % 2).astype(np.uint8), ((np.arange(T) + 1) % 2).astype(np.uint8)]
        for p in (0.02, 0.3, 0.9):
            r = (rng.random(T) < 0.5).astype(np.uint8)
            for _ in range(3 if T >= 5 else 0):   # smooth into runs of assorted lengths
                r = (np.convolve(r, np.ones(5), "same") > 5 * p).astype(np.uint8)
            rows.append(r)
        tail = np.zeros(T, np.uint8); tail[T // 2:] = 1
        rows.append(tail)
        lab = np.stack(rows)
        got = uvad_amd.labels_to_intervals_batch(torch.from_numpy(lab).to(dev), 0.01)
        runs, counts = uvad_amd.postprocess._shared_runtime(dev).label_runs(torch.from_numpy(lab).to(dev))
        for b in range(lab.shape[0]):
            assert got[b] == uvad_amd.labels_to_intervals(lab[b], 0.01), (T, b)
            assert got[b] == [tuple(x) for x in c_oracle.intervals(lab[b], 0.01)], (T, b)
            d = np.diff(np.concatenate(([0], lab[b].astype(np.int8), [0])))
            assert int(counts[b]) == int((d == 1).sum())
    # overflow is reported, not silently truncated: counts holds the true number of runs
    alt = torch.from_numpy((np.arange(200) % 2).astype(np.uint8)[None]).to(dev)
    runs, counts = uvad_amd.postprocess._shared_runtime(dev).label_runs(alt, max_runs=8)
    assert int(counts[0]) == 100 and runs.shape == (1, 8, 2)
    assert runs[0, :, 0].cpu().tolist() == [1, 3, 5, 7, 9, 11, 13, 15]


def test_detection_error_counts_match_numpy():
    """predict.py:666-673 on the GPU: FA = #(gt==0 & pred==1)/N, MD = #(gt==1 & pred==0)/N, DER = FA + MD."""
    import uvad_amd
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(9)
    for B, T in [(5, 1000), (3, 37), (1, 16), (7, 3001)]:
        pred = (rng.random((B, T)) < 0.4).astype(np.uint8)
        gt = (rng.random((B, T)) < 0.5).astype(np.uint8)
        out = uvad_amd.detection_error(torch.from_numpy(pred).to(dev), torch.from_numpy(gt).to(dev))
        fa = ((gt == 0) & (pred == 1)).sum(1) / T
        md = ((gt == 1) & (pred == 0)).sum(1) / T
        assert np.allclose(out["false_alarm"].cpu().numpy(), fa, atol=0, rtol=1e-12)
        assert np.allclose(out["missed_detection"].cpu().numpy(), md, atol=0, rtol=1e-12)
        assert np.allclose(out["detection_error_rate"].cpu().numpy(), fa + md, atol=0, rtol=1e-12)


@pytest.mark.parametrize("cfg", [
    dict(F=64, H=64, L=2, bi=True, lin_h=32, lin_l=1, B=5, T=37),      # 64-unit kernel (4 waves), odd batch, short T
    dict(F=80, H=128, L=1, bi=False, lin_h=128, lin_l=0, B=1, T=1),    # single frame, no feed-forward, causal
    dict(F=768, H=128, L=2, bi=True, lin_h=128, lin_l=2, B=3, T=50),   # SSL-feature width of the reference (encoding_dim=768)
    dict(F=60, H=128, L=4, bi=True, lin_h=64, lin_l=3, B=9, T=5),      # K not a multiple of 32 (SincNet width), 3 FC layers
    dict(F=64, H=128, L=4, bi=True, lin_h=128, lin_l=2, B=2, T=300),   # T >= 256 (chunk-capable length)
    # hidden sizes without a register-resident kernel (VERDICT r3 missing #6: the reference constructor takes any): lstm_rec_any_kernel
    dict(F=64, H=32, L=2, bi=True, lin_h=128, lin_l=2, B=6, T=40),
    dict(F=80, H=96, L=3, bi=False, lin_h=64, lin_l=1, B=3, T=33),     # one direction: 96 units = 384 gate columns
    dict(F=64, H=256, L=2, bi=True, lin_h=128, lin_l=2, B=5, T=25),    # more units than threads' first pass covers evenly
    dict(F=40, H=48, L=1, bi=True, lin_h=128, lin_l=0, B=2, T=9),      # 48 x 2 directions = 384 gate columns, no feed-forward
])
def test_model_variants_vs_oracle(cfg):
    """Constructor variants the reference allows (PyanNet2.py:69-139) on seeded weights, against the C oracle
    (the oracle itself is pinned to the reference class by the goldens for the default / uni / 1-layer shapes)."""
    import uvad_amd
    from uvad_amd.synth import seed_weights
    from oracle import c_oracle as co
    dev = torch.device("cuda:0")
    m = uvad_amd.PyanNet2(lstm={"hidden_size": cfg["H"], "num_layers": cfg["L"], "bidirectional": cfg["bi"]},
                          linear={"hidden_size": cfg["lin_h"], "num_layers": cfg["lin_l"]}, encoding_dim=cfg["F"])
    m.build()
    seed_weights(m, 77, 2.0)
    m = m.to(dev).eval()
    g = torch.Generator().manual_seed(5)
    feats = torch.randn(cfg["B"], cfg["T"], cfg["F"], generator=g) * 2.0 - 3.0
    logits, probs = m.forward_logits(feats.to(dev))
    sd = {k: v.cpu().numpy() for k, v in m.state_dict().items()}
    mc = co.ModelCfg(cfg["F"], cfg["H"], cfg["L"], int(cfg["bi"]), cfg["lin_h"], cfg["lin_l"], 0.01)
    want, wantp = co.classify(sd, mc, feats.numpy())
    err = np.abs(logits.cpu().numpy() - want).max()
    print(f"variant {cfg}: logit err {err:.2e}")
    assert err < LOGIT_TOL
    assert np.abs(probs.cpu().numpy() - wantp).max() < LOGIT_TOL
    if cfg["H"] not in (64, 128):
        rt = m.runtime(dev)
        rt.set_gemm_mode("f32")                                  # the exact-f32 projections in front of the same generic recurrence
        l32 = m.forward_logits(feats.to(dev))[0]
        assert np.abs(l32.cpu().numpy() - want).max() < LOGIT_TOL
        rt.set_gemm_mode("f16p")
        with pytest.raises(RuntimeError, match="hidden_size 128"):
            rt.set_recurrent_tile(16)                           # the 16-sequence form stays an H = 128 kernel


def test_hidden_sizes_the_gate_matrix_layout_cannot_take_are_refused_loudly():
    """hidden_size x directions must be a multiple of 32 (the gate matrix comes in whole 128-column tiles): anything else is an error at
    construction time, never a silent fallback."""
    import uvad_amd
    dev = torch.device("cuda:0")
    for lstm in ({"hidden_size": 100}, {"hidden_size": 72, "bidirectional": False}, {"hidden_size": 2048}):
        m = uvad_amd.PyanNet2(lstm=lstm, encoding_dim=64)
        m.build()
        with pytest.raises(RuntimeError, match="multiple of 32"):
            m.to(dev).eval().runtime(dev)


def test_tile16_throughput_kernel_matches_tile4():
    """The 16-sequence recurrent kernel (chosen by itself from B >= 1024; forced here with uvad_set_recurrent_tile) against
    the 4-sequence kernel and the reference golden, incl. a partial last workgroup (21 sequences = 6 tiles)."""
    import uvad_amd
    g, sd, case = load_golden("pyannet2_f64_T1000")
    dev = torch.device("cuda:0")
    m = _model(case, sd, dev)
    rt = m.runtime(dev)
    x = torch.from_numpy(g["feats"]).to(dev)
    extra = (torch.randn(19, 1000, 64, generator=torch.Generator().manual_seed(1)) * 4 - 8).to(dev)
    xs = torch.cat([x, extra])
    outs = {}
    for tile in (4, 16):
        rt.set_recurrent_tile(tile)
        l, _ = m.forward_logits(xs)
        assert rt.recurrent_tile() == tile
        err = float(np.abs(l[:2].cpu().numpy() - g["logits"]).max())
        print(f"tile {tile}: logit err vs reference golden {err:.2e}")
        assert err < LOGIT_TOL
        outs[tile] = l.clone()
    rt.set_recurrent_tile(0)
    l, _ = m.forward_logits(xs)
    assert rt.recurrent_tile() == 4 and torch.equal(l, outs[4])     # 21 sequences: the latency form by default
    d = float((outs[4] - outs[16]).abs().max())
    print(f"tile16 vs tile4: max diff {d:.2e}")
    assert d < LOGIT_TOL


def test_throughput_recurrence_fourth_product_on_the_fp8_pipe_equals_its_f16_form():
    """lstm_rec16h_kernel<.., 4, true>: the P2 x h product on v_mfma_scale_f32_16x16x128_f8f6f4 (uvad_finalize verifies that every element of
    every layer's P2 plane is exactly a bf8 number; uvad_get_p2_on_fp8).  GEMM mode 2 -- the kernel set kept for comparisons -- runs the same
    recurrence with P2 from its f16 image (the form every mode ran before round 4): both must reproduce the reference's golden logits, and
    they must agree with each other far inside the bound (same sums; the fp8 form rounds h to 4 bits in a term that is 2^-22 of the sum)."""
    import uvad_amd
    g, sd, case = load_golden("pyannet2_f64_T1000")
    dev = torch.device("cuda:0")
    m = _model(case, sd, dev)
    rt = m.runtime(dev)
    x = torch.from_numpy(g["feats"]).to(dev).repeat(10, 1, 1)[:19]         # 19 sequences: a full and a partial 16-sequence workgroup per direction
    outs = {}
    for mode, want_fp8 in (("f16p", True), ("f16p_stream", False)):
        rt.set_gemm_mode(mode)
        rt.set_recurrent_tile(16)
        assert rt.p2_on_fp8() == want_fp8, mode
        l, _ = m.forward_logits(x)
        assert rt.recurrent_tile() == 16
        err = float(np.abs(l[:2].cpu().numpy() - g["logits"]).max())
        print(f"mode {mode}: p2_on_fp8 {rt.p2_on_fp8()}, logit err vs reference golden {err:.2e}")
        assert err < LOGIT_TOL
        outs[mode] = l.clone()
    rt.set_gemm_mode("f16p")
    rt.set_recurrent_tile(0)
    d = float((outs["f16p"] - outs["f16p_stream"]).abs().max())
    print(f"fp8 form vs f16 form of the P2 product: max logit diff {d:.2e}")
    assert d < 0.05 * LOGIT_TOL


@pytest.mark.parametrize("B,T,bidir,fc", [(40, 37, True, 2), (1, 1, True, 2), (33, 2, True, 2), (64, 129, False, 2), (17, 300, True, 0),
                                         (32, 50, True, 0), (64, 129, False, 0), (17, 3, True, 0)])
def test_recurrent_forms_agree_on_ragged_shapes(B, T, bidir, fc):
    """4- and 16-sequence recurrent kernels on batches that are not multiples of either tile, frame counts around the 128-row
    blocks of the gate matrix, one direction, and the last LSTM layer's exact-f32 output (no feed-forward layers): they agree to
    rounding, and each is bit-reproducible run to run (the kernels keep no branch between an MFMA and the read of its result)."""
    import uvad_amd
    from uvad_amd.synth import seed_weights
    dev = torch.device("cuda:0")
    m = uvad_amd.PyanNet2(encoding_dim=64, lstm={"bidirectional": bidir}, linear={"num_layers": fc})
    m.build()
    seed_weights(m, 1234, 2.0)
    m = m.to(dev).eval()
    rt = m.runtime(dev)
    x = torch.randn(B, T, 64, device=dev, generator=torch.Generator(device=dev).manual_seed(B * 1000 + T)) * 3
    outs = {}
    for tile in (4, 16):
        rt.set_recurrent_tile(tile)
        runs = [rt.classify(x, want_probs=False)[0].clone() for _ in range(3)]
        assert rt.recurrent_tile() == tile
        assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2]), f"tile {tile} differs run to run"
        outs[tile] = runs[0]
    rt.set_recurrent_tile(0)
    d = float((outs[4] - outs[16]).abs().max())
    assert torch.isfinite(outs[16]).all() and d < 1e-5, d


def test_forward_pipeline_results_equal_sequential_path():
    """uvad_amd.ForwardPipeline (several uvad_forward calls in flight on HIP streams proven concurrent by uvad_streams_overlap): every batch's logits are
    bit-identical to the ones the plain sequential path produces, whatever slot ran them."""
    import uvad_amd
    from uvad_amd.synth import seed_weights, synth_pcm_device
    dev = torch.device("cuda:0")
    m = uvad_amd.PyanNet2(encoding_dim=64)
    m.build()
    seed_weights(m, 1234, 4.0)
    m.attach_fbank(uvad_amd.FbankConfig(num_filters=64, window_type="hamming"))
    m = m.to(dev).eval()
    batches = [synth_pcm_device(24, 48000, 70 + i, dev) for i in range(5)]
    want = [m.forward_waveform(b)[0].clone() for b in batches]
    pipe = uvad_amd.ForwardPipeline(m, dev, depth=2)
    pend = [pipe.submit(b) for b in batches]
    got = [p.result()[0] for p in pend]
    assert pipe.streams is not None and len(pipe.streams) == 2 and pipe.streams[0] != pipe.streams[1]
    for w, g in zip(want, got):
        assert torch.equal(w, g)
    pipe.close()
    # the throughput form of the recurrence in every slot (what bench.py runs): same bits as that form one call at a time
    rt = m.runtime(dev)
    rt.set_recurrent_tile(16)
    want16 = [m.forward_waveform(b)[0].clone() for b in batches]
    rt.set_recurrent_tile(0)
    pipe = uvad_amd.ForwardPipeline(m, dev, depth=4, recurrent_tile=16)
    got16 = [p.result()[0] for p in [pipe.submit(b) for b in batches]]
    assert all(r.recurrent_tile() == 16 for r in pipe.runtimes)
    for w, g in zip(want16, got16):
        assert torch.equal(w, g)
    pipe.close()
    with pytest.raises(RuntimeError, match="attach_fbank"):
        m2 = uvad_amd.PyanNet2(encoding_dim=64)
        m2.build()
        uvad_amd.ForwardPipeline(m2, dev)


def test_random_shape_sweep_vs_oracle():
    """Seeded sweep over the constructor space and ragged shapes (batch not a multiple of the 4-sequence tile, frame counts
    around the 4 / 16 / 32 / 128 tile edges, widths that are only multiples of 4, 0..3 feed-forward layers, both hidden
    sizes, both directions) against the C oracle, in all three GEMM modes for the first cases."""
    import uvad_amd
    from uvad_amd.synth import seed_weights
    from oracle import c_oracle as co
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(2024)
    worst = 0.0
    for case in range(24):
        H = int(rng.choice([64, 128]))
        L = int(rng.integers(1, 5))
        bi = bool(rng.integers(0, 2))
        F = int(rng.choice([4, 12, 40, 60, 64, 80, 100, 132]))
        lin_l = int(rng.integers(0, 4))
        lin_h = int(rng.choice([4, 32, 100, 128]))
        B = int(rng.choice([1, 2, 3, 5, 7, 8, 13, 33]))
        T = int(rng.choice([1, 2, 3, 15, 16, 17, 31, 33, 63, 127, 129, 200]))
        m = uvad_amd.PyanNet2(lstm={"hidden_size": H, "num_layers": L, "bidirectional": bi},
                              linear={"hidden_size": lin_h, "num_layers": lin_l}, encoding_dim=F)
        m.build()
        seed_weights(m, 100 + case, 2.0)
        m = m.to(dev).eval()
        g = torch.Generator().manual_seed(case)
        feats = torch.randn(B, T, F, generator=g) * 2.0 - 3.0
        sd = {k: v.cpu().numpy() for k, v in m.state_dict().items()}
        want, _ = co.classify(sd, co.ModelCfg(F, H, L, int(bi), lin_h, lin_l, 0.01), feats.numpy())
        for mode in (("f16p", "f32") if case < 8 else ("f16p",)):
            m.runtime(dev).set_gemm_mode(mode)
            logits, _ = m.forward_logits(feats.to(dev))
            err = float(np.abs(logits.cpu().numpy() - want).max())
            worst = max(worst, err)
            assert err < LOGIT_TOL, (case, mode, dict(H=H, L=L, bi=bi, F=F, lin_l=lin_l, lin_h=lin_h, B=B, T=T), err)
    print(f"shape sweep: worst logit err {worst:.2e}")


def test_fbank_random_config_sweep_vs_oracle():
    """Seeded sweep over the FbankConfig space the kernel implements (frame length / hop, filters, window, pre-emphasis,
    DC removal, snip_edges, low / high cut-offs) and ragged lengths, float and int16 PCM, against the float64-DFT C oracle."""
    import uvad_amd
    from oracle import c_oracle as co, torch_ref as tr
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(77)
    worst = 0.0
    for case in range(20):
        flen = float(rng.choice([0.025, 0.02, 0.032, 0.03]))
        fshift = float(rng.choice([0.01, 0.005, 0.016]))
        F = int(rng.choice([8, 23, 40, 64, 80, 128]))
        window = str(rng.choice(["povey", "hamming", "hanning", "rectangular"]))
        pre = float(rng.choice([0.97, 0.0, 0.5]))
        dc = bool(rng.integers(0, 2))
        snip = bool(rng.integers(0, 2))
        low, high = float(rng.choice([20.0, 0.0, 100.0])), float(rng.choice([-400.0, 0.0, 6000.0]))
        B = int(rng.integers(1, 5))
        L, sh = int(round(flen * 16000)), int(round(fshift * 16000))
        S = int(rng.integers(max(L, 600), 40000))
        cfg = uvad_amd.FbankConfig(frame_length=flen, frame_shift=fshift, num_filters=F, window_type=window, preemph_coeff=pre,
                                   remove_dc_offset=dc, snip_edges=snip, low_freq=low, high_freq=high)
        oc = co.default_fbank_cfg(F, frame_len=L, frame_shift=sh, preemph=pre, remove_dc=int(dc), snip_edges=int(snip), low_hz=low, high_hz=high)
        pcm = tr.synth_pcm(B, S, seed=900 + case)
        win, melm = co.window(window, L), co.mel_banks(oc)
        want = co.fbank(pcm, oc, win, melm)
        rt = uvad_amd.Fbank(cfg)._runtime(dev)
        got = rt.fbank(torch.from_numpy(pcm).to(dev)).cpu().numpy()
        assert got.shape == want.shape, (case, got.shape, want.shape)
        err = float(np.abs(got - want).max()) if want.size else 0.0
        worst = max(worst, err)
        # Off the reference geometry (rectangular windows, no pre-emphasis, 128 narrow bands ...) a weak mel bin can sit 1e-3 of the
        # frame's energy below f32 resolution for ANY fp32 transform, so the bound follows the classifier's standard: against the
        # float64-throughout evaluation the kernel may be no further than FEAT_TOL, or than REL x the fp32 torch-CPU rfft path on
        # the same configuration where that path is itself beyond FEAT_TOL; never beyond the 2e-3 this sweep used to allow.
        params = dict(flen=flen, fshift=fshift, F=F, window=window, pre=pre, dc=dc, snip=snip, low=low, high=high, B=B, S=S)
        if want.size:
            truth = co.fbank_f64(pcm, oc, win, melm, threads=4)
            cpu = tr.torch_fbank(pcm, torch.from_numpy(win), torch.from_numpy(melm), frame_shift=sh, n_fft=512, preemph=pre, remove_dc=dc,
                                 snip_edges=snip).numpy()
            e_gpu, e_cpu = float(np.abs(got - truth).max()), float(np.abs(cpu - truth).max())
            r_gpu, r_cpu = float(np.sqrt(((got - truth) ** 2).mean())), float(np.sqrt(((cpu - truth) ** 2).mean()))
            # rms (stable on a sample of a few 10^4 values) within 1.5 x the CPU path's; the single worst value within 2 x: on these
            # small samples it is one weak bin, and two equally accurate fp32 transforms differ by that much there (the torch rfft
            # paths of two hosts gave 3.6e-4 and 4.9e-4 on case 1, the kernel 6.0e-4; at cfg-2 size -- 16 M values -- the ratio of
            # the worst values is 1.06, tests/test_gpu_scale.py holds it to 1.5)
            assert r_gpu <= max(1.5 * r_cpu, 1e-6), (case, params, r_gpu, r_cpu)
            assert e_gpu < max(FEAT_TOL, 2.0 * e_cpu) and e_gpu < 2e-3, (case, params, e_gpu, e_cpu)
            assert err < max(FEAT_TOL, 2.0 * e_cpu + 2e-4), (case, params, err, e_cpu)
        if case % 4 == 0:   # int16 ingest of the same signal
            q = np.round(pcm * 32767.0).astype(np.int16)
            want16 = co.fbank(q.astype(np.float32) / 32768.0, oc, co.window(window, L), co.mel_banks(oc))
            got16 = rt.fbank(torch.from_numpy(q).to(dev)).cpu().numpy()
            assert np.abs(got16 - want16).max() < FEAT_TOL, case
    print(f"fbank sweep: worst log-mel err {worst:.2e}")


def test_streaming_random_chunk_sweep_equals_offline():
    """Lock-step streams with awkward chunk sizes (smaller than a hop, not a multiple of the hop, larger than a second),
    stream counts off the 4-sequence tile, 1- and 2-layer causal models at both hidden sizes: every emitted frame equals
    the offline logits of the whole signal and the number of emitted frames is exactly the frames whose last sample arrived."""
    import uvad_amd
    from uvad_amd.synth import synth_pcm, seed_weights
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(31)
    for case in range(8):
        chunk = int(rng.choice([80, 200, 333, 480, 1024, 4800, 20000]))
        B = int(rng.choice([1, 3, 5, 8]))
        H = int(rng.choice([64, 128]))
        L = int(rng.integers(1, 3))
        F = int(rng.choice([40, 64, 80]))
        steps = max(3, int(40000 // chunk))
        S = steps * chunk
        pcm = synth_pcm(B, S, seed=400 + case)
        m = uvad_amd.PyanNet2(lstm={"bidirectional": False, "hidden_size": H, "num_layers": L}, encoding_dim=F)
        m.build()
        seed_weights(m, 50 + case, 3.0)
        m.attach_fbank(uvad_amd.FbankConfig(num_filters=F))
        m = m.to(dev).eval()
        rt = m.runtime(dev)
        x = torch.from_numpy(pcm).to(dev)
        offline, _ = rt.forward(x)
        st = rt.stream_open(B, chunk)
        outs = [rt.stream_step(st, x[:, i * chunk:(i + 1) * chunk].contiguous()).clone() for i in range(steps)]
        got = torch.cat(outs, dim=1)
        n = got.shape[1]
        assert n == max(0, (S + 120 - 400) // 160 + 1), (case, chunk, n)
        err = float((got - offline[:, :n]).abs().max())
        print(f"stream sweep case {case}: chunk {chunk} B {B} H {H} L {L} F {F}: {n} frames, max diff {err:.2e}")
        assert err < LOGIT_TOL, (case, chunk, B, H, L, F, err)


def test_fbank_full_size_properties_shift_and_gain():
    """Size-independent properties of the feature stage at the full BASELINE cfg-2 size (256 x 10 s, 64 filters):
    (i) dropping the first hop of samples shifts the interior frames by exactly one frame;
    (ii) doubling the signal adds log(4) to every log-mel value (power spectrum x 4), away from the log floor."""
    import uvad_amd
    from uvad_amd.synth import synth_pcm_device
    dev = torch.device("cuda:0")
    rt = uvad_amd.Fbank(uvad_amd.FbankConfig(num_filters=64, window_type="hamming"))._runtime(dev)
    pcm = synth_pcm_device(256, 160000, 7, dev) * 0.5
    f0 = rt.fbank(pcm)
    f1 = rt.fbank(pcm[:, 160:].contiguous())
    assert f0.shape == (256, 1000, 64) and f1.shape == (256, 999, 64)
    shift_err = float((f1[:, 2:-2] - f0[:, 3:-2]).abs().max())      # interior frames only (edges are reflect-padded)
    f2 = rt.fbank(pcm * 2.0)
    gain_err = float((f2 - f0 - float(np.log(4.0))).abs().max())
    print(f"fbank full size: shift err {shift_err:.2e}, gain err {gain_err:.2e}")
    # a frame shares its complex FFT with a different neighbour after the shift: fp32 rounding differs, and the weakest of
    # the 16 M log-mel values amplifies that most (same bound as the oracle comparison); the gain property is exact
    # up to the rounding of log().
    assert shift_err < FEAT_TOL and gain_err < 1e-5
    assert torch.isfinite(f0).all()


def test_long_utterance_ten_minutes():
    """One 10-minute recording in a single call (T = 60 000 frames, 240 000 sequential recurrent steps): frame count,
    finiteness and logits against the torch-CPU reference path on identical features (weights x2: a contractive
    network, so the comparison measures the kernels and not chaotic error growth)."""
    import uvad_amd
    from uvad_amd.synth import seed_weights, synth_pcm_device
    from oracle import torch_ref as tr
    dev = torch.device("cuda:0")
    m = uvad_amd.PyanNet2(encoding_dim=64)
    m.build()
    seed_weights(m, 1234, 2.0)
    m.attach_fbank(uvad_amd.FbankConfig(num_filters=64, window_type="hamming"))
    m = m.to(dev).eval()
    rt = m.runtime(dev)
    pcm = synth_pcm_device(1, 16000 * 600, 3, dev)
    feats = rt.fbank(pcm)
    logits, probs = rt.classify(feats)
    torch.cuda.synchronize()
    assert logits.shape == (1, 60000) and torch.isfinite(logits).all()
    cpu = tr.TorchPyanNet2(64)
    cpu.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    torch.set_num_threads(min(16, torch.get_num_threads()))
    want = cpu(feats.cpu())[0]
    err = float((logits.cpu() - want).abs().max())
    print(f"10-minute utterance: max |GPU - CPU| logit err {err:.2e} (logit range {want.min():.3f}..{want.max():.3f})")
    assert err < LOGIT_TOL
    fused, _ = rt.forward(pcm)
    assert torch.equal(fused, logits)       # uvad_forward == uvad_fbank + uvad_classify
