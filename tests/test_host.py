"""CPU tests of the host-side mirror of the reference interface (no compute calls)."""
import os

import numpy as np
import pytest
import torch

import uvad_amd
from oracle import c_oracle as co, torch_ref as tr

EXPECTED_KEYS_L0 = ["lstm.weight_ih_l0", "lstm.weight_hh_l0", "lstm.bias_ih_l0", "lstm.bias_hh_l0",
                    "lstm.weight_ih_l0_reverse", "lstm.weight_hh_l0_reverse", "lstm.bias_ih_l0_reverse", "lstm.bias_hh_l0_reverse"]


def test_pyannet2_state_dict_contract():
    m = uvad_amd.PyanNet2(encoding_dim=80)
    m.build()
    sd = m.state_dict()
    keys = list(sd)
    assert keys[:8] == EXPECTED_KEYS_L0
    assert keys[-6:] == ["linear.0.weight", "linear.0.bias", "linear.1.weight", "linear.1.bias", "classifier.weight", "classifier.bias"]
    assert sd["lstm.weight_ih_l0"].shape == (512, 80) and sd["lstm.weight_ih_l1"].shape == (512, 256)
    assert sd["lstm.weight_hh_l3_reverse"].shape == (512, 128)
    assert sd["linear.0.weight"].shape == (128, 256) and sd["classifier.weight"].shape == (1, 128)
    assert sum(p.numel() for p in m.parameters()) == 1450369
    assert m.encoding_dim == 80
    assert m.hparams.lstm["batch_first"] is True and m.hparams.lstm["num_layers"] == 4
    assert m.hparams.linear == {"hidden_size": 128, "num_layers": 2}
    # same keys / shapes as the oracle restatement, hence as the reference class the goldens came from
    ref = tr.TorchPyanNet2(80).state_dict()
    assert list(ref) == keys and all(ref[k].shape == sd[k].shape for k in keys)
    assert uvad_amd.PyanNet2().encoding_dim == 768                   # reference default


def test_pyannet2_variants():
    m = uvad_amd.PyanNet2(lstm={"bidirectional": False, "num_layers": 2}, linear={"num_layers": 0}, encoding_dim=64)
    m.build()
    assert m.classifier.in_features == 128 and not hasattr(m, "linear")
    m2 = uvad_amd.PyanNet2(lstm={"monolithic": False}, encoding_dim=64)
    m2.build()
    assert "lstm.3.weight_hh_l0_reverse" in m2.state_dict()
    with pytest.raises(RuntimeError, match="build"):
        uvad_amd.PyanNet2(encoding_dim=64).runtime(torch.device("cuda:0"))


def test_pyannet_mirror_state_dict_and_frame_count():
    """PyanNet (SincNet front end): the reference's state_dict names / shapes, 293 frames per 5 s cut
    (src/datasets/custom_vad.py:47) and no CPU path."""
    m = uvad_amd.PyanNet()
    m.build()
    sd = m.state_dict()
    for k, shape in {"sincnet.wav_norm1d.weight": (1,), "sincnet.conv1d.0.filterbank.low_hz_": (40, 1),
                     "sincnet.conv1d.0.filterbank.band_hz_": (40, 1), "sincnet.conv1d.1.weight": (60, 80, 5),
                     "sincnet.conv1d.2.bias": (60,), "sincnet.norm1d.0.weight": (80,), "sincnet.norm1d.2.bias": (60,),
                     "lstm.weight_ih_l0": (512, 60), "classifier.weight": (1, 128)}.items():
        assert tuple(sd[k].shape) == shape, k
    assert m.hparams.sincnet == {"stride": 10, "sample_rate": 16000}
    assert m.num_frames(80000) == 293 and tr.sincnet_num_frames(80000) == 293
    f = m.sincnet.conv1d[0].filterbank.filters()
    assert tuple(f.shape) == (80, 1, 251)
    # host filter construction == the oracle's restatement of ParamSincFB.filters()
    assert torch.equal(f[:, 0, :], tr.sinc_filters(m.sincnet.conv1d[0].filterbank.low_hz_.detach(), m.sincnet.conv1d[0].filterbank.band_hz_.detach()))
    # cos filters are even, sin filters odd, DC gain of a band-pass is ~0 relative to its peak
    assert torch.allclose(f[:40, 0], torch.flip(f[:40, 0], dims=[1])) and torch.allclose(f[40:, 0], -torch.flip(f[40:, 0], dims=[1]))
    with pytest.raises(RuntimeError, match="GPU"):
        m(torch.zeros(1, 1, 16000))
    with pytest.raises(RuntimeError, match="PyanNet"):
        m.sincnet.__class__()(torch.zeros(1, 1, 16000))


def test_seed_weights_equals_golden_generator_weights():
    from uvad_amd.synth import seed_weights
    m = uvad_amd.PyanNet2(encoding_dim=64)
    m.build()
    seed_weights(m, 1234, 4.0)
    want = tr.seeded_state_dict(64, seed=1234, scale=4.0)
    for k, v in m.state_dict().items():
        assert torch.equal(v, want[k]), k


def test_vadmodel_surface_and_checkpoint_roundtrip(tmp_path):
    vm = uvad_amd.VadModel(model_name="PyanNet2", model_dict={"encoding_dim": 80})
    assert vm.model_name == "PyanNet2" and isinstance(vm.model, uvad_amd.PyanNet2) and hasattr(vm.model, "classifier")
    for name in ("forward", "predict_step", "_common_step", "load_from_checkpoint"):
        assert hasattr(vm, name)
    # Lightning-style checkpoint: {"state_dict": {"model.<key>": tensor}}
    ck = tmp_path / "checkpoint-epoch=11.ckpt"
    torch.save({"state_dict": {k: v for k, v in vm.state_dict().items()}, "epoch": 11}, ck)
    assert all(k.startswith("model.") for k in vm.state_dict())
    vm2 = uvad_amd.VadModel.load_from_checkpoint(checkpoint_path=str(ck), model_dict={"encoding_dim": 80})
    for k, v in vm.state_dict().items():
        assert torch.equal(v, vm2.state_dict()[k])
    with pytest.raises(RuntimeError):   # reference pitfall (predict.py:77): default encoding_dim=768 does not fit a fbank ckpt
        uvad_amd.VadModel.load_from_checkpoint(checkpoint_path=str(ck))
    with pytest.raises(NotImplementedError):
        vm.training_step({}, 0)


def test_config_and_main_dispatch():
    import main as entry
    from config.config import load_config
    cfg = load_config()
    for key in ("task", "function", "seed", "device", "feature_extractor", "frame_shift", "model_name", "model_dict",
                "supported_models", "checkpoint_path", "load_checkpoint", "max_duration", "predict_output_dir"):
        assert key in cfg
    assert cfg.model_name == "PyanNet2" and cfg.model_dict.encoding_dim == 80 and cfg.frame_shift == 0.01
    assert dict(**cfg)["task"] == "run"                     # **config splatting, main.py:34-44
    cfg.task = "prepare"
    with pytest.raises(NotImplementedError):
        entry.main(cfg)
    cfg = load_config()
    cfg.device = "cpu"
    with pytest.raises(RuntimeError, match="HIP kernels only"):
        entry.main(cfg)


def test_reference_import_paths_resolve():
    from src.models import PyanNet2 as A
    from src.models.segmentation.PyanNet2 import PyanNet2 as B
    from src.engines.vad_engine import VadModel
    from src.utils.helper import median_filter
    from src.scripts import predict_vad
    assert A is B is uvad_amd.PyanNet2 and VadModel is uvad_amd.VadModel
    assert callable(median_filter) and callable(predict_vad)
    from src.models import PyanNet as C
    from src.models.segmentation.PyanNet import PyanNet as D
    from src.models.blocks.sincnet import SincNet
    assert C is D is uvad_amd.PyanNet and SincNet is uvad_amd.SincNet


def test_fbank_config_defaults_are_lhotse_defaults():
    c = uvad_amd.FbankConfig(sampling_rate=16000, device="cuda")     # the reference's call, ami/utils.py:153
    assert (c.frame_len_samples, c.frame_shift_samples, c.n_fft, c.num_filters) == (400, 160, 512, 80)
    assert c.window_type == "povey" and c.preemph_coeff == 0.97 and c.remove_dc_offset and not c.snip_edges
    assert c.low_freq == 20.0 and c.high_freq == -400.0 and c.dither == 0.0
    assert abs(c.energy_floor - np.finfo(np.float32).eps) < 1e-12
    assert uvad_amd.FbankConfig(num_mel_bins=64).num_filters == 64
    assert uvad_amd.Fbank(c).feature_dim(16000) == 80 and uvad_amd.Fbank(c).frame_shift == 0.01


def test_median_window_and_intervals():
    assert uvad_amd.median_window(0.01) == 49 and uvad_amd.median_window(0.02) == 25      # helper.py:85-87
    rng = np.random.default_rng(1)
    for _ in range(20):
        lab = (rng.random(rng.integers(1, 300)) < 0.5).astype(np.uint8)
        assert uvad_amd.labels_to_intervals(lab, 0.01) == co.intervals(lab, 0.01)
    assert uvad_amd.labels_to_intervals(np.ones(10, np.uint8), 0.01) == [(0.0, 0.09)]
    assert uvad_amd.labels_to_intervals(np.zeros(10, np.uint8), 0.01) == []
    assert uvad_amd.labels_to_intervals(np.array([], np.uint8), 0.01) == []


def test_synth_is_shard_invariant():
    from uvad_amd.synth import synth_pcm
    full = synth_pcm(6, 8000, seed=1000)
    assert np.array_equal(full[4:6], synth_pcm(2, 8000, seed=1000, first=4))
    assert np.abs(full).max() <= 1.0 and full.std() > 0.05


def test_interval_scoring_helpers_follow_the_reference():
    # predict.py:614-634: buffer, clip to [0, duration], merge when the next start <= running end
    assert uvad_amd.merge_intervals_with_buffer([(1, 2), (2.5, 3), (10, 11)], 12, 0.3) == [[0.7, 3.3], [9.7, 11.3]]
    assert uvad_amd.merge_intervals_with_buffer([(0.1, 0.2)], 5, 0.5) == [[0, 0.7]]
    assert uvad_amd.merge_intervals_with_buffer([(4, 4.9), (1, 2)], 5, 0.25) == [[0.75, 2.25], [3.75, 5]]
    assert uvad_amd.merge_intervals_with_buffer([], 5, 0.5) == []
    # predict.py:638-647: split long intervals into 10 s windows, drop remainders <= 0.1 s
    assert uvad_amd.split_into_windows([[0, 25.05]]) == [[0, 10], [10, 20], [20, 25.05]]
    assert uvad_amd.split_into_windows([[0, 20.05]]) == [[0, 10], [10, 20]]
    assert uvad_amd.split_into_windows([[3, 3.05], [5, 6]]) == [[5, 6]]
    # predict.py:654-663: ceil(duration/shift) frames, [int(s/shift), int(e/shift)) = 1
    lab = uvad_amd.intervals_to_labels([(0.02, 0.05), (0.08, 0.2)], 0.1, 0.01)
    assert lab.tolist() == [0, 0, 1, 1, 1, 0, 0, 0, 1, 1]
    assert len(uvad_amd.intervals_to_labels([], 1.005, 0.01)) == 101


def test_bench_line_contract_on_the_committed_sample():
    """The bench contract (driver side): one JSON object with the required keys, the metric / unit of BASELINE.json, a roofline
    and a cpu_baseline object.  Checked on the line committed under profiles/ (bench.py itself needs a GPU)."""
    import glob
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = sorted(glob.glob(os.path.join(root, "profiles", "r0*bench.json")))[-1]
    o = json.load(open(path))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in o, k
    base = json.load(open(os.path.join(root, "BASELINE.json")))
    assert o["unit"] == "frames/s" and "frames/sec" in o["metric"] and "frames/sec" in base["metric"]
    assert o["higher_is_better"] is True and o["scaling"] == "weak" and o["vs_baseline"] is None and o["data"] == "synthetic"
    assert "workload" in o["config"] and not any(k in o["config"] for k in ("model", "seq_len", "global_batch"))
    r = o["roofline"]
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["unit"] in ("GB/s", "TFLOP/s")
    c = o["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "frames/s"
    assert abs(o["value"] - o["config"]["utterances_per_gpu"] * o["config"]["frames_per_utterance"] * o["n_gpus"] / (o["ms_per_step"] * 1e-3)) / o["value"] < 1e-6


def test_predict_vad_pipeline_falls_back_when_streams_do_not_overlap(monkeypatch):
    """ForwardPipeline raises when it cannot prove `depth` concurrent HIP streams (GPU_MAX_HW_QUEUES = 1, a shared
    device); predict_vad must then run with fewer batches in flight -- down to batch after batch -- not abort."""
    from uvad_amd import scripts
    tried = []

    class NoStreams:
        def __init__(self, model, device, depth=2, recurrent_tile=0):
            tried.append(depth)
            raise RuntimeError(f"only 1 concurrent HIP streams found in 48 tries; lower depth (GPU_MAX_HW_QUEUES?)")

    monkeypatch.setattr(scripts, "ForwardPipeline", NoStreams)
    assert scripts.open_pipeline(object(), "cuda:0", 3) is None and tried == [3]
    tried.clear()
    assert scripts.open_pipeline(object(), "cuda:0", 8) is None and tried == [8, 4, 2]

    class TwoStreams:
        def __init__(self, model, device, depth=2, recurrent_tile=0):
            if depth > 2:
                raise RuntimeError("only 2 concurrent HIP streams found in 48 tries; lower depth (GPU_MAX_HW_QUEUES?)")
            self.depth = depth

    monkeypatch.setattr(scripts, "ForwardPipeline", TwoStreams)
    assert scripts.open_pipeline(object(), "cuda:0", 4).depth == 2

    class Broken:
        def __init__(self, *a, **k):
            raise RuntimeError("out of memory")

    monkeypatch.setattr(scripts, "ForwardPipeline", Broken)
    with pytest.raises(RuntimeError, match="out of memory"):   # anything else is not swallowed
        scripts.open_pipeline(object(), "cuda:0", 3)


def test_sincnet_frame_time_geometry_matches_the_reference_fixture():
    """VERDICT r2 #8.  tests/golden/sincnet_predict_geometry.json holds outputs of the REFERENCE's own code (tools/gen_golden_sincnet_predict.py:
    receptive_field.py imported from its file; get_timestamp_from_sample_boundary, the run-length walk of get_new_cuts, the
    merge / split helpers compiled out of predict_sincnet.py's syntax tree).  Frame k is centred on sample 270 k + 496 (round(0.5 * 991) in Python) and the
    reference rounds to whole seconds -- it does NOT multiply by frame_shift."""
    import json
    from uvad_amd import postprocess as pp
    from uvad_amd.sincnet import SincNet
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "sincnet_predict_geometry.json")))
    assert [pp.SINC_RF_1, pp.SINC_RF_2] == g["receptive_field"] and pp.SINC_STEP == 270 and pp.SINC_HALF == 496   # the reference's comment says 495, its round(495.5) is 496
    for n, frames in g["num_frames"].items():
        assert SincNet.num_frames(int(n)) == frames, n
    for a, b, d, s, e in g["timestamps"]:
        assert pp.sincnet_frame_times(a, b, d) == (s, e), (a, b, d)
    for w in g["walks"]:
        lab = np.frombuffer(w["labels"].encode(), np.uint8) - ord("0")
        got = pp.sincnet_labels_to_intervals(lab, w["duration"])
        assert [list(x) for x in got] == w["intervals"], (w["duration"], got[:4], w["intervals"][:4])
    for m in g["merge"]:
        assert pp.merge_intervals_with_buffer(m["intervals"], m["duration"], m["buffer"]) == m["merged"]
    for m in g["split"]:
        assert pp.split_into_windows(m["intervals"], m["window"]) == m["split"]


def test_residue_plane_of_the_three_way_f16_split_is_always_exactly_bf8():
    """The claim behind lstm_rec16h_kernel's fp8 product (csrc/lstm.hip pack_whh16h_p2q): with w scaled so that the
    largest magnitude lies in [2^13, 2^14), p0 = f16(ws), t2 = (ws - p0) * 2^11, p1 = f16(t2), the residue p2 = f16(t2 - p1) -- the
    third plane, which makes p0 + (p1 + p2) / 2^11 reproduce ws -- times 2^13 is ALWAYS exactly a bf8 (E5M2) number: E5M2 is the upper
    byte of an f16, so the low byte of f16(p2 * 2^13) must be zero and the product must not round.  Restated in numpy (float16 casts
    round to nearest even, as _Float16 casts do) over 4 million weights: uniform, normal, log-uniform over 30 binades, denormal-prone
    tiny ones and exact powers of two."""
    rng = np.random.default_rng(20)
    parts = [rng.uniform(-1, 1, 1 << 20), rng.normal(0, 0.2, 1 << 20),
             np.exp2(rng.uniform(-30, 0, 1 << 20)) * rng.choice([-1.0, 1.0], 1 << 20),
             np.exp2(rng.integers(-40, 1, 1 << 19).astype(np.float64)), rng.uniform(-1e-6, 1e-6, 1 << 19)]
    w = np.concatenate(parts).astype(np.float32)
    w[0] = 0.999   # the largest magnitude decides the scale
    e = int(np.frexp(np.abs(w).max())[1])
    ws = (w * np.float32(2.0 ** (14 - e))).astype(np.float32)
    assert 2.0 ** 13 <= np.abs(ws).max() < 2.0 ** 14
    p0 = ws.astype(np.float16)
    t2 = ((ws - p0.astype(np.float32)) * np.float32(2048.0)).astype(np.float32)
    p1 = t2.astype(np.float16)
    p2 = (t2 - p1.astype(np.float32)).astype(np.float16)
    assert np.abs(p2.astype(np.float32)).max() <= 4.0
    sh32 = p2.astype(np.float32) * np.float32(8192.0)
    sh = sh32.astype(np.float16)
    assert np.array_equal(sh.astype(np.float32), sh32)                       # the shift itself never rounds or overflows
    bits = sh.view(np.uint16)
    assert not (bits & 0xFF).any(), "a residue with more than two significant bits: not a bf8 number"
    # and the three planes do reproduce the scaled weight wherever f16 can hold the residue at all (|ws| >= 2^-1: p2's unit >= 2^-24 x 2^11)
    big = np.abs(ws) >= 0.5
    rec = p0.astype(np.float64) + (p1.astype(np.float64) + p2.astype(np.float64)) / 2048.0
    assert np.array_equal(rec[big], ws.astype(np.float64)[big])
