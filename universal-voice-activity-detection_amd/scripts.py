"""``predict_vad(**config)``: the reference's predict entry point (src/scripts/predict.py:22-110)
with its lhotse / Lightning plumbing replaced by a manifest-free source.  Flow kept from the
reference: seed -> model (checkpoint or seeded weights) -> batches bounded by ``max_duration``
seconds -> ``VadModel.predict_step`` (probabilities -> 0/1 labels) -> per-recording speech
intervals (predict.py:472-490).  Feature extraction, which the reference runs offline in
``task="prepare"`` (ami/utils.py:153-163), is fused in front of the classifier here."""
import json
import os
import wave
from typing import List

import numpy as np
import torch

from .engine import VadModel
from .features import Fbank, FbankConfig
from .postprocess import labels_to_intervals_batch, median_filter
from .synth import seed_weights, synth_pcm


def read_wav_int16(path: str, sample_rate: int = 16000) -> np.ndarray:
    with wave.open(path, "rb") as w:
        if w.getframerate() != sample_rate or w.getsampwidth() != 2:
            raise ValueError(f"{path}: need {sample_rate} Hz 16-bit PCM")
        pcm = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")
        if w.getnchannels() > 1:
            pcm = pcm.reshape(-1, w.getnchannels())[:, 0]
    return np.ascontiguousarray(pcm)


def _resolve_device(name: str) -> torch.device:
    if name in ("gpu", "cuda"):
        if not torch.cuda.is_available():
            raise RuntimeError("config.device='gpu' but no HIP device is visible; this package has no CPU path")
        return torch.device("cuda", torch.cuda.current_device())
    raise RuntimeError(f"config.device={name!r}: the VAD path runs in HIP kernels only (use 'gpu'); "
                       "the CPU restatement lives in oracle/ and is test infrastructure")


def predict_vad(**kwargs):
    assert kwargs["model_name"] in kwargs["supported_models"], \
        f"Invalid model {kwargs['model_name']}. Model should be one of {kwargs['supported_models']}"
    if kwargs["feature_extractor"] not in ("fbank", "sincnet"):
        raise NotImplementedError("feature_extractor must be 'fbank' (log-mel + PyanNet2) or 'sincnet' (waveform PyanNet); "
                                  "the wav2vec2 / hubert encoders are outside the accelerated path")
    sincnet = kwargs["feature_extractor"] == "sincnet"   # the reference's custom_vad path: the model consumes raw audio
    torch.manual_seed(kwargs["seed"])
    np.random.seed(kwargs["seed"])
    device = _resolve_device(kwargs["device"])
    frame_shift = kwargs["frame_shift"]
    model_dict = dict(kwargs["model_dict"])

    if kwargs["load_checkpoint"]:
        model = VadModel.load_from_checkpoint(checkpoint_path=kwargs["checkpoint_path"],
                                              model_name=kwargs["model_name"], model_dict=model_dict)
    else:
        model = VadModel(model_name=kwargs["model_name"], model_dict=model_dict)
        seed_weights(model.model, kwargs.get("weights_seed", 1234), kwargs.get("weights_scale", 4.0))
    model = model.to(device).eval()

    extractor = None
    if not sincnet:
        fb_cfg = FbankConfig(sampling_rate=16000, num_filters=model.model.encoding_dim,
                             window_type=kwargs.get("window_type", "povey"), frame_shift=frame_shift, device="cuda")
        extractor = Fbank(fb_cfg)

    src = kwargs["input"]
    recs: List[dict] = []
    if src["kind"] == "wav":
        for p in src["paths"]:
            recs.append({"id": os.path.basename(p), "pcm": read_wav_int16(p).astype(np.float32) / 32768.0})
    elif src["kind"] == "synthetic":
        S = int(round(src["seconds"] * 16000))
        pcm = synth_pcm(src["num_utterances"], S, seed=src["seed"])
        recs = [{"id": f"synthetic-{src['seed'] + i}", "pcm": pcm[i]} for i in range(pcm.shape[0])]
    else:
        raise ValueError(f"unknown input kind {src['kind']!r}")

    # batches: equal-length recordings together, at most max_duration seconds per batch
    results = []
    order = sorted(range(len(recs)), key=lambda i: len(recs[i]["pcm"]))
    i = 0
    while i < len(order):
        n = len(recs[order[i]]["pcm"])
        group = [order[i]]
        while (i + len(group) < len(order) and len(recs[order[i + len(group)]]["pcm"]) == n
               and (len(group) + 1) * n / 16000.0 <= kwargs["max_duration"]):
            group.append(order[i + len(group)])
        i += len(group)
        batch_pcm = torch.from_numpy(np.stack([recs[j]["pcm"] for j in group])).to(device)
        # one forward pass per batch; labels exactly as VadModel.predict_step derives them from the probabilities
        # (vad_engine.py:204-211: threshold 0.5 + median filter, 49 taps unless encoding_dim == 768)
        if sincnet:   # (batch, samples); the model consumes raw audio, channel axis added as in vad_engine.py:252-255
            probs = model(batch_pcm.unsqueeze(1)).squeeze(-1)
        else:
            feats = torch.stack(extractor.extract_batch(list(batch_pcm), sampling_rate=16000))
            probs = model(feats).squeeze(-1)
        labels = median_filter(probs, window=0.02 if model.model.encoding_dim == 768 else 0.01)   # (B, T) 0/1
        intervals = labels_to_intervals_batch(labels, frame_shift)   # run-length walk on the GPU (uvad_label_runs)
        labels_h, probs_h = labels.cpu().numpy(), probs.cpu().numpy()
        for r, j in enumerate(group):
            results.append({"recording_id": recs[j]["id"], "num_frames": int(labels_h.shape[1]),
                            "labels": labels_h[r].astype(np.uint8), "probs": probs_h[r],
                            "intervals": intervals[r]})
    results.sort(key=lambda r: r["recording_id"])

    out_dir = kwargs.get("predict_output_dir") or ""
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "predictions.json"), "w") as f:
            json.dump([{"recording_id": r["recording_id"], "num_frames": r["num_frames"],
                        "speech_frames": int(r["labels"].sum()), "intervals": r["intervals"]} for r in results], f, indent=1)
    for r in results:
        print(f"{r['recording_id']}: {r['num_frames']} frames, {int(r['labels'].sum())} speech, {len(r['intervals'])} intervals")
    return results
