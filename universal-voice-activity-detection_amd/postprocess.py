"""Post-processing next to the hot path: the reference's ``median_filter``
(src/utils/helper.py:66-97) and the run-length interval extraction of ``get_new_cuts``
(src/scripts/predict.py:472-490).  The filter runs on the GPU (``uvad_median_filter``) instead of
the reference's device -> CPU -> scipy -> "cuda" round trip; the interval walk is vectorised."""
from typing import List, Tuple

import numpy as np
import torch


def median_window(window: float, speech_window: float = 0.5) -> int:
    """Tap count the reference derives: int(SPEECH_WINDOW / window), made odd (49 @ 10 ms, 25 @ 20 ms)."""
    k = int(speech_window / window)
    return k - 1 if k % 2 == 0 else k


def median_filter(x: torch.Tensor, SPEECH_WINDOW: float = 0.5, window: float = 0.02, runtime=None) -> torch.Tensor:
    """(batch, frames) probabilities -> (batch, frames) int64 0/1, same contract as the reference
    (which returns the tensor on "cuda"); thresholds at 0.5 then applies the odd binary median
    with zero-padded edges."""
    if not x.is_cuda:
        raise RuntimeError("median_filter runs on the GPU only")
    if runtime is None:
        from .runtime import VadRuntime
        runtime = _shared_runtime(x.device)
    return runtime.median_filter(x, median_window(window, SPEECH_WINDOW)).to(torch.int64)


_RT = {}


def _shared_runtime(device):
    from .runtime import VadRuntime
    key = str(device)
    if key not in _RT:
        _RT[key] = VadRuntime(device=device, fbank=None, model=None)
    return _RT[key]


def labels_to_intervals(labels, frame_shift: float) -> List[Tuple[float, float]]:
    """One row of 0/1 frame labels -> [(start_s, end_s)], predict.py:472-490: a run that starts at
    frame k and whose first non-speech frame is k2 gives (round(k*shift, 2), round((k2-1)*shift, 2)),
    kept only if end - start > 0; a run still open at the end closes at (len-1)*shift."""
    v = np.asarray(labels.cpu() if torch.is_tensor(labels) else labels).astype(np.int8).ravel()
    if v.size == 0:
        return []
    d = np.diff(np.concatenate(([0], v, [0])))
    starts = np.flatnonzero(d == 1)
    stops = np.flatnonzero(d == -1)       # first non-speech frame after each run (== len for an open run)
    out = []
    for k, k2 in zip(starts, stops):
        last = (len(v) - 1) if k2 >= len(v) else (k2 - 1)
        s, e = round(float(k * frame_shift), 2), round(float(last * frame_shift), 2)
        if e - s > 0.0:
            out.append((s, e))
    return out


def labels_to_intervals_batch(labels: torch.Tensor, frame_shift: float, runtime=None) -> List[List[Tuple[float, float]]]:
    """(B, T) 0/1 labels ON THE GPU -> per row [(start_s, end_s)], same values as ``labels_to_intervals`` row by row.
    The run-length walk runs in ``uvad_label_runs``; only the (start, stop) frame pairs cross to the host (one copy
    for the whole batch), where the reference's rounding / empty-interval rule is applied."""
    if not (torch.is_tensor(labels) and labels.is_cuda):
        raise RuntimeError("labels_to_intervals_batch runs on the GPU only (use labels_to_intervals for host rows)")
    if labels.dim() == 3:
        labels = labels.squeeze(-1)
    rt = runtime or _shared_runtime(labels.device)
    T = labels.shape[1]
    runs, counts = rt.label_runs(labels)
    counts_h = counts.cpu().numpy()
    runs_h = runs[:, : max(int(counts_h.max()), 1)].cpu().numpy()
    out = []
    for b in range(labels.shape[0]):
        row = []
        for k, k2 in runs_h[b, : counts_h[b]]:
            last = (T - 1) if k2 >= T else (k2 - 1)
            s, e = round(float(k * frame_shift), 2), round(float(last * frame_shift), 2)
            if e - s > 0.0:
                row.append((s, e))
        out.append(row)
    return out


# ---- SincNet (PyanNet) frames -> seconds, src/scripts/predict_sincnet.py:492-504 ----------------------------------
SINC_RF_1, SINC_RF_2 = 991, 1261           # receptive field of 1 and of 2 output frames (src/utils/receptive_field.py:197-215)
SINC_STEP = SINC_RF_2 - SINC_RF_1          # 270 samples between frame centres
SINC_HALF = round(0.5 * SINC_RF_1)         # the reference's comment says 495; its code, round(495.5), gives 496 (round-half-even) -- the code wins


def sincnet_frame_times(start: int, end: int, duration: float, sample_rate: int = 16000):
    """get_timestamp_from_sample_boundary (predict_sincnet.py:492-504): frame k of the SincNet front end is centred on sample
    k * 270 + 496 (SINC_HALF); the reference rounds that to WHOLE seconds (Python round), clamps the start at 0 and the end at the
    recording's duration."""
    s = round((start * SINC_STEP + SINC_HALF) / sample_rate)
    e = round((end * SINC_STEP + SINC_HALF) / sample_rate)
    return max(s, 0), min(e, duration)


def sincnet_labels_to_intervals(labels, duration: float, runtime=None) -> List[Tuple[float, float]]:
    """One recording's 0/1 frame labels (SincNet frame rate) -> [(start_s, end_s)] as get_new_cuts of predict_sincnet.py
    walks them (:348-370): a run of frames [k, k2) maps through sincnet_frame_times(k, k2 - 1, duration), kept only if
    end - start > 0.  Labels on the GPU go through uvad_label_runs; host rows are walked with numpy."""
    if torch.is_tensor(labels) and labels.is_cuda:
        rt = runtime or _shared_runtime(labels.device)
        runs, counts = rt.label_runs(labels.reshape(1, -1))
        n = int(counts.cpu()[0])
        pairs = runs[0, :n].cpu().numpy()
    else:
        v = np.asarray(labels.cpu() if torch.is_tensor(labels) else labels).astype(np.int8).ravel()
        d = np.diff(np.concatenate(([0], v, [0])))
        pairs = np.stack([np.flatnonzero(d == 1), np.flatnonzero(d == -1)], axis=1) if v.size else np.zeros((0, 2), np.int64)
    out = []
    for k, k2 in pairs:
        s, e = sincnet_frame_times(int(k), int(k2) - 1, duration)
        if e - s > 0.0:
            out.append((s, e))
    return out


# ---- scoring side of get_new_cuts (src/scripts/predict.py:500-509, 612-673) -----------------------------------

def merge_intervals_with_buffer(intervals, total_duration: float, buffer: float):
    """predict.py:614-634: widen every interval by `buffer` (clipped to the recording) and merge overlaps."""
    if len(intervals) == 0:
        return []
    iv = sorted(([max(s - buffer, 0), min(e + buffer, total_duration)] for s, e in intervals), key=lambda x: x[0])
    out = [list(iv[0])]
    for s, e in iv[1:]:
        if s <= out[-1][1]:
            out[-1][1] = e          # as the reference: the later interval's end replaces (not max) the running end
        else:
            out.append([s, e])
    return out


def split_into_windows(intervals, window: float = 10):
    """predict.py:638-647: cut intervals longer than `window` seconds; drop remainders of 0.1 s or less."""
    out = []
    for s, e in intervals:
        while e - s > window:
            out.append([s, s + window])
            s += window
        if e - s > 0.1:
            out.append([s, e])
    return out


def intervals_to_labels(intervals, total_duration: float, frame_shift: float) -> np.ndarray:
    """predict.py:654-663 (get_binary_tensor): ceil(duration/shift) frames, [int(s/shift), int(e/shift)) set to 1."""
    import math
    lab = np.zeros(math.ceil(total_duration / frame_shift), np.uint8)
    for s, e in intervals:
        lab[int(s / frame_shift):int(e / frame_shift)] = 1
    return lab


def detection_error(pred_labels: torch.Tensor, gt_labels: torch.Tensor, runtime=None):
    """(B, T) 0/1 predictions vs ground truth on the GPU -> dict of per-row FA, MD, DER fractions
    (predict.py:666-673 / vad_engine.py:102-105).  The counting runs in uvad_der_counts."""
    if not pred_labels.is_cuda:
        raise RuntimeError("detection_error runs on the GPU only")
    rt = runtime or _shared_runtime(pred_labels.device)
    counts = rt.der_counts(pred_labels, gt_labels).to(torch.float64)
    n = float(pred_labels.shape[1])
    fa, md = counts[:, 0] / n, counts[:, 1] / n
    return {"false_alarm": fa, "missed_detection": md, "detection_error_rate": fa + md}
