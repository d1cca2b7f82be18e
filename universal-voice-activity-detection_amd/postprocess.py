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
