"""ForwardPipeline: keep several uvad_forward calls in flight.

Why: the recurrence of one batch is a serial chain of 4 x T steps that cannot use the chip -- at B = 256 its latency form
occupies 128 of the 256 CUs for 5.3 ms, its throughput form (recurrent_tile=16) 32 CUs for 7.4 ms.  Independent batches
submitted on other HIP streams run their feature kernels, projections and recurrences on the CUs one batch leaves idle
(bench.py at BASELINE cfg 2: 32 M frames/s one step at a time, 68 M with twelve in flight and the throughput form).  Each slot is
a full VadRuntime (own weights copy and workspace) bound to its own stream; results are identical to the sequential path with
the same recurrent form (same kernels, same launch shapes).

Stream choice: HIP maps streams onto a small pool of hardware queues (four by default; the HIP runtime reads
GPU_MAX_HW_QUEUES when it starts, so a caller that wants depth > 4 sets it before the first HIP call) and streams that
land on one queue serialise.  ``select_streams`` therefore creates streams until ``depth`` of them are PROVEN pairwise
concurrent by ``uvad_streams_overlap`` (a spinning wave on one, an empty kernel on the other): an observation made once at
construction, not a timing heuristic."""
from typing import List, Optional

import torch

from .runtime import VadRuntime


class Pending:
    """Result of ForwardPipeline.submit: tensors that are complete once ``result()`` (or ``wait()``) returns."""

    def __init__(self, event, logits, probs, start=None):
        self._event, self._logits, self._probs, self._start = event, logits, probs, start

    def elapsed_ms(self) -> float:
        """Device time from the step's first kernel being allowed to start to its last one finishing (submit(timed=True) only):
        the latency of ONE batch while the other slots' steps share the GPU."""
        if self._start is None:
            raise RuntimeError("submit(..., timed=True) to measure a step")
        self._event.synchronize()
        return self._start.elapsed_time(self._event)

    def wait(self):
        self._event.synchronize()

    def result(self):
        self._event.synchronize()
        return self._logits, self._probs


class ForwardPipeline:
    MAX_STREAM_TRIES = 48

    def __init__(self, model, device, depth: int = 2, recurrent_tile: int = 0):
        """model: a built uvad_amd.PyanNet2 with attach_fbank(...) done (weights are copied into every slot).
        recurrent_tile: 0 = the library's per-call choice (fastest single call), 16 = the throughput form of the recurrence
        (16 sequences per workgroup: ~3.5x fewer CU-cycles per sequence than the latency form, so more of the chip is free
        for the other slots' kernels; a single call gets slower), 4 = the latency form."""
        if getattr(model, "_fbank_cfg", None) is None:
            raise RuntimeError("attach_fbank(FbankConfig(...)) first: the pipeline runs the fused PCM -> logits path")
        self.device = torch.device(device)
        self.depth = max(1, int(depth))
        cfg = {"encoding_dim": model.encoding_dim, "lstm": model.hparams.lstm, "linear": model.hparams.linear}
        self.runtimes: List[VadRuntime] = []
        self.streams: Optional[List[torch.cuda.Stream]] = None
        self.streams_tried = 0
        self._k = 0
        self._active = self.depth
        try:   # every slot holds a weights copy and (later) a workspace: a failure while building slot k must not leave 0..k-1 behind
            for _ in range(self.depth):
                r = VadRuntime(device=self.device, fbank=model._fbank_cfg, model=cfg)
                self.runtimes.append(r)
                r.load_state_dict(model.state_dict())
                if recurrent_tile:
                    r.set_recurrent_tile(recurrent_tile)
                if self.depth > 1:
                    # time-chunked layers were designed and measured for ONE batch alone on the GPU (its side stream takes the CUs the
                    # recurrence leaves idle); with several steps in flight those CUs belong to the other slots, and a slot's first
                    # submit would run the side-stream probe beside busy neighbours.  One launch per layer here.
                    r.set_time_chunks(1)
            self.select_streams()
        except Exception:
            self.close()
            raise

    # ------------------------------------------------------------------ streams
    def select_streams(self) -> List[torch.cuda.Stream]:
        """Pick ``depth`` HIP streams that are pairwise concurrent (each new candidate is probed against the ones already
        kept).  Raises if the device offers fewer concurrent queues than ``depth``."""
        rt = self.runtimes[0]
        kept: List[torch.cuda.Stream] = []
        tried = 0
        with torch.cuda.device(self.device):
            while len(kept) < self.depth and tried < self.MAX_STREAM_TRIES:
                s = torch.cuda.Stream(device=self.device)
                tried += 1
                if all(rt.streams_overlap(k, s) for k in kept):
                    kept.append(s)
        self.streams_tried = tried
        if len(kept) < self.depth:
            raise RuntimeError(f"only {len(kept)} concurrent HIP streams found in {tried} tries; lower depth (GPU_MAX_HW_QUEUES?)")
        self.streams = kept
        return kept

    # ------------------------------------------------------------------ use
    def submit(self, pcm: torch.Tensor, want_logits: bool = True, want_probs: bool = False, timed: bool = False) -> Pending:
        """pcm (B, S) f32 on the device, ready on the CURRENT stream.  Returns at once; the step runs on the next slot's stream.
        timed: bracket the step with timing events on its own stream (Pending.elapsed_ms)."""
        i = self._k % self._active
        self._k += 1
        s = self.streams[i]
        s.wait_stream(torch.cuda.current_stream(self.device))   # the input was produced on the caller's stream
        with torch.cuda.stream(s):
            start = None
            if timed:
                start = torch.cuda.Event(enable_timing=True)
                start.record(s)
            logits, probs = self.runtimes[i].forward(pcm, want_logits=want_logits, want_probs=want_probs)
            ev = torch.cuda.Event(enable_timing=timed)
            ev.record(s)
        pcm.record_stream(s)
        return Pending(ev, logits, probs, start)

    def slot_of_next_submit(self) -> int:
        return self._k % self._active

    def set_active_depth(self, n: int):
        """Submit round-robin to the first n slots only (1 <= n <= depth): fewer steps in flight with the same contexts and
        streams.  Call between steps (after synchronize())."""
        n = int(n)
        if not 1 <= n <= self.depth:
            raise ValueError(f"active depth {n} outside 1..{self.depth}")
        self._active = n
        self._k = 0

    def synchronize(self):
        for s in self.streams or []:
            s.synchronize()

    def close(self):
        for r in self.runtimes:
            r.close()
        self.runtimes = []
