"""VadRuntime: owns one ``uvad_ctx`` (device, model) and the device workspace; torch tensors are
used only as device-memory containers whose ``data_ptr()`` is handed to the C ABI together with
torch's current HIP stream.  Every method fails loudly if libuvad.so or the GPU is missing."""
import ctypes as C
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from .features import FbankConfig, make_mel_matrix, make_window


def _model_cfg(encoding_dim: int, lstm: dict, linear: dict, leaky_slope: float = 0.01) -> _lib.ModelCfg:
    return _lib.ModelCfg(int(encoding_dim), int(lstm["hidden_size"]), int(lstm["num_layers"]),
                         int(bool(lstm["bidirectional"])), int(linear.get("hidden_size", 128)),
                         int(linear.get("num_layers", 0)), float(leaky_slope))


class VadRuntime:
    def __init__(self, device, fbank: Optional[FbankConfig] = None, model: Optional[dict] = None, sincnet: Optional[dict] = None):
        """model: {"encoding_dim": int, "lstm": {...merged defaults...}, "linear": {...}} or None.
        sincnet: SincNet.config() (stride, n_filters, kernel_size, c2, k2, c3, k3, leaky_slope, eps) or None."""
        self.lib = _lib.load()
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError(f"VadRuntime needs a GPU device, got {dev}; there is no CPU path")
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible to torch; libuvad has no CPU fallback")
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        self.device = dev
        self.fbank_cfg = fbank
        self.model_cfg = model
        fb_c = mc_c = None
        if fbank is not None:
            fb_c = _lib.FbankCfg(fbank.sampling_rate, fbank.frame_len_samples, fbank.frame_shift_samples, fbank.n_fft,
                                 fbank.num_filters, fbank.preemph_coeff, fbank.low_freq, fbank.high_freq,
                                 fbank.energy_floor, int(fbank.remove_dc_offset), int(fbank.snip_edges))
        if model is not None:
            mc_c = _model_cfg(model["encoding_dim"], model["lstm"], model["linear"], model.get("leaky_slope", 0.01))
        self._fb_c, self._mc_c = fb_c, mc_c
        self.ctx = C.c_void_p()
        code = self.lib.uvad_create(dev.index, C.byref(fb_c) if fb_c else None, C.byref(mc_c) if mc_c else None,
                                    C.byref(self.ctx))
        try:
            _lib.check(self.lib, self.ctx, code)
        except Exception:
            self.close()
            raise
        if fbank is not None:
            win = make_window(fbank.window_type, fbank.frame_len_samples)
            mel = make_mel_matrix(fbank.num_filters, fbank.n_fft, fbank.sampling_rate, fbank.low_freq,
                                  fbank.high_freq, fbank.norm_filters)
            self.set_tables(win, mel)
        self._ws = None
        self._finalized = False
        self._sn_c = None
        if sincnet is not None:
            self._sn_c = _lib.SincNetCfg(int(sincnet["stride"]), int(sincnet["n_filters"]), int(sincnet["kernel_size"]),
                                         int(sincnet["c2"]), int(sincnet["k2"]), int(sincnet["c3"]), int(sincnet["k3"]),
                                         float(sincnet.get("leaky_slope", 0.01)), float(sincnet.get("eps", 1e-5)))
            try:
                self._check(self.lib.uvad_sincnet_configure(self.ctx, C.byref(self._sn_c)))
            except Exception:
                self.close()
                raise

    # ------------------------------------------------------------------ setup
    def set_tables(self, window: np.ndarray, mel: np.ndarray):
        window = np.ascontiguousarray(window, np.float32)
        mel = np.ascontiguousarray(mel, np.float32)
        assert window.shape == (self._fb_c.frame_len,), window.shape
        assert mel.shape == (self._fb_c.n_mels, self._fb_c.n_fft // 2 + 1), mel.shape
        self._check(self.lib.uvad_set_tables(self.ctx, window.ctypes.data, mel.ctypes.data))

    def load_state_dict(self, sd: Dict[str, "torch.Tensor"]):
        """torch-keyed tensors (CPU or GPU; a Lightning ``model.`` prefix is accepted)."""
        for k, v in sd.items():
            if ".filterbank." in k:
                continue   # band edges / buffers of ParamSincFB: the materialised bank is sent as "...conv1d.0.filters"
            a = np.ascontiguousarray(v.detach().to("cpu", torch.float32).numpy() if torch.is_tensor(v) else v, np.float32)
            shape = (C.c_int64 * a.ndim)(*a.shape)
            self._check(self.lib.uvad_set_weight(self.ctx, k.encode(), a.ctypes.data, shape, a.ndim))
        self._check(self.lib.uvad_finalize(self.ctx))
        self._finalized = True
        self._weights_gen = getattr(self, "_weights_gen", 0) + 1   # uvad_finalize re-allocates every weight buffer: graphs captured before are stale

    # ------------------------------------------------------------------ helpers
    def _check(self, code):
        _lib.check(self.lib, self.ctx, code)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _dev_f32(self, t: "torch.Tensor", name: str) -> "torch.Tensor":
        if not torch.is_tensor(t) or t.device != self.device:
            raise RuntimeError(f"{name} must be a tensor on {self.device} (got {getattr(t, 'device', type(t))})")
        if t.dtype != torch.float32:
            t = t.float()
        return t.contiguous()

    def num_frames(self, S: int) -> int:
        return int(self.lib.uvad_num_frames(self.ctx, S))

    def workspace(self, B: int, T: int) -> "torch.Tensor":
        need = int(self.lib.uvad_workspace_bytes(self.ctx, B, T))
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    # ------------------------------------------------------------------ compute
    def fbank(self, pcm: "torch.Tensor") -> "torch.Tensor":
        """pcm (B,S) f32 or int16 on the GPU -> (B,T,n_mels) f32."""
        with torch.cuda.device(self.device):
            if pcm.dtype == torch.int16:
                if pcm.device != self.device:
                    raise RuntimeError(f"pcm must be on {self.device}")
                pcm = pcm.contiguous()
                fn = self.lib.uvad_fbank_i16
            else:
                pcm = self._dev_f32(pcm, "pcm")
                fn = self.lib.uvad_fbank
            B, S = pcm.shape
            T = self.num_frames(S)
            feats = torch.empty((B, T, self._fb_c.n_mels), dtype=torch.float32, device=self.device)
            self._check(fn(self.ctx, pcm.data_ptr(), B, S, feats.data_ptr(), self._stream()))
            return feats

    def classify(self, feats: "torch.Tensor", want_logits=True, want_probs=True):
        """feats (B,T,F) on the GPU -> (logits (B,T) | None, probs (B,T) | None)."""
        with torch.cuda.device(self.device):
            feats = self._dev_f32(feats, "feats")
            B, T, F = feats.shape
            if F != self._mc_c.in_dim:
                raise ValueError(f"feature dim {F} != encoding_dim {self._mc_c.in_dim}")
            ws = self.workspace(B, T)
            logits = torch.empty((B, T), dtype=torch.float32, device=self.device) if want_logits else None
            probs = torch.empty((B, T), dtype=torch.float32, device=self.device) if want_probs else None
            self._check(self.lib.uvad_classify(self.ctx, feats.data_ptr(), B, T,
                                               logits.data_ptr() if want_logits else None,
                                               probs.data_ptr() if want_probs else None,
                                               ws.data_ptr(), ws.numel(), self._stream()))
            self._last_bt = (B, T)
            return logits, probs

    def forward(self, pcm: "torch.Tensor", want_logits=True, want_probs=True):
        """pcm (B,S) f32 (or int16, as read from a wav file) on the GPU -> (logits, probs); features never leave the workspace."""
        with torch.cuda.device(self.device):
            if pcm.dtype == torch.int16:
                if pcm.device != self.device:
                    raise RuntimeError(f"pcm must be on {self.device}")
                pcm = pcm.contiguous()
                fn = self.lib.uvad_forward_i16
            else:
                pcm = self._dev_f32(pcm, "pcm")
                fn = self.lib.uvad_forward
            B, S = pcm.shape
            T = self.num_frames(S)
            ws = self.workspace(B, T)
            logits = torch.empty((B, T), dtype=torch.float32, device=self.device) if want_logits else None
            probs = torch.empty((B, T), dtype=torch.float32, device=self.device) if want_probs else None
            self._check(fn(self.ctx, pcm.data_ptr(), B, S,
                           logits.data_ptr() if want_logits else None,
                           probs.data_ptr() if want_probs else None,
                           ws.data_ptr(), ws.numel(), self._stream()))
            self._last_bt = (B, T)
            return logits, probs

    # ------------------------------------------------------------------ SincNet front end (PyanNet)
    def sincnet_num_frames(self, S: int) -> int:
        return int(self.lib.uvad_sincnet_num_frames(self.ctx, S))

    def _wav_ws(self, B: int, S: int, T: int, with_classifier: bool) -> "torch.Tensor":
        need = int(self.lib.uvad_sincnet_workspace_bytes(self.ctx, B, S))
        if with_classifier:
            need += int(self.lib.uvad_workspace_bytes(self.ctx, B, T))
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def sincnet(self, wav: "torch.Tensor") -> "torch.Tensor":
        """wav (B,S) f32 on the GPU -> SincNet features (B, frames, c3)."""
        if self._sn_c is None:
            raise RuntimeError("this runtime was created without a SincNet configuration")
        with torch.cuda.device(self.device):
            wav = self._dev_f32(wav, "wav")
            B, S = wav.shape
            T = self.sincnet_num_frames(S)
            if T <= 0:
                raise ValueError(f"{S} samples are too short for one SincNet frame")
            ws = self._wav_ws(B, S, T, False)
            feats = torch.empty((B, T, self._sn_c.c3), dtype=torch.float32, device=self.device)
            self._check(self.lib.uvad_sincnet(self.ctx, wav.data_ptr(), B, S, feats.data_ptr(), ws.data_ptr(), ws.numel(), self._stream()))
            return feats

    def forward_wav(self, wav: "torch.Tensor", want_logits=True, want_probs=True):
        """wav (B,S) f32 on the GPU -> (logits, probs) of PyanNet (SincNet -> LSTM stack -> head)."""
        if self._sn_c is None:
            raise RuntimeError("this runtime was created without a SincNet configuration")
        with torch.cuda.device(self.device):
            wav = self._dev_f32(wav, "wav")
            B, S = wav.shape
            T = self.sincnet_num_frames(S)
            if T <= 0:
                raise ValueError(f"{S} samples are too short for one SincNet frame")
            ws = self._wav_ws(B, S, T, True)
            logits = torch.empty((B, T), dtype=torch.float32, device=self.device) if want_logits else None
            probs = torch.empty((B, T), dtype=torch.float32, device=self.device) if want_probs else None
            self._check(self.lib.uvad_forward_wav(self.ctx, wav.data_ptr(), B, S,
                                                  logits.data_ptr() if want_logits else None,
                                                  probs.data_ptr() if want_probs else None,
                                                  ws.data_ptr(), ws.numel(), self._stream()))
            self._last_bt = (B, T)
            return logits, probs

    def taps(self, lin: bool = True):
        """(lstm_out (B,T,H*D), lin_out (B,T,lin_hidden) | None) of the last classify/forward.  lin=False: the LSTM tap only (the
        feed-forward tap is not available in GEMM mode "f16p3" where the fused head ran: include/uvad.h)."""
        with torch.cuda.device(self.device):
            B, T = self._last_bt
            W = self._mc_c.hidden * (2 if self._mc_c.bidirectional else 1)
            y = torch.empty((B, T, W), dtype=torch.float32, device=self.device)
            z = None
            if lin and self._mc_c.lin_layers > 0:
                z = torch.empty((B, T, self._mc_c.lin_hidden), dtype=torch.float32, device=self.device)
            self._check(self.lib.uvad_get_taps(self.ctx, B, T, y.data_ptr(), z.data_ptr() if z is not None else None,
                                               self._ws.data_ptr(), self._stream()))
            return y, z

    def median_filter(self, probs: "torch.Tensor", kernel: int) -> "torch.Tensor":
        with torch.cuda.device(self.device):
            probs = self._dev_f32(probs, "probs")
            B, T = probs.shape
            out = torch.empty((B, T), dtype=torch.uint8, device=self.device)
            self._check(self.lib.uvad_median_filter(self.ctx, probs.data_ptr(), B, T, int(kernel), out.data_ptr(), self._stream()))
            return out

    # ------------------------------------------------------------------ streaming (BASELINE cfg 5)
    def stream_open(self, B: int, chunk: int, graphs: bool = False):
        """Allocate and reset the carried state of B lock-step streams fed `chunk` samples per step.
        graphs: replay each distinct step shape as a hipGraph (see stream_step).  Off by default: a step of a causal 128-unit model
        is one launch (lstm_stack_kernel: feature stage, every layer and the head), 0.059 ms at BASELINE cfg 5 (512 feeds,
        20 ms chunks); a graph has nothing left to shorten (with the 14 per-layer launches of round 2 a replayed step took 0.133 ms
        against 0.125 ms enqueued kernel by kernel)."""
        with torch.cuda.device(self.device):
            nbytes = int(self.lib.uvad_stream_state_bytes(self.ctx, B))
            if nbytes == 0:
                raise RuntimeError("streaming needs a runtime built with both a FbankConfig and a model")
            state = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            ws = torch.empty(int(self.lib.uvad_stream_workspace_bytes(self.ctx, B, chunk)), dtype=torch.uint8, device=self.device)
            self._check(self.lib.uvad_stream_reset(self.ctx, state.data_ptr(), B, self._stream()))
            kmax = chunk // self._fb_c.frame_shift + 1
            return {"state": state, "ws": ws, "B": B, "chunk": chunk,
                    "out": torch.empty((B, kmax), dtype=torch.float32, device=self.device),
                    "in": torch.empty((B, chunk), dtype=torch.float32, device=self.device),
                    "graphs": {} if graphs else None, "weights_gen": getattr(self, "_weights_gen", 0)}

    def stream_step(self, st, pcm_chunk: "torch.Tensor") -> "torch.Tensor":
        """pcm_chunk (B, chunk) f32 on the GPU -> logits (B, k) of the k frames completed by this chunk (a view of a buffer that
        the next step overwrites).

        With stream_open(graphs=True): what a step enqueues depends on the stream group's host-side counters only through
        (k, offset, parity, first) (uvad_stream_peek), so the first step with a given combination is captured into a hipGraph
        while it is enqueued and later ones replay that graph and move the counters with uvad_stream_advance.  At the reference
        geometry (20 ms chunks, 10 ms shift) a group settles into two graphs."""
        with torch.cuda.device(self.device):
            pcm_chunk = self._dev_f32(pcm_chunk, "pcm_chunk")
            if tuple(pcm_chunk.shape) != (st["B"], st["chunk"]):
                raise ValueError(f"expected a ({st['B']}, {st['chunk']}) chunk, got {tuple(pcm_chunk.shape)}")
            out = st["out"]

            def enqueue(src):
                return self.lib.uvad_stream_step(self.ctx, src.data_ptr(), st["B"], st["chunk"], st["state"].data_ptr(),
                                                 out.data_ptr(), out.shape[1], st["ws"].data_ptr(), st["ws"].numel(), self._stream())

            graphs = st.get("graphs")
            if graphs is not None and st.get("weights_gen") != getattr(self, "_weights_gen", 0):
                graphs.clear()           # captured before a weight hot-swap: their kernel nodes point at freed buffers
                st["weights_gen"] = getattr(self, "_weights_gen", 0)
            if graphs is None:
                k = enqueue(pcm_chunk)
            else:
                kk, off, par, first = C.c_int(), C.c_int64(), C.c_int(), C.c_int()
                self._check(self.lib.uvad_stream_peek(self.ctx, st["state"].data_ptr(), st["chunk"], C.byref(kk), C.byref(off),
                                                      C.byref(par), C.byref(first)))
                key = (kk.value, off.value, par.value, first.value)
                st["in"].copy_(pcm_chunk)                    # the graphs read their input from a fixed buffer
                g = graphs.get(key)
                if g is None:
                    cur = torch.cuda.current_stream(self.device)
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):                # capture: the step's launches are recorded, not run; its counters advance
                        k = enqueue(st["in"])
                    if k < 0:
                        self._check(k)
                    graphs[key] = g
                    torch.cuda.current_stream(self.device).wait_stream(cur)
                    g.replay()                               # now run it
                else:
                    g.replay()
                    k = self.lib.uvad_stream_advance(self.ctx, st["state"].data_ptr(), st["chunk"])
            if k < 0:
                self._check(k)
            return out[:, :k]

    def set_gemm_mode(self, mode: str):
        """"f32": exact f32 MFMA; "f16p": split-f16 on the f16 matrix cores (default: the weight-stationary kernel for the large
        projections, the tile-streaming one elsewhere); "f16p_stream": split-f16 with the tile-streaming kernel only (same bits,
        kept for A/B runs and as the reference of the bit-identity test); "f16p3": "f16p" with three instead of four MFMA products
        per f32-equivalent product in the large-launch kernels (weights rounded to 22 bits; faster, see include/uvad.h)."""
        self._check(self.lib.uvad_set_gemm_mode(self.ctx, {"f32": 0, "f16p": 1, "f16p_stream": 2, "f16p3": 3}[mode]))

    def set_recurrent_tile(self, sequences: int):
        """Sequences per recurrent workgroup: 0 (default) = chosen from the batch size, 4 = latency form, 16 = throughput form."""
        self._check(self.lib.uvad_set_recurrent_tile(self.ctx, int(sequences)))

    def recurrent_tile_for(self, batch: int) -> int:
        """What the default choice would launch for `batch` sequences (4 or 16)."""
        return int(self.lib.uvad_recurrent_tile_for(self.ctx, int(batch)))

    def recurrent_tile(self) -> int:
        """What the most recent classify / forward launched (4 or 16)."""
        return int(self.lib.uvad_get_recurrent_tile(self.ctx))

    def p2_on_fp8(self) -> bool:
        """True if the 16-sequence recurrence runs its P2 x h product on the 8-bit matrix pipe (every P2 element exactly bf8: include/uvad.h)."""
        return bool(self.lib.uvad_get_p2_on_fp8(self.ctx))

    def weights_shared_by(self) -> int:
        """How many contexts of this process use this context's packed weights on the device (include/uvad.h: contexts finalized with
        identical tensors share them; 1 = not shared)."""
        return int(self.lib.uvad_weights_shared_by(self.ctx))

    def sincnet_form(self) -> str:
        """What the most recent sincnet() / forward_wav() ran: "f16p" (the split-f16 stages of sincnet_f16p.hip) or "f32" (sincnet.hip)."""
        return "f16p" if int(self.lib.uvad_get_sincnet_form(self.ctx)) == 1 else "f32"

    def set_time_chunks(self, chunks: int):
        """Time chunks per layer for a batch that runs alone (include/uvad.h): 0 = automatic (default), 1 = off, n = that many.  The
        projection of chunk i + 1 runs on a stream of the library's own beside the recurrence of chunk i; outputs are bit-identical."""
        self._check(self.lib.uvad_set_time_chunks(self.ctx, int(chunks)))

    def time_chunks(self) -> int:
        """What the most recent classify / forward ran (1 = not chunked)."""
        return int(self.lib.uvad_get_time_chunks(self.ctx))

    def der_counts(self, pred: "torch.Tensor", gt: "torch.Tensor") -> "torch.Tensor":
        """pred, gt (B, T) uint8 0/1 on the GPU -> (B, 2) int32 counts {false alarm, missed detection}."""
        with torch.cuda.device(self.device):
            pred = pred.to(torch.uint8).contiguous()
            gt = gt.to(self.device, torch.uint8).contiguous()
            B, T = pred.shape
            out = torch.empty((B, 2), dtype=torch.int32, device=self.device)
            self._check(self.lib.uvad_der_counts(self.ctx, pred.data_ptr(), gt.data_ptr(), B, T, out.data_ptr(), self._stream()))
            return out

    def label_runs(self, labels: "torch.Tensor", max_runs: int = 0):
        """labels (B, T) uint8 0/1 on the GPU -> (runs (B, max_runs, 2) int32, counts (B,) int32), both on the GPU."""
        with torch.cuda.device(self.device):
            if not torch.is_tensor(labels) or labels.device != self.device:
                raise RuntimeError(f"labels must be a tensor on {self.device}")
            labels = labels.to(torch.uint8).contiguous()
            B, T = labels.shape
            max_runs = int(max_runs) if max_runs > 0 else (T + 1) // 2
            runs = torch.empty((B, max_runs, 2), dtype=torch.int32, device=self.device)
            counts = torch.empty((B,), dtype=torch.int32, device=self.device)
            self._check(self.lib.uvad_label_runs(self.ctx, labels.data_ptr(), B, T, max_runs, runs.data_ptr(), counts.data_ptr(), self._stream()))
            return runs, counts

    def streams_overlap(self, a: "torch.cuda.Stream", b: "torch.cuda.Stream") -> bool:
        """True if kernels on the two HIP streams run concurrently (they sit on different hardware queues)."""
        r = self.lib.uvad_streams_overlap(self.ctx, C.c_void_p(a.cuda_stream), C.c_void_p(b.cuda_stream))
        if r < 0:
            self._check(r)
        return r == 1

    def set_timing(self, on: bool):
        self._check(self.lib.uvad_set_timing(self.ctx, int(on)))

    def timing_ms(self):
        buf = (C.c_float * 5)()
        self._check(self.lib.uvad_get_timing(self.ctx, buf))
        return dict(zip(("fbank", "proj", "recurrent", "head", "total"), [float(x) for x in buf]))

    def layer_timing_ms(self):
        """[(projection ms, recurrence ms)] per LSTM layer of the last timed call."""
        n = 2 * self._mc_c.num_layers
        buf = (C.c_float * n)()
        got = self.lib.uvad_get_layer_timing(self.ctx, buf, n)
        if got < 0:
            self._check(got)
        return [(float(buf[2 * k]), float(buf[2 * k + 1])) for k in range(got // 2)]

    # ------------------------------------------------------------------ teardown
    def close(self):
        if getattr(self, "ctx", None) is not None and self.ctx:
            self.lib.uvad_destroy(self.ctx)
            self.ctx = C.c_void_p()
        self._ws = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
