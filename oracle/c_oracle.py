"""ctypes binding of oracle/uvad_oracle.c (CPU ORACLE -- test infrastructure only)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class FbankCfg(C.Structure):
    _fields_ = [("sample_rate", C.c_int), ("frame_len", C.c_int), ("frame_shift", C.c_int),
                ("n_fft", C.c_int), ("n_mels", C.c_int),
                ("preemph", C.c_float), ("low_hz", C.c_float), ("high_hz", C.c_float),
                ("log_floor", C.c_float), ("remove_dc", C.c_int), ("snip_edges", C.c_int)]


class ModelCfg(C.Structure):
    _fields_ = [("in_dim", C.c_int), ("hidden", C.c_int), ("num_layers", C.c_int),
                ("bidirectional", C.c_int), ("lin_hidden", C.c_int), ("lin_layers", C.c_int),
                ("leaky_slope", C.c_float)]


def build(force=False):
    """Compile liborc.so next to the source (gcc only; a few hundred ms)."""
    so = os.path.join(_HERE, "liborc.so")
    src = os.path.join(_HERE, "uvad_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liborc.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_num_frames.restype = C.c_int64
        _LIB.orc_num_frames.argtypes = [C.c_int64, C.c_int, C.c_int, C.c_int]
        _LIB.orc_weight_count.restype = C.c_size_t
        _LIB.orc_intervals.restype = C.c_int
    return _LIB


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def default_fbank_cfg(n_mels=80, **kw):
    d = dict(sample_rate=16000, frame_len=400, frame_shift=160, n_fft=512, n_mels=n_mels,
             preemph=0.97, low_hz=20.0, high_hz=-400.0, log_floor=float(np.finfo(np.float32).eps),
             remove_dc=1, snip_edges=0)
    d.update(kw)
    return FbankCfg(**d)


def num_frames(S, cfg):
    return int(lib().orc_num_frames(S, cfg.frame_len, cfg.frame_shift, cfg.snip_edges))


WINDOW_KINDS = {"povey": 0, "hamming": 1, "hanning": 2, "rectangular": 3}


def window(kind, n):
    out = np.empty(n, np.float32)
    lib().orc_window(C.c_int(WINDOW_KINDS[kind]), C.c_int(n), _fp(out))
    return out


def mel_banks(cfg):
    out = np.empty((cfg.n_mels, cfg.n_fft // 2 + 1), np.float32)
    lib().orc_mel_banks(C.c_int(cfg.n_mels), C.c_int(cfg.n_fft), C.c_float(cfg.sample_rate),
                        C.c_float(cfg.low_hz), C.c_float(cfg.high_hz), _fp(out))
    return out


def fbank(pcm, cfg, win, mel):
    pcm = np.ascontiguousarray(pcm, np.float32)
    B, S = pcm.shape
    T = num_frames(S, cfg)
    out = np.empty((B, T, cfg.n_mels), np.float32)
    win = np.ascontiguousarray(win, np.float32)
    mel = np.ascontiguousarray(mel, np.float32)
    lib().orc_fbank(_fp(pcm), C.c_int(B), C.c_int64(S), C.byref(cfg), _fp(win), _fp(mel), _fp(out))
    return out


def fbank_f64(pcm, cfg, win, mel, threads=1):
    """Float64-throughout evaluation of the feature stage on the f32 samples / tables: (B, T, n_mels) float64.
    threads > 1: one utterance per thread (ctypes releases the GIL)."""
    pcm = np.ascontiguousarray(pcm, np.float32)
    B, S = pcm.shape
    T = num_frames(S, cfg)
    out = np.empty((B, T, cfg.n_mels), np.float64)
    win = np.ascontiguousarray(win, np.float32)
    mel = np.ascontiguousarray(mel, np.float32)
    fn = lib().orc_fbank_f64

    def one(i):
        fn(_fp(pcm[i:i + 1]), C.c_int(1), C.c_int64(S), C.byref(cfg), _fp(win), _fp(mel),
           out[i:i + 1].ctypes.data_as(C.POINTER(C.c_double)))

    if threads > 1 and B > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(int(threads)) as ex:
            list(ex.map(one, range(B)))
    else:
        for i in range(B):
            one(i)
    return out


def flatten_state_dict(sd, mcfg):
    """torch-keyed dict of numpy arrays -> flat blob in the order orc_classify expects."""
    parts = []
    for k in range(mcfg.num_layers):
        for suf in ([""] + (["_reverse"] if mcfg.bidirectional else [])):
            for nm in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                parts.append(np.asarray(sd[f"lstm.{nm}_l{k}{suf}"], np.float32).ravel())
    for j in range(mcfg.lin_layers):
        parts.append(np.asarray(sd[f"linear.{j}.weight"], np.float32).ravel())
        parts.append(np.asarray(sd[f"linear.{j}.bias"], np.float32).ravel())
    parts.append(np.asarray(sd["classifier.weight"], np.float32).ravel())
    parts.append(np.asarray(sd["classifier.bias"], np.float32).ravel())
    blob = np.ascontiguousarray(np.concatenate(parts))
    assert blob.size == lib().orc_weight_count(C.byref(mcfg)), (blob.size, lib().orc_weight_count(C.byref(mcfg)))
    return blob


def classify(sd, mcfg, feats, taps=False):
    feats = np.ascontiguousarray(feats, np.float32)
    B, T, F = feats.shape
    assert F == mcfg.in_dim
    blob = flatten_state_dict(sd, mcfg)
    W = mcfg.hidden * (2 if mcfg.bidirectional else 1)
    lin_w = mcfg.lin_hidden if mcfg.lin_layers > 0 else W
    logits = np.empty((B, T), np.float32)
    probs = np.empty((B, T), np.float32)
    lstm_out = np.empty((B, T, W), np.float32) if taps else None
    lin_out = np.empty((B, T, lin_w), np.float32) if taps else None
    lib().orc_classify(C.byref(mcfg), _fp(blob), _fp(feats), C.c_int(B), C.c_int(T),
                       _fp(lstm_out) if taps else None, _fp(lin_out) if taps else None,
                       _fp(logits), _fp(probs))
    if taps:
        return logits, probs, lstm_out, lin_out
    return logits, probs


def classify_f64(sd, mcfg, feats):
    """Float64-throughout evaluation of the same network on the f32 weights / features: logits (B, T) float64."""
    feats = np.ascontiguousarray(feats, np.float32)
    B, T, F = feats.shape
    assert F == mcfg.in_dim
    blob = flatten_state_dict(sd, mcfg)
    logits = np.empty((B, T), np.float64)
    lib().orc_classify_f64(C.byref(mcfg), _fp(blob), _fp(feats), C.c_int(B), C.c_int(T),
                           logits.ctypes.data_as(C.POINTER(C.c_double)))
    return logits


def median_filter(probs, kernel):
    probs = np.ascontiguousarray(probs, np.float32)
    B, T = probs.shape
    out = np.empty((B, T), np.uint8)
    lib().orc_median_filter(_fp(probs), C.c_int(B), C.c_int(T), C.c_int(kernel),
                            out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out


def intervals(labels, shift):
    labels = np.ascontiguousarray(labels, np.uint8)
    T = labels.shape[0]
    s = np.empty(T // 2 + 2, np.float64)
    e = np.empty(T // 2 + 2, np.float64)
    n = lib().orc_intervals(labels.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_int(T), C.c_double(shift),
                            s.ctypes.data_as(C.POINTER(C.c_double)), e.ctypes.data_as(C.POINTER(C.c_double)),
                            C.c_int(s.size))
    return list(zip(s[:n].tolist(), e[:n].tolist()))
