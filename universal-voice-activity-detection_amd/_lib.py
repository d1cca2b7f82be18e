"""ctypes binding of libuvad.so (include/uvad.h).  No torch types cross this boundary: only raw
device pointers, sizes and a hipStream_t.  There is no CPU fallback: if the shared object is
missing, ``load()`` raises with the build command instead of degrading to a torch path."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libuvad.so")

UVAD_OK = 0
ERR_NAMES = {-1: "UVAD_E_ARG", -2: "UVAD_E_HIP", -3: "UVAD_E_STATE", -4: "UVAD_E_WORKSPACE", -5: "UVAD_E_UNSUPPORTED"}
ABI_VERSION = 4


class FbankCfg(C.Structure):
    _fields_ = [("sample_rate", C.c_int), ("frame_len", C.c_int), ("frame_shift", C.c_int),
                ("n_fft", C.c_int), ("n_mels", C.c_int),
                ("preemph", C.c_float), ("low_hz", C.c_float), ("high_hz", C.c_float),
                ("log_floor", C.c_float), ("remove_dc", C.c_int), ("snip_edges", C.c_int)]


class ModelCfg(C.Structure):
    _fields_ = [("in_dim", C.c_int), ("hidden", C.c_int), ("num_layers", C.c_int),
                ("bidirectional", C.c_int), ("lin_hidden", C.c_int), ("lin_layers", C.c_int),
                ("leaky_slope", C.c_float)]


class SincNetCfg(C.Structure):
    _fields_ = [("stride", C.c_int), ("n_filters", C.c_int), ("kernel_size", C.c_int), ("c2", C.c_int), ("k2", C.c_int),
                ("c3", C.c_int), ("k3", C.c_int), ("leaky_slope", C.c_float), ("eps", C.c_float)]


class UvadError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{ERR_NAMES.get(code, code)}: {msg}")
        self.code = code


# name -> (restype, argtypes); this table is also what tests check against include/uvad.h
SIGNATURES = {
    "uvad_abi_version": (C.c_int, []),
    "uvad_create": (C.c_int, [C.c_int, C.POINTER(FbankCfg), C.POINTER(ModelCfg), C.POINTER(C.c_void_p)]),
    "uvad_set_tables": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "uvad_set_weight": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int]),
    "uvad_finalize": (C.c_int, [C.c_void_p]),
    "uvad_num_frames": (C.c_int64, [C.c_void_p, C.c_int64]),
    "uvad_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int64]),
    "uvad_fbank": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]),
    "uvad_fbank_i16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]),
    "uvad_classify": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_size_t, C.c_void_p]),
    "uvad_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p,
                               C.c_void_p, C.c_size_t, C.c_void_p]),
    "uvad_forward_i16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_size_t, C.c_void_p]),
    "uvad_sincnet_configure": (C.c_int, [C.c_void_p, C.POINTER(SincNetCfg)]),
    "uvad_sincnet_num_frames": (C.c_int64, [C.c_void_p, C.c_int64]),
    "uvad_sincnet_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int64]),
    "uvad_sincnet": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "uvad_forward_wav": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_size_t, C.c_void_p]),
    "uvad_get_taps": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "uvad_stream_state_bytes": (C.c_size_t, [C.c_void_p, C.c_int]),
    "uvad_stream_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int]),
    "uvad_stream_reset": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "uvad_stream_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                   C.c_void_p, C.c_size_t, C.c_void_p]),
    "uvad_stream_peek": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_int),
                                   C.POINTER(C.c_int)]),
    "uvad_stream_advance": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "uvad_median_filter": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "uvad_label_runs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "uvad_der_counts": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "uvad_set_gemm_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "uvad_set_recurrent_tile": (C.c_int, [C.c_void_p, C.c_int]),
    "uvad_get_recurrent_tile": (C.c_int, [C.c_void_p]),
    "uvad_get_p2_on_fp8": (C.c_int, [C.c_void_p]),
    "uvad_get_sincnet_form": (C.c_int, [C.c_void_p]),
    "uvad_weights_shared_by": (C.c_int, [C.c_void_p]),
    "uvad_set_time_chunks": (C.c_int, [C.c_void_p, C.c_int]),
    "uvad_get_time_chunks": (C.c_int, [C.c_void_p]),
    "uvad_recurrent_tile_for": (C.c_int, [C.c_void_p, C.c_int]),
    "uvad_streams_overlap": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "uvad_set_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "uvad_get_timing": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "uvad_get_layer_timing": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_int]),
    "uvad_last_error": (C.c_char_p, [C.c_void_p]),
    "uvad_destroy": (None, [C.c_void_p]),
}

_lib = None


def load():
    """dlopen libuvad.so and declare every prototype.  Raises (never falls back) if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C universal-voice-activity-detection_amd/csrc`. There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    got = lib.uvad_abi_version()
    if got != ABI_VERSION:
        raise RuntimeError(f"libuvad.so ABI {got} != binding ABI {ABI_VERSION}; rebuild the library")
    _lib = lib
    return lib


def check(lib, ctx, code):
    if code != UVAD_OK:
        msg = lib.uvad_last_error(ctx)
        raise UvadError(code, msg.decode() if msg else "")
