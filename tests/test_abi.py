"""CPU tests of the C-ABI boundary: libuvad.so loads, exports every symbol include/uvad.h declares,
the ctypes table covers exactly that set, and (without a GPU) it fails loudly instead of falling back."""
import ctypes as C
import os
import re
import subprocess

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "uvad.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(uvad_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from uvad_amd import _lib
    return _lib


def test_library_exports_every_declared_symbol(built):
    lib = built.load()
    syms = _header_symbols()
    assert len(syms) >= 19
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/uvad.h but not exported"
    assert sorted(built.SIGNATURES) == syms, "ctypes table and header disagree"
    assert lib.uvad_abi_version() == built.ABI_VERSION
    out = subprocess.check_output(["nm", "-D", "--defined-only", built.LIB_PATH], text=True)
    exported = sorted(set(re.findall(r" T (uvad_[a-z0-9_]+)", out)))
    assert exported == syms


def test_code_object_targets_gfx950_only(built):
    blob = open(built.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    for other in (b"gfx90a", b"gfx942", b"sm_90", b"nvptx"):
        assert other not in blob


def test_struct_layout_matches_header(built):
    assert C.sizeof(built.FbankCfg) == 11 * 4
    assert C.sizeof(built.ModelCfg) == 7 * 4


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_gpu_is_a_loud_error_not_a_fallback(built):
    lib = built.load()
    ctx = C.c_void_p()
    mc = built.ModelCfg(64, 128, 4, 1, 128, 2, 0.01)
    code = lib.uvad_create(0, None, C.byref(mc), C.byref(ctx))
    assert code == -2
    assert b"no CPU fallback" in lib.uvad_last_error(ctx)
    lib.uvad_destroy(ctx)
    import uvad_amd
    m = uvad_amd.PyanNet2(encoding_dim=64)
    m.build()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 8, 64))
    with pytest.raises(RuntimeError):
        uvad_amd.Fbank(uvad_amd.FbankConfig(device="cuda")).extract_batch([torch.zeros(1600)], 16000)


def test_missing_library_is_a_loud_error(built, monkeypatch):
    monkeypatch.setattr(built, "_lib", None)
    monkeypatch.setattr(built, "LIB_PATH", os.path.join(ROOT, "does_not_exist.so"))
    with pytest.raises(RuntimeError, match="has not been built"):
        built.load()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "universal-voice-activity-detection_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, fn)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "liborc" not in txt, fn
    for fn in ("main.py", "uvad_amd.py"):
        assert "oracle" not in open(os.path.join(ROOT, fn)).read()
