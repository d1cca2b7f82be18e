#!/usr/bin/env python3
"""Timing-only ablation builds of csrc/sincnet_f16p.hip (patched COPIES under build/sinc_abl/, never the shipped source): which part of a
tile's time is the epilogue, the staging (registers -> LDS), the prefetch (global -> registers).  Results of these builds are wrong by
construction; only their durations are read.
    python tools/sinc_ablate.py build            # here (hipcc cross-compiles): build/sinc_abl/libuvad_<variant>.so
    python tools/sinc_ablate.py run              # on the GPU box: sincnet_ms of every variant, one JSON line each"""
import json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "universal-voice-activity-detection_amd", "csrc")
OUT = os.path.join(ROOT, "build", "sinc_abl")
VARIANTS = {
    "base": [],
    "noepi": ["epi"],
    "nostage": ["stage"],
    "noprefetch": ["prefetch"],
    "mfma_only": ["epi", "stage", "prefetch"],
}


def patch(src, what):
    if "epi" in what:
        i = src.index("            // lane (n, q): positions 12 sigma(g, q) + 4 rb + i")
        j = src.index("            return sig_q;\n        };")
        src = src[:i] + ("            float z = 0.f;\n#pragma unroll\n            for (int rb = 0; rb < 3; ++rb)\n#pragma unroll\n"
                         "                for (int i = 0; i < 4; ++i) z += hi[rb][i] + lo[rb][i];\n"
                         "            m[0] = m[1] = m[2] = m[3] = z;\n            if (z == 123456.75f) out_tile[n] = z;\n            const int sig_q = 0;\n") + src[j:]
        src = src.replace("            float s = sl + __shfl_xor(sl, 16);", "            if (cnt != -12345) { if (sl == 123456.75f) a.partials[n] = sl; return; }\n            float s = sl + __shfl_xor(sl, 16);")
    if "stage" in what:
        src = src.replace("        stage(b, tile);\n", "        if (a.Lpool == -7) stage(b, tile);\n")
    if "prefetch" in what:
        src = src.replace("    prefetch(g_begin);\n", "#pragma unroll\n    for (int i = 0; i < NPRE; ++i) pre[i] = make_float4(0.f, 0.f, 0.f, 0.f);\n    if (a.Lpool == -7) prefetch(g_begin);\n")
        src = src.replace("        if (gi + 1 < g_end) prefetch(gi + 1);\n", "        if (a.Lpool == -7) prefetch(gi + 1);\n")
    return src


def build():
    os.makedirs(OUT, exist_ok=True)
    src = open(os.path.join(CSRC, "sincnet_f16p.hip")).read()
    objs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".o") and f != "sincnet_f16p.o"]
    for name, what in VARIANTS.items():
        p = os.path.join(OUT, f"sincnet_f16p_{name}.hip")
        open(p, "w").write(patch(src, what))
        o = p[:-4] + ".o"
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math", "-mllvm", "-amdgpu-mfma-vgpr-form=1",
                               "-fno-slp-vectorize", f"-I{CSRC}", f"-I{os.path.join(ROOT, 'include')}", "-c", p, "-o", o])
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(OUT, f"libuvad_{name}.so"), o] + objs)
        print("built", name)


def run():
    for name in VARIANTS:
        lib = os.path.join(OUT, f"libuvad_{name}.so")
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_sincnet.py"), "--lib", lib, "--check", "0", "--reps", "20", "--stages"],
                           capture_output=True, text=True, timeout=300)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        d = json.loads(line[-1]) if line else {"error": p.stderr[-400:]}
        print(json.dumps({"variant": name, "sincnet_ms": d.get("sincnet_ms"), "stage_ms": d.get("stage_ms")}), flush=True)


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]]()
