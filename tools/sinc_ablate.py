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
    "stamps": ["stamps"],      # s_memtime around the parts of a tile; block 0 / wave 0 prints the sums (cycles) at the end of each launch
}


def patch(src, what):
    if "epi" in what:
        i = src.index("            // lane (n, q): positions 12 sigma(g, q) + 4 rb + i")
        j = src.index("            return sig_q;\n        };")
        src = src[:i] + ("            float z = 0.f;\n#pragma unroll\n            for (int rb = 0; rb < 3; ++rb)\n#pragma unroll\n"
                         "                for (int i = 0; i < 4; ++i) z += hi[rb][i] + lo[rb][i];\n"
                         "            m[0] = m[1] = m[2] = m[3] = z;\n            if (z == 123456.75f) out_tile[n] = z;\n            const int sig_q = 0;\n") + src[j:]
        src = src.replace("            float s = sl + __shfl_xor(sl, 16);", "            if (cnt != -12345) { if (sl == 123456.75f) a.partials[n] = sl; return; }\n            float s = sl + __shfl_xor(sl, 16);")
    if "stamps" in what:
        src = src.replace("    int cur_b = -1;\n    prefetch(g_begin);", "    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};\n    int st_tiles = 0;\n#define STAMP() __builtin_amdgcn_s_memtime()\n    int cur_b = -1;\n    prefetch(g_begin);")
        src = src.replace("        __syncthreads();   // the previous tile's fragment reads are complete\n", "        const unsigned long long s0 = STAMP();\n        __syncthreads();   // the previous tile's fragment reads are complete\n        const unsigned long long s1 = STAMP();\n")
        src = src.replace("        stage(b, tile);\n        __syncthreads();\n        if (gi + 1 < g_end) prefetch(gi + 1);\n",
                          "        stage(b, tile);\n        const unsigned long long s2 = STAMP();\n        __syncthreads();\n        const unsigned long long s3 = STAMP();\n        if (gi + 1 < g_end) prefetch(gi + 1);\n        const unsigned long long s4 = STAMP();\n        unsigned long long mf = 0, ep = 0;\n")
        src = src.replace("            // lane (n, q): positions 12 sigma(g, q) + 4 rb + i, i = 0 .. 3.  u = hi + lo * 2^-11;", "            const unsigned long long m1 = STAMP();\n            // lane (n, q): positions 12 sigma(g, q) + 4 rb + i, i = 0 .. 3.  u = hi + lo * 2^-11;")
        src = src.replace("            f32x4 hi[3], lo[3];\n            f16x8 fh[2][3], fl[2][3];", "            const unsigned long long m0 = STAMP();\n            f32x4 hi[3], lo[3];\n            f16x8 fh[2][3], fl[2][3];")
        src = src.replace("            return sig_q;\n        };", "            const unsigned long long m2 = STAMP();\n            mf += m1 - m0; ep += m2 - m1;\n            return sig_q;\n        };")
        # end of tile: after the last put_stats -> accumulate
        src = src.replace("            }, wave, 4);\n        }\n    }\n}", "            }, wave, 4);\n        }\n        UVAD_TILE_END\n    }\n    UVAD_KERNEL_END\n}")
        src = src.replace("#include \"uvad_internal.h\"", "#include \"uvad_internal.h\"\n#include <cstdio>\n#define UVAD_TILE_END { const unsigned long long s5 = STAMP(); st_acc[0] += s1 - s0; st_acc[1] += s2 - s1; st_acc[2] += s3 - s2; st_acc[3] += s4 - s3; st_acc[4] += mf; st_acc[5] += ep; st_acc[6] += s5 - s4 - mf - ep; st_acc[7] += s5 - s0; ++st_tiles; }\n#define UVAD_KERNEL_END if (blockIdx.x == 0 && threadIdx.x == 0) printf(\"STAMPS stage %d tiles %d barrier1 %llu stage %llu barrier2 %llu prefetch %llu mfma %llu epilogue %llu stats %llu total %llu\\n\", ST, st_tiles, st_acc[0] / st_tiles, st_acc[1] / st_tiles, st_acc[2] / st_tiles, st_acc[3] / st_tiles, st_acc[4] / st_tiles, st_acc[5] / st_tiles, st_acc[6] / st_tiles, st_acc[7] / st_tiles);")
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
        for l in sorted(set(l for l in p.stdout.splitlines() if l.startswith("STAMPS"))):
            print(l, flush=True)


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]]()
