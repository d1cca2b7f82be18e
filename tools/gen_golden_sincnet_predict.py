#!/usr/bin/env python3
"""Generate tests/golden/sincnet_predict_geometry.json by EXECUTING the reference's own frame -> time code for the SincNet predict
path.  Build container only (needs /root/reference); nothing of the reference's text is written anywhere.

What runs:
  * src/utils/receptive_field.py is pure Python and is imported from its file (get_num_frames, receptive_field_size).
  * src/scripts/predict_sincnet.py cannot be imported as a module (pytorch_lightning, lhotse, wandb, the data module), so the
    pieces that matter are taken out of its syntax tree and compiled as they stand:
      - the function get_timestamp_from_sample_boundary (:492-504), merge_intervals_with_buffer (:507-528), split_into_windows
        (:531-540);
      - the run-length walk inside get_new_cuts (:348-370: `start = None`, the `for k, value in enumerate(obj["tensor"])` loop
        and the `if start is not None` tail), executed with obj = {"tensor": labels, "duration": seconds} and an empty
        pred_intervals list in scope.
The fixture stores inputs and the outputs of that code only."""
import ast
import importlib.util
import json
import os
import random
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
SCRIPT = os.path.join(REF, "src", "scripts", "predict_sincnet.py")


def _load_receptive_field():
    spec = importlib.util.spec_from_file_location("ref_receptive_field", os.path.join(REF, "src", "utils", "receptive_field.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _compile_pieces():
    tree = ast.parse(open(SCRIPT).read(), SCRIPT)
    ns = {}
    wanted = {"get_timestamp_from_sample_boundary", "merge_intervals_with_buffer", "split_into_windows"}
    fns = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in wanted]
    assert {f.name for f in fns} == wanted
    exec(compile(ast.Module(body=fns, type_ignores=[]), SCRIPT, "exec"), ns)
    # the walk: inside get_new_cuts, the `for i, obj in enumerate(recording_tensor)` loop that builds pred_intervals
    gnc = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "get_new_cuts")
    walk = None
    for node in gnc.body:
        if isinstance(node, ast.For):
            body = node.body
            for j, st in enumerate(body):
                if (isinstance(st, ast.Assign) and isinstance(st.targets[0], ast.Name) and st.targets[0].id == "start"
                        and isinstance(st.value, ast.Constant) and st.value.value is None
                        and isinstance(body[j + 1], ast.For) and isinstance(body[j + 2], ast.If)):
                    walk = body[j:j + 3]
    assert walk is not None, "the run-length walk of get_new_cuts was not found"
    code = compile(ast.Module(body=walk, type_ignores=[]), SCRIPT, "exec")

    def run_walk(labels, duration):
        scope = dict(ns)
        scope.update(obj={"tensor": labels, "duration": duration}, pred_intervals=[])
        exec(code, scope)
        return [list(iv) for iv in scope["pred_intervals"]]

    return ns, run_walk


def main():
    rf = _load_receptive_field()
    ns, run_walk = _compile_pieces()
    ts = ns["get_timestamp_from_sample_boundary"]
    rng = random.Random(20261004)
    out = {"receptive_field": [rf.receptive_field_size(1), rf.receptive_field_size(2)], "num_frames": {}, "timestamps": [], "walks": [],
           "merge": [], "split": []}
    for n in (80000, 48001, 64000, 79999, 160000, 379200, 991, 1261, 1531, 16000 * 30):
        out["num_frames"][str(n)] = rf.get_num_frames(n)
    for _ in range(200):
        a = rng.randrange(0, 4000)
        b = a + rng.randrange(0, 600)
        d = rng.choice([5, 5.0, 12.0, 23.7, 30.0, 61.25, 4.2])
        s, e = ts(a, b, d)
        out["timestamps"].append([a, b, d, s, e])
    for case in range(24):
        d = rng.choice([5.0, 12.0, 23.7, 30.0, 61.25, 4.2, 100.0])
        n = rf.get_num_frames(int(16000 * d)) + 1
        if case == 0:
            lab = [0] * n
        elif case == 1:
            lab = [1] * n
        else:
            lab, v = [], rng.randrange(2)
            while len(lab) < n:
                lab += [v] * rng.randrange(1, rng.choice([8, 80, 400]))
                v ^= 1
            lab = lab[:n]
        out["walks"].append({"duration": d, "labels": "".join(map(str, lab)), "intervals": run_walk(lab, d)})
    for _ in range(12):
        iv = []
        for _ in range(rng.randrange(0, 7)):
            a = rng.randrange(0, 50)
            iv.append([a, a + rng.randrange(1, 15)])
        buf = rng.choice([0, 0.5, 2])
        out["merge"].append({"intervals": iv, "duration": 60, "buffer": buf, "merged": ns["merge_intervals_with_buffer"](iv, 60, buf)})
        out["split"].append({"intervals": [list(x) for x in out["merge"][-1]["merged"]], "window": 10,
                             "split": ns["split_into_windows"]([list(x) for x in out["merge"][-1]["merged"]], window=10)})
    path = os.path.join(REPO, "tests", "golden", "sincnet_predict_geometry.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print(f"{path}: {os.path.getsize(path)} bytes; RF {out['receptive_field']}, 5 s -> {out['num_frames']['80000']} frames")


if __name__ == "__main__":
    main()
