#!/usr/bin/env python3
"""bench.py against another build of the library (A/B on one box): python tools/bench_lib.py <lib.so> [bench args...]"""
import os, runpy, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import uvad_amd
from uvad_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(root, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
