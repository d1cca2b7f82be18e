#!/usr/bin/env python3
"""bench.py against another build of the library (A/B of kernel variants):  python tools/bench_with_lib.py path/to/lib.so [bench.py flags]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from uvad_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(root, "bench.py")] + sys.argv[2:]
import bench
bench.main()
