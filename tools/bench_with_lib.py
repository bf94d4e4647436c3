#!/usr/bin/env python3
"""A/B helper: `python tools/bench_with_lib.py path/to/libfavit.so [bench.py arguments]` runs bench.py against another
build of the library (e.g. tools/ab_build/libfavit_head.so built from a git worktree of the previous commit) so that
two builds can be timed back to back on ONE box -- boxes differ by +-1.5 %, more than most single changes."""
import importlib, os, runpy, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
lib = os.path.abspath(sys.argv[1])
pkg = importlib.import_module("focused-attention-vit_amd")
pkg._abi.LIB_PATH = lib
sys.argv = [os.path.join(root, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
