#!/usr/bin/env python3
"""Runs another tool against an alternative build of the library:
   python tools/try_lib.py <path/to/libretinahead.so> tools/bench_stages.py [args...]"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pytorch-retinanet_amd")):
    sys.path.insert(0, p)
import retinahead._lib as L  # noqa: E402

L.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
