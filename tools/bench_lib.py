#!/usr/bin/env python3
"""bench.py against another build of the library (A/B runs of kernel variants inside ONE gpurun call -- boxes differ by
several per cent, so numbers from different calls do not compare):  MI355ENC_LIB=/path/libmi355enc_x.so python tools/bench_lib.py [bench.py flags]"""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ceracoder_amd import enc as E
E.LIB_PATH = os.environ.get("MI355ENC_LIB", E.LIB_PATH)
import runpy
sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
