"""Triage of one fuzz_ops simdrop case: FUZZ_VERBOSE=1 python tools/probes/repro_simdrop.py B img patch D H F blocks p"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("FUZZ_VERBOSE", "1")
import fuzz_ops
a = sys.argv[1:]
args = (int(a[0]), int(a[1]), int(a[2]), int(a[3]), int(a[4]), int(a[5]), int(a[6]), float(a[7]))
print("simdrop", args, flush=True)
fuzz_ops.simmim_drop_case(fuzz_ops.ops, *args)
