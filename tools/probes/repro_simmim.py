"""Triage of fuzz_ops simmim cases: python tools/probes/repro_simmim.py "B img patch D H F" ["B img patch D H F" ...]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("FUZZ_VERBOSE", "1")
import fuzz_ops
for c in sys.argv[1:]:
    args = tuple(int(v) for v in c.split())
    print("simmim", args, flush=True)
    try:
        fuzz_ops.simmim_case(fuzz_ops.ops, *args)
    except AssertionError as e:
        print("  FAILED", str(e)[:300], flush=True)
