"""Does ONE long-running workgroup on another stream slow a full grid of lean episodes?  (The hoisted replay's workgroup runs 0.3-1.4 ms
beside the lean grid: profiles/r04_ab_notes.log.)  One idle workgroup (eg_debug_occupy: 1 KB or 150 KB of LDS, few or 200+ registers a lane) on the library's side stream, then
16 384 sampled episodes on the null stream; the lean grid's duration from the library's events, with and without the sleeper.  python scripts/side_kernel_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from eirgrid_amd import synthetic_world
from eirgrid_amd.engine import ActionWeights, Engine

eng = Engine(synthetic_world())
w = ActionWeights()
eng.upload_snapshot(w)
from eirgrid_amd import _native as N
cases = [("no side workgroup", -1), ("1 KB LDS, few registers", 0), ("150 KB LDS", 1), ("200+ registers a lane", 2), ("150 KB LDS + 200+ registers", 3), ("the same, busy", 4), ("the same, busy, LDS writes + barriers", 5), ("no side workgroup", -1)]
for label, variant in cases:
    for rep in range(3):
        eng.launch(12345, rep * 16384, 16384)
    eng.sync(); eng.timing_reset()
    for rep in range(10):
        torch.cuda.synchronize()
        if variant >= 0:
            N.check(N.lib().eg_debug_occupy(eng.h, variant, 3_000_000))      # ~1.3 ms at 2.4 GHz
        eng.launch(12345, rep * 16384, 16384)
    eng.sync()
    ms, n = eng.timing_read()
    print(f"{label:30s} lean grid {ms / n:.3f} ms", flush=True)
