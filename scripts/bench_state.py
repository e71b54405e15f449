"""Diagnostic: what do the episodes of the bench's steady state look like?  python scripts/bench_state.py [B] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from eirgrid_amd import synthetic_world
from eirgrid_amd.engine import ActionWeights, Engine
from eirgrid_amd.parallel import BatchTrainer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 23
torch.cuda.set_device(0)
eng = Engine(synthetic_world()); pol = ActionWeights()
tr = BatchTrainer(eng, pol, B, 12345)
full = len(sys.argv) > 3
for k in range(steps):
    tr.sync(); stall_before = pol.get("iterations_without_improvement")
    eng.timing_reset(); tr.step(); eng.sync()
    ms, n = eng.timing_read()
    if not full:
        print(f"step {k:2d} kernel {ms / max(n, 1):.3f} ms  stall before {stall_before:.0f}", flush=True); continue
    res = eng.fetch(B)
    print(f"step {k:2d} stall before {stall_before:.0f} kernel {ms / max(n, 1):.3f} ms  gens/ep mean {res.n_gens.mean():5.1f} max {res.n_gens.max():3d}  offsets/ep {res.n_offsets.mean():4.1f}  "
          f"acts/ep {res.n_act.sum(1).mean():5.1f}  deficit acts/ep {res.n_def.sum(1).mean():5.1f}  draws/ep {res.n_draws.mean():6.1f}  ok {int((res.status == 0).sum())}", flush=True)
tr.sync()
print("stall", pol.get("iterations_without_improvement"), "improvements", tr.improvements)
