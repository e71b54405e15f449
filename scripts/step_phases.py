"""Diagnostic: wall time of each phase of BatchTrainer.step() (1 GPU): python scripts/step_phases.py [B] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from eirgrid_amd import synthetic_world
from eirgrid_amd.engine import ActionWeights, Engine, apply_reduced
from eirgrid_amd import parallel as P
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
torch.cuda.set_device(0)
eng = Engine(synthetic_world()); pol = ActionWeights()
tr = P.BatchTrainer(eng, pol, B, 12345)
for _ in range(5): tr.step()
acc = dict(upload=0.0, launch=0.0, wait_kernel=0.0, d2h_parse=0.0, apply=0.0)
t_all = time.perf_counter()
for k in range(steps):
    first = tr.step_index * B
    t0 = time.perf_counter(); eng.upload_snapshot(pol)
    t1 = time.perf_counter(); eng.launch_update(12345, first, B, tr.packet.data_ptr(), None)
    t2 = time.perf_counter(); torch.cuda.synchronize()
    t3 = time.perf_counter(); stats, cand = P.exchange_packet(tr.packet, None)
    t4 = time.perf_counter(); apply_reduced(pol, stats, cand, noise_seed=12345 + tr.step_index)
    t5 = time.perf_counter(); tr.step_index += 1
    for n, d in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)): acc[n] += d
tot = time.perf_counter() - t_all
print(f"B={B}: {tot / steps * 1e3:.3f} ms/step;  " + "  ".join(f"{n} {v / steps * 1e6:.0f} us" for n, v in acc.items()))
ms, n = eng.timing_read(); print(f"kernel avg {ms / max(n, 1):.3f} ms over {n} launches; stall now {pol.get('iterations_without_improvement')}")
