import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from eirgrid_amd import synthetic_world, _native as N
from eirgrid_amd.engine import ActionWeights, Engine
eng = Engine(synthetic_world())
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
def timeit(pol, fused, label):
    eng.upload_snapshot(pol)
    packet = torch.zeros(N.PACKET_BYTES, dtype=torch.uint8, device="cuda")
    for rep in range(2):
        eng.sync(); eng.timing_reset()
        for k in range(5):
            if fused: eng.launch_update(12345, k * B, B, packet.data_ptr())
            else: eng.launch(12345, k * B, B)
        eng.sync(); ms, n = eng.timing_read()
    res = eng.fetch(B)
    print(f"{label:50s} kernel {ms/n:.3f} ms  gens/ep {res.n_gens.mean():.1f} draws/ep {res.n_draws.mean():.1f} acts/ep {res.n_act.sum(1).mean():.1f}", flush=True)
pol = ActionWeights(); timeit(pol, False, "fresh, plain")
timeit(pol, True, "fresh, fused stats (no best)")
first = eng.run_iteration(0, pol, False, 12345)
pol.apply_episode(first.metrics[0], first.n_run[0], first.run_log[0], first.n_def[0], first.def_log[0])
timeit(pol, False, "has best, plain"); timeit(pol, True, "has best, fused stats")
for stall in (200, 600, 2000, 20000):
    pol.set("iterations_without_improvement", stall); timeit(pol, False, f"stall {stall}, plain")
pol.set("iterations_without_improvement", 2000); timeit(pol, True, "stall 2000, fused stats")
