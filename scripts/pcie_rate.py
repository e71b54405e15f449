"""What the host-buffer form of the boundary costs (eg_rollout_batch: snapshot upload, rollout, EVERY output copied to caller memory),
next to the device-resident loop bench.py times.  The caller keeps its buffers, as a host program in a loop would.
   python scripts/pcie_rate.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eirgrid_amd import synthetic_world
from eirgrid_amd.engine import ActionWeights, BatchResult, Engine
eng = Engine(synthetic_world()); pol = ActionWeights()
for n in (1024, 16384):
    out = BatchResult.alloc(n)
    for k in range(2): eng.rollout_batch(pol, 1 + k, n, out=out)
    t0 = time.perf_counter()
    for k in range(5): res = eng.rollout_batch(pol, 100 + k, n, out=out)
    dt = (time.perf_counter() - t0) / 5
    t0 = time.perf_counter()
    for k in range(5): res = eng.rollout_batch(pol, 100 + k, n, out=out, write_yearly=False)
    dt2 = (time.perf_counter() - t0) / 5
    print(f"eg_rollout_batch, every output fetched into the caller's buffers: {n} episodes {dt * 1e3:.2f} ms per call ({n / dt / 1e6:.2f} M eps/s); "
          f"without the yearly rows {dt2 * 1e3:.2f} ms; most generators in an episode {res.n_gens.max()}")
