import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eirgrid_amd import synthetic_world
from eirgrid_amd.engine import ActionWeights, Engine
w = synthetic_world(); eng = Engine(w); pol = ActionWeights()
eng.upload_snapshot(pol)
for B in (64, 1024, 4096, 16384):
    eng.launch(12345, 0, B); eng.sync(); eng.timing_reset()
    t = time.time()
    for k in range(3): eng.launch(12345, k * B, B)
    eng.sync(); dt = time.time() - t
    ms, n = eng.timing_read()
    res = eng.fetch(B)
    print(f"B={B} wall {dt/3*1e3:.2f} ms/batch kernel {ms/n:.2f} ms -> {B/(ms/n)*1e3:.0f} eps/s ; status!=0: {(res.status!=0).sum()} gens mean {res.n_gens.mean():.1f} bytes/ep {res.bytes_moved.mean():.0f}", flush=True)
