"""Diagnostic: where does an episode spend its cycles?  Needs the -DEG_STAMPS build (make -C eirgrid_amd/csrc stamps):
   EIRGRID_LIB=eirgrid_amd/libeirgrid_hip_stamps.so python scripts/stamps.py
Shares only; never quote this build's run time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eirgrid_amd import synthetic_world
from eirgrid_amd.engine import ActionWeights, Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
eng = Engine(synthetic_world()); pol = ActionWeights()
eng.upload_snapshot(pol); eng.launch(12345, 0, B); eng.sync()
res = eng.fetch(B)
st = res.act_log[:, -64:].copy().view(np.uint64).astype(np.float64)   # [B, 8]
names = ["year-start aggregates", "placement search", "sampling (rng + walks)", "deficit evaluate + nudges", "yearly metrics + stores", "-", "-", "episode total"]
tot = st[:, 7].mean()
print(f"B={B}  mean episode cycles {tot:.0f}  (min {st[:,7].min():.0f} max {st[:,7].max():.0f})  gens/ep {res.n_gens.mean():.1f}")
for i, n in enumerate(names[:5]):
    print(f"  {n:28s} {st[:, i].mean():10.0f} cycles  {100 * st[:, i].mean() / tot:5.1f} %")
print(f"  {'unaccounted':28s} {tot - st[:, :5].sum(1).mean():10.0f} cycles  {100 * (tot - st[:, :5].sum(1).mean()) / tot:5.1f} %")
