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
st = res.act_log[:, -128:].copy().view(np.uint64).astype(np.float64)   # [B, 16]
names = ["year-start aggregates", "placement search", "sampling (rng + walks)", "deficit evaluate + nudges", "yearly metrics + stores", "-", "-", "episode total"]
tot = st[:, 7].mean()
print(f"B={B}  mean episode cycles {tot:.0f}  (min {st[:,7].min():.0f} max {st[:,7].max():.0f})  gens/ep {res.n_gens.mean():.1f}")
for i, n in enumerate(names[:5]):
    print(f"  {n:28s} {st[:, i].mean():10.0f} cycles  {100 * st[:, i].mean() / tot:5.1f} %")
print(f"  {'policy rows -> LDS':28s} {st[:, 5].mean():10.0f} cycles  {100 * st[:, 5].mean() / tot:5.1f} %")
acc = st[:, :6].sum(1).mean()
print(f"  {'unaccounted':28s} {tot - acc:10.0f} cycles  {100 * (tot - acc) / tot:5.1f} %")
print(f"  placement detail: searches/ep {st[:, 11].mean():.1f}  chunks/search {st[:, 8].sum() / st[:, 11].sum():.2f}  "
      f"generator loop {st[:, 9].mean():.0f} cyc/ep ({st[:, 9].sum() / st[:, 8].sum():.0f}/chunk)  reduce+select {st[:, 10].mean():.0f} cyc/ep ({st[:, 10].sum() / st[:, 8].sum():.0f}/chunk)  "
      f"rest (loads, setup, exit test) {(st[:, 1] - st[:, 9] - st[:, 10]).mean():.0f} cyc/ep")
k = int(np.argmax(st[:, 7]))
print(f"  slowest episode {k}: total {st[k, 7]:.0f}, placement {st[k, 1]:.0f}, searches {st[k, 11]:.0f}, chunks {st[k, 8]:.0f}, gens {res.n_gens[k]}, year-start {st[k, 0]:.0f}, sampling {st[k, 2]:.0f}")
