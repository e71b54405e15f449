import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from eirgrid_amd import synthetic_world
from eirgrid_amd.engine import ActionWeights, Engine
from eirgrid_amd.parallel import BatchTrainer
torch.cuda.set_device(0)
eng = Engine(synthetic_world()); pol = ActionWeights()
tr = BatchTrainer(eng, pol, 1024, 12345, replay_fraction=0.0)
for k in range(6): tr.step()
tr.sync()
print("best list lengths per year:", [len(l) for l in pol.lists(0)], "deficit:", [len(l) for l in pol.lists(1)])
for frac in (0.0, 0.25, 1.0):
    mask = (np.random.default_rng(0).uniform(size=1024) < frac).astype(np.uint8)
    eng.upload_snapshot(pol); eng.timing_reset()
    eng.launch(777, 0, 1024, mask); eng.sync()
    ms, n = eng.timing_read(); res = eng.fetch(1024)
    r = mask == 1
    print(f"replay fraction {frac}: kernel {ms/n:.3f} ms; status counts {np.bincount(res.status, minlength=4)}; gens/ep replay {res.n_gens[r].mean() if r.any() else 0:.1f} others {res.n_gens[~r].mean() if (~r).any() else 0:.1f}; "
          f"acts/ep replay {res.n_act.sum(1)[r].mean() if r.any() else 0:.1f}; run len replay {res.n_run.sum(1)[r].mean() if r.any() else 0:.1f} others {res.n_run.sum(1)[~r].mean() if (~r).any() else 0:.1f}; draws replay {res.n_draws[r].mean() if r.any() else 0:.1f}")
