"""Replay runaway in numbers: the device-resident training loop with a replay fraction, step by step.
   python scripts/replay_study.py [episodes_per_step] [replay_fraction] [steps] [out.json]
Per step: wall ms (synchronised for the measurement), k_rollout ms (HIP events), episodes by status, generators per episode
(seeded / replay), best-list length, stall counter, improvements."""
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from eirgrid_amd import synthetic_world, _native as N
from eirgrid_amd.engine import ActionWeights, Engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
out_path = sys.argv[4] if len(sys.argv) > 4 else None
period = max(1, int(round(1.0 / frac))) if frac > 0 else 0
torch.cuda.set_device(0)
eng = Engine(synthetic_world()); pol = ActionWeights()
eng.push(pol)
rows = []
t_all = time.perf_counter()
for step in range(steps):
    first = step * B
    eng.sync(); eng.timing_reset()
    t0 = time.perf_counter()
    eng.device_step(12345, first, B, period, 12345 + step)
    eng.sync()
    wall = (time.perf_counter() - t0) * 1e3
    kms, _ = eng.timing_read()
    st = np.zeros(B, np.int32); ng = np.zeros(B, np.int32); nr = np.zeros((B, 26), np.int32)
    import ctypes as C
    o = N.EgEpisodeOut(); o.status = st.ctypes.data_as(C.POINTER(C.c_int32)); o.n_gens = ng.ctypes.data_as(C.POINTER(C.c_int32))
    o.n_run = nr.ctypes.data_as(C.POINTER(C.c_int32))
    rc = N.lib().eg_fetch(eng.h, C.byref(o))
    eng.pull(pol)
    has = pol.get("has_best_actions") == 1
    idx = np.arange(first, first + B)
    rep = (idx % period == 0) if (period and step > 0) else np.zeros(B, bool)      # (replays need a best strategy: from step 1)
    ok = st == 0
    row = dict(step=step, wall_ms=round(wall, 3), rollout_ms=round(kms, 3), ok=int(ok.sum()), overflow=int((st == -1).sum()),
               other=int(((st != 0) & (st != -1)).sum()), fetch_rc=int(rc),
               gens_seeded=float(ng[ok & ~rep].mean()) if (ok & ~rep).any() else None,
               gens_replay=float(ng[ok & rep].mean()) if (ok & rep).any() else None, gens_max=int(ng.max()),
               run_len_replay=float(nr.sum(1)[ok & rep].mean()) if (ok & rep).any() else None,
               best_len=int(sum(len(l) for l in pol.lists(0))), best_def_len=int(sum(len(l) for l in pol.lists(1))),
               stall=int(pol.get("iterations_without_improvement")), improvements=int(pol.get("improvement_history_len")))
    rows.append(row)
    print(json.dumps(row), flush=True)
total = time.perf_counter() - t_all
summary = dict(episodes_per_step=B, replay_fraction=frac, steps=steps, total_s=round(total, 2),
               mean_rollout_ms=float(np.mean([r["rollout_ms"] for r in rows])), max_rollout_ms=float(max(r["rollout_ms"] for r in rows)),
               total_overflow=int(sum(r["overflow"] for r in rows)), total_ok=int(sum(r["ok"] for r in rows)))
print("SUMMARY", json.dumps(summary))
if out_path:
    json.dump(dict(summary=summary, steps=rows), open(out_path, "w"), indent=1)
