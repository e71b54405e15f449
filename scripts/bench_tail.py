"""Diagnostic (stamps build): per-step kernel time vs the distribution of episode cycle counts along the bench trajectory.
   EIRGRID_LIB=eirgrid_amd/libeirgrid_hip_stamps.so python scripts/bench_tail.py [B] [steps]"""
import os, sys
os.environ["EIRGRID_FETCH_FULL"] = "1"      # (the stamps sit at the end of act_log: whole rows, please)
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
for k in range(steps):
    eng.timing_reset(); tr.step(); eng.sync()
    ms, n = eng.timing_read()
    res = eng.fetch(B)
    st = res.act_log[:, -256:].copy().view(np.uint64).astype(np.float64)
    tot = st[:, 7]
    srt = np.sort(tot)
    hw = res.act_log[:, -256:].copy().view(np.uint64)[:, 31]
    hwid = (hw & 0xFFFFFFFF).astype(np.int64); xcc = ((hw >> 32) & 0xF).astype(np.int64)
    cu = (hwid >> 8) & 0xF; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 0x7
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    counts = np.bincount(key.astype(np.int64))
    counts = counts[counts > 0]
    simd = (hwid >> 4) & 3
    per_simd = np.bincount((key * 4 + simd).astype(np.int64)); per_simd = per_simd[per_simd > 0]
    print(f"step {k:2d} episode waves per SIMD: " + " ".join(f"{c}:{int((per_simd == c).sum())}" for c in sorted(set(per_simd.tolist()))) + f"  (SIMDs hosting one: {len(per_simd)} of {4 * len(counts)})")
    print(f"step {k:2d} CUs used {len(counts)}  workgroups per CU: " + " ".join(f"{c}:{int((counts == c).sum())}" for c in sorted(set(counts.tolist()))))
    print(f"step {k:2d} kernel {ms / max(n, 1):.3f} ms  episode cycles mean {tot.mean():8.0f} p50 {srt[B // 2]:8.0f} p99 {srt[int(B * 0.99)]:8.0f} max {srt[-1]:8.0f}  "
          f"max/2.4GHz {srt[-1] / 2.4e6:.3f} ms  placement of slowest {st[int(np.argmax(tot)), 1]:8.0f} gens {res.n_gens[int(np.argmax(tot))]}", flush=True)

names = {0: "year-start aggregates", 1: "placement search", 2: "sampling", 3: "deficit evaluate + nudges", 4: "yearly metrics", 5: "policy block",
         12: "apply gen", 14: "totals", 15: "initial state", 16: "episode start", 17: "glue year back edge", 18: "glue before aggregates",
         19: "glue loop top", 20: "glue sampled->search", 21: "glue search->bookkeeping", 22: "glue apply->evaluate", 23: "phase-1 logs", 24: "glue before n_add", 25: "glue exit->metrics"}
k = int(np.argmax(tot))
print(f"slowest episode of the last step: {tot[k]:.0f} cycles, gens {res.n_gens[k]}, searches {st[k, 11]:.0f}, chunks {st[k, 8]:.0f}, gen loop {st[k, 9]:.0f}, reduce+merge {st[k, 10]:.0f}, spin {st[k, 6]:.0f}")
print("  " + "  ".join(f"{n} {100 * st[k, i] / tot[k]:.1f}%" for i, n in names.items()))
print("mean episode: " + "  ".join(f"{n} {100 * st[:, i].mean() / tot.mean():.1f}%" for i, n in names.items()))
