"""Summarise rocprofv3 SQ counter passes of bench.py into <out_dir>/<tag>_sq_counters.json.
   python scripts/sq_summarize.py <tag> <out_dir> <workload>:<dir> [...]
Per-batch averages over the k_rollout dispatches of the timed region (the last 8 of each variant; a batch with replay
episodes is the two replay-variant grids — one of them returns at once — plus a lean one: their counters add up)."""
import csv, glob, json, os, sys
tag, out_dir, specs = sys.argv[1], sys.argv[2], sys.argv[3:]
TIMED = 8
doc = {"note": "rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU "
               "SQ_WAIT_ANY -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --batches-per-step 2 [...] (every batch from the same policy: bench.py's "
               "pinned workloads); per-batch averages over the k_rollout dispatches of the timed region, heavy-variant grid + lean grid. "
               "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are in quad-cycles summed over the SIMDs: valu_busy = SQ_ACTIVE_INST_VALU * 4 / "
               "(1024 SIMDs * duration * 2.4 GHz), duration = the grids one after the other (counter collection serialises dispatches; "
               "the un-profiled launch overlaps them). 1024 episodes per launch run k_rollout<1,.> (episode wave + helper wave), 16384 k_rollout<0,.>.",
       "runs": {}}
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eirgrid_amd import _native as N
doc["build_hash"] = N.lib().eg_build_hash().decode()      # the library the counters were taken on (bench.py checks it)
for spec in specs:
    wl, d = spec.split(":")
    per = {"heavy": {}, "short": {}, "lean": {}, "coop": {}, "books": {}, "bcast": {}, "solo": {}}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            nm = r["Kernel_Name"]
            hoist = [v for key, v in (("k_replay_coop", "coop"), ("k_replay_books", "books"), ("k_replay_broadcast", "bcast"), ("k_replay_solo", "solo")) if key in nm]
            if "k_rollout" not in nm and not hoist: continue
            v = hoist[0] if hoist else ("heavy" if ", 2>" in nm else ("short" if ", 1>" in nm else "lean"))      # k_rollout<helpers, kind>; the replay hoist's kernels
            e = per[v].setdefault(int(r["Dispatch_Id"]), {"grid": int(r["Grid_Size"])})
            e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            e["duration_ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    run = {}
    for v in ("heavy", "short", "lean", "coop", "books", "bcast", "solo"):
        ids = sorted(per[v])
        if not ids: continue
        grid = per[v][ids[-1]]["grid"]
        ids = [i for i in ids if per[v][i]["grid"] == grid][-TIMED:]
        part = {k: sum(per[v][i][k] for i in ids) / len(ids) for k in per[v][ids[0]] if k != "grid"}
        run[v] = part
        for k, x in part.items():
            if k == "duration_ns" and v in ("coop", "books", "bcast"): continue      # (one workgroup / tiny grids beside the lean grid: not a share of the chip's time)
            run[k] = run.get(k, 0.0) + x
    eps = int(wl.split("r")[0])      # ("16384r0.1", "16384r0.1grown", "16384r0.1grownhoist", "1024")
    run["episodes_per_launch"] = eps; run["clock_hz"] = 2.4e9
    run["valu_busy"] = run.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0 / (1024 * run["duration_ns"] * 1e-9 * 2.4e9)
    for v in ("heavy", "short", "lean", "coop", "books", "bcast", "solo"):
        if v in run: run[v]["valu_busy"] = run[v].get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0 / (1024 * run[v]["duration_ns"] * 1e-9 * 2.4e9)
    doc["runs"][wl] = run
out = os.path.join(out_dir, f"{tag}_sq_counters.json")
json.dump(doc, open(out, "w"), indent=1)
for wl, r in doc["runs"].items():
    e = r["episodes_per_launch"]
    print(wl, {k: round(v / e) for k, v in r.items() if k.startswith("SQ_INSTS")}, "valu_busy", round(r["valu_busy"], 3), "kernel ms (serialised)", round(r["duration_ns"] * 1e-6, 3),
          {v: (round(r[v]["valu_busy"], 3), round(r[v]["duration_ns"] * 1e-6, 3)) for v in ("heavy", "solo", "short", "lean") if v in r})
