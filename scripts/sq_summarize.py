"""Summarise rocprofv3 SQ counter passes of scripts/pmc_launch.py into profiles/<tag>_sq_counters.json.
   python scripts/sq_summarize.py <tag> <episodes>:<dir> [...]"""
import csv, glob, json, os, sys
tag, specs = sys.argv[1], sys.argv[2:]
doc = {"note": "rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU "
               "SQ_WAIT_ANY -- python3 scripts/pmc_launch.py B 3 (plain k_rollout launches, fresh ActionWeights); per-launch averages. "
               "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are in quad-cycles. B=1024 runs k_rollout<1> (episode wave + helper wave), "
               "B=16384 k_rollout<0>.", "runs": {}}
for spec in specs:
    eps, d = spec.split(":")
    acc, n = {}, {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if "k_rollout" not in r["Kernel_Name"]: continue
            acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"]); n[r["Counter_Name"]] = n.get(r["Counter_Name"], 0) + 1
            acc["duration_ns"] = acc.get("duration_ns", 0.0) + int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); n["duration_ns"] = n.get("duration_ns", 0) + 1
    run = {k: acc[k] / n[k] for k in acc}
    run["episodes_per_launch"] = int(eps)
    doc["runs"][eps] = run
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", f"{tag}_sq_counters.json")
json.dump(doc, open(out, "w"), indent=1)
for eps, r in doc["runs"].items():
    e = r["episodes_per_launch"]
    print(eps, {k: round(v / e) for k, v in r.items() if k.startswith("SQ_INSTS")}, "VALU-active share of wave cycles", round(r.get("SQ_ACTIVE_INST_VALU", 0) / max(r.get("SQ_WAVE_CYCLES", 1), 1), 3))
