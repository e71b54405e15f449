"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into <out_dir>/<tag>_pmc_hbm_traffic.json.

  python scripts/pmc_summarize.py <tag> <out_dir> <workload>:<fetch_dir>:<write_dir> [...]

Each dir is the -d directory of one counter pass of `python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline ...` of that
workload ("16384r0.1" = BASELINE configs[2], the default; "1024" = configs[1]).  Only the k_rollout dispatches of the
TIMED region (the last steps x batches_per_step = 8 of them) are averaged: the loop's cost per batch changes while the
first best strategies are found.
HBM bytes per k_rollout launch = 2 * FETCH_SIZE * 1024 (gfx950: FETCH_SIZE counts half of a wide stream,
MI355X_MICROARCH.md HBM section; an upper bound for narrow gathers) + WRITE_SIZE * 1024.
"""
import csv, glob, json, os, sys

TIMED = 8


def rows(d, counter):
    out = []
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter and ("k_rollout(" in r["Kernel_Name"] or "k_rollout<" in r["Kernel_Name"]):
                out.append(dict(dispatch=int(r["Dispatch_Id"]), grid=int(r["Grid_Size"]), value_kb=float(r["Counter_Value"]),
                                dur_ns=int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), vgpr=int(r["VGPR_Count"]),
                                lds=int(r["LDS_Block_Size"])))
    out = sorted(out, key=lambda r: r["dispatch"])
    grid = out[-1]["grid"] if out else 0
    return [r for r in out if r["grid"] == grid][-TIMED:]      # (the seeding episode of configs[2] is a grid of its own)


def main():
    tag, out_dir, specs = sys.argv[1], sys.argv[2], sys.argv[3:]
    doc = {"note": __doc__.strip().split("\n\n", 1)[1], "configs": {}}
    for spec in specs:
        wl, fd, wd = spec.split(":")
        f, w = rows(fd, "FETCH_SIZE"), rows(wd, "WRITE_SIZE")
        if not f or not w:
            raise SystemExit(f"no k_rollout rows under {fd} / {wd}")
        fa, wa = sum(r["value_kb"] for r in f) / len(f), sum(r["value_kb"] for r in w) / len(w)
        doc["configs"][wl] = dict(workload=wl, launches=len(f), fetch_size_kb_avg=fa, write_size_kb_avg=wa,
                                  hbm_bytes_per_launch=2 * fa * 1024 + wa * 1024, fetch_rows=f, write_rows=w)
    path = os.path.join(out_dir, f"{tag}_pmc_hbm_traffic.json")
    json.dump(doc, open(path, "w"), indent=1)
    print(path, {k: round(v["hbm_bytes_per_launch"]) for k, v in doc["configs"].items()})


if __name__ == "__main__":
    main()
