"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into <out_dir>/<tag>_pmc_hbm_traffic.json.

  python scripts/pmc_summarize.py <tag> <out_dir> <workload>:<fetch_dir>:<write_dir> [...]

Each dir is the -d directory of one counter pass of `python3 bench.py --steps 4 --warmup 20 --no-cpu-baseline --batches-per-step 2 ...` of that
workload ("16384r0.1" = BASELINE configs[2], the default; "1024" = configs[1]).  A batch with replay episodes is two
k_rollout grids (the heavy-capable variant `k_rollout<.., true>` for the replays, the lean one for the rest); the counters
of a batch are the sum of its grids.  Only the dispatches of the TIMED region (the last steps x batches_per_step = 8 per
variant) are averaged, after 40 warm-up batches: the loop's cost per batch changes while the first best strategies are found.
HBM bytes per batch = 2 * FETCH_SIZE * 1024 (gfx950: FETCH_SIZE counts half of a wide stream, MI355X_MICROARCH.md HBM
section; an upper bound for narrow gathers) + WRITE_SIZE * 1024.  Counter collection serialises the dispatches, so
`kernel_ns` (heavy + lean, one after the other) is longer than the overlapped launch of an un-profiled run.
"""
import csv, glob, json, os, sys

TIMED = 8


def variant_of(name):
    if "k_rollout" not in name: return None
    return "heavy" if ", true>" in name else "lean"


def rows(d, counter):
    out = {"heavy": [], "lean": []}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            v = variant_of(r["Kernel_Name"])
            if v and r["Counter_Name"] == counter:
                out[v].append(dict(dispatch=int(r["Dispatch_Id"]), grid=int(r["Grid_Size"]), value_kb=float(r["Counter_Value"]),
                                   dur_ns=int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), vgpr=int(r["VGPR_Count"]),
                                   lds=int(r["LDS_Block_Size"])))
    for v in out:
        rs = sorted(out[v], key=lambda r: r["dispatch"])
        grid = rs[-1]["grid"] if rs else 0
        out[v] = [r for r in rs if r["grid"] == grid][-TIMED:]      # (the seeding episode of configs[2] is a grid of its own)
    return out


def avg(rs, key):
    return sum(r[key] for r in rs) / len(rs) if rs else 0.0


def main():
    tag, out_dir, specs = sys.argv[1], sys.argv[2], sys.argv[3:]
    doc = {"note": __doc__.strip().split("\n\n", 1)[1], "configs": {}}
    for spec in specs:
        wl, fd, wd = spec.split(":")
        f, w = rows(fd, "FETCH_SIZE"), rows(wd, "WRITE_SIZE")
        if not f["lean"] and not f["heavy"]:
            raise SystemExit(f"no k_rollout rows under {fd}")
        fa = avg(f["heavy"], "value_kb") + avg(f["lean"], "value_kb"); wa = avg(w["heavy"], "value_kb") + avg(w["lean"], "value_kb")
        doc["configs"][wl] = dict(workload=wl, launches=max(len(f["lean"]), len(f["heavy"])), fetch_size_kb_avg=fa, write_size_kb_avg=wa,
                                  hbm_bytes_per_launch=2 * fa * 1024 + wa * 1024,
                                  kernel_ns=avg(f["heavy"], "dur_ns") + avg(f["lean"], "dur_ns"),
                                  kernel_ns_heavy=avg(f["heavy"], "dur_ns"), kernel_ns_lean=avg(f["lean"], "dur_ns"),
                                  fetch_rows=f, write_rows=w)
    path = os.path.join(out_dir, f"{tag}_pmc_hbm_traffic.json")
    json.dump(doc, open(path, "w"), indent=1)
    print(path, {k: round(v["hbm_bytes_per_launch"]) for k, v in doc["configs"].items()})


if __name__ == "__main__":
    main()
