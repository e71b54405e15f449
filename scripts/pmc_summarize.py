"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into <out_dir>/<tag>_pmc_hbm_traffic.json.

  python scripts/pmc_summarize.py <tag> <out_dir> <workload>:<fetch_dir>:<write_dir> [...]

Each dir is the -d directory of one counter pass of `python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --batches-per-step 2 ...` of that
workload ("16384r0.1" = BASELINE configs[2], the default; "16384r0.1grown" = the same from the grown-replay state, --grown;
"16384r0.1grownhoist" = the grown state with the replay hoist on, --replay-hoist: k_replay_coop / _books / _broadcast are counted too;
"1024" = configs[1]).  A batch with replay episodes is two
k_rollout grids for the replays (`k_rollout<.., 2>`, the heavy-capable variant, and `k_rollout<.., 1>`, the short-replay one:
whichever the length of the best list does not call for returns at once) and the lean one (`k_rollout<.., 0>`) for the rest; the
counters of a batch are the sum of its grids.  Only the dispatches of the TIMED region (the last steps x batches_per_step = 8 per
variant) are averaged (every batch starts from the same policy, so they are all alike).
HBM bytes per batch = 2 * FETCH_SIZE * 1024 (gfx950: FETCH_SIZE counts half of a wide stream, MI355X_MICROARCH.md HBM
section; an upper bound for narrow gathers) + WRITE_SIZE * 1024.  Counter collection serialises the dispatches, so
`kernel_ns` (heavy + lean, one after the other) is longer than the overlapped launch of an un-profiled run.
"""
import csv, glob, json, os, sys

TIMED = 8


VARIANTS = ("heavy", "short", "lean", "coop", "books", "bcast", "solo")


def variant_of(name):      # k_rollout<helpers, kind>: kind 2 = long-replay (heavy-capable) variant, 1 = short-replay, 0 = lean
    for key, v in (("k_replay_coop", "coop"), ("k_replay_books", "books"), ("k_replay_broadcast", "bcast"),      # the replay hoist's kernels
                   ("k_replay_solo", "solo")):      # the per-episode replay kernel (k_rollout<.., 2> behind it then has nothing to do)
        if key in name: return v
    if "k_rollout" not in name: return None
    return "heavy" if ", 2>" in name else ("short" if ", 1>" in name else "lean")


def rows(d, counter):
    out = {v: [] for v in VARIANTS}
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
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from eirgrid_amd import _native as N
    # the library the counters were taken on: bench.py only quotes a profile whose hash is the loaded library's
    doc = {"note": __doc__.strip().split("\n\n", 1)[1], "build_hash": N.lib().eg_build_hash().decode(), "configs": {}}
    for spec in specs:
        wl, fd, wd = spec.split(":")
        f, w = rows(fd, "FETCH_SIZE"), rows(wd, "WRITE_SIZE")
        if not any(f[v] for v in ("lean", "heavy", "short")):
            raise SystemExit(f"no k_rollout rows under {fd}")
        fa = sum(avg(f[v], "value_kb") for v in f); wa = sum(avg(w[v], "value_kb") for v in w)
        doc["configs"][wl] = dict(workload=wl, launches=max(len(f[v]) for v in f), fetch_size_kb_avg=fa, write_size_kb_avg=wa,
                                  hbm_bytes_per_launch=2 * fa * 1024 + wa * 1024,
                                  kernel_ns=sum(avg(f[v], "dur_ns") for v in ("heavy", "short", "lean", "solo")),
                                  kernel_ns_heavy=avg(f["heavy"], "dur_ns") + avg(f["solo"], "dur_ns"), kernel_ns_short=avg(f["short"], "dur_ns"), kernel_ns_lean=avg(f["lean"], "dur_ns"),
                                  kernel_ns_hoist=sum(avg(f[v], "dur_ns") for v in ("coop", "books", "bcast")),
                                  fetch_rows=f, write_rows=w)
    path = os.path.join(out_dir, f"{tag}_pmc_hbm_traffic.json")
    json.dump(doc, open(path, "w"), indent=1)
    print(path, {k: round(v["hbm_bytes_per_launch"]) for k, v in doc["configs"].items()})


if __name__ == "__main__":
    main()
