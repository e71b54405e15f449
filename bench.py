#!/usr/bin/env python3
"""Headline benchmark: full 26-year episodes/s of the rollout hot path on N MI355X (BASELINE.json `metric`).

A step = one pass of the hot path over one batch, with the policy resident on the device: rollout of `--episodes`
episodes per GPU against the current policy (k_rollout, batch-update statistics in its epilogue), best-episode pick
(k_pick_best), when N > 1 the per-update exchange (ONE RCCL all-gather of every rank's 32 KB update packet), the batch
update itself (k_apply_update) and the stalled-sampler tables (k_stalled_tables) — stream-ordered launches, no host
synchronisation inside the timed region.  Inputs (world tables, policy) are resident in HBM when the timed region starts.
Weak scaling: every rank runs `--episodes` episodes per step with streams keyed by global episode index.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P bench.py --gpus 8 ...
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md (≈6.3 TB/s achievable)


def host_cores() -> int:
    """Threads this process may really use: the cgroup CPU quota if there is one, else the affinity mask (capped at 64)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


def pmc_traffic(episodes: int):
    """HBM bytes per k_rollout launch from the newest committed PMC profile for this batch size (profiles/*pmc_hbm_traffic.json:
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of this same command, gfx950 correction applied) or None.
    Counters cannot be collected from inside the process, so the number comes from the committed profile of the same build."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_hbm_traffic.json"))):
        try:
            cfg = json.load(open(path))["configs"].get(str(episodes))
        except (OSError, ValueError, KeyError):
            continue
        if cfg:
            best = float(cfg["hbm_bytes_per_launch"])
    return best


def cpu_baseline(world, seconds_budget: float = 20.0):
    """The CPU oracle (C restatement of the reference algorithm — NOT the Rust/rayon binary, which cannot be built here)
    timed on this box's host cores: literal mode (same work as the reference: 100x100 candidate search with sqrt+div per
    factor, G x S opinion loops) and tabled mode (same results from the host tables).  One episode per thread."""
    from concurrent.futures import ThreadPoolExecutor
    from eirgrid_amd.engine import HostTables
    from oracle import api as O
    O.build()
    cores = host_cores()
    ow = O.OracleWorld(world)
    t0 = time.perf_counter(); O.run_episode(ow, O.OracleWeights(), 1); one = time.perf_counter() - t0
    n_lit = max(cores, int(seconds_budget * 0.75 / max(one, 1e-3)))     # ≈ 15 s of CPU work
    n_lit = (n_lit // cores) * cores

    def lit(e):
        return O.run_episode(ow, O.OracleWeights(), 12345 + e)[0]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        st = list(ex.map(lit, range(n_lit)))
    t_lit = time.perf_counter() - t0
    assert all(s == 0 for s in st)
    tb = O.OracleTables(HostTables(world), len(world.existing_x))
    n_tab = 2000 * cores

    def tab(chunk):
        for e in chunk:
            O.run_episode_tabled(tb, O.OracleWeights(), 12345 + e)
    chunks = [range(i, n_tab, cores) for i in range(cores)]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(tab, chunks))
    t_tab = time.perf_counter() - t0
    return {"value": n_lit / t_lit, "unit": "episodes/s", "cores": cores, "kind": "port",
            "sample": f"{n_lit} literal-mode episodes (seeds 12345+e, same synthetic world), one per thread on {cores} threads, "
                      f"{t_lit:.1f} s wall; tabled mode: {n_tab} episodes in {t_tab:.1f} s",
            "tabled_value": n_tab / t_tab,
            "note": "C restatement of the reference algorithm (oracle/eg_oracle.c), not the Rust/rayon binary"}


def measured_peak():
    """Triad bandwidth measured on an MI355X of this pool (scripts/hbm_probe.py -> profiles/*hbm_probe.json), reported beside
    the datasheet peak that `frac` is priced against; None when no probe result is committed."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*hbm_probe.json"))):
        try:
            best = float(json.load(open(path))["triad_GBps"])
        except Exception:
            pass
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--episodes", type=int, default=1024, help="episodes per GPU per step (BASELINE configs[1] = 1024)")
    ap.add_argument("--seed", type=int, default=12345)
    ap.add_argument("--replay-fraction", type=float, default=0.0, help="configs[2]: 0.1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only for rehearsal)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--force-collectives", action="store_true",
                    help="rehearsal: initialise torch.distributed and issue the per-update collectives even with one rank")
    ap.add_argument("--no-yearly", action="store_true", help="skip the 26x21 yearly rows (the reference always produces them)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:   # convenience: relaunch under torchrun as a child process
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", "29511", os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    import numpy as np
    import torch
    import torch.distributed as dist
    from eirgrid_amd import synthetic_world
    from eirgrid_amd.engine import ActionWeights, Engine
    from eirgrid_amd.parallel import BatchTrainer

    rank = int(os.environ.get("RANK", "0")); local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the rollout engine has no CPU path")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    use_dist = world_size > 1 or args.force_collectives
    # RCCL prints a version banner on stdout when its first communicator comes up (at the first collective).  stdout is
    # reserved for the one JSON line: until the warm-up is over, file descriptor 1 points at stderr.
    saved_stdout = None
    if use_dist:
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")      # only matters for --force-collectives outside torchrun
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world_size, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world_size)

    world = synthetic_world()
    eng = Engine(world, device=local_rank)
    weights = ActionWeights()
    trainer = BatchTrainer(eng, weights, args.episodes, args.seed, rank, world_size, dist if use_dist else None,
                           replay_fraction=args.replay_fraction, write_yearly=not args.no_yearly,
                           force_collectives=args.force_collectives)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.step()
    eng.sync(); eng.timing_reset()
    fence()
    if saved_stdout is not None:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1); os.close(saved_stdout)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        trainer.step()
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    trainer.sync()                     # device-resident policy -> host copy (after the timed region)
    kernel_ms, n_launch = eng.timing_read()
    res = eng.fetch(args.episodes)
    ok = int((res.status == 0).sum())
    bytes_per_launch = float(res.bytes_moved.sum())
    avg_kernel_s = kernel_ms / max(n_launch, 1) * 1e-3
    achieved = bytes_per_launch / avg_kernel_s / 1e9 if avg_kernel_s > 0 else 0.0

    if rank == 0:
        total_eps = args.episodes * world_size * args.steps
        line = {
            "metric": "26-year episodes/sec", "value": total_eps / elapsed, "unit": "episodes/s",
            "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1]: {args.episodes} parallel 2025-2050 episodes per GPU per step "
                                   "(grid step + tabular policy sampling + batch policy update), synthetic world "
                                   "S=130 settlements / G0=59 existing plant / P=200 coast points, fresh ActionWeights, seed 12345",
                       "episodes_per_gpu_per_step": args.episodes, "replay_fraction": args.replay_fraction,
                       "parallelism": f"episode-sharded dp{world_size}, policy resident on every GPU, one 32 KB all-gather per update "
                                      "(integer statistics summed in the update kernel)",
                       "episodes_ok_last_batch": ok, "strategy_improvements": trainer.improvements},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "peak_measured": measured_peak(), "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(args.episodes), "kernel": "k_rollout",
                         "avg_kernel_ms": avg_kernel_s * 1e3, "algorithmic_bytes_per_launch": bytes_per_launch,
                         "bytes_per_episode": bytes_per_launch / max(args.episodes, 1),
                         "kernel_only_episodes_per_s": args.episodes / avg_kernel_s if avg_kernel_s > 0 else 0.0},
        }
        if world_size == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(world)
        print(json.dumps(line), flush=True)
    eng.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
