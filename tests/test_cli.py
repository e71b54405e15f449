"""The eirgrid-hip command-line driver: flags of cli/cli.rs, checkpoint layout of run_multi_simulation."""
import json
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "eirgrid_amd", "eirgrid-hip")
WORLD = os.path.join(ROOT, "tests", "golden", "world_v1.json")


def run(*args, cwd=None):
    return subprocess.run([CLI, *args], capture_output=True, text=True, timeout=600, cwd=cwd)


def test_help_lists_every_reference_flag(built):
    out = run("--help")
    assert out.returncode == 0
    for flag in ("--iterations", "--parallel", "--no-continue", "--checkpoint-dir", "--checkpoint-interval", "--progress-interval",
                 "--cache-dir", "--force-full-simulation", "--enable-timing", "--seed", "--verbose-state-logging", "--cost-only",
                 "--enable-energy-sales", "--enable-csv-export", "--debug-logging", "--debug-weights", "--enable-construction-delays",
                 "--track-weight-history"):                                   # cli/cli.rs:5-59
        assert flag in out.stdout, flag
    assert run("--no-such-flag").returncode == 2


def test_fails_loudly_without_gpu(built, tmp_path):
    from eirgrid_amd import _native as N
    if N.lib().eg_device_count() > 0:
        pytest.skip("a GPU is present")
    out = run("--world", WORLD, "-n", "4", "-c", str(tmp_path / "ck"))
    assert out.returncode == 1 and "no HIP device" in out.stderr
    assert "World: 130 settlements, 59 existing generators, 200 coastline points" in out.stdout


def test_reads_reference_asset_formats(built, tmp_path):
    """settlements.json / ireland_generators.csv / coastline_points.json in the reference's own formats (main.rs:74-193),
    exercised on a tiny hand-written data set."""
    d = tmp_path / "assets"; d.mkdir()
    json.dump({"settlements": [{"name": "A", "lat": 53.3, "lon": -6.3, "population": 1000, "grid_x": 0, "grid_y": 0},
                               {"name": "outside", "lat": 60.0, "lon": -6.3, "population": 5}]}, open(d / "settlements.json", "w"))
    open(d / "ireland_generators.csv", "w").write("capacity_mw,latitude,longitude,primary_fuel\n100.0,52.0,-8.0,Gas\n7.5,54.0,-8.1,Wind\n")
    json.dump({"original_coords": [], "grid_coords": [[1.0, 2.0], [3.0, 4.0], [5.0, 6.0]]}, open(d / "coastline_points.json", "w"))
    out = run("--assets-dir", str(d), "-n", "1", "-c", str(tmp_path / "ck"))
    assert "World: 1 settlements, 2 existing generators, 3 coastline points" in out.stdout


@pytest.mark.gpu
def test_training_run_checkpoints_and_resume(built, tmp_path):
    ck = str(tmp_path / "checkpoints")
    out = run("--world", WORLD, "-n", "96", "--batch", "32", "--seed", "7", "-c", ck, "-i", "40", "-r", "1000")
    assert out.returncode == 0, out.stdout + out.stderr
    runs = os.listdir(ck)
    assert len(runs) == 1 and re.fullmatch(r"2024\d{4}_\d{6}", runs[0])           # multi_simulation.rs:162
    rd = os.path.join(ck, runs[0])
    assert {"latest_weights.json", "thread_0_weights.json", "checkpoint_iteration.txt", "best_weights.json"} <= set(os.listdir(rd))
    assert open(os.path.join(rd, "checkpoint_iteration.txt")).read() == "96"
    d = json.load(open(os.path.join(rd, "latest_weights.json")))
    assert d["iteration_count"] == 96 and d["best_metrics"] is not None and len(d["weights"]["2025"]) == 61
    # best-run export (csv_export.rs:114-152): improvement history + the summary of the best episode
    exports = os.listdir(os.path.join(rd, "enhanced_csv"))
    assert len(exports) == 1 and re.fullmatch(r"\d{8}_\d{6}", exports[0])
    ed = os.path.join(rd, "enhanced_csv", exports[0])
    assert {"improvement_history.csv", "simulation_summary.csv", "yearly_details", "operation_logs"} <= set(os.listdir(ed))
    assert set(os.listdir(os.path.join(ed, "yearly_details"))) == {"settlements.csv", "generators.csv", "carbon_offsets.csv"}
    # the detail files against the literal restatement, rebuilt from the summary's own action list (sampled actions) and
    # the generators.csv ids (the episode's generator list): every file byte for byte
    from eirgrid_amd import synthetic_world
    from oracle import api as O, csv_export as OC
    world = synthetic_world()
    gens = open(os.path.join(ed, "yearly_details", "generators.csv"), encoding="utf-8").read().split("\n")
    added = []
    for line in gens[1:]:
        gid = line.split(",")[1] if line else ""
        if gid.startswith("Gen_") and gid not in [g[0] for g in added]:
            _, t, y, k = gid.split("_")
            added.append((gid, OC.GENERATOR_TYPES.index(t), int(y) - 2025, int(k)))
    added.sort(key=lambda g: g[3])
    assert [g[3] for g in added] == list(range(59, 59 + len(added)))
    summary_lines = open(os.path.join(ed, "simulation_summary.csv"), encoding="utf-8").read().split("\n")
    i0 = summary_lines.index("Actions Taken") + 2
    n_act = [0] * 26; act_log = []
    for line in summary_lines[i0:summary_lines.index("Yearly Summary Metrics") - 1]:
        year, kind, gen, _, _, off, _ = line.split(",")
        a = {"AddGenerator": lambda: 3 * OC.GENERATOR_TYPES.index(gen), "AddCarbonOffset": lambda: 45 + 3 * OC.OFFSET_TYPES.index(off),
             "UpgradeEfficiency": lambda: 57, "AdjustOperation": lambda: 58, "CloseGenerator": lambda: 59, "DoNothing": lambda: 60}[kind]()
        n_act[int(year) - 2025] += 1; act_log.append(a)      # (the multiplier is not in the summary: offsets are re-priced below)
    want = OC.detail_files(world, None, O.OracleWorld(world).existing_online(), [g[1] for g in added], [g[2] for g in added], n_act, act_log, 7)
    for rel in ("yearly_details/settlements.csv", "yearly_details/generators.csv", "operation_logs/generator_operation_logs.csv"):
        assert open(os.path.join(ed, rel), encoding="utf-8").read() == want[rel], rel
    offs = open(os.path.join(ed, "yearly_details", "carbon_offsets.csv"), encoding="utf-8").read().split("\n")
    woffs = want["yearly_details/carbon_offsets.csv"].split("\n")
    assert len(offs) == len(woffs) and [l.split(",")[:10] for l in offs] == [l.split(",")[:10] for l in woffs]      # ids, coordinates, sizes
    summary = open(os.path.join(ed, "simulation_summary.csv"), encoding="utf-8").read().split("\n")
    assert summary[0] == "Simulation Summary" and summary[1] == "Timestamp," + exports[0]
    assert summary[-2].startswith("2050,") and summary[-27].startswith("2025,5149136,")
    # WHICH run is exported: the reference's `best_result` — a fold over the process's iterations in iteration order with
    # evaluate_action_impact(result -> best) > 0 (core/multi_simulation.rs:384, :613-620) — not the policy's best strategy.
    # The same training loop driven through the library, the oracle's fold over its episodes, and the files of that run:
    import numpy as np
    from eirgrid_amd.engine import ActionWeights, Engine, score_metrics
    eng = Engine(world, device=0)
    try:
        pol = ActionWeights(); eng.push(pol); eng.track_best_result(False)
        fold = O.BestResultFold(False)
        for done in (0, 32, 64):
            eng.device_step(7, done, 32, 1, 7 + done)      # no location cache: every run is "full" (replays once a best exists)
            res = eng.fetch(32)
            fold.feed(res.status, res.metrics, done)
        idx, rec = eng.fetch_best_result()
        eng.pull(pol)
    finally:
        eng.close()
    assert idx == fold.winner and rec.metrics[0].tobytes() == fold.best.tobytes()
    assert f"BEST SIMULATION RESULTS SUMMARY (iteration {idx})" in out.stdout, out.stdout
    best_strategy = [d["best_metrics"][k] for k in ("final_net_emissions", "average_public_opinion", "total_cost", "power_reliability")]
    assert best_strategy == [pol.get(k) for k in ("best_net_emissions", "best_opinion", "best_cost", "best_reliability")]
    assert rec.metrics[0].tolist() != best_strategy and score_metrics(rec.metrics[0]) < score_metrics(best_strategy), \
        "the exported run is not the best-scoring episode (and in this run it differs from it)"
    rec.export_summary_csv(str(tmp_path / "want_summary.csv"), exports[0])
    assert open(os.path.join(ed, "simulation_summary.csv"), "rb").read() == open(tmp_path / "want_summary.csv", "rb").read()
    names = json.load(open(WORLD)).get("settlement_names")
    rec.export_run_details(world, str(tmp_path / "want"), names, offset_seed=7)
    for rel in ("yearly_details/settlements.csv", "yearly_details/generators.csv", "yearly_details/carbon_offsets.csv",
                "operation_logs/generator_operation_logs.csv"):
        assert open(os.path.join(ed, rel), "rb").read() == open(tmp_path / "want" / rel, "rb").read(), rel
    # resume: the newest run directory is picked up and only the remaining iterations run
    out2 = run("--world", WORLD, "-n", "160", "--batch", "32", "--seed", "7", "-c", ck, "-r", "1000")
    assert out2.returncode == 0 and "Loaded weights from" in out2.stdout and "(96 completed, 64 remaining)" in out2.stdout
    assert json.load(open(os.path.join(rd, "latest_weights.json")))["iteration_count"] == 160
    # --no-continue starts a fresh directory
    import time; time.sleep(1.1)
    out3 = run("--world", WORLD, "-n", "8", "--batch", "8", "--seed", "7", "-c", ck, "--no-continue", "--update", "sequential")
    assert out3.returncode == 0 and len(os.listdir(ck)) == 2


@pytest.mark.gpu
def test_sequential_mode_matches_the_library_driven_loop(built, tmp_path, engine):
    """--update sequential == rollout batches + eg_policy_apply_episode in index order, driven from Python."""
    import numpy as np
    from eirgrid_amd.engine import ActionWeights
    ck = str(tmp_path / "ck")
    out = run("--world", WORLD, "-n", "48", "--batch", "16", "--seed", "99", "-c", ck, "--update", "sequential", "-C", str(tmp_path / "nocache"))
    assert out.returncode == 0, out.stdout + out.stderr
    rd = os.path.join(ck, os.listdir(ck)[0])
    cli_pol = ActionWeights.load_from_file(os.path.join(rd, "latest_weights.json"))
    pol = ActionWeights()
    for first in range(0, 48, 16):
        mask = np.full(16, 1 if pol.get("has_best_actions") else 0, dtype=np.uint8)     # no cache dir => every run is "full"
        res = engine.rollout_batch(pol, 99, 16, first_episode_index=first, replay_mask=mask, write_yearly=False)
        for e in range(16):
            nr, nd = res.n_run[e], res.n_def[e]
            pol.apply_episode(res.metrics[e], nr, res.run_log[e], nd, res.def_log[e], noise_seed=99 + first + e)
    for a, b in zip(pol.tables()[:2], cli_pol.tables()[:2]):
        assert a.tobytes() == b.tobytes()
    assert pol.lists(0) == cli_pol.lists(0) and pol.get("iterations_without_improvement") == cli_pol.get("iterations_without_improvement")
