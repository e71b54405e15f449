"""Known answers the REFERENCE holds (SURVEY §8(c) pins 2 and 3), checked against its own data files read in place:
README.md:96 "Power Generation 2025 = 7390.91 MW" and the shapes / fuel mix of aiSimulator/assets/*.  The assets may not be
redistributed, so nothing of them is stored here — only the README's numbers (tests/golden/readme_generation.json).  These tests
run in the build container, where /root/reference exists, and fail loudly there if an asset is missing; the GPU box has no
reference and skips them."""
import json
import os
import subprocess

import numpy as np
import pytest

from oracle import api as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "eirgrid_amd", "eirgrid-hip")
REF = "/root/reference"
ASSETS = os.path.join(REF, "aiSimulator", "assets")
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "readme_generation.json")))
UTILITY_SOLAR_100 = 3 * 4      # canonical action index: AddGenerator(UtilitySolar, 100 %)


def _reference_world(tmp_path, at_start):
    """The reference's assets through the CLI's own loader (csrc/eg_cli.cpp load_reference_assets = main.rs:74-193,
    data/generators_loader.rs:47-57, :133-206, data/settlements_loader.rs:23-41), dumped in the --world JSON form."""
    if not os.path.isdir(REF):
        pytest.skip("no /root/reference on this machine (the GPU box): the pin runs in the build container")
    for f in ("settlements.json", "ireland_generators.csv", "coastline_points.json"):
        assert os.path.exists(os.path.join(ASSETS, f)), f"the build container must have the reference's {f}: it is a reference-held pin of the oracle"
    path = str(tmp_path / f"reference_world_{int(at_start)}.json")
    args = [CLI, "--assets-dir", ASSETS, "--dump-world", path] + (["--existing-operational-at-start"] if at_start else [])
    out = subprocess.run(args, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    from eirgrid_amd.world import World
    return out.stdout, World.from_json_dict(json.load(open(path)))


def test_cli_reads_the_reference_assets_in_place(built, tmp_path):
    """130 settlements / 59 existing generators / 200 coastline points, fuel mix 38-10-6-3-1-1 (SURVEY §8 'Sizes at BASELINE
    configs'; data/generators_loader.rs:47-57 maps gas -> GasCombinedCycle, oil -> GasPeaker)."""
    stdout, world = _reference_world(tmp_path, False)
    assert f"World: {GOLD['settlements']} settlements, {GOLD['existing_generators']} existing generators, {GOLD['coastline_points']} coastline points" in stdout
    line = [l for l in stdout.splitlines() if l.startswith("Existing generators by type: ")][0]
    mix = dict((k, int(v)) for k, v in (item.rsplit(" ", 1) for item in line.split(": ", 1)[1].split(", ")))
    assert mix == GOLD["type_mix"]
    assert (len(world.settlement_x), len(world.existing_x), len(world.coast_x)) == (130, 59, 200)
    assert int(world.settlement_pop.sum()) == 5149136      # README.md:96 Pop. 2025
    assert world.settlement_x.min() >= 0.0 and world.settlement_x.max() <= 50000.0 and not world.existing_operational_at_start


def test_readme_generation_2025(built, tmp_path):
    """README.md:96: Power Generation 2025 = 7390.91 MW = the output of the 59 existing plants of ireland_generators.csv
    (operational in 2025, as the code behind the README had them: SURVEY Q1 switch) + the twelve UtilitySolar additions of the
    README's 2025 action list at 59.4 MW each (models/generator.rs:523-554).  Held against (a) the product's host tables, (b) the
    literal oracle replaying exactly those twelve additions, (c) the tabled oracle on the product's tables, (d) a sampled
    literal episode minus what it added itself."""
    from eirgrid_amd.engine import HostTables
    _, world = _reference_world(tmp_path, True)
    assert world.existing_operational_at_start
    assert "%.2f" % (float(GOLD["generation_2025_mw"]) - GOLD["utility_solar_added_2025"] * GOLD["utility_solar_output_mw"]) == GOLD["existing_generation_mw"]
    assert "%.1f" % O.lib().og_type_power_output(4) == "%.1f" % GOLD["utility_solar_output_mw"]
    # (a) existing-plant prefix of the 2025 aggregates (eg_tables.cpp pre_tg / pre_ig / pre_sg, folded in list order)
    H = HostTables(world)
    existing = (H.f64("pre_tg")[0] + H.f64("pre_ig")[0]) + H.f64("pre_sg")[0]
    assert "%.2f" % existing == GOLD["existing_generation_mw"]
    assert all(int(y) == 0 for y in H.i32("existing_online"))
    # (b), (c) the README's 2025 additions replayed: twelve UtilitySolar, nothing else in the year
    ow = O.OracleWorld(world)
    pol = O.OracleWeights()
    pol.set("has_best", 1); pol.set("has_best_actions", 1); pol.set("has_best_deficit_actions", 1)
    pol.set_list(0, 0, [UTILITY_SOLAR_100] * GOLD["utility_solar_added_2025"])
    st, ep = O.run_episode(ow, pol.clone(), 12345, replay=True)
    assert st == 0 and ep.n_def[0] == 0 and ep.n_act[0] == 12      # no deficit with the existing plant online: exactly the twelve
    assert "%.2f" % ep.yearly[0][3] == GOLD["generation_2025_mw"]      # EG_Y_GEN
    st, et = O.run_episode_tabled(O.OracleTables(H, len(world.existing_x)), pol.clone(), 12345, replay=True)
    assert st == 0 and "%.2f" % et.yearly[0][3] == GOLD["generation_2025_mw"]
    assert np.array(ep.yearly).tobytes() == np.array(et.yearly).tobytes()      # literal == tabled on the reference's own data
    # (d) a sampled episode: 2025 generation minus its own 2025 additions is the existing plant's
    st, es = O.run_episode(ow, O.OracleWeights(), 12345)
    own = sum(O.lib().og_type_power_output(int(t)) for t, y in zip(es.gen_type[:es.n_gens], es.gen_year[:es.n_gens]) if y == 0)
    assert st == 0 and "%.2f" % (es.yearly[0][3] - own) == GOLD["existing_generation_mw"]
