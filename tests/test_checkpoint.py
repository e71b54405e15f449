"""CPU: checkpoint JSON in the reference's SerializableWeights schema."""
import json

import numpy as np
import pytest

from eirgrid_amd import _native as N
from eirgrid_amd.engine import ActionWeights, HostTables
from oracle import api as O

FIELDS = ["weights", "learning_rate", "best_metrics", "best_weights", "best_actions", "iteration_count",
          "iterations_without_improvement", "exploration_rate", "deficit_weights", "best_deficit_actions", "optimization_mode",
          "improvement_history"]          # ai/learning/serialization.rs:38-51, in declaration order
ACTION_KEYS = ["action_type", "generator_type", "generator_id", "operation_percentage", "offset_type", "cost_multiplier"]


def _trained_policy(world, n=12):
    ot = O.OracleTables(HostTables(world), len(world.existing_x))
    pol = ActionWeights()
    for it in range(n):
        st, out = O.run_episode_tabled(ot, O.OracleWeights(), 900 + it)
        lists = (O.split_log(out.run_log, out.n_run), O.split_log(out.def_log, out.n_def))
        pol.apply_episode(list(out.metrics), [len(l) for l in lists[0]], [a for l in lists[0] for a in l],
                          [len(l) for l in lists[1]], [a for l in lists[1] for a in l], noise_seed=it)
    return pol


def test_fresh_checkpoint_schema(tmp_path, built):
    path = tmp_path / "latest_weights.json"
    ActionWeights().save_to_file(path)
    d = json.load(open(path))
    assert list(d.keys()) == FIELDS
    assert d["best_metrics"] is None and d["best_actions"] is None and d["best_weights"] is None and d["improvement_history"] is None
    assert sorted(d["weights"].keys()) == [str(y) for y in range(2025, 2051)]
    row = d["weights"]["2025"]
    assert len(row) == 61 and all(len(e) == 2 and list(e[0].keys()) == ACTION_KEYS for e in row)
    assert row[0] == [{"action_type": "AddGenerator", "generator_type": "OnshoreWind", "generator_id": None,
                       "operation_percentage": None, "offset_type": None, "cost_multiplier": 100}, 0.08]
    assert row[58][0] == {"action_type": "AdjustOperation", "generator_type": None, "generator_id": "", "operation_percentage": 0,
                          "offset_type": None, "cost_multiplier": None}
    assert row[60][0]["action_type"] == "DoNothing" and row[60][1] == 0.1
    assert len(d["deficit_weights"]["2050"]) == 15
    assert d["deficit_weights"]["2025"][0][0]["generator_type"] == "GasPeaker" and d["deficit_weights"]["2025"][14][0]["action_type"] == "DoNothing"
    assert open(path).read().startswith('{\n  "weights": {\n    "2025": [\n      [\n        {\n          "action_type"')   # pretty, 2 spaces
    assert "action_count_weights" not in d


def test_round_trip_is_exact(tmp_path, world):
    pol = _trained_policy(world)
    path = tmp_path / "ck.json"
    pol.save_to_file(path)
    back = ActionWeights.load_from_file(path)
    for a, b in zip(pol.tables()[:2], back.tables()[:2]):
        assert a.tobytes() == b.tobytes()           # shortest round-trip float formatting
    for name in ("learning_rate", "exploration_rate", "iterations_without_improvement", "iteration_count", "has_best",
                 "best_net_emissions", "best_opinion", "best_cost", "best_reliability", "has_best_actions", "has_best_deficit_actions"):
        assert pol.get(name) == back.get(name), name
    assert pol.lists(0) == back.lists(0) and pol.lists(1) == back.lists(1)
    assert back.get("has_count_weights") == 0      # dropped by the loader, like the reference (serialization.rs:474)
    d = json.load(open(path))
    assert d["improvement_history"] and set(d["improvement_history"][0]) == {"iteration", "score", "net_emissions", "total_cost",
                                                                             "public_opinion", "power_reliability", "timestamp"}
    path2 = tmp_path / "ck2.json"
    back.save_to_file(path2)
    assert open(path).read() == open(path2).read()


def test_loads_a_reference_style_file(tmp_path, built):
    """HashMap order is arbitrary in files written by the reference; unknown generator types are an error there too."""
    ActionWeights().save_to_file(tmp_path / "a.json")
    d = json.load(open(tmp_path / "a.json"))
    rng = np.random.default_rng(0)
    for table in ("weights", "deficit_weights"):
        keys = list(d[table].keys()); rng.shuffle(keys)
        d[table] = {k: [d[table][k][i] for i in rng.permutation(len(d[table][k]))] for k in keys}
    d["weights"]["2031"] = [[a, 0.5 if a["action_type"] == "DoNothing" else w] for a, w in d["weights"]["2031"]]
    json.dump(d, open(tmp_path / "b.json", "w"))          # compact, different order
    pol = ActionWeights.load_from_file(tmp_path / "b.json")
    w, dw, _ = pol.tables()
    w0, dw0, _ = ActionWeights().tables()
    assert w[6, 60] == 0.5 and (np.delete(w, 6, 0) == np.delete(w0, 6, 0)).all() and (dw == dw0).all()
    next(e for e in d["weights"]["2025"] if e[0]["action_type"] == "AddGenerator")[0]["generator_type"] = "FusionReactor"
    json.dump(d, open(tmp_path / "c.json", "w"))
    with pytest.raises(N.EirgridError):
        ActionWeights.load_from_file(tmp_path / "c.json")
    with pytest.raises(N.EirgridError):
        ActionWeights.load_from_file(tmp_path / "missing.json")


def test_weight_history_snapshots(built, tmp_path):
    """--track-weight-history: weight_history.json is a JSON array that grows by one ActionWeights::to_json() snapshot per
    checkpoint; objects list their keys in byte order (serde_json's BTreeMap), table keys are the actions' Display strings."""
    import json
    from eirgrid_amd import _native as N
    from eirgrid_amd.engine import ActionWeights
    pol = ActionWeights()
    path = str(tmp_path / "weight_history.json")
    (tmp_path / "weight_history.json").write_text("[]")            # the reference creates it like this
    for it in (4, 9, 14):
        N.check(N.lib().eg_policy_append_weight_history(pol.h, path.encode(), it))
    text = open(path).read()
    hist = json.loads(text)
    assert [h["iteration"] for h in hist] == [4, 9, 14]
    snap = hist[0]
    assert list(snap.keys()) == sorted(snap.keys()) == ["best_score", "iteration", "timestamp", "weights"]
    w = snap["weights"]
    assert list(w.keys()) == sorted(w.keys())
    assert set(w.keys()) == {"action_count_weights", "best_score", "deficit_weights", "exploration_rate", "force_best_actions",
                             "guaranteed_best_actions", "iteration_count", "iterations_without_improvement", "learning_rate",
                             "optimization_mode", "weights"}
    y = w["weights"]["2025"]
    assert len(y) == 61 and list(y.keys()) == sorted(y.keys())
    assert "AddGenerator(OnshoreWind, 100%)" in y and "AddCarbonOffset(Forest, 150%)" in y and "AdjustOperation(, 0%)" in y and "DoNothing" in y
    table, _, counts = pol.tables()
    assert y["AddGenerator(OnshoreWind, 100%)"] == table[0, 0] and y["DoNothing"] == table[0, 60]
    assert list(w["action_count_weights"]["2030"].keys()) == sorted(str(i) for i in range(21))
    assert w["action_count_weights"]["2030"]["7"] == counts[5, 7]
    assert len(w["deficit_weights"]["2050"]) == 15 and "AddGenerator(GasPeaker, 100%)" in w["deficit_weights"]["2050"]
    assert text.startswith("[\n  {\n    \"best_score\": 0.0,\n    \"iteration\": 4,") and text.endswith("\n  }\n]")
    assert snap["timestamp"][10] == "T" and snap["timestamp"][-6] in "+-"


def test_improvement_history_csv(built, tmp_path):
    """utils/csv_export.rs:155-207: header, one row per improvement, score improvement in per cent of the previous score."""
    from eirgrid_amd import _native as N
    from eirgrid_amd.engine import ActionWeights
    pol = ActionWeights()
    path = tmp_path / "improvement_history.csv"
    N.check(N.lib().eg_policy_export_improvement_csv(pol.h, str(path).encode()))
    assert not path.exists()                              # no history, no file
    ones, none = np.ones(26, np.int32), np.zeros(26, np.int32)
    pol.apply_episode([5e5, 0.6, 9e11, 1.0], ones, np.full(26, 60, np.uint8), none, np.zeros(0, np.uint8))       # score 0.5
    pol.apply_episode([2.5e5, 0.7, 8e11, 1.0], ones, np.full(26, 60, np.uint8), none, np.zeros(0, np.uint8))     # score 0.75
    N.check(N.lib().eg_policy_export_improvement_csv(pol.h, str(path).encode()))
    lines = path.read_text(encoding="utf-8").splitlines()
    assert lines[0] == "Iteration,Score,Net Emissions (tonnes),Total Cost (€),Public Opinion (%),Power Reliability (%),Score Improvement (%),Timestamp"
    r1, r2 = lines[1].split(","), lines[2].split(",")
    assert r1[:7] == ["1", "0.500000", "500000.00", "900000000000.00", "60.00", "100.00", "0.00"]
    assert r2[:7] == ["2", "0.750000", "250000.00", "800000000000.00", "70.00", "100.00", "50.00"]
    assert len(lines) == 3
