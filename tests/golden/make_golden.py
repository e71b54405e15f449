"""Regenerates the committed fixtures under tests/golden/ (run from the repo root: python tests/golden/make_golden.py).

  world_v1.json      the synthetic world of SURVEY.md §8(d) (eirgrid_amd.world.synthetic_world, seed 0xE16D0001)
  episode_v1.json    BASELINE config 1: one 2025-2050 episode, seed 12345, fresh ActionWeights, CPU oracle (literal mode)
  readme_demand.json the Pop. / Power Usage columns of the reference README.md:93-121 (known answers of the demand step
                     for the reference's own settlements.json, which is not redistributable and therefore not stored)
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import numpy as np  # noqa: E402

from eirgrid_amd.world import synthetic_world  # noqa: E402
from oracle import api as O  # noqa: E402

README_ROWS = [  # year, Pop., Power Usage (MW) — /root/reference/README.md:96-121
    (2025, 5149136, 5252.12), (2026, 5200628, 5516.83), (2027, 5252636, 5792.73), (2028, 5305160, 6080.27),
    (2029, 5358215, 6379.89), (2030, 5411800, 6692.07), (2031, 5465919, 7017.28), (2032, 5520574, 7356.03),
    (2033, 5575778, 7708.84), (2034, 5631527, 8076.24), (2035, 5687845, 8458.81), (2036, 5744726, 8857.13),
    (2037, 5802180, 9271.79), (2038, 5860199, 9703.41), (2039, 5918800, 10152.65), (2040, 5977982, 10620.16),
    (2041, 6037760, 11106.66), (2042, 6098141, 11612.86), (2043, 6159121, 12139.5), (2044, 6220709, 12687.36),
    (2045, 6282917, 13257.24), (2046, 6345748, 13849.97), (2047, 6409208, 14466.42), (2048, 6473298, 15107.45),
    (2049, 6538030, 15774.02), (2050, 6603409, 16467.06),
]


def main():
    world = synthetic_world()
    with open(os.path.join(HERE, "world_v1.json"), "w") as f:
        json.dump(world.to_json_dict(), f)
    with open(os.path.join(HERE, "readme_demand.json"), "w") as f:
        json.dump({"source": "README.md:96-121 of ETM-Code/eirgrid", "rows": README_ROWS}, f, indent=1)
    ow = O.OracleWorld(world)
    wts = O.OracleWeights()
    st, out = O.run_episode(ow, wts, 12345)
    assert st == 0
    w, dw, cw = wts.tables()
    ep = {
        "seed": 12345, "status": st, "metrics": [float.hex(v) for v in out.metrics],
        "yearly": [[float.hex(v) for v in row] for row in out.yearly],
        "run": O.split_log(out.run_log, out.n_run), "deficit": O.split_log(out.def_log, out.n_def),
        "actions": O.split_log(out.act_log, out.n_act),
        "gen_cell": list(out.gen_cell[:out.n_gens]), "gen_type": list(out.gen_type[:out.n_gens]),
        "gen_year": list(out.gen_year[:out.n_gens]), "gen_mult": list(out.gen_mult[:out.n_gens]),
        "off_type": list(out.off_type[:out.n_offsets]), "off_year": list(out.off_year[:out.n_offsets]),
        "n_draws": int(out.n_draws),
        "nudged_weights_sha": __import__("hashlib").sha256(w.tobytes() + dw.tobytes()).hexdigest(),
        "stream_head": [int(v) for v in O.rng_stream(12345, 8)],
    }
    with open(os.path.join(HERE, "episode_v1.json"), "w") as f:
        json.dump(ep, f)
    print("golden fixtures written")


if __name__ == "__main__":
    main()
