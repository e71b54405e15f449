"""Synthetic world of the reference's shape.

The reference's assets (130 settlements, 59 existing generators, 200 coastline
points: /root/reference/aiSimulator/assets/) may not be redistributed, so every
benchmark and parity test runs on a synthetic world with the same shapes, value
ranges and type mix.  The generator is integer-exact (splitmix64) except for a
few libm calls whose results are rounded to a coarse grid before use, so the
same world is produced on every machine; tests/golden/world_v1.json pins it.

Type indices follow the reference enum order (models/generator.rs:11-36).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

WORLD_SEED = 0xE16D0001
MAP_MAX = 50_000.0

# generator type indices (models/generator.rs:11-36)
ONSHORE_WIND, OFFSHORE_WIND, DOMESTIC_SOLAR, COMMERCIAL_SOLAR, UTILITY_SOLAR, NUCLEAR, COAL_PLANT, \
    GAS_COMBINED_CYCLE, GAS_PEAKER, BIOMASS, HYDRO_DAM, PUMPED_STORAGE, BATTERY_STORAGE, TIDAL_GENERATOR, \
    WAVE_ENERGY = range(15)

GENERATOR_TYPE_NAMES = [
    "OnshoreWind", "OffshoreWind", "DomesticSolar", "CommercialSolar", "UtilitySolar", "Nuclear", "CoalPlant",
    "GasCombinedCycle", "GasPeaker", "Biomass", "HydroDam", "PumpedStorage", "BatteryStorage", "TidalGenerator",
    "WaveEnergy",
]


@dataclass
class World:
    """Plain SoA view of what `initialize_map` (main.rs:74-193) loads into the reference `Map`."""
    settlement_x: np.ndarray      # f64 [S]  grid metres
    settlement_y: np.ndarray      # f64 [S]
    settlement_pop: np.ndarray    # u32 [S]  2025 population
    existing_x: np.ndarray        # f64 [G0]
    existing_y: np.ndarray        # f64 [G0]
    existing_type: np.ndarray     # i32 [G0] generator type index
    existing_capacity: np.ndarray  # f64 [G0] MW (CSV `capacity_mw`)
    coast_x: np.ndarray           # f64 [P]
    coast_y: np.ndarray           # f64 [P]
    existing_operational_at_start: bool = False   # Q1 switch; False = HEAD behaviour

    def to_json_dict(self) -> dict:
        return {
            "settlement_x": self.settlement_x.tolist(), "settlement_y": self.settlement_y.tolist(),
            "settlement_pop": self.settlement_pop.tolist(),
            "existing_x": self.existing_x.tolist(), "existing_y": self.existing_y.tolist(),
            "existing_type": self.existing_type.tolist(), "existing_capacity": self.existing_capacity.tolist(),
            "coast_x": self.coast_x.tolist(), "coast_y": self.coast_y.tolist(),
            "existing_operational_at_start": bool(self.existing_operational_at_start),
        }

    @staticmethod
    def from_json_dict(d: dict) -> "World":
        return World(
            np.asarray(d["settlement_x"], dtype=np.float64), np.asarray(d["settlement_y"], dtype=np.float64),
            np.asarray(d["settlement_pop"], dtype=np.uint32),
            np.asarray(d["existing_x"], dtype=np.float64), np.asarray(d["existing_y"], dtype=np.float64),
            np.asarray(d["existing_type"], dtype=np.int32), np.asarray(d["existing_capacity"], dtype=np.float64),
            np.asarray(d["coast_x"], dtype=np.float64), np.asarray(d["coast_y"], dtype=np.float64),
            bool(d.get("existing_operational_at_start", False)),
        )


class _SplitMix64:
    def __init__(self, seed: int):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next(self) -> int:
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def uniform(self) -> float:
        return (self.next() >> 11) * (1.0 / 9007199254740992.0)


def synthetic_world(seed: int = WORLD_SEED, n_settlements: int = 130, n_coast: int = 200,
                    existing_operational_at_start: bool = False) -> World:
    rng = _SplitMix64(seed)
    # settlements: uniform coordinates (1 mm grid), log-uniform populations rescaled to the reference's 2025 total
    sx = np.array([round(rng.uniform() * MAP_MAX, 3) for _ in range(n_settlements)], dtype=np.float64)
    sy = np.array([round(rng.uniform() * MAP_MAX, 3) for _ in range(n_settlements)], dtype=np.float64)
    lo, hi, total = 318.0, 590_898.0, 5_149_136
    raw = [math.exp(round(math.log(lo) + rng.uniform() * (math.log(hi) - math.log(lo)), 6)) for _ in range(n_settlements)]
    scale = total / sum(raw)
    pops = [max(318, int(round(r * scale))) for r in raw]
    pops[pops.index(max(pops))] += total - sum(pops)
    # existing plant: the reference's fuel mix (generators_loader.rs:47-57 applied to ireland_generators.csv)
    types = [ONSHORE_WIND] * 38 + [GAS_COMBINED_CYCLE] * 10 + [GAS_PEAKER] * 6 + [HYDRO_DAM] * 3 + [COAL_PLANT] + [BIOMASS]
    for i in range(len(types) - 1, 0, -1):
        j = rng.next() % (i + 1)
        types[i], types[j] = types[j], types[i]
    gx = np.array([round(rng.uniform() * MAP_MAX, 3) for _ in types], dtype=np.float64)
    gy = np.array([round(rng.uniform() * MAP_MAX, 3) for _ in types], dtype=np.float64)
    cap = np.array([round(math.exp(math.log(1.6) + rng.uniform() * (math.log(915.0) - math.log(1.6))), 1) for _ in types],
                   dtype=np.float64)
    # coastline: star-shaped polygon around the map centre, radius 12-22 km
    cx, cy = [], []
    for k in range(n_coast):
        r = 12_000.0 + rng.uniform() * 10_000.0
        a = 2.0 * math.pi * k / n_coast
        cx.append(round(25_000.0 + r * math.cos(a), 3))
        cy.append(round(25_000.0 + r * math.sin(a), 3))
    return World(sx, sy, np.array(pops, dtype=np.uint32), gx, gy, np.array(types, dtype=np.int32), cap,
                 np.array(cx, dtype=np.float64), np.array(cy, dtype=np.float64), existing_operational_at_start)
