"""eirgrid_amd — MI355X-native rollout engine for GridAI-style grid-planning episodes.

Only the hot path of ETM-Code/eirgrid's aiSimulator lives here: the 2025-2050 episode rollout (demand, dispatch,
emissions/cost/opinion reward, deficit repair, tabular policy sampling, placement arg-max) as hand-written HIP
kernels for gfx950 behind a C ABI (include/eirgrid_hip.h), plus the host-side mirror of the reference interface.
"""
from .world import World, synthetic_world  # noqa: F401

__all__ = ["World", "synthetic_world"]
