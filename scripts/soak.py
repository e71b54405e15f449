"""Soak test on the GPU box (not part of pytest: minutes, not seconds).
  python scripts/soak.py [trials]
Every trial draws a random policy (perturbed tables, stall, rates, random best lists — every fourth trial long ones, so that
the replay episodes place hundreds of generators, every sixteenth 20-40 additions per year: 500-1 000 generators per replay
episode, beyond the 512 the kernels keep in LDS), runs 1,536 episodes through BOTH kernels (helper-wave and single-wave),
through an engine without the penalty-field pool (every search the exact scan) and through two engines with the replay hoist on (as it
decides, and with every hoisted search forced down its slow path), and demands identical bytes, checks 48
random episodes against the tabled CPU oracle bit for bit, and then takes 6 training steps on the device and on the host
from that policy and demands identical policies, and the independent restatement of the batch update within 1e-12 of them.
Exit code 0 = everything matched."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eirgrid_amd import synthetic_world
from eirgrid_amd.engine import ActionWeights, Engine, HostTables
from oracle import api as O
from tests.helpers import assert_episode_equal, oracle_weights_like

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 20
world = synthetic_world()
engines = {}
for mode in ("0", "all"):
    os.environ["EIRGRID_HELPER_WAVES"] = mode
    engines[mode] = Engine(world, device=0)
os.environ["EIRGRID_HELPER_WAVES"] = "0"; os.environ["EIRGRID_HEAVY_SLOTS"] = "0"
engines["exact"] = Engine(world, device=0)
del os.environ["EIRGRID_HELPER_WAVES"]; del os.environ["EIRGRID_HEAVY_SLOTS"]
# the replay hoist (eg_replay_hoist): the replay episodes of a batch computed once — as it decides by itself, and with every hoisted
# search forced down its slow path (EIRGRID_COOP_FORCE=1: exact evaluation of the candidates)
engines["hoist"] = Engine(world, device=0); engines["hoist"].replay_hoist(True)
os.environ["EIRGRID_COOP_FORCE"] = "1"
engines["hoist_slow"] = Engine(world, device=0); engines["hoist_slow"].replay_hoist(True)
del os.environ["EIRGRID_COOP_FORCE"]
# the throughput kernels without k_replay_solo (eg_replay_solo.h: engine "0" runs long replay episodes through it): k_rollout<0,2> alone
os.environ["EIRGRID_HELPER_WAVES"] = "0"; os.environ["EIRGRID_REPLAY_SOLO"] = "0"
engines["classic"] = Engine(world, device=0)
del os.environ["EIRGRID_HELPER_WAVES"]; del os.environ["EIRGRID_REPLAY_SOLO"]
served = 0
dev = Engine(world, device=0)
tb = O.OracleTables(HostTables(world), len(world.existing_x))
rng = np.random.default_rng(int(time.time()) if len(sys.argv) > 2 else 20261004)
t0 = time.time()
for trial in range(trials):
    pol = ActionWeights()
    heavy = trial % 4 == 3
    run = [rng.integers(0, 61, int(rng.choice([0, 0, 1, 2, 5, 9]))).tolist() for _ in range(26)]
    if heavy:
        types = rng.choice(15, int(rng.integers(1, 6)), replace=False)
        lo, hi = (20, 41) if trial % 16 == 15 else (5, 15)
        run = [[int(3 * rng.choice(types) + rng.integers(0, 3)) for _ in range(int(rng.integers(lo, hi)))] for _ in range(26)]
    dfl = [(3 * rng.choice([8, 7, 12, 11, 9, 0, 1, 4, 10, 5, 2, 3, 13, 14], int(rng.choice([0, 1, 2, 3])))).tolist() for _ in range(26)]
    if trial % 2 == 1:      # a script a replay can follow to its end without a fallback draw (four repair actions a year): the hoist takes it
        dfl = [[int(rng.choice([3 * 8, 3 * 7, 3 * 12, 3 * 11, 3 * 0, 48, 60])) for _ in range(4)] for _ in range(26)]
    nr = np.array([len(l) for l in run], np.int32); nd = np.array([len(l) for l in dfl], np.int32)
    pol.apply_episode([float(rng.choice([-5e4, 3e5])), 0.7, float(rng.choice([4e10, 9e11])), 1.0], nr,
                      np.array([a for l in run for a in l], np.uint8), nd, np.array([a for l in dfl for a in l], np.uint8))
    w, dw, cw = pol.tables()
    pol.set_tables(np.clip(w * 10 ** rng.uniform(-1.5, 1.0, w.shape), 1e-4, 0.999), np.clip(dw * 10 ** rng.uniform(-1.5, 1.0, dw.shape), 1e-4, 0.999),
                   cw * rng.uniform(0.2, 3.0, cw.shape))
    pol.set("iterations_without_improvement", int(rng.choice([0, 50, 150, 480, 520, 1400, 4000])))
    pol.set("learning_rate", float(rng.uniform(0.05, 0.5))); pol.set("exploration_rate", float(rng.uniform(0.0, 0.9)))
    if trial % 3 == 2:
        pol.set("has_count_weights", 0)
    n = 1536; seed = int(rng.integers(1, 2**40)); first = int(rng.integers(0, 2**20))
    mask = (rng.uniform(size=n) < 0.2).astype(np.uint8)
    a = engines["0"].rollout_batch(pol, seed, n, first_episode_index=first, replay_mask=mask)
    b = engines["all"].rollout_batch(pol, seed, n, first_episode_index=first, replay_mask=mask)
    c = engines["exact"].rollout_batch(pol, seed, n, first_episode_index=first, replay_mask=mask)
    for name in ("status", "metrics", "yearly", "n_run", "n_def", "n_act", "n_gens", "n_offsets", "n_draws", "bytes_moved"):
        assert getattr(a, name).tobytes() == getattr(b, name).tobytes(), (trial, name)
        assert getattr(a, name).tobytes() == getattr(c, name).tobytes(), (trial, name, "field path vs exact scan")
    for key in ("hoist", "hoist_slow", "classic"):
        h = engines[key].rollout_batch(pol, seed, n, first_episode_index=first, replay_mask=mask)
        for name in ("status", "metrics", "yearly", "n_run", "n_def", "n_act", "n_gens", "n_offsets", "n_draws", "bytes_moved"):
            assert getattr(a, name).tobytes() == getattr(h, name).tobytes(), (trial, name, key)
        for name, cnt in (("gen_cell", a.n_gens), ("gen_pack", a.n_gens), ("off_pack", a.n_offsets), ("run_log", a.n_run.sum(axis=1)), ("def_log", a.n_def.sum(axis=1)), ("act_log", a.n_act.sum(axis=1))):
            lv = np.arange(getattr(a, name).shape[1])[None, :] < cnt[:, None]
            assert (getattr(a, name)[lv] == getattr(h, name)[lv]).all(), (trial, name, key)
    served += int(engines["hoist"].replay_hoist_stats()[1])
    live = np.arange(a.gen_cell.shape[1])[None, :] < a.n_gens[:, None]      # (the buffers are not cleared between batches)
    for name in ("gen_cell", "gen_pack"):
        assert (getattr(a, name)[live] == getattr(b, name)[live]).all() and (getattr(a, name)[live] == getattr(c, name)[live]).all(), (trial, name)
    ok = np.flatnonzero(a.status == 0)
    for e in rng.choice(ok, min(48, len(ok)), replace=False):
        st, ref = O.run_episode_tabled(tb, oracle_weights_like(pol), seed + first + int(e), replay=bool(mask[e]))
        assert_episode_equal(b, int(e), ref, f"soak trial {trial}")
    # six training steps: device-resident vs host update
    host = ActionWeights(); host.set_tables(*pol.tables())
    for name in ("iterations_without_improvement", "learning_rate", "exploration_rate", "has_count_weights"):
        host.set(name, pol.get(name))
    devp = ActionWeights(); devp.set_tables(*host.tables())
    for name in ("iterations_without_improvement", "learning_rate", "exploration_rate", "has_count_weights"):
        devp.set(name, host.get(name))
    dev.push(devp)
    ow = oracle_weights_like(devp)
    for step in range(6):
        f = first + step * 256
        m = ((np.arange(f, f + 256) % 3) == 0).astype(np.uint8) if host.get("has_best_actions") == 1 else None
        engines["0"].train_step(host, seed, f, 256, m, noise_seed=seed + step)
        dev.device_step(seed, f, 256, 3, seed + step)
        res = dev.fetch(256)
        O.reduced_batch_update(ow, res.status, res.metrics, res.n_run, res.n_def, res.run_log, res.def_log, noise_seed=seed + step)
    dev.pull(devp)
    for x, y in zip(devp.tables()[:2], ow.tables()[:2]):
        # (the statistics are Q32 logarithms rounded with llrint on the device (ocml log) and in the restatement (glibc log): at a rounding
        #  boundary an episode's contribution differs by one unit, 2.3e-10 relative in a weight — seen once in 4 729 trials x 6 steps)
        np.testing.assert_allclose(x, y, rtol=1e-7, atol=0, err_msg=f"trial {trial}: device policy vs independent restatement")
    assert devp.lists(0) == ow.lists(0) and devp.get("iterations_without_improvement") == ow.get("stall")
    for x, y in zip(host.tables(), devp.tables()):
        assert x.tobytes() == y.tobytes(), (trial, "device vs host policy")
    assert host.lists(0) == devp.lists(0) and host.get("iterations_without_improvement") == devp.get("iterations_without_improvement")
    print(f"trial {trial:3d} ok  ({time.time() - t0:.0f} s; failed episodes in batch: {int((a.status != 0).sum())}; most generators in an episode {int(a.n_gens.max())})", flush=True)
print("soak: all", trials, "trials matched; the replay hoist served", served, "of them (the others needed a fallback draw or hit a capacity)")
