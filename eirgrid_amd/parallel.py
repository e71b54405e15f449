"""Data-parallel rollout over the GPUs of one node: one process per GPU, torch.distributed ("nccl" = RCCL over xGMI).

Episodes are independent given a policy snapshot, so a batch is sharded by global episode index with no data-path
collective.  The only exchange is the per-update one of SURVEY.md §8(e):
  * device-resident policy (BatchTrainer's default under RCCL): ONE all-gather of every rank's 37 008-byte update packet (N.PACKET_BYTES)
    (int64 statistics + the rank's best-candidate record); k_apply_update on every rank adds the statistics (integers:
    the sum is the all-reduce) and picks the winning candidate — no host synchronisation in the step;
  * host-side policy (gloo rehearsals, exchange_packet_raw / exchange_update): a sum all-reduce of the statistics and an
    all-gather of the candidate records, then eg_policy_apply_packet on every rank.
Either way every rank applies the identical update to its own copy of the policy (integer statistics ⇒ bit-identical
replicas, no weight broadcast).  torch is used for device memory, the stream and the collectives only.
"""
from __future__ import annotations

import numpy as np
# torch is imported HERE, not lazily: the torch wheel brings its own HIP runtime, and a process that has already initialised the
# system's one through libeirgrid_hip.so (an Engine created before the first `import torch`) finds "No HIP GPUs" in torch's —
# measured on the GPU box.  Importing this module before creating an Engine puts torch's runtime first, where both share it.
import torch  # noqa: F401

from . import _native as N
from .engine import ActionWeights, Engine, apply_packet, apply_reduced


def shard_range(total: int, rank: int, world_size: int):
    """Contiguous shard [first, first+count) of `total` episodes for `rank` (first ranks take the remainder)."""
    base, rem = divmod(total, world_size)
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


def pick_candidate(pairs):
    """pairs: iterable of (score, global_index) per rank (index < 0 = no candidate).  Highest score wins, ties go to the
    lowest global index.  Returns (rank, score, index) or None."""
    best = None
    for r, (s, i) in enumerate(pairs):
        if i < 0:
            continue
        if best is None or s > best[1] or (s == best[1] and i < best[2]):
            best = (r, float(s), int(i))
    return best


def pack_candidate(metrics, n_run, run_log, n_def, def_log) -> np.ndarray:
    """Flat uint8 payload of the best-candidate broadcast."""
    return np.concatenate([np.asarray(metrics, np.float64).view(np.uint8), np.asarray(n_run, np.int32).view(np.uint8),
                           np.asarray(n_def, np.int32).view(np.uint8), np.asarray(run_log, np.uint8), np.asarray(def_log, np.uint8)])


def unpack_candidate(buf: np.ndarray):
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    o = 0
    m = buf[o:o + 32].view(np.float64).copy(); o += 32
    nr = buf[o:o + 4 * N.YEARS].view(np.int32).copy(); o += 4 * N.YEARS
    nd = buf[o:o + 4 * N.YEARS].view(np.int32).copy(); o += 4 * N.YEARS
    rl = buf[o:o + N.RUN_CAP].copy(); o += N.RUN_CAP
    dl = buf[o:o + N.DEF_CAP].copy()
    return m, nr, rl, nd, dl


CANDIDATE_BYTES = 32 + 8 * N.YEARS + N.RUN_CAP + N.DEF_CAP   # payload without the (score, index) header


def parse_candidate(buf: np.ndarray):
    """Candidate record of the update packet (include/eirgrid_hip.h): returns (score, global index, candidate tuple)."""
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    score = float(buf[0:8].view(np.float64)[0]); index = int(buf[8:16].view(np.int64)[0])
    return score, index, unpack_candidate(buf[16:])


def exchange_packet_raw(packet, dist=None):
    """The whole per-update exchange on an update packet (torch uint8 tensor [PACKET_BYTES] on the device):
    ONE sum all-reduce of the int64 statistics part + an all-gather of the per-rank candidate records (3.3 KB each),
    then ONE copy to the host.  Returns (stats int64 ndarray [STATS_LEN], candidates uint8 ndarray [W, CANDIDATE_BYTES]),
    the inputs of eg_policy_apply_packet; every rank holds the same data."""
    import torch
    nstat = 8 * N.STATS_LEN
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        host = packet.cpu().numpy()
        return host[:nstat].view(np.int64).copy(), host[nstat:].reshape(1, N.CANDIDATE_BYTES)
    ws = dist.get_world_size()
    if dist.get_backend() == "gloo":          # CPU rehearsal of the same exchange (tests; no RCCL involved)
        host = packet.cpu()
        stats = host[:nstat].view(torch.int64).clone()
        stats[3] = 0                                            # slot 3 is a maximum (best-score key), not a sum: unused here
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
        mine = host[nstat:].clone()
        gathered = [torch.empty_like(mine) for _ in range(ws)]
        dist.all_gather(gathered, mine)
        return stats.numpy().copy(), torch.stack(gathered).numpy()
    stats = packet[:nstat].view(torch.int64)
    stats[3] = 0                                                # slot 3 is a maximum (best-score key), not a sum: unused here
    dist.all_reduce(stats, op=dist.ReduceOp.SUM)                # the one all-reduce of the update (RCCL over xGMI)
    both = torch.empty(nstat + ws * N.CANDIDATE_BYTES, dtype=torch.uint8, device=packet.device)
    dist.all_gather_into_tensor(both[nstat:], packet[nstat:].contiguous())
    both[:nstat] = packet[:nstat]
    host = both.cpu().numpy()                                   # one device-to-host copy
    return host[:nstat].view(np.int64).copy(), host[nstat:].reshape(ws, N.CANDIDATE_BYTES)


def exchange_packet(packet, dist=None):
    """exchange_packet_raw with the winning candidate already parsed: (stats ndarray, candidate tuple | None)."""
    stats, cands = exchange_packet_raw(packet, dist)
    parsed = [parse_candidate(cands[r]) for r in range(cands.shape[0])]
    win = pick_candidate([(p[0], p[1]) for p in parsed])
    return stats, (parsed[win[0]][2] if win is not None else None)


def exchange_update(stats, local_pair, local_payload_fn, dist=None, device=None):
    """Exchange for callers that hold the pieces separately (used by the CPU/gloo tests): `stats` torch int64 tensor
    [STATS_LEN] (summed in place), `local_pair` = (score, global index) of this rank's best episode (index -1 if none),
    `local_payload_fn()` → uint8 ndarray [CANDIDATE_BYTES - 16] of that episode."""
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        cand = unpack_candidate(local_payload_fn()) if local_pair[1] >= 0 else None
        return stats.cpu().numpy(), cand
    ws, rank = dist.get_world_size(), dist.get_rank()
    dist.all_reduce(stats, op=dist.ReduceOp.SUM)
    pair = torch.tensor([float(local_pair[0]), float(local_pair[1])], dtype=torch.float64, device=device)
    gathered = [torch.empty_like(pair) for _ in range(ws)]
    dist.all_gather(gathered, pair)
    pairs = [(float(g[0]), int(g[1])) for g in torch.stack(gathered).cpu()]
    win = pick_candidate(pairs)
    cand = None
    if win is not None:
        if rank == win[0]:
            payload = torch.from_numpy(local_payload_fn()).to(device)
        else:
            payload = torch.empty(CANDIDATE_BYTES, dtype=torch.uint8, device=device)
        dist.broadcast(payload, src=win[0])
        cand = unpack_candidate(payload.cpu().numpy())
    return stats.cpu().numpy(), cand


class BatchTrainer:
    """Rollout + batch update loop of one rank (the driver of configs 2-4).

    device_resident (default on one GPU and under RCCL): the policy is pushed to the device once and every step is
    enqueued without a host synchronisation — rollout with the statistics epilogue, best pick, (one all-gather,)
    k_apply_update, stalled tables; `weights` is refreshed by sync().  Otherwise (gloo rehearsals on the CPU side of
    the exchange) each step uploads the snapshot, copies the packet to the host and updates `weights` there.  Both
    leave the same policy behind, bit for bit (tests/test_gpu_update.py)."""

    def __init__(self, engine: Engine, weights: ActionWeights, episodes_per_rank: int, seed: int, rank: int = 0,
                 world_size: int = 1, dist=None, replay_fraction: float = 0.0, write_yearly: bool = True,
                 device_resident=None, force_collectives: bool = False):
        import torch
        self.torch, self.dist = torch, dist
        self.eng, self.w = engine, weights
        self.n, self.seed, self.rank, self.ws = episodes_per_rank, seed, rank, world_size
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.packet = torch.zeros(N.PACKET_BYTES, dtype=torch.uint8, device=self.device)
        self.replay_fraction = replay_fraction
        self.replay_period = max(1, int(round(1.0 / replay_fraction))) if replay_fraction > 0.0 else 0
        self.write_yearly = write_yearly
        self.step_index = 0
        self.pinned = False
        # force_collectives: run the all-gather even with one rank (exercises the RCCL path on one GPU)
        multi = dist is not None and (world_size > 1 or force_collectives)
        self.multi = multi
        if device_resident is None:
            device_resident = (not multi) or dist.get_backend() == "nccl"
        self.device_resident = bool(device_resident)
        self._history0 = int(weights.get("improvement_history_len"))
        self._improvements_host = 0
        if self.device_resident:
            if multi:
                self.gathered = torch.zeros(world_size * N.PACKET_BYTES, dtype=torch.uint8, device=self.device)
            self.eng.push(weights, write_yearly=write_yearly)

    @property
    def improvements(self) -> int:
        """New best strategies since construction (device-resident mode: as of the last sync())."""
        if self.device_resident:
            return int(self.w.get("improvement_history_len")) - self._history0
        return self._improvements_host

    def failed_episodes(self) -> int:
        """Episodes (of all ranks) that ended with a status other than EG_EP_OK since the trainer started, as counted by the
        batch updates (capacity overflows of replay-doubled lists).  Synchronises."""
        self.sync()
        return int(self.w.get("failed_episodes"))

    def pin_policy(self):
        """From now on every step starts from the policy as it is on the device at this moment (eg_policy_hold / eg_policy_rewind):
        the update of a step runs in full, the next step does not build on it.  bench.py measures that way: every batch is
        then the same work on any number of GPUs, where the free-running loop changes what a replay episode costs (Q15)."""
        if not self.device_resident:
            raise RuntimeError("pin_policy needs the device-resident mode")
        self.eng.hold()
        self.pinned = True

    def sync(self):
        """Wait for the enqueued steps and bring `weights` up to date (device-resident mode)."""
        if self.device_resident:
            self.eng.pull(self.w)
        else:
            self.eng.sync()

    def step(self):
        """One pass of the hot path: rollout of this rank's shard, update statistics, the exchange, the update.
        Returns whether a new best strategy was installed (None in device-resident mode: nothing is read back)."""
        total = self.n * self.ws
        first = self.step_index * total + self.rank * self.n           # global episode index of this shard
        noise = self.seed + self.step_index
        nstat = 8 * N.STATS_LEN
        if self.device_resident:
            if self.pinned:
                self.eng.rewind()
            if not self.multi:
                self.eng.device_step(self.seed, first, self.n, self.replay_period, noise)
            else:
                self.eng.device_rollout(self.seed, first, self.n, self.replay_period, self.packet.data_ptr())
                # the one collective of the update (RCCL over xGMI): every rank receives every rank's 37 008-byte packet (N.PACKET_BYTES); the
                # statistics are integers, so summing them inside k_apply_update is the all-reduce
                if self.dist.get_backend() == "gloo":      # rehearsal without RCCL: the same exchange through the host
                    mine = self.packet.cpu()
                    parts = [self.torch.empty_like(mine) for _ in range(self.ws)]
                    self.dist.all_gather(parts, mine)
                    self.gathered.copy_(self.torch.cat(parts))
                else:
                    self.dist.all_gather_into_tensor(self.gathered, self.packet)
                self.eng.device_apply(self.gathered.data_ptr(), self.ws, self.packet.data_ptr(), noise)
            self.step_index += 1
            return None
        mask = None
        if self.replay_period > 0 and self.w.get("has_best_actions") == 1:
            mask = ((np.arange(first, first + self.n) % self.replay_period) == 0).astype(np.uint8)
        if not self.multi:      # one GPU: the whole step is one library call
            improved = self.eng.train_step(self.w, self.seed, first, self.n, mask, noise_seed=noise, write_yearly=self.write_yearly)
        else:
            self.eng.upload_snapshot(self.w, write_yearly=self.write_yearly)
            self.eng.launch_update(self.seed, first, self.n, self.packet.data_ptr(), mask)   # rollout + stats + best pick
            stats, cands = exchange_packet_raw(self.packet, self.dist)                        # all-reduce + one D2H copy
            improved = apply_packet(self.w, stats, cands, noise_seed=noise)
        self._improvements_host += int(improved)
        self.step_index += 1
        return improved
