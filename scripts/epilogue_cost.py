"""What the fused statistics epilogue of k_rollout costs: B sampled episodes against a policy that has a best strategy (so the
epilogue has work), launched without and with the statistics.   python scripts/epilogue_cost.py [B] [launches]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eirgrid_amd import synthetic_world, _native as N
from eirgrid_amd.engine import ActionWeights, Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
L = int(sys.argv[2]) if len(sys.argv) > 2 else 200
eng = Engine(synthetic_world()); pol = ActionWeights()
first = eng.run_iteration(0, pol, False, 12345)
pol.apply_episode(first.metrics[0], first.n_run[0], first.run_log[0, :first.n_run[0].sum()], first.n_def[0], first.def_log[0, :first.n_def[0].sum()])
eng.upload_snapshot(pol, write_yearly=True)
packet = torch.zeros(N.PACKET_BYTES, dtype=torch.uint8, device="cuda")
for mode in ("plain", "stats", "plain", "stats"):
    for k in range(10):
        eng.launch(12345, k * B, B) if mode == "plain" else eng.launch_update(12345, k * B, B, packet.data_ptr())
    eng.sync(); eng.timing_reset()
    for k in range(L):
        eng.launch(777, k * B, B) if mode == "plain" else eng.launch_update(777, k * B, B, packet.data_ptr())
    eng.sync()
    ms, n = eng.timing_read()
    print(f"B={B} {mode}: k_rollout {ms / n:.4f} ms")
