"""Diagnostic (-DEG_STAMPS build): wall-clock phases of k_apply_update (100 MHz counter), averaged over training steps.
   make -C eirgrid_amd/csrc stamps && EIRGRID_LIB=eirgrid_amd/libeirgrid_hip_stamps.so python scripts/apply_phases.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from eirgrid_amd import synthetic_world, _native as N
from eirgrid_amd.engine import ActionWeights, Engine
eng = Engine(synthetic_world()); eng.push(ActionWeights())
packet = torch.zeros(N.PACKET_BYTES, dtype=torch.uint8, device="cuda")
acc = np.zeros(4); cnt = 0
for step in range(80):
    eng.device_rollout(1, step * 1024, 1024, 0, packet.data_ptr())
    eng.device_apply(packet.data_ptr(), 1, packet.data_ptr(), step)
    eng.sync()
    t = packet[:64].cpu().numpy().view(np.int64)[4:8].astype(np.float64)
    packet[32:64] = 0
    if step >= 10: acc += t; cnt += 1
names = ["state + winner + noise key/blocks", "contrast step (main table) + best-strategy bookkeeping", "list copies / deficit contrast", "row sums, derive_state, zeroing"]
for n, v in zip(names, acc / cnt / 100.0): print(f"  {n:42s} {v:6.2f} us")
print(f"  total inside the kernel {acc.sum() / cnt / 100.0:.2f} us")
