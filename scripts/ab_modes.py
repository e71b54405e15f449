"""Kernel time of a fresh-policy batch in three configurations: plain, with yearly rows, with yearly rows + the statistics epilogue."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eirgrid_amd import synthetic_world, _native as N
from eirgrid_amd.engine import ActionWeights, Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
L = int(sys.argv[2]) if len(sys.argv) > 2 else 200
eng = Engine(synthetic_world()); pol = ActionWeights()
packet = torch.zeros(N.PACKET_BYTES, dtype=torch.uint8, device="cuda")
for name, yearly, stats in (("plain", False, False), ("yearly rows", True, False), ("yearly rows + statistics epilogue", True, True)):
    eng.upload_snapshot(pol, write_yearly=yearly)
    for k in range(10): eng.launch(12345, k * B, B)
    eng.sync(); eng.timing_reset()
    for k in range(L):
        if stats: eng.launch_update(777, k * B, B, packet.data_ptr(), None)
        else: eng.launch(777, k * B, B)
    eng.sync()
    ms, n = eng.timing_read()
    print(f"B={B} {name:36s} kernel ms {ms / n:.4f}")
