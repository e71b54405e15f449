"""A plain loop of lean batches (16 384 sampled episodes, no replay, no update) for profilers: python scripts/lean_only.py [batches=20]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eirgrid_amd import synthetic_world
from eirgrid_amd.engine import ActionWeights, Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
eng = Engine(synthetic_world()); w = ActionWeights()
eng.upload_snapshot(w)
for i in range(n):
    eng.launch(12345, i * 16384, 16384)
eng.sync()
ms, k = eng.timing_read()
print(f"lean grid {ms / k:.3f} ms")
