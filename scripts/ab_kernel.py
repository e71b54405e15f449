"""A/B helper: average k_rollout time over many launches of B episodes (fresh policy), for the library in EIRGRID_LIB.
   EIRGRID_LIB=eirgrid_amd/libeirgrid_hip_base.so python scripts/ab_kernel.py [B] [launches]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eirgrid_amd import synthetic_world
from eirgrid_amd.engine import ActionWeights, Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
L = int(sys.argv[2]) if len(sys.argv) > 2 else 200
eng = Engine(synthetic_world()); pol = ActionWeights()
eng.upload_snapshot(pol, write_yearly=False)
for k in range(10): eng.launch(12345, k * B, B)
eng.sync()
out = []
for rep in range(3):
    eng.timing_reset()
    for k in range(L): eng.launch(777, (rep * L + k) * B, B)
    eng.sync()
    ms, n = eng.timing_read()
    out.append(ms / n)
print(f"{os.path.basename(os.environ.get('EIRGRID_LIB', 'libeirgrid_hip.so'))}: B={B} kernel ms " + " ".join(f"{v:.4f}" for v in out))
