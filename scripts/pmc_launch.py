"""Plain k_rollout launches for counter collection: python scripts/pmc_launch.py B [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eirgrid_amd import synthetic_world
from eirgrid_amd.engine import ActionWeights, Engine
B = int(sys.argv[1]); reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
eng = Engine(synthetic_world()); pol = ActionWeights(); eng.upload_snapshot(pol)
for k in range(reps):
    eng.launch(12345, k * B, B)
eng.sync()
print("done", B, reps)
