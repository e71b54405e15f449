#!/bin/bash
# Run on the GPU box: SQ counters of the lean grid alone (scripts/ab_kernel.py: sampled episodes, fresh policy, no statistics epilogue) at
# 16 384 and 131 072 episodes per launch — how busy the vector ALUs are while every SIMD has its four waves, against the batch size's
# own ramp and tail.   bash scripts/lean_occupancy.sh <outdir>
set -eo pipefail
O=${1:-gpurun_out/lean_occ}
mkdir -p $O
export TMPDIR=/tmp EIRGRID_HELPER_WAVES=0
SQ="SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"
for b in 16384 131072; do
  rocprofv3 --pmc $SQ -d $O/sq_$b --output-format csv -- python3 scripts/ab_kernel.py $b 6 > $O/sq_$b.log 2>&1
done
python - <<PY
import csv, glob, collections
for b in (16384, 131072):
    rows = collections.defaultdict(dict)
    for path in glob.glob("$O/sq_%d/**/*counter_collection.csv" % b, recursive=True):
        for r in csv.DictReader(open(path)):
            if "k_rollout" in r["Kernel_Name"] and int(r["Grid_Size"]) == b * 64:
                d = rows[int(r["Dispatch_Id"])]
                d[r["Counter_Name"]] = float(r["Counter_Value"]); d["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    ids = sorted(rows)[-8:]
    avg = {k: sum(rows[i][k] for i in ids) / len(ids) for k in rows[ids[0]]}
    simd_quads = 1024 * avg["ns"] * 1e-9 * 2.4e9 / 4
    print(b, "episodes: kernel %.3f ms under counters; VALU busy %.3f of the span; waves per SIMD on average %.2f (of 4); VALU busy per resident wave-slot %.3f; "
          "VALU %.0f + SALU %.0f instructions per episode" % (avg["ns"] * 1e-6, avg["SQ_ACTIVE_INST_VALU"] / simd_quads, avg["SQ_WAVE_CYCLES"] / simd_quads,
          avg["SQ_ACTIVE_INST_VALU"] * 4 / avg["SQ_WAVE_CYCLES"], avg["SQ_INSTS_VALU"] / b, avg["SQ_INSTS_SALU"] / b))
PY
rm -rf $O/sq_16384 $O/sq_131072
