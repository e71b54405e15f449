#!/bin/bash
# Run on the GPU box: configs[2] in the sustained state with the per-episode replay kernel (eg_replay_solo.h) off and on, then the
# kernel trace of the latter.   bash scripts/solo_probe.sh <outdir>
set -eo pipefail
O=${1:-gpurun_out/solo}
mkdir -p $O
export TMPDIR=/tmp
CG="--steps 20 --warmup 5 --no-cpu-baseline --no-config1"
EIRGRID_REPLAY_SOLO=0 python bench.py $CG > $O/bench_classic.json 2> $O/bench_classic.err
python bench.py $CG > $O/bench_solo.json 2> $O/bench_solo.err
EIRGRID_REPLAY_SOLO=0 python bench.py $CG > $O/bench_classic2.json 2>> $O/bench_classic.err
python bench.py $CG > $O/bench_solo2.json 2>> $O/bench_solo.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py $CG > $O/prof.log 2>&1
cp $(ls $O/prof/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
python - <<PY
import json
for n in ("classic", "solo", "classic2", "solo2"):
    l = json.loads(open("$O/bench_%s.json" % n).read().strip().splitlines()[-1])
    print(n, round(l["value"]), "eps/s", round(l["ms_per_batch"], 4), "ms/batch;", l["config"]["per_episode_replay_kernel"][:14])
PY
head -12 $O/kernel_stats.csv | cut -c1-200
rm -rf $O/prof
