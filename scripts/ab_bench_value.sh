#!/bin/bash
# A/B on the GPU box: the headline batch (bench.py, sustained state of configs[2], every episode on its own) for the shipped library and
# the named `make ab` builds, interleaved, twice.   bash scripts/ab_bench_value.sh solo6 ...
CG="--steps 20 --warmup 5 --no-cpu-baseline --no-config1"
for rep in 1 2; do
  for v in "" "$@"; do
    if [ -z "$v" ]; then lib=eirgrid_amd/libeirgrid_hip.so; else lib=eirgrid_amd/libeirgrid_hip_ab_$v.so; fi
    EIRGRID_LIB=$lib python bench.py $CG 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('${v:-shipped}', round(l['value']), 'eps/s', round(l['ms_per_batch'],4), 'ms/batch; hoisted', round(l['config2_replay_hoisted']['value']) if 'config2_replay_hoisted' in l else '')"
  done
done
