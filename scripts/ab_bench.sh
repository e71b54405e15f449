#!/bin/bash
# A/B on the GPU box at the headline workload (bench.py defaults: 16 384 episodes, every 10th a replay, sustained state): interleaved
# runs for the shipped library and the named ab builds (make -C eirgrid_amd/csrc ab AB=<name> ABFLAGS=<flags>; or any other build of
# the same ABI copied to eirgrid_amd/libeirgrid_hip_ab_<name>.so).  AB_ARGS: further bench.py arguments, e.g. AB_ARGS=--seeded
#   bash scripts/ab_bench.sh prio3 prio1
for rep in 1 2 3; do
  for v in "" "$@"; do
    if [ -z "$v" ]; then unset EIRGRID_LIB; name=shipped; else export EIRGRID_LIB=eirgrid_amd/libeirgrid_hip_ab_$v.so; name=$v; fi
    python bench.py --no-cpu-baseline --no-config1 $AB_ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', '$AB_ARGS', 'ms/batch %.3f' % d['ms_per_batch'], 'kernel %.3f' % d['roofline']['avg_kernel_ms'], '%.3f M eps/s' % (d['value']/1e6))"
  done
done
