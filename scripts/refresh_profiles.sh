#!/bin/bash
# Run on the GPU box (gpurun): regenerates everything under profiles/ for tag $1 into gpurun_out/.
set -eo pipefail
tag=${1:-r01d}
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
python bench.py > $O/${tag}_bench_b1024.json 2> $O/bench_default.err
python bench.py --no-cpu-baseline --episodes 16384 > $O/${tag}_bench_b16384.json 2>> $O/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${tag} -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/prof_${tag}.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_${tag}_f1k --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/pmc_${tag}.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_${tag}_w1k --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline >> $O/pmc_${tag}.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_${tag}_f16k --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --episodes 16384 >> $O/pmc_${tag}.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_${tag}_w16k --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --episodes 16384 >> $O/pmc_${tag}.log 2>&1
echo refreshed $tag
