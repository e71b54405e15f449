#!/bin/bash
# Run on the GPU box (gpurun): regenerates everything under profiles/ for tag $1 into gpurun_out/<tag>_*.
#   bash scripts/refresh_profiles.sh r02a      then copy gpurun_out/r02a_* into profiles/
# Workloads: "16384r0.1grown" = BASELINE configs[2] in the state the training loop sustains (bench.py default: 228 generators per
# replay episode), "16384r0.1grownhoist" = the same with the replay hoist on (--replay-hoist), "16384r0.1" = the same batches from the
# seeded policy (--seeded), "1024" = configs[1] (--episodes 1024 --replay-fraction 0).
set -eo pipefail
tag=${1:-r02a}
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
CG="--steps 20 --warmup 5 --no-cpu-baseline --no-config1"
C2="$CG --seeded"
CH="$CG --replay-hoist"
C1="--steps 20 --warmup 5 --no-cpu-baseline --episodes 1024 --replay-fraction 0 --batches-per-step 8"
python bench.py > $O/${tag}_bench.json 2> $O/${tag}_bench.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${tag}_c2 -- python3 bench.py $C2 > $O/prof_${tag}.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${tag}_cg -- python3 bench.py $CG >> $O/prof_${tag}.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${tag}_c1 -- python3 bench.py $C1 >> $O/prof_${tag}.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${tag}_ch -- python3 bench.py $CH >> $O/prof_${tag}.log 2>&1
cp $(ls $O/prof_${tag}_ch/*/*kernel_stats.csv | head -1) $O/${tag}_bench_c2grownhoist_kernel_stats.csv
cp $(ls $O/prof_${tag}_c2/*/*kernel_stats.csv | head -1) $O/${tag}_bench_c2_kernel_stats.csv
cp $(ls $O/prof_${tag}_cg/*/*kernel_stats.csv | head -1) $O/${tag}_bench_c2grown_kernel_stats.csv
cp $(ls $O/prof_${tag}_c1/*/*kernel_stats.csv | head -1) $O/${tag}_bench_c1_kernel_stats.csv
echo "kernel stats done"
# counters: separate passes, nothing but --pmc (MI355X_MICROARCH.md, HBM / rocprofv3 section)
PG="--steps 4 --warmup 2 --no-cpu-baseline --no-config1 --batches-per-step 2"
P2="$PG --seeded"
PH="$PG --replay-hoist"
P1="--steps 4 --warmup 2 --no-cpu-baseline --episodes 1024 --replay-fraction 0 --batches-per-step 2"
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_${tag}_f_c2 --output-format csv -- python3 bench.py $P2 > $O/pmc_${tag}.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_${tag}_w_c2 --output-format csv -- python3 bench.py $P2 >> $O/pmc_${tag}.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_${tag}_f_cg --output-format csv -- python3 bench.py $PG >> $O/pmc_${tag}.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_${tag}_w_cg --output-format csv -- python3 bench.py $PG >> $O/pmc_${tag}.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_${tag}_f_ch --output-format csv -- python3 bench.py $PH >> $O/pmc_${tag}.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_${tag}_w_ch --output-format csv -- python3 bench.py $PH >> $O/pmc_${tag}.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_${tag}_f_c1 --output-format csv -- python3 bench.py $P1 >> $O/pmc_${tag}.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_${tag}_w_c1 --output-format csv -- python3 bench.py $P1 >> $O/pmc_${tag}.log 2>&1
echo "hbm counters done"
SQ="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY"
rocprofv3 --pmc $SQ -d $O/sq_${tag}_c2 --output-format csv -- python3 bench.py $P2 >> $O/pmc_${tag}.log 2>&1
rocprofv3 --pmc $SQ -d $O/sq_${tag}_cg --output-format csv -- python3 bench.py $PG >> $O/pmc_${tag}.log 2>&1
rocprofv3 --pmc $SQ -d $O/sq_${tag}_ch --output-format csv -- python3 bench.py $PH >> $O/pmc_${tag}.log 2>&1
rocprofv3 --pmc $SQ -d $O/sq_${tag}_c1 --output-format csv -- python3 bench.py $P1 >> $O/pmc_${tag}.log 2>&1
echo "sq counters done"
python scripts/pmc_summarize.py $tag $O 16384r0.1:$O/pmc_${tag}_f_c2:$O/pmc_${tag}_w_c2 16384r0.1grown:$O/pmc_${tag}_f_cg:$O/pmc_${tag}_w_cg 16384r0.1grownhoist:$O/pmc_${tag}_f_ch:$O/pmc_${tag}_w_ch 1024:$O/pmc_${tag}_f_c1:$O/pmc_${tag}_w_c1
python scripts/sq_summarize.py $tag $O 16384r0.1:$O/sq_${tag}_c2 16384r0.1grown:$O/sq_${tag}_cg 16384r0.1grownhoist:$O/sq_${tag}_ch 1024:$O/sq_${tag}_c1
rm -rf $O/prof_${tag}_c1 $O/prof_${tag}_c2 $O/prof_${tag}_cg $O/prof_${tag}_ch $O/pmc_${tag}_f_* $O/pmc_${tag}_w_* $O/sq_${tag}_c1 $O/sq_${tag}_c2 $O/sq_${tag}_cg $O/sq_${tag}_ch
echo refreshed $tag
