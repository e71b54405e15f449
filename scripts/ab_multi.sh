#!/bin/bash
# A/B on the GPU box: interleaved runs of scripts/ab_kernel.py at B episodes for the shipped library and the named ab builds
#   bash scripts/ab_multi.sh 16384 200 drglobal occ4
B=$1; L=$2; shift 2
for rep in 1 2; do
  python scripts/ab_kernel.py $B $L
  for v in "$@"; do EIRGRID_LIB=eirgrid_amd/libeirgrid_hip_ab_$v.so python scripts/ab_kernel.py $B $L; done
done
