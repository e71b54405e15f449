#!/bin/bash
# A/B on the GPU box: interleaved runs of scripts/ab_kernel.py for the shipped library and eirgrid_amd/libeirgrid_hip_ab_$1.so
#   bash scripts/ab_pair.sh noheavy [launches]
L=${2:-200}
for B in 1024 16384; do
  for rep in 1 2; do
    python scripts/ab_kernel.py $B $L
    EIRGRID_LIB=eirgrid_amd/libeirgrid_hip_ab_$1.so python scripts/ab_kernel.py $B $L
  done
done
