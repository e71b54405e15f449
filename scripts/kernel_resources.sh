#!/bin/bash
# Registers, spills, LDS and occupancy of every kernel of eg_rollout.hip as the compiler reports them
# (-Rpass-analysis=kernel-resource-usage; device code only, nothing is linked or installed).  Runs without a GPU.
# The file is compiled twice (csrc/Makefile): eg_rollout.o, and eg_rollout_tp.o — the throughput kernels, with KTP's options.
#   scripts/kernel_resources.sh [extra hipcc flags] > profiles/rNN_kernel_resources.txt
set -e
cd "$(dirname "$0")/../eirgrid_amd/csrc"
KTP=$(sed -n 's/^KTP = //p' Makefile)
for tu in "" "$KTP"; do
  echo "# eg_rollout.hip ${tu:-(eg_rollout.o)}"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -I../../include -I. \
    -mllvm -amdgpu-sched-strategy=iterative-ilp --offload-device-only -Rpass-analysis=kernel-resource-usage $tu "$@" \
    -c eg_rollout.hip -o /tmp/eg_rollout_resources.o 2>&1 |
    grep -E "Function Name|VGPRs:|Spill|LDS Size|Occupancy|ScratchSize|SGPRs:" | paste - - - - - - - - |
    sed -E 's/eg_rollout.hip:[0-9]+:[0-9]+: remark: //g; s/\[-Rpass-analysis=kernel-resource-usage\]//g; s/[ \t]+/ /g; s/Function Name: _ZN2eg(12_GLOBAL__N_1)?[0-9]+//; s/ENS_[^ ]*//'
done
