"""What the compiler makes of the rollout kernels (no GPU needed: device code only, -Rpass-analysis=kernel-resource-usage).

Two properties of the throughput kernels are performance contracts that a source edit or a compiler update can silently break
(DESIGN.md §2.2 "The long-replay variant at four waves per SIMD", §7): they use NO scratch memory — a lean kernel with any scratch
lost 7-11 % beside the long-replay grid while losing 1 % alone, measured twice —, and the long-replay variant fits the 128 vector
registers of four waves per SIMD, which rests on the register budget its field code inherits from a kernel that is never launched
(k_heavy_register_budget) and on the aggregates parked in LDS around those calls."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_throughput_kernels_keep_their_register_and_scratch_budget():
    out = subprocess.run(["bash", os.path.join(ROOT, "scripts", "kernel_resources.sh")], capture_output=True, text=True, timeout=900).stdout
    rows = {}
    for line in out.splitlines():
        m = re.match(r"k_rolloutILi(\d)ELi(\d)E.*?VGPRs: (\d+) ScratchSize \[bytes/lane\]: (\d+).*?LDS Size \[bytes/block\]: (\d+)", line)
        if m:
            rows[(int(m.group(1)), int(m.group(2)))] = tuple(int(m.group(k)) for k in (3, 4, 5))
    assert set(rows) == {(h, k) for h in (0, 1) for k in (0, 1, 2)}, out
    for kind in (0, 1, 2):      # one wave per episode: lean, short-replay, long-replay
        vgprs, scratch, lds = rows[(0, kind)]
        assert vgprs <= 128, (kind, vgprs)          # four waves per SIMD
        assert scratch == 0, (kind, scratch)
        assert lds <= (9 if kind == 2 else 7) * 1280, (kind, lds)      # LDS granules: sixteen workgroups per CU beside each other
    # small-batch kernels (episode wave + helper wave): three waves per SIMD, no scratch on the path configs[1] runs
    assert rows[(1, 0)][0] <= 168 and rows[(1, 0)][1] == 0, rows[(1, 0)]
