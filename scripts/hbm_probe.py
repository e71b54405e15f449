"""HBM bandwidth actually reachable on the box (report the measured peak next to the datasheet's 8 TB/s).
   python scripts/hbm_probe.py > gpurun_out/hbm_probe.json
Plain device-to-device copy and a triad (a = b + s * c) over buffers far larger than the 256 MB Infinity Cache, torch
ops only (this is a property of the machine, not of this repository's kernels)."""
import json, time
import torch
dev = torch.device("cuda", 0)
n = 1 << 30                                  # 1 Gi float32 = 4 GiB per buffer
a = torch.empty(n, dtype=torch.float32, device=dev); b = torch.ones_like(a); c = torch.ones_like(a)
def timed(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps
t_copy = timed(lambda: a.copy_(b))
t_triad = timed(lambda: torch.add(b, c, alpha=3.0, out=a))
out = {"device": torch.cuda.get_device_name(0), "buffer_bytes": 4 * n,
       "copy_GBps": 2 * 4 * n / t_copy / 1e9, "triad_GBps": 3 * 4 * n / t_triad / 1e9,
       "datasheet_GBps": 8000.0, "note": "read + write bytes over wall time; torch device copy / torch.add(out=)"}
print(json.dumps(out))
