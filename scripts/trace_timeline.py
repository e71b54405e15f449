"""Timeline of one batch from a rocprofv3 kernel trace (run on the GPU box):
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-config1 --batches-per-step 2
  python scripts/trace_timeline.py gpurun_out/tl
Prints, for the last batches, when each kernel started and ended relative to the batch's first kernel (microseconds)."""
import csv, glob, sys
path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Stream_Id", r.get("Queue_Id", "?")), r.get("Grid_Size", "?"), r.get("VGPR_Count", r.get("Arch_VGPR_Count", "?"))) for r in rows]
tail = ev[-24:]
t0 = tail[0][0]
for s, e, name, q, grid, vg in tail:
    print(f"{(s - t0) / 1e3:10.1f} .. {(e - t0) / 1e3:10.1f} us  ({(e - s) / 1e3:8.1f})  q{q} grid {grid} vgpr {vg}  {name}")
