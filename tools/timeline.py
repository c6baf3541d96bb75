"""development: print the kernel timeline of the last bench step from a rocprofv3 --kernel-trace CSV directory.
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr -o t -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline
    python tools/timeline.py gpurun_out/tr"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
last = [i for i, r in enumerate(rows) if "camera_fwd" in r["Kernel_Name"]][-1]
t0 = int(rows[last]["Start_Timestamp"])
prev_end = t0
for r in rows[last:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f  gap %5.1f  dur %7.1f  %s" % ((s - t0) / 1000, (s - prev_end) / 1000, (e - s) / 1000, r["Kernel_Name"][:80]))
    prev_end = e
