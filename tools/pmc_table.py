"""Print per-kernel averages of every counter in rocprofv3 counter_collection CSVs.
    python tools/pmc_table.py <csv> [<csv> ...] [--filter spf_render]"""
import csv
import re
import sys
from collections import defaultdict

files = [a for a in sys.argv[1:] if not a.startswith("--")]
flt = None
if "--filter" in sys.argv:
    flt = sys.argv[sys.argv.index("--filter") + 1]
    files = [f for f in files if f != flt]
for path in files:
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for row in csv.DictReader(open(path)):
        m = re.search(r"(spf_[a-z0-9_]+)", row["Kernel_Name"])
        if not m:
            continue
        k = m.group(1)
        if flt and flt not in k:
            continue
        a = acc[k][row["Counter_Name"]]
        a[0] += float(row["Counter_Value"])
        a[1] += 1
    print("==", path)
    for k, cs in acc.items():
        print(f"  {k}")
        for cn, (tot, n) in sorted(cs.items()):
            print(f"      {cn:26s} {tot / n:16.0f}   ({n} launches)")
