"""Summarise rocprofv3 --pmc SQ-counter passes (tools/pmc_passes.sh) into profiles/sq_summary.json: per kernel, the
average of every collected counter per launch, plus the ratios the docs quote.

    python tools/sq_summary.py <counter_collection.csv> [...] profiles/sq_summary.json

`SQ_INSTS_VALU_per_launch` is what bench.py's `roofline.valu` reads: wave-level VALU instructions per launch; a wave64
VALU instruction occupies its SIMD for 4 cycles, so the VALU-issue roofline is
insts * 4 / (256 CUs * 4 SIMDs * clock * launch time)."""
import csv
import json
import re
import sys
from collections import defaultdict

files, out = sys.argv[1:-1], sys.argv[-1]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for path in files:
    for row in csv.DictReader(open(path)):
        m = re.search(r"(spf_[a-z0-9_]+)", row["Kernel_Name"])
        if not m:
            continue
        a = acc[m.group(1)][row["Counter_Name"]]
        a[0] += float(row["Counter_Value"])
        a[1] += 1
res = {}
for k, cs in sorted(acc.items()):
    d = {f"{c}_per_launch": v[0] / v[1] for c, v in sorted(cs.items())}
    g = lambda n: d.get(n + "_per_launch")
    if g("SQ_LDS_BANK_CONFLICT") is not None and g("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_frac"] = round(g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE"), 4)
    if g("SQ_ACTIVE_INST_VALU") is not None and g("SQ_BUSY_CU_CYCLES"):
        d["valu_busy_frac"] = round(g("SQ_ACTIVE_INST_VALU") / g("SQ_BUSY_CU_CYCLES"), 4)
    if g("SQ_THREAD_CYCLES_VALU") is not None and g("SQ_INSTS_VALU"):
        # the counter adds the active lanes of every VALU instruction: / instructions = average active lanes (of 64)
        d["active_lanes_per_valu_inst"] = round(g("SQ_THREAD_CYCLES_VALU") / g("SQ_INSTS_VALU"), 1)
    res[k] = {kk: (round(vv, 1) if isinstance(vv, float) and vv > 10 else vv) for kk, vv in d.items()}
json.dump(res, open(out, "w"), indent=1)
for k, d in res.items():
    print(k, {kk: vv for kk, vv in d.items() if not kk.endswith("_per_launch")},
          "INSTS_VALU", d.get("SQ_INSTS_VALU_per_launch"))
