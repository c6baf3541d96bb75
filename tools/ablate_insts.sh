#!/bin/bash
# development: VALU instruction count of spf_render_bwd_lists_kernel per ablation cut (profiling build in _C_abl)
export TMPDIR=/tmp SPF_LIB_DIR=_C_abl SPF_HIPCC_EXTRA=-DSPF_ABLATE
OUT=$PWD/gpurun_out/ablwork; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT -o a -- python -u tools/ablate_bwd.py > $OUT/log.txt 2>&1
python - "$1" <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/ablwork/**/*counter_collection.csv", recursive=True)[0]
import sys
which = "spf_render_fwd_lists" if len(sys.argv) > 1 and sys.argv[1] == "fwd" else "spf_render_bwd_lists"
rows = [r for r in csv.DictReader(open(f)) if which in r["Kernel_Name"]]
by = collections.defaultdict(dict)
for r in rows:
    by[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(by)
per = len(ids) // 13
for c in range(13):
    sel = ids[c * per:(c + 1) * per]
    out = {k: sum(by[i][k] for i in sel) / len(sel) for k in by[sel[0]]}
    print("cut", c, {k: round(v / 1e6, 2) for k, v in out.items()})
PY
grep cut $OUT/log.txt
rm -rf $OUT
