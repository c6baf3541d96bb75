"""Summarise rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; one counter per pass) into profiles/pmc_summary.json.

    python tools/pmc_summary.py gpurun_out/pmc_fetch/fetch_counter_collection.csv \
                                gpurun_out/pmc_write/write_counter_collection.csv profiles/pmc_summary.json

Units / corrections (MI355X_MICROARCH.md, HBM section): the counters are in KiB; on gfx950 FETCH_SIZE reports half
of the bytes of a wide coalesced read, so hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 per launch.  The raw
values are kept next to the corrected one.  Infinity-Cache hits are counted, so this is an upper bound on DRAM bytes.
"""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    m = re.search(r"(spf_[a-z0-9_]+)", name)
    return m.group(1) if m else name.split("(")[0][:60]


def load(path, counter):
    acc = defaultdict(lambda: [0.0, 0])
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            k = short(row["Kernel_Name"])
            acc[k][0] += float(row["Counter_Value"])
            acc[k][1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}


def main():
    fetch = load(sys.argv[1], "FETCH_SIZE")
    write = load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("spf_"):
            continue
        f, nf = fetch.get(k, (0.0, 0))
        w, nw = write.get(k, (0.0, 0))
        out[k] = {"launches": [nf, nw], "FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1),
                  "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in out.items():
        print(f"{k:32s} fetch {v['FETCH_SIZE_KiB']/1024:9.1f} MiB  write {v['WRITE_SIZE_KiB']/1024:9.1f} MiB  "
              f"corrected {v['hbm_bytes_per_launch']/1e6:9.1f} MB/launch")


if __name__ == "__main__":
    main()
