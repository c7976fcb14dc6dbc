#!/usr/bin/python3
"""Per-kernel averages of two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, collected in separate runs as the MI355X
guide prescribes) -> a table on stdout and gpurun_out/pmc_per_kernel.json.  Counter units are KiB (raw, uncorrected)."""
import collections
import csv
import glob
import json
import os
import sys


def per_kernel(d, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter:
                k = r.get("Kernel_Name", "")
                k = k.split("(")[0]
                a = acc[k]
                a[0] += float(r["Counter_Value"])
                a[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items() if v[1]}


def main():
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(k, (None, 0))
        w, nw = write.get(k, (None, 0))
        out[k] = {"FETCH_SIZE_KiB_raw": f, "launches_fetch": nf, "WRITE_SIZE_KiB_raw": w, "launches_write": nw}
        print("%-60s fetch %12s KiB (%d)  write %12s KiB (%d)" % (k[:60], "%.0f" % f if f is not None else "-", nf,
                                                                   "%.0f" % w if w is not None else "-", nw))
    json.dump(out, open(sys.argv[3] if len(sys.argv) > 3 else os.path.join("gpurun_out", "pmc_per_kernel.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
