#!/usr/bin/python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as the MI355X guide's
HBM section prescribes) into per-launch HBM bytes for one kernel and write profiles/pmc_traffic.json.

  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py ...
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py ...
  python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write msm_accum_kernel msm_2p20 [read_factor] [note]

Units/corrections (MI355X_MICROARCH.md, HBM): the counters are in KiB; on gfx950 FETCH_SIZE
reports half the bytes of a wide coalesced read, so the read side is doubled; WRITE_SIZE is exact
for 16-B-per-lane stores.  The guide calls other access shapes uncalibrated and asks for a
calibration on a known byte count: pass read_factor = 1 where that calibration says the raw
counter is already exact (say why in `note`)."""
import csv
import glob
import json
import os
import sys


def per_launch(d, counter, kernel):
    tot, n = 0.0, 0
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter and kernel in r.get("Kernel_Name", ""):
                tot += float(r["Counter_Value"])
                n += 1
    return (tot / n if n else None), n


def main():
    fetch_dir, write_dir, kernel, workload = sys.argv[1:5]
    factor = float(sys.argv[5]) if len(sys.argv) > 5 else 2.0
    note = sys.argv[6] if len(sys.argv) > 6 else ""

    f, nf = per_launch(fetch_dir, "FETCH_SIZE", kernel)
    w, nw = per_launch(write_dir, "WRITE_SIZE", kernel)
    out_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
    try:
        allv = json.load(open(out_path))
    except (OSError, ValueError):
        allv = {}
    rec = {"kernel": kernel, "launches_fetch": nf, "launches_write": nw,
           "FETCH_SIZE_KiB_raw": f, "WRITE_SIZE_KiB_raw": w,
           "read_factor": factor, "note": note,
           "read_bytes": None if f is None else f * 1024 * factor,
           "write_bytes": None if w is None else w * 1024,
           "hbm_bytes_per_launch": None if (f is None or w is None) else f * 1024 * factor + w * 1024}
    allv[workload] = rec
    json.dump(allv, open(out_path, "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
