#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, collected separately with --kernel-trace only,
as /opt/skills/guides/MI355X_MICROARCH.md 'HBM' prescribes) into profiles/traffic_<workload>.json.

    python tools/make_traffic.py cfg4 <fetch_counter_collection.csv> <write_counter_collection.csv> [kernel substring]

Corrections applied (same guide): counters are in KiB; on gfx950 FETCH_SIZE reports exactly half the bytes of
a wide (16 B/lane) coalesced read stream, so it is doubled; WRITE_SIZE is exact for 16 B/lane stores.
"""
import csv
import json
import os
import sys


def per_launch(path, kernel_sub):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if kernel_sub in r["Kernel_Name"]]
    assert vals, "no rows for kernel %r in %s" % (kernel_sub, path)
    vals.sort()
    return vals[len(vals) // 2], len(vals)


def main():
    workload, fetch_csv, write_csv = sys.argv[1:4]
    kernel_sub = sys.argv[4] if len(sys.argv) > 4 else "k_reproject_dma"
    fetch_kib, nf = per_launch(fetch_csv, kernel_sub)
    write_kib, nw = per_launch(write_csv, kernel_sub)
    fetch_bytes = 2.0 * fetch_kib * 1024.0
    write_bytes = write_kib * 1024.0
    out = {
        "workload": workload, "kernel": kernel_sub, "launches_sampled": [nf, nw],
        "FETCH_SIZE_KiB_raw": fetch_kib, "WRITE_SIZE_KiB_raw": write_kib,
        "fetch_bytes_corrected_x2": fetch_bytes, "write_bytes": write_bytes,
        "hbm_bytes_per_launch": fetch_bytes + write_bytes,
        "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; KiB -> bytes; "
                  "FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B for 16 B/lane streams); median over launches",
    }
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "profiles", "traffic_%s.json" % workload), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
