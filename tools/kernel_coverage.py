#!/usr/bin/env python3
"""Which kernel instantiations of libpixell_hip.so do the GPU tests launch?

    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cov -o cov -- python3 -m pytest tests -m gpu -q
    make -C pixell.jl_amd/csrc asm
    python tools/kernel_coverage.py gpurun_out/cov/cov_kernel_stats.csv

Compares the kernel names in the rocprofv3 statistics with the kernels in the ISA listing of the library."""
import csv
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def norm(s):
    return re.sub(r"\s+", "", re.sub(r"\(.*$", "", s.replace("void ", "")))


def main():
    stats = sys.argv[1]
    rows = list(csv.DictReader(open(stats)))
    calls = {norm(r["Name"]): int(r["Calls"]) for r in rows}
    txt = open(os.path.join(ROOT, "pixell.jl_amd", "csrc", "pxl_kernels.gfx950.s")).read()
    syms = sorted(set(re.findall(r"\.amdhsa_kernel (\S+)", txt)))
    dem = subprocess.run(["c++filt"] + syms, capture_output=True, text=True).stdout.splitlines()
    missing = [d for d in dem if norm(d) not in calls]
    print("%d kernels in the library, %d launched by the profiled run, %d never launched" % (len(dem), len(dem) - len(missing), len(missing)))
    for d in dem:
        print("%8s  %s" % (calls.get(norm(d), "-"), re.sub(r"\(.*$", "", d.replace("void ", ""))))
    sys.exit(1 if missing else 0)


if __name__ == "__main__":
    main()
