#!/usr/bin/env python3
"""Per-variant table of a tools/research/r04_first.sh collection (three interleaved variants of one kernel name):
    python tools/research/sq_by_variant.py gpurun_out/r04_first > profiles/r04_cfg3_sq_counters_by_variant.txt"""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
print("# k_reproject_dma<double,4,3> on cfg3 (placed maps: destination across a class boundary), SQ counters per launch, median of 5 launches per variant")
print("# collected by tools/research/r04_first.sh: one rocprofv3 --kernel-trace --pmc pass per group; variants interleaved in one process (tools/tune_reproject.py)")
print("# variant A = the product launch; B = flags=2 (everything but the stores); C = flags=64 (the tile's stores alone)")
names = {0: 'A full', 1: 'B no stores', 2: 'C stores only'}
tab = collections.OrderedDict()
for p in sorted(glob.glob(os.path.join(root, 'pmc*/runc/*_counter_collection.csv'))):
    rows = [r for r in csv.DictReader(open(p)) if 'k_reproject_dma' in r['Kernel_Name']]
    by = collections.OrderedDict()
    for r in rows:
        by.setdefault(int(r['Dispatch_Id']), {})[r['Counter_Name']] = float(r['Counter_Value'])
    ids = sorted(by)
    for v in range(3):
        sel = [by[i] for k, i in enumerate(ids) if k % 3 == v][1:]
        for n in sorted(sel[0]):
            tab.setdefault(n, {})[v] = sorted(s[n] for s in sel)[len(sel) // 2]
print("%-24s %16s %16s %16s" % ("counter", names[0], names[1], names[2]))
for n, v in tab.items():
    print("%-24s %16.0f %16.0f %16.0f" % (n, v[0], v[1], v[2]))
w = 57464
rows = w * 32
print()
print("# per output row of a wave (57 464 waves x 32 rows = %d wave-rows), variant A: SALU %.0f, VALU %.0f, LDS %.1f, VMEM_RD %.2f, VMEM_WR %.2f instructions" % (
    rows, tab['SQ_INSTS_SALU'][0] / rows, tab['SQ_INSTS_VALU'][0] / rows, tab['SQ_INSTS_LDS'][0] / rows, tab['SQ_INSTS_VMEM_RD'][0] / rows, tab['SQ_INSTS_VMEM_WR'][0] / rows))
for v in range(3):
    wc = tab['SQ_WAVE_CYCLES'][v]
    print("# %-14s wave cycles %.3g: issuing %.0f %%, issue-stalled (SQ_WAIT_INST_ANY: the next instruction cannot issue -- for a store, the memory pipeline is not taking it) %.0f %%, parked in s_waitcnt (SQ_WAIT_ANY) %.0f %%" % (
        names[v], wc, 100 * tab['SQ_ACTIVE_INST_ANY'][v] / wc, 100 * tab['SQ_WAIT_INST_ANY'][v] / wc, 100 * tab['SQ_WAIT_ANY'][v] / wc))
print("# reading: the stores-only launch already spends 64 % of its wave cycles issue-stalled on its stores; the full launch carries the same ~1e9 stalled cycles.")
print("# SQ_WAIT_INST_LDS is ~6e4 cycles in all: the LDS is never the stall.  The wave is not a latency chain: it queues behind the memory pipeline.")
