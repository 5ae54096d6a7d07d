#!/usr/bin/env python3
"""Does the scattered sampler care which memory classes its map lies in?  The row-pair copy of the 0.5-arcmin map (20 GB) is built
at chosen offsets of one big allocation whose classes have been mapped (pj.map_classes): inside ONE class, across a boundary
between two, and where it covers three if the allocation has such a place; 5e8 uniform-on-sphere points, HIP-event medians.
Sequential reads do not care about classes and multi-front stores do; random 64-byte reads were untested."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pixell_jl_amd as pj
GiB = 1 << 30
dev = torch.device("cuda:0")
shape, wcs = pj.fullsky_geometry(2 * 3.141592653589793 / 43200)
nx, ny = shape
m = torch.empty((ny, nx), dtype=torch.float64, device=dev)
pj.fill_random_(m, 7)
em = pj.Enmap(m, wcs)
n = int(float(os.environ.get("PXL_N", "5e8")))
sky = torch.empty((n, 2), dtype=torch.float64, device=dev)
pj.fill_sphere_points_(sky, 11)
probe = pj.SamplePairs(em)
nel = probe.data.numel()
del probe
torch.cuda.empty_cache()
free, _ = torch.cuda.mem_get_info(dev)
total = min(200 * GiB, free - 16 * GiB) // (2 << 20) * (2 << 20)
arena = torch.empty(total, dtype=torch.uint8, device=dev)
offs, labels, info = pj.map_classes(arena, step_gib=2)
runs = []
for o, l in zip(offs, labels):
    if not runs or runs[-1][0] != l:
        runs.append([l, o, o])
    runs[-1][2] = o + 2 * GiB
print(json.dumps({"allocation_GiB": total // GiB, "classes": info["classes"], "runs_label_from_to_GiB": [[l, a // GiB, b // GiB] for l, a, b in runs]}), flush=True)
nbytes = nel * 8
need = -(-nbytes // (2 << 20)) * (2 << 20)


def hist(o):
    h = {}
    for oo, l in zip(offs, labels):
        if oo + GiB > o and oo < o + need:
            h[l] = h.get(l, 0) + 1
    tot = sum(h.values())
    return {k: v / tot for k, v in h.items()}


cands = [o for o in range(0, total - need + 1, 2 * GiB)]
one = [o for o in cands if len(hist(o)) == 1]
two = sorted((o for o in cands if len(hist(o)) == 2), key=lambda o: abs(max(hist(o).values()) - 0.5))
three = sorted((o for o in cands if len(hist(o)) >= 3), key=lambda o: max(hist(o).values()))
picks = []
if one:
    picks.append(("one class", one[0]))
    if len(one) > 1:
        picks.append(("one class (another place)", one[-1]))
if two:
    picks.append(("two classes, about half each", two[0]))
if three:
    picks.append(("three classes", three[0]))


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


for rnd in range(2):
    for name, o in picks:
        buf = arena[o:o + nbytes].view(torch.float64)
        pairs = pj.SamplePairs(em, out=buf)
        ms = timed(lambda: pj.sample_bilinear(None, sky, pairs=pairs))
        print(json.dumps({"pair copy in": name, "offset_GiB": o // GiB, "class_shares": {str(k): round(v, 2) for k, v in hist(o).items()},
                          "k_sample_pairs_ms": round(ms, 3), "Gpts_s": round(n / ms / 1e6, 2)}), flush=True)
        del pairs, buf
# the direct sampler on the map itself, inside one class and across a boundary
mb = m.numel() * 8
for name, lst in (("one class", one), ("two classes", two)):
    if not lst:
        continue
    o = lst[0]
    mm = arena[o:o + mb].view(torch.float64).view(ny, nx)
    mm.copy_(m)
    e2 = pj.Enmap(mm, wcs)
    ms = timed(lambda: pj.sample_bilinear(e2, sky), reps=3)
    print(json.dumps({"direct sampler, map in": name, "offset_GiB": o // GiB, "ms": round(ms, 3)}), flush=True)
