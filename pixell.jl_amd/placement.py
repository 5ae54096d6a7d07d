"""Class-aware placement of a (source, destination) pair of maps in MI355X HBM.

Round 3 found why the same reprojection ran 10-17 % faster with its destination "in some places" (docs/DESIGN_history_r01-r03.md 9 item 6;
profiles/r03_xcd_classes.txt, r03_scan_placement_*.jsonl): the memory of a hipMalloc'ed allocation falls into three classes
(thirds of the 288 GiB), and a kernel with several far-apart WRITE fronts -- the reprojection keeps one per XCD -- stores at
5.8-6.0 TB/s when all fronts lie in one class and at 6.8-7.1 TB/s when they are split over two.  A plain allocation is made of
large physically contiguous blocks and normally lies inside one class.

`place_pair` makes ONE allocation with head-room, maps its classes with the library's store probe (pxl_mem_probe_pair: one
0.3 ms probe per 2 GiB against a reference window of each class found so far) and returns a destination that straddles a class
boundary (half of the XCD fronts on either side) and a source elsewhere.  This is topology discovery -- like asking which NUMA
node a page lives on -- not a search over timings of the caller's kernel: nothing about the workload is measured, and the rule
"destination across a boundary, boundary in the middle" is fixed.  The head-room stays allocated as long as the maps live
(hipMalloc cannot give part of an allocation back), so this is for hosts with memory to spare: a 45 GB pair is placed inside a
186 GiB allocation (144 GiB of head-room: a fresh device hands out one class for 64 GiB and the next for another 64, so the third
class -- where the source should go: with the source in a class the destination also uses the launch loses 3-4 % -- only shows
up beyond 128 GiB).
"""
import ctypes as C

import torch

from . import _lib

GiB = 1 << 30


def _probe(base_ptr, off_a, off_b, window, stream, reps=3):
    us = C.c_float()
    _lib.check(_lib.load().pxl_mem_probe_pair(C.c_void_p(base_ptr + off_a), C.c_void_p(base_ptr + off_b), window, reps, C.byref(us), stream))
    return float(us.value)


def map_classes(arena: torch.Tensor, step_gib=2, window_gib=1):
    """Label 1 GiB windows of `arena` (a uint8 / float64 device tensor; OVERWRITTEN with zeros where probed) every `step_gib`:
    returns (offsets in bytes, labels, info).  Two windows get the same label when the eight-front store probe between them runs
    at the slow one of its two rates."""
    assert arena.is_cuda and arena.is_contiguous()
    nbytes = arena.numel() * arena.element_size()
    base = arena.data_ptr()
    window, step = int(window_gib * GiB), int(step_gib * GiB)
    offs = list(range(0, nbytes - window + 1, step))
    with torch.cuda.device(arena.device):
        stream = C.c_void_p(torch.cuda.current_stream(arena.device).cuda_stream)
        if len(offs) < 2:
            return offs, [0] * len(offs), {"classes": 1, "probes": 0}
        # calibration: every window against window 0 -- the times are bimodal (same class: slow, different: fast)
        t0 = [None] + [_probe(base, o, offs[0], window, stream) for o in offs[1:]]
        lo, hi = min(t0[1:]), max(t0[1:])
        nprobes = len(offs) - 1
        if hi >= 1.10 * lo:
            thr = (lo * hi) ** 0.5
        else:
            # one rate only (everything of window 0's class, or nothing else): fall back to the part's two absolute rates,
            # 5.6-5.9 TB/s within a class and 6.6-7.0 TB/s across classes -- the dividing line is 6.25 TB/s
            thr = 2.0 * window / 6.25e6
        labels = [0] + [-1] * (len(offs) - 1)
        refs = [0]
        for k in range(1, len(offs)):
            if t0[k] > thr:
                labels[k] = 0
        for k in range(1, len(offs)):
            if labels[k] >= 0:
                continue
            for c in range(1, len(refs)):
                t = _probe(base, offs[k], offs[refs[c]], window, stream)
                nprobes += 1
                if t > thr:
                    labels[k] = c
                    break
            if labels[k] < 0:
                labels[k] = len(refs)
                refs.append(k)
    return offs, labels, {"classes": len(refs), "probes": nprobes, "probe_us_same_class": round(hi, 1), "probe_us_different_classes": round(lo, 1)}


def _separate_in_other_class(nelem, dtype, dev, ref_ptrs, budget_bytes):
    """A separate allocation of `nelem` elements none of whose first GiB shares a class with the 1 GiB reference windows at
    `ref_ptrs` (candidates that do are held as ballast while the search goes on, and freed before returning), or None when none
    turns up within the budget / the free memory."""
    esz = torch.empty((), dtype=dtype).element_size()
    if nelem * esz < GiB:
        return None, 0
    thr = 2.0 * GiB / 6.25e6
    ballast, held, found = [], 0, None
    with torch.cuda.device(dev):
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        while True:
            free, _t = torch.cuda.mem_get_info(dev)
            if free < nelem * esz + 8 * GiB or held + nelem * esz > budget_bytes:
                break
            cand = torch.empty(nelem, dtype=dtype, device=dev)
            same = False
            for rp in ref_ptrs:
                us = C.c_float()
                _lib.check(_lib.load().pxl_mem_probe_pair(C.c_void_p(cand.data_ptr()), C.c_void_p(rp), GiB, 3, C.byref(us), stream))
                same = same or us.value > thr
            if not same:
                found = cand
                break
            ballast.append(cand)
            held += nelem * esz
        tried = len(ballast) + (1 if found is not None else 0)
        cand = None                      # the last rejected candidate must not outlive the ballast (empty_cache below returns it)
        del ballast
        torch.cuda.empty_cache()
    return found, tried


def place_pair(src_shape, dst_shape, dtype=torch.float64, device="cuda", headroom_gib=144, step_gib=2):
    """(src, dst, info): a zero-filled source and a destination inside one allocation of (pair size + headroom_gib), the
    destination centred on a boundary between two memory classes whenever the allocation contains one with enough room on both
    sides, the source in a class the destination does not touch if there is one.  Keep `info["arena"]` alive as long as the maps
    are in use (the tensors are views of it)."""
    import math
    dev = torch.device(device)
    esz = torch.empty((), dtype=dtype).element_size()
    ns, nd = math.prod(src_shape), math.prod(dst_shape)
    al = (2 << 20)
    bs, bd = -(-ns * esz // al) * al, -(-nd * esz // al) * al
    free, _total = torch.cuda.mem_get_info(dev)
    want = bs + bd + int(headroom_gib * GiB)
    total = min(want, max(bs + bd + al, free - 6 * GiB))
    total = total // al * al
    arena = torch.empty(total, dtype=torch.uint8, device=dev)
    offs, labels, cinfo = map_classes(arena, step_gib=step_gib)
    step = int(step_gib * GiB)
    # runs of equal label; a boundary sits between the last window of one run and the first of the next: take it half-way
    bounds = []
    for k in range(1, len(offs)):
        if labels[k] != labels[k - 1]:
            bounds.append((offs[k - 1] + offs[k] + GiB) // 2)
    edges = [0] + bounds + [total]
    best = None
    for i, b in enumerate(bounds):
        left, right = b - edges[i], edges[i + 2] - b          # room of the two classes either side of this boundary
        half = bd // 2
        score = min(left, right, half)
        if best is None or score > best[0]:
            best = (score, b, left, right)
    how = "no class boundary inside the allocation: destination above the source (one class)"
    dst_off = (total - bd) // al * al
    src_off = 0
    if best is not None and best[0] >= min(bd // 8, 1 * GiB):
        score, b, left, right = best
        # centre on the boundary, shifted as far as needed to stay inside the allocation
        dst_off = b - bd // 2
        dst_off = max(0, min(dst_off, total - bd)) // al * al
        split = (b - dst_off) / bd
        how = "destination across a class boundary (%.0f %% / %.0f %% of it on either side)" % (100 * split, 100 * (1 - split))
        # source: the free stretch (before or after the destination) that fits, preferring labels the destination does not touch
        dst_labels = {labels[k] for k in range(len(offs)) if offs[k] + GiB > dst_off and offs[k] < dst_off + bd}
        cands = []
        if dst_off >= bs:
            cands += [o for o in range(0, dst_off - bs + 1, step)] + [dst_off - bs]
        if total - (dst_off + bd) >= bs:
            start = -(-(dst_off + bd) // al) * al
            cands += [o for o in range(start, total - bs + 1, step)] + [total - bs]
        if not cands:
            raise RuntimeError("place_pair: head-room too small to hold the source beside the destination")

        def foreign(o):          # share of the source's windows whose class the destination does not touch
            ks = [k for k in range(len(offs)) if offs[k] + GiB > o and offs[k] < o + bs]
            return sum(1 for k in ks if labels[k] not in dst_labels) / max(1, len(ks))
        src_off = max(cands, key=lambda o: (round(foreign(o), 2), abs(o - dst_off)))
        src_off = src_off // al * al
    else:
        src_off = 0
    assert src_off + bs <= dst_off or dst_off + bd <= src_off
    view = arena.view(dtype)
    dst = view[dst_off // esz: dst_off // esz + nd].view(tuple(dst_shape))
    src = None
    src_how = "inside the allocation"
    if best is not None and best[0] >= min(bd // 8, 1 * GiB):
        ks = [k for k in range(len(offs)) if offs[k] + GiB > src_off and offs[k] < src_off + bs]
        if any(labels[k] in dst_labels for k in ks):
            # the allocation holds no third class with room for the source (it would share a class with half of the destination:
            # 3-4 % on the 2x refinement): look for one outside, in a separate allocation
            refs = {}
            for k in range(len(offs)):
                if labels[k] in dst_labels:
                    refs.setdefault(labels[k], arena.data_ptr() + offs[k])
            found, tried = _separate_in_other_class(ns, dtype, dev, list(refs.values()), 96 * GiB)
            if found is not None:
                src, src_how = found.view(tuple(src_shape)), "a separate allocation in a class the destination does not touch (%d tried)" % tried
            else:
                src_how = "inside the allocation, in a class the destination also uses (no third class within reach, %d separate allocations tried)" % tried
    if src is None:
        src = view[src_off // esz: src_off // esz + ns].view(tuple(src_shape))
    src.zero_()
    runs = []
    for k in range(len(offs)):
        if not runs or runs[-1][0] != labels[k]:
            runs.append([labels[k], offs[k] // GiB, offs[k] // GiB])
        runs[-1][2] = offs[k] // GiB + 1
    info = {"arena": arena, "allocation_GiB": round(total / GiB, 1), "src_offset_GiB": round(src_off / GiB, 2),
            "dst_offset_GiB": round(dst_off / GiB, 2), "placement": how, "source": src_how, "class_runs_label_from_to_GiB": runs}
    info.update(cinfo)
    return src, dst, info


def place_pair_shifted(src_shape, dst_shape, dtype=torch.float64, device="cuda", scout_gib=64, min_share=0.3, step_gib=2):
    """(src zero-filled, dst, info): the pair in ONE allocation of exactly its own size -- no head-room kept -- whose destination
    lies across a boundary between two memory classes.  How: a scout allocation of (pair + scout_gib) is mapped with the store
    probe (map_classes) to find where the class boundaries lie from its start; it is returned to the driver; a ballast of the size
    that pushes the pair's destination onto the chosen boundary is allocated in its place, the pair right after it (consecutive
    allocations walk through the device's memory), and the ballast is returned too.  The pair's destination is then probed again:
    at least `min_share` of its 1-GiB windows must lie in a second class, otherwise the pair is kept as it fell (and info says so).
    Topology discovery by a fixed rule, as place_pair: nothing about the caller's kernel is timed.  Keep info["arena"] alive."""
    import math
    import time
    t0 = time.perf_counter()
    dev = torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    esz = torch.empty((), dtype=dtype).element_size()
    ns, nd = math.prod(src_shape), math.prod(dst_shape)
    al = 2 << 20
    bs, bd = -(-ns * esz // al) * al, -(-nd * esz // al) * al
    pair = bs + bd
    info = {"policy": "class-aware pair: scout, ballast, exact allocation", "pair_GiB": round(pair / GiB, 2)}

    def carve(arena, src_first):
        view = arena.view(dtype)
        so, do = (0, bs) if src_first else (bd, 0)
        return view[so // esz: so // esz + ns].view(tuple(src_shape)), view[do // esz: do // esz + nd].view(tuple(dst_shape))

    with torch.cuda.device(dev):
        free, _t = torch.cuda.mem_get_info(dev)
        plan = None
        if bd >= 3 * GiB and free >= pair + 12 * GiB:
            total = min(pair + int(scout_gib * GiB), free - 8 * GiB) // al * al
            scout = torch.empty(total, dtype=torch.uint8, device=dev)
            offs, labels, cinfo = map_classes(scout, step_gib=step_gib)
            info["scout_GiB"] = round(total / GiB, 1)
            info["probes"] = cinfo.get("probes", 0)
            bounds = [(offs[k - 1] + offs[k] + GiB) // 2 for k in range(1, len(offs)) if labels[k] != labels[k - 1]]
            info["boundaries_GiB_from_scout_start"] = [round(b / GiB, 1) for b in bounds]
            # the pair's start s: destination centred on a boundary b, source below it (s = b - bs - bd/2) or above it (s = b - bd/2);
            # the smallest ballast wins
            cands = []
            for b in bounds:
                for src_first, s in ((True, b - bs - bd // 2), (False, b - bd // 2)):
                    if s >= 0 and s + pair <= total:
                        cands.append((s, src_first, b))
            del scout
            torch.cuda.empty_cache()
            if cands:
                plan = min(cands)
        if plan is None:
            arena = torch.empty(pair, dtype=torch.uint8, device=dev)
            src, dst = carve(arena, True)
            info.update({"placement": "plain (no boundary within the scout allocation, or a destination below 3 GiB)", "arena": arena})
        else:
            s, src_first, b = plan
            s = s // al * al
            ballast = torch.empty(s, dtype=torch.uint8, device=dev) if s >= al else None
            arena = torch.empty(pair, dtype=torch.uint8, device=dev)
            src, dst = carve(arena, src_first)
            del ballast
            torch.cuda.empty_cache()
            # where did it fall?  (probed windows of the destination are overwritten with zeros: it is uninitialised anyway)
            _o, dlab, dinfo = map_classes(dst.view(-1).view(torch.uint8), step_gib=1)
            _major, share = _split_of(dlab)
            info["probes"] = info.get("probes", 0) + dinfo.get("probes", 0)
            info.update({"ballast_GiB": round(s / GiB, 2), "layout": "source below the destination" if src_first else "destination below the source",
                         "destination_minor_class_share": round(share, 2), "arena": arena,
                         "placement": "destination across a class boundary (%.0f %% of its windows in the second class)" % (100 * share)})
            if share < min_share:
                # the allocator did not put the pair behind the ballast (seen with ballasts of tens of GiB): the candidate search
                # of place_pair_compact instead -- two allocations, nothing kept beyond them either
                del src, dst, arena
                info.pop("arena")
                torch.cuda.empty_cache()
                src, dst, cinfo2 = place_pair_compact(src_shape, dst_shape, dtype=dtype, device=dev, budget_gib=96)
                info.update({"placement": "pair not behind its ballast (%.0f %% in a second class); candidate search instead: %s" % (100 * share, cinfo2.get("placement")),
                             "source": cinfo2.get("source"), "candidates_minor_share": cinfo2.get("candidates_minor_share"), "arena": None})
                info["seconds"] = round(time.perf_counter() - t0, 3)
                return src, dst, info
        src.zero_()
    info["seconds"] = round(time.perf_counter() - t0, 3)
    return src, dst, info


def place_streams(shapes, dtype=torch.float64, device="cuda", headroom_gib=144, step_gib=2):
    """(tensors, info): one buffer per shape inside ONE allocation, each in a memory class of its own as far as the allocation
    has classes with room (buffer k goes into the longest free run of the class used least so far).  For kernels that write several
    streams at once -- `posmap` writes its RA and its DEC map together: 6.8 TB/s with the two in different classes, 5.3 TB/s with
    both in one (profiles/r03_class_streams.jsonl) -- or that should keep their reads away from their writes.  Keep info["arena"]
    alive as long as the tensors are in use."""
    import math
    dev = torch.device(device)
    esz = torch.empty((), dtype=dtype).element_size()
    al = 2 << 20
    sizes = [-(-math.prod(sh) * esz // al) * al for sh in shapes]
    free, _total = torch.cuda.mem_get_info(dev)
    total = min(sum(sizes) + int(headroom_gib * GiB), max(sum(sizes) + al, free - 6 * GiB)) // al * al
    arena = torch.empty(total, dtype=torch.uint8, device=dev)
    offs, labels, cinfo = map_classes(arena, step_gib=step_gib)
    step = int(step_gib * GiB)
    # free runs of one label: [label, start, end) in bytes (a window stands for the step after it)
    runs = []
    for k in range(len(offs)):
        end = min(offs[k] + step, total)
        if runs and runs[-1][0] == labels[k]:
            runs[-1][2] = end
        else:
            runs.append([labels[k], offs[k], end])
    used = {}
    out, where = [], []
    view = arena.view(dtype)
    for sh, size in zip(shapes, sizes):
        fits = [r for r in runs if r[2] - r[1] >= size]
        if not fits:
            raise RuntimeError("place_streams: no run of one memory class holds %d MiB; more head-room needed" % (size >> 20))
        r = min(fits, key=lambda r: (used.get(r[0], 0), -(r[2] - r[1])))
        o = -(-r[1] // al) * al
        n = math.prod(sh)
        out.append(view[o // esz: o // esz + n].view(tuple(sh)))
        where.append({"class": r[0], "offset_GiB": round(o / GiB, 2)})
        used[r[0]] = used.get(r[0], 0) + 1
        r[1] = o + size
    info = {"arena": arena, "allocation_GiB": round(total / GiB, 1), "buffers": where}
    info.update(cinfo)
    return out, info


def _split_of(labels):
    """(majority label, share of the windows NOT carrying it)."""
    from collections import Counter
    c = Counter(labels)
    major, n = c.most_common(1)[0]
    return major, 1.0 - n / len(labels)


def place_pair_compact(src_shape, dst_shape, dtype=torch.float64, device="cuda", budget_gib=200, min_minor_share=0.25, max_tries=40):
    """The same goal as place_pair -- a destination whose pages lie in two memory classes, a source in another class if there is
    one -- WITHOUT keeping any head-room: the destination is allocated on its own, its classes are mapped with the store probe,
    and while it lies inside one class (or has less than `min_minor_share` of its windows in a second one) it is kept as ballast
    and the next one is tried -- consecutive allocations walk through the device's memory and reach a class boundary within
    64-96 GiB.  The ballast is returned to the driver before the function returns: afterwards exactly the two maps are allocated.
    The same topology probe, the same fixed acceptance rule; nothing about the caller's workload is timed.  Returns
    (src zero-filled, dst, info); falls back to the last candidate (and says so) when the budget runs out."""
    import math
    dev = torch.device(device)
    esz = torch.empty((), dtype=dtype).element_size()
    ns, nd = math.prod(src_shape), math.prod(dst_shape)
    ballast, tried = [], []
    held = 0
    dst = None
    dst_labels = None
    note = None
    with torch.cuda.device(dev):
        if nd * esz < 3 * GiB:
            dst = torch.empty(nd, dtype=dtype, device=dev)
            note = "destination below 3 GiB: plain allocation"
        while dst is None:
            cand = torch.empty(nd, dtype=dtype, device=dev)
            offs, labels, cinfo = map_classes(cand, step_gib=1)
            major, minor_share = _split_of(labels)
            tried.append(round(minor_share, 2))
            free, _t = torch.cuda.mem_get_info(dev)
            if minor_share >= min_minor_share:
                dst, dst_labels = cand, labels
                note = "destination in two classes (%.0f %% of its windows in the second) after %d allocation(s)" % (100 * minor_share, len(tried))
            elif len(tried) >= max_tries or held + nd * esz > budget_gib * GiB or free < nd * esz + ns * esz + 8 * GiB:
                dst, dst_labels = cand, labels
                note = "no two-class destination within the budget (%d allocations): the last candidate, %.0f %% in a second class" % (len(tried), 100 * minor_share)
            else:
                ballast.append(cand)
                held += nd * esz
        # the source: prefer an allocation none of whose windows shares a class with the destination
        src = None
        src_note = "plain"
        sballast = []
        if dst_labels is not None and ns * esz >= GiB:
            base_d = dst.data_ptr()
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            thr = 2.0 * GiB / 6.25e6
            # one reference window of the destination per class it has
            refs = {}
            for k, lab in enumerate(dst_labels):
                refs.setdefault(lab, k)
            sheld = 0
            while True:
                cand = torch.empty(ns, dtype=dtype, device=dev)
                shares = []
                for lab, k in refs.items():
                    us = C.c_float()
                    _lib.check(_lib.load().pxl_mem_probe_pair(C.c_void_p(cand.data_ptr()), C.c_void_p(base_d + k * GiB), GiB, 3, C.byref(us), stream))
                    shares.append(us.value > thr)              # True: same class as that part of the destination
                free, _t = torch.cuda.mem_get_info(dev)
                if not any(shares):
                    src, src_note = cand, "source in a class the destination does not touch"
                    break
                if free < ns * esz + 8 * GiB or sheld + ns * esz > budget_gib * GiB:
                    src, src_note = cand, "source shares a class with the destination (no other class within the budget)"
                    break
                sballast.append(cand)
                sheld += ns * esz
        if src is None:
            src = torch.empty(ns, dtype=dtype, device=dev)
        del ballast, sballast
        torch.cuda.empty_cache()
    src = src.view(tuple(src_shape))
    dst = dst.view(tuple(dst_shape))
    src.zero_()
    return src, dst, {"placement": note, "source": src_note, "candidates_minor_share": tried, "allocation_GiB": round((ns + nd) * esz / GiB, 1)}


class _NativePair:
    """Owns a struct pxl_mem_pair; device memory is handed to torch through __cuda_array_interface__ views that keep this alive."""

    def __init__(self, pair):
        self.pair = pair

    def view(self, ptr, shape, dtype):
        typestr = {torch.float64: "<f8", torch.float32: "<f4", torch.uint8: "|u1"}[dtype]
        owner = self

        class _View:
            __cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}
            keep = owner
        return torch.as_tensor(_View(), device="cuda")

    def __del__(self):
        try:
            _lib.load().pxl_mem_pair_free(C.byref(self.pair))
        except Exception:           # noqa: BLE001 -- interpreter shutdown
            pass


def place_pair_native(src_shape, dst_shape, dtype=torch.float64, device="cuda", headroom_gib=144):
    """The same placement through the C ABI (pxl_mem_pair_alloc: what a Julia or C host calls): (src, dst, info).  The library owns
    the allocation; it is freed when the last of the two tensors (and `info["owner"]`) is gone."""
    import math
    dev = torch.device(device)
    esz = torch.empty((), dtype=dtype).element_size()
    pair = _lib.MemPair()
    with torch.cuda.device(dev):
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(_lib.load().pxl_mem_pair_alloc(math.prod(src_shape) * esz, math.prod(dst_shape) * esz, int(headroom_gib * GiB),
                                                  C.byref(pair), stream))
        owner = _NativePair(pair)
        src = owner.view(pair.src, src_shape, dtype)
        dst = owner.view(pair.dst, dst_shape, dtype)
    info = {"owner": owner, "allocation_GiB": round(pair.arena_bytes / GiB, 1), "src_offset_GiB": round(pair.src_offset / GiB, 2),
            "dst_offset_GiB": round(pair.dst_offset / GiB, 2), "classes": pair.classes, "dst_two_classes": bool(pair.dst_two_classes),
            "src_own_class": bool(pair.src_own_class), "probes": pair.probes, "separate_source_allocation": bool(pair.src_alloc),
            "separate_tried": pair.separate_tried}
    return src, dst, info


# ---- the library's default allocation policy for map-sized outputs ---------------------------------------------------------
#
# What `pj.reproject(m, shape_out, wcs_out)` (no `out=`), `DecStripReprojector.alloc_pair()` and the Julia `reproject` / `similar`
# allocate through.  "class-aware" (the default): a destination of 3 GiB or more is allocated on its own and its 1 GiB windows
# are labelled with the store probe; a candidate that lies inside ONE memory class is kept as ballast while the next one is
# tried (consecutive allocations walk through the device's memory: a class boundary is at most one class run away, 4-32 GiB on
# the boxes seen; a candidate with 30 % of its windows in a second class is taken at once -- with six windows in a 7 GiB map the
# shares come in sixths, and holding out for 40 % cost 14 tries and 6 s where the first try already had 33 %), and all ballast goes back to the driver before the call returns -- afterwards exactly the map is allocated,
# no head-room.  "plain": torch.empty, nothing probed.  PXL_ALLOC_POLICY / set_allocation_policy() choose; PXL_ALLOC_BUDGET_GIB
# bounds the transient ballast (default 96, and never more than the free memory less 8 GiB).
import os as _os

_POLICY = _os.environ.get("PXL_ALLOC_POLICY", "class-aware")
_LAST_INFO = {}       # what the most recent empty_map() call did (diagnostics: pj.last_allocation_info())
_LABELS = {}          # (device index, data_ptr, nbytes) -> (labels, num_device_free when they were measured)
MIN_PLACED_BYTES = 3 * GiB


def set_allocation_policy(policy: str):
    """'class-aware' (default) or 'plain'; returns the previous policy."""
    global _POLICY
    if policy not in ("class-aware", "plain"):
        raise ValueError("allocation policy must be 'class-aware' or 'plain'")
    old, _POLICY = _POLICY, policy
    return old


def allocation_policy() -> str:
    return _POLICY


def last_allocation_info() -> dict:
    """The report of the most recent empty_map() call in this process (policy, tries, class shares, seconds, ballast)."""
    return dict(_LAST_INFO)


def _device_frees(dev):
    try:
        return int(torch.cuda.memory_stats(dev).get("num_device_free", 0))
    except Exception:           # noqa: BLE001
        return -1


def _labels_of(t: torch.Tensor):
    """Class labels of the 1 GiB windows of `t` (probed once per block: torch's caching allocator hands the same block out again
    with the same address, and its physical pages stay put until the allocator returns memory to the driver -- the cache entry
    is dropped as soon as the device's free counter has moved).  OVERWRITES the probed windows with zeros."""
    dev = t.device
    key = (dev.index, t.data_ptr(), t.numel() * t.element_size())
    frees = _device_frees(dev)
    hit = _LABELS.get(key)
    if hit is not None and hit[1] == frees and frees >= 0:
        return hit[0], 0
    offs, labels, cinfo = map_classes(t, step_gib=1)
    if len(_LABELS) > 64:
        _LABELS.clear()
    _LABELS[key] = (labels, frees)
    return labels, cinfo.get("probes", 0)


def empty_map(shape, dtype=torch.float64, device="cuda", policy=None, budget_gib=None, accept_share=0.3, min_share=0.2, max_tries=24):
    """(tensor, info): an uninitialised device tensor of `shape` allocated by the library's policy (above).  Contents are
    unspecified (probed windows hold zeros).  info says what was done: policy, tries, the share of the buffer's windows in its
    second class, seconds spent, bytes of ballast held transiently."""
    import math
    import time
    dev = torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    policy = policy or _POLICY
    esz = torch.empty((), dtype=dtype).element_size()
    n = math.prod(shape)
    nbytes = n * esz
    if policy == "plain" or nbytes < MIN_PLACED_BYTES:
        info = {"policy": "plain" if policy == "plain" else "class-aware: below %d GiB, plain allocation" % (MIN_PLACED_BYTES // GiB), "tries": 1}
        _LAST_INFO.clear(); _LAST_INFO.update(info)
        return torch.empty(tuple(shape), dtype=dtype, device=dev), info
    t0 = time.perf_counter()
    if budget_gib is None:
        budget_gib = float(_os.environ.get("PXL_ALLOC_BUDGET_GIB", "96"))
    ballast, shares = [], []
    best = None                  # (share, tensor)
    held = 0
    probes = 0
    with torch.cuda.device(dev):
        while True:
            cand = torch.empty(n, dtype=dtype, device=dev)
            labels, p = _labels_of(cand)
            probes += p
            _major, share = _split_of(labels)
            shares.append(round(share, 2))
            if best is None or share > best[0]:
                if best is not None:
                    ballast.append(best[1])
                best = (share, cand)
            else:
                ballast.append(cand)
            cand = None
            held = sum(b.numel() * esz for b in ballast)
            free, _t = torch.cuda.mem_get_info(dev)
            # a map of 16 GiB or more is built by the driver from several blocks and shows ~20 % of its windows in a second class in
            # EVERY candidate (five tries, 83 GiB of ballast and 3 s bought nothing on the 21 GiB IQU map): the first such one is taken
            big_enough = nbytes >= 16 * GiB and best[0] >= min_share
            if best[0] >= accept_share or big_enough or len(shares) >= max_tries or held + nbytes > budget_gib * GiB or free < nbytes + 8 * GiB:
                break
        chosen = best[1]
        freed = bool(ballast)
        peak = held
        del ballast
        best = None
        if freed:
            torch.cuda.empty_cache()         # the ballast goes back to the driver: only the map stays allocated
    share = max(shares)
    how = ("two classes (%.0f %% of its windows in the second)" % (100 * share)) if share >= min_share else "one class (no boundary within the budget)"
    info = {"policy": "class-aware", "placement": how, "tries": len(shares), "candidates_minor_share": shares, "probes": probes,
            "seconds": round(time.perf_counter() - t0, 3), "transient_ballast_GiB": round(peak / GiB, 1), "held_GiB": round(nbytes / GiB, 2)}
    _LAST_INFO.clear(); _LAST_INFO.update(info)
    return chosen.view(tuple(shape)), info
