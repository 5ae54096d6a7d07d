"""Declination-strip sharding of a reprojection across the GPUs of one node.

One process per GPU.  The OUTPUT map is partitioned into contiguous DEC strips (rows), the SOURCE map
into strips by the same even split; because source row depends only on output row (CAR -> CAR is
separable), an output strip needs its own source strip plus a few halo rows owned by the neighbouring
ranks (one row each side when source and output share DEC boundaries).  Halo rows travel as
point-to-point send/recv over torch.distributed (RCCL over xGMI on GPUs, gloo in the CPU tests); there
is no reduction and no all-gather on this path.  The RA seam stays inside every shard.

Shard descriptors are windows (row0, nrows) into the FULL geometry -- not re-derived WCS -- so the
sharded result is bit-identical to the single-GPU one.
"""
import numpy as np
import torch
import torch.distributed as dist

from .wcs import TWOPI


def strip_bounds(n: int, world: int, rank: int):
    """Even contiguous split of n rows: rank r owns [lo, hi)."""
    return (n * rank) // world, (n * (rank + 1)) // world


def _np_jl_mod(x, y: float):
    r = np.fmod(x, y)
    zero = r == 0.0
    flip = (r > 0.0) != (y > 0.0)
    r = np.where(flip & ~zero, r + y, r)
    return np.where(zero, np.copysign(0.0, y), r)


def _np_rewind(a, period: float, ref: float):
    half = period / 2
    return (ref + _np_jl_mod((a - ref) + half, period)) - half


def host_row_cells(shape_in, wcs_in, shape_out, wcs_out):
    """Source cell j0 (1-based, int64) of every output row: numpy restatement, operation for operation,
    of the device row table (pix2sky safe=false [car_proj.jl:147] then scalar sky2pix safe=true
    [car_proj.jl:226,230]).  Host-side planning only."""
    ny_in = int(shape_in[1])
    j = np.arange(1, int(shape_out[1]) + 1, dtype=np.float64)
    d = wcs_out.crval[1] * wcs_out.unit + (j - wcs_out.crpix[1]) * (wcs_out.cdelt[1] * wcs_out.unit)
    dd = wcs_in.cdelt[1] * wcs_in.unit
    y = wcs_in.crpix[1] + (d - wcs_in.crval[1] * wcs_in.unit) / dd
    y = _np_rewind(y, abs(TWOPI / dd), ny_in / 2 + 1)
    f = np.floor(y)
    f = np.minimum(np.maximum(f, -1073741824.0), 1073741824.0)
    return f.astype(np.int64)


def _rows_needed(cells, ny_in, lo, hi):
    """0-based half-open source row range touched by output rows [lo, hi): taps j0 and j0+1 inside the map."""
    if hi <= lo:
        return 0, 0
    c = cells[lo:hi]
    taps = np.concatenate([c, c + 1])
    taps = taps[(taps >= 1) & (taps <= ny_in)]
    if taps.size == 0:
        return 0, 0
    return int(taps.min()) - 1, int(taps.max())


class DecStripLayout:
    """Pure host logic: who owns which rows, which halo rows move between which ranks."""

    def __init__(self, shape_in, wcs_in, shape_out, wcs_out, rank: int, world: int):
        self.shape_in = (int(shape_in[0]), int(shape_in[1]), int(shape_in[2]) if len(shape_in) > 2 else 1)
        self.shape_out = (int(shape_out[0]), int(shape_out[1]))
        self.wcs_in, self.wcs_out = wcs_in, wcs_out
        self.rank, self.world = rank, world
        nx, ny, nc = self.shape_in
        nyo = self.shape_out[1]
        self.cells = host_row_cells(self.shape_in, wcs_in, self.shape_out, wcs_out)
        self.own = [strip_bounds(ny, world, r) for r in range(world)]           # source rows owned
        self.out = [strip_bounds(nyo, world, r) for r in range(world)]          # output rows owned
        self.need = [_rows_needed(self.cells, ny, *self.out[r]) for r in range(world)]
        o_lo, o_hi = self.own[rank]
        n_lo, n_hi = self.need[rank]
        if n_hi <= n_lo:
            n_lo, n_hi = o_lo, o_hi
        self.buf_lo, self.buf_hi = min(o_lo, n_lo), max(o_hi, n_hi)             # rows resident on this rank
        # halo traffic: (peer, row_lo, row_hi) absolute 0-based rows
        self.recvs, self.sends = [], []
        for q in range(world):
            if q == rank:
                continue
            lo, hi = max(self.need[rank][0], self.own[q][0]), min(self.need[rank][1], self.own[q][1])
            if hi > lo:
                self.recvs.append((q, lo, hi))
            lo, hi = max(self.need[q][0], self.own[rank][0]), min(self.need[q][1], self.own[rank][1])
            if hi > lo:
                self.sends.append((q, lo, hi))
        # interior: longest run of my output rows computable from my own source rows alone
        self.interior = self._interior()

    # -- windows for the kernels
    @property
    def src_window(self):
        return self.buf_lo, self.buf_hi - self.buf_lo

    @property
    def dst_window(self):
        lo, hi = self.out[self.rank]
        return lo, hi - lo

    def src_tensor_shape(self):
        return (self.shape_in[2], self.buf_hi - self.buf_lo, self.shape_in[0])

    def dst_tensor_shape(self):
        return (self.shape_in[2], self.out[self.rank][1] - self.out[self.rank][0], self.shape_out[0])

    def own_slice(self):
        """Slice of the resident buffer's row axis holding the rows this rank owns."""
        lo, hi = self.own[self.rank]
        return slice(lo - self.buf_lo, hi - self.buf_lo)

    def halo_bytes(self):
        nx, _, nc = self.shape_in
        return sum((hi - lo) * nx * nc * 8 for _, lo, hi in self.recvs)

    def _interior(self):
        lo, hi = self.out[self.rank]
        own_lo, own_hi = self.own[self.rank]
        ny = self.shape_in[1]
        c = self.cells[lo:hi]
        ok = np.ones(c.shape, dtype=bool)
        for t in (c, c + 1):
            inmap = (t >= 1) & (t <= ny)
            ok &= ~inmap | ((t - 1 >= own_lo) & (t - 1 < own_hi))
        best = (0, 0)
        start = None
        for k in range(len(ok) + 1):
            if k < len(ok) and ok[k]:
                if start is None:
                    start = k
            elif start is not None:
                if k - start > best[1] - best[0]:
                    best = (start, k)
                start = None
        return best          # relative to the dst window

    # -- halo exchange (backend-agnostic: RCCL for cuda tensors, gloo for cpu tensors)
    def make_staging(self, like: torch.Tensor, group=None):
        """Packed (nc, rows, nx) message buffers.  They live where the transport can reach them: on the
        GPU for RCCL, in host memory when a gloo group moves GPU data (1-GPU rehearsals of the N-rank flow)."""
        nx, _, nc = self.shape_in
        dev = like.device
        if like.is_cuda and dist.is_initialized() and dist.get_backend(group) == "gloo":
            dev = torch.device("cpu")
        send = [torch.empty((nc, hi - lo, nx), dtype=like.dtype, device=dev) for _, lo, hi in self.sends]
        recv = [torch.empty((nc, hi - lo, nx), dtype=like.dtype, device=dev) for _, lo, hi in self.recvs]
        return send, recv

    def start_halo_exchange(self, src: torch.Tensor, staging, group=None):
        """Post all sends/recvs; returns the work handles.  src is the (nc, buf rows, nx) resident buffer."""
        send_bufs, recv_bufs = staging
        ops = []
        for (peer, lo, hi), buf in zip(self.sends, send_bufs):
            buf.copy_(src[:, lo - self.buf_lo:hi - self.buf_lo, :])
            ops.append(dist.P2POp(dist.isend, buf, peer, group))
        for (peer, lo, hi), buf in zip(self.recvs, recv_bufs):
            ops.append(dist.P2POp(dist.irecv, buf, peer, group))
        return dist.batch_isend_irecv(ops) if ops else []

    def finish_halo_exchange(self, src: torch.Tensor, staging, works):
        for w in works:
            w.wait()
        _, recv_bufs = staging
        for (peer, lo, hi), buf in zip(self.recvs, recv_bufs):
            src[:, lo - self.buf_lo:hi - self.buf_lo, :].copy_(buf)


class DecStripReprojector(DecStripLayout):
    """The sharded operator on one rank's GPU: halo exchange on RCCL overlapped with the interior rows,
    then the boundary rows.  Produces rows out[rank] of the output map."""

    def __init__(self, shape_in, wcs_in, shape_out, wcs_out, rank, world, device, group=None):
        super().__init__(shape_in, wcs_in, shape_out, wcs_out, rank, world)
        from .ops import ReprojectPlan
        self.device = torch.device(device)
        self.group = group
        self.plan = ReprojectPlan(self.shape_in, wcs_in, self.shape_out, wcs_out, src_rows=self.src_window,
                                  dst_rows=self.dst_window, device=self.device)
        self._staging = None
        self._xfers = None

    def alloc_src(self):
        return torch.zeros(self.src_tensor_shape(), dtype=torch.float64, device=self.device)

    def alloc_dst(self):
        return torch.empty(self.dst_tensor_shape(), dtype=torch.float64, device=self.device)

    def alloc_maps(self, dtype=torch.float64, policy=None):
        """(src zero-filled, dst, info): this rank's resident source strip and its output strip, allocated by the library's policy
        (placement.py: class-aware by default -- place_pair_shifted: ONE allocation of exactly the pair's size whose destination
        lies across a class boundary, nothing kept beyond the two maps; 'plain' = alloc_pair; 'compact' = round 3's candidate search).  What a host that lets the library allocate gets."""
        from . import placement
        policy = policy or placement.allocation_policy()
        if policy == "plain":
            src, dst, arena = self.alloc_pair(dtype)
            return src, dst, {"policy": "plain", "arena": arena}
        if policy == "compact":
            src, dst, info = placement.place_pair_compact(self.src_tensor_shape(), self.dst_tensor_shape(), dtype=dtype, device=self.device)
            info["policy"] = "class-aware (compact search)"
            return src, dst, info
        return placement.place_pair_shifted(self.src_tensor_shape(), self.dst_tensor_shape(), dtype=dtype, device=self.device)

    def alloc_pair(self, dtype=torch.float64):
        """Source and destination buffers carved out of ONE device allocation, destination above the source on a 2 MiB
        boundary.  Where the write stream lands physically moves the reprojection by up to 8 % (stores alone: 3.2 vs 3.7
        ms for 22 GB, same kernel; tools/research/exp_placement_vmm.cpp, profiles/r02_placement_arena.jsonl).  For the
        0.5-arcmin IQU map this arrangement measured 7.2-7.37 ms in 20 of 24 processes on 9 boxes (two allocations: 7.2-8.1,
        about half of them slow, on some boxes every time); for the 2x-refinement workloads it is within 1 % of two
        allocations either way (profiles/r02_ab_arena_*.txt).  bench.py allocates through it.
        Returns (src zero-filled, dst, arena) -- keep `arena` alive."""
        ns = 1
        for d in self.src_tensor_shape():
            ns *= d
        nd = 1
        for d in self.dst_tensor_shape():
            nd *= d
        esz = torch.empty((), dtype=dtype).element_size()
        al = (2 << 20) // esz
        off = (ns + al - 1) // al * al
        arena = torch.empty(off + nd, dtype=dtype, device=self.device)
        src = arena[:ns].view(self.src_tensor_shape())
        dst = arena[off:off + nd].view(self.dst_tensor_shape())
        src.zero_()
        return src, dst, arena

    def rccl_comm_ptr(self):
        """The ncclComm_t of this job's RCCL process group, for the native step.  `ProcessGroupNCCL._comm_ptr` is a PRIVATE torch
        API (present in torch 2.4 ... 2.10, this image's); it is looked up by name, and a torch without it raises a RuntimeError
        that says which transport the caller takes instead -- `native_ready()` turns that into (False, reason) on every rank before
        anything is posted, and bench.py / a host then uses a communicator made through the library's own ABI (`make_own_comm`:
        pxl_comm_unique_id + pxl_comm_init_rank, no torch internals), or torch.distributed P2P (`step`)."""
        pg = self.group if self.group is not None else dist.distributed_c10d._get_default_group()
        get_backend = getattr(pg, "_get_backend", None)
        if get_backend is None:
            raise RuntimeError("torch %s: ProcessGroup has no _get_backend(); use make_own_comm() (the library's own RCCL communicator) "
                               "or step() (torch P2P) instead of torch's communicator" % torch.__version__)
        backend = get_backend(self.device)
        comm_ptr = getattr(backend, "_comm_ptr", None)
        if comm_ptr is None:
            raise RuntimeError("torch %s: %s has no _comm_ptr() (a private API; not an RCCL group, or removed in this version); use "
                               "make_own_comm() (the library's own RCCL communicator) or step() (torch P2P) instead"
                               % (torch.__version__, type(backend).__name__))
        return int(comm_ptr())

    def make_own_comm(self, ctrl_group=None):
        """A communicator made through the library's own ABI (pxl_comm_unique_id / pxl_comm_init_rank) -- what a Julia
        or C host does; here the 128-byte id travels over torch.distributed (any backend, e.g. a gloo control group).
        Collective over the job's ranks.  Returns the ncclComm_t as an int; step_native(comm_ptr=...) takes it."""
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        ident = C.create_string_buffer(128)
        # rank 0 ALWAYS broadcasts -- the id or the reason it could not make one -- so that a failure there (RCCL not
        # found, a bad PXL_RCCL_LIB) reaches every rank as an exception instead of leaving them in the broadcast
        box = [None]
        if self.rank == 0:
            try:
                with torch.cuda.device(self.device):
                    _lib.check(lib.pxl_comm_unique_id(ident))
                box = [(True, bytes(ident.raw))]
            except Exception as e:                  # noqa: BLE001 -- re-raised on every rank below
                box = [(False, "%s: %s" % (type(e).__name__, (str(e).splitlines() or [""])[0][:200]))]
        dist.broadcast_object_list(box, src=0, group=ctrl_group)
        ok, payload = box[0]
        if not ok:
            raise RuntimeError("rank 0 could not create the communicator id: " + payload)
        ident = C.create_string_buffer(payload, 128)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(lib.pxl_comm_init_rank(ident, self.rank, self.world, C.byref(h)))
        self._own_comm = h
        return int(h.value)

    def close_own_comm(self):
        from . import _lib
        if getattr(self, "_own_comm", None):
            _lib.check(_lib.load().pxl_comm_destroy(self._own_comm))
            self._own_comm = None

    def native_ready(self):
        """Local pre-flight of the native step, before any rank posts anything: is the job's RCCL communicator
        reachable from here?  Returns (ok, reason)."""
        if not (self.sends or self.recvs):
            return True, ""
        try:
            return (self.rccl_comm_ptr() != 0), "ProcessGroupNCCL has no communicator yet"
        except Exception as e:                      # noqa: BLE001 -- reported by the caller
            return False, "%s: %s" % (type(e).__name__, (str(e).splitlines() or [""])[0][:160])

    def step_native(self, src: torch.Tensor, dst: torch.Tensor, comm_ptr=None):
        """The same pass through the library's own sharded entry (pxl_reproject_sharded_step_*): RCCL send/recv
        issued by the library straight from/into the resident buffer (no staging copies, no Python in the loop),
        interior rows while the halo travels.  comm_ptr: an ncclComm_t (default: this job's RCCL group)."""
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        if self._xfers is None:
            self._xfers = (_lib.xfer_arr(self.sends), _lib.xfer_arr(self.recvs))
        if comm_ptr is None and (self.sends or self.recvs):
            comm_ptr = self.rccl_comm_ptr()
        fn = lib.pxl_reproject_sharded_step_f32 if src.dtype == torch.float32 else lib.pxl_reproject_sharded_step_f64
        own_lo, own_hi = self.own[self.rank]
        with torch.cuda.device(self.device):
            s = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            _lib.check(fn(self.plan._h, C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), own_lo, own_hi - own_lo,
                          C.cast(self._xfers[0], C.c_void_p), len(self.sends), C.cast(self._xfers[1], C.c_void_p), len(self.recvs),
                          C.c_void_p(comm_ptr or 0), s))
        return dst

    def step(self, src: torch.Tensor, dst: torch.Tensor, events=None):
        """One pass: exchange halos, reproject this rank's output strip.  Asynchronous on the current stream.
        events = (start, end) torch.cuda.Events recorded around the dominant kernel launch (the whole
        strip on one rank, the interior rows when sharded) for per-kernel timing."""
        nrows = self.dst_window[1]
        if self.world == 1 or (not self.sends and not self.recvs):
            self.plan.build_tables()
            if events:
                events[0].record()
            self.plan.execute_rows(src, dst, 0, nrows)
            if events:
                events[1].record()
            return dst
        if self._staging is None:
            self._staging = self.make_staging(src, self.group)
        works = self.start_halo_exchange(src, self._staging, self.group)
        self.plan.build_tables()
        i_lo, i_hi = self.interior
        if i_hi > i_lo:
            if events:
                events[0].record()
            self.plan.execute_rows(src, dst, i_lo, i_hi - i_lo)          # overlaps the halo transfer
            if events:
                events[1].record()
        self.finish_halo_exchange(src, self._staging, works)
        if i_hi > i_lo:
            if i_lo > 0:
                self.plan.execute_rows(src, dst, 0, i_lo)
            if i_hi < nrows:
                self.plan.execute_rows(src, dst, i_hi, nrows - i_hi)
        else:
            if events:
                events[0].record()
            self.plan.execute_rows(src, dst, 0, nrows)
            if events:
                events[1].record()
        return dst
