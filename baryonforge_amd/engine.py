"""
Device engine: owns one libbfg_mi355 context per GPU and turns the Python-level
objects of the reference API (catalog, shell, tabulated model) into the
device-resident inputs of the C-ABI (include/bfg_mi355.h).

PyTorch is used for plumbing only: HBM allocations (torch tensors), the HIP
stream (torch's current stream is handed to the library) and, for multi-GPU
runs, torch.distributed (RCCL).  All arithmetic of the hot path runs in the
hand-written HIP kernels; there is no CPU or eager-PyTorch fallback.
"""
import ctypes as C
import threading
import warnings

import numpy as np

from . import _lib
from .background import Background, RHO_CRITICAL, massdef_params

_contexts = {}
_lock = threading.Lock()


def _torch():
    import torch
    return torch


try:                              # a fast non-cryptographic 128-bit hash where it is installed (20 GB/s), else hashlib (1 GB/s)
    from xxhash import xxh3_128_digest as _digest
except ImportError:               # pragma: no cover
    import hashlib

    def _digest(buf):
        return hashlib.blake2b(buf, digest_size=16).digest()


def _fingerprint(arr):
    """content stamp of an array: shape, dtype and a 128-bit hash of EVERY byte: an in-place edit of any element between two
    process() calls must miss the device-table cache -- the reference re-reads its table on every call.  xxh3 runs at ~20 GB/s
    (0.01 ms for the 240 KB table of the headline, ~1.5 ms for a 30 MB N-dimensional table), and only when the identity check
    of Context.table() has passed.  BFG_TABLE_CACHE=sampled stamps arrays over 4 MiB from 128 pages spread over them plus both
    ends instead (the default of round 5, which an edit of a single cell could slip through: ADVICE r5); BFG_TABLE_CACHE=0 keeps
    no device tables between calls at all; Context.invalidate_tables() drops them explicitly."""
    import os
    a = np.ascontiguousarray(arr)
    b = a.view(np.uint8).reshape(-1)
    n = b.size
    if n > (4 << 20) and os.environ.get("BFG_TABLE_CACHE") == "sampled":
        page, k = 4096, 128
        starts = (np.arange(k, dtype=np.int64) * ((n - page) // (k - 1))) // 8 * 8
        b = np.concatenate([b[s0:s0 + page] for s0 in starts] + [b[n - page:]])
    return (a.shape, a.dtype.str, n, _digest(b))


def require_gpu():
    torch = _torch()
    if not torch.cuda.is_available():
        raise _lib.BFGError("baryonforge_amd needs an AMD MI355X (gfx950) GPU: torch.cuda.is_available() is False. "
                            "There is no CPU fallback for the shell paint / baryonify path.")
    return torch


class Table(object):
    """An interpolation table resident in HBM (bfg_table): (ln(1+z), ln M, ln r) + up to 10 p_keys axes.  Up to three extra axes
    are read by the shell kernels directly; with more, the library blends every halo's radial row first (csrc/bfg_ndtable.hpp)
    and runs the same kernels on those rows -- the same bfg_paint_shell / bfg_baryonify_offsets calls either way."""

    def __init__(self, ctx, axes, values, log_values):
        self.ctx = ctx
        axes = [np.ascontiguousarray(a, dtype=np.float64) for a in axes]
        values = np.ascontiguousarray(values, dtype=np.float64)
        if values.ndim != len(axes) or any(a.ndim != 1 or a.size != s for a, s in zip(axes, values.shape)):
            raise ValueError("table axes do not match the shape of the table values")
        if values.ndim - 1 > _lib.BFG_ND_MAX_OUTER:
            raise NotImplementedError(f"tables with more than {_lib.BFG_ND_MAX_OUTER - 2} extra (p_keys) dimensions "
                                      "are not supported on the MI355X path")
        self.ndim = values.ndim
        self.shape = values.shape
        shape = (C.c_int64 * self.ndim)(*values.shape)
        ax_ptrs = (C.POINTER(C.c_double) * self.ndim)(*[_lib.dptr(a) for a in axes])
        handle = C.c_void_p()
        flags = _lib.BFG_TABLE_LOG_VALUES if log_values else 0
        ctx._on_current_stream()
        _lib.check(ctx.lib.bfg_table_create(ctx.handle, self.ndim, shape, ax_ptrs, _lib.dptr(values), flags,
                                            C.byref(handle)), "bfg_table_create")
        self.handle = handle
        self.log_values = bool(log_values)

    def eval(self, coords):
        """stand-alone read-out at coords [npts, ndim] = (ln(1+z), ln M, ln r, extras...)"""
        coords = np.ascontiguousarray(coords, dtype=np.float64).reshape(-1, self.ndim)
        out = np.empty(coords.shape[0])
        self.ctx._on_current_stream()
        _lib.check(self.ctx.lib.bfg_table_eval(self.ctx.handle, self.handle, coords.shape[0], _lib.dptr(coords),
                                               _lib.dptr(out)), "bfg_table_eval")
        return out

    def __del__(self):
        try:
            if self.handle and self.ctx.handle:
                self.ctx.lib.bfg_table_destroy(self.ctx.handle, self.handle)
        except Exception:
            pass


class Spline(object):
    """D_A(z) cubic spline resident in HBM (bfg_spline); HealpixRunner.py:297-299."""

    def __init__(self, ctx, knots, coef):
        knots = np.ascontiguousarray(knots, dtype=np.float64)
        coef = np.ascontiguousarray(coef, dtype=np.float64)
        assert coef.shape == (4, knots.size - 1)
        self.ctx = ctx
        handle = C.c_void_p()
        ctx._on_current_stream()
        _lib.check(ctx.lib.bfg_spline_create(ctx.handle, knots.size, _lib.dptr(knots), _lib.dptr(coef),
                                             C.byref(handle)), "bfg_spline_create")
        self.handle = handle

    def __del__(self):
        try:
            if self.handle and self.ctx.handle:
                self.ctx.lib.bfg_spline_destroy(self.ctx.handle, self.handle)
        except Exception:
            pass


class Context(object):
    """One bfg_ctx per GPU.  Every call that enqueues work first binds the context to torch's CURRENT stream on that
    GPU (`_on_current_stream`), so the kernels are ordered with the torch ops around them (zero-fills, copies,
    collectives) also inside `with torch.cuda.stream(s)` blocks or after the caller changed the current stream; the
    library orders its own work across such a change with an event (bfg_ctx_set_stream)."""

    def __init__(self, device_index):
        torch = require_gpu()
        self.lib = _lib.load()
        self.device_index = int(device_index)
        self.device = torch.device("cuda", self.device_index)
        with torch.cuda.device(self.device):
            self._stream_ptr = int(torch.cuda.current_stream(self.device).cuda_stream)
            handle = C.c_void_p()
            _lib.check(self.lib.bfg_ctx_create(self.device_index, C.c_void_p(self._stream_ptr),
                                               C.byref(handle)), "bfg_ctx_create")
        self.handle = handle
        self._table_cache = {}
        self._spline_cache = {}
        self._inflight = {}                # ticket -> tensor a collective on the communication stream still uses
        self.comm_world = 1
        self.comm_rank = 0

    def _on_current_stream(self):
        ptr = int(_torch().cuda.current_stream(self.device).cuda_stream)
        if ptr != self._stream_ptr:
            _lib.check(self.lib.bfg_ctx_set_stream(self.handle, C.c_void_p(ptr)), "bfg_ctx_set_stream")
            self._stream_ptr = ptr

    # ---- multi-GPU: RCCL communicator of this context (bfg_comm_*) ----------------------
    def comm_init(self, dist=None, group=None):
        """Build the context's RCCL communicator over the ranks of a torch.distributed process group (any backend: the
        128-byte id travels through the group's object broadcast).  Needs one GPU per rank."""
        if dist is None:
            import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        ident = [None]
        if rank == 0:
            buf = C.create_string_buffer(_lib.BFG_COMM_ID_BYTES)
            _lib.check(self.lib.bfg_comm_unique_id(buf, _lib.BFG_COMM_ID_BYTES), "bfg_comm_unique_id")
            ident[0] = buf.raw
        dist.broadcast_object_list(ident, src=0, group=group)
        self.comm_init_id(ident[0], rank, world)

    def comm_init_id(self, ident, rank, world):
        self._on_current_stream()
        _lib.check(self.lib.bfg_comm_init(self.handle, ident, len(ident), int(rank), int(world)), "bfg_comm_init")
        self.comm_world, self.comm_rank = int(world), int(rank)

    def comm_destroy(self):
        _lib.check(self.lib.bfg_comm_destroy(self.handle), "bfg_comm_destroy")
        self.comm_world, self.comm_rank = 1, 0

    def allreduce(self, d_tensor):
        """in-place sum over the ranks of the context's communicator, asynchronous on the current stream"""
        assert d_tensor.is_contiguous() and d_tensor.dtype == _torch().float64
        self._on_current_stream()
        _lib.check(self.lib.bfg_allreduce_f64(self.handle, C.c_void_p(d_tensor.data_ptr()), d_tensor.numel()),
                   "bfg_allreduce_f64")

    def allreduce_begin(self, d_tensor):
        """the same sum on the context's communication stream (overlaps the work enqueued next); returns the collective's
        ticket for comm_wait.  The tensor is kept referenced until a comm_wait covers the ticket (the communication stream
        is not one torch's caching allocator knows about)."""
        assert d_tensor.is_contiguous() and d_tensor.dtype == _torch().float64
        self._on_current_stream()
        ticket = C.c_int64(0)
        _lib.check(self.lib.bfg_allreduce_f64_begin(self.handle, C.c_void_p(d_tensor.data_ptr()), d_tensor.numel(),
                                                    C.byref(ticket)), "bfg_allreduce_f64_begin")
        if ticket.value:
            self._inflight[ticket.value] = d_tensor
        return ticket.value

    def reduce_scatter_begin(self, d_tensor):
        """in place on the communication stream: rank r ends up owning [r n / world, (r + 1) n / world); returns a ticket"""
        assert d_tensor.is_contiguous() and d_tensor.dtype == _torch().float64
        self._on_current_stream()
        ticket = C.c_int64(0)
        _lib.check(self.lib.bfg_reduce_scatter_f64_begin(self.handle, C.c_void_p(d_tensor.data_ptr()), d_tensor.numel(),
                                                         C.byref(ticket)), "bfg_reduce_scatter_f64_begin")
        if ticket.value:
            self._inflight[ticket.value] = d_tensor
        return ticket.value

    def comm_wait(self, ticket=0):
        """the current stream waits for the collective `ticket` (and those begun before it); 0: every collective begun so far"""
        self._on_current_stream()
        _lib.check(self.lib.bfg_comm_wait(self.handle, int(ticket)), "bfg_comm_wait")
        # the stream now orders every later use (and torch's reuse of the memory) after the collective
        for k in [k for k in self._inflight if ticket == 0 or k <= ticket]:
            del self._inflight[k]

    def reduce_scatter(self, d_tensor):
        """in place: rank r ends up owning the summed elements [r n / world, (r + 1) n / world)"""
        assert d_tensor.is_contiguous() and d_tensor.dtype == _torch().float64
        self._on_current_stream()
        _lib.check(self.lib.bfg_reduce_scatter_f64(self.handle, C.c_void_p(d_tensor.data_ptr()), d_tensor.numel()),
                   "bfg_reduce_scatter_f64")

    def allgather(self, d_tensor):
        assert d_tensor.is_contiguous() and d_tensor.dtype == _torch().float64
        self._on_current_stream()
        _lib.check(self.lib.bfg_allgather_f64(self.handle, C.c_void_p(d_tensor.data_ptr()), d_tensor.numel()),
                   "bfg_allgather_f64")

    # ---- device info -------------------------------------------------------------
    def device_info(self):
        name = C.create_string_buffer(256)
        ncu, lds, hbm = C.c_int(), C.c_int(), C.c_int64()
        _lib.check(self.lib.bfg_ctx_device_info(self.handle, name, 256, C.byref(ncu), C.byref(lds), C.byref(hbm)))
        return {"name": name.value.decode(), "n_cu": ncu.value, "lds_per_cu": lds.value, "hbm_bytes": hbm.value}

    def synchronize(self):
        self._on_current_stream()
        _lib.check(self.lib.bfg_ctx_synchronize(self.handle), "bfg_ctx_synchronize")

    # ---- uploads ---------------------------------------------------------------------
    def to_device(self, array):
        torch = _torch()
        return torch.from_numpy(np.ascontiguousarray(array, dtype=np.float64)).to(self.device)

    def zeros(self, *shape):
        torch = _torch()
        return torch.zeros(*shape, dtype=torch.float64, device=self.device)

    def empty(self, *shape):
        """uninitialised HBM for outputs a call defines completely (BFG_SHELL_OUT_OVERWRITE); BFG_POISON=1 (the GPU tests)
        fills it with NaN so that a pixel the kernels fail to write cannot go unnoticed"""
        import os
        torch = _torch()
        if os.environ.get("BFG_POISON"):
            return torch.full(shape, float("nan"), dtype=torch.float64, device=self.device)
        return torch.empty(*shape, dtype=torch.float64, device=self.device)

    def copy_stream(self):
        """a stream of this context's GPU for device <-> host copies that overlap the kernels of the current stream"""
        if getattr(self, "_copy_stream", None) is None:
            self._copy_stream = _torch().cuda.Stream(device=self.device)
        return self._copy_stream

    def upload_stream(self):
        """a second copy stream, for host -> device copies that overlap kernels AND device -> host copies (PCIe is full duplex)"""
        if getattr(self, "_upload_stream", None) is None:
            self._upload_stream = _torch().cuda.Stream(device=self.device)
        return self._upload_stream

    def copy_to_pinned(self, h_pinned, d_tensor):
        """d_tensor -> h_pinned (a page-locked torch CPU tensor) by a copy kernel on torch's CURRENT stream
        (bfg_copy_to_mapped_host; the context stays bound to the stream its kernels run on): unlike a second DMA copy it
        overlaps a host -> device copy running on another stream"""
        assert h_pinned.is_pinned() and h_pinned.is_contiguous() and d_tensor.is_contiguous()
        assert h_pinned.numel() * h_pinned.element_size() == d_tensor.numel() * d_tensor.element_size()
        stream = int(_torch().cuda.current_stream(self.device).cuda_stream)
        _lib.check(self.lib.bfg_copy_to_mapped_host(self.handle, C.c_void_p(stream), C.c_void_p(h_pinned.data_ptr()),
                                                    C.c_void_p(d_tensor.data_ptr()),
                                                    d_tensor.numel() * d_tensor.element_size()), "bfg_copy_to_mapped_host")

    def to_host(self, d_tensor):
        """device tensor -> numpy array.  Up to 1 GiB the copy lands in page-locked memory from torch's caching host
        allocator and the array is a view of it: 1.8 ms instead of 11 ms for a 101 MB map once a block is being
        recycled (the caller dropped an earlier result); a first-time block costs what the pageable copy costs."""
        torch = _torch()
        if d_tensor.numel() * d_tensor.element_size() > (1 << 30):
            return d_tensor.cpu().numpy()
        try:
            h = torch.empty(d_tensor.shape, dtype=d_tensor.dtype, pin_memory=True)
        except RuntimeError:                                   # no page-locked memory to be had (memlock limit)
            return d_tensor.cpu().numpy()
        h.copy_(d_tensor)
        return h.numpy()

    def table(self, axes, values, log_values, cache_key=None):
        """values: ndarray, or a zero-argument callable producing it (only called on a cache miss).

        cache_key: a tuple of the objects the table was made from (model, tag string, raw table array).  The entry
        keeps those objects alive and is only reused for the very same objects (`is`) with unchanged contents (a
        fingerprint of the array and of the axes): keys made of bare id() values would hand a stale table to a new model
        that happens to be allocated where a collected one used to live."""
        import os
        axes = [np.ascontiguousarray(a, dtype=np.float64) for a in axes]
        if cache_key is not None and os.environ.get("BFG_TABLE_CACHE") == "0":
            cache_key = None
        if cache_key is not None:
            ident = tuple(k if isinstance(k, str) else id(k) for k in cache_key)
            stamp = tuple(_fingerprint(k) for k in cache_key if isinstance(k, np.ndarray)) + \
                tuple(_fingerprint(a) for a in axes) + (bool(log_values),)
            hit = self._table_cache.get(ident)
            if hit is not None and hit[2] == stamp and all(a is b for a, b in zip(hit[1], cache_key)):
                return hit[0]
        if callable(values):
            values = values()
        t = Table(self, axes, values, log_values)
        if cache_key is not None:
            if len(self._table_cache) > 8:
                self._table_cache.clear()
            self._table_cache[ident] = (t, tuple(cache_key), stamp)
        return t

    def invalidate_tables(self):
        """forget every device table cached by table(): the next process() call of any model uploads its table afresh"""
        self._table_cache.clear()

    def da_spline(self, background, z_max):
        """D_a = CubicSpline(linspace(0, z_max + 0.1, 1000), D_A(1/(1+z)))   (HealpixRunner.py:297-299); cached per
        (background parameters, z_max): consecutive shells of one light cone reuse it"""
        key = (float(z_max),) + tuple(float(getattr(background, k)) for k in ("Omega_m", "Omega_l", "Omega_r", "w0", "h"))
        hit = self._spline_cache.get(key)
        if hit is not None:
            return hit
        from scipy import interpolate
        z_t = np.linspace(0, z_max + 0.1, 1000)
        cs = interpolate.CubicSpline(z_t, background.angular_diameter_distance(1 / (1 + z_t)))
        sp = Spline(self, cs.x, cs.c)
        if len(self._spline_cache) > 16:
            self._spline_cache.clear()
        self._spline_cache[key] = sp
        return sp

    # ---- hot path --------------------------------------------------------------------
    @staticmethod
    def massdef_struct(background, mass_def):
        Delta, rho_type = massdef_params(mass_def)
        s = _lib.MassDefStruct()
        s.Omega_m, s.Omega_l, s.Omega_r = background.Omega_m, background.Omega_l, background.Omega_r
        s.w0, s.h, s.rho_crit0_h2 = background.w0, background.h, RHO_CRITICAL
        s.Delta, s.rho_type = Delta, 0 if rho_type == "critical" else 1
        return s

    def shell_args(self, nside, d_catalog, n_halo, cat_stride, n_extra, epsilon_max, runner_md, model_md=None,
                   model_epsilon_max=0.0, rdelta_sampling=False, include_pixel_size=False, variant="auto",
                   out_is_zero=False, out_overwrite=False, reuse_plan=False):
        a = _lib.ShellArgs()
        a.nside, a.n_halo = int(nside), int(n_halo)
        a.d_catalog = d_catalog.data_ptr() if n_halo else None
        a.cat_stride, a.n_extra = int(cat_stride), int(n_extra)
        a.epsilon_max = float(epsilon_max)
        a.runner_md = runner_md
        a.model_md = model_md if model_md is not None else runner_md
        a.model_epsilon_max = float(model_epsilon_max)
        a.rdelta_sampling = int(bool(rdelta_sampling))
        a.include_pixel_size = int(bool(include_pixel_size))
        a.variant = _lib.VARIANTS[variant]
        # out_is_zero: the caller cleared the output (tiles are stored, not added); out_overwrite: the output is uninitialised
        # memory and the call defines all of it (no clearing pass at all on the tile path)
        # reuse_plan: the caller vouches that d_catalog holds the records of this context's previous shell call (BFG_SHELL_REUSE_PLAN:
        # another model on the same grid over the same catalog runs the tile kernels only); same_catalog() is how the runners know
        a.flags = (_lib.SHELL_OUT_IS_ZERO if out_is_zero else 0) | (_lib.SHELL_OUT_OVERWRITE if out_overwrite else 0) | \
            (_lib.SHELL_REUSE_PLAN if reuse_plan else 0)
        return a

    def same_catalog(self, d_catalog):
        """True if `d_catalog` is the very tensor object the previous shell call of this context was given (and so, the containers
        never writing to their device copies, the same records): what entitles a runner to set reuse_plan.  The tensor is kept
        referenced, so its address cannot be recycled for other data in between.  BFG_PLAN_REUSE=0 switches the short cut off."""
        import os
        same = getattr(self, "_plan_cat", None) is d_catalog and os.environ.get("BFG_PLAN_REUSE", "1") != "0"
        self._plan_cat = d_catalog
        return same

    def plan_reuses(self):
        """shell calls of this context that ran on a reused plan (bfg_plan_reuses)"""
        n = C.c_int64(0)
        _lib.check(self.lib.bfg_plan_reuses(self.handle, C.byref(n)), "bfg_plan_reuses")
        return n.value

    def _sliced(self, fn_name, args, table, spline, d_out, slices, on_slice):
        """bfg_*_sliced: on_slice(k, n, elem_begin, elem_end) is called on this thread after the k-th slice of the output has
        been enqueued (flat element range of d_out; final once the current stream gets there)"""
        failure = []

        def cb(_user, k, n, lo, hi):
            try:
                on_slice(int(k), int(n), int(lo), int(hi))
                return 0
            except BaseException as exc:          # never unwind through the C frames
                failure.append(exc)
                return 1
        c_cb = _lib.SLICE_FN(cb)
        status = getattr(self.lib, fn_name)(self.handle, C.byref(args), table.handle, spline.handle,
                                            C.c_void_p(d_out.data_ptr()), int(slices), c_cb, None)
        if failure:
            raise failure[0]
        _lib.check(status, fn_name)

    def paint_shell(self, args, table, spline, d_map, slices=1, on_slice=None):
        self._on_current_stream()
        if on_slice is not None:
            return self._sliced("bfg_paint_shell_sliced", args, table, spline, d_map, slices, on_slice)
        _lib.check(self.lib.bfg_paint_shell(self.handle, C.byref(args), table.handle, spline.handle,
                                            C.c_void_p(d_map.data_ptr())), "bfg_paint_shell")

    def baryonify_offsets(self, args, table, spline, d_offsets, slices=1, on_slice=None):
        self._on_current_stream()
        if on_slice is not None:
            return self._sliced("bfg_baryonify_offsets_sliced", args, table, spline, d_offsets, slices, on_slice)
        _lib.check(self.lib.bfg_baryonify_offsets(self.handle, C.byref(args), table.handle, spline.handle,
                                                  C.c_void_p(d_offsets.data_ptr())), "bfg_baryonify_offsets")

    # ---- models that are not tabulated: the geometry around a host-evaluated callable (csrc/bfg_enum.hpp) --------------
    def disc_count(self, args, spline, fallback4=False):
        """int64[n_halo] device tensor: pixels of every halo's disc (bfg_disc_enumerate_count)"""
        torch = _torch()
        self._on_current_stream()
        counts = torch.zeros(int(args.n_halo), dtype=torch.int64, device=self.device)
        _lib.check(self.lib.bfg_disc_enumerate_count(self.handle, C.byref(args), spline.handle, int(bool(fallback4)),
                                                     C.c_void_p(counts.data_ptr())), "bfg_disc_enumerate_count")
        return counts

    def disc_enumerate(self, args, spline, fallback4=False):
        """(counts, base, pix, r_com, halo) device tensors for the halos of `args`: entry e of halo j, base[j] <= e <
        base[j] + counts[j], is RING pixel pix[e] of its disc at comoving distance r_com[e] = r_sep / a_j
        (HealpixRunner.py:460-469; fallback4: the < 4 pixel rule of :333-334).  Synchronises once (the total)."""
        torch = _torch()
        n = int(args.n_halo)
        counts = self.disc_count(args, spline, fallback4)
        incl = torch.cumsum(counts, 0)
        base = incl - counts
        total = int(incl[-1].item()) if n else 0
        pix = torch.empty(total, dtype=torch.int64, device=self.device)
        r_com = torch.empty(total, dtype=torch.float64, device=self.device)
        halo = torch.empty(total, dtype=torch.int32, device=self.device)
        if total:
            _lib.check(self.lib.bfg_disc_enumerate(self.handle, C.byref(args), spline.handle, int(bool(fallback4)),
                                                   C.c_void_p(base.data_ptr()), C.c_void_p(pix.data_ptr()),
                                                   C.c_void_p(r_com.data_ptr()), C.c_void_p(halo.data_ptr())),
                       "bfg_disc_enumerate")
        return counts, base, pix, r_com, halo

    def map_add_values(self, d_map, d_pix, d_val):
        self._on_current_stream()
        _lib.check(self.lib.bfg_map_add_values(self.handle, C.c_void_p(d_map.data_ptr()), C.c_void_p(d_pix.data_ptr()),
                                               C.c_void_p(d_val.data_ptr()), int(d_pix.numel())), "bfg_map_add_values")

    def offsets_add_displacements(self, args, spline, d_pix, d_halo, d_disp, d_offsets):
        self._on_current_stream()
        _lib.check(self.lib.bfg_offsets_add_displacements(self.handle, C.byref(args), spline.handle,
                                                          C.c_void_p(d_pix.data_ptr()), C.c_void_p(d_halo.data_ptr()),
                                                          C.c_void_p(d_disp.data_ptr()), int(d_pix.numel()),
                                                          C.c_void_p(d_offsets.data_ptr())),
                   "bfg_offsets_add_displacements")

    def regrid_shell(self, nside, d_offsets, d_in_map, d_out_map, d_sums=None):
        self._on_current_stream()
        _lib.check(self.lib.bfg_regrid_shell(self.handle, int(nside), C.c_void_p(d_offsets.data_ptr()),
                                             C.c_void_p(d_in_map.data_ptr()), C.c_void_p(d_out_map.data_ptr()),
                                             C.c_void_p(d_sums.data_ptr()) if d_sums is not None else None),
                   "bfg_regrid_shell")

    def regrid_shell_bands(self, nside, d_offsets, d_in_map, d_out_map, d_sums3, band_lo, band_hi, clear_sums=False):
        """bfg_regrid_shell_bands: the sources of ring bands [band_lo, band_hi); d_sums3 = {sum(in), sum(deposits), far deposits}"""
        self._on_current_stream()
        _lib.check(self.lib.bfg_regrid_shell_bands(self.handle, int(nside), C.c_void_p(d_offsets.data_ptr()),
                                                   C.c_void_p(d_in_map.data_ptr()), C.c_void_p(d_out_map.data_ptr()),
                                                   C.c_void_p(d_sums3.data_ptr()) if d_sums3 is not None else None,
                                                   int(band_lo), int(band_hi), 1 if clear_sums else 0), "bfg_regrid_shell_bands")

    def baryonify_snapshot(self, d_part, d_halo, ndim, L, a, epsilon_max, runner_md, model_md, model_epsilon_max,
                           rdelta_sampling, n_extra, table, d_out):
        """bfg_baryonify_snapshot_strided: d_part / d_out float64[n, ndim] tensors or column views of wider record
        tensors (inner stride 1, any row stride), d_halo float64[n_halo, 5 + n_extra] (M, lnM, x, y, z, extras)"""
        self._on_current_stream()
        assert d_part.stride(1) == 1 and d_out.stride(1) == 1 and d_out.shape == d_part.shape
        args = _lib.SnapshotArgs()
        args.ndim, args.rdelta_sampling = int(ndim), int(bool(rdelta_sampling))
        args.n_part, args.n_halo = int(d_part.shape[0]), int(d_halo.shape[0])
        args.L, args.a = float(L), float(a)
        args.d_part = d_part.data_ptr() if args.n_part else None
        args.d_halo = d_halo.data_ptr() if args.n_halo else None
        args.halo_stride, args.n_extra = int(d_halo.shape[1]), int(n_extra)
        args.epsilon_max = float(epsilon_max)
        args.runner_md, args.model_md = runner_md, model_md
        args.model_epsilon_max = float(model_epsilon_max)
        _lib.check(self.lib.bfg_baryonify_snapshot_strided(self.handle, C.byref(args), table.handle,
                                                           C.c_void_p(d_out.data_ptr()), int(d_part.stride(0)),
                                                           int(d_out.stride(0))), "bfg_baryonify_snapshot_strided")

    def grid_args(self, ndim, npix, d_bins, d_halo, a, epsilon_max, runner_md, model_md=None, model_epsilon_max=0.0,
                  rdelta_sampling=False, n_extra=0, d_rmat=None):
        g = _lib.GridArgs()
        g.ndim, g.rdelta_sampling, g.n_halo, g.npix = int(ndim), int(bool(rdelta_sampling)), int(d_halo.shape[0]), int(npix)
        g.a = float(a)
        g.d_bins = d_bins.data_ptr()
        g.d_halo = d_halo.data_ptr() if g.n_halo else None
        g.halo_stride, g.n_extra = int(d_halo.shape[1]), int(n_extra)
        g.epsilon_max = float(epsilon_max)
        g.runner_md = runner_md
        g.model_md = model_md if model_md is not None else runner_md
        g.model_epsilon_max = float(model_epsilon_max)
        g.d_rmat = d_rmat.data_ptr() if d_rmat is not None else None
        return g

    def paint_grid(self, args, table, d_map):
        self._on_current_stream()
        _lib.check(self.lib.bfg_paint_grid(self.handle, C.byref(args), table.handle, C.c_void_p(d_map.data_ptr())),
                   "bfg_paint_grid")

    def baryonify_grid_offsets(self, args, table, d_offsets):
        self._on_current_stream()
        _lib.check(self.lib.bfg_baryonify_grid_offsets(self.handle, C.byref(args), table.handle,
                                                       C.c_void_p(d_offsets.data_ptr())), "bfg_baryonify_grid_offsets")

    def regrid_grid(self, ndim, npix, d_offsets, d_in_map, d_out_map):
        self._on_current_stream()
        _lib.check(self.lib.bfg_regrid_grid(self.handle, int(ndim), int(npix), C.c_void_p(d_offsets.data_ptr()),
                                            C.c_void_p(d_in_map.data_ptr()), C.c_void_p(d_out_map.data_ptr())),
                   "bfg_regrid_grid")

    def deposit_grid(self, d_pos, d_mass, L, n_grid, mode="ngp"):
        """mass map float64[n_grid]*ndim of particles d_pos float64[n, ndim] (d_mass float64[n] or None); both may be
        column views of one record tensor (bfg_deposit_grid_strided)"""
        self._on_current_stream()
        ndim = int(d_pos.shape[1])
        assert d_pos.stride(1) == 1
        d_grid = self.zeros(*([int(n_grid)] * ndim))
        _lib.check(self.lib.bfg_deposit_grid_strided(
            self.handle, ndim, int(d_pos.shape[0]), C.c_void_p(d_pos.data_ptr()), int(d_pos.stride(0)),
            C.c_void_p(d_mass.data_ptr()) if d_mass is not None else None,
            int(d_mass.stride(0)) if d_mass is not None else 1, float(L), int(n_grid), {"ngp": 0, "cic": 1}[mode],
            C.c_void_p(d_grid.data_ptr())), "bfg_deposit_grid_strided")
        return d_grid

    def build_displacement_table(self, geometry, r_int, dens_dmo, dens_dmb, r, rdelta=None, rdelta_range=None):
        """bfg_build_displacement_table: dens_* float64[n_rows, n_int] (host), returns (d[n_rows, nr], status[n_rows])"""
        self._on_current_stream()
        r_int = np.ascontiguousarray(r_int, dtype=np.float64)
        r = np.ascontiguousarray(r, dtype=np.float64)
        d_dmo, d_dmb = self.to_device(dens_dmo), self.to_device(dens_dmb)
        n_rows, n_int = d_dmo.shape
        assert d_dmb.shape == d_dmo.shape and r_int.size == n_int
        d_out = self.zeros(n_rows, r.size)
        status = np.zeros(n_rows, dtype=np.int32)
        dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
        if rdelta is not None:
            rdelta = np.ascontiguousarray(rdelta, dtype=np.float64)
            rdelta_range = np.ascontiguousarray(rdelta_range, dtype=np.float64)
            assert rdelta.size == n_rows and rdelta_range.size == r.size
        _lib.check(self.lib.bfg_build_displacement_table(
            self.handle, int(geometry), int(n_rows), int(n_int), dp(r_int), C.c_void_p(d_dmo.data_ptr()),
            C.c_void_p(d_dmb.data_ptr()), int(r.size), dp(r), dp(rdelta) if rdelta is not None else None,
            dp(rdelta_range) if rdelta is not None else None, C.c_void_p(d_out.data_ptr()),
            status.ctypes.data_as(C.POINTER(C.c_int32))), "bfg_build_displacement_table")
        return d_out.cpu().numpy(), status

    def absmax_sum(self, d_x):
        self._on_current_stream()
        amax, s = C.c_double(), C.c_double()
        _lib.check(self.lib.bfg_reduce_absmax_sum(self.handle, d_x.numel(), C.c_void_p(d_x.data_ptr()),
                                                  C.byref(amax), C.byref(s)), "bfg_reduce_absmax_sum")
        return amax.value, s.value

    def stats_reset(self):
        self._on_current_stream()
        _lib.check(self.lib.bfg_stats_reset(self.handle))

    def stats(self):
        self._on_current_stream()
        st = _lib.Stats()
        _lib.check(self.lib.bfg_stats_read(self.handle, C.byref(st)))
        return {"pixel_updates": int(st.pixel_updates), "halos_out_of_table": int(st.halos_out_of_table),
                "pixels_out_of_table": int(st.pixels_out_of_table), "halos_fallback4": int(st.halos_fallback4),
                "warn_mask": int(st.warn_mask), "fallback_halos": int(st.halos_scatter_fallback)}

    def timing_enable(self, on=True, which=None):
        """hipEvents around the context's kernels; `which`: the kernel classes (bfg_timing_read's indices) to time, default all"""
        _lib.check(self.lib.bfg_timing_enable(self.handle, int(bool(on))))
        if on and which is not None:
            _lib.check(self.lib.bfg_timing_select(self.handle, sum(1 << int(k) for k in which)))

    def timing_read(self, which):
        ms, n = C.c_double(), C.c_int64()
        _lib.check(self.lib.bfg_timing_read(self.handle, int(which), C.byref(ms), C.byref(n)))
        return ms.value, n.value


def pinned_empty(n):
    """float64 numpy array of n elements in page-locked host memory (for PaintProfilesShell.process(out=...)): device -> host
    copies into it run at PCIe speed and asynchronously"""
    return require_gpu().empty(int(n), dtype=_torch().float64, pin_memory=True).numpy()


_registered = {}      # data address -> (bytes, the array: kept alive while its pages are locked)
_PAGE = 4096


def aligned_empty(n, dtype=np.float64):
    """numpy array of n elements in page-aligned anonymous memory of its own (mmap), a whole number of pages long: what pin()
    accepts.  (pinned_empty() is the simpler choice where the array can be allocated page-locked in the first place.)"""
    import mmap
    dtype = np.dtype(dtype)
    nbytes = max(1, int(n) * dtype.itemsize)
    buf = mmap.mmap(-1, (nbytes + _PAGE - 1) // _PAGE * _PAGE)
    return np.frombuffer(buf, dtype=dtype, count=int(n))


def pin_ok(array):
    """True if pin() may page-lock this array in place: C-contiguous, starting on a page boundary, inside a buffer that owns every
    page it touches (aligned_empty, a private anonymous mmap, posix_memalign'd whole pages)"""
    a = array
    if not (isinstance(a, np.ndarray) and a.flags["C_CONTIGUOUS"] and a.nbytes > 0 and a.ctypes.data % _PAGE == 0):
        return False
    if a.nbytes % _PAGE == 0:
        return True
    import mmap
    base = a                                          # a view of a whole-page buffer (aligned_empty: ndarray -> memoryview -> mmap)
    while True:
        nxt = base.base if isinstance(base, np.ndarray) else base.obj if isinstance(base, memoryview) else None
        if nxt is None:
            break
        base = nxt
    return isinstance(base, mmap.mmap)


def pin(array):
    """Page-lock the memory of a C-contiguous numpy array IN PLACE (hipHostRegister) and return it: host <-> device copies of it
    then run asynchronously at PCIe speed instead of through the runtime's staging buffers.  Registering costs about as much as one
    copy of the array, so it pays for maps that are used more than once.  unpin(array) releases the pages; the registry keeps the
    array alive until then.
    ONLY page-aligned buffers that own all their pages are accepted (aligned_empty(), a private mmap; pin_ok() tells): registered
    pageable memory reaches the GPU through the kernel's user-pointer mapping of whole pages, and an ordinary numpy array shares
    its first and last page -- below glibc's mmap threshold all of them -- with other heap objects, which the allocator may trim or
    the kernel migrate under the mapping.  In round 5's soak two of ~5000 shells registered from the HEAP ended in a GPU memory
    fault inside an asynchronous DMA copy (no kernel running, faulting address in the process heap: profiles/r05_soak.txt); heap
    arrays are therefore refused with a ValueError (LightconeShell(pinned="inplace") then makes a page-locked copy instead)."""
    torch = require_gpu()
    a = array
    if not (isinstance(a, np.ndarray) and a.flags["C_CONTIGUOUS"] and a.nbytes > 0):
        raise ValueError("pin() needs a non-empty C-contiguous numpy array")
    ptr = a.ctypes.data
    if ptr in _registered or torch.from_numpy(a.reshape(-1).view(np.uint8)).is_pinned():
        return array
    if not pin_ok(a):
        raise ValueError("pin() only page-locks page-aligned buffers that own every page they touch (engine.aligned_empty(n), a "
                         "private mmap); this array lives in the allocator's heap -- use pinned_copy() / pinned_empty() instead")
    nbytes = (a.nbytes + _PAGE - 1) // _PAGE * _PAGE
    # drain the device first: the runtime page-locks the source of a large pageable copy in place and may still hold that lock (it
    # lets go at the queue's next fence) -- a registration nested inside such a transient lock would share its GPU mapping
    torch.cuda.synchronize()
    rc = torch.cuda.cudart().cudaHostRegister(ptr, nbytes, 0)
    if int(rc) != 0:
        raise _lib.BFGError(f"hipHostRegister of {nbytes} bytes failed (error {int(rc)})")
    _registered[ptr] = (nbytes, a)
    return array


def unpin(array):
    """release the pages pin() locked (no-op for arrays it did not register)"""
    ptr = array.ctypes.data
    if ptr not in _registered:
        return
    _torch().cuda.synchronize()                      # no copy of ours may still be reading or writing these pages
    rc = _torch().cuda.cudart().cudaHostUnregister(ptr)
    if int(rc) != 0:                                  # the pages stay mapped for the GPU: the registry keeps the array alive
        raise _lib.BFGError(f"hipHostUnregister failed (error {int(rc)})")
    del _registered[ptr]


def pinned_copy(array, dtype=np.float64):
    """a copy of `array` (same shape) in page-locked host memory; float64 by default (what the kernels read), dtype=None keeps the
    array's own dtype"""
    a = np.asarray(array)
    dtype = np.dtype(a.dtype if dtype is None else dtype).newbyteorder("=")      # (FITS columns arrive big-endian: native order here)
    torch = require_gpu()
    out = torch.empty(a.shape, dtype=torch.from_numpy(np.empty(0, dtype)).dtype, pin_memory=True).numpy()
    out[...] = a
    return out


def is_pinned(array):
    """True if host <-> device copies of this numpy array can run asynchronously (page-locked memory)"""
    try:
        return bool(_torch().from_numpy(np.ascontiguousarray(array).reshape(-1)).is_pinned())
    except Exception:
        return False


def get_context(device=None):
    """The process-wide Context of a GPU (default: torch's current device)."""
    torch = require_gpu()
    if device is None:
        index = torch.cuda.current_device()
    elif isinstance(device, int):
        index = device
    else:
        index = torch.device(device).index or 0
    with _lock:
        ctx = _contexts.get(index)
        if ctx is None:
            ctx = _contexts[index] = Context(index)
    return ctx


def emit_fallback_warning(stats):
    """the tile path's silent degradation made visible: halos that took the global-atomic scatter kernel although they had
    work to do (bfg_stats.halos_scatter_fallback)"""
    n = stats.get("fallback_halos", 0)
    if n > 0:
        warnings.warn(f"{n} halos were painted by the slower global-atomic scatter kernel (disc over more than 64 sky "
                      f"tiles, pixel factor outside the fast exp range, or tile pair buffer exhausted); the result is "
                      f"unaffected, the run time is not", UserWarning)


def emit_range_warnings(stats, what="table"):
    """The reference warns per halo when a query leaves the table (BaryonCorrection.py:382-394);
    the kernels return one bit-mask per process() instead."""
    m = stats["warn_mask"]
    if m & _lib.WARN_Z_RANGE:
        warnings.warn(f"Requested redshift range outside {what}'s range "
                      f"({stats['halos_out_of_table']} halos outside the (z, M) hull contribute nothing)", UserWarning)
    if m & _lib.WARN_M_RANGE:
        warnings.warn(f"Requested log_Mass range outside {what}'s range "
                      f"({stats['halos_out_of_table']} halos outside the (z, M) hull contribute nothing)", UserWarning)
    if m & _lib.WARN_R_RANGE:
        warnings.warn(f"Requested Radius range outside {what}'s range "
                      f"({stats['pixels_out_of_table']} pixel queries contribute nothing)", UserWarning)
