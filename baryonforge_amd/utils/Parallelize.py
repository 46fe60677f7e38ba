"""
Parallel wrappers, mirroring BaryonForge/utils/Parallelize.py: `SimpleParallel`
(:8-113) and `SplitJoinParallel` (:116-320).

The reference forks joblib/loky worker processes and pickles Runner objects to
them.  Here the workers are the ranks of a torch.distributed process group --
one process per MI355X, backend "nccl" (= RCCL over xGMI) -- launched with
`python -m torch.distributed.run --nproc-per-node N ...`:

  * SplitJoinParallel shards the halo catalog BY SKY PATCH across the ranks
    (baryonforge_amd.sharding), every rank paints its shard onto a full-size
    private map in its own HBM, and an all-reduce(sum, f64, Npix) replaces the
    parent-side np.sum(outputs, axis=0) of Parallelize.py:318.  The exchange
    is hidden behind the painting in two ways:
      - inside one call the map is handed to the all-reduce in `slices` pieces
        as the tile kernel finishes them (bfg_paint_shell_sliced), so only the
        last piece's exchange is exposed;
      - given a LIST of shell runners (what the reference's SimpleParallel
        takes, Parallelize.py:92-113: the shells of a light cone, or several
        models on one shell) every shell is split over all ranks and shell
        k + 1 is painted while shell k's all-reduce and its copy to the host
        are still in flight (rotating map buffers).
  * Unlike the reference (Parallelize.py:206-209) Baryonify runners ARE
    supported: the offset field Npix x 3 is linear in halos (HealpixRunner.py:355),
    so it is summed across ranks -- a reduce-scatter: every rank only needs the
    summed offsets of the pixel range it regrids --, every rank regrids its own
    pixel range and the output maps are all-reduced.
  * `include_pixel_size` is forwarded to the per-rank runners (the reference
    drops it, Parallelize.py:271 -- a conscious divergence).

Without an initialised process group both wrappers simply run on the one GPU.
"""
import numpy as np

from ..sharding import disc_radius, estimate_disc_pixels, shard_by_sky_patch, shard_by_stripes, stripe_extent

__all__ = ["SimpleParallel", "SplitJoinParallel", "Exchange", "OwnerExchange"]


def _dist():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist
    except Exception:
        pass
    return None


class Exchange(object):
    """The exchange step between the ranks of a torch.distributed process group, on float64 tensors, in place.

    collective = "torch": torch.distributed's own collectives (backend "nccl" = RCCL over xGMI on device tensors; "gloo"
                          for CPU tensors and for rehearsals of the multi-rank path on a one-GPU box, where device
                          tensors are staged through the host);
    collective = "bfg":   the library's communicator (bfg_comm_init / bfg_allreduce_f64[_begin] / bfg_reduce_scatter_f64 of
                          include/bfg_mi355.h: RCCL on the context's streams) -- what a C-ABI consumer without torch uses;
                          needs one GPU per rank.
    """

    def __init__(self, dist, collective="torch", ctx=None):
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.backend = dist.get_backend()
        self.collective = collective
        self.ctx = ctx
        if collective == "bfg":
            if ctx is None:
                from ..engine import get_context
                self.ctx = ctx = get_context()
            if ctx.comm_world != self.world:
                ctx.comm_init(dist)
        elif collective != "torch":
            raise ValueError(f"unknown collective {collective!r}")

    def own_range(self, n):
        """the slice of n elements (n a multiple of the world size) this rank owns after reduce_scatter"""
        assert n % self.world == 0, "the exchanged array must split evenly over the ranks"
        return n // self.world * self.rank, n // self.world * (self.rank + 1)

    def _staged(self, t):
        return t.is_cuda and self.backend == "gloo"

    def allreduce(self, t):
        """t <- sum over ranks (in place), ordered on the current stream"""
        if self.collective == "bfg" and t.is_cuda:
            self.ctx.allreduce(t)
        elif self._staged(t):
            h = t.cpu()
            self.dist.all_reduce(h, op=self.dist.ReduceOp.SUM)
            t.copy_(h)
        else:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t

    def allreduce_begin(self, t):
        """start t <- sum over ranks so that it overlaps what is enqueued next on the current stream; returns a handle for
        `wait`.  t must stay alive (and untouched) until then."""
        if self.collective == "bfg" and t.is_cuda:
            return ("bfg", self.ctx.allreduce_begin(t))         # on the context's communication stream; a ticket
        if self._staged(t):                                     # gloo rehearsal on one GPU: synchronous
            self.allreduce(t)
            return None
        return ("torch", self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, async_op=True))

    def reduce_scatter_begin(self, t):
        """start the in-place reduce-scatter of the flat tensor t (numel a multiple of the world size: rank r ends up owning the
        r-th part); returns a handle for `wait`"""
        n = t.numel()
        assert n % self.world == 0
        if self.collective == "bfg" and t.is_cuda:
            return ("bfg", self.ctx.reduce_scatter_begin(t))
        if self.backend == "nccl" and t.is_cuda:
            import torch
            lo, hi = self.own_range(n)
            out = torch.empty(hi - lo, dtype=t.dtype, device=t.device)
            work = self.dist.reduce_scatter_tensor(out, t, op=self.dist.ReduceOp.SUM, async_op=True)
            return ("torch_rs", (work, out, t[lo:hi]))
        return self.allreduce_begin(t)                             # gloo has no reduce-scatter

    def wait(self, handle):
        """order the current stream (CPU tensors: the host) after the collective of `handle`"""
        if handle is None:
            return
        kind, h = handle
        if kind == "bfg":
            self.ctx.comm_wait(h)
        elif kind == "torch_rs":
            work, out, dest = h
            work.wait()
            dest.copy_(out)
        else:
            h.wait()

    def reduce_scatter(self, t):
        """in place: afterwards the elements own_range(t.numel()) of the flattened t hold the sum over ranks; the rest of
        t is unspecified (partial sums).  Moves half the bytes of an all-reduce."""
        flat = t.view(-1)
        lo, hi = self.own_range(flat.numel())
        if self.collective == "bfg" and t.is_cuda:
            self.ctx.reduce_scatter(flat)
        elif self.backend == "nccl" and t.is_cuda:
            import torch
            out = torch.empty(hi - lo, dtype=flat.dtype, device=flat.device)
            self.dist.reduce_scatter_tensor(out, flat, op=self.dist.ReduceOp.SUM)
            flat[lo:hi] = out
        else:                                                      # gloo has no reduce-scatter
            self.allreduce(t)
        return t


class OwnerExchange(object):
    """The owner-computes join of per-rank maps (the alternative to a full all-reduce; DESIGN.md section 5, column C1).

    The RING-ordered map is cut into `world` equal parts; rank r OWNS part r.  With the catalog sharded by declination stripes
    (sharding.shard_by_stripes) a rank's discs touch its own part plus a border of a few rings -- its EXTENT, known on the host
    from the catalog alone.  The join is then
      1. border exchange: every rank sends what it painted inside another rank's part (extent_s intersected with part_r: a few
         per cent of the map, neighbours only) point to point, and adds what it receives to its own part;
      2. all-gather of the owned parts, if every rank is to hold the whole map (`gather=True`) -- (N - 1) / N of the map per
         rank, HALF the bytes of the all-reduce it replaces.
    Correct for ANY extents (a rank whose discs reach three stripes just sends more); needs Npix divisible by the world size.
    Collectives: torch.distributed point-to-point (batch_isend_irecv) + all_gather_into_tensor; gloo with device tensors (the
    one-GPU rehearsal) is staged through the host.
    """

    def __init__(self, exchange, npix, extent):
        self.ex = exchange
        self.dist = exchange.dist
        self.rank, self.world = exchange.rank, exchange.world
        if npix % self.world:
            raise ValueError("the owner-computes join needs 12 NSIDE^2 divisible by the number of ranks")
        self.npix, self.part = int(npix), int(npix) // self.world
        extents = [None] * self.world
        self.dist.all_gather_object(extents, (int(extent[0]), int(extent[1])))
        self.extents = extents
        # what this rank sends to / receives from every other rank: RING pixel ranges
        self.sends, self.recvs = [], []
        for s in range(self.world):
            if s == self.rank:
                continue
            lo, hi = max(extents[self.rank][0], s * self.part), min(extents[self.rank][1], (s + 1) * self.part)
            if hi > lo:
                self.sends.append((s, lo, hi))
            lo, hi = max(extents[s][0], self.rank * self.part), min(extents[s][1], (self.rank + 1) * self.part)
            if hi > lo:
                self.recvs.append((s, lo, hi))
        self.border_bytes = 8 * sum(hi - lo for _, lo, hi in self.sends)

    def begin(self, t, gather=True):
        """t: this rank's map float64[npix], defined everywhere (zeros where it painted nothing).  Starts the join; after
        wait(handle) t holds the summed map (gather=True) or its owned part t[rank * part : (rank + 1) * part] does."""
        import torch
        staged = t.is_cuda and self.ex.backend == "gloo"
        w = t.cpu() if staged else t
        dist = self.dist
        ops, bufs = [], []
        for s, lo, hi in self.sends:
            ops.append(dist.P2POp(dist.isend, w[lo:hi], s))
        for s, lo, hi in self.recvs:
            b = torch.empty(hi - lo, dtype=w.dtype, device=w.device)
            bufs.append((lo, hi, b))
            ops.append(dist.P2POp(dist.irecv, b, s))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()                                  # nccl: the current stream waits; gloo: the host
        for lo, hi, b in bufs:
            w[lo:hi] += b
        work = None
        if gather:
            own = w[self.rank * self.part:(self.rank + 1) * self.part].clone()
            if staged or not w.is_cuda:                     # gloo: the list form
                parts = [w[r * self.part:(r + 1) * self.part] for r in range(self.world)]
                dist.all_gather(parts, own)
            else:
                work = dist.all_gather_into_tensor(w, own, async_op=True)
        if staged:
            t.copy_(w)
        return ("owner", (work, bufs))                      # (the receive buffers live until the wait)

    def wait(self, handle):
        work, _ = handle[1]
        if work is not None:
            work.wait()


class _DeviceOps(object):
    """The GPU side of one rank: HBM tensors, C-ABI calls through the runners, copies to the host (the default `ops` of
    SplitJoinParallel.process / process_device)."""

    def __init__(self):
        from ..engine import get_context
        self.ctx = get_context()
        self._copy_stream = None

    def new_map(self, npix):
        return self.ctx.empty(npix)

    def paint(self, runner, d_map, slices, on_slice):
        runner.process_device(d_map=d_map, overwrite=True, slices=slices, on_slice=on_slice, sync_stats=False)

    def collect(self, runners):
        """the counters of everything painted since the last collect (one read-back = one synchronisation)"""
        stats = runners[0].collect_stats()
        for r in runners[1:]:
            r.last_stats = stats
        self.ctx.stats_reset()
        return stats

    def reset_stats(self):
        self.ctx.stats_reset()

    def to_host(self, d_map):
        return self.ctx.to_host(d_map)

    def to_host_begin(self, d_map):
        """copy d_map to page-locked host memory on a copy stream of its own, ordered after what the current stream holds;
        returns (host tensor, event): the event marks the end of the copy (before the buffer may be repainted)"""
        import torch
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(device=self.ctx.device)
        try:
            h = torch.empty(d_map.shape, dtype=d_map.dtype, pin_memory=True)
        except RuntimeError:
            h = torch.empty(d_map.shape, dtype=d_map.dtype)
        self._copy_stream.wait_stream(torch.cuda.current_stream(self.ctx.device))
        with torch.cuda.stream(self._copy_stream):
            h.copy_(d_map, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._copy_stream)
        return h, ev

    def wait_event(self, ev):
        """the current stream waits for `ev` (a buffer is about to be repainted)"""
        import torch
        torch.cuda.current_stream(self.ctx.device).wait_event(ev)

    def host_ready(self, h, ev):
        ev.synchronize()
        return h.numpy()


def _plain_baryonify_process():
    from ..Runners.HealpixRunner import BaryonifyShell
    return BaryonifyShell.process


class SimpleParallel(object):
    """Run several independent Runners (Parallelize.py:8-113).

    split=False (default, the reference's scheme): whole runners are dealt round-robin to the ranks of the process group
    and every rank returns the full list of outputs (gathered through the host).
    split=True: every runner's catalog is split over ALL ranks instead (SplitJoinParallel over the list): shell k + 1 is
    painted while shell k's all-reduce is in flight, no rank idles when there are fewer runners than GPUs, and the
    results never leave HBM before they are final."""

    def __init__(self, Runner_list, njobs=-1, seed=42, split=False, **splitjoin_args):
        self.Runner_list = list(Runner_list)
        self.njobs = njobs
        self.seed = seed
        self.split = split
        self.splitjoin_args = splitjoin_args

    def single_run(self, Runner):
        return Runner.process()

    def process(self):
        dist = _dist()
        if self.split:
            return SplitJoinParallel(self.Runner_list, self.njobs, self.seed, **self.splitjoin_args).process()
        if dist is None or dist.get_world_size() == 1:
            if len(self.Runner_list) > 1 and type(self).single_run is SimpleParallel.single_run and \
                    all(hasattr(R, "offsets_device") and type(R).process is _plain_baryonify_process() for R in self.Runner_list):
                # a list of BaryonifyShell runners on one GPU: uploads, kernels and downloads of consecutive shells overlap
                from ..Runners.HealpixRunner import _baryonify_pipelined
                return _baryonify_pipelined(self.Runner_list)
            return [self.single_run(R) for R in self.Runner_list]
        rank, world = dist.get_rank(), dist.get_world_size()
        mine = {i: self.single_run(R) for i, R in enumerate(self.Runner_list) if i % world == rank}
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        out = {}
        for g in gathered:
            out.update(g)
        return [out[i] for i in range(len(self.Runner_list))]


class SplitJoinParallel(object):
    """
    Split a Runner's halo catalog across GPUs and join (sum) the results (Parallelize.py:116-320).

    Parameters
    ----------
    Runner : PaintProfilesShell or BaryonifyShell, or a list of them (every one is split over all ranks; `process()`
        then returns the list of maps)
    njobs : ignored when a process group exists (the world size is used); kept for API parity
    seed : kept for API parity (the sky-patch split is deterministic and needs no shuffle)
    nside_patch, layout : NSIDE of the sky patches and how they are dealt to the ranks (sharding.shard_by_sky_patch)
    collective : "torch" (torch.distributed: RCCL with the nccl backend) or "bfg" (the library's own RCCL communicator,
        bfg_allreduce_f64 & co.); see Exchange
    slices : pieces in which a painted map is handed to the all-reduce while the rest is still being painted
        (bfg_paint_shell_sliced); 1 = one all-reduce after the call
    exchange : "allreduce" (default) or "owner": the owner-computes join (OwnerExchange: declination-stripe shards, border
        exchange + all-gather -- half the bytes of the all-reduce; paint runners; `layout` is then "stripes")
    """

    def __init__(self, Runner, njobs=-1, seed=42, nside_patch=64, layout="interleaved", collective="torch", slices=4,
                 exchange="allreduce"):
        self.is_list = isinstance(Runner, (list, tuple))
        self.Runners = list(Runner) if self.is_list else [Runner]
        self.Runner = Runner
        self.seed = seed
        self.njobs = njobs
        self.nside_patch = nside_patch
        self.layout = layout
        self.collective = collective
        self.slices = max(1, int(slices))
        if exchange not in ("allreduce", "owner"):
            raise ValueError("exchange must be 'allreduce' or 'owner'")
        self.exchange = exchange
        if exchange == "owner":
            self.layout = "stripes"
        self._owner = {}                                                 # id(catalog) -> OwnerExchange
        dist = _dist()
        self.rank = dist.get_rank() if dist else 0
        self.world = dist.get_world_size() if dist else 1
        self.shard_indices_list = []
        self._shards = {}                                                # id(catalog) -> (indices, this rank's sub-catalog)
        self.Runner_list = [self.split_run(R) for R in self.Runners]     # this rank's runner of every input runner
        self.shard_indices = self.shard_indices_list[0] if self.shard_indices_list else np.arange(0)

    def split_run(self, Runner):
        """the Runner of THIS rank for one input Runner"""
        HaloCat = Runner.HaloLightConeCatalog
        cat = HaloCat.cat
        if self.world == 1:
            self.shard_indices_list.append(np.arange(cat.size))
            self._owner_of_runner = getattr(self, "_owner_of_runner", [])
            self._owner_of_runner.append(None)
            return Runner
        # runners that share a catalog object (several models on one shell) share its shard -- and its one device copy
        key = (id(HaloCat), float(Runner.epsilon_max), int(Runner.LightconeShell.NSIDE), id(Runner.mass_def))
        if key not in self._shards:
            w = estimate_disc_pixels(Runner.cosmo, cat["M"], cat["z"], Runner.epsilon_max, Runner.LightconeShell.NSIDE,
                                     Runner.mass_def)
            if self.layout == "stripes":
                shards = shard_by_stripes(cat["ra"], cat["dec"], self.world)
            else:
                shards = shard_by_sky_patch(cat["ra"], cat["dec"], w, self.world, self.nside_patch, layout=self.layout)
            self._shards[key] = (shards[self.rank], HaloCat[shards[self.rank]], HaloCat)   # (keeps HaloCat alive: id() stays unique)
            if self.exchange == "owner" and not hasattr(Runner, "offsets_device"):
                idx = shards[self.rank]
                th = disc_radius(Runner.cosmo, cat["M"][idx], cat["z"][idx], Runner.epsilon_max, Runner.mass_def)
                nside = int(Runner.LightconeShell.NSIDE)
                self._owner[key] = OwnerExchange(Exchange(_dist(), "torch"), 12 * nside * nside,
                                                 stripe_extent(nside, cat["dec"][idx], th))
        idx, New_HaloCatalog, _ = self._shards[key]
        self.shard_indices_list.append(idx)
        self._owner_of_runner = getattr(self, "_owner_of_runner", [])
        self._owner_of_runner.append(self._owner.get(key))
        New_Runner = type(Runner)(New_HaloCatalog, Runner.LightconeShell, Runner.epsilon_max, Runner.model,
                                  Runner.use_ellipticity, Runner.mass_def,
                                  include_pixel_size=Runner.include_pixel_size, verbose=False)
        if hasattr(Runner, "variant"):
            New_Runner.variant = Runner.variant
        # every rank builds the D_A spline the serial run would build (knots up to max(z) of the WHOLE catalog,
        # HealpixRunner.py:297-299): a halo then gets bit-identical scalars whichever rank it lands on
        New_Runner._spline_z_max = float(np.max(cat["z"])) if cat.size else 0.0
        return New_Runner

    def single_run(self, Runner):
        return Runner.process()

    # ---- the pipeline ---------------------------------------------------------------------------------
    def _exchange(self):
        dist = _dist()
        if dist is None or self.world == 1:
            return None
        return Exchange(dist, self.collective)

    def process_device(self, consume=None, ops=None, exchange=True):
        """Paint every runner's shard of this rank, summed over the ranks, leaving the maps ON THE DEVICE.
        (exchange=False: no sum over the ranks -- every rank keeps the map of its own shard.)

        consume=None: returns the list of device maps (one buffer per runner), ordered on the current stream.
        consume=callable(k, d_map): called once per runner, in order, when the current stream holds everything map k
        needs (painting and exchange); the map's buffer is recycled afterwards (two buffers rotate), so the callable must
        enqueue or finish what it does with it.  Returns None.
        Only PaintProfilesShell runners (a BaryonifyShell's regrid needs the exchanged offsets first: see process())."""
        ops = ops or _DeviceOps()
        ex = self._exchange() if exchange else None
        n = len(self.Runner_list)
        nbuf = n if consume is None else min(2, n)
        bufs, pend = [None] * nbuf, [None] * nbuf

        def retire(b):
            if pend[b] is None:
                return
            k, handles = pend[b]
            for h in handles:
                if h is not None and h[0] == "owner":
                    self._owner_of_runner[k].wait(h)
                else:
                    ex.wait(h)
            pend[b] = None
            if consume is not None:
                consume(k, bufs[b])
        ops.reset_stats()
        for k, R in enumerate(self.Runner_list):
            b = k % nbuf
            retire(b)
            if bufs[b] is None:
                bufs[b] = ops.new_map(12 * R.LightconeShell.NSIDE ** 2)
            handles = []
            owner = self._owner_of_runner[k] if ex is not None else None
            if ex is None:
                ops.paint(R, bufs[b], 1, None)
            elif owner is not None:                                 # owner-computes join: borders + all-gather after the call
                ops.paint(R, bufs[b], 1, None)
                handles.append(owner.begin(bufs[b]))
            else:
                buf = bufs[b]
                ops.paint(R, buf, self.slices, lambda i, m, lo, hi, buf=buf, handles=handles:
                          handles.append(ex.allreduce_begin(buf[lo:hi])))
            pend[b] = (k, handles)
        for k in range(max(0, n - nbuf), n):
            retire(k % nbuf)
        ops.collect(self.Runner_list)
        return list(bufs) if consume is None else None

    def process(self, ops=None):
        """The summed map(s) on the host: float64[Npix] for one Runner, a list for a list of Runners."""
        first = self.Runner_list[0] if self.Runner_list else None
        if first is not None and hasattr(first, "offsets_device"):
            if ops is None and self.world == 1:                        # one GPU: the transfers of consecutive shells overlap
                from ..Runners.HealpixRunner import _baryonify_pipelined
                outs = _baryonify_pipelined(self.Runner_list)
            else:
                outs = [self._baryonify(R, ops) for R in self.Runner_list]
            return outs if self.is_list else outs[0]
        ops = ops or _DeviceOps()
        n = len(self.Runner_list)
        host = [None] * n
        if n == 1:
            d_map = self.process_device(ops=ops)[0]
            host[0] = ops.to_host(d_map)
        else:
            # copies to the host run on their own stream: shell k's copy overlaps shell k + 1's painting; a buffer is only
            # repainted after its copy has finished
            copied = {}

            def consume(k, d_map):
                h, ev = ops.to_host_begin(d_map)
                host[k] = (h, ev)
                copied[id(d_map)] = ev
            orig_paint = ops.paint

            def paint_after_copy(R, d_map, slices, on_slice):
                ev = copied.pop(id(d_map), None)
                if ev is not None:
                    ops.wait_event(ev)
                orig_paint(R, d_map, slices, on_slice)
            ops.paint = paint_after_copy
            try:
                self.process_device(consume=consume, ops=ops)
            finally:
                ops.paint = orig_paint
            host = [ops.host_ready(h, ev) for h, ev in host]
        outs = [m.reshape(np.shape(R.LightconeShell.map)) for m, R in zip(host, self.Runner_list)]
        return outs if self.is_list else outs[0]

    def _baryonify(self, local, ops):
        from ..Runners.HealpixRunner import _baryonify_process, _BaryonifyDeviceOps
        self.last_ops = ops = ops or _BaryonifyDeviceOps(local)           # (its h2d_bytes: what this rank uploaded of the input map)
        return _baryonify_process(local, ops, self._exchange(), slices=self.slices)
