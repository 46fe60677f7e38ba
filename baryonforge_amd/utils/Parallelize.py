"""
Parallel wrappers, mirroring BaryonForge/utils/Parallelize.py: `SimpleParallel`
(:8-113) and `SplitJoinParallel` (:116-320).

The reference forks joblib/loky worker processes and pickles Runner objects to
them.  Here the workers are the ranks of a torch.distributed process group --
one process per MI355X, backend "nccl" (= RCCL over xGMI) -- launched with
`python -m torch.distributed.run --nproc-per-node N ...`:

  * SplitJoinParallel shards the halo catalog BY SKY PATCH across the ranks
    (baryonforge_amd.sharding), every rank paints its shard onto a full-size
    private map in its own HBM, and ONE all-reduce(sum, f64, Npix) replaces
    the parent-side np.sum(outputs, axis=0) of Parallelize.py:318.
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

from ..sharding import estimate_disc_pixels, shard_by_sky_patch

__all__ = ["SimpleParallel", "SplitJoinParallel", "Exchange", "HostOps"]


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
    collective = "bfg":   the library's communicator (bfg_comm_init / bfg_allreduce_f64 / bfg_reduce_scatter_f64 of
                          include/bfg_mi355.h: RCCL on the context's stream) -- what a C-ABI consumer without torch uses;
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
        """t <- sum over ranks (in place)"""
        if self.collective == "bfg" and t.is_cuda:
            self.ctx.allreduce(t)
        elif self._staged(t):
            h = t.cpu()
            self.dist.all_reduce(h, op=self.dist.ReduceOp.SUM)
            t.copy_(h)
        else:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t

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


class SimpleParallel(object):
    """Run several independent Runners (Parallelize.py:8-113).  Runners are dealt round-robin to the
    ranks of the process group; every rank returns the full list of outputs (gathered)."""

    def __init__(self, Runner_list, njobs=-1, seed=42):
        self.Runner_list = list(Runner_list)
        self.njobs = njobs
        self.seed = seed

    def single_run(self, Runner):
        return Runner.process()

    def process(self):
        dist = _dist()
        if dist is None or dist.get_world_size() == 1:
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
    Split one Runner's halo catalog across GPUs and join (sum) the results (Parallelize.py:116-320).

    Parameters
    ----------
    Runner : PaintProfilesShell or BaryonifyShell
    njobs : ignored when a process group exists (the world size is used); kept for API parity
    seed : kept for API parity (the sky-patch split is deterministic and needs no shuffle)
    nside_patch, layout : NSIDE of the sky patches and how they are dealt to the ranks (sharding.shard_by_sky_patch)
    collective : "torch" (torch.distributed: RCCL with the nccl backend) or "bfg" (the library's own RCCL communicator,
        bfg_allreduce_f64 & co.); see Exchange
    local_ops : test seam -- an object replacing the GPU work of the per-rank runner (`paint(runner)`,
        `offsets(runner)`, `regrid(nside, offsets, in_map)`, all on numpy arrays), so that the sharding and the
        exchange steps -- the very code the GPU ranks run -- can be exercised with the gloo backend on CPU-only machines.
    local_process : older form of the seam, paint only: callable(runner_for_this_rank) -> np.ndarray
    """

    def __init__(self, Runner, njobs=-1, seed=42, nside_patch=64, local_process=None, layout="interleaved",
                 collective="torch", local_ops=None):
        self.Runner = Runner
        self.seed = seed
        self.njobs = njobs
        self.nside_patch = nside_patch
        self.layout = layout
        self.collective = collective
        if local_process is not None and local_ops is None:
            local_ops = _PaintOnlyOps(local_process)
        self.local_ops = local_ops
        dist = _dist()
        self.rank = dist.get_rank() if dist else 0
        self.world = dist.get_world_size() if dist else 1
        self.Runner_list = self.split_run(self.Runner)

    def split_run(self, Runner):
        """the Runner of THIS rank (list of length 1, mirroring the reference's attribute name)"""
        HaloCat = Runner.HaloLightConeCatalog
        cat = HaloCat.cat
        if self.world == 1:
            self.shard_indices = np.arange(cat.size)
            return [Runner]
        w = estimate_disc_pixels(Runner.cosmo, cat["M"], cat["z"], Runner.epsilon_max, Runner.LightconeShell.NSIDE,
                                 Runner.mass_def)
        shards = shard_by_sky_patch(cat["ra"], cat["dec"], w, self.world, self.nside_patch, layout=self.layout)
        self.shard_indices = shards[self.rank]
        New_HaloCatalog = HaloCat[self.shard_indices]
        New_Runner = type(Runner)(New_HaloCatalog, Runner.LightconeShell, Runner.epsilon_max, Runner.model,
                                  Runner.use_ellipticity, Runner.mass_def,
                                  include_pixel_size=Runner.include_pixel_size, verbose=False)
        if hasattr(Runner, "variant"):
            New_Runner.variant = Runner.variant
        # every rank builds the D_A spline the serial run would build (knots up to max(z) of the WHOLE catalog,
        # HealpixRunner.py:297-299): a halo then gets bit-identical scalars whichever rank it lands on
        New_Runner._spline_z_max = float(np.max(cat["z"])) if cat.size else 0.0
        return [New_Runner]

    def single_run(self, Runner):
        return Runner.process()

    def process(self):
        dist = _dist()
        local = self.Runner_list[0]
        ops = self.local_ops
        is_baryonify = hasattr(local, "offsets_device")
        if self.world == 1:
            if ops is None:
                return local.process()
            return ops.baryonify(local, None) if is_baryonify else ops.paint(local)
        ex = Exchange(dist, self.collective if ops is None else "torch")
        if ops is not None:                                         # CPU test seam (gloo): same steps, numpy arrays
            import torch
            if is_baryonify:
                return ops.baryonify(local, ex)
            part = torch.from_numpy(np.ascontiguousarray(ops.paint(local), dtype=np.float64))
            return ex.allreduce(part).numpy()
        if is_baryonify:
            return local.process(distributed=ex)
        d_map = local.process_device()
        ex.allreduce(d_map)                                         # RCCL over xGMI
        from ..engine import get_context
        return get_context().to_host(d_map).reshape(np.shape(local.LightconeShell.map))


class _PaintOnlyOps(object):
    def __init__(self, fn):
        self.paint = fn


class HostOps(object):
    """Test seam of SplitJoinParallel (`local_ops`): the per-rank GPU work replaced by host callables on numpy arrays --
    paint(runner) -> [Npix], offsets(runner) -> [Npix, 3], regrid(nside, offsets, in_map) -> [Npix] -- e.g. the CPU oracle
    in tests/.  Everything else (sharding, the reduce-scatter / all-reduce sequence, the pixel ranges, the mass
    assertion) is the code the GPU ranks run."""

    def __init__(self, paint=None, offsets=None, regrid=None):
        self.paint, self._offsets, self._regrid = paint, offsets, regrid

    def baryonify(self, runner, exchange):
        from ..Runners.HealpixRunner import _baryonify_process
        return _baryonify_process(runner, _BoundHostOps(self, runner), exchange)


class _BoundHostOps(object):
    def __init__(self, host, runner):
        self.host, self.runner = host, runner

    def upload(self, flat):
        import torch
        return torch.from_numpy(np.array(flat, dtype=np.float64))

    def zeros(self, *shape):
        import torch
        return torch.zeros(*shape, dtype=torch.float64)

    def absmax_sum(self, t):
        return float(t.abs().max()), float(t.sum())

    def offsets(self):
        import torch
        return torch.from_numpy(np.ascontiguousarray(self.host._offsets(self.runner), dtype=np.float64))

    def regrid(self, nside, d_off, d_in, d_out):
        import torch
        d_out += torch.from_numpy(np.ascontiguousarray(self.host._regrid(nside, d_off.numpy(), d_in.numpy())))

    def to_host(self, t):
        return t.numpy()
