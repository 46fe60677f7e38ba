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
    so it is all-reduced, then every rank regrids its own pixel range and the
    output maps are all-reduced.
  * `include_pixel_size` is forwarded to the per-rank runners (the reference
    drops it, Parallelize.py:271 -- a conscious divergence).

Without an initialised process group both wrappers simply run on the one GPU.
"""
import numpy as np

from ..sharding import estimate_disc_pixels, shard_by_sky_patch

__all__ = ["SimpleParallel", "SplitJoinParallel"]


def _dist():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist
    except Exception:
        pass
    return None


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
    local_process : test seam -- callable(runner_for_this_rank) -> np.ndarray replacing the GPU
        process() of the per-rank paint runner, so the shard + all-reduce logic can be
        exercised with the gloo backend on CPU-only machines.
    """

    def __init__(self, Runner, njobs=-1, seed=42, nside_patch=8, local_process=None, layout="contiguous"):
        self.Runner = Runner
        self.seed = seed
        self.njobs = njobs
        self.nside_patch = nside_patch
        self.layout = layout
        self.local_process = local_process
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
        return [New_Runner]

    def single_run(self, Runner):
        return Runner.process()

    def process(self):
        dist = _dist()
        local = self.Runner_list[0]
        if self.world == 1:
            return self.local_process(local) if self.local_process else local.process()
        import torch
        if self.local_process is not None:                      # CPU test seam (gloo)
            part = torch.from_numpy(np.ascontiguousarray(self.local_process(local), dtype=np.float64))
            dist.all_reduce(part, op=dist.ReduceOp.SUM)
            return part.numpy()
        if hasattr(local, "offsets_device"):                    # BaryonifyShell
            return local.process(distributed=dist)
        d_map = local.process_device()
        dist.all_reduce(d_map, op=dist.ReduceOp.SUM)            # RCCL over xGMI
        from ..engine import get_context
        return get_context().to_host(d_map).reshape(np.shape(local.LightconeShell.map))
