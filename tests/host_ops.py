"""
Test infrastructure: a stand-in for the GPU side of one rank (baryonforge_amd.utils.Parallelize._DeviceOps and
Runners.HealpixRunner._BaryonifyDeviceOps) that computes with the CPU oracle on torch CPU tensors, so that the multi-rank
control flow of the product -- sharding, the sliced / pipelined exchange, buffer rotation, pixel ranges, the mass assertion --
runs under gloo on machines without a GPU.  Never imported by the product.
"""
import numpy as np
import torch


class OraclePaintOps(object):
    """paint(runner) -> float64[Npix] computed by the caller's oracle function"""

    def __init__(self, paint_fn, n_cuts=None):
        self.paint_fn = paint_fn
        self.n_cuts = n_cuts            # None: as many slices as asked for
        self.log = []                   # (what, ...) records for the tests
        self.collected = 0

    def new_map(self, npix):
        return torch.full((npix,), float("nan"), dtype=torch.float64)       # poisoned: every pixel must be defined

    def paint(self, runner, d_map, slices, on_slice):
        self.log.append(("paint", id(d_map), slices))
        d_map.copy_(torch.from_numpy(np.ascontiguousarray(self.paint_fn(runner), dtype=np.float64)))
        if on_slice is None:
            return
        n = d_map.numel()
        k = min(self.n_cuts or slices, n)
        # uneven cuts on purpose: the product must take whatever ranges the library reports
        cuts = [0] + sorted(set(int(n * (i / k) ** 1.3) for i in range(1, k))) + [n]
        cuts = sorted(set(cuts))
        for i in range(len(cuts) - 1):
            on_slice(i, len(cuts) - 1, cuts[i], cuts[i + 1])

    def reset_stats(self):
        pass

    def collect(self, runners):
        self.collected += 1
        return {}

    def to_host(self, d_map):
        return d_map.numpy().copy()

    def to_host_begin(self, d_map):
        self.log.append(("copy", id(d_map)))
        return d_map.clone(), ("copied", id(d_map))

    def wait_event(self, ev):
        assert ev[0] == "copied"
        self.log.append(("wait_copy", ev[1]))

    def host_ready(self, h, ev):
        return h.numpy()


class OracleBaryonifyOps(object):
    """the ops object of Runners.HealpixRunner._baryonify_process, on CPU tensors"""

    def __init__(self, runner, offsets_fn, regrid_fn):
        self.runner, self.offsets_fn, self.regrid_fn = runner, offsets_fn, regrid_fn

    def upload(self, flat):
        return torch.from_numpy(np.array(flat, dtype=np.float64))

    def zeros(self, *shape):
        return torch.zeros(*shape, dtype=torch.float64)

    def absmax_sum(self, t):
        return float(t.abs().max()), float(t.sum())

    def offsets(self, slices=1, on_slice=None):
        d = torch.from_numpy(np.ascontiguousarray(self.offsets_fn(self.runner), dtype=np.float64))
        if on_slice is not None:
            flat = d.view(-1)
            npix = d.shape[0]
            # cuts on pixel boundaries, some lengths divisible by the world size and some not (both exchange paths)
            cuts = sorted(set([0, npix] + [int(npix * (i / slices) ** 1.2) // 4 * 4 + (i % 2) for i in range(1, slices)]))
            self.cuts = cuts
            for k in range(len(cuts) - 1):
                on_slice(k, len(cuts) - 1, 3 * cuts[k], 3 * cuts[k + 1], flat)
        return d

    def regrid(self, nside, d_off, d_in, d_out):
        d_out += torch.from_numpy(np.ascontiguousarray(self.regrid_fn(nside, d_off.numpy(), d_in.numpy())))

    def to_host(self, t):
        return t.numpy()
