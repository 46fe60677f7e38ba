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
        self.h2d_bytes = 0
        self.cuts = None

    def slice_cuts(self, nside, slices):
        npix = 12 * nside * nside
        if slices <= 1:
            return [0, 3 * npix]
        # cuts on pixel boundaries, some lengths divisible by the world size and some not (both exchange paths)
        cuts = sorted(set([0, npix] + [int(npix * (i / slices) ** 1.2) // 4 * 4 + (i % 2) for i in range(1, slices)]))
        self.cuts = cuts
        return [3 * c for c in cuts]

    def upload_ranges(self, flat, ranges, npix):
        d = torch.zeros(npix, dtype=torch.float64)
        for lo, hi in ranges:
            d[lo:hi] = torch.from_numpy(np.array(flat[lo:hi], dtype=np.float64))
            self.h2d_bytes += 8 * (hi - lo)
        return d

    def zeros(self, *shape):
        return torch.zeros(*shape, dtype=torch.float64)

    def count_above(self, d_in, ranges, threshold, d_dst):
        d_dst[0] = float(sum(int((~(d_in[lo:hi].abs() <= threshold)).sum()) for lo, hi in ranges))

    def offsets(self, slices=1, on_slice=None):
        d = torch.from_numpy(np.ascontiguousarray(self.offsets_fn(self.runner), dtype=np.float64))
        if on_slice is not None:
            flat = d.view(-1)
            cuts = self.slice_cuts(self.runner.LightconeShell.NSIDE, slices)
            for k in range(len(cuts) - 1):
                on_slice(k, len(cuts) - 1, cuts[k], cuts[k + 1], flat)
        return d

    def regrid(self, nside, d_off, d_in, d_out, d_sums=None):
        dep = np.ascontiguousarray(self.regrid_fn(nside, d_off.numpy(), d_in.numpy()))
        d_out += torch.from_numpy(dep)
        if d_sums is not None:
            d_sums[0] = float(d_in.sum())
            d_sums[1] = float(dep.sum())

    def to_host(self, t):
        return t.numpy()
