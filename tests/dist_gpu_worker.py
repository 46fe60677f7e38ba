"""
One rank of tests/test_gpu_distributed.py: a FRESH python process (started with subprocess, never a re-exec of a process
that has touched the GPU) that joins a torch.distributed group and runs the reference-shaped multi-GPU API --
SplitJoinParallel(PaintProfilesShell) and SplitJoinParallel(BaryonifyShell) -- through the real HIP kernels (no test
seam).  On a one-GPU box every rank uses cuda:0 and the group is gloo; with one GPU per rank the group is nccl (= RCCL)
and the library's own communicator (collective="bfg") is exercised too.  Results go to <out>/..._<rank>.npy.
"""
import argparse
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

NSIDE, N_PAINT, N_BARY, EPS = 256, 4000, 3000, 10.0
LIST_CUTS = [(0, 1500), (1500, 4000), (500, 2500), (0, 4000)]


def inputs():
    from baryonforge_amd import synthetic as syn
    ra, dec, M, z = syn.catalog(N_PAINT, seed=2024)
    zax, Max, rax, T = syn.pressure_table()
    dz, dM, dr, dtab = syn.displacement_table()
    m_in = syn.mass_map(NSIDE)
    m_in[::9] = 0.0
    return dict(cosmo=dict(syn.COSMO), ra=ra, dec=dec, M=M, z=z, paint=(zax, Max, rax, T), disp=(dz, dM, dr, dtab), m_in=m_in)


def north_mask(I, n=1200):
    """the first n halos north of dec = +10 deg"""
    return np.flatnonzero(I["dec"] > 10.0)[:n]


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--rank", type=int, required=True)
    p.add_argument("--world", type=int, required=True)
    p.add_argument("--port", type=int, required=True)
    p.add_argument("--out", required=True)
    p.add_argument("--backend", default="gloo")
    a = p.parse_args()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(a.port), RANK=str(a.rank), WORLD_SIZE=str(a.world))
    import torch
    import torch.distributed as dist
    one_device = a.backend != "nccl"
    dev = 0 if one_device else a.rank
    torch.cuda.set_device(dev)
    if a.backend == "nccl":
        dist.init_process_group("nccl", rank=a.rank, world_size=a.world, device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group(a.backend, rank=a.rank, world_size=a.world)
    import baryonforge_amd as bfg
    I = inputs()
    cosmo = I["cosmo"]
    zax, Max, rax, T = I["paint"]
    Cat = bfg.HaloLightConeCatalog(I["ra"], I["dec"], I["M"], I["z"], cosmo)
    npix = 12 * NSIDE * NSIDE
    info = {"rank": a.rank, "world": a.world, "backend": a.backend, "device": dev}
    collectives = ["torch"] + (["bfg"] if a.backend == "nccl" else [])
    for coll in collectives:
        R = bfg.PaintProfilesShell(Cat, bfg.LightconeShell(map=np.zeros(npix), cosmo=cosmo), EPS,
                                   bfg.TabulatedProfile.from_arrays(zax, Max, rax, T), verbose=False)
        SJ = bfg.SplitJoinParallel(R, collective=coll)
        assert SJ.world == a.world and SJ.rank == a.rank
        out = SJ.process()
        np.save(os.path.join(a.out, f"paint_{coll}_{a.rank}.npy"), out)
        np.save(os.path.join(a.out, f"idx_{coll}_{a.rank}.npy"), SJ.shard_indices)
        info[f"paint_{coll}_pixel_updates"] = int(SJ.Runner_list[0].last_stats["pixel_updates"])

        # a LIST of shells, every one split over all ranks and pipelined (sliced exchange, rotating buffers, async copies)
        model = bfg.TabulatedProfile.from_arrays(zax, Max, rax, T)
        shells = []
        for lo, hi in LIST_CUTS:
            c = bfg.HaloLightConeCatalog(I["ra"][lo:hi], I["dec"][lo:hi], I["M"][lo:hi], I["z"][lo:hi], cosmo)
            shells.append(bfg.PaintProfilesShell(c, bfg.LightconeShell(map=np.zeros(npix), cosmo=cosmo), EPS, model, verbose=False))
        LSJ = bfg.SplitJoinParallel(shells, collective=coll, slices=3)
        louts = LSJ.process()
        info[f"list_{coll}_pixel_updates"] = int(LSJ.Runner_list[0].last_stats["pixel_updates"])
        for k, o in enumerate(louts):
            np.save(os.path.join(a.out, f"list{k}_{coll}_{a.rank}.npy"), o)
        louts2 = bfg.SimpleParallel(shells, split=True, collective=coll, slices=1).process()
        assert all(np.allclose(x, y, rtol=1e-12, atol=0) for x, y in zip(louts, louts2))   # atomics reorder: rounding only

        if coll == "torch":
            # the owner-computes join with the real kernels: declination stripes, border exchange, all-gather
            OSJ = bfg.SplitJoinParallel(shells, exchange="owner")
            oouts = OSJ.process()
            assert all(np.allclose(x, y, rtol=1e-12, atol=0) for x, y in zip(oouts, louts))
            assert all(np.array_equal(x != 0, y != 0) for x, y in zip(oouts, louts))
            own = OSJ._owner_of_runner[-1]
            info["owner_border_fraction"] = own.border_bytes / (8.0 * npix)
            info["owner_pixel_updates"] = int(OSJ.Runner_list[0].last_stats["pixel_updates"])

        dz, dM, dr, dtab = I["disp"]
        sub = bfg.HaloLightConeCatalog(I["ra"][:N_BARY], I["dec"][:N_BARY], I["M"][:N_BARY], I["z"][:N_BARY], cosmo)
        BR = bfg.BaryonifyShell(sub, bfg.LightconeShell(map=I["m_in"].copy(), cosmo=cosmo), EPS,
                                bfg.Baryonification2D.from_arrays(dz, dM, dr, dtab, cosmo, epsilon_max=20), verbose=False)
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            BSJ = bfg.SplitJoinParallel(BR, collective=coll)
            bout = BSJ.process()
        info[f"bary_{coll}_h2d_bytes"] = int(BSJ.last_ops.h2d_bytes)      # of the input map: the pixels this rank owns, nothing else
        np.save(os.path.join(a.out, f"bary_{coll}_{a.rank}.npy"), bout)
    # one rank's shard EMPTY (a partial-sky catalog cut into declination stripes: every halo north of dec = +10 deg, so rank 1 of 2
    # gets none): the sliced exchange must still issue the same collectives on both ranks (bfg_*_sliced reports slices that do
    # not depend on the catalog)
    nn = north_mask(I)
    ncat = bfg.HaloLightConeCatalog(I["ra"][nn], I["dec"][nn], I["M"][nn], I["z"][nn], cosmo)
    for coll in collectives:
        R = bfg.PaintProfilesShell(ncat, bfg.LightconeShell(map=np.zeros(npix), cosmo=cosmo), EPS,
                                   bfg.TabulatedProfile.from_arrays(zax, Max, rax, T), verbose=False)
        SJ = bfg.SplitJoinParallel(R, collective=coll, layout="stripes", slices=4)
        info[f"north_{coll}_shard"] = int(SJ.shard_indices.size)
        np.save(os.path.join(a.out, f"north_paint_{coll}_{a.rank}.npy"), SJ.process())
        dz, dM, dr, dtab = I["disp"]
        BR = bfg.BaryonifyShell(ncat, bfg.LightconeShell(map=I["m_in"].copy(), cosmo=cosmo), EPS,
                                bfg.Baryonification2D.from_arrays(dz, dM, dr, dtab, cosmo, epsilon_max=20), verbose=False)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            bout = bfg.SplitJoinParallel(BR, collective=coll, layout="stripes", slices=4).process()
        np.save(os.path.join(a.out, f"north_bary_{coll}_{a.rank}.npy"), bout)
    from baryonforge_amd import _lib
    info["so"] = _lib.so_path()
    info["maps"] = [m.split()[-1] for m in open("/proc/self/maps") if "libbfg_mi355" in m][:1]
    json.dump(info, open(os.path.join(a.out, f"info_{a.rank}.json"), "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
