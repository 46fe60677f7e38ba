"""
World-size-2 (and 3) tests of the multi-GPU path on CPU (gloo): sky-patch sharding of the catalog across ranks +
the exchange steps (paint: all-reduce of the per-rank maps; baryonify: reduce-scatter of the offset field, regrid of the
rank's own pixel range, all-reduce of the output maps), exactly the code path SplitJoinParallel takes under RCCL, with the
per-rank device work done by the CPU oracle (tests/host_ops.py passed as `ops`).  Covered: the single-runner call with the
map exchanged in slices, the pipelined call over a LIST of shell runners (rotating buffers, exchange of shell k behind the
painting of shell k + 1, asynchronous copies to the host), SimpleParallel(split=True).
Checks: every halo handled exactly once, reduced map == serial oracle map on every rank, mass conserved.
(tests/test_gpu_distributed.py runs the same with the real HIP kernels in the ranks.)
"""
import os
import socket
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import baryonforge_amd as bfg
    from baryonforge_amd import synthetic as syn
    from oracle import oracle as orc
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cosmo = dict(syn.COSMO)
    nside = 64
    ra, dec, M, z = syn.catalog(600, seed=9, z=(0.05, 0.3))
    zax, Max, rax, T = syn.pressure_table(6, 9, 40)
    model = bfg.TabulatedProfile.from_arrays(zax, Max, rax, T)
    Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
    Shell = bfg.LightconeShell(map=np.zeros(12 * nside * nside), cosmo=cosmo)
    Runner = bfg.PaintProfilesShell(Cat, Shell, 10, model, include_pixel_size=True, verbose=False)

    def oracle_local(runner):
        c = runner.HaloLightConeCatalog.cat
        assert runner.include_pixel_size is True                 # forwarded (unlike Parallelize.py:271)
        if c.size == 0:
            return np.zeros(12 * nside * nside)
        # per-halo scalars from the FULL catalog's spline range, as every rank's runner would build them
        a, R, D = orc.halo_scalars(cosmo, c["M"], c["z"])
        m, _ = orc.paint_shell(nside, c["ra"], c["dec"], c["M"], a, D, R, (zax, Max, rax), np.log(T), 10,
                               include_pixel_size=True)
        return m
    from host_ops import OracleBaryonifyOps, OraclePaintOps
    ops = OraclePaintOps(oracle_local)
    SJ = bfg.SplitJoinParallel(Runner, njobs=-1, slices=5)
    assert SJ.world == world and SJ.rank == rank
    out = SJ.process(ops=ops)
    assert ops.log[0][0] == "paint" and ops.log[0][2] == 5 and ops.collected == 1   # sliced exchange inside the one call
    np.save(os.path.join(out_dir, f"map_{rank}.npy"), out)
    np.save(os.path.join(out_dir, f"idx_{rank}.npy"), SJ.shard_indices)

    # ---- a LIST of shells (three catalogs): every shell split over all ranks, pipelined
    runners, subs = [], [slice(0, 250), slice(250, 600), slice(100, 400)]
    for sl in subs:
        sub = bfg.HaloLightConeCatalog(ra[sl], dec[sl], M[sl], z[sl], cosmo)
        runners.append(bfg.PaintProfilesShell(sub, Shell, 10, model, include_pixel_size=True, verbose=False))
    ops = OraclePaintOps(oracle_local)
    LSJ = bfg.SplitJoinParallel(runners, slices=3)
    outs = LSJ.process(ops=ops)
    assert isinstance(outs, list) and len(outs) == 3
    paints = [e for e in ops.log if e[0] == "paint"]
    assert len(paints) == 3 and len({e[1] for e in paints}) == 2          # two map buffers rotate
    assert sum(e[0] == "copy" for e in ops.log) == 3 and sum(e[0] == "wait_copy" for e in ops.log) == 1
    for k, o in enumerate(outs):
        np.save(os.path.join(out_dir, f"lmap{k}_{rank}.npy"), o)
    # the same through SimpleParallel(split=True), maps left on the "device"
    dev = bfg.SplitJoinParallel(runners, slices=2).process_device(ops=OraclePaintOps(oracle_local))
    assert len(dev) == 3 and all(np.array_equal(d.numpy(), o) for d, o in zip(dev, outs))
    consumed = []
    bfg.SplitJoinParallel(runners, slices=1).process_device(consume=lambda k, d: consumed.append((k, d.numpy().copy())),
                                                             ops=OraclePaintOps(oracle_local))
    assert [k for k, _ in consumed] == [0, 1, 2] and all(np.array_equal(c, o) for (_, c), o in zip(consumed, outs))

    # ---- the owner-computes join (declination stripes, border exchange + all-gather) gives the same maps
    if (12 * nside * nside) % world == 0:
        OSJ = bfg.SplitJoinParallel(runners + [Runner], exchange="owner")
        assert OSJ.layout == "stripes" and all(o is not None for o in OSJ._owner_of_runner)
        oouts = OSJ.process(ops=OraclePaintOps(oracle_local))
        for k, o in enumerate(oouts):
            np.save(os.path.join(out_dir, f"omap{k}_{rank}.npy"), o)
        np.save(os.path.join(out_dir, f"oinfo_{rank}.npy"), np.array([OSJ._owner_of_runner[-1].border_bytes, 8 * 12 * nside * nside,
                                                                      OSJ.shard_indices_list[-1].size]))

    # ---- BaryonifyShell, which the reference's splitter refuses (Parallelize.py:206-209): offsets are linear in halos
    dz, dM, dr, dtab = syn.displacement_table(5, 8, 50)
    bmodel = bfg.Baryonification2D.from_arrays(dz, dM, dr, dtab, cosmo, epsilon_max=20)
    m_in = syn.mass_map(nside)
    m_in[::5] = 0.0
    BR = bfg.BaryonifyShell(Cat, bfg.LightconeShell(map=m_in.copy(), cosmo=cosmo), 10, bmodel, verbose=False)
    calls = {"offsets": 0, "regrid_sources": None}

    def oracle_offsets(runner):
        c = runner.HaloLightConeCatalog.cat
        calls["offsets"] += 1
        a, R, D = orc.halo_scalars(cosmo, c["M"], c["z"])
        return orc.baryonify_offsets(nside, c["ra"], c["dec"], c["M"], a, D, R, R / a, (dz, dM, dr), dtab, 10, 20.0, False,
                                     None)[0]

    def oracle_regrid(ns, off, in_map):
        calls["regrid_sources"] = np.flatnonzero(in_map)
        return orc.regrid_shell(ns, off, in_map)
    BSJ = bfg.SplitJoinParallel(BR, slices=1)
    bout = BSJ.process(ops=OracleBaryonifyOps(BSJ.Runner_list[0], oracle_offsets, oracle_regrid))
    # the same with the offset field exchanged in 5 slices (owned pixels = one part of every slice)
    calls2 = dict(calls)
    BSJ5 = bfg.SplitJoinParallel(BR, slices=5)
    bops = OracleBaryonifyOps(BSJ5.Runner_list[0], oracle_offsets, oracle_regrid)
    bout5 = BSJ5.process(ops=bops)
    assert len(bops.cuts) == 6
    np.save(os.path.join(out_dir, f"bh2d_{rank}.npy"), np.array([bops.h2d_bytes]))      # bytes of the input map this rank uploaded
    np.testing.assert_allclose(bout5, bout, rtol=1e-12, atol=1e-12 * np.abs(bout).max())
    np.save(os.path.join(out_dir, f"bsrc5_{rank}.npy"), calls["regrid_sources"])
    calls["regrid_sources"] = calls2["regrid_sources"]
    np.save(os.path.join(out_dir, f"bmap_{rank}.npy"), bout)
    np.save(os.path.join(out_dir, f"bidx_{rank}.npy"), BSJ.shard_indices)
    np.save(os.path.join(out_dir, f"bsrc_{rank}.npy"), calls["regrid_sources"])
    assert calls["offsets"] == 1 + 1
    # SimpleParallel: runners dealt round-robin, every rank gets every output
    class Fake(object):
        def __init__(self, k):
            self.k = k

        def process(self):
            return np.full(3, float(self.k))
    outs = bfg.SimpleParallel([Fake(k) for k in range(5)]).process()
    assert [o[0] for o in outs] == [0.0, 1.0, 2.0, 3.0, 4.0]
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_splitjoin_ranks_gloo(tmp_path, world):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from baryonforge_amd import synthetic as syn
    from util import oracle_paint
    ra, dec, M, z = syn.catalog(600, seed=9, z=(0.05, 0.3))
    zax, Max, rax, T = syn.pressure_table(6, 9, 40)
    ref, _ = oracle_paint(dict(syn.COSMO), ra, dec, M, z, (zax, Max, rax), T, 64, 10, include_pixel_size=True)
    maps = [np.load(tmp_path / f"map_{r}.npy") for r in range(world)]
    idx = [np.load(tmp_path / f"idx_{r}.npy") for r in range(world)]
    assert np.array_equal(np.sort(np.concatenate(idx)), np.arange(600))      # each halo on exactly one rank
    assert min(i.size for i in idx) > 400 // world                             # every rank got real work (600 halos)
    for m in maps:                                                            # all-reduce: same map everywhere
        np.testing.assert_allclose(m, ref, rtol=1e-9, atol=0)
    assert np.array_equal(maps[0], maps[1])
    for k, sl in enumerate([slice(0, 250), slice(250, 600), slice(100, 400)]):     # the list API: every shell == its serial oracle
        lref, _ = oracle_paint(dict(syn.COSMO), ra[sl], dec[sl], M[sl], z[sl], (zax, Max, rax), T, 64, 10,
                               include_pixel_size=True)
        for r in range(world):
            np.testing.assert_allclose(np.load(tmp_path / f"lmap{k}_{r}.npy"), lref, rtol=1e-9, atol=0)
    # the owner-computes join: the same maps as the all-reduce; only a border travels point to point
    if (12 * 64 * 64) % world == 0:
        for r in range(world):
            for k in range(3):
                np.testing.assert_allclose(np.load(tmp_path / f"omap{k}_{r}.npy"), np.load(tmp_path / f"lmap{k}_{r}.npy"), rtol=1e-12, atol=0)
            np.testing.assert_allclose(np.load(tmp_path / f"omap3_{r}.npy"), ref, rtol=1e-9, atol=0)
        info = [np.load(tmp_path / f"oinfo_{r}.npy") for r in range(world)]
        assert sum(int(i[2]) for i in info) == 600
        assert all(0 < i[0] < 0.6 * i[1] for i in info)                           # (NSIDE 64: the discs are large against a stripe)
    # distributed BaryonifyShell == the serial oracle run, on every rank; mass conserved; every source pixel regridded once
    from util import oracle_baryonify
    dz, dM, dr, dtab = syn.displacement_table(5, 8, 50)
    m_in = syn.mass_map(64)
    m_in[::5] = 0.0
    bref = oracle_baryonify(dict(syn.COSMO), ra, dec, M, z, (dz, dM, dr), dtab, 64, 10, 20.0, m_in)
    bmaps = [np.load(tmp_path / f"bmap_{r}.npy") for r in range(world)]
    bidx = [np.load(tmp_path / f"bidx_{r}.npy") for r in range(world)]
    bsrc = [np.load(tmp_path / f"bsrc_{r}.npy") for r in range(world)]
    assert np.array_equal(np.sort(np.concatenate(bidx)), np.arange(600))
    assert np.array_equal(np.sort(np.concatenate(bsrc)), np.flatnonzero(m_in))   # pixel ranges partition the sources
    bsrc5 = [np.load(tmp_path / f"bsrc5_{r}.npy") for r in range(world)]
    assert np.array_equal(np.sort(np.concatenate(bsrc5)), np.flatnonzero(m_in))  # ... also when every slice is cut N ways
    assert all(s.size > 0 for s in bsrc)
    # every rank uploads the pixels it owns and nothing else: together exactly one copy of the map (8 Npix / N bytes each)
    h2d = [int(np.load(tmp_path / f"bh2d_{r}.npy")[0]) for r in range(world)]
    assert sum(h2d) == 8 * m_in.size and max(h2d) <= 8 * (m_in.size // world + 8 * world)
    assert not np.allclose(bref, m_in)
    for m in bmaps:
        np.testing.assert_allclose(m, bref, rtol=1e-9, atol=1e-9 * np.abs(bref).max())
        assert np.isclose(m.sum(), m_in.sum(), rtol=1e-12)


def test_splitjoin_single_process_is_passthrough():
    import baryonforge_amd as bfg
    from baryonforge_amd import synthetic as syn
    Cat = bfg.HaloLightConeCatalog(*syn.catalog(10), syn.COSMO)
    Shell = bfg.LightconeShell(map=np.zeros(12), cosmo=syn.COSMO)
    R = bfg.PaintProfilesShell(Cat, Shell, 10, None, verbose=False)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from host_ops import OraclePaintOps
    SJ = bfg.SplitJoinParallel(R, njobs=4)
    assert SJ.world == 1 and SJ.Runner_list[0] is R
    assert np.array_equal(SJ.process(ops=OraclePaintOps(lambda r: np.arange(12.0))), np.arange(12.0))
    outs = bfg.SplitJoinParallel([R, R, R]).process(ops=OraclePaintOps(lambda r: np.arange(12.0)))
    assert len(outs) == 3 and all(np.array_equal(o, np.arange(12.0)) for o in outs)
