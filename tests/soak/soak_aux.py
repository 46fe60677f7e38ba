#!/usr/bin/env python3
"""Randomised soak of the widened rows against the oracle: BaryonifySnapshot (2D / 3D, both passes), the mass deposit
(NGP / CIC, direct and tiled, forced small slot counts) and the periodic-grid runners (paint / baryonify, per-halo and tile
passes).  usage: soak_aux.py [seconds] [seed]"""
import os, sys, time, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import baryonforge_amd as bfg
from baryonforge_amd import synthetic as syn
from oracle import oracle as orc
from util import assert_maps_close

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 99)
cosmo = dict(syn.COSMO)
t_end = time.time() + budget
case = 0
ENV = ("BFG_SNAPSHOT", "BFG_DEPOSIT", "BFG_DEPOSIT_CAP", "BFG_GRID")


def setenv(**kw):
    for k in ENV:
        os.environ.pop(k, None)
    for k, v in kw.items():
        if v is not None:
            os.environ[k] = str(v)


def pclose(a, b, L, atol, tag):
    d = np.abs(a - b); d = np.minimum(d, L - d)
    assert d.max() <= atol, (tag, d.max())


while time.time() < t_end:
    case += 1
    kind = rng.choice(["snapshot", "deposit", "grid"])
    is2D = bool(rng.uniform() < 0.4)
    nd = 2 if is2D else 3
    zs = float(rng.uniform(0.05, 0.8))
    if kind == "snapshot":
        L = float(rng.choice([40.0, 150.0, 600.0]))
        npart, nhalo = int(10 ** rng.uniform(2, 5.3)), int(10 ** rng.uniform(0, 3.3))
        P = rng.uniform(0, L, (npart, nd))
        H = rng.uniform(0, L, (nhalo, nd))
        k = max(1, nhalo // 10)
        H[:k] = np.where(rng.uniform(size=(k, nd)) < 0.5, rng.uniform(0, 0.3, (k, nd)), L - rng.uniform(0, 0.3, (k, nd)))
        if rng.uniform() < 0.3:
            P[: npart // 3] = (H[rng.integers(0, nhalo, npart // 3)] + rng.normal(0, 0.5, (npart // 3, nd))) % L   # clustered
        hM = 10 ** rng.uniform(12.5, 15.4, nhalo)
        rdelta = bool(rng.uniform() < 0.3)
        shape = [(10, 30, 100), (2, 30, 2000)][int(rng.integers(2))]
        zax, Max, rax, d = syn.displacement_table(*shape)
        eps = float(rng.choice([3, 10, 25]))
        path = rng.choice(["direct", "cell"])
        setenv(BFG_SNAPSHOT=path)
        Cat = bfg.HaloNDCatalog(H[:, 0], H[:, 1], hM, zs, cosmo, z=None if is2D else H[:, 2])
        Part = bfg.ParticleSnapshot(x=P[:, 0], y=P[:, 1], z=None if is2D else P[:, 2], M=np.ones(npart), L=L, redshift=zs,
                                    cosmo=cosmo)
        model = bfg.Baryonification2D.from_arrays(zax, Max, rax, d, cosmo, epsilon_max=20, Rdelta_sampling=rdelta)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            new = bfg.BaryonifySnapshot(Cat, Part, epsilon_max=eps, model=model, verbose=False).process()
            # the oracle takes the float32 halo columns the catalogue holds (io.py:204)
            c = Cat.cat
            ref = orc.baryonify_snapshot(cosmo, L, zs, P[:, 0], P[:, 1], None if is2D else P[:, 2], c["M"], c["x"], c["y"],
                                         None if is2D else c["z"], (zax, Max, rax), d, eps, 20, rdelta)
        got = np.stack([new["x"], new["y"]] + ([] if is2D else [new["z"]]), axis=1)
        tag = f"case {case}: snapshot {nd}D L {L} npart {npart} nhalo {nhalo} eps {eps} rdelta {rdelta} table {shape} {path}"
        pclose(got, ref, L, 1e-9, tag)
    elif kind == "deposit":
        L = float(rng.choice([10.0, 75.0]))
        n = int(10 ** rng.uniform(1, 5.5))
        N = int(rng.integers(3, 90 if not is2D else 300))
        P = rng.uniform(-0.1 * L, 1.1 * L, (n, nd)) if rng.uniform() < 0.3 else rng.uniform(0, L, (n, nd))
        if rng.uniform() < 0.5:
            P[: n // 2] = rng.normal(L / 3, L / 200, (n // 2, nd))
        edges = np.linspace(0, L, N + 1)
        P[: min(n, 50)] = edges[rng.integers(0, N + 1, (min(n, 50), nd))]
        M = rng.uniform(0.5, 2.0, n)
        mode = rng.choice(["ngp", "cic"])
        path = rng.choice(["direct", "tile"])
        cap = rng.choice([None, 1, 7]) if path == "tile" else None
        setenv(BFG_DEPOSIT=path, BFG_DEPOSIT_CAP=cap)
        from baryonforge_amd.engine import get_context
        ctx = get_context(0)
        got = ctx.deposit_grid(ctx.to_device(P), ctx.to_device(M), L, N, mode).cpu().numpy()
        ref = orc.make_map(P, M, L, N, mode)
        tag = f"case {case}: deposit {nd}D n {n} N {N} {mode} {path} cap {cap}"
        np.testing.assert_allclose(got, ref, rtol=1e-11, atol=1e-11, err_msg=tag)
    else:
        N = int(rng.integers(8, 400)) if is2D else int(rng.integers(8, 72))
        N -= N % 4        # Npix // 2 even: with an odd clip value the reference's cut-out arrays disagree in length (it raises)
        L = float(rng.choice([60.0, 300.0]))
        nhalo = int(10 ** rng.uniform(0, 2.6))
        bins = (np.arange(N) + 0.5) * (L / N)
        H = rng.uniform(0, L, (nhalo, 3))
        hM = 10 ** rng.uniform(13.0, 15.3, nhalo)
        eps = float(rng.choice([2, 6, 12]))
        path = rng.choice(["direct", "tile", "small"])
        setenv(BFG_GRID=path)
        Cat = bfg.HaloNDCatalog(H[:, 0], H[:, 1], hM, zs, cosmo, z=None if is2D else H[:, 2])
        c = Cat.cat
        hpos = np.stack([c["x"], c["y"]] + ([] if is2D else [c["z"]]), axis=1).astype(np.float64)
        zax, Max, rax, T = syn.pressure_table()
        model = bfg.TabulatedProfile.from_arrays(zax, Max, rax, T, T * 0.5)
        got = bfg.PaintProfilesGrid(Cat, bfg.GriddedMap(map=np.zeros((N,) * nd), redshift=zs, bins=bins, cosmo=cosmo), eps, model,
                                    verbose=False).process()
        ref = orc.paint_grid(cosmo, bins, (N,) * nd, zs, hpos, c["M"], (zax, Max, rax), T if is2D else T * 0.5, eps, True)
        tag = f"case {case}: grid {nd}D N {N} nhalo {nhalo} eps {eps} {path}"
        assert_maps_close(got, ref, 1e-5, what=tag + " paint")
        zd, Md, rd, d = syn.displacement_table()
        dm = bfg.Baryonification2D.from_arrays(zd, Md, rd, d * 3, cosmo, epsilon_max=20)
        m_in = rng.uniform(0, 10, (N,) * nd)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            gotb = bfg.BaryonifyGrid(Cat, bfg.GriddedMap(map=m_in.copy(), redshift=zs, bins=bins, cosmo=cosmo), eps, dm,
                                     verbose=False).process()
            refb = orc.baryonify_grid(cosmo, bins, m_in, zs, hpos, c["M"], (zd, Md, rd), d * 3, eps, 20)
        assert_maps_close(gotb, refb, 1e-5, floor=1e-9, what=tag + " baryonify")
    print("ok", tag, flush=True)
setenv()
print(f"{case} cases passed")
