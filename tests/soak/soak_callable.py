#!/usr/bin/env python3
"""Randomised soak of the callable-model path (models that are not tabulated, tables with more than three p_keys axes) against the
oracle's line-by-line restatement of the reference loops: NSIDE, catalog size and layout, eps, batch size, paint and baryonify.
usage: soak_callable.py [seconds] [seed]"""
import os, sys, time, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import baryonforge_amd as bfg
from baryonforge_amd import synthetic as syn
from util import assert_maps_close
from oracle import oracle as orc

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 77)
cosmo = dict(syn.COSMO)


class Model(object):
    def __init__(self, k):
        self.k = k

    def projected(self, cosmo, r, M, a):
        r = np.asarray(r, dtype=np.float64)
        rc = self.k * (M / 1e14) ** (1 / 3)
        out = 1e-6 * (M / 1e14) ** (5 / 3) / a ** 2 * (1 + (r / rc) ** 2) ** -1.5
        return np.where(r > 9 * rc, np.nan, out)

    def displacement(self, r, M, a):
        x = np.asarray(r, dtype=np.float64) / (self.k * 3 * (M / 1e14) ** (1 / 3) / a)
        return 0.1 * (M / 1e14) ** (1 / 3) * x * (1 - x / 4) * np.exp(-x)


t_end, case = time.time() + budget, 0
while time.time() < t_end:
    case += 1
    nside = int(rng.choice([8, 13, 32, 64, 128, 256]))
    n = int(10 ** rng.uniform(0, 3.0))
    eps = float(rng.choice([2, 5, 10]))
    ra, dec, M, z = syn.catalog(n, seed=int(rng.integers(1 << 30)), logM=(12.0, 15.3))
    if rng.uniform() < 0.3:
        dec[: max(1, n // 10)] = rng.choice([-89.97, 89.97])              # discs over the poles
    os.environ["BFG_CALLABLE_BATCH"] = str(int(rng.choice([500, 5000, 1 << 24])))
    model = Model(float(rng.uniform(0.1, 0.5)))
    ips = bool(rng.uniform() < 0.5)
    Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
    tag = f"case {case}: nside {nside} n {n} eps {eps} ips {ips} batch {os.environ['BFG_CALLABLE_BATCH']}"
    ref, ptot = orc.paint_shell_callable(cosmo, nside, ra, dec, M, z, eps, lambda r, Mj, aj: model.projected(None, r, Mj, aj), ips)
    R = bfg.PaintProfilesShell(Cat, bfg.LightconeShell(map=np.zeros(12 * nside * nside), cosmo=cosmo), eps, model,
                               include_pixel_size=ips, verbose=False)
    got = R.process()
    assert R.last_stats["pixel_updates"] == ptot, tag
    assert np.array_equal(got != 0, ref != 0), tag
    assert_maps_close(got, ref, 1e-9, what=tag)
    if rng.uniform() < 0.6:
        off, ptb = orc.baryonify_offsets_callable(cosmo, nside, ra, dec, M, z, eps, model.displacement)
        m_in = syn.mass_map(nside)
        refb = orc.regrid_shell(nside, off, m_in)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            Rb = bfg.BaryonifyShell(Cat, bfg.LightconeShell(map=m_in.copy(), cosmo=cosmo), eps, model, verbose=False)
            gotb = Rb.process()
        assert Rb.last_stats["pixel_updates"] == ptb, tag
        assert_maps_close(gotb, refb, 1e-5, floor=1e-9, what=tag + " baryonify")
        tag += " +baryonify"
    print("ok", tag, flush=True)
print(f"{case} cases passed")
