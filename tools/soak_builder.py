#!/usr/bin/env python3
"""Randomised soak of the device-side displacement-table builder against the host (scipy) builder: random profile
shapes incl. ringing / negative densities and holes, grid sizes, Rdelta sampling, 2D / 3D.  usage: soak_builder.py [seconds] [seed]"""
import os, sys, time, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import baryonforge_amd as bfg
from baryonforge_amd import synthetic as syn
from baryonforge_amd.background import MassDef

md = MassDef(200, "critical")


class Prof(object):
    def __init__(self, core, slope, ring=0.0, freq=6.0, hole=None):
        self.core, self.slope, self.ring, self.freq, self.hole, self.cutoff = core, slope, ring, freq, hole, None

    def set_parameter(self, k, v):
        setattr(self, k, v)

    def projected(self, cosmo, r, M, a):
        M = np.atleast_1d(M); r = np.atleast_1d(r)
        R = (md.get_radius(dict(syn.COSMO), M, a) / a)[:, None]
        x = r[None, :] / (self.core * R)
        S = M[:, None] / (2 * np.pi * (self.core * R) ** 2) * (1 + x * x) ** (-self.slope) * np.exp(-r[None, :] / (30 * R))
        S = S * (1 + self.ring * np.sin(self.freq * np.log(r))[None, :] * (r[None, :] / R) ** 1.5)
        if self.hole is not None:
            S = np.where((r[None, :] > self.hole[0]) & (r[None, :] < self.hole[1]), 0.0, S)
        return S

    real = projected


budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
t_end = time.time() + budget
case = 0
n_err = n_warn = 0
while time.time() < t_end:
    case += 1
    cls = bfg.Baryonification2D if rng.uniform() < 0.6 else bfg.Baryonification3D
    pars = []
    for _ in range(2):
        hole = None
        if rng.uniform() < 0.3:
            lo = 10 ** rng.uniform(-2.5, 1.0); hole = (lo, lo * rng.uniform(1.2, 4))
        pars.append(dict(core=rng.uniform(0.1, 0.8), slope=rng.uniform(1.2, 2.2), ring=rng.choice([0.0, 0.5, 1.5]) * rng.uniform(0, 1.3),
                         freq=rng.uniform(2, 12), hole=hole))
    if rng.uniform() < 0.15:
        pars[1] = dict(pars[0])                      # DMB == DMO: every row defaults to zero
    kw = dict(z_min=0.05, z_max=float(rng.uniform(0.3, 1.5)), N_samples_z=int(rng.integers(2, 4)), M_min=1e12, M_max=1e16,
              N_samples_Mass=int(rng.integers(3, 7)), R_min=1e-3, R_max=1e2, N_samples_R=int(rng.integers(20, 160)), verbose=False)
    if rng.uniform() < 0.3:
        kw.update(Rdelta_sampling=True, Rdelta_min=1e-2, Rdelta_max=float(rng.uniform(5, 30)))
    N_int = int(rng.integers(100, 900))
    out, warned = {}, {}
    for dev in (False, True):
        B = cls(Prof(**pars[0]), Prof(**pars[1]), dict(syn.COSMO), epsilon_max=20, N_int=N_int)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            try:
                B.setup_interpolator(device=dev, **kw)
                out[dev] = B.raw_input_d
            except ValueError as e:
                out[dev] = "ValueError"
        warned[dev] = sorted(str(x.message)[:60] for x in w if issubclass(x.category, UserWarning))
    tag = f"case {case}: {cls.__name__} N_int {N_int} NR {kw['N_samples_R']} rdelta {kw.get('Rdelta_sampling', False)} {pars}"
    if isinstance(out[False], str) or isinstance(out[True], str):
        assert isinstance(out[False], str) and isinstance(out[True], str), tag
        n_err += 1
    else:
        scale = max(np.abs(out[False]).max(), 1e-3)
        assert np.allclose(out[True], out[False], rtol=1e-7, atol=1e-10 * scale), (tag, np.abs(out[True] - out[False]).max())
        assert warned[True] == warned[False], (tag, warned)
        n_warn += bool(warned[True])
    print("ok", tag[:150], flush=True)
print(f"{case} cases passed ({n_err} where scipy and the device builder both refuse the profile, {n_warn} with warnings / zeroed rows)")
