#!/usr/bin/env python3
"""Soak of LightconeShell(pinned="inplace") -- hipHostRegister of the caller's own map (engine.pin) -- after round 6's rule that only
page-aligned buffers that own their pages are registered (engine.aligned_empty / engine.pin_ok).  Round 5's soak saw two GPU memory
faults in ~5000 shells whose HEAP arrays were registered in place (profiles/r05_soak.txt): the hypothesis was that a user-pointer
mapping of partial pages shared with other heap objects goes stale when the allocator trims or the kernel migrates them.  Every case
here: a fresh aligned buffer, registered, BaryonifyShell.process() (asynchronous upload slices from it, result compared with the
pageable run of the same shell), unregistered, dropped -- while the process heap is churned with allocations and frees of random sizes
between and during the cases.  usage: soak_inplace.py [cases] [seed]"""
import os, sys, time, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import baryonforge_amd as bfg
from baryonforge_amd import engine, synthetic as syn

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 606)
cosmo = dict(syn.COSMO)
zd, Md, rd, d = syn.displacement_table()
model = bfg.Baryonification2D.from_arrays(zd, Md, rd, d, cosmo, epsilon_max=20)
configs = []
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for nside, n in ((64, 300), (128, 1500), (200, 3000), (256, 4000), (512, 8000)):
        ra, dec, M, z = syn.catalog(n, seed=int(rng.integers(1 << 30)), logM=(13.0, 15.0))
        Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
        m_in = syn.mass_map(nside, seed=int(rng.integers(1 << 30)))
        ref = bfg.BaryonifyShell(Cat, bfg.LightconeShell(map=m_in.copy(), cosmo=cosmo), 10, model, verbose=False).process().copy()
        configs.append((nside, Cat, m_in, ref))
churn, t0, done = [], time.time(), 0
for case in range(cases):
    nside, Cat, m_in, ref = configs[int(rng.integers(len(configs)))]
    # heap churn: allocate and free arrays of 1 KB .. 8 MB (below and above glibc's mmap threshold) around the registration
    for _ in range(int(rng.integers(0, 4))):
        churn.append(np.ones(int(2 ** rng.uniform(7, 20)), dtype=np.float64))
    while len(churn) > 6:
        churn.pop(int(rng.integers(len(churn))))
    src = engine.aligned_empty(m_in.size)
    src[:] = m_in
    with warnings.catch_warnings():
        warnings.simplefilter("error")                       # a fallback to a page-locked COPY would warn: it must not happen here
        shell = bfg.LightconeShell(map=src, cosmo=cosmo, pinned="inplace")
    assert shell.map is src and engine.is_pinned(src)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = bfg.BaryonifyShell(Cat, shell, 10, model, verbose=False).process()
    if rng.uniform() < 0.3:
        churn.append(np.ones(int(2 ** rng.uniform(7, 20)), dtype=np.float64))
    err = float(np.max(np.abs(got - ref)) / np.max(np.abs(ref)))
    assert err < 1e-9 and np.isclose(got.sum(), m_in.sum()), (case, nside, err)
    assert np.array_equal(src, m_in)
    engine.unpin(src)
    assert not engine.is_pinned(src)
    del shell, src, got
    done += 1
    if done % 500 == 0:
        print(f"{done} in-place shells ok, {time.time() - t0:.0f} s", flush=True)
print(f"{done} cases passed ({time.time() - t0:.0f} s): page-aligned buffers registered in place, no fault")
