#!/usr/bin/env python3
"""Python API with page-locked host maps: BaryonifyShell.process() at BASELINE configs[2] with the shell's map pageable / registered in
place (LightconeShell(pinned="inplace")) / a page-locked copy (pinned=True), PaintProfilesShell.process() at the headline size with the
default output and with out= a page-locked array; results must be bit-identical to the pageable run."""
import os, sys, time, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import baryonforge_amd as bfg
from baryonforge_amd import engine, synthetic as syn

cosmo = dict(syn.COSMO)
nside = 1024
sync = torch.cuda.synchronize


def best(fn, reps=7):
    fn(); sync()
    b, out = 1e9, None
    for _ in range(reps):
        t0 = time.perf_counter(); out = fn(); sync(); b = min(b, time.perf_counter() - t0)
    return b * 1e3, out


ra, dec, M, z = syn.catalog(100_000, seed=42)
zd, Md, rd, d = syn.displacement_table()
m_in = syn.mass_map(nside)
Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
bm = bfg.Baryonification2D.from_arrays(zd, Md, rd, d, cosmo, epsilon_max=20)
ref = None
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for label, kw in (("pageable", {}), ("pinned='inplace' (registered in place)", {"pinned": "inplace"}), ("pinned=True (page-locked copy)", {"pinned": True})):
        t0 = time.perf_counter()
        shell = bfg.LightconeShell(map=m_in.copy(), cosmo=cosmo, **kw)
        t_make = (time.perf_counter() - t0) * 1e3
        R = bfg.BaryonifyShell(Cat, shell, 10, bm, verbose=False)
        for sl in (8, 4, 1):
            os.environ["BFG_BARY_SLICES"] = str(sl)
            ms, out = best(R.process)
            if ref is None:
                ref = out.copy()
            print(f"BaryonifyShell.process() {label:36s} slices {sl}: {ms:6.2f} ms   (shell made in {t_make:5.1f} ms)  max rel diff {float(np.max(np.abs(out - ref) / np.abs(ref).max())):.1e}  pinned {engine.is_pinned(shell.map)}", flush=True)
        os.environ.pop("BFG_BARY_SLICES")
        if kw.get("pinned"):
            os.environ["BFG_BARY_DOWN"] = "kernel"
            ms, out = best(R.process)
            print(f"BaryonifyShell.process() {label:36s} slices 8, download by copy kernel: {ms:6.2f} ms", flush=True)
            os.environ.pop("BFG_BARY_DOWN")
            runners = [bfg.BaryonifyShell(Cat, bfg.LightconeShell(map=m_in.copy(), cosmo=cosmo, **kw), 10, bm, verbose=False) for _ in range(8)]
            for dn in ("kernel", "dma"):
                os.environ["BFG_BARY_DOWN"] = dn
                ms8, outs = best(lambda: bfg.SimpleParallel(runners).process(), reps=3)
                print(f"SimpleParallel(8 shells, {label}) download by {dn}: {ms8 / 8:.2f} ms per shell", flush=True)
            os.environ.pop("BFG_BARY_DOWN")
        else:
            runners = [bfg.BaryonifyShell(Cat, bfg.LightconeShell(map=m_in.copy(), cosmo=cosmo), 10, bm, verbose=False) for _ in range(8)]
            ms8, outs = best(lambda: bfg.SimpleParallel(runners).process(), reps=3)
            print(f"SimpleParallel(8 shells, pageable): {ms8 / 8:.2f} ms per shell", flush=True)
        if kw.get("pinned") is True:
            engine.unpin(shell.map)

ra, dec, M, z = syn.catalog(1_000_000, seed=42)
zax, Max, rax, T = syn.pressure_table()
Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
R = bfg.PaintProfilesShell(Cat, bfg.LightconeShell(map=np.zeros(12 * nside * nside), cosmo=cosmo), 10,
                           bfg.TabulatedProfile.from_arrays(zax, Max, rax, T), verbose=False)
ms, ref = best(R.process)
ref = ref.copy()
print(f"PaintProfilesShell.process()            : {ms:6.2f} ms")
out = engine.pinned_empty(12 * nside * nside)
ms, o = best(lambda: R.process(out=out))
print(f"PaintProfilesShell.process(out=pinned)  : {ms:6.2f} ms  identical {bool(np.array_equal(o, ref))}")
for sl in (2, 4, 16):
    os.environ["BFG_D2H_SLICES"] = str(sl)
    ms, o = best(lambda: R.process(out=out))
    print(f"   BFG_D2H_SLICES={sl:2d}                     : {ms:6.2f} ms")
os.environ.pop("BFG_D2H_SLICES")
ms, _ = best(lambda: R.process_device())
print(f"PaintProfilesShell.process_device()     : {ms:6.2f} ms")
