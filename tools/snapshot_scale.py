#!/usr/bin/env python3
"""BASELINE config 4 at scale: BaryonifySnapshot 3D on n^3 particles (default 512^3) + 1e5 halos + CIC deposit.
Times the C-ABI calls with inputs resident in HBM (parity: tests/test_gpu_parity.py, tests/soak/soak_aux.py)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import baryonforge_amd as bfg
from baryonforge_amd import synthetic as syn
from baryonforge_amd.background import Background
from baryonforge_amd.engine import get_context

n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 512
nhalo = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
L = float(sys.argv[3]) if len(sys.argv) > 3 else 1000.0
ngrid = int(sys.argv[4]) if len(sys.argv) > 4 else 512
cosmo = dict(syn.COSMO)
ctx = get_context(0)
dev = ctx.device
g = torch.Generator(device=dev); g.manual_seed(7)
npart = n1 ** 3
# a jittered lattice (what an N-body IC looks like), built on the device
ax = (torch.arange(n1, device=dev, dtype=torch.float64) + 0.5) * (L / n1)
P = torch.stack(torch.meshgrid(ax, ax, ax, indexing="ij"), dim=-1).reshape(-1, 3)
P = (P + (torch.rand(P.shape, generator=g, device=dev, dtype=torch.float64) - 0.5) * (L / n1)) % L
if os.environ.get('SNAP_SHUFFLE'):
    P = P[torch.randperm(P.shape[0], device=dev, generator=g)].contiguous()      # arbitrary particle order
rng = np.random.default_rng(3)
H = rng.uniform(0, L, (nhalo, 3)).astype(">f4").astype(np.float64)
hM = (10 ** rng.uniform(13.0, 15.3, nhalo)).astype(">f4")
zs = 0.25
halos = np.stack([hM.astype(np.float64), np.log(hM).astype(np.float64), H[:, 0], H[:, 1], H[:, 2]], axis=1)
d_halo = ctx.to_device(halos)
zax, Max, rax, d = syn.displacement_table()
table = ctx.table([zax, Max, rax], d, log_values=False)
bg = Background(cosmo)
md = ctx.massdef_struct(bg, None)
d_out = torch.empty_like(P)
a = 1 / (1 + zs)

def run():
    ctx.baryonify_snapshot(P, d_halo, 3, L, a, 10.0, md, md, 20.0, False, 0, table, d_out)
    return ctx.deposit_grid(d_out, None, L, ngrid, "cic")

ctx.stats_reset(); grid = run(); torch.cuda.synchronize(); st = ctx.stats()
t0 = time.perf_counter(); reps = 3
for _ in range(reps):
    t1 = time.perf_counter(); ctx.baryonify_snapshot(P, d_halo, 3, L, a, 10.0, md, md, 20.0, False, 0, table, d_out); torch.cuda.synchronize()
    t2 = time.perf_counter(); grid = ctx.deposit_grid(d_out, None, L, ngrid, "cic"); torch.cuda.synchronize()
    t3 = time.perf_counter()
    print(f"displace {1e3*(t2-t1):8.2f} ms   deposit {1e3*(t3-t2):7.2f} ms")
dt = (time.perf_counter() - t0) / reps
print(f"{npart} particles, {nhalo} halos, L {L}: {1e3*dt:.2f} ms per pass = {nhalo/dt:.3e} halos/s; (halo, particle) pairs {st['pixel_updates']:.4g}; grid sum {float(grid.sum()):.6e} vs {npart}")
