#!/usr/bin/env python3
"""Profiling aid: barrier-to-barrier cycle breakdown of shell_tile_kernel.

Needs a library built with -DBFG_STAGE_TIMING=1 (BFG_SO=path/to/that.so).  Prints, for the default bench workload,
the share of workgroup time spent in each stage of the chunk loop (measured by one thread per workgroup)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (one HIP runtime per process)
from baryonforge_amd import _lib, synthetic as syn
from baryonforge_amd.background import Background
from baryonforge_amd.engine import get_context

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
nside = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
workload = sys.argv[3] if len(sys.argv) > 3 else "paint"
cosmo = dict(syn.COSMO)
ra, dec, M, z = syn.catalog(n, seed=42, steep=bool(os.environ.get("BFG_ST_STEEP")))
ctx = get_context(0)
bg = Background(cosmo)
d_cat = ctx.to_device(np.stack([M, z, ra, dec], axis=1))
spline = ctx.da_spline(bg, float(np.max(z)))
md = ctx.massdef_struct(bg, None)
npix = 12 * nside * nside
if workload == "paint":
    zax, Max, rax, T = syn.pressure_table(10, 30, 100)
    table = ctx.table([zax, Max, rax], np.log(T), log_values=True)
    d_map = ctx.zeros(npix)
    sargs = ctx.shell_args(nside, d_cat, n, 4, 0, 10.0, md, out_overwrite=True)
    run = lambda: ctx.paint_shell(sargs, table, spline, d_map)
else:
    zax, Max, rax, T = syn.displacement_table(10, 30, 100)
    table = ctx.table([zax, Max, rax], T, log_values=False)
    d_off = ctx.zeros(npix, 3)
    sargs = ctx.shell_args(nside, d_cat, n, 4, 0, 10.0, md, model_md=md, model_epsilon_max=20.0, out_overwrite=True)
    run = lambda: ctx.baryonify_offsets(sargs, table, spline, d_off)
L = _lib.load()
fn = L.bfg_debug_stage_cycles
fn.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
out = (ctypes.c_ulonglong * 16)()
run(); run()
fn(ctx.handle, out, 1)
reps = 3
for _ in range(reps):
    run()
fn(ctx.handle, out, 1)
v = np.array(list(out), dtype=np.float64) / reps
names = ["a: pair records (wave 0) + barrier", "b: ring windows -> segments + barrier", "p: next chunk's record loads issued",
         "c1: pixel->segment table + barrier", "c2: pixel loop (own pixels)", "c3: wait at the end-of-chunk barrier",
         "prologue: LDS clear, tables, ring rows, first records", "epilogue: final drain, counters, write-back"]
sub = v[8:]
v = v[:8]
tot = v.sum()
print(f"workload {workload} n={n} nside={nside}: {tot:.4g} cycles summed over workgroups per launch")
for nm, x in zip(names, v):
    if x:
        print(f"  {nm:45s} {x:12.4g}  {100 * x / tot:5.1f} %")
if os.environ.get("BFG_ST_MODE") == "4":
    allv = np.array(list(out), dtype=np.float64) / reps
    lab = ["t0: top of the item: counter atomic issued, work record of item k + 2 arrived, list prefetch issued",
           "t1: accumulator clear, ring rows (wave 1), first records (wave 0)", "t2: stage a + its barrier (first chunk incl. wait for the records)",
           "t3: stage b: blend / window DMA issue, halo records, ring windows, segments", "t4: barrier after stage b (vmcnt(0): loads, DMA, earlier stores)",
           "t5: next records issued, pixel loop", "t6: end-of-chunk barrier (+ queue drain check)", "t7: write-back: addresses, deferred pixels, stores issued",
           "t8: end-of-item barrier", "t7a: (of the write-back) look-ahead parked, deferred pixels", "t7b: (of the write-back) LDS reads of the thread's pixels"]
    tot = allv[:11].sum()
    print(f"workload {workload} n={n} nside={nside}: thread {os.environ.get('BFG_ST_TID', '64')}: {tot:.4g} cycles per launch summed over workgroups")
    for nm, x in zip(lab, allv[:11]):
        print(f"    {nm:100s} {x:12.4g}  {100 * x / tot:5.1f} %")
    sys.exit(0)
if sub.sum() and os.environ.get("BFG_ST_MODE") == "3":
    subn = ["i0: top of the item (counter atomic, work record, list prefetch)", "i1: accumulator clear", "i2: ring rows",
            "i3: first records / wave-0 priming", "i4: chunk loop", "i5: write-back addresses, deferred pixels",
            "i6: write-back stores", "i7: end-of-item barrier"]
    print(f"  inside a work item (thread {os.environ.get('BFG_ST_TID', '64')} of every workgroup; a -DBFG_STAGE_TIMING=3 build):")
    for nm, x in zip(subn, sub):
        print(f"    {nm:62s} {x:12.4g}  {100 * x / sub.sum():5.1f} %")
elif sub.sum():
    subn = ["b0: window DMA issued, pair of the slot found", "b1: halo record arrived", "b2: ring window, clipping, segment constants",
            "b3: pixel-list offsets (scan + LDS atomics)", "b4: segment records + pixel->segment table", "b5: barrier (other waves, window DMA)"]
    print("  inside stage b (wave 1 of every workgroup; a -DBFG_STAGE_TIMING=2 build):")
    for nm, x in zip(subn, sub):
        print(f"    {nm:50s} {x:12.4g}  {100 * x / sub.sum():5.1f} %")
