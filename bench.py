#!/usr/bin/env python3
"""
bench.py -- headline benchmark of the MI355X shell-paint hot path (BASELINE.json metric:
halos/s + achieved HBM GB/s, NSIDE = 1024 shell, 1e6 halos, at 1/2/4/8 GPUs).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one full pass of the hot path over one synthetic catalog that is already
resident in HBM as float64 (M, z, ra, dec) records: zero the map, halo preparation kernel,
shell paint kernel, and -- for N > 1 -- the RCCL all-reduce of the per-rank maps.
Weak scaling: every rank paints its own sky-patch shard (sharding.shard_by_sky_patch) of `--halos` halos PER GPU
(N x halos in total); value = all halos painted by all ranks / max-over-ranks time.

Rank 0 prints ONE JSON line (see the contract in the task statement) with two extra objects:
  "roofline":     algorithmic bytes of the dominant kernel / its mean duration (HIP events on the
                  kernel's own stream, live in this process) against the 8 TB/s HBM peak
  "cpu_baseline": the CPU oracle (a C port of the reference loop; kind "port") timed on this
                  box's host cores on a bounded sample of the same workload (N = 1 only)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK = 8.0e12   # B/s, MI355X_MICROARCH.md "HBM3E peak BW 8.0 TB/s spec"


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--halos", type=int, default=1_000_000, help="halos per GPU")
    p.add_argument("--nside", type=int, default=1024)
    p.add_argument("--eps", type=float, default=10.0)
    p.add_argument("--workload", choices=["paint", "baryonify"], default="paint")
    p.add_argument("--variant", default="auto")
    p.add_argument("--table", choices=["default", "stress"], default="default")
    p.add_argument("--steep", action="store_true", help="dn/dlnM ~ M^-0.9 catalog instead of uniform log M")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of each baseline leg")
    return p.parse_args()


def cpu_baseline(args, cosmo, ra, dec, M, z, axes, T):
    """The oracle (oracle/bfg_oracle.c: a plain-C port of HealpixRunner.py:449-481) on a bounded sample of the
    same catalog, (1) single thread = Runner.process(), (2) split-join over the host cores =
    SplitJoinParallel (Parallelize.py:218-320: private full-size map per worker, summed by the parent)."""
    from oracle import oracle as orc
    cores = os.cpu_count() or 1
    a, R, D = orc.halo_scalars(cosmo, M, z)
    lnT = np.log(T)

    def run(n, njobs):
        t0 = time.perf_counter()
        _, ptot = orc.paint_shell(args.nside, ra[:n], dec[:n], M[:n], a[:n], D[:n], R[:n], axes, lnT, args.eps,
                                  njobs=njobs)
        return time.perf_counter() - t0, ptot
    n0 = min(5000, M.size)
    t, _ = run(n0, None)
    n1 = int(min(M.size, max(n0, n0 * args.cpu_seconds / max(t, 1e-3))))
    t1, p1 = run(n1, None)
    single = n1 / t1
    njobs = min(cores, 32)     # each worker owns a full-size float64 map (101 MB at NSIDE 1024)
    out = {"value": single, "unit": "halos/s", "cores": 1, "kind": "port",
           "sample": f"first {n1} halos of the same catalog, NSIDE {args.nside}, eps {args.eps:g}, "
                     f"oracle/bfg_oracle.c single thread, {t1:.1f} s, {p1 / t1:.3g} pixel-updates/s",
           "host_cores": cores, "single_thread_halos_per_s": single}
    if njobs > 1:
        rng = np.random.default_rng(42)                           # Parallelize.py:255 shuffle
        n2 = int(min(M.size, n1 * min(njobs, 8)))
        perm = rng.choice(n2, size=n2, replace=False)
        t0 = time.perf_counter()
        orc.paint_shell(args.nside, ra[:n2][perm], dec[:n2][perm], M[:n2][perm], a[:n2][perm], D[:n2][perm],
                        R[:n2][perm], axes, lnT, args.eps, njobs=njobs)
        t2 = time.perf_counter() - t0
        out["splitjoin_halos_per_s"] = n2 / t2
        out["splitjoin_workers"] = njobs
        out["splitjoin_sample"] = f"{n2} halos, {njobs} workers, {t2:.1f} s incl. per-worker map zeroing and the join"
        if n2 / t2 > single:
            out.update(value=n2 / t2, cores=njobs,
                       sample=f"first {n2} halos (seed-42 shuffled) of the same catalog, NSIDE {args.nside}, "
                              f"eps {args.eps:g}, split-join over {njobs} worker threads, {t2:.1f} s")
    return out


def _mark(msg):
    if os.environ.get("BFG_BENCH_VERBOSE"):
        print(f"[bench rank {os.environ.get('RANK', '0')}] {msg}", file=sys.stderr, flush=True)


def main():
    args = parse()
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X GPU (no CPU fallback for the product path)")
    # BFG_BENCH_BACKEND=gloo + BFG_BENCH_ONE_DEVICE=1: rehearsal of the multi-rank path on a one-GPU box
    # (all ranks on cuda:0, gloo all-reduce); the driver's runs use one GPU per rank and nccl (= RCCL).
    backend = os.environ.get("BFG_BENCH_BACKEND", "nccl")
    if os.environ.get("BFG_BENCH_ONE_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    _mark("process group up")
    import baryonforge_amd as bfg
    from baryonforge_amd import sharding, synthetic as syn
    from baryonforge_amd.background import Background
    from baryonforge_amd.engine import get_context

    cosmo = dict(syn.COSMO)
    nside, npix = args.nside, 12 * args.nside * args.nside
    n_total = args.halos * world
    ra, dec, M, z = syn.catalog(n_total, seed=42, steep=args.steep)
    shape = (10, 30, 100) if args.table == "default" else (2, 30, 2000)
    ctx = get_context(local_rank)
    bg = Background(cosmo)

    # this rank's sky-patch shard (the whole catalog at N = 1), resident in HBM before timing starts
    if world > 1:
        w = sharding.estimate_disc_pixels(cosmo, M, z, args.eps, nside)
        idx = sharding.shard_by_sky_patch(ra, dec, w, world)[rank]
    else:
        idx = np.arange(n_total)
    recs = np.stack([M[idx], z[idx], ra[idx], dec[idx]], axis=1)
    d_cat = ctx.to_device(recs)
    spline = ctx.da_spline(bg, float(np.max(z)))
    md = ctx.massdef_struct(bg, None)

    if args.workload == "paint":
        zax, Max, rax, T = syn.pressure_table(*shape)
        with np.errstate(all="ignore"):
            table = ctx.table([zax, Max, rax], np.log(T), log_values=True)
        sargs = ctx.shell_args(nside, d_cat, idx.size, 4, 0, args.eps, md, variant=args.variant)
        # N > 1: consecutive shells go to alternating map buffers, so the all-reduce of shell k (async, on RCCL's own
        # stream) overlaps the painting of shell k+1; every collective is waited for before its buffer is reused and
        # before the timed region ends (finish()).
        nbuf = 2 if dist is not None else 1
        d_maps = [ctx.zeros(npix) for _ in range(nbuf)]
        pending = [None] * nbuf
        counter = [0]

        def step():
            b = counter[0] % nbuf
            counter[0] += 1
            if pending[b] is not None:
                pending[b].wait()            # the current stream waits for that buffer's all-reduce
                pending[b] = None
            d_maps[b].zero_()
            ctx.paint_shell(sargs, table, spline, d_maps[b])
            if dist is not None:
                pending[b] = dist.all_reduce(d_maps[b], op=dist.ReduceOp.SUM, async_op=True)

        def finish():
            for b in range(nbuf):
                if pending[b] is not None:
                    pending[b].wait()
                    pending[b] = None
    else:
        zax, Max, rax, T = syn.displacement_table(*shape)
        table = ctx.table([zax, Max, rax], T, log_values=False)
        d_off = ctx.zeros(npix, 3)
        d_in = ctx.to_device(syn.mass_map(nside))
        d_map = ctx.zeros(npix)
        sargs = ctx.shell_args(nside, d_cat, idx.size, 4, 0, args.eps, md, model_md=md, model_epsilon_max=20.0,
                               variant=args.variant)

        def step():
            d_off.zero_()
            d_map.zero_()
            ctx.baryonify_offsets(sargs, table, spline, d_off)
            if dist is not None:
                dist.all_reduce(d_off, op=dist.ReduceOp.SUM)
            ctx.regrid_shell(nside, d_off, d_in, d_map, None)

        def finish():
            pass

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    _mark("inputs resident")
    for _ in range(args.warmup):
        step()
    finish()
    _mark("warmup issued")
    barrier()
    _mark("warmup done")
    ctx.stats_reset()
    ctx.timing_enable(True)          # hipEvents around each kernel, on the kernels' own stream
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    finish()                         # every outstanding collective has completed inside the timed region
    barrier()
    dt = time.perf_counter() - t0
    _mark("timed region done")
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    stats = ctx.stats()
    k_ms, k_n = ctx.timing_read(1)
    p_ms, p_n = ctx.timing_read(0)
    r_ms, r_n = ctx.timing_read(2)
    b_ms, b_n = ctx.timing_read(3)
    l_ms, l_n = ctx.timing_read(4)
    ctx.timing_enable(False)
    ptot_step = stats["pixel_updates"] / max(args.steps, 1)
    if dist is not None:
        pt = torch.tensor([ptot_step], dtype=torch.float64, device="cuda")
        dist.all_reduce(pt, op=dist.ReduceOp.SUM)
        ptot_all = float(pt.item())
    else:
        ptot_all = ptot_step

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    ms_per_step = dt / args.steps * 1e3
    value = n_total / (dt / args.steps)
    # ---- roofline of the dominant kernel (rank 0's launch) ------------------------------------
    # algorithmic bytes: 32 B catalog record per halo + one f64 atomic RMW (8 B read +
    # 8 B write) per (halo, pixel) [x3 for the offset field]; the map zero-fill / read-out term
    # 16 * Npix moves in hipMemset / D2H, outside this kernel, and is listed separately.
    per_px = 16.0 if args.workload == "paint" else 48.0
    kernel_bytes = 32.0 * idx.size + per_px * ptot_step
    kernel_s = (k_ms / max(k_n, 1)) * 1e-3
    achieved = kernel_bytes / kernel_s if kernel_s > 0 else 0.0
    traffic = None
    tfile = os.path.join(REPO, "profiles", "pmc_traffic.json")
    if os.path.exists(tfile):
        try:
            traffic = json.load(open(tfile)).get(f"{args.workload}_{args.variant}_n{args.halos}_nside{nside}")
        except Exception:
            traffic = None
    tile = args.variant in ("auto", "tile_lds")
    roofline = {"bound": "hbm", "kernel": "shell_tile_kernel" if tile else "shell_scatter_kernel",
                "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9,
                "unit": "GB/s", "frac": achieved / HBM_PEAK, "traffic": traffic,
                "kernel_ms": k_ms / max(k_n, 1), "kernel_launches": k_n,
                "algorithmic_bytes_per_launch": kernel_bytes,
                "bytes_per_halo": kernel_bytes / max(idx.size, 1),
                "pixel_updates_per_launch": ptot_step,
                "pixel_updates_per_s": ptot_step / kernel_s if kernel_s > 0 else 0.0,
                "prep_kernel_ms": p_ms / max(p_n, 1),
                "tile_binning_ms": (b_ms / b_n) if b_n else None,
                "leftover_scatter_kernel_ms": (l_ms / l_n) if l_n else None,
                "regrid_kernel_ms": (r_ms / r_n) if r_n else None,
                "step_algorithmic_GBps": (kernel_bytes + 16.0 * npix * (1 if args.workload == "paint" else 8)) /
                                         (dt / args.steps) / 1e9}
    out = {
        "metric": "halos_per_s", "value": value, "unit": "halos/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{'PaintProfilesShell' if args.workload == 'paint' else 'BaryonifyShell'}: "
                               f"{args.halos} halos per GPU ({n_total} total), NSIDE={nside}, epsilon_max={args.eps:g}, "
                               f"{'TabulatedProfile(Pressure)' if args.workload == 'paint' else 'Baryonification2D'} "
                               f"table {shape[0]}x{shape[1]}x{shape[2]}, catalog "
                               f"{'dn/dlnM~M^-0.9' if args.steep else 'log10M~U(12,15.5)'}, z~U(0.4,0.5), seed 42",
                   "variant": args.variant, "halos_per_gpu": args.halos, "nside": nside,
                   "sharding": "sky patch (contiguous NEST ranges of nside-8 patches, balanced by pixel work) + RCCL all-reduce of the map, overlapped with the next shell (two map buffers)" if world > 1 else "none",
                   "pixel_updates_total_per_step": ptot_all},
        "roofline": roofline,
    }
    if world == 1 and not args.no_cpu_baseline and args.workload == "paint":
        out["cpu_baseline"] = cpu_baseline(args, cosmo, ra, dec, M, z, (zax, Max, rax), T)
        out["cpu_baseline"]["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
