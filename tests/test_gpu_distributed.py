"""
The multi-GPU path with the REAL kernels in every rank: two fresh child processes (subprocess + tests/dist_gpu_worker.py)
join a process group and run SplitJoinParallel(PaintProfilesShell) and SplitJoinParallel(BaryonifyShell) -- sharding by
sky patch, HIP paint / offsets / regrid kernels, reduce-scatter / all-reduce -- and every rank's result is compared with
the serial CPU oracle at the north-star tolerance (1e-5 relative on non-zero pixels), plus mass conservation.

One-GPU box: both ranks on cuda:0, gloo backend (device tensors staged through the host).  Two or more GPUs: one GPU
per rank, nccl (= RCCL over xGMI), and additionally the library's own communicator (bfg_allreduce_f64 & co.).
The reference forbids splitting Baryonify runners (utils/Parallelize.py:206-209); this build relies on the linearity of
pix_offsets in halos (Runners/HealpixRunner.py:355) -- which is what this test verifies end to end.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_real_kernels_paint_and_baryonify(tmp_path):
    import torch
    from util import assert_maps_close, oracle_baryonify, oracle_paint
    import dist_gpu_worker as W
    world = 2
    backend = "nccl" if torch.cuda.device_count() >= world else "gloo"
    port = _free_port()
    env = dict(os.environ, PYTHONPATH=REPO + os.pathsep + os.environ.get("PYTHONPATH", ""))
    procs = [subprocess.Popen([sys.executable, os.path.join(REPO, "tests", "dist_gpu_worker.py"), "--rank", str(r), "--world",
                               str(world), "--port", str(port), "--out", str(tmp_path), "--backend", backend],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = []
    for pr in procs:
        try:
            o, _ = pr.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o.decode(errors="replace"))
    for r, pr in enumerate(procs):
        assert pr.returncode == 0, f"rank {r} failed:\n{outs[r][-3000:]}"

    I = W.inputs()
    zax, Max, rax, T = I["paint"]
    ref, ptot = oracle_paint(I["cosmo"], I["ra"], I["dec"], I["M"], I["z"], (zax, Max, rax), T, W.NSIDE, W.EPS)
    dz, dM, dr, dtab = I["disp"]
    nb = W.N_BARY
    bref = oracle_baryonify(I["cosmo"], I["ra"][:nb], I["dec"][:nb], I["M"][:nb], I["z"][:nb], (dz, dM, dr), dtab, W.NSIDE,
                            W.EPS, 20.0, I["m_in"])
    assert not np.allclose(bref, I["m_in"])
    infos = [json.load(open(tmp_path / f"info_{r}.json")) for r in range(world)]
    for info in infos:
        assert info["maps"] and info["maps"][0].endswith("libbfg_mi355.so")       # the HIP library did the work
    collectives = ["torch"] + (["bfg"] if backend == "nccl" else [])
    for coll in collectives:
        idx = [np.load(tmp_path / f"idx_{coll}_{r}.npy") for r in range(world)]
        assert np.array_equal(np.sort(np.concatenate(idx)), np.arange(W.N_PAINT))  # every halo on exactly one rank
        assert min(i.size for i in idx) > W.N_PAINT // 8
        assert sum(info[f"paint_{coll}_pixel_updates"] for info in infos) == ptot  # ... and painted exactly once
        for r in range(world):
            got = np.load(tmp_path / f"paint_{coll}_{r}.npy")
            assert np.array_equal(got != 0, ref != 0)
            assert_maps_close(got, ref, 1e-5, what=f"2-rank paint ({backend}/{coll}), rank {r}")
            bgot = np.load(tmp_path / f"bary_{coll}_{r}.npy")
            assert_maps_close(bgot, bref, 1e-5, floor=1e-9, what=f"2-rank baryonify ({backend}/{coll}), rank {r}")
            assert np.isclose(bgot.sum(), I["m_in"].sum(), rtol=1e-10)            # mass conservation (:368-370)
        assert np.array_equal(np.load(tmp_path / f"paint_{coll}_0.npy"), np.load(tmp_path / f"paint_{coll}_1.npy"))
        # distributed BaryonifyShell: every rank uploaded the pixels it owns -- together one copy of the input map
        assert sum(info[f"bary_{coll}_h2d_bytes"] for info in infos) == 8 * 12 * W.NSIDE ** 2
        assert max(info[f"bary_{coll}_h2d_bytes"] for info in infos) <= 8 * (12 * W.NSIDE ** 2 // world + 64)
        # the list API: every shell equals its own serial oracle run on every rank; every (halo, pixel) pair painted once
        ptot_list = 0
        for k, (lo, hi) in enumerate(W.LIST_CUTS):
            lref, lp = oracle_paint(I["cosmo"], I["ra"][lo:hi], I["dec"][lo:hi], I["M"][lo:hi], I["z"][lo:hi],
                                    (zax, Max, rax), T, W.NSIDE, W.EPS)
            ptot_list += lp
            for r in range(world):
                lgot = np.load(tmp_path / f"list{k}_{coll}_{r}.npy")
                assert np.array_equal(lgot != 0, lref != 0)
                assert_maps_close(lgot, lref, 1e-5, what=f"2-rank list shell {k} ({backend}/{coll}), rank {r}")
        assert sum(info[f"list_{coll}_pixel_updates"] for info in infos) == ptot_list
    # one rank's shard empty (all halos in rank 0's declination stripe): same collectives on both ranks, right answers
    nn = W.north_mask(I)
    nref, _ = oracle_paint(I["cosmo"], I["ra"][nn], I["dec"][nn], I["M"][nn], I["z"][nn], (zax, Max, rax), T, W.NSIDE, W.EPS)
    nbref = oracle_baryonify(I["cosmo"], I["ra"][nn], I["dec"][nn], I["M"][nn], I["z"][nn], (dz, dM, dr), dtab, W.NSIDE,
                             W.EPS, 20.0, I["m_in"])
    for coll in collectives:
        assert sorted(info[f"north_{coll}_shard"] for info in infos) == [0, nn.size]
        for r in range(world):
            got = np.load(tmp_path / f"north_paint_{coll}_{r}.npy")
            assert np.array_equal(got != 0, nref != 0)
            assert_maps_close(got, nref, 1e-5, what=f"2-rank paint, one shard empty ({backend}/{coll}), rank {r}")
            bgot = np.load(tmp_path / f"north_bary_{coll}_{r}.npy")
            assert_maps_close(bgot, nbref, 1e-5, floor=1e-9, what=f"2-rank baryonify, one shard empty ({backend}/{coll}), rank {r}")
    # the owner-computes join painted every (halo, pixel) pair once as well, and only a border travelled point to point
    assert sum(info["owner_pixel_updates"] for info in infos) == ptot_list
    assert all(0 < info["owner_border_fraction"] < 0.2 for info in infos)
