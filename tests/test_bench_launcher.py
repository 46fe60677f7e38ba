"""
bench.py as its own launcher (`python bench.py --gpus N`, N > 1, no WORLD_SIZE): the parent spawns N fresh ranks before any GPU
call, relays rank 0's line and fails with one line when a rank does -- plus the deadline watchdog, which must fire while the
main thread is blocked in a C call.  CPU only: the ranks here are stub scripts (or bench.py itself refusing to run without GPUs).
"""
import io
import json
import os
import subprocess
import sys
import textwrap
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def _stub(tmp_path, body):
    path = tmp_path / "rank_stub.py"
    path.write_text(textwrap.dedent(body))
    return str(path)


def test_launcher_sets_the_rank_environment_and_relays_rank0_only(tmp_path):
    import bench
    stub = _stub(tmp_path, """
        import json, os, sys
        r = int(os.environ["RANK"])
        env = {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "BFG_BENCH_SPAWNED")}
        json.dump(env, open(os.path.join(sys.argv[1], f"env_{r}.json"), "w"))
        print(json.dumps({"rank": r, "argv": sys.argv[2:]}))          # every rank prints: only rank 0's line may reach stdout
    """)
    out, err = io.StringIO(), io.StringIO()
    rc = bench.spawn_ranks(3, [str(tmp_path), "--gpus", "3", "--steps", "7"], script=stub, out=out, err=err)
    assert rc == 0
    lines = [l for l in out.getvalue().splitlines() if l.strip()]
    assert len(lines) == 1 and json.loads(lines[0]) == {"rank": 0, "argv": ["--gpus", "3", "--steps", "7"]}
    assert sorted(json.loads(l)["rank"] for l in err.getvalue().splitlines() if l.strip()) == [1, 2]
    envs = [json.load(open(tmp_path / f"env_{r}.json")) for r in range(3)]
    assert [e["RANK"] for e in envs] == ["0", "1", "2"] and [e["LOCAL_RANK"] for e in envs] == ["0", "1", "2"]
    assert all(e["WORLD_SIZE"] == "3" and e["MASTER_ADDR"] == "127.0.0.1" and e["BFG_BENCH_SPAWNED"] == "1" for e in envs)
    assert len({e["MASTER_PORT"] for e in envs}) == 1 and int(envs[0]["MASTER_PORT"]) > 0


def test_launcher_fails_with_the_first_failing_rank_and_kills_the_survivors(tmp_path):
    import bench
    stub = _stub(tmp_path, """
        import os, sys, time
        r = int(os.environ["RANK"])
        open(os.path.join(sys.argv[1], f"pid_{r}"), "w").write(str(os.getpid()))
        if r == 1:
            time.sleep(0.3)
            sys.exit(7)
        time.sleep(600)                                               # a peer stuck in a collective
    """)
    out, err = io.StringIO(), io.StringIO()
    t0 = time.monotonic()
    rc = bench.spawn_ranks(3, [str(tmp_path)], script=stub, grace=0.5, out=out, err=err)
    assert rc == 7 and time.monotonic() - t0 < 30
    assert out.getvalue() == ""
    assert "FAILED (launcher, 3 ranks): rank 1 exited with code 7" in err.getvalue()
    for r in (0, 2):                                                  # the stuck ranks are gone
        pid = int(open(tmp_path / f"pid_{r}").read())
        try:
            os.kill(pid, 0)
            alive = open(f"/proc/{pid}/stat").read().split()[2] != "Z"
        except (ProcessLookupError, FileNotFoundError):
            alive = False
        assert not alive


def test_launcher_deadline(tmp_path):
    import bench
    stub = _stub(tmp_path, "import time; time.sleep(600)")
    err = io.StringIO()
    t0 = time.monotonic()
    rc = bench.spawn_ranks(2, [], script=stub, deadline=1.0, out=io.StringIO(), err=err)
    assert rc == 3 and time.monotonic() - t0 < 30 and "deadline" in err.getvalue()


def test_plain_python_gpus2_spawns_ranks_that_refuse_without_gpus():
    """the review's case: `python3 bench.py --gpus 2` as plain python.  On this box (no GPU) both ranks must come up as ranks of a
    world of two (not die on WORLD_SIZE) and refuse with the device count; the launcher reports the failure and exits non-zero."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["HIP_VISIBLE_DEVICES"] = ""                                   # (also without GPUs on a GPU box)
    env["ROCR_VISIBLE_DEVICES"] = ""
    pr = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                        capture_output=True, text=True, timeout=300)
    assert pr.returncode != 0 and pr.stdout.strip() == ""
    assert "0 GPU(s) visible, 2 needed" in pr.stderr and "WORLD_SIZE is" not in pr.stderr
    assert "FAILED (launcher, 2 ranks)" in pr.stderr


def test_watchdog_fires_while_the_main_thread_is_blocked_in_a_c_call(tmp_path):
    """a SIGALRM handler would wait for the interpreter; the watchdog thread does not"""
    stub = _stub(tmp_path, f"""
        import ctypes, sys
        sys.path.insert(0, {REPO!r})
        import bench

        def overdue():
            print("deadline line", file=sys.stderr, flush=True)
            return 5
        bench.WATCHDOG.arm(0.5, overdue)
        ctypes.CDLL(None).sleep(600)                                  # blocked outside the interpreter, GIL released
    """)
    t0 = time.monotonic()
    pr = subprocess.run([sys.executable, stub], capture_output=True, text=True, timeout=120)
    assert pr.returncode == 5 and "deadline line" in pr.stderr and time.monotonic() - t0 < 60


def test_unknown_legs_are_refused_before_anything_runs():
    """a typo in --legs ends the run at argument parsing: exit 2, usage line, no GPU touched, no rank spawned"""
    t0 = time.monotonic()
    pr = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--legs", "weak,confgs3"],
                        capture_output=True, text=True, timeout=120)
    assert pr.returncode == 2 and "unknown --legs entries ['confgs3']" in pr.stderr and pr.stdout == ""
    assert time.monotonic() - t0 < 30
    import bench
    assert set(bench.N1_ONLY_LEGS) <= set(bench.LEG_ARGS) and {"weak", "owner", "configs3"} <= set(bench.LEG_ARGS)


def test_the_line_is_printed_once_when_the_legs_deadline_races_the_main_thread(tmp_path):
    """the legs' deadline fires on the watchdog's thread while the main thread reaches its own emit(): ONE line.
    The stub runs bench._main's leg loop with a run_config stand-in (no GPU): the single leg ends just as the deadline fires."""
    stub = _stub(tmp_path, f"""
        import sys, time, types, json
        sys.path.insert(0, {REPO!r})
        import bench
        calls = []

        def fake_run_config(args, torch, dist, rank, local_rank, world, backend, main=True):
            if not main:
                time.sleep(0.5)                                      # the leg ends when the legs' deadline (0.5 s) fires
            return {{"value": 1.0, "unit": "halos/s", "ms_per_step": 1.0, "scaling": "strong", "steps": 1, "ramp_steps": 0,
                    "config": {{"workload": "w", "halos_total": 1, "sharding": "none"}}, "roofline": {{"frac": 0.5}}}}
        bench.run_config = fake_run_config
        fake_torch = types.SimpleNamespace(cuda=types.SimpleNamespace(empty_cache=lambda: None))
        sys.argv = ["bench.py", "--legs", "configs1"]
        args = bench.parse()
        import os
        os.environ["BFG_BENCH_LEGS_DEADLINE_S"] = "0.5"
        bench._main(args, fake_torch, None, 0, 0, 1, "nccl")
        time.sleep(0.3)
    """)
    for _ in range(3):
        pr = subprocess.run([sys.executable, stub], capture_output=True, text=True, timeout=120)
        lines = [ln for ln in pr.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1, pr.stdout + pr.stderr
        assert "legs" in json.loads(lines[0])


def test_eight_ranks_rehearsal_on_cpu_prints_the_strong_scaling_line_with_all_legs():
    """`bench.py --gpus 8` as the driver runs it, rehearsed without a GPU: the launcher's eight ranks (tests/bench_host_stub.py: bench.py
    unchanged, the device side replaced by the CPU oracle) form a gloo group, shard a small catalog, run the product API's sliced
    all-reduce path, then the weak / owner / configs3 legs -- and rank 0 prints ONE line with `ranks` of length 8, `vs_n1` and the
    three legs.  The maps themselves are checked in tests/test_distributed_cpu.py; here the point is that every N-rank code path of
    bench.py runs to its end."""
    import bench
    out, err = io.StringIO(), io.StringIO()
    env = {"BFG_BENCH_BACKEND": "gloo", "BFG_BENCH_ONE_DEVICE": "1", "BFG_BENCH_RAMP_S": "0", "BFG_BENCH_DEADLINE_S": "500",
           "BFG_BENCH_LEGS_DEADLINE_S": "300", "OMP_NUM_THREADS": "1"}
    rc = bench.spawn_ranks(8, ["--gpus", "8", "--steps", "2", "--warmup", "1", "--halos", "480", "--nside", "32", "--slices", "3",
                               "--legs", "weak,owner,configs3"],
                           script=os.path.join(REPO, "tests", "bench_host_stub.py"), extra_env=env, deadline=560, out=out, err=err)
    assert rc == 0, err.getvalue()[-2000:]
    lines = [ln for ln in out.getvalue().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.getvalue()
    r = json.loads(lines[0])
    assert r["n_gpus"] == 8 and r["scaling"] == "strong" and r["metric"] == "halos_per_s" and r["value"] > 0
    assert len(r["ranks"]) == 8 and sorted(x["rank"] for x in r["ranks"]) == list(range(8))
    assert sum(x["shard_halos"] for x in r["ranks"]) == 480
    assert r["vs_n1"] is not None and r["n1"]["halos"] == 480
    assert r["rccl_ranks"] == 0 and r["backend"] == "gloo"            # a rehearsal: RCCL took no part
    # the owner-computes join is either the guarded leg or -- if the all-reduce run was exchange-bound and the join passed its
    # self-check and was faster -- the main line, with the all-reduce's measurement kept as the leg `allreduce` (promote_owner)
    choice = r["exchange_choice"]
    assert choice["picked"] in ("owner", "allreduce") and choice["owner_selfcheck"] == "passed"
    other = "allreduce" if choice["picked"] == "owner" else "owner"
    assert set(r["legs"]) == {"weak", other, "configs3"}
    for name, leg in r["legs"].items():
        assert "error" not in leg and leg["value"] > 0 and len(leg["ranks"]) == 8, (name, leg)
    assert r["legs"]["weak"]["scaling"] == "weak" and r["legs"]["weak"]["halos_total"] == 8 * 480
    owner_rec = r if choice["picked"] == "owner" else r["legs"]["owner"]
    assert "owner-computes" in owner_rec["exchange"]["mode"]
    assert ("stripes" in r["config"]["sharding"]) == (choice["picked"] == "owner")
    assert r["legs"]["configs3"]["workload"].startswith("BaryonifyShell")


def test_owner_join_is_promoted_only_when_the_run_is_exchange_bound_and_the_join_checked_out():
    """bench.promote_owner on stand-in records: promotion needs (i) all-reduce alone slower than painting alone, (ii) the owner leg's
    self-check passed, (iii) the owner leg faster; the all-reduce measurement is then kept as the leg `allreduce`"""
    import argparse
    import bench

    def rec(value, exch, selfcheck=None, a_ms=2.0, c_ms=1.0):
        return {"value": value, "unit": "halos/s", "ms_per_step": 1.0, "scaling": "strong", "steps": 2, "ramp_steps": 0, "dtype": "f64",
                "n_gpus": 8, "warmup": 1, "config": {"workload": "w", "halos_total": 10, "sharding": exch}, "roofline": {"frac": 0.5},
                "ranks": [{"rank": r, "shard_halos": 1, "compute_ms": c_ms, "allreduce_ms": a_ms} for r in range(2)],
                "exchange": {"mode": exch, "selfcheck": selfcheck}, "rccl_ranks": 8, "n1": {"value": 5.0}, "vs_n1": value / 5.0}
    args = argparse.Namespace(exchange="allreduce", workload="paint")
    # exchange-bound, checked, faster: promoted
    out, done = rec(10.0, "all-reduce"), {"owner": {"value": 20.0}}
    bench.promote_owner(out, rec(20.0, "owner-computes stripes", "passed"), done, args)
    assert out["value"] == 20.0 and out["exchange_choice"]["picked"] == "owner" and out["vs_n1"] == 4.0 and out["rccl_ranks"] == 8
    assert set(done) == {"allreduce"} and done["allreduce"]["value"] == 10.0 and "stripes" in out["config"]["sharding"]
    # compute-bound / self-check not passed / slower / forced exchange: the all-reduce stays
    for main, leg, a in ((rec(10.0, "all-reduce", a_ms=0.5), rec(20.0, "owner", "passed"), args),
                         (rec(10.0, "all-reduce"), rec(20.0, "owner", "failed (maps differ)"), args),
                         (rec(10.0, "all-reduce"), rec(9.0, "owner", "passed"), args),
                         (rec(10.0, "all-reduce"), rec(20.0, "owner", "passed"), argparse.Namespace(exchange="owner", workload="paint"))):
        done = {"owner": {"value": leg["value"]}}
        bench.promote_owner(main, leg, done, a)
        assert main["value"] == 10.0 and main["exchange_choice"]["picked"] == "allreduce" and set(done) == {"owner"}


def test_roofline_block_cannot_pass_one_and_keeps_the_algorithmic_figure():
    """bench.finish_roofline: `algorithmic_frac` is always SURVEY 8(d) bytes / time / 8 TB/s; `frac` is that only
    where the bytes travel, otherwise a counted issue fraction (<= 1 by construction) in the unit of the resource that bounds the kernel"""
    import bench

    def block(alg_bytes, kernel_s):
        a = alg_bytes / kernel_s
        return {"bound": "hbm", "achieved": a / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": a / 8e12, "algorithmic_bytes_per_launch": alg_bytes}
    sq = {"raw": {"SQ_INSTS_VALU": 3.7756e8, "SQ_LDS_IDX_ACTIVE": 3.2365e8}, "valu_issue_frac": 0.70, "lds_pipe_frac": 0.59}
    # the headline tile kernel: 4.469 GB algorithmic in 0.92 ms, 0.32 GB measured -> the bytes do not travel: VALU-counted
    r = bench.finish_roofline(block(4.469e9, 0.92e-3), sq, 0.92e-3, 0.32e9)
    assert r["bound"] == "valu" and abs(r["algorithmic_frac"] - 4.469e9 / 0.92e-3 / 8e12) < 1e-12
    assert abs(r["frac"] - 3.7756e8 * 64 / 0.92e-3 / (256 * 4 * 16 * 2.4e9)) < 1e-12 and r["frac"] <= 1.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["unit"] == "Tlane-op/s"
    assert r["hbm"]["frac"] == r["algorithmic_frac"] and r["hbm"]["traffic"] == 0.32e9 and r["lds_counted_frac"] < r["valu_counted_frac"]
    # the offsets kernel at NSIDE 2048: algorithmic fraction 1.25 -> never reported as `frac`
    sq3 = {"raw": {"SQ_INSTS_VALU": 2.8626e9, "SQ_LDS_IDX_ACTIVE": 2.4086e9}}
    r = bench.finish_roofline(block(66.5e9, 6.65e-3), sq3, 6.65e-3, 3.9e9)
    assert r["algorithmic_frac"] > 1.0 and r["bound"] == "valu" and 0.0 < r["frac"] <= 1.0
    # no stored counters and an algorithmic fraction above 1: withheld, not invented
    r = bench.finish_roofline(block(66.5e9, 6.65e-3), None, 6.65e-3, None)
    assert r["frac"] is None and r["algorithmic_frac"] > 1.0 and "no stored counters" in r["bound"]
    # a kernel whose bytes do travel (the deposit: 11.3 GB measured in 2.6 ms) stays on the HBM yardstick
    r = bench.finish_roofline(block(20.4e9, 2.6e-3), sq, 2.6e-3, 11.3e9)
    assert r["bound"] == "hbm" and r["frac"] == r["algorithmic_frac"] <= 1.0 and r["unit"] == "GB/s"
    # LDS-bound when the LDS pipe is the fuller one
    r = bench.finish_roofline(block(4.0e9, 1e-3), {"raw": {"SQ_INSTS_VALU": 1e8, "SQ_LDS_IDX_ACTIVE": 5e8}}, 1e-3, 0.1e9)
    assert r["bound"] == "lds" and abs(r["frac"] - 5e8 / 256 / (1e-3 * 2.4e9)) < 1e-12


def test_single_gpu_legs_rehearsal_on_cpu_every_leg_has_a_cpu_baseline_and_no_frac_above_one():
    """`bench.py` at N = 1 with its paint / baryonify legs, rehearsed without a GPU (tests/bench_host_stub.py: bench.py unchanged, the
    device side replaced by the oracle, timings meaningless): ONE line; the main line and every leg carry a non-null `cpu_baseline`
    with value / unit / cores / kind / sample; every roofline block keeps `algorithmic_frac` and never reports a `frac` above 1
    ("BENCH_r06 legs all carry non-null cpu_baseline, no frac > 1")"""
    env = dict(os.environ, BFG_BENCH_RAMP_S="0", BFG_STUB_SMALL_LEGS="1", OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    pr = subprocess.run([sys.executable, os.path.join(REPO, "tests", "bench_host_stub.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                         "--halos", "300", "--nside", "32", "--no-e2e", "--cpu-seconds", "0.2", "--cpu-seconds-leg", "0.2",
                         "--legs", "configs1,configs2,steep"], env=env, capture_output=True, text=True, timeout=600)
    assert pr.returncode == 0, pr.stderr[-2000:]
    lines = [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, pr.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 1 and r["metric"] == "halos_per_s" and r["vs_baseline"] is None and r["dtype"] == "f64"
    assert set(r["legs"]) == {"configs1", "configs2", "steep"}
    for name, rec in [("main", r)] + list(r["legs"].items()):
        assert "error" not in rec, (name, rec)
        cb = rec["cpu_baseline"]
        assert cb and cb["value"] > 0 and cb["unit"] == "halos/s" and cb["cores"] >= 1 and cb["kind"] == "port" and cb["sample"], (name, cb)
        rf = rec["roofline"]
        assert rf["algorithmic_frac"] > 0 and (rf["frac"] is None or 0 < rf["frac"] <= 1.0), (name, rf["frac"], rf["algorithmic_frac"])
        assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(rf)
    assert r["legs"]["configs2"]["workload"].startswith("BaryonifyShell") and r["legs"]["configs2"]["cpu_baseline"]["cores"] == 1
