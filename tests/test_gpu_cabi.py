"""
The drop-in boundary on its own: libbfg_mi355.so driven through ctypes with plain pointers and sizes only
(no torch tensors, no numpy device arrays): context on its own stream, bfg_dev_malloc / bfg_memcpy_*,
table + spline upload, bfg_paint_shell, bfg_baryonify_offsets + bfg_regrid_shell, bfg_stats_read --
checked against the CPU oracle.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from baryonforge_amd import _lib, synthetic as syn
from baryonforge_amd.background import Background, RHO_CRITICAL
from oracle import oracle as orc
from util import assert_maps_close


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Dev(object):
    def __init__(self, L, ctx, nbytes):
        self.L, self.ctx, self.n = L, ctx, nbytes
        self.p = C.c_void_p()
        _lib.check(L.bfg_dev_malloc(ctx, nbytes, C.byref(self.p)), "malloc")

    def up(self, arr):
        arr = np.ascontiguousarray(arr) if np.asarray(arr).dtype in (np.int32, np.uint32, np.int64) else \
            np.ascontiguousarray(arr, dtype=np.float64)
        _lib.check(self.L.bfg_memcpy_h2d(self.ctx, self.p, arr.ctypes.data, arr.nbytes), "h2d")
        return self

    def zero(self):
        _lib.check(self.L.bfg_dev_memset_zero(self.ctx, self.p, self.n), "memset")
        return self

    def down(self, shape, dtype=np.float64):
        out = np.empty(shape, dtype=dtype)
        _lib.check(self.L.bfg_memcpy_d2h(self.ctx, out.ctypes.data, self.p, out.nbytes), "d2h")
        return out

    def free(self):
        _lib.check(self.L.bfg_dev_free(self.ctx, self.p), "free")


def _massdef(bg):
    m = _lib.MassDefStruct()
    m.Omega_m, m.Omega_l, m.Omega_r, m.w0, m.h = bg.Omega_m, bg.Omega_l, bg.Omega_r, bg.w0, bg.h
    m.rho_crit0_h2, m.Delta, m.rho_type = RHO_CRITICAL, 200.0, 0
    return m


def _table(L, ctx, axes, values, flags):
    axes = [np.ascontiguousarray(a, dtype=np.float64) for a in axes]
    values = np.ascontiguousarray(values, dtype=np.float64)
    h = C.c_void_p()
    _lib.check(L.bfg_table_create(ctx, 3, (C.c_int64 * 3)(*values.shape), (C.POINTER(C.c_double) * 3)(*map(_dp, axes)),
                                  _dp(values), flags, C.byref(h)), "table")
    return h


@pytest.mark.parametrize("variant", [0, 2, 3])
def test_cabi_paint_and_baryonify_without_torch_types(cosmo, variant):
    from scipy import interpolate
    L = _lib.load()
    ctx = C.c_void_p()
    _lib.check(L.bfg_ctx_create(0, C.c_void_p(-1), C.byref(ctx)), "ctx")          # BFG_STREAM_OWN
    name = C.create_string_buffer(128)
    ncu = C.c_int()
    _lib.check(L.bfg_ctx_device_info(ctx, name, 128, C.byref(ncu), None, None))
    assert b"gfx950" in name.value and ncu.value >= 64

    nside, npix, eps = 128, 12 * 128 * 128, 10.0
    ra, dec, M, z = syn.catalog(800, seed=21, z=(0.1, 0.4))
    bg = Background(cosmo)
    z_t = np.linspace(0, z.max() + 0.1, 1000)
    cs = interpolate.CubicSpline(z_t, bg.angular_diameter_distance(1 / (1 + z_t)))
    spl = C.c_void_p()
    knots, coef = np.ascontiguousarray(cs.x), np.ascontiguousarray(cs.c)
    _lib.check(L.bfg_spline_create(ctx, knots.size, _dp(knots), _dp(coef), C.byref(spl)), "spline")
    d_cat = Dev(L, ctx, 800 * 32).up(np.stack([M, z, ra, dec], 1))
    a, R, D = orc.halo_scalars(cosmo, M, z)

    args = _lib.ShellArgs()
    args.nside, args.n_halo, args.d_catalog, args.cat_stride, args.n_extra = nside, 800, d_cat.p.value, 4, 0
    args.epsilon_max, args.runner_md, args.model_md = eps, _massdef(bg), _massdef(bg)
    args.model_epsilon_max, args.variant = 20.0, variant

    # ---- paint
    zax, Max, rax, T = syn.pressure_table()
    tab = _table(L, ctx, (zax, Max, rax), np.log(T), _lib.BFG_TABLE_LOG_VALUES)
    d_map = Dev(L, ctx, npix * 8).zero()
    _lib.check(L.bfg_stats_reset(ctx))
    _lib.check(L.bfg_paint_shell(ctx, C.byref(args), tab, spl, d_map.p), "paint")
    st = _lib.Stats()
    _lib.check(L.bfg_stats_read(ctx, C.byref(st)))
    ref, ptot = orc.paint_shell(nside, ra, dec, M, a, D, R, (zax, Max, rax), np.log(T), eps)
    assert st.pixel_updates == ptot
    assert_maps_close(d_map.down(npix), ref, 1e-5, what="C-ABI paint")
    # wrong table kind is refused, bad arguments are refused
    lin = _table(L, ctx, (zax, Max, rax), T, 0)
    assert L.bfg_paint_shell(ctx, C.byref(args), lin, spl, d_map.p) == -1
    assert L.bfg_paint_shell(ctx, C.byref(args), tab, spl, None) == -1
    _lib.check(L.bfg_table_destroy(ctx, lin))

    # ---- baryonify: offsets + regrid
    zax, Max, rax, d = syn.displacement_table()
    dtab = _table(L, ctx, (zax, Max, rax), d, 0)
    m_in = syn.mass_map(nside)
    d_off, d_in, d_out, d_sums = Dev(L, ctx, npix * 24).zero(), Dev(L, ctx, npix * 8).up(m_in), \
        Dev(L, ctx, npix * 8).zero(), Dev(L, ctx, 16)
    _lib.check(L.bfg_baryonify_offsets(ctx, C.byref(args), dtab, spl, d_off.p), "offsets")
    _lib.check(L.bfg_regrid_shell(ctx, nside, d_off.p, d_in.p, d_out.p, d_sums.p), "regrid")
    _lib.check(L.bfg_ctx_synchronize(ctx))
    got = d_out.down(npix)
    ref = orc.baryonify_shell(nside, m_in, ra, dec, M, a, D, R, R / a, (zax, Max, rax), d, eps, 20.0)
    assert_maps_close(got, ref, 1e-5, floor=1e-9, what="C-ABI baryonify")
    sums = d_sums.down(2)
    assert np.isclose(sums[0], m_in.sum(), rtol=1e-12) and np.isclose(sums[1], sums[0], rtol=1e-10)
    amax, ssum = C.c_double(), C.c_double()
    _lib.check(L.bfg_reduce_absmax_sum(ctx, npix, d_in.p, C.byref(amax), C.byref(ssum)))
    assert amax.value == m_in.max() and np.isclose(ssum.value, m_in.sum(), rtol=1e-12)

    for b in (d_cat, d_map, d_off, d_in, d_out, d_sums):
        b.free()
    _lib.check(L.bfg_table_destroy(ctx, tab))
    _lib.check(L.bfg_table_destroy(ctx, dtab))
    _lib.check(L.bfg_spline_destroy(ctx, spl))
    _lib.check(L.bfg_ctx_destroy(ctx))


def test_cabi_context_lifecycle_frees_its_memory(cosmo):
    """create -> paint + offsets + regrid (all workspaces grow) -> destroy, six times: bfg_ctx_destroy gives back what the
    context took, results do not depend on the context's age"""
    import torch
    from scipy import interpolate
    L = _lib.load()
    nside, npix, n = 512, 12 * 512 * 512, 200000
    ra, dec, M, z = syn.catalog(n, seed=5)
    bg = Background(cosmo)
    z_t = np.linspace(0, z.max() + 0.1, 1000)
    cs = interpolate.CubicSpline(z_t, bg.angular_diameter_distance(1 / (1 + z_t)))
    knots, coef = np.ascontiguousarray(cs.x), np.ascontiguousarray(cs.c)
    zax, Max, rax, T = syn.pressure_table()
    dz, dM, dr, d = syn.displacement_table()
    m_in = syn.mass_map(nside)
    free_after, maps = [], []
    for it in range(6):
        ctx = C.c_void_p()
        _lib.check(L.bfg_ctx_create(0, C.c_void_p(-1), C.byref(ctx)), "ctx")
        spl = C.c_void_p()
        _lib.check(L.bfg_spline_create(ctx, knots.size, _dp(knots), _dp(coef), C.byref(spl)), "spline")
        d_cat = Dev(L, ctx, n * 32).up(np.stack([M, z, ra, dec], 1))
        args = _lib.ShellArgs()
        args.nside, args.n_halo, args.d_catalog, args.cat_stride, args.n_extra = nside, n, d_cat.p.value, 4, 0
        args.epsilon_max, args.runner_md, args.model_md = 10.0, _massdef(bg), _massdef(bg)
        args.model_epsilon_max, args.variant = 20.0, 0
        tab = _table(L, ctx, (zax, Max, rax), np.log(T), _lib.BFG_TABLE_LOG_VALUES)
        dtab = _table(L, ctx, (dz, dM, dr), d, 0)
        d_map = Dev(L, ctx, npix * 8).zero()
        _lib.check(L.bfg_paint_shell(ctx, C.byref(args), tab, spl, d_map.p), "paint")
        d_off, d_in, d_out, d_sums = Dev(L, ctx, npix * 24).zero(), Dev(L, ctx, npix * 8).up(m_in), \
            Dev(L, ctx, npix * 8).zero(), Dev(L, ctx, 16)
        _lib.check(L.bfg_baryonify_offsets(ctx, C.byref(args), dtab, spl, d_off.p), "offsets")
        _lib.check(L.bfg_regrid_shell(ctx, nside, d_off.p, d_in.p, d_out.p, d_sums.p), "regrid")
        _lib.check(L.bfg_ctx_synchronize(ctx))
        maps.append((d_map.down(npix), d_out.down(npix)))
        for b in (d_cat, d_map, d_off, d_in, d_out, d_sums):
            b.free()
        _lib.check(L.bfg_table_destroy(ctx, tab))
        _lib.check(L.bfg_table_destroy(ctx, dtab))
        _lib.check(L.bfg_spline_destroy(ctx, spl))
        _lib.check(L.bfg_ctx_destroy(ctx))
        free_after.append(torch.cuda.mem_get_info(0)[0])
    assert abs(free_after[-1] - free_after[0]) <= (8 << 20), free_after       # nothing accumulates from context to context
    for p, b in maps[1:]:
        assert_maps_close(p, maps[0][0], 1e-10, what="paint, fresh context")
        assert_maps_close(b, maps[0][1], 1e-9, floor=1e-9, what="baryonify, fresh context")


def test_cabi_comm_world_of_one_and_collectives():
    """the multi-GPU entry points on a one-GPU box: RCCL is dlopen'ed, a communicator of world size 1 is built from a
    unique id, the three collectives run on the context's stream and leave the buffer as the sum over one rank; a
    context without a communicator treats them as no-ops; bad arguments are refused."""
    L = _lib.load()
    ctx = C.c_void_p()
    _lib.check(L.bfg_ctx_create(0, C.c_void_p(-1), C.byref(ctx)), "ctx")
    x = np.random.default_rng(3).normal(size=4096)
    d = Dev(L, ctx, x.nbytes).up(x)
    rank, world = C.c_int(-1), C.c_int(-1)
    _lib.check(L.bfg_comm_info(ctx, C.byref(rank), C.byref(world)))
    assert (rank.value, world.value) == (0, 1)
    _lib.check(L.bfg_allreduce_f64(ctx, d.p, x.size), "allreduce without communicator")
    ident = C.create_string_buffer(_lib.BFG_COMM_ID_BYTES)
    assert L.bfg_comm_unique_id(ident, 16) == -1                                   # buffer too small
    _lib.check(L.bfg_comm_unique_id(ident, _lib.BFG_COMM_ID_BYTES), "unique id")
    assert L.bfg_comm_init(ctx, ident.raw, _lib.BFG_COMM_ID_BYTES, 1, 1) == -1       # rank outside [0, world)
    _lib.check(L.bfg_comm_init(ctx, ident.raw, _lib.BFG_COMM_ID_BYTES, 0, 1), "comm init")
    _lib.check(L.bfg_comm_info(ctx, C.byref(rank), C.byref(world)))
    assert (rank.value, world.value) == (0, 1)
    _lib.check(L.bfg_allreduce_f64(ctx, d.p, x.size), "allreduce")
    _lib.check(L.bfg_reduce_scatter_f64(ctx, d.p, x.size), "reduce-scatter")
    _lib.check(L.bfg_allgather_f64(ctx, d.p, x.size), "all-gather")
    assert L.bfg_allreduce_f64(ctx, None, 8) == -1
    # the overlapped forms: in a world of one there is nothing to exchange, the ticket is 0 and waiting is a no-op
    tk = C.c_int64(-5)
    _lib.check(L.bfg_allreduce_f64_begin(ctx, d.p, x.size, C.byref(tk)), "allreduce begin")
    assert tk.value == 0
    _lib.check(L.bfg_reduce_scatter_f64_begin(ctx, d.p, x.size, None), "reduce-scatter begin, no ticket wanted")
    _lib.check(L.bfg_comm_wait(ctx, 0), "wait for everything")
    assert L.bfg_comm_wait(ctx, 7) in (0, -1)                                      # a ticket never issued
    _lib.check(L.bfg_ctx_synchronize(ctx))
    assert np.array_equal(d.down(x.shape), x)
    _lib.check(L.bfg_comm_destroy(ctx))
    d.free()
    _lib.check(L.bfg_ctx_destroy(ctx))


def test_cabi_reduce_scatter_begin_and_allgather_world_of_one():
    """bfg_reduce_scatter_f64_begin (with a ticket) and bfg_allgather_f64 on a communicator of ONE rank: rank 0 owns [0, count), so both
    leave every element as it was; the ticket is 0 (nothing was enqueued on the communication stream) and bfg_comm_wait takes it;
    bad arguments are refused before RCCL is touched"""
    L = _lib.load()
    ctx = C.c_void_p()
    _lib.check(L.bfg_ctx_create(0, C.c_void_p(-1), C.byref(ctx)), "ctx")
    ident = C.create_string_buffer(_lib.BFG_COMM_ID_BYTES)
    _lib.check(L.bfg_comm_unique_id(ident, _lib.BFG_COMM_ID_BYTES), "unique id")
    _lib.check(L.bfg_comm_init(ctx, ident.raw, _lib.BFG_COMM_ID_BYTES, 0, 1), "comm init")
    x = np.random.default_rng(11).normal(size=3 * 12 * 16 * 16)                     # an offset field of NSIDE 16
    d = Dev(L, ctx, x.nbytes).up(x)
    for count in (x.size, 12, 1):
        tk = C.c_int64(-7)
        _lib.check(L.bfg_reduce_scatter_f64_begin(ctx, d.p, count, C.byref(tk)), "reduce-scatter begin")
        assert tk.value == 0
        _lib.check(L.bfg_comm_wait(ctx, tk.value), "wait")
        _lib.check(L.bfg_allgather_f64(ctx, d.p, count), "all-gather")
        _lib.check(L.bfg_ctx_synchronize(ctx))
        assert np.array_equal(d.down(x.shape), x)
    assert L.bfg_reduce_scatter_f64_begin(ctx, None, 8, None) == -1
    assert L.bfg_allgather_f64(ctx, None, 8) == -1
    assert L.bfg_reduce_scatter_f64_begin(ctx, d.p, -1, None) == -1
    _lib.check(L.bfg_reduce_scatter_f64_begin(ctx, d.p, 0, None), "nothing to exchange")
    _lib.check(L.bfg_comm_destroy(ctx))
    d.free()
    _lib.check(L.bfg_ctx_destroy(ctx))


def test_cabi_slice_cuts_are_what_the_sliced_calls_report(cosmo):
    """bfg_shell_slice_cuts (no GPU call) == the ranges bfg_paint_shell_sliced / bfg_baryonify_offsets_sliced hand to their callback,
    with halos and without"""
    from baryonforge_amd.engine import get_context
    ctx = get_context()
    bg = Background(cosmo)
    spline = ctx.da_spline(bg, 0.6)
    md = ctx.massdef_struct(bg, None)
    zax, Max, rax, T = syn.pressure_table()
    zd, Md, rd, dd = syn.displacement_table()
    with np.errstate(all="ignore"):
        tp = ctx.table([zax, Max, rax], np.log(T), log_values=True)
    td = ctx.table([zd, Md, rd], dd, log_values=False)
    for nside in (8, 64, 256):
        npix = 12 * nside * nside
        for n in (0, 300):
            ra, dec, M, z = syn.catalog(max(n, 1), seed=21)
            d_cat = ctx.to_device(np.stack([M, z, ra, dec], axis=1)[:n])
            for slices in (1, 2, 5, 16, 40):
                for offsets in (False, True):
                    want = _lib.shell_slice_cuts(nside, offsets, slices)
                    assert want[0] == 0 and want[-1] == (3 if offsets else 1) * npix and all(a < b for a, b in zip(want, want[1:]))
                    seen = []
                    cb = lambda k, m, lo, hi: seen.append((k, m, lo, hi))
                    if offsets:
                        args = ctx.shell_args(nside, d_cat, n, 4, 0, 10.0, md, model_md=md, model_epsilon_max=20.0, out_overwrite=True)
                        ctx.baryonify_offsets(args, td, spline, ctx.empty(npix, 3), slices=slices, on_slice=cb)
                    else:
                        args = ctx.shell_args(nside, d_cat, n, 4, 0, 10.0, md, out_overwrite=True)
                        ctx.paint_shell(args, tp, spline, ctx.empty(npix), slices=slices, on_slice=cb)
                    assert [(lo, hi) for _, _, lo, hi in seen] == list(zip(want, want[1:])), (nside, n, slices, offsets)
                    assert [k for k, _, _, _ in seen] == list(range(len(want) - 1)) and all(m == len(want) - 1 for _, m, _, _ in seen)


def test_cabi_set_stream_orders_work_across_streams(cosmo):
    """bfg_ctx_set_stream: a context created on its own stream is moved to a second stream; the paint enqueued there
    sees the zero-fill enqueued on the first one (event ordering, no host synchronisation in between)"""
    import torch
    from scipy import interpolate
    L = _lib.load()
    ctx = C.c_void_p()
    _lib.check(L.bfg_ctx_create(0, C.c_void_p(-1), C.byref(ctx)), "ctx")
    nside, npix, n = 256, 12 * 256 * 256, 5000
    ra, dec, M, z = syn.catalog(n, seed=8)
    bg = Background(cosmo)
    z_t = np.linspace(0, z.max() + 0.1, 1000)
    cs = interpolate.CubicSpline(z_t, bg.angular_diameter_distance(1 / (1 + z_t)))
    knots, coef = np.ascontiguousarray(cs.x), np.ascontiguousarray(cs.c)
    spl = C.c_void_p()
    _lib.check(L.bfg_spline_create(ctx, knots.size, _dp(knots), _dp(coef), C.byref(spl)), "spline")
    zax, Max, rax, T = syn.pressure_table()
    tab = _table(L, ctx, (zax, Max, rax), np.log(T), _lib.BFG_TABLE_LOG_VALUES)
    d_cat = Dev(L, ctx, n * 32).up(np.stack([M, z, ra, dec], 1))
    args = _lib.ShellArgs()
    args.nside, args.n_halo, args.d_catalog, args.cat_stride, args.n_extra = nside, n, d_cat.p.value, 4, 0
    args.epsilon_max, args.runner_md, args.model_md, args.variant = 10.0, _massdef(bg), _massdef(bg), 0
    d_map = Dev(L, ctx, npix * 8).up(np.full(npix, 5.0))
    d_map.zero()                                                    # async, on the context's first stream
    s2 = torch.cuda.Stream()
    assert L.bfg_ctx_set_stream(ctx, C.c_void_p(-1)) == -1          # BFG_STREAM_OWN is not a stream to move to
    _lib.check(L.bfg_ctx_set_stream(ctx, C.c_void_p(s2.cuda_stream)), "set_stream")
    _lib.check(L.bfg_paint_shell(ctx, C.byref(args), tab, spl, d_map.p), "paint")
    _lib.check(L.bfg_ctx_synchronize(ctx))
    a, R, D = orc.halo_scalars(cosmo, M, z)
    ref, _ = orc.paint_shell(nside, ra, dec, M, a, D, R, (zax, Max, rax), np.log(T), 10.0)
    assert_maps_close(d_map.down(npix), ref, 1e-5, what="paint after a stream switch")
    d_cat.free(); d_map.free()
    _lib.check(L.bfg_table_destroy(ctx, tab))
    _lib.check(L.bfg_spline_destroy(ctx, spl))
    _lib.check(L.bfg_ctx_destroy(ctx))


def test_cabi_sliced_paint_through_ctypes(cosmo):
    """bfg_paint_shell_sliced driven from plain ctypes: the callback sees n ascending slices that cover the map; copying
    each slice out (bfg_memcpy_d2h is synchronous on the context's stream) the moment it is reported gives the plain map"""
    from scipy import interpolate
    L = _lib.load()
    ctx = C.c_void_p()
    _lib.check(L.bfg_ctx_create(0, C.c_void_p(-1), C.byref(ctx)), "ctx")
    nside, npix, n = 128, 12 * 128 * 128, 3000
    ra, dec, M, z = syn.catalog(n, seed=18)
    bg = Background(cosmo)
    z_t = np.linspace(0, z.max() + 0.1, 1000)
    cs = interpolate.CubicSpline(z_t, bg.angular_diameter_distance(1 / (1 + z_t)))
    knots, coef = np.ascontiguousarray(cs.x), np.ascontiguousarray(cs.c)
    spl = C.c_void_p()
    _lib.check(L.bfg_spline_create(ctx, knots.size, _dp(knots), _dp(coef), C.byref(spl)), "spline")
    zax, Max, rax, T = syn.pressure_table()
    tab = _table(L, ctx, (zax, Max, rax), np.log(T), _lib.BFG_TABLE_LOG_VALUES)
    d_cat = Dev(L, ctx, n * 32).up(np.stack([M, z, ra, dec], 1))
    args = _lib.ShellArgs()
    args.nside, args.n_halo, args.d_catalog, args.cat_stride, args.n_extra = nside, n, d_cat.p.value, 4, 0
    args.epsilon_max, args.runner_md, args.model_md, args.variant = 10.0, _massdef(bg), _massdef(bg), 0
    args.flags = _lib.SHELL_OUT_OVERWRITE
    d_map = Dev(L, ctx, npix * 8).up(np.full(npix, np.nan))
    out = np.full(npix, np.nan)
    seen = []

    def cb(user, k, nsl, lo, hi):
        seen.append((k, nsl, lo, hi))
        L.bfg_memcpy_d2h(ctx, out[lo:hi].ctypes.data_as(C.c_void_p), C.c_void_p(d_map.p.value + 8 * lo), 8 * (hi - lo))
        return 0
    fn = _lib.SLICE_FN(cb)
    _lib.check(L.bfg_paint_shell_sliced(ctx, C.byref(args), tab, spl, d_map.p, 4, fn, None), "sliced paint")
    assert L.bfg_paint_shell_sliced(ctx, C.byref(args), tab, spl, d_map.p, 0, fn, None) == -1     # n_slices < 1
    _lib.check(L.bfg_ctx_synchronize(ctx))
    assert [s[0] for s in seen] == [0, 1, 2, 3] and seen[0][2] == 0 and seen[-1][3] == npix
    a, R, D = orc.halo_scalars(cosmo, M, z)
    ref, _ = orc.paint_shell(nside, ra, dec, M, a, D, R, (zax, Max, rax), np.log(T), 10.0)
    assert_maps_close(out, ref, 1e-5, what="sliced paint through ctypes")
    assert np.array_equal(d_map.down(npix), out)
    fail = _lib.SLICE_FN(lambda user, k, nsl, lo, hi: 1)
    assert L.bfg_paint_shell_sliced(ctx, C.byref(args), tab, spl, d_map.p, 4, fail, None) == -1  # the callback's veto
    d_cat.free(); d_map.free()
    _lib.check(L.bfg_table_destroy(ctx, tab))
    _lib.check(L.bfg_spline_destroy(ctx, spl))
    _lib.check(L.bfg_ctx_destroy(ctx))


def test_cabi_disc_enumeration_and_value_scatter(cosmo):
    """the four entry points around a host-evaluated model (bfg_disc_enumerate_count / _enumerate / bfg_map_add_values /
    bfg_offsets_add_displacements) through ctypes with plain pointers: the pixel lists equal hp.query_disc + the < 4 pixel rule as
    the oracle restates them, the distances equal |vec D - vec_j D| / a_j, and the two scatter kernels reproduce the oracle's
    loops for a closed-form model"""
    from scipy import interpolate
    L = _lib.load()
    ctx = C.c_void_p()
    _lib.check(L.bfg_ctx_create(0, C.c_void_p(-1), C.byref(ctx)), "ctx")
    nside, npix, eps, n = 64, 12 * 64 * 64, 6.0, 300
    ra, dec, M, z = syn.catalog(n, seed=33, z=(0.1, 0.4), logM=(12.0, 15.3))
    ra[0], dec[0] = 123.0, -89.95                                          # a disc over the south pole
    bg = Background(cosmo)
    z_t = np.linspace(0, z.max() + 0.1, 1000)
    cs = interpolate.CubicSpline(z_t, bg.angular_diameter_distance(1 / (1 + z_t)))
    spl = C.c_void_p()
    knots, coef = np.ascontiguousarray(cs.x), np.ascontiguousarray(cs.c)
    _lib.check(L.bfg_spline_create(ctx, knots.size, _dp(knots), _dp(coef), C.byref(spl)), "spline")
    d_cat = Dev(L, ctx, n * 32).up(np.stack([M, z, ra, dec], 1))
    args = _lib.ShellArgs()
    args.nside, args.n_halo, args.d_catalog, args.cat_stride, args.n_extra = nside, n, d_cat.p.value, 4, 0
    args.epsilon_max, args.runner_md, args.model_md = eps, _massdef(bg), _massdef(bg)
    a, R, D = orc.halo_scalars(cosmo, M, z)

    for fallback4 in (0, 1):
        d_counts = Dev(L, ctx, n * 8)
        _lib.check(L.bfg_disc_enumerate_count(ctx, C.byref(args), spl, fallback4, d_counts.p), "count")
        counts = np.empty(n, dtype=np.int64)
        _lib.check(L.bfg_memcpy_d2h(ctx, counts.ctypes.data, d_counts.p, counts.nbytes))
        base = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.int64)
        total = int(counts.sum())
        d_base = Dev(L, ctx, n * 8)
        _lib.check(L.bfg_memcpy_h2d(ctx, d_base.p, base.ctypes.data, base.nbytes))
        d_pix, d_r, d_halo = Dev(L, ctx, total * 8), Dev(L, ctx, total * 8), Dev(L, ctx, total * 4)
        _lib.check(L.bfg_disc_enumerate(ctx, C.byref(args), spl, fallback4, d_base.p, d_pix.p, d_r.p, d_halo.p), "enumerate")
        pix, halo = np.empty(total, dtype=np.int64), np.empty(total, dtype=np.int32)
        _lib.check(L.bfg_memcpy_d2h(ctx, pix.ctypes.data, d_pix.p, pix.nbytes))
        _lib.check(L.bfg_memcpy_d2h(ctx, halo.ctypes.data, d_halo.p, halo.nbytes))
        r_com = d_r.down(total)
        for j in range(n):
            vec_j = orc.ang2vec(ra[j], dec[j], lonlat=True)
            ref_pix = orc.query_disc(nside, vec_j, R[j] * eps / D[j])
            if fallback4 and ref_pix.size < 4:
                ref_pix = orc.get_interp_weights(nside, ra[j], dec[j], lonlat=True)[0]
            sl = slice(base[j], base[j] + counts[j])
            assert counts[j] == ref_pix.size, (j, fallback4)
            assert np.array_equal(np.sort(pix[sl]), np.sort(ref_pix)) and np.all(halo[sl] == j)
            vec = np.stack(orc.pix2vec(nside, pix[sl]), axis=1)
            np.testing.assert_allclose(r_com[sl], np.sqrt(np.sum((vec * D[j] - vec_j * D[j]) ** 2, axis=1)) / a[j], rtol=1e-11)
        if not fallback4:
            # paint: values = a closed form of (r, M)
            vals = 1e-3 * (M[halo] / 1e14) / (1 + r_com ** 2)
            d_val, d_map = Dev(L, ctx, total * 8).up(vals), Dev(L, ctx, npix * 8).zero()
            _lib.check(L.bfg_map_add_values(ctx, d_map.p, d_pix.p, d_val.p, total), "add values")
            ref = np.zeros(npix)
            np.add.at(ref, pix, vals)
            assert_maps_close(d_map.down(npix), ref, 1e-12, what="bfg_map_add_values")
            d_val.free(); d_map.free()
        else:
            disp = 0.05 * (M[halo] / 1e14) ** (1 / 3) * r_com * np.exp(-r_com)
            d_disp, d_off = Dev(L, ctx, total * 8).up(disp), Dev(L, ctx, npix * 24).zero()
            _lib.check(L.bfg_offsets_add_displacements(ctx, C.byref(args), spl, d_pix.p, d_halo.p, d_disp.p, total, d_off.p), "disp")
            ref, _ = orc.baryonify_offsets_callable(cosmo, nside, ra, dec, M, z, eps,
                                                    lambda r, Mj, aj: 0.05 * (Mj / 1e14) ** (1 / 3) * r * np.exp(-r))
            np.testing.assert_allclose(d_off.down((npix, 3)), ref, rtol=1e-7, atol=1e-15 + 1e-9 * np.abs(ref).max())
            d_disp.free(); d_off.free()
        for b in (d_counts, d_base, d_pix, d_r, d_halo):
            b.free()
    # bad arguments are refused
    assert L.bfg_disc_enumerate_count(ctx, C.byref(args), spl, 0, None) == -1
    assert L.bfg_map_add_values(ctx, None, None, None, 5) == -1
    d_cat.free()
    _lib.check(L.bfg_spline_destroy(ctx, spl))
    _lib.check(L.bfg_ctx_destroy(ctx))


def test_cabi_ndtable_rows_and_read_equal_scipy_rgi():
    """bfg_ndtable_create / _rows / _read through plain ctypes pointers: a 7-dimensional table (z, M, r + four parameter axes) read
    out at (halo, r) entries == scipy's RegularGridInterpolator (the reference's read-out, utils/Tabulate.py:585-590) to 1e-13,
    NaN for halos outside an axis and for radii outside the radial axis; the exp / scale / cut variants"""
    from scipy.interpolate import RegularGridInterpolator
    L = _lib.load()
    ctx = C.c_void_p()
    _lib.check(L.bfg_ctx_create(0, C.c_void_p(-1), C.byref(ctx)), "ctx")
    rng = np.random.default_rng(12)
    axes = [np.log(1 + np.array([0.0, 0.3, 0.7, 1.2])), np.log(np.geomspace(1e12, 1e16, 6)), np.log(np.geomspace(1e-2, 50, 40)),
            np.array([0.5, 1.0, 2.0]), np.array([-1.0, 1.0]), np.array([3.0, 4.0, 6.0]), np.array([0.0, 0.1, 0.25, 1.0])]
    vals = rng.normal(size=[a.size for a in axes])
    vals[1, 2, 5, 0, 0, 1, 2] = np.nan
    outer = [axes[0], axes[1]] + axes[3:]
    rlast = np.ascontiguousarray(np.moveaxis(vals, 2, -1))
    shape = (C.c_int64 * len(outer))(*[a.size for a in outer])
    ptrs = (C.POINTER(C.c_double) * len(outer))(*[_lib.dptr(a) for a in outer])
    t = C.c_void_p()
    assert L.bfg_ndtable_create(ctx, 13, shape, ptrs, 40, _lib.dptr(axes[2]), _lib.dptr(rlast), C.byref(t)) == -4      # unsupported
    _lib.check(L.bfg_ndtable_create(ctx, len(outer), shape, ptrs, 40, _lib.dptr(axes[2]), _lib.dptr(rlast), C.byref(t)), "ndtable")
    n = 500
    M = 10 ** rng.uniform(11.8, 16.2, n)
    z = rng.uniform(-0.05, 1.3, n)
    ex = np.stack([rng.uniform(0.4, 2.1, n), rng.uniform(-1.1, 1.1, n), rng.uniform(3, 6, n), rng.uniform(0, 1, n)], 1)
    cat = np.ascontiguousarray(np.concatenate([np.stack([M, z, rng.uniform(0, 360, n), rng.uniform(-90, 90, n)], 1), ex], 1))
    d_cat = Dev(L, ctx, cat.nbytes).up(cat)
    d_rows = Dev(L, ctx, n * 40 * 8)
    assert L.bfg_ndtable_rows(ctx, t, d_cat.p, n, 7, d_rows.p) == -1                 # stride too small for the table's axes
    _lib.check(L.bfg_ndtable_rows(ctx, t, d_cat.p, n, 8, d_rows.p), "rows")
    ne = 4000
    halo = rng.integers(0, n, ne).astype(np.int32)
    r = np.exp(rng.uniform(np.log(5e-3), np.log(80), ne))
    shift = rng.uniform(-0.3, 0.3, n)
    rcut = np.exp(rng.uniform(0, 3, n))
    scale = rng.uniform(1, 2, n)
    d_halo, d_r = Dev(L, ctx, halo.nbytes).up(halo), Dev(L, ctx, r.nbytes).up(r)
    d_shift, d_rcut, d_scale = Dev(L, ctx, shift.nbytes).up(shift), Dev(L, ctx, rcut.nbytes).up(rcut), Dev(L, ctx, scale.nbytes).up(scale)
    d_out = Dev(L, ctx, ne * 8)
    oob = np.zeros(1, dtype=np.uint32)
    d_oob = Dev(L, ctx, 4).up(oob)
    rgi = RegularGridInterpolator(tuple(axes), vals, method="linear", bounds_error=False, fill_value=np.nan)

    def ref(shifted):
        lnr = np.log(r) - (shift[halo] if shifted else 0.0)
        pts = np.concatenate([np.log(1.0 / (1.0 / (1.0 + z[halo])))[:, None], np.log(M[halo])[:, None], lnr[:, None], ex[halo]], 1)
        return rgi(pts)
    _lib.check(L.bfg_ndtable_read(ctx, t, d_rows.p, ne, d_halo.p, d_r.p, None, None, None, 0, d_out.p, d_oob.p), "read")
    _lib.check(L.bfg_ctx_synchronize(ctx))
    got, want = d_out.down((ne,)), ref(False)
    assert np.array_equal(np.isnan(got), np.isnan(want)) and 0.2 * ne < np.isnan(want).sum() < 0.8 * ne
    np.testing.assert_allclose(got[~np.isnan(got)], want[~np.isnan(want)], rtol=1e-13, atol=1e-14)
    lnr = np.log(r)
    assert d_oob.down((1,), np.uint32)[0] == np.count_nonzero((lnr < axes[2][0]) | (lnr > axes[2][-1]))
    # displacement flavour: shifted radial coordinate, zero beyond the cut (NaN inside it stays NaN)
    _lib.check(L.bfg_ndtable_read(ctx, t, d_rows.p, ne, d_halo.p, d_r.p, d_shift.p, d_rcut.p, None, 0, d_out.p, None), "read (cut)")
    _lib.check(L.bfg_ctx_synchronize(ctx))
    got, want = d_out.down((ne,)), np.where(r < rcut[halo], ref(True), 0.0)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    np.testing.assert_allclose(got[~np.isnan(got)], want[~np.isnan(want)], rtol=1e-13, atol=1e-14)
    # paint flavour: exp, non-finite -> 0, times the halo's scale
    _lib.check(L.bfg_ndtable_read(ctx, t, d_rows.p, ne, d_halo.p, d_r.p, None, None, d_scale.p, 1, d_out.p, None), "read (exp)")
    _lib.check(L.bfg_ctx_synchronize(ctx))
    e = np.exp(ref(False))
    np.testing.assert_allclose(d_out.down((ne,)), np.where(np.isfinite(e), e, 0.0) * scale[halo], rtol=1e-13, atol=0)
    assert L.bfg_ndtable_read(ctx, t, None, ne, d_halo.p, d_r.p, None, None, None, 0, d_out.p, None) == -1
    for d in (d_cat, d_rows, d_halo, d_r, d_shift, d_rcut, d_scale, d_out, d_oob):
        d.free()
    _lib.check(L.bfg_ndtable_destroy(ctx, t))
    _lib.check(L.bfg_ctx_destroy(ctx))


def test_device_distances_reproduce_what_live_pyccl_printed_in_the_reference_notebooks():
    """a9 on the DEVICE against the real libccl: two halos at the notebooks' min_z / max_z, their discs enumerated through the C-ABI;
    every entry's r_com / |vec_pix - vec_j| is the comoving distance the device used for that halo (D_A spline on the device,
    HealpixRunner.py:297-299, :464-469), and chi(max_z) - chi(min_z) reproduces the 17-digit number pyccl printed in the reference's
    example notebooks (tests/golden/pyccl_notebook_outputs.json) to 3e-7"""
    import json
    import os
    from scipy import interpolate
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pyccl_notebook_outputs.json")) as f:
        cases = json.load(f)["cases"]
    L = _lib.load()
    ctx = C.c_void_p()
    _lib.check(L.bfg_ctx_create(0, C.c_void_p(-1), C.byref(ctx)), "ctx")
    for case in cases:
        c = case["cosmology"]
        cosmo = {"Omega_m": c["Omega_c"] + c["Omega_b"], "Omega_b": c["Omega_b"], "h": c["h"], "sigma8": c["sigma8"], "n_s": c["n_s"], "w0": -1.0}
        bg = Background(cosmo)
        nside, n = 256, 2
        z = np.array([case["min_z"], case["max_z"]])
        M, ra, dec = np.array([4e14, 4e14]), np.array([40.0, 200.0]), np.array([10.0, -35.0])
        z_t = np.linspace(0, z.max() + 0.1, 1000)                               # the runners' spline (HealpixRunner.py:297-299)
        cs = interpolate.CubicSpline(z_t, bg.angular_diameter_distance(1 / (1 + z_t)))
        spl = C.c_void_p()
        knots, coef = np.ascontiguousarray(cs.x), np.ascontiguousarray(cs.c)
        _lib.check(L.bfg_spline_create(ctx, knots.size, _dp(knots), _dp(coef), C.byref(spl)), "spline")
        d_cat = Dev(L, ctx, n * 32).up(np.stack([M, z, ra, dec], 1))
        args = _lib.ShellArgs()
        args.nside, args.n_halo, args.d_catalog, args.cat_stride, args.n_extra = nside, n, d_cat.p.value, 4, 0
        args.epsilon_max, args.runner_md, args.model_md = 8.0, _massdef(bg), _massdef(bg)
        d_counts = Dev(L, ctx, n * 8)
        _lib.check(L.bfg_disc_enumerate_count(ctx, C.byref(args), spl, 0, d_counts.p), "count")
        counts = d_counts.down(n, np.int64)
        assert counts.min() > 20
        base = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.int64)
        total = int(counts.sum())
        d_base = Dev(L, ctx, n * 8).up(base)
        d_pix, d_r, d_halo = Dev(L, ctx, total * 8), Dev(L, ctx, total * 8), Dev(L, ctx, total * 4)
        _lib.check(L.bfg_disc_enumerate(ctx, C.byref(args), spl, 0, d_base.p, d_pix.p, d_r.p, d_halo.p), "enumerate")
        pix, r_com = d_pix.down(total, np.int64), d_r.down(total)
        chi = np.empty(n)
        for j in range(n):
            sl = slice(base[j], base[j] + counts[j])
            vec = np.stack(orc.pix2vec(nside, pix[sl]), axis=1)
            chord = np.sqrt(np.sum((vec - orc.ang2vec(ra[j], dec[j], lonlat=True)) ** 2, axis=1))
            ratio = r_com[sl] / chord
            assert np.ptp(ratio) < 1e-9 * ratio.mean()                          # one distance per halo
            chi[j] = ratio.mean()
        assert abs((chi[1] - chi[0]) / case["shell_thickness_mpc"] - 1) < 3e-7, (case["source"], chi)
        for b in (d_cat, d_counts, d_base, d_pix, d_r, d_halo):
            b.free()
        _lib.check(L.bfg_spline_destroy(ctx, spl))
    _lib.check(L.bfg_ctx_destroy(ctx))


def test_cabi_reuse_plan_flag_against_the_oracle(cosmo):
    """BFG_SHELL_REUSE_PLAN at the boundary itself (ABI 6): three tables on one grid painted over one catalog -- the first call plans,
    the two that carry the flag run the tile kernels only (bfg_plan_reuses) -- each map and each counter against the oracle's run of
    THAT table; then a table on another grid with the flag set: ignored, full call, still the oracle's map.  Offsets likewise."""
    from scipy import interpolate
    L = _lib.load()
    ctx = C.c_void_p()
    _lib.check(L.bfg_ctx_create(0, C.c_void_p(-1), C.byref(ctx)), "ctx")
    nside, npix, eps, n = 256, 12 * 256 * 256, 10.0, 6000
    ra, dec, M, z = syn.catalog(n, seed=33, z=(0.1, 0.4))
    M[3] = 7e16                                                     # outside the hull: counted by the planning call AND by the reusing ones
    bg = Background(cosmo)
    z_t = np.linspace(0, z.max() + 0.1, 1000)
    cs = interpolate.CubicSpline(z_t, bg.angular_diameter_distance(1 / (1 + z_t)))
    spl = C.c_void_p()
    knots, coef = np.ascontiguousarray(cs.x), np.ascontiguousarray(cs.c)
    _lib.check(L.bfg_spline_create(ctx, knots.size, _dp(knots), _dp(coef), C.byref(spl)), "spline")
    d_cat = Dev(L, ctx, n * 32).up(np.stack([M, z, ra, dec], 1))
    a, R, D = orc.halo_scalars(cosmo, M, z)
    args = _lib.ShellArgs()
    args.nside, args.n_halo, args.d_catalog, args.cat_stride, args.n_extra = nside, n, d_cat.p.value, 4, 0
    args.epsilon_max, args.runner_md, args.model_md = eps, _massdef(bg), _massdef(bg)
    args.model_epsilon_max, args.variant = 20.0, 0

    def reuses():
        k = C.c_int64()
        _lib.check(L.bfg_plan_reuses(ctx, C.byref(k)))
        return k.value
    zax, Max, rax, T = syn.pressure_table()
    d_map = Dev(L, ctx, npix * 8)
    st = _lib.Stats()
    with np.errstate(all="ignore"):
        for k, scale in enumerate((1.0, 2.5, 0.3)):
            Tk = T * scale * (1.0 + 0.2 * k * np.tanh(np.exp(rax)))[None, None, :]
            tab = _table(L, ctx, (zax, Max, rax), np.log(Tk), _lib.BFG_TABLE_LOG_VALUES)
            args.flags = _lib.SHELL_OUT_OVERWRITE | (_lib.SHELL_REUSE_PLAN if k else 0)
            r0 = reuses()
            _lib.check(L.bfg_stats_reset(ctx))
            _lib.check(L.bfg_paint_shell(ctx, C.byref(args), tab, spl, d_map.p), "paint")
            _lib.check(L.bfg_stats_read(ctx, C.byref(st)))
            assert reuses() - r0 == (1 if k else 0)
            ref, ptot = orc.paint_shell(nside, ra, dec, M, a, D, R, (zax, Max, rax), np.log(Tk), eps)
            assert st.pixel_updates == ptot and st.halos_out_of_table == 1 and (st.warn_mask & 2)
            got = d_map.down(npix)
            assert np.array_equal(got != 0, ref != 0)
            assert_maps_close(got, ref, 1e-5, what=f"C-ABI paint on a reused plan, table {k}")
            _lib.check(L.bfg_table_destroy(ctx, tab))
        # another grid (nine redshift nodes), flag set: the library notices and does the whole call
        z9, M9, r9, T9 = syn.pressure_table(9, 30, 100)
        tab = _table(L, ctx, (z9, M9, r9), np.log(T9), _lib.BFG_TABLE_LOG_VALUES)
        r0 = reuses()
        _lib.check(L.bfg_paint_shell(ctx, C.byref(args), tab, spl, d_map.p), "paint")
        assert reuses() == r0
        ref, _ = orc.paint_shell(nside, ra, dec, M, a, D, R, (z9, M9, r9), np.log(T9), eps)
        assert_maps_close(d_map.down(npix), ref, 1e-5, what="C-ABI paint, flag ignored for another grid")
        _lib.check(L.bfg_table_destroy(ctx, tab))
    # offsets: the pre-blended row windows are rebuilt from the table the call is given
    zd, Md, rd, d = syn.displacement_table()
    d_off = Dev(L, ctx, npix * 24)
    for k, scale in enumerate((1.0, 1.7)):
        dtab = _table(L, ctx, (zd, Md, rd), d * scale, 0)
        args.flags = _lib.SHELL_OUT_OVERWRITE | (_lib.SHELL_REUSE_PLAN if k else 0)
        r0 = reuses()
        _lib.check(L.bfg_baryonify_offsets(ctx, C.byref(args), dtab, spl, d_off.p), "offsets")
        assert reuses() - r0 == (1 if k else 0)
        ref, _ = orc.baryonify_offsets(nside, ra, dec, M, a, D, R, R / a, (zd, Md, rd), d * scale, eps, 20.0)
        got = d_off.down((npix, 3))
        assert np.max(np.abs(got - ref)) <= 1e-5 * np.max(np.abs(ref)) * 1e-3 + 1e-15, f"offsets on a reused plan, table {k}"
        _lib.check(L.bfg_table_destroy(ctx, dtab))
    for dv in (d_cat, d_map, d_off):
        dv.free()
    _lib.check(L.bfg_spline_destroy(ctx, spl))
    _lib.check(L.bfg_ctx_destroy(ctx))
