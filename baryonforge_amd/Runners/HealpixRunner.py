"""
Full-sky HEALPix shell runners, mirroring BaryonForge/Runners/HealpixRunner.py:
`DefaultRunner` (:78-232), `BaryonifyShell` (:235-373), `PaintProfilesShell`
(:376-483) and `regrid_pixels_hpix` (:17-71).

Same constructor, attributes, error behaviour and `process() -> float64[Npix]`;
the serial per-halo python loop of the reference is replaced by the fused HIP
kernels of libbfg_mi355.so (one launch for all halos):

    catalog records -> HBM -> halo_prep_kernel (a, R_delta, D_A spline, disc ring
    range, table cell) -> shell kernel (query_disc ring windows, LDS-staged
    profile row, trilinear read-out, f64 atomic scatter-add) [-> regrid_kernel]

There is no CPU fallback for the loops.  The fast path is the tabulated one
(TabulatedProfile, ParamTabulatedProfile, BaryonificationClass -- ours or objects
from the real BaryonForge exposing the same raw_input_* attributes).  A model that
is NOT tabulated -- any object with .projected(cosmo, r, M, a) / .displacement(r, M, a),
which the reference calls once per halo (:472, :345) -- is still a Python callable and
is evaluated on the host, per halo, exactly as the reference does; the GPU does
everything around it (disc enumeration and distances, the scatter-add, the offset
geometry, the regrid): `_callable_batches`.
"""
import numpy as np

from ..background import Background, MassDef
from ..engine import emit_fallback_warning, emit_range_warnings, get_context
from ..utils.Tabulate import ParamTabulatedProfile, _get_parameter
from ..Profiles.BaryonCorrection import BaryonificationClass

__all__ = ["DefaultRunner", "BaryonifyShell", "PaintProfilesShell", "PaintProfilesAnisShell", "regrid_pixels_hpix"]


def regrid_pixels_hpix(hmap, parent_pix_vals, child_pix, child_weights):
    """hmap[child_pix[i, j]] += child_weights[i, j] * parent_pix_vals[i]   (HealpixRunner.py:17-71)

    Host utility kept for API parity (vectorised, unbuffered so repeated indices accumulate);
    the shell runner itself regrids on the GPU (bfg_regrid_shell)."""
    np.add.at(hmap, np.asarray(child_pix).ravel(),
              (np.asarray(child_weights) * np.asarray(parent_pix_vals)[:, None]).ravel())
    return hmap


def _is_paint_table(model):
    return all(hasattr(model, k) for k in ("raw_input_2D", "raw_input_z_range", "raw_input_M_range",
                                           "raw_input_r_range"))


def _is_disp_table(model):
    return all(hasattr(model, k) for k in ("raw_input_d", "raw_input_z_range", "raw_input_M_range",
                                           "raw_input_r_range"))


def _table_axes(model, keys):
    return [np.asarray(model.raw_input_z_range, dtype=np.float64), np.asarray(model.raw_input_M_range, dtype=np.float64),
            np.asarray(model.raw_input_r_range, dtype=np.float64)] + \
           [np.asarray(getattr(model, "raw_input_%s_range" % k), dtype=np.float64) for k in keys]


class DefaultRunner(object):
    """
    Base runner (HealpixRunner.py:78-232): stores the catalog, the shell, the cut-out
    size `epsilon_max` (in halo radii), the `model`, and the mass definition.

    Extra keyword (not in the reference): `variant` selects the kernel variant
    ('auto', 'scatter_wave', 'scatter_quarter', 'tile_lds').
    """

    def __init__(self, HaloLightConeCatalog, LightconeShell, epsilon_max, model, use_ellipticity=False,
                 mass_def=None, include_pixel_size=False, verbose=True, variant="auto"):
        self.HaloLightConeCatalog = HaloLightConeCatalog
        self.LightconeShell = LightconeShell
        self.cosmo = HaloLightConeCatalog.cosmology
        self.model = model
        self.epsilon_max = epsilon_max
        self.mass_def = MassDef(200, "critical") if mass_def is None else mass_def
        self.verbose = verbose
        self.use_ellipticity = use_ellipticity
        self.include_pixel_size = include_pixel_size
        self.variant = variant
        self.last_stats = None
        if use_ellipticity:
            raise NotImplementedError("You have set use_ellipticity = True, but this not yet implemented for HealpixRunner")

    def build_Rmat(self, A, ref):
        """2x2 rotation aligning A with ref (HealpixRunner.py:180-209)"""
        A = A / np.linalg.norm(A)
        ref = ref / np.linalg.norm(ref)
        ang = np.arccos(np.dot(A, ref))
        return np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]])

    def coord_array(self, *args):
        """HealpixRunner.py:212-232"""
        return np.vstack([a.flatten() for a in args]).T

    # ---- shared set-up of one process() call ----------------------------------------------
    def _keys_checked(self):
        keys = vars(self.model).get("p_keys", []) if self.model is not None else []
        if len(keys) > 0:                                               # HealpixRunner.py:304-311 / :436-443
            txt = (f"You asked to use {keys} properties in Baryonification. You must pass a ParamTabulatedProfile "
                   f"pr BaryonificationClass as the model. You have passed {type(self.model)} instead. "
                   f"If you did pass in a BaryonificationClass make sure you passed in addition params using "
                   f"the other_params option.")
            ok = isinstance(self.model, (ParamTabulatedProfile, BaryonificationClass)) or \
                type(self.model).__name__ in ("ParamTabulatedProfile", "BaryonificationClass", "Baryonification2D",
                                              "Baryonification3D")
            assert ok, txt
        return list(keys)

    def _device_inputs(self, ctx, keys):
        cat = self.HaloLightConeCatalog.cat
        if hasattr(self.HaloLightConeCatalog, "z_max"):
            z_m = self.HaloLightConeCatalog.z_max()
        else:
            z_m = np.max(cat["z"]) if cat.size else 0.0
        assert z_m <= 30, f"We assume max(z) = 30, but your catalog has max(z) = {z_m}"   # :301 / :433
        z_m = max(z_m, getattr(self, "_spline_z_max", z_m))              # a shard of a split catalog: the whole catalog's max(z)
        bg = Background(self.cosmo)
        spline = ctx.da_spline(bg, z_m)                                  # :297-299 / :429-431
        if hasattr(self.HaloLightConeCatalog, "device_records"):         # uploaded once per catalog object (utils/io.py)
            d_cat, stride = self.HaloLightConeCatalog.device_records(ctx, keys)
        else:                                                            # a catalog object of the real BaryonForge
            recs = np.stack([np.asarray(cat[c], dtype=np.float64) for c in ["M", "z", "ra", "dec"] + keys], axis=1)
            d_cat, stride = ctx.to_device(recs), recs.shape[1]
        return bg, spline, d_cat, stride


def _callable_batches(runner, ctx, fallback4, keys=()):
    """The geometry of the reference loops for a model that is a Python callable (csrc/bfg_enum.hpp): yields, batch of halos by
    batch, (j0, args, counts, base, pix, r_com, halo, D_j) -- host arrays counts / base / r_com / D_j and device tensors pix / halo
    -- where entry e of halo j0 + j (base[j] <= e < base[j] + counts[j]) is pixel pix[e] of its disc (:463 / :330-334) at
    r_sep / a_j = r_com[e] (:464-469).  A batch holds at most BFG_CALLABLE_BATCH entries (default 2^24: 320 MB of lists)."""
    import os
    from scipy import interpolate
    NSIDE = runner.LightconeShell.NSIDE
    bg, spline, d_cat, stride = runner._device_inputs(ctx, list(keys))
    cat = runner.HaloLightConeCatalog.cat
    n = int(d_cat.shape[0])
    md = ctx.massdef_struct(bg, runner.mass_def)
    z_m = max(float(np.max(cat["z"])) if n else 0.0, getattr(runner, "_spline_z_max", 0.0))
    z_t = np.linspace(0, z_m + 0.1, 1000)
    D_a = interpolate.CubicSpline(z_t, bg.angular_diameter_distance(1 / (1 + z_t)))     # :297-299 / :429-431 (the host copy)
    if n == 0:
        return
    counts_all = ctx.disc_count(ctx.shell_args(NSIDE, d_cat, n, stride, 0, runner.epsilon_max, md), spline, fallback4).cpu().numpy()
    cap = int(os.environ.get("BFG_CALLABLE_BATCH", str(1 << 24)))
    j0 = 0
    while j0 < n:
        csum = np.cumsum(counts_all[j0:])
        j1 = j0 + max(1, int(np.searchsorted(csum, cap, side="right")))
        args = ctx.shell_args(NSIDE, d_cat[j0:j1], j1 - j0, stride, 0, runner.epsilon_max, md)
        counts, base, pix, r_com, halo = ctx.disc_enumerate(args, spline, fallback4)
        yield j0, args, spline, counts.cpu().numpy(), base.cpu().numpy(), pix, r_com.cpu().numpy(), halo, \
            D_a(np.asarray(cat["z"][j0:j1], dtype=np.float64))
        j0 = j1


class PaintProfilesShell(DefaultRunner):
    """Paint a tabulated projected profile around every halo onto the shell (HealpixRunner.py:376-483)."""

    def _validated_keys(self):
        """the reference's argument checks, before anything touches the GPU"""
        assert self.model is not None, "You must provide a model"         # :446
        keys = self._keys_checked()
        self._callable_model = False
        if not _is_paint_table(self.model):
            if hasattr(self.model, "setup_interpolator"):
                raise NameError("No Table created. Run setup_interpolator() method first")
            if not callable(getattr(self.model, "projected", None)):
                raise TypeError(f"PaintProfilesShell needs a tabulated model (TabulatedProfile / ParamTabulatedProfile with "
                                f"raw_input_2D) or an object with a .projected(cosmo, r, M, a) method; got {type(self.model)}")
            self._callable_model = True                                   # (no p_keys here: :436-443 asserted above)
        # (a table with more p_keys axes than the shell kernels read -- Tabulate.py:497-650 is N-dimensional -- is the same call:
        # the library blends every halo's radial row first and runs the tile path on the rows, csrc/bfg_ndtable.hpp)
        return keys

    def _paint_callable(self, d_map, fresh, keys=()):
        """HealpixRunner.py:449-481 with `Baryons.projected` called per halo on the host (it is a Python callable): the GPU lists
        every disc's pixels and distances and adds the returned values to the map"""
        ctx = get_context()
        NSIDE = self.LightconeShell.NSIDE
        npix = 12 * NSIDE * NSIDE
        pixarea = 4.0 * np.pi / npix
        if d_map is None:
            d_map = ctx.zeros(npix)                                       # :424
        elif fresh:
            d_map.zero_()
        cat = self.HaloLightConeCatalog.cat
        total = 0
        for j0, args, spline, counts, base, pix, r_com, halo, D in _callable_batches(self, ctx, False):
            vals = np.zeros(r_com.size)
            for j in range(counts.size):                                  # (an empty disc still gets its call, as in the reference)
                sl = slice(base[j], base[j] + counts[j])
                M_j, a_j = cat["M"][j0 + j], 1 / (1 + cat["z"][j0 + j])
                o_j = {key: cat[key][j0 + j] for key in keys}             # :456
                Paint = np.asarray(self.model.projected(self.cosmo, r_com[sl], M_j, a_j, **o_j), dtype=np.float64).reshape(-1)   # :472
                Paint = np.where(np.isfinite(Paint), Paint, 0)            # :473
                if self.include_pixel_size:
                    Paint = Paint * (pixarea * D[j] ** 2)                 # :478
                vals[sl] = Paint
            ctx.map_add_values(d_map, pix, ctx.to_device(vals))           # :481
            total += int(r_com.size)
        self.last_stats = dict(ctx.stats(), pixel_updates=total)
        return d_map

    def process_device(self, d_map=None, overwrite=None, slices=1, on_slice=None, sync_stats=True):
        """Paint into a device map (float64[Npix] torch tensor) and return it.

        d_map=None: a fresh map, defined entirely by the call.  A given map is accumulated INTO unless overwrite=True
        (its previous contents are then ignored: BFG_SHELL_OUT_OVERWRITE).
        slices, on_slice: bfg_paint_shell_sliced -- on_slice(k, n, lo, hi) is called after the k-th of n slices of the map,
        d_map[lo:hi], has been enqueued: it is final once the current stream gets there, so an exchange of that slice can
        start while the next one is painted (utils.Parallelize.SplitJoinParallel).
        sync_stats=False: do not read the counters back (that synchronises the stream); `collect_stats()` does it later --
        what a pipeline over several shells wants."""
        keys = self._validated_keys()
        if self._callable_model:
            fresh = d_map is None or bool(overwrite)
            d_map = self._paint_callable(d_map, fresh, keys)
            if on_slice is not None:                                      # nothing to cut: the whole map as one slice
                on_slice(0, 1, 0, int(d_map.numel()))
            return d_map
        ctx = get_context()
        NSIDE = self.LightconeShell.NSIDE
        bg, spline, d_cat, stride = self._device_inputs(ctx, keys)

        def log_table():                                                  # Tabulate.py:270-271 keeps ln T
            with np.errstate(all="ignore"):
                return np.log(np.asarray(self.model.raw_input_2D, dtype=np.float64))
        table = ctx.table(_table_axes(self.model, keys), log_table, log_values=True,
                          cache_key=(self.model, "2D", self.model.raw_input_2D))
        fresh = d_map is None
        if fresh:
            d_map = ctx.empty(12 * NSIDE * NSIDE)                         # :424 -- the zeros come from the kernels (OUT_OVERWRITE)
        args = ctx.shell_args(NSIDE, d_cat, d_cat.shape[0], stride, len(keys), self.epsilon_max,
                              ctx.massdef_struct(bg, self.mass_def), include_pixel_size=self.include_pixel_size,
                              variant=self.variant, out_overwrite=fresh if overwrite is None else bool(overwrite),
                              # one plan, K models: if the context's previous shell call was given this very catalog tensor, its
                              # per-halo records and pair lists serve this model too where everything but the table's values agrees
                              # (the library checks that much: BFG_SHELL_REUSE_PLAN) -- examples/05_Paint_tSZ_shell.ipynb:303-324
                              reuse_plan=ctx.same_catalog(d_cat))
        if sync_stats:
            ctx.stats_reset()
        ctx.paint_shell(args, table, spline, d_map, slices=slices, on_slice=on_slice)
        self.last_stats = None
        if sync_stats:
            self.collect_stats()
        return d_map

    def collect_stats(self):
        """read the device counters (synchronises the stream), keep them in `last_stats`, emit the warnings"""
        if getattr(self, "_callable_model", False) and self.last_stats is not None:
            return self.last_stats                                        # counted on the host (_paint_callable)
        self.last_stats = get_context().stats()
        emit_fallback_warning(self.last_stats)
        return self.last_stats

    def process(self, out=None):
        """returns new_map : float64[Npix] (RING), the sum over halos of the painted profiles.

        out: optional float64 numpy array of the map's size to receive the result -- page-locked memory if the copy is to run
        at PCIe speed (engine.pinned_empty; a pageable array costs ~3 ms more per 101 MB); by default the result lands in
        page-locked memory from torch's caching host allocator.
        The map leaves the GPU in slices while it is being painted (bfg_paint_shell_sliced + a copy stream): the copy of slice k
        overlaps the painting of slice k + 1 (measured at the headline size: 3.9 ms with one slice, 2.6 ms with eight)."""
        import os
        import torch
        self._validated_keys()
        ctx = get_context()
        npix = 12 * self.LightconeShell.NSIDE ** 2
        shape = np.shape(self.LightconeShell.map)
        if out is not None:
            if not (isinstance(out, np.ndarray) and out.dtype == np.float64 and out.size == npix and out.flags["C_CONTIGUOUS"]):
                raise ValueError("out must be a C-contiguous float64 array with one element per pixel")
            h = torch.from_numpy(out.reshape(-1))
        else:
            try:
                h = torch.empty(npix, dtype=torch.float64, pin_memory=True)
            except RuntimeError:                                           # no page-locked memory to be had
                h = torch.empty(npix, dtype=torch.float64)
        d_map = ctx.empty(npix)
        main, side = torch.cuda.current_stream(ctx.device), ctx.copy_stream()

        def on_slice(k, n, lo, hi):
            ev = torch.cuda.Event()
            ev.record(main)                                                # slice k has been painted when this fires
            side.wait_event(ev)
            with torch.cuda.stream(side):
                h[lo:hi].copy_(d_map[lo:hi], non_blocking=True)
        ctx.stats_reset()
        self.process_device(d_map=d_map, overwrite=True, slices=int(os.environ.get("BFG_D2H_SLICES", "8")), on_slice=on_slice,
                            sync_stats=False)
        side.synchronize()                                                 # every slice is on the host; d_map may go
        self.collect_stats()
        return (out if out is not None else h.numpy()).reshape(shape)


class _ProductTable(object):
    """ln-space product of two tabulated profiles on one grid: exp(L1) * exp(L2) = exp(L1 + L2), and a multilinear
    read-out is linear in the node values, so the table of ln T1 + ln T2 reads out the product exactly."""

    def __init__(self, a, b, keys):
        for k in ["raw_input_z_range", "raw_input_M_range", "raw_input_r_range"] + ["raw_input_%s_range" % q for q in keys]:
            if not np.array_equal(np.asarray(getattr(a, k)), np.asarray(getattr(b, k))):
                raise ValueError(f"PaintProfilesAnisShell: model and Tracer_model must be tabulated on the same grid ({k} differs)")
            setattr(self, k, np.asarray(getattr(a, k), dtype=np.float64))
        with np.errstate(all="ignore"):
            self.ln_product = np.log(np.asarray(a.raw_input_2D, dtype=np.float64)) + \
                np.log(np.asarray(b.raw_input_2D, dtype=np.float64))


class PaintProfilesAnisShell(DefaultRunner):
    """
    Tracer-weighted painting (HealpixRunner.py:486-640): every halo paints `model` x `Tracer_model`, weighted per
    pixel by the input map over the total-mass map painted from `Mtot_model` (plus a uniform background):

        new[p] = orig[p] / Mtot[p] * sum_j Paint_j(r) Tracer_j(r) [pixarea D_j^2]
                 + background_val * global_tracer_fraction * (dV drho_m / Mtot[p]) * orig[p]

    The per-pixel weights factor out of the halo sum, so the job is two launches of the paint kernel (Mtot_model with
    include_pixel_size, and the ln-space product table of model and Tracer_model) and element-wise work on the GPU.
    """

    def __init__(self, HaloLightConeCatalog, LightConeShell, epsilon_max, model, Tracer_model, Mtot_model,
                 background_val, global_tracer_fraction, mass_def=None, include_pixel_size=False,
                 use_ellipticity=False, verbose=True, variant="auto"):
        self.Tracer_model = Tracer_model
        self.Mtot_model = Mtot_model
        self.background_val = background_val
        self.global_tracer_fraction = global_tracer_fraction
        super().__init__(HaloLightConeCatalog, LightConeShell, epsilon_max, model, use_ellipticity, mass_def,
                         include_pixel_size, verbose, variant)

    def process(self):
        import warnings
        from scipy import interpolate
        assert self.model is not None, "You must provide a model"
        keys = self._keys_checked()
        for m in (self.model, self.Tracer_model, self.Mtot_model):
            if not _is_paint_table(m):
                if hasattr(m, "setup_interpolator"):
                    raise NameError("No Table created. Run setup_interpolator() method first")
                raise TypeError(f"PaintProfilesAnisShell on the MI355X path needs tabulated models; got {type(m)}")
        import torch
        ctx = get_context()
        orig_map = np.asarray(self.LightconeShell.map)
        NSIDE = self.LightconeShell.NSIDE
        npix = 12 * NSIDE * NSIDE
        pixarea = 4.0 * np.pi / npix
        bg = Background(self.cosmo)
        cat = self.HaloLightConeCatalog.cat
        z_m = np.max(cat["z"]) if cat.size else 0.0
        z_t = np.linspace(0, z_m + 0.1, 1000)                              # :553-555
        D_a = interpolate.CubicSpline(z_t, bg.angular_diameter_distance(1 / (1 + z_t)))

        # total mass distribution of the halos (:565-571), then the uniform background (:573-582)
        d_mtot = PaintProfilesShell(self.HaloLightConeCatalog, self.LightconeShell, self.epsilon_max, self.Mtot_model,
                                    use_ellipticity=self.use_ellipticity, mass_def=self.mass_def,
                                    include_pixel_size=True, verbose=self.verbose, variant=self.variant).process_device()
        dL = 2 * _get_parameter(self.Mtot_model, "proj_cutoff")           # proj_cutoff == Lproj / 2
        dD = float(D_a(self.LightconeShell.redshift))
        dV = pixarea * ((dD + dL) ** 3 - dD ** 3)
        rho_halos = float(d_mtot.sum().item()) / (dV * npix)
        rho_m = float(bg.rho_x(1 / (self.LightconeShell.redshift + 1), "matter"))
        drho_m = float(np.clip(rho_m - rho_halos, 0, None))
        d_mtot += dV * drho_m
        if self.verbose:
            print(f"Inputted halos contribute {100*(rho_halos/rho_m):0.2f}% of the total matter density.")
            print("Remaining density is assigned to a uniform background.")
        if rho_halos > rho_m:
            warnings.warn("Inputted halos contribute more mass than is available for this mean matter density."
                          "Your Mtot_model profiles are either too extended or you are using the wrong cosmology.")

        # sum over halos of Paint x Tracer [x pixarea D^2]: one paint of the product table
        prod = _ProductTable(self.model, self.Tracer_model, keys)
        bgc, spline, d_cat, stride = self._device_inputs(ctx, keys)
        table = ctx.table(_table_axes(prod, keys), prod.ln_product, log_values=True)
        d_sum = ctx.empty(npix)                                          # defined by the call (the zeros come from the kernels), as the
        args = ctx.shell_args(NSIDE, d_cat, d_cat.shape[0], stride, len(keys), self.epsilon_max,   # Mtot paint above: the same plan where
                              ctx.massdef_struct(bgc, self.mass_def), include_pixel_size=self.include_pixel_size,   # include_pixel_size agrees
                              variant=self.variant, out_overwrite=True, reuse_plan=ctx.same_catalog(d_cat))
        ctx.stats_reset()
        ctx.paint_shell(args, table, spline, d_sum)
        self.last_stats = ctx.stats()

        d_orig = ctx.to_device(np.ascontiguousarray(orig_map, dtype=np.float64).reshape(-1))
        pos = d_mtot > 0
        safe = torch.where(pos, d_mtot, torch.ones_like(d_mtot))
        w = torch.where(pos, d_orig / safe, torch.zeros_like(d_orig))      # Mfrac weights (:619-620)
        new_map = d_sum * w
        new_map += (self.background_val * self.global_tracer_fraction) * torch.where(pos, (dV * drho_m) / safe,
                                                                                     torch.zeros_like(safe)) * d_orig
        return get_context().to_host(new_map).reshape(orig_map.shape)


class BaryonifyShell(DefaultRunner):
    """Baryonify a MASS map on the shell with a tabulated displacement model (HealpixRunner.py:235-373)."""

    def _checked_model_keys(self):
        """the reference's argument checks (:304-311, BaryonCorrection.py:454-455), before anything touches the GPU"""
        keys = self._keys_checked()
        self._callable_model = False
        if not _is_disp_table(self.model):
            ours = isinstance(self.model, BaryonificationClass) or hasattr(self.model, "setup_interpolator")
            if self.model is not None and ours:
                raise NameError("No Table created. Run setup_interpolator() method first")
            if not callable(getattr(self.model, "displacement", None)):
                raise TypeError(f"BaryonifyShell needs a BaryonificationClass model with a displacement table, or an object "
                                f"with a .displacement(r, M, a) method; got {type(self.model)}")
            self._callable_model = True
        return keys

    def _offsets_callable(self, keys=()):
        """HealpixRunner.py:313-355 with `model.displacement` called per halo on the host (a Python callable): the GPU lists every
        disc's pixels and distances (with the < 4 pixel rule) and turns the returned displacements into unit-vector offsets"""
        ctx = get_context()
        NSIDE = self.LightconeShell.NSIDE
        d_off = ctx.zeros(12 * NSIDE * NSIDE, 3)                          # :313
        cat = self.HaloLightConeCatalog.cat
        total, fb4 = 0, 0
        for j0, args, spline, counts, base, pix, r_com, halo, _D in _callable_batches(self, ctx, True):
            disp = np.zeros(r_com.size)
            for j in range(counts.size):
                if counts[j] == 0:                                        # only a NaN record: the reference would have raised in healpy
                    continue
                sl = slice(base[j], base[j] + counts[j])
                M_j, a_j = cat["M"][j0 + j], 1 / (1 + cat["z"][j0 + j])
                o_j = {key: cat[key][j0 + j] for key in keys}             # :322
                disp[sl] = np.asarray(self.model.displacement(r_com[sl], M_j, a_j, **o_j), dtype=np.float64).reshape(-1)   # :345
            ctx.offsets_add_displacements(args, spline, pix, halo, ctx.to_device(disp), d_off)                       # :345-355
            total += int(r_com.size)
        self.last_stats = dict(ctx.stats(), pixel_updates=total)
        return d_off

    def offsets_device(self, slices=1, on_slice=None, sync_stats=True):
        """Accumulate the unit-vector offsets of all halos (:313-355); returns float64[Npix, 3] on the device.
        sync_stats=False: the counters are not read back (that synchronises the stream); `collect_stats()` does it later.
        slices, on_slice: bfg_baryonify_offsets_sliced -- on_slice(k, n, lo, hi) after the k-th band slice of the field has been
        enqueued; lo / hi are ELEMENT indices of the flattened field (3 per pixel)."""
        keys = self._checked_model_keys()
        if self._callable_model:
            d_off = self._offsets_callable(keys)
            if on_slice is not None:                                      # nothing to cut: the whole field as one slice
                on_slice(0, 1, 0, int(d_off.numel()), d_off.view(-1))
            return d_off
        ctx = get_context()
        NSIDE = self.LightconeShell.NSIDE
        bg, spline, d_cat, stride = self._device_inputs(ctx, keys)
        model = self.model
        table = ctx.table(_table_axes(model, keys), lambda: np.asarray(model.raw_input_d, dtype=np.float64),
                          log_values=False, cache_key=(model, "d", model.raw_input_d))
        model_bg = Background(model.cosmo) if getattr(model, "cosmo", None) is not None else bg
        model_md = ctx.massdef_struct(model_bg, getattr(model, "mass_def", None))
        args = ctx.shell_args(NSIDE, d_cat, d_cat.shape[0], stride, len(keys), self.epsilon_max,
                              ctx.massdef_struct(bg, self.mass_def), model_md=model_md,
                              model_epsilon_max=model.epsilon_max,
                              rdelta_sampling=getattr(model, "Rdelta_sampling", False), variant=self.variant,
                              out_overwrite=True, reuse_plan=ctx.same_catalog(d_cat))
        d_off = ctx.empty(12 * NSIDE * NSIDE, 3)                          # :313 -- the zeros come from the kernels
        if sync_stats:
            ctx.stats_reset()
        if on_slice is not None:
            flat = d_off.view(-1)
            ctx.baryonify_offsets(args, table, spline, d_off, slices=slices,
                                  on_slice=lambda k, n, lo, hi: on_slice(k, n, lo, hi, flat))
        else:
            ctx.baryonify_offsets(args, table, spline, d_off)             # :315-355
        self.last_stats = None
        if sync_stats:
            self.collect_stats()
        return d_off

    def collect_stats(self):
        """read the device counters (synchronises the stream), keep them in `last_stats`, emit the warnings"""
        if getattr(self, "_callable_model", False) and self.last_stats is not None:
            return self.last_stats
        self.last_stats = get_context().stats()
        emit_range_warnings(self.last_stats, "table")                     # BaryonCorrection.py:382-394
        emit_fallback_warning(self.last_stats)
        return self.last_stats

    def process(self, distributed=None):
        """returns new_map : float64[Npix]; the input array itself if the map is all zeros (:293-294).

        distributed: a utils.Parallelize.Exchange (set by SplitJoinParallel; a torch.distributed module with an initialised
        group is wrapped into one): this runner then holds one sky-patch shard of the halos.  The offset field is linear
        in halos (:355), so it is summed across the ranks -- by a reduce-scatter, because every rank regrids only the
        sources of the pixel range it owns (:357-365 on that range) --, and the regridded maps, whose deposits can cross
        the range borders, are all-reduced."""
        if distributed is None:
            return _baryonify_pipelined([self])[0]
        return _baryonify_process(self, _BaryonifyDeviceOps(self), distributed)


class _BaryonifyDeviceOps(object):
    """the GPU side of BaryonifyShell.process: HBM tensors and C-ABI calls"""

    def __init__(self, runner):
        self.runner = runner
        self.h2d_bytes = 0                                                # of the input map, for tools/bary_api_probe.py and the tests

    def slice_cuts(self, nside, slices):
        """ELEMENT cuts (3 per pixel) of the slices offsets() will report: known before the call (bfg_shell_slice_cuts), so the pixels
        this rank will own can be uploaded while the offsets are still being accumulated"""
        from .._lib import shell_slice_cuts
        if getattr(self.runner, "_callable_model", False) or slices <= 1:
            return [0, 3 * 12 * nside * nside]                            # (a host-evaluated model reports its field in one piece)
        return shell_slice_cuts(nside, True, slices)

    def upload_ranges(self, flat, ranges, npix):
        """float64[npix] on the device: flat[lo:hi] inside the given pixel ranges, zeros elsewhere -- only those bytes cross PCIe"""
        import torch
        ctx = get_context()
        d = ctx.zeros(npix)
        for lo, hi in ranges:
            if hi > lo:
                d[lo:hi].copy_(torch.from_numpy(flat[lo:hi]))
                self.h2d_bytes += 8 * (hi - lo)
        return d

    def zeros(self, *shape):
        return get_context().zeros(*shape)

    def offsets(self, slices=1, on_slice=None):
        return self.runner.offsets_device(slices=slices, on_slice=on_slice)

    def regrid(self, nside, d_off, d_in, d_out, d_sums=None):
        get_context().regrid_shell(nside, d_off, d_in, d_out, d_sums)

    def count_above(self, d_in, ranges, threshold, d_dst):
        """d_dst[0] = number of pixels of the ranges that are NOT within `threshold` of zero, left on the device: |value| > threshold
        or NaN -- exactly the pixels that make np.allclose(map, 0) False"""
        total = None
        for lo, hi in ranges:
            if hi > lo:
                c = (~(d_in[lo:hi].abs() <= threshold)).sum()
                total = c if total is None else total + c
        if total is not None:
            d_dst[0] = total

    def to_host(self, t):
        return get_context().to_host(t)


def _checks_or_zero_map(runner, orig_map):
    """The reference returns an all-zero input map BEFORE it looks at the model (HealpixRunner.py:293-294 precede :304-311 and the
    table lookup).  Here the argument checks come first because they are free and the zero test of a large map is a device
    reduction; so when they fail, the (host) zero test decides whether the reference would have raised at all.
    Returns True if the input map is to be handed back unchanged."""
    try:
        runner._checked_model_keys()
    except Exception:
        if np.allclose(runner.LightconeShell.map, 0):
            return True
        raise
    return False


def _regrid_band_groups(NSIDE, n_groups):
    """the source bands of bfg_regrid_shell_bands cut into n_groups runs: (band cuts, RING pixel cuts)"""
    from .._lib import load
    from ..sharding import _ring_start
    tr = int(load().bfg_regrid_band_rings())
    nbands = (4 * NSIDE - 1 + tr - 1) // tr
    g = max(1, min(int(n_groups), nbands))
    cuts_b = [nbands * k // g for k in range(g + 1)]
    return cuts_b, [_ring_start(NSIDE, 1 + b * tr) for b in cuts_b]


def _baryonify_pipelined(runners, in_flight=2):
    """BaryonifyShell.process (HealpixRunner.py:252-373) for one or several shells on ONE GPU with the host transfers off the
    critical path.  Per shell the reference does: map to the device, offsets, regrid, map back -- 1.8 + 0.6 + 0.3 + 1.8 ms at
    BASELINE configs[2], the two PCIe legs being three quarters of it.  Here the offset kernels are enqueued BEFORE the upload
    (they do not need the map; the pageable copy blocks the host thread, not the GPU, and runs on a stream of its own), nothing
    is read back in between (the mass sums come out of the regrid kernel, the all-zero test out of a device reduction, the
    counters once at the end), and the copy of shell k's result to page-locked memory overlaps the upload and the kernels of
    shell k + 1 (PCIe is full duplex -- with a copy kernel for one of the two directions): 4.7 -> 2.8 ms per shell of a list,
    4.7 -> 4.4 ms for a single shell (tools/bary_api_probe.py, tools/bary_pipe_probe.py).  Returns the list of new maps (the input array itself where it is all zeros, :293-294)."""
    n = len(runners)
    results, pend = [None] * n, []
    todo = []
    for k, R in enumerate(runners):                                       # the reference's checks first, before anything touches the GPU
        orig = R.LightconeShell.map
        if orig.size < (1 << 16) and np.allclose(orig, 0):                # small maps: decided on the host, as the reference (:293-294)
            results[k] = orig
            continue
        if _checks_or_zero_map(R, orig):
            results[k] = orig
            continue
        todo.append(k)
    if not todo:
        return results
    import os
    import torch
    ctx = get_context()
    dev = ctx.device
    up, down = ctx.upload_stream(), ctx.copy_stream()

    def finish(item):
        k, orig, h, h_small, ev, d_out = item
        ev.synchronize()
        absmax, old_sum, new_sum, far = (float(x) for x in h_small.tolist())
        if far > 0:                                                       # a displacement of more than 3 rings somewhere: slices that had
            h.copy_(d_out)                                                # left may have received deposits since -- the whole map again
        if not (absmax > 1e-8) and np.allclose(orig, 0):                  # :293-294 (False for NaN maps)
            results[k] = orig
            return
        assert np.isclose(new_sum, old_sum), \
            "ERROR in pixel regridding, sum(new_map) [%0.14e] != sum(oldmap) [%0.14e]" % (new_sum, old_sum)  # :368-370
        results[k] = h.numpy().reshape(np.shape(orig))

    ctx.stats_reset()
    ran = []
    n_slices = int(os.environ.get("BFG_BARY_SLICES", "8"))
    for k in todo:
        R = runners[k]
        orig = R.LightconeShell.map
        NSIDE = R.LightconeShell.NSIDE
        npix = 12 * NSIDE * NSIDE
        while len(pend) >= in_flight:                                     # bounds the HBM in flight (~0.5 GB per shell at NSIDE 1024)
            finish(pend.pop(0))
        main = torch.cuda.current_stream(dev)
        flat = np.ascontiguousarray(orig, dtype=np.float64).ravel()
        try:
            h = torch.empty(npix, dtype=torch.float64, pin_memory=True)
            h_small = torch.empty(4, dtype=torch.float64, pin_memory=True)
        except RuntimeError:                                              # no page-locked memory to be had
            h, h_small = torch.empty(npix, dtype=torch.float64), torch.empty(4, dtype=torch.float64)
        d_out, d_small = ctx.zeros(npix), ctx.zeros(4)                    # d_small = {max |in|, sum(in), sum(deposits), far deposits}
        # The map in band slices (bfg_regrid_shell_bands): slice s is regridded as soon as it has arrived, and leaves once slice
        # s + 1 has been regridded too (deposits reach 3 rings beyond a band; anything farther is counted and, should it ever
        # happen, the whole map is copied again at the end) -- the upload, the regrid and the download of one shell overlap
        # instead of following each other: 1.8 + 0.3 + 1.8 ms at BASELINE configs[2] (tools/bary_api_probe.py).
        if NSIDE >= 32 and n_slices > 1 and n == 1:                      # (in a list the shells overlap each other: whole maps are faster)
            cuts_b, cuts_p = _regrid_band_groups(NSIDE, n_slices)
        else:
            cuts_b, cuts_p = None, [0, npix]
        S = len(cuts_p) - 1
        # A page-locked input map (LightconeShell(pinned=True), engine.pin / pinned_empty) goes up as asynchronous DMA slices that
        # never block the host thread; its result then comes down by DMA as well -- measured on the single-shell path (tools/
        # bary_timeline_probe.py, eight slices): 3.17 ms with both directions on the DMA engines, 3.40 ms with the download as a copy
        # kernel (whose stores slow the upload's slices from 0.29 to 0.40 ms each).  A pageable map keeps the copy kernel.
        h_src = torch.from_numpy(flat)
        src_pinned = h_src.is_pinned()
        down_by_kernel = not src_pinned
        if os.environ.get("BFG_BARY_DOWN"):
            down_by_kernel = os.environ["BFG_BARY_DOWN"] == "kernel"
        with torch.cuda.stream(up):
            d_orig = torch.empty(npix, dtype=torch.float64, device=dev)
        d_orig.record_stream(main)

        def upload(sl):                                                   # slice sl of the map -> d_orig, on the upload stream
            with torch.cuda.stream(up):
                d_orig[cuts_p[sl]:cuts_p[sl + 1]].copy_(h_src[cuts_p[sl]:cuts_p[sl + 1]], non_blocking=True)
                e = torch.cuda.Event()
                e.record(up)
            return e
        # A page-locked map: every upload slice is enqueued BEFORE the offset kernels are even set up (an asynchronous DMA copy costs
        # the host ~10 us; setting up the offsets call -- catalog stamp, table and spline lookup, arguments -- takes it ~0.25 ms, during
        # which the PCIe link would idle).  A pageable map: the offsets first, because its copies block the host while the GPU works.
        ev_ups = [upload(sl) for sl in range(S)] if src_pinned else None
        d_off = R.offsets_device(sync_stats=False)                        # :313-355, enqueued; the GPU works while the host copies
        ran.append(R)

        def send(lo, hi):                                                 # d_out[lo:hi] -> h[lo:hi] on the download stream
            if hi <= lo:
                return
            if h.is_pinned() and (n > 1 or S > 1) and down_by_kernel:
                # a copy KERNEL, not the DMA engine: an upload is (or will be) running as a DMA copy, and two DMA copies in opposite
                # directions take turns on this platform (tools/copy_probe.py: 3.7 ms for the pair, 2.3 ms with the kernel)
                ctx.copy_to_pinned(h[lo:hi], d_out[lo:hi])
            else:
                h[lo:hi].copy_(d_out[lo:hi], non_blocking=True)
        prev_ev = None
        for sl in range(S):
            lo, hi = cuts_p[sl], cuts_p[sl + 1]
            ev_up = ev_ups[sl] if ev_ups is not None else upload(sl)
            main.wait_event(ev_up)
            d_small[0] = torch.maximum(d_small[0], d_orig[lo:hi].abs().max())   # np.allclose(orig_map, 0) <=> max |map| <= 1e-8; NaN sticks
            if cuts_b is None:
                ctx.regrid_shell(NSIDE, d_off, d_orig, d_out, d_small[1:3])     # :357-365; {sum(in), sum(deposits)} from the kernel
            else:
                ctx.regrid_shell_bands(NSIDE, d_off, d_orig, d_out, d_small[1:], cuts_b[sl], cuts_b[sl + 1])
            ev_rg = torch.cuda.Event()
            ev_rg.record(main)
            if sl >= 1:                                                   # slice sl - 1 is final now
                down.wait_event(ev_rg)
                with torch.cuda.stream(down):
                    send(cuts_p[sl - 1], cuts_p[sl])
            prev_ev = ev_rg
        down.wait_event(prev_ev)
        with torch.cuda.stream(down):
            send(cuts_p[S - 1], cuts_p[S])
            h_small.copy_(d_small, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(down)
        d_out.record_stream(down)
        d_small.record_stream(down)
        pend.append((k, orig, h, h_small, ev, d_out))
    for item in pend:
        finish(item)
    up.synchronize()                                                      # (idle by now) the runtime drops its page locks on the callers' maps
    if ran:
        stats = ran[0].collect_stats()                                    # one read-back for the whole list
        for R in ran[1:]:
            if not getattr(R, "_callable_model", False):
                R.last_stats = stats
    return results


def _baryonify_process(runner, ops, exchange, slices=1):
    """BaryonifyShell.process (HealpixRunner.py:252-373) over an `ops` object (the device side: _BaryonifyDeviceOps) and an
    optional Exchange between ranks.

    Several ranks: every rank holds a shard of the halos and accumulates their offsets over the whole sky (:313-355); the field is
    summed by reduce-scatter -- in `slices` band slices while the rest is still being accumulated: rank r ends up owning the r-th
    part of EVERY slice (a slice whose length does not divide by the world size is all-reduced instead) -- and a rank regrids the
    sources of the pixels it owns (:357-365 on that range).  Which pixels those are is known before the call (the slices depend
    on NSIDE and the slice count only: bfg_shell_slice_cuts), so ONLY THEY are uploaded: 8 Npix / N bytes of input map per rank.
    The regridded maps, whose deposits cross the range borders, are all-reduced together with three scalars -- sum(in),
    sum(deposits), count(|in| > 1e-8) of the rank's sources -- so the mass assertion (:368-370) and the all-zero test (:293-294)
    need no collective or read-back of their own: with several ranks nothing comes back from the device before the final map.  On
    ONE rank the count is read back (one 8-byte copy, one synchronisation) before the offsets are accumulated, so that an all-zero
    shell costs its upload and nothing else, as in the reference.  The count includes NaN pixels (count_above), so it alone
    decides np.allclose(orig_map, 0): no second pass over the map on the host."""
    orig_map = runner.LightconeShell.map
    NSIDE = runner.LightconeShell.NSIDE
    if orig_map.size < (1 << 16) and np.allclose(orig_map, 0):         # small maps: decided on the host, as the reference
        return orig_map
    if _checks_or_zero_map(runner, orig_map):
        return orig_map
    npix = 12 * NSIDE * NSIDE
    flat = np.ascontiguousarray(orig_map, dtype=np.float64).ravel()
    if exchange is not None and not hasattr(exchange, "reduce_scatter"):
        from ..utils.Parallelize import Exchange
        exchange = Exchange(exchange)
    multi = exchange is not None and exchange.world > 1
    rank, world = (exchange.rank, exchange.world) if multi else (0, 1)
    # the slices of the offset field and, in each, the pixels this rank owns after its exchange
    cuts = [int(c) for c in ops.slice_cuts(NSIDE, slices)] if multi else [0, 3 * npix]
    assert cuts[0] == 0 and cuts[-1] == 3 * npix and all(c % 3 == 0 for c in cuts)
    plan, owned = [], []
    for lo, hi in zip(cuts, cuts[1:]):
        plo, npx = lo // 3, (hi - lo) // 3
        scatter = multi and npx % world == 0                          # reduce-scatter if the slice splits evenly, else all-reduce
        plan.append((lo, hi, scatter))
        owned.append((plo + npx * rank // world, plo + npx * (rank + 1) // world))
    assert sum(hi - lo for lo, hi in owned) > 0 or npix < world
    d_in = ops.upload_ranges(flat, owned, npix)                        # sources this rank does not own have no mass here
    d_out = ops.zeros(npix + 3)
    ops.count_above(d_in, owned, 1e-8, d_out[npix + 2:])              # np.allclose(orig_map, 0) <=> no pixel with |value| > 1e-8 or NaN
    if not multi:
        # one rank: nobody waits for this rank's collectives, so the all-zero test (:293-294) is read back BEFORE the offsets are
        # accumulated and the map regridded -- an empty shell costs its upload and one small copy, as in the reference; with several
        # ranks the count rides on the final all-reduce instead (a read-back here would cost every rank a synchronisation per shell)
        if float(ops.to_host(d_out[npix + 2:])[0]) == 0:
            return orig_map
    handles, seen = [], []

    def on_slice(k, n, lo, hi, field):                                # element range [lo, hi) of the flattened field: final
        seen.append((lo, hi))
        if (lo, hi) != plan[k][:2] or n != len(plan):
            raise RuntimeError(f"slice {k} of {n} is [{lo}, {hi}), planned {plan[k][:2]} of {len(plan)}")
        if plan[k][2]:
            handles.append(exchange.reduce_scatter_begin(field[lo:hi]))
        else:
            handles.append(exchange.allreduce_begin(field[lo:hi]))
    if multi:
        d_off = ops.offsets(slices=len(plan) if len(plan) > 1 else 1, on_slice=on_slice)   # :313-355, this rank's halos
        assert len(seen) == len(plan), "the offsets call reported fewer slices than planned"
        for h in handles:
            exchange.wait(h)
    else:
        d_off = ops.offsets()
    ops.regrid(NSIDE, d_off, d_in, d_out[:npix], d_out[npix:npix + 2])  # :357-365; {sum(in), sum(deposits)} of this rank's sources
    if multi:
        exchange.allreduce(d_out)                                     # the map and the three scalars in one collective
    host = ops.to_host(d_out)                                          # the one read-back
    old_sum, new_sum, n_above = (float(x) for x in host[npix:])
    if n_above == 0:                                                   # :293-294 (NaN pixels are counted: a NaN map is not all-zero)
        return orig_map
    assert np.isclose(new_sum, old_sum), \
        "ERROR in pixel regridding, sum(new_map) [%0.14e] != sum(oldmap) [%0.14e]" % (new_sum, old_sum)  # :368-370
    return host[:npix]
