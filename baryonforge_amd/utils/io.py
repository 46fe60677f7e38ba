"""
Host-side containers of the shell path, mirroring the reference's
BaryonForge/utils/io.py: `HaloLightConeCatalog` (:9-140) and `LightconeShell`
(:290-379).  Same constructor signatures, attributes and error behaviour;
healpy is not needed (npix2nside is arithmetic, FITS reading is out of scope).
"""
import warnings

import numpy as np

from ..background import check_cosmology_dict

__all__ = ["HaloLightConeCatalog", "LightconeShell", "HaloNDCatalog", "GriddedMap", "ParticleSnapshot", "npix2nside",
           "nside2npix"]


def nside2npix(nside):
    return 12 * int(nside) * int(nside)


def npix2nside(npix):
    """hp.npix2nside (io.py:353): raises ValueError unless npix = 12 nside^2"""
    nside = int(round(np.sqrt(npix / 12.0)))
    if nside < 1 or 12 * nside * nside != npix:
        raise ValueError("Wrong pixel number (it is not 12*nside**2)")
    return nside


class HaloLightConeCatalog(object):
    """
    Halo lightcone catalog: structured float64 array `cat` with fields
    M, z, ra, dec (+ any extra per-halo properties, e.g. `cdelta`), and the
    cosmology dict.  Mirrors utils/io.py:9-140.

    Parameters
    ----------
    ra, dec : array_like   degrees
    M : array_like         halo mass, Msun
    z : array_like         redshift
    cosmo : dict           needs Omega_m, sigma8, h, Omega_b, n_s, w0
    **arrays               extra per-halo columns
    """

    def __init__(self, ra, dec, M, z, cosmo, **arrays):
        t = np.float64
        dtype = [("M", t), ("z", t), ("ra", t), ("dec", t)]
        dtype = dtype + [(name, t) for name, arr in arrays.items()]
        cat = np.zeros(len(ra), dtype)

        if np.any(np.abs(dec) == 90):                                    # io.py:65-68
            dec = np.asarray(dec).astype(t)
            warnings.warn("Some halos found with declination exactly at the poles. Offsetting these by 4e-5 arcsec")
            dec = np.clip(dec, -90 + 1e-8, 90 - 1e-8)

        cat["ra"] = ra
        cat["dec"] = dec
        cat["z"] = z
        cat["M"] = M
        for name, arr in arrays.items():
            cat[name] = arr
        self.cat = cat

        check_cosmology_dict(cosmo)                                      # io.py:79-85
        self.cosmo = cosmo

    @property
    def data(self):
        return self.cat

    @property
    def cosmology(self):
        return self.cosmo

    def __getitem__(self, key):                                          # io.py:104-129
        other = {k: self.cat[k][key] for k in self.cat.dtype.names if k not in ["ra", "dec", "M", "z"]}
        return HaloLightConeCatalog(ra=self.cat["ra"][key], dec=self.cat["dec"][key], M=self.cat["M"][key],
                                    z=self.cat["z"][key], cosmo=self.cosmo, **other)

    def __len__(self):
        return self.cat.size

    def records(self, extra_keys=()):
        """float64 [n, 4 + len(extra_keys)] record matrix (M, z, ra, dec, extras...) for the device."""
        cols = ["M", "z", "ra", "dec"] + list(extra_keys)
        names = self.cat.dtype.names
        if (self.cat.flags["C_CONTIGUOUS"] and tuple(names[:len(cols)]) == tuple(cols) and
                all(self.cat.dtype[nm] == np.float64 for nm in names) and self.cat.dtype.itemsize == 8 * len(names)):
            # the structured array IS the record matrix (all fields float64, packed, M z ra dec first, utils/io.py:58-60):
            # a view, no copy; columns beyond the requested ones just widen the stride
            return self.cat.view(np.float64).reshape(self.cat.size, len(names))
        out = np.empty((self.cat.size, len(cols)), dtype=np.float64)
        for i, c in enumerate(cols):
            out[:, i] = self.cat[c]
        return out

    def _sample_stamp(self):
        """a cheap content stamp of `cat`: its size and the bytes of ~256 records spread over it.  Writes through a view taken BEFORE
        the array was locked stay possible (numpy cannot revoke them); a bulk edit through such a view -- m = Cat.cat['M'] before the
        first process(), m *= 2 after it -- changes every record and so this stamp, and the device copy is refreshed.  An edit of a
        single element through such a view is not detectable short of re-reading all 32 B per halo on every call: invalidate() /
        unlock() (or BFG_CATALOG_CACHE=0: upload on every call; BFG_CATALOG_CACHE=full: a checksum of every byte on every call) is
        the documented way to edit a catalog between calls."""
        import os
        cat = self.cat
        if os.environ.get("BFG_CATALOG_CACHE", "1") == "full":
            # every byte, every call: the 128-bit hash the table cache uses (engine._digest: xxh3, ~20 GB/s -- ~1.6 ms per 1e6 halos,
            # more than the painting itself takes on the GPU, which is why it is not the default).  Position-dependent: catches
            # single-element edits AND permutations (a swap of two halos' masses, np.random.shuffle through an old view), which a
            # sum / xor of the words would not
            from ..engine import _digest
            return (cat.size, cat.dtype.str, _digest(np.ascontiguousarray(cat).view(np.uint8).reshape(-1)))
        step = max(1, cat.size // 256)
        sample = np.ascontiguousarray(cat[::step])
        return (cat.size, cat.dtype.str, sample.tobytes())

    def device_records(self, ctx, extra_keys=()):
        """(device record matrix, doubles per record) on the GPU of `ctx`, uploaded once per catalog.

        The reference re-reads `cat` on every process() call.  To keep that guarantee without re-uploading 32 B per halo
        per call, the structured array is made READ-ONLY while a device copy exists (numpy then refuses in-place edits
        with a ValueError instead of letting the copy go stale); `unlock()` makes it writable again and drops the copies,
        replacing `self.cat` by another array is noticed by identity, and a sampled content stamp catches bulk edits made through
        views that existed before the lock (see _sample_stamp).  BFG_CATALOG_CACHE=0 uploads on every call."""
        import os
        recs = self.records(extra_keys)
        if os.environ.get("BFG_CATALOG_CACHE", "1") == "0":
            return ctx.to_device(recs), recs.shape[1]
        cache = self.__dict__.setdefault("_device_copies", {})
        key = (ctx.device_index, tuple(extra_keys))
        hit = cache.get(key)
        stamp = self._sample_stamp()
        if hit is not None and hit[0] is self.cat and not self.cat.flags.writeable and hit[3] == stamp:
            return hit[1], hit[2]
        cache.clear()                                                    # another array, writable in between, or edited through a view
        self.__dict__.pop("_zmax", None)
        d = ctx.to_device(recs)
        self.cat.setflags(write=False)                                   # (never raises for write=False)
        cache[key] = (self.cat, d, recs.shape[1], stamp)
        return d, recs.shape[1]

    def z_max(self):
        """max(z) of the catalog (HealpixRunner.py:297 / :429 take it on every call: 0.5 ms per 1e6 halos); remembered while the
        array is locked by device_records() and its sampled stamp is unchanged, recomputed otherwise"""
        hit = self.__dict__.get("_zmax")
        if hit is not None and hit[0] is self.cat and not self.cat.flags.writeable and hit[2] == self._sample_stamp():
            return hit[1]
        z_m = float(np.max(self.cat["z"])) if self.cat.size else 0.0
        self.__dict__["_zmax"] = (self.cat, z_m, self._sample_stamp())
        return z_m

    def unlock(self):
        """make `cat` writable again and forget its device copies (see device_records)"""
        self.__dict__.pop("_device_copies", None)
        self.__dict__.pop("_zmax", None)
        self.cat.setflags(write=True)

    def invalidate(self):
        """tell the container that `cat` was (or is about to be) edited in place: the device copy and the cached max(z) are dropped
        and the array is writable again; the next process() call uploads the catalog afresh (= unlock())"""
        self.unlock()

    def __getstate__(self):                                              # device tensors do not travel
        st = dict(self.__dict__)
        st.pop("_device_copies", None)
        st.pop("_zmax", None)
        return st

    def __str__(self):
        return (f"HaloLightConeCatalog with {self.cat.size} halos; fields {self.cat.dtype.names}; "
                f"cosmology {self.cosmo}")


class LightconeShell(object):
    """
    Full-sky HEALPix (RING) shell: `map`, `NSIDE`, `redshift`, `cosmo`.
    Mirrors utils/io.py:290-379.  `path`: a HEALPix FITS file, read as `hp.read_map(path)` reads it (:346-347: first map column,
    RING order -- a NESTED file is reordered --, the file's dtype) by utils/fits.py on numpy alone, or by healpy itself where it is
    installed; a .npy file is accepted as well.
    """

    def __init__(self, map=None, path=None, cosmo=None, redshift=None, pinned=False):
        """pinned (not in the reference): True (or "copy") replaces the map by a page-locked copy of it (torch's host allocator:
        hipHostMalloc), so that the runners' host <-> device transfers of this shell run asynchronously, in slices behind the kernels
        (BaryonifyShell.process() at BASELINE configs[2]: 4.06 -> 3.15 ms, 2.95 -> 2.46 ms per shell of a list; tools/pinned_probe.py).
        "inplace" page-locks the caller's own array instead (engine.pin: hipHostRegister) -- same speed, no copy -- but ONLY if the array
        is a page-aligned buffer that owns its pages (engine.aligned_empty, a private mmap; engine.pin_ok): an array in the
        allocator's heap shares pages with other objects, and in round 5's soak two of ~5000 heap arrays registered in place ended in
        a GPU memory fault inside an asynchronous DMA copy (profiles/r05_soak.txt) -- such a map gets a page-locked COPY and a
        UserWarning.  The copy is float64 (what the kernels read), whatever the map's dtype.  Needs the GPU: without one the map
        stays pageable and a UserWarning says so; any other failure is raised."""
        if (path is None) & (map is None):
            raise ValueError("Need to provide either path to map, or provide map values in healpix ring configuration")
        elif isinstance(path, str):
            if path.endswith(".npy"):
                self.map = np.load(path)
            else:
                from .fits import read_healpix_map
                self.map = read_healpix_map(path)                         # io.py:347 hp.read_map(path)
        elif isinstance(map, np.ndarray):
            self.map = map

        self.NSIDE = npix2nside(self.map.size)
        self.redshift = redshift
        if pinned:
            import warnings
            from .. import engine, _lib
            try:
                if pinned == "inplace" and self.map.dtype == np.float64 and engine.pin_ok(self.map):
                    engine.pin(self.map)
                else:
                    if pinned == "inplace":
                        warnings.warn("LightconeShell(pinned='inplace'): the map is not a page-aligned float64 buffer that owns its "
                                      "pages (engine.aligned_empty / engine.pin_ok); a page-locked copy is used instead", UserWarning)
                    self.map = engine.pinned_copy(self.map)
            except (_lib.BFGError, RuntimeError, MemoryError) as exc:    # no GPU / no page-locked memory: the map stays pageable
                warnings.warn(f"LightconeShell(pinned=...): the map stays in pageable memory ({exc})", UserWarning)

        if cosmo is None:
            raise ValueError("Not all cosmology parameters provided. I need Omega_m, sigma8, h, sigma8, Omega_b, n_s, w0")
        check_cosmology_dict(cosmo)                                      # io.py:357-363
        self.cosmo = cosmo

    @property
    def data(self):
        return self.map

    @property
    def cosmology(self):
        return self.cosmo


class HaloNDCatalog(object):
    """
    Halo catalog in a periodic 2D or 3D box (io.py:143-287): `cat` with fields M, x, y, z (+extras) stored, as the
    reference does, in big-endian float32 (io.py:204); `redshift`; `cosmology`.  z=None means a 2D field (z column 0).
    """

    def __init__(self, x, y, M, redshift, cosmo, z=None, **arrays):
        dtype = [("M", ">f"), ("x", ">f"), ("y", ">f"), ("z", ">f")]
        dtype = dtype + [(name, ">f", np.asarray(arr).shape[1:] if np.ndim(arr) > 1 else "") for name, arr in arrays.items()]
        N = 1 if not isinstance(x, (list, np.ndarray, tuple)) else len(x)
        cat = np.zeros(N, dtype)
        cat["x"] = x
        cat["y"] = y
        cat["z"] = 0 if z is None else z
        cat["M"] = M
        for name, arr in arrays.items():
            cat[name] = arr
        self.cat = cat
        self.redshift = redshift
        check_cosmology_dict(cosmo)
        self.cosmo = cosmo

    def __str__(self):
        return "HaloNDCatalog with %d halos at z = %s. Cosmology: %s" % (self.cat.size, str(self.redshift), str(self.cosmo))

    __repr__ = __str__

    @property
    def data(self):
        return self.cat

    @property
    def cosmology(self):
        return self.cosmo


class GriddedMap(object):
    """
    Periodic 2D / 3D mass or density grid (io.py:382-495): `map` (square / cubic), `bins` (the Npix pixel centres),
    `res`, `L` (centre of the last bin + res / 2), `Npix`, `is2D`, `grid` (np.meshgrid of the bins, 'xy') and `inds`.
    """

    def __init__(self, map=None, redshift=None, bins=None, cosmo=None):
        self.map = map
        self.redshift = redshift
        self.Npix = self.map.shape[0]
        self.res = bins[1] - bins[0]
        self.bins = bins
        self.L = bins[-1] + self.res / 2
        self.is2D = True if len(self.map.shape) == 2 else False
        if self.is2D:
            assert self.map.shape[0] == self.map.shape[1]
            self.grid = np.meshgrid(bins, bins, indexing="xy")
        else:
            assert (self.map.shape[0] == self.map.shape[1]) & (self.map.shape[1] == self.map.shape[2])
            self.grid = np.meshgrid(bins, bins, bins, indexing="xy")
        assert self.Npix == self.bins.size, f"Map has {self.Npix} pixels a side, but you passed {self.bins.size} bins"
        self.inds = np.arange(self.grid[0].size).reshape(self.grid[0].shape)
        check_cosmology_dict(cosmo)
        self.cosmo = cosmo

    @property
    def data(self):
        return self.map

    @property
    def cosmology(self):
        return self.cosmo


class ParticleSnapshot(object):
    """
    Particle snapshot of a periodic box (io.py:497-677): `cat` (float64 fields M, x, y, z), box size `L` [comoving
    Mpc], `redshift`, `is2D` (z=None), `cosmology`; `make_map(N_grid)` = nearest-grid-point mass histogram (:629-677).
    """

    def __init__(self, x=None, y=None, z=None, M=None, L=None, redshift=None, cosmo=None):
        dtype = [("M", np.float64), ("x", np.float64), ("y", np.float64), ("z", np.float64)]
        cat = np.zeros(len(x), dtype)
        cat["x"] = x
        cat["y"] = y
        cat["z"] = 0 if z is None else z
        cat["M"] = M
        self.L = L
        self.cat = cat
        self.redshift = redshift
        self.is2D = True if z is None else False
        check_cosmology_dict(cosmo)
        self.cosmo = cosmo

    @property
    def data(self):
        return self.cat

    @property
    def cosmology(self):
        return self.cosmo

    @classmethod
    def from_catalog(cls, cat, L, redshift, cosmo, is2D=False):
        """adopt a structured particle array (fields M, x, y, z; e.g. the output of BaryonifySnapshot.process()) without
        copying its columns (not in the reference, whose constructor always rebuilds the array)"""
        self = cls.__new__(cls)
        self.L, self.cat, self.redshift, self.is2D = L, cat, redshift, bool(is2D)
        check_cosmology_dict(cosmo)
        self.cosmo = cosmo
        return self

    def records(self):
        """the catalogue as a float64[n, 4] matrix (M, x, y, z) -- a view when `cat` is the packed float64 record array
        this class builds (io.py:553-560), else None"""
        names = self.cat.dtype.names
        if (names == ("M", "x", "y", "z") and self.cat.flags["C_CONTIGUOUS"] and self.cat.dtype.itemsize == 32 and
                all(self.cat.dtype[nm] == np.float64 for nm in names)):
            return self.cat.view(np.float64).reshape(self.cat.size, 4)
        return None

    def make_map(self, N_grid, mode="ngp", device=False):
        """mode 'ngp' (the reference's histogram) or 'cic' (cloud-in-cell, periodic); device=True deposits on the GPU
        (bfg_deposit_grid) -- the host path is numpy, as in the reference, and only does 'ngp'."""
        assert np.isnan(self.cat["M"]).sum() == 0, "If you want to make a map, provide a value for the particle mass"
        if device or mode != "ngp":
            from ..engine import get_context
            ctx = get_context()
            cols = ("x", "y") if self.is2D else ("x", "y", "z")
            recs = self.records()
            if recs is not None:                          # one upload of the records; positions and masses are column views
                d_recs = ctx.to_device(recs)
                d_pos, d_mass = d_recs[:, 1:1 + len(cols)], d_recs[:, 0]
            else:
                d_pos = ctx.to_device(np.stack([self.cat[c] for c in cols], axis=1))
                d_mass = ctx.to_device(self.cat["M"])
            return ctx.to_host(ctx.deposit_grid(d_pos, d_mass, self.L, N_grid, mode))
        bins = np.linspace(0, self.L, N_grid + 1)
        if self.is2D:
            coords = np.vstack([self.cat["x"], self.cat["y"]]).T
            bins = (bins, bins)
        else:
            coords = np.vstack([self.cat["x"], self.cat["y"], self.cat["z"]]).T
            bins = (bins, bins, bins)
        return np.histogramdd(coords, bins=bins, weights=self.cat["M"])[0]

