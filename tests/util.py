"""shared helpers for the parity tests (test-side only: this is where the oracle is allowed)"""
import numpy as np

from oracle import oracle as orc


def oracle_paint(cosmo, ra, dec, M, z, axes, T2D, nside, eps, include_pixel_size=False, extra=None,
                 Delta=200, rho_type="critical"):
    a, R, D = orc.halo_scalars(cosmo, M, z, Delta, rho_type)
    with np.errstate(all="ignore"):
        return orc.paint_shell(nside, ra, dec, M, a, D, R, axes, np.log(T2D), eps,
                               include_pixel_size=include_pixel_size, extra=extra)


def oracle_baryonify(cosmo, ra, dec, M, z, axes, d, nside, eps, eps_model, map_in, rdelta=False, extra=None,
                     offsets_only=False):
    a, R, D = orc.halo_scalars(cosmo, M, z)
    if offsets_only:
        return orc.baryonify_offsets(nside, ra, dec, M, a, D, R, R / a, axes, d, eps, eps_model, rdelta, extra)
    return orc.baryonify_shell(nside, map_in, ra, dec, M, a, D, R, R / a, axes, d, eps, eps_model, rdelta, extra)


def assert_maps_close(got, ref, rtol=1e-5, floor=1e-12, max_edge_pixels=0, what=""):
    """<= rtol relative on the non-zero pixels of the reference (north-star tolerance); pixels whose
    reference value is below floor * max|ref| are compared absolutely."""
    got, ref = np.asarray(got).ravel(), np.asarray(ref).ravel()
    assert got.shape == ref.shape
    scale = np.max(np.abs(ref)) if ref.size else 0.0
    err = np.abs(got - ref)
    tol = rtol * np.abs(ref) + floor * scale
    bad = np.where(err > tol)[0]
    assert bad.size <= max_edge_pixels, (f"{what}: {bad.size} pixels differ beyond rtol={rtol}: "
                                         f"first {bad[:5]}, got {got[bad[:5]]}, ref {ref[bad[:5]]}")
    return bad.size
