"""
Test infrastructure: HEALPix RING pixel centres from the closed-form ring formulae of Gorski et al. 2005,
written with numpy / torch array arithmetic only -- independent of oracle/bfg_oracle.c and of csrc/bfg_device.hpp, which the
tests compare against it.
"""
import numpy as np


def ring_pixel_z_phi(nside, xp=np, with_tmp=False):
    """(z, phi) of every RING pixel centre, float64 arrays of length 12 nside^2 (xp = numpy or torch); with_tmp: also 1 - |z|
    without cancellation (i^2 / 3 nside^2 in the caps), from which sin(theta) = sqrt(tmp (2 - tmp)) keeps its digits at the poles"""
    npix = 12 * nside * nside
    ncap = 2 * nside * (nside - 1)
    if xp is np:
        p = np.arange(npix, dtype=np.int64)
        f = lambda a: a.astype(np.float64)
        i64 = lambda a: a.astype(np.int64)
        where, sqrt, empty = np.where, np.sqrt, lambda: np.empty(npix)
    else:
        import torch
        dev = "cuda" if torch.cuda.is_available() else "cpu"
        p = torch.arange(npix, dtype=torch.int64, device=dev)
        f = lambda a: a.to(torch.float64)
        i64 = lambda a: a.to(torch.int64)
        where, sqrt = torch.where, torch.sqrt
        empty = lambda: torch.empty(npix, dtype=torch.float64, device=dev)

    def cap_ring(q):                       # ring index i >= 1 of cap pixel q (counted from the pole): 2 i (i - 1) <= q < 2 i (i + 1)
        i = i64((1 + sqrt(1 + 2 * f(q))) / 2)
        i = where(2 * i * (i - 1) > q, i - 1, i)
        i = where(2 * (i + 1) * i <= q, i + 1, i)
        return i
    z, phi, tmp = empty(), empty(), empty()
    north = p < ncap
    q = p[north]
    i = cap_ring(q)
    j = q - 2 * i * (i - 1)
    tmp[north] = f(i * i) / (3.0 * nside * nside)
    z[north] = 1 - tmp[north]
    phi[north] = (f(j) + 0.5) * (np.pi / 2) / f(i)
    belt = (p >= ncap) & (p < npix - ncap)
    q = p[belt] - ncap
    i = q // (4 * nside) + nside
    j = q % (4 * nside)
    z[belt] = f(2 * nside - i) * 2.0 / (3.0 * nside)
    tmp[belt] = 1 - abs(z[belt])
    phi[belt] = (f(j) + 0.5 * f(((i - nside) & 1) == 0)) * (np.pi / 2) / nside
    south = p >= npix - ncap
    q = npix - 1 - p[south]                 # the south cap mirrors the north cap
    i = cap_ring(q)
    j = 4 * i - 1 - (q - 2 * i * (i - 1))
    tmp[south] = f(i * i) / (3.0 * nside * nside)
    z[south] = -(1 - tmp[south])
    phi[south] = (f(j) + 0.5) * (np.pi / 2) / f(i)
    return (z, phi, tmp) if with_tmp else (z, phi)


def ring_pixel_vectors(nside, xp=np):
    z, phi, tmp = ring_pixel_z_phi(nside, xp, with_tmp=True)
    if xp is np:
        s = np.sqrt(tmp * (2 - tmp))
        return np.stack([s * np.cos(phi), s * np.sin(phi), z], axis=1)
    import torch
    s = torch.sqrt(tmp * (2 - tmp))
    return torch.stack([s * torch.cos(phi), s * torch.sin(phi), z], dim=1)


def ring_theta(nside):
    """colatitude of ring 1 .. 4 nside - 1"""
    i = np.arange(1, 4 * nside, dtype=np.float64)
    z = np.where(i < nside, 1 - i * i / (3.0 * nside * nside),
                 np.where(i <= 3 * nside, (2 * nside - i) * 2.0 / (3.0 * nside), -(1 - (4 * nside - i) ** 2 / (3.0 * nside * nside))))
    return np.arccos(z)
