// bfg_device.hpp -- device-side building blocks shared by the gfx950 kernels:
// HEALPix RING geometry, disc/ring windows, table cells.  CDNA4 only (wave64).
//
// The HEALPix formulas are the published RING-scheme algorithms (Gorski et al.
// 2005; healpix_cxx T_Healpix_Base) that the reference reaches through healpy
// at Runners/HealpixRunner.py:327,330,334,336,357,358,361.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bfg {

constexpr double kPi = 3.141592653589793238462643383279502884197;
constexpr double kTwoPi = 6.283185307179586476925286766559005768394;
constexpr double kHalfPi = 1.570796326794896619231321691639751442099;
constexpr double kInvTwoPi = 1.0 / kTwoPi;
constexpr double kTwoThird = 2.0 / 3.0;
constexpr double kDeg2Rad = kPi / 180.0;

// (sin, cos) of an angle in [0, 2 pi] without libm: Cody-Waite reduction by pi/2, degree-13 / -14 series on
// [-pi/4, pi/4] (<= 2 ulp); libm's version carries a large-argument path that is never needed here
__device__ inline void sincos_2pi(double a, double &s, double &c)
{
    const double qf = rint(a * 0.63661977236758134308);                       // 2 / pi
    const int q = (int)qf;
    double r = fma(qf, -1.57079632679489655800e+00, a);                       // pi/2 hi
    r = fma(qf, -6.12323399573676603587e-17, r);                              // pi/2 lo
    const double r2 = r * r;
    double ps = fma(r2, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma(r2, ps, 2.75573137070700676789e-06);
    ps = fma(r2, ps, -1.98412698298579493134e-04);
    ps = fma(r2, ps, 8.33333333332248946124e-03);
    ps = fma(r2, ps, -1.66666666666666324348e-01);
    const double sn = fma(r * r2, ps, r);
    double pc = fma(r2, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma(r2, pc, -2.75573143513906633035e-07);
    pc = fma(r2, pc, 2.48015872894767294178e-05);
    pc = fma(r2, pc, -1.38888888888741095749e-03);
    pc = fma(r2, pc, 4.16666666666666019037e-02);
    const double cs = fma(r2 * r2, pc, fma(r2, -0.5, 1.0));
    const double s0 = (q & 1) ? cs : sn, c0 = (q & 1) ? sn : cs;
    s = (q & 2) ? -s0 : s0;
    c = ((q + 1) & 2) ? -c0 : c0;
}

// sin / cos of any argument: the lean routine inside [0, 2 pi] (|x| for the even / odd symmetry), libm outside (and for NaN)
__device__ inline void sincos_range(double a, double &s, double &c)
{
    if (a >= 0.0 && a <= kTwoPi) sincos_2pi(a, s, c);
    else sincos(a, &s, &c);
}
__device__ inline double cos_range(double a)
{
    const double x = fabs(a);
    if (x <= kTwoPi) { double s, c; sincos_2pi(x, s, c); return c; }
    return cos(a);
}
__device__ inline double sin_range(double a)
{
    const double x = fabs(a);
    if (x <= kTwoPi) { double s, c; sincos_2pi(x, s, c); return (a < 0.0) ? -s : s; }
    return sin(a);
}

struct Hpx {
    int64_t nside, npix, ncap;
    double fact1, fact2;
};

__host__ __device__ inline Hpx make_hpx(int64_t nside)
{
    Hpx h;
    h.nside = nside;
    h.npix = 12 * nside * nside;
    h.ncap = 2 * nside * (nside - 1);
    h.fact2 = 4.0 / (double)h.npix;
    h.fact1 = (double)(nside << 1) * h.fact2;
    return h;
}

__device__ inline int64_t ring_above(const Hpx &h, double z)
{
    double az = fabs(z);
    if (az <= kTwoThird) return (int64_t)((double)h.nside * (2.0 - 1.5 * z));
    int64_t iring = (int64_t)((double)h.nside * sqrt(3.0 * (1.0 - az)));
    return (z > 0) ? iring : 4 * h.nside - iring - 1;
}

// first pixel, pixel count and half-pixel shift of a ring (1 .. 4 nside - 1)
__device__ inline void ring_info_small(const Hpx &h, int64_t ring, int64_t &startpix, int64_t &ringpix,
                                       bool &shifted)
{
    if (ring < h.nside) {
        shifted = true;
        ringpix = 4 * ring;
        startpix = 2 * ring * (ring - 1);
    } else if (ring < 3 * h.nside) {
        shifted = ((ring - h.nside) & 1) == 0;
        ringpix = 4 * h.nside;
        startpix = h.ncap + (ring - h.nside) * ringpix;
    } else {
        int64_t nr = 4 * h.nside - ring;
        shifted = true;
        ringpix = 4 * nr;
        startpix = h.npix - 2 * nr * (nr + 1);
    }
}

__device__ inline double ring2z(const Hpx &h, int64_t ring)
{
    if (ring < h.nside) return 1.0 - (double)(ring * ring) * h.fact2;
    if (ring <= 3 * h.nside) return (double)(2 * h.nside - ring) * h.fact1;
    ring = 4 * h.nside - ring;
    return (double)(ring * ring) * h.fact2 - 1.0;
}

// Everything the pixel loop needs about one ring: z, sin(theta) (with the
// polar-cap sqrt(tmp(2-tmp)) form of pix2loc), and phi(ip) = (ip + phioff) * phistep.
struct RingGeom {
    int64_t start;
    int32_t nr;
    double z, sth, phistep, phioff;
};

__device__ inline RingGeom ring_geom(const Hpx &h, int64_t ring)
{
    RingGeom g;
    if (ring < h.nside) {
        double tmp = (double)(ring * ring) * h.fact2;
        g.z = 1.0 - tmp;
        g.sth = (g.z > 0.99) ? sqrt(tmp * (2.0 - tmp)) : sqrt((1.0 - g.z) * (1.0 + g.z));
        g.nr = (int32_t)(4 * ring);
        g.start = 2 * ring * (ring - 1);
        g.phistep = kHalfPi / (double)ring;
        g.phioff = 0.5;
    } else if (ring <= 3 * h.nside) {
        g.z = (double)(2 * h.nside - ring) * h.fact1;
        g.sth = sqrt((1.0 - g.z) * (1.0 + g.z));
        g.nr = (int32_t)(4 * h.nside);
        g.start = h.ncap + (ring - h.nside) * (int64_t)g.nr;
        g.phistep = kPi * 0.75 * h.fact1;
        g.phioff = ((ring + h.nside) & 1) ? 0.0 : 0.5;  // iphi - fodd with iphi = ip + 1
    } else {
        int64_t ir = 4 * h.nside - ring;
        double tmp = (double)(ir * ir) * h.fact2;
        g.z = tmp - 1.0;
        g.sth = (g.z < -0.99) ? sqrt(tmp * (2.0 - tmp)) : sqrt((1.0 - g.z) * (1.0 + g.z));
        g.nr = (int32_t)(4 * ir);
        g.start = h.npix - 2 * ir * (ir + 1);
        g.phistep = kHalfPi / (double)ir;
        g.phioff = 0.5;
    }
    return g;
}

__device__ inline int64_t isqrt64(int64_t v)
{
    int64_t r = (int64_t)sqrt((double)v + 0.5);
    while (r * r > v) --r;
    while ((r + 1) * (r + 1) <= v) ++r;
    return r;
}

// RING pixel -> (ring, index in ring)
__device__ inline void pix2ring(const Hpx &h, int64_t pix, int64_t &ring, int64_t &ip)
{
    if (pix < h.ncap) {
        ring = (1 + isqrt64(1 + 2 * pix)) >> 1;
        ip = pix - 2 * ring * (ring - 1);
    } else if (pix < h.npix - h.ncap) {
        int64_t q = pix - h.ncap;
        int64_t nl4 = 4 * h.nside;
        int64_t t = q / nl4;
        ring = t + h.nside;
        ip = q - nl4 * t;
    } else {
        int64_t q = h.npix - pix;
        int64_t ir = (1 + isqrt64(2 * q - 1)) >> 1;
        ring = 4 * h.nside - ir;
        ip = 4 * ir - (q - 2 * ir * (ir - 1));
    }
}

// get_ring_info2: ring colatitude as get_interpol wants it
__device__ inline void ring_info2(const Hpx &h, int64_t ring, int64_t &startpix, int64_t &ringpix,
                                  double &theta, bool &shifted)
{
    int64_t northring = (ring > 2 * h.nside) ? 4 * h.nside - ring : ring;
    if (northring < h.nside) {
        double tmp = (double)(northring * northring) * h.fact2;
        double costheta = 1.0 - tmp;
        double sintheta = sqrt(tmp * (2.0 - tmp));
        theta = atan2(sintheta, costheta);
        ringpix = 4 * northring;
        shifted = true;
        startpix = 2 * northring * (northring - 1);
    } else {
        theta = acos((double)(2 * h.nside - northring) * h.fact1);
        ringpix = 4 * h.nside;
        shifted = ((northring - h.nside) & 1) == 0;
        startpix = h.ncap + (northring - h.nside) * ringpix;
    }
    if (northring != ring) {
        theta = kPi - theta;
        startpix = h.npix - startpix - ringpix;
    }
}

// healpix_cxx get_interpol: the 4 bilinear neighbours and weights of (theta, phi)
__device__ inline void get_interpol(const Hpx &h, double theta, double phi, int64_t pix[4], double wgt[4])
{
    double z = cos(theta);
    int64_t ir1 = ring_above(h, z);
    int64_t ir2 = ir1 + 1;
    double theta1 = 0, theta2 = 0;
    int64_t sp, nr;
    bool shift;
    pix[0] = pix[1] = pix[2] = pix[3] = 0;
    wgt[0] = wgt[1] = wgt[2] = wgt[3] = 0;
    if (ir1 > 0) {
        ring_info2(h, ir1, sp, nr, theta1, shift);
        double dphi = kTwoPi / (double)nr;
        double tmp = (phi / dphi - 0.5 * (shift ? 1.0 : 0.0));
        int64_t i1 = (tmp < 0) ? (int64_t)tmp - 1 : (int64_t)tmp;
        double w1 = (phi - ((double)i1 + 0.5 * (shift ? 1.0 : 0.0)) * dphi) / dphi;
        int64_t i2 = i1 + 1;
        if (i1 < 0) i1 += nr;
        if (i2 >= nr) i2 -= nr;
        pix[0] = sp + i1; pix[1] = sp + i2;
        wgt[0] = 1 - w1; wgt[1] = w1;
    }
    if (ir2 < 4 * h.nside) {
        ring_info2(h, ir2, sp, nr, theta2, shift);
        double dphi = kTwoPi / (double)nr;
        double tmp = (phi / dphi - 0.5 * (shift ? 1.0 : 0.0));
        int64_t i1 = (tmp < 0) ? (int64_t)tmp - 1 : (int64_t)tmp;
        double w1 = (phi - ((double)i1 + 0.5 * (shift ? 1.0 : 0.0)) * dphi) / dphi;
        int64_t i2 = i1 + 1;
        if (i1 < 0) i1 += nr;
        if (i2 >= nr) i2 -= nr;
        pix[2] = sp + i1; pix[3] = sp + i2;
        wgt[2] = 1 - w1; wgt[3] = w1;
    }
    if (ir1 == 0) {
        double wtheta = theta / theta2;
        wgt[2] *= wtheta; wgt[3] *= wtheta;
        double fac = (1 - wtheta) * 0.25;
        wgt[0] = fac; wgt[1] = fac; wgt[2] += fac; wgt[3] += fac;
        pix[0] = (pix[2] + 2) & 3;
        pix[1] = (pix[3] + 2) & 3;
    } else if (ir2 == 4 * h.nside) {
        double wtheta = (theta - theta1) / (kPi - theta1);
        wgt[0] *= 1 - wtheta; wgt[1] *= 1 - wtheta;
        double fac = wtheta * 0.25;
        wgt[0] += fac; wgt[1] += fac; wgt[2] = fac; wgt[3] = fac;
        pix[2] = ((pix[0] + 2) & 3) + h.npix - 4;
        pix[3] = ((pix[1] + 2) & 3) + h.npix - 4;
    } else {
        double wtheta = (theta - theta1) / (theta2 - theta1);
        wgt[0] *= 1 - wtheta; wgt[1] *= 1 - wtheta;
        wgt[2] *= wtheta; wgt[3] *= wtheta;
    }
}

// scipy find_interval_ascending: largest i with g[i] <= x, clipped to [0, n-2]
__device__ inline int find_interval(const double *__restrict__ g, int n, double x)
{
    if (!(x >= g[0])) return 0;
    if (x >= g[n - 1]) return n - 2;
    int lo = 0, hi = n - 1;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (g[mid] <= x) lo = mid; else hi = mid;
    }
    return lo;
}

// The same cell for ANY ascending axis, found from a guess instead of a bisection: i0 = (x - g[0]) * inv_h with inv_h the inverse
// of the axis' MEAN spacing (host side: (n - 1) / (g[n-1] - g[0]); 0 = unknown).  The guess is then moved by the very comparisons
// the bisection makes (g[i] <= x < g[i+1]) -- at most two cells either way, which covers every axis that is a linspace to
// rounding (the D_A knots of HealpixRunner.py:297, the ln M axis of a geomspace mass grid) -- and anything farther off finishes
// with a bisection of the remaining bracket.  Two dependent loads instead of log2(n).
__device__ inline int find_interval_hint(const double *__restrict__ g, int n, double x, double inv_h)
{
    if (!(x >= g[0])) return 0;
    if (x >= g[n - 1]) return n - 2;
    int i = (int)((x - g[0]) * inv_h);
    i = min(max(i, 0), n - 2);
    int lo, hi;
    if (g[i] > x) {                                        // g[0] <= x: i >= 1 here
        --i;
        if (g[i] <= x) return i;
        --i;
        if (g[i] <= x) return i;
        lo = 0; hi = i;
    } else {
        if (x < g[i + 1]) return i;
        ++i;                                               // x < g[n-1]: i + 1 <= n - 1 here
        if (x < g[i + 1]) return i;
        ++i;
        if (x < g[i + 1]) return i;
        lo = i + 1; hi = n - 1;
    }
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (g[mid] <= x) lo = mid; else hi = mid;
    }
    return lo;
}

// inclusive prefix sum over the G consecutive lanes of a group (G = 16 or 64)
template <int G>
__device__ inline int group_inclusive_scan(int v, int lane)
{
#pragma unroll
    for (int d = 1; d < G; d <<= 1) {
        int o = __shfl_up(v, d, G);
        if (lane >= d) v += o;
    }
    return v;
}

// ---- sky tiles of the LDS-privatised variant (bfg_tile.hpp) ----------------------------------
constexpr int kTileWidth = 32;      // TW: max pixels of one ring inside one sector
constexpr int kMaxPairsPerHalo = 64;
constexpr int kLogTab = 128;
constexpr int kExpTab = 64;
constexpr int kAtanTab = 72;       // atan(k / 64), k = 0 .. 64 (padded to a multiple of 8)

struct TileGeom {
    int tr;                      // rings per band
    int tw;                      // max pixels of one ring inside one sector
    int nbands;
    int ntiles;
    const int32_t *band_ns;      // [nbands]   sectors per band
    const int32_t *band_tile0;   // [nbands+1] first tile id of a band
    const int32_t *band_nrmin;   // [nbands]   shortest ring of the band
    const int32_t *tile_band;    // [ntiles]
};


// what the halo -> tile binning needs (count pass inside halo_prep_kernel, fill pass in tile_fill_kernel)
// counters behind the ntiles tile counts of a set: [0] left-over halos, [1] needs_scan, then what the planning call added to the
// statistics (bfg_tile.hpp: plan_reinit_kernel adds them again for a call that reuses the plan)
constexpr int kTileTail = 8, kPlanOob = 2, kPlanWarn = 3, kPlanFallback = 4;

struct BinCtx {
    TileGeom geo;
    int32_t *tile_count;          // [ntiles] counts, then fill cursors
    const int32_t *tile_start;    // [ntiles+1]
    int32_t *pairs;               // [ntiles * cap_direct] fixed slots per tile | [pair_cap] overflow lists
    unsigned long long *pair_total;
    long long pair_cap;           // capacity of the overflow region
    int mode;                     // MODE_PAINT / MODE_BARYONIFY
    int cap_direct;               // slots per tile that the COUNT pass fills directly (see tile_bin_halo)
    unsigned long long *ovf_mask; // [n_halo] bit i: the halo's i-th pair found its tile's slots full
    int32_t *needs_scan;          // set when a tile gets more pairs than direct_limit: the work list then needs the scan kernel
    int direct_limit;             // min(cap_direct, 512): up to here a tile is one plain work item
};

}  // namespace bfg
