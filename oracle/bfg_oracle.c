/*
 * bfg_oracle.c -- CPU ORACLE for the BaryonForge shell paint / baryonify hot path.
 *
 * THIS FILE IS TEST INFRASTRUCTURE.  It is a deliberately plain, serial C
 * restatement of the reference algorithm and is only ever used as the checker
 * (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).  The product
 * path (baryonforge_amd/ + libbfg_mi355.so) never links, imports or calls it.
 *
 * What it restates (citations relative to /root/reference):
 *   - Runners/HealpixRunner.py:449-481   PaintProfilesShell.process loop body
 *   - Runners/HealpixRunner.py:315-355   BaryonifyShell.process loop body
 *   - Runners/HealpixRunner.py:357-365   displaced-pixel regrid (+ :17-71 regrid_pixels_hpix)
 *   - utils/Tabulate.py:305-319, :621-650  (Param)TabulatedProfile._readout
 *       = exp( scipy RegularGridInterpolator(method='linear', bounds_error=False,
 *              fill_value=nan) ) on axes (ln(1+z), ln M, ln r[, extras])
 *   - Profiles/BaryonCorrection.py:331-419  BaryonificationClass._readout
 *       (linear table, optional ln r - ln R_delta axis, zero beyond eps*R)
 *
 * Third-party arithmetic that is NOT under /root/reference and is restated
 * from the published algorithms (healpy is unpinned in pyproject.toml:19):
 *   - HEALPix RING scheme, healpix_cxx T_Healpix_Base: ring_above, ring2z,
 *     get_ring_info_small/2, pix2loc, query_disc_internal (fact = 0, i.e.
 *     healpy query_disc(inclusive=False, nest=False)), get_interpol;
 *     healpy.ang2vec / vec2ang / lonlat conversions (Gorski et al. 2005).
 *   - scipy RegularGridInterpolator._evaluate_linear + find_indices
 *     (scipy is importable in the build container, so this one is pinned
 *     bit-for-bit by tests/test_oracle_*.py against the real scipy).
 *
 * PARITY STATUS: the loop glue and the table read-out are pinned by golden
 * vectors generated from the reference's own modules (tests/golden/);
 * HEALPix geometry is pinned only by analytic known answers and by the
 * healpy doc-string examples recalled in tests/test_oracle_healpix.py
 * ("parity unpinned" against a live healpy: none is installed here).  The
 * background (oracle.py) is pinned against LIVE pyccl by the two comoving-distance
 * differences the reference's example notebooks store as printed cell outputs
 * (tests/golden/pyccl_notebook_outputs.json; agreement 1.4e-7).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#define ORC_PI      3.141592653589793238462643383279502884197
#define ORC_TWOPI   6.283185307179586476925286766559005768394
#define ORC_HALFPI  1.570796326794896619231321691639751442099
#define ORC_INV_TWOPI (1.0 / ORC_TWOPI)
#define ORC_TWOTHIRD (2.0 / 3.0)
#define ORC_DEG2RAD (ORC_PI / 180.0)
#define ORC_RAD2DEG (180.0 / ORC_PI)
#define ORC_MAXDIM 8

typedef struct {
    int64_t nside, npix, ncap;
    double fact1, fact2;
} hpx_t;

static hpx_t hpx_make(int64_t nside)
{
    hpx_t h;
    h.nside = nside;
    h.npix = 12 * nside * nside;
    h.ncap = 2 * nside * (nside - 1);
    h.fact2 = 4.0 / (double)h.npix;
    h.fact1 = (double)(nside << 1) * h.fact2;
    return h;
}

/* healpix_cxx T_Healpix_Base::ring_above */
static int64_t hpx_ring_above(const hpx_t *h, double z)
{
    double az = fabs(z);
    if (az <= ORC_TWOTHIRD)
        return (int64_t)((double)h->nside * (2.0 - 1.5 * z));
    int64_t iring = (int64_t)((double)h->nside * sqrt(3.0 * (1.0 - az)));
    return (z > 0) ? iring : 4 * h->nside - iring - 1;
}

/* healpix_cxx get_ring_info_small */
static void hpx_ring_info_small(const hpx_t *h, int64_t ring, int64_t *startpix,
                                int64_t *ringpix, int *shifted)
{
    if (ring < h->nside) {
        *shifted = 1;
        *ringpix = 4 * ring;
        *startpix = 2 * ring * (ring - 1);
    } else if (ring < 3 * h->nside) {
        *shifted = ((ring - h->nside) & 1) == 0;
        *ringpix = 4 * h->nside;
        *startpix = h->ncap + (ring - h->nside) * (*ringpix);
    } else {
        int64_t nr = 4 * h->nside - ring;
        *shifted = 1;
        *ringpix = 4 * nr;
        *startpix = h->npix - 2 * nr * (nr + 1);
    }
}

/* healpix_cxx ring2z */
static double hpx_ring2z(const hpx_t *h, int64_t ring)
{
    if (ring < h->nside)
        return 1.0 - (double)(ring * ring) * h->fact2;
    if (ring <= 3 * h->nside)
        return (double)(2 * h->nside - ring) * h->fact1;
    ring = 4 * h->nside - ring;
    return (double)(ring * ring) * h->fact2 - 1.0;
}

/* healpix_cxx get_ring_info2: also the colatitude of the ring */
static void hpx_ring_info2(const hpx_t *h, int64_t ring, int64_t *startpix,
                           int64_t *ringpix, double *theta, int *shifted)
{
    int64_t northring = (ring > 2 * h->nside) ? 4 * h->nside - ring : ring;
    if (northring < h->nside) {
        double tmp = (double)(northring * northring) * h->fact2;
        double costheta = 1.0 - tmp;
        double sintheta = sqrt(tmp * (2.0 - tmp));
        *theta = atan2(sintheta, costheta);
        *ringpix = 4 * northring;
        *shifted = 1;
        *startpix = 2 * northring * (northring - 1);
    } else {
        *theta = acos((double)(2 * h->nside - northring) * h->fact1);
        *ringpix = 4 * h->nside;
        *shifted = ((northring - h->nside) & 1) == 0;
        *startpix = h->ncap + (northring - h->nside) * (*ringpix);
    }
    if (northring != ring) { /* southern hemisphere */
        *theta = ORC_PI - *theta;
        *startpix = h->npix - *startpix - *ringpix;
    }
}

static int64_t isqrt64(int64_t v)
{
    int64_t r = (int64_t)sqrt((double)v + 0.5);
    while (r * r > v) --r;
    while ((r + 1) * (r + 1) <= v) ++r;
    return r;
}

/* healpix_cxx pix2loc (RING) followed by the vec3 construction of pix2vec */
static void hpx_pix2vec_ring(const hpx_t *h, int64_t pix, double *v)
{
    double z, phi, sth;
    int have_sth = 0;
    if (pix < h->ncap) {
        int64_t iring = (1 + isqrt64(1 + 2 * pix)) >> 1;
        int64_t iphi = (pix + 1) - 2 * iring * (iring - 1);
        double tmp = (double)(iring * iring) * h->fact2;
        z = 1.0 - tmp;
        if (z > 0.99) { sth = sqrt(tmp * (2.0 - tmp)); have_sth = 1; }
        phi = ((double)iphi - 0.5) * ORC_HALFPI / (double)iring;
    } else if (pix < h->npix - h->ncap) {
        int64_t nl4 = 4 * h->nside;
        int64_t ip = pix - h->ncap;
        int64_t tmp = ip / nl4;
        int64_t iring = tmp + h->nside;
        int64_t iphi = ip - nl4 * tmp + 1;
        double fodd = ((iring + h->nside) & 1) ? 1.0 : 0.5;
        z = (double)(2 * h->nside - iring) * h->fact1;
        phi = ((double)iphi - fodd) * ORC_PI * 0.75 * h->fact1;
    } else {
        int64_t ip = h->npix - pix;
        int64_t iring = (1 + isqrt64(2 * ip - 1)) >> 1;
        int64_t iphi = 4 * iring + 1 - (ip - 2 * iring * (iring - 1));
        double tmp = (double)(iring * iring) * h->fact2;
        z = tmp - 1.0;
        if (z < -0.99) { sth = sqrt(tmp * (2.0 - tmp)); have_sth = 1; }
        phi = ((double)iphi - 0.5) * ORC_HALFPI / (double)iring;
    }
    if (!have_sth) sth = sqrt((1.0 - z) * (1.0 + z));
    v[0] = sth * cos(phi);
    v[1] = sth * sin(phi);
    v[2] = z;
}

/* healpy.ang2vec(lon, lat, lonlat=True): theta = pi/2 - radians(lat), phi = radians(lon) */
static void hpx_ang2vec_lonlat(double lon_deg, double lat_deg, double *v)
{
    double theta = ORC_HALFPI - lat_deg * ORC_DEG2RAD;
    double phi = lon_deg * ORC_DEG2RAD;
    double st = sin(theta);
    v[0] = st * cos(phi);
    v[1] = st * sin(phi);
    v[2] = cos(theta);
}

/* healpix_cxx query_disc_internal, RING scheme, fact = 0 (non-inclusive).
 * Emits pixels in ascending order.  Returns the number of pixels in the disc;
 * writes at most cap of them to out (out may be NULL to count only). */
static int64_t hpx_query_disc(const hpx_t *h, const double *vec, double radius,
                              int64_t *out, int64_t cap)
{
    int64_t n = 0;
#define EMIT_RANGE(a, b)                                                     \
    do {                                                                     \
        for (int64_t q_ = (a); q_ < (b); ++q_) {                             \
            if (out && n < cap) out[n] = q_;                                 \
            ++n;                                                             \
        }                                                                    \
    } while (0)

    /* pointing(vec3): theta = atan2(sqrt(x^2+y^2), z), phi = atan2(y, x) in [0, 2pi) */
    double ptheta = atan2(sqrt(vec[0] * vec[0] + vec[1] * vec[1]), vec[2]);
    double pphi = (vec[0] == 0.0 && vec[1] == 0.0) ? 0.0 : atan2(vec[1], vec[0]);
    if (pphi < 0.0) pphi += ORC_TWOPI;

    double rsmall = radius, rbig = radius;
    if (rsmall >= ORC_PI) { EMIT_RANGE(0, h->npix); return n; }
    if (rbig > ORC_PI) rbig = ORC_PI;

    double cosrbig = cos(rbig);
    double z0 = cos(ptheta);
    double xa = 1.0 / sqrt((1.0 - z0) * (1.0 + z0));

    double rlat1 = ptheta - rsmall;
    double zmax = cos(rlat1);
    int64_t irmin = hpx_ring_above(h, zmax) + 1;

    if (rlat1 <= 0 && irmin > 1) { /* north pole in the disc */
        int64_t sp, rp; int dummy;
        hpx_ring_info_small(h, irmin - 1, &sp, &rp, &dummy);
        EMIT_RANGE(0, sp + rp);
    }

    double rlat2 = ptheta + rsmall;
    double zmin = cos(rlat2);
    int64_t irmax = hpx_ring_above(h, zmin);

    for (int64_t iring = irmin; iring <= irmax; ++iring) {
        double z = hpx_ring2z(h, iring);
        double x = (cosrbig - z * z0) * xa;
        double ysq = 1.0 - z * z - x * x;
        double dphi = (ysq <= 0.0) ? 0.0 : atan2(sqrt(ysq), x);
        if (dphi > 0.0) {
            int64_t nr, ipix1; int shifted;
            hpx_ring_info_small(h, iring, &ipix1, &nr, &shifted);
            double shift = shifted ? 0.5 : 0.0;
            int64_t ipix2 = ipix1 + nr - 1;
            int64_t ip_lo = (int64_t)floor((double)nr * ORC_INV_TWOPI * (pphi - dphi) - shift) + 1;
            int64_t ip_hi = (int64_t)floor((double)nr * ORC_INV_TWOPI * (pphi + dphi) - shift);
            if (ip_hi >= nr) { ip_lo -= nr; ip_hi -= nr; }
            if (ip_lo < 0) {
                EMIT_RANGE(ipix1, ipix1 + ip_hi + 1);
                EMIT_RANGE(ipix1 + ip_lo + nr, ipix2 + 1);
            } else {
                EMIT_RANGE(ipix1 + ip_lo, ipix1 + ip_hi + 1);
            }
        }
    }
    if (rlat2 >= ORC_PI && irmax + 1 < 4 * h->nside) { /* south pole in the disc */
        int64_t sp, rp; int dummy;
        hpx_ring_info_small(h, irmax + 1, &sp, &rp, &dummy);
        EMIT_RANGE(sp, h->npix);
    }
#undef EMIT_RANGE
    return n;
}

/* healpix_cxx get_interpol for pointing (theta, phi) */
static void hpx_get_interpol(const hpx_t *h, double theta, double phi,
                             int64_t *pix, double *wgt)
{
    double z = cos(theta);
    int64_t ir1 = hpx_ring_above(h, z);
    int64_t ir2 = ir1 + 1;
    double theta1 = 0, theta2 = 0, w1, tmp, dphi;
    int64_t sp, nr, i1, i2;
    int shift;
    if (ir1 > 0) {
        hpx_ring_info2(h, ir1, &sp, &nr, &theta1, &shift);
        dphi = ORC_TWOPI / (double)nr;
        tmp = (phi / dphi - 0.5 * shift);
        i1 = (tmp < 0) ? (int64_t)tmp - 1 : (int64_t)tmp;
        w1 = (phi - ((double)i1 + 0.5 * shift) * dphi) / dphi;
        i2 = i1 + 1;
        if (i1 < 0) i1 += nr;
        if (i2 >= nr) i2 -= nr;
        pix[0] = sp + i1; pix[1] = sp + i2;
        wgt[0] = 1 - w1; wgt[1] = w1;
    }
    if (ir2 < 4 * h->nside) {
        hpx_ring_info2(h, ir2, &sp, &nr, &theta2, &shift);
        dphi = ORC_TWOPI / (double)nr;
        tmp = (phi / dphi - 0.5 * shift);
        i1 = (tmp < 0) ? (int64_t)tmp - 1 : (int64_t)tmp;
        w1 = (phi - ((double)i1 + 0.5 * shift) * dphi) / dphi;
        i2 = i1 + 1;
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
    } else if (ir2 == 4 * h->nside) {
        double wtheta = (theta - theta1) / (ORC_PI - theta1);
        wgt[0] *= 1 - wtheta; wgt[1] *= 1 - wtheta;
        double fac = wtheta * 0.25;
        wgt[0] += fac; wgt[1] += fac; wgt[2] = fac; wgt[3] = fac;
        pix[2] = ((pix[0] + 2) & 3) + h->npix - 4;
        pix[3] = ((pix[1] + 2) & 3) + h->npix - 4;
    } else {
        double wtheta = (theta - theta1) / (theta2 - theta1);
        wgt[0] *= 1 - wtheta; wgt[1] *= 1 - wtheta;
        wgt[2] *= wtheta; wgt[3] *= wtheta;
    }
}

/* healpy.get_interp_weights(nside, lon, lat, lonlat=True) */
static void hpx_get_interp_weights_lonlat(const hpx_t *h, double lon_deg, double lat_deg,
                                          int64_t *pix, double *wgt)
{
    double theta = ORC_HALFPI - lat_deg * ORC_DEG2RAD;
    double phi = lon_deg * ORC_DEG2RAD;
    hpx_get_interpol(h, theta, phi, pix, wgt);
}

/* healpy.vec2ang(v, lonlat=True) */
static void hpx_vec2ang_lonlat(const double *v, double *lon_deg, double *lat_deg)
{
    double dnorm = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    double theta = acos(v[2] / dnorm);
    double phi = atan2(v[1], v[0]);
    if (phi < 0) phi += ORC_TWOPI;
    *lon_deg = phi * ORC_RAD2DEG;
    *lat_deg = 90.0 - theta * ORC_RAD2DEG;
}

/* ------------------------------------------------------------------ */
/* exported HEALPix entry points (arrays)                               */
/* ------------------------------------------------------------------ */
int64_t orc_nside2npix(int64_t nside) { return 12 * nside * nside; }
double orc_nside2pixarea(int64_t nside) { return 4.0 * ORC_PI / (double)(12 * nside * nside); }

int64_t orc_ring_above(int64_t nside, double z)
{
    hpx_t h = hpx_make(nside);
    return hpx_ring_above(&h, z);
}

void orc_ang2vec_lonlat(int64_t n, const double *lon, const double *lat, double *out)
{
    for (int64_t i = 0; i < n; ++i) hpx_ang2vec_lonlat(lon[i], lat[i], out + 3 * i);
}

void orc_pix2vec_ring(int64_t nside, int64_t n, const int64_t *pix, double *out)
{
    hpx_t h = hpx_make(nside);
    for (int64_t i = 0; i < n; ++i) hpx_pix2vec_ring(&h, pix[i], out + 3 * i);
}

int64_t orc_query_disc_ring(int64_t nside, const double *vec, double radius,
                            int64_t *out, int64_t cap)
{
    hpx_t h = hpx_make(nside);
    return hpx_query_disc(&h, vec, radius, out, cap);
}

/* pix, wgt laid out [n][4] (healpy returns the transpose, shape (4, n)) */
void orc_get_interp_weights_lonlat(int64_t nside, int64_t n, const double *lon,
                                   const double *lat, int64_t *pix, double *wgt)
{
    hpx_t h = hpx_make(nside);
    for (int64_t i = 0; i < n; ++i)
        hpx_get_interp_weights_lonlat(&h, lon[i], lat[i], pix + 4 * i, wgt + 4 * i);
}

void orc_get_interp_weights_thetaphi(int64_t nside, int64_t n, const double *theta,
                                     const double *phi, int64_t *pix, double *wgt)
{
    hpx_t h = hpx_make(nside);
    for (int64_t i = 0; i < n; ++i)
        hpx_get_interpol(&h, theta[i], phi[i], pix + 4 * i, wgt + 4 * i);
}

void orc_vec2ang_lonlat(int64_t n, const double *vec, double *lon, double *lat)
{
    for (int64_t i = 0; i < n; ++i) hpx_vec2ang_lonlat(vec + 3 * i, lon + i, lat + i);
}

/* ------------------------------------------------------------------ */
/* N-linear table read-out = scipy RegularGridInterpolator(linear)      */
/* ------------------------------------------------------------------ */
typedef struct {
    int ndim;
    int64_t shape[ORC_MAXDIM];
    int64_t stride[ORC_MAXDIM];
    const double *axes[ORC_MAXDIM];
    const double *values;
} orc_table_t;

static void table_init(orc_table_t *t, int ndim, const int64_t *shape,
                       const double *axes_concat, const double *values)
{
    t->ndim = ndim;
    const double *p = axes_concat;
    for (int d = 0; d < ndim; ++d) {
        t->shape[d] = shape[d];
        t->axes[d] = p;
        p += shape[d];
    }
    int64_t s = 1;
    for (int d = ndim - 1; d >= 0; --d) { t->stride[d] = s; s *= shape[d]; }
    t->values = values;
}

/* scipy find_indices / find_interval_ascending: largest i with grid[i] <= x,
 * clipped to [0, n-2]; NaN -> falls through (result is NaN anyway). */
static int64_t find_interval(const double *g, int64_t n, double x)
{
    if (!(x >= g[0])) return 0;
    if (x >= g[n - 1]) return n - 2;
    int64_t lo = 0, hi = n - 1; /* g[lo] <= x < g[hi] */
    while (hi - lo > 1) {
        int64_t mid = (lo + hi) >> 1;
        if (g[mid] <= x) lo = mid; else hi = mid;
    }
    return lo;
}

/* scipy RegularGridInterpolator.__call__ (method='linear', bounds_error=False,
 * fill_value=nan) -> _evaluate_linear: hypercube corners in itertools.product
 * order, weight = ((1*w_0)*w_1)*..., value += V[corner]*weight. */
static double table_eval(const orc_table_t *t, const double *x)
{
    int64_t idx[ORC_MAXDIM];
    double yi[ORC_MAXDIM];
    int oob = 0;
    for (int d = 0; d < t->ndim; ++d) {
        const double *g = t->axes[d];
        int64_t n = t->shape[d];
        if (x[d] < g[0] || x[d] > g[n - 1]) oob = 1;
        int64_t i = find_interval(g, n, x[d]);
        idx[d] = i;
        yi[d] = (x[d] - g[i]) / (g[i + 1] - g[i]);
    }
    double value = 0.0;
    int ncorner = 1 << t->ndim;
    for (int c = 0; c < ncorner; ++c) {
        double weight = 1.0;
        int64_t off = 0;
        for (int d = 0; d < t->ndim; ++d) {
            int bit = (c >> (t->ndim - 1 - d)) & 1; /* product(): last dim fastest */
            weight = weight * (bit ? yi[d] : 1.0 - yi[d]);
            off += (idx[d] + bit) * t->stride[d];
        }
        value = value + t->values[off] * weight;
    }
    if (oob) return NAN;
    return value;
}

void orc_interp_linear(int ndim, const int64_t *shape, const double *axes_concat,
                       const double *values, int64_t npts, const double *coords,
                       double *out)
{
    orc_table_t t;
    table_init(&t, ndim, shape, axes_concat, values);
    for (int64_t i = 0; i < npts; ++i) out[i] = table_eval(&t, coords + (int64_t)ndim * i);
}

/* ------------------------------------------------------------------ */
/* Runner loops                                                          */
/* ------------------------------------------------------------------ */
typedef struct {
    int64_t nside;
    int64_t n_halo;
    const double *ra, *dec;   /* degrees                              */
    const double *M;          /* Msun                                  */
    const double *a;          /* scale factor 1/(1+z)                  */
    const double *D;          /* angular diameter distance, phys Mpc   */
    const double *R;          /* halo radius (runner mass_def), phys Mpc */
    const double *extra;      /* [n_halo][n_extra] extra table coords  */
    int n_extra;
    double eps_run;
    orc_table_t table;
} orc_job_t;

/* HealpixRunner.py:449-481.  new_map must be zero-initialised by the caller
 * (":424"); returns P_tot = sum_j |disc_j|. */
static int64_t paint_range(const orc_job_t *job, int64_t j0, int64_t j1,
                           int include_pixel_size, double *new_map)
{
    hpx_t h = hpx_make(job->nside);
    double pixarea = 4.0 * ORC_PI / (double)h.npix;
    int64_t cap = 1024, ptot = 0;
    int64_t *pixind = (int64_t *)malloc(sizeof(int64_t) * cap);
    double x[ORC_MAXDIM];
    for (int64_t j = j0; j < j1; ++j) {
        double M_j = job->M[j], a_j = job->a[j], R_j = job->R[j], D_j = job->D[j];
        double vec_j[3];
        hpx_ang2vec_lonlat(job->ra[j], job->dec[j], vec_j);            /* :458-460 */
        double radius = R_j * job->eps_run / D_j;                        /* :462 */
        int64_t np = hpx_query_disc(&h, vec_j, radius, pixind, cap);     /* :463 */
        if (np > cap) {
            cap = np + np / 2;
            pixind = (int64_t *)realloc(pixind, sizeof(int64_t) * cap);
            np = hpx_query_disc(&h, vec_j, radius, pixind, cap);
        }
        ptot += np;
        double pos_j[3] = { vec_j[0] * D_j, vec_j[1] * D_j, vec_j[2] * D_j }; /* :466 */
        x[0] = log(1.0 / a_j);                                           /* Tabulate.py:308 */
        x[1] = log(M_j);                                                 /* :312 */
        for (int e = 0; e < job->n_extra; ++e) x[3 + e] = job->extra[j * job->n_extra + e];
        for (int64_t k = 0; k < np; ++k) {
            double vec[3];
            hpx_pix2vec_ring(&h, pixind[k], vec);                        /* :464 */
            double d0 = vec[0] * D_j - pos_j[0];                         /* :467-468 */
            double d1 = vec[1] * D_j - pos_j[1];
            double d2 = vec[2] * D_j - pos_j[2];
            double r_sep = sqrt(d0 * d0 + d1 * d1 + d2 * d2);            /* :469 */
            x[2] = log(r_sep / a_j);                                     /* :472, Tabulate.py:309 */
            double v = exp(table_eval(&job->table, x));                  /* Tabulate.py:314-315 */
            if (!isfinite(v)) v = 0.0;                                   /* :473 */
            if (include_pixel_size) v = v * (pixarea * (D_j * D_j));     /* :478 */
            new_map[pixind[k]] += v;                                     /* :481 */
        }
    }
    free(pixind);
    return ptot;
}

static void job_init(orc_job_t *job, int64_t nside, int64_t n_halo, const double *ra,
                     const double *dec, const double *M, const double *a, const double *D,
                     const double *R, const double *extra, int n_extra, double eps_run,
                     int ndim, const int64_t *shape, const double *axes_concat,
                     const double *values)
{
    job->nside = nside; job->n_halo = n_halo;
    job->ra = ra; job->dec = dec; job->M = M; job->a = a; job->D = D; job->R = R;
    job->extra = extra; job->n_extra = n_extra; job->eps_run = eps_run;
    table_init(&job->table, ndim, shape, axes_concat, values);
}

int64_t orc_paint_shell(int64_t nside, int64_t n_halo, const double *ra, const double *dec,
                        const double *M, const double *a, const double *D, const double *R,
                        const double *extra, int n_extra, double eps_run,
                        int include_pixel_size, int ndim, const int64_t *shape,
                        const double *axes_concat, const double *values, double *new_map)
{
    orc_job_t job;
    job_init(&job, nside, n_halo, ra, dec, M, a, D, R, extra, n_extra, eps_run, ndim, shape,
             axes_concat, values);
    return paint_range(&job, 0, n_halo, include_pixel_size, new_map);
}

/* utils/Parallelize.py:218-275, :297-320 (SplitJoinParallel): the catalog is
 * cut into njobs contiguous slices of ceil(N/njobs) halos, each slice painted
 * on its own zero map by its own worker, and the maps summed by the parent.
 * (The seed-42 shuffle of :255 is applied by the caller.)  Threads stand in
 * for loky processes.  include_pixel_size is NOT forwarded (:271) -> 0. */
typedef struct {
    const orc_job_t *job; int64_t j0, j1; double *map; int64_t ptot;
} orc_worker_t;

static void *paint_worker(void *arg)
{
    orc_worker_t *w = (orc_worker_t *)arg;
    /* the worker owns its zero map (empty_shell of Parallelize.py:257); touching it here keeps
     * the page faults inside the worker, as they are in the reference's separate processes */
    memset(w->map, 0, sizeof(double) * (size_t)(12 * w->job->nside * w->job->nside));
    w->ptot = paint_range(w->job, w->j0, w->j1, 0, w->map);
    return NULL;
}

int64_t orc_paint_shell_splitjoin(int njobs, int64_t nside, int64_t n_halo, const double *ra,
                                  const double *dec, const double *M, const double *a,
                                  const double *D, const double *R, const double *extra,
                                  int n_extra, double eps_run, int ndim, const int64_t *shape,
                                  const double *axes_concat, const double *values,
                                  double *map_out)
{
    orc_job_t job;
    job_init(&job, nside, n_halo, ra, dec, M, a, D, R, extra, n_extra, eps_run, ndim, shape,
             axes_concat, values);
    int64_t npix = 12 * nside * nside;
    int64_t per = (n_halo + njobs - 1) / njobs;
    orc_worker_t *w = (orc_worker_t *)calloc((size_t)njobs, sizeof(orc_worker_t));
    pthread_t *th = (pthread_t *)calloc((size_t)njobs, sizeof(pthread_t));
    for (int i = 0; i < njobs; ++i) {
        w[i].job = &job;
        w[i].j0 = (int64_t)i * per; if (w[i].j0 > n_halo) w[i].j0 = n_halo;
        w[i].j1 = (int64_t)(i + 1) * per; if (w[i].j1 > n_halo) w[i].j1 = n_halo;
        w[i].map = (double *)malloc((size_t)npix * sizeof(double));
        pthread_create(&th[i], NULL, paint_worker, &w[i]);
    }
    int64_t ptot = 0;
    memset(map_out, 0, sizeof(double) * (size_t)npix);
    for (int i = 0; i < njobs; ++i) {
        pthread_join(th[i], NULL);
        ptot += w[i].ptot;
        for (int64_t p = 0; p < npix; ++p) map_out[p] += w[i].map[p];  /* np.sum(outputs, axis=0) :318 */
        free(w[i].map);
    }
    free(w); free(th);
    return ptot;
}

/* HealpixRunner.py:315-355 with BaryonCorrection.py:331-419 inlined.
 * The table holds the LINEAR displacement d (comoving Mpc), fill_value = nan.
 * R_model_com[j] = model.mass_def.get_radius(model.cosmo, M, a)/a  (:399).
 * pix_offsets is [npix][3], accumulated in place.  Returns P_tot. */
int64_t orc_baryonify_offsets(int64_t nside, int64_t n_halo, const double *ra, const double *dec,
                              const double *M, const double *a, const double *D, const double *R,
                              const double *R_model_com, const double *extra, int n_extra,
                              double eps_run, double eps_model, int rdelta_sampling, int ndim,
                              const int64_t *shape, const double *axes_concat,
                              const double *values, double *pix_offsets)
{
    orc_job_t job;
    job_init(&job, nside, n_halo, ra, dec, M, a, D, R, extra, n_extra, eps_run, ndim, shape,
             axes_concat, values);
    hpx_t h = hpx_make(nside);
    int64_t cap = 1024, ptot = 0;
    int64_t *pixind = (int64_t *)malloc(sizeof(int64_t) * cap);
    double x[ORC_MAXDIM];
    for (int64_t j = 0; j < n_halo; ++j) {
        double M_j = M[j], a_j = a[j], R_j = R[j], D_j = D[j];
        double vec_j[3];
        hpx_ang2vec_lonlat(ra[j], dec[j], vec_j);                        /* :325-327 */
        double radius = R_j * eps_run / D_j;                             /* :329 */
        int64_t np = hpx_query_disc(&h, vec_j, radius, pixind, cap);     /* :330 */
        if (np > cap) {
            cap = np + np / 2;
            pixind = (int64_t *)realloc(pixind, sizeof(int64_t) * cap);
            np = hpx_query_disc(&h, vec_j, radius, pixind, cap);
        }
        if (np < 4) {                                                    /* :333-334 */
            double wdummy[4];
            hpx_get_interp_weights_lonlat(&h, ra[j], dec[j], pixind, wdummy);
            np = 4;
        }
        ptot += np;
        double pos_j[3] = { vec_j[0] * D_j, vec_j[1] * D_j, vec_j[2] * D_j }; /* :338 */
        double Rm = R_model_com[j];
        x[0] = log(1.0 / a_j);                                           /* BaryonCorrection.py:367 */
        x[1] = log(M_j);                                                 /* :397 */
        for (int e = 0; e < n_extra; ++e) x[3 + e] = extra[j * n_extra + e];
        for (int64_t k = 0; k < np; ++k) {
            double vec[3], pos[3], diff[3];
            hpx_pix2vec_ring(&h, pixind[k], vec);                        /* :336 */
            for (int c = 0; c < 3; ++c) { pos[c] = vec[c] * D_j; diff[c] = pos[c] - pos_j[c]; } /* :339-340 */
            double r_sep = sqrt(diff[0] * diff[0] + diff[1] * diff[1] + diff[2] * diff[2]); /* :341 */
            double r_com = r_sep / a_j;                                  /* :345 */
            double lr = log(r_com);                                      /* BaryonCorrection.py:368 */
            x[2] = rdelta_sampling ? lr - log(Rm) : lr;                  /* :403-408 */
            double d = table_eval(&job.table, x);
            if (!(r_com < eps_model * Rm)) d = 0.0;                      /* :410-411 */
            d = d * a_j;                                                 /* HealpixRunner.py:345 */
            double off[3], nw[3];
            for (int c = 0; c < 3; ++c) {
                off[c] = d * (diff[c] / r_sep);                          /* :346 */
                if (!isfinite(off[c])) off[c] = 0.0;                     /* :347 */
                nw[c] = pos[c] + off[c];                                 /* :350 */
            }
            double nrm = sqrt(nw[0] * nw[0] + nw[1] * nw[1] + nw[2] * nw[2]); /* :351 */
            for (int c = 0; c < 3; ++c)
                pix_offsets[3 * pixind[k] + c] += nw[c] / nrm - vec[c];  /* :352, :355 */
        }
    }
    free(pixind);
    return ptot;
}

/* HealpixRunner.py:357-365 + regrid_pixels_hpix :17-71.  new_map zeroed by caller. */
void orc_regrid_shell(int64_t nside, const double *pix_offsets, const double *orig_map,
                      double *new_map)
{
    hpx_t h = hpx_make(nside);
    for (int64_t p = 0; p < h.npix; ++p) {
        if (orig_map[p] == 0.0) continue;                                /* :359 */
        double v[3], lon, lat, w[4];
        int64_t c[4];
        hpx_pix2vec_ring(&h, p, v);                                      /* :357 */
        for (int k = 0; k < 3; ++k) v[k] += pix_offsets[3 * p + k];
        hpx_vec2ang_lonlat(v, &lon, &lat);                               /* :358 */
        hpx_get_interp_weights_lonlat(&h, lon, lat, c, w);               /* :361 */
        for (int k = 0; k < 4; ++k) new_map[c[k]] += w[k] * orig_map[p]; /* :64-68 */
    }
}

/* regrid_pixels_hpix verbatim (HealpixRunner.py:62-68); child arrays are [N][4] */
void orc_regrid_pixels_hpix(double *hmap, int64_t n, const double *parent_pix_vals,
                            const int64_t *child_pix, const double *child_weights)
{
    for (int64_t i = 0; i < n; ++i)
        for (int j = 0; j < 4; ++j)
            hmap[child_pix[4 * i + j]] += child_weights[4 * i + j] * parent_pix_vals[i];
}
