// bfg_enum.hpp -- the geometry half of the shell loops for models that are NOT tabulated.
//
// The reference paints any object with a .projected(cosmo, r, M, a) / .displacement(r, M, a) method
// (Runners/HealpixRunner.py:472, :345): a Python callable, evaluated once per halo on the distances of the halo's disc
// pixels.  That evaluation cannot leave the host, but everything around it can: the kernels here produce, for a batch of
// halos, the flat lists (pixel, r_sep / a_j, halo) of HealpixRunner.py:460-469 / :327-342 (query_disc, the < 4 pixel rule,
// pix2vec, |vec D - vec_j D|), and add the values the host computed from them to the map (:481) or turn them into
// unit-vector offsets (:345-355).  Same ring-window enumeration as shell_scatter_kernel (query_disc_internal,
// non-inclusive), one wavefront per halo.
#pragma once

namespace bfg {

struct EnumParams {
    PrepParams prep;                 // catalog, runner mass definition, D_A spline, eps_run, hpx (tab.nouter = 0: no table)
    int fallback4;                   // BaryonifyShell: a disc of < 4 pixels becomes the 4 bilinear neighbours (HealpixRunner.py:333-334)
    int64_t *counts;                 // count pass: [n_halo] entries of every halo
    const int64_t *base;             // fill pass: exclusive prefix sum of counts
    int64_t *pix;                    // fill pass: [total] RING pixel
    double *r_com;                   //            [total] r_sep / a_j (comoving Mpc)
    int32_t *halo;                   //            [total] index of the halo in this batch
};

struct EnumLds {
    int32_t cum[64], nr[64], iplo[64], pad[64];
    int64_t start[64];
    double z[64], sth[64], phistep[64], phioff[64];
};

// r_sep / a of pixel (z, sth, phi) around halo `o` -- the arithmetic of scatter_halo::process_pixel
__device__ inline double enum_r_com(const HaloCalc &o, double z, double sth, double phi)
{
    double sphi, cphi;
    sincos(phi, &sphi, &cphi);
    const double p0 = sth * cphi * o.D, p1 = sth * sphi * o.D, p2 = z * o.D;             // pos = vec * D_j   (:465)
    const double d0 = p0 - o.x0 * o.D, d1 = p1 - o.y0 * o.D, d2 = p2 - o.z0v * o.D;      // diff          (:466)
    return sqrt(d0 * d0 + d1 * d1 + d2 * d2) / o.a;                                      // r_sep / a_j   (:467, :472)
}

template <bool COUNT>
__global__ __launch_bounds__(256) void disc_enum_kernel(const EnumParams P)
{
    __shared__ EnumLds lds[4];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    EnumLds *rl = lds + grp;
    const Hpx &hp = P.prep.hpx;
    for (int64_t j = (int64_t)blockIdx.x * 4 + grp; j < P.prep.n_halo; j += (int64_t)gridDim.x * 4) {
        HaloCalc o;
        halo_calc(P.prep, j, P.prep.spl_knots, [](int) -> const double * { return nullptr; }, [](int, int, double) {}, o);
        if (o.flags & HF_SKIP) {
            if (COUNT && lane == 0) P.counts[j] = 0;
            continue;
        }
        const double cosrbig = cos(o.radius), z0 = cos(o.ptheta);
        const double xa = 1.0 / sqrt((1.0 - z0) * (1.0 + z0));
        const int64_t out0 = COUNT ? 0 : P.base[j];
        int64_t done = 0;
        const int nrings = o.rlast - o.rfirst + 1;
        for (int ring0 = o.rfirst; ring0 <= o.rlast || ring0 == o.rfirst; ring0 += 64) {
            const int ring = ring0 + lane;
            int cnt = 0, iplo = 0;
            RingGeom g;
            g.start = 0; g.nr = 1; g.z = 0; g.sth = 0; g.phistep = 0; g.phioff = 0;
            if (ring <= o.rlast) {
                g = ring_geom(hp, ring);
                if (ring < o.irmin || ring > o.irmax) { cnt = g.nr; iplo = 0; }      // ring completely inside the disc
                else {
                    const double z = ring2z(hp, ring);
                    const double x = (cosrbig - z * z0) * xa;
                    const double ysq = 1.0 - z * z - x * x;
                    const double dphi = (ysq <= 0.0) ? 0.0 : atan2(sqrt(ysq), x);
                    if (dphi > 0.0) {
                        const double shift = (g.phioff != 0.0) ? 0.5 : 0.0;
                        const int64_t lo = (int64_t)floor((double)g.nr * kInvTwoPi * (o.pphi - dphi) - shift) + 1;
                        const int64_t hi = (int64_t)floor((double)g.nr * kInvTwoPi * (o.pphi + dphi) - shift);
                        int64_t c = hi - lo + 1;
                        if (c > g.nr) c = g.nr;
                        if (c > 0) { cnt = (int)c; iplo = (int)lo; }
                    }
                }
            }
            const int cum = group_inclusive_scan<64>(cnt, lane);
            const int total = __shfl(cum, 63, 64);
            if (P.fallback4 && nrings <= 64 && total < 4) {
                // fewer than 4 pixels: the 4 bilinear neighbours of the halo centre, in get_interp_weights' order
                if (!COUNT && lane < 4) {
                    const double *c = P.prep.cat + j * (int64_t)P.prep.cat_stride;
                    int64_t fp[4]; double fw[4];
                    get_interpol(hp, kHalfPi - c[3] * kDeg2Rad, c[2] * kDeg2Rad, fp, fw);
                    int64_t fring, fip;
                    pix2ring(hp, fp[lane], fring, fip);
                    const RingGeom fg = ring_geom(hp, fring);
                    P.pix[out0 + lane] = fp[lane];
                    P.r_com[out0 + lane] = enum_r_com(o, fg.z, fg.sth, ((double)fip + fg.phioff) * fg.phistep);
                    P.halo[out0 + lane] = (int32_t)j;
                }
                done = 4;
                break;
            }
            if (!COUNT) {
                rl->cum[lane] = cum; rl->nr[lane] = g.nr; rl->iplo[lane] = iplo; rl->start[lane] = g.start;
                rl->z[lane] = g.z; rl->sth[lane] = g.sth; rl->phistep[lane] = g.phistep; rl->phioff[lane] = g.phioff;
                __builtin_amdgcn_wave_barrier();
                for (int t = lane; t < total; t += 64) {
                    int lo = 0, hi = 63;                                  // smallest jr with cum[jr] > t
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        if (rl->cum[mid] > t) hi = mid; else lo = mid + 1;
                    }
                    const int jr = lo;
                    const int before = (jr == 0) ? 0 : rl->cum[jr - 1];
                    const int nr = rl->nr[jr];
                    int ip = rl->iplo[jr] + (t - before);
                    if (ip < 0) ip += nr;
                    if (ip >= nr) ip -= nr;
                    if (ip >= nr) ip -= nr;
                    const int64_t e = out0 + done + t;
                    P.pix[e] = rl->start[jr] + ip;
                    P.r_com[e] = enum_r_com(o, rl->z[jr], rl->sth[jr], ((double)ip + rl->phioff[jr]) * rl->phistep[jr]);
                    P.halo[e] = (int32_t)j;
                }
                __builtin_amdgcn_wave_barrier();
            }
            done += total;
        }
        if (COUNT && lane == 0) P.counts[j] = done;
    }
}

// new_map[pixind] += Paint (HealpixRunner.py:481); zeros add nothing
__global__ __launch_bounds__(256) void values_add_kernel(double *__restrict__ out, const int64_t *__restrict__ pix,
                                                         const double *__restrict__ val, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double v = val[i];
        if (v != 0.0) unsafeAtomicAdd(out + pix[i], v);
    }
}

// HealpixRunner.py:345-355 for one (halo, pixel) entry: disp = model.displacement(r_sep / a_j, M_j, a_j) as the host computed it
struct DispParams {
    PrepParams prep;
    const int64_t *pix;
    const int32_t *halo;
    const double *disp;              // comoving displacement per entry
    int64_t n;
    double *out;                     // [npix][3]
};

__global__ __launch_bounds__(256) void displacements_add_kernel(const DispParams P)
{
    const Hpx &hp = P.prep.hpx;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P.n; i += (int64_t)gridDim.x * blockDim.x) {
        HaloCalc o;
        halo_calc(P.prep, (int64_t)P.halo[i], P.prep.spl_knots, [](int) -> const double * { return nullptr; }, [](int, int, double) {}, o);
        const int64_t pix = P.pix[i];
        int64_t ring, ip;
        pix2ring(hp, pix, ring, ip);
        const RingGeom g = ring_geom(hp, ring);
        double sphi, cphi;
        sincos(((double)ip + g.phioff) * g.phistep, &sphi, &cphi);
        const double vx = g.sth * cphi, vy = g.sth * sphi, vz = g.z;
        const double p0 = vx * o.D, p1 = vy * o.D, p2 = vz * o.D;
        const double d0 = p0 - o.x0 * o.D, d1 = p1 - o.y0 * o.D, d2 = p2 - o.z0v * o.D;
        const double r_sep = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
        const double d = P.disp[i] * o.a;                                          // :345
        double o0 = d * (d0 / r_sep), o1 = d * (d1 / r_sep), o2 = d * (d2 / r_sep);   // :346
        if (!isfinite(o0)) o0 = 0.0;                                               // :347
        if (!isfinite(o1)) o1 = 0.0;
        if (!isfinite(o2)) o2 = 0.0;
        const double n0 = p0 + o0, n1 = p1 + o1, n2 = p2 + o2;                     // :350
        const double nn = sqrt(n0 * n0 + n1 * n1 + n2 * n2);                       // :351
        if (o0 != 0.0 || o1 != 0.0 || o2 != 0.0) {                                 // zero offsets add exactly 0
            double *q = P.out + 3 * pix;
            unsafeAtomicAdd(q + 0, n0 / nn - vx);                                  // :352, :355
            unsafeAtomicAdd(q + 1, n1 / nn - vy);
            unsafeAtomicAdd(q + 2, n2 / nn - vz);
        }
    }
}

}  // namespace bfg
