// bfg_snapshot.hpp -- BaryonifySnapshot (BaryonForge/Runners/SnapshotRunner.py:176-275) on the GPU.
//
// The reference finds, halo by halo, the particles inside R_q = min(eps R / a, L / 2) with a periodic scipy KDTree
// (:99, :225 / :240), reads the radial displacement off the model's table and accumulates offset * unit-vector on
// every particle found (:232 / :249); at the end positions are shifted and wrapped into the box once (:260-273).
//
// Here the search is turned around (particles outnumber halos by 10^3): a coarse periodic cell grid holds, per cell,
// the list of halos whose sphere can reach the cell (built halo by halo: count -> scan -> fill over each sphere's
// bounding box; spheres spanning more than kSnapBigCells cells go to one global list instead).  Then ONE pass over the
// particles, one thread per particle in whatever order they come: look up the particle's cell, test its handful of
// candidate halos (+ the global list), read the displacement off the halo's blended radial row (hrow), sum the
// offsets in registers, shift, wrap, write.  No particle sort, no atomics on particle data, one read and one write of
// every particle.
// Included by bfg_mi355.hip after DevTable / massdef_radius are defined.
#pragma once

namespace bfg {

constexpr int kSnapBigCells = 4096;
constexpr int kSnapOverlapLanes = 16;    // lanes that share one halo's bounding box in snap_overlap_kernel

struct __align__(16) SnapHalo {      // per-halo constants (written by snap_halo_kernel)
    double x, y, z, rq;              // centre [comoving Mpc], query radius
    double xcut;                     // model.epsilon_max * R_model_com: no displacement at or beyond it
    double lnshift;                  // ln(R_model_com) for Rdelta_sampling tables, else 0
    int32_t flags, pad;              // HF_OOB: (z, M, extras) outside the table hull -> contributes nothing
    double pad2;
};

struct __align__(16) SnapCand {      // one (cell, halo) entry of the overlap lists (64 B)
    double x, y, z, rq;
    double xcut, lnshift;            // copies of SnapHalo's: a hit needs no second per-halo record
    int32_t halo, pad[3];
};

struct SnapParams {
    int ndim;                        // 2 or 3
    int rdelta;
    int64_t n_part, n_halo;
    double L, a, eps_run, eps_model;
    bfg_massdef md_run, md_model;
    const double *part;              // particle k, coordinate c at part[k * pstride + c]
    int64_t pstride, ostride;        // doubles per particle record of part / out (ndim when packed)
    const double *halo;              // [n_halo][stride]: M, lnM (table coordinate), x, y, z, extras...
    int halo_stride, n_extra;
    DevTable tab;
    int ncell;                       // cells per dimension
    int64_t ncell_tot;
    int32_t *cell_count;             // [ncell_tot] (count, then fill cursor)
    int32_t *cell_start;             // [ncell_tot + 1]
    SnapCand *cand;                  // [cand_cap] candidates grouped by cell: test data inline, no second indirection
    int64_t cand_cap;
    int32_t *big;                    // [0] = count, [1..] halos whose sphere spans too many cells (tested by every particle)
    SnapHalo *hs;                    // [n_halo]
    double *hrow;                    // [n_halo][tab.nr] blended radial rows
    double *out;                     // [n_part][ndim] displaced, wrapped coordinates
    bfg_stats *stats;
    const double2 *logtab;           // [kLogTab] {1/c, ln c} of fast_log (bfg_tile.hpp), staged in LDS
    // cell-grouped pass: particle indices grouped by cell (counting sort of 4-byte indices)
    int32_t *pkey;                   // [n_part] cell of every particle
    int32_t *pcount;                 // [ncell_tot] counts, then fill cursors
    const int32_t *pstart;           // [ncell_tot + 1]
    int32_t *perm;                   // [n_part]
    int xcd_map;                     // snap_particle_kernel: XCD-contiguous chunk order (below)
};

// particle coordinates from / to a record array with a runtime row stride; the packed case (stride == NDIM) keeps its
// compile-time stride so that the three loads / stores of a particle stay one wide access
template <int NDIM>
__device__ inline void load_coords(const double *base, int64_t ip, int64_t stride, double *p)
{
    if (stride == NDIM) { for (int k = 0; k < NDIM; ++k) p[k] = base[ip * NDIM + k]; }
    else { for (int k = 0; k < NDIM; ++k) p[k] = base[ip * stride + k]; }
}
template <int NDIM>
__device__ inline void store_coords(double *base, int64_t ip, int64_t stride, const double *v)
{
    if (stride == NDIM) { for (int k = 0; k < NDIM; ++k) base[ip * NDIM + k] = v[k]; }
    else { for (int k = 0; k < NDIM; ++k) base[ip * stride + k] = v[k]; }
}

__device__ inline int snap_cell_of(double x, double inv_cell, int n)
{
    int i = (int)floor(x * inv_cell);
    return min(max(i, 0), n - 1);    // x in [0, L]; x == L lands in the last cell
}

// exclusive scan of n int32 in three phases (block sums -> scan of the sums -> add), 1024 elements per block
__global__ __launch_bounds__(256) void snap_scan_block_kernel(int64_t n, const int32_t *in, int32_t *out, int32_t *bsum)
{
    __shared__ int32_t wsum[4];
    const int64_t base = (int64_t)blockIdx.x * 1024 + threadIdx.x * 4;
    int v[4], s = 0;
    for (int k = 0; k < 4; ++k) { v[k] = (base + k < n) ? in[base + k] : 0; s += v[k]; }
    int incl = s;
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d, 64); if ((threadIdx.x & 63) >= d) incl += o; }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) woff += wsum[w];
    int run = woff + incl - s;
    for (int k = 0; k < 4; ++k) { if (base + k < n) out[base + k] = run; run += v[k]; }
    if (threadIdx.x == 255) bsum[blockIdx.x] = woff + incl;
}

__global__ __launch_bounds__(1024) void snap_scan_sums_kernel(int nb, int32_t *bsum, int32_t *total)
{
    __shared__ int32_t wsum[16];
    __shared__ int32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nb; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = (i < nb) ? bsum[i] : 0;
        int incl = v;
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d, 64); if ((threadIdx.x & 63) >= d) incl += o; }
        if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) woff += wsum[w];
        const int c = carry;
        if (i < nb) bsum[i] = c + woff + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = c + woff + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}

__global__ __launch_bounds__(256) void snap_scan_add_kernel(int64_t n, int32_t *out, const int32_t *bsum, int32_t *count,
                                                            const int32_t *total)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { out[i] += bsum[i >> 10]; count[i] = 0; }      // count becomes the fill cursor
    if (i == 0) out[n] = *total;
}

// ---- grouping particle indices by an integer key (cell / tile): counting sort of 4-byte indices --------------------
// keys[] -> counts (wave-aggregated atomics) -> exclusive scan (kernels above) -> slots (wave-aggregated atomics).
// wave_group_slot adds 1 to count[key] for every active lane and returns the lane's slot in its key's range.  Lanes of
// a wavefront that share a key are merged into ONE atomic: up to 16 groups are found with ballots (spatially coherent
// particle orders have 1-4 keys per wavefront), the rest stay singletons; every group leader then issues its atomic in
// the same instruction, so a wavefront waits for one atomic round trip, not one per group.
__device__ inline int wave_group_slot(int32_t *count, int key, bool active)
{
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(active);
    unsigned long long mine = 1ull << lane;
    int leader = lane;
    for (int round = 0; round < 16 && todo; ++round) {
        const int l0 = __ffsll((long long)todo) - 1;
        const int k0 = __shfl(key, l0, 64);
        const unsigned long long same = __ballot(active && key == k0) & todo;
        if ((same >> lane) & 1ull) { mine = same; leader = l0; }
        todo &= ~same;
    }
    int base = 0;
    if (active && lane == leader) base = atomicAdd(&count[key], __popcll(mine));
    base = __shfl(base, leader, 64);
    return base + __popcll(mine & ((1ull << lane) - 1ull));
}

// perm[start[key] + slot] = particle index, for every particle with key >= 0 (count[] holds the fill cursors, zeroed)
__global__ __launch_bounds__(256) void group_fill_kernel(int64_t n, const int32_t *key, int32_t *count, const int32_t *start,
                                                         int32_t *perm)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t base = (int64_t)blockIdx.x * blockDim.x; base < n; base += stride) {          // wave-uniform trip count
        const int64_t ip = base + threadIdx.x;
        const int k = (ip < n) ? key[ip] : -1;
        const int slot = wave_group_slot(count, k, k >= 0);
        if (k >= 0) perm[(int64_t)start[k] + slot] = (int32_t)ip;
    }
}

// per halo: radii, table cell in the outer dimensions, full blended radial row
__global__ __launch_bounds__(64) void snap_halo_kernel(const SnapParams P)
{
    const int64_t j = blockIdx.x;
    const int lane = threadIdx.x;
    const double *c = P.halo + j * P.halo_stride;
    const double M = c[0], lnM = c[1];
    const DevTable &T = P.tab;
    __shared__ double s_w[kMaxCorner];
    __shared__ int64_t s_off[kMaxCorner];
    __shared__ int s_oob;
    if (lane == 0) {
        const double R = massdef_radius(P.md_run, M, P.a);                          // physical Mpc (:222)
        double rq = P.eps_run * R / P.a;                                            // comoving (:223)
        rq = fmin(fmax(rq, 0.0), 0.5 * P.L);                                        // :224 (NaN -> NaN: finds nothing)
        const double Rm = massdef_radius(P.md_model, M, P.a) / P.a;                 // BaryonCorrection.py:399
        bool oob = false;
        uint32_t warn = 0;
        int idx[BFG_MAX_DIM];
        double wt[BFG_MAX_DIM];
        for (int k = 0; k < T.nouter; ++k) {
            const double x = (k == 0) ? log(1.0 / P.a) : (k == 1) ? lnM : c[5 + (k - 2)];
            const double *g = T.oaxis[k];
            const int n = T.oshape[k];
            if (!(x >= g[0]) || !(x <= g[n - 1])) {
                oob = true;
                if (k == 0) warn |= BFG_WARN_Z_RANGE;
                if (k == 1) warn |= BFG_WARN_M_RANGE;
            }
            idx[k] = find_interval(g, n, x);
            wt[k] = (x - g[idx[k]]) / (g[idx[k] + 1] - g[idx[k]]);
        }
        const int ncorner = 1 << T.nouter;
        for (int cc = 0; cc < ncorner; ++cc) {
            double w = 1.0;
            int64_t off = 0;
            for (int k = 0; k < T.nouter; ++k) {
                const int bit = (cc >> (T.nouter - 1 - k)) & 1;
                w *= bit ? wt[k] : 1.0 - wt[k];
                off += (int64_t)(idx[k] + bit) * T.ostride[k];
            }
            s_w[cc] = w; s_off[cc] = off;
        }
        s_oob = oob ? 1 : 0;
        if (oob) {
            atomicAdd((unsigned long long *)&P.stats->halos_out_of_table, 1ull);
            atomicOr(&P.stats->warn_mask, warn);
        }
        SnapHalo h;
        h.x = c[2]; h.y = c[3]; h.z = (P.ndim == 3) ? c[4] : 0.0; h.rq = rq;
        h.xcut = P.eps_model * Rm;
        h.lnshift = P.rdelta ? log(Rm) : 0.0;
        h.flags = oob ? HF_OOB : 0; h.pad = 0; h.pad2 = 0.0;
        P.hs[j] = h;
    }
    __syncthreads();
    const int ncorner = 1 << T.nouter;
    for (int i = lane; i < T.nr; i += 64) {
        double b = 0.0;
        for (int cc = 0; cc < ncorner; ++cc) b = fma(T.values[s_off[cc] + i], s_w[cc], b);
        P.hrow[j * T.nr + i] = s_oob ? nan("") : b;
    }
}

// cells of the (periodic) grid a halo's sphere can reach: count pass (fill = false) / list fill (fill = true);
// spheres spanning more than kSnapBigCells cells are appended to the global list in the count pass
template <int NDIM>
__global__ __launch_bounds__(256) void snap_overlap_kernel(const SnapParams P, int fill)
{
    // kSnapOverlapLanes lanes per halo share the cells of its bounding box (one thread per halo: the lanes of a wavefront waited for
    // the largest sphere among 64 -- 0.285 ms per pass at BASELINE configs[4], 250 cells per halo on average, up to 4096)
    const int64_t j = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kSnapOverlapLanes;
    const int sub = (int)(threadIdx.x % kSnapOverlapLanes);
    if (j >= P.n_halo) return;
    const SnapHalo h = P.hs[j];
    if ((h.flags & HF_OOB) || !(h.rq > 0.0)) return;           // NaN displacement everywhere -> 0 (:231 / :248)
    const int n = P.ncell;
    const double cell = P.L / (double)n, inv_cell = (double)n / P.L;
    const double hc[3] = {h.x, h.y, h.z};
    int lo[3] = {0, 0, 0}, cnt[3] = {1, 1, 1};
    for (int k = 0; k < NDIM; ++k) {
        // 1e-9 cells of slack: a particle's own cell index is floor(x / cell) of its wrapped coordinate
        const int a = (int)floor((hc[k] - h.rq) * inv_cell - 1e-9), b = (int)floor((hc[k] + h.rq) * inv_cell + 1e-9);
        lo[k] = a; cnt[k] = min(b - a + 1, n);                 // at most every cell once
    }
    const int64_t nbox = (int64_t)cnt[0] * cnt[1] * cnt[2];
    // A list entry carries the halo's IMAGE as the particles of that cell see it (centre shifted by a multiple of L, below), so the
    // particle pass needs no periodic wrap per candidate.  That image is unique only while the sphere covers at most half the box
    // along every axis; larger spheres go to the global list, whose test wraps (compute_distance / enforce_periodicity).
    bool half_box = false;
    for (int k = 0; k < NDIM; ++k) half_box = half_box || 2 * cnt[k] > n;
    if (nbox > kSnapBigCells || half_box) {
        if (!fill && sub == 0) P.big[1 + atomicAdd(&P.big[0], 1)] = (int32_t)j;
        return;
    }
    const double rq2 = h.rq * h.rq * (1.0 + 1e-12);
    for (int64_t ci = sub; ci < nbox; ci += kSnapOverlapLanes) {
        int64_t rem = ci, cid = 0;
        int ic[3];
        for (int k = NDIM - 1; k >= 0; --k) { ic[k] = lo[k] + (int)(rem % cnt[k]); rem /= cnt[k]; }
        double dmin2 = 0.0;
        double shift[3] = {0.0, 0.0, 0.0};
        for (int k = 0; k < NDIM; ++k) {
            const double c0 = (double)ic[k] * cell, c1 = c0 + cell;
            double gap = 0.0;
            if (cnt[k] < n - 1) {                              // (nearly) full wrap: another image of the cell may be closer
                if (hc[k] < c0) gap = c0 - hc[k]; else if (hc[k] > c1) gap = hc[k] - c1;
            }
            dmin2 += gap * gap;
            int w = ic[k] % n; if (w < 0) w += n;
            cid = cid * n + w;
            shift[k] = (double)((ic[k] - w) / n) * P.L;        // the unwrapped cell is the wrapped one + this many box lengths
        }
        if (dmin2 > rq2) continue;                             // the sphere cannot reach this cell
        if (!fill) atomicAdd(&P.cell_count[cid], 1);
        else {
            const int64_t pos = (int64_t)P.cell_start[cid] + atomicAdd(&P.cell_count[cid], 1);
            if (pos < P.cand_cap) {
                SnapCand e;
                e.x = h.x - shift[0]; e.y = h.y - shift[1]; e.z = h.z - shift[2];     // the image next to this cell
                e.rq = h.rq; e.xcut = h.xcut; e.lnshift = h.lnshift;
                e.halo = (int32_t)j; e.pad[0] = e.pad[1] = e.pad[2] = 0;
                P.cand[pos] = e;
            }
        }
    }
}

// One (halo, particle) hit: BaryonificationClass._readout (BaryonCorrection.py:331-419) on the halo's blended radial
// row -- linear table in ln r, NaN outside the hull (contributes nothing, :231 / :248), 0 at or beyond
// epsilon_max * R -- and the radial unit vector.  d2 = squared periodic distance (<= rq^2, checked by the caller).
// Lean arithmetic: 1/sqrt by v_rsq_f64 + two coupled Newton steps (d and 1/d to ~1 ulp, no sqrt / division),
// ln d = ln(d2)/2 from the table-driven fast_log (abs err < 6e-11; libm within 1e-9 of the axis ends so that the
// in / out-of-table decision is the reference's), the cell on a geomspace axis by arithmetic.
struct SnapHit {
    const DevTable *T;
    const double *hrow;
    const double2 *logtab;           // LDS
    double r_lo, r_hi;
};

__device__ inline void snap_hit(const SnapHit &H, double d2, const double *dd, int ndim, int j, double xcut, double lnshift,
                                double *off, unsigned long long &n_oob)
{
    const DevTable &T = *H.T;
    double rin, d, rinv;
    if (d2 >= 1e-290 && d2 <= 1e290) {
        double y = __builtin_amdgcn_rsq(d2);
        double g = d2 * y, h = 0.5 * y;
        double r = fma(-h, g, 0.5);
        g = fma(g, r, g); h = fma(h, r, h);
        r = fma(-h, g, 0.5);
        g = fma(g, r, g); h = fma(h, r, h);
        d = g; rinv = h + h;
        rin = 0.5 * fast_log(d2, H.logtab) - lnshift;
        if (fabs(rin - H.r_lo) < 1e-9 || fabs(rin - H.r_hi) < 1e-9) rin = log(sqrt(d2)) - lnshift;
    } else {
        if (d2 == 0.0) {
            // a particle exactly on the halo centre: the reference's unit vector is 0 / 0 and its (zeroed) offset times that
            // is NaN (SnapshotRunner.py:228-232 / :244-249) -- the particle's new position is NaN there, so it is here.
            // (Tested in this rare branch: the same test ahead of the fast path cost the kernel 7 %.)
            for (int k = 0; k < ndim; ++k) off[k] += nan("");
            ++n_oob;                                   // r = 0 lies below the table's radial axis (BaryonCorrection.py:391-394)
            return;
        }
        d = sqrt(d2); rinv = 1.0 / d;
        rin = log(d) - lnshift;
    }
    if (!(rin >= H.r_lo) || !(rin <= H.r_hi)) { ++n_oob; return; }
    if (!(d < xcut)) return;
    const double *row = H.hrow + (int64_t)j * T.nr;
    int i;
    double f;
    if (T.r_uniform) {
        const double t = (rin - T.r0) * T.inv_dr;
        i = min(max((int)t, 0), T.nr - 2);
        f = t - (double)i;
    } else {
        i = find_interval(T.raxis, T.nr, rin);
        f = (rin - T.raxis[i]) / (T.raxis[i + 1] - T.raxis[i]);
    }
    const double val = row[i] * (1.0 - f) + row[i + 1] * f;
    if (!(fabs(val) < 1.0e300)) return;
    const double sc = val * rinv;
    for (int k = 0; k < ndim; ++k) off[k] += sc * dd[k];
}

// one thread per particle: candidates of its cell (+ the global list) -> summed offset -> shift, wrap, write
template <int NDIM>
__global__ __launch_bounds__(256) void snap_particle_kernel(const SnapParams P)
{
    // grid-stride over the particles: a few thousand fat workgroups, so that the two statistics counters see a few
    // thousand same-address atomics instead of one per wavefront (2e6 of those cost 20 ms)
    __shared__ double2 s_logtab[kLogTab];
    if (threadIdx.x < kLogTab) s_logtab[threadIdx.x] = P.logtab[threadIdx.x];
    __syncthreads();
    const SnapHit H = {&P.tab, P.hrow, s_logtab, P.tab.raxis[0], P.tab.raxis[P.tab.nr - 1]};
    unsigned long long hits = 0, n_oob = 0;
    // Chunks of 256 consecutive particles.  Workgroups are dealt to the eight XCDs round-robin (blockIdx mod 8), each XCD with an L2 of
    // its own: with chunk = blockIdx the neighbouring chunks of a snapshot stored in a spatially coherent order -- whose particles
    // sit in the same cells and read the same candidate lists -- went to eight different L2s, every one of which fetched those lists
    // again.  xcd_map: XCD x works through the x-th EIGHTH of the chunks, so chunks that follow each other in memory follow each
    // other in one L2 (gridDim is a multiple of 8: a workgroup's XCD is the same in every trip of the grid-stride loop).
    const int64_t nchunk = (P.n_part + blockDim.x - 1) / blockDim.x, per_xcd = (nchunk + 7) / 8;
    for (int64_t it = blockIdx.x; it < (P.xcd_map ? 8 * per_xcd : nchunk); it += gridDim.x) {
        const int64_t chunk = P.xcd_map ? (it & 7) * per_xcd + (it >> 3) : it;
        const int64_t ip = chunk * blockDim.x + threadIdx.x;
        if ((P.xcd_map && (it >> 3) >= per_xcd) || ip >= P.n_part) continue;
        const DevTable &T = P.tab;
        const double L = P.L, halfL = 0.5 * P.L;
        const int n = P.ncell;
        const double inv_cell = (double)n / L;
        double p[3] = {0.0, 0.0, 0.0}, off[3] = {0.0, 0.0, 0.0};
        int64_t cid = 0;
        load_coords<NDIM>(P.part, ip, P.pstride, p);
        for (int k = 0; k < NDIM; ++k) cid = cid * n + snap_cell_of(p[k], inv_cell, n);
        // a candidate: periodic distance test (compute_distance / enforce_periodicity, :104-158; KDTree radius :225 / :240)
        auto test = [&](double hx_, double hy_, double hz_, double rq_, double *dd, double &d2) -> bool {
            const double hc[3] = {hx_, hy_, hz_};
            d2 = 0.0;
            for (int k = 0; k < NDIM; ++k) {
                double dx = p[k] - hc[k];
                dx = (dx > halfL) ? dx - L : dx;
                dx = (dx < -halfL) ? dx + L : dx;
                dd[k] = dx; d2 += dx * dx;
            }
            return d2 <= rq_ * rq_;
        };
        // ... of a cell list: the entry is the halo's image next to this cell (snap_overlap_kernel), the separation needs no wrap
        auto test_image = [&](double hx_, double hy_, double hz_, double rq_, double *dd, double &d2) -> bool {
            const double hc[3] = {hx_, hy_, hz_};
            d2 = 0.0;
            for (int k = 0; k < NDIM; ++k) { const double dx = p[k] - hc[k]; dd[k] = dx; d2 += dx * dx; }
            return d2 <= rq_ * rq_;
        };
        const int c0 = P.cell_start[cid], c1 = (int)min((int64_t)P.cell_start[cid + 1], P.cand_cap);
        // two phases per batch of 64 candidates, so that the expensive read-out runs on the lanes' own hits only
        // (a lane hits ~2 of its ~8 candidates; done inline, every trip of the test loop would pay for the read-out
        // because some lane of the wavefront hits): (1) distance tests -> hit mask, (2) the set bits in ascending order
        for (int qb = c0; qb < c1; qb += 64) {
            const int qe = min(qb + 64, c1);
            unsigned long long mask = 0;
            // three independent record loads in flight per trip (four: 87 VGPRs = five wavefronts per SIMD; three: 78 = six, and the
            // sixth wavefront hides more than the fourth load did -- 5.05 -> 4.79 ms at BASELINE configs[4]; two: 4.87)
            constexpr int U = 3;
            for (int q = qb; q < qe; q += U) {
                double4 e[U];
                for (int u = 0; u < U; ++u) e[u] = *reinterpret_cast<const double4 *>(&P.cand[min(q + u, qe - 1)]);   // x, y, z, rq
                for (int u = 0; u < U; ++u) {
                    double dd[3], d2;
                    if (test_image(e[u].x, e[u].y, e[u].z, e[u].w, dd, d2) && q + u < qe) mask |= 1ull << (q + u - qb);
                }
            }
            while (mask) {
                const int q = qb + __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                const SnapCand e = P.cand[q];
                double dd[3] = {0, 0, 0}, d2;
                (void)test_image(e.x, e.y, e.z, e.rq, dd, d2);
                ++hits;
                snap_hit(H, d2, dd, NDIM, e.halo, e.xcut, e.lnshift, off, n_oob);
            }
        }
        const int nbig = P.big[0];
        for (int q = 0; q < nbig; ++q) {
            const int j = P.big[1 + q];
            const SnapHalo h = P.hs[j];
            double dd[3] = {0, 0, 0}, d2;
            if (test(h.x, h.y, h.z, h.rq, dd, d2)) { ++hits; snap_hit(H, d2, dd, NDIM, j, h.xcut, h.lnshift, off, n_oob); }
        }
        double pn[3] = {0.0, 0.0, 0.0};
        for (int k = 0; k < NDIM; ++k) {
            double v = p[k] + off[k];                          // :262-265
            v = (v > L) ? v - L : v;                           // :268-273
            v = (v < 0.0) ? v + L : v;
            pn[k] = v;
        }
        store_coords<NDIM>(P.out, ip, P.ostride, pn);
    }
    __shared__ unsigned long long s_red[2][4];
    for (int o = 32; o > 0; o >>= 1) { hits += __shfl_down(hits, o, 64); n_oob += __shfl_down(n_oob, o, 64); }
    if ((threadIdx.x & 63) == 0) { s_red[0][threadIdx.x >> 6] = hits; s_red[1][threadIdx.x >> 6] = n_oob; }
    __syncthreads();
    if (threadIdx.x == 0) {
        hits = s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3];
        n_oob = s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3];
        if (hits) atomicAdd((unsigned long long *)&P.stats->pixel_updates, hits);
        if (n_oob) {
            atomicAdd((unsigned long long *)&P.stats->pixels_out_of_table, n_oob);
            if (!P.rdelta) atomicOr(&P.stats->warn_mask, BFG_WARN_R_RANGE);       // BaryonCorrection.py:391-394
        }
    }
}

// ---- cell-grouped pass ----------------------------------------------------------------------------------------------
// The one-thread-per-particle kernel above reads its candidate records lane by lane: a wavefront of 64 consecutive
// particles straddles several cells, so every trip of the candidate loop is a divergent 64-byte gather (8 tests per
// particle on average: ~70 GB of L2 -> CU traffic for 512^3 particles).  Here the particle indices are grouped by cell
// first (key -> count -> scan -> fill, 4 bytes per particle), then one wavefront works on one cell: the cell's
// candidate list is wave-uniform (scalar loads, one per candidate for 64 particles), every lane tests its own
// particle, and results go to out[] in the caller's particle order.
template <int NDIM>
__global__ __launch_bounds__(256) void snap_key_kernel(const SnapParams P)
{
    const int n = P.ncell;
    const double inv_cell = (double)n / P.L;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t base = (int64_t)blockIdx.x * blockDim.x; base < P.n_part; base += stride) {
        const int64_t ip = base + threadIdx.x;
        int key = -1;
        if (ip < P.n_part) {
            key = 0;
            double pk[3];
            load_coords<NDIM>(P.part, ip, P.pstride, pk);
            for (int k = 0; k < NDIM; ++k) key = key * n + snap_cell_of(pk[k], inv_cell, n);
            P.pkey[ip] = key;
        }
        (void)wave_group_slot(P.pcount, key, key >= 0);
    }
}

template <int NDIM>
__global__ __launch_bounds__(256) void snap_cell_kernel(const SnapParams P)
{
    __shared__ double2 s_logtab[kLogTab];
    if (threadIdx.x < kLogTab) s_logtab[threadIdx.x] = P.logtab[threadIdx.x];
    __syncthreads();
    const SnapHit H = {&P.tab, P.hrow, s_logtab, P.tab.raxis[0], P.tab.raxis[P.tab.nr - 1]};
    const int lane = threadIdx.x & 63;
    const double L = P.L, halfL = 0.5 * P.L;
    const int nbig = P.big[0];
    unsigned long long hits = 0, n_oob = 0;
    for (int64_t cell = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); cell < P.ncell_tot; cell += (int64_t)gridDim.x * 4) {
        const int64_t cid = __builtin_amdgcn_readfirstlane((int)cell);                       // wave-uniform
        const int p0 = __builtin_amdgcn_readfirstlane(P.pstart[cid]), p1 = __builtin_amdgcn_readfirstlane(P.pstart[cid + 1]);
        if (p0 == p1) continue;
        const int c0 = __builtin_amdgcn_readfirstlane(P.cell_start[cid]);
        const int c1 = __builtin_amdgcn_readfirstlane((int)min((int64_t)P.cell_start[cid + 1], P.cand_cap));
        for (int qb = p0; qb < p1; qb += 64) {
            const bool valid = qb + lane < p1;
            const int64_t ip = P.perm[valid ? qb + lane : p1 - 1];
            double p[3] = {0.0, 0.0, 0.0}, off[3] = {0.0, 0.0, 0.0};
            load_coords<NDIM>(P.part, ip, P.pstride, p);
            // one candidate against this lane's particle: periodic distance (compute_distance / enforce_periodicity,
            // :104-158; KDTree radius :225 / :240), then the read-out
            auto visit = [&](double hx_, double hy_, double hz_, double rq_, double xcut_, double lnshift_, int j, bool wrap) {
                const double hc[3] = {hx_, hy_, hz_};
                double dd[3] = {0.0, 0.0, 0.0}, d2 = 0.0;
                for (int k = 0; k < NDIM; ++k) {
                    double dx = p[k] - hc[k];
                    if (wrap) {                                // (list entries are images next to the cell: no wrap; uniform per call)
                        dx = (dx > halfL) ? dx - L : dx;
                        dx = (dx < -halfL) ? dx + L : dx;
                    }
                    dd[k] = dx; d2 += dx * dx;
                }
                if (!valid || !(d2 <= rq_ * rq_)) return;
                ++hits;
                snap_hit(H, d2, dd, NDIM, j, xcut_, lnshift_, off, n_oob);
            };
            for (int q = c0; q < c1; ++q) {                                                  // wave-uniform: scalar loads
                const SnapCand &e = P.cand[q];
                visit(e.x, e.y, e.z, e.rq, e.xcut, e.lnshift, e.halo, false);
            }
            for (int q = 0; q < nbig; ++q) {
                const int j = P.big[1 + q];
                const SnapHalo &h = P.hs[j];
                visit(h.x, h.y, h.z, h.rq, h.xcut, h.lnshift, j, true);
            }
            if (valid) {
                double pn[3] = {0.0, 0.0, 0.0};
                for (int k = 0; k < NDIM; ++k) {
                    double v = p[k] + off[k];                  // :262-265
                    v = (v > L) ? v - L : v;                   // :268-273
                    v = (v < 0.0) ? v + L : v;
                    pn[k] = v;
                }
                store_coords<NDIM>(P.out, ip, P.ostride, pn);
            }
        }
    }
    __shared__ unsigned long long s_red[2][4];
    for (int o = 32; o > 0; o >>= 1) { hits += __shfl_down(hits, o, 64); n_oob += __shfl_down(n_oob, o, 64); }
    if ((threadIdx.x & 63) == 0) { s_red[0][threadIdx.x >> 6] = hits; s_red[1][threadIdx.x >> 6] = n_oob; }
    __syncthreads();
    if (threadIdx.x == 0) {
        hits = s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3];
        n_oob = s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3];
        if (hits) atomicAdd((unsigned long long *)&P.stats->pixel_updates, hits);
        if (n_oob) {
            atomicAdd((unsigned long long *)&P.stats->pixels_out_of_table, n_oob);
            if (!P.rdelta) atomicOr(&P.stats->warn_mask, BFG_WARN_R_RANGE);       // BaryonCorrection.py:391-394
        }
    }
}

// Mass deposit of particles on a periodic N^ndim grid: mode 0 = nearest grid point with numpy.histogramdd's bin
// rule on edges linspace(0, L, N + 1) (ParticleSnapshot.make_map, io.py:629-677: right-open bins, the last one
// closed); mode 1 = cloud-in-cell on cell centres (i + 1/2) L / N with periodic wrap.  mass == nullptr: unit masses.
struct DepositParams {
    int ndim, mode, N;
    int64_t n_part;
    double L;
    const double *pos;               // particle k, coordinate c at pos[k * pstride + c]
    const double *mass;              // particle k at mass[k * mstride], or nullptr
    int64_t pstride, mstride;
    double *grid;                    // [N^ndim], C order (x slowest), accumulated into
};

__device__ inline int ngp_bin(double x, double step, int N, double L)
{
    // np.histogramdd: searchsorted(edges, x, 'right') - 1, with x == edges[-1] put into the last bin; outside -> dropped
    if (!(x >= 0.0) || !(x <= L)) return -1;
    int i = (int)floor(x / step);
    i = min(max(i, 0), N - 1);
    // edges as numpy.linspace builds them: i * step, the last one exactly L
    const double e0 = (double)i * step, e1 = (i + 1 == N) ? L : (double)(i + 1) * step;
    if (x < e0) --i; else if (x >= e1 && i + 1 < N) ++i;
    return i;
}

template <int NDIM>
__global__ __launch_bounds__(256) void deposit_kernel(const DepositParams P)
{
    const int64_t ip = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (ip >= P.n_part) return;
    const double m = P.mass ? P.mass[ip * P.mstride] : 1.0;
    const double step = P.L / (double)P.N;
    double xp[3];
    load_coords<NDIM>(P.pos, ip, P.pstride, xp);
    if (P.mode == 0) {
        int64_t c = 0;
        for (int k = 0; k < NDIM; ++k) {
            const int i = ngp_bin(xp[k], step, P.N, P.L);
            if (i < 0) return;
            c = c * P.N + i;
        }
        unsafeAtomicAdd(P.grid + c, m);
    } else {
        int i0[NDIM];
        double w1[NDIM];
        for (int k = 0; k < NDIM; ++k) {
            const double u = xp[k] / step - 0.5;                     // in units of cells, relative to cell centres
            const double f = floor(u);
            w1[k] = u - f;
            int i = (int)f % P.N; if (i < 0) i += P.N;
            i0[k] = i;
        }
        for (int corner = 0; corner < (1 << NDIM); ++corner) {
            double w = m;
            int64_t c = 0;
            for (int k = 0; k < NDIM; ++k) {
                const int bit = (corner >> k) & 1;
                w *= bit ? w1[k] : 1.0 - w1[k];
                int i = i0[k] + bit; if (i >= P.N) i -= P.N;
                c = c * P.N + i;
            }
            if (w != 0.0) unsafeAtomicAdd(P.grid + c, w);
        }
    }
}


// ---- tile-privatised deposit ------------------------------------------------------------------------------------
// The direct kernel above issues 2^ndim scattered f64 atomics per particle (1.1e9 for 512^3 particles: 24 ms, bound
// by the L2 atomic rate of ~4.5e10/s).  Here the particles are first grouped by the grid tile (16^3 / 64^2 cells)
// of their lower CIC corner in one pass over the particles (the wave-aggregated counting atomic returns the particle's slot in
// its tile's region of a 4-byte index array), then one workgroup per tile accumulates its particles in LDS
// (ds_add_f64, ~2e12/s) and flushes the (T + 1)^ndim block once: cells no other tile can touch by a plain
// read-add-write, the shared faces by global atomics.  Weights are computed exactly as in deposit_kernel.
template <int NDIM> struct DepTile {
    static constexpr int T = (NDIM == 3) ? 16 : 64;
    static constexpr int E = T + 1;
    static constexpr int NE = (NDIM == 3) ? E * E * E : E * E;
};

struct DepSortParams {
    DepositParams d;
    int nt;                          // tiles per dimension = ceil(N / T)
    int cap;                         // particle slots per tile in perm[]
    int32_t *count;                  // [ntile] particles per tile (may exceed cap)
    int32_t *perm;                   // [ntile][cap] particle indices, slot = the particle's rank in its tile
    unsigned long long *ovf_n;       // number of particles whose tile had no slot left ...
    int32_t *ovf;                    // [ovf_cap] ... and their indices
    int64_t ovf_cap;
};

// lower corner (CIC) / bin (NGP) of a particle along one axis; false = dropped
template <int MODE>
__device__ inline bool dep_cell(double x, double step, int N, double L, int &i0, double &w1)
{
    if (MODE == BFG_DEPOSIT_NGP) {
        i0 = ngp_bin(x, step, N, L); w1 = 0.0;
        return i0 >= 0;
    }
    const double u = x / step - 0.5;
    const double f = floor(u);
    w1 = u - f;
    int i = (int)f % N; if (i < 0) i += N;
    i0 = i;
    return true;
}

template <int NDIM, int MODE>
__global__ __launch_bounds__(256) void dep_key_kernel(const DepSortParams S)
{
    const DepositParams &P = S.d;
    constexpr int T = DepTile<NDIM>::T;
    const double step = P.L / (double)P.N;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t base = (int64_t)blockIdx.x * blockDim.x; base < P.n_part; base += stride) {     // wave-uniform trip count
        const int64_t ip = base + threadIdx.x;
        int key = -1;
        if (ip < P.n_part) {
            key = 0;
            double xp[3];
            load_coords<NDIM>(P.pos, ip, P.pstride, xp);
            for (int k = 0; k < NDIM; ++k) {
                int i0; double w1;
                if (!dep_cell<MODE>(xp[k], step, P.N, P.L, i0, w1)) { key = -1; break; }
                key = key * S.nt + i0 / T;
            }
        }
        // the counting atomic hands the particle its rank in the tile = its slot in the tile's region of perm[]: grouped
        // in this one pass (no scan, no fill pass, no key array).  Tiles denser than cap (2x the mean) spill into a list
        // that dep_overflow_kernel deposits with global atomics.
        const int slot = wave_group_slot(S.count, key, key >= 0);
        if (key >= 0) {
            if (slot < S.cap) S.perm[(int64_t)key * S.cap + slot] = (int32_t)ip;
            else {
                const int64_t o = (int64_t)atomicAdd(S.ovf_n, 1ull);
                if (o < S.ovf_cap) S.ovf[o] = (int32_t)ip;
            }
        }
    }
}

// particles that found their tile's slots full: plain deposit with global atomics (as deposit_kernel)
template <int NDIM>
__global__ __launch_bounds__(256) void dep_overflow_kernel(const DepSortParams S)
{
    const DepositParams &P = S.d;
    const int64_t n = min((int64_t)*S.ovf_n, S.ovf_cap);
    const double step = P.L / (double)P.N;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) {
        const int64_t ip = S.ovf[q];
        const double m = P.mass ? P.mass[ip * P.mstride] : 1.0;
        double xp[3];
        load_coords<NDIM>(P.pos, ip, P.pstride, xp);
        if (P.mode == BFG_DEPOSIT_NGP) {
            int64_t c = 0;
            bool ok = true;
            for (int k = 0; k < NDIM; ++k) {
                const int i = ngp_bin(xp[k], step, P.N, P.L);
                if (i < 0) { ok = false; break; }
                c = c * P.N + i;
            }
            if (ok) unsafeAtomicAdd(P.grid + c, m);
        } else {
            int i0[NDIM];
            double w1[NDIM];
            for (int k = 0; k < NDIM; ++k) (void)dep_cell<BFG_DEPOSIT_CIC>(xp[k], step, P.N, P.L, i0[k], w1[k]);
            for (int corner = 0; corner < (1 << NDIM); ++corner) {
                double w = m;
                int64_t c = 0;
                for (int k = 0; k < NDIM; ++k) {
                    const int bit = (corner >> k) & 1;
                    w *= bit ? w1[k] : 1.0 - w1[k];
                    int i = i0[k] + bit; if (i >= P.N) i -= P.N;
                    c = c * P.N + i;
                }
                if (w != 0.0) unsafeAtomicAdd(P.grid + c, w);
            }
        }
    }
}

constexpr int kDepThreads = 512;

template <int NDIM, int MODE>
__global__ __launch_bounds__(kDepThreads) void dep_tile_kernel(const DepSortParams S)
{
    const DepositParams &P = S.d;
    constexpr int T = DepTile<NDIM>::T, E = DepTile<NDIM>::E, NE = DepTile<NDIM>::NE;
    __shared__ double acc[NE];
    const int tile = blockIdx.x;
    const int64_t q0 = (int64_t)tile * S.cap, q1 = q0 + min(S.count[tile], S.cap);
    if (q0 == q1) return;
    int tc[3] = {0, 0, 0};
    { int rem = tile; for (int k = NDIM - 1; k >= 0; --k) { tc[k] = rem % S.nt; rem /= S.nt; } }
    for (int e = threadIdx.x; e < NE; e += kDepThreads) acc[e] = 0.0;
    __syncthreads();
    const double step = P.L / (double)P.N;
    constexpr int U = 2;                                       // particles per thread and trip: independent gathers (4: the same time)
    for (int64_t qb = q0 + threadIdx.x; qb < q1; qb += U * kDepThreads) {
        int64_t ip[U];
        double x[U][NDIM], m[U];
        for (int u = 0; u < U; ++u) ip[u] = (qb + u * kDepThreads < q1) ? S.perm[qb + u * kDepThreads] : -1;
        for (int u = 0; u < U; ++u) {
            if (ip[u] < 0) continue;
            load_coords<NDIM>(P.pos, ip[u], P.pstride, x[u]);
            m[u] = P.mass ? P.mass[ip[u] * P.mstride] : 1.0;
        }
        for (int u = 0; u < U; ++u) {
            if (ip[u] < 0) continue;
            int l0[NDIM];
            double w1[NDIM];
            for (int k = 0; k < NDIM; ++k) {
                int i0;
                (void)dep_cell<MODE>(x[u][k], step, P.N, P.L, i0, w1[k]);
                l0[k] = i0 - tc[k] * T;
            }
            if (MODE == BFG_DEPOSIT_NGP) {
                int e = 0;
                for (int k = 0; k < NDIM; ++k) e = e * E + l0[k];
                unsafeAtomicAdd(&acc[e], m[u]);
            } else {
                for (int corner = 0; corner < (1 << NDIM); ++corner) {
                    double w = m[u];
                    int e = 0;
                    for (int k = 0; k < NDIM; ++k) {
                        const int bit = (corner >> k) & 1;
                        w *= bit ? w1[k] : 1.0 - w1[k];
                        e = e * E + l0[k] + bit;
                    }
                    if (w != 0.0) unsafeAtomicAdd(&acc[e], w);
                }
            }
        }
    }
    __syncthreads();
    // Flush in two passes: (1) the cells no other tile writes (the interior of the block) by a plain read-add-write, (2) the shared
    // faces by atomics that nothing waits for.  (In one loop over all cells every read-add-write waited -- s_waitcnt vmcnt(0) --
    // for the acknowledgement of the atomic the wavefront had issued just before it: dep_tile_kernel 2.60 -> 1.5 ms at 512^3
    // particles.  Reading several owned cells ahead adds nothing on top.)
    auto cell_of = [&](int e, bool &own) -> int64_t {
        int rem = e;
        int64_t cc = 0, mul = 1;
        own = true;
        for (int k = NDIM - 1; k >= 0; --k) {
            const int l = rem % E; rem /= E;
            int gk = tc[k] * T + l;
            own = own && (l >= 1) && (l <= T - 1) && (gk < P.N);
            if (gk >= P.N) gk -= P.N;                          // only g == N carries weight (the periodic +1 neighbour)
            cc += (int64_t)gk * mul; mul *= P.N;
        }
        return cc;
    };
    for (int e = threadIdx.x; e < NE; e += kDepThreads) {
        const double v = acc[e];
        if (v == 0.0) continue;
        bool own;
        const int64_t c = cell_of(e, own);
        if (own) P.grid[c] += v;                               // no other tile writes this cell
    }
    for (int e = threadIdx.x; e < NE; e += kDepThreads) {
        const double v = acc[e];
        if (v == 0.0) continue;
        bool own;
        const int64_t c = cell_of(e, own);
        if (!own) unsafeAtomicAdd(P.grid + c, v);
    }
}

}  // namespace bfg
