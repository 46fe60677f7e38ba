// bfg_wtile.hpp -- shell_wave_kernel: the sky-tile kernel with WAVE-PRIVATE chunks.
//
// Same tiles, same binning, same work list, same per-pixel arithmetic as shell_tile_kernel (bfg_tile.hpp); what changes
// is who does what inside the workgroup.  shell_tile_kernel runs every chunk of (halo, tile) pairs through three
// workgroup-wide stages separated by barriers (pair records by wave 0, one thread per (pair, ring) slot, one thread per
// pixel): round 1 measured ~15 % of the workgroup's cycles in those barriers and in the wave-0-only stage, and at 1e5
// halos (one or two chunks per tile) the serial chain prologue -> a -> b -> c -> epilogue IS the kernel.
//
// Here every wavefront owns its chunks: it takes the next KP pairs of the tile's list from a cursor in LDS, forms their
// (pair, ring) slots, segment records and pixel list in its own slice of LDS, paints them with its own 64 lanes, and comes
// back for more.  Inside the chunk loop there is no s_barrier: LDS operations of one wavefront execute in order, so a
// wave's own writes are visible to its later reads, and the tile accumulator is shared through ds_add_f64 as before.  The
// eight wavefronts of a workgroup only meet twice -- after the prologue (accumulator cleared, ln / exp tables, ring
// rows) and before the write-back.  Latencies (pair list -> halo record -> row-window DMA) are hidden by prefetching one
// and two chunks ahead in registers and by the other three wavefronts on the SIMD.
//
// Pixels whose table cell lies below the staged row window (two dependent rounds of global loads) go to a small
// wave-private queue that is drained once per round of slots.  Windows must be the 32-node LDS-staged ones.
#pragma once
#include "bfg_tile.hpp"

namespace bfg {

constexpr int kWaveThreads = 512;
constexpr int kWaveWaves = kWaveThreads / 64;

template <int MODE> struct WaveCfg;
template <> struct WaveCfg<MODE_PAINT> {
    // pairs per chunk, segment records per round (64 slots + second pieces of wrapped windows), pixel -> segment
    // table entries per pass, deferred-pixel queue entries
    static constexpr int TR = TileCfg<MODE_PAINT>::TR, TW = TileCfg<MODE_PAINT>::TW, NACC = 1, KP = 8, NSEG = 72, PIXR = 1024, QW = 16;
    using Pair = PairInfo;
};
template <> struct WaveCfg<MODE_BARYONIFY> {
    static constexpr int TR = TileCfg<MODE_BARYONIFY>::TR, TW = TileCfg<MODE_BARYONIFY>::TW, NACC = 3, KP = 7, NSEG = 68, PIXR = 512, QW = 0;
    using Pair = PairInfoDisp;
};

template <int MODE>
struct WaveLayout {
    using Cfg = WaveCfg<MODE>;
    static constexpr int acc_off = 0;                                                            // double [TR*TW*NACC]
    static constexpr int logtab_off = acc_off + Cfg::TR * Cfg::TW * Cfg::NACC * (int)sizeof(double);
    static constexpr int exptab_off = logtab_off + kLogTab * (int)sizeof(double2);
    static constexpr int rows_off = exptab_off + kExpTab * (int)sizeof(double);                  // RingRow [TR]
    static constexpr int ctl_off = rows_off + Cfg::TR * (int)sizeof(RingRow);                    // int32 [4]: pair cursor
    static constexpr int wave0_off = ctl_off + 16;
    // one wavefront's slice
    static constexpr int ws_seg = 0;                                                             // Seg [NSEG]
    static constexpr int ws_win = ws_seg + Cfg::NSEG * (int)sizeof(Seg);                         // double [KP][kWinLds]
    static constexpr int ws_pair = ws_win + Cfg::KP * kWinLds * (int)sizeof(double);             // Pair [KP]
    static constexpr int ws_ptab = ws_pair + Cfg::KP * (int)sizeof(typename Cfg::Pair);          // uint8 [PIXR]
    static constexpr int ws_q = ws_ptab + Cfg::PIXR;                                             // DeferredPixel [QW]
    static constexpr int wave_bytes = (ws_q + Cfg::QW * (int)sizeof(DeferredPixel) + 15) & ~15;
    static constexpr int total = wave0_off + kWaveWaves * wave_bytes;
};

template <int MODE>
__host__ __device__ constexpr size_t wave_lds_bytes() { return (size_t)WaveLayout<MODE>::total; }
static_assert(wave_lds_bytes<MODE_PAINT>() <= 81920 && wave_lds_bytes<MODE_BARYONIFY>() <= 81920, "two workgroups per CU");

template <int MODE>
__global__ __launch_bounds__(kWaveThreads, 4) void shell_wave_kernel(const TileParams P)
{
    using Cfg = WaveCfg<MODE>;
    using Lay = WaveLayout<MODE>;
    using Pair = typename Cfg::Pair;
    constexpr int TR = Cfg::TR, TW = Cfg::TW, NT = kWaveThreads, NACC = Cfg::NACC;
    constexpr int KP = Cfg::KP, NSEG = Cfg::NSEG, PIXR = Cfg::PIXR, QW = Cfg::QW;
    static_assert(KP * 16 <= 128 && NSEG <= 255 && NSEG % 4 == 0 && PIXR % 64 == 0, "chunk shape");
    static_assert(sizeof(RingRow) % 16 == 0 && sizeof(Pair) % 16 == 0 && Lay::wave_bytes % 16 == 0, "16-byte aligned LDS records");
    extern __shared__ __align__(16) unsigned char smem_raw[];
    double *acc = reinterpret_cast<double *>(smem_raw + Lay::acc_off);
    double2 *logtab = reinterpret_cast<double2 *>(smem_raw + Lay::logtab_off);
    double *exptab = reinterpret_cast<double *>(smem_raw + Lay::exptab_off);
    RingRow *rows = reinterpret_cast<RingRow *>(smem_raw + Lay::rows_off);
    int32_t *ctl = reinterpret_cast<int32_t *>(smem_raw + Lay::ctl_off);

    if ((int)blockIdx.x >= *P.n_work || (long long)P.tile_start[P.geo.ntiles] > P.pair_cap) return;
    const int4 wk = P.work[2 * blockIdx.x], wg = P.work[2 * blockIdx.x + 1];
    const int n_pairs = wk.z - wk.y;
    const Hpx &hp = P.hpx;
    const DevTable &T = P.tab;
    const int band = wg.x, sector = wg.y, NS = wg.z;
    const int ring_lo = 1 + band * TR;
    const int ring_hi = min((int)(4 * hp.nside - 1), ring_lo + TR - 1);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int32_t *plist = P.pairs + wk.y;

    // this wavefront's first two chunks and their pair ids are requested before the prologue's barrier
    if (tid == 0) { ctl[0] = 2 * kWaveWaves * KP; ctl[1] = 0; ctl[2] = 0; ctl[3] = 0; }   // chunks 0 .. 2 * waves - 1 are handed out statically
    int base_cur = wave * KP, base_nxt = (kWaveWaves + wave) * KP;
    auto load_id = [&](int base) -> int { return (lane < KP && base + lane < n_pairs) ? plist[base + lane] : -1; };
    int id_cur = load_id(base_cur);
    int id_nxt = load_id(base_nxt);

    for (int i = tid; i < TR * TW * NACC; i += NT) acc[i] = 0.0;
    if (tid < kLogTab) logtab[tid] = P.logtab[tid];
    if (tid < kExpTab) exptab[tid] = P.exptab[tid];
    if (tid < TR) {
        const int ring = ring_lo + tid;
        RingRow rr;
        rr.z = 0; rr.sth = 0; rr.phistep = 0; rr.phioff = 0; rr.nr = 1; rr.k0 = 0; rr.k1 = 0; rr.rowoff = 0;
        if (ring <= ring_hi) {
            const RingGeom g = ring_geom(hp, ring);
            rr.z = g.z;                // identical formula to ring2z, which query_disc uses
            rr.sth = g.sth; rr.phistep = g.phistep; rr.phioff = g.phioff; rr.nr = g.nr;
            rr.k0 = (int)(((int64_t)sector * g.nr) / NS);
            rr.k1 = (int)(((int64_t)(sector + 1) * g.nr) / NS);
            rr.rowoff = tid * TW - rr.k0;
        }
        rows[tid] = rr;
    }
    const double inv_dr = T.inv_dr;
    const double t_c = (-T.r0) * inv_dr, t_m = 0.5 * inv_dr;     // cell coordinate t = ln(x) * t_m + t_c
    const double t_c1 = t_c + 1.0 - kLogBias * t_m;              // t + 1 from the exponent-biased logarithm
    // segment records carry absolute LDS byte addresses: the dynamic LDS block (the kernel has no static __shared__)
    // starts at address 0; refuse to run otherwise
    if ((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem_raw != 0u) {
        if (threadIdx.x == 0) atomicOr(&P.stats->warn_mask, 0x80000000u);
        return;
    }
    constexpr unsigned lds_base = 0u;
    const int NRm1 = T.nr - 1;
    constexpr int W = kWinLds;
    const unsigned wbase = Lay::wave0_off + wave * Lay::wave_bytes;
    Seg *wsegs = reinterpret_cast<Seg *>(smem_raw + wbase + Lay::ws_seg);
    double *wwin = reinterpret_cast<double *>(smem_raw + wbase + Lay::ws_win);
    Pair *wpair = reinterpret_cast<Pair *>(smem_raw + wbase + Lay::ws_pair);
    uint8_t *wptab = smem_raw + wbase + Lay::ws_ptab;
    [[maybe_unused]] DeferredPixel *wq = reinterpret_cast<DeferredPixel *>(smem_raw + wbase + Lay::ws_q);
    const int win_byte0 = (int)wbase + Lay::ws_win;

    unsigned long long my_pixels = 0;
    unsigned int n_oob32 = 0;
    int qcount = 0;                                              // wave-uniform fill of the deferred-pixel queue
    const int qcap = (P.debug & 32) ? min(2, QW) : QW;           // debug bit 32: tiny queue (tests the inline fallback)

    // the halo record fields the pair lanes need (prefetched one chunk ahead)
    int r_first = 0, r_last = -1, r_wl = 0;
    [[maybe_unused]] double r_lnpf = 0.0;
    auto load_rec = [&](int id, int &first, int &last, int &wl, double &lnpf) {
        first = 0; last = -1; wl = 0; lnpf = 0.0;
        if (id >= 0) {
            const HaloTile &h = P.ht[id];
            first = h.rfirst; last = h.rlast; wl = h.win_lo;
            if constexpr (MODE == MODE_PAINT) lnpf = h.spare[0];
        }
    };
    load_rec(id_cur, r_first, r_last, r_wl, r_lnpf);

    // rare: a pixel whose table cell lies outside the pair's staged row window -> blend the corners directly
    auto direct_row_halo = [&](int64_t j, double t) -> double {
        const int i = min(max((int)t, 0), NRm1 - 1);
        const double f = t - (double)i;
        double c0v, c1v;
        halo_row_pair(T, P.ht, P.cidx, P.cw, P.cap, j, i, c0v, c1v);
        return fma(f, c1v - c0v, c0v);
    };
    auto direct_row = [&](int pidx, double t) -> double {
        double L = direct_row_halo(wpair[pidx].halo, t);
        if constexpr (MODE == MODE_PAINT) L += wpair[pidx].lnpf;
        return L;
    };
    // interpolant of the pair's blended row; see shell_tile_kernel::window_row
    auto window_row = [&](const Seg &sg, int wl1, double t1, bool &in) -> double {
        const int i1 = (int)t1;                                           // saturating conversion
        const int ic = med3_i32(i1, wl1, wl1 + W - 2);
        in = (ic == i1);
        const double f = t1 - (double)ic;
        const lds_double *wp = lds_ptr<double>(lds_base + sg.wbyte + 8 * ic);
        const double B0 = wp[0], B1 = wp[1];
        return fma(f, B1 - B0, B0);
    };
    // one pixel: segment record -> chord^2 -> ln -> row read-out -> accumulate in LDS (same arithmetic as shell_tile_kernel)
    auto do_pixel = [&](int q, const Seg &sg) {
        const int k = q - sg.excl;                                         // pixel index inside the segment
        const double h = fma((double)k, sg.hstep, sg.c0);
        const double h2 = h * h;
        if constexpr (MODE == MODE_PAINT) {
            double s2 = sin_squared_small(h2);
            if (__any(h2 > kSinSmall)) {                                   // wave-uniform branch: only near the poles
                if (h2 > kSinSmall) s2 = sin_squared_wide(h);
            }
            const double x = fma(sg.Bq, s2, sg.Aq);                        // r_com^2
            const double t1 = fma(fast_log_biased(x, logtab), t_m, t_c1);
            const int wl1 = sg.pk & 0xFFFFFF;
            bool in;
            double L = window_row(sg, wl1, t1, in);
            bool slow = false;
            double tq = 0.0;
            if (!in) {
                const double t = t1 - 1.0;
                if ((t >= 0.0) && (t <= (double)NRm1)) { slow = true; tq = t; }
                else n_oob32 += 1;
            }
            if (__any(slow)) {                                             // rare: queue for the batch at the end of the round
                const unsigned long long m = __ballot(slow);
                const int pidx = sg.pk >> 24;
                if (slow) {
                    const int qi = qcount + __popcll(m & ((1ull << lane) - 1ull));
                    if (qi < qcap) {
                        DeferredPixel e;
                        e.halo = wpair[pidx].halo; e.abyte = sg.abyte + 8 * k; e.t = tq;
                        wq[qi] = e;
                    } else { L = direct_row(pidx, tq); in = true; }        // queue full: inline
                }
                qcount = min(qcount + __popcll(m), QW + 64);
            }
            const bool go = in && (fabs(L) < 709.0);                       // false for NaN too
            const double v = fast_exp(L, exptab);                          // garbage when !go, never added
            if (go) __hip_atomic_fetch_add(lds_ptr<double>(lds_base + sg.abyte + 8 * k), v, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            // HealpixRunner.py:336-355 for one pixel
            const int pidx = sg.pk & 63, wl1 = sg.pk >> 12;
            const Pair &pi = wpair[pidx];
            const RingRow &rr = rows[(sg.pk >> 6) & 63];
            double sh = sin_small(h, h2), ch = sqrt(1.0 - sh * sh);        // sin, cos of dphi/2 (cos >= 0)
            if (__any(h2 > kSinSmall)) { if (h2 > kSinSmall) sincos_wide(h, sh, ch); }
            const double s2 = sh * sh;
            const double x = fma(sg.Bq, s2, sg.Aq);                        // r_com^2
            const double sd = 2.0 * sh * ch, cd = 1.0 - 2.0 * s2;          // sin, cos of dphi
            const double cphi = pi.cp0 * cd - pi.sp0 * sd, sphi = pi.sp0 * cd + pi.cp0 * sd;
            const double vx = rr.sth * cphi, vy = rr.sth * sphi, vz = rr.z;                 // pixel unit vector
            const double dx = vx - pi.st * pi.cp0, dy = vy - pi.st * pi.sp0, dz = vz - pi.ct;   // vec - vec_j
            const double t1 = fma(fast_log_biased(x, logtab), t_m, t_c1 + pi.tshift);
            bool in;
            double d = window_row(sg, wl1, t1, in);                        // comoving displacement; table holds d
            if (!in) {
                const double t = t1 - 1.0;
                if ((t >= 0.0) && (t <= (double)NRm1)) { d = direct_row(pidx, t); in = true; }
                else n_oob32 += 1;
            }
            const bool use = in && (x < pi.xcut) && (x > 0.0) && (fabs(d) < 1.0e300);
            d = use ? d * pi.a : 0.0;                                      // physical (HealpixRunner.py:345)
            if (d != 0.0) {
                const double rc = sqrt(x);                                 // r_com; chord = rc a / D
                const double chord = rc * pi.a / pi.D;
                const double qq = pi.D * d * chord + d * d;                // |pos + off|^2 - D^2
                const double nwn = sqrt(fma(pi.D, pi.D, qq));
                const double g = -qq / (nwn * (nwn + pi.D));               // D / |nw| - 1 without cancellation
                const double kk = d / (chord * nwn);                       // offset along (vec - vec_j) / chord, / |nw|
                lds_double *ap = lds_ptr<double>(lds_base + sg.abyte + 24 * k);
                __hip_atomic_fetch_add(ap + 0, fma(vx, g, dx * kk), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(ap + 1, fma(vy, g, dy * kk), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(ap + 2, fma(vz, g, dz * kk), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    };

    __syncthreads();                                             // accumulator, tables and ring rows are in place

    while (base_cur < n_pairs) {
        // ---- requests for later chunks: base of chunk n + 2 (cursor in LDS), its pair ids, the records of chunk n + 1 ----
        int base_nn = 0;
        if (lane == 0) base_nn = atomicAdd(&ctl[0], KP);
        base_nn = __builtin_amdgcn_readfirstlane(base_nn);
        const int id_nn = load_id(base_nn);
        int n_first, n_last, n_wl;
        double n_lnpf;
        load_rec(id_nxt, n_first, n_last, n_wl, n_lnpf);

        // ---- pair records of this chunk: one lane per pair ----------------------------------------------------------
        const int j = id_cur;
        const bool valid = j >= 0;
        const int ra = max(r_first, ring_lo);
        const int nrings = valid ? max(0, min(r_last, ring_hi) - ra + 1) : 0;
        const int cum = wave_scan_incl(nrings);
        const int S = __builtin_amdgcn_readlane(cum, 63);        // (pair, ring) slots of the chunk
        const int myoff = valid ? cum - nrings : 0x7fffffff;     // first slot of the lane's pair
        if (lane < KP) {
            Pair &pi = wpair[lane];
            pi.hoff = (int64_t)j * W; pi.win_lo = r_wl; pi.halo = j; pi.ra = ra; pi.pad = myoff;
            if constexpr (MODE == MODE_PAINT) pi.lnpf = r_lnpf;
            else if (valid) {
                const HaloTile &h = P.ht[j];
                const HaloDisp &hd = P.hd[j];
                pi.cp0 = hd.cp0; pi.sp0 = hd.sp0; pi.st = h.st; pi.ct = h.ct;
                pi.a = hd.a; pi.D = hd.D; pi.xcut = hd.xcut; pi.tshift = hd.tshift; pi.a_over_D = hd.a / hd.D;
            }
        }
        // row windows of the chunk's pairs: LDS-DMA, 16 lanes x 16 B per pair, asynchronous (waited for before the first pixel)
        if (S > 0 && !(P.debug & 64)) {
#pragma unroll
            for (int it = 0; it < (KP * 16 + 63) / 64; ++it) {
                const int i = it * 64 + lane;
                const int pj = __shfl(j, (i >> 4) & 63, 64);
                if (i < KP * 16 && pj >= 0) {
                    const double *src = P.hwin + (int64_t)pj * W + 2 * (i & 15);
                    double *dst = wwin + 2 * (it * 64);                      // wave-uniform; lane * 16 B added by the hardware
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                     (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
                }
            }
        }
        bool windows_pending = true;

        // ---- rounds of 64 slots: one lane per (pair, ring) slot, then one lane per pixel ----------------------------------
        for (int r0 = 0; r0 < S; r0 += 64) {
            const int slot = r0 + lane;
            const bool live = slot < S;
            int p = 0;
#pragma unroll
            for (int k = 1; k < KP; ++k) p += (slot >= __builtin_amdgcn_readlane(myoff, k)) ? 1 : 0;
            int cnt1 = 0, cnt2 = 0, aa1 = 0, aa2 = 0, ab1 = 0, ab2 = 0;
            Seg sg;
            sg.excl = 0; sg.abyte = 0; sg.wbyte = 0; sg.pk = 0; sg.hstep = 0; sg.c0 = 0; sg.Aq = 0; sg.Bq = 0;
            if (live) {
                const Pair &pp = wpair[p];
                const int pj = pp.halo;
                const int ring = pp.ra + (slot - pp.pad);
                const int row = ring - ring_lo;
                const RingRow rr = rows[row];
                const HaloTile &h = P.ht[pj];
                const double st = h.st, ct = h.ct, pphi = h.pphi, S2 = h.S;
                const int nr = rr.nr;
                const int irmin = h.irmin, irmax = h.irmax;
                int lo = 0, cnt = 0;
                {
                    const double x = (h.cosr - rr.z * h.z0) * h.xa;
                    const double ysq = 1.0 - rr.z * rr.z - x * x;
                    const double dphi = (ysq > 0.0) ? atan2_upper(sqrt_unit(ysq), x) : 0.0;
                    if (dphi > 0.0) {
                        const double shift = (rr.phioff != 0.0) ? 0.5 : 0.0;
                        const double fn = (double)nr * kInvTwoPi;
                        const int l32 = (int)floor(fn * (pphi - dphi) - shift) + 1;
                        const int h32 = (int)floor(fn * (pphi + dphi) - shift);
                        const int c = min(h32 - l32 + 1, nr);
                        if (c > 0) { cnt = c; lo = l32; }                // unwrapped: lo in (-nr, 1.5 nr)
                    }
                }
                if (ring < irmin || ring > irmax) { cnt = nr; lo = 0; }  // ring entirely inside the disc
                const int wl = pp.win_lo;
                sg.wbyte = win_byte0 + p * (kWinLds * 8) - 8 * (wl + 1);
                if constexpr (MODE == MODE_PAINT) sg.pk = (wl + 1) | (p << 24);
                else sg.pk = p | (row << 6) | ((wl + 1) << 12);
                sg.hstep = 0.5 * rr.phistep;
                sg.c0 = 0.5 * (rr.phioff * rr.phistep - pphi);          // + first pixel * hstep, below
                const double ds = rr.sth - st, dz = rr.z - ct;
                sg.Aq = (ds * ds + dz * dz) * S2;
                sg.Bq = 4.0 * rr.sth * st * S2;
                if (cnt > 0) {
                    const bool wraps = (lo < 0) || (lo + cnt > nr);
                    if (!__any(wraps)) {                                // the usual case for a whole wavefront
                        const int aa = max(lo, rr.k0), bb = min(lo + cnt, rr.k1);
                        if (bb > aa) { cnt1 = bb - aa; aa1 = aa; ab1 = Lay::acc_off + 8 * NACC * (rr.rowoff + aa); }
                    } else {
#pragma unroll
                        for (int mi = 0; mi < 3; ++mi) {
                            const int m = (mi == 0) ? 0 : (mi == 1 ? -1 : 1);
                            const int aa = max(lo, rr.k0 + m * nr), bb = min(lo + cnt, rr.k1 + m * nr);
                            if (bb > aa) {
                                const int ab = Lay::acc_off + 8 * NACC * (rr.rowoff - m * nr + aa);
                                if (cnt1 == 0) { cnt1 = bb - aa; aa1 = aa; ab1 = ab; }
                                else if (cnt2 == 0) { cnt2 = bb - aa; aa2 = aa; ab2 = ab; }
                            }
                        }
                    }
                }
            }
            // a second piece (a window that wraps around inside the sector) takes one of the extra segment records; should
            // the round run out of them the piece is painted right here through the direct read-out
            const unsigned long long m2 = __ballot(cnt2 > 0);
            const int idx2 = 64 + __popcll(m2 & ((1ull << lane) - 1ull));
            const bool spill2 = (cnt2 > 0) && (idx2 >= NSEG || (P.debug & 16));
            const int tot = cnt1 + (spill2 ? 0 : cnt2);
            const int incl = wave_scan_incl(tot);
            const int ptotal = (P.debug & 2) ? 0 : __builtin_amdgcn_readlane(incl, 63);
            const int e1 = incl - tot, e2 = e1 + cnt1;
            const double c0b = sg.c0;
            // pixel -> segment table of pixels [pbase, pbase + PIXR) of the round
            auto fill = [&](int pbase) {
                if (cnt1 > 0) {
                    const int a0 = max(e1, pbase) - pbase, a1 = min(e1 + cnt1, pbase + PIXR) - pbase;
                    for (int q = a0; q < a1; ++q) wptab[q] = (uint8_t)lane;
                    if (cnt2 > 0 && !spill2) {
                        const int b0 = max(e2, pbase) - pbase, b1 = min(e2 + cnt2, pbase + PIXR) - pbase;
                        for (int q = b0; q < b1; ++q) wptab[q] = (uint8_t)idx2;
                    }
                }
            };
            if (cnt1 > 0) {
                sg.excl = e1; sg.abyte = ab1; sg.c0 = fma((double)aa1, sg.hstep, c0b);
                wsegs[lane] = sg;
                if (cnt2 > 0 && !spill2) {
                    Seg s2 = sg;
                    s2.excl = e2; s2.abyte = ab2; s2.c0 = fma((double)aa2, sg.hstep, c0b);
                    wsegs[idx2] = s2;
                }
            }
            fill(0);
            if (windows_pending) {                               // the chunk's row windows (and the prefetches) have landed
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                windows_pending = false;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (cnt1 > 0 && cnt2 > 0 && spill2) {
                constexpr int wlf = 1 << 18;                   // a window no cell falls into -> direct_row()
                Seg s2 = sg;
                s2.excl = 0; s2.abyte = ab2; s2.c0 = fma((double)aa2, sg.hstep, c0b);
                s2.wbyte = win_byte0 + p * (kWinLds * 8) - 8 * wlf;
                if constexpr (MODE == MODE_PAINT) s2.pk = wlf | (p << 24);
                else s2.pk = (sg.pk & 0xFFF) | (wlf << 12);
                const int q_save = qcount;
                for (int k = 0; k < cnt2; ++k) {
                    // inline, lane by lane: the queue bookkeeping of do_pixel is wave-uniform, so this path reads the row itself
                    const double hh = fma((double)k, s2.hstep, s2.c0);
                    if constexpr (MODE == MODE_PAINT) {
                        const double ss = sin_squared(hh);
                        const double x = fma(s2.Bq, ss, s2.Aq);
                        const double t = fma(fast_log_biased(x, logtab), t_m, t_c1) - 1.0;
                        if ((t >= 0.0) && (t <= (double)NRm1)) {
                            const double L = direct_row(p, t);
                            if (fabs(L) < 709.0)
                                __hip_atomic_fetch_add(lds_ptr<double>(lds_base + s2.abyte + 8 * k), fast_exp(L, exptab),
                                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        } else n_oob32 += 1;
                    } else {
                        qcount = q_save;
                        do_pixel(k, s2);
                    }
                }
                qcount = q_save;
                my_pixels += (unsigned long long)cnt2;
            }
            my_pixels += (lane == 0) ? (unsigned long long)ptotal : 0ull;

            for (int pbase = 0; pbase < ptotal; pbase += PIXR) {
                if (pbase > 0) {                                 // rare: more than PIXR pixels in one round of slots
                    __builtin_amdgcn_wave_barrier();
                    fill(pbase);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
                const int pend = min(ptotal, pbase + PIXR);
                for (int q = pbase + lane; q - lane < pend; q += 64) {       // whole wavefronts step together
                    if (q < pend) {
                        const int sidx = wptab[q - pbase];
                        const Seg sgq = wsegs[sidx];
                        do_pixel(q, sgq);
                    }
                    // lanes past the end skipped the queue bookkeeping; lane 0 never does
                    if constexpr (QW > 0) qcount = __builtin_amdgcn_readfirstlane(qcount);
                }
            }
            // the round's deferred pixels, all lanes at once
            if constexpr (QW > 0) {
                if (qcount > 0) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    const int n = min(qcount, qcap);
                    for (int i = lane; i < n; i += 64) {
                        const DeferredPixel e = wq[i];
                        const double L = direct_row_halo(e.halo, e.t) + P.ht[e.halo].spare[0];
                        if (fabs(L) < 709.0)
                            __hip_atomic_fetch_add(lds_ptr<double>(lds_base + e.abyte), fast_exp(L, exptab), __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    qcount = 0;
                }
            }
            __builtin_amdgcn_wave_barrier();                     // the next round reuses the segment records and the table
        }
        // rotate the prefetch registers
        base_cur = base_nxt; base_nxt = base_nn;
        id_cur = id_nxt; id_nxt = id_nn;
        r_first = n_first; r_last = n_last; r_wl = n_wl; r_lnpf = n_lnpf;
    }
    // counters: one global atomic per workgroup (eight same-address atomics per tile cost 0.15 ms per launch)
    if (my_pixels) atomicAdd(reinterpret_cast<unsigned long long *>(ctl + 2), my_pixels);
    if (n_oob32) atomicAdd(reinterpret_cast<unsigned int *>(ctl + 1), n_oob32);
    __syncthreads();                                             // every wavefront has added its last pixel
    // write the tile back: every pixel belongs to exactly one tile -> plain read-modify-write (atomics where the tile's
    // pair list was cut into several work items)
    constexpr int kPerThread = (TR * TW + NT - 1) / NT;
#pragma unroll
    for (int u = 0; u < kPerThread; ++u) {
        const int i = tid + u * NT;
        if (i >= TR * TW) continue;
        const int row = i / TW, col = i % TW;
        const int ring = ring_lo + row;
        if (ring > ring_hi || rows[row].k0 + col >= rows[row].k1) continue;
        int64_t start, nr64; bool shifted;
        ring_info_small(hp, ring, start, nr64, shifted);
        const int64_t pix = start + rows[row].k0 + col;
#pragma unroll
        for (int c = 0; c < NACC; ++c) {
            const double v = acc[NACC * i + c];
            if (v != 0.0) {
                if (wk.w) unsafeAtomicAdd(P.out + NACC * pix + c, v);   // the tile is shared with other workgroups
                else if (P.out_zero) P.out[NACC * pix + c] = v;         // the caller cleared the map
                else P.out[NACC * pix + c] += v;
            }
        }
    }
    if (tid == 0) {
        const unsigned long long px = *reinterpret_cast<unsigned long long *>(ctl + 2);
        if (px) atomicAdd((unsigned long long *)&P.stats->pixel_updates, px);
        const unsigned int oob = *reinterpret_cast<unsigned int *>(ctl + 1);
        if (oob) {
            atomicAdd((unsigned long long *)&P.stats->pixels_out_of_table, (unsigned long long)oob);
            atomicOr(&P.stats->warn_mask, BFG_WARN_R_RANGE);
        }
    }
}

}  // namespace bfg
