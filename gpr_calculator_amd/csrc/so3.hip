// so3.hip -- SO(3) power-spectrum descriptor and its position derivatives on the GPU.
//
// Replaces the numpy/scipy pipeline of gpr_calc/SO3.py: compute_dcs (:608-727: Gauss-Chebyshev radial
// quadrature x modified spherical Bessel x Y_lm x Gaussian x cosine cutoff and their gradients) and
// the per-centre assembly loop of SO3.calculate (:243-273: c_tot, power spectrum P, dP per neighbour
// pair, scatter into the (i, j) rows of `seq`, self row = -sum of the others).
//
// One 256-thread workgroup per centre atom.  Phase 1: each wavefront takes neighbour pairs of the
// centre in turn -- lanes over quadrature nodes for the radial integrals, over m for the Legendre
// recurrences, over (n, l, m) for the expansion coefficients -- stores dC of the pair in a global
// scratch and accumulates c_tot in wavefront-private LDS.  Phase 2: the four partial c_tot are summed
// in a fixed order, lanes over coefficients emit x_i, and one wavefront per seq row contracts the
// stored dC with conj(c_tot).  No atomics; the summation order is fixed.
#include "gpr_internal.h"

namespace {

constexpr int SO3_LCAP = 7;      // lmax <= 6  (tables hold l = 0 .. lmax+1)
constexpr int SO3_NQCAP = 128;   // quadrature nodes: 10*(nmax+lmax+1) <= 128
constexpr int SO3_WAVES = 4;

struct so3_args {
    int nmax, lmax, nq, d;
    double rcut, alpha;
    const double *G;            // [nmax][nq] radial weights (SO3.py:623-655)
    const double *rq;           // [nq] quadrature radii
    const double *vec;          // [P][3] neighbour displacement
    const double *wj;           // [P] neighbour weight (Z_j, SO3.py:385)
    const int *centre_pair0;    // [nc+1]
    const int *centre_row0;     // [nc+1] seq-row range of each centre
    const int *row_pair0;       // [S+1]  pair range of each seq row
    const int *self_row;        // [nc]   seq row (i, i)
    double *x;                  // [nc][d]
    double *dxdr;               // [S][d][3]
    double2 *dc;                // scratch [P][nmax*L*M][3]
};

struct cplx { double re, im; };
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ cplx cscale(cplx a, double s) { return {a.re * s, a.im * s}; }
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return {a.re + b.re, a.im + b.im}; }

// modified spherical Bessel functions of the first kind i_0 .. i_lt(x), x > 0
// (scipy.special.spherical_in in the reference, SO3.py:563-565).  Miller's downward recurrence for small x,
// closed forms + upward recurrence once x is large enough for the recurrence to be well conditioned.
__device__ void sph_in_table(double x, int lt, double *b) {
    const double xs = fmax(6.0, 2.0 * lt);
    if (x < xs) {
        // Miller's downward recurrence t_{l-1} = t_{l+1} + (2l+1)/x t_l from l = lt+16 (all terms positive, so
        // no cancellation), normalised by i_0 = sinh(x)/x; agrees with scipy to ~3e-14 relative for x < 12
        const double ix = 1.0 / x;
        double tp = 0.0, t = 1e-300;
#pragma unroll
        for (int l = SO3_LCAP - 1 + 16; l > 0; l--) {        // fully unrolled: every index below is static
            if (l <= lt + 16) {
                const double tm = tp + (2 * l + 1) * ix * t;
                tp = t; t = tm;
                if (l - 1 <= SO3_LCAP - 1 && l - 1 <= lt) b[l - 1 <= SO3_LCAP - 1 ? l - 1 : 0] = t;
            }
        }
        double s0;
        if (x > 0.5) { const double e = exp(x); s0 = 0.5 * (e - 1.0 / e) * ix; }
        else { const double x2 = x * x; s0 = 1.0 + x2 / 6.0 * (1.0 + x2 / 20.0 * (1.0 + x2 / 42.0 * (1.0 + x2 / 72.0 * (1.0 + x2 / 110.0)))); }
        const double sc = s0 / b[0];
#pragma unroll
        for (int l = 0; l < SO3_LCAP; l++) if (l <= lt) b[l] *= sc;
    } else {
        // two divisions in all: 1/x and 1/e^x
        const double ix = 1.0 / x;
        const double e = exp(x), em = 1.0 / e, sh = 0.5 * (e - em), ch = 0.5 * (e + em);
        b[0] = sh * ix;
        if (lt >= 1) b[1] = (ch - b[0]) * ix;
#pragma unroll
        for (int l = 1; l < SO3_LCAP - 1; l++) if (l < lt) b[l + 1] = b[l - 1] - (2 * l + 1) * ix * b[l];
    }
}

// dynamic LDS layout (doubles), sized for the actual (nmax, lmax, nq):
//   sG [nmax][nq] | srq [nq] | per wave: sB [L][nq], sD [L][nq] | sI, sdI [4][32] | sY [4][LT+1][LT+1] cplx |
//   sGY [4][L][M][3] cplx | sCw [4][NLM] cplx | sSelf [4][96];   phase 2 re-uses the sB/sD area per wave as
//   the staging buffer of one pair's dC (NLM*3 complex), which the host checks to fit.
struct so3_lds {
    int oG, orq, oB, oD, oI, odI, oY, oGY, oC, oSelf, oNY, oCG, oNL, total, bstride, ystride, gystride, cstride;
};
__host__ __device__ inline so3_lds so3_layout(int nmax, int lmax, int nq) {
    const int L = lmax + 1, M = 2 * lmax + 1, LT = lmax + 2, NLM = nmax * L * M;
    so3_lds o;
    int off = 0;
    o.oG = off; off += nmax * nq;
    o.orq = off; off += nq;
    // per-wave Bessel tables sB | sD (contiguous areas of SO3_WAVES * bs doubles each); in phase 2 the two areas
    // together are re-cut into SO3_WAVES staging buffers of NLM*6 doubles (one pair's dC as complex triples)
    int bs = L * nq;
    const int need = (NLM * 6 * SO3_WAVES + 2 * SO3_WAVES - 1) / (2 * SO3_WAVES);
    if (bs < need) bs = need;
    o.bstride = bs;
    o.oB = off; off += SO3_WAVES * bs;
    o.oD = off; off += SO3_WAVES * bs;
    o.oI = off; off += SO3_WAVES * 32;
    o.odI = off; off += SO3_WAVES * 32;
    o.ystride = LT * LT * 2;
    o.oY = off; off += SO3_WAVES * o.ystride;
    o.gystride = L * M * 3 * 2;
    o.oGY = off; off += SO3_WAVES * o.gystride;
    o.cstride = NLM * 2;
    o.oC = off; off += SO3_WAVES * o.cstride;
    o.oSelf = off; off += SO3_WAVES * 96;
    o.oNY = off; off += LT * LT;            // Y_l^m normalisation, l <= lmax+1, m >= 0
    o.oCG = off; off += L * M * 6;          // the six coefficients of grad Y_l^m (SO3.py:682-707)
    o.oNL = off; off += L;                  // norm_l = sqrt(2 sqrt2 pi / sqrt(2l+1)) (SO3.py:206)
    o.total = off;
    return o;
}

__global__ void __launch_bounds__(256) so3_centre_kernel(so3_args A) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int L = A.lmax + 1, M = 2 * A.lmax + 1, NLM = A.nmax * L * M;
    const int LT = A.lmax + 1;                  // highest l in the Y table (gradients need l+1)
    const int YW = LT + 1;                      // row width of the Y table
    const so3_lds lay = so3_layout(A.nmax, A.lmax, A.nq);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    double *sG = smem + lay.oG, *srq = smem + lay.orq;
    double *sB = smem + lay.oB + wave * lay.bstride, *sD = smem + lay.oD + wave * lay.bstride;
    double *sI = smem + lay.oI + wave * 32, *sdI = smem + lay.odI + wave * 32;
    cplx *sY = (cplx *)(smem + lay.oY + wave * lay.ystride);
    cplx *sGY = (cplx *)(smem + lay.oGY + wave * lay.gystride);
    cplx *sCw = (cplx *)(smem + lay.oC + wave * lay.cstride);
    double *sSelf = smem + lay.oSelf;
    const int ic = blockIdx.x;
    const int p0 = A.centre_pair0[ic], p1 = A.centre_pair0[ic + 1];
    double *sNY = smem + lay.oNY, *sCG = smem + lay.oCG, *sNL = smem + lay.oNL;
    // constant tables (square roots and factorial ratios) evaluated once per workgroup
    for (int e = tid; e < (LT + 1) * (LT + 1); e += 256) {
        const int l = e / (LT + 1), m = e % (LT + 1);
        double v = 0.0;
        if (m <= l) {
            double ratio = 1.0;                          // (l-m)! / (l+m)!
            for (int k = l - m + 1; k <= l + m; k++) ratio /= k;
            v = sqrt((2 * l + 1) / (4.0 * M_PI) * ratio);
        }
        sNY[e] = v;
    }
    for (int e = tid; e < L * M; e += 256) {
        const int l = e / M, m = e % M - A.lmax;
        double c[6] = {0, 0, 0, 0, 0, 0};
        if (l >= 1 && m >= -l && m <= l) {
            const double dl = l;
            c[0] = -sqrt(((dl + 1) * (dl + 1) - m * m) / (2 * dl + 1) / (2 * dl + 3)) * dl;
            if (abs(m) <= l - 1) c[1] = sqrt((dl * dl - m * m) / (2 * dl - 1) / (2 * dl + 1)) * (dl + 1);
            c[2] = -sqrt((dl + m + 1) * (dl + m + 2) / 2 / (2 * dl + 1) / (2 * dl + 3)) * dl;
            if (abs(m + 1) <= l - 1) c[3] = -sqrt((dl - m - 1) * (dl - m) / 2 / (2 * dl - 1) / (2 * dl + 1)) * (dl + 1);
            c[4] = -sqrt((dl - m + 1) * (dl - m + 2) / 2 / (2 * dl + 1) / (2 * dl + 3)) * dl;
            if (abs(m - 1) <= l - 1) c[5] = -sqrt((dl + m - 1) * (dl + m) / 2 / (2 * dl - 1) / (2 * dl + 1)) * (dl + 1);
        }
        for (int k = 0; k < 6; k++) sCG[e * 6 + k] = c[k];
    }
    for (int e = tid; e < L; e += 256) sNL[e] = sqrt(2.0 * sqrt(2.0) * M_PI / sqrt(2.0 * e + 1.0));
    for (int e = tid; e < A.nmax * A.nq; e += 256) sG[e] = A.G[e];
    for (int e = tid; e < A.nq; e += 256) srq[e] = A.rq[e];
    for (int e = lane; e < NLM; e += 64) sCw[e] = {0.0, 0.0};
    __syncthreads();

    auto Ylm = [&](int l, int m) -> cplx {     // any m; zero outside |m| <= l
        if (m > l || -m > l) return {0.0, 0.0};
        if (m >= 0) return sY[l * YW + m];
        cplx v = sY[l * YW - m];
        const double sgn = ((-m) & 1) ? -1.0 : 1.0;
        return {sgn * v.re, -sgn * v.im};
    };
    // Y_l^m (m = lane >= 0, l = m .. lmax+1) and grad Y_l^m of the unit vector (ux, uy, uz), |r| = r
    auto angular_tables = [&](double ux, double uy, double uz, double r) {
        if (lane <= LT) {
            const int m = lane;
            const double ct = uz, rho = sqrt(ux * ux + uy * uy), st = rho;
            const double cphi = rho > 0.0 ? ux / rho : 1.0, sphi = rho > 0.0 ? uy / rho : 0.0;
            double cm = 1.0, sm = 0.0;                       // cos(m phi), sin(m phi)
            for (int k = 0; k < m; k++) { const double c2 = cm * cphi - sm * sphi; sm = sm * cphi + cm * sphi; cm = c2; }
            double pmm = 1.0;                                // P_m^m with the Condon-Shortley phase
            for (int k = 1; k <= m; k++) pmm *= -(2 * k - 1) * st;
            double plm2 = 0.0, plm1 = pmm;
            for (int l = m; l <= LT; l++) {
                double plm;
                if (l == m) plm = pmm;
                else if (l == m + 1) plm = (2 * m + 1) * ct * pmm;
                else plm = ((2 * l - 1) * ct * plm1 - (l + m - 1) * plm2) / (l - m);
                if (l > m) { plm2 = plm1; plm1 = plm; }
                const double nrm = sNY[l * YW + m];
                sY[l * YW + m] = {nrm * plm * cm, nrm * plm * sm};
            }
        }
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
        // gradient of Y_l^m from its covariant spherical components (SO3.py:682-707)
        const double inv_r = 1.0 / r;
        for (int e = lane; e < L * M; e += 64) {
            const int l = e / M, m = e % M - A.lmax;
            cplx g0 = {0, 0}, gp = {0, 0}, gm = {0, 0};
            if (l >= 1 && m >= -l && m <= l) {
                const double *cg = sCG + e * 6;
                g0 = cadd(cscale(Ylm(l + 1, m), cg[0] * inv_r), cscale(Ylm(l - 1, m), cg[1] * inv_r));
                gp = cadd(cscale(Ylm(l + 1, m + 1), cg[2] * inv_r), cscale(Ylm(l - 1, m + 1), cg[3] * inv_r));
                gm = cadd(cscale(Ylm(l + 1, m - 1), cg[4] * inv_r), cscale(Ylm(l - 1, m - 1), cg[5] * inv_r));
            }
            const double is2 = 0.70710678118654752440;
            cplx dmp = {gm.re - gp.re, gm.im - gp.im}, smp = {gm.re + gp.re, gm.im + gp.im};
            sGY[e * 3 + 0] = cscale(dmp, is2);
            sGY[e * 3 + 1] = {-is2 * smp.im, is2 * smp.re};          // i/sqrt(2) * (gm + gp)
            sGY[e * 3 + 2] = g0;
        }
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
    };

    // ---- phase 1: every wavefront walks its share of the centre's neighbour pairs (no block barriers inside:
    //      all tables are wavefront private, ordered by wave_barrier + LDS fences)
    for (int p = p0 + wave; p < p1; p += SO3_WAVES) {
        const double vx = A.vec[3 * p], vy = A.vec[3 * p + 1], vz = A.vec[3 * p + 2];
        const double r = sqrt(vx * vx + vy * vy + vz * vz);
        const double ux = vx / r, uy = vy / r, uz = vz / r;
        // A1: Bessel values at the quadrature nodes (lanes over nodes)
        const int lt = A.lmax > 1 ? A.lmax : 1;
        for (int q = lane; q < A.nq; q += 64) {
            double b[SO3_LCAP + 1];
            const double rqq = srq[q];
            const double xarg = 2.0 * A.alpha * r * rqq;
            sph_in_table(xarg, lt, b);
            const double chain = 2.0 * A.alpha * rqq;           // d/dr of the argument
            const double ixarg = 1.0 / xarg;
#pragma unroll
            for (int l = 0; l <= SO3_LCAP - 1; l++)
                if (l <= A.lmax) {
                    sB[l * A.nq + q] = b[l];
                    const double db = (l == 0) ? b[1] : b[l - 1] - (l + 1) * ixarg * b[l];   // scipy _spherical_in_d
                    sD[l * A.nq + q] = db * chain;
                }
        }
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
        // A2: radial integrals I[n][l], dI/dr: lanes = (n, l) x 4 node quarters, then a 4-lane reduction
        {
            const int e = lane >> 2, part = lane & 3, nl = A.nmax * L;
            for (int e0 = 0; e0 < nl; e0 += 16) {
                const int ee = e0 + e;
                double s0 = 0.0, s1 = 0.0;
                if (ee < nl) {
                    const int n = ee / L, l = ee % L;
                    for (int q = part; q < A.nq; q += 4) {
                        const double g = sG[n * A.nq + q];
                        s0 += g * sB[l * A.nq + q];
                        s1 += g * sD[l * A.nq + q];
                    }
                }
                s0 += __shfl_xor(s0, 1, 64); s0 += __shfl_xor(s0, 2, 64);
                s1 += __shfl_xor(s1, 1, 64); s1 += __shfl_xor(s1, 2, 64);
                if (ee < nl && part == 0) { sI[ee] = s0; sdI[ee] = s1; }
            }
        }
        angular_tables(ux, uy, uz, r);
        // C: expansion coefficients and gradients of this pair (lanes over (n, l, m))
        {
            const double w = A.wj[p];
            const double E = 4.0 * M_PI * exp(-A.alpha * r * r);
            const double gE = -2.0 * A.alpha * r * E;                            // d/dr
            const double fc = 0.5 * (cos(M_PI * r / A.rcut) + 1.0);
            const double dfc = -0.5 * M_PI / A.rcut * sin(M_PI * r / A.rcut);
            const double u[3] = {ux, uy, uz};
            for (int e = lane; e < NLM; e += 64) {
                const int n = e / (L * M), l = (e / M) % L, mi = e % M, m = mi - A.lmax;
                cplx cval = {0, 0}, dcv[3] = {{0, 0}, {0, 0}, {0, 0}};
                if (m >= -l && m <= l) {
                    const double norm = sNL[l] * w;                                               // SO3.py:206, 221
                    const double In = sI[n * L + l], dIn = sdI[n * L + l];
                    const cplx Y = Ylm(l, m);
                    const cplx yin = cscale(Y, In);
                    cval = cscale(yin, E);
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        cplx t = cscale(yin, gE * u[k]);
                        cplx h = cadd(cscale(sGY[(l * M + mi) * 3 + k], In), cscale(Y, dIn * u[k]));
                        t = cadd(t, cscale(h, E));
                        t = cadd(cscale(t, fc), cscale(cval, dfc * u[k]));
                        dcv[k] = cscale(t, norm);
                    }
                    cval = cscale(cval, fc * norm);
                }
                double2 *o = A.dc + ((size_t)p * NLM + e) * 3;
#pragma unroll
                for (int k = 0; k < 3; k++) o[k] = make_double2(dcv[k].re, dcv[k].im);
                sCw[e] = cadd(sCw[e], cval);
            }
        }
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
    }
    __syncthreads();

    // ---- phase 2: c_tot, x_i, dP rows
    cplx *sC0 = (cplx *)(smem + lay.oC);
    for (int e = tid; e < NLM; e += 256) {
        cplx s = sC0[e];
        for (int w2 = 1; w2 < SO3_WAVES; w2++) s = cadd(s, ((cplx *)(smem + lay.oC + w2 * lay.cstride))[e]);
        sC0[e] = s;
    }
    __syncthreads();
    const cplx *ctot = sC0;
    auto decode = [&](int c, int &n1, int &n2, int &l) {
        const int pi = c / L;
        l = c % L;
        n1 = 0;
        while ((n1 + 1) * (n1 + 2) / 2 <= pi) n1++;
        n2 = pi - n1 * (n1 + 1) / 2;
    };
    for (int c = tid; c < A.d; c += 256) {
        int n1, n2, l;
        decode(c, n1, n2, l);
        double s = 0.0;
        for (int mi = 0; mi < M; mi++) {
            const cplx a = ctot[(n1 * L + l) * M + mi], b = ctot[(n2 * L + l) * M + mi];
            s += a.re * b.re + a.im * b.im;
        }
        A.x[(size_t)ic * A.d + c] = s;
    }
    const int r0 = A.centre_row0[ic], r1 = A.centre_row0[ic + 1], rself = A.self_row[ic];
    const int ne = A.d * 3;
    double self0 = 0.0, self1 = 0.0;
    // per lane: the two (coefficient, component) outputs it owns and their c_tot index pairs
    int e_idx[2], i1b[2], i2b[2];
#pragma unroll
    for (int rnd = 0; rnd < 2; rnd++) {
        const int e = lane + 64 * rnd;
        e_idx[rnd] = e;
        int n1 = 0, n2 = 0, l = 0;
        if (e < ne) decode(e / 3, n1, n2, l);
        i1b[rnd] = (n1 * L + l) * M; i2b[rnd] = (n2 * L + l) * M;
    }
    double2 *stage = (double2 *)(smem + lay.oB + wave * NLM * 6);   // wave-private staging of one pair's dC
    for (int row = r0 + wave; row < r1; row += SO3_WAVES) {
        if (row == rself) continue;
        const int q0 = A.row_pair0[row], q1 = A.row_pair0[row + 1];
        double s[2] = {0.0, 0.0};
        for (int p = q0; p < q1; p++) {
            const double2 *dp = A.dc + (size_t)p * NLM * 3;
            __builtin_amdgcn_wave_barrier();
            {   // coalesced copy of the pair's dC: all loads in flight before the first LDS store
                double2 tmp[8];
                const int nt = NLM * 3;
#pragma unroll
                for (int u = 0; u < 8; u++) { const int t = lane + 64 * u; if (t < nt) tmp[u] = dp[t]; }
#pragma unroll
                for (int u = 0; u < 8; u++) { const int t = lane + 64 * u; if (t < nt) stage[t] = tmp[u]; }
                for (int t = lane + 512; t < nt; t += 64) stage[t] = dp[t];
            }
            __builtin_amdgcn_wave_barrier();
            __threadfence_block();
#pragma unroll
            for (int rnd = 0; rnd < 2; rnd++) {
                const int e = e_idx[rnd];
                if (e < ne) {
                    const int k = e % 3;
                    double acc = 0.0;
                    for (int mi = 0; mi < M; mi++) {
                        const int i1 = i1b[rnd] + mi, i2 = i2b[rnd] + mi;
                        const double2 d1 = stage[i1 * 3 + k], d2 = stage[i2 * 3 + k];
                        acc += d1.x * ctot[i2].re + d1.y * ctot[i2].im + d2.x * ctot[i1].re + d2.y * ctot[i1].im;
                    }
                    s[rnd] += acc;
                }
            }
        }
#pragma unroll
        for (int rnd = 0; rnd < 2; rnd++)
            if (e_idx[rnd] < ne) A.dxdr[(size_t)row * ne + e_idx[rnd]] = s[rnd];
        self0 += s[0]; self1 += s[1];
    }
    sSelf[wave * 96 + lane] = self0;
    if (lane < 32) sSelf[wave * 96 + 64 + lane] = self1;
    __syncthreads();
    for (int e = tid; e < ne; e += 256) {
        double s = 0.0;
        for (int w2 = 0; w2 < SO3_WAVES; w2++) s += sSelf[w2 * 96 + e];
        A.dxdr[(size_t)rself * ne + e] = -s;             // SO3.py:266-270
    }
}

}  // namespace

// x[nc, d], dxdr[S, d, 3] from neighbour pairs grouped by centre and, inside a centre, by seq row.
// All pointers are device pointers; scratch must hold P * nmax*(lmax+1)*(2*lmax+1) * 3 complex doubles.
extern "C" int gpr_so3_power_spectrum(void *stream, int nmax, int lmax, double rcut, double alpha, int nq,
                                      const double *G, const double *rq, int ncentres, int npairs, int nrows,
                                      const double *vec, const double *wj, const int *centre_pair0,
                                      const int *centre_row0, const int *row_pair0, const int *self_row,
                                      double *x, double *dxdr, void *scratch) {
    hipStream_t st = (hipStream_t)stream;
    const int d = nmax * (nmax + 1) / 2 * (lmax + 1);
    GPR_REQUIRE(nmax >= 1 && lmax >= 0, "gpr_so3_power_spectrum: bad nmax/lmax");
    if (lmax > SO3_LCAP - 1 || nq > SO3_NQCAP || nmax * (lmax + 1) > 32 || d * 3 > 96 || nq < 1) {
        gpr_set_error("gpr_so3_power_spectrum: (nmax=%d, lmax=%d, nq=%d) outside this build's limits "
                      "(lmax <= %d, nq <= %d, ncoef <= 32)", nmax, lmax, nq, SO3_LCAP - 1, SO3_NQCAP);
        return GPR_ERR_UNSUPPORTED;
    }
    if (ncentres <= 0) return GPR_OK;
    GPR_REQUIRE(G && rq && centre_pair0 && centre_row0 && row_pair0 && self_row && x && dxdr,
                "gpr_so3_power_spectrum: null pointer");
    GPR_REQUIRE(npairs == 0 || (vec && wj && scratch), "gpr_so3_power_spectrum: null pair arrays");
    (void)nrows;
    so3_args A;
    A.nmax = nmax; A.lmax = lmax; A.nq = nq; A.d = d; A.rcut = rcut; A.alpha = alpha;
    A.G = G; A.rq = rq; A.vec = vec; A.wj = wj; A.centre_pair0 = centre_pair0; A.centre_row0 = centre_row0;
    A.row_pair0 = row_pair0; A.self_row = self_row; A.x = x; A.dxdr = dxdr; A.dc = (double2 *)scratch;
    const so3_lds lay = so3_layout(nmax, lmax, nq);
    const size_t lds_bytes = (size_t)lay.total * sizeof(double);
    if (lds_bytes > 150 * 1024) {
        gpr_set_error("gpr_so3_power_spectrum: (nmax=%d, lmax=%d, nq=%d) needs %zu bytes of LDS", nmax, lmax, nq, lds_bytes);
        return GPR_ERR_UNSUPPORTED;
    }
    static bool attr_set = false;
    if (!attr_set) {
        GPR_HIP_TRY(hipFuncSetAttribute((const void *)so3_centre_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        attr_set = true;
    }
    const int pid = gpr_prof_begin(st, GPR_PROF_SO3);
    hipLaunchKernelGGL(so3_centre_kernel, dim3(ncentres), dim3(256), lds_bytes, st, A);
    gpr_prof_end(st, pid);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { gpr_set_error("so3_centre_kernel launch failed: %s", hipGetErrorString(e)); return GPR_ERR_HIP; }
    return GPR_OK;
}
