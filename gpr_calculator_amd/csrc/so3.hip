// so3.hip -- SO(3) power-spectrum descriptor and its position derivatives on the GPU.
//
// Replaces the numpy/scipy pipeline of gpr_calc/SO3.py: compute_dcs (:608-727: Gauss-Chebyshev radial
// quadrature x modified spherical Bessel x Y_lm x Gaussian x cosine cutoff and their gradients) and
// the per-centre assembly loop of SO3.calculate (:243-273: c_tot, power spectrum P, dP per neighbour
// pair, scatter into the (i, j) rows of `seq`, self row = -sum of the others).
//
// Radial part.  For a neighbour at distance r the reference integrates, per (n, l),
//     I_nl(r) = sum_q G[n][q] i_l(2 alpha r r_q)          (80-node quadrature, SO3.py:623-655)
// and only ever uses F0_nl(r) = 4 pi exp(-alpha r^2) I_nl(r) and its derivative F1_nl = dF0_nl/dr.  Both are
// smooth functions of the single scalar r, so gpr_so3_tables evaluates that same quadrature once, at
// Chebyshev nodes of `nseg` equal segments of [0, rcut] (Bessel functions by Miller's recurrence, below), and
// stores piecewise Chebyshev coefficients; the descriptor kernel then evaluates 2 nmax (lmax+1) short Clenshaw
// sums per pair instead of 80 Bessel tables.  The interpolation error (8 segments x 16 coefficients at
// alpha = 2, rcut = 5) is 5e-15 of each function's scale -- below the rounding of the quadrature itself; it is
// measured against the direct quadrature in tests/test_gpu_so3.py.
//
// Angular part and assembly.  With A_nl = norm_l w (fc F1 + fc' F0), B_nl = norm_l w fc F0 the pair's
// coefficients are c_nlm = B_nl Y_lm and dc_nlm/dr_k = A_nl Y_lm u_k + B_nl (grad Y_lm)_k, so the derivative of
// the power spectrum row needs only the 15 + 45 real numbers
//     R_nl = sum_m Re[Y_lm conj(ctot_nlm)],   S_nlk = sum_m Re[(grad Y_lm)_k conj(ctot_nlm)]:
//     dP[(n1,n2,l), k] = u_k (A_n1l R_n2l + A_n2l R_n1l) + B_n1l S_n2lk + B_n2l S_n1lk
// and dC itself is never formed; S is evaluated without grad Y either, from Y_{l+-1} and per-centre tables that
// contract the gradient coefficients (SO3.py:682-707) with c_tot once.  One 256-thread workgroup per centre atom:
// the per-pair tables (geometry, A, B, Y) are built with lanes over pairs into LDS records (radial sums on three
// wavefronts, angular recurrences on the fourth), pass 1 sums c_tot from them, pass 2 forms the dP entries with one
// lane per (pair, l) and adds them over the pairs of a seq row in an LDS row buffer that is written out as the
// centre's slab of dxdr (the stress variant walks the seq rows instead, one row per wavefront).  No floating-point
// atomics; the summation order is fixed.
#include <atomic>
#include "gpr_internal.h"

namespace {

constexpr int SO3_LCAP = 7;      // lmax <= 6  (tables hold l = 0 .. lmax+1)
constexpr int SO3_NCAP = 6;      // nmax <= 6  (nmax = 5, 6 with lmax <= 3: the <4, 6> instance of the centre kernel)
constexpr int SO3_NQCAP = 128;   // quadrature nodes: 10*(nmax+lmax+1) <= 128
constexpr int SO3_WAVES = 4;
constexpr int SO3_NCHEB = 16;    // Chebyshev coefficients per segment (compile-time: the Clenshaw sum is fully unrolled)

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

// ---------------------------------------------------------------- table layout (doubles, device global)
//   cheb [nseg][ncheb][NF]   NF = 2 nmax (lmax+1): function f = 2 (n L + l) + {0: F0, 1: F1}
//   NY   [(lmax+3)^2]        Y_l^m normalisation, l <= lmax+1, m >= 0 (row width lmax+3)
//   CG   [L][M][6]           the six coefficients of grad Y_l^m (SO3.py:682-707)
//   NL   [L]                 norm_l = sqrt(2 sqrt2 pi / sqrt(2l+1)) (SO3.py:206)
//   RI   [(lmax+3)^2]        1 / (l - m) for l > m (Legendre recurrence without divisions)
//   GT   [(L^2+L) 3][22]     the linear map c_tot -> gradient tables T of one n (SO3.py:682-707 folded): per compact T entry
//                            [l, nterm, 4 x (source m, p, q, r, s)]:  T.re += p c.re + q c.im,  T.im += r c.re + s c.im
struct so3_tab { int NF, oNY, oCG, oNL, oRI, oGT, total; };
constexpr int SO3_GT_W = 22;
__host__ __device__ inline so3_tab so3_table_layout(int nmax, int lmax, int nseg, int ncheb) {
    const int L = lmax + 1, M = 2 * lmax + 1, YW = lmax + 3;
    so3_tab t;
    t.NF = 2 * nmax * L;
    int off = nseg * ncheb * t.NF;
    t.oNY = off; off += YW * YW;
    t.oCG = off; off += L * M * 6;
    t.oNL = off; off += L;
    t.oRI = off; off += YW * YW;
    off = (off + 1) & ~1;                                       // entries of GT are read 16 bytes at a time
    t.oGT = off; off += (L * L + L) * 3 * SO3_GT_W;
    t.total = off;
    return t;
}

// direct quadrature of F0_nl(r), F1_nl(r) (SO3.py:608-727 restricted to the radial factors): one wavefront per
// radius, lanes over quadrature nodes.  out[nr][NF].
__global__ void __launch_bounds__(64) so3_radial_direct_kernel(int nmax, int lmax, double alpha, int nq, const double *G,
                                                              const double *rq, int nr, const double *rs, double *out) {
    const int ir = blockIdx.x, lane = threadIdx.x, L = lmax + 1;
    if (ir >= nr) return;
    const double r = rs[ir];
    const int lt = lmax > 1 ? lmax : 1;
    double s0[SO3_LCAP * SO3_NCAP], s1[SO3_LCAP * SO3_NCAP];          // [n][l], nmax <= SO3_NCAP (checked on the host)
#pragma unroll
    for (int e = 0; e < SO3_LCAP * SO3_NCAP; e++) { s0[e] = 0.0; s1[e] = 0.0; }
    for (int q = lane; q < nq; q += 64) {
        double b[SO3_LCAP + 1];
        const double rqq = rq[q], xarg = 2.0 * alpha * r * rqq, chain = 2.0 * alpha * rqq, ixarg = 1.0 / xarg;
        sph_in_table(xarg, lt, b);
#pragma unroll
        for (int l = 0; l < SO3_LCAP; l++)
            if (l <= lmax) {
                const double db = ((l == 0) ? b[1] : b[l - 1] - (l + 1) * ixarg * b[l]) * chain;   // scipy _spherical_in_d
#pragma unroll
                for (int n = 0; n < SO3_NCAP; n++)
                    if (n < nmax) { const double g = G[n * nq + q]; s0[n * SO3_LCAP + l] += g * b[l]; s1[n * SO3_LCAP + l] += g * db; }
            }
    }
    const double E = 4.0 * M_PI * exp(-alpha * r * r);
#pragma unroll
    for (int n = 0; n < SO3_NCAP; n++)
#pragma unroll
        for (int l = 0; l < SO3_LCAP; l++)
            if (n < nmax && l <= lmax) {
                double a = s0[n * SO3_LCAP + l], d = s1[n * SO3_LCAP + l];
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) { a += __shfl_xor(a, o); d += __shfl_xor(d, o); }
                if (lane == 0) {
                    out[(size_t)ir * 2 * nmax * L + 2 * (n * L + l)] = E * a;
                    out[(size_t)ir * 2 * nmax * L + 2 * (n * L + l) + 1] = E * (d - 2.0 * alpha * r * a);
                }
            }
}

__global__ void so3_cheb_nodes_kernel(int nseg, int ncheb, double rcut, double *rs) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nseg * ncheb) return;
    const int seg = t / ncheb, j = t % ncheb;
    const double half = 0.5 * rcut / nseg, mid = (seg + 0.5) * rcut / nseg;
    rs[t] = mid + half * cos(M_PI * (j + 0.5) / ncheb);
}

// vals[seg][j][f] at the Chebyshev nodes -> coef[seg][k][f], f(t) = sum_k coef_k T_k(t) (coef_0 already halved)
__global__ void so3_cheb_coef_kernel(int nseg, int ncheb, int NF, const double *vals, double *coef) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nseg * ncheb * NF) return;
    const int f = t % NF, k = (t / NF) % ncheb, seg = t / (NF * ncheb);
    double s = 0.0;
    for (int j = 0; j < ncheb; j++) s += vals[((size_t)seg * ncheb + j) * NF + f] * cos(M_PI * k * (j + 0.5) / ncheb);
    coef[t] = s * (k == 0 ? 1.0 : 2.0) / ncheb;
}

__global__ void so3_const_tables_kernel(int lmax, double *NY, double *CG, double *NL, double *RI) {
    const int L = lmax + 1, M = 2 * lmax + 1, YW = lmax + 3, tid = threadIdx.x;
    for (int e = tid; e < YW * YW; e += blockDim.x) {
        const int l = e / YW, m = e % YW;
        double v = 0.0;
        if (m <= l && l <= lmax + 1) {
            double ratio = 1.0;                          // (l-m)! / (l+m)!
            for (int k = l - m + 1; k <= l + m; k++) ratio /= k;
            v = sqrt((2 * l + 1) / (4.0 * M_PI) * ratio);
        }
        NY[e] = v;
        RI[e] = l > m ? 1.0 / (l - m) : 0.0;
    }
    for (int e = tid; e < L * M; e += blockDim.x) {
        const int l = e / M, m = e % M - lmax;
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
        for (int k = 0; k < 6; k++) CG[e * 6 + k] = c[k];
    }
    for (int e = tid; e < L; e += blockDim.x) NL[e] = sqrt(2.0 * sqrt(2.0) * M_PI / sqrt(2.0 * e + 1.0));
}

// One entry of the per-centre gradient tables: the coefficient of Y_{l', mp} (l' = l -/+ 1 for side 0 / 1, mp >= 0, the
// m' < 0 half folded on) in  sum_m (grad Y_lm)_k conj(ctot_nlm)  -- SO3.py:682-707.  ctv(m): c_tot[n, l, m] (zero beyond
// |m| <= l), cgv(m, which): the six coupling coefficients of (l, m).
template <class CT, class CG>
__device__ __forceinline__ cplx so3_T_entry(int l, int side, int k, int mp, CT ctv, CG cgv) {
    auto Tval = [&](int m1) -> cplx {                          // coefficient of Y_{lp, m1}, any sign of m1
        const double is2 = 0.70710678118654752440;
        if (k == 2) return cscale(ctv(m1), cgv(m1, side ? 0 : 1));
        const cplx up = cscale(ctv(m1 + 1), cgv(m1 + 1, side ? 4 : 5) * is2);      // from gm of (l, m1+1)
        const cplx dn = cscale(ctv(m1 - 1), cgv(m1 - 1, side ? 2 : 3) * is2);      // from gp of (l, m1-1)
        if (k == 0) return {up.re - dn.re, up.im - dn.im};
        const cplx sm = {up.re + dn.re, up.im + dn.im};
        return {sm.im, -sm.re};                                // -i (up + dn)
    };
    cplx out = Tval(mp);
    if (mp > 0) {
        const cplx t2 = Tval(-mp);
        const double sg = (mp & 1) ? -1.0 : 1.0;
        out.re += sg * t2.re; out.im -= sg * t2.im;
    }
    return out;
}

// compact index inside one n-block of the T tables -> (l, k, side, mp); false for the unused l = 0 entries
__host__ __device__ inline bool so3_tf_decode(int idx, int L, int &l, int &k, int &side, int &mp) {
    l = 0;
    while (l + 1 < L && 3 * (l + 1) * (l + 2) <= idx) l++;
    const int rem = idx - 3 * l * (l + 1), w = 2 * l + 2;
    k = rem / w;
    const int r2 = rem - k * w;
    side = r2 >= l ? 1 : 0;
    mp = side ? r2 - l : r2;
    return l >= 1;
}

// The map c_tot -> T is real-linear and the same for every n and every centre: it is extracted ONCE per descriptor
// configuration by probing so3_T_entry with unit inputs (c = 1 and c = i at every source m), so that the descriptor kernel
// evaluates each table entry as at most four 2 x 2 real products instead of walking the case analysis per entry.
__global__ void so3_grad_map_kernel(int lmax, const double *CG, double *GT) {
    const int L = lmax + 1, M = 2 * lmax + 1, NB = (L * L + L) * 3;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= NB) return;
    double *g = GT + (size_t)idx * SO3_GT_W;
    int l, k, side, mp;
    const bool used = so3_tf_decode(idx, L, l, k, side, mp);
    int nterm = 0;
    for (int t = 0; t < 4; t++) { g[2 + 5 * t] = 0.0; g[3 + 5 * t] = 0.0; g[4 + 5 * t] = 0.0; g[5 + 5 * t] = 0.0; g[6 + 5 * t] = 0.0; }
    if (used) {
        auto cgv = [&](int m, int which) -> double { return (m >= -l && m <= l) ? CG[(l * M + (m + lmax)) * 6 + which] : 0.0; };
        for (int ms = -l; ms <= l; ms++) {
            const cplx a = so3_T_entry(l, side, k, mp, [&](int m) -> cplx { return m == ms ? cplx{1.0, 0.0} : cplx{0.0, 0.0}; }, cgv);
            const cplx b = so3_T_entry(l, side, k, mp, [&](int m) -> cplx { return m == ms ? cplx{0.0, 1.0} : cplx{0.0, 0.0}; }, cgv);
            if (a.re != 0.0 || a.im != 0.0 || b.re != 0.0 || b.im != 0.0) {
                if (nterm < 4) { double *t = g + 2 + 5 * nterm; t[0] = (double)ms; t[1] = a.re; t[2] = b.re; t[3] = a.im; t[4] = b.im; }
                nterm++;
            }
        }
    }
    g[0] = (double)l;
    g[1] = (double)nterm;                                      // > 4 would not fit: the descriptor kernel then takes the formula itself
}

// Clenshaw evaluation of function f of the piecewise Chebyshev table at radius r
__device__ __noinline__ double cheb_eval(const double *cheb, int nseg, int NF, double rcut, double r, int f) {
    int seg = (int)(r * nseg / rcut);
    seg = seg < 0 ? 0 : (seg > nseg - 1 ? nseg - 1 : seg);
    const double inv_half = 2.0 * nseg / rcut, mid = (seg + 0.5) * rcut / nseg;
    const double t = (r - mid) * inv_half, t2 = 2.0 * t;
    const double *c = cheb + (size_t)seg * SO3_NCHEB * NF + f;
    double cf[SO3_NCHEB];
#pragma unroll
    for (int k = 0; k < SO3_NCHEB; k++) cf[k] = c[k * NF];          // all loads in flight before the recurrence
    double b1 = 0.0, b2 = 0.0;
#pragma unroll
    for (int k = SO3_NCHEB - 1; k >= 1; k--) { const double b0 = cf[k] + t2 * b1 - b2; b2 = b1; b1 = b0; }
    return cf[0] + t * b1 - b2;
}

// F0 and F1 of one (n, l) -- functions f and f + 1, f even -- in one pass: their coefficients are neighbours in the
// table, one 16-byte load fetches both, and the two recurrences are independent chains
__device__ __forceinline__ void cheb_eval2(const double *cheb, int nseg, int NF, double rcut, double r, int f, double &v0, double &v1) {
    int seg = (int)(r * nseg / rcut);
    seg = seg < 0 ? 0 : (seg > nseg - 1 ? nseg - 1 : seg);
    const double inv_half = 2.0 * nseg / rcut, mid = (seg + 0.5) * rcut / nseg;
    const double t = (r - mid) * inv_half, t2 = 2.0 * t;
    const double2 *c = (const double2 *)(cheb + (size_t)seg * SO3_NCHEB * NF + f);
    // the coefficients come in two halves of eight (the upper one first: the recurrence runs downwards): sixteen 16-byte
    // loads in flight at once cost 64 registers, which is what pushed the kernel into scratch
    constexpr int H = SO3_NCHEB / 2;
    double2 hi[H], lo[H];
#pragma unroll
    for (int k = 0; k < H; k++) hi[k] = c[(H + k) * (NF / 2)];
    double a1 = 0.0, a2 = 0.0, b1 = 0.0, b2 = 0.0;
#pragma unroll
    for (int k = 0; k < H; k++) lo[k] = c[k * (NF / 2)];
#pragma unroll
    for (int k = H - 1; k >= 0; k--) {
        const double a0 = hi[k].x + t2 * a1 - a2; a2 = a1; a1 = a0;
        const double b0 = hi[k].y + t2 * b1 - b2; b2 = b1; b1 = b0;
    }
#pragma unroll
    for (int k = H - 1; k >= 1; k--) {
        const double a0 = lo[k].x + t2 * a1 - a2; a2 = a1; a1 = a0;
        const double b0 = lo[k].y + t2 * b1 - b2; b2 = b1; b1 = b0;
    }
    v0 = lo[0].x + t * a1 - a2;
    v1 = lo[0].y + t * b1 - b2;
}

__global__ void so3_radial_eval_kernel(const double *cheb, int nseg, int ncheb, int NF, double rcut, int nr, const double *rs,
                                       double *out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nr * NF) return;
    out[t] = cheb_eval(cheb, nseg, NF, rcut, rs[t / NF], t % NF);
}

// ---------------------------------------------------------------- descriptor kernel
// Ordering of wavefront-private LDS traffic between lanes of one wavefront: the LDS queue is in order, so draining
// it (and stopping the compiler from moving memory operations across) is enough.  A workgroup fence would also
// wait for the outstanding global stores of the previous row (vmcnt), i.e. put HBM write latency into the loop.
__device__ __forceinline__ void so3_wave_lds_sync() {
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

struct so3_args {
    int nmax, lmax, d, nseg, ncheb, dbg;
    double rcut, alpha;
    const double *table;        // so3_table_layout
    const double *vec;          // [P][3] neighbour displacement
    const double *wj;           // [P] neighbour weight (Z_j, SO3.py:385)
    const int *centre_pair0;    // [nc+1]
    const int *centre_row0;     // [nc+1] seq-row range of each centre
    const int *row_pair0;       // [S+1]  pair range of each seq row
    const int *self_row;        // [nc]   seq row (i, i)
    double *x;                  // [nc][d]
    double *dxdr;               // [S][d][3]
    // optional stress rows (SO3.py:240-306): rdxdr[S][d][3][3] = -pstress / volume
    const double *positions;    // [nc][3] centre positions as the caller gave them (not wrapped)
    const double *inv_vol;      // [nc] 1 / cell volume of the centre's structure
    double *rdxdr;              // NULL: not wanted
    // large batches (so3_fixed_kernel only): geometry and radial channels of every pair, [P][8 + 2 NLP] = u, 1/r, r | A_nl | B_nl, formed by
    // so3_radial_pairs_kernel in a streaming pass of its own; NULL: the centre kernel forms them itself
    const double *ab;
};

constexpr int SO3_PC = 44;       // pairs whose tables are resident in LDS at a time (with the tables below: < 40 KB,
                                 // four workgroups per CU)
constexpr int SO3_NLP = 16;      // padded number of radial-angular channels nmax (lmax+1) when there are at most 16
constexpr int SO3_NLP_BIG = 32;  // ... when there are more (nmax <= 4, lmax <= 6: up to 28)
constexpr int SO3_SELFW = 256;   // stride of the (i, i)-row partial sums: 3 d <= 256 (one thread per column of a dxdr row)
constexpr int SO3_LDS_DOUBLES = 96 * 1024 / 8;   // dynamic LDS the centre kernels may ask for
__host__ __device__ inline int so3_nlp(int nmax, int lmax) { return nmax * (lmax + 1) <= SO3_NLP ? SO3_NLP : SO3_NLP_BIG; }

// dynamic LDS layout (doubles).  Per-centre tables: NY | RI | NL constants, c_tot [NLM] cplx, Rf [NL][L] cplx
// (c_tot folded onto m >= 0), Tf [NL][3][2][LT+1] cplx (gradient coefficients contracted with c_tot, folded onto
// m' >= 0; compact: 2 l + 2 entries per (n, l, k)), per wavefront R [16] | S [48], with stress output the (i, i)
// image sums [4][96]; then SO3_PC pair records of `rec` doubles (the self sums [4][96] reuse that area at the end):
//   ux uy uz 1/r r - - - | A [16] | B [16] | Y_l^m, 0 <= m <= l <= lmax+1, triangular, cplx
// (sII: with stress output, dP and R (x) dP summed over the periodic images of the centre itself, [4][96])
struct so3_lds { int oNY, oRI, oNL, oCG, oPart, oCtot, oRf, oTf, oSelf, oII, oRS, oRec, rec, tri, pc, total; };
// compact Tf index: per n the l-blocks hold 3 (2 l + 2) entries, l (l + 1) 3 entries precede block l
__host__ __device__ inline int so3_tf_base(int n, int l, int L) { return (n * (L * L + L) + l * (l + 1)) * 3; }
// Without stress rows pass 2 runs one lane per (pair, l): at most 256 / (lmax + 1) pairs per chunk, no per-wavefront
// R/S exchange area, and the record area is reused as the staging area of the chunk's dP rows ([pc + 1][3 d]: the
// rows of pc pairs plus the empty (i, i) row between them) with the pair -> row map behind the records.
__host__ __device__ inline so3_lds so3_layout(int nmax, int lmax, bool stress) {
    const int L = lmax + 1, M = 2 * lmax + 1, YW = lmax + 3, LT = lmax + 1, NLM = nmax * L * M, NL = nmax * L;
    const int ne = nmax * (nmax + 1) / 2 * L * 3;
    so3_lds o;
    int off = 0;
    o.oNY = off; off += YW * YW;
    o.oRI = off; off += YW * YW;
    o.oNL = off; off += L;
    o.oCG = off; off += L * M * 6;            // gradient coupling coefficients (SO3.py:682-707)
    off = (off + 1) & ~1;
    o.oPart = off; off += 256 * 2;            // c_tot partial sums of the pair groups of pass 1
    o.oCtot = off; off += NLM * 2;
    o.oRf = off; off += NL * L * 2;
    o.oTf = off; off += nmax * (L * L + L) * 3 * 2;
    o.oII = off; if (stress) off += 96 * 4;   // stress only: dP sum and R (x) dP sum of the (i, i) image pairs
    o.oRS = off; if (stress) off += SO3_WAVES * 64;
    o.tri = (LT + 1) * (LT + 2) / 2;
    o.rec = 8 + 2 * so3_nlp(nmax, lmax) + 2 * o.tri;
    o.pc = stress ? SO3_PC : (SO3_PC < 256 / L ? SO3_PC : 256 / L);
    // long descriptors (round 4: up to 85 components, e.g. SO3(4, 6), SO3(5, 4), SO3(6, 3)): fewer pairs per chunk so that the
    // staging area of the chunk's rows still fits
    while (!stress && o.pc > 8 && off + (o.pc + 1) * ne > SO3_LDS_DOUBLES) o.pc--;
    o.oRec = off;
    int area_rec = o.pc * o.rec + o.pc, area_rows = (o.pc + 1) * ne;    // records + pair->row map | row staging
    if (!stress && area_rec < 2 * SO3_SELFW) area_rec = 2 * SO3_SELFW;   // the (i, i)-row partial sums end up here
    off += stress ? o.pc * o.rec : (area_rec > area_rows ? area_rec : area_rows);
    o.oSelf = o.oRec;                         // the records are dead when the self sums are formed
    o.total = off;
    (void)NL;
    return o;
}

// LC, NM: compile-time bounds on lmax + 1 and nmax (5, 3 serve the usual nmax <= 3, lmax <= 4; the R/S sums below are
// unrolled to them)
template <int LC, int NM, bool STRESS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 4))) so3_centre_kernel(so3_args P) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    // record stride of the channel tables: a constant 16 in the instance of the usual configurations (nmax (lmax+1) <= 15)
    const int NLP = (NM <= 3 && LC <= 5) ? SO3_NLP : so3_nlp(P.nmax, P.lmax);
    const int L = P.lmax + 1, M = 2 * P.lmax + 1, NLM = P.nmax * L * M, NL = P.nmax * L;
    const int YW = P.lmax + 3, LT = P.lmax + 1;
    const so3_lds lay = so3_layout(P.nmax, P.lmax, STRESS);
    const int PC = lay.pc;                        // pairs per chunk
    const so3_tab tab = so3_table_layout(P.nmax, P.lmax, P.nseg, P.ncheb);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const double *sNY = smem + lay.oNY, *sRI = smem + lay.oRI, *sNL = smem + lay.oNL;
    cplx *ctot = (cplx *)(smem + lay.oCtot), *sRf = (cplx *)(smem + lay.oRf), *sTf = (cplx *)(smem + lay.oTf);
    double *sSelf = smem + lay.oSelf;
    double *sR = smem + lay.oRS + wave * 64, *sS = sR + 16;      // stress variant only
    double *recs = smem + lay.oRec;
    const int REC = lay.rec;
    const int ic = blockIdx.x;
    const int p0 = P.centre_pair0[ic], p1 = P.centre_pair0[ic + 1];
    const double *sCG = smem + lay.oCG;
    for (int e = tid; e < YW * YW; e += 256) { smem[lay.oNY + e] = P.table[tab.oNY + e]; smem[lay.oRI + e] = P.table[tab.oRI + e]; }
    for (int e = tid; e < L; e += 256) smem[lay.oNL + e] = P.table[tab.oNL + e];
    for (int e = tid; e < L * M * 6; e += 256) smem[lay.oCG + e] = P.table[tab.oCG + e];
    __syncthreads();

    // Tables of the pairs [c0, c0 + npc): lanes over pairs.  Wavefronts 0-2, four threads per pair: distance, cutoff
    // and the radial Chebyshev sums (A_nl, B_nl); wavefront 3, one thread per pair: unit vector and the Legendre /
    // azimuth recurrences (Y_l^m, l <= lmax+1) -- two code paths of about equal length on different wavefronts, neither
    // waiting for the other.  Ends with a block barrier.
    auto build_records = [&](int c0, int npc) {
        if (wave < 3) {
            const int pr = tid >> 2, part = tid & 3;
            if (pr < npc) {
                const int p = c0 + pr;
                double *rec = recs + pr * REC;
                const double vx = P.vec[3 * (size_t)p], vy = P.vec[3 * (size_t)p + 1], vz = P.vec[3 * (size_t)p + 2];
                const double r = sqrt(vx * vx + vy * vy + vz * vz), w = P.wj[p];
                double sarg, carg;
                sincospi(r / P.rcut, &sarg, &carg);         // cutoff 0.5 (cos(pi r / rcut) + 1) and its derivative
                const double fc = 0.5 * (carg + 1.0), dfc = -0.5 * M_PI / P.rcut * sarg;
                for (int nl = part; nl < NL; nl += 4) {
                    double F0, F1;
                    cheb_eval2(P.table, P.nseg, tab.NF, P.rcut, r, 2 * nl, F0, F1);
                    const double norm = sNL[nl % L] * w;                 // SO3.py:206, 221
                    rec[8 + nl] = norm * (fc * F1 + dfc * F0);
                    rec[8 + NLP + nl] = norm * fc * F0;
                }
            }
        } else if (lane < npc) {
            const int p = c0 + lane;
            double *rec = recs + lane * REC;
            const double vx = P.vec[3 * (size_t)p], vy = P.vec[3 * (size_t)p + 1], vz = P.vec[3 * (size_t)p + 2];
            const double r = sqrt(vx * vx + vy * vy + vz * vz), inv_r = 1.0 / r;
            const double ux = vx * inv_r, uy = vy * inv_r, ct = vz * inv_r;
            rec[0] = ux; rec[1] = uy; rec[2] = ct; rec[3] = inv_r; rec[4] = r;
            const double rho = sqrt(ux * ux + uy * uy);
            const double cphi = rho > 0.0 ? ux / rho : 1.0, sphi = rho > 0.0 ? uy / rho : 0.0;
            cplx *Y = (cplx *)(rec + 8 + 2 * NLP);
            double cm = 1.0, sm = 0.0, pmm = 1.0;            // cos(m phi), sin(m phi), P_m^m (Condon-Shortley)
            for (int m = 0; m <= LT; m++) {
                if (m > 0) {
                    const double c2 = cm * cphi - sm * sphi;
                    sm = sm * cphi + cm * sphi; cm = c2;
                    pmm *= -(2 * m - 1) * rho;
                }
                double plm2 = 0.0, plm1 = pmm;
                for (int l = m; l <= LT; l++) {
                    double plm;
                    if (l == m) plm = pmm;
                    else if (l == m + 1) plm = (2 * m + 1) * ct * pmm;
                    else plm = ((2 * l - 1) * ct * plm1 - (l + m - 1) * plm2) * sRI[l * YW + m];
                    if (l > m) { plm2 = plm1; plm1 = plm; }
                    const double nrm = sNY[l * YW + m] * plm;
                    Y[l * (l + 1) / 2 + m] = {nrm * cm, nrm * sm};
                }
            }
        }
        __syncthreads();
    };

    // ---- pass 1: c_tot[nl][m >= 0] = sum over pairs of B_nl Y_l^m.  Task = one (n, l, m <= l); the block splits into
    //      G groups of 256 / G threads, group g sums the pairs q = g, g + G, ... of a chunk, and the G partial sums are
    //      added in group order.
    const int nchunks = (p1 - p0 + PC - 1) / PC;
    {
        const int tri_l = L * (L + 1) / 2, ntask = P.nmax * tri_l;
        const int G = ntask <= 64 ? 4 : (ntask <= 128 ? 2 : 1), slots = 256 / G;
        const int task = tid % slots, grp = tid / slots;
        int n = task / tri_l, l = 0;
        const int rem = task - n * tri_l;
        while ((l + 1) * (l + 2) / 2 <= rem) l++;
        const int m = rem - l * (l + 1) / 2, nl = n * L + l;
        const bool mine = task < ntask;
        cplx acc = {0.0, 0.0};
        for (int c0 = p0; c0 < p1; c0 += PC) {
            const int npc = min(PC, p1 - c0);
            build_records(c0, npc);
            GPR_PHASE_EXIT(P.dbg, 1)
            if (mine)
                for (int q = grp; q < npc; q += G) {
                    const double *rec = recs + q * REC;
                    const cplx y = ((const cplx *)(rec + 8 + 2 * NLP))[l * (l + 1) / 2 + m];
                    const double b = rec[8 + NLP + nl];
                    acc.re += b * y.re; acc.im += b * y.im;
                }
            if (nchunks > 1) __syncthreads();            // records are rebuilt by the next chunk
        }
        cplx *sPart = (cplx *)(smem + lay.oPart);
        if (mine) sPart[grp * slots + task] = acc;
        for (int e = tid; e < NLM; e += 256) ctot[e] = {0.0, 0.0};
        __syncthreads();
        if (mine && grp == 0) {
            for (int g = 1; g < G; g++) { const cplx o = sPart[g * slots + task]; acc.re += o.re; acc.im += o.im; }
            ctot[nl * M + P.lmax + m] = acc;
            if (m > 0) { const double sg = (m & 1) ? -1.0 : 1.0; ctot[nl * M + P.lmax - m] = {sg * acc.re, -sg * acc.im}; }
            sRf[nl * L + m] = (m == 0) ? acc : cplx{2.0 * acc.re, 2.0 * acc.im};
        }
    }
    __syncthreads();

    // ---- power spectrum x_i
    auto decode = [&](int c, int &n1, int &n2, int &l) {
        const int pi = c / L;
        l = c % L;
        n1 = 0;
        while ((n1 + 1) * (n1 + 2) / 2 <= pi) n1++;
        n2 = pi - n1 * (n1 + 1) / 2;
    };
    for (int c = tid; c < P.d; c += 256) {
        int n1, n2, l;
        decode(c, n1, n2, l);
        double s = 0.0;
        for (int mi = 0; mi < M; mi++) {
            const cplx a = ctot[(n1 * L + l) * M + mi], b = ctot[(n2 * L + l) * M + mi];
            s += a.re * b.re + a.im * b.im;
        }
        P.x[(size_t)ic * P.d + c] = s;
    }

    // ---- gradient coefficients contracted with c_tot (SO3.py:682-707 folded into per-centre tables):
    //      S_nlk(pair) = (1/r) sum_{side} sum_{m' >= 0} Re[Y_{l', m'} conj(Tf[nl][k][side][m'])], l' = l -/+ 1
    {
#ifdef GPR_PHASE_ABLATION
        const int NB = (L * L + L) * 3, ntf = (P.dbg == 5) ? 0 : P.nmax * NB;      // dbg 5: timing without the tables (wrong numbers)
#else
        const int NB = (L * L + L) * 3, ntf = P.nmax * NB;
#endif
        const double *GT = P.table + tab.oGT;
        for (int e = tid; e < ntf; e += 256) {
            const int n = e / NB, idx = e - n * NB;
            const double *g = GT + (size_t)idx * SO3_GT_W;
            const int l = (int)g[0], nterm = (int)g[1];
            if (l < 1) continue;                                  // the l = 0 slots of the compact layout are never read
            const int nl = n * L + l;
            cplx out = {0.0, 0.0};
            if (nterm <= 4) {
#pragma unroll
                for (int t = 0; t < 4; t++)
                    if (t < nterm) {
                        const double *q = g + 2 + 5 * t;
                        const cplx c = ctot[nl * M + P.lmax + (int)q[0]];
                        out.re += q[1] * c.re + q[2] * c.im;
                        out.im += q[3] * c.re + q[4] * c.im;
                    }
            } else {
                int l2, k, side, mp;
                so3_tf_decode(idx, L, l2, k, side, mp);
                out = so3_T_entry(l, side, k, mp,
                                  [&](int m) -> cplx { return (m >= -l && m <= l) ? ctot[nl * M + P.lmax + m] : cplx{0.0, 0.0}; },
                                  [&](int m, int which) -> double { return (m >= -l && m <= l) ? sCG[(l * M + (m + P.lmax)) * 6 + which] : 0.0; });
            }
            sTf[n * NB + idx] = out;
        }
    }
    __syncthreads();

    GPR_PHASE_EXIT(P.dbg, 2)
    // ---- pass 2 (stress variant below): dP rows, one row per wavefront at a time.  Lane t < NL owns R_nl, lanes NL .. 4 NL - 1 own S_nlk;
    //      lanes e < 3 d then combine into the row's outputs.
    const int r0 = P.centre_row0[ic], r1 = P.centre_row0[ic + 1];
    if (r0 == r1) return;                         // no rows at all: the emptied table of an overflowed capped build (nbr.hip)
    const int rself = P.self_row[ic];
    const int ne = P.d * 3;
    if constexpr (!STRESS) {
        // ---- pass 2 without stress rows: one lane per (pair q of the chunk, l).  The lane forms R_nl and S_nlk of its
        // pair for every n from the pair's Y values and the per-centre tables (coefficient reads are broadcasts: all
        // lanes with the same l read the same address), combines them into the nmax (nmax + 1) / 2 x 3 dP entries of
        // that l, and the entries are summed over the pairs of a seq row (periodic images of one neighbour) in pair
        // order in an LDS row buffer that takes the place of the dead pair records.  The buffer is the chunk's slab
        // of dxdr: it is written out flat (coalesced), and its column sums give the (i, i) row (SO3.py:266-270).
        constexpr int NPI = NM * (NM + 1) / 2;
        int *sMap = (int *)(recs + PC * REC);                // [PC] row of the pair (centre-local), [PC] index in its row, max
        double *rowOut = recs;
        const int nsp = 256 / ne >= 2 ? 2 : 1;                // (i, i) row: column sums, in two interleaved halves when
        const int spart = tid / ne, scol = tid - spart * ne;  // the block has two threads per column
        double selfacc = 0.0;
        for (int c0 = p0; c0 < p1; c0 += PC) {
            const int npc = min(PC, p1 - c0), c1 = c0 + npc;
            if (nchunks > 1) build_records(c0, npc);          // a single chunk is still resident from pass 1
            if (tid == 0) sMap[2 * PC] = 0;
            __syncthreads();
            for (int row = r0 + tid; row < r1; row += 256) {
                const int a = max(P.row_pair0[row], c0), b = min(P.row_pair0[row + 1], c1);
                for (int p = a; p < b; p++) { sMap[p - c0] = row - r0; sMap[PC + p - c0] = p - a; }
                if (b > a) atomicMax(&sMap[2 * PC], b - a);
            }
            // lanes ordered by l, then pair: a wavefront sees one or two values of l, so its coefficient reads are
            // (nearly) uniform and it skips the terms none of its lanes has (lw = the largest l in this wavefront)
            const int l = tid / npc, q = tid - l * npc;
            const bool active = l < L;
            const int lw = __builtin_amdgcn_readfirstlane(min(L - 1, (64 * wave + 63) / npc));
            double outv[NPI * 3];
            if (active) {
                const double *rec = recs + q * REC;
                const cplx *Y = (const cplx *)(rec + 8 + 2 * NLP);
                double Rn[NM], Sn[NM][3];
#pragma unroll
                for (int n = 0; n < NM; n++) { Rn[n] = 0.0; Sn[n][0] = 0.0; Sn[n][1] = 0.0; Sn[n][2] = 0.0; }
                {
                    const cplx *yl = Y + l * (l + 1) / 2, *rf = sRf + l * L;
#pragma unroll
                    for (int m = 0; m < LC; m++)
                        if (m <= lw && m <= l) {
                            const cplx y = yl[m];
#pragma unroll
                            for (int n = 0; n < NM; n++)
                                if (n < P.nmax) { const cplx c = rf[n * L * L + m]; Rn[n] += y.re * c.re + y.im * c.im; }
                        }
                }
                if (l >= 1) {
                    const cplx *y0 = Y + (l - 1) * l / 2, *y1 = Y + (l + 1) * (l + 2) / 2;
                    const cplx *tf = sTf + so3_tf_base(0, l, L);
                    const int nstep = (L * L + L) * 3, kstep = 2 * l + 2;
#pragma unroll
                    for (int m = 0; m < LC - 1; m++)
                        if (m < lw && m < l) {
                            const cplx y = y0[m];
#pragma unroll
                            for (int n = 0; n < NM; n++)
                                if (n < P.nmax)
#pragma unroll
                                    for (int k = 0; k < 3; k++) {
                                        const cplx c = tf[n * nstep + k * kstep + m];
                                        Sn[n][k] += y.re * c.re + y.im * c.im;
                                    }
                        }
#pragma unroll
                    for (int m = 0; m < LC + 1; m++)
                        if (m < lw + 2 && m < l + 2) {
                            const cplx y = y1[m];
#pragma unroll
                            for (int n = 0; n < NM; n++)
                                if (n < P.nmax)
#pragma unroll
                                    for (int k = 0; k < 3; k++) {
                                        const cplx c = tf[n * nstep + k * kstep + l + m];
                                        Sn[n][k] += y.re * c.re + y.im * c.im;
                                    }
                        }
                }
                const double inv_r = rec[3], u[3] = {rec[0], rec[1], rec[2]};
                double An[NM], Bn[NM];
#pragma unroll
                for (int n = 0; n < NM; n++) {
                    An[n] = n < P.nmax ? rec[8 + n * L + l] : 0.0;
                    Bn[n] = n < P.nmax ? rec[8 + NLP + n * L + l] * inv_r : 0.0;
                }
#pragma unroll
                for (int n1 = 0; n1 < NM; n1++)
#pragma unroll
                    for (int n2 = 0; n2 <= n1; n2++) {
                        const double ar = An[n1] * Rn[n2] + An[n2] * Rn[n1];
#pragma unroll
                        for (int k = 0; k < 3; k++)
                            outv[(n1 * (n1 + 1) / 2 + n2) * 3 + k] = u[k] * ar + Bn[n1] * Sn[n2][k] + Bn[n2] * Sn[n1][k];
                    }
            }
            __syncthreads();                                  // records read, pair -> row map complete
#ifdef GPR_PHASE_ABLATION
            if (P.dbg == 3) { if (active && outv[0] == 1.234567e300) P.x[0] = 0.0; return; }   // ablation: keep outv alive
#endif
            const int myrow = active ? sMap[q] : 0, myj = active ? sMap[PC + q] : -1;
            const int rowb = sMap[0], nrows_c = sMap[npc - 1] - rowb + 1, maxj = sMap[2 * PC];
            const bool cont = P.row_pair0[r0 + rowb] < c0;    // the chunk's first row began in the previous chunk
            __syncthreads();                                  // map read: the row buffer may take its place
            const bool mine = active && r0 + myrow != rself;  // periodic images of the centre itself cancel (SO3.py:266-273)
            double *dst = rowOut + (myrow - rowb) * ne + l * 3;
            for (int j = 0; j < maxj; j++) {
                if (mine && myj == j) {
#pragma unroll
                    for (int pi = 0; pi < NPI; pi++)
                        if (pi < P.d / L)
#pragma unroll
                            for (int k = 0; k < 3; k++) {
                                double *o = dst + pi * L * 3 + k;
                                *o = j == 0 ? outv[pi * 3 + k] : *o + outv[pi * 3 + k];
                            }
                }
                __syncthreads();
            }
            GPR_PHASE_EXIT(P.dbg, 4)
            for (int f = tid; f < nrows_c * ne; f += 256) {
                const int rl = f / ne, row = r0 + rowb + rl;
                if (row == rself) continue;
                double *o = P.dxdr + (size_t)(r0 + rowb) * ne + f;
                *o = (cont && rl == 0) ? *o + rowOut[f] : rowOut[f];
            }
            if (spart < nsp)
                for (int rl = spart; rl < nrows_c; rl += nsp)
                    if (r0 + rowb + rl != rself) selfacc += rowOut[rl * ne + scol];
            __syncthreads();                                  // row buffer consumed (the next chunk rebuilds the records)
        }
        if (spart < nsp) sSelf[spart * SO3_SELFW + scol] = selfacc;
        __syncthreads();
        for (int e = tid; e < ne; e += 256) P.dxdr[(size_t)rself * ne + e] = -(sSelf[e] + (nsp == 2 ? sSelf[SO3_SELFW + e] : 0.0));
    } else {
        int t_nl = 0, t_k = -1;                       // task of this lane: k = -1 -> R, 0..2 -> S
        if (lane < NL) t_nl = lane;
        else if (lane < 4 * NL) { t_nl = (lane - NL) / 3; t_k = (lane - NL) % 3; }
        const bool has_task = lane < 4 * NL;
        const int t_l = t_nl % L;
        // R lane: Y_l . Rf (l + 1 terms).  S lane: Y_{l-1} . Tf (l terms, m' <= l - 1) then Y_{l+1} . Tf (l + 2 terms); the
        // compact Tf block of (n, l, k) holds them back to back
        const cplx *t_c1 = sRf + t_nl * L, *t_c2 = sRf;
        int t_y1 = t_l * (t_l + 1) / 2, t_y2 = 0, t_n1 = t_l + 1, t_n2 = 0, t_dst = t_nl;
        if (t_k >= 0) {
            t_c1 = sTf + so3_tf_base(t_nl / L, t_l, L) + t_k * (2 * t_l + 2);
            t_c2 = t_c1 + t_l;
            t_y1 = t_l >= 2 ? (t_l - 1) * t_l / 2 : 0; t_y2 = (t_l + 1) * (t_l + 2) / 2;
            t_n1 = t_l; t_n2 = t_l >= 1 ? t_l + 2 : 0;
            t_dst = 16 + t_nl * 3 + t_k;
        }
        int e_k[2], iA1[2], iA2[2];
    #pragma unroll
        for (int rnd = 0; rnd < 2; rnd++) {
            const int e = lane + 64 * rnd;
            int n1 = 0, n2 = 0, l = 0;
            if (e < ne) decode(e / 3, n1, n2, l);
            e_k[rnd] = e % 3; iA1[rnd] = n1 * L + l; iA2[rnd] = n2 * L + l;
        }
        double self0 = 0.0, self1 = 0.0;
        constexpr bool stress = true;
        double *sII = smem + lay.oII;
        double Ri[3] = {0.0, 0.0, 0.0};
        if (stress) {
            Ri[0] = P.positions[3 * (size_t)ic]; Ri[1] = P.positions[3 * (size_t)ic + 1]; Ri[2] = P.positions[3 * (size_t)ic + 2];
            for (int e = tid; e < 96 * 4; e += 256) sII[e] = 0.0;
            __syncthreads();
        }
        const double ivol = stress ? P.inv_vol[ic] : 0.0;
        for (int c0 = p0; c0 < p1; c0 += PC) {
            const int npc = min(PC, p1 - c0), c1 = c0 + npc;
            if (nchunks > 1) build_records(c0, npc);        // a single chunk is still resident from pass 1
            for (int row = r0 + wave; row < r1; row += SO3_WAVES) {
                const bool is_self = row == rself;
                if (is_self && !stress) continue;
                const int q0 = P.row_pair0[row], q1 = P.row_pair0[row + 1];
                const int a = max(q0, c0), b = min(q1, c1);
                if (a >= b) continue;
                double s[2] = {0.0, 0.0};
                double rs[2][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};       // sum over pairs of R_j[a] * dP[c, k]
                for (int p = a; p < b; p++) {
                    const double *rec = recs + (p - c0) * REC;
                    const cplx *Y = (const cplx *)(rec + 8 + 2 * NLP);
                    if (has_task) {
                        // one code path for R and S lanes: two fixed-length, predicated segments Re[Y conj(c)]; every LDS
                        // load of the task is issued before the first add
                        const cplx *y1 = Y + t_y1, *y2 = Y + t_y2;
                        double acc = 0.0;
    #pragma unroll
                        for (int m = 0; m < LC; m++)
                            if (m < t_n1) { const cplx y = y1[m], c = t_c1[m]; acc += y.re * c.re + y.im * c.im; }
    #pragma unroll
                        for (int m = 0; m < LC + 1; m++)
                            if (m < t_n2) { const cplx y = y2[m], c = t_c2[m]; acc += y.re * c.re + y.im * c.im; }
                        sR[t_dst] = t_k < 0 ? acc : acc * rec[3];
                    }
                    so3_wave_lds_sync();
    #pragma unroll
                    for (int rnd = 0; rnd < 2; rnd++) {
                        if (lane + 64 * rnd < ne) {
                            const int a1 = iA1[rnd], a2 = iA2[rnd], k = e_k[rnd];
                            const double *A = rec + 8, *B = rec + 8 + NLP;
                            const double term = rec[k] * (A[a1] * sR[a2] + A[a2] * sR[a1]) + B[a1] * sS[a2 * 3 + k] + B[a2] * sS[a1 * 3 + k];
                            s[rnd] += term;
                            if (stress) {                       // R_j = R_i + r u (SO3.py:225-226, 254)
                                const double rr = rec[4];
                                rs[rnd][0] += (Ri[0] + rr * rec[0]) * term;
                                rs[rnd][1] += (Ri[1] + rr * rec[1]) * term;
                                rs[rnd][2] += (Ri[2] + rr * rec[2]) * term;
                            }
                        }
                    }
                    so3_wave_lds_sync();
                }
    #pragma unroll
                for (int rnd = 0; rnd < 2; rnd++)
                    if (lane + 64 * rnd < ne) {
                        const int e = lane + 64 * rnd;
                        if (is_self) {      // periodic images of the centre itself: only the stress rows use them (SO3.py:264, 272-273)
                            sII[e] += s[rnd];
                            sII[96 + e] += rs[rnd][0]; sII[192 + e] += rs[rnd][1]; sII[288 + e] += rs[rnd][2];
                            continue;
                        }
                        double *o = P.dxdr + (size_t)row * ne + e;
                        // a row whose pairs straddle a chunk boundary: this lane wrote the earlier part itself
                        *o = (q0 < c0) ? *o + s[rnd] : s[rnd];
                        if (stress) {
                            double *so = P.rdxdr + (size_t)row * ne * 3 + (e / 3) * 9 + (e % 3);
    #pragma unroll
                            for (int a3 = 0; a3 < 3; a3++) {
                                const double v = rs[rnd][a3] * ivol;        // rdxdr = -pstress / vol, pstress = -sum R_j (x) dP
                                so[3 * a3] = (q0 < c0) ? so[3 * a3] + v : v;
                            }
                        }
                    }
                if (!is_self) { self0 += s[0]; self1 += s[1]; }
            }
            if (nchunks > 1) __syncthreads();
        }
        __syncthreads();                                   // every wavefront is done with the records: their area takes the self sums
        sSelf[wave * 96 + lane] = self0;
        if (lane < 32) sSelf[wave * 96 + 64 + lane] = self1;
        __syncthreads();
        for (int e = tid; e < ne; e += 256) {
            double s = 0.0;
            for (int w2 = 0; w2 < SO3_WAVES; w2++) s += sSelf[w2 * 96 + e];
            P.dxdr[(size_t)rself * ne + e] = -s;             // SO3.py:266-270
            if (stress) {
                // pstress(i, i) = -sum_{images of i} R_j (x) dP + R_i (x) sum_{all pairs} dP  (SO3.py:264, 272-273)
                const double tot = s + sII[e];
                double *so = P.rdxdr + (size_t)rself * ne * 3 + (e / 3) * 9 + (e % 3);
    #pragma unroll
                for (int a3 = 0; a3 < 3; a3++) so[3 * a3] = (sII[96 * (a3 + 1) + e] - Ri[a3] * tot) * ivol;
            }
        }
    }
}

// ---------------------------------------------------------------- the same kernel with (nmax, lmax) known at compile time
// The generic kernel above carries nmax / lmax as run-time values, so every index -- record strides, table offsets, the LDS
// layout, triangular decodes -- is integer arithmetic at run time (more than half of its vector instructions are not
// floating point, and its scalar unit spills to vector lanes).  The reference uses ONE configuration everywhere
// (SO3(nmax = 3, lmax = 4): gaussianprocess.py:1028, examples/database/pd4-RBF.json:24-31): this instance has every size as a
// constant and its phases re-mapped so that nearly all 256 lanes work in each of them:
//   radial part   lane = (pair, l): the three channels (n, l) of one l, as dot products of the Chebyshev coefficients with
//                 T_k(t) formed once per lane (one multiply-add per coefficient instead of the two of the Clenshaw recurrence);
//   angular part  lane = (pair, m): Y_l^m for l = m .. lmax + 1 as N_lm Q_l^m(cos theta) (u_x + i u_y)^m -- the azimuth and the
//                 sin^m theta factor of the associated Legendre function are one complex power, no square root, no division
//                 and no special case on the axis;
//   c_tot         45 tasks (n, l, m >= 0) x 5 pair groups;
//   gradient map  lane = table entry, the three n in a row (the map is the same for every n: one read of its coefficients);
//   pass 2        lane = (pair, l) with the heavy l first (a wavefront pays for the largest l it holds).
// Same arithmetic as the generic kernel up to the order of a few sums; tests/test_gpu_so3.py holds both to the goldens of the
// reference's own SO3.calculate and to each other.
template <int NMAX, int LMAX, bool PRE = false>
struct so3_fx {
    static constexpr int L = LMAX + 1, M = 2 * LMAX + 1, NL = NMAX * L, YW = LMAX + 3, LT = LMAX + 1;
    static constexpr int NPI = NMAX * (NMAX + 1) / 2, D = NPI * L, NE = 3 * D;
    static constexpr int TRI = (LT + 1) * (LT + 2) / 2;       // Y_l^m, 0 <= m <= l <= LT
    static constexpr int TRIL = L * (L + 1) / 2;              // (l, m >= 0) with l <= LMAX
    static constexpr int NLP = (NL + 1) & ~1;                 // channel tables padded to an even length (16-byte records)
    // record of a pair in LDS: [ux uy uz 1/r r - - -] | A [NLP] | B [NLP] | Y_l^m; with the radial pre-pass (PRE) A stays in
    // global memory (pass 2 reads its three values per lane from there): [geometry] | B | Y
    static constexpr int OA = 8, OB = PRE ? 8 : 8 + NLP, OY = OB + NLP;
    static constexpr int REC = OY + 2 * TRI;
    static constexpr int HEAD = 8 + 2 * NLP;                  // record head the pre-pass writes per pair: geometry | B | A
    // rows of dxdr staged at a time: all rows of a chunk, or -- PRE, to fit five workgroups on a CU -- slabs of 31
    static constexpr int RCAP = PRE ? 31 : 256 / (LT + 1) + 1;
    static constexpr int PC = 256 / (LT + 1);                 // pairs per chunk: (pair, m) lanes for m = 0 .. LT
    static constexpr int NB = (L * L + L) * 3;                // gradient-table entries per n
    static constexpr int NLM = NL * M;
    static constexpr int NTASK = NMAX * TRIL, NGRP = 256 / NTASK;          // c_tot tasks and pair groups
    static constexpr int NF = 2 * NL;
    // LDS (doubles)
    static constexpr int oNY = 0, oRI = YW * YW, oQB = 2 * YW * YW, oNLc = 3 * YW * YW;
    static constexpr int oCtot = (oNLc + L + 1) & ~1;
    static constexpr int oRf = oCtot + 2 * NLM;
    static constexpr int oTf = oRf + 2 * NL * L;
    static constexpr int oPart = oTf;                           // the partial sums of c_tot are consumed before T is written
    static constexpr int oRec = oTf + (2 * NMAX * NB > 2 * NTASK * NGRP ? 2 * NMAX * NB : 2 * NTASK * NGRP);
    static constexpr int AREA_REC = PC * REC + PC + 2, AREA_ROWS = RCAP * NE;
    static constexpr int AREA = AREA_REC > AREA_ROWS ? AREA_REC : AREA_ROWS;
    static constexpr int TOTAL = oRec + (AREA > 2 * SO3_SELFW ? AREA : 2 * SO3_SELFW);
    static_assert(L * PC <= 256 && (LT + 1) * PC <= 256 && NE <= SO3_SELFW && NB <= 256 && D <= 64 && NGRP >= 1, "lane maps");
};

// The radial channels of all pairs of a batch in a pass of their own.  Inside the centre kernel this is the half of a workgroup's
// life that does not shrink: 48 sixteen-byte coefficient loads per lane behind the chain load -> root -> segment -> load, at three
// workgroups per CU.  Per pair the work is independent of the centre, so for large batches it runs as a streaming kernel -- lane =
// (pair, l), no LDS, no barrier, as many wavefronts per SIMD as the registers allow -- and leaves the head of the pair's record,
// geometry | A_nl | B_nl ([P][8 + 2 NLP] doubles), for the centre kernel to copy (13 KB per centre, contiguous).  Same arithmetic as the radial half of build_records below.
// The Chebyshev table of the radial functions is small (nseg x 16 x 30 doubles = 34.5 KB): every workgroup copies it into LDS once
// and then walks its share of the (pair, l) lanes -- the 48 sixteen-byte coefficient reads of a lane come from LDS (twice the rate of
// the texture path, no address arithmetic per load) instead of through L1.  A segment's block is padded to SO3_RSEG 16-byte pieces
// (245: an odd multiple of five, so the five l of pairs in neighbouring segments fall into different bank groups).
constexpr int SO3_RSEG = 245;
template <int NMAX, int LMAX>
__global__ void __launch_bounds__(256) so3_radial_pairs_kernel(so3_args P, int ncentres, double *ab) {
    using F = so3_fx<NMAX, LMAX>;
    constexpr int L = F::L, NLP = F::NLP, NF = F::NF;
    static_assert(SO3_NCHEB * (NF / 2) <= SO3_RSEG, "a segment's coefficients fit its padded block");
    extern __shared__ __attribute__((aligned(16))) double2 rtab[];              // [nseg][SO3_RSEG]
    const int npairs = P.centre_pair0[ncentres];              // (the caller's pair count may be a capacity: nbr.hip, capped builds)
    const long long nlanes = (long long)npairs * L;
    if ((long long)blockIdx.x * 256 >= nlanes) return;       // (uniform per workgroup: nobody is left behind at the barrier)
    {
        const double2 *src = (const double2 *)P.table;
        const int per_seg = SO3_NCHEB * (NF / 2);
        for (int e = threadIdx.x; e < P.nseg * per_seg; e += 256) {
            const int sg = e / per_seg;
            rtab[sg * SO3_RSEG + (e - sg * per_seg)] = src[e];
        }
    }
    __syncthreads();
    const so3_tab tab = so3_table_layout(NMAX, LMAX, P.nseg, SO3_NCHEB);
    const double seg_scale = (double)P.nseg / P.rcut, inv_rcut = 1.0 / P.rcut;
    for (long long t_ = (long long)blockIdx.x * 256 + threadIdx.x; t_ < nlanes; t_ += (long long)gridDim.x * 256) {
        const int p = (int)(t_ / L), l = (int)(t_ - (long long)p * L);
        const double vx = P.vec[3 * (size_t)p], vy = P.vec[3 * (size_t)p + 1], vz = P.vec[3 * (size_t)p + 2];
        const double r2 = vx * vx + vy * vy + vz * vz, w = P.wj[p];
        int seg = (int)(__builtin_sqrtf((float)r2) * (float)seg_scale);
        seg = seg < 0 ? 0 : (seg > P.nseg - 1 ? P.nseg - 1 : seg);
        const double2 *cb = rtab + seg * SO3_RSEG + l;
        const double r = sqrt(r2);
        double sarg, carg;
        sincospi(r * inv_rcut, &sarg, &carg);
        const double fc = 0.5 * (carg + 1.0), dfc = -0.5 * M_PI * inv_rcut * sarg;
        const double t = (r * seg_scale - (seg + 0.5)) * 2.0, t2 = 2.0 * t;
        const double nw = P.table[tab.oNL + l] * w;
        double *out = ab + (size_t)p * (8 + 2 * NLP) + 8;           // the pair's record head: [ux uy uz 1/r r - - -] | B | A
        if (l == 0) {
            const double inv_r = 1.0 / r;
            out[-8] = vx * inv_r; out[-7] = vy * inv_r; out[-6] = vz * inv_r; out[-5] = inv_r; out[-4] = r;
        }
        // T_k(t) once, k = 0 .. 15 (the same for the three channels of the lane), then three dot products with the coefficients
        double T[SO3_NCHEB];
        T[0] = 1.0; T[1] = t;
#pragma unroll
        for (int k = 2; k < SO3_NCHEB; k++) T[k] = fma(t2, T[k - 1], -T[k - 2]);
#pragma unroll
        for (int n = 0; n < NMAX; n++) {
            double f0 = 0.0, f1 = 0.0;
#pragma unroll
            for (int k = 0; k < SO3_NCHEB; k++) {
                const double2 c = cb[n * L + k * (NF / 2)];
                f0 = fma(c.x, T[k], f0); f1 = fma(c.y, T[k], f1);
            }
            out[NLP + n * L + l] = nw * (fc * f1 + dfc * f0);
            out[n * L + l] = nw * fc * f0;
        }
    }
}

#ifndef GPR_SO3_FIXED_WAVES
#define GPR_SO3_FIXED_WAVES 3
#endif
template <int NMAX, int LMAX, bool PREPASS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PREPASS ? 5 : GPR_SO3_FIXED_WAVES, PREPASS ? 5 : 4)))
so3_fixed_kernel(so3_args P) {      // with the pre-pass: <= 96 registers and 31 KB of LDS, five workgroups per CU
    using F = so3_fx<NMAX, LMAX, PREPASS>;
    constexpr int L = F::L, M = F::M, YW = F::YW, LT = F::LT, NPI = F::NPI, NE = F::NE, NLP = F::NLP, REC = F::REC,
                  PC = F::PC, NB = F::NB, NLM = F::NLM, NTASK = F::NTASK, NGRP = F::NGRP, NF = F::NF, TRIL = F::TRIL;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const double *sNY = smem + F::oNY, *sQA = smem + F::oRI, *sQB = smem + F::oQB, *sNL = smem + F::oNLc;
    cplx *ctot = (cplx *)(smem + F::oCtot), *sRf = (cplx *)(smem + F::oRf), *sTf = (cplx *)(smem + F::oTf);
    double *recs = smem + F::oRec, *sSelf = recs;
    const int tid = threadIdx.x;
    // Wavefront w of every workgroup runs on SIMD w, and the phases below give the wavefronts unequal work (the heavy l of pass 2,
    // the long m chains, the table entries).  The roles are therefore rotated from workgroup to workgroup -- `vt` is the thread
    // id all task maps use -- so that the workgroups resident on a CU (ids 256 apart) load its four SIMDs evenly.
    const int wave = ((tid >> 6) + (blockIdx.x >> 8) + (blockIdx.x >> 3)) & 3, vt = (wave << 6) | (tid & 63);
    const int ic = blockIdx.x;
    const int p0 = P.centre_pair0[ic], p1 = P.centre_pair0[ic + 1];
    const int r0 = P.centre_row0[ic], r1 = P.centre_row0[ic + 1];
    const so3_tab tab = so3_table_layout(NMAX, LMAX, P.nseg, SO3_NCHEB);
    // angular constants in the form the recurrence below wants them: N_lm (-1)^m (2m - 1)!!, (2l - 1) / (l - m), (l + m - 1) / (l - m)
    for (int e = tid; e < YW * YW; e += 256) {
        const int l = e / YW, m = e - l * YW;
        double pref = 1.0;
        for (int k = 1; k <= m; k++) pref *= -(double)(2 * k - 1);
        const double ri = P.table[tab.oRI + e];                 // 1 / (l - m) for l > m, else 0
        smem[F::oNY + e] = P.table[tab.oNY + e] * pref;
        smem[F::oRI + e] = (2 * l - 1) * ri;
        smem[F::oQB + e] = (l + m - 1) * ri;
    }
    if (tid < L) smem[F::oNLc + tid] = P.table[tab.oNL + tid];
    __syncthreads();

    // lane maps of the two halves of a chunk's tables: (l or m) = vt / PC, pair = vt % PC
    const int hi = vt / PC, q_lane = vt - hi * PC;
    const double seg_scale = (double)P.nseg / P.rcut, inv_rcut = 1.0 / P.rcut;
    // (hi_ / q_: the lane's coordinates behind an empty asm, so that the addresses derived from them are formed where they are
    //  used -- hoisted out of the chunk loops, which almost always run once, they only sit in registers and spill)
    auto build_records = [&](int c0, int npc) {
        int hi_ = hi, q_ = q_lane;
        asm volatile("" : "+v"(hi_), "+v"(q_));
        if constexpr (PREPASS) {
            // geometry and B_nl from the pre-pass: the first 8 + NLP doubles of every pair's head (A_nl stays behind them in
            // global memory for pass 2)
            constexpr int COPY2 = (8 + NLP) / 2;                 // double2 per pair
            const double2 *src = (const double2 *)(P.ab + (size_t)c0 * F::HEAD);
            for (int e = tid; e < npc * COPY2; e += 256) {
                const int q = e / COPY2, j = e - q * COPY2;
                *(double2 *)(recs + q * REC + 2 * j) = src[q * (F::HEAD / 2) + j];
            }
        } else if (hi_ < L && q_ < npc) {                       // radial channels (n, l = hi) of pair q  (not PREPASS)
            const int l = hi_, p = c0 + q_;
            double *rec = recs + q_ * REC;
            const double vx = P.vec[3 * (size_t)p], vy = P.vec[3 * (size_t)p + 1], vz = P.vec[3 * (size_t)p + 2];
            const double r2 = vx * vx + vy * vy + vz * vz, w = P.wj[p];
            // The segment of the Chebyshev table from a single-precision root: the coefficient loads -- the long latency of this
            // phase -- then do not wait for the double-precision one.  A radius within 1e-7 of a segment boundary may land in
            // the neighbouring segment, whose polynomial is just as good there (t a hair outside [-1, 1]).
            int seg = (int)(__builtin_sqrtf((float)r2) * (float)seg_scale);
            seg = seg < 0 ? 0 : (seg > P.nseg - 1 ? P.nseg - 1 : seg);
            const double2 *cb = (const double2 *)(P.table + (size_t)seg * SO3_NCHEB * NF) + l;
            // coefficients of F0 / F1 (one 16-byte load for both) in halves of eight: the halves of channel n + 1 are
            // requested while those of channel n are multiplied, 16 loads in flight at any time
            constexpr int H = SO3_NCHEB / 2;
            double2 ca[H], cl[H];
#pragma unroll
            for (int k = 0; k < H; k++) ca[k] = cb[(H + k) * (NF / 2)];
#pragma unroll
            for (int k = 0; k < H; k++) cl[k] = cb[k * (NF / 2)];
            const double r = sqrt(r2);
            double sarg, carg;
            sincospi(r * inv_rcut, &sarg, &carg);              // cutoff 0.5 (cos(pi r / rcut) + 1) and its derivative
            const double fc = 0.5 * (carg + 1.0), dfc = -0.5 * M_PI * inv_rcut * sarg;
            const double t = (r * seg_scale - (seg + 0.5)) * 2.0, t2 = 2.0 * t;
            const double nw = sNL[l] * w;                       // SO3.py:206, 221
            // sum_k c_k T_k(t) with T_k formed on the way (T_{k+1} = 2 t T_k - T_{k-1}: two registers instead of a table of
            // sixteen per lane); the upper half of the coefficients is multiplied first, so T_8, T_9 are carried over
            double T8, T9;
#pragma unroll
            for (int n = 0; n < NMAX; n++) {
                double f0 = 0.0, f1 = 0.0;
                if (n == 0) {
                    double ta = 1.0, tb = t;                    // T_0, T_1 -> T_8, T_9 once
#pragma unroll
                    for (int k = 2; k <= H + 1; k++) { const double tc = fma(t2, tb, -ta); ta = tb; tb = tc; }
                    T8 = ta; T9 = tb;
                }
                {
                    double ta = T8, tb = T9;
#pragma unroll
                    for (int k = 0; k < H; k++) {
                        f0 = fma(ca[k].x, ta, f0); f1 = fma(ca[k].y, ta, f1);
                        const double tc = fma(t2, tb, -ta); ta = tb; tb = tc;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);              // (the scheduler would otherwise hoist every load of every
                if (n + 1 < NMAX) {                             //  channel to the top and spill half of them)
#pragma unroll
                    for (int k = 0; k < H; k++) ca[k] = cb[(n + 1) * L + (H + k) * (NF / 2)];
                }
                __builtin_amdgcn_sched_barrier(0);
                {
                    double ta = 1.0, tb = t;
#pragma unroll
                    for (int k = 0; k < H; k++) {
                        f0 = fma(cl[k].x, ta, f0); f1 = fma(cl[k].y, ta, f1);
                        const double tc = fma(t2, tb, -ta); ta = tb; tb = tc;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (n + 1 < NMAX) {
#pragma unroll
                    for (int k = 0; k < H; k++) cl[k] = cb[(n + 1) * L + k * (NF / 2)];
                }
                __builtin_amdgcn_sched_barrier(0);
                rec[F::OA + n * L + l] = nw * (fc * f1 + dfc * f0);
                rec[F::OB + n * L + l] = nw * fc * f0;
            }
            if (l == 0) {
                const double inv_r = 1.0 / r;
                rec[0] = vx * inv_r; rec[1] = vy * inv_r; rec[2] = vz * inv_r; rec[3] = inv_r; rec[4] = r;
            }
        }
        __syncthreads();
        asm volatile("" : "+v"(hi_), "+v"(q_));
        if (hi_ <= LT && q_ < npc) {                            // Y_l^m, m = hi, of pair q
            const int m = hi_;
            double *rec = recs + q_ * REC;
            const double ux = rec[0], uy = rec[1], ct = rec[2];
            double er = 1.0, ei = 0.0;                          // (u_x + i u_y)^m
#pragma unroll
            for (int k = 1; k <= LT; k++)
                if (k <= m) {
                    const double e2 = er * ux - ei * uy;
                    ei = er * uy + ei * ux; er = e2;
                }
            cplx *Y = (cplx *)(rec + F::OY);
            // Q_m^m = 1, Q_l^m = ((2l - 1) cos(theta) Q_{l-1}^m - (l + m - 1) Q_{l-2}^m) / (l - m) with Q_{m-1}^m = 0
            double q2 = 0.0, q1 = 1.0;
            { const double nrm = sNY[m * YW + m]; Y[m * (m + 1) / 2 + m] = {nrm * er, nrm * ei}; }
#pragma unroll
            for (int s_ = 1; s_ <= LT; s_++) {
                const int l = m + s_;
                if (l <= LT) {
                    const double ql = sQA[l * YW + m] * ct * q1 - sQB[l * YW + m] * q2;
                    q2 = q1; q1 = ql;
                    const double nrm = sNY[l * YW + m] * ql;
                    Y[l * (l + 1) / 2 + m] = {nrm * er, nrm * ei};
                }
            }
        }
        __syncthreads();
    };

    // ---- pass 1: c_tot[nl][m >= 0] = sum over pairs of B_nl Y_l^m: task = (n, l, m), NGRP groups of lanes take the pairs
    //      q = g, g + NGRP, ... and their partial sums are added in group order
    const int nchunks = (p1 - p0 + PC - 1) / PC;
    {
        const int grp = vt / NTASK, task = vt - grp * NTASK;
        const int n = task / TRIL, rem = task - n * TRIL;
        int l = 0;
        while ((l + 1) * (l + 2) / 2 <= rem) l++;
        const int m = rem - l * (l + 1) / 2, nl = n * L + l;
        const bool mine = grp < NGRP;
        cplx acc = {0.0, 0.0};
        for (int c0 = p0; c0 < p1; c0 += PC) {
            const int npc = min(PC, p1 - c0);
            build_records(c0, npc);
            GPR_PHASE_EXIT(P.dbg, 1)
            if (mine)
                for (int q = grp; q < npc; q += NGRP) {
                    const double *rec = recs + q * REC;
                    const cplx y = ((const cplx *)(rec + F::OY))[l * (l + 1) / 2 + m];
                    const double b = rec[F::OB + nl];
                    acc.re = fma(b, y.re, acc.re); acc.im = fma(b, y.im, acc.im);
                }
            if (nchunks > 1) __syncthreads();                  // records are rebuilt by the next chunk
        }
        cplx *sPart = (cplx *)(smem + F::oPart);
        if (mine) sPart[grp * NTASK + task] = acc;
        for (int e = tid; e < NLM; e += 256) ctot[e] = {0.0, 0.0};
        __syncthreads();
        if (grp == 0) {
#pragma unroll
            for (int g = 1; g < NGRP; g++) { const cplx o = sPart[g * NTASK + task]; acc.re += o.re; acc.im += o.im; }
            ctot[nl * M + LMAX + m] = acc;
            if (m > 0) { const double sg = (m & 1) ? -1.0 : 1.0; ctot[nl * M + LMAX - m] = {sg * acc.re, -sg * acc.im}; }
            sRf[nl * L + m] = (m == 0) ? acc : cplx{2.0 * acc.re, 2.0 * acc.im};
        }
    }
    __syncthreads();

    // ---- power spectrum x_i (the last lanes of the block) and the gradient tables (the first NB) side by side
    if (vt >= 256 - F::D) {
        const int c = vt - (256 - F::D), pi = c / L, l = c - pi * L;
        int n1 = 0;
        while ((n1 + 1) * (n1 + 2) / 2 <= pi) n1++;
        const int n2 = pi - n1 * (n1 + 1) / 2;
        const cplx *a = ctot + (n1 * L + l) * M, *b = ctot + (n2 * L + l) * M;
        double s = 0.0;
#pragma unroll
        for (int mi = 0; mi < M; mi++) s += a[mi].re * b[mi].re + a[mi].im * b[mi].im;
        P.x[(size_t)ic * F::D + c] = s;
    }
    if (vt < NB) {
        // entry `vt` of the linear map c_tot -> T (so3_grad_map_kernel): [l, nterm <= 4, 4 x (source m, p, q, r, s)], applied to
        // the three n one after the other
        const double2 *g2 = (const double2 *)(P.table + tab.oGT + (size_t)vt * SO3_GT_W);
        const double2 head = g2[0];
        const int l = (int)head.x, nterm = (int)head.y;
        if (l >= 1) {
            double2 tm[10];
#pragma unroll
            for (int k = 0; k < 10; k++) tm[k] = g2[1 + k];
            const double *tv = (const double *)tm;              // 4 x [m, p, q, r, s]
#pragma unroll
            for (int n = 0; n < NMAX; n++) {
                cplx out = {0.0, 0.0};
#pragma unroll
                for (int t = 0; t < 4; t++)
                    if (t < nterm) {
                        const cplx c = ctot[(n * L + l) * M + LMAX + (int)tv[5 * t]];
                        out.re += tv[5 * t + 1] * c.re + tv[5 * t + 2] * c.im;
                        out.im += tv[5 * t + 3] * c.re + tv[5 * t + 4] * c.im;
                    }
                sTf[n * NB + vt] = out;
            }
        }
    }
    __syncthreads();
    GPR_PHASE_EXIT(P.dbg, 2)

    // ---- pass 2: dP rows.  Lane = (pair q, l), heavy l first: l = L - 1 - vt / PC
    if (r0 == r1) return;                         // no rows at all: the emptied table of an overflowed capped build (nbr.hip)
    const int rself = P.self_row[ic];
    int *sMap = (int *)(recs + PC * REC);                    // [PC] row of the pair (centre-local), [PC] index in its row, max
    double *rowOut = recs;
    constexpr int NSP = 256 / NE >= 2 ? 2 : 1;               // (i, i) row: column sums in two interleaved halves
    const int spart = tid / NE, scol = tid - spart * NE;
    double selfacc = 0.0;
    const int l_lane = L - 1 - hi;                           // (negative l: idle lane)
    // the largest l among the lanes of this wavefront: terms beyond it are skipped by the whole wavefront
    const int lw = __builtin_amdgcn_readfirstlane(max(0, L - 1 - (64 * wave) / PC));
    for (int c0 = p0; c0 < p1; c0 += PC) {
        const int npc = min(PC, p1 - c0), c1 = c0 + npc;
        if (nchunks > 1) build_records(c0, npc);              // a single chunk is still resident from pass 1
        if (tid == 0) sMap[2 * PC] = 0;
        __syncthreads();
        for (int row = r0 + tid; row < r1; row += 256) {
            const int a = max(P.row_pair0[row], c0), b = min(P.row_pair0[row + 1], c1);
            for (int p = a; p < b; p++) { sMap[p - c0] = row - r0; sMap[PC + p - c0] = p - a; }
            if (b > a) atomicMax(&sMap[2 * PC], b - a);
        }
        int l = l_lane, q = q_lane;
        asm volatile("" : "+v"(l), "+v"(q));                 // (see build_records)
        const bool active = l >= 0 && q < npc;
        double outv[NPI * 3];
        if (active) {
            const double *rec = recs + q * REC;
            const cplx *Y = (const cplx *)(rec + F::OY);
            double Rn[NMAX], Sn[NMAX][3];
#pragma unroll
            for (int n = 0; n < NMAX; n++) { Rn[n] = 0.0; Sn[n][0] = 0.0; Sn[n][1] = 0.0; Sn[n][2] = 0.0; }
            {
                const cplx *yl = Y + l * (l + 1) / 2, *rf = sRf + l * L;
#pragma unroll
                for (int m = 0; m < L; m++)
                    if (m <= lw && m <= l) {
                        const cplx y = yl[m];
#pragma unroll
                        for (int n = 0; n < NMAX; n++) { const cplx c = rf[n * L * L + m]; Rn[n] += y.re * c.re + y.im * c.im; }
                    }
            }
            if (l >= 1) {
                const cplx *y0 = Y + (l - 1) * l / 2, *y1 = Y + (l + 1) * (l + 2) / 2;
                const cplx *tf = sTf + so3_tf_base(0, l, L);
                const int kstep = 2 * l + 2;
#pragma unroll
                for (int m = 0; m < L - 1; m++)
                    if (m < lw && m < l) {
                        const cplx y = y0[m];
#pragma unroll
                        for (int n = 0; n < NMAX; n++)
#pragma unroll
                            for (int k = 0; k < 3; k++) {
                                const cplx c = tf[n * NB + k * kstep + m];
                                Sn[n][k] += y.re * c.re + y.im * c.im;
                            }
                    }
#pragma unroll
                for (int m = 0; m < L + 1; m++)
                    if (m < lw + 2 && m < l + 2) {
                        const cplx y = y1[m];
#pragma unroll
                        for (int n = 0; n < NMAX; n++)
#pragma unroll
                            for (int k = 0; k < 3; k++) {
                                const cplx c = tf[n * NB + k * kstep + l + m];
                                Sn[n][k] += y.re * c.re + y.im * c.im;
                            }
                    }
            }
            const double inv_r = rec[3], u[3] = {rec[0], rec[1], rec[2]};
            double An[NMAX], Bn[NMAX];
#pragma unroll
            for (int n = 0; n < NMAX; n++) {
                An[n] = PREPASS ? P.ab[(size_t)(c0 + q) * F::HEAD + 8 + NLP + n * L + l] : rec[F::OA + n * L + l];
                Bn[n] = rec[F::OB + n * L + l] * inv_r;
            }
#pragma unroll
            for (int n1 = 0; n1 < NMAX; n1++)
#pragma unroll
                for (int n2 = 0; n2 <= n1; n2++) {
                    const double ar = An[n1] * Rn[n2] + An[n2] * Rn[n1];
#pragma unroll
                    for (int k = 0; k < 3; k++)
                        outv[(n1 * (n1 + 1) / 2 + n2) * 3 + k] = u[k] * ar + Bn[n1] * Sn[n2][k] + Bn[n2] * Sn[n1][k];
                }
        }
        __syncthreads();                                  // records read, pair -> row map complete
#ifdef GPR_PHASE_ABLATION
        if (P.dbg == 3) { if (active && outv[0] == 1.234567e300) P.x[0] = 0.0; return; }   // ablation: keep outv alive
#endif
        const int myrow = active ? sMap[q] : 0, myj = active ? sMap[PC + q] : -1;
        const int rowb = sMap[0], nrows_c = sMap[npc - 1] - rowb + 1, maxj = sMap[2 * PC];
        const bool cont = P.row_pair0[r0 + rowb] < c0;    // the chunk's first row began in the previous chunk
        __syncthreads();                                  // map read: the row buffer may take its place
        const bool mine = active && r0 + myrow != rself;  // periodic images of the centre itself cancel (SO3.py:266-273)
        static_assert(NE % 2 == 0, "rows are whole 16-byte pieces");
        // the chunk's rows go through the staging buffer in slabs of at most RCAP rows (one slab unless the buffer was made
        // smaller than a chunk's rows: PREPASS)
        for (int rb = 0; rb < nrows_c; rb += F::RCAP) {
            const int nr = min(F::RCAP, nrows_c - rb), lrow = myrow - rowb - rb;
            const bool here = mine && lrow >= 0 && lrow < nr;
            double *dst = rowOut + lrow * NE + l * 3;
            for (int j = 0; j < maxj; j++) {
                if (here && myj == j) {
#pragma unroll
                    for (int pi = 0; pi < NPI; pi++)
#pragma unroll
                        for (int k = 0; k < 3; k++) {
                            double *o = dst + pi * L * 3 + k;
                            *o = j == 0 ? outv[pi * 3 + k] : *o + outv[pi * 3 + k];
                        }
                }
                __syncthreads();
            }
            GPR_PHASE_EXIT(P.dbg, 4)
            // the slab of dxdr, flat and 16 bytes at a time (a row is NE * 8 = 720 bytes); the (i, i) row is skipped
            {
                double2 *gdst = (double2 *)(P.dxdr + (size_t)(r0 + rowb + rb) * NE);
                const double2 *src = (const double2 *)rowOut;
                for (int f = tid; f < nr * (NE / 2); f += 256) {
                    const int rl = f / (NE / 2);
                    if (r0 + rowb + rb + rl == rself) continue;
                    double2 v = src[f];
                    if (cont && rb + rl == 0) { const double2 o = gdst[f]; v.x += o.x; v.y += o.y; }
                    gdst[f] = v;
                }
            }
            if (spart < NSP)
                for (int rl = spart; rl < nr; rl += NSP)
                    if (r0 + rowb + rb + rl != rself) selfacc += rowOut[rl * NE + scol];
            __syncthreads();                              // staging buffer consumed (the next slab / chunk reuses it)
        }
    }
    if (spart < NSP) sSelf[spart * SO3_SELFW + scol] = selfacc;
    __syncthreads();
    for (int e = tid; e < NE; e += 256) P.dxdr[(size_t)rself * NE + e] = -(sSelf[e] + (NSP == 2 ? sSelf[SO3_SELFW + e] : 0.0));
}

int so3_check_limits(int nmax, int lmax, int nq, int nseg, int ncheb, const char *who) {
    const int d = nmax * (nmax + 1) / 2 * (lmax + 1), L = lmax + 1;
    if (nmax < 1 || lmax < 0 || lmax > SO3_LCAP - 1 || nmax > SO3_NCAP || (nmax == 5 && lmax > 4) || (nmax > 5 && lmax > 3) || nq > SO3_NQCAP ||
        nmax * L > so3_nlp(nmax, lmax) || d * 3 > SO3_SELFW || nmax * L * (L + 1) / 2 > 256 || nseg < 1 || ncheb != SO3_NCHEB) {
        gpr_set_error("%s: (nmax=%d, lmax=%d, nq=%d, nseg=%d, ncheb=%d) outside this build's limits "
                      "(nmax <= 4 with lmax <= %d, nmax = 5 with lmax <= 4, nmax <= %d with lmax <= 3, nq <= %d, ncoef <= 85, ncheb == %d)",
                      who, nmax, lmax, nq, nseg, ncheb, SO3_LCAP - 1, SO3_NCAP, SO3_NQCAP, SO3_NCHEB);
        return GPR_ERR_UNSUPPORTED;
    }
    return GPR_OK;
}

}  // namespace

// 0 (default): configurations that have a compile-time instance take it; 1: always the generic kernel (tests hold the
// two against each other)
static std::atomic<int> g_so3_variant{0};
extern "C" int gpr_so3_set_variant(int generic_only) { g_so3_variant.store(generic_only ? 1 : 0); return GPR_OK; }
// smallest number of centres for which the radial channels are formed by the streaming pre-pass (tests / A-B: 0 always,
// a huge number never)
static std::atomic<int> g_so3_prepass_min{getenv("GPR_SO3_PREPASS_MIN") ? atoi(getenv("GPR_SO3_PREPASS_MIN")) : 512};    // centres from which the radial pre-pass pays (A/B: env or gpr_so3_set_prepass_min)
extern "C" int gpr_so3_set_prepass_min(int ncentres) { g_so3_prepass_min.store(ncentres < 0 ? 0 : ncentres); return GPR_OK; }

extern "C" long long gpr_so3_table_size(int nmax, int lmax, int nseg, int ncheb) {
    return so3_table_layout(nmax, lmax, nseg, ncheb).total;
}

// Builds the radial Chebyshev table and the angular constant tables on the device (once per descriptor
// configuration).  G[nmax, nq], rq[nq]: the reference's quadrature (SO3.py:446-453, 623-655), device.
extern "C" int gpr_so3_tables(void *stream, int nmax, int lmax, double rcut, double alpha, int nq, const double *G,
                              const double *rq, int nseg, int ncheb, double *table) {
    hipStream_t st = (hipStream_t)stream;
    int rc = so3_check_limits(nmax, lmax, nq, nseg, ncheb, "gpr_so3_tables");
    if (rc) return rc;
    GPR_REQUIRE(G && rq && table && nq >= 1 && rcut > 0.0, "gpr_so3_tables: null pointer or bad sizes");
    const so3_tab tab = so3_table_layout(nmax, lmax, nseg, ncheb);
    const int nr = nseg * ncheb;
    void *p;
    if ((rc = gpr_dev_alloc(&p, (size_t)nr * (1 + tab.NF) * sizeof(double), st))) return rc;
    double *rs = (double *)p, *vals = rs + nr;
    hipLaunchKernelGGL(so3_cheb_nodes_kernel, dim3((nr + 255) / 256), dim3(256), 0, st, nseg, ncheb, rcut, rs);
    hipLaunchKernelGGL(so3_radial_direct_kernel, dim3(nr), dim3(64), 0, st, nmax, lmax, alpha, nq, G, rq, nr, rs, vals);
    hipLaunchKernelGGL(so3_cheb_coef_kernel, dim3((nr * tab.NF + 255) / 256), dim3(256), 0, st, nseg, ncheb, tab.NF, vals, table);
    hipLaunchKernelGGL(so3_const_tables_kernel, dim3(1), dim3(256), 0, st, lmax, table + tab.oNY, table + tab.oCG, table + tab.oNL, table + tab.oRI);
    {
        const int NB = ((lmax + 1) * (lmax + 1) + lmax + 1) * 3;
        hipLaunchKernelGGL(so3_grad_map_kernel, dim3((NB + 63) / 64), dim3(64), 0, st, lmax, (const double *)(table + tab.oCG), table + tab.oGT);
    }
    hipError_t e = hipGetLastError();
    gpr_dev_free(p, st);
    if (e != hipSuccess) { gpr_set_error("gpr_so3_tables: launch failed: %s", hipGetErrorString(e)); return GPR_ERR_HIP; }
    return GPR_OK;
}

// F0_nl(r), F1_nl(r) at arbitrary radii rs[nr] (device): mode 0 = direct quadrature, 1 = Chebyshev table.
// out[nr, 2 nmax (lmax+1)].  Test / validation entry point.
extern "C" int gpr_so3_radial(void *stream, int mode, int nmax, int lmax, double rcut, double alpha, int nq, const double *G,
                              const double *rq, int nseg, int ncheb, const double *table, int nr, const double *rs,
                              double *out) {
    hipStream_t st = (hipStream_t)stream;
    int rc = so3_check_limits(nmax, lmax, nq, nseg, ncheb, "gpr_so3_radial");
    if (rc) return rc;
    if (nr <= 0) return GPR_OK;
    GPR_REQUIRE(rs && out, "gpr_so3_radial: null pointer");
    const int NF = 2 * nmax * (lmax + 1);
    if (mode == 0) {
        GPR_REQUIRE(G && rq, "gpr_so3_radial: direct mode needs G and rq");
        hipLaunchKernelGGL(so3_radial_direct_kernel, dim3(nr), dim3(64), 0, st, nmax, lmax, alpha, nq, G, rq, nr, rs, out);
    } else {
        GPR_REQUIRE(table, "gpr_so3_radial: table mode needs the table");
        hipLaunchKernelGGL(so3_radial_eval_kernel, dim3((nr * NF + 255) / 256), dim3(256), 0, st, table, nseg, ncheb, NF, rcut, nr, rs, out);
    }
    GPR_HIP_TRY(hipGetLastError());
    return GPR_OK;
}

// x[nc, d], dxdr[S, d, 3] from neighbour pairs grouped by centre and, inside a centre, by seq row.
// All pointers are device pointers; `table` comes from gpr_so3_tables with the same (nmax, lmax, rcut, alpha).
extern "C" int gpr_so3_power_spectrum(void *stream, int nmax, int lmax, double rcut, double alpha, int nseg, int ncheb,
                                      const double *table, int ncentres, int npairs, int nrows,
                                      const double *vec, const double *wj, const int *centre_pair0,
                                      const int *centre_row0, const int *row_pair0, const int *self_row,
                                      double *x, double *dxdr) {
    return gpr_so3_power_spectrum_stress(stream, nmax, lmax, rcut, alpha, nseg, ncheb, table, ncentres, npairs, nrows, vec, wj,
                                         centre_pair0, centre_row0, row_pair0, self_row, x, dxdr, nullptr, nullptr, nullptr);
}

// same with the stress rows rdxdr[S, d, 3, 3] = -pstress / volume of SO3.calculate(stress=True) (SO3.py:240-306);
// positions[nc, 3] = centre positions as given by the caller, inv_vol[nc] = 1 / cell volume per centre (device);
// rdxdr == NULL switches the stress rows off
extern "C" int gpr_so3_power_spectrum_stress(void *stream, int nmax, int lmax, double rcut, double alpha, int nseg, int ncheb,
                                             const double *table, int ncentres, int npairs, int nrows,
                                             const double *vec, const double *wj, const int *centre_pair0,
                                             const int *centre_row0, const int *row_pair0, const int *self_row,
                                             double *x, double *dxdr, const double *positions, const double *inv_vol,
                                             double *rdxdr) {
    hipStream_t st = (hipStream_t)stream;
    GPR_REQUIRE(!rdxdr || (positions && inv_vol), "gpr_so3_power_spectrum_stress: rdxdr needs positions and inv_vol");
    const int d = nmax * (nmax + 1) / 2 * (lmax + 1);
    if (rdxdr && (nmax * (lmax + 1) > 16 || d * 3 > 96)) {     // the stress rows keep the narrower lane layout
        gpr_set_error("gpr_so3_power_spectrum_stress: stress rows need nmax (lmax + 1) <= 16 and ncoef <= 32 (nmax=%d, lmax=%d)", nmax, lmax);
        return GPR_ERR_UNSUPPORTED;
    }
    int rc = so3_check_limits(nmax, lmax, 1, nseg, ncheb, "gpr_so3_power_spectrum");
    if (rc) return rc;
    if (rdxdr && nmax > 4) {
        gpr_set_error("gpr_so3_power_spectrum_stress: stress rows are built for nmax <= 4 (nmax=%d)", nmax);
        return GPR_ERR_UNSUPPORTED;
    }
    if (ncentres <= 0) return GPR_OK;
    GPR_REQUIRE(table && centre_pair0 && centre_row0 && row_pair0 && self_row && x && dxdr,
                "gpr_so3_power_spectrum: null pointer");
    GPR_REQUIRE(npairs == 0 || (vec && wj), "gpr_so3_power_spectrum: null pair arrays");
    (void)nrows;
    so3_args A;
    A.nmax = nmax; A.lmax = lmax; A.d = d; A.nseg = nseg; A.ncheb = ncheb; A.rcut = rcut; A.alpha = alpha;
    A.dbg = GPR_PHASE_ENV("GPR_SO3_DBG");          // phase ablation: tools build only (gpr_internal.h); 0 otherwise
    A.table = table; A.vec = vec; A.wj = wj; A.centre_pair0 = centre_pair0; A.centre_row0 = centre_row0;
    A.row_pair0 = row_pair0; A.self_row = self_row; A.x = x; A.dxdr = dxdr;
    A.positions = positions; A.inv_vol = inv_vol; A.rdxdr = rdxdr; A.ab = nullptr;
    const so3_lds lay = so3_layout(nmax, lmax, rdxdr != nullptr);
    const size_t lds_bytes = (size_t)lay.total * sizeof(double);
    if (lds_bytes > 96 * 1024) {
        gpr_set_error("gpr_so3_power_spectrum: (nmax=%d, lmax=%d) needs %zu bytes of LDS", nmax, lmax, lds_bytes);
        return GPR_ERR_UNSUPPORTED;
    }
    {
        int rc_ = gpr_func_lds_once((const void *)so3_centre_kernel<5, 3, false>, 96 * 1024);
        if (!rc_) rc_ = gpr_func_lds_once((const void *)so3_centre_kernel<SO3_LCAP, 4, false>, 96 * 1024);
        if (!rc_) rc_ = gpr_func_lds_once((const void *)so3_centre_kernel<5, 3, true>, 96 * 1024);
        if (!rc_) rc_ = gpr_func_lds_once((const void *)so3_centre_kernel<SO3_LCAP, 4, true>, 96 * 1024);
        if (!rc_) rc_ = gpr_func_lds_once((const void *)so3_centre_kernel<4, SO3_NCAP, false>, 96 * 1024);
        if (!rc_) rc_ = gpr_func_lds_once((const void *)so3_centre_kernel<5, 5, false>, 96 * 1024);
        if (rc_) return rc_;
    }
    int pid = -1;                                            // the profiler's bracket around the centre kernel (one bracket, one kernel)
    const bool small_l = lmax + 1 <= 5 && nmax <= 3;
    if (!rdxdr && nmax == 3 && lmax == 4 && g_so3_variant.load() == 0) {
        // the configuration of the reference's defaults: every size a compile-time constant (so3_fixed_kernel)
        using F = so3_fx<3, 4, false>;
        using FP = so3_fx<3, 4, true>;
        if ((rc = gpr_func_lds_once((const void *)so3_fixed_kernel<3, 4, false>, F::TOTAL * (int)sizeof(double)))) return rc;
        if ((rc = gpr_func_lds_once((const void *)so3_fixed_kernel<3, 4, true>, FP::TOTAL * (int)sizeof(double)))) return rc;
        // large batches: the radial channels in a streaming pass of their own (so3_radial_pairs_kernel); a few structures are
        // quicker in one launch
        void *ab = nullptr;
        if (ncentres >= g_so3_prepass_min.load() && npairs > 0) {
            if ((rc = gpr_dev_alloc(&ab, (size_t)npairs * (8 + 2 * F::NLP) * sizeof(double), st))) return rc;
            const long long lanes = (long long)npairs * F::L;
            // workgroups that stay: each copies the coefficient table into LDS once (four per CU by its 35 KB) and walks its share
            const long long want = (lanes + 255) / 256, cap = 4LL * gpr_cu_count();
            const size_t rlds = (size_t)nseg * SO3_RSEG * sizeof(double2);
            const int rpid = gpr_prof_begin(st, GPR_PROF_SO3_RADIAL);       // a kind of its own: one bracket, one kernel
            hipLaunchKernelGGL((so3_radial_pairs_kernel<3, 4>), dim3((unsigned)std::min(want, cap)), dim3(256), rlds, st, A, ncentres, (double *)ab);
            gpr_prof_end(st, rpid);
            A.ab = (const double *)ab;
        }
        pid = gpr_prof_begin(st, GPR_PROF_SO3);
        if (ab) hipLaunchKernelGGL((so3_fixed_kernel<3, 4, true>), dim3(ncentres), dim3(256), FP::TOTAL * sizeof(double), st, A);
        else hipLaunchKernelGGL((so3_fixed_kernel<3, 4, false>), dim3(ncentres), dim3(256), F::TOTAL * sizeof(double), st, A);
        if (ab) gpr_dev_free(ab, st);
    } else if (rdxdr) {
        pid = gpr_prof_begin(st, GPR_PROF_SO3);
        if (small_l) hipLaunchKernelGGL((so3_centre_kernel<5, 3, true>), dim3(ncentres), dim3(256), lds_bytes, st, A);
        else hipLaunchKernelGGL((so3_centre_kernel<SO3_LCAP, 4, true>), dim3(ncentres), dim3(256), lds_bytes, st, A);
    } else {
        pid = gpr_prof_begin(st, GPR_PROF_SO3);
        if (small_l) hipLaunchKernelGGL((so3_centre_kernel<5, 3, false>), dim3(ncentres), dim3(256), lds_bytes, st, A);
        else if (nmax <= 4) hipLaunchKernelGGL((so3_centre_kernel<SO3_LCAP, 4, false>), dim3(ncentres), dim3(256), lds_bytes, st, A);
        else if (lmax <= 3) hipLaunchKernelGGL((so3_centre_kernel<4, SO3_NCAP, false>), dim3(ncentres), dim3(256), lds_bytes, st, A);   // nmax 5, 6: lmax <= 3
        else hipLaunchKernelGGL((so3_centre_kernel<5, 5, false>), dim3(ncentres), dim3(256), lds_bytes, st, A);           // nmax = 5, lmax = 4 (d = 75)
    }
    gpr_prof_end(st, pid);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { gpr_set_error("so3_centre_kernel launch failed: %s", hipGetErrorString(e)); return GPR_ERR_HIP; }
    return GPR_OK;
}
