// features.hip -- explicit feature map of the Dot kernel at zeta = 2.
//
// For zeta = 2 the reference's Dot kernel sigma^2 (d^2 + sigma0^2) (dot_kernel.cpp:5-336) is a polynomial
// kernel of degree 2 in the normalised descriptor rows, so every block of the covariance matrix is an inner
// product of per-point feature vectors (same notation as pack.hip: a~ = x/|x|, B' the projected derivative
// columns; vech packs a symmetric d x d matrix as [diagonal ; sqrt2 * strict lower], 465 numbers for d = 30):
//     energy point i :  Phi_i^s = (1/n_i) sum_{rows a of species s} vech(a~ a~^T),   f_i^s = n_i^s / n_i
//     force point j,q:  S_j^s[q] = sum_{rows b of species s} vech(b~ B'_q^T + B'_q b~^T)
//     K_ee = sigma^2 (Phi.Phi' + sigma0^2 f.f'),  K_ef = -sigma^2 Phi.S[q],  K_ff = sigma^2 S[p].S'[q]
// i.e. K = sigma^2 Psi Psi'^T with rows  Psi_E = [Phi, sigma0 f],  Psi_F = [-S, 0]:  the N x N assembly becomes
// one FP64 MFMA GEMM with inner dimension 466 * n_species instead of a sum over all row pairs.  (Identities:
// d^2 = vech(a~a~^T).vech(b~b~^T);  2 d (a~.B'_q) = vech(a~a~^T).S[q];  d G_pq + w_p z_q = (1/2) S[p].S[q].)
#include <algorithm>
#include "gpr_internal.h"

namespace {

constexpr int FEAT_MAX_SPECIES = 4;
constexpr int FEAT_MAX_EXTRA = 8;

struct feat_args {
    const double *ops; const int *ele; const gpr_chunk *chunks; const int *pt_chunk0; const int *perm;
    const double *inv_cnt;
    int npts, nc, d, npair, ns;
    int species[FEAT_MAX_SPECIES];
    double sigma0;
    double *out; long long ld;
};

// one workgroup per packed point; thread e owns the (k, m) pair e of the packed symmetric index
__global__ void __launch_bounds__(576) dot2_features_kernel(feat_args A) {
    __shared__ __attribute__((aligned(16))) double sT[GPR_CHUNK_TILES * GPR_PLANE];
    __shared__ int sE[GPR_CHUNK_TILES * 16];
    const int p = blockIdx.x, tid = threadIdx.x, e = tid;
    const bool act = e < A.npair;
    int k = 0, m = 0;                                  // e -> (k >= m): e = k(k+1)/2 + m
    if (act) { while ((k + 1) * (k + 2) / 2 <= e) k++; m = e - k * (k + 1) / 2; }
    const double wgt = (k == m) ? 1.0 : 1.4142135623730951;
    const int rows_per_tile = (A.nc == 4) ? 4 : 16;
    double acc[FEAT_MAX_SPECIES][3];
    double cnt[FEAT_MAX_SPECIES];
#pragma unroll
    for (int s = 0; s < FEAT_MAX_SPECIES; s++) { cnt[s] = 0.0; acc[s][0] = acc[s][1] = acc[s][2] = 0.0; }

    const int c0 = A.pt_chunk0[p], c1 = A.pt_chunk0[p + 1];
    for (int c = c0; c < c1; c++) {
        const gpr_chunk ch = A.chunks[c];
        const int ndbl = ch.ntiles * GPR_PLANE;
        const double *src = A.ops + (size_t)ch.tile0 * GPR_PLANE;
        __syncthreads();
        for (int i = tid; i < ndbl; i += blockDim.x) sT[i] = src[i];
        for (int i = tid; i < ch.ntiles * rows_per_tile; i += blockDim.x) sE[i] = A.ele[ch.tile0 * rows_per_tile + i];
        __syncthreads();
        if (!act) continue;
        for (int t = 0; t < ch.ntiles; t++) {
            const double *T = sT + t * GPR_PLANE;
            for (int r = 0; r < rows_per_tile; r++) {
                const int el = sE[t * rows_per_tile + r];
                double v[3] = {0.0, 0.0, 0.0};
                if (A.nc == 4) {
                    // element (row r, comp c, index k) of a stream force tile: T[(k>>2)*64 + r + 4c + 16(k&3)]
                    const double ak = T[(k >> 2) * 64 + r + 12 + 16 * (k & 3)], am = T[(m >> 2) * 64 + r + 12 + 16 * (m & 3)];
#pragma unroll
                    for (int q = 0; q < 3; q++) {
                        const double bk = T[(k >> 2) * 64 + r + 4 * q + 16 * (k & 3)], bm = T[(m >> 2) * 64 + r + 4 * q + 16 * (m & 3)];
                        v[q] = wgt * (ak * bm + am * bk);
                    }
                } else {
                    const double ak = T[(k >> 2) * 64 + r + 16 * (k & 3)], am = T[(m >> 2) * 64 + r + 16 * (m & 3)];
                    v[0] = wgt * ak * am;
                }
#pragma unroll
                for (int s = 0; s < FEAT_MAX_SPECIES; s++)
                    if (s < A.ns && el == A.species[s]) { acc[s][0] += v[0]; acc[s][1] += v[1]; acc[s][2] += v[2]; cnt[s] += 1.0; }
            }
        }
    }
    if (!act) return;
    const int orig = A.perm[p];
    const double scale = (A.nc == 4) ? -1.0 : A.inv_cnt[p];
    const int nq = (A.nc == 4) ? 3 : 1;
    for (int q = 0; q < nq; q++) {
        double *o = A.out + (long long)(orig * nq + q) * A.ld;
#pragma unroll
        for (int s = 0; s < FEAT_MAX_SPECIES; s++)
            if (s < A.ns) {
                o[s * A.npair + e] = scale * acc[s][q];
                if (e == 0) o[A.ns * A.npair + s] = (A.nc == 4) ? 0.0 : A.sigma0 * cnt[s] * A.inv_cnt[p];
            }
    }
}


// ---------------------------------------------------------------- direct variant (prediction path)
// Same Psi rows, computed straight from the descriptor kernel's x / dxdr (rows addressed through the index
// arrays of gpr_nbr_force_rows) without a packed set, plus the prior variance k(x*, x*) of every predicted
// quantity in the convention of the reference's numpy helpers (Dot_mb.py:45-85, 177-258: norm + 1e-8, rows
// never skipped):  force point  diag[3p+q] = sigma^2 |Psi_eps[p, q]|^2  (the identity of the header applied to
// the eps-normalised rows; a~_eps = s a~, A'_eps = s (D/n - s^2 a~ proj), s = n / (n + 1e-8));  energy points
// go through dot2_diag_ee_direct_kernel (their cosine carries a second eps and is not an inner product).
struct dfeat_args {
    const double *x, *dxdr; const int *ele;
    const int *pt_row0, *x_index, *d_index;
    int npts, nc, d, npair, ns;
    int species[FEAT_MAX_SPECIES];
    int nsx, species_x[FEAT_MAX_EXTRA];   // species present in the predicted points but not in the model: they have no
                                          // feature block, but their same-species row pairs belong to the prior variance
    double sigma0, sigma2;
    double *out; long long ld;
    double *diag;
};

constexpr int DF_ROWS = 32;      // rows staged per pass (16 lanes per row: the first 512 threads stage)
constexpr int DF_LD = 33;        // padded row stride of the staged operands (doubles)

// energy points: Phi_i^s = (1/n_i) sum_{rows of species s} vech(a~ a~^T) and the constant column sigma0 n_i^s / n_i;
// one workgroup per point, thread e owns monomial e.  (Force points: dot2_force_features_mfma_kernel below.)
template <int NS, int THREADS>
__global__ void __launch_bounds__(THREADS) dot2_features_direct_kernel(dfeat_args A) {
    __shared__ double sA[DF_ROWS * DF_LD];            // a~ (native norm)
    __shared__ int sE[DF_ROWS];                       // species code, INT_MIN for rows that do not exist or are skipped
    const int p = blockIdx.x, tid = threadIdx.x, e = tid;
    const bool act = e < A.npair;
    int k = 0, m = 0;
    if (act) { while ((k + 1) * (k + 2) / 2 <= e) k++; m = e - k * (k + 1) / 2; }
    const double wgt = (k == m) ? 1.0 : 1.4142135623730951;
    double acc[NS], cnt[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) { cnt[s] = 0.0; acc[s] = 0.0; }
    const int r0 = A.pt_row0[p], r1 = A.pt_row0[p + 1];
    const int slot = tid >> 4, g = tid & 15;          // staging: 16 lanes per row
    for (int rb = r0; rb < r1; rb += DF_ROWS) {
        __syncthreads();
        {
            const int r = rb + slot;
            const bool stager = slot < DF_ROWS;
            const bool have = stager && r < r1;
            const int xs = have ? (A.x_index ? A.x_index[r] : r) : -1;
            const bool ok = have && xs >= 0;
            double x0 = 0.0, x1 = 0.0;
            if (ok) {
                const double *xr = A.x + (size_t)xs * A.d;
                if (g < A.d) x0 = xr[g];
                if (g + 16 < A.d) x1 = xr[g + 16];
            }
            double ss = x0 * x0 + x1 * x1;
#pragma unroll
            for (int o = 8; o >= 1; o >>= 1) ss += __shfl_xor(ss, o, 16);
            const double n = sqrt(ss);
            const bool live = ok && n > 1e-8;          // dot_kernel.cpp:28-44 skip
            const double inv = live ? 1.0 / n : 0.0;
            if (stager) {
                sA[slot * DF_LD + g] = x0 * inv;
                sA[slot * DF_LD + g + 16] = x1 * inv;
                if (g == 0) sE[slot] = live ? A.ele[xs] : INT_MIN;
            }
        }
        __syncthreads();
        if (!act) continue;
        const int nr = min(DF_ROWS, r1 - rb);
        for (int r = 0; r < nr; r++) {
            const int el = sE[r];
            if (el == INT_MIN) continue;
            const double v = wgt * sA[r * DF_LD + k] * sA[r * DF_LD + m];
#pragma unroll
            for (int s = 0; s < NS; s++)
                if (el == A.species[s]) { acc[s] += v; cnt[s] += 1.0; }
        }
    }
    const int npt_rows = r1 - r0;
    const double inv_cnt = npt_rows > 0 ? 1.0 / npt_rows : 0.0;
    if (act) {
        double *o = A.out + (long long)p * A.ld;
#pragma unroll
        for (int s = 0; s < NS; s++) {
            o[s * A.npair + e] = inv_cnt * acc[s];
            if (e == 0) o[A.ns * A.npair + s] = A.sigma0 * cnt[s] * inv_cnt;
        }
    }
}

// prior variance of energy points, numpy-helper convention (Dot_mb.py:177-202): rows of one point against each
// other, same species only, cosine = x.x' / (1e-8 + (|x| + 1e-8)(|x'| + 1e-8)), mean of sigma^2 (cos^2 + sigma0^2)
__device__ __forceinline__ void dot2_diag_ee_body(const dfeat_args &A, int p, double *sRed) {
    const int tid = threadIdx.x;
    const int r0 = A.pt_row0[p], r1 = A.pt_row0[p + 1], n = r1 - r0;
    double part = 0.0;
    for (long long t = tid; t < (long long)n * n; t += blockDim.x) {
        const int i = (int)(t / n), j = (int)(t - (long long)i * n);
        const int xi = A.x_index ? A.x_index[r0 + i] : r0 + i, xj = A.x_index ? A.x_index[r0 + j] : r0 + j;
        if (A.ele[xi] != A.ele[xj]) continue;
        const double *a = A.x + (size_t)xi * A.d, *b = A.x + (size_t)xj * A.d;
        double s = 0.0, na = 0.0, nb = 0.0;
        for (int q = 0; q < A.d; q++) { s += a[q] * b[q]; na += a[q] * a[q]; nb += b[q] * b[q]; }
        const double c = s / (1e-8 + (sqrt(na) + 1e-8) * (sqrt(nb) + 1e-8));
        part += c * c + A.sigma0 * A.sigma0;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) part += __shfl_xor(part, o);
    if ((tid & 63) == 0) sRed[tid >> 6] = part;
    __syncthreads();
    if (tid == 0) {
        const double tot = sRed[0] + sRed[1] + sRed[2] + sRed[3];
        A.diag[p] = n > 0 ? A.sigma2 * tot / ((double)n * (double)n) : 0.0;
    }
}

__global__ void __launch_bounds__(256) dot2_diag_ee_direct_kernel(dfeat_args A) {
    __shared__ double sRed[4];
    dot2_diag_ee_body(A, blockIdx.x, sRed);
}


// ---------------------------------------------------------------- force points on the matrix cores
// The feature rows of a force point are a symmetric rank-2k update: with the rows r of the point,
//     S_q = sum_r (a~_r A'_qr^T + A'_qr a~_r^T) = M_q + M_q^T,   M_q = A~^T A'_q   ([d x R] [R x d], q = x, y, z),
// Psi_F[q] = -vech(S_q) -- an FP64 MFMA product with the rows as the K dimension (four rows per
// v_mfma_f64_16x16x4_f64 step).  One wavefront per point, two passes (native operands for Psi, numpy-eps operands for
// the prior variance sigma^2 |vech(S_eps)|^2 = sigma^2 sum_{k,m} (M_km + M_mk)^2), per species block (rows of other
// species enter as zeros).  Row staging is the one of the VALU kernel above: 16 lanes per row, four rows per step.
typedef double fm_d4 __attribute__((ext_vector_type(4)));
constexpr int FM_LD = 33;                 // padded operand row stride (doubles)

// sum over the 16 lanes of a DPP row (the 16 lanes that stage one descriptor row): four rotate-and-add steps on the
// VALU (row_ror:8,4,2,1), every lane ends with the total; no LDS traffic, unlike a shuffle-based butterfly
template <int CTRL>
__device__ __forceinline__ double fm_row_ror(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double fm_row_sum16(double v) {
    v += fm_row_ror<0x128>(v);
    v += fm_row_ror<0x124>(v);
    v += fm_row_ror<0x122>(v);
    v += fm_row_ror<0x121>(v);
    return v;
}

__device__ __forceinline__ void fm_lds_sync() {
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// One species block of one force point.  P0: native operands -> Psi rows; P1: numpy-eps operands -> prior variance.
// Both products are formed in ONE pass over the rows: what is expensive per step is the staging of four rows (gathered
// loads, the norm and the projection by DPP sums, three LDS round trips), and the two operand sets differ by scale
// factors only (a~_eps = s a~, A'_eps = s (D/n - s^2 a~ proj), s = n / (n + 1e-8)).  The accumulators hold S = M + M^T
// directly -- S_ab += A~_a^T A'_b + A'_a^T A~_b: an LDS operand tile serves as the A or the B operand of the MFMA alike
// -- so a component needs three tiles (S_00, S_01, S_11) instead of the four of M, and both sets fit the register file.
//
// SPLIT: the four wavefronts of the workgroup share ONE point (each takes a quarter of its rows, [r0, r1) is that
// quarter) and their partial S tiles are summed on the way out -- the form for a handful of points (one structure: 32
// force points), where the time of the launch is the time of one point's chain and the chip is empty anyway.
template <bool P0, bool P1, bool SPLIT>
__device__ __forceinline__ void fm_species_block(const dfeat_args &A, int p, int si, int zs, int r0, int r1, double *sm,
                                                 const double *sm_all, const int *km, double (&dacc)[3]) {
    const int lane = threadIdx.x & 63;
    const int slot = lane >> 4, g = lane & 15;
    double *opA0 = sm, *opB0 = sm + 4 * FM_LD, *opA1 = sm + 16 * FM_LD, *opB1 = sm + 20 * FM_LD;   // op?[row][k], opB?[q][row][k]
    fm_d4 acc0[3][3], acc1[3][3];                               // [component q][tile 00, 10, 11] (lower triangle of S)
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
        for (int t = 0; t < 3; t++) { acc0[q][t] = fm_d4{0.0, 0.0, 0.0, 0.0}; acc1[q][t] = fm_d4{0.0, 0.0, 0.0, 0.0}; }
    // Row indices of up to 64 rows at a time sit in registers (lane j: row cb + j) and reach the staging lanes by
    // shuffle; the x / dxdr values of the next four rows are requested before the current four are processed, so
    // no dependent global load sits on the loop's critical path.
    for (int cb = r0; cb < r1; cb += 64) {
        int my_xs = -1, my_ds = -1;
        {
            const int rr = cb + lane;
            if (rr < r1) {
                my_xs = A.x_index ? A.x_index[rr] : rr;
                my_ds = A.d_index ? A.d_index[rr] : rr;
                if (my_xs < 0 || my_ds < 0 || A.ele[my_xs] != zs) { my_xs = -1; my_ds = -1; }   // absent or other species
            }
        }
        const int cend = min(r1, cb + 64);
        double nx0 = 0.0, nx1 = 0.0, ndv[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        bool nok = false;
#define FM_FETCH(RB)                                                                                       \
        {   const int j_ = (RB) - cb + slot;                                                                \
            const int xs_ = __shfl(my_xs, j_ & 63), ds_ = __shfl(my_ds, j_ & 63);                           \
            nok = ((RB) + slot < cend) && xs_ >= 0;                                                         \
            nx0 = 0.0; nx1 = 0.0;                                                                           \
            _Pragma("unroll") for (int u_ = 0; u_ < 6; u_++) ndv[u_] = 0.0;                                 \
            if (nok) {                                                                                      \
                const double *xr_ = A.x + (size_t)xs_ * A.d;                                                \
                if (g < A.d) nx0 = xr_[g];                                                                  \
                if (g + 16 < A.d) nx1 = xr_[g + 16];                                                        \
                const double *D_ = A.dxdr + (size_t)ds_ * A.d * 3;                                          \
                _Pragma("unroll") for (int u_ = 0; u_ < 6; u_++) { const int el_ = g + 16 * u_; if (el_ < 3 * A.d) ndv[u_] = D_[el_]; } } }
        FM_FETCH(cb)
        for (int rb = cb; rb < cend; rb += 4) {
            // ---- stage four rows: 16 lanes per row (values fetched one step ahead)
            const bool ok = nok;
            const double x0 = nx0, x1 = nx1;
            double raw[6];
#pragma unroll
            for (int u = 0; u < 6; u++) raw[u] = ndv[u];
            if (rb + 4 < cend) FM_FETCH(rb + 4)
            const double n = sqrt(fm_row_sum16(x0 * x0 + x1 * x1));
            // native: rows with |x| <= 1e-8 are skipped (dot_kernel.cpp); numpy-eps: norm + 1e-8, nothing skipped
            const double inv = (ok && n > 0.0) ? 1.0 / n : 0.0;
            const double m0 = (ok && n > 1e-8) ? 1.0 : 0.0;                   // native operands: the row counts or not
            const double sc = n / (n + 1e-8), s2 = sc * sc;
            const double a0 = x0 * inv, a1 = x1 * inv;
            // a~ goes to LDS first: the lanes that hold dxdr element (k, c) fetch a~_k from there
            opA1[slot * FM_LD + g] = a0;
            opA1[slot * FM_LD + g + 16] = a1;
            fm_lds_sync();
            double dv[6], ak[6], pr[3] = {0.0, 0.0, 0.0};
#pragma unroll
            for (int u = 0; u < 6; u++) {
                const int el = g + 16 * u, kk = el / 3, c = el - 3 * kk;
                dv[u] = raw[u] * inv;                                        // zero beyond 3 d and for absent rows
                ak[u] = (kk < 32) ? opA1[slot * FM_LD + kk] : 0.0;
                const double t = ak[u] * dv[u];
                pr[0] += (c == 0) ? t : 0.0; pr[1] += (c == 1) ? t : 0.0; pr[2] += (c == 2) ? t : 0.0;
            }
#pragma unroll
            for (int c = 0; c < 3; c++) pr[c] = fm_row_sum16(pr[c]);
            fm_lds_sync();                                                   // every lane has its a~_k: the row may be overwritten
            if (P0) { opA0[slot * FM_LD + g] = m0 * a0; opA0[slot * FM_LD + g + 16] = m0 * a1; }
            if (P1) { opA1[slot * FM_LD + g] = sc * a0; opA1[slot * FM_LD + g + 16] = sc * a1; }
            // A'_c = D/n - a~ proj_c  (native)  /  s (D/n - s^2 a~ proj_c)  (numpy-eps)
#pragma unroll
            for (int u = 0; u < 6; u++) {
                const int el = g + 16 * u, kk = el / 3, c = el - 3 * kk;
                if (kk < 32) {
                    const double prc = (c == 0) ? pr[0] : (c == 1 ? pr[1] : pr[2]);
                    const double ap = ak[u] * prc;
                    if (P0) opB0[(c * 4 + slot) * FM_LD + kk] = m0 * (dv[u] - ap);
                    if (P1) opB1[(c * 4 + slot) * FM_LD + kk] = sc * (dv[u] - s2 * ap);
                }
            }
            fm_lds_sync();
            // ---- S_q += A~^T A'_q + A'_q^T A~ for these four rows (operand tile [k = row][16 columns], either role)
            const int krow = lane >> 4, col = lane & 15;
            if (P0) {
                const double av0 = opA0[krow * FM_LD + col], av1 = opA0[krow * FM_LD + 16 + col];
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    const double b0 = opB0[(q * 4 + krow) * FM_LD + col], b1 = opB0[(q * 4 + krow) * FM_LD + 16 + col];
                    acc0[q][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av0, b0, acc0[q][0], 0, 0, 0);
                    acc0[q][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, av0, acc0[q][0], 0, 0, 0);
                    acc0[q][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av1, b0, acc0[q][1], 0, 0, 0);
                    acc0[q][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, av0, acc0[q][1], 0, 0, 0);
                    acc0[q][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(av1, b1, acc0[q][2], 0, 0, 0);
                    acc0[q][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, av1, acc0[q][2], 0, 0, 0);
                }
            }
            if (P1) {
                const double av0 = opA1[krow * FM_LD + col], av1 = opA1[krow * FM_LD + 16 + col];
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    const double b0 = opB1[(q * 4 + krow) * FM_LD + col], b1 = opB1[(q * 4 + krow) * FM_LD + 16 + col];
                    acc1[q][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av0, b0, acc1[q][0], 0, 0, 0);
                    acc1[q][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, av0, acc1[q][0], 0, 0, 0);
                    acc1[q][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av1, b0, acc1[q][1], 0, 0, 0);
                    acc1[q][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, av0, acc1[q][1], 0, 0, 0);
                    acc1[q][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(av1, b1, acc1[q][2], 0, 0, 0);
                    acc1[q][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, av1, acc1[q][2], 0, 0, 0);
                }
            }
            fm_lds_sync();
        }
#undef FM_FETCH
    }
    // ---- vech(S_q) with sqrt(2) on the off-diagonal entries.  A lane holds S[k][m] for k = tile row base + (lane >> 4) + 4 r,
    // m = tile column base + (lane & 15): the sixteen lanes of a row write sixteen consecutive entries of the packed lower
    // triangle (k (k + 1) / 2 + m), so the three tiles go out straight from the accumulators -- no transpose through LDS, no
    // index table.  The prior variance is the sum over the FULL symmetric matrix: tiles 00 and 11 as they are, tile 10 twice.
    if (!SPLIT) {
        const int rl = lane >> 4, cl = lane & 15;
        if (P0) {
#pragma unroll
            for (int q = 0; q < 3; q++) {
                double *o = A.out + (long long)(p * 3 + q) * A.ld + si * A.npair;
#pragma unroll
                for (int t = 0; t < 3; t++) {
                    const int kb = (t == 0) ? 0 : 16, mb = (t == 2) ? 16 : 0;
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int k = kb + rl + 4 * r, m = mb + cl;
                        const double sv = acc0[q][t][r];
                        if (m <= k && k < A.d) o[k * (k + 1) / 2 + m] = -((k == m) ? sv : 1.4142135623730951 * sv);
                    }
                }
            }
        }
        if (P1) {
#pragma unroll
            for (int q = 0; q < 3; q++) {
                double part = 0.0;
#pragma unroll
                for (int r = 0; r < 4; r++)
                    part += acc1[q][0][r] * acc1[q][0][r] + 2.0 * acc1[q][1][r] * acc1[q][1][r] + acc1[q][2][r] * acc1[q][2][r];
                dacc[q] += part;
            }
        }
        return;
    }
    // SPLIT: the partial tiles of the four wavefronts meet in LDS, entry by entry in packed order
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        if ((pass == 0 && !P0) || (pass == 1 && !P1)) continue;
        for (int q = 0; q < 3; q++) {
#pragma unroll
            for (int t = 0; t < 3; t++) {
                const int ra = (t == 0) ? 0 : 16, cbq = (t == 2) ? 16 : 0;
#pragma unroll
                for (int r = 0; r < 4; r++)
                    sm[(ra + (lane >> 4) + 4 * r) * FM_LD + cbq + (lane & 15)] = (pass == 0) ? acc0[q][t][r] : acc1[q][t][r];
            }
            if (SPLIT) __syncthreads(); else fm_lds_sync();
            double part = 0.0;
            const int t0 = (int)(threadIdx.x >> 6);                                  // SPLIT: the wavefronts share the entries
#pragma unroll
            for (int tt = 0; tt < 9; tt++) {
                if (SPLIT && (tt & 3) != t0) continue;
                const int t = tt;
                const int kmt = km[64 * t + lane];
                if (kmt >= 0) {
                    const int k = kmt >> 8, m = kmt & 255, e = lane + 64 * t;         // m <= k
                    const int at = k * FM_LD + m;
                    double sv;
                    if (SPLIT) sv = (sm_all[at] + sm_all[32 * FM_LD + at]) + (sm_all[2 * 32 * FM_LD + at] + sm_all[3 * 32 * FM_LD + at]);
                    else sv = sm[at];
                    if (pass == 0) A.out[(long long)(p * 3 + q) * A.ld + si * A.npair + e] = -((k == m) ? sv : 1.4142135623730951 * sv);
                    else part += (k == m) ? sv * sv : 2.0 * sv * sv;
                }
            }
            if (pass == 1) dacc[q] += part;
            if (SPLIT) __syncthreads(); else fm_lds_sync();
        }
    }
}

template <bool SPLIT>
__device__ __forceinline__ void dot2_force_features_mfma_body(const dfeat_args &A, int blk, double (*smem)[32 * FM_LD], int (*skm)[9 * 64]) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int p = SPLIT ? blk : blk * 4 + wave;
    if (p >= A.npts) return;                                    // wavefront-uniform (SPLIT: workgroup-uniform, never taken)
    double *sm = smem[wave];
    int r0 = A.pt_row0[p], r1 = A.pt_row0[p + 1];
    if (SPLIT) {                                                // this wavefront's quarter of the rows, whole steps of four
        const int per = ((r1 - r0 + 15) >> 4) << 2;
        r0 = min(r1, r0 + wave * per);
        r1 = min(r1, r0 + per);
    }
    double dacc[3] = {0.0, 0.0, 0.0};                           // sum over species and (k, m) of (M_km + M_mk)^2, eps operands
    // SPLIT: this lane's packed (k, m) indices e = lane + 64 t of the vech layout, decoded once (kept in LDS)
    int *km = skm[wave];
#pragma unroll
    for (int t = 0; t < 9; t++) {
        if (!SPLIT || (t & 3) != wave) continue;                // SPLIT only: the entries this wavefront writes out
        const int e = lane + 64 * t;
        int k = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
        while ((k + 1) * (k + 2) / 2 <= e) k++;
        while (k * (k + 1) / 2 > e) k--;
        km[64 * t + lane] = (e < A.npair) ? ((k << 8) | (e - k * (k + 1) / 2)) : -1;
    }
    if (!SPLIT || wave == 0)
        for (int q = 0; q < 3; q++) {                           // the constant column of force rows is zero
            double *o = A.out + (long long)(p * 3 + q) * A.ld;
            if (lane < A.ns) o[A.ns * A.npair + lane] = 0.0;
        }
    for (int si = 0; si < A.ns; si++) {                         // species of the model: Psi rows (+ their share of the prior variance)
        if (A.diag) fm_species_block<true, true, SPLIT>(A, p, si, A.species[si], r0, r1, sm, smem[0], km, dacc);
        else fm_species_block<true, false, SPLIT>(A, p, si, A.species[si], r0, r1, sm, smem[0], km, dacc);
    }
    if (A.diag)
        for (int sx = 0; sx < A.nsx; sx++)                      // species the model does not know: prior variance only
            fm_species_block<false, true, SPLIT>(A, p, A.ns + sx, A.species_x[sx], r0, r1, sm, smem[0], km, dacc);
    if (A.diag) {
#pragma unroll
        for (int q = 0; q < 3; q++) {
            double t = dacc[q];
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) t += __shfl_xor(t, o);
            if (!SPLIT) { if (lane == 0) A.diag[3 * (size_t)p + q] = A.sigma2 * t; }
            else if (lane == 0) sm[q] = t;                      // the species blocks end on a workgroup barrier: sm is free
        }
        if (SPLIT) {
            __syncthreads();
            if (threadIdx.x < 3) {
                const int q = threadIdx.x;
                A.diag[3 * (size_t)p + q] = A.sigma2 * ((smem[0][q] + smem[1][q]) + (smem[2][q] + smem[3][q]));
            }
        }
    }
}

template <bool SPLIT>
__global__ void __launch_bounds__(256) dot2_force_features_mfma_kernel(dfeat_args A) {
    __shared__ double smem[4][32 * FM_LD];                     // per wavefront: operands of one step / transpose buffer
    __shared__ int skm[4][9 * 64];
    dot2_force_features_mfma_body<SPLIT>(A, blockIdx.x, smem, skm);
}

// energy-point feature rows with 256 threads (thread t owns monomials t, t + 256, t + 512): the block role of the
// fused launch below; same arithmetic as dot2_features_direct_kernel, 16 rows staged per pass
template <int NS>
__device__ __forceinline__ void dot2_features_direct256_body(const dfeat_args &A, int p, double *sA, int *sE) {
    constexpr int ROWS = 16, MPT = 3;
    const int tid = threadIdx.x;
    int kk[MPT], mm[MPT];
    double wgt[MPT];
    bool act[MPT];
#pragma unroll
    for (int u = 0; u < MPT; u++) {
        const int e = tid + 256 * u;
        act[u] = e < A.npair;
        int k = 0;
        if (act[u]) { k = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5); while ((k + 1) * (k + 2) / 2 <= e) k++; while (k * (k + 1) / 2 > e) k--; }
        kk[u] = k; mm[u] = act[u] ? e - k * (k + 1) / 2 : 0;
        wgt[u] = (kk[u] == mm[u]) ? 1.0 : 1.4142135623730951;
    }
    double acc[NS][MPT], cnt[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) { cnt[s] = 0.0; for (int u = 0; u < MPT; u++) acc[s][u] = 0.0; }
    const int r0 = A.pt_row0[p], r1 = A.pt_row0[p + 1];
    const int slot = tid >> 4, g = tid & 15;          // staging: 16 lanes per row, 16 rows per pass
    for (int rb = r0; rb < r1; rb += ROWS) {
        __syncthreads();
        {
            const int r = rb + slot;
            const bool have = r < r1;
            const int xs = have ? (A.x_index ? A.x_index[r] : r) : -1;
            const bool ok = have && xs >= 0;
            double x0 = 0.0, x1 = 0.0;
            if (ok) {
                const double *xr = A.x + (size_t)xs * A.d;
                if (g < A.d) x0 = xr[g];
                if (g + 16 < A.d) x1 = xr[g + 16];
            }
            double ss = x0 * x0 + x1 * x1;
#pragma unroll
            for (int o = 8; o >= 1; o >>= 1) ss += __shfl_xor(ss, o, 16);
            const double n = sqrt(ss);
            const bool live = ok && n > 1e-8;          // dot_kernel.cpp:28-44 skip
            const double inv = live ? 1.0 / n : 0.0;
            sA[slot * DF_LD + g] = x0 * inv;
            sA[slot * DF_LD + g + 16] = x1 * inv;
            if (g == 0) sE[slot] = live ? A.ele[xs] : INT_MIN;
        }
        __syncthreads();
        const int nr = min(ROWS, r1 - rb);
        for (int r = 0; r < nr; r++) {
            const int el = sE[r];
            if (el == INT_MIN) continue;
#pragma unroll
            for (int s = 0; s < NS; s++)
                if (el == A.species[s]) {
                    cnt[s] += 1.0;
#pragma unroll
                    for (int u = 0; u < MPT; u++)
                        if (act[u]) acc[s][u] += wgt[u] * sA[r * DF_LD + kk[u]] * sA[r * DF_LD + mm[u]];
                }
        }
    }
    const int npt_rows = r1 - r0;
    const double inv_cnt = npt_rows > 0 ? 1.0 / npt_rows : 0.0;
    double *o = A.out + (long long)p * A.ld;
#pragma unroll
    for (int s = 0; s < NS; s++) {
#pragma unroll
        for (int u = 0; u < MPT; u++)
            if (act[u]) o[s * A.npair + tid + 256 * u] = inv_cnt * acc[s][u];
        if (tid == 0) o[A.ns * A.npair + s] = A.sigma0 * cnt[s] * inv_cnt;
    }
}

// One launch for all feature rows of a prediction batch: blocks [0, nblkF) take four force points each (matrix
// cores), the next AE.npts blocks one energy point each, the last AE.npts blocks the prior variance of one energy
// point.  The three roles have nothing in common but their inputs; side by side they fill the chip where three
// launches in a row each left most of it idle (64 energy points = 64 workgroups).
template <int NS, bool SPLIT>                                  // SPLIT: a workgroup per force point (few points)
__global__ void __launch_bounds__(256, 2) dot2_features_ef_kernel(dfeat_args AF, dfeat_args AE, int nblkF) {
    __shared__ double smem[4][32 * FM_LD];
    __shared__ int skm[4][9 * 64];
    const int b = blockIdx.x;
    if (b < nblkF) dot2_force_features_mfma_body<SPLIT>(AF, b, smem, skm);
    else if (b < nblkF + AE.npts) dot2_features_direct256_body<NS>(AE, b - nblkF, &smem[0][0], (int *)&smem[1][0]);
    else dot2_diag_ee_body(AE, b - nblkF - AE.npts, &smem[0][0]);
}

}  // namespace

// A workgroup per force point (its rows split over the four wavefronts) when the points are few enough that a
// wavefront per point leaves the chip empty: up to 256 points (a workgroup per CU; equal at 512, slower beyond).  GPR_FEAT_SPLIT=0/1 forces it.
static bool fm_split_points(int npts) {
    static const int force = getenv("GPR_FEAT_SPLIT") ? atoi(getenv("GPR_FEAT_SPLIT")) : -1;
    if (force >= 0) return force != 0;
    return npts <= 256;
}

// Psi rows of the points of a STREAM-packed set: energy set -> out[npts, F]; force set -> out[3*npts, F]
// (row 3*j+q), F = ns * (d(d+1)/2 + 1), in the caller's point order.  species[ns] lists the atomic numbers
// that get a feature block (rows of other species contribute nothing).  Valid for the Dot kernel at zeta = 2.
extern "C" int gpr_dot2_features(void *stream, const gpr_set *set, int ns, const int *species_host, double sigma0,
                                 double *out, long long ld) {
    hipStream_t st = (hipStream_t)stream;
    GPR_REQUIRE(!set || set->ks == GPR_KSTEPS, "gpr_dot2_features: the degree-2 feature map covers descriptors of length <= %d", GPR_KPAD);
    GPR_REQUIRE(set && set->role == GPR_ROLE_STREAM, "gpr_dot2_features: needs a STREAM-packed set");
    GPR_REQUIRE(ns >= 1 && ns <= FEAT_MAX_SPECIES, "gpr_dot2_features: 1..%d species supported (got %d)", FEAT_MAX_SPECIES, ns);
    GPR_REQUIRE(!(set->flags & GPR_SET_NUMPY_EPS), "gpr_dot2_features: native (no-eps) sets only");
    const int npair = set->d * (set->d + 1) / 2, F = ns * (npair + 1);
    GPR_REQUIRE(out && ld >= F, "gpr_dot2_features: output leading dimension %lld < %d", ld, F);
    if (set->npts == 0) return GPR_OK;
    const long long rows = (long long)set->npts * (set->nc == 4 ? 3 : 1);
    { const int zrc_ = gpr_zero_2d(st, out, ld, rows, F); if (zrc_) return zrc_; }   // points without rows
    feat_args A;
    A.ops = set->ops; A.ele = set->ele; A.chunks = set->chunks; A.pt_chunk0 = set->pt_chunk0; A.perm = set->perm;
    A.inv_cnt = set->inv_cnt; A.npts = set->npts; A.nc = set->nc; A.d = set->d; A.npair = npair; A.ns = ns;
    for (int s = 0; s < FEAT_MAX_SPECIES; s++) A.species[s] = s < ns ? species_host[s] : 0;
    A.sigma0 = sigma0; A.out = out; A.ld = ld;
    const int threads = ((npair + 63) / 64) * 64;
    const int pid = gpr_prof_begin(st, GPR_PROF_FEATURES);
    hipLaunchKernelGGL(dot2_features_kernel, dim3(set->npts), dim3(threads), 0, st, A);
    gpr_prof_end(st, pid);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { gpr_set_error("dot2_features launch failed: %s", hipGetErrorString(e)); return GPR_ERR_HIP; }
    return GPR_OK;
}

// Direct variant for prediction: Psi rows and prior variances of points given as row ranges pt_row0[npts+1]
// (device) over index arrays into the descriptor kernel's outputs.  width = 0: energy points (rows r read
// x[x_index[r]]; psi[npts, F], diag[npts]); width = 3: force points (rows read x[x_index[r]], dxdr[dxdr_index[r]];
// psi[3 npts, F], diag[3 npts]).  Index arrays may be NULL (identity); diag may be NULL.  d <= 32.
extern "C" int gpr_dot2_features_direct(void *stream, int d, int width, int npts, const int *pt_row0_dev,
                                        const int *x_index_dev, const int *dxdr_index_dev, const double *x_dev,
                                        const double *dxdr_dev, const int *ele_dev, int ns, const int *species_host,
                                        int nsx, const int *species_extra_host,
                                        double sigma, double sigma0, double *psi, long long ld, double *diag) {
    hipStream_t st = (hipStream_t)stream;
    GPR_REQUIRE(width == 0 || width == 3, "gpr_dot2_features_direct: width must be 0 or 3 (got %d)", width);
    GPR_REQUIRE(d >= 1 && d <= GPR_KPAD, "gpr_dot2_features_direct: d=%d unsupported (1..%d)", d, GPR_KPAD);
    GPR_REQUIRE(ns >= 1 && ns <= FEAT_MAX_SPECIES, "gpr_dot2_features_direct: 1..%d species supported (got %d)", FEAT_MAX_SPECIES, ns);
    GPR_REQUIRE(nsx >= 0 && nsx <= FEAT_MAX_EXTRA && (nsx == 0 || species_extra_host),
                "gpr_dot2_features_direct: 0..%d extra species supported (got %d)", FEAT_MAX_EXTRA, nsx);
    const int npair = d * (d + 1) / 2, F = ns * (npair + 1);
    GPR_REQUIRE(psi && ld >= F, "gpr_dot2_features_direct: output leading dimension %lld < %d", ld, F);
    if (npts <= 0) return GPR_OK;
    GPR_REQUIRE(pt_row0_dev && x_dev && ele_dev && (width == 0 || dxdr_dev), "gpr_dot2_features_direct: null device pointer");
    dfeat_args A;
    A.x = x_dev; A.dxdr = dxdr_dev; A.ele = ele_dev; A.pt_row0 = pt_row0_dev; A.x_index = x_index_dev; A.d_index = dxdr_index_dev;
    A.npts = npts; A.nc = width ? 4 : 1; A.d = d; A.npair = npair; A.ns = ns;
    for (int s = 0; s < FEAT_MAX_SPECIES; s++) A.species[s] = s < ns ? species_host[s] : 0;
    A.nsx = nsx;
    for (int s = 0; s < FEAT_MAX_EXTRA; s++) A.species_x[s] = s < nsx ? species_extra_host[s] : 0;
    A.sigma0 = sigma0; A.sigma2 = sigma * sigma; A.out = psi; A.ld = ld; A.diag = diag;
    const int pid = gpr_prof_begin(st, GPR_PROF_FEATURES);
    if (width == 3) {       // force points: symmetric rank-2k update on the matrix cores, one wavefront per point
        if (fm_split_points(npts)) hipLaunchKernelGGL(dot2_force_features_mfma_kernel<true>, dim3(npts), dim3(256), 0, st, A);
        else hipLaunchKernelGGL(dot2_force_features_mfma_kernel<false>, dim3((npts + 3) / 4), dim3(256), 0, st, A);
        gpr_prof_end(st, pid);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { gpr_set_error("dot2_force_features_mfma launch failed: %s", hipGetErrorString(e)); return GPR_ERR_HIP; }
        return GPR_OK;
    }
    const bool wide = npair > 512;
#define GPR_DF_LAUNCH(NSV)                                                                                         \
    do {                                                                                                           \
        if (wide) hipLaunchKernelGGL((dot2_features_direct_kernel<NSV, 576>), dim3(npts), dim3(576), 0, st, A);   \
        else hipLaunchKernelGGL((dot2_features_direct_kernel<NSV, 512>), dim3(npts), dim3(512), 0, st, A);        \
    } while (0)
    switch (ns) {
        case 1: GPR_DF_LAUNCH(1); break;
        case 2: GPR_DF_LAUNCH(2); break;
        case 3: GPR_DF_LAUNCH(3); break;
        default: GPR_DF_LAUNCH(4); break;
    }
#undef GPR_DF_LAUNCH
    if (width == 0 && diag) hipLaunchKernelGGL(dot2_diag_ee_direct_kernel, dim3(npts), dim3(256), 0, st, A);
    gpr_prof_end(st, pid);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { gpr_set_error("dot2_features_direct launch failed: %s", hipGetErrorString(e)); return GPR_ERR_HIP; }
    return GPR_OK;
}

// Energy and force feature rows (and prior variances) of a prediction batch in one launch: psi = [nE energy rows |
// 3 nF force rows], diag likewise (may be NULL).  Energy point s = the x rows [e_row0[s], e_row0[s+1]); force point a =
// rows f_row0[a] .. f_row0[a+1] of the index arrays (gpr_nbr_force_rows).  Same arithmetic as two calls of
// gpr_dot2_features_direct (width 0, then width 3).
extern "C" int gpr_dot2_features_ef(void *stream, int d, int nE, const int *e_row0_dev, int nF, const int *f_row0_dev,
                                    const int *x_index_dev, const int *dxdr_index_dev, const double *x_dev,
                                    const double *dxdr_dev, const int *ele_dev, int ns, const int *species_host,
                                    int nsx, const int *species_extra_host, double sigma, double sigma0, double *psi,
                                    long long ld, double *diag) {
    hipStream_t st = (hipStream_t)stream;
    GPR_REQUIRE(d >= 1 && d <= GPR_KPAD, "gpr_dot2_features_ef: d=%d unsupported (1..%d)", d, GPR_KPAD);
    GPR_REQUIRE(ns >= 1 && ns <= FEAT_MAX_SPECIES, "gpr_dot2_features_ef: 1..%d species supported (got %d)", FEAT_MAX_SPECIES, ns);
    GPR_REQUIRE(nsx >= 0 && nsx <= FEAT_MAX_EXTRA && (nsx == 0 || species_extra_host),
                "gpr_dot2_features_ef: 0..%d extra species supported (got %d)", FEAT_MAX_EXTRA, nsx);
    const int npair = d * (d + 1) / 2, F = ns * (npair + 1);
    GPR_REQUIRE(psi && ld >= F, "gpr_dot2_features_ef: output leading dimension %lld < %d", ld, F);
    GPR_REQUIRE(nE >= 0 && nF >= 0, "gpr_dot2_features_ef: negative sizes");
    if (nE == 0 && nF == 0) return GPR_OK;
    GPR_REQUIRE(x_dev && ele_dev && (nE == 0 || e_row0_dev) && (nF == 0 || (f_row0_dev && dxdr_dev)),
                "gpr_dot2_features_ef: null device pointer");
    dfeat_args AE, AF;
    AE.x = x_dev; AE.dxdr = nullptr; AE.ele = ele_dev; AE.pt_row0 = e_row0_dev; AE.x_index = nullptr; AE.d_index = nullptr;
    AE.npts = nE; AE.nc = 1; AE.d = d; AE.npair = npair; AE.ns = ns;
    for (int s = 0; s < FEAT_MAX_SPECIES; s++) AE.species[s] = s < ns ? species_host[s] : 0;
    AE.nsx = nsx;
    for (int s = 0; s < FEAT_MAX_EXTRA; s++) AE.species_x[s] = s < nsx ? species_extra_host[s] : 0;
    AE.sigma0 = sigma0; AE.sigma2 = sigma * sigma; AE.out = psi; AE.ld = ld; AE.diag = diag;
    AF = AE;
    AF.dxdr = dxdr_dev; AF.pt_row0 = f_row0_dev; AF.x_index = x_index_dev; AF.d_index = dxdr_index_dev;
    AF.npts = nF; AF.nc = 4; AF.sigma0 = 0.0; AF.out = psi + (size_t)nE * ld; AF.diag = diag ? diag + nE : nullptr;
    const int split = fm_split_points(nF) ? 1 : 0;
    const int nblkF = split ? nF : (nF + 3) / 4, nblk = nblkF + nE * (diag ? 2 : 1);
    const int pid = gpr_prof_begin(st, GPR_PROF_FEATURES);
#define GPR_EF_LAUNCH(NSV)                                                                                         \
    do {                                                                                                           \
        if (split) hipLaunchKernelGGL((dot2_features_ef_kernel<NSV, true>), dim3(nblk), dim3(256), 0, st, AF, AE, nblkF);   \
        else hipLaunchKernelGGL((dot2_features_ef_kernel<NSV, false>), dim3(nblk), dim3(256), 0, st, AF, AE, nblkF);      \
    } while (0)
    switch (ns) {
        case 1: GPR_EF_LAUNCH(1); break;
        case 2: GPR_EF_LAUNCH(2); break;
        case 3: GPR_EF_LAUNCH(3); break;
        default: GPR_EF_LAUNCH(4); break;
    }
#undef GPR_EF_LAUNCH
    gpr_prof_end(st, pid);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { gpr_set_error("dot2_features_ef launch failed: %s", hipGetErrorString(e)); return GPR_ERR_HIP; }
    return GPR_OK;
}
