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
#include "gpr_internal.h"

namespace {

constexpr int FEAT_MAX_SPECIES = 4;

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

}  // namespace

// Psi rows of the points of a STREAM-packed set: energy set -> out[npts, F]; force set -> out[3*npts, F]
// (row 3*j+q), F = ns * (d(d+1)/2 + 1), in the caller's point order.  species[ns] lists the atomic numbers
// that get a feature block (rows of other species contribute nothing).  Valid for the Dot kernel at zeta = 2.
extern "C" int gpr_dot2_features(void *stream, const gpr_set *set, int ns, const int *species_host, double sigma0,
                                 double *out, long long ld) {
    hipStream_t st = (hipStream_t)stream;
    GPR_REQUIRE(set && set->role == GPR_ROLE_STREAM, "gpr_dot2_features: needs a STREAM-packed set");
    GPR_REQUIRE(ns >= 1 && ns <= FEAT_MAX_SPECIES, "gpr_dot2_features: 1..%d species supported (got %d)", FEAT_MAX_SPECIES, ns);
    GPR_REQUIRE(!(set->flags & GPR_SET_NUMPY_EPS), "gpr_dot2_features: native (no-eps) sets only");
    const int npair = set->d * (set->d + 1) / 2, F = ns * (npair + 1);
    GPR_REQUIRE(out && ld >= F, "gpr_dot2_features: output leading dimension %lld < %d", ld, F);
    if (set->npts == 0) return GPR_OK;
    const long long rows = (long long)set->npts * (set->nc == 4 ? 3 : 1);
    GPR_HIP_TRY(hipMemset2DAsync(out, (size_t)ld * 8, 0, (size_t)F * 8, (size_t)rows, st));   // points without rows
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
