// capi.hip -- L1 covariance-block entry points (wrapper-level semantics of kee_C / kef_C / kff_C,
// gpr_calc/kernels/rbf_kernel.py:7-337 and dot_kernel.py:9-270) on top of gpr_launch_block.
#include "gpr_internal.h"

static int zero_block(hipStream_t st, double *p, long long ld, long long rows, long long cols) {
    if (rows <= 0 || cols <= 0) return GPR_OK;
    GPR_REQUIRE(p != nullptr, "block: null output pointer");
    GPR_REQUIRE(ld >= cols, "block: leading dimension %lld < %lld columns", ld, cols);
    return gpr_zero_2d(st, p, ld, rows, cols);
}

static void fill_kernel(gpr_block_call &c, const gpr_kernel_params *kp) {
    c.kind = kp->kind; c.zeta = kp->zeta; c.sigma2 = kp->sigma * kp->sigma; c.l = kp->l; c.l2 = kp->l * kp->l;
    c.sigma02 = kp->sigma0 * kp->sigma0; c.tol = kp->tol;
    c.symmetric = 0; c.diag_only = 0; c.eps_mode = 0; c.grad = 0;
    c.scale = 1.0; c.dl_scale = 1.0; c.norm1 = 0; c.norm2 = 0;
    c.dl = nullptr; c.d_i = c.d_j = c.d_a = c.d_b = 0;
    c.s_i = c.s_j = c.s_a = c.s_b = 0;
}

static int check_kp(const gpr_kernel_params *kp, const double *dl_out) {
    GPR_REQUIRE(kp != nullptr, "block: null kernel params");
    GPR_REQUIRE(kp->kind == GPR_KERNEL_RBF || kp->kind == GPR_KERNEL_DOT, "block: unknown kernel kind %d", kp->kind);
    GPR_REQUIRE(kp->kind != GPR_KERNEL_RBF || kp->l > 0.0, "block: RBF needs l > 0");
    GPR_REQUIRE(!(dl_out && kp->kind != GPR_KERNEL_RBF), "block: dl_out is defined for the RBF kernel only");
    return GPR_OK;
}

extern "C" int gpr_block_ee(void *stream, const gpr_kernel_params *kp, const gpr_set *s1,
                            const gpr_set *s2, int flags, double *out, long long ld,
                            double *dl_out, long long dl_ld) {
    hipStream_t st = (hipStream_t)stream;
    int rc = check_kp(kp, dl_out); if (rc) return rc;
    GPR_REQUIRE(s1 && s2, "gpr_block_ee: null set");
    const bool tr = (flags & GPR_BLOCK_TRANSPOSE) != 0;
    const long long rows = tr ? s2->npts : s1->npts, cols = tr ? s1->npts : s2->npts;
    if ((rc = zero_block(st, out, ld, rows, cols))) return rc;
    if (dl_out && (rc = zero_block(st, dl_out, dl_ld, rows, cols))) return rc;
    gpr_block_call c; fill_kernel(c, kp);
    c.mode = GPR_MODE_EE; c.s1 = s1; c.s2 = s2; c.symmetric = (flags & GPR_BLOCK_SYMMETRIC) ? 1 : 0;
    c.norm1 = 1; c.norm2 = 1;                                    // C /= n_i n_j (rbf_kernel.py:70)
    c.scale = (kp->kind == GPR_KERNEL_DOT) ? c.sigma2 : 1.0;     // dot_kernel.cpp:49
    c.out = out; c.s_i = tr ? 1 : ld; c.s_j = tr ? ld : 1;
    if (dl_out) {                                                // C_l *= 1/l^3 (rbf_kernel.py:47,60)
        c.grad = 1; c.dl = dl_out; c.d_i = tr ? 1 : dl_ld; c.d_j = tr ? dl_ld : 1;
        c.dl_scale = 1.0 / (kp->l * kp->l * kp->l);
    }
    return gpr_launch_block(st, c);
}

extern "C" int gpr_block_ef(void *stream, const gpr_kernel_params *kp, const gpr_set *s1,
                            const gpr_set *s2, int transpose, double *out, long long ld,
                            double *dl_out, long long dl_ld) {
    hipStream_t st = (hipStream_t)stream;
    int rc = check_kp(kp, dl_out); if (rc) return rc;
    GPR_REQUIRE(s1 && s2, "gpr_block_ef: null set");
    const long long rows = transpose ? 3LL * s2->npts : s1->npts;
    const long long cols = transpose ? s1->npts : 3LL * s2->npts;
    if ((rc = zero_block(st, out, ld, rows, cols))) return rc;
    if (dl_out && (rc = zero_block(st, dl_out, dl_ld, rows, cols))) return rc;
    gpr_block_call c; fill_kernel(c, kp);
    c.mode = GPR_MODE_EF; c.s1 = s1; c.s2 = s2;
    c.norm1 = 1;                                                 // out /= n_i (rbf_kernel.py:161)
    c.scale = (kp->kind == GPR_KERNEL_DOT) ? -c.sigma2 : 1.0;    // dot_kernel.py:131
    c.out = out;
    if (transpose) { c.s_i = 1; c.s_j = 3 * ld; c.s_b = ld; }
    else           { c.s_i = ld; c.s_j = 3; c.s_b = 1; }
    if (dl_out) {
        c.grad = 1; c.dl = dl_out;
        if (transpose) { c.d_i = 1; c.d_j = 3 * dl_ld; c.d_b = dl_ld; }
        else           { c.d_i = dl_ld; c.d_j = 3; c.d_b = 1; }
    }
    return gpr_launch_block(st, c);
}

extern "C" int gpr_block_ff(void *stream, const gpr_kernel_params *kp, const gpr_set *s1,
                            const gpr_set *s2, int flags, double *out, long long ld,
                            double *dl_out, long long dl_ld) {
    hipStream_t st = (hipStream_t)stream;
    int rc = check_kp(kp, dl_out); if (rc) return rc;
    GPR_REQUIRE(s1 && s2, "gpr_block_ff: null set");
    const bool tr = (flags & GPR_BLOCK_TRANSPOSE) != 0;
    const long long rows = 3LL * (tr ? s2->npts : s1->npts), cols = 3LL * (tr ? s1->npts : s2->npts);
    if ((rc = zero_block(st, out, ld, rows, cols))) return rc;
    if (dl_out && (rc = zero_block(st, dl_out, dl_ld, rows, cols))) return rc;
    gpr_block_call c; fill_kernel(c, kp);
    c.mode = GPR_MODE_FF; c.s1 = s1; c.s2 = s2; c.symmetric = (flags & GPR_BLOCK_SYMMETRIC) ? 1 : 0;
    c.scale = (kp->kind == GPR_KERNEL_DOT) ? c.sigma2 * kp->zeta : 1.0;   // dot_kernel.py:256
    c.out = out;
    if (tr) { c.s_i = 3; c.s_a = 1; c.s_j = 3 * ld; c.s_b = ld; }
    else    { c.s_i = 3 * ld; c.s_a = ld; c.s_j = 3; c.s_b = 1; }
    if (dl_out) {
        c.grad = 1; c.dl = dl_out;
        if (tr) { c.d_i = 3; c.d_a = 1; c.d_j = 3 * dl_ld; c.d_b = dl_ld; }
        else    { c.d_i = 3 * dl_ld; c.d_a = dl_ld; c.d_j = 3; c.d_b = 1; }
        c.tol = -1.0;   // the gradient native has no skip (rbf_kernel.cpp:534-536)
    }
    return gpr_launch_block(st, c);
}

extern "C" int gpr_diag_ff(void *stream, const gpr_kernel_params *kp, const gpr_set *s1,
                           const gpr_set *s2, double *out) {
    hipStream_t st = (hipStream_t)stream;
    int rc = check_kp(kp, nullptr); if (rc) return rc;
    GPR_REQUIRE(s1 && s2, "gpr_diag_ff: null set");
    if ((rc = zero_block(st, out, 3LL * s2->npts, 1, 3LL * s2->npts))) return rc;
    gpr_block_call c; fill_kernel(c, kp);
    c.mode = GPR_MODE_FF; c.s1 = s1; c.s2 = s2; c.diag_only = 1;
    c.scale = (kp->kind == GPR_KERNEL_DOT) ? c.sigma2 * kp->zeta : 1.0;
    c.out = out; c.s_j = 3; c.s_b = 1;
    return gpr_launch_block(st, c);
}

extern "C" int gpr_diag_ee(void *stream, const gpr_kernel_params *kp, const gpr_set *s1,
                           const gpr_set *s2, double *out) {
    hipStream_t st = (hipStream_t)stream;
    int rc = check_kp(kp, nullptr); if (rc) return rc;
    GPR_REQUIRE(s1 && s2, "gpr_diag_ee: null set");
    if ((rc = zero_block(st, out, s2->npts, 1, s2->npts))) return rc;
    gpr_block_call c; fill_kernel(c, kp);
    c.mode = GPR_MODE_EE; c.s1 = s1; c.s2 = s2; c.diag_only = 1;
    c.norm1 = 1; c.norm2 = 1;
    c.scale = (kp->kind == GPR_KERNEL_DOT) ? c.sigma2 : 1.0;
    // the reference evaluates the energy diagonal with the numpy helpers, whose cosine carries an
    // extra eps in the denominator (base.py:121, Dot_mb.py:191); sets must be packed NUMPY_EPS
    c.eps_mode = ((s1->flags & GPR_SET_NUMPY_EPS) && (s2->flags & GPR_SET_NUMPY_EPS)) ? 1 : 0;
    c.out = out; c.s_j = 1; c.s_b = 0;
    return gpr_launch_block(st, c);
}
