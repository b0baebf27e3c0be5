// legacy_abi.hip -- L0: the 13 natives of the reference (gpr_calc/kernels/rbf_kernel.h:4-38,
// dot_kernel.h:4-27) with their exact signatures and semantics, executed on the GPU.
// Host pointers in/out; results are ACCUMULATED into the caller's (zeroed) buffers exactly as
// the reference does.  No CPU fallback: a call that cannot be executed on the device (an unsupported shape such as a
// descriptor longer than this build packs, a HIP error) prints the reason to stderr and RETURNS with the caller's
// output buffers untouched -- the natives are `void` (rbf_kernel.h:4-38), so the status is also kept for
// gpr_l0_last_status(); a library entry never takes the host process down.
#include <algorithm>
#include <numeric>
#include "gpr_internal.h"

namespace {

struct l0_failure { int rc; };
thread_local int g_l0_status = GPR_OK;

[[noreturn]] void die(const char *who, int rc) {
    fprintf(stderr, "libgprhip: %s failed (status %d): %s; the output buffers are left untouched\n", who, rc, gpr_last_error());
    fflush(stderr);
    g_l0_status = rc;
    throw l0_failure{rc};
}

#define L0_TRY(expr) do { int _rc = (expr); if (_rc) die(who, _rc); } while (0)
#define L0_HIP(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) {                          \
        gpr_set_error("%s: %s", #expr, hipGetErrorString(_e)); die(who, GPR_ERR_HIP); } } while (0)

// One side of a native call: rows [r0, r1) of (x, dxdr, ele, inds) uploaded and packed.
struct Side {
    double *x = nullptr, *dx = nullptr;
    int *ele = nullptr;
    std::vector<int> counts;     // rows per point id (length = number of point ids)
    std::vector<gpr_set *> sets; // one per derivative triplet
    int nrows = 0;
    ~Side() {                    // also on the unwinding path of a failed call
        for (gpr_set *p : sets) gpr_set_destroy(p, nullptr);
        if (x) (void)hipFree(x);
        if (dx) (void)hipFree(dx);
        if (ele) (void)hipFree(ele);
    }
};

// inds need not be sorted in principle (the reference only ever passes sorted ones,
// rbf_kernel.py:32-36); rows are gathered into point order on the host when they are not.
void build_side(const char *who, Side &s, int role, int d, int width, int r0, int r1, int npts_hint,
                const double *x, const double *dx, const int *ele, const int *inds) {
    const int n = std::max(0, r1 - r0);
    s.nrows = n;
    int npts = npts_hint;
    for (int r = r0; r < r1; r++) {
        if (inds[r] < 0) { gpr_set_error("negative point index"); die(who, GPR_ERR_ARG); }
        npts = std::max(npts, inds[r] + 1);
    }
    s.counts.assign(npts, 0);
    bool sorted = true;
    for (int r = r0; r < r1; r++) {
        s.counts[inds[r]]++;
        if (r > r0 && inds[r] < inds[r - 1]) sorted = false;
    }
    std::vector<double> gx, gdx;
    std::vector<int> gele;
    const double *px = x + (size_t)r0 * d;
    const double *pdx = dx ? dx + (size_t)r0 * d * width : nullptr;
    const int *pele = ele + r0;
    if (!sorted) {
        std::vector<int> order(n);
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return inds[r0 + a] < inds[r0 + b]; });
        gx.resize((size_t)n * d); gele.resize(n);
        if (dx) gdx.resize((size_t)n * d * width);
        for (int k = 0; k < n; k++) {
            int src = order[k];
            std::copy(px + (size_t)src * d, px + (size_t)(src + 1) * d, gx.begin() + (size_t)k * d);
            if (dx) std::copy(pdx + (size_t)src * d * width, pdx + (size_t)(src + 1) * d * width,
                              gdx.begin() + (size_t)k * d * width);
            gele[k] = pele[src];
        }
        px = gx.data(); pdx = dx ? gdx.data() : nullptr; pele = gele.data();
    }
    if (n > 0) {
        L0_HIP(hipMalloc(&s.x, (size_t)n * d * 8));
        L0_HIP(hipMemcpy(s.x, px, (size_t)n * d * 8, hipMemcpyHostToDevice));
        L0_HIP(hipMalloc(&s.ele, (size_t)n * 4));
        L0_HIP(hipMemcpy(s.ele, pele, (size_t)n * 4, hipMemcpyHostToDevice));
        if (dx) {
            L0_HIP(hipMalloc(&s.dx, (size_t)n * d * width * 8));
            L0_HIP(hipMemcpy(s.dx, pdx, (size_t)n * d * width * 8, hipMemcpyHostToDevice));
        }
    }
    const int ntrip = dx ? width / 3 : 1;
    for (int t = 0; t < ntrip; t++) {
        gpr_set *set = nullptr;
        L0_TRY(gpr_set_create(&set, nullptr, role, d, dx ? width : 0, 3 * t, n, npts, s.counts.data(),
                              s.x, s.dx, s.ele, 0));
        s.sets.push_back(set);
    }
}

void free_side(Side &) {}       // the destructor releases everything

struct DevOut {
    double *d = nullptr;
    size_t n = 0;
    ~DevOut() { if (d) (void)hipFree(d); }
    void alloc(const char *who, size_t count) {
        n = count;
        if (n == 0) return;
        L0_HIP(hipMalloc(&d, n * 8));
        L0_HIP(hipMemset(d, 0, n * 8));
    }
    void add_into(const char *who, double *host) {
        if (n == 0) return;
        std::vector<double> tmp(n);
        L0_HIP(hipDeviceSynchronize());
        L0_HIP(hipMemcpy(tmp.data(), d, n * 8, hipMemcpyDeviceToHost));
        for (size_t k = 0; k < n; k++) host[k] += tmp[k];
        (void)hipFree(d);
        d = nullptr;
    }
};

// lpar is l when `grad` (the gradient natives take l) and l2 otherwise (rbf_kernel.h)
gpr_block_call base_call(int mode, int kind, double zeta, double sigma2, double lpar, bool lpar_is_l, double sigma02, double tol) {
    gpr_block_call c;
    c.mode = mode; c.kind = kind; c.zeta = zeta; c.sigma2 = sigma2; c.sigma02 = sigma02; c.tol = tol;
    if (lpar_is_l) { c.l = lpar; c.l2 = lpar * lpar; } else { c.l2 = lpar; c.l = sqrt(lpar); }
    c.grad = 0; c.symmetric = 0; c.diag_only = 0; c.eps_mode = 0; c.norm1 = 0; c.norm2 = 0;
    c.scale = 1.0; c.dl_scale = 1.0;
    c.out = nullptr; c.s_i = c.s_j = c.s_a = c.s_b = 0;
    c.dl = nullptr; c.d_i = c.d_j = c.d_a = c.d_b = 0;
    return c;
}

// ---- energy-energy
void ee_native(const char *who, int kind, bool grad, int n1, int n2, int d, int x2i, double zeta,
               double sigma2, double l2_or_s02, double *x1, int *ele1, int *i1, double *x2, int *ele2,
               int *i2, double *pout, double *dpout) {
    g_l0_status = GPR_OK;
    try {
        if (n1 <= 0 || n2 <= 0) return;
        Side a, b;
        build_side(who, a, GPR_ROLE_STREAM, d, 0, 0, n1, 0, x1, nullptr, ele1, i1);
        build_side(who, b, GPR_ROLE_RESIDENT, d, 0, 0, n2, x2i, x2, nullptr, ele2, i2);
        if ((int)b.counts.size() != x2i) { gpr_set_error("x2_inds exceed x2i"); die(who, GPR_ERR_ARG); }
        const size_t m1 = a.counts.size();
        DevOut o, o2;
        o.alloc(who, m1 * x2i);
        if (grad) o2.alloc(who, m1 * x2i);
        gpr_block_call c = base_call(GPR_MODE_EE, kind, zeta, sigma2,
                                     kind == GPR_KERNEL_RBF ? l2_or_s02 : 1.0, false,
                                     kind == GPR_KERNEL_DOT ? l2_or_s02 : 0.0, 0.0);
        c.s1 = a.sets[0]; c.s2 = b.sets[0];
        c.scale = (kind == GPR_KERNEL_DOT) ? sigma2 : 1.0;
        c.out = o.d; c.s_i = x2i; c.s_j = 1;
        if (grad) { c.grad = 1; c.dl = o2.d; c.d_i = x2i; c.d_j = 1; }
        L0_TRY(gpr_launch_block(nullptr, c));
        o.add_into(who, pout);
        if (grad) o2.add_into(who, dpout);
        free_side(a); free_side(b);
    } catch (const l0_failure &) {}      // reported by die(); outputs untouched
}

// ---- energy-force (w = 3 or 9 derivative columns on set 2)
void ef_native(const char *who, int kind, bool grad, int w, int n1, int n2, int d, int x2i, double zeta,
               double sigma2, double lpar, double *x1, int *ele1, int *i1, double *x2, double *dx2dr,
               int *ele2, int *i2, double *pout) {
    g_l0_status = GPR_OK;
    try {
        if (n1 <= 0 || n2 <= 0) return;
        Side a, b;
        build_side(who, a, GPR_ROLE_STREAM, d, 0, 0, n1, 0, x1, nullptr, ele1, i1);
        build_side(who, b, GPR_ROLE_RESIDENT, d, w, 0, n2, x2i, x2, dx2dr, ele2, i2);
        if ((int)b.counts.size() != x2i) { gpr_set_error("x2_inds exceed x2i"); die(who, GPR_ERR_ARG); }
        const size_t m1 = a.counts.size();
        const int ow = grad ? 6 : w;
        DevOut o;
        o.alloc(who, m1 * x2i * ow);
        for (size_t t = 0; t < b.sets.size(); t++) {
            gpr_block_call c = base_call(GPR_MODE_EF, kind, zeta, sigma2, lpar, grad, 0.0, 0.0);
            c.s1 = a.sets[0]; c.s2 = b.sets[t];
            c.out = o.d + 3 * t; c.s_i = (long long)x2i * ow; c.s_j = ow; c.s_b = 1;
            if (grad) { c.grad = 1; c.dl = o.d + 3; c.d_i = c.s_i; c.d_j = ow; c.d_b = 1; }
            L0_TRY(gpr_launch_block(nullptr, c));
        }
        o.add_into(who, pout);
        free_side(a); free_side(b);
    } catch (const l0_failure &) {}      // reported by die(); outputs untouched
}

// ---- force-force (w1 = 3 or 9 derivative columns on set 1)
void ff_native(const char *who, int kind, bool grad, int w1, int n1, int n2, int n2s, int n2e, int d,
               int x2i, double zeta, double sigma2, double lpar, double tol, double *x1, double *dx1dr,
               int *ele1, int *i1, double *x2, double *dx2dr, int *ele2, int *i2, double *pout,
               double *dpout) {
    g_l0_status = GPR_OK;
    try {
        n2s = std::max(n2s, 0); n2e = std::min(n2e, n2);
        if (n1 <= 0 || n2e <= n2s) return;
        Side a, b;
        build_side(who, a, GPR_ROLE_STREAM, d, w1, 0, n1, 0, x1, dx1dr, ele1, i1);
        build_side(who, b, GPR_ROLE_RESIDENT, d, 3, n2s, n2e, x2i, x2, dx2dr, ele2, i2);
        if ((int)b.counts.size() != x2i) { gpr_set_error("x2_inds exceed x2i"); die(who, GPR_ERR_ARG); }
        const size_t m1 = a.counts.size();
        DevOut o, o2;
        o.alloc(who, m1 * w1 * x2i * 3);
        if (grad) o2.alloc(who, m1 * w1 * x2i * 3);
        for (size_t t = 0; t < a.sets.size(); t++) {
            gpr_block_call c = base_call(GPR_MODE_FF, kind, zeta, sigma2, lpar, grad, 0.0, grad ? -1.0 : tol);
            c.s1 = a.sets[t]; c.s2 = b.sets[0];
            // pout[((q + i*w1)*x2i + j)*3 + b], q = 3t + a
            c.out = o.d + (size_t)3 * t * x2i * 3;
            c.s_i = (long long)w1 * x2i * 3; c.s_a = (long long)x2i * 3; c.s_j = 3; c.s_b = 1;
            if (grad) { c.grad = 1; c.dl = o2.d; c.d_i = c.s_i; c.d_a = c.s_a; c.d_j = 3; c.d_b = 1; }
            L0_TRY(gpr_launch_block(nullptr, c));
        }
        o.add_into(who, pout);
        if (grad) o2.add_into(who, dpout);
        free_side(a); free_side(b);
    } catch (const l0_failure &) {}      // reported by die(); outputs untouched
}

}  // namespace

extern "C" {

void rbf_kee_many(int n1, int n2, int d, int x2i, double zeta, double sigma2, double l2, double *x1,
                  int *ele1, int *x1_inds, double *x2, int *ele2, int *x2_inds, double *pout) {
    ee_native("rbf_kee_many", GPR_KERNEL_RBF, false, n1, n2, d, x2i, zeta, sigma2, l2, x1, ele1, x1_inds,
              x2, ele2, x2_inds, pout, nullptr);
}
void rbf_kee_many_with_grad(int n1, int n2, int d, int x2i, double zeta, double sigma2, double l2,
                            double *x1, int *ele1, int *x1_inds, double *x2, int *ele2, int *x2_inds,
                            double *pout, double *dpout_dl) {
    ee_native("rbf_kee_many_with_grad", GPR_KERNEL_RBF, true, n1, n2, d, x2i, zeta, sigma2, l2, x1, ele1,
              x1_inds, x2, ele2, x2_inds, pout, dpout_dl);
}
void dot_kee_many(int n1, int n2, int d, int x2i, double zeta, double sigma2, double sigma02, double *x1,
                  int *ele1, int *x1_inds, double *x2, int *ele2, int *x2_inds, double *pout) {
    ee_native("dot_kee_many", GPR_KERNEL_DOT, false, n1, n2, d, x2i, zeta, sigma2, sigma02, x1, ele1,
              x1_inds, x2, ele2, x2_inds, pout, nullptr);
}

void rbf_kef_many(int n1, int n2, int d, int x2i, double zeta, double sigma2, double l2, double *x1,
                  int *ele1, int *x1_inds, double *x2, double *dx2dr, int *ele2, int *x2_inds,
                  double *pout) {
    ef_native("rbf_kef_many", GPR_KERNEL_RBF, false, 3, n1, n2, d, x2i, zeta, sigma2, l2, x1, ele1,
              x1_inds, x2, dx2dr, ele2, x2_inds, pout);
}
void rbf_kef_many_with_grad(int n1, int n2, int d, int x2i, double zeta, double sigma2, double l,
                            double *x1, int *ele1, int *x1_inds, double *x2, double *dx2dr, int *ele2,
                            int *x2_inds, double *pout) {
    ef_native("rbf_kef_many_with_grad", GPR_KERNEL_RBF, true, 3, n1, n2, d, x2i, zeta, sigma2, l, x1, ele1,
              x1_inds, x2, dx2dr, ele2, x2_inds, pout);
}
void rbf_kef_many_stress(int n1, int n2, int d, int x2i, double zeta, double sigma2, double l2, double *x1,
                         int *ele1, int *x1_inds, double *x2, double *dx2dr, int *ele2, int *x2_inds,
                         double *pout) {
    ef_native("rbf_kef_many_stress", GPR_KERNEL_RBF, false, 9, n1, n2, d, x2i, zeta, sigma2, l2, x1,
              ele1, x1_inds, x2, dx2dr, ele2, x2_inds, pout);
}
void dot_kef_many(int n1, int n2, int d, int x2i, double zeta, double *x1, int *ele1, int *x1_inds,
                  double *x2, double *dx2dr, int *ele2, int *x2_inds, double *pout) {
    ef_native("dot_kef_many", GPR_KERNEL_DOT, false, 3, n1, n2, d, x2i, zeta, 1.0, 1.0, x1, ele1, x1_inds,
              x2, dx2dr, ele2, x2_inds, pout);
}
void dot_kef_many_stress(int n1, int n2, int d, int x2i, double zeta, double *x1, int *ele1, int *x1_inds,
                         double *x2, double *dx2dr, int *ele2, int *x2_inds, double *pout) {
    ef_native("dot_kef_many_stress", GPR_KERNEL_DOT, false, 9, n1, n2, d, x2i, zeta, 1.0, 1.0, x1, ele1,
              x1_inds, x2, dx2dr, ele2, x2_inds, pout);
}

void rbf_kff_many(int n1, int n2, int n2_start, int n2_end, int d, int x2i, double zeta, double sigma2,
                  double l2, double tol, double *x1, double *dx1dr, int *ele1, int *x1_inds, double *x2,
                  double *dx2dr, int *ele2, int *x2_inds, double *pout) {
    ff_native("rbf_kff_many", GPR_KERNEL_RBF, false, 3, n1, n2, n2_start, n2_end, d, x2i, zeta, sigma2,
              l2, tol, x1, dx1dr, ele1, x1_inds, x2, dx2dr, ele2, x2_inds, pout, nullptr);
}
void rbf_kff_many_with_grad(int n1, int n2, int n2_start, int n2_end, int d, int x2i, double zeta,
                            double sigma2, double l, double *x1, double *dx1dr, int *ele1, int *x1_inds,
                            double *x2, double *dx2dr, int *ele2, int *x2_inds, double *pout,
                            double *dpout_dl) {
    ff_native("rbf_kff_many_with_grad", GPR_KERNEL_RBF, true, 3, n1, n2, n2_start, n2_end, d, x2i, zeta,
              sigma2, l, 0.0, x1, dx1dr, ele1, x1_inds, x2, dx2dr, ele2, x2_inds, pout, dpout_dl);
}
void rbf_kff_many_stress(int n1, int n2, int n2_start, int n2_end, int d, int x2i, double zeta,
                         double sigma2, double l2, double tol, double *x1, double *dx1dr, int *ele1,
                         int *x1_inds, double *x2, double *dx2dr, int *ele2, int *x2_inds, double *pout) {
    ff_native("rbf_kff_many_stress", GPR_KERNEL_RBF, false, 9, n1, n2, n2_start, n2_end, d, x2i, zeta,
              sigma2, l2, tol, x1, dx1dr, ele1, x1_inds, x2, dx2dr, ele2, x2_inds, pout, nullptr);
}
void dot_kff_many(int n1, int n2, int n2_start, int n2_end, int d, int x2i, double zeta, double *x1,
                  double *dx1dr, int *ele1, int *x1_inds, double *x2, double *dx2dr, int *ele2,
                  int *x2_inds, double *pout) {
    ff_native("dot_kff_many", GPR_KERNEL_DOT, false, 3, n1, n2, n2_start, n2_end, d, x2i, zeta, 1.0, 1.0,
              0.0, x1, dx1dr, ele1, x1_inds, x2, dx2dr, ele2, x2_inds, pout, nullptr);
}
void dot_kff_many_stress(int n1, int n2, int n2_start, int n2_end, int d, int x2i, double zeta, double *x1,
                         double *dx1dr, int *ele1, int *x1_inds, double *x2, double *dx2dr, int *ele2,
                         int *x2_inds, double *pout) {
    ff_native("dot_kff_many_stress", GPR_KERNEL_DOT, false, 9, n1, n2, n2_start, n2_end, d, x2i, zeta, 1.0,
              1.0, 0.0, x1, dx1dr, ele1, x1_inds, x2, dx2dr, ele2, x2_inds, pout, nullptr);
}

// status of the last L0 native on this thread (GPR_OK, or why it returned without touching its outputs)
int gpr_l0_last_status(void) { return g_l0_status; }

}  // extern "C"
