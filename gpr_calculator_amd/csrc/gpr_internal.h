// gpr_internal.h -- shared declarations of libgprhip.so (not part of the public C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "gpr_hip.h"

// ---------------------------------------------------------------- error plumbing
void gpr_set_error(const char *fmt, ...);

#define GPR_HIP_TRY(expr)                                                                    \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            gpr_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr,                 \
                          hipGetErrorString(_e));                                            \
            return GPR_ERR_HIP;                                                              \
        }                                                                                    \
    } while (0)

#define GPR_REQUIRE(cond, ...)                                                               \
    do {                                                                                     \
        if (!(cond)) {                                                                       \
            gpr_set_error(__VA_ARGS__);                                                      \
            return GPR_ERR_ARG;                                                              \
        }                                                                                    \
    } while (0)

// ---------------------------------------------------------------- geometry constants
constexpr int GPR_WAVE = 64;
constexpr int GPR_KSTEPS = 8;          // usual case: descriptor length padded to 4*KSTEPS = 32 (d <= 32)
constexpr int GPR_KPAD = 4 * GPR_KSTEPS;
constexpr int GPR_PLANE = GPR_KSTEPS * GPR_WAVE;   // doubles per 16x(4*KSTEPS) operand plane (4 KiB)
constexpr int GPR_CHUNK_TILES = 8;     // stream tiles staged per LDS buffer (32 KiB)
// longer descriptors (32 < d <= 64, e.g. SO3(nmax=4, lmax=4): d = 50): 16 k-steps per plane (8 KiB) and 4 tiles per
// staged chunk -- the same 32 KiB of LDS per buffer.  A set records which form it was packed in (gpr_set::ks).
constexpr int GPR_KSTEPS_MAX = 16;
// still longer ones (64 < d <= 128: SO3(4, 6) has d = 70, SO3(5, 4) 75, SO3(6, 4) 105 -- the reference loops over any d,
// rbf_kernel.cpp:341-473): 32 k-steps per plane (16 KiB), 2 tiles per staged chunk.  The resident operands of a group are
// then 256 registers per lane, so these instances run one wavefront per SIMD: correct first, about half the rate.
constexpr int GPR_KSTEPS_WIDE = 32;
constexpr int GPR_KPAD_MAX = 4 * GPR_KSTEPS_WIDE;
constexpr int GPR_GROUP_ROWS = 16;     // rows per resident group (one MFMA N extent)
constexpr int GPR_WU_MAX_GROUPS = 8;   // resident groups one wavefront owns at most (a wave unit: consecutive points)
constexpr int GPR_ELE_PAD_STREAM = INT_MIN;        // species code of padded / skipped stream rows
constexpr int GPR_ELE_PAD_RESIDENT = INT_MIN + 1;  // ... resident rows (never equal to the above)

// chunk descriptor of the stream side: tiles [tile0, tile0+ntiles) all belong to packed point `pt`
struct gpr_chunk { int tile0, ntiles, pt, last; };
// resident wave-unit descriptor: groups [group0, group0+ngroups) hold the rows of the unit's points back to back
struct gpr_rpoint { int group0, ngroups; };

struct gpr_set {
    int role, d, width, col0, nc, flags;
    int ks = GPR_KSTEPS;   // k-steps per operand plane: 8 (d <= 32), 16 (d <= 64) or 32 (d <= 128); plane = ks * 64 doubles
    int npts, nrows, rpad;
    int tile_rows;     // rows per operand plane on this side (stream force: 4, otherwise 16)
    int nunits;        // number of planes-groups: stream tiles or resident groups
    // device arrays
    double *ops = nullptr;     // stream: [ntiles][KSTEPS][64]; resident: [ngroups][nc][KSTEPS][64]
    int *ele = nullptr;        // [rpad]
    double *norms = nullptr;   // [rpad] effective row norms (used by the numpy-eps ee variant)
    int *perm = nullptr;       // [npts] packed position -> caller's point index
    double *inv_cnt = nullptr; // [npts] 1/rows of the point, packed order
    gpr_chunk *chunks = nullptr;    // stream only
    int *pt_chunk0 = nullptr;       // stream only [npts+1]
    int *ir_chunk0 = nullptr;       // stream only [nranges+1] chunk range of each launch range
    int *ir_pt0 = nullptr;          // stream only [nranges+1] first packed point of each range
    int nranges = 0;
    gpr_rpoint *rpts = nullptr;     // resident only [nwu] wave units
    int *wu_pt0 = nullptr;          // resident only [nwu+1] first packed point of each wave unit
    int *rowpt = nullptr;           // resident only [rpad] packed point of every padded row (-1: padding)
    int nwu = 0;
    int nchunks = 0, maxgroups = 0;
    int max_unit_pts = 0;           // resident only: points in the largest wave unit (kblock's split mode takes up to 8)
    // host mirrors used for launch planning
    std::vector<int> h_perm, h_cnt, h_pt_chunk0, h_pt_tiles;
    // host sources of the asynchronous uploads: kept alive with the set so that no stream
    // synchronisation is needed at creation time
    std::vector<int> hk_src_row, hk_irc, hk_irp, hk_rowpt, hk_wu_pt0;
    std::vector<double> hk_inv_cnt;
    std::vector<gpr_chunk> hk_chunks;
    std::vector<gpr_rpoint> hk_rpts;
    // every device array above lives in ONE allocation, and the index arrays reach it in ONE copy (a set used to cost a dozen
    // allocations and seven small copies: 58 copies of 4.5 us per single-structure prediction of the pair-kernel route)
    void *blob = nullptr;
    std::vector<char> hk_blob;
    long long bytes = 0;
};

// ---------------------------------------------------------------- internal block launcher
enum { GPR_MODE_FF = 0, GPR_MODE_EF = 1, GPR_MODE_EE = 2 };

struct gpr_block_call {
    int mode;
    const gpr_set *s1, *s2;
    int kind;                 // gpr_kernel_kind
    double zeta, sigma2, l, l2, sigma02, tol;   // l2 = l*l as the caller computed it (natives take l2)
    int grad;                 // also produce the l-derivative (RBF only)
    int symmetric, diag_only;
    double scale, dl_scale;   // global factors on the value / derivative outputs
    int norm1, norm2;         // multiply by 1/count of the stream / resident point
    int eps_mode;             // numpy-eps cosine (ee diag only)
    double *out; long long s_i, s_j, s_a, s_b;      // element strides of the value output
    double *dl;  long long d_i, d_j, d_a, d_b;      // ... of the derivative output
};
// zero-fills nothing: the caller has already zeroed the output; bins are accumulated with RMW adds
int gpr_launch_block(hipStream_t st, const gpr_block_call &c);

// Phase-ablation exits of the large kernels (profiling: "how long does the kernel take up to here").  They exist only in a
// tools build (make ABLATION=1 -> -DGPR_PHASE_ABLATION, then GPR_SO3_DBG / GPR_NBR_DBG / GPR_TAIL_DBG select the phase); the
// production library has no such branch and does not read those variables -- a stray one cannot change a result.
#ifdef GPR_PHASE_ABLATION
#define GPR_PHASE_EXIT(dbg_, k_) if ((dbg_) == (k_)) return;
#define GPR_PHASE_ENV(name_) (getenv(name_) ? atoi(getenv(name_)) : 0)
#else
#define GPR_PHASE_EXIT(dbg_, k_)
#define GPR_PHASE_ENV(name_) 0
#endif

// profiling kinds of gpr_profile_fetch (kernel-only HIP-event timing, see include/gpr_hip.h)
enum gpr_prof_kind { GPR_PROF_FF = 0, GPR_PROF_EF = 1, GPR_PROF_EE = 2, GPR_PROF_DIAG = 3, GPR_PROF_TRMM = 4,
                     GPR_PROF_GEMM = 5, GPR_PROF_SO3 = 6, GPR_PROF_FEATURES = 7, GPR_PROF_PACK = 8, GPR_PROF_GEMV = 9,
                     GPR_PROF_SO3_RADIAL = 10, GPR_PROF_NKINDS = 11 };
int gpr_prof_begin(hipStream_t st, int kind);
void gpr_prof_end(hipStream_t st, int id);

// tiled FP64 GEMM of dense.hip: C = alpha op(A) op(B) + beta C; lower_only bit 0: skip tiles above the diagonal,
// bit 1: op(A) lower triangular, bit 2: op(B) = B^T with B lower triangular, bit 3: 64 x 64 tiles, bit 4: op(B) = B lower triangular
int gpr_syrk_bulk_internal(hipStream_t st, int m, int k, const double *P, long long ldp, double *C, long long ldc, int nwg);   // C[lower tiles] -= P P^T on nwg CUs at most
int gpr_gemm_internal(hipStream_t st, int ta, int tb, int m, int n, int k, double alpha, const double *a, long long lda,
                      const double *b, long long ldb, double beta, double *c, long long ldc, int lower_only);

// gpr_ws_tail_status with standard deviations instead of variances in the second half of `out` (std_out != 0)
int gpr_ws_tail_internal(void *stream, int m, int n, const double *P, long long ldp, const double *Cinv, long long ldc,
                         const double *w, const double *diag, double *out, double vsq, const int *status_src,
                         double *status_dst, int std_out);

// small helpers
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device): code objects are loaded per device, so a
// process-wide "done" flag would leave the second device of a process at the 64 KB default
int gpr_func_lds_once(const void *fn, int bytes);
int gpr_cu_count();                                 // compute units of the current device (cached per device)
int gpr_dev_alloc(void **p, size_t bytes, hipStream_t st);
// zeros a [rows x cols] block of doubles with leading dimension ld (one memset when it is contiguous, a coalesced fill kernel otherwise:
// the runtime's 2-D fill does 0.8 TB/s on a 6063 x 6063 block)
int gpr_zero_2d(hipStream_t st, double *p, long long ld, long long rows, long long cols);
int gpr_dev_free(void *p, hipStream_t st);
int gpr_dev_cache_drop_stream(hipStream_t st);      // before hipStreamDestroy: cached temporaries of that stream go back to the driver
void gpr_reap_sets();   // frees the host side of destroyed sets whose stream work has completed
