// dense.hip -- FP64 dense linear algebra of GP.fit / GP.predict on the matrix cores:
// Cholesky factor, triangular solves, GEMM, and the small reductions around them.
// Replaces scipy.linalg.cholesky / cho_solve / solve_triangular and the numpy products of
// gaussianprocess.py:128-131, 288-299, 880, 903-907.  Row-major storage throughout.
#include <cstdlib>
#include <cstdint>
#include <map>
#include <mutex>
#include <unordered_map>
#include <utility>
#include "gpr_internal.h"

typedef double d4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BM = 64, BN = 64, BK = 16, LDS_LD = BM + 16;   // +16 doubles: rows k, k+1 hit disjoint banks
constexpr int NB = 64;                                        // Cholesky / TRSM panel width
constexpr int GPR_FEW_COLS = 256;                             // up to here the column reductions / the variance product take their few-column forms

// C[m,n] = alpha * op(A) op(B) + beta * C ;  op(A) is m x k, op(B) is k x n
// one 64x64 C tile per 256-thread workgroup; each wavefront owns a 32x32 quadrant = 2x2 MFMA tiles
template <bool TA, bool TB>
__global__ void __launch_bounds__(256) gemm_kernel(int m, int n, int k, double alpha,
                                                    const double *__restrict__ A, long long lda,
                                                    const double *__restrict__ B, long long ldb,
                                                    double beta, double *__restrict__ C, long long ldc,
                                                    int lower_only) {
    __shared__ double As[BK][LDS_LD];
    __shared__ double Bs[BK][LDS_LD];
    const int bm0 = blockIdx.y * BM, bn0 = blockIdx.x * BN;
    if ((lower_only & 1) && bn0 > bm0 + BM - 1) return;      // tile strictly above the diagonal
    if (lower_only & 2) k = min(k, bm0 + BM);                // op(A) lower triangular: zero beyond the row block
    if (lower_only & 4) k = min(k, bn0 + BN);                // op(B) = B^T with B lower triangular: zero beyond the column block
    if (lower_only & (16 | 64)) {                            // op(B) = B lower triangular: zero before the column block (16);
        const int kbeg = (((lower_only & 64) ? bm0 : bn0) / BK) * BK;   // op(A) = A^T with A lower triangular: before the row block (64)
        A += TA ? (long long)kbeg * lda : (long long)kbeg; B += TB ? (long long)kbeg : (long long)kbeg * ldb; k -= kbeg;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    d4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};

    for (int k0 = 0; k0 < k; k0 += BK) {
        // the eight loads of a k tile carry no predicate (under a lane-dependent branch each is waited for at the join: eight round
        // trips per tile): addresses clamped into the operand, what lies outside it becomes zero when it is stored to LDS
        double va[4], vb[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int mm, kk;
            if (!TA) { kk = tid & 15; mm = (tid >> 4) + 16 * i; }
            else     { mm = tid & 63; kk = (tid >> 6) + 4 * i; }
            const int gm = min(bm0 + mm, m - 1), gk = min(k0 + kk, k - 1);
            va[i] = TA ? A[(long long)gk * lda + gm] : A[(long long)gm * lda + gk];
            int nn, kb;
            if (!TB) { nn = tid & 63; kb = (tid >> 6) + 4 * i; }
            else     { kb = tid & 15; nn = (tid >> 4) + 16 * i; }
            const int gn = min(bn0 + nn, n - 1), gk2 = min(k0 + kb, k - 1);
            vb[i] = TB ? B[(long long)gn * ldb + gk2] : B[(long long)gk2 * ldb + gn];
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int mm, kk;
            if (!TA) { kk = tid & 15; mm = (tid >> 4) + 16 * i; }
            else     { mm = tid & 63; kk = (tid >> 6) + 4 * i; }
            As[kk][mm] = (bm0 + mm < m && k0 + kk < k) ? va[i] : 0.0;
            int nn, kb;
            if (!TB) { nn = tid & 63; kb = (tid >> 6) + 4 * i; }
            else     { kb = tid & 15; nn = (tid >> 4) + 16 * i; }
            Bs[kb][nn] = (bn0 + nn < n && k0 + kb < k) ? vb[i] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < BK / 4; ks++) {
            const int kr = ks * 4 + (lane >> 4);
            double a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; i++) a[i] = As[kr][wm + i * 16 + (lane & 15)];
#pragma unroll
            for (int j = 0; j < 2; j++) b[j] = Bs[kr][wn + j * 16 + (lane & 15)];
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    // C/D layout of v_mfma_f64_16x16x4_f64: col = lane&15, row = (lane>>4) + 4*reg
    double cold[2][2][4];
    if (beta != 0.0) {                               // the reads of C issued together (see the k loop)
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int r = 0; r < 4; r++)
                    cold[i][j][r] = C[(long long)min(bm0 + wm + i * 16 + (lane >> 4) + 4 * r, m - 1) * ldc + min(bn0 + wn + j * 16 + (lane & 15), n - 1)];
    }
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int gm = bm0 + wm + i * 16 + (lane >> 4) + 4 * r;
                const int gn = bn0 + wn + j * 16 + (lane & 15);
                if (gm < m && gn < n) {
                    const double v = alpha * acc[i][j][r];
                    C[(long long)gm * ldc + gn] = (beta == 0.0) ? v : v + beta * cold[i][j][r];
                }
            }
}

// Large-tile variant: 128x128 C tile per 256-thread workgroup, each wavefront a 64x64 quadrant =
// 4x4 MFMA tiles (16 accumulators), k-tiles of 16 double-buffered in LDS with the next tile's global
// loads in flight during the MFMAs.  LDS images keep the operand's storage order (so the stores are
// conflict free) and are padded so that the 64 lanes of an MFMA operand read hit 64 distinct banks:
//   stored [row][k]  (A not transposed, B transposed):  row stride 17 doubles
//   stored [k][col]  (A transposed, B not transposed):  row stride 144 doubles
constexpr int GM = 128, GN = 128, GK = 16, GLDK = GM + 16, GLDR = GK + 1;
// C rows = rows lb0*nb.. of the row blocks (size nb, a multiple of 128) rank `rank` of `nranks` owns under a 1-D
// block-cyclic distribution (local block b = global block b*nranks + rank); C column j = global column col0 + j
struct cyc_rule { int nb, nranks, rank, lb0, col0; };
constexpr int GBUF = (GK * GLDK > GM * GLDR) ? GK * GLDK : GM * GLDR;

// FAST (leading dimensions below 2^21): the tile loads carry no predicate.  Rows or columns beyond the matrix are clamped to its
// last one (their products are never stored), the uniform part of the address (tile origin + k0) sits in scalar registers -- buffer
// loads: descriptor + 32-bit lane offset + scalar offset -- and each lane keeps sixteen 32-bit offsets; a k range that is not a
// multiple of 16 ends in one tile whose loads are clamped into the range and whose A values beyond the end become zeros.  The generic
// form spends ~150 instructions per k tile on predicates and 64-bit pointers, under lane-dependent branches; with the loads removed
// altogether the kernel ran at 63 TF instead of 55 (k = 4096).  NT at m = n = 5807: k = 256, beta = 1 0.499 -> 0.413 ms;
// k = 4096 55 -> 60 TF.
template <bool TA, bool TB, bool FAST = false>
__global__ void __launch_bounds__(256) gemm128_kernel(int m, int n, int k, double alpha,
                                                       const double *__restrict__ A, long long lda,
                                                       const double *__restrict__ B, long long ldb,
                                                       double beta, double *__restrict__ C, long long ldc,
                                                       int lower_only, int ksplit, double *__restrict__ ws, cyc_rule cyc) {
    __shared__ double As[2][GBUF];
    __shared__ double Bs[2][GBUF];
    // With a triangular operand the k range of a tile depends on its row or column block: the tiles are handed out longest
    // first (a tile with the full k range started in the last round of workgroups was the tail of the launch -- the two
    // top-level products of the triangular inverse at n = 6063 ran at 33 TF): rows in descending order when k ends at
    // the row block (2), column-major with the long columns first when it ends (4) or begins (16) at the column block.
    int bx = blockIdx.x, by = blockIdx.y;
    if (ksplit == 0 && cyc.nb == 0) {
        if (lower_only & 2) by = (int)gridDim.y - 1 - by;
        else if (lower_only & (4 | 16)) {
            const int lin = by * (int)gridDim.x + bx, c = lin / (int)gridDim.y;
            by = lin - c * (int)gridDim.y;
            bx = (lower_only & 4) ? (int)gridDim.x - 1 - c : c;
        }
    }
    const int bm0 = by * GM, bn0 = bx * GN;
    if (cyc.nb > 0) {            // rows are the block-cyclic rows of one rank: skip tiles above the GLOBAL diagonal
        const int last = min(bm0 + GM, m) - 1;                                   // last local row of the tile
        const long long grow = ((long long)(cyc.lb0 + last / cyc.nb) * cyc.nranks + cyc.rank) * cyc.nb + last % cyc.nb;
        if ((long long)cyc.col0 + bn0 > grow) return;
    }
    if (ksplit > 0) {            // split-K slice blockIdx.z: rows [k0, k0 + ksplit) of op(A), op(B); plain sums into ws[z]
        const int k0 = blockIdx.z * ksplit;
        A += TA ? (long long)k0 * lda : (long long)k0;
        B += TB ? (long long)k0 : (long long)k0 * ldb;
        k = min(k - k0, ksplit);
        C = ws + (size_t)blockIdx.z * m * n; ldc = n; alpha = 1.0; beta = 0.0;
    }
    if ((lower_only & 1) && bn0 > bm0 + GM - 1) return;
    if (lower_only & 2) k = min(k, bm0 + GM);
    if (lower_only & 4) k = min(k, bn0 + GN);
    if (lower_only & (16 | 64)) {                                // op(B) = B lower triangular: zero before the column block (16);
        const int kbeg = (((lower_only & 64) ? bm0 : bn0) / GK) * GK;   // op(A) = A^T with A lower triangular: before the row block (64)
        A += TA ? (long long)kbeg * lda : (long long)kbeg; B += TB ? (long long)kbeg : (long long)kbeg * ldb; k -= kbeg;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    d4 acc[4][4];
    // alpha = +-1 with beta != 0 (the trailing update of the factorisation, C -= A B^T): the accumulators START from
    // (beta / alpha) C, requested before the first operand tile, so the tile's read of C is in flight during the k loop's
    // prologue and the epilogue is a plain store -- instead of a read-modify-write whose read nothing hides (at k = 256 that
    // read was a quarter of the kernel: 0.514 ms with beta = 1 against 0.387 ms with beta = 0 at m = n = 5807)
    const bool init_c = beta != 0.0 && (alpha == 1.0 || alpha == -1.0);
    if (init_c) {
        const double sc = alpha * beta;
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    // no predicate on the loads (64 loads under lane-dependent branches are 64 round trips one after the other:
                    // the compiler drains the counter at every join); entries outside the matrix start from some element of
                    // its last row or column and are never stored
                    const int gm = min(bm0 + wm + i * 16 + (lane >> 4) + 4 * r, m - 1);
                    const int gn = min(bn0 + wn + j * 16 + (lane & 15), n - 1);
                    acc[i][j][r] = sc * C[(long long)gm * ldc + gn];
                }
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};
    }

    double ra[8], rb[8];
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            int mm, kk;
            if (!TA) { kk = tid & 15; mm = (tid >> 4) + 16 * i; }
            else     { mm = tid & 127; kk = (tid >> 7) + 2 * i; }
            const int gm = bm0 + mm, gk = k0 + kk;
            ra[i] = (gm < m && gk < k) ? (TA ? A[(long long)gk * lda + gm] : A[(long long)gm * lda + gk]) : 0.0;
            int nn, kb;
            if (!TB) { nn = tid & 127; kb = (tid >> 7) + 2 * i; }
            else     { kb = tid & 15; nn = (tid >> 4) + 16 * i; }
            const int gn = bn0 + nn, gk2 = k0 + kb;
            rb[i] = (gn < n && gk2 < k) ? (TB ? B[(long long)gn * ldb + gk2] : B[(long long)gk2 * ldb + gn]) : 0.0;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (!TA) { const int kk = tid & 15, mm = (tid >> 4) + 16 * i; As[buf][mm * GLDR + kk] = ra[i]; }
            else     { const int mm = tid & 127, kk = (tid >> 7) + 2 * i; As[buf][kk * GLDK + mm] = ra[i]; }
            if (!TB) { const int nn = tid & 127, kb = (tid >> 7) + 2 * i; Bs[buf][kb * GLDK + nn] = rb[i]; }
            else     { const int kb = tid & 15, nn = (tid >> 4) + 16 * i; Bs[buf][nn * GLDR + kb] = rb[i]; }
        }
    };
    const int nkt = (k + GK - 1) / GK;
    // FAST: byte offsets of this lane's sixteen elements from the tile origins (k = 0).  An operand stored [row][k] (A not
    // transposed, B transposed) advances along k by a scalar offset; one stored [k][row] by whole rows of the array, more than 32
    // bits can hold, so its descriptor is rebuilt per k tile (scalar arithmetic)
    unsigned oa[8], ob[8];
    const char *Ab = nullptr, *Bb = nullptr;
    if constexpr (FAST) {
        Ab = (const char *)(TA ? A + bm0 : A + (long long)bm0 * lda);
        Bb = (const char *)(TB ? B + (long long)bn0 * ldb : B + bn0);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            oa[i] = TA ? (unsigned)(((tid >> 7) + 2 * i) * (int)lda + min(tid & 127, m - 1 - bm0)) * 8u
                       : (unsigned)(min((tid >> 4) + 16 * i, m - 1 - bm0) * (int)lda + (tid & 15)) * 8u;
            ob[i] = TB ? (unsigned)(min((tid >> 4) + 16 * i, n - 1 - bn0) * (int)ldb + (tid & 15)) * 8u
                       : (unsigned)(((tid >> 7) + 2 * i) * (int)ldb + min(tid & 127, n - 1 - bn0)) * 8u;
        }
    }
    // buffer loads: descriptor (scalar) + 32-bit lane offset + scalar k offset -- no vector arithmetic per load
    auto rsrc = [&](const char *base, unsigned long long bytes) {
        return __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)(unsigned)(bytes > 0xFFFFFFFFull ? 0xFFFFFFFFull : bytes), 0x00020000);
    };
    auto ld8 = [&](__amdgpu_buffer_rsrc_t r, unsigned off, int soff) {
        return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, off, soff, 0));
    };
    const int rows_a = min(GM, m - bm0), rows_b = min(GN, n - bn0);
    // [row][k] operands: one descriptor for the whole panel
    const __amdgpu_buffer_rsrc_t pa_src = rsrc(Ab, ((unsigned long long)(rows_a - 1) * (unsigned long long)lda + (unsigned long long)k) * 8ull);
    const __amdgpu_buffer_rsrc_t pb_src = rsrc(Bb, ((unsigned long long)(rows_b - 1) * (unsigned long long)ldb + (unsigned long long)k) * 8ull);
    auto tile_src = [&](const char *base, long long ld, int rows, int k0) {          // [k][row] operands: the k tile at k0
        return rsrc(base + (long long)k0 * ld * 8, ((unsigned long long)(min(GK, k - k0) - 1) * (unsigned long long)ld + (unsigned long long)rows) * 8ull);
    };
    auto load_full = [&](int k0) {                       // every k of the tile inside the range
        const __amdgpu_buffer_rsrc_t sa = TA ? tile_src(Ab, lda, rows_a, k0) : pa_src, sb = TB ? pb_src : tile_src(Bb, ldb, rows_b, k0);
        const int ka = TA ? 0 : k0 * 8, kb = TB ? k0 * 8 : 0;
#pragma unroll
        for (int i = 0; i < 8; i++) { ra[i] = ld8(sa, oa[i], ka); rb[i] = ld8(sb, ob[i], kb); }
    };
    auto load_last = [&](int k0) {                       // the tile the range ends in: k clamped into it, A's surplus zeroed
        const __amdgpu_buffer_rsrc_t sa = TA ? tile_src(Ab, lda, rows_a, k0) : pa_src, sb = TB ? pb_src : tile_src(Bb, ldb, rows_b, k0);
        const int ka = TA ? 0 : k0 * 8, kb = TB ? k0 * 8 : 0, klast = k - 1 - k0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int over_a = max((TA ? (tid >> 7) + 2 * i : (tid & 15)) - klast, 0);
            const int over_b = max((TB ? (tid & 15) : (tid >> 7) + 2 * i) - klast, 0);
            const double va = ld8(sa, oa[i] - 8u * (unsigned)over_a * (TA ? (unsigned)lda : 1u), ka);
            ra[i] = over_a ? 0.0 : va;
            rb[i] = ld8(sb, ob[i] - 8u * (unsigned)over_b * (TB ? 1u : (unsigned)ldb), kb);
        }
    };
    if constexpr (FAST) {
        if (nkt > 0) { if (nkt == 1) load_last(0); else load_full(0); store_tile(0); }
    } else {
        if (nkt > 0) { load_tile(0); store_tile(0); }
    }
    __syncthreads();
    for (int kt = 0; kt < nkt; kt++) {
        const int buf = kt & 1;
        if constexpr (FAST) {
            if (kt + 2 < nkt) load_full((kt + 1) * GK);
            else if (kt + 1 < nkt) load_last((kt + 1) * GK);
        } else {
            if (kt + 1 < nkt) load_tile((kt + 1) * GK);
        }
#pragma unroll
        for (int ks = 0; ks < GK / 4; ks++) {
            const int kr = ks * 4 + (lane >> 4);
            double a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int mm = wm + i * 16 + (lane & 15);
                a[i] = TA ? As[buf][kr * GLDK + mm] : As[buf][mm * GLDR + kr];
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int nn = wn + j * 16 + (lane & 15);
                b[j] = TB ? Bs[buf][nn * GLDR + kr] : Bs[buf][kr * GLDK + nn];
            }
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nkt) store_tile(buf ^ 1);
        __syncthreads();
    }
    const bool rmw = !(beta == 0.0 || init_c);       // general alpha with beta != 0: C is read here, a row of tiles at a time,
#pragma unroll                                       // every load issued (addresses clamped into the matrix) before the first use
    for (int i = 0; i < 4; i++) {
        double cold[4][4];
        if (rmw) {
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int gm = min(bm0 + wm + i * 16 + (lane >> 4) + 4 * r, m - 1);
                    const int gn = min(bn0 + wn + j * 16 + (lane & 15), n - 1);
                    cold[j][r] = C[(long long)gm * ldc + gn];
                }
        }
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int gm = bm0 + wm + i * 16 + (lane >> 4) + 4 * r;
                const int gn = bn0 + wn + j * 16 + (lane & 15);
                if (gm < m && gn < n) {
                    const double v = alpha * acc[i][j][r];
                    C[(long long)gm * ldc + gn] = rmw ? v + beta * cold[j][r] : v;
                }
            }
    }
}

// diagonal-block solve of the blocked TRSM: one thread per right-hand-side column
// trans = 0: L x = b (forward) ; trans = 1: L^T x = b (backward)
__global__ void __launch_bounds__(256) trsm_diag_kernel(int trans, const double *lkk, long long ldl, int kb,
                                                         double *b, long long ldb, int nrhs) {
    __shared__ double L[NB][NB + 1];
    const int tid = threadIdx.x;
    for (int e = tid; e < kb * kb; e += 256) { int i = e / kb, j = e % kb; L[i][j] = lkk[(long long)i * ldl + j]; }
    __syncthreads();
    const int col = blockIdx.x * 256 + tid;
    if (col >= nrhs) return;
    double x[NB];
    if (!trans) {
        for (int i = 0; i < kb; i++) {
            double v = b[(long long)i * ldb + col];
            for (int q = 0; q < i; q++) v -= L[i][q] * x[q];
            x[i] = v / L[i][i];
        }
    } else {
        for (int i = kb - 1; i >= 0; i--) {
            double v = b[(long long)i * ldb + col];
            for (int q = i + 1; q < kb; q++) v -= L[q][i] * x[q];
            x[i] = v / L[i][i];
        }
    }
    for (int i = 0; i < kb; i++) b[(long long)i * ldb + col] = x[i];
}

// 64-wide slabs times a 64 x 64 block W (= L_kk^-1, zero padded):
//   mode 0: P[rows, kb] <- P W^T        (Cholesky panel: X L^T = P), one workgroup per 64 rows
//   mode 1: B[kb, ncols] <- W B         (forward solve of a diagonal block), one workgroup per 64 columns
//   mode 2: B[kb, ncols] <- W^T B       (backward solve)
__global__ void __launch_bounds__(256) tri_apply_kernel(int mode, const double *w, double *t, long long ld, int kb,
                                                         int extent) {
    __shared__ double Ws[NB][NB + 1];
    __shared__ double Ts[NB][NB + 1];
    const int tid = threadIdx.x, c = tid & 63, ty = tid >> 6, s0 = blockIdx.x * NB;
    for (int i = ty; i < NB; i += 4) {
        Ws[i][c] = w[i * NB + c];
        double v = 0.0;
        if (mode == 0) { if (s0 + i < extent && c < kb) v = t[(long long)(s0 + i) * ld + c]; }
        else           { if (i < kb && s0 + c < extent) v = t[(long long)i * ld + s0 + c]; }
        Ts[i][c] = v;
    }
    __syncthreads();
    double acc[16];
#pragma unroll
    for (int u = 0; u < 16; u++) acc[u] = 0.0;
    for (int q = 0; q < NB; q++) {
        if (mode == 0) {
            const double wv = Ws[c][q];
#pragma unroll
            for (int u = 0; u < 16; u++) acc[u] += Ts[ty + 4 * u][q] * wv;
        } else {
            const double tv = Ts[q][c];
#pragma unroll
            for (int u = 0; u < 16; u++) acc[u] += (mode == 1 ? Ws[ty + 4 * u][q] : Ws[q][ty + 4 * u]) * tv;
        }
    }
#pragma unroll
    for (int u = 0; u < 16; u++) {
        const int i = ty + 4 * u;
        if (mode == 0) { if (s0 + i < extent && c < kb) t[(long long)(s0 + i) * ld + c] = acc[u]; }
        else           { if (i < kb && s0 + c < extent) t[(long long)i * ld + s0 + c] = acc[u]; }
    }
}

__global__ void zero_upper_kernel(int n, double *a, long long lda) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j < n && j > i) a[(long long)i * lda + j] = 0.0;
}

__global__ void add_diag_kernel(int n, double *a, long long lda, int n_first, double v1, double v2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[(long long)i * lda + i] += (i < n_first) ? v1 : v2;
}

// column sums of squares: thread per column, rows split over blockIdx.y with a fixed-order second pass
__global__ void colsumsq_partial_kernel(int n, int ncols, const double *v, long long ldv, double *part, int rows_per) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ncols) return;
    const int r0 = blockIdx.y * rows_per, r1 = min(n, r0 + rows_per);
    double acc = 0.0;
    for (int i = r0; i < r1; i++) { double t = v[(long long)i * ldv + j]; acc += t * t; }
    part[(long long)blockIdx.y * ncols + j] = acc;
}
__global__ void colsum_final_kernel(int nparts, int ncols, const double *part, double *out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ncols) return;
    double acc = 0.0;
    for (int p = 0; p < nparts; p++) acc += part[(long long)p * ncols + j];
    out[j] = acc;
}

// ---- few columns (one structure: 1 + 3 n_atoms quantities): the thread-per-column kernels above leave most of the chip idle
// and walk their rows serially.  Here a block of 256 threads is (256 / CW row lanes) x (CW columns); the row lanes are added in
// lane order through LDS, the blocks by colsum_final_kernel in block order.
template <int CW, bool SQ>
__global__ void __launch_bounds__(256) colred_few_kernel(int n, int ncols, const double *__restrict__ a, long long lda,
                                                        const double *__restrict__ x, double *__restrict__ part, int rows_per) {
    constexpr int RL = 256 / CW;
    __shared__ double red[256];
    const int j = blockIdx.x * CW + (threadIdx.x % CW), rl = threadIdx.x / CW;
    const int r0 = blockIdx.y * rows_per, r1 = min(n, r0 + rows_per);
    double acc = 0.0;
    if (j < ncols)
        for (int i = r0 + rl; i < r1; i += RL) {
            const double t = a[(long long)i * lda + j];
            acc += SQ ? t * t : t * x[i];
        }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (rl == 0 && j < ncols) {
#pragma unroll
        for (int k = 1; k < RL; k++) acc += red[k * CW + threadIdx.x];
        part[(long long)blockIdx.y * ncols + j] = acc;
    }
}

// second pass of the few-column reductions: four lanes per column take every fourth part, then are added in lane order
__global__ void __launch_bounds__(256) colsum_final_few_kernel(int nparts, int ncols, const double *__restrict__ part, long long ldp,
                                                               double *__restrict__ out) {
    __shared__ double red[256];
    const int j = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    double acc = 0.0;
    if (j < ncols)
        for (int p = rl; p < nparts; p += 4) acc += part[(long long)p * ldp + j];
    red[threadIdx.x] = acc;
    __syncthreads();
    if (rl == 0 && j < ncols) out[j] = ((acc + red[64 + threadIdx.x]) + red[128 + threadIdx.x]) + red[192 + threadIdx.x];
}

// Bp[i][0 .. ldp) = B[i][0 .. ncols) followed by zeros: the few-column product below reads whole, aligned rows
__global__ void pad_cols_kernel(int n, int ncols, const double *__restrict__ B, long long ldb, double *__restrict__ Bp, int ldp) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)n * ldp) return;
    const int i = (int)(e / ldp), j = (int)(e % ldp);
    Bp[e] = j < ncols ? B[(long long)i * ldb + j] : 0.0;
}

// part[w][j] = sum over the rows i of block w of (sum_k T[i][k] Bp[k][j])^2 for a LOWER-triangular T [n x n] and FEW columns
// (16 CT per blockIdx.y): the subtracted term of the predictive variance, |L^-1 k_j|^2 (gaussianprocess.py:903-907), without
// V = T B ever reaching memory.  T is streamed from HBM exactly once (n^2 / 2 doubles, the traffic that bounds the kernel);
// Bp (n x ldp, a few MB, zero-padded to whole tiles by pad_cols_kernel) is re-read from L2 by every block.  A block owns
// 16-row blocks of T; triangular rows cost in proportion to their index, so block w takes the pair (NB - 1 - w, w),
// every pair the same length.  Its eight wavefronts split the k range in runs of 16, and both operands go straight from
// memory into the MFMA operand registers with one wide load per lane: lane (r = l & 15, q = l >> 4) takes
// T[i0 + r][k0 + 4q .. + 3] (a full 128-byte line per row and run) and, for each of those four k, Bp[k][CT r .. CT r + CT - 1]
// -- so MFMA step e of a run multiplies k = k0 + 4q + e, and column tile t holds the columns CT r + t (a permutation that
// the column sums do not care about).  Rows of T are only 8-byte aligned (odd n), which the wide global loads accept.
// Every load of the main loop is unconditional (no divergent control flow, so the loads of later runs stay in flight
// across the MFMA steps of earlier ones: the T loads of four runs and the Bp loads of two per wavefront, 64 KB of T per
// CU); only the run that holds the diagonal block masks what lies above the diagonal.
// The eight partial tiles are folded in a fixed order through LDS; wavefront 0 squares and adds the 16 rows.
template <int W> struct dvec_u;
template <> struct dvec_u<1> { typedef double type; };
template <> struct dvec_u<2> { typedef double type __attribute__((ext_vector_type(2), aligned(8))); };
template <> struct dvec_u<4> { typedef double type __attribute__((ext_vector_type(4), aligned(8))); };
template <int W> __device__ __forceinline__ double dvec_get(const typename dvec_u<W>::type &v, int e) { return v[e]; }
template <> __device__ __forceinline__ double dvec_get<1>(const double &v, int) { return v; }

template <int CT>                                  // columns per lane: 1, 2 or 4 (16 CT columns per blockIdx.y)
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))     // one block per CU is all a launch has (n <= 8192)
trmm_sumsq_few_kernel(int n, const double *__restrict__ T, long long ldt, const double *__restrict__ Bp, int ldp,
                      double *__restrict__ part) {
    constexpr int NWV = 8, RUN = 16, DA = 4;
    typedef typename dvec_u<4>::type avec;
    typedef typename dvec_u<CT>::type bvec;
    __shared__ double sR[4][16 * 16 * CT];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    const int nb = (n + 15) >> 4;
    const int j0 = blockIdx.y * 16 * CT + CT * lr;                      // first column of this lane
    double colacc[CT];
#pragma unroll
    for (int t = 0; t < CT; t++) colacc[t] = 0.0;
    for (int leg = 0; leg < 2; leg++) {
        const int b = leg == 0 ? nb - 1 - (int)blockIdx.x : (int)blockIdx.x;
        if (leg == 1 && b >= nb - 1 - (int)blockIdx.x) break;            // the middle block of an odd count comes once
        const int i0 = b * 16, irow = i0 + lr;
        const int nrun = b + 1;                                          // run b holds the diagonal block
        const double *tp = T + (long long)min(irow, n - 1) * ldt + 4 * lq;      // rows past the end repeat the last one; dropped below
        const double *bp = Bp + (long long)(4 * lq) * ldp + j0;
        d4 acc[CT];
#pragma unroll
        for (int t = 0; t < CT; t++) acc[t] = d4{0.0, 0.0, 0.0, 0.0};
        auto load_a = [&](int run, double (&a)[4]) {                    // run < b: entirely below the diagonal
            const avec v = *(const avec *)(tp + run * RUN);
#pragma unroll
            for (int e = 0; e < 4; e++) a[e] = v[e];
        };
        auto load_b = [&](int run, double (&bb)[4][CT]) {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const bvec v = *(const bvec *)(bp + (long long)(run * RUN + e) * ldp);
#pragma unroll
                for (int t = 0; t < CT; t++) bb[e][t] = dvec_get<CT>(v, t);
            }
        };
        auto mma = [&](const double (&a)[4], const double (&bb)[4][CT]) {
#pragma unroll
            for (int e = 0; e < 4; e++)
#pragma unroll
                for (int t = 0; t < CT; t++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[e], bb[e][t], acc[t], 0, 0, 0);
        };
        // runs 0 .. b - 1 through the pipeline.  The prefetches past the last run repeat it (unused), so that the loop body has
        // no conditional load and the only branches are the scalar loop exits.
        const int nfull = nrun - 1, last = max(nfull - 1, 0);
        double aq[DA][4], bq[2][4][CT];
#pragma unroll
        for (int u = 0; u < DA; u++) load_a(min(wave + u * NWV, last), aq[u]);
        load_b(min(wave, last), bq[0]);
        // (groups of DA runs with no exit inside a group -- an exit is a place below which the compiler sinks the refills, see
        //  ws_tail_rows_kernel --: a run beyond the last full one multiplies by a zero A operand)
        for (int run = wave; run < nfull; run += DA * NWV) {
#pragma unroll
            for (int u = 0; u < DA; u++) {
                const int r_ = run + u * NWV;
                const bool live = r_ < nfull;
                load_b(min(r_ + NWV, last), bq[(u + 1) & 1]);
                double a[4];
#pragma unroll
                for (int e = 0; e < 4; e++) a[e] = live ? aq[u][e] : 0.0;
                load_a(min(r_ + DA * NWV, last), aq[u]);
                mma(a, bq[u & 1]);
            }
        }
        if (wave == (b & (NWV - 1))) {                                  // the diagonal run: k <= i only, and k < n
            double a[4], bb[4][CT];
            const int k = b * RUN + 4 * lq;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int kc = min(k + e, n - 1);
                const double av = T[(long long)min(irow, n - 1) * ldt + kc];
                const bvec v = *(const bvec *)(Bp + (long long)kc * ldp + j0);
                a[e] = (k + e <= irow && irow < n) ? av : 0.0;
#pragma unroll
                for (int t = 0; t < CT; t++) bb[e][t] = dvec_get<CT>(v, t);
            }
            mma(a, bb);
        }
        // fold the eight partial tiles: 4..7 onto 0..3, 2..3 onto 0..1, 1 onto 0
#pragma unroll
        for (int half = 4; half >= 1; half >>= 1) {
            if (wave >= half && wave < 2 * half)
#pragma unroll
                for (int t = 0; t < CT; t++)
#pragma unroll
                    for (int r = 0; r < 4; r++) sR[wave - half][(t * 4 + r) * 64 + lane] = acc[t][r];
            __syncthreads();
            if (wave < half)
#pragma unroll
                for (int t = 0; t < CT; t++)
#pragma unroll
                    for (int r = 0; r < 4; r++) acc[t][r] += sR[wave][(t * 4 + r) * 64 + lane];
            __syncthreads();
        }
        if (wave == 0) {                                                // acc[t][r] = V[row i0 + (l >> 4) + 4 r][column j0 + t]
#pragma unroll
            for (int t = 0; t < CT; t++) {
                double q = 0.0;
#pragma unroll
                for (int r = 0; r < 4; r++) q += (i0 + lq + 4 * r < n) ? acc[t][r] * acc[t][r] : 0.0;
                q += __shfl_xor(q, 16, 64);
                q += __shfl_xor(q, 32, 64);
                colacc[t] += q;
            }
        }
    }
    if (wave == 0 && lane < 16)
#pragma unroll
        for (int t = 0; t < CT; t++) part[(long long)blockIdx.x * ldp + j0 + t] = colacc[t];
}

__global__ void __launch_bounds__(256) sum_log_diag_kernel(int n, const double *a, long long lda, double *out) {
    __shared__ double red[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += log(a[(long long)i * lda + i]);
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
}

// sum_ij a[i*lda+j] * b[i*ldb+j]: per-block partials in a fixed order, then one block sums them
__global__ void __launch_bounds__(256) frob_partial_kernel(int m, int n, const double *a, long long lda,
                                                            const double *b, long long ldb, double *part) {
    __shared__ double red[256];
    double acc = 0.0;
    const long long total = (long long)m * n;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long i = e / n, j = e % n;
        acc += a[i * lda + j] * b[i * ldb + j];
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
__global__ void __launch_bounds__(256) sum_partials_kernel(int n, const double *part, double *out) {
    __shared__ double red[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += part[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
}

// out[i] = sum_j a[i*lda+j] * b[i*ldb+j]: one wavefront per row, fixed summation order
__global__ void __launch_bounds__(256) rowdot_kernel(int m, int n, const double *a, long long lda, const double *b,
                                                      long long ldb, double *out) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= m) return;
    double acc = 0.0;
    for (int j = lane; j < n; j += 64) acc += a[(long long)row * lda + j] * b[(long long)row * ldb + j];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) out[row] = acc;
}

// weight-space prediction tail: mean[i] = P[i,:] . w and var[i] = max(diag[i] - Q[i,:] . P[i,:], 0) in one pass over
// the rows (Q = P M); one wavefront per row, fixed summation order.  out = [mean (m) | var (m)]
// vsq > 0 selects the factored form: Q = P C^-T (G = C C^T) and var = diag - vsq (|p|^2 - |q|^2), vsq = sigma^2
__global__ void __launch_bounds__(256) ws_finish_kernel(int m, int n, const double *P, long long ldp, const double *Q,
                                                         long long ldq, const double *w, const double *diag, double *out,
                                                         double vsq) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= m) return;
    double a1 = 0.0, a2 = 0.0;
    for (int j = lane; j < n; j += 64) {
        const double pv = P[(long long)row * ldp + j];
        a1 += pv * w[j];
        if (Q) {
            const double qv = Q[(long long)row * ldq + j];
            a2 += (vsq > 0.0) ? (pv * pv - qv * qv) : pv * qv;
        }
    }
    for (int o = 32; o > 0; o >>= 1) { a1 += __shfl_xor(a1, o, 64); a2 += __shfl_xor(a2, o, 64); }
    if (lane == 0) {
        out[row] = a1;
        if (Q) out[m + row] = fmax(diag[row] - ((vsq > 0.0) ? vsq * a2 : a2), 0.0);
    }
}

// Weight-space prediction tail in one kernel, factored variance form: for the rows of P (feature rows psi of the
// predicted quantities) mean = P w and var = max(diag - vsq (|p|^2 - |p C^-T|^2), 0) with C^-1 lower triangular
// (G = C C^T).  The product Q = P C^-T is never stored: a workgroup owns WT_R rows, walks the column blocks of Q
// (the k range of a block ends at its diagonal block), keeps the squares of its Q entries in registers and reduces
// them per row in a fixed order; P w and |p|^2 are gathered from the A tiles of the last column block, whose k range
// is the whole row.  FP64 MFMA 16x16x4, tiles two steps ahead in registers, every workgroup does the same work.
#ifndef WT_ROWS
#define WT_ROWS 32
#endif
constexpr int WT_FEW_MAX_RB = 64;                        // row blocks of the few-rows form (its per-stream arrival counters)
constexpr int WT_C = 128, WT_K = 32, WT_LD = WT_K + 2;   // row stride 34: operand reads of a half-wave hit 32 distinct bank pairs
template <int WT_R, int NW = 4> constexpr size_t wt_lds_bytes() { return (size_t)(2 * (WT_R + WT_C) * WT_LD + NW * WT_R) * sizeof(double); }

// SPLIT = 1 (few rows, e.g. one structure): grid (column blocks, row blocks), a workgroup takes ONE column block so that
// the k chains of a row block run side by side; it leaves its sums of Q^2 in part[cb][row] (the owner of the last
// column block also P w and |p|^2 in mn), and whichever workgroup of the row block finishes last (cnt) adds the parts in
// column-block order and writes mean and variance.  SPLIT = 2 (many rows): grid (ceil(ncb / 2), row blocks), a
// workgroup takes column blocks x and ncb - 1 - x; ws_tail_combine_kernel adds the parts (a device-scope fence per
// workgroup, as the last-arriver scheme needs, costs more than that launch when there are hundreds of workgroups).
// SPLIT = 0: one workgroup walks all column blocks (a single column block, or as a reference).
// NW wavefronts per workgroup (4 or 8): with 8, two wavefronts share a SIMD and one multiplies while the other waits
// for its tile loads or at the barrier -- same tiles, same traffic, half the columns per wavefront.
template <int WT_R, int SPLIT, int NW = 4>
__global__ void __launch_bounds__(64 * NW) ws_tail_kernel(int m, int n, const double *__restrict__ P, long long ldp,
                                                       const double *__restrict__ Ci, long long ldc,
                                                       const double *__restrict__ w, const double *__restrict__ diag,
                                                       double *__restrict__ out, double vsq, double *__restrict__ part,
                                                       double *__restrict__ mn, int *__restrict__ cnt, int std_out,
                                                       const int *status_src, double *status_dst) {
    constexpr int RT = WT_R / 16, AI = WT_R / (2 * NW);          // 16-row MFMA tiles, A loads per thread and tile
    constexpr int NC = WT_C / (16 * NW), BI = WT_C / (2 * NW), LR = 2 * NW;   // column tiles per wavefront, B loads per thread and tile, rows per load pass
    __shared__ int s_last;
    extern __shared__ __attribute__((aligned(16))) double wt_smem[];
    double *As0 = wt_smem, *As1 = As0 + WT_R * WT_LD, *Bs0 = As1 + WT_R * WT_LD, *Bs1 = Bs0 + WT_C * WT_LD;
    double *red = Bs1 + WT_C * WT_LD;                            // [NW wavefronts][WT_R]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bm0 = (SPLIT ? blockIdx.y : blockIdx.x) * WT_R;
    const int wj = wave * (WT_C / NW);                           // this wavefront: all WT_R rows x columns wj .. wj + WT_C / NW - 1
    const int lk = tid & 31, lr = tid >> 5;                      // tile loads: k offset, row / column offset (+ LR i)
    double sq[RT][4];                                            // sum over columns of Q^2, rows 16 i + (lane >> 4) + 4 r
    double mean2[AI], norm2[AI];                                 // P w, |p|^2 partials of rows lr + 8 i
#pragma unroll
    for (int i = 0; i < RT; i++) { sq[i][0] = 0.0; sq[i][1] = 0.0; sq[i][2] = 0.0; sq[i][3] = 0.0; }
#pragma unroll
    for (int i = 0; i < AI; i++) { mean2[i] = 0.0; norm2[i] = 0.0; }
    const int ncb = (n + WT_C - 1) / WT_C;
    // SPLIT 2: `cnt` is the arrival counter of the combine launch that follows on the stream; it lives in this call's
    // workspace (two predictions in flight on two streams must not share one) and is zeroed here, a launch earlier
    if (SPLIT == 2 && cnt && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) *cnt = 0;
    // SPLIT 0: all column blocks; 1: column block blockIdx.x; 2: blocks blockIdx.x and ncb - 1 - blockIdx.x (a short k
    // range with a long one)
    const int nturn = SPLIT == 0 ? ncb : (SPLIT == 1 ? 1 : ((int)blockIdx.x * 2 + 1 == ncb ? 1 : 2));
    for (int turn = 0; turn < nturn; turn++) {
        const int cb = SPLIT == 0 ? turn : (turn == 0 ? (int)blockIdx.x : ncb - 1 - (int)blockIdx.x);
        const int bn0 = cb * WT_C;
        const int kend = min(n, bn0 + WT_C);                     // C^-1[j][k] = 0 for k > j
        const bool last = cb == ncb - 1;
        d4 acc[RT][NC];                                          // independent accumulation chains
#pragma unroll
        for (int i = 0; i < RT; i++)
#pragma unroll
            for (int j = 0; j < NC; j++) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};
        // two tiles ahead: the loads issued in step kt are stored at the end of step kt + 1; two register sets, the kt
        // loop is unrolled by two
        double ra0[AI], rb0[BI], ra1[AI], rb1[BI], wv0 = 0.0, wv1 = 0.0;
        auto load_tile = [&](int k0, double *ra, double *rb, double &wv) {
            const int gk = k0 + lk;
            if (last) wv = gk < n ? w[gk] : 0.0;
#pragma unroll
            for (int i = 0; i < AI; i++) {
                const int gm = bm0 + lr + LR * i;
                ra[i] = (gm < m && gk < kend) ? P[(long long)gm * ldp + gk] : 0.0;
            }
#pragma unroll
            for (int i = 0; i < BI; i++) {
                const int gn = bn0 + lr + LR * i;
                rb[i] = (gn < n && gk < kend) ? Ci[(long long)gn * ldc + gk] : 0.0;
            }
        };
        auto store_tile = [&](double *As, double *Bs, const double *ra, const double *rb, double wv) {
#pragma unroll
            for (int i = 0; i < AI; i++) As[(lr + LR * i) * WT_LD + lk] = ra[i];
#pragma unroll
            for (int i = 0; i < BI; i++) Bs[(lr + LR * i) * WT_LD + lk] = rb[i];
            if (last) {
#pragma unroll
                for (int i = 0; i < AI; i++) { mean2[i] += ra[i] * wv; norm2[i] += ra[i] * ra[i]; }
            }
        };
        auto mma_tile = [&](const double *As, const double *Bs) {
#pragma unroll
            for (int ks = 0; ks < WT_K / 4; ks++) {
                const int kr = ks * 4 + (lane >> 4);
                double bv[NC];
#pragma unroll
                for (int j = 0; j < NC; j++) bv[j] = Bs[(wj + 16 * j + (lane & 15)) * WT_LD + kr];
#pragma unroll
                for (int i = 0; i < RT; i++) {
                    const double a = As[(16 * i + (lane & 15)) * WT_LD + kr];
#pragma unroll
                    for (int j = 0; j < NC; j++) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv[j], acc[i][j], 0, 0, 0);
                }
            }
        };
        const int nkt = (kend + WT_K - 1) / WT_K;
        load_tile(0, ra0, rb0, wv0);
        if (nkt > 1) load_tile(WT_K, ra1, rb1, wv1);
        store_tile(As0, Bs0, ra0, rb0, wv0);
        __syncthreads();
        for (int kt = 0; kt < nkt; kt += 2) {
            // even step: LDS buffer 0 holds tile kt, set 1 holds tile kt + 1, set 0 is free for tile kt + 2
            if (kt + 2 < nkt) load_tile((kt + 2) * WT_K, ra0, rb0, wv0);
            mma_tile(As0, Bs0);
            if (kt + 1 < nkt) store_tile(As1, Bs1, ra1, rb1, wv1);
            __syncthreads();
            if (kt + 1 >= nkt) break;
            // odd step: buffer 1 holds tile kt + 1, set 0 holds tile kt + 2, set 1 is free for tile kt + 3
            if (kt + 3 < nkt) load_tile((kt + 3) * WT_K, ra1, rb1, wv1);
            mma_tile(As1, Bs1);
            if (kt + 2 < nkt) store_tile(As0, Bs0, ra0, rb0, wv0);
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < RT; i++)
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int j = 0; j < NC; j++) sq[i][r] += acc[i][j][r] * acc[i][j][r];
    }
    // rows of sq: 16 lanes (columns) share a row -> butterfly inside the 16-lane group; the four wavefronts hold
    // different columns of the same rows and are added in wavefront order
#pragma unroll
    for (int i = 0; i < RT; i++)
#pragma unroll
        for (int r = 0; r < 4; r++)
            for (int o = 8; o > 0; o >>= 1) sq[i][r] += __shfl_xor(sq[i][r], o, 64);
#pragma unroll
    for (int i = 0; i < AI; i++)
        for (int o = 16; o > 0; o >>= 1) { mean2[i] += __shfl_xor(mean2[i], o, 64); norm2[i] += __shfl_xor(norm2[i], o, 64); }
    if ((lane & 15) == 0)
#pragma unroll
        for (int i = 0; i < RT; i++)
#pragma unroll
            for (int r = 0; r < 4; r++) red[wave * WT_R + 16 * i + (lane >> 4) + 4 * r] = sq[i][r];
    __syncthreads();
    if (!SPLIT) {
        if (lk == 0)
#pragma unroll
            for (int i = 0; i < AI; i++) {
                const int rl = lr + LR * i, row = bm0 + rl;
                if (row < m) {
                    out[row] = mean2[i];
                    double q2 = 0.0;
#pragma unroll
                    for (int wv = 0; wv < NW; wv++) q2 += red[wv * WT_R + rl];        // wavefront order
                    { const double v = fmax(diag[row] - vsq * (norm2[i] - q2), 0.0); out[m + row] = std_out ? sqrt(v) : v; }
                }
            }
        return;
    }
    if (tid < WT_R && bm0 + tid < m) {
        double q2 = 0.0;
#pragma unroll
        for (int wv = 0; wv < NW; wv++) q2 += red[wv * WT_R + tid];                    // wavefront order
        // SPLIT 1: the sums travel between workgroups that are running at the same time, possibly on different XCDs (each
        // with an L2 of its own): stores that go through, loads that do not stop at a cache (as in potrf.hip's hand-overs)
        if (SPLIT == 1) __hip_atomic_store(&part[(size_t)blockIdx.x * m + bm0 + tid], q2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else part[(size_t)blockIdx.x * m + bm0 + tid] = q2;
    }
    if ((SPLIT == 2 ? (int)blockIdx.x == 0 : (int)blockIdx.x == ncb - 1) && lk == 0)     // owner of the last column block: full k range seen
#pragma unroll
        for (int i = 0; i < AI; i++) {
            const int row = bm0 + lr + LR * i;
            if (row < m) {
                if (SPLIT == 1) {
                    __hip_atomic_store(&mn[row], mean2[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&mn[(size_t)m + row], norm2[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else { mn[row] = mean2[i]; mn[(size_t)m + row] = norm2[i]; }
            }
        }
    if (SPLIT == 2) return;                                      // ws_tail_combine_kernel adds the parts
    // last workgroup of this row block: every workgroup counts itself in only after its own stores have been acknowledged
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) s_last = __hip_atomic_fetch_add(&cnt[blockIdx.y], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ncb - 1;
    __syncthreads();
    if (!s_last) return;
    // the counters belong to the stream (launches of a stream are ordered): the last arriver leaves them at zero
    if (tid == 0) __hip_atomic_store(&cnt[blockIdx.y], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid < WT_R && bm0 + tid < m) {
        const int row = bm0 + tid;
        double q2 = 0.0;
        for (int x = 0; x < ncb; x++) q2 += __hip_atomic_load(&part[(size_t)x * m + row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double mu = __hip_atomic_load(&mn[row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double v = fmax(diag[row] - vsq * (__hip_atomic_load(&mn[(size_t)m + row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - q2), 0.0);
        const double sd = std_out ? sqrt(v) : v;
        if (status_dst) {                                        // a host may be polling for these (page-locked memory)
            __hip_atomic_store(&out[row], mu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&out[m + row], sd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else { out[row] = mu; out[m + row] = sd; }
    }
    if (!status_dst) return;
    // status words by the row block whose results go out last (as in ws_tail_rows_kernel): own stores acknowledged, then
    // counted in; cnt[WT_FEW_MAX_RB] is that second counter
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) s_last = __hip_atomic_fetch_add(&cnt[WT_FEW_MAX_RB], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.y - 1;
    __syncthreads();
    if (s_last && tid < 4) __hip_atomic_store(&status_dst[tid], (double)status_src[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (s_last && tid == 0) __hip_atomic_store(&cnt[WT_FEW_MAX_RB], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- the many-row form again, with the tiles brought into LDS by the DMA path (global_load_lds_dwordx4) instead of
// through registers.  Same decomposition as ws_tail_kernel<64, 2, 8>: a workgroup of eight wavefronts owns 64 rows of P and
// the column blocks x and ncb - 1 - x of Q = P C^-T; ws_tail_combine_kernel adds the parts.  What changes is the staging:
//   * a k tile (32 columns) of an operand is an LDS image of 256-byte rows, [row][16 slots][2 doubles]; one DMA
//     instruction moves 4 rows x 256 B (whole cache lines), lane (rr, jj) -> LDS offset lane x 16 B, as the instruction
//     demands, and the bank spread comes from the SOURCE side: slot jj of row r holds the column pair jj ^ (r & 15);
//   * an MFMA operand read is then one ds_read_b128 per lane and TWO matrix steps -- lane (c, q) of step pair t reads
//     slot (4 t + q) ^ c of row c: sixteen rows hit sixteen different slots, no bank conflict -- with the k order
//     inside a tile permuted identically for both operands (k = 2 (4 t + q) + e);
//   * no tile ever sits in VGPRs, so the loads of tile t + 1 are issued before the products of tile t and nothing but
//     the wait at the end of the step depends on them (the register-staged form drained its loads at the top of every
//     step: the compiler places the wait at the first predicated use).
// Columns beyond kend and rows beyond the matrix are read from a block of zeros (the source address is per lane).
// Requires n, ldp, ldc even and 16-byte aligned operands (a pair never straddles the end of a row); other shapes
// keep the register-staged kernel.
typedef __attribute__((address_space(3))) void lds_void_t;
__device__ __attribute__((aligned(16))) double g_tail_zero[2] = {0.0, 0.0};
constexpr int TG_R = 64, TG_C = 128, TG_K = 32, TG_NW = 8, TG_NS = 3;        // three stages: two tiles in flight
constexpr int TG_STAGE = (TG_R + TG_C) * TG_K;                       // doubles per stage: A image then B image
constexpr size_t tg_lds_bytes() { return (size_t)(TG_NS * TG_STAGE + TG_NW * TG_R + 512) * sizeof(double); }
typedef double d2v __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(64 * TG_NW) ws_tail_glds_kernel(int m, int n, const double *__restrict__ P, long long ldp,
                                                                 const double *__restrict__ Ci, long long ldc,
                                                                 const double *__restrict__ w, double *__restrict__ part,
                                                                 double *__restrict__ mn, int *__restrict__ cnt, int dbg) {
    // dbg (tools/tail_bench.py, GPR_TAIL_DBG): 1 no DMA loads, 2 no matrix products, 4 no operand reads from LDS
    extern __shared__ __attribute__((aligned(16))) double tg_smem[];
    double *red = tg_smem + TG_NS * TG_STAGE, *sW = red + TG_NW * TG_R;  // [NW][64] | w, zero padded to 512
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bm0 = blockIdx.y * TG_R;
    const int ncb = (n + TG_C - 1) / TG_C;
    if (cnt && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) *cnt = 0;       // arrival counter of the combine launch
    sW[tid] = tid < n ? w[tid] : 0.0;                                           // n <= 512 (host checks)
    const int c = lane & 15, q = lane >> 4;
    // the six DMA instructions of this wavefront per tile: ids wave, wave + 8, ...; id < 16: rows 4 id .. of the A image,
    // else rows 4 (id - 16) .. of the B image.  Lane (rr = lane >> 4, jj = lane & 15): row 4 id' + rr, column pair jj ^ (row & 15)
    const int rr = lane >> 4, jj = lane & 15;
    double sq[4][4];
    double mean2 = 0.0, norm2 = 0.0;
#pragma unroll
    for (int i = 0; i < 4; i++) { sq[i][0] = 0.0; sq[i][1] = 0.0; sq[i][2] = 0.0; sq[i][3] = 0.0; }
    const int nturn = ((int)blockIdx.x * 2 + 1 == ncb) ? 1 : 2;
    // Barriers are raw s_barrier (+ lgkmcnt) and the DMA queue is counted by hand: __syncthreads() would drain it
    // (vmcnt(0)) and with it the tile that is meant to stay in flight across the barrier.
    auto barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); };
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int turn = 0; turn < nturn; turn++) {
        const int cb = turn == 0 ? (int)blockIdx.x : ncb - 1 - (int)blockIdx.x;
        const int bn0 = cb * TG_C;
        const int kend = min(n, bn0 + TG_C);
        const bool last = cb == ncb - 1;
        const int nkt = (kend + TG_K - 1) / TG_K;
        // per-instruction source rows (fixed over the k loop)
        const double *src[6];
        int pk[6];                                                   // first column of this lane's pair inside a tile
        bool rok[6];
#pragma unroll
        for (int u = 0; u < 6; u++) {
            const int id = wave + 8 * u;
            const bool isA = id < 16;
            const int row = 4 * (isA ? id : id - 16) + rr;
            pk[u] = 2 * (jj ^ (row & 15));
            if (isA) { const int g = bm0 + row; rok[u] = g < m; src[u] = P + (long long)min(g, m - 1) * ldp + pk[u]; }
            else { const int g = bn0 + row; rok[u] = g < n; src[u] = Ci + (long long)min(g, n - 1) * ldc + pk[u]; }
        }
        auto issue = [&](int kt) {
            double *stage = tg_smem + (kt % TG_NS) * TG_STAGE;
            const int k0 = kt * TG_K;
            if (dbg & 1) return;
#pragma unroll
            for (int u = 0; u < 6; u++) {
                const int id = wave + 8 * u;
                double *dst = stage + 4 * id * TG_K;                 // the B image follows the A image: row 4 id of the stage
                const double *g = (rok[u] && k0 + pk[u] < kend) ? src[u] + k0 : g_tail_zero;
                __builtin_amdgcn_global_load_lds((const void *)g, (lds_void_t *)dst, 16, 0, 0);
            }
        };
        d4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; i++) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
        // two tiles ahead.  Iteration kt: queue tile kt + 2 (its stage was last read in iteration kt - 1, which every
        // wavefront has left through the barrier), multiply tile kt, wait until this wavefront's pieces of tile kt + 1
        // have landed (all but the six newest DMA instructions), barrier: tile kt + 1 is then complete for everybody.
        issue(0);
        if (nkt > 1) { issue(1); asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        barrier();
        for (int kt = 0; kt < nkt; kt++) {
            if (kt + 2 < nkt) issue(kt + 2);
            const double *sA = tg_smem + (kt % TG_NS) * TG_STAGE, *sB = sA + TG_R * TG_K;
            d2v a[2][4], b[2];
            auto frag = [&](int t, int f) {
                const int slot = 2 * ((4 * t + q) ^ c);
                b[f] = *(const d2v *)(sB + (16 * wave + c) * TG_K + slot);
#pragma unroll
                for (int i = 0; i < 4; i++) a[f][i] = *(const d2v *)(sA + (16 * i + c) * TG_K + slot);
            };
            if (dbg & 4) { b[0] = b[1] = d2v{1.0 + kt, 2.0}; for (int i = 0; i < 4; i++) a[0][i] = a[1][i] = d2v{0.5 * i, 1.5}; }
            else frag(0, 0);
#pragma unroll
            for (int t = 0; t < 4; t++) {
                if (t < 3 && !(dbg & 4)) frag(t + 1, (t + 1) & 1);   // the next operands are on their way during these products
                if (dbg & 2) { for (int i = 0; i < 4; i++) acc[i][0] += a[t & 1][i].x * b[t & 1].y; continue; }
#pragma unroll
                for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t & 1][i].x, b[t & 1].x, acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t & 1][i].y, b[t & 1].y, acc[i], 0, 0, 0);
            }
            if (last) {
                // P w and |p|^2 off the A image: thread (row = tid >> 3, slots 2 g, 2 g + 1), four columns per tile
                const int row = tid >> 3, g = tid & 7, k0 = kt * TG_K;
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const int slot = 2 * g + e;
                    const d2v v = *(const d2v *)(sA + row * TG_K + 2 * slot);
                    const int k = k0 + 2 * (slot ^ (row & 15));
                    mean2 += v.x * sW[k] + v.y * sW[k + 1];
                    norm2 += v.x * v.x + v.y * v.y;
                }
            }
            if (kt + 2 < nkt) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            barrier();
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int r = 0; r < 4; r++) sq[i][r] += acc[i][r] * acc[i][r];
    }
    // rows of sq: the 16 lanes of a q group hold 16 columns of the same rows
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int r = 0; r < 4; r++)
            for (int o = 8; o > 0; o >>= 1) sq[i][r] += __shfl_xor(sq[i][r], o, 64);
    for (int o = 4; o > 0; o >>= 1) { mean2 += __shfl_xor(mean2, o, 64); norm2 += __shfl_xor(norm2, o, 64); }
    if (c == 0)
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int r = 0; r < 4; r++) red[wave * TG_R + 16 * i + q + 4 * r] = sq[i][r];
    barrier();
    if (tid < TG_R && bm0 + tid < m) {
        double q2 = 0.0;
#pragma unroll
        for (int wv = 0; wv < TG_NW; wv++) q2 += red[wv * TG_R + tid];                   // wavefront order
        part[(size_t)blockIdx.x * m + bm0 + tid] = q2;
    }
    if (blockIdx.x == 0 && (tid & 7) == 0) {                        // owner of the last column block: full k range seen
        const int row = bm0 + (tid >> 3);
        if (row < m) { mn[row] = mean2; mn[(size_t)m + row] = norm2; }
    }
}

// ---- the many-row form, third shape: the rows stay, the columns stream.  A workgroup owns TRW_ROWS = 32 rows of P and
// keeps ALL their columns in LDS (32 x 466 doubles = 117 KB; one contiguous DMA copy, P is read exactly once); each of its
// sixteen wavefronts takes column tiles of Q = P C^-T (16 columns each) in pairs (j, ntile - 1 - j) -- k ranges 16 (j + 1)
// and 16 (ntile - j): every pair costs the same, and the triangular structure of C^-1 is used at the granularity of ONE
// column tile instead of a 128-column block (a quarter fewer matrix products than the paired-block forms) -- and streams
// the C^-1 operand straight from L2 into registers, four steps ahead (16-byte pieces: lane (c, q) reads the column pair
// 4 t + q of row 16 j + c and feeds two matrix steps, k order permuted like the A reads).  No tile is shared between
// wavefronts, so the main loop has no barrier at all; the rows are complete inside the workgroup, so there is no
// second launch: the sums of Q^2 meet in LDS in wavefront order, mean / variance (or standard deviation) are written,
// and the last workgroup to finish (a counter) delivers the caller's status words behind the results.
// The A operand needs no swizzle: rows are n = 2 x odd doubles apart (466: 233 pairs), so the sixteen rows a b128 read
// touches fall into sixteen different 16-byte bank groups.  Needs n even, n <= TRW_NMAX, ldp == n (contiguous rows).
constexpr int TRW_ROWS = 32, TRW_NW = 16, TRW_NMAX = 576, TRW_D = 4;
__host__ __device__ constexpr size_t trw_lds_doubles(int n) { return (size_t)TRW_ROWS * n + 128 + TRW_NW * TRW_ROWS + (n + 8) + 2 * TRW_ROWS; }

__global__ void __launch_bounds__(64 * TRW_NW) ws_tail_rows_kernel(int m, int n, const double *__restrict__ P,
                                                                  const double *__restrict__ Ci, long long ldc,
                                                                  const double *__restrict__ w, const double *__restrict__ diag,
                                                                  double *__restrict__ out, double vsq, const int *status_src,
                                                                  double *status_dst, int *done, int std_out) {
    extern __shared__ __attribute__((aligned(16))) double trw_smem[];
    double *sA = trw_smem;                                          // [32][n], then 128 doubles of slack (DMA tail, k tail)
    double *red = sA + (size_t)TRW_ROWS * n + 128;                  // [NW][32]
    double *sW = red + TRW_NW * TRW_ROWS;                           // [n + 8]
    double *sMN = sW + n + 8;                                       // mean [32] | norm [32]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bm0 = blockIdx.x * TRW_ROWS;
    const int rows = min(TRW_ROWS, m - bm0);
    // ---- the row block by LDS DMA, in COLUMN chunks of 128: a wavefront's first tile of a pair is the short one (k < 16 (j + 1)
    // <= 256 for j <= 15), so it may start when the first two chunks of every row have landed, while the other half of
    // the block (coming from HBM when the feature kernel has just written it) is still on its way.  One DMA instruction
    // moves 128 columns of one row (1 KB, contiguous on both sides); the last chunk of a row is shorter and the lanes
    // beyond it are switched off (they would land in the next row).  Wavefront w issues rows w and w + 16, chunk by chunk.
    const int nchunk = (n + 127) / 128;
    {
        for (int k = tid; k < 128; k += 64 * TRW_NW) sA[(size_t)TRW_ROWS * n + k] = 0.0;      // the slack behind the last row
        for (int ch = 0; ch < nchunk; ch++) {
            const int c0 = 128 * ch, np = min(64, (n - c0) / 2);          // 16-byte pieces of this chunk
#pragma unroll
            for (int rr = 0; rr < 2; rr++) {
                const int r = wave + 16 * rr;
                if (lane < np) {
                    const double *g = r < rows ? P + (long long)(bm0 + r) * n + c0 + 2 * lane : g_tail_zero;
                    __builtin_amdgcn_global_load_lds((const void *)g, (lds_void_t *)(sA + (size_t)r * n + c0), 16, 0, 0);
                }
            }
        }
    }
    for (int k = tid; k < n + 8; k += 64 * TRW_NW) sW[k] = k < n ? w[k] : 0.0;
    // the first two chunks of this wavefront's rows (all but the DMA instructions of the later chunks), then everybody's
    if (nchunk > 2) {
        if (nchunk == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // ---- column tiles
    const int c = lane & 15, q = lane >> 4;
    const int ntile = (n + 15) / 16, npair = (ntile + 1) / 2, ntmax = (n + 7) / 8;
    double sq[2][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
    const double *a0p = sA + (size_t)c * n + 2 * q, *a1p = sA + (size_t)(16 + c) * n + 2 * q;
    auto tile = [&](int j) {
        const int col = 16 * j + c;
        const int nts = min(2 * (j + 1), ntmax);                     // steps of 8 columns: k < 16 (j + 1)
        const bool cok = col < n;
        const double *bp = Ci + (long long)min(col, n - 1) * ldc;
        // Every load is unconditional -- the address is clamped into the row, what lies beyond the row's end (or belongs to a
        // column past n) becomes zero by a select on the loaded value WHEN IT IS USED, and the prefetches past the last step repeat it -- so
        // the loop has no divergent branch and no load under a branch: the compiler then counts the loads in flight
        // (s_waitcnt vmcnt(3) in the steady state) instead of draining them at every join, which is what made the ring
        // four deep in name only.  (The pair never straddles n: n is even.)
        auto loadb = [&](int t) -> d2v { return *(const d2v *)(bp + min(8 * t + 2 * q, n - 2)); };     // raw: the select waits until use
        d2v ring[TRW_D];
#pragma unroll
        for (int u = 0; u < TRW_D; u++) {
            ring[u] = loadb(min(u, nts - 1));
            __builtin_amdgcn_sched_barrier(0);                       // issued in ring order: the loop waits for the OLDEST load only
        }
        d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
        // The steps go in groups of TRW_D with no exit inside a group (a step beyond nts multiplies by a zero B: the A read
        // stays inside the block or its slack): an exit between two steps is a place below which the compiler sinks the
        // refills -- they are not needed on the way out -- and the ring is then refilled in a burst at the end of the group.
        for (int t0 = 0; t0 < nts; t0 += TRW_D) {
#pragma unroll
            for (int u = 0; u < TRW_D; u++) {
                const int t = t0 + u;
                const bool ok = cok && t < nts && 8 * t + 2 * q < n;
                const d2v b = d2v{ok ? ring[u].x : 0.0, ok ? ring[u].y : 0.0};
                ring[u] = loadb(min(t + TRW_D, nts - 1));
                const d2v a0 = *(const d2v *)(a0p + 8 * t), a1 = *(const d2v *)(a1p + 8 * t);
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.x, b.x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.x, b.x, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.y, b.y, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.y, b.y, acc1, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; r++) { sq[0][r] += acc0[r] * acc0[r]; sq[1][r] += acc1[r] * acc1[r]; }
    };
    // first the tiles that need the first two column chunks only (the short one of every pair, j <= 15) ...
    const bool early = nchunk > 2;
    auto is_early = [&](int j) { return early && 16 * (j + 1) <= 256; };
    for (int pr = wave; pr < npair; pr += TRW_NW)
        if (is_early(pr)) tile(pr);
    // ... then, once the whole block is there (every wavefront passes here exactly once), all the others.  (The wait as a
    // builtin, not as inline assembly: the compiler has to SEE that no LDS DMA is pending any more -- otherwise it guards
    // every LDS read of the loops below with a full vmcnt(0), which also drains the ring of C^-1 loads.)
    __builtin_amdgcn_s_waitcnt(0x0F70);                              // vmcnt(0)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    for (int pr = wave; pr < npair; pr += TRW_NW) {
        if (!is_early(pr)) tile(pr);
        const int j2 = ntile - 1 - pr;
        if (j2 != pr) tile(j2);                                      // odd number of tiles: the middle one stands alone
    }
    // ---- P w and |p|^2: 32 lanes per row (after the products: the wavefronts that finish their tiles early do it while
    // the others still multiply)
    {
        const int row = tid >> 5, l32 = tid & 31;
        double mean = 0.0, nrm = 0.0;
        const double *ar = sA + (size_t)row * n;
        for (int k = l32; k < n; k += 32) { const double v = ar[k]; mean += v * sW[k]; nrm += v * v; }
        for (int o = 16; o > 0; o >>= 1) { mean += __shfl_xor(mean, o, 64); nrm += __shfl_xor(nrm, o, 64); }
        if (l32 == 0) { sMN[row] = mean; sMN[TRW_ROWS + row] = nrm; }
    }
    // rows of sq: the 16 lanes of a q group hold 16 columns of rows 16 i + q + 4 r
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int r = 0; r < 4; r++)
            for (int o = 8; o > 0; o >>= 1) sq[i][r] += __shfl_xor(sq[i][r], o, 64);
    if (c == 0)
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int r = 0; r < 4; r++) red[wave * TRW_ROWS + 16 * i + q + 4 * r] = sq[i][r];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (tid < rows) {
        const int row = bm0 + tid;
        double q2 = 0.0;
#pragma unroll
        for (int wv = 0; wv < TRW_NW; wv++) q2 += red[wv * TRW_ROWS + tid];              // wavefront order
        const double v = fmax(diag[row] - vsq * (sMN[TRW_ROWS + tid] - q2), 0.0), sd = std_out ? sqrt(v) : v;
        if (status_dst) {
            // results a host may be polling for (page-locked memory): stores that go through to system scope
            __hip_atomic_store(&out[row], sMN[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&out[m + row], sd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else { out[row] = sMN[tid]; out[m + row] = sd; }
    }
    if (!status_dst) return;
    // The last workgroup delivers the status words once every workgroup's results are out: a workgroup waits for the
    // acknowledgement of its own write-through stores (vmcnt) and only then counts itself in -- no cache write-back, which
    // is what a system-scope fence per workgroup would cost (194 of them doubled the kernel's time).
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) s_last = __hip_atomic_fetch_add(done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
    __syncthreads();
    if (s_last && tid < 4) __hip_atomic_store(&status_dst[tid], (double)status_src[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (s_last && tid == 0) __hip_atomic_store(done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch on this stream
}

// second launch of the paired scheme: parts added in order, mean and variance written.  With status_dst the workgroup
// that finishes last (counter) also delivers the caller's four status words (ints -> doubles behind the results) after
// every workgroup has made its results visible system-wide: a host polling those words in page-locked memory may read
// the results as soon as it sees them, and no separate launch is needed for them
__global__ void __launch_bounds__(256) ws_tail_combine_kernel(int m, int npart, const double *__restrict__ part,
                                                              const double *__restrict__ mn, const double *__restrict__ diag,
                                                              double *__restrict__ out, double vsq, const int *status_src,
                                                              double *status_dst, int *done, int std_out) {
    __shared__ int s_last;
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row < m) {
        double q2 = 0.0;
        for (int x = 0; x < npart; x++) q2 += part[(size_t)x * m + row];
        out[row] = mn[row];
        { const double v = fmax(diag[row] - vsq * (mn[(size_t)m + row] - q2), 0.0); out[m + row] = std_out ? sqrt(v) : v; }
    }
    if (!status_dst) return;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(done, 1) == (int)gridDim.x - 1;
    __syncthreads();
    if (s_last && threadIdx.x < 4) status_dst[threadIdx.x] = (double)status_src[threadIdx.x];
}

// ---- few rows (one structure per call: m = 97): the time of the tail is the latency of one workgroup's chain, so the chain is
// made short.  Grid (column slices of 32, row blocks of 32); the eight wavefronts of a workgroup take the k steps of its 2 x 2
// tiles in turn (k step ks belongs to wavefront ks mod 8), request ALL their operands at once -- straight from global memory in
// MFMA operand order, up to 16 k steps per wavefront and batch: one memory latency for the whole product where a staged k loop
// pays one per tile --, multiply, and add their partial Q tiles through LDS in wavefront order before the squares are taken.
// The workgroups of the last slice (full k range) also produce P w and |p|^2.  Row blocks are completed by the last arriver,
// which adds the slices in order, and the status words by the last row block -- the hand-over of ws_tail_kernel<., 1>.
constexpr int TF_NW = 8, TF_KB = 16, TF_R = 32, TF_C = 32;
constexpr size_t tf_lds_bytes() { return (size_t)(TF_NW * 4 * 256 + TF_R * (TF_C + 1) + TF_NW * 2 * 2 * 16) * sizeof(double); }
__global__ void __launch_bounds__(64 * TF_NW) ws_tail_few_kernel(int m, int n, const double *__restrict__ P, long long ldp,
                                                                 const double *__restrict__ Ci, long long ldc,
                                                                 const double *__restrict__ w, const double *__restrict__ diag,
                                                                 double *__restrict__ out, double vsq, double *__restrict__ part,
                                                                 double *__restrict__ mn, int *__restrict__ cnt, int std_out,
                                                                 const int *status_src, double *status_dst) {
    extern __shared__ __attribute__((aligned(16))) double tf_smem[];
    double *red = tf_smem;                                   // [wavefront][tile][r * 64 + lane]
    double *sq = red + TF_NW * 4 * 256;                      // [row][TF_C + 1]
    double *mred = sq + TF_R * (TF_C + 1);                   // [wavefront][mean | norm][row tile][16]
    __shared__ int s_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int nslice = gridDim.x, sl = blockIdx.x;
    const int bm0 = blockIdx.y * TF_R, bn0 = sl * TF_C;
    const int kend = min(n, bn0 + TF_C);                     // C^-1[j][k] = 0 for k > j
    const int nks = (kend + 3) >> 2;
    const bool last = sl == nslice - 1;
    d4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) acc[a][b] = d4{0.0, 0.0, 0.0, 0.0};
    double mean_p[2] = {0.0, 0.0}, norm_p[2] = {0.0, 0.0};
    const double *prow[2], *crow[2];
    bool rok[2], cok[2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int row = bm0 + 16 * t + i, col = bn0 + 16 * t + i;
        rok[t] = row < m; cok[t] = col < n;
        prow[t] = P + (long long)(rok[t] ? row : 0) * ldp;
        crow[t] = Ci + (long long)(cok[t] ? col : 0) * ldc;
    }
    for (int ks0 = wave; ks0 < nks; ks0 += TF_NW * TF_KB) {
        double av[TF_KB][2], bv[TF_KB][2], wv[TF_KB];
        const int kbase = 4 * ks0 + kq;
        // every operand of the batch is requested before the first product.  A k step that exists (wavefront-uniform test)
        // starts inside the range; only the lanes of its upper quarters can lie beyond kend (kend not a multiple of four):
        // they read the last valid element instead and are zeroed afterwards, so no load is predicated per lane.
#pragma unroll
        for (int u = 0; u < TF_KB; u++) {
            if (ks0 + TF_NW * u < nks) {
                const int kc = min(kbase + 4 * TF_NW * u, kend - 1);
#pragma unroll
                for (int t = 0; t < 2; t++) { av[u][t] = prow[t][kc]; bv[u][t] = crow[t][kc]; }
                wv[u] = last ? w[kc] : 0.0;
            } else {
#pragma unroll
                for (int t = 0; t < 2; t++) { av[u][t] = 0.0; bv[u][t] = 0.0; }
                wv[u] = 0.0;
            }
        }
#pragma unroll
        for (int u = 0; u < TF_KB; u++) {
            if (ks0 + TF_NW * u < nks) {                      // wavefront-uniform
                const bool kin = kbase + 4 * TF_NW * u < kend;
#pragma unroll
                for (int t = 0; t < 2; t++) { av[u][t] = (kin && rok[t]) ? av[u][t] : 0.0; bv[u][t] = (kin && cok[t]) ? bv[u][t] : 0.0; }
#pragma unroll
                for (int a = 0; a < 2; a++)
#pragma unroll
                    for (int b = 0; b < 2; b++) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][a], bv[u][b], acc[a][b], 0, 0, 0);
                if (last) {
#pragma unroll
                    for (int a = 0; a < 2; a++) { mean_p[a] = fma(av[u][a], wv[u], mean_p[a]); norm_p[a] = fma(av[u][a], av[u][a], norm_p[a]); }
                }
            }
        }
    }
    // partial tiles -> LDS; P w and |p|^2: the four k quarters of a row sit in lanes i, i + 16, i + 32, i + 48
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 4; r++) red[(wave * 4 + a * 2 + b) * 256 + r * 64 + lane] = acc[a][b][r];
    if (last) {
#pragma unroll
        for (int a = 0; a < 2; a++) {
            mean_p[a] += __shfl_xor(mean_p[a], 16, 64); mean_p[a] += __shfl_xor(mean_p[a], 32, 64);
            norm_p[a] += __shfl_xor(norm_p[a], 16, 64); norm_p[a] += __shfl_xor(norm_p[a], 32, 64);
            if (lane < 16) { mred[((wave * 2 + 0) * 2 + a) * 16 + lane] = mean_p[a]; mred[((wave * 2 + 1) * 2 + a) * 16 + lane] = norm_p[a]; }
        }
    }
    __syncthreads();
    // Q entries: sum over the wavefronts in order, squared, placed by (row, column)
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int e = tid + 512 * h, tile = e >> 8, idx = e & 255, r = idx >> 6, ln = idx & 63;
        double q = 0.0;
#pragma unroll
        for (int wv_ = 0; wv_ < TF_NW; wv_++) q += red[(wv_ * 4 + tile) * 256 + idx];
        const int row = 16 * (tile >> 1) + (ln >> 4) + 4 * r, col = 16 * (tile & 1) + (ln & 15);
        sq[row * (TF_C + 1) + col] = q * q;
    }
    __syncthreads();
    if (tid < TF_R && bm0 + tid < m) {
        double q2 = 0.0;
#pragma unroll
        for (int c = 0; c < TF_C; c++) q2 += sq[tid * (TF_C + 1) + c];
        __hip_atomic_store(&part[(size_t)sl * m + bm0 + tid], q2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (last) {
            double mu = 0.0, nn = 0.0;
            const int a = tid >> 4, ii = tid & 15;
#pragma unroll
            for (int wv_ = 0; wv_ < TF_NW; wv_++) { mu += mred[((wv_ * 2 + 0) * 2 + a) * 16 + ii]; nn += mred[((wv_ * 2 + 1) * 2 + a) * 16 + ii]; }
            __hip_atomic_store(&mn[bm0 + tid], mu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&mn[(size_t)m + bm0 + tid], nn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // last workgroup of this row block: every workgroup counts itself in only after its own stores have been acknowledged
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) s_last = __hip_atomic_fetch_add(&cnt[blockIdx.y], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nslice - 1;
    __syncthreads();
    if (!s_last) return;
    if (tid == 0) __hip_atomic_store(&cnt[blockIdx.y], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid < TF_R && bm0 + tid < m) {
        const int row = bm0 + tid;
        double q2 = 0.0;
        const double mu = __hip_atomic_load(&mn[row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double nn = __hip_atomic_load(&mn[(size_t)m + row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // the slices' sums in order, eight loads in flight at a time (one after the other they were a third of the kernel)
        for (int x0 = 0; x0 < nslice; x0 += 8) {
            double pv[8];
#pragma unroll
            for (int u = 0; u < 8; u++)
                pv[u] = __hip_atomic_load(&part[(size_t)min(x0 + u, nslice - 1) * m + row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int u = 0; u < 8; u++) q2 += (x0 + u < nslice) ? pv[u] : 0.0;
        }
        const double v = fmax(diag[row] - vsq * (nn - q2), 0.0);
        const double sd = std_out ? sqrt(v) : v;
        if (status_dst) {
            __hip_atomic_store(&out[row], mu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&out[m + row], sd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else { out[row] = mu; out[m + row] = sd; }
    }
    if (!status_dst) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) s_last = __hip_atomic_fetch_add(&cnt[WT_FEW_MAX_RB], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.y - 1;
    __syncthreads();
    if (s_last && tid < 4) __hip_atomic_store(&status_dst[tid], (double)status_src[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (s_last && tid == 0) __hip_atomic_store(&cnt[WT_FEW_MAX_RB], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void ws_status_kernel(const int *src, double *dst) { if (threadIdx.x < 4) dst[threadIdx.x] = (double)src[threadIdx.x]; }

// partial[b][j] = sum_{i in row block b} a[i*lda+j] * x[i]; fixed-order second pass sums the blocks
__global__ void gemv_t_partial_kernel(int n, int ncols, const double *a, long long lda, const double *x, double *part,
                                      int rows_per) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ncols) return;
    const int r0 = blockIdx.y * rows_per, r1 = min(n, r0 + rows_per);
    double acc = 0.0;
    for (int i = r0; i < r1; i++) acc += a[(long long)i * lda + j] * x[i];
    part[(long long)blockIdx.y * ncols + j] = acc;
}

// C = alpha sum_z ws[z] + beta C, slices added in z order
__global__ void gemm_splitk_reduce_kernel(int m, int n, int nz, const double *__restrict__ ws, double alpha, double beta,
                                          double *__restrict__ C, long long ldc) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)m * n) return;
    double s = 0.0;
    for (int z = 0; z < nz; z++) s += ws[(size_t)z * m * n + e];
    double *c = C + (e / n) * ldc + (e % n);
    *c = (beta == 0.0) ? alpha * s : alpha * s + beta * *c;
}

// c[i] = alpha * sum_j A[i][j] x[j] + beta c[i]: a wavefront per row, lanes along the row (coalesced), four partial sums per lane
// added in a fixed order.  The products with one right-hand side (K^-1-weighted vectors of the LML gradient, Psi* w, C^-1 u) went through
// the 64 x 64-tile kernel: 95 workgroups walking 379 k tiles with two barriers each, 0.35 ms at N = 6063 for 294 MB
__global__ void __launch_bounds__(256) gemv_n_kernel(int m, int k, double alpha, const double *__restrict__ A, long long lda,
                                                      const double *__restrict__ x, long long ldx, double beta, double *__restrict__ c, long long ldc) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= m) return;
    const double *a = A + (long long)row * lda;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int j = lane;
    for (; j + 192 < k; j += 256) {
        s0 += a[j] * x[(long long)j * ldx]; s1 += a[j + 64] * x[(long long)(j + 64) * ldx];
        s2 += a[j + 128] * x[(long long)(j + 128) * ldx]; s3 += a[j + 192] * x[(long long)(j + 192) * ldx];
    }
    for (; j < k; j += 64) s0 += a[j] * x[(long long)j * ldx];
    double sum = (s0 + s1) + (s2 + s3);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o);
    if (lane == 0) {
        double *dst = c + (long long)row * ldc;
        *dst = (beta == 0.0) ? alpha * sum : alpha * sum + beta * *dst;
    }
}

int launch_gemm(hipStream_t st, int ta, int tb, int m, int n, int k, double alpha, const double *a, long long lda,
                const double *b, long long ldb, double beta, double *c, long long ldc, int lower_only) {
    if (m <= 0 || n <= 0) return GPR_OK;
    if (n == 1 && !ta && !lower_only && k >= 64) {            // one right-hand side: a matrix-vector product
        hipLaunchKernelGGL(gemv_n_kernel, dim3((m + 3) / 4), dim3(256), 0, st, m, k, alpha, a, lda, b, tb ? 1LL : ldb, beta, c, ldc);
        const hipError_t e0 = hipGetLastError();
        if (e0 != hipSuccess) { gpr_set_error("gemv launch failed: %s", hipGetErrorString(e0)); return GPR_ERR_HIP; }
        return GPR_OK;
    }
    if (m >= 256 && n >= 256 && !(lower_only & 8)) {      // enough tiles to fill the chip with the 128x128 kernel
        // (bit 3 of lower_only: the caller wants the 64 x 64 tiles -- narrow updates on a latency-critical chain)
        dim3 g2((n + GN - 1) / GN, (m + GM - 1) / GM), b2(256);
        // tall-skinny A^T B (the F x F Gram matrices of the weight-space fit: a handful of tiles, k in the thousands):
        // slices of k on blockIdx.z, partial tiles to a workspace, summed in slice order
        const int tiles = (int)(g2.x * g2.y);
        static const bool generic_env = getenv("GPR_GEMM_GENERIC") != nullptr;
        const bool fast = lda < (1 << 21) && ldb < (1 << 21) && !generic_env;
        if (ta && !tb && !lower_only && tiles <= 64 && k >= 2048) {
            int nz = std::min(256 / tiles, k / 512);
            const int ksplit = ((k + nz - 1) / nz + GK - 1) / GK * GK;
            nz = (k + ksplit - 1) / ksplit;
            void *wsp;
            int rc = gpr_dev_alloc(&wsp, (size_t)nz * m * n * sizeof(double), st);
            if (rc) return rc;
            g2.z = nz;
            if (fast) hipLaunchKernelGGL((gemm128_kernel<true, false, true>), g2, b2, 0, st, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc, 0, ksplit, (double *)wsp, cyc_rule{0, 0, 0, 0, 0});
            else hipLaunchKernelGGL((gemm128_kernel<true, false>), g2, b2, 0, st, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc, 0, ksplit, (double *)wsp, cyc_rule{0, 0, 0, 0, 0});
            hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)(((long long)m * n + 255) / 256)), dim3(256), 0, st, m, n, nz,
                               (const double *)wsp, alpha, beta, c, ldc);
            const hipError_t e3 = hipGetLastError();
            gpr_dev_free(wsp, st);
            if (e3 != hipSuccess) { gpr_set_error("split-K gemm launch failed: %s", hipGetErrorString(e3)); return GPR_ERR_HIP; }
            return GPR_OK;
        }
#define GPR_GEMM128(TA_, TB_, FAST_) hipLaunchKernelGGL((gemm128_kernel<TA_, TB_, FAST_>), g2, b2, 0, st, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc, lower_only, 0, (double *)nullptr, cyc_rule{0, 0, 0, 0, 0})
        if (!ta && !tb) { if (fast) GPR_GEMM128(false, false, true); else GPR_GEMM128(false, false, false); }
        else if (!ta && tb) { if (fast) GPR_GEMM128(false, true, true); else GPR_GEMM128(false, true, false); }
        else if (ta && !tb) { if (fast) GPR_GEMM128(true, false, true); else GPR_GEMM128(true, false, false); }
        else { if (fast) GPR_GEMM128(true, true, true); else GPR_GEMM128(true, true, false); }
#undef GPR_GEMM128
        hipError_t e2 = hipGetLastError();
        if (e2 != hipSuccess) { gpr_set_error("gemm128 launch failed: %s", hipGetErrorString(e2)); return GPR_ERR_HIP; }
        return GPR_OK;
    }
    dim3 grid((n + BN - 1) / BN, (m + BM - 1) / BM), block(256);
    if (!ta && !tb) hipLaunchKernelGGL((gemm_kernel<false, false>), grid, block, 0, st, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc, lower_only);
    else if (!ta && tb) hipLaunchKernelGGL((gemm_kernel<false, true>), grid, block, 0, st, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc, lower_only);
    else if (ta && !tb) hipLaunchKernelGGL((gemm_kernel<true, false>), grid, block, 0, st, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc, lower_only);
    else hipLaunchKernelGGL((gemm_kernel<true, true>), grid, block, 0, st, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc, lower_only);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { gpr_set_error("gemm launch failed: %s", hipGetErrorString(e)); return GPR_ERR_HIP; }
    return GPR_OK;
}

#define LAUNCH_CHECK(name)                                                                   \
    do { hipError_t _e = hipGetLastError();                                                  \
         if (_e != hipSuccess) { gpr_set_error("%s launch failed: %s", name, hipGetErrorString(_e)); return GPR_ERR_HIP; } } while (0)

}  // namespace

__global__ void mirror_lower_kernel(int n, double *a, long long lda) {       // a[i][j] = a[j][i] for j > i
    __shared__ double t[32][33];
    const int bi = blockIdx.y * 32, bj = blockIdx.x * 32;
    if (bj < bi) return;                                                        // tiles with some j > i only
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) { const int j = bj + r, i = bi + tx; t[r][tx] = (j < n && i < n) ? a[(long long)j * lda + i] : 0.0; }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) { const int i = bi + r, j = bj + tx; if (i < n && j < n && j > i) a[(long long)i * lda + j] = t[tx][r]; }
}

// upper triangle <- transpose of the lower one, in place (a covariance matrix assembled as row slabs that hold the
// lower trapezoid only)
__global__ void __launch_bounds__(256) zero2d_kernel(double *p, long long ld, long long cols) {
    double *row = p + (long long)blockIdx.y * ld;
    for (long long c = (long long)blockIdx.x * 1024 + threadIdx.x; c < min(cols, ((long long)blockIdx.x + 1) * 1024); c += 256) row[c] = 0.0;
}

int gpr_zero_2d(hipStream_t st, double *p, long long ld, long long rows, long long cols) {
    if (rows <= 0 || cols <= 0) return GPR_OK;
    if (ld == cols || rows == 1) { GPR_HIP_TRY(hipMemsetAsync(p, 0, (size_t)(rows == 1 ? cols : rows * cols) * 8, st)); return GPR_OK; }
    if (rows > 65535) { GPR_HIP_TRY(hipMemset2DAsync(p, (size_t)ld * 8, 0, (size_t)cols * 8, (size_t)rows, st)); return GPR_OK; }
    hipLaunchKernelGGL(zero2d_kernel, dim3((unsigned)((cols + 1023) / 1024), (unsigned)rows), dim3(256), 0, st, p, ld, cols);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { gpr_set_error("zero2d launch failed: %s", hipGetErrorString(e)); return GPR_ERR_HIP; }
    return GPR_OK;
}

extern "C" int gpr_mirror_lower(void *stream, int n, double *a, long long lda) {
    GPR_REQUIRE(n >= 0 && (n == 0 || a) && lda >= n, "gpr_mirror_lower: bad arguments");
    if (n < 2) return GPR_OK;
    const unsigned nb = (n + 31) / 32;
    hipLaunchKernelGGL(mirror_lower_kernel, dim3(nb, nb), dim3(256), 0, (hipStream_t)stream, n, a, lda);
    LAUNCH_CHECK("mirror_lower");
    return GPR_OK;
}

// C[m, n] -= A[m, k] B[n, k]^T restricted to the lower triangle of the GLOBAL matrix, where the m rows of C (and A)
// are the block-cyclic rows lb0*nb.. of rank `rank` (cyc_rule above) and column j of C is global column col0 + j.
// The trailing update of the row-distributed Cholesky (distributed.py); nb must be a multiple of 128.
extern "C" int gpr_gemm_nt_cyclic_lower(void *stream, int m, int n, int k, double alpha, const double *a, long long lda,
                                        const double *b, long long ldb, double beta, double *c, long long ldc, int nb,
                                        int nranks, int rank, int lb0, int col0) {
    GPR_REQUIRE(m >= 0 && n >= 0 && k >= 0 && nb > 0 && nb % GM == 0 && nranks >= 1 && rank >= 0 && rank < nranks && lb0 >= 0,
                "gpr_gemm_nt_cyclic_lower: bad arguments (nb=%d must be a multiple of %d)", nb, GM);
    if (m == 0 || n == 0) return GPR_OK;
    GPR_REQUIRE(a && b && c, "gpr_gemm_nt_cyclic_lower: null pointer");
    const int pid = gpr_prof_begin((hipStream_t)stream, GPR_PROF_GEMM);
    dim3 g2((n + GN - 1) / GN, (m + GM - 1) / GM), b2(256);
    if (lda < (1 << 21) && ldb < (1 << 21))
        hipLaunchKernelGGL((gemm128_kernel<false, true, true>), g2, b2, 0, (hipStream_t)stream, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc, 0, 0,
                           (double *)nullptr, cyc_rule{nb, nranks, rank, lb0, col0});
    else
        hipLaunchKernelGGL((gemm128_kernel<false, true>), g2, b2, 0, (hipStream_t)stream, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc, 0, 0,
                           (double *)nullptr, cyc_rule{nb, nranks, rank, lb0, col0});
    gpr_prof_end((hipStream_t)stream, pid);
    LAUNCH_CHECK("gemm128(cyclic)");
    return GPR_OK;
}

extern "C" int gpr_gemm(void *stream, int transa, int transb, int m, int n, int k, double alpha,
                        const double *a, long long lda, const double *b, long long ldb, double beta,
                        double *c, long long ldc) {
    GPR_REQUIRE(m >= 0 && n >= 0 && k >= 0, "gpr_gemm: negative dimension");
    GPR_REQUIRE(m == 0 || n == 0 || (c && (k == 0 || (a && b))), "gpr_gemm: null pointer");
    const int pid = gpr_prof_begin((hipStream_t)stream, GPR_PROF_GEMM);
    const int rc = launch_gemm((hipStream_t)stream, transa, transb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc, 0);
    gpr_prof_end((hipStream_t)stream, pid);
    return rc;
}

// C = X^T X for a lower-triangular X (n x n, zeros above the diagonal), C symmetric and written in full: only the tiles on
// or below the diagonal are computed, their k range starts at their row block (X[k][i] = 0 for k < i) -- a sixth of
// the multiply-adds of the plain product -- and the upper triangle is mirrored.  K^-1 = L^-T L^-1 from X = L^-1.
extern "C" int gpr_xtx_lower(void *stream, int n, const double *x, long long ldx, double *c, long long ldc) {
    GPR_REQUIRE(n >= 0 && (n == 0 || (x && c)) && ldx >= n && ldc >= n, "gpr_xtx_lower: bad arguments");
    if (n == 0) return GPR_OK;
    const int pid = gpr_prof_begin((hipStream_t)stream, GPR_PROF_GEMM);
    const int rc = launch_gemm((hipStream_t)stream, 1, 0, n, n, n, 1.0, x, ldx, x, ldx, 0.0, c, ldc, 1 | 64);
    gpr_prof_end((hipStream_t)stream, pid);
    if (rc) return rc;
    return gpr_mirror_lower(stream, n, c, ldc);
}

// C[n, nrhs] = T B with T an n x n LOWER-triangular matrix (zeros above the diagonal): the k loop of
// each 128-row block stops at the end of its diagonal block.  Used with T = L^-1 for the predictive
// variance, V = L^-1 K_trans^T (gaussianprocess.py:903-907 evaluates the same quantity through K^-1).
extern "C" int gpr_trmm_lower(void *stream, int n, int nrhs, const double *t, long long ldt, const double *b,
                              long long ldb, double *c, long long ldc) {
    GPR_REQUIRE(n >= 0 && nrhs >= 0, "gpr_trmm_lower: negative dimension");
    if (n == 0 || nrhs == 0) return GPR_OK;
    GPR_REQUIRE(t && b && c && ldt >= n && ldb >= nrhs && ldc >= nrhs, "gpr_trmm_lower: bad arguments");
    const int pid = gpr_prof_begin((hipStream_t)stream, GPR_PROF_TRMM);
    const int rc = launch_gemm((hipStream_t)stream, 0, 0, n, nrhs, n, 1.0, t, ldt, b, ldb, 0.0, c, ldc, 2);
    gpr_prof_end((hipStream_t)stream, pid);
    return rc;
}

// (the Cholesky factorisation itself lives in potrf.hip; it calls back into the tiled GEMM through this)
// ---- trailing update of the Cholesky look-ahead as a PERSISTENT kernel that leaves CUs free:  C[lower 128 x 128 tiles] -= P P^T
// (P = the panel(s) just solved, [m x k] row-major, k a multiple of 16).  The tiled GEMM above fills every CU with its
// workgroups, and a diagonal-block kernel of the dependent chain that shares its CU with them runs three times longer
// (its factor wavefront waits behind 64-cycle matrix instructions it cannot pre-empt).  Here `nwg` workgroups of eight
// wavefronts, each too large for a second one to fit beside it (the LDS request), loop over the tiles: they settle on
// nwg different CUs and the chain kernels are dispatched onto the CUs that stay free.
constexpr int SB_T = 128, SB_K = 16, SB_LD = SB_K + 1;
__global__ void __launch_bounds__(512) syrk_bulk_kernel(int m, int k, const double *__restrict__ P, long long ldp,
                                                         double *__restrict__ C, long long ldc) {
    __shared__ double As[2][SB_T * SB_LD];
    __shared__ double Bs[2][SB_T * SB_LD];
    extern __shared__ double sb_pad[];                       // never touched: keeps a second workgroup off the CU
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 64;
    const int nbt = (m + SB_T - 1) / SB_T, ntile = nbt * (nbt + 1) / 2;
    const int lk = tid & 15, lr = tid >> 4;                  // tile loads: k offset, row offset (+ 32 i)
    for (int t = blockIdx.x; t < ntile; t += gridDim.x) {
        int bi = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
        while ((bi + 1) * (bi + 2) / 2 <= t) bi++;
        while (bi * (bi + 1) / 2 > t) bi--;
        const int bj = t - bi * (bi + 1) / 2;
        const int bm0 = bi * SB_T, bn0 = bj * SB_T;
        d4 acc[2][4];
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};
        double ra[4], rb[4];
        // no predicate on any load of this kernel (a load under a lane-dependent branch is waited for at the join, one round trip
        // after the other): rows beyond the panel are clamped to its last row -- their products are never stored
        auto load_tile = [&](int k0) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int gm = min(bm0 + lr + 32 * i, m - 1), gn = min(bn0 + lr + 32 * i, m - 1);
                ra[i] = P[(long long)gm * ldp + k0 + lk];
                rb[i] = P[(long long)gn * ldp + k0 + lk];
            }
        };
        auto store_tile = [&](int buf) {
#pragma unroll
            for (int i = 0; i < 4; i++) { As[buf][(lr + 32 * i) * SB_LD + lk] = ra[i]; Bs[buf][(lr + 32 * i) * SB_LD + lk] = rb[i]; }
        };
        const int nkt = k / SB_K;
        load_tile(0); store_tile(0);
        __syncthreads();
        for (int kt = 0; kt < nkt; kt++) {
            const int buf = kt & 1;
            if (kt + 1 < nkt) load_tile((kt + 1) * SB_K);
#pragma unroll
            for (int ks = 0; ks < SB_K / 4; ks++) {
                const int kr = ks * 4 + (lane >> 4);
                double a[2], b[4];
#pragma unroll
                for (int i = 0; i < 2; i++) a[i] = As[buf][(wm + 16 * i + (lane & 15)) * SB_LD + kr];
#pragma unroll
                for (int j = 0; j < 4; j++) b[j] = Bs[buf][(wn + 16 * j + (lane & 15)) * SB_LD + kr];
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
            }
            if (kt + 1 < nkt) store_tile(buf ^ 1);
            __syncthreads();
        }
        // C -= acc: the 32 reads of C issued together, then the stores (one by one under the predicate they were 32 round trips,
        // longer than the k = 256 product of the tile)
        double cold[2][4][4];
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int gm = min(bm0 + wm + 16 * i + (lane >> 4) + 4 * r, m - 1), gn = min(bn0 + wn + 16 * j + (lane & 15), m - 1);
                    cold[i][j][r] = C[(long long)gm * ldc + gn];
                }
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int gm = bm0 + wm + 16 * i + (lane >> 4) + 4 * r, gn = bn0 + wn + 16 * j + (lane & 15);
                    if (gm < m && gn < m) C[(long long)gm * ldc + gn] = cold[i][j][r] - acc[i][j][r];
                }
    }
}

// C[m x m, lower tiles] -= P P^T on `nwg` CUs at most (see syrk_bulk_kernel)
int gpr_syrk_bulk_internal(hipStream_t st, int m, int k, const double *P, long long ldp, double *C, long long ldc, int nwg) {
    if (m <= 0 || k <= 0) return GPR_OK;
    GPR_REQUIRE(k % SB_K == 0 && nwg >= 1, "gpr_syrk_bulk: k must be a multiple of %d", SB_K);
    const int nbt = (m + SB_T - 1) / SB_T, ntile = nbt * (nbt + 1) / 2;
    constexpr int pad_bytes = 16 * 1024;                      // static 69.6 KB + 16 KB: two of these do not fit 160 KB
    { const int rc_ = gpr_func_lds_once((const void *)syrk_bulk_kernel, pad_bytes); if (rc_) return rc_; }
    hipLaunchKernelGGL(syrk_bulk_kernel, dim3(std::min(ntile, nwg)), dim3(512), pad_bytes, st, m, k, P, ldp, C, ldc);
    LAUNCH_CHECK("syrk_bulk");
    return GPR_OK;
}

int gpr_gemm_internal(hipStream_t st, int ta, int tb, int m, int n, int k, double alpha, const double *a, long long lda,
                      const double *b, long long ldb, double beta, double *c, long long ldc, int lower_only) {
    return launch_gemm(st, ta, tb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc, lower_only);
}

// Blocked triangular solve L X = B (trans 0) / L^T X = B (trans 1), in place.  With `dinv` (from
// gpr_potrf_lower_dinv) the diagonal-block solves are 64 x 64 products; without it they fall back to
// per-column substitution.
// One step of the blocked triangular solve for a handful of right-hand sides (alpha = K^-1 y is one): with two launches
// per 64-row block (diagonal apply, then a GEMM update whose N is the number of right-hand sides) the solve is nothing
// but launch latency.  Here every workgroup forms x_k = W_k b_k (W_k = stored inverse of the diagonal block) itself;
// workgroup 0 stores it to `xs`, workgroup w >= 1 subtracts L_ik x_k from one 64-row slab of the pending right-hand
// side.  b_k is only read in the step, so no workgroup depends on another one of the same launch.
constexpr int TRSV_MAX_RHS = 8;

template <bool TRANS>
__global__ void __launch_bounds__(256) trsv_step_kernel(const double *__restrict__ w, const double *__restrict__ l, long long ldl,
                                                         double *b, long long ldb, double *xs, int nrhs, int k0, int kb, int n) {
    __shared__ double T[NB][NB + 1];
    __shared__ double X[NB][TRSV_MAX_RHS + 1];
    __shared__ double Bk[NB][TRSV_MAX_RHS + 1];
    const int tid = threadIdx.x, c = tid & 63, ty = tid >> 6;
    for (int i = ty; i < NB; i += 4) T[i][c] = w[i * NB + c];
    for (int e = tid; e < NB * nrhs; e += 256) {
        const int r = e / nrhs, j = e - r * nrhs;
        Bk[r][j] = (r < kb) ? b[(long long)(k0 + r) * ldb + j] : 0.0;
    }
    __syncthreads();
    for (int e = tid; e < NB * nrhs; e += 256) {              // x_k = W b_k (forward) or W^T b_k (backward)
        const int r = e % NB, j = e / NB;
        double acc = 0.0;
        for (int q = 0; q < NB; q++) acc += (TRANS ? T[q][r] : T[r][q]) * Bk[q][j];
        X[r][j] = acc;
    }
    __syncthreads();
    if (blockIdx.x == 0) {
        for (int e = tid; e < kb * nrhs; e += 256) { const int r = e / nrhs, j = e - r * nrhs; xs[(long long)(k0 + r) * nrhs + j] = X[r][j]; }
        return;
    }
    // slab of 64 pending rows: below the block (forward) or above it (backward)
    const int s0 = TRANS ? (blockIdx.x - 1) * NB : k0 + kb + (blockIdx.x - 1) * NB;
    const int send = TRANS ? k0 : n;
    for (int i = ty; i < NB; i += 4) {
        double v = 0.0;
        if (!TRANS) { if (s0 + i < send && c < kb) v = l[(long long)(s0 + i) * ldl + k0 + c]; }      // L[s0+i][k0+c]
        else        { if (i < kb && s0 + c < send) v = l[(long long)(k0 + i) * ldl + s0 + c]; }      // L[k0+i][s0+c]
        T[i][c] = v;
    }
    __syncthreads();
    for (int e = tid; e < NB * nrhs; e += 256) {
        const int r = e % NB, j = e / NB;
        if (s0 + r < send) {
            double acc = 0.0;
            for (int q = 0; q < NB; q++) acc += (TRANS ? T[q][r] : T[r][q]) * X[q][j];
            b[(long long)(s0 + r) * ldb + j] -= acc;
        }
    }
}

// The same solve as ONE launch: workgroup `id` owns the id-th 64-row block in the order of the substitution, applies the blocks
// solved before it as their solutions appear, solves its own block with the stored inverse and publishes it.  Publication needs
// no flag and no acknowledgement: the solutions go to a side buffer `xs` that starts out as NaNs (0xFF bytes), written with
// agent-scope stores (the workgroups run on different XCDs), and a consumer polls the very elements it needs until they are
// numbers.  The L tile of the next step and the inverse of the own diagonal block depend on no solution and are requested ahead;
// between two publications sit one poll, the last 64 x 64 update and the diagonal product.  A step of the launch-per-block form
// costs a launch (15 us, 95 of them at N = 6063).  Workgroups only wait for lower ids, which the dispatcher starts first; a wait
// that outlasts the poll budget or finds the abort word set gives up and raises it; its rows of b stay what they were (so do
// those of every later block, which all wait for it) and trsv_fixup_kernel, queued right behind, finishes the solve from there.
constexpr int TRSV_CHAIN_POLLS = 1 << 22;
static int g_trsv_max_polls = TRSV_CHAIN_POLLS;
template <bool TRANS>
__global__ void __launch_bounds__(256) trsv_chain_kernel(const double *__restrict__ dinv, const double *__restrict__ l, long long ldl,
                                                          double *b, long long ldb, int nrhs, int n, double *xs, int *abort_word,
                                                          int max_polls) {
    __shared__ double T[NB][NB + 1];
    __shared__ double X[NB][TRSV_MAX_RHS + 1];
    __shared__ int s_bad;
    const int nblk = gridDim.x, id = blockIdx.x, blk = TRANS ? nblk - 1 - id : id;
    const int k0 = blk * NB, kb = min(NB, n - k0);
    const int tid = threadIdx.x, c = tid & 63, jg = tid >> 6;
    const int lrow = tid >> 2, lcol = (tid & 3) * 16;            // tile loads: 16 consecutive elements of one row per thread
    const int j0 = jg, j1 = jg + 4;
    if (tid == 0) s_bad = 0;
    double acc0 = (c < kb && j0 < nrhs) ? b[(long long)(k0 + c) * ldb + j0] : 0.0;
    double acc1 = (c < kb && j1 < nrhs) ? b[(long long)(k0 + c) * ldb + j1] : 0.0;
    double regs[16], wreg[16];
    auto load_tile = [&](int kk) {                               // forward: L[k0 + i][64 kk + q]; backward: L[64 kk + q][k0 + i]
        const int grow = (TRANS ? kk * NB : k0) + lrow, gcol = (TRANS ? k0 : kk * NB) + lcol;
        const double *src = l + (long long)grow * ldl + gcol;
#pragma unroll
        for (int u = 0; u < 16; u++) regs[u] = (grow < n && gcol + u < n) ? src[u] : 0.0;
    };
#pragma unroll
    for (int u = 0; u < 16; u++) wreg[u] = dinv[(size_t)blk * NB * NB + lrow * NB + lcol + u];
    if (id > 0) load_tile(TRANS ? nblk - 1 : 0);
    __syncthreads();
    for (int step = 0; step < id; step++) {
        const int kk = TRANS ? nblk - 1 - step : step;
#pragma unroll
        for (int u = 0; u < 16; u++) T[lrow][lcol + u] = regs[u];
        for (int e = tid; e < NB * nrhs; e += 256) {            // the solution of block kk, element by element as it lands
            const int q = e / nrhs, j = e - q * nrhs, row = kk * NB + q;
            double v = 0.0;
            if (row < n) {
                int polls = 0;
                v = __hip_atomic_load(&xs[(size_t)row * nrhs + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (v != v) {
                    if (++polls > max_polls || ((polls & 63) == 0 && __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                        s_bad = 1;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                    v = __hip_atomic_load(&xs[(size_t)row * nrhs + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            X[q][j] = v;
        }
        if (step + 1 < id) load_tile(TRANS ? kk - 1 : kk + 1);
        __syncthreads();
        if (s_bad) break;                                        // (uniform: written before the barrier)
        if (j0 < nrhs) {
            double a0 = 0.0, a1 = 0.0;
#pragma unroll 8
            for (int q = 0; q < NB; q++) {
                const double t = TRANS ? T[q][c] : T[c][q];
                a0 = fma(t, X[q][j0], a0);
                if (j1 < nrhs) a1 = fma(t, X[q][j1], a1);
            }
            acc0 -= a0; acc1 -= a1;
        }
        __syncthreads();                                         // T and X are rewritten by the next step
    }
    const bool ok = s_bad == 0;
    // x = W acc (forward) / W^T acc (backward), W the stored inverse of the diagonal block
    if (j0 < nrhs) X[c][j0] = acc0;
    if (j1 < nrhs) X[c][j1] = acc1;
#pragma unroll
    for (int u = 0; u < 16; u++) T[lrow][lcol + u] = wreg[u];
    __syncthreads();
    if (c < kb) {
        double x0 = 0.0, x1 = 0.0;
        for (int q = 0; q < NB; q++) {
            const double t = TRANS ? T[q][c] : T[c][q];
            if (j0 < nrhs) x0 = fma(t, X[q][j0], x0);
            if (j1 < nrhs) x1 = fma(t, X[q][j1], x1);
        }
        // a NaN of the data itself (a factor that is not one) is published as an infinity: "not there yet" stays unambiguous
        // and the consumers end with non-finite numbers instead of waiting
        const double inf = __builtin_inf();
        if (j0 < nrhs && ok) {
            b[(long long)(k0 + c) * ldb + j0] = x0;
            __hip_atomic_store(&xs[(size_t)(k0 + c) * nrhs + j0], x0 == x0 ? x0 : inf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (j1 < nrhs && ok) {
            b[(long long)(k0 + c) * ldb + j1] = x1;
            __hip_atomic_store(&xs[(size_t)(k0 + c) * nrhs + j1], x1 == x1 ? x1 : inf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (!ok && tid == 0) __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Queued behind every trsv_chain_kernel: one workgroup that returns at once unless the abort word is up.  If it is, some block
// of the chain gave up waiting; that block and all later ones have left their rows of b as they were and their part of xs
// as NaNs, the earlier ones are solved and published.  The substitution is finished here block by block, in order, from
// exactly that state -- slower than the chain (one workgroup), but the caller gets the solution, never NaNs, and the count of
// such rescues (`fallbacks`, page-locked host memory) is there for anyone who wants to know (gpr_trsv_chain_fallbacks).
template <bool TRANS>
__global__ void __launch_bounds__(256) trsv_fixup_kernel(const double *__restrict__ dinv, const double *__restrict__ l, long long ldl,
                                                          double *b, long long ldb, int nrhs, int n, double *xs, const int *abort_word,
                                                          int *fallbacks) {
    if (*abort_word == 0) return;
    __shared__ double A[NB][TRSV_MAX_RHS + 1];
    const int tid = threadIdx.x, nblk = (n + NB - 1) / NB;
    for (int id = 0; id < nblk; id++) {
        const int blk = TRANS ? nblk - 1 - id : id, k0 = blk * NB, kb = min(NB, n - k0);
        const double first = __hip_atomic_load(&xs[(size_t)k0 * nrhs], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (first == first) continue;                            // solved and published by the chain
        for (int e = tid; e < NB * nrhs; e += 256) {
            const int r = e / nrhs, j = e - r * nrhs;
            double acc = 0.0;
            if (r < kb) {
                acc = b[(long long)(k0 + r) * ldb + j];
                const int q0 = TRANS ? k0 + kb : 0, q1 = TRANS ? n : k0;    // rows solved before this block
                for (int q = q0; q < q1; q++) {
                    const double lv = TRANS ? l[(long long)q * ldl + k0 + r] : l[(long long)(k0 + r) * ldl + q];
                    acc = fma(-lv, __hip_atomic_load(&xs[(size_t)q * nrhs + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), acc);
                }
            }
            A[r][j] = acc;
        }
        __syncthreads();
        const double *W = dinv + (size_t)blk * NB * NB;
        for (int e = tid; e < NB * nrhs; e += 256) {
            const int r = e / nrhs, j = e - r * nrhs;
            if (r < kb) {
                double x = 0.0;
                for (int q = 0; q < NB; q++) x = fma(TRANS ? W[q * NB + r] : W[r * NB + q], A[q][j], x);
                b[(long long)(k0 + r) * ldb + j] = x;
                __hip_atomic_store(&xs[(size_t)(k0 + r) * nrhs + j], x == x ? x : __builtin_inf(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __threadfence();
        __syncthreads();
    }
    if (tid == 0) __hip_atomic_fetch_add(fallbacks, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// how often trsv_fixup_kernel had to finish a solve (process-wide; page-locked, read by the host without a device call)
static int *g_trsv_fallbacks = nullptr, *g_trsv_fallbacks_dev = nullptr;
static std::mutex g_trsv_mu;
static int trsv_fallback_word(int **dev_ptr) {
    std::lock_guard<std::mutex> lk(g_trsv_mu);
    if (!g_trsv_fallbacks) {
        GPR_HIP_TRY(hipHostMalloc((void **)&g_trsv_fallbacks, sizeof(int), hipHostMallocDefault));
        *g_trsv_fallbacks = 0;
        GPR_HIP_TRY(hipHostGetDevicePointer((void **)&g_trsv_fallbacks_dev, g_trsv_fallbacks, 0));
    }
    *dev_ptr = g_trsv_fallbacks_dev;
    return GPR_OK;
}
extern "C" int gpr_trsv_chain_fallbacks(void) { return g_trsv_fallbacks ? __atomic_load_n(g_trsv_fallbacks, __ATOMIC_ACQUIRE) : 0; }
// tests: polls a block of the one-launch solve may spend on a predecessor before it gives up (<= 0: the default)
extern "C" int gpr_trsv_debug_max_polls(int polls) { g_trsv_max_polls = polls > 0 ? polls : TRSV_CHAIN_POLLS; return GPR_OK; }

__global__ void trsv_copy_kernel(int n, int nrhs, const double *xs, double *b, long long ldb) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n * nrhs) b[(long long)(e / nrhs) * ldb + e % nrhs] = xs[e];
}

extern "C" int gpr_trsm_lower_dinv(void *stream, int trans, int n, int nrhs, const double *l, long long ldl,
                                   const double *dinv, double *b, long long ldb) {
    hipStream_t st = (hipStream_t)stream;
    GPR_REQUIRE(n >= 0 && nrhs >= 0, "gpr_trsm_lower: negative dimension");
    if (n == 0 || nrhs == 0) return GPR_OK;
    GPR_REQUIRE(l && b && ldl >= n && ldb >= nrhs, "gpr_trsm_lower: bad arguments");
    const int nblk = (n + NB - 1) / NB;
    static const bool no_chain = getenv("GPR_TRSV_NO_CHAIN") != nullptr;
    if (dinv && nrhs <= TRSV_MAX_RHS && nblk >= 2 && nblk <= 1024 && !no_chain) {   // few right-hand sides: the whole solve in one launch
        void *p;
        const size_t xs_bytes = (size_t)n * nrhs * sizeof(double);
        int rc = gpr_dev_alloc(&p, xs_bytes + 16, st);
        if (rc) return rc;
        double *xs = (double *)p;
        int *abort_word = (int *)((char *)p + xs_bytes);
        int *fb = nullptr;
        if ((rc = trsv_fallback_word(&fb))) { gpr_dev_free(p, st); return rc; }
        // NaNs in xs: "not published yet"; the 16 bytes behind it hold the abort word and must start as zero -- one fill for
        // both would need two patterns, so the abort word gets its own
        hipError_t me = hipMemsetAsync(xs, 0xFF, xs_bytes, st);
        if (me == hipSuccess) me = hipMemsetAsync(abort_word, 0, sizeof(int), st);
        if (me == hipSuccess) {
            const int polls = g_trsv_max_polls;
            if (trans) {
                hipLaunchKernelGGL(trsv_chain_kernel<true>, dim3(nblk), dim3(256), 0, st, dinv, l, ldl, b, ldb, nrhs, n, xs, abort_word, polls);
                hipLaunchKernelGGL(trsv_fixup_kernel<true>, dim3(1), dim3(256), 0, st, dinv, l, ldl, b, ldb, nrhs, n, xs, abort_word, fb);
            } else {
                hipLaunchKernelGGL(trsv_chain_kernel<false>, dim3(nblk), dim3(256), 0, st, dinv, l, ldl, b, ldb, nrhs, n, xs, abort_word, polls);
                hipLaunchKernelGGL(trsv_fixup_kernel<false>, dim3(1), dim3(256), 0, st, dinv, l, ldl, b, ldb, nrhs, n, xs, abort_word, fb);
            }
        }
        const hipError_t e = me != hipSuccess ? me : hipGetLastError();
        gpr_dev_free(p, st);
        if (e != hipSuccess) { gpr_set_error("trsv_chain launch failed: %s", hipGetErrorString(e)); return GPR_ERR_HIP; }
        return GPR_OK;
    }
    if (dinv && nrhs <= TRSV_MAX_RHS) {              // (GPR_TRSV_NO_CHAIN, one block, very large n) one fused launch per block
        void *p;
        int rc = gpr_dev_alloc(&p, (size_t)n * nrhs * sizeof(double), st);
        if (rc) return rc;
        double *xs = (double *)p;
        for (int ib = 0; ib < nblk; ib++) {
            const int blk = trans ? nblk - 1 - ib : ib;
            const int k0 = blk * NB, kb = (n - k0 < NB) ? n - k0 : NB;
            const int pending = trans ? k0 : n - k0 - kb;
            const dim3 grid(1 + (pending + NB - 1) / NB);
            if (trans) hipLaunchKernelGGL(trsv_step_kernel<true>, grid, dim3(256), 0, st, dinv + (size_t)blk * NB * NB, l, ldl, b, ldb, xs, nrhs, k0, kb, n);
            else hipLaunchKernelGGL(trsv_step_kernel<false>, grid, dim3(256), 0, st, dinv + (size_t)blk * NB * NB, l, ldl, b, ldb, xs, nrhs, k0, kb, n);
        }
        hipLaunchKernelGGL(trsv_copy_kernel, dim3((n * nrhs + 255) / 256), dim3(256), 0, st, n, nrhs, xs, b, ldb);
        hipError_t e = hipGetLastError();
        gpr_dev_free(p, st);
        if (e != hipSuccess) { gpr_set_error("trsv_step launch failed: %s", hipGetErrorString(e)); return GPR_ERR_HIP; }
        return GPR_OK;
    }
    for (int ib = 0; ib < nblk; ib++) {
        const int blk = trans ? nblk - 1 - ib : ib;
        const int k0 = blk * NB, kb = (n - k0 < NB) ? n - k0 : NB;
        const double *lkk = l + (long long)k0 * ldl + k0;
        double *bk = b + (long long)k0 * ldb;
        if (dinv) {
            hipLaunchKernelGGL(tri_apply_kernel, dim3((nrhs + NB - 1) / NB), dim3(256), 0, st, trans ? 2 : 1,
                               dinv + (size_t)blk * NB * NB, bk, ldb, kb, nrhs);
            LAUNCH_CHECK("tri_apply(solve)");
        } else {
            hipLaunchKernelGGL(trsm_diag_kernel, dim3((nrhs + 255) / 256), dim3(256), 0, st, trans, lkk, ldl, kb, bk, ldb, nrhs);
            LAUNCH_CHECK("trsm_diag");
        }
        int rc = GPR_OK;
        if (!trans) {
            const int rows = n - k0 - kb;     // B[k0+kb:, :] -= L[k0+kb:, k0:k0+kb] X_k
            if (rows > 0)
                rc = launch_gemm(st, 0, 0, rows, nrhs, kb, -1.0, l + (long long)(k0 + kb) * ldl + k0, ldl, bk, ldb,
                                 1.0, b + (long long)(k0 + kb) * ldb, ldb, 0);
        } else {
            if (k0 > 0)                        // B[:k0, :] -= L[k0:k0+kb, :k0]^T X_k
                rc = launch_gemm(st, 1, 0, k0, nrhs, kb, -1.0, l + (long long)k0 * ldl, ldl, bk, ldb, 1.0, b, ldb, 0);
        }
        if (rc) return rc;
    }
    return GPR_OK;
}

extern "C" int gpr_trsm_lower(void *stream, int trans, int n, int nrhs, const double *l, long long ldl,
                              double *b, long long ldb) {
    return gpr_trsm_lower_dinv(stream, trans, n, nrhs, l, ldl, nullptr, b, ldb);
}

extern "C" int gpr_add_diag(void *stream, int n, double *a, long long lda, int n_first, double add_first,
                            double add_rest) {
    if (n <= 0) return GPR_OK;
    GPR_REQUIRE(a && lda >= n, "gpr_add_diag: bad arguments");
    hipLaunchKernelGGL(add_diag_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, a, lda, n_first,
                       add_first, add_rest);
    LAUNCH_CHECK("add_diag");
    return GPR_OK;
}

extern "C" int gpr_colsumsq(void *stream, int n, int ncols, const double *v, long long ldv, double *out) {
    hipStream_t st = (hipStream_t)stream;
    if (ncols <= 0) return GPR_OK;
    GPR_REQUIRE(out && (n == 0 || v), "gpr_colsumsq: null pointer");
    const bool few = ncols <= GPR_FEW_COLS;
    const int rows_per = few ? 64 : 512;
    const int nparts = (n + rows_per - 1) / rows_per > 0 ? (n + rows_per - 1) / rows_per : 1;
    void *part;
    int rc = gpr_dev_alloc(&part, (size_t)nparts * ncols * 8, st);
    if (rc) return rc;
    if (few) hipLaunchKernelGGL((colred_few_kernel<64, true>), dim3((ncols + 63) / 64, nparts), dim3(256), 0, st, n, ncols, v, ldv,
                                (const double *)nullptr, (double *)part, rows_per);
    else hipLaunchKernelGGL(colsumsq_partial_kernel, dim3((ncols + 255) / 256, nparts), dim3(256), 0, st, n, ncols, v, ldv,
                            (double *)part, rows_per);
    LAUNCH_CHECK("colsumsq_partial");
    if (few) hipLaunchKernelGGL(colsum_final_few_kernel, dim3((ncols + 63) / 64), dim3(256), 0, st, nparts, ncols, (const double *)part, (long long)ncols, out);
    else hipLaunchKernelGGL(colsum_final_kernel, dim3((ncols + 255) / 256), dim3(256), 0, st, nparts, ncols,
                            (const double *)part, out);
    LAUNCH_CHECK("colsum_final");
    return gpr_dev_free(part, st);
}

// out[j] = |T B[:, j]|^2 with T lower triangular: gpr_trmm_lower followed by gpr_colsumsq, in one kernel when B has few
// columns (a single structure's 1 + 3 n quantities) and T is large enough for the product to matter
extern "C" int gpr_trmm_lower_sumsq(void *stream, int n, int nrhs, const double *t, long long ldt, const double *b,
                                    long long ldb, double *out) {
    hipStream_t st = (hipStream_t)stream;
    GPR_REQUIRE(n >= 0 && nrhs >= 0, "gpr_trmm_lower_sumsq: negative dimension");
    if (nrhs == 0) return GPR_OK;
    GPR_REQUIRE(out, "gpr_trmm_lower_sumsq: null output");
    if (n == 0) { GPR_HIP_TRY(hipMemsetAsync(out, 0, (size_t)nrhs * 8, st)); return GPR_OK; }
    GPR_REQUIRE(t && b && ldt >= n && ldb >= nrhs, "gpr_trmm_lower_sumsq: bad arguments");
    if (nrhs > GPR_FEW_COLS || n < 64) {             // wide (or tiny): the tiled product, then the column reduction
        void *v;
        int rc = gpr_dev_alloc(&v, (size_t)n * nrhs * 8, st);
        if (rc) return rc;
        rc = gpr_trmm_lower(stream, n, nrhs, t, ldt, b, ldb, (double *)v, nrhs);
        if (!rc) rc = gpr_colsumsq(stream, n, nrhs, (const double *)v, nrhs, out);
        const int rf = gpr_dev_free(v, st);
        return rc ? rc : rf;
    }
    const int nb = (n + 15) / 16, nwg = (nb + 1) / 2;
    const int ct = nrhs <= 16 ? 1 : (nrhs <= 32 ? 2 : 4);             // columns per lane; 16 ct columns per blockIdx.y
    const int ngy = (nrhs + 16 * ct - 1) / (16 * ct), ldp = ngy * 16 * ct;
    void *buf;                                                         // [n x ldp] padded copy of B, then [nwg x ldp] partial sums
    int rc = gpr_dev_alloc(&buf, ((size_t)n + nwg) * ldp * 8, st);
    if (rc) return rc;
    double *bp = (double *)buf, *part = bp + (size_t)n * ldp;
    const int pid = gpr_prof_begin(st, GPR_PROF_TRMM);
    const long long npad = (long long)n * ldp;
    hipLaunchKernelGGL(pad_cols_kernel, dim3((unsigned)((npad + 255) / 256)), dim3(256), 0, st, n, nrhs, b, ldb, bp, ldp);
    LAUNCH_CHECK("pad_cols");
    const dim3 grid(nwg, ngy), block(512);
    switch (ct) {
        case 1: hipLaunchKernelGGL((trmm_sumsq_few_kernel<1>), grid, block, 0, st, n, t, ldt, (const double *)bp, ldp, part); break;
        case 2: hipLaunchKernelGGL((trmm_sumsq_few_kernel<2>), grid, block, 0, st, n, t, ldt, (const double *)bp, ldp, part); break;
        default: hipLaunchKernelGGL((trmm_sumsq_few_kernel<4>), grid, block, 0, st, n, t, ldt, (const double *)bp, ldp, part); break;
    }
    LAUNCH_CHECK("trmm_sumsq_few");
    hipLaunchKernelGGL(colsum_final_few_kernel, dim3((nrhs + 63) / 64), dim3(256), 0, st, nwg, nrhs, (const double *)part, (long long)ldp, out);
    LAUNCH_CHECK("colsum_final");
    gpr_prof_end(st, pid);
    return gpr_dev_free(buf, st);
}

extern "C" int gpr_sum_log_diag(void *stream, int n, const double *a, long long lda, double *out) {
    GPR_REQUIRE(out && (n == 0 || a), "gpr_sum_log_diag: null pointer");
    hipLaunchKernelGGL(sum_log_diag_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, n, a, lda, out);
    LAUNCH_CHECK("sum_log_diag");
    return GPR_OK;
}

extern "C" int gpr_frob_dot(void *stream, int m, int n, const double *a, long long lda, const double *b,
                            long long ldb, double *out) {
    hipStream_t st = (hipStream_t)stream;
    GPR_REQUIRE(out != nullptr, "gpr_frob_dot: null output");
    if (m <= 0 || n <= 0) { GPR_HIP_TRY(hipMemsetAsync(out, 0, 8, st)); return GPR_OK; }
    GPR_REQUIRE(a && b && lda >= n && ldb >= n, "gpr_frob_dot: bad arguments");
    const long long total = (long long)m * n;
    int nblk = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
    void *part;
    int rc = gpr_dev_alloc(&part, (size_t)nblk * 8, st);
    if (rc) return rc;
    hipLaunchKernelGGL(frob_partial_kernel, dim3(nblk), dim3(256), 0, st, m, n, a, lda, b, ldb, (double *)part);
    LAUNCH_CHECK("frob_partial");
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, st, nblk, (const double *)part, out);
    LAUNCH_CHECK("sum_partials");
    return gpr_dev_free(part, st);
}

extern "C" int gpr_rowdot(void *stream, int m, int n, const double *a, long long lda, const double *b, long long ldb,
                          double *out) {
    if (m <= 0) return GPR_OK;
    GPR_REQUIRE(out && (n == 0 || (a && b && lda >= n && ldb >= n)), "gpr_rowdot: bad arguments");
    hipLaunchKernelGGL(rowdot_kernel, dim3((m + 3) / 4), dim3(256), 0, (hipStream_t)stream, m, n, a, lda, b, ldb, out);
    LAUNCH_CHECK("rowdot");
    return GPR_OK;
}

extern "C" int gpr_ws_finish(void *stream, int m, int n, const double *P, long long ldp, const double *Q, long long ldq,
                             const double *w, const double *diag, double *out, double vsq) {
    if (m <= 0) return GPR_OK;
    GPR_REQUIRE(out && P && w && ldp >= n && (!Q || (diag && ldq >= n)), "gpr_ws_finish: bad arguments");
    hipLaunchKernelGGL(ws_finish_kernel, dim3((m + 3) / 4), dim3(256), 0, (hipStream_t)stream, m, n, P, ldp, Q, ldq, w, diag, out, vsq);
    LAUNCH_CHECK("ws_finish");
    return GPR_OK;
}

// out = [mean (m) | var (m)] of the weight-space prediction with the factored variance, one launch, Q never stored
extern "C" int gpr_ws_tail(void *stream, int m, int n, const double *P, long long ldp, const double *Cinv, long long ldc,
                           const double *w, const double *diag, double *out, double vsq) {
    return gpr_ws_tail_status(stream, m, n, P, ldp, Cinv, ldc, w, diag, out, vsq, nullptr, nullptr);
}

// same, and afterwards status_dst[0..3] = (double)status_src[0..3] (device ints, e.g. the counters of a neighbour table):
// written strictly after the results, by the kernel that wrote them last where that is possible
extern "C" int gpr_ws_tail_status(void *stream, int m, int n, const double *P, long long ldp, const double *Cinv, long long ldc,
                                  const double *w, const double *diag, double *out, double vsq, const int *status_src,
                                  double *status_dst) {
    return gpr_ws_tail_internal(stream, m, n, P, ldp, Cinv, ldc, w, diag, out, vsq, status_src, status_dst, 0);
}

// std_out: the second half of `out` holds standard deviations (square roots of the clamped variances) instead of variances
// Arrival counters of the tail kernels: a block per (device, stream) -- launches of a stream are ordered and the last
// arriver leaves a counter at zero, so two predictions in flight (two streams) never share one and no launch has to
// clear them.  [0, WT_FEW_MAX_RB): row blocks of the few-rows form, [WT_FEW_MAX_RB]: its status delivery,
// [WT_FEW_MAX_RB + 1]: status delivery of the row-resident form.
static int *tail_stream_counters(hipStream_t st) {
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, int *> counters;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { gpr_set_error("gpr_ws_tail: hipGetDevice failed"); return nullptr; }
    std::lock_guard<std::mutex> lk(mu);
    int *&slot = counters[{dev, st}];
    if (!slot) {
        const size_t bytes = (WT_FEW_MAX_RB + 2) * sizeof(int);
        // cleared on the stream that uses them (a plain hipMemset goes to the null stream, which a non-blocking stream does not
        // wait for: the first launch could count arrivals on top of whatever the memory held, or be cleared half way)
        if (hipMalloc((void **)&slot, bytes) != hipSuccess || hipMemsetAsync(slot, 0, bytes, st) != hipSuccess) {
            gpr_set_error("gpr_ws_tail: arrival counters could not be allocated");
            slot = nullptr;
        }
    }
    return slot;
}

int gpr_ws_tail_internal(void *stream, int m, int n, const double *P, long long ldp, const double *Cinv, long long ldc,
                         const double *w, const double *diag, double *out, double vsq, const int *status_src,
                         double *status_dst, int std_out) {
    hipStream_t st = (hipStream_t)stream;
    GPR_REQUIRE((status_src == nullptr) == (status_dst == nullptr), "gpr_ws_tail_status: status source and destination go together");
    auto status_only = [&]() {
        if (!status_dst) return GPR_OK;
        hipLaunchKernelGGL(ws_status_kernel, dim3(1), dim3(64), 0, st, status_src, status_dst);
        LAUNCH_CHECK("ws_status");
        return GPR_OK;
    };
    if (m <= 0) return status_only();
    GPR_REQUIRE(n > 0 && out && P && Cinv && w && diag && ldp >= n && ldc >= n && vsq > 0.0, "gpr_ws_tail: bad arguments");
    constexpr int R = WT_ROWS;
    { int rc_ = gpr_func_lds_once((const void *)ws_tail_kernel<R, 0>, (int)wt_lds_bytes<R>());
      if (!rc_) rc_ = gpr_func_lds_once((const void *)ws_tail_kernel<R, 1>, (int)wt_lds_bytes<R>());
      if (rc_) return rc_; }
    const int nrb = (m + R - 1) / R, ncb = (n + WT_C - 1) / WT_C;
    // The row-resident kernel needs no exchange between workgroups, so it also serves a handful of rows (one structure:
    // four workgroups) whenever the shape allows it; the split form below stays for the other shapes.
    const bool rows_ok = ncb >= 2 && n % 2 == 0 && n <= TRW_NMAX && ldp == n && ldc % 2 == 0 && ((uintptr_t)P & 15) == 0 &&
                         ((uintptr_t)Cinv & 15) == 0;
    // few rows (one structure: m = 97): the split form spreads a row block's column blocks over workgroups (33 us against 43
    // for four row-resident workgroups); GPR_TAIL_NO_FEW sends them through the row-resident kernel as well
    static const bool no_few = getenv("GPR_TAIL_NO_FEW") != nullptr;
    if (nrb <= WT_FEW_MAX_RB && ncb > 1 && (!rows_ok || !no_few)) {
        // few rows (a single structure): one workgroup per (column block, row block) and a last-arriver reduction; one
        // launch -- arrival counters that belong to the stream, status words by the last row block
        int *cnt = tail_stream_counters(st);
        if (!cnt) return GPR_ERR_HIP;
        // the short-chain form (operands straight from memory, k steps spread over eight wavefronts); GPR_TAIL_FEW_STAGED=1
        // selects the staged k loop of ws_tail_kernel<., 1> (A/B, tests)
        static const bool staged_env = getenv("GPR_TAIL_FEW_STAGED") != nullptr;
        const bool staged = staged_env || nrb > 32;              // from about sixteen structures on the staged loop is ahead
        const int nslice = (n + TF_C - 1) / TF_C, nparts = staged ? ncb : nslice;
        void *wsp;
        const size_t ws_bytes = (size_t)(nparts + 2) * m * sizeof(double);
        int rc = gpr_dev_alloc(&wsp, ws_bytes, st);
        if (rc) return rc;
        double *part = (double *)wsp, *mn = part + (size_t)nparts * m;
        if ((rc = gpr_func_lds_once((const void *)ws_tail_few_kernel, (int)tf_lds_bytes()))) { gpr_dev_free(wsp, st); return rc; }
        const int pid = gpr_prof_begin(st, GPR_PROF_GEMM);
        if (staged)
            hipLaunchKernelGGL((ws_tail_kernel<R, 1>), dim3(ncb, nrb), dim3(256), wt_lds_bytes<R>(), st, m, n, P, ldp, Cinv, ldc, w, diag,
                               out, vsq, part, mn, cnt, std_out, status_src, status_dst);
        else
            hipLaunchKernelGGL(ws_tail_few_kernel, dim3(nslice, (m + TF_R - 1) / TF_R), dim3(64 * TF_NW), tf_lds_bytes(), st, m, n, P, ldp,
                               Cinv, ldc, w, diag, out, vsq, part, mn, cnt, std_out, status_src, status_dst);
        gpr_prof_end(st, pid);
        const hipError_t le = hipGetLastError();
        gpr_dev_free(wsp, st);
        if (le != hipSuccess) { gpr_set_error("ws_tail launch failed: %s", hipGetErrorString(le)); return GPR_ERR_HIP; }
        return GPR_OK;
    }
    static const int tail_form = getenv("GPR_TAIL_FORM") ? atoi(getenv("GPR_TAIL_FORM")) : 0;   // 1: paired blocks (LDS DMA), 2: register-staged
    if (rows_ok && tail_form == 0) {
        // many rows, row-resident form: one launch, P read once, no barrier in the product loop
        const size_t lds = trw_lds_doubles(n) * sizeof(double);
        { const int rc_ = gpr_func_lds_once((const void *)ws_tail_rows_kernel, (int)(trw_lds_doubles(TRW_NMAX) * sizeof(double))); if (rc_) return rc_; }
        int *done = nullptr;
        if (status_dst) {
            done = tail_stream_counters(st);
            if (!done) return GPR_ERR_HIP;
            done += WT_FEW_MAX_RB + 1;
        }
        const int pid = gpr_prof_begin(st, GPR_PROF_GEMM);
        hipLaunchKernelGGL(ws_tail_rows_kernel, dim3((m + TRW_ROWS - 1) / TRW_ROWS), dim3(64 * TRW_NW), lds, st, m, n, P, Cinv, ldc, w,
                           diag, out, vsq, status_src, status_dst, done, std_out);
        gpr_prof_end(st, pid);
        LAUNCH_CHECK("ws_tail_rows");
        return GPR_OK;
    }
    if (ncb >= 2) {
        // many rows: 64-row workgroups take column blocks x and ncb - 1 - x (equal work for every workgroup, half the
        // C^-1 bytes per flop of the 32-row form), their sums are combined by a second small launch
        constexpr int R2 = 64, NW2 = 8;
        constexpr size_t lds2 = wt_lds_bytes<R2, NW2>();
        { int rc_ = gpr_func_lds_once((const void *)ws_tail_kernel<R2, 2, NW2>, (int)lds2);
          if (!rc_) rc_ = gpr_func_lds_once((const void *)ws_tail_glds_kernel, (int)tg_lds_bytes());
          if (rc_) return rc_; }
        // tiles by LDS DMA where the shape allows 16-byte pieces (always, for the degree-2 feature rows: F is even)
        static const bool no_glds = getenv("GPR_TAIL_NO_GLDS") != nullptr;
        const bool glds = !no_glds && n <= 512 && n % 2 == 0 && ldp % 2 == 0 && ldc % 2 == 0 &&
                          ((uintptr_t)P & 15) == 0 && ((uintptr_t)Cinv & 15) == 0;
        const int npart = (ncb + 1) / 2, nrb2 = (m + R2 - 1) / R2;
        void *wsp;
        int rc = gpr_dev_alloc(&wsp, (size_t)(npart + 2) * m * sizeof(double) + 16, st);
        if (rc) return rc;
        double *part = (double *)wsp, *mn = part + (size_t)npart * m;
        // arrival counter of the combine launch: in this call's workspace, zeroed by the tail kernel
        int *done = (int *)(mn + 2 * (size_t)m);
        const int pid2 = gpr_prof_begin(st, GPR_PROF_GEMM);
        static const int tail_dbg = GPR_PHASE_ENV("GPR_TAIL_DBG");     // tools build only (gpr_internal.h); 0 otherwise
        if (glds)
            hipLaunchKernelGGL(ws_tail_glds_kernel, dim3(npart, nrb2), dim3(64 * TG_NW), tg_lds_bytes(), st, m, n, P, ldp, Cinv, ldc, w,
                               part, mn, done, tail_dbg);
        else
            hipLaunchKernelGGL((ws_tail_kernel<R2, 2, NW2>), dim3(npart, nrb2), dim3(64 * NW2), lds2, st, m, n, P, ldp, Cinv, ldc, w, diag,
                               out, vsq, part, mn, done, std_out, (const int *)nullptr, (double *)nullptr);
        hipLaunchKernelGGL(ws_tail_combine_kernel, dim3((m + 255) / 256), dim3(256), 0, st, m, npart, (const double *)part,
                           (const double *)mn, diag, out, vsq, status_src, status_dst, done, std_out);
        gpr_prof_end(st, pid2);
        const hipError_t le = hipGetLastError();
        gpr_dev_free(wsp, st);
        if (le != hipSuccess) { gpr_set_error("ws_tail launch failed: %s", hipGetErrorString(le)); return GPR_ERR_HIP; }
        return GPR_OK;
    }
    const int pid = gpr_prof_begin(st, GPR_PROF_GEMM);
    hipLaunchKernelGGL((ws_tail_kernel<R, 0>), dim3(nrb), dim3(256), wt_lds_bytes<R>(), st, m, n, P, ldp, Cinv, ldc, w, diag, out, vsq,
                       (double *)nullptr, (double *)nullptr, (int *)nullptr, std_out, (const int *)nullptr, (double *)nullptr);
    gpr_prof_end(st, pid);
    LAUNCH_CHECK("ws_tail");
    return status_only();
}

// C[m, n] = alpha A[m, n] B^T with B n x n lower triangular: the k range of every column block stops at its diagonal block
extern "C" int gpr_gemm_nt_lower(void *stream, int m, int n, double alpha, const double *a, long long lda, const double *bl,
                                 long long ldb, double *c, long long ldc) {
    GPR_REQUIRE(m >= 0 && n >= 0, "gpr_gemm_nt_lower: negative dimension");
    if (m == 0 || n == 0) return GPR_OK;
    GPR_REQUIRE(a && bl && c && lda >= n && ldb >= n && ldc >= n, "gpr_gemm_nt_lower: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int pid = gpr_prof_begin(st, GPR_PROF_GEMM);
    const int rc = launch_gemm(st, 0, 1, m, n, n, alpha, a, lda, bl, ldb, 0.0, c, ldc, 4);
    gpr_prof_end(st, pid);
    return rc;
}

// out[j] = sum_i a[i*lda+j] * x[i]  (A^T x for a row-major n x ncols matrix): the predictive mean K_trans alpha
// with K_trans^T stored [N, m] (gaussianprocess.py:880); bandwidth bound, deterministic summation order
extern "C" int gpr_gemv_t(void *stream, int n, int ncols, const double *a, long long lda, const double *x, double *out) {
    hipStream_t st = (hipStream_t)stream;
    if (ncols <= 0) return GPR_OK;
    GPR_REQUIRE(out && (n == 0 || (a && x && lda >= ncols)), "gpr_gemv_t: bad arguments");
    const bool few = ncols <= GPR_FEW_COLS;
    const int rows_per = few ? 64 : 128;
    const int nparts = n > 0 ? (n + rows_per - 1) / rows_per : 1;
    void *part;
    int rc = gpr_dev_alloc(&part, (size_t)nparts * ncols * 8, st);
    if (rc) return rc;
    const int pid = gpr_prof_begin(st, GPR_PROF_GEMV);
    if (few) hipLaunchKernelGGL((colred_few_kernel<64, false>), dim3((ncols + 63) / 64, nparts), dim3(256), 0, st, n, ncols, a, lda, x,
                                (double *)part, rows_per);
    else hipLaunchKernelGGL(gemv_t_partial_kernel, dim3((ncols + 255) / 256, nparts), dim3(256), 0, st, n, ncols, a, lda, x,
                            (double *)part, rows_per);
    LAUNCH_CHECK("gemv_t_partial");
    if (few) hipLaunchKernelGGL(colsum_final_few_kernel, dim3((ncols + 63) / 64), dim3(256), 0, st, nparts, ncols, (const double *)part, (long long)ncols, out);
    else hipLaunchKernelGGL(colsum_final_kernel, dim3((ncols + 255) / 256), dim3(256), 0, st, nparts, ncols,
                            (const double *)part, out);
    LAUNCH_CHECK("colsum_final");
    gpr_prof_end(st, pid);
    return gpr_dev_free(part, st);
}
