// potrf.hip -- blocked right-looking lower Cholesky, FP64, built so that the sequential part (the diagonal blocks)
// is a handful of launches instead of one per 64 columns.  Replaces scipy.linalg.cholesky of
// gaussianprocess.py:175, 297 (K + noise -> L).  Row-major storage.
//
// One outer step (256 columns):
//   potrf_diag256_kernel   one 512-thread workgroup factors the 256 x 256 diagonal block with its lower triangle
//                          resident in registers as 16 x 16 tiles in MFMA accumulator layout (136 tiles over 8
//                          wavefronts).  16 stages: chol16 + inverse of the diagonal tile in one wavefront (lane = row,
//                          pivots and multipliers by v_readlane), panel tiles L_ik = A_ik W_k^T and trailing tiles
//                          A_ij -= L_ik L_jk^T on the matrix cores with operands exchanged through LDS in operand order.
//                          Emits L, and W_k / L_jk once more in MFMA A-operand order for the panel kernel.
//   potrf_panel256_kernel  rows below the block: one workgroup per 16 rows keeps its 16 x 256 slab in the registers of
//                          four wavefronts and runs the same 16 stages (X_k = P_k W_k^T, P_j -= X_k L_jk^T).
//   gemm128_kernel         trailing update A22 -= L21 L21^T (lower tiles), k = 256.
// tri_inv64_kernel then forms the inverse of every 64 x 64 diagonal block of L (`dinv`, used by gpr_trsm_lower_dinv).
#include "gpr_internal.h"

typedef double d4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int DT = 16;                 // base tile
constexpr int DB = 256;                // diagonal block = 16 x 16 tiles
constexpr int NTILE = 136;             // lower tiles of a 16 x 16 tile grid
constexpr int DWAVES = 12;             // wavefronts of the diagonal kernel: 1 factor + 2 idle (same SIMD) + 9 tile waves
constexpr int TILE_DOUBLES = 256;      // one tile in operand order [s = 0..3][lane = 0..63]

// tiles ordered by column descending, row ascending inside a column: the tiles still active at stage k (column > k)
// are a prefix of the order, so dealing tile t to wavefront t % 8 keeps every stage balanced; t = base(j) + (i - j)
__host__ __device__ constexpr int tile_base(int j) { return (15 - j) * (16 - j) / 2; }
__host__ __device__ constexpr int tile_index(int i, int j) { return tile_base(j) + (i - j); }

// ij[t] = i | j << 8: whole words, so that a wave-uniform t is a scalar load (there is no scalar byte load on this chip, and a
// vector load of the table sat in front of the block's own loads, one round trip per tile)
struct tile_map { unsigned char i[NTILE], j[NTILE]; int ij[NTILE]; };
constexpr tile_map make_tile_map() {
    tile_map m{};
    for (int j = 0; j < 16; j++)
        for (int i = j; i < 16; i++) {
            m.i[tile_index(i, j)] = (unsigned char)i; m.j[tile_index(i, j)] = (unsigned char)j; m.ij[tile_index(i, j)] = i | (j << 8);
        }
    return m;
}
__constant__ tile_map c_tiles = make_tile_map();

__device__ __forceinline__ double readlane_d(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// LDS layout of the diagonal kernel
struct diag_lds {
    double lt[16][TILE_DOUBLES];       // L_ik of the current stage in operand order, indexed by tile row i
    double wt[TILE_DOUBLES];           // W_k = L_kk^-1 in operand order
    double dg[16][17];                 // diagonal tile, row-major, for the factor wavefront
};

// Cholesky of a 16 x 16 tile and the inverse of its factor in ONE wavefront, one instruction stream for both:
//   lanes  0..15 hold the rows of A      (v[c] of lane i      = A[i][c]),
//   lanes 16..31 hold the columns of W   (v[r] of lane 16 + c = W[r][c], starting from the identity; W = L^-1),
// and the right-looking step  v[j] *= 1/sqrt(a_jj);  v[k] -= v[j] L[k][j] (k > j)  is at once the elimination of
// column j of A and the forward substitution of every column of W.  The pivot a_jj and the multipliers L[k][j] are
// wave-uniform (v_readlane from lanes j / k).  Out: lanes 0..15 v[c] = L[i][c] for c <= i (garbage above the
// diagonal), lanes 16..31 v[r] = W[r][c] (zero for r < c).  Returns the first bad pivot (1-based) or 0.
__device__ __forceinline__ int chol16_inv(double (&v)[16]) {
    int bad = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const double ajj = readlane_d(v[j], j);
        if (!(ajj > 0.0) && bad == 0) bad = j + 1;
        const double inv = rsqrt(ajj);
        v[j] = v[j] * inv;
#pragma unroll
        for (int k = j + 1; k < 16; k++) v[k] = __builtin_fma(-v[j], readlane_d(v[j], k), v[k]);
    }
    return bad;
}

// ---------------------------------------------------------------------------------------------------------------------
// diagonal block: a = &A[k0][k0], kb <= 256 valid rows / columns (identity padding beyond)
//   wop [16][256]  : W_k in operand order;   lop [136][256] : L_ik in operand order (tile_index(i, k), i > k)
// Wavefront 0 only factors diagonal tiles.  The FP64 VALU and the FP64 matrix core share one datapath, and a
// v_mfma_f64_16x16x4 holds it for 64 cycles, so the factor wavefront must not share its SIMD with matrix-core work:
// wavefronts 4 and 8 (same SIMD as 0 under the round-robin placement) only keep the barriers company, the nine others
// own the 136 tiles (tile t -> tile wave t % 9).  Per stage k three block barriers:  B1 tile (k, k) is in LDS | B2 W_k is in LDS | B3 the L_ik of the stage are in LDS.
// Between B1 and B2 the factor wavefront works on tile (k, k) while the tile wavefronts finish the trailing updates of
// stage k - 1 (tile (k, k) itself was brought up to date right after B3 of stage k - 1), so the sequential factor and
// the matrix-core updates overlap.  The two roles are separate code paths: neither holds the other's registers.
constexpr int TWAVES = DWAVES - 3;                  // 9 tile wavefronts
constexpr int TSLOTS = (NTILE + TWAVES - 1) / TWAVES;   // 16 tiles per wavefront at most

// ---- hand-over of the operand tiles between two kernels that run AT THE SAME TIME (the diagonal block and the first
// rows of its panel, gpr_potrf_lower_dinv): the producer stores tiles with agent-scope atomic stores (they go through
// to where the other XCDs' loads see them, no cache write-back needed), waits for its own stores (vmcnt) before the
// stage's barrier, and one lane then raises a counter; the consumer polls the counter and reads with agent-scope
// atomic loads.  `seq` grows by one per diagonal block for the life of the process; the counter holds seq * 32 + stages done.
__device__ __forceinline__ void coherent_store(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double coherent_load(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void wait_own_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
constexpr int PIPE_MAX_POLLS = 1 << 23;                  // ~10 s: a consumer whose producer never ran gives up instead of hanging the GPU
__device__ int g_pipe_max_polls = PIPE_MAX_POLLS;        // run-time copy (gpr_potrf_debug_max_polls: the tests force a time-out with it)
// ... and says so in abort word `flag[4]`, which every later poll of the same factorisation looks at first: one missing
// producer costs one time-out, not one per kernel.  poll_until: lane-0 helper, true = the counter got there.
__device__ __forceinline__ bool poll_until(const int *ctr, int target, const int *abort_word) {
    int polls = 0;
    const int max_polls = g_pipe_max_polls;
    while ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
        if (++polls >= max_polls || ((polls & 1023) == 0 && __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
            __hip_atomic_store(const_cast<int *>(abort_word), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
        __builtin_amdgcn_s_sleep(3);
    }
    return true;
}

// TRACE: debug instance (tools/potrf_stage_trace.py) that stamps the phases of every stage with the 100 MHz clock
// PUB:   the operand tiles are published stage by stage through `flag` (see above)
__device__ long long *g_diag_trace = nullptr;
template <bool TRACE, bool PUB>
__global__ void __launch_bounds__(64 * DWAVES) potrf_diag256_kernel(double *__restrict__ a, long long lda, int kb, int k0,
                                                             int *info, double *__restrict__ wop,
                                                             double *__restrict__ lop, int *flag, int base,
                                                             const int *wait_ctr, int wait_target) {
    __shared__ diag_lds S;
    auto stamp = [&](int who, int k, int slot) {
        if (TRACE && (threadIdx.x & 63) == 0) g_diag_trace[(who * 16 + k) * 8 + slot] = wall_clock64();
    };
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nt = (kb + DT - 1) / DT;
    const int lr = lane & 15, lq = lane >> 4;
    if (PUB && wait_ctr) {       // the block is still receiving its last update from a kernel running beside this one
        if (tid == 0 && !poll_until(wait_ctr, wait_target, wait_ctr + 3) && info) atomicExch(info, -1);
        __syncthreads();
    }
    if (wave == 0) {
        __builtin_amdgcn_s_setprio(3);
        for (int k = 0; k < nt; k++) {
            __syncthreads();                                 // B1: tile (k, k), row-major, in S.dg
            stamp(0, k, 0);
            double v[16];
            const bool wl = (lane & 48) == 16;               // lanes 16..31: columns of W
#pragma unroll
            for (int c = 0; c < 16; c++) v[c] = wl ? ((c == lr) ? 1.0 : 0.0) : S.dg[lr][c];
            const int bad = chol16_inv(v);
            stamp(0, k, 1);
            if (bad && lane == 0 && info) atomicCAS(info, 0, k0 + k * DT + bad);
            if (wl) {                                        // W[r][c = lr] straight into operand order:
#pragma unroll                                               //   (q, lane') = W[lane' & 15][(lane' >> 4) + 4 q]
                for (int r = 0; r < 16; r++) S.wt[(lr >> 2) * 64 + 16 * (lr & 3) + r] = v[r];
            }
            stamp(0, k, 2);
            __syncthreads();                                 // B2: W_k in S.wt
            stamp(0, k, 3);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (PUB) coherent_store(&wop[(size_t)k * TILE_DOUBLES + q * 64 + lane], S.wt[q * 64 + lane]);
                else wop[(size_t)k * TILE_DOUBLES + q * 64 + lane] = S.wt[q * 64 + lane];
            }
            if (lane < 16) {                                 // L_kk to global, zeros above the diagonal
                const int row = k * DT + lr;
#pragma unroll
                for (int c = 0; c < 16; c++) {
                    const int col = k * DT + c;
                    if (row < kb && col < kb) a[(long long)row * lda + col] = (c <= lr) ? v[c] : 0.0;
                }
            }
            stamp(0, k, 4);
            if (PUB) wait_own_stores();                      // W_k is out before B3 (this wavefront is early there anyway)
            __syncthreads();                                 // B3
            stamp(0, k, 5);
        }
        return;
    }
    if (wave == 4 || wave == 8) {
        // the publisher: after B1 of stage k every store of stage k - 1 has been waited for by the wavefront that issued it
        for (int k = 0; k < nt; k++) {
            __syncthreads();
            if (PUB && wave == 4 && lane == 0 && k > 0) __hip_atomic_store(flag, base + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads(); __syncthreads();
        }
        if (PUB && wave == 4 && lane == 0) __hip_atomic_store(flag, base + nt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const int tw = wave - 1 - (wave > 4) - (wave > 8);      // 0..8
    __builtin_amdgcn_s_setprio(2);
    // ---- tile wavefronts.  Register tiles, transposed storage: reg r of lane l = A[16 i + (l & 15)][16 j + (l >> 4) + 4 r]
    d4 acc[TSLOTS];
    int tij[TSLOTS];                                         // i | j << 8, or -1
#pragma unroll
    for (int s = 0; s < TSLOTS; s++) {
        const int t = s * TWAVES + tw;
        int v = -1;
        if (t < NTILE) { v = c_tiles.ij[t]; if ((v & 255) >= nt) v = -1; }
        tij[s] = v;
    }
    // the block's 64 loads per lane carry no predicate (a load under a lane-dependent branch is waited for at the join: 64 round
    // trips one after the other, a quarter of this kernel's time): addresses are clamped into the block, padding selected afterwards
#pragma unroll
    for (int s = 0; s < TSLOTS; s++) {
        const int v = max(tij[s], 0), row = min((v & 255) * DT + lr, kb - 1);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int col = min((v >> 8) * DT + lq + 4 * r, kb - 1);
            const double *src = (const double *)((const char *)a + (size_t)((unsigned)(row * (int)lda + col) * 8u));   // 32-bit lane offset
            acc[s][r] = PUB ? coherent_load(src) : *src;
        }
        if (s == TSLOTS / 2 - 1) __builtin_amdgcn_sched_barrier(0);      // two batches of 32: the addresses of 64 do not fit the registers
    }
    // tij[s] is read through an opaque move inside the stage loop: otherwise every address derived from it (LDS tile
    // bases, global rows, operand-order slots; 16 slots of each) is hoisted out of the loop and spilled to scratch
    auto slot = [&](int s) { int v = __builtin_amdgcn_readfirstlane(tij[s]); asm volatile("" : "+s"(v)); return v; };
#pragma unroll
    for (int s = 0; s < TSLOTS; s++) {
        const int v0 = slot(s), v = max(v0, 0), row = (v & 255) * DT + lr;       // (opaque: recomputed, not kept across the loads)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int col = (v >> 8) * DT + lq + 4 * r;
            if (v0 < 0) acc[s][r] = 0.0;
            else if (!(row < kb && col < kb)) acc[s][r] = (row == col) ? 1.0 : 0.0;
        }
    }
    auto hand_over_diag = [&](int k) {                       // the owner of tile (k, k) puts it into S.dg
#pragma unroll
        for (int s = 0; s < TSLOTS; s++)
            if (slot(s) == (k | (k << 8))) {
#pragma unroll
                for (int r = 0; r < 4; r++) S.dg[lr][lq + 4 * r] = acc[s][r];
            }
    };
    auto update = [&](int s, int v) {                        // S_ij -= L_jk L_ik^T from the stage-k tiles in S.lt
        const int ti = v & 255, tj = v >> 8;
#pragma unroll
        for (int q = 0; q < 4; q++)
            acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(-S.lt[tj][q * 64 + lane], S.lt[ti][q * 64 + lane], acc[s], 0, 0, 0);
    };
    hand_over_diag(0);
    for (int k = 0; k < nt; k++) {
        if (tw == 0) stamp(1, k, 0);
        if (PUB) wait_own_stores();                          // the L tiles of stage k - 1 (issued a trailing update ago)
        __syncthreads();                                     // B1
        if (tw == 0) stamp(1, k, 1);
        if (k > 0) {                                         // rest of the trailing update of stage k - 1
#pragma unroll
            for (int s = 0; s < TSLOTS; s++) {
                const int v = slot(s);
                if (v >= 0 && (v >> 8) > k - 1 && v != (k | (k << 8))) update(s, v);
            }
        }
        if (tw == 0) stamp(1, k, 2);
        __syncthreads();                                     // B2
        if (tw == 0) stamp(1, k, 3);
        // panel tiles (i, k), i > k:  X^T = W A_ik^T  (A operand W from LDS, B operand = the register tile)
#pragma unroll
        for (int s = 0; s < TSLOTS; s++) {
            const int v = slot(s);
            if (v >= 0 && (v >> 8) == k && (v & 255) > k) {
                const int ti = v & 255;
                d4 d = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int q = 0; q < 4; q++) d = __builtin_amdgcn_mfma_f64_16x16x4f64(S.wt[q * 64 + lane], acc[s][q], d, 0, 0, 0);
                acc[s] = d;
                const int row = ti * DT + lr;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    S.lt[ti][r * 64 + lane] = d[r];          // register tile == operand order
                    if (PUB) coherent_store(&lop[(size_t)tile_index(ti, k) * TILE_DOUBLES + r * 64 + lane], d[r]);
                    else lop[(size_t)tile_index(ti, k) * TILE_DOUBLES + r * 64 + lane] = d[r];
                    const int col = k * DT + lq + 4 * r;
                    if (row < kb && col < kb) a[(long long)row * lda + col] = d[r];
                }
            }
        }
        if (tw == 0) stamp(1, k, 4);
        __syncthreads();                                     // B3
        if (tw == 0) stamp(1, k, 5);
        if (k + 1 < nt) {                                    // next diagonal tile first, so that its factor can start
#pragma unroll
            for (int s = 0; s < TSLOTS; s++) {
                const int v = slot(s);
                if (v == ((k + 1) | ((k + 1) << 8))) update(s, v);
            }
            hand_over_diag(k + 1);
        }
    }
}

// rows below the diagonal block: p = &A[r0][k0], `rows` rows.  One workgroup (4 wavefronts) per 16 rows: the slab's
// 16 column tiles are dealt to the wavefronts (tile j -> wavefront j % 4), so the 136 tile updates of a slab run four
// abreast; what is sequential per stage is one tile solve X_k = P_k W_k^T in the owner of tile k, handed to the others
// through LDS in B-operand order (= the accumulator layout it is produced in).  The operand tiles of a stage (W_k and
// L_jk, j > k) are staged through LDS, stage k + 1 loading while stage k computes.
// FULL: kb == 256, every bound on the column tiles is a compile-time constant (the guarded form branches per tile).
// PIPE: the diagonal block is being factored by another kernel right now; stage k may read the tiles of stage k once the
// counter says so (see coherent_store above); the prefetch of stage k + 1 waits for that stage, so the slab runs two
// stages behind the factorisation at most.
template <bool FULL, bool PIPE>
__global__ void __launch_bounds__(256) potrf_panel256_kernel(double *__restrict__ p, long long lda, int rows, int kb_,
                                                              const double *__restrict__ wop,
                                                              const double *__restrict__ lop, const int *flag, int base, int *info,
                                                              long long p_batch, long long ops_batch) {
    // blockIdx.y: independent problems (gpr_trtri_lower inverts all diagonal blocks at once), slab / operand tiles
    // p_batch / ops_batch doubles apart
    p += (long long)blockIdx.y * p_batch; wop += (long long)blockIdx.y * ops_batch; lop += (long long)blockIdx.y * ops_batch;
    __shared__ double ops[2][16 * TILE_DOUBLES];             // [buffer][slot = j - k][operand order]; slot 0 = W_k
    __shared__ double xk[2][TILE_DOUBLES];                   // X_k^T of the stage, B-operand order
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r0 = blockIdx.x * DT;
    const int kb = FULL ? DB : kb_;
    const int nt = FULL ? 16 : (kb + DT - 1) / DT;
    const int lr = lane & 15, lq = lane >> 4;
    const bool rok = r0 + lr < rows;
    __builtin_amdgcn_s_setprio(2);                           // a link of the sequential chain: ahead of the bulk updates on its SIMD
    d4 t[4];                                                 // column tiles wave, wave + 4, wave + 8, wave + 12
#pragma unroll
    for (int c = 0; c < 4; c++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int col = (wave + 4 * c) * DT + lq + 4 * r;             // no predicate on the load (see the diagonal kernel)
            const double v = p[(long long)min(r0 + lr, rows - 1) * lda + min(col, kb - 1)];
            t[c][r] = (rok && col < kb) ? v : 0.0;
        }
    }
    // tile u of stage k: u = 0 -> W_k, u >= 1 -> L_{k+u, k}
    auto src = [&](int k, int u) { return u == 0 ? wop + (size_t)k * TILE_DOUBLES : lop + (size_t)tile_index(k + u, k) * TILE_DOUBLES; };
    auto tile_elem = [&](int k, int u) { return PIPE ? coherent_load(src(k, u) + tid) : src(k, u)[tid]; };
    __shared__ int gave_up;
    auto await_stages = [&](int done) {                      // until `done` stages of the diagonal block are published
        if (!PIPE) return;
        if (tid == 0) gave_up = !poll_until(flag, base + done, flag + 4);
        __syncthreads();
    };
    await_stages(nt < 2 ? nt : 2);
    if (PIPE && gave_up) { if (tid == 0 && info) atomicExch(info, -1); return; }
    {   // stage 0's tiles: all sixteen requested before the first is stored (one by one they were sixteen round trips at the head of
        // every workgroup of the panel, a third of a top panel's 40 us)
        double f[16];
#pragma unroll
        for (int u = 0; u < 16; u++) f[u] = tile_elem(0, (FULL || u < nt) ? u : 0);
#pragma unroll
        for (int u = 0; u < 16; u++)
            if (FULL || u < nt) ops[0][u * TILE_DOUBLES + tid] = f[u];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; k++) {
        if (k < nt) {
            if (PIPE && k > 0) {
                await_stages(k + 2 < nt ? k + 2 : nt);
                if (gave_up) { if (tid == 0 && info) atomicExch(info, -1); return; }
            }
            const double *cur = ops[k & 1];
            double pf[15];
#pragma unroll
            for (int u = 0; u < 15 - k; u++)                 // stage k + 1 has nt - k - 1 tiles (a tile beyond them: its W again)
                if (FULL || k + 1 < nt) pf[u] = tile_elem(k + 1, (FULL || k + 1 + u < nt) ? u : 0);
            if (wave == (k & 3)) {                           // X_k^T = W_k P_k^T
                d4 d = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int q = 0; q < 4; q++) d = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[q * 64 + lane], t[k >> 2][q], d, 0, 0, 0);
                t[k >> 2] = d;
#pragma unroll
                for (int q = 0; q < 4; q++) xk[k & 1][q * 64 + lane] = d[q];
            }
            __syncthreads();
            double xb[4];
#pragma unroll
            for (int q = 0; q < 4; q++) xb[q] = xk[k & 1][q * 64 + lane];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int j = wave + 4 * c;
                if (j > k && j < nt) {
                    const double *l = cur + (j - k) * TILE_DOUBLES + lane;
#pragma unroll
                    for (int q = 0; q < 4; q++) t[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(-l[q * 64], xb[q], t[c], 0, 0, 0);
                }
            }
#pragma unroll
            for (int u = 0; u < 15 - k; u++)
                if (k + 1 + u < nt) ops[(k + 1) & 1][u * TILE_DOUBLES + tid] = pf[u];
            __syncthreads();
        }
    }
#pragma unroll
    for (int c = 0; c < 4; c++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int col = (wave + 4 * c) * DT + lq + 4 * r;
            if (rok && col < kb) p[(long long)(r0 + lr) * lda + col] = t[c][r];
        }
    }
}

// inverse of one 64 x 64 lower-triangular diagonal block of L -> dinv block (zero padded); one wavefront per block,
// lane c = column c of the inverse.  Right-looking forward substitution with the 64 partial right-hand sides of a column
// in registers: finishing x[q] updates acc[r] -= T[r][q] x[q] for r > q with independent multiply-adds (T[r][q] is a
// broadcast LDS read), so the 64-step dependency chain carries one multiply per step instead of a dot product.
__global__ void __launch_bounds__(64) tri_inv64_kernel(const double *__restrict__ a, long long lda, int n,
                                                        double *__restrict__ dinv) {
    __shared__ double T[64][65];
    const int b = blockIdx.x, k0 = b * 64, kb = min(64, n - k0), c = threadIdx.x;
    {   // the block's 64 loads per lane without predicates (one by one under a branch: 64 round trips, most of this kernel's 35 us)
        double t[64];
#pragma unroll
        for (int i = 0; i < 64; i++) t[i] = a[(long long)(k0 + min(i, kb - 1)) * lda + k0 + min(c, kb - 1)];
#pragma unroll
        for (int i = 0; i < 64; i++) T[i][c] = (i < kb && c < kb) ? ((c <= i) ? t[i] : 0.0) : ((i == c) ? 1.0 : 0.0);
    }
    __syncthreads();
    double acc[64];
#pragma unroll
    for (int r = 0; r < 64; r++) acc[r] = (r == c) ? 1.0 : 0.0;
#pragma unroll
    for (int q = 0; q < 64; q++) {
        const double x = acc[q] / T[q][q];
        acc[q] = x;
#pragma unroll
        for (int r = q + 1; r < 64; r++) acc[r] = __builtin_fma(-T[r][q], x, acc[r]);
    }
    double *out = dinv + (size_t)b * 4096;
#pragma unroll
    for (int i = 0; i < 64; i++) out[i * 64 + c] = (i < kb && c < kb && i >= c) ? acc[i] : 0.0;
}

// Narrow trailing updates of the look-ahead, C[m x n] -= A[m x 256] B[n x 256]^T (n <= 256; A, B rows of the panel just
// solved).  Both operands are stored with k contiguous, so a lane loads 4 consecutive k of ONE row straight into operand
// registers (lane (r, q) takes k0 + 4 q .. + 3 and feeds them to four consecutive v_mfma_f64_16x16x4: the k order inside
// a 16-chunk is permuted, identically for A and B).  No operand staging through LDS and no k loop to pipeline: a
// workgroup owns a 32 x 32 tile of C and its four wavefronts take a quarter of k each (64 = four chunks, all of whose
// loads are issued before the first MFMA, so the memory latency is paid once), then add their partial tiles through LDS.
// The product is only 256 deep: a tiled GEMM spends its time filling and draining its pipeline (the 64 x 64-tile kernel
// took 50 us for this shape whatever m; one wavefront per 32 x 32 tile running over all of k took 19 us).
typedef double d4u __attribute__((ext_vector_type(4), aligned(8)));
__global__ void __launch_bounds__(256) syrk_narrow_kernel(int m, int n, const double *__restrict__ A, long long lda,
                                                           const double *__restrict__ B, long long ldb,
                                                           double *__restrict__ C, long long ldc, int lower_only, int *done) {
    // done: C is handed to a kernel that is already running and polls this counter (the next diagonal block): stores
    // that go through to it, and one count per workgroup once they are out
    __shared__ double part[4][4][256];                       // [wave][tile][reg * 64 + lane]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int row0 = blockIdx.y * 32, col0 = blockIdx.x * 32;
    if (lower_only && col0 > row0 + 31) {                    // tile strictly above the diagonal of C
        if (done && threadIdx.x == 0) __hip_atomic_fetch_add(done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    __builtin_amdgcn_s_setprio(2);                           // a link of the sequential chain: ahead of the bulk updates on its SIMD
    const int k0 = wave * 64 + 4 * q;
    const double *a0 = A + (long long)min(row0 + r, m - 1) * lda + k0;
    const double *a1 = A + (long long)min(row0 + 16 + r, m - 1) * lda + k0;
    const double *b0 = B + (long long)min(col0 + r, n - 1) * ldb + k0;
    const double *b1 = B + (long long)min(col0 + 16 + r, n - 1) * ldb + k0;
    d4 xa[4][2], xb[4][2];                                   // [chunk][tile]
#pragma unroll
    for (int c = 0; c < 4; c++) {
        xa[c][0] = *(const d4u *)(a0 + 16 * c); xa[c][1] = *(const d4u *)(a1 + 16 * c);
        xb[c][0] = *(const d4u *)(b0 + 16 * c); xb[c][1] = *(const d4u *)(b1 + 16 * c);
    }
    // the four entries of C this lane will update, requested with the operands (under the predicate of the epilogue they were
    // four round trips one after the other at the end of a link of the chain); nothing else writes them while this kernel runs
    double cold[4];
#pragma unroll
    for (int e = 0; e < 4; e++)
        cold[e] = C[(long long)min(row0 + 16 * (wave >> 1) + q + 4 * e, m - 1) * ldc + min(col0 + 16 * (wave & 1) + r, n - 1)];
    d4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int s = 0; s < 4; s++)
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[c][i][s], xb[c][j][s], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 4; e++) part[wave][2 * i + j][e * 64 + lane] = acc[i][j][e];
    __syncthreads();
    const int ti = wave >> 1, tj = wave & 1;                 // wavefront w finishes tile w
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const int gm = row0 + 16 * ti + q + 4 * e, gn = col0 + 16 * tj + r;
        const double v = (part[0][wave][e * 64 + lane] + part[1][wave][e * 64 + lane]) + (part[2][wave][e * 64 + lane] + part[3][wave][e * 64 + lane]);
        if (gm < m && gn < n) {
            double *c = C + (long long)gm * ldc + gn;
            if (done) coherent_store(c, cold[e] - v);
            else *c = cold[e] - v;
        }
    }
    if (done) {
        wait_own_stores();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ void zero_upper2_kernel(int n, double *a, long long lda) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j < n && j > i) a[(long long)i * lda + j] = 0.0;
}

}  // namespace

#include <cstdlib>
#include <mutex>
#define LAUNCH_CHECK(name)                                                                   \
    do { hipError_t _e = hipGetLastError();                                                  \
         if (_e != hipSuccess) { gpr_set_error("%s launch failed: %s", name, hipGetErrorString(_e)); return GPR_ERR_HIP; } } while (0)

// Can two kernels on different streams run at the same time here?  The consumer is enqueued first and polls (bounded) for
// a flag the producer sets.  Under a tool that serialises kernels (rocprofv3 --pmc does), or when the two streams land on
// one hardware queue, the consumer runs alone and gives up: the factorisation then keeps every hand-over at kernel
// boundaries (events) instead of polling.
__global__ void pipe_probe_wait_kernel(const int *flag, int *seen) {
    int polls = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 && polls < 4096) { __builtin_amdgcn_s_sleep(8); polls++; }
    *seen = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
}
__global__ void pipe_probe_set_kernel(int *flag) { __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// second stream + events of the look-ahead: one set per DEVICE (streams, events and the flag words belong to the device
// that was current when they were made).  The factorisation is not re-entrant -- it owns the flag words and the two
// streams -- so calls on one device are serialised: the mutex covers the host-side enqueue, and the event recorded at
// the end of a call is what the next call's stream waits for before its first kernel (a second factorisation enqueued
// from another host thread / on another stream while the first is still RUNNING would otherwise publish its own stage
// counters into flag[0] under the first one's polls).
#include <atomic>
#include <chrono>
struct lookahead_ctx {
    hipStream_t side = nullptr, mid = nullptr;
    hipStream_t aux[3] = {nullptr, nullptr, nullptr};   // more lanes for the independent joins of the triangular inverse
    std::vector<hipEvent_t> ev;
    hipEvent_t end_ev = nullptr;         // end of the last factorisation that used this context
    bool end_valid = false;
    size_t next = 0;
    int *flag = nullptr;                 // [0] stages published by the diagonal-block kernel in flight (see coherent_store)
                                         // [1] workgroups of the updates of diagonal blocks that have finished
                                         // [2], [3] overlap probe   [4] abort word of the polls
    unsigned seq = 0;                    // diagonal blocks launched so far
    unsigned done_total = 0;             // value flag[1] reaches when every update launched so far is through
    bool ready = false;                  // every resource below exists (set last)
    bool failed = false;                 // creation failed once: stay on the no-look-ahead path, do not retry every call
    bool concurrent = false;             // kernels of two streams overlap in time (probed): polled hand-overs are safe
    std::chrono::steady_clock::time_point probed;
    std::mutex mu;                       // one factorisation with look-ahead at a time per device
    void release() {
        for (auto &e : ev) if (e) (void)hipEventDestroy(e);
        ev.clear();
        if (end_ev) (void)hipEventDestroy(end_ev);
        if (flag) (void)hipFree(flag);
        if (mid) { (void)gpr_dev_cache_drop_stream(mid); (void)hipStreamDestroy(mid); }
        if (side) { (void)gpr_dev_cache_drop_stream(side); (void)hipStreamDestroy(side); }
        for (hipStream_t &a : aux) { if (a) { (void)gpr_dev_cache_drop_stream(a); (void)hipStreamDestroy(a); } a = nullptr; }
        end_ev = nullptr; flag = nullptr; mid = side = nullptr; ready = false; end_valid = false;
        (void)hipGetLastError();
    }
    // Can two kernels on different streams run at the same time here?  The consumer is enqueued first and polls (bounded)
    // for a flag the producer sets.  Asked again when the last answer is older than a second (a profiler attached later,
    // another process on the device): the answer costs two tiny launches and a wait for the two private streams.
    int probe() {
        int seen = 0;
        if (hipMemsetAsync(flag + 2, 0, 2 * sizeof(int), mid) != hipSuccess || hipStreamSynchronize(mid) != hipSuccess) return GPR_ERR_HIP;
        hipLaunchKernelGGL(pipe_probe_wait_kernel, dim3(1), dim3(1), 0, mid, (const int *)(flag + 2), flag + 3);
        hipLaunchKernelGGL(pipe_probe_set_kernel, dim3(1), dim3(1), 0, side, flag + 2);
        if (hipStreamSynchronize(mid) != hipSuccess || hipStreamSynchronize(side) != hipSuccess ||
            hipMemcpy(&seen, flag + 3, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return GPR_ERR_HIP;
        concurrent = seen != 0 && !getenv("GPR_POTRF_NO_POLL");
        probed = std::chrono::steady_clock::now();
        return GPR_OK;
    }
    int init() {
        if (ready) return GPR_OK;
        if (failed) return GPR_ERR_HIP;
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);              // lo = least urgent: the bulk updates must not crowd out the chain
        bool ok = false;
        if (!ok) ok = hipStreamCreateWithPriority(&side, hipStreamNonBlocking, lo) == hipSuccess;
        if (!ok) side = nullptr;
        if (ok && hipStreamCreateWithPriority(&mid, hipStreamNonBlocking, hi) != hipSuccess) { mid = nullptr; ok = false; }
        for (hipStream_t &a : aux)
            if (ok && hipStreamCreateWithFlags(&a, hipStreamNonBlocking) != hipSuccess) { a = nullptr; ok = false; }
        if (ok && hipMalloc((void **)&flag, 8 * sizeof(int)) != hipSuccess) { flag = nullptr; ok = false; }
        ok = ok && hipMemset(flag, 0, 8 * sizeof(int)) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&end_ev, hipEventDisableTiming) == hipSuccess;
        if (ok) {
            ev.assign(256, nullptr);
            for (auto &e : ev)
                if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { e = nullptr; ok = false; break; }
        }
        ok = ok && probe() == GPR_OK;
        if (!ok) { release(); failed = true; (void)hipGetLastError(); return GPR_ERR_HIP; }
        ready = true;
        return GPR_OK;
    }
    hipEvent_t take() { return ev[next++ % ev.size()]; }
};
constexpr int LA_MAX_DEVICES = 16;
static lookahead_ctx g_la_dev[LA_MAX_DEVICES];
static std::atomic<int> g_poll_mode{-1};                 // -1: polled hand-overs where the probe allows them, 0: events only

// 0: every hand-over of the look-ahead is an event from now on (what the host asks for after a poll time-out);
// -1: back to the probed answer, and probe again at the next call
extern "C" int gpr_potrf_set_polling(int mode) {
    g_poll_mode.store(mode == 0 ? 0 : -1);
    if (mode != 0)
        for (auto &c : g_la_dev) { std::lock_guard<std::mutex> lk(c.mu); c.probed = std::chrono::steady_clock::time_point(); }
    return GPR_OK;
}
// 1: the next factorisation on the current device may hand tiles over between running kernels; 0: it will not
extern "C" int gpr_potrf_polling(void) {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= LA_MAX_DEVICES) return 0;
    std::lock_guard<std::mutex> lk(g_la_dev[d].mu);
    return g_la_dev[d].ready && g_la_dev[d].concurrent && g_poll_mode.load() != 0;
}
// tests: the poll budget of the hand-overs on the current device (<= 0 restores the default); a tiny budget makes the
// first hand-over give up, which is how the time-out path is exercised
extern "C" int gpr_potrf_debug_max_polls(int polls) {
    const int v = polls > 0 ? polls : PIPE_MAX_POLLS;
    GPR_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_pipe_max_polls), &v, sizeof(v)));
    return GPR_OK;
}

// temporaries of one call, handed back on every way out
struct scratch_guard {
    void *p = nullptr; hipStream_t st = nullptr;
    ~scratch_guard() { if (p) (void)gpr_dev_free(p, st); }
};

static void launch_panel(hipStream_t st, double *p, long long lda, int rows, int kb, const double *wop, const double *lop) {
    if (kb == DB) hipLaunchKernelGGL((potrf_panel256_kernel<true, false>), dim3((rows + DT - 1) / DT), dim3(256), 0, st, p, lda, rows, kb, wop, lop, (const int *)nullptr, 0, (int *)nullptr, 0LL, 0LL);
    else hipLaunchKernelGGL((potrf_panel256_kernel<false, false>), dim3((rows + DT - 1) / DT), dim3(256), 0, st, p, lda, rows, kb, wop, lop, (const int *)nullptr, 0, (int *)nullptr, 0LL, 0LL);
}
// C[m x n] -= A[m x 256] B[n x 256]^T
static int launch_narrow(hipStream_t st, int m, int n, const double *A, const double *B, double *C, long long lda, int lower_only,
                         int *done = nullptr) {
    const dim3 grid((n + 31) / 32, (m + 31) / 32);
    hipLaunchKernelGGL(syrk_narrow_kernel, grid, dim3(256), 0, st, m, n, A, lda, B, lda, C, lda, lower_only, done);
    return (int)(grid.x * grid.y);                           // what `done` grows by
}

// Right-looking, 256 columns per step, with look-ahead.  D_K = diagonal block K; P_K = the panel below it, cut into
// top (the rows of block K + 1) and rest; U_K(J) = update of column block J by P_K, cut the same way.
//   main (the caller's stream)  D_K, D_{K+1}, ... back to back.  D_{K+1} polls a counter until U_K(K + 1) top, its last
//                               update, is through: the chain D -> P top -> U top -> D crosses no stream event.
//   mid  (urgent)               P_K top WHILE D_K runs, two stages behind it (operand tiles handed over by coherent stores
//                               and a counter) -> U_K(K + 1) top (counted) -> P_K rest -> U_K(K + 1) rest
//   side (lowest)               U_K(K + 2) -> U_K(K + 3 ...): the bulk of the flops (gemm128), beside the chain of step K + 1
// Events:
//   mid  U_K(K + 1) top / rest  after  side U_{K-1}(K + 1)
//   side U_K(K + 2)             after  mid U_K(K + 1) rest       (all of P_K is there)
//   the last, unpiped diagonal block and the end of the call wait for everything.
// Operand-tile buffers: three, used in turn (D_{K+2} starts after U_{K+1} top, which follows P_K rest on mid; the third
// is headroom).
// D_K is always enqueued before the P_K top that polls its counter, and depends on nothing enqueued after it, so a
// polling kernel cannot starve its producer; if a producer never runs (a failed launch) the consumer gives up after
// PIPE_MAX_POLLS, reports info = -1 and raises the abort word (flag[4]) that ends every later poll of the call at once.
// N = 6063: 14 ms (round 1: 64-wide steps) -> 5.4 ms (256-wide steps, two streams, tiled narrow GEMMs) -> 4.5 ms
// (syrk_narrow_kernel) -> 4.2 ms (top panel beside the diagonal block, polled hand-overs).  Tried on top of this and
// slower (4.8 ms): the tail of the panel (rows beyond block K + 2) on a fourth stream -- every hop between streams costs
// 15-20 us of event latency, and U_K(K + 1) top then waits for side <- fourth <- mid.
extern "C" int gpr_potrf_lower_dinv(void *stream, int n, double *a, long long lda, int *info_dev, double *dinv) {
    hipStream_t st = (hipStream_t)stream;
    GPR_REQUIRE(n >= 0 && (n == 0 || (a && dinv)) && lda >= n, "gpr_potrf_lower: bad arguments (n=%d lda=%lld)", n, lda);
    if (n == 0) return GPR_OK;
    if (info_dev) GPR_HIP_TRY(hipMemsetAsync(info_dev, 0, sizeof(int), st));
    const size_t ops_doubles = (size_t)(16 + NTILE) * TILE_DOUBLES;
    scratch_guard sg;
    sg.st = st;
    int rc = gpr_dev_alloc(&sg.p, 3 * ops_doubles * sizeof(double), st);
    if (rc) return rc;
    void *scratch = sg.p;
    int devid = 0;
    if (hipGetDevice(&devid) != hipSuccess || devid < 0 || devid >= LA_MAX_DEVICES) { (void)hipGetLastError(); devid = -1; }
    lookahead_ctx &g_la = g_la_dev[devid < 0 ? 0 : devid];
    std::unique_lock<std::mutex> lock(g_la.mu);
    const bool la = devid >= 0 && n > 3 * DB && !getenv("GPR_POTRF_NO_LOOKAHEAD") && g_la.init() == GPR_OK;
    if (!la) lock.unlock();
    if (la && g_poll_mode.load() != 0 && std::chrono::steady_clock::now() - g_la.probed > std::chrono::seconds(1))
        GPR_REQUIRE(g_la.probe() == GPR_OK, "gpr_potrf_lower: the overlap probe failed");
    const bool polled = la && g_la.concurrent && g_poll_mode.load() != 0;
    auto at = [&](int r, int c) { return a + (long long)r * lda + c; };
    auto wait = [&](hipStream_t s, hipEvent_t e) { return e ? hipStreamWaitEvent(s, e, 0) : hipSuccess; };
    auto mark = [&](hipStream_t s) { hipEvent_t e = g_la.take(); return hipEventRecord(e, s) == hipSuccess ? e : nullptr; };
    hipEvent_t bulk_first = nullptr, bulk_last = nullptr, mid_last = nullptr;
    if (la) {                                                // the other streams start behind what the caller queued
        if (g_la.end_valid) GPR_HIP_TRY(hipStreamWaitEvent(st, g_la.end_ev, 0));   // ... and behind the previous factorisation
        // the counters start from zero in every call: a hand-over that gave up in an earlier call (time-out) leaves flag[1]
        // short of done_total for good.  Safe: everything that read them has finished before this point of `st`.
        GPR_HIP_TRY(hipMemsetAsync(g_la.flag, 0, 2 * sizeof(int), st));
        GPR_HIP_TRY(hipMemsetAsync(g_la.flag + 4, 0, sizeof(int), st));      // abort word of the polls
        g_la.seq = 0; g_la.done_total = 0;
        hipEvent_t e0 = mark(st);
        GPR_HIP_TRY(wait(g_la.mid, e0)); GPR_HIP_TRY(wait(g_la.side, e0));
    }
    int step = 0;
    int deferred = 0;                                        // panels K - deferred .. K - 1 have not been applied beyond column block K + 1 yet
    // pairing pays where the bulk dominates (N = 24 250: 148 -> 127 ms); at N = 6063 the chain of diagonal blocks does, and
    // a bulk launch twice as long beside it only lengthens the steps it overlaps (4.2 -> 4.4 ms): pairs from 8192 rows on
    static const bool pair_env = !getenv("GPR_POTRF_NO_PAIR");
    const bool pair_bulk = pair_env && n >= 8192;
    // ... and FOUR panels at a time (k = 1024: 49.5 TF against 43 at k = 512, tools/gemm_k.py) where the bulk is nearly all there is
    static const int group_env = getenv("GPR_POTRF_GROUP") ? atoi(getenv("GPR_POTRF_GROUP")) : 0;
    const int group = !pair_bulk ? 1 : (group_env > 0 ? group_env : (n >= 16384 ? 4 : 2));
    static const int reserve_cus = getenv("GPR_POTRF_RESERVE_CUS") ? atoi(getenv("GPR_POTRF_RESERVE_CUS")) : 24;
    for (int K0 = 0; K0 < n && rc == GPR_OK; K0 += DB, step++) {
        const int KB = (n - K0 < DB) ? n - K0 : DB;
        double *wop = (double *)scratch + (size_t)(step % 3) * ops_doubles, *lop = wop + 16 * TILE_DOUBLES;
        const int c1 = K0 + KB, R = n - c1;                      // first row / column after the block
        if (!la) {
            hipLaunchKernelGGL((potrf_diag256_kernel<false, false>), dim3(1), dim3(64 * DWAVES), 0, st, at(K0, K0), lda, KB, K0, info_dev, wop, lop,
                               (int *)nullptr, 0, (const int *)nullptr, 0);
            if (R <= 0) break;
            launch_panel(st, at(c1, K0), lda, R, KB, wop, lop);
            rc = gpr_gemm_internal(st, 0, 1, R, R, KB, -1.0, at(c1, K0), lda, at(c1, K0), lda, 1.0, at(c1, c1), lda, 1);
            continue;
        }
        if (R <= 0) {                                            // last block: nothing polls, nothing is handed over
            GPR_HIP_TRY(wait(st, mid_last)); GPR_HIP_TRY(wait(st, bulk_last));
            hipLaunchKernelGGL((potrf_diag256_kernel<false, false>), dim3(1), dim3(64 * DWAVES), 0, st, at(K0, K0), lda, KB, K0, info_dev, wop, lop,
                               (int *)nullptr, 0, (const int *)nullptr, 0);
            break;
        }
        double *p = at(c1, K0);
        const int KB1 = (R < DB) ? R : DB, c2 = c1 + KB1, Rrest = n - c2;   // top = rows of block K + 1
        if (polled) {
            const int base = (int)(++g_la.seq * 32u);
            hipLaunchKernelGGL((potrf_diag256_kernel<false, true>), dim3(1), dim3(64 * DWAVES), 0, st, at(K0, K0), lda, KB, K0, info_dev, wop, lop,
                               g_la.flag, base, (const int *)(g_la.flag + 1), (int)g_la.done_total);
            // mid: P top beside D, U top counted into flag[1]
            hipLaunchKernelGGL((potrf_panel256_kernel<true, true>), dim3((KB1 + DT - 1) / DT), dim3(256), 0, g_la.mid, p, lda, KB1, KB, wop, lop,
                               (const int *)g_la.flag, base, info_dev, 0LL, 0LL);
            GPR_HIP_TRY(wait(g_la.mid, bulk_first));
            g_la.done_total += (unsigned)launch_narrow(g_la.mid, KB1, KB1, p, p, at(c1, c1), lda, 1, g_la.flag + 1);
        } else {
            // kernels of different streams do not overlap here (see pipe_probe_wait_kernel): same schedule, every hand-over
            // an event -- D_K, then P top on mid, then U top on the main stream again
            GPR_HIP_TRY(wait(st, mid_last));
            hipLaunchKernelGGL((potrf_diag256_kernel<false, false>), dim3(1), dim3(64 * DWAVES), 0, st, at(K0, K0), lda, KB, K0, info_dev, wop, lop,
                               (int *)nullptr, 0, (const int *)nullptr, 0);
            GPR_HIP_TRY(wait(g_la.mid, mark(st)));
            launch_panel(g_la.mid, p, lda, KB1, KB, wop, lop);
            GPR_HIP_TRY(wait(st, mark(g_la.mid)));
            GPR_HIP_TRY(wait(st, bulk_first));
            launch_narrow(st, KB1, KB1, p, p, at(c1, c1), lda, 1);
            GPR_HIP_TRY(wait(g_la.mid, mark(st)));                // (the rest of the panel reads the top rows; keeps mid behind U top too)
            GPR_HIP_TRY(wait(g_la.mid, bulk_first));
        }
        mid_last = mark(g_la.mid);
        if (Rrest <= 0) { bulk_first = nullptr; continue; }
        launch_panel(g_la.mid, at(c2, K0), lda, Rrest, KB, wop, lop);                          // P rest, U rest
        launch_narrow(g_la.mid, Rrest, KB1, at(c2, K0), p, at(c2, c1), lda, 0);
        mid_last = mark(g_la.mid);
        // side: the bulk
        GPR_HIP_TRY(wait(g_la.side, mid_last));
        const int KB2 = (Rrest < DB) ? Rrest : DB, c3 = c2 + KB2;
        // The bulk of the trailing update is applied for TWO panels at a time: after an even step only the next two column
        // blocks have received panel K (K + 1 on mid, K + 2 here); the odd step that follows brings block K + 2 up to date
        // with both panels and then updates everything beyond with k = 512 -- the tiled GEMM runs at more than twice the
        // rate there than at k = 256, and it has two steps of the chain to hide behind instead of one.
        if (pair_bulk && Rrest >= 12288) {
            // long panels: column block K + 2 takes its pending panels and panel K in ONE tiled product (k up to `group` x 256;
            // the 32 x 32-tile kernel does 22 TF on these shapes -- 30 of 119 ms at N = 24 250 -- the 128 x 128 one 35-49 once the
            // panel has enough 128-row tiles to fill the chip twice over; below that the small tiles win)
            rc = gpr_gemm_internal(g_la.side, 0, 1, Rrest, KB2, KB + deferred * DB, -1.0, at(c2, K0 - deferred * DB), lda,
                                   at(c2, K0 - deferred * DB), lda, 1.0, at(c2, c2), lda, 1);
        } else {
            for (int b = deferred; b >= 1; b--)                                                // U_{K-b}(K + 2), oldest first
                launch_narrow(g_la.side, Rrest, KB2, at(c2, K0 - b * DB), at(c2, K0 - b * DB), at(c2, c2), lda, 1);
            launch_narrow(g_la.side, Rrest, KB2, at(c2, K0), at(c2, K0), at(c2, c2), lda, 1); // U_K(K + 2)
        }
        bulk_last = bulk_first = mark(g_la.side);
        if (c3 < n) {                                                                          // U(K + 3 ...)
            if (deferred + 1 < group && KB == DB) deferred++;                                  // wait for the next panel(s)
            else {
                const int k0 = K0 - deferred * DB, kk = KB + deferred * DB;
                // where the chain of diagonal blocks decides the time (not pair_bulk) the bulk runs as a persistent kernel
                // on all but a few CUs, which stay free for the chain; elsewhere the plain tiled GEMM on every CU
                if (reserve_cus > 0 && !pair_bulk && kk % 16 == 0 && n - c3 >= 1024)
                    rc = gpr_syrk_bulk_internal(g_la.side, n - c3, kk, at(c3, k0), lda, at(c3, c3), lda, std::max(gpr_cu_count() - reserve_cus, 1));
                else
                    rc = gpr_gemm_internal(g_la.side, 0, 1, n - c3, n - c3, kk, -1.0, at(c3, k0), lda, at(c3, k0), lda, 1.0, at(c3, c3), lda, 1);
                bulk_last = mark(g_la.side);
                deferred = 0;
            }
        } else deferred = 0;
    }
    if (la) { GPR_HIP_TRY(wait(st, mid_last)); GPR_HIP_TRY(wait(st, bulk_last)); }
    hipLaunchKernelGGL(tri_inv64_kernel, dim3((n + 63) / 64), dim3(64), 0, st, a, lda, n, dinv);
    if (n > 1) hipLaunchKernelGGL(zero_upper2_kernel, dim3((n + 255) / 256, n), dim3(256), 0, st, n, a, lda);
    if (la) g_la.end_valid = hipEventRecord(g_la.end_ev, st) == hipSuccess;
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { gpr_set_error("potrf launch failed: %s", hipGetErrorString(e)); return GPR_ERR_HIP; }
    return rc;
}

// ---- building blocks of a factorisation whose rows live on several GPUs (gpr_calculator_amd/distributed.py):
// the owner of a diagonal block factors it and hands the operand tiles (W_k and L_jk in MFMA operand order, 311 KB) to
// the other ranks, which apply them to their own rows of the panel.
extern "C" int gpr_potrf_ops_doubles(void) { return (16 + NTILE) * TILE_DOUBLES; }

extern "C" int gpr_potrf_diag_block(void *stream, int kb, double *a, long long lda, int k0, int *info_dev, double *ops) {
    GPR_REQUIRE(kb >= 1 && kb <= DB && a && ops && lda >= kb, "gpr_potrf_diag_block: bad arguments (kb=%d)", kb);
    hipLaunchKernelGGL((potrf_diag256_kernel<false, false>), dim3(1), dim3(64 * DWAVES), 0, (hipStream_t)stream, a, lda, kb, k0, info_dev, ops,
                       ops + 16 * TILE_DOUBLES, (int *)nullptr, 0, (const int *)nullptr, 0);
    LAUNCH_CHECK("potrf_diag256");
    return GPR_OK;
}

extern "C" int gpr_potrf_panel_rows(void *stream, int kb, const double *ops, double *p, long long lda, int rows) {
    GPR_REQUIRE(kb >= 1 && kb <= DB && rows >= 0, "gpr_potrf_panel_rows: bad arguments (kb=%d rows=%d)", kb, rows);
    if (rows == 0) return GPR_OK;
    GPR_REQUIRE(ops && p && lda >= kb, "gpr_potrf_panel_rows: null pointer / short rows");
    const double *wop = ops, *lop = ops + 16 * TILE_DOUBLES;
    launch_panel((hipStream_t)stream, p, lda, rows, kb, wop, lop);
    LAUNCH_CHECK("potrf_panel256");
    return GPR_OK;
}

// debug: one diagonal block with phase stamps; trace_host[2][16][8] (factor wavefront, first tile wavefront) in 10 ns ticks
extern "C" int gpr_potrf_diag_trace(void *stream, int kb, double *a, long long lda, long long *trace_host) {
    hipStream_t st = (hipStream_t)stream;
    long long *tr; double *ops;
    GPR_HIP_TRY(hipMalloc(&tr, 256 * sizeof(long long)));
    GPR_HIP_TRY(hipMalloc(&ops, (size_t)(16 + NTILE) * TILE_DOUBLES * sizeof(double)));
    GPR_HIP_TRY(hipMemset(tr, 0, 256 * sizeof(long long)));
    GPR_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_diag_trace), &tr, sizeof(tr)));
    hipLaunchKernelGGL((potrf_diag256_kernel<true, false>), dim3(1), dim3(64 * DWAVES), 0, st, a, lda, kb, 0, (int *)nullptr, ops, ops + 16 * TILE_DOUBLES, (int *)nullptr, 0, (const int *)nullptr, 0);
    GPR_HIP_TRY(hipStreamSynchronize(st));
    GPR_HIP_TRY(hipMemcpy(trace_host, tr, 256 * sizeof(long long), hipMemcpyDeviceToHost));
    (void)hipFree(tr); (void)hipFree(ops);
    return GPR_OK;
}

// dinv of an already factored matrix (inverse of every 64 x 64 diagonal block of L) and the zeroing of the strict upper
// triangle: the tail of gpr_potrf_lower_dinv, for a factor that was put together from row blocks
extern "C" int gpr_potrf_finish(void *stream, int n, double *l, long long ldl, double *dinv) {
    GPR_REQUIRE(n >= 0 && (n == 0 || l) && ldl >= n, "gpr_potrf_finish: bad arguments");
    if (n == 0) return GPR_OK;
    if (dinv) hipLaunchKernelGGL(tri_inv64_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, l, ldl, n, dinv);
    if (n > 1) hipLaunchKernelGGL(zero_upper2_kernel, dim3((n + 255) / 256, n), dim3(256), 0, (hipStream_t)stream, n, l, ldl);
    LAUNCH_CHECK("potrf_finish");
    return GPR_OK;
}

// X = L^-1 for a lower-triangular L (n x n, e.g. from gpr_potrf_lower): the lower triangle of X and the zeros above the
// diagonal inside the 256 x 256 diagonal blocks are written; the blocks above them are left as they are (callers
// hand over a zeroed matrix).
//   1. all diagonal blocks at once: operand tiles of each block (tri_tiles256_kernel), the panel kernel applied to the
//      identity (I L_kk^-T), stored transposed.
//   2. recursively by halves:  X21 = -X22 (L21 X11)  -- two products on the matrix cores whose three dimensions are all
//      n / 2, n / 4, ...: the top levels hold most of the work and run as large GEMMs (block rows of 256 against all n
//      columns, the first form of this routine, were 2 x n / 128 workgroups per launch: 14 ms at n = 6063).  Both products
//      skip the zero halves of their triangular operands.
__global__ void __launch_bounds__(256) trtri_store_kernel(int n, const double *__restrict__ t, double *__restrict__ x, long long ldx) {
    // diagonal block blockIdx.z: x[i][j] = t[j][i] (t = L_kk^-T, ld DB); zero above the diagonal
    const int K0 = blockIdx.z * DB, kb = min(DB, n - K0);
    const double *tb = t + (size_t)blockIdx.z * DB * DB;
    double *xb = x + (long long)K0 * ldx + K0;
    const int i = blockIdx.x * 16 + (threadIdx.x >> 4), j = blockIdx.y * 16 + (threadIdx.x & 15);
    if (i < kb && j < kb) xb[(long long)i * ldx + j] = (j <= i) ? tb[(size_t)j * DB + i] : 0.0;
}
// operand tiles (as the diagonal kernel emits them) of already factored lower-triangular diagonal blocks (block
// blockIdx.x of an n x n matrix): W_k = inverse of its k-th 16 x 16 diagonal tile (one wavefront per tile: the tiles are
// independent), L_jk copied into operand order
__global__ void __launch_bounds__(1024) tri_tiles256_kernel(const double *__restrict__ l, long long ldl, int n,
                                                             double *__restrict__ ops_all) {
    __shared__ double W[16][16][17];
    const int K0 = blockIdx.x * DB, kb = min(DB, n - K0);
    l += (long long)K0 * ldl + K0;
    double *wop = ops_all + (size_t)blockIdx.x * (16 + NTILE) * TILE_DOUBLES, *lop = wop + 16 * TILE_DOUBLES;
    const int tid = threadIdx.x, lane = tid & 63, k = tid >> 6, lr = lane & 15, lq = lane >> 4;
    const int nt = (kb + DT - 1) / DT;
    if (k < nt && lane < 16) {
        double t[16], acc[16];                                   // row `lane` of the tile; column `lane` of its inverse
#pragma unroll
        for (int c = 0; c < 16; c++) {
            const int row = k * DT + lane, col = k * DT + c;
            t[c] = (row < kb && col < kb) ? (c <= lane ? l[(long long)row * ldl + col] : 0.0) : (row == col ? 1.0 : 0.0);
            acc[c] = (c == lane) ? 1.0 : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 16; q++) {                           // T[r][q] lives in lane r: wave-uniform through v_readlane
            const double x = acc[q] / readlane_d(t[q], q);
            acc[q] = x;
#pragma unroll
            for (int r = q + 1; r < 16; r++) acc[r] = __builtin_fma(-readlane_d(t[q], r), x, acc[r]);
        }
#pragma unroll
        for (int r = 0; r < 16; r++) W[k][r][lane] = (r >= lane) ? acc[r] : 0.0;
    }
    __syncthreads();
    if (k < nt) {
#pragma unroll
        for (int q = 0; q < 4; q++) wop[(size_t)k * TILE_DOUBLES + q * 64 + lane] = W[k][lr][lq + 4 * q];
    }
    for (int t = k; t < NTILE; t += 16) {                        // tiles (j, kk), j > kk, in operand order
        const int j = c_tiles.i[t], kk = c_tiles.j[t];
        if (j == kk || j >= nt) continue;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int row = j * DT + lr, col = kk * DT + lq + 4 * q;
            lop[(size_t)t * TILE_DOUBLES + q * 64 + lane] = (row < kb && col < kb) ? l[(long long)row * ldl + col] : 0.0;
        }
    }
}

__global__ void __launch_bounds__(256) trtri_eye_kernel(double *t, size_t total) {      // identities, one per diagonal block
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e < total) { const size_t w = e % ((size_t)DB * DB); t[e] = (w / DB == w % DB) ? 1.0 : 0.0; }
}

// X[rows, rows] for the block rows [b0, b1) whose two halves are already inverted:
//   tmp[m, k] = L21 X11 (X11 lower triangular: the k range of a column block starts at its own rows)
//   X21 = -X22 tmp      (X22 lower triangular: the k range of a row block ends at its own rows)
struct trtri_node { int b0, mid, b1; };
static int trtri_join(hipStream_t st, int n, const double *l, long long ldl, double *x, long long ldx, const trtri_node &nd, double *tmp) {
    const int r0 = nd.b0 * DB, r1 = nd.mid * DB, r2 = std::min(nd.b1 * DB, n), m = r2 - r1, k = r1 - r0;
    int rc = gpr_gemm_internal(st, 0, 0, m, k, k, 1.0, l + (long long)r1 * ldl + r0, ldl, x + (long long)r0 * ldx + r0, ldx, 0.0, tmp, k, 16);
    if (rc) return rc;
    return gpr_gemm_internal(st, 0, 0, m, k, m, -1.0, x + (long long)r1 * ldx + r1, ldx, tmp, k, 0.0, x + (long long)r1 * ldx + r0, ldx, 2);
}
static void trtri_collect(int b0, int b1, size_t depth, std::vector<std::vector<trtri_node>> &levels) {
    if (b1 - b0 < 2) return;
    const int mid = b0 + (b1 - b0 + 1) / 2;
    if (levels.size() <= depth) levels.resize(depth + 1);
    levels[depth].push_back(trtri_node{b0, mid, b1});
    trtri_collect(b0, mid, depth + 1, levels);
    trtri_collect(mid, b1, depth + 1, levels);
}
// The halves are joined level by level, deepest first.  The joins of one level touch disjoint blocks, and below the top
// they are small products (at n = 6063: 4, 8, 36, 144 workgroups on 256 CUs, 3.6 of the routine's 5.6 ms one after the
// other), so they are spread over up to six streams -- the caller's and the private ones of the device's look-ahead
// context -- between a fork and a join event per level; each gets its own piece of `tmp` (cap doubles in all).
static int trtri_levels(hipStream_t st, lookahead_ctx *la, int n, const double *l, long long ldl, double *x, long long ldx, int nblk,
                        double *tmp, size_t cap) {
    std::vector<std::vector<trtri_node>> levels;
    trtri_collect(0, nblk, 0, levels);
    for (size_t d = levels.size(); d-- > 0;) {
        const std::vector<trtri_node> &nodes = levels[d];
        size_t need = 0;
        for (const trtri_node &nd : nodes) need += (size_t)(std::min(nd.b1 * DB, n) - nd.mid * DB) * (size_t)((nd.mid - nd.b0) * DB);
        const bool spread = la && nodes.size() > 1 && need <= cap;
        hipStream_t lanes[6] = {st, st, st, st, st, st};
        int nlanes = 1;
        if (spread) {
            lanes[1] = la->mid; lanes[2] = la->side; lanes[3] = la->aux[0]; lanes[4] = la->aux[1]; lanes[5] = la->aux[2];
            nlanes = (int)std::min<size_t>(6, nodes.size());
            hipEvent_t e = la->take();
            GPR_HIP_TRY(hipEventRecord(e, st));
            for (int q = 1; q < nlanes; q++) GPR_HIP_TRY(hipStreamWaitEvent(lanes[q], e, 0));
        }
        size_t off = 0;
        for (size_t i = 0; i < nodes.size(); i++) {
            const trtri_node &nd = nodes[i];
            const int rc = trtri_join(lanes[i % nlanes], n, l, ldl, x, ldx, nd, tmp + (spread ? off : 0));
            if (rc) return rc;
            off += (size_t)(std::min(nd.b1 * DB, n) - nd.mid * DB) * (size_t)((nd.mid - nd.b0) * DB);
        }
        for (int q = 1; q < nlanes; q++) {
            hipEvent_t e = la->take();
            GPR_HIP_TRY(hipEventRecord(e, lanes[q]));
            GPR_HIP_TRY(hipStreamWaitEvent(st, e, 0));
        }
    }
    return GPR_OK;
}

extern "C" int gpr_trtri_lower(void *stream, int n, const double *l, long long ldl, double *x, long long ldx) {
    hipStream_t st = (hipStream_t)stream;
    GPR_REQUIRE(n >= 0 && (n == 0 || (l && x)) && ldl >= n && ldx >= n, "gpr_trtri_lower: bad arguments");
    if (n == 0) return GPR_OK;
    const int nblk = (n + DB - 1) / DB, nfull = n / DB;
    const size_t ops = (size_t)(16 + NTILE) * TILE_DOUBLES, half = (size_t)((nblk + 1) / 2) * DB;
    void *scratch;
    int rc = gpr_dev_alloc(&scratch, ((size_t)nblk * (ops + (size_t)DB * DB) + half * half) * sizeof(double), st);
    if (rc) return rc;
    double *ops_all = (double *)scratch, *eye = ops_all + (size_t)nblk * ops, *tmp = eye + (size_t)nblk * DB * DB;
    // 1. diagonal blocks: eye[b] <- L_bb^-T, X[b, b] = its transpose
    hipLaunchKernelGGL(tri_tiles256_kernel, dim3(nblk), dim3(1024), 0, st, l, ldl, n, ops_all);
    const size_t eyes = (size_t)nblk * DB * DB;
    hipLaunchKernelGGL(trtri_eye_kernel, dim3((unsigned)((eyes + 255) / 256)), dim3(256), 0, st, eye, eyes);
    if (nfull > 0)
        hipLaunchKernelGGL((potrf_panel256_kernel<true, false>), dim3(DB / DT, nfull), dim3(256), 0, st, eye, (long long)DB, DB, DB, ops_all,
                           ops_all + 16 * TILE_DOUBLES, (const int *)nullptr, 0, (int *)nullptr, (long long)DB * DB, (long long)ops);
    if (nblk > nfull)
        hipLaunchKernelGGL((potrf_panel256_kernel<false, false>), dim3(DB / DT), dim3(256), 0, st, eye + (size_t)nfull * DB * DB, (long long)DB, DB,
                           n - nfull * DB, ops_all + (size_t)nfull * ops, ops_all + (size_t)nfull * ops + 16 * TILE_DOUBLES, (const int *)nullptr, 0,
                           (int *)nullptr, 0LL, 0LL);
    hipLaunchKernelGGL(trtri_store_kernel, dim3(DB / 16, DB / 16, nblk), dim3(256), 0, st, n, eye, x, ldx);
    // 2. the blocks below the diagonal, by halves; the device's look-ahead context lends its two streams (its mutex and
    // end-of-call event order this call against a factorisation or another inverse in flight on the device)
    {
        int devid = 0;
        if (hipGetDevice(&devid) != hipSuccess || devid < 0 || devid >= LA_MAX_DEVICES) { (void)hipGetLastError(); devid = -1; }
        static const bool serial_env = getenv("GPR_TRTRI_SERIAL") != nullptr;
        lookahead_ctx *la = (devid >= 0 && nblk > 3 && !serial_env) ? &g_la_dev[devid] : nullptr;
        std::unique_lock<std::mutex> lock;
        if (la) {
            lock = std::unique_lock<std::mutex>(la->mu);
            if (la->init() != GPR_OK) { lock.unlock(); la = nullptr; }
        }
        if (la && la->end_valid && hipStreamWaitEvent(st, la->end_ev, 0) != hipSuccess) { (void)hipGetLastError(); la = nullptr; }
        rc = trtri_levels(st, la, n, l, ldl, x, ldx, nblk, tmp, half * half);
        if (la) la->end_valid = hipEventRecord(la->end_ev, st) == hipSuccess;
    }
    const hipError_t e = hipGetLastError();
    const int rc2 = gpr_dev_free(scratch, st);
    if (e != hipSuccess) { gpr_set_error("gpr_trtri_lower launch failed: %s", hipGetErrorString(e)); return GPR_ERR_HIP; }
    return rc ? rc : rc2;
}

extern "C" int gpr_potrf_lower(void *stream, int n, double *a, long long lda, int *info_dev) {
    hipStream_t st = (hipStream_t)stream;
    if (n <= 0) return GPR_OK;
    void *tmp;
    int rc = gpr_dev_alloc(&tmp, (size_t)((n + 63) / 64) * 4096 * sizeof(double), st);
    if (rc) return rc;
    rc = gpr_potrf_lower_dinv(stream, n, a, lda, info_dev, (double *)tmp);
    const int rc2 = gpr_dev_free(tmp, st);
    return rc ? rc : rc2;
}
