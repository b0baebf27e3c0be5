// kblock.hip -- covariance-block kernels (energy-energy, energy-force, force-force; RBF and Dot)
// on the FP64 matrix cores of gfx950.
//
// Replaces the row-pair loop nests of rbf_kee/kef/kff_many(+_with_grad) (rbf_kernel.cpp:5-640) and
// dot_kee/kef/kff_many (dot_kernel.cpp:5-336).  Formulation (restated for
// the projected operands of pack.hip): for a row pair (a, b) the 4x4 product
//        [A'^T ; a~^T] (4 x d)  *  [B' , b~] (d x 4)
// yields G = A'^T B' (3x3), w = A'^T b~ (3), z = a~^T B' (3) and the cosine c, and
//        force-force 3x3 block = alpha * G + beta * w z^T
//        energy-force 3-vector = coef * z
//        energy-energy scalar  = k(c)
// with alpha, beta, coef, k scalar functions of c (one exp for RBF).  The 4x4 products of a
// 4-row x 16-row tile are four v_mfma_f64_16x16x4_f64 accumulators whose C/D layout leaves all 16
// numbers of ONE row pair in ONE lane, so the scalar epilogue needs no cross-lane traffic.
//
// Work decomposition: one wavefront owns one resident ("set 2") point j and keeps the MFMA B
// operands of one 16-row group of it in registers; the workgroup streams the tiles of the
// stream ("set 1") points through double-buffered LDS; each wavefront accumulates the bin (i, j)
// in registers and, at the last tile of point i, reduces over its 64 lanes and adds the 3x3
// (3, 1) result to the output with no-return FP64 atomic adds (exactly one wavefront ever touches a
// bin and adds to it in program order, so the summation order is deterministic).
//
// Small launches (a handful of long points: one 220-atom structure against a training set of a few dozen points -- the
// on-the-fly loop) leave the chip empty in that decomposition: a wavefront walks the up to fourteen groups of its unit one
// after the other and the launch is as long as that walk (170 us for a few thousand tiles).  In SPLIT mode a block takes ONE
// unit and deals its groups to its wavefronts (wavefront w: groups w, w + NW, ...); all of them stream the same staged
// tiles, and at the end of a stream point their per-point totals meet in LDS and wavefront 0 adds them up in wavefront
// order and issues the adds -- still one writer per bin per pass, still a fixed summation order.
#include "gpr_internal.h"

typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int KB_SPLIT_PTS = 8;          // points a unit may hold in split mode (the LDS exchange is sized for it)

struct kb_args {
    const double *a_ops; const int *a_ele; const double *a_norm; const gpr_chunk *a_chunks;
    const int *a_perm; const double *a_inv; const int *a_pt_chunk0;
    const double *b_ops; const int *b_ele; const double *b_norm; const gpr_rpoint *b_wu;
    const int *b_wu_pt0, *b_rowpt;   // first packed point of every wave unit [nwu+1]; packed point of every resident row
    const int *b_perm; const double *b_inv;
    int b_npts, a_npts, b_nwu;
    const int *ir_chunk0;   // [nir+1] chunk range of each stream range
    const int *ir_pt0;      // [nir+1] first packed point of each stream range
    int njb, nir;
    int shard_mod, shard_rem;
    int symmetric, diag_only, eps_mode, norm1, norm2;
    int split;              // small launches: one wave unit per block, its groups dealt to the block's wavefronts (see below)
    int xcd;                // stream ranges dealt to the XCDs (block index -> (jb, ir), see the kernel)
    int merge, npacked;     // packed stream ranges per launch range; packed ranges in all
    int dparts;             // diagonal launches: blocks per unit
    double zeta, sigma2, inv2l2, inv_l, inv_l3, sigma02, tol;
    double scale, dl_scale;
    double *out; long long s_i, s_j, s_a, s_b;
    double *dl;  long long d_i, d_j, d_a, d_b;
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// scalar functions of the cosine c shared by all blocks
template <int KIND, bool Z2>
__device__ __forceinline__ void pair_scalars(double c, const kb_args &P, double &D, double &d1,
                                             double &d2, double &Kv, double &g) {
    if (Z2) { d1 = c; D = c * c; d2 = 1.0; }
    else {                                   // rbf_kernel.cpp:400-402
        double cz2 = pow(c, P.zeta - 2.0);
        d1 = c * cz2; D = c * d1; d2 = cz2 * (P.zeta - 1.0);
    }
    if (KIND == GPR_KERNEL_RBF) {
        Kv = P.sigma2 * exp((D - 1.0) * P.inv2l2);   // sigma2*exp(-(1-D)/(2 l2))
        g = Kv * P.inv2l2;                           // dK_dD
    } else { Kv = 0.0; g = 1.0; }
}

// KS: k-steps per operand plane, 8 (d <= 32), 16 (d <= 64; four stream tiles per staged chunk instead of eight) or 32 (d <= 128; two)
template <int MODE, int KIND, bool Z2, bool GRAD, int NW, int KS>
__global__ void __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(KS <= GPR_KSTEPS_MAX ? 2 : 1)))     // two wavefronts per SIMD
kblock_kernel(kb_args P) {                                                                                // (<= 256 registers) up to d = 64
    constexpr int PLANE = KS * GPR_WAVE, CTILES = GPR_CHUNK_TILES * GPR_KSTEPS / KS;
    constexpr int NCB = (MODE == GPR_MODE_EE) ? 1 : 4;            // resident planes per group
    constexpr int NOUT = (MODE == GPR_MODE_FF) ? 9 : (MODE == GPR_MODE_EF ? 3 : 1);
    constexpr int AROWS = (MODE == GPR_MODE_FF) ? 4 : 16;         // stream rows per tile
    constexpr int NTHREADS = NW * 64;
    constexpr int CHUNK_BYTES = CTILES * PLANE * 8;
    constexpr int NLD = CHUNK_BYTES / (NTHREADS * 16);
    static_assert(NLD * NTHREADS * 16 == CHUNK_BYTES, "chunk must divide evenly");

    __shared__ __attribute__((aligned(16))) double sA[2][CTILES * PLANE];
    __shared__ int sEle[2][GPR_CHUNK_TILES * 16];
    __shared__ double sNorm[2][GPR_CHUNK_TILES * 16];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // block -> (resident block jb, stream range ir).  Workgroups are dealt to the eight XCDs in turn, and every XCD has an L2
    // of its own: with `xcd` set the ranges are dealt to the XCDs (range ir lives on XCD ir % 8) and an XCD walks its ranges
    // one after the other, all resident blocks of a range side by side -- the range's stream tiles (re-read by every group
    // pass of every block) then stay in ONE L2 instead of being pulled into all eight and pushed out again by the ranges
    // the other blocks in flight are working on.
    // Symmetric launches skip what lies below the diagonal, so a range's work shrinks with its index: there the ranges are dealt
    // back and forth (0 .. 7, 7 .. 0, ...), which gives every XCD the same share.  `merge` packed ranges make one launch range
    // (a block reloads its resident operands once per launch range: longer ranges, fewer reloads; the symmetric launches
    // keep the short ones, whose triangular skipping is finer).
    int jb, ir;
    // The ranges left over after the last full round of eight are dealt block by block again (an XCD without a range would idle:
    // workgroups are dispatched in order).
    const int nfull = (P.nir >> 3) * 8 * P.njb;                // blocks of the full rounds
    if (P.xcd && (int)blockIdx.x < nfull) {
        const int x = blockIdx.x & 7, sq = blockIdx.x >> 3, rnd = sq / P.njb;
        jb = sq % P.njb;
        ir = rnd * 8 + ((P.symmetric && (rnd & 1)) ? 7 - x : x);
    } else {
        const int t = P.xcd ? (int)blockIdx.x - nfull : (int)blockIdx.x, r0_ = P.xcd ? (P.nir >> 3) * 8 : 0;
        jb = t % P.njb; ir = r0_ + t / P.njb;
    }
    const int ir0 = ir * P.merge, ir1 = min(ir0 + P.merge, P.npacked);
    // resident wave unit of this wavefront (consecutive packed points).  Diagonal launches are tiny, so there every
    // block takes ONE unit.  Usually they run in split form (the unit's groups dealt to the wavefronts, below); a unit of
    // more than KB_SPLIT_PTS points keeps the older form: its wavefronts share the unit's points (wave w flushes points
    // w, w + NW, ...: the MFMA passes are repeated per wave, but each bin still has a single writer)
    const bool one_unit = P.diag_only || P.split;
    const int ju = one_unit ? jb : jb * NW + wave;
    const bool active = ju < P.b_nwu;
    const int u_first = one_unit ? jb : jb * NW, u_end = one_unit ? jb + 1 : min((jb + 1) * NW, P.b_nwu);
    // split mode: per-point totals of every wavefront, [NW][KB_SPLIT_PTS][value | derivative]
    constexpr int RW = (GRAD ? 2 : 1) * NOUT;
    __shared__ double sRed[NW * KB_SPLIT_PTS * RW];
    const int blk_pt0 = P.b_wu_pt0[u_first], blk_pt1 = P.b_wu_pt0[u_end];                           // points of this block
    const int jlast = blk_pt1 - 1;
    if ((jb + ir) % P.shard_mod != P.shard_rem) return;   // multi-GPU: this task belongs to another rank

    int c0, c1;
    if (P.diag_only) {                                 // only the chunks of this block's own points -- of its share of them:
        // `dparts` blocks per unit (blockIdx.y) take consecutive points of the unit each (a point against itself is independent
        // of the others, and a block's time is the walk over its chunks whichever wavefront owns them)
        const int npt = blk_pt1 - blk_pt0, part = blockIdx.y;
        const int lo = blk_pt0 + (int)((long long)npt * part / P.dparts), hi = blk_pt0 + (int)((long long)npt * (part + 1) / P.dparts);
        c0 = P.a_pt_chunk0[lo];
        c1 = P.a_pt_chunk0[hi];
    } else {
        if (P.symmetric && P.ir_pt0[ir0] > jlast) return;  // range entirely below the diagonal
        c0 = P.ir_chunk0[ir0];
        c1 = P.ir_chunk0[ir1];
    }
    if (c0 >= c1) return;

    gpr_rpoint pj = active ? P.b_wu[ju] : gpr_rpoint{0, 0};
    __shared__ int sMaxg;                              // the wavefronts of a block walk their groups in lock step
    if (tid == 0) sMaxg = 0;
    __syncthreads();
    if (lane == 0 && active) atomicMax(&sMaxg, P.split ? (pj.ngroups + NW - 1) / NW : pj.ngroups);
    __syncthreads();
    const int maxg = sMaxg;
    // lane roles of the segmented reductions below (fixed for the kernel): after a reduction lane l holds the total of value
    // `ridx`; the first NH values sit in the lanes with l % SHARE == 0, the ninth (if any) in lane 1
    constexpr int NH = (NOUT >= 8) ? 8 : (NOUT >= 2 ? 2 : 1);
    constexpr int SHARE = 64 / NH;
    const bool w_main = (lane & (SHARE - 1)) == 0;
    const bool w_extra = (NOUT > NH) && lane == 1;
    const int ridx = w_extra ? NH : ((NH == 8) ? ((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1)
                                               : (NH == 2 ? (lane >> 5) & 1 : 0));

    for (int gp = 0; gp < maxg; ++gp) {
        const int g = P.split ? gp * NW + wave : gp;
        const bool has = active && g < pj.ngroups;
        double Breg[NCB][KS];
        int eleB = GPR_ELE_PAD_RESIDENT;
        double normB = 0.0;
        int my_pt = -1;                                // packed resident point of this lane's column (-1: padding)
        if (has) {
            const double *bp = P.b_ops + (size_t)(pj.group0 + g) * NCB * PLANE;
#pragma unroll
            for (int t = 0; t < NCB; t++)
#pragma unroll
                for (int s = 0; s < KS; s++) Breg[t][s] = bp[(t * KS + s) * 64 + lane];
            eleB = P.b_ele[(pj.group0 + g) * 16 + (lane & 15)];
            my_pt = P.b_rowpt[(pj.group0 + g) * 16 + (lane & 15)];
            if (MODE == GPR_MODE_EE) normB = P.b_norm[(pj.group0 + g) * 16 + (lane & 15)];
        } else {
#pragma unroll
            for (int t = 0; t < NCB; t++)
#pragma unroll
                for (int s = 0; s < KS; s++) Breg[t][s] = 0.0;
        }

        // the rows of a group belong to a contiguous run of points [pt_lo, pt_hi] (one point unless a boundary falls
        // inside the group); every lane accumulates the bin of its own column's point
        int pt_lo = my_pt < 0 ? INT_MAX : my_pt, pt_hi = my_pt;
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) {
            pt_lo = min(pt_lo, __shfl_xor(pt_lo, o, 64));
            pt_hi = max(pt_hi, __shfl_xor(pt_hi, o, 64));
        }
        double bin[NOUT], bin2[GRAD ? NOUT : 1];
#pragma unroll
        for (int k = 0; k < NOUT; k++) bin[k] = 0.0;
#pragma unroll
        for (int k = 0; k < (GRAD ? NOUT : 1); k++) bin2[k] = 0.0;

        // ---- stage the first chunk
        float4 regs[NLD];
        int ereg = 0; double nreg = 0.0;
        // global -> registers (issued before the tile loop) and registers -> LDS (after it); macros rather
        // than lambdas so that the staging registers are never address-taken (no scratch)
#define GPR_STAGE_LOAD(C_)                                                                              \
        {   const gpr_chunk ch_ = P.a_chunks[(C_)];                                                      \
            const char *src_ = (const char *)(P.a_ops + (size_t)ch_.tile0 * PLANE);                      \
            const int bytes_ = ch_.ntiles * PLANE * 8;                                                   \
            _Pragma("unroll") for (int i_ = 0; i_ < NLD; i_++) {                                         \
                const int off_ = (i_ * NTHREADS + tid) * 16;                                             \
                regs[i_] = (off_ < bytes_) ? *(const float4 *)(src_ + off_) : make_float4(0.f, 0.f, 0.f, 0.f); } \
            if (tid < ch_.ntiles * AROWS) {                                                              \
                ereg = P.a_ele[ch_.tile0 * AROWS + tid];                                                 \
                if (MODE == GPR_MODE_EE) nreg = P.a_norm[ch_.tile0 * AROWS + tid]; } }
#define GPR_STAGE_STORE(C_, BUF_)                                                                       \
        {   const gpr_chunk ch_ = P.a_chunks[(C_)];                                                      \
            const int bytes_ = ch_.ntiles * PLANE * 8;                                                   \
            char *dst_ = (char *)sA[(BUF_)];                                                             \
            _Pragma("unroll") for (int i_ = 0; i_ < NLD; i_++) {                                         \
                const int off_ = (i_ * NTHREADS + tid) * 16;                                             \
                if (off_ < bytes_) *(float4 *)(dst_ + off_) = regs[i_]; }                                \
            if (tid < ch_.ntiles * AROWS) {                                                              \
                sEle[(BUF_)][tid] = ereg;                                                                \
                if (MODE == GPR_MODE_EE) sNorm[(BUF_)][tid] = nreg; } }
        GPR_STAGE_LOAD(c0)
        GPR_STAGE_STORE(c0, 0)
        __syncthreads();

        for (int c = c0; c < c1; ++c) {
            const int buf = (c - c0) & 1;
            const gpr_chunk ch = P.a_chunks[c];
            if (c + 1 < c1) GPR_STAGE_LOAD(c + 1)

            bool mine = has;
            if (P.symmetric && ch.pt > pt_hi) mine = false;  // every point of the group is below the diagonal
            if (P.diag_only && (ch.pt < pt_lo || ch.pt > pt_hi)) mine = false;
            if (mine) {
                for (int tile = 0; tile < ch.ntiles; ++tile) {
                    const double *At = sA[buf] + tile * PLANE;
                    d4 acc[NCB];
#pragma unroll
                    for (int t = 0; t < NCB; t++) acc[t] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int s = 0; s < KS; s++) {
                        const double a = At[s * 64 + lane];
#pragma unroll
                        for (int t = 0; t < NCB; t++)
                            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Breg[t][s], acc[t], 0, 0, 0);
                    }
                    if (MODE == GPR_MODE_FF) {
                        // lane = (stream row lane>>4 of the tile, resident row lane&15);
                        // acc[t][r]: r = stream component (A'x,A'y,A'z,a~), t = resident component
                        const int eleA = sEle[buf][tile * 4 + (lane >> 4)];
                        bool ok = (eleA == eleB);
                        const double c_ = acc[3][3];
                        double D, d1, d2, Kv, gg;
                        pair_scalars<KIND, Z2>(c_, P, D, d1, d2, Kv, gg);
                        double alpha, beta, alpha2 = 0.0, beta2 = 0.0;
                        if (KIND == GPR_KERNEL_RBF) {
                            const double zd1 = P.zeta * d1;
                            const double cross = zd1 * zd1 * P.inv2l2;   // (1/2l2) dD/dx1 dD/dx2 scale
                            alpha = gg * zd1;
                            beta = gg * (P.zeta * d2 + cross);
                            if (GRAD) {                                   // rbf_kernel.cpp:622-630
                                const double f = (D - 1.0) * P.inv_l3 + 2.0 * P.inv_l;
                                alpha2 = -f * alpha;
                                beta2 = -(f * beta + 2.0 * P.inv_l * gg * cross);
                            } else {
                                ok = ok && (gg > P.tol);                  // rbf_kernel.cpp:395
                            }
                        } else {                                          // dot_kernel.cpp:285-289
                            alpha = d1; beta = d2;
                        }
                        alpha = ok ? alpha : 0.0; beta = ok ? beta : 0.0;
                        if (GRAD) { alpha2 = ok ? alpha2 : 0.0; beta2 = ok ? beta2 : 0.0; }
#pragma unroll
                        for (int p = 0; p < 3; p++) {
                            const double bw = beta * acc[3][p];
                            const double bw2 = GRAD ? beta2 * acc[3][p] : 0.0;
#pragma unroll
                            for (int q = 0; q < 3; q++) {      // two fused multiply-adds per output
                                bin[p * 3 + q] = fma(bw, acc[q][3], fma(alpha, acc[q][p], bin[p * 3 + q]));
                                if (GRAD) bin2[p * 3 + q] = fma(bw2, acc[q][3], fma(alpha2, acc[q][p], bin2[p * 3 + q]));
                            }
                        }
                    } else {
                        // lane = (stream rows (lane>>4) + 4 r, r < 4 ; resident row lane&15)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int arow = tile * 16 + (lane >> 4) + 4 * r;
                            const int eleA = sEle[buf][arow];
                            const bool ok = (eleA == eleB);
                            double c_ = acc[NCB - 1][r];
                            if (MODE == GPR_MODE_EE && P.eps_mode) {       // base.py:121, Dot_mb.py:191
                                const double nn = sNorm[buf][arow] * normB;
                                c_ = c_ * nn / (1e-8 + nn);
                            }
                            double D, d1, d2, Kv, gg;
                            pair_scalars<KIND, Z2>(c_, P, D, d1, d2, Kv, gg);
                            if (MODE == GPR_MODE_EF) {
                                // rbf_kernel.cpp:143-165 (pout -= ...), dot_kernel.cpp:104-122 (+=)
                                double coef = (KIND == GPR_KERNEL_RBF) ? -gg * P.zeta * d1 : P.zeta * d1;
                                coef = ok ? coef : 0.0;
#pragma unroll
                                for (int q = 0; q < 3; q++) bin[q] += coef * acc[q][r];
                                if (GRAD) {                                 // rbf_kernel.cpp:245-247
                                    const double f = (D - 1.0) * P.inv_l3 + 2.0 * P.inv_l;
                                    const double coef2 = -coef * f;
#pragma unroll
                                    for (int q = 0; q < 3; q++) bin2[q] += coef2 * acc[q][r];
                                }
                            } else {
                                double v = (KIND == GPR_KERNEL_RBF) ? Kv : (D + P.sigma02);
                                bin[0] += ok ? v : 0.0;
                                if (GRAD) bin2[0] += ok ? Kv * (1.0 - D) : 0.0;   // rbf_kernel.cpp:93
                            }
                        }
                    }
                }
            }
            if (ch.last) {
                const int i_orig = P.a_perm[ch.pt];
                const double fac_i = P.norm1 ? P.a_inv[ch.pt] : 1.0;
                // halving reduction on a masked copy of the bins (lanes of other points contribute zeros): the first
                // NH (power of two) values are reduced in log2(NH) exchange steps that each keep half of them, then
                // plain xor steps; the value(s) beyond NH take the plain 6-step butterfly.  Afterwards lane l holds
                // the total of value `ridx`, and every lane the totals of the values >= NH.  Value and derivative bins go
                // through it one after the other (register pressure).
#define GPR_HALVE(ARR, OFF, HALF)                                                                 \
                {   const bool up_ = (lane & (OFF)) != 0;                                          \
                    _Pragma("unroll") for (int k_ = 0; k_ < (HALF); k_++) {                        \
                        const double send_ = up_ ? ARR[k_] : ARR[k_ + (HALF)];                     \
                        const double keep_ = up_ ? ARR[k_ + (HALF)] : ARR[k_];                     \
                        ARR[k_] = keep_ + __shfl_xor(send_, (OFF), 64); } }
#define GPR_PLAIN(X, FROM)                                                                        \
                { _Pragma("unroll") for (int o_ = (FROM); o_ > 0; o_ >>= 1) X += __shfl_xor(X, o_, 64); }
#define GPR_SEG_REDUCE(SRC, VAL)                                                                  \
                {   double tb_[NOUT];                                                              \
                    _Pragma("unroll") for (int k_ = 0; k_ < NOUT; k_++) tb_[k_] = inseg ? SRC[k_] : 0.0;  \
                    if (NH == 8) { GPR_HALVE(tb_, 32, 4) GPR_HALVE(tb_, 16, 2) GPR_HALVE(tb_, 8, 1) GPR_PLAIN(tb_[0], 4) } \
                    else if (NH == 2) { GPR_HALVE(tb_, 32, 1) GPR_PLAIN(tb_[0], 16) }              \
                    else { GPR_PLAIN(tb_[0], 32) }                                                 \
                    _Pragma("unroll") for (int k_ = NH; k_ < NOUT; k_++) GPR_PLAIN(tb_[k_], 32)           \
                    VAL = w_extra ? tb_[NOUT - 1] : tb_[0]; }
                // one lane per value adds it to the output.  The hardware FP64 atomic add returns nothing, so
                // the wavefront does not wait for memory; exactly one wavefront ever adds to a given element
                // and its adds are issued in program order, so the result does not depend on scheduling.
                auto emit = [&](int sp, double v, double v2) {
                    if (!((w_main || w_extra) && ridx < NOUT)) return;
                    const int j_orig = P.b_perm[sp];
                    const double fac = fac_i * (P.norm2 ? P.b_inv[sp] : 1.0);
                    const int a = (MODE == GPR_MODE_FF) ? ridx / 3 : 0;
                    const int b = (MODE == GPR_MODE_FF) ? ridx % 3 : ridx;
                    const double val = v * fac * P.scale;
                    if (P.diag_only) {
                        if (a == b) unsafeAtomicAdd(P.out + (long long)j_orig * P.s_j + b * P.s_b, val);
                    } else {
                        unsafeAtomicAdd(P.out + i_orig * P.s_i + j_orig * P.s_j + a * P.s_a + b * P.s_b, val);
                        if (P.symmetric && ch.pt != sp)
                            unsafeAtomicAdd(P.out + j_orig * P.s_i + i_orig * P.s_j + b * P.s_a + a * P.s_b, val);
                        if (GRAD) {
                            const double vald = v2 * fac * P.dl_scale;
                            unsafeAtomicAdd(P.dl + i_orig * P.d_i + j_orig * P.d_j + a * P.d_a + b * P.d_b, vald);
                            if (P.symmetric && ch.pt != sp)
                                unsafeAtomicAdd(P.dl + j_orig * P.d_i + i_orig * P.d_j + b * P.d_a + a * P.d_b, vald);
                        }
                    }
                };
                if (!P.split) {
                    if (mine)
                        // one reduction and one set of adds per resident point present in the group
                        for (int sp = pt_lo; sp <= pt_hi; ++sp) {
                            if (P.symmetric && ch.pt > sp) continue;          // this point's pair lies below the diagonal
                            if (P.diag_only && (ch.pt != sp || ((sp - blk_pt0) & (NW - 1)) != wave)) continue;
                            const bool inseg = (my_pt == sp);
                            double v, v2 = 0.0;
                            GPR_SEG_REDUCE(bin, v)
                            if (GRAD) GPR_SEG_REDUCE(bin2, v2)
                            emit(sp, v, v2);
                        }
                } else {
                    // split mode: every wavefront leaves the totals of its own group for the points of the unit in LDS, then
                    // wavefront 0 adds them in wavefront order and issues the adds (the loops are uniform over the block)
                    for (int sp = blk_pt0; sp <= jlast; ++sp) {
                        if (P.symmetric && ch.pt > sp) continue;
                        if (P.diag_only && ch.pt != sp) continue;             // a point against itself only
                        const bool here = mine && sp >= pt_lo && sp <= pt_hi;
                        double v = 0.0, v2 = 0.0;
                        if (here) {
                            const bool inseg = (my_pt == sp);
                            GPR_SEG_REDUCE(bin, v)
                            if (GRAD) GPR_SEG_REDUCE(bin2, v2)
                        }
                        if ((w_main || w_extra) && ridx < NOUT) {
                            double *r_ = sRed + (wave * KB_SPLIT_PTS + (sp - blk_pt0)) * RW;
                            r_[ridx] = v;
                            if (GRAD) r_[NOUT + ridx] = v2;
                        }
                    }
                    __syncthreads();
                    if (wave == 0)
                        for (int sp = blk_pt0; sp <= jlast; ++sp) {
                            if (P.symmetric && ch.pt > sp) continue;
                            if (P.diag_only && ch.pt != sp) continue;
                            double v = 0.0, v2 = 0.0;
                            if ((w_main || w_extra) && ridx < NOUT) {
#pragma unroll
                                for (int w_ = 0; w_ < NW; w_++) {
                                    const double *r_ = sRed + (w_ * KB_SPLIT_PTS + (sp - blk_pt0)) * RW;
                                    v += r_[ridx];
                                    if (GRAD) v2 += r_[NOUT + ridx];
                                }
                            }
                            emit(sp, v, v2);
                        }
                    __syncthreads();
                }
#undef GPR_SEG_REDUCE
#undef GPR_HALVE
#undef GPR_PLAIN
#pragma unroll
                for (int k = 0; k < NOUT; k++) { bin[k] = 0.0; if (GRAD) bin2[k] = 0.0; }
            }
            if (c + 1 < c1) GPR_STAGE_STORE(c + 1, buf ^ 1)
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------- host launcher
template <int MODE, int KIND, bool Z2, bool GRAD>
static void launch_nw(hipStream_t st, int nw, int ks, int nblocks_, const kb_args &a) {
    const dim3 nblocks(nblocks_, a.diag_only ? a.dparts : 1);
    if (ks == GPR_KSTEPS_WIDE)      // d <= 128: the resident operands are 256 registers per lane, one wavefront per SIMD
        hipLaunchKernelGGL((kblock_kernel<MODE, KIND, Z2, GRAD, 4, GPR_KSTEPS_WIDE>), nblocks, dim3(256), 0, st, a);
    else if (ks == GPR_KSTEPS_MAX)  // long descriptors: four wavefronts (the resident operands alone are 128 VGPRs per lane)
        hipLaunchKernelGGL((kblock_kernel<MODE, KIND, Z2, GRAD, 4, GPR_KSTEPS_MAX>), nblocks, dim3(256), 0, st, a);
    else if (nw == 8) hipLaunchKernelGGL((kblock_kernel<MODE, KIND, Z2, GRAD, 8, GPR_KSTEPS>), nblocks, dim3(512), 0, st, a);
    else              hipLaunchKernelGGL((kblock_kernel<MODE, KIND, Z2, GRAD, 4, GPR_KSTEPS>), nblocks, dim3(256), 0, st, a);
}

template <int MODE, int KIND>
static void launch_zg(hipStream_t st, bool z2, bool grad, int nw, int ks, int nblocks, const kb_args &a) {
    if (z2) { if (grad) launch_nw<MODE, KIND, true, true>(st, nw, ks, nblocks, a); else launch_nw<MODE, KIND, true, false>(st, nw, ks, nblocks, a); }
    else    { if (grad) launch_nw<MODE, KIND, false, true>(st, nw, ks, nblocks, a); else launch_nw<MODE, KIND, false, false>(st, nw, ks, nblocks, a); }
}

static thread_local int g_shard_mod = 1, g_shard_rem = 0;
struct prof_rec { int mode; hipEvent_t start, stop; };
static thread_local bool g_prof_on = false;
static thread_local std::vector<prof_rec> g_prof;

extern "C" int gpr_set_shard(int nshards, int shard) {
    GPR_REQUIRE(nshards >= 1 && shard >= 0 && shard < nshards, "gpr_set_shard: bad shard %d of %d", shard, nshards);
    g_shard_mod = nshards; g_shard_rem = shard;
    return GPR_OK;
}

int gpr_launch_block(hipStream_t st, const gpr_block_call &c) {
    const gpr_set *s1 = c.s1, *s2 = c.s2;
    GPR_REQUIRE(s1 && s2, "block: null set");
    GPR_REQUIRE(s1->role == GPR_ROLE_STREAM && s2->role == GPR_ROLE_RESIDENT,
                "block: set1 must be packed with role STREAM and set2 with role RESIDENT");
    GPR_REQUIRE(s1->d == s2->d, "block: descriptor lengths differ (%d vs %d)", s1->d, s2->d);
    const int nc1 = (c.mode == GPR_MODE_FF) ? 4 : 1, nc2 = (c.mode == GPR_MODE_EE) ? 1 : 4;
    GPR_REQUIRE(s1->nc == nc1 && s2->nc == nc2, "block: sets have the wrong kind (energy/force) for this block");
    GPR_REQUIRE(!(c.grad && c.kind != GPR_KERNEL_RBF), "block: length-scale gradient exists for RBF only");
    GPR_REQUIRE(!c.grad || c.dl, "block: grad requested without dl output");
    if (c.symmetric || c.diag_only) {
        GPR_REQUIRE(s1->npts == s2->npts && s1->h_perm == s2->h_perm && s1->h_cnt == s2->h_cnt,
                    "block: symmetric/diag needs the same points packed in both roles");
    }
    if (s1->npts == 0 || s2->npts == 0 || s1->nchunks == 0 || s2->maxgroups == 0) return GPR_OK;

    // workgroup width: 8 wavefronts when that still leaves >= 2 blocks per CU, else 4
    GPR_REQUIRE(s1->ks == s2->ks, "block: sets packed with different plane widths");
    int nw = (s2->nwu >= 8 * 512) ? 8 : 4;
    if (c.diag_only || s1->ks >= GPR_KSTEPS_MAX) nw = 4;
    // a diagonal launch is one block per unit; in split form the unit's groups are dealt to the wavefronts, so eight of them
    // finish a full unit in one pass over the unit's own chunks
    const bool diag_split = c.diag_only && s2->max_unit_pts <= KB_SPLIT_PTS && s2->maxgroups >= 2;
    if (diag_split && s2->maxgroups > 4 && s1->ks < GPR_KSTEPS_MAX) nw = 8;
    // stream ranges (consecutive packed points, ~128 tiles each; finer for small sets) were laid out at pack time
    const int nranges = c.diag_only ? 1 : s1->nranges;
    // Split mode (see the top of the file): when the usual decomposition would not even give every CU a block and the units
    // have several groups, a block takes one unit and its wavefronts share the unit's groups.
    static const int split_env = getenv("GPR_KBLOCK_SPLIT") ? atoi(getenv("GPR_KBLOCK_SPLIT")) : -1;     // 0 / 1 force it (A/B, tests)
    bool split = !c.diag_only && s2->maxgroups >= 2 && s2->max_unit_pts <= KB_SPLIT_PTS &&
                 (long long)((s2->nwu + nw - 1) / nw) * nranges < gpr_cu_count();
    if (diag_split) split = true;
    if (split_env == 0) split = false;
    if (split_env == 1) split = s2->max_unit_pts <= KB_SPLIT_PTS;
    const int njb = (c.diag_only || split) ? s2->nwu : (s2->nwu + nw - 1) / nw;

    // GPR_KBLOCK_MERGE=2: predict-shaped launches take the packed ranges two at a time (see the kernel) -- the resident reloads,
    // most of what is left of the fetch traffic, halve (3.35 -> 2.17 GB per launch at the cfg3 shape) but the tasks are twice
    // as long and the last round of workgroups leaves more of the chip idle: +1.2 % on the RBF bench, so it is not the default
    static const int merge_env = getenv("GPR_KBLOCK_MERGE") ? atoi(getenv("GPR_KBLOCK_MERGE")) : 0;
    const int merge = (c.diag_only || c.symmetric || merge_env <= 1) ? 1 : merge_env;
    const int nir = (nranges + merge - 1) / merge;

    kb_args a;
    a.a_ops = s1->ops; a.a_ele = s1->ele; a.a_norm = s1->norms; a.a_chunks = s1->chunks;
    a.a_perm = s1->perm; a.a_inv = s1->inv_cnt; a.a_pt_chunk0 = s1->pt_chunk0;
    a.b_ops = s2->ops; a.b_ele = s2->ele; a.b_norm = s2->norms; a.b_wu = s2->rpts;
    a.b_wu_pt0 = s2->wu_pt0; a.b_rowpt = s2->rowpt; a.b_nwu = s2->nwu;
    a.b_perm = s2->perm; a.b_inv = s2->inv_cnt;
    a.b_npts = s2->npts; a.a_npts = s1->npts;
    a.ir_chunk0 = s1->ir_chunk0; a.ir_pt0 = s1->ir_pt0;
    a.njb = njb; a.nir = nir;
    a.shard_mod = g_shard_mod; a.shard_rem = g_shard_rem;
    a.symmetric = c.symmetric; a.diag_only = c.diag_only; a.eps_mode = c.eps_mode; a.split = split ? 1 : 0;
    a.norm1 = c.norm1; a.norm2 = c.norm2;
    a.zeta = c.zeta; a.sigma2 = c.sigma2; a.sigma02 = c.sigma02; a.tol = c.tol;
    if (c.kind == GPR_KERNEL_RBF) {
        a.inv2l2 = 1.0 / (2.0 * c.l2); a.inv_l = 1.0 / c.l; a.inv_l3 = a.inv_l / (c.l * c.l);
    } else { a.inv2l2 = 0.0; a.inv_l = 0.0; a.inv_l3 = 0.0; }
    a.scale = c.scale; a.dl_scale = c.dl_scale;
    a.out = c.out; a.s_i = c.s_i; a.s_j = c.s_j; a.s_a = c.s_a; a.s_b = c.s_b;
    a.dl = c.dl; a.d_i = c.d_i; a.d_j = c.d_j; a.d_a = c.d_a; a.d_b = c.d_b;

    const int prof_id = gpr_prof_begin(st, c.diag_only ? GPR_PROF_DIAG : c.mode);
    const bool z2 = (c.zeta == 2.0);
    static const int xcd_env = getenv("GPR_KBLOCK_XCD") ? atoi(getenv("GPR_KBLOCK_XCD")) : -1;          // 0 / 1 force it (A/B)
    const bool xcd = !c.diag_only && (xcd_env >= 0 ? xcd_env != 0 : nir >= 16);
    a.xcd = xcd ? 1 : 0; a.merge = merge; a.npacked = nranges;
    a.dparts = (c.diag_only && s2->max_unit_pts >= 2 && njb < 4 * gpr_cu_count()) ? std::min(4, s2->max_unit_pts) : 1;
    const int nblocks = c.diag_only ? njb : njb * nir;
    const bool grad = c.grad != 0;
    switch (c.mode * 2 + c.kind) {
        case GPR_MODE_FF * 2 + GPR_KERNEL_RBF: launch_zg<GPR_MODE_FF, GPR_KERNEL_RBF>(st, z2, grad, nw, s1->ks, nblocks, a); break;
        case GPR_MODE_FF * 2 + GPR_KERNEL_DOT: launch_zg<GPR_MODE_FF, GPR_KERNEL_DOT>(st, z2, false, nw, s1->ks, nblocks, a); break;
        case GPR_MODE_EF * 2 + GPR_KERNEL_RBF: launch_zg<GPR_MODE_EF, GPR_KERNEL_RBF>(st, z2, grad, nw, s1->ks, nblocks, a); break;
        case GPR_MODE_EF * 2 + GPR_KERNEL_DOT: launch_zg<GPR_MODE_EF, GPR_KERNEL_DOT>(st, z2, false, nw, s1->ks, nblocks, a); break;
        case GPR_MODE_EE * 2 + GPR_KERNEL_RBF: launch_zg<GPR_MODE_EE, GPR_KERNEL_RBF>(st, z2, grad, nw, s1->ks, nblocks, a); break;
        case GPR_MODE_EE * 2 + GPR_KERNEL_DOT: launch_zg<GPR_MODE_EE, GPR_KERNEL_DOT>(st, z2, false, nw, s1->ks, nblocks, a); break;
        default: gpr_set_error("block: bad mode/kind"); return GPR_ERR_ARG;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { gpr_set_error("kblock launch failed: %s", hipGetErrorString(e)); return GPR_ERR_HIP; }
    gpr_prof_end(st, prof_id);
    return GPR_OK;
}

// HIP events around one launch (or launch sequence) on its stream; no-ops unless profiling is enabled
// Events come from a pool that is only ever grown: creating a pair per launch put ~10 us of host time into every
// profiled step.
static thread_local std::vector<hipEvent_t> g_event_pool;
static thread_local size_t g_event_next = 0;
static hipEvent_t prof_event() {
    if (g_event_next == g_event_pool.size()) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        g_event_pool.push_back(e);
    }
    return g_event_pool[g_event_next++];
}
static thread_local int g_prof_period = 1;
static thread_local int g_prof_seen[GPR_PROF_NKINDS];
int gpr_prof_begin(hipStream_t st, int kind) {
    if (!g_prof_on) return -1;
    if ((g_prof_seen[kind]++) % g_prof_period != 0) return -1;          // sampled: every period-th launch of a kind
    prof_rec r{kind, prof_event(), prof_event()};
    if (!r.start || !r.stop) return -1;
    (void)hipEventRecord(r.start, st);
    g_prof.push_back(r);
    return (int)g_prof.size() - 1;
}
void gpr_prof_end(hipStream_t st, int id) {
    if (id >= 0 && id < (int)g_prof.size()) (void)hipEventRecord(g_prof[id].stop, st);
}

extern "C" int gpr_profile_enable(int on) {
    g_prof_on = on != 0;
    g_prof_period = on > 1 ? on : 1;
    for (int k = 0; k < GPR_PROF_NKINDS; k++) g_prof_seen[k] = 0;
    g_prof.clear();
    g_event_next = 0;
    if (on && g_event_pool.size() < 4096) {          // enough for a long timed region without a create inside it
        g_event_pool.reserve(4096);
        while (g_event_pool.size() < 4096) {
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess) break;
            g_event_pool.push_back(e);
        }
    }
    return GPR_OK;
}

// sums the kernel-only durations (ms) and launch counts recorded since gpr_profile_enable(1), per block
// kind (gpr_prof_kind, GPR_PROF_NKINDS entries); clears the records
extern "C" int gpr_profile_fetch(double *ms_by_mode, int *launches_by_mode) {
    for (int k = 0; k < GPR_PROF_NKINDS; k++) { ms_by_mode[k] = 0.0; launches_by_mode[k] = 0; }
    for (auto &r : g_prof) {
        GPR_HIP_TRY(hipEventSynchronize(r.stop));
        float ms = 0.f;
        GPR_HIP_TRY(hipEventElapsedTime(&ms, r.start, r.stop));
        ms_by_mode[r.mode] += ms; launches_by_mode[r.mode] += 1;
    }
    g_prof.clear();
    g_event_next = 0;
    return GPR_OK;
}

// launches of every kind seen since gpr_profile_enable, recorded or not (with sampling the two differ)
extern "C" int gpr_profile_counts(int *launches_seen_by_kind) {
    for (int k = 0; k < GPR_PROF_NKINDS; k++) launches_seen_by_kind[k] = g_prof_seen[k];
    return GPR_OK;
}
