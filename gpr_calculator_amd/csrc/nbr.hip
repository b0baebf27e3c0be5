// nbr.hip -- periodic neighbour search and `seq` bookkeeping for batches of structures: the index
// arithmetic SO3.build_neighbor_list does through ase.neighborlist.NeighborList
// (gpr_calc/SO3.py:348-407: cutoffs rcut/2 per atom, bothways, no self interaction, skin 0).
// Two builders with identical output: gpr_nbr_build (host threads, host arrays) and
// gpr_nbr_build_device (one wavefront per centre atom, results stay in HBM and feed
// gpr_so3_power_spectrum and the force-point gather of gpr_set_create_gather without a host round
// trip).  No floating-point model arithmetic happens here beyond the distance predicate.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <map>
#include <mutex>
#include <thread>
#include <unordered_map>
#include "gpr_internal.h"

struct nbr_sinfo { int atom0, n, img0, nimg, zero_img, pad; };

struct gpr_nbr {
    int nstruct = 0, natoms = 0;
    std::vector<double> vec, w;                 // [P,3], [P]
    std::vector<int> centre_pair0, centre_row0; // [natoms+1]
    std::vector<int> row_pair0;                 // [S+1]
    std::vector<int> self_row;                  // [natoms]
    std::vector<int> seq;                       // [S,2] structure-local atom ids
    std::vector<int> struct_row0;               // [nstruct+1]
    // device-built variant: everything above except centre_row0 / struct_row0 stays empty on the host
    bool on_device = false;
    hipStream_t st = nullptr;
    int P = 0, S = 0;
    char *d_in = nullptr;                       // uploaded geometry blob
    const double *d_wpos = nullptr, *d_shift = nullptr;
    const int *d_numbers = nullptr, *d_sid = nullptr, *d_atom0 = nullptr;
    const nbr_sinfo *d_sinfo = nullptr;
    int *d_off = nullptr;                       // [centre_row0 (natoms+1) | centre_pair0 (natoms+1) | counts | err]
    double *d_vec = nullptr, *d_w = nullptr;
    int *d_rp0 = nullptr, *d_self = nullptr, *d_seq = nullptr, *d_err = nullptr;
    bool capped = false;                        // P, S are capacities; the true sizes live in d_err[1..3] only
    bool seq_in_vec = false;                    // d_seq shares d_vec's allocation (capped build)
    const int *d_extra = nullptr;               // caller's extra ints uploaded with the geometry (or NULL)
    bool fused_rows = false;                    // the one-launch build has written the force-point regrouping already
};

static void cross3(const double *a, const double *b, double *c) {
    c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
}

// wrapped positions and the periodic image table of one structure (shared by both builders).  The cell analysis
// (normals, volume, image counts, shift table) is kept when the next structure has the same cell and pbc -- the usual
// case in a band of images -- and only the positions are wrapped again.
struct nbr_geom {
    std::vector<double> wrapped, shifts;
    int nimg = 0, zero_img = 0;
    bool singular = false;
    // cell analysis of the last call
    bool have_cell = false, any = false, reused = false;      // reused: same cell and pbc as the call before (same shift table)
    double cell[9], perp[3][3], det = 0.0, inv_det = 0.0;
    int pb[3] = {0, 0, 0};
};

// dst: where the wrapped positions go (3 n doubles); NULL: into g.wrapped
static void nbr_geometry(int n, const double *positions, const double *cell, const int *pb, double rcut, nbr_geom &g,
                         double *dst = nullptr) {
    const bool same = g.have_cell && !g.singular && memcmp(g.cell, cell, sizeof(g.cell)) == 0 && g.pb[0] == pb[0] &&
                      g.pb[1] == pb[1] && g.pb[2] == pb[2];
    if (!dst) { g.wrapped.resize(3 * (size_t)n); dst = g.wrapped.data(); }
    memcpy(dst, positions, 3 * (size_t)n * sizeof(double));
    g.reused = same;
    if (!same) {
        g.shifts.clear(); g.nimg = 0; g.zero_img = 0; g.singular = false;
        memcpy(g.cell, cell, sizeof(g.cell)); g.pb[0] = pb[0]; g.pb[1] = pb[1]; g.pb[2] = pb[2];
        g.have_cell = true;
        g.any = pb[0] || pb[1] || pb[2];
        int nimg[3] = {0, 0, 0};
        if (g.any) {
            cross3(cell + 3, cell + 6, g.perp[0]); cross3(cell + 6, cell, g.perp[1]); cross3(cell, cell + 3, g.perp[2]);
            g.det = cell[0] * g.perp[0][0] + cell[1] * g.perp[0][1] + cell[2] * g.perp[0][2];
            const double vol = std::fabs(g.det);
            if (vol < 1e-12) { g.singular = true; return; }
            g.inv_det = 1.0 / g.det;
            for (int k = 0; k < 3; k++)          // perp[k]: normal of the plane spanned by the other two cell vectors
                if (pb[k]) {
                    const double area = std::sqrt(g.perp[k][0] * g.perp[k][0] + g.perp[k][1] * g.perp[k][1] + g.perp[k][2] * g.perp[k][2]);
                    nimg[k] = (int)std::ceil(rcut / (vol / area));
                }
        }
        for (int a = -nimg[0]; a <= nimg[0]; a++)
            for (int b = -nimg[1]; b <= nimg[1]; b++)
                for (int c = -nimg[2]; c <= nimg[2]; c++) {
                    if (a == 0 && b == 0 && c == 0) g.zero_img = g.nimg;
                    for (int k = 0; k < 3; k++) g.shifts.push_back(a * cell[k] + b * cell[3 + k] + c * cell[6 + k]);
                    g.nimg++;
                }
    }
    if (g.any) {
        // quick test with the rows of the inverse cell (within a few ulp of the quotients below): every periodic
        // fractional coordinate safely inside [0, 1) -> nothing to wrap, no division
        double c[3][3];
        for (int k = 0; k < 3; k++)
            for (int q = 0; q < 3; q++) c[k][q] = g.perp[k][q] * g.inv_det;
        const double lo[3] = {pb[0] ? 1e-9 : -1e300, pb[1] ? 1e-9 : -1e300, pb[2] ? 1e-9 : -1e300};
        const double hi[3] = {pb[0] ? 1.0 - 1e-9 : 1e300, pb[1] ? 1.0 - 1e-9 : 1e300, pb[2] ? 1.0 - 1e-9 : 1e300};
        for (int i = 0; i < n; i++) {
            double *p = dst + 3 * i;
            const double q0 = p[0] * c[0][0] + p[1] * c[0][1] + p[2] * c[0][2];
            const double q1 = p[0] * c[1][0] + p[1] * c[1][1] + p[2] * c[1][2];
            const double q2 = p[0] * c[2][0] + p[1] * c[2][1] + p[2] * c[2][2];
            if (q0 > lo[0] && q0 < hi[0] && q1 > lo[1] && q1 < hi[1] && q2 > lo[2] && q2 < hi[2]) continue;
            double f[3];
            for (int k = 0; k < 3; k++) f[k] = (p[0] * g.perp[k][0] + p[1] * g.perp[k][1] + p[2] * g.perp[k][2]) / g.det;   // the quotient itself
            for (int k = 0; k < 3; k++)
                if (pb[k]) {
                    const double fl = std::floor(f[k]);
                    for (int cc = 0; cc < 3; cc++) p[cc] -= fl * cell[3 * k + cc];
                }
        }
    }
}

// per-structure result (structure-local pair / row numbering), merged afterwards
struct nbr_local {
    std::vector<double> vec, w;
    std::vector<int> centre_pair0, centre_row0, row_pair0, self_row, seq;   // offsets start at 0
    bool singular = false;
};

static void nbr_one(int n, const double *positions, const double *cell, const int *pb, const int *numbers,
                    double rcut, int weight_on, nbr_local &out) {
    const double rc2 = rcut * rcut;
    nbr_geom g;
    nbr_geometry(n, positions, cell, pb, rcut, g);
    if (g.singular) { out.singular = true; return; }
    const std::vector<double> &wrapped = g.wrapped, &shifts = g.shifts;
    const int nI = g.nimg;
    out.centre_pair0.push_back(0); out.centre_row0.push_back(0); out.row_pair0.push_back(0);
    for (int i = 0; i < n; i++) {
        const double *pi = &wrapped[3 * i];
        for (int j = 0; j < n; j++) {
            const double *pj = &wrapped[3 * j];
            int found = 0;
            for (int im = 0; im < nI; im++) {
                if (i == j && im == g.zero_img) continue;
                // (pj - pi) + shift: exactly antisymmetric under (i <-> j, image -> -image), so the neighbour
                // relation is symmetric bit for bit (the device path relies on it)
                const double dx = (pj[0] - pi[0]) + shifts[3 * im], dy = (pj[1] - pi[1]) + shifts[3 * im + 1],
                             dz = (pj[2] - pi[2]) + shifts[3 * im + 2];
                if (dx * dx + dy * dy + dz * dz < rc2) {
                    out.vec.push_back(dx); out.vec.push_back(dy); out.vec.push_back(dz);
                    double wgt = numbers[j];
                    if (weight_on && numbers[j] != numbers[i]) wgt = -wgt;     // SO3.py:381-385
                    out.w.push_back(wgt);
                    found++;
                }
            }
            if (found || i == j) {          // one seq row per distinct neighbour, plus the (i, i) row
                if (i == j) out.self_row.push_back((int)out.seq.size() / 2);
                out.seq.push_back(i); out.seq.push_back(j);
                out.row_pair0.push_back((int)out.w.size());
            }
        }
        out.centre_pair0.push_back((int)out.w.size());
        out.centre_row0.push_back((int)out.seq.size() / 2);
    }
}

extern "C" int gpr_nbr_build(gpr_nbr **out, int nstruct, const int *atom0, const double *positions,
                             const double *cells, const int *pbc, const int *numbers, double rcut,
                             int weight_on) {
    GPR_REQUIRE(out && atom0 && (nstruct == 0 || (positions && cells && pbc && numbers)), "gpr_nbr_build: null pointer");
    GPR_REQUIRE(rcut > 0.0, "gpr_nbr_build: rcut must be positive");
    std::vector<nbr_local> loc(nstruct);
    auto work = [&](int s0, int s1) {
        for (int s = s0; s < s1; s++)
            nbr_one(atom0[s + 1] - atom0[s], positions + 3 * (size_t)atom0[s], cells + 9 * s, pbc + 3 * s,
                    numbers + atom0[s], rcut, weight_on, loc[s]);
    };
    // structures are independent: a few host threads (the search is O(n^2 images) per structure)
    int nthr = (int)std::thread::hardware_concurrency();
    if (nthr > 16) nthr = 16;
    if (nthr > nstruct / 4) nthr = nstruct / 4;
    if (nthr <= 1) work(0, nstruct);
    else {
        std::vector<std::thread> pool;
        for (int t = 0; t < nthr; t++)
            pool.emplace_back(work, (int)((long long)nstruct * t / nthr), (int)((long long)nstruct * (t + 1) / nthr));
        for (auto &th : pool) th.join();
    }
    gpr_nbr *nb = new gpr_nbr();
    nb->nstruct = nstruct;
    nb->natoms = nstruct ? atom0[nstruct] : 0;
    nb->centre_pair0.push_back(0); nb->centre_row0.push_back(0); nb->row_pair0.push_back(0); nb->struct_row0.push_back(0);
    for (int s = 0; s < nstruct; s++) {
        const nbr_local &l = loc[s];
        if (l.singular) { delete nb; gpr_set_error("gpr_nbr_build: periodic structure %d has a singular cell", s); return GPR_ERR_ARG; }
        const int p0 = (int)nb->w.size(), r0 = (int)nb->seq.size() / 2;
        nb->vec.insert(nb->vec.end(), l.vec.begin(), l.vec.end());
        nb->w.insert(nb->w.end(), l.w.begin(), l.w.end());
        nb->seq.insert(nb->seq.end(), l.seq.begin(), l.seq.end());
        for (size_t k = 1; k < l.centre_pair0.size(); k++) nb->centre_pair0.push_back(l.centre_pair0[k] + p0);
        for (size_t k = 1; k < l.centre_row0.size(); k++) nb->centre_row0.push_back(l.centre_row0[k] + r0);
        for (size_t k = 1; k < l.row_pair0.size(); k++) nb->row_pair0.push_back(l.row_pair0[k] + p0);
        for (int v : l.self_row) nb->self_row.push_back(v + r0);
        nb->struct_row0.push_back((int)nb->seq.size() / 2);
    }
    nb->P = (int)nb->w.size(); nb->S = (int)nb->seq.size() / 2;
    *out = nb;
    return GPR_OK;
}

// ================================================================ device builder
namespace {

struct nbrk_args {
    const double *wpos, *shift;
    const int *numbers, *sid;
    const nbr_sinfo *sinfo;
    int *cnt_pairs, *cnt_rows;       // count pass: per-centre totals
    const int *cr0, *cp0;            // fill pass: exclusive offsets
    double *vec, *w;
    int *rp0, *self_row, *seq;
    double rc2;
    int natoms, weight_on;
    const int *ovf;                  // fill pass of a capped build: non-zero = capacities exceeded, write nothing
};

__device__ inline int wave_incl_scan(int v, int lane) {
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(v, o);
        if (lane >= o) v += t;
    }
    return v;
}

// One wavefront per centre atom i; lane l looks at neighbours j = l, l + 64, ...; pairs of a centre are
// ordered by (j, image) and rows by j, like the host builder.  FILL = false only counts.
template <bool FILL>
__global__ void __launch_bounds__(64) nbr_centre_kernel(nbrk_args a) {
    const int i = blockIdx.x, lane = threadIdx.x;
    if (FILL && a.ovf && a.ovf[0]) return;
    const nbr_sinfo si = a.sinfo[a.sid[i]];
    const int il = i - si.atom0;
    const double pix = a.wpos[3 * (size_t)i], piy = a.wpos[3 * (size_t)i + 1], piz = a.wpos[3 * (size_t)i + 2];
    const int zi = a.numbers[i];
    const double *sh = a.shift + 3 * (size_t)si.img0;
    int pair_run = FILL ? a.cp0[i] : 0, row_run = FILL ? a.cr0[i] : 0;
    for (int jb = 0; jb < si.n; jb += 64) {
        const int j = jb + lane;
        const bool live = j < si.n;
        double dx0 = 0.0, dy0 = 0.0, dz0 = 0.0;
        if (live) {
            const double *pj = a.wpos + 3 * (size_t)(si.atom0 + j);
            dx0 = pj[0] - pix; dy0 = pj[1] - piy; dz0 = pj[2] - piz;
        }
        int found = 0;
        if (live)
            for (int im = 0; im < si.nimg; im++) {
                if (j == il && im == si.zero_img) continue;
                const double dx = dx0 + sh[3 * im], dy = dy0 + sh[3 * im + 1], dz = dz0 + sh[3 * im + 2];
                // no fused multiply-add: the predicate has to agree with the host builder to the bit
                const double r2 = __dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz));
                if (r2 < a.rc2) found++;
            }
        const int isrow = (live && (found > 0 || j == il)) ? 1 : 0;
        const int pincl = wave_incl_scan(found, lane), rincl = wave_incl_scan(isrow, lane);
        if (FILL) {
            int p = pair_run + pincl - found;
            const int row = row_run + rincl - isrow;
            if (isrow) {
                a.seq[2 * (size_t)row] = il; a.seq[2 * (size_t)row + 1] = j;
                a.rp0[row] = p;
                if (j == il) a.self_row[i] = row;
            }
            if (found) {
                const int zj = a.numbers[si.atom0 + j];
                const double wgt = (a.weight_on && zj != zi) ? -(double)zj : (double)zj;     // SO3.py:381-385
                for (int im = 0; im < si.nimg; im++) {
                    if (j == il && im == si.zero_img) continue;
                    const double dx = dx0 + sh[3 * im], dy = dy0 + sh[3 * im + 1], dz = dz0 + sh[3 * im + 2];
                    const double r2 = __dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz));
                    if (r2 < a.rc2) {
                        a.vec[3 * (size_t)p] = dx; a.vec[3 * (size_t)p + 1] = dy; a.vec[3 * (size_t)p + 2] = dz;
                        a.w[p] = wgt;
                        p++;
                    }
                }
            }
        }
        pair_run += __shfl(pincl, 63);
        row_run += __shfl(rincl, 63);
    }
    if (!FILL && lane == 0) { a.cnt_pairs[i] = pair_run; a.cnt_rows[i] = row_run; }
}

// geometry blob from page-locked host memory to the device by a kernel (a copy-engine transfer costs ~20 us of queue
// hand-over before the first compute kernel may start); also clears the missing-row counter
__global__ void __launch_bounds__(256) nbr_upload_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16, int *err) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n16) dst[i] = src[i];
    if (i == 0) err[0] = 0;
}

// exclusive scans of the per-centre counts: off = [cr0 (n+1) | cp0 (n+1)]; one workgroup.
// info = [overflow, pairs, rows] (true totals).  Capped build (cap_rows > 0): the arrays downstream were sized from a
// guess; when the totals exceed it every offset is zeroed -- each centre then has no pairs and no rows, so whatever is
// already queued behind this kernel runs over empty ranges -- and the caller repeats the build with exact sizes.
__global__ void __launch_bounds__(1024) nbr_scan_kernel(int n, const int *cnt_rows, const int *cnt_pairs, int *off, int *info,
                                                        int cap_pairs, int cap_rows, int *rp0) {
    __shared__ int wsum[2][16];
    __shared__ int carry[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 2) carry[tid] = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int k = base + tid;
        const int vr = k < n ? cnt_rows[k] : 0, vp = k < n ? cnt_pairs[k] : 0;
        const int ir = wave_incl_scan(vr, lane), ip = wave_incl_scan(vp, lane);
        if (lane == 63) { wsum[0][wave] = ir; wsum[1][wave] = ip; }
        __syncthreads();
        int wr = 0, wp = 0;
        for (int q = 0; q < wave; q++) { wr += wsum[0][q]; wp += wsum[1][q]; }
        const int cr = carry[0], cp = carry[1];
        if (k < n) { off[k] = cr + wr + ir - vr; off[n + 1 + k] = cp + wp + ip - vp; }
        __syncthreads();
        if (tid == 1023) { carry[0] = cr + wr + ir; carry[1] = cp + wp + ip; }
        __syncthreads();
    }
    const int S = carry[0], P = carry[1];
    const bool over = cap_rows > 0 && (S > cap_rows || P > cap_pairs);
    if (over)
        for (int k = tid; k < 2 * n + 2; k += 1024) off[k] = 0;
    else if (tid == 0) {
        off[n] = S; off[2 * n + 1] = P;
        if (rp0) rp0[S] = P;
    }
    if (tid == 0) { info[0] = over ? 1 : 0; info[1] = P; info[2] = S; }
}

// row offsets of a list of moving atoms: out[k] = sum_{k' < k} rows(atoms[k']), out[npts] = total; one workgroup
__global__ void __launch_bounds__(1024) nbr_point_rows_kernel(int npts, const int *atoms, const int *cr0, int *out) {
    __shared__ int wsum[16];
    __shared__ int carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < npts; base += 1024) {
        const int k = base + tid;
        int v = 0;
        if (k < npts) { const int a = atoms[k]; v = cr0[a + 1] - cr0[a]; }
        const int incl = wave_incl_scan(v, lane);
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int w0 = 0;
        for (int q = 0; q < wave; q++) w0 += wsum[q];
        const int c = carry;
        if (k < npts) out[k] = c + w0 + incl - v;
        __syncthreads();
        if (tid == 1023) carry = c + w0 + incl;
        __syncthreads();
    }
    if (tid == 0) out[npts] = carry;
}

struct frow_args {
    const int *atoms, *pt_row0;      // [npts], [npts+1]
    const int *cr0, *seq, *sid;
    const nbr_sinfo *sinfo;
    int *row_index, *x_index, *err;
    int npts, nrows;
};

// output row q = k-th row of force point p (moving atom a): the k-th neighbour i of a (rows of centre a are
// ordered by neighbour) and the seq row (i, a) found among the rows of centre i
__global__ void __launch_bounds__(256) nbr_force_rows_kernel(frow_args f) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= f.nrows || q >= f.pt_row0[f.npts]) return;      // nrows may be a capacity (capped build)
    int lo = 0, hi = f.npts;                   // last p with pt_row0[p] <= q
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (f.pt_row0[mid] <= q) lo = mid; else hi = mid;
    }
    const int a = f.atoms ? f.atoms[lo] : lo, k = q - f.pt_row0[lo];
    const int atom0 = f.sinfo[f.sid[a]].atom0, al = a - atom0;
    const int i = atom0 + f.seq[2 * (size_t)(f.cr0[a] + k) + 1];
    int r0 = f.cr0[i], r1 = f.cr0[i + 1], found = -1;
    while (r0 < r1) {
        const int mid = (r0 + r1) >> 1, jm = f.seq[2 * (size_t)mid + 1];
        if (jm == al) { found = mid; break; }
        if (jm < al) r0 = mid + 1; else r1 = mid;
    }
    if (found < 0) atomicAdd(f.err, 1);
    f.row_index[q] = found;
    f.x_index[q] = i;
}

}  // namespace

// ================================================================ the whole build in ONE launch
// One workgroup (16 wavefronts) per STRUCTURE does what upload + count + scan + fill + force-row regrouping did in five
// launches (each boundary of that dependent chain cost ~6 us of drain and dispatch on a 36 us path):
//   geometry of the structure from the page-locked blob into LDS (and on to the device copy later kernels read),
//   count per centre (one wavefront per centre, the loops of nbr_centre_kernel), exclusive scan inside the workgroup,
//   the structure's offsets among all structures by LOOKING BACK: every workgroup publishes its totals (tagged with the
//   call's epoch, so that nobody has to clear the slots -- which are memory of their own and never hold anything else) right
//   after its count, and sums the totals of its PREDECESSORS -- only of them: workgroups are dispatched in index order, so
//   whoever a workgroup waits for is resident or done, and the launch makes progress however few of its workgroups the
//   device holds at a time (the first form also read the totals of its successors, to decide everywhere alike whether the
//   guessed capacities hold; with four batches in flight two such launches, each partly resident, could then hold each
//   other's CUs until the polls ran out -- seen once under a profiler as a 1.4 s kernel),
//   fill (pairs ordered by (j, image), rows by j: bit-identical to the other builders), and -- the neighbour relation
//   being exactly symmetric -- the force-point regrouping by table look-up: row k of force point a is the seq row
//   (i, a) of a's k-th neighbour i, whose position among i's rows centre i noted in an n x n table in LDS while counting.
// Capacity overflow: a structure whose running totals (its predecessors' and its own) exceed the guess fills nothing and
// closes every one of its centres at the totals of the last structure that fitted -- as do all structures after it -- so
// the tables describe a prefix of the batch and empty ranges behind it, whatever is queued behind this kernel stays
// inside the arrays, and the last structure (which has seen every total) raises the flag and reports the true sizes:
// the caller discards the call and repeats the build with exact sizes.
struct nbrf_args {
    const char *blob_host;           // device alias of the page-locked blob [wpos | shifts | sinfo | numbers | sid | atom0]
    char *blob_dev;                  // its device copy
    long long off_shift, off_sinfo, off_num, off_sid, off_a0, bytes;
    int nstruct, natoms, weight_on, cap_pairs, cap_rows, nmax, nimg_max, dbg, use_mask, mask_off;
    unsigned epoch;
    double rc2;
    int *cr0, *cp0, *info;           // [natoms + 1] each; info = [missing, overflow, pairs, rows]
    unsigned long long *agg;         // [nstruct][3]: rows | pairs | missing, each tagged (epoch << 32) | value
    unsigned *ticket;                // counts the workgroups of every launch on this stream; this launch's start at ticket_base
    unsigned ticket_base;
    double *vec, *w;
    int *rp0, *self_row, *seq;
    int *row_index, *x_index;        // force-point regrouping (may be NULL)
    nbr_sinfo si[96];
};
constexpr int NBRF_WAVES = 16, NBRF_MAX_N = 128, NBRF_MAX_STRUCT = 96, NBRF_POLLS = 1 << 22;

__device__ __forceinline__ void so3_nbr_wave_lds_sync() {
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ bool nbrf_wait(const unsigned long long *p, unsigned epoch, unsigned &value) {
    for (int polls = 0; polls < NBRF_POLLS; polls++) {
        const unsigned long long v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(v >> 32) == epoch) { value = (unsigned)v; return true; }
        __builtin_amdgcn_s_sleep(2);
    }
    value = 0;
    return false;
}

__global__ void __launch_bounds__(64 * NBRF_WAVES) nbr_fused_kernel(nbrf_args a) {
    extern __shared__ __attribute__((aligned(16))) double nf_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // The structure a workgroup takes is the order in which it STARTED (a ticket), not its index: whoever it will wait for --
    // the holders of the earlier tickets -- is then resident or done by construction, wherever the dispatcher put the
    // workgroups and whatever else fills the device (an index only promises that per XCD).
    __shared__ int s_ticket;
    if (tid == 0) s_ticket = (int)(atomicAdd(a.ticket, 1u) - a.ticket_base);
    __syncthreads();
    const int s = s_ticket;
    if ((unsigned)s >= (unsigned)a.nstruct) {                 // the counter and the host's count disagree: say so instead of reading past the arguments
        if (tid == 0) { a.info[0] = 0x3ffffff0; a.info[1] = 1; a.info[2] = s; a.info[3] = (int)a.ticket_base; }
        return;
    }
    const nbr_sinfo si = a.si[s];
    const int n = si.n, nimg = si.nimg;
    double *pos = nf_smem;                                   // [3 nmax]
    double *sh = pos + 3 * a.nmax;                           // [3 nimg_max]
    int *num = (int *)(sh + 3 * a.nimg_max);                 // [nmax]
    int *cntR = num + a.nmax, *cntP = cntR + a.nmax, *offR = cntP + a.nmax, *offP = offR + a.nmax;
    int *misc = offP + a.nmax;                               // [8]: TR, TP, baseR, baseP, ok, missing
    unsigned short *li = (unsigned short *)(misc + 8);       // [n][n] local row of neighbour j among centre i's rows, 0xffff: none
    unsigned char *fnd = (unsigned char *)(li + a.nmax * a.nmax);   // [n][n] images of j within the cutoff of i
    // [n][n] which images, when they fit a 64-bit mask and the table fits LDS (else NULL: the fill counts again)
    unsigned long long *msk = a.use_mask ? (unsigned long long *)(nf_smem + a.mask_off) : nullptr;
    // ---- geometry: host blob -> LDS (+ device copy)
    {
        const double *hw = (const double *)a.blob_host + 3 * (size_t)si.atom0, *hs = (const double *)(a.blob_host + a.off_shift) + 3 * (size_t)si.img0;
        const int *hn = (const int *)(a.blob_host + a.off_num) + si.atom0;
        double *dw = (double *)a.blob_dev + 3 * (size_t)si.atom0;
        int *dn = (int *)(a.blob_dev + a.off_num) + si.atom0, *dsid = (int *)(a.blob_dev + a.off_sid) + si.atom0;
        // the three reads cross PCIe: all of them are requested before anything waits for one (the usual structure needs
        // one pass of each loop), and the LDS tables are cleared while they are on their way
        const bool p0 = tid < 3 * n, p1 = tid < 3 * nimg, p2 = tid < n;
        const double v0 = p0 ? hw[tid] : 0.0, v1 = p1 ? hs[tid] : 0.0;
        const int z0 = p2 ? hn[tid] : 0;
        for (int e = tid; e < n * n; e += blockDim.x) { li[e] = 0xffff; fnd[e] = 0; }
        if (p0) { pos[tid] = v0; dw[tid] = v0; }
        if (p1) sh[tid] = v1;
        if (p2) { num[tid] = z0; dn[tid] = z0; dsid[tid] = s; }
        for (int e = tid + blockDim.x; e < 3 * n; e += blockDim.x) { const double v = hw[e]; pos[e] = v; dw[e] = v; }
        for (int e = tid + blockDim.x; e < 3 * nimg; e += blockDim.x) sh[e] = hs[e];
        for (int e = tid + blockDim.x; e < n; e += blockDim.x) { const int z = hn[e]; num[e] = z; dn[e] = z; dsid[e] = s; }
        if (s == 0) {            // what is shared by all structures: shift table, structure table, atom offsets
            const long long lo = a.off_shift, hi = a.off_num;
            for (long long e = lo + 8LL * tid; e < hi; e += 8LL * blockDim.x) *(double *)(a.blob_dev + e) = *(const double *)(a.blob_host + e);
            for (long long e = a.off_a0 + 4LL * tid; e < a.bytes; e += 4LL * blockDim.x) *(int *)(a.blob_dev + e) = *(const int *)(a.blob_host + e);
        }
        if (tid < 8) misc[tid] = 0;
    }
    __syncthreads();
    GPR_PHASE_EXIT(a.dbg, 1)
    // ---- count: one wavefront per centre, a lane per neighbour j walking its images; which images are inside the cutoff is
    // kept as a bit mask per (centre, neighbour) where it fits (<= 64 images, <= 64 atoms): the fill then tests nothing twice
    for (int il = wave; il < n; il += NBRF_WAVES) {
        const double pix = pos[3 * il], piy = pos[3 * il + 1], piz = pos[3 * il + 2];
        int pair_run = 0, row_run = 0;
        for (int jb = 0; jb < n; jb += 64) {
            const int j = jb + lane;
            const bool live = j < n;
            int found = 0;
            unsigned long long bits = 0ull;
            if (live) {
                const double dx0 = pos[3 * j] - pix, dy0 = pos[3 * j + 1] - piy, dz0 = pos[3 * j + 2] - piz;
                for (int im = 0; im < nimg; im++) {
                    if (j == il && im == si.zero_img) continue;
                    const double dx = dx0 + sh[3 * im], dy = dy0 + sh[3 * im + 1], dz = dz0 + sh[3 * im + 2];
                    // no fused multiply-add: the predicate has to agree with the host builder to the bit
                    const double r2 = __dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz));
                    if (r2 < a.rc2) { found++; bits |= 1ull << (im & 63); }
                }
                fnd[il * n + j] = (unsigned char)(found > 0);
                if (msk) msk[il * n + j] = bits;
            }
            const int isrow = (live && (found > 0 || j == il)) ? 1 : 0;
            const int pincl = wave_incl_scan(found, lane), rincl = wave_incl_scan(isrow, lane);
            if (isrow) li[il * n + j] = (unsigned short)(row_run + rincl - isrow);
            pair_run += __shfl(pincl, 63);
            row_run += __shfl(rincl, 63);
        }
        if (lane == 0) { cntP[il] = pair_run; cntR[il] = row_run; }
    }
    __syncthreads();
    GPR_PHASE_EXIT(a.dbg, 2)
    // ---- offsets inside the structure, totals out, predecessors' totals in
    if (wave == 0) {
        int runR = 0, runP = 0;
        for (int b = 0; b < n; b += 64) {
            const int i = b + lane;
            const int vr = i < n ? cntR[i] : 0, vp = i < n ? cntP[i] : 0;
            const int ir = wave_incl_scan(vr, lane), ip = wave_incl_scan(vp, lane);
            if (i < n) { offR[i] = runR + ir - vr; offP[i] = runP + ip - vp; }
            runR += __shfl(ir, 63); runP += __shfl(ip, 63);
        }
        const unsigned long long tag = (unsigned long long)a.epoch << 32;
        if (lane == 0) {
            __hip_atomic_store(&a.agg[3 * s], tag | (unsigned)runR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&a.agg[3 * s + 1], tag | (unsigned)runP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // running totals over the structures 0 .. s (the predecessors' from their slots, this one's own): the exclusive value at s
        // is this structure's offset; the first structure whose inclusive totals exceed a capacity ends the part of the batch
        // that fits, at the exclusive totals there
        int baseR = 0, baseP = 0, validR = 0, validP = 0, over = 0, ok = 1, carryR = 0, carryP = 0;
        for (int b = 0; b <= s && !over; b += 64) {
            const int q = b + lane;
            unsigned vr = 0, vp = 0;
            if (q < s) { if (!nbrf_wait(&a.agg[3 * q], a.epoch, vr)) ok = 0; if (!nbrf_wait(&a.agg[3 * q + 1], a.epoch, vp)) ok = 0; }
            else if (q == s) { vr = (unsigned)runR; vp = (unsigned)runP; }
            const int inR = carryR + wave_incl_scan((int)vr, lane), inP = carryP + wave_incl_scan((int)vp, lane);
            const unsigned long long bad = __ballot(q <= s && (inR > a.cap_rows || inP > a.cap_pairs));
            if (bad) {
                const int f = __ffsll((long long)bad) - 1;
                validR = __shfl(inR - (int)vr, f); validP = __shfl(inP - (int)vp, f);
                over = 1;
            } else if (s - b < 64) {
                baseR = __shfl(inR - (int)vr, s - b); baseP = __shfl(inP - (int)vp, s - b);
            }
            carryR = __shfl(inR, 63); carryP = __shfl(inP, 63);
        }
        for (int o = 32; o > 0; o >>= 1) ok &= __shfl_xor(ok, o);
        if (lane == 0) { misc[0] = runR; misc[1] = runP; misc[2] = over ? validR : baseR; misc[3] = over ? validP : baseP; misc[4] = ok; misc[6] = over; }
    }
    __syncthreads();
    GPR_PHASE_EXIT(a.dbg, 3)
    const int TR = misc[0], TP = misc[1], baseR = misc[2], baseP = misc[3];
    // does not fit (or a wait gave up): every centre of this structure is closed at the totals of the part that fitted
    const bool over = !misc[4] || misc[6];
    {
        // a workgroup writes the END of each of its centres (entry i + 1): the entry between two structures then comes from
        // the earlier one, which knows it whether or not the later one fits
        for (int i = tid; i < n; i += blockDim.x) {
            const int er = i + 1 < n ? offR[i + 1] : TR, ep = i + 1 < n ? offP[i + 1] : TP;
            a.cr0[si.atom0 + i + 1] = over ? baseR : baseR + er;
            a.cp0[si.atom0 + i + 1] = over ? baseP : baseP + ep;
        }
        if (s == 0 && tid == 0) { a.cr0[0] = 0; a.cp0[0] = 0; }
        if (s == a.nstruct - 1 && tid == 0) {
            // the last structure has seen every total: true sizes (its prefix + itself when everything fitted; else unknown
            // beyond the point of overflow -- the totals are then summed from the slots below)
            a.info[1] = over ? 1 : 0;
        }
        if (!over && tid == 0) a.rp0[baseR + TR] = baseP + TP;
    }
    int missing = 0;
    if (!over) {
        // ---- fill: pairs ordered by (j, image), rows by j
        if (msk) {
            for (int il = wave; il < n; il += NBRF_WAVES) {
                const int i = si.atom0 + il, zi = num[il];
                const double pix = pos[3 * il], piy = pos[3 * il + 1], piz = pos[3 * il + 2];
                int pair_run = baseP + offP[il];
                const int row0 = baseR + offR[il];
                for (int jb = 0; jb < n; jb += 64) {
                    const int j = jb + lane;
                    const bool live = j < n;
                    unsigned long long bits = live ? msk[il * n + j] : 0ull;
                    const int found = __popcll(bits);
                    const int pincl = wave_incl_scan(found, lane);
                    const unsigned short lr = live ? li[il * n + j] : (unsigned short)0xffff;
                    if (lr != 0xffff) {
                        const int row = row0 + lr;
                        int p = pair_run + pincl - found;
                        a.seq[2 * (size_t)row] = il; a.seq[2 * (size_t)row + 1] = j;
                        a.rp0[row] = p;
                        if (j == il) a.self_row[i] = row;
                        if (found) {
                            const int zj = num[j];
                            const double wgt = (a.weight_on && zj != zi) ? -(double)zj : (double)zj;     // SO3.py:381-385
                            const double dx0 = pos[3 * j] - pix, dy0 = pos[3 * j + 1] - piy, dz0 = pos[3 * j + 2] - piz;
                            while (bits) {                        // images in ascending order: the order of the pairs
                                const int im = __ffsll((long long)bits) - 1;
                                bits &= bits - 1;
                                a.vec[3 * (size_t)p] = dx0 + sh[3 * im]; a.vec[3 * (size_t)p + 1] = dy0 + sh[3 * im + 1];
                                a.vec[3 * (size_t)p + 2] = dz0 + sh[3 * im + 2];
                                a.w[p] = wgt;
                                p++;
                            }
                        }
                    }
                    pair_run += __shfl(pincl, 63);
                }
            }
        } else {
            for (int il = wave; il < n; il += NBRF_WAVES) {
                const int i = si.atom0 + il, zi = num[il];
                const double pix = pos[3 * il], piy = pos[3 * il + 1], piz = pos[3 * il + 2];
                int pair_run = baseP + offP[il];
                const int row0 = baseR + offR[il];
                for (int jb = 0; jb < n; jb += 64) {
                    const int j = jb + lane;
                    const bool live = j < n;
                    // the images are counted again (the table keeps one bit per pair of atoms): same arithmetic, same answer
                    int found = 0;
                    double dx0 = 0.0, dy0 = 0.0, dz0 = 0.0;
                    if (live && fnd[il * n + j]) {
                        dx0 = pos[3 * j] - pix; dy0 = pos[3 * j + 1] - piy; dz0 = pos[3 * j + 2] - piz;
                        for (int im = 0; im < nimg; im++) {
                            if (j == il && im == si.zero_img) continue;
                            const double dx = dx0 + sh[3 * im], dy = dy0 + sh[3 * im + 1], dz = dz0 + sh[3 * im + 2];
                            const double r2 = __dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz));
                            if (r2 < a.rc2) found++;
                        }
                    }
                    const int pincl = wave_incl_scan(found, lane);
                    const unsigned short lr = live ? li[il * n + j] : (unsigned short)0xffff;
                    if (lr != 0xffff) {
                        const int row = row0 + lr;
                        int p = pair_run + pincl - found;
                        a.seq[2 * (size_t)row] = il; a.seq[2 * (size_t)row + 1] = j;
                        a.rp0[row] = p;
                        if (j == il) a.self_row[i] = row;
                        if (found) {
                            const int zj = num[j];
                            const double wgt = (a.weight_on && zj != zi) ? -(double)zj : (double)zj;     // SO3.py:381-385
                            for (int im = 0; im < nimg; im++) {
                                if (j == il && im == si.zero_img) continue;
                                const double dx = dx0 + sh[3 * im], dy = dy0 + sh[3 * im + 1], dz = dz0 + sh[3 * im + 2];
                                const double r2 = __dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz));
                                if (r2 < a.rc2) {
                                    a.vec[3 * (size_t)p] = dx; a.vec[3 * (size_t)p + 1] = dy; a.vec[3 * (size_t)p + 2] = dz;
                                    a.w[p] = wgt;
                                    p++;
                                }
                            }
                        }
                    }
                    pair_run += __shfl(pincl, 63);
                }
            }
        }
        GPR_PHASE_EXIT(a.dbg, 4)
        // ---- force points: row k of point a (atom a) is the seq row (i, a) of a's k-th neighbour i
        if (a.row_index) {
            for (int e = tid; e < n * n; e += blockDim.x) {
                const int al = e / n, il = e - al * n;
                const unsigned short k = li[al * n + il];
                if (k == 0xffff) continue;
                const unsigned short back = li[il * n + al];
                const int q = baseR + offR[al] + k;
                a.row_index[q] = back == 0xffff ? -1 : baseR + offR[il] + back;
                a.x_index[q] = si.atom0 + il;
                if (back == 0xffff) missing++;
            }
        }
    }
    // ---- rows the symmetry check could not find (zero unless something is broken), summed by the last structure
    for (int o = 32; o > 0; o >>= 1) missing += __shfl_xor(missing, o);
    if (lane == 0 && missing) atomicAdd(&misc[5], missing);
    __syncthreads();
    if (tid == 0 && !misc[4]) misc[5] = 0x3fffffff;          // gave up waiting for a predecessor: fail loudly (the host reports the count)
    __syncthreads();
    if (tid == 0)
        __hip_atomic_store(&a.agg[3 * s + 2], ((unsigned long long)a.epoch << 32) | (unsigned)misc[5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (s == a.nstruct - 1 && wave == 0) {
        int tot = 0, ok = 1, allR = 0, allP = 0;
        for (int b = 0; b < a.nstruct; b += 64) {
            const int q = b + lane;
            unsigned v = 0, vr = 0, vp = 0;
            if (q < a.nstruct) {
                if (!nbrf_wait(&a.agg[3 * q + 2], a.epoch, v)) ok = 0;
                if (!nbrf_wait(&a.agg[3 * q], a.epoch, vr)) ok = 0;
                if (!nbrf_wait(&a.agg[3 * q + 1], a.epoch, vp)) ok = 0;
            }
            tot += (int)v; allR += (int)vr; allP += (int)vp;
        }
        for (int o = 32; o > 0; o >>= 1) {
            tot += __shfl_xor(tot, o); ok &= __shfl_xor(ok, o); allR += __shfl_xor(allR, o); allP += __shfl_xor(allP, o);
        }
        if (lane == 0) { a.info[0] = tot; a.info[2] = allP; a.info[3] = allR; if (!ok) a.info[1] = 1; }       // [missing, overflow, pairs, rows]
    }
}

namespace {
struct stage_buf { void *p = nullptr, *dev = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; bool pending = false; };   // dev: device alias of p
// as many as a prediction session may have in flight (predict.hip: SESSION_SLOTS): a buffer is re-used only after the
// submission that filled it has been waited for, or else after its event
constexpr int NSTAGE = 8;
thread_local stage_buf g_stage[NSTAGE];
thread_local int g_stage_next = 0;
}

// ---- when may a build take the one-launch kernel?  Its workgroups wait for their PREDECESSORS only (ticket order, see the
// kernel), so a launch needs no particular number of them resident at the same time and any number of such launches may
// share the device with each other and with anything else.  What the host still does:
//   * raises the kernel's dynamic-LDS limit when a build needs more than the 64 KB default and the device has it, and
//     declines the build when it does not (a declined build takes the five-launch path: same tables, bit for bit);
//   * keeps the residency figure (occupancy x CU count for the build's LDS need) and the count of launches in flight that the
//     first form of the kernel -- whose workgroups waited for ALL the others -- needed as an admission rule; the rule
//     (all such workgroups in flight within half of the residency) applies only while a test pretends a capacity
//     (gpr_nbr_debug_fused_capacity), which is how the declined path is exercised.
namespace {
struct fused_flight { hipEvent_t ev = nullptr; int wgs = 0; bool busy = false; };
std::mutex g_fused_mu;
fused_flight g_fused_flight[32];
std::map<std::pair<int, size_t>, int> g_fused_capacity;       // (device, lds) -> resident workgroups
std::map<int, size_t> g_fused_lds_attr;                      // device -> dynamic-LDS limit set on the kernel so far
std::atomic<int> g_fused_forced_capacity{-1};
}
// tests: pretend the device holds `workgroups` of the one-launch kernel at a time (< 0: ask the runtime, the default)
extern "C" int gpr_nbr_debug_fused_capacity(int workgroups) { g_fused_forced_capacity.store(workgroups); return GPR_OK; }

// admits `nstruct` workgroups with `lds` bytes each or says no (0); on yes *slot names the record to arm after the launch
static int fused_admit(int nstruct, size_t lds, int *slot) {
    int devid = 0;
    if (hipGetDevice(&devid) != hipSuccess) { (void)hipGetLastError(); return 0; }
    std::lock_guard<std::mutex> lk(g_fused_mu);
    if (lds > (size_t)64 * 1024) {
        size_t &have = g_fused_lds_attr[devid];
        if (have < lds) {
            int limit = 0;
            if (hipDeviceGetAttribute(&limit, hipDeviceAttributeMaxSharedMemoryPerBlock, devid) != hipSuccess || (size_t)limit < lds ||
                hipFuncSetAttribute((const void *)nbr_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
                (void)hipGetLastError();
                return 0;
            }
            have = lds;
        }
    }
    const size_t lds_key = (lds + 4095) / 4096 * 4096;
    auto it = g_fused_capacity.find({devid, lds_key});
    if (it == g_fused_capacity.end()) {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, nbr_fused_kernel, 64 * NBRF_WAVES, lds_key) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, devid) != hipSuccess) { (void)hipGetLastError(); per_cu = 0; }
        it = g_fused_capacity.emplace(std::make_pair(devid, lds_key), per_cu * cus).first;
    }
    const int forced_cap = g_fused_forced_capacity.load();          // tests (gpr_nbr_debug_fused_capacity)
    if (forced_cap < 0) { *slot = -1; return 1; }                   // no admission rule (see above)
    const int capacity = forced_cap;
    // launches counted as in flight: first by the records alone (an upper bound -- finished launches are only struck off when
    // somebody asks the runtime), and only if that bound does not admit the build by asking (an event query costs about a
    // microsecond; a prediction step keeps two or three launches in flight and never gets here)
    int flying = 0, free_slot = -1;
    for (int pass = 0; pass < 2; pass++) {
        flying = 0; free_slot = -1;
        for (int k = 0; k < 32; k++) {
            fused_flight &f = g_fused_flight[k];
            if (pass == 1 && f.busy && hipEventQuery(f.ev) != hipErrorNotReady) f.busy = false;      // finished (or failed: the stream will say)
            if (f.busy) flying += f.wgs; else if (free_slot < 0) free_slot = k;
        }
        if (free_slot >= 0 && 2 * (flying + nstruct) <= capacity) break;
    }
    (void)hipGetLastError();
    if (free_slot < 0 || 2 * (flying + nstruct) > capacity) return 0;
    *slot = free_slot;
    return 1;
}
// `ev`: the event the builder records behind the launch anyway (it guards the staging buffer; a later re-recording of it
// only makes this record look busy for longer)
static void fused_arm(int slot, int nstruct, hipEvent_t ev) {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_fused_mu);
    fused_flight &f = g_fused_flight[slot];
    f.ev = ev; f.wgs = nstruct; f.busy = true;
}
// resident workgroups the runtime reported for the last admitted shape on this device, launches counted as in flight (tests)
extern "C" int gpr_nbr_fused_state(int *capacity, int *in_flight) {
    int devid = 0;
    (void)hipGetDevice(&devid);
    std::lock_guard<std::mutex> lk(g_fused_mu);
    int cap = 0, fly = 0;
    for (auto &kv : g_fused_capacity) if (kv.first.first == devid) cap = std::max(cap, kv.second);
    for (fused_flight &f : g_fused_flight) {
        if (f.busy && hipEventQuery(f.ev) != hipErrorNotReady) f.busy = false;
        if (f.busy) fly += f.wgs;
    }
    (void)hipGetLastError();
    if (capacity) *capacity = cap;
    if (in_flight) *in_flight = fly;
    return GPR_OK;
}

static std::atomic<unsigned> g_nbr_epoch{1};
static std::atomic<int> g_nbr_fused{1};
// tests / A-B timing: 0 keeps every capped build on the five-launch path, 1 (default) lets the one-launch builder take
// the builds it can
extern "C" int gpr_nbr_set_fused(int on) { g_nbr_fused.store(on ? 1 : 0); return GPR_OK; }

static int nbr_build_device_impl(gpr_nbr **out, void *stream, int nstruct, const int *atom0, const double *positions,
                                 const double *cells, const int *pbc, const int *numbers, double rcut,
                                 int weight_on, int cap_pairs, int cap_rows, int n_extra = 0, const int *extra = nullptr,
                                 int *row_index_dev = nullptr, int *x_index_dev = nullptr) {
    GPR_REQUIRE(out && atom0 && (nstruct == 0 || (positions && cells && pbc && numbers)), "gpr_nbr_build_device: null pointer");
    GPR_REQUIRE(rcut > 0.0, "gpr_nbr_build_device: rcut must be positive");
    hipStream_t st = (hipStream_t)stream;
    const int natoms = nstruct ? atom0[nstruct] : 0;
    // host: wrap positions, image tables; one blob [wpos | shifts | sinfo | numbers | sid]
    std::vector<nbr_sinfo> sinfo(nstruct);
    std::vector<double> wpos(3 * (size_t)natoms), shifts;
    std::vector<int> sid(natoms);
    nbr_geom g;
    for (int s = 0; s < nstruct; s++) {
        const int n = atom0[s + 1] - atom0[s];
        GPR_REQUIRE(n >= 0, "gpr_nbr_build_device: atom0 not ascending at structure %d", s);
        nbr_geometry(n, positions + 3 * (size_t)atom0[s], cells + 9 * s, pbc + 3 * s, rcut, g, wpos.data() + 3 * (size_t)atom0[s]);
        if (g.singular) { gpr_set_error("gpr_nbr_build_device: periodic structure %d has a singular cell", s); return GPR_ERR_ARG; }
        // structures with the cell and pbc of their predecessor (a band of images) share its shift table
        const int img0 = (g.reused && s > 0) ? sinfo[s - 1].img0 : (int)(shifts.size() / 3);
        sinfo[s] = nbr_sinfo{atom0[s], n, img0, g.nimg, g.zero_img, 0};
        if (!(g.reused && s > 0)) shifts.insert(shifts.end(), g.shifts.begin(), g.shifts.end());
        for (int k = 0; k < n; k++) sid[atom0[s] + k] = s;
    }
    const size_t b_wpos = wpos.size() * 8, b_shift = shifts.size() * 8, b_sinfo = sinfo.size() * sizeof(nbr_sinfo),
                 b_num = (size_t)natoms * 4;
    const size_t b_a0 = (size_t)(nstruct + 1) * 4;
    const size_t b_extra = (size_t)n_extra * 4;         // caller's ints riding along with the blob (gpr_nbr_extra_ptr)
    const size_t b_total = b_wpos + b_shift + b_sinfo + 2 * b_num + b_a0 + b_extra;
    // the blob is assembled in page-locked memory (two buffers used in turn, each guarded by the event of its last
    // copy): the upload is then a real asynchronous copy instead of a staged one
    stage_buf &sb = g_stage[g_stage_next];
    g_stage_next = (g_stage_next + 1) % NSTAGE;
    if (sb.pending) { (void)hipEventSynchronize(sb.ev); sb.pending = false; }
    if (sb.cap < b_total) {
        if (sb.p) (void)hipHostFree(sb.p);
        sb.p = nullptr; sb.dev = nullptr; sb.cap = 0;
        const size_t want = std::max(b_total * 2, (size_t)1 << 16);
        if (hipHostMalloc(&sb.p, want, hipHostMallocDefault) != hipSuccess) {
            gpr_set_error("gpr_nbr_build_device: cannot allocate %zu bytes of page-locked memory", want);
            return GPR_ERR_HIP;
        }
        sb.cap = want;
    }
    if (!sb.ev && hipEventCreateWithFlags(&sb.ev, hipEventDisableTiming) != hipSuccess) {
        gpr_set_error("gpr_nbr_build_device: cannot create an event");
        return GPR_ERR_HIP;
    }
    char *bp = (char *)sb.p;
    if (b_wpos) memcpy(bp, wpos.data(), b_wpos);
    if (b_shift) memcpy(bp + b_wpos, shifts.data(), b_shift);
    if (b_sinfo) memcpy(bp + b_wpos + b_shift, sinfo.data(), b_sinfo);
    if (b_num) { memcpy(bp + b_wpos + b_shift + b_sinfo, numbers, b_num); memcpy(bp + b_wpos + b_shift + b_sinfo + b_num, sid.data(), b_num); }
    memcpy(bp + b_wpos + b_shift + b_sinfo + 2 * b_num, atom0, b_a0);
    if (b_extra) memcpy(bp + b_wpos + b_shift + b_sinfo + 2 * b_num + b_a0, extra, b_extra);

    gpr_nbr *nb = new gpr_nbr();
    nb->on_device = true; nb->st = st; nb->nstruct = nstruct; nb->natoms = natoms;
    auto fail = [&](int code) { gpr_nbr_destroy(nb); return code; };
    int rc;
    void *p;
    // one allocation for [geometry blob | offsets and counters | self rows]
    const size_t sz_in = (b_total + 16 + 255) / 256 * 256,
                 sz_off = ((size_t)(4 * natoms + 8) * sizeof(int) + (size_t)3 * (nstruct + 1) * sizeof(unsigned long long) + 16 + 255) / 256 * 256;
    if ((rc = gpr_dev_alloc(&p, sz_in + sz_off + (size_t)std::max(natoms, 1) * sizeof(int), st))) return fail(rc);
    nb->d_in = (char *)p;
    nb->d_off = (int *)(nb->d_in + sz_in);      // [cr0 | cp0 | cnt_rows | cnt_pairs | err | overflow, pairs, rows]
    nb->d_self = (int *)(nb->d_in + sz_in + sz_off);
    int *cnt_rows = nb->d_off + 2 * natoms + 2, *cnt_pairs = cnt_rows + natoms;
    nb->d_err = cnt_pairs + natoms;
    // ---- everything in one launch (nbr_fused_kernel) when the sizes are guessed, every atom moves and the structures
    // are of ordinary size; otherwise the launches below
    int nmax_atoms = 0, nimg_max = 1;
    for (int s = 0; s < nstruct; s++) { nmax_atoms = std::max(nmax_atoms, sinfo[s].n); nimg_max = std::max(nimg_max, sinfo[s].nimg); }
    static const bool no_fused = getenv("GPR_NBR_NO_FUSED") != nullptr;
    size_t lds = 0;
    int use_mask = 0, mask_off = 0, flight_slot = -1;
    bool fused = !no_fused && g_nbr_fused.load() && cap_pairs > 0 && cap_rows > 0 && natoms > 0 && n_extra == 0 && nstruct <= NBRF_MAX_STRUCT &&
                 nmax_atoms <= NBRF_MAX_N && nimg_max <= 512;
    if (fused) {
        lds = (size_t)(3 * nmax_atoms + 3 * nimg_max) * sizeof(double) + (size_t)(5 * nmax_atoms + 8) * sizeof(int) +
              (size_t)nmax_atoms * nmax_atoms * 3 + 16;
        use_mask = (nmax_atoms <= 64 && nimg_max <= 64) ? 1 : 0;
        lds = (lds + 7) / 8 * 8;
        mask_off = (int)(lds / 8);                           // in doubles from the start of the dynamic LDS
        if (use_mask) lds += (size_t)nmax_atoms * nmax_atoms * 8;
        fused = fused_admit(nstruct, lds, &flight_slot) != 0;      // residency and LDS: see above; no -> the five launches below
    }
    if (fused) {
        if (!sb.dev && hipHostGetDevicePointer(&sb.dev, sb.p, 0) != hipSuccess) { gpr_set_error("gpr_nbr_build_device: no device alias of the staging buffer"); return fail(GPR_ERR_HIP); }
        nb->capped = true; nb->P = cap_pairs; nb->S = cap_rows;
        if ((rc = gpr_dev_alloc(&p, (size_t)cap_pairs * 4 * sizeof(double) + ((size_t)3 * cap_rows + 4) * sizeof(int), st))) return fail(rc);
        nb->d_vec = (double *)p; nb->d_w = nb->d_vec + 3 * (size_t)cap_pairs;
        nb->d_seq = (int *)(nb->d_vec + 4 * (size_t)cap_pairs); nb->d_rp0 = nb->d_seq + 2 * (size_t)cap_rows;
        nb->seq_in_vec = true;
        nb->d_wpos = (const double *)nb->d_in;
        nb->d_shift = (const double *)(nb->d_in + b_wpos);
        nb->d_sinfo = (const nbr_sinfo *)(nb->d_in + b_wpos + b_shift);
        nb->d_numbers = (const int *)(nb->d_in + b_wpos + b_shift + b_sinfo);
        nb->d_sid = nb->d_numbers + natoms;
        nb->d_atom0 = nb->d_sid + natoms;
        nbrf_args fa{};
        unsigned *ticket_count = nullptr;
        fa.blob_host = (const char *)sb.dev; fa.blob_dev = nb->d_in;
        fa.off_shift = (long long)b_wpos; fa.off_sinfo = (long long)(b_wpos + b_shift); fa.off_num = (long long)(b_wpos + b_shift + b_sinfo);
        fa.off_sid = fa.off_num + (long long)b_num; fa.off_a0 = fa.off_sid + (long long)b_num; fa.bytes = (long long)b_total;
        fa.nstruct = nstruct; fa.natoms = natoms; fa.weight_on = weight_on; fa.cap_pairs = cap_pairs; fa.cap_rows = cap_rows;
        fa.nmax = nmax_atoms; fa.nimg_max = nimg_max; fa.rc2 = rcut * rcut;
        static const int nbr_dbg = GPR_PHASE_ENV("GPR_NBR_DBG");      // tools build only (gpr_internal.h); 0 otherwise
        fa.dbg = nbr_dbg;
        unsigned ep = g_nbr_epoch.fetch_add(1);
        if (ep == 0) ep = g_nbr_epoch.fetch_add(1);
        fa.epoch = ep;
        fa.cr0 = nb->d_off; fa.cp0 = nb->d_off + natoms + 1; fa.info = nb->d_err;
        // The look-back slots are tagged with the call's epoch instead of being cleared, so they must never hold anything
        // but earlier calls' tags: memory of their own, one small array per stream (launches of a stream are ordered),
        // zeroed once.  (They used to sit in the call's recycled temporaries, where an old row count in the upper half of
        // a word could equal a small epoch: a stale total was then taken for a published one.)
        // (the host's count of tickets handed out and the launch that consumes them go together: two host threads building on
        //  the same stream must not both start from the same count -- the lock is held until the launch is queued)
        struct lookback { unsigned long long *ag = nullptr; unsigned issued = 0; };     // slots + ticket word, tickets handed out so far
        static std::mutex lb_mu;
        static std::map<std::pair<int, hipStream_t>, lookback> lb_slots;                 // (device, stream): the null stream exists on every device
        int lb_dev = 0;
        if (hipGetDevice(&lb_dev) != hipSuccess) { (void)hipGetLastError(); lb_dev = 0; }
        std::unique_lock<std::mutex> lb_lock(lb_mu);
        {
            lookback &lb = lb_slots[{lb_dev, st}];
            if (!lb.ag) {
                // zeroed ON THE STREAM that will use them: a plain hipMemset runs on the null stream, which a non-blocking stream
                // does not wait for -- the first launch could then see the words before they were cleared, or have them cleared
                // under its feet (stale totals taken for published ones, or a wait that never ends)
                if (hipMalloc((void **)&lb.ag, (3 * NBRF_MAX_STRUCT + 1) * sizeof(unsigned long long)) != hipSuccess ||
                    hipMemsetAsync(lb.ag, 0, (3 * NBRF_MAX_STRUCT + 1) * sizeof(unsigned long long), st) != hipSuccess) {
                    lb.ag = nullptr;
                    gpr_set_error("gpr_nbr_build_device: cannot allocate the look-back slots");
                    return fail(GPR_ERR_HIP);
                }
                lb.issued = 0;
            }
            fa.agg = lb.ag;
            fa.ticket = (unsigned *)(lb.ag + 3 * NBRF_MAX_STRUCT);
            fa.ticket_base = lb.issued;
            ticket_count = &lb.issued;
        }
        fa.vec = nb->d_vec; fa.w = nb->d_w; fa.rp0 = nb->d_rp0; fa.self_row = nb->d_self; fa.seq = nb->d_seq;
        fa.row_index = row_index_dev; fa.x_index = row_index_dev ? x_index_dev : nullptr;
        for (int s = 0; s < nstruct; s++) fa.si[s] = sinfo[s];
        fa.use_mask = use_mask;
        fa.mask_off = mask_off;
        hipLaunchKernelGGL(nbr_fused_kernel, dim3(nstruct), dim3(64 * NBRF_WAVES), lds, st, fa);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) *ticket_count += (unsigned)nstruct;
        lb_lock.unlock();
        if (e == hipSuccess) { e = hipEventRecord(sb.ev, st); sb.pending = e == hipSuccess; }
        if (e == hipSuccess) fused_arm(flight_slot, nstruct, sb.ev);
        if (e != hipSuccess) { gpr_set_error("gpr_nbr_build_device: fused launch failed: %s", hipGetErrorString(e)); return fail(GPR_ERR_HIP); }
        nb->fused_rows = row_index_dev != nullptr;
        *out = nb;
        return GPR_OK;
    }
    {
        void *src_dev = nullptr;
        hipError_t e = hipSuccess;
        if (!sb.dev) e = hipHostGetDevicePointer(&sb.dev, sb.p, 0);
        src_dev = sb.dev;
        if (e == hipSuccess) {
            const size_t n16 = (b_total + 15) / 16;
            hipLaunchKernelGGL(nbr_upload_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, st, (const uint4 *)src_dev,
                               (uint4 *)nb->d_in, n16, nb->d_err);
            e = hipGetLastError();
        }
        if (e != hipSuccess) { gpr_set_error("gpr_nbr_build_device: upload failed: %s", hipGetErrorString(e)); return fail(GPR_ERR_HIP); }
    }
    // the event that guards the staging buffer is recorded behind the count pass (the stream is in order): between the
    // upload and the first kernel that needs it the host does nothing but launch
    auto guard_stage = [&]() {
        const hipError_t e = hipEventRecord(sb.ev, st);
        sb.pending = e == hipSuccess;
        return e;
    };
    nb->d_wpos = (const double *)nb->d_in;
    nb->d_shift = (const double *)(nb->d_in + b_wpos);
    nb->d_sinfo = (const nbr_sinfo *)(nb->d_in + b_wpos + b_shift);
    nb->d_numbers = (const int *)(nb->d_in + b_wpos + b_shift + b_sinfo);
    nb->d_sid = nb->d_numbers + natoms;
    nb->d_atom0 = nb->d_sid + natoms;
    nb->d_extra = n_extra ? nb->d_atom0 + nstruct + 1 : nullptr;

    nbrk_args a{};
    a.wpos = nb->d_wpos; a.shift = nb->d_shift; a.numbers = nb->d_numbers; a.sid = nb->d_sid; a.sinfo = nb->d_sinfo;
    a.cnt_pairs = cnt_pairs; a.cnt_rows = cnt_rows; a.rc2 = rcut * rcut; a.natoms = natoms; a.weight_on = weight_on;
    const bool capped = cap_pairs > 0 && cap_rows > 0 && natoms > 0;
    if (capped) {
        // sizes guessed by the caller: everything is queued without waiting for the count pass; the true sizes and the
        // overflow flag stay on the device (gpr_nbr_info_to)
        nb->capped = true; nb->P = cap_pairs; nb->S = cap_rows;
        hipLaunchKernelGGL(nbr_centre_kernel<false>, dim3(natoms), dim3(64), 0, st, a);       // the count pass needs no pair arrays
        hipError_t e = guard_stage();
        if (e != hipSuccess) { gpr_set_error("gpr_nbr_build_device: event failed: %s", hipGetErrorString(e)); return fail(GPR_ERR_HIP); }
        if ((rc = gpr_dev_alloc(&p, (size_t)cap_pairs * 4 * sizeof(double) + ((size_t)3 * cap_rows + 4) * sizeof(int), st))) return fail(rc);
        nb->d_vec = (double *)p; nb->d_w = nb->d_vec + 3 * (size_t)cap_pairs;
        nb->d_seq = (int *)(nb->d_vec + 4 * (size_t)cap_pairs); nb->d_rp0 = nb->d_seq + 2 * (size_t)cap_rows;
        nb->seq_in_vec = true;
        hipLaunchKernelGGL(nbr_scan_kernel, dim3(1), dim3(1024), 0, st, natoms, cnt_rows, cnt_pairs, nb->d_off, nb->d_err + 1,
                           cap_pairs, cap_rows, nb->d_rp0);
        a.cr0 = nb->d_off; a.cp0 = nb->d_off + natoms + 1; a.ovf = nb->d_err + 1;
        a.vec = nb->d_vec; a.w = nb->d_w; a.rp0 = nb->d_rp0; a.self_row = nb->d_self; a.seq = nb->d_seq;
        hipLaunchKernelGGL(nbr_centre_kernel<true>, dim3(natoms), dim3(64), 0, st, a);
        e = hipGetLastError();
        if (e != hipSuccess) { gpr_set_error("gpr_nbr_build_device: launch failed: %s", hipGetErrorString(e)); return fail(GPR_ERR_HIP); }
        *out = nb;
        return GPR_OK;
    }
    nb->centre_row0.assign(natoms + 1, 0);
    std::vector<int> h_off(2 * (size_t)natoms + 2, 0);
    {
        hipError_t e = guard_stage();
        if (e == hipSuccess && natoms > 0) {
            hipLaunchKernelGGL(nbr_centre_kernel<false>, dim3(natoms), dim3(64), 0, st, a);
            hipLaunchKernelGGL(nbr_scan_kernel, dim3(1), dim3(1024), 0, st, natoms, cnt_rows, cnt_pairs, nb->d_off, nb->d_err + 1,
                               0, 0, (int *)nullptr);
            e = hipMemcpyAsync(h_off.data(), nb->d_off, h_off.size() * sizeof(int), hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);      // the sizes decide every later allocation
        } else if (e == hipSuccess) e = hipMemsetAsync(nb->d_err + 1, 0, 3 * sizeof(int), st);
        if (e != hipSuccess) { gpr_set_error("gpr_nbr_build_device: count pass failed: %s", hipGetErrorString(e)); return fail(GPR_ERR_HIP); }
    }
    std::copy(h_off.begin(), h_off.begin() + natoms + 1, nb->centre_row0.begin());
    nb->S = h_off[natoms]; nb->P = h_off[2 * natoms + 1];
    nb->struct_row0.resize(nstruct + 1);
    for (int s = 0; s <= nstruct; s++) nb->struct_row0[s] = nb->centre_row0[atom0[s]];
    if ((rc = gpr_dev_alloc(&p, (size_t)std::max(nb->P, 1) * 4 * sizeof(double), st))) return fail(rc);
    nb->d_vec = (double *)p; nb->d_w = nb->d_vec + 3 * (size_t)std::max(nb->P, 1);
    if ((rc = gpr_dev_alloc(&p, ((size_t)3 * nb->S + 4) * sizeof(int), st))) return fail(rc);
    nb->d_seq = (int *)p; nb->d_rp0 = nb->d_seq + 2 * (size_t)nb->S;
    if (natoms > 0) {
        a.cr0 = nb->d_off; a.cp0 = nb->d_off + natoms + 1;
        a.vec = nb->d_vec; a.w = nb->d_w; a.rp0 = nb->d_rp0; a.self_row = nb->d_self; a.seq = nb->d_seq;
        hipLaunchKernelGGL(nbr_centre_kernel<true>, dim3(natoms), dim3(64), 0, st, a);
        // row_pair0[S] = P
        hipError_t e = hipMemcpyAsync(nb->d_rp0 + nb->S, nb->d_off + 2 * natoms + 1, sizeof(int), hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess) e = hipGetLastError();
        if (e != hipSuccess) { gpr_set_error("gpr_nbr_build_device: fill pass failed: %s", hipGetErrorString(e)); return fail(GPR_ERR_HIP); }
    }
    *out = nb;
    return GPR_OK;
}

extern "C" int gpr_nbr_build_device(gpr_nbr **out, void *stream, int nstruct, const int *atom0, const double *positions,
                                    const double *cells, const int *pbc, const int *numbers, double rcut,
                                    int weight_on) {
    return nbr_build_device_impl(out, stream, nstruct, atom0, positions, cells, pbc, numbers, rcut, weight_on, 0, 0);
}

// Same with caller-guessed capacities (pairs, seq rows): no host wait inside; gpr_nbr_sizes reports the capacities.
// When the guess was too small the handle describes an empty neighbour table (every centre has zero pairs and rows)
// and gpr_nbr_info_to reports overflow = 1 next to the true sizes; repeat with gpr_nbr_build_device or larger caps.
extern "C" int gpr_nbr_build_device_capped(gpr_nbr **out, void *stream, int nstruct, const int *atom0,
                                           const double *positions, const double *cells, const int *pbc,
                                           const int *numbers, double rcut, int weight_on, int cap_pairs, int cap_rows) {
    GPR_REQUIRE(cap_pairs > 0 && cap_rows > 0, "gpr_nbr_build_device_capped: capacities must be positive");
    return nbr_build_device_impl(out, stream, nstruct, atom0, positions, cells, pbc, numbers, rcut, weight_on, cap_pairs, cap_rows);
}

// builders with n_extra caller ints appended to the geometry upload (they reach the device with it, no copy of their own);
// cap_pairs = cap_rows = 0 selects the exact, synchronising build
extern "C" int gpr_nbr_build_device_extra(gpr_nbr **out, void *stream, int nstruct, const int *atom0, const double *positions,
                                          const double *cells, const int *pbc, const int *numbers, double rcut, int weight_on,
                                          int cap_pairs, int cap_rows, int n_extra, const int *extra_host) {
    GPR_REQUIRE(n_extra >= 0 && (n_extra == 0 || extra_host), "gpr_nbr_build_device_extra: bad extra payload");
    GPR_REQUIRE((cap_pairs > 0) == (cap_rows > 0), "gpr_nbr_build_device_extra: capacities go together");
    return nbr_build_device_impl(out, stream, nstruct, atom0, positions, cells, pbc, numbers, rcut, weight_on, cap_pairs, cap_rows,
                                 n_extra, extra_host);
}
// capped build of a batch in which every atom moves, with the force-point regrouping of gpr_nbr_force_rows(nb, natoms,
// NULL, NULL, cap_rows, row_index, x_index) written by the same launch where the one-launch builder applies;
// *rows_done tells whether it did (0: call gpr_nbr_force_rows as before)
extern "C" int gpr_nbr_build_device_rows(gpr_nbr **out, void *stream, int nstruct, const int *atom0, const double *positions,
                                         const double *cells, const int *pbc, const int *numbers, double rcut, int weight_on,
                                         int cap_pairs, int cap_rows, int *row_index_dev, int *x_index_dev, int *rows_done) {
    GPR_REQUIRE(cap_pairs > 0 && cap_rows > 0 && row_index_dev && x_index_dev && rows_done, "gpr_nbr_build_device_rows: bad arguments");
    const int rc = nbr_build_device_impl(out, stream, nstruct, atom0, positions, cells, pbc, numbers, rcut, weight_on, cap_pairs, cap_rows,
                                         0, nullptr, row_index_dev, x_index_dev);
    *rows_done = (rc == GPR_OK && *out && (*out)->fused_rows) ? 1 : 0;
    return rc;
}
extern "C" const int *gpr_nbr_extra_ptr(const gpr_nbr *nb) { return (nb && nb->on_device) ? nb->d_extra : nullptr; }

extern "C" int gpr_nbr_device_arrays(const gpr_nbr *nb, const double **vec, const double **w, const int **centre_pair0,
                                     const int **centre_row0, const int **row_pair0, const int **self_row,
                                     const int **seq, const int **numbers, const int **atom0) {
    GPR_REQUIRE(nb && nb->on_device, "gpr_nbr_device_arrays: handle was not built by gpr_nbr_build_device");
    if (atom0) *atom0 = nb->d_atom0;
    if (vec) *vec = nb->d_vec;
    if (w) *w = nb->d_w;
    if (centre_pair0) *centre_pair0 = nb->d_off + nb->natoms + 1;
    if (centre_row0) *centre_row0 = nb->d_off;
    if (row_pair0) *row_pair0 = nb->d_rp0;
    if (self_row) *self_row = nb->d_self;
    if (seq) *seq = nb->d_seq;
    if (numbers) *numbers = nb->d_numbers;
    return GPR_OK;
}

extern "C" int gpr_nbr_force_rows(const gpr_nbr *nb, int npts, const int *atoms_dev, const int *pt_row0_dev, int nrows,
                                  int *row_index_dev, int *x_index_dev) {
    GPR_REQUIRE(nb && nb->on_device, "gpr_nbr_force_rows: handle was not built by gpr_nbr_build_device");
    GPR_REQUIRE(npts >= 0 && nrows >= 0, "gpr_nbr_force_rows: negative sizes");
    if (nrows == 0 || npts == 0) return GPR_OK;
    GPR_REQUIRE(row_index_dev && x_index_dev, "gpr_nbr_force_rows: null device pointer");
    GPR_REQUIRE((atoms_dev && pt_row0_dev) || (!atoms_dev && !pt_row0_dev && npts == nb->natoms && nrows == nb->S),
                "gpr_nbr_force_rows: atoms and pt_row0 go together; both NULL means every atom moves (npts = natoms, nrows = S)");
    if (!pt_row0_dev) pt_row0_dev = nb->d_off;          // all atoms free: point a is atom a, its row range is centre_row0
    frow_args f{atoms_dev, pt_row0_dev, nb->d_off, nb->d_seq, nb->d_sid, nb->d_sinfo, row_index_dev, x_index_dev, nb->d_err,
                npts, nrows};
    hipLaunchKernelGGL(nbr_force_rows_kernel, dim3((nrows + 255) / 256), dim3(256), 0, nb->st, f);
    GPR_HIP_TRY(hipGetLastError());
    return GPR_OK;
}

// pt_row0_out_dev[npts + 1]: exclusive row offsets of the force points of the moving atoms atoms_dev[npts] (device),
// the input gpr_nbr_force_rows wants next to that list -- computed on the device so that nobody needs centre_row0 on
// the host
extern "C" int gpr_nbr_point_rows(const gpr_nbr *nb, int npts, const int *atoms_dev, int *pt_row0_out_dev) {
    GPR_REQUIRE(nb && nb->on_device, "gpr_nbr_point_rows: handle was not built by gpr_nbr_build_device");
    GPR_REQUIRE(npts >= 0 && pt_row0_out_dev && (npts == 0 || atoms_dev), "gpr_nbr_point_rows: bad arguments");
    hipLaunchKernelGGL(nbr_point_rows_kernel, dim3(1), dim3(1024), 0, nb->st, npts, atoms_dev, nb->d_off, pt_row0_out_dev);
    GPR_HIP_TRY(hipGetLastError());
    return GPR_OK;
}

namespace {
__global__ void nbr_status_kernel(const int *err, double *dst) { dst[0] = (double)err[0]; }
__global__ void nbr_info_kernel(const int *err, double *dst) { if (threadIdx.x < 4) dst[threadIdx.x] = (double)err[threadIdx.x]; }
}

extern "C" const int *gpr_nbr_info_ptr(const gpr_nbr *nb) { return (nb && nb->on_device) ? nb->d_err : nullptr; }

// [missing force rows, overflow, pairs, rows] as doubles into dst[0..3] on the handle's stream (see gpr_nbr_status_to)
extern "C" int gpr_nbr_info_to(const gpr_nbr *nb, double *dst_dev) {
    GPR_REQUIRE(nb && nb->on_device && dst_dev, "gpr_nbr_info_to: needs a device-built handle and a device pointer");
    hipLaunchKernelGGL(nbr_info_kernel, dim3(1), dim3(64), 0, nb->st, nb->d_err, dst_dev);
    GPR_HIP_TRY(hipGetLastError());
    return GPR_OK;
}

// device-side variant of gpr_nbr_check: writes the count (as a double) into dst[0] on the handle's stream, so that it
// can travel with the caller's own result download instead of costing a synchronisation of its own
extern "C" int gpr_nbr_status_to(const gpr_nbr *nb, double *dst_dev) {
    GPR_REQUIRE(nb && nb->on_device && dst_dev, "gpr_nbr_status_to: needs a device-built handle and a device pointer");
    hipLaunchKernelGGL(nbr_status_kernel, dim3(1), dim3(1), 0, nb->st, nb->d_err, dst_dev);
    GPR_HIP_TRY(hipGetLastError());
    return GPR_OK;
}

extern "C" int gpr_nbr_check(const gpr_nbr *nb, int *nmissing) {
    GPR_REQUIRE(nb && nb->on_device && nmissing, "gpr_nbr_check: needs a device-built handle");
    GPR_HIP_TRY(hipMemcpyAsync(nmissing, nb->d_err, sizeof(int), hipMemcpyDeviceToHost, nb->st));
    GPR_HIP_TRY(hipStreamSynchronize(nb->st));
    return GPR_OK;
}

extern "C" int gpr_nbr_sizes(const gpr_nbr *nb, int *npairs, int *nrows) {
    GPR_REQUIRE(nb, "gpr_nbr_sizes: null handle");
    if (npairs) *npairs = nb->P;
    if (nrows) *nrows = nb->S;
    return GPR_OK;
}

extern "C" int gpr_nbr_export(const gpr_nbr *nb, double *vec, double *w, int *centre_pair0, int *centre_row0,
                              int *row_pair0, int *self_row, int *seq, int *struct_row0) {
    GPR_REQUIRE(nb, "gpr_nbr_export: null handle");
    GPR_REQUIRE(!nb->capped, "gpr_nbr_export: a capped build keeps its sizes on the device; use gpr_nbr_build_device");
    auto cp = [](auto *dst, const auto &src) { if (dst && !src.empty()) std::copy(src.begin(), src.end(), dst); };
    cp(centre_row0, nb->centre_row0); cp(struct_row0, nb->struct_row0);
    if (!nb->on_device) {
        cp(vec, nb->vec); cp(w, nb->w); cp(centre_pair0, nb->centre_pair0);
        cp(row_pair0, nb->row_pair0); cp(self_row, nb->self_row); cp(seq, nb->seq);
        return GPR_OK;
    }
    // device-built: download what was asked for (a synchronising debug / single-structure path)
    if (!vec && !w && !centre_pair0 && !row_pair0 && !self_row && !seq) return GPR_OK;
    const int n = nb->natoms, P = nb->P, S = nb->S;
    hipStream_t st = nb->st;
    auto dl = [&](void *dst, const void *src, size_t bytes) -> hipError_t {
        if (!dst || bytes == 0) return hipSuccess;
        return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st);
    };
    GPR_HIP_TRY(dl(vec, nb->d_vec, (size_t)P * 24));
    GPR_HIP_TRY(dl(w, nb->d_w, (size_t)P * 8));
    GPR_HIP_TRY(dl(centre_pair0, nb->d_off + n + 1, (size_t)(n + 1) * 4));
    GPR_HIP_TRY(dl(row_pair0, nb->d_rp0, (size_t)(S + 1) * 4));
    GPR_HIP_TRY(dl(self_row, nb->d_self, (size_t)n * 4));
    GPR_HIP_TRY(dl(seq, nb->d_seq, (size_t)S * 8));
    GPR_HIP_TRY(hipStreamSynchronize(st));
    return GPR_OK;
}

extern "C" int gpr_nbr_destroy(gpr_nbr *nb) {
    if (!nb) return GPR_OK;
    if (nb->on_device) {      // stream-ordered frees: kernels already queued on the stream still see the arrays
        gpr_dev_free(nb->d_in, nb->st);               // d_off and d_self live in the same block
        gpr_dev_free(nb->d_vec, nb->st);
        if (!nb->seq_in_vec) gpr_dev_free(nb->d_seq, nb->st);
    }
    delete nb;
    return GPR_OK;
}
