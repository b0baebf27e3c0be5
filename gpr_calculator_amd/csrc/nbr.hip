// nbr.hip -- host-side (C++) periodic neighbour search and `seq` bookkeeping for batches of
// structures: the index arithmetic SO3.build_neighbor_list does through ase.neighborlist.NeighborList
// (gpr_calc/SO3.py:348-407: cutoffs rcut/2 per atom, bothways, no self interaction, skin 0).
// No floating-point model arithmetic happens here beyond the distance predicate; the outputs feed
// gpr_so3_power_spectrum.
#include <algorithm>
#include <cmath>
#include <thread>
#include "gpr_internal.h"

struct gpr_nbr {
    int nstruct = 0, natoms = 0;
    std::vector<double> vec, w;                 // [P,3], [P]
    std::vector<int> centre_pair0, centre_row0; // [natoms+1]
    std::vector<int> row_pair0;                 // [S+1]
    std::vector<int> self_row;                  // [natoms]
    std::vector<int> seq;                       // [S,2] structure-local atom ids
    std::vector<int> struct_row0;               // [nstruct+1]
};

static void cross3(const double *a, const double *b, double *c) {
    c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
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
    const bool any = pb[0] || pb[1] || pb[2];
    std::vector<double> wrapped(positions, positions + 3 * (size_t)n), shifts;
    std::vector<int> imgs;
    int nimg[3] = {0, 0, 0};
    if (any) {
        double c01[3], c12[3], c20[3];
        cross3(cell, cell + 3, c01); cross3(cell + 3, cell + 6, c12); cross3(cell + 6, cell, c20);
        const double det = cell[0] * c12[0] + cell[1] * c12[1] + cell[2] * c12[2];
        const double vol = std::fabs(det);
        if (vol < 1e-12) { out.singular = true; return; }
        const double *perp[3] = {c12, c20, c01};     // normal of the plane spanned by the other two
        for (int k = 0; k < 3; k++)
            if (pb[k]) {
                const double area = std::sqrt(perp[k][0] * perp[k][0] + perp[k][1] * perp[k][1] + perp[k][2] * perp[k][2]);
                nimg[k] = (int)std::ceil(rcut / (vol / area));
            }
        for (int i = 0; i < n; i++) {
            double *p = &wrapped[3 * i];
            double f[3];
            for (int k = 0; k < 3; k++) f[k] = (p[0] * perp[k][0] + p[1] * perp[k][1] + p[2] * perp[k][2]) / det;
            for (int k = 0; k < 3; k++)
                if (pb[k]) {
                    const double fl = std::floor(f[k]);
                    for (int c = 0; c < 3; c++) p[c] -= fl * cell[3 * k + c];
                }
        }
    }
    for (int a = -nimg[0]; a <= nimg[0]; a++)
        for (int b = -nimg[1]; b <= nimg[1]; b++)
            for (int c = -nimg[2]; c <= nimg[2]; c++) {
                imgs.push_back((a == 0 && b == 0 && c == 0) ? 1 : 0);
                for (int k = 0; k < 3; k++) shifts.push_back(a * cell[k] + b * cell[3 + k] + c * cell[6 + k]);
            }
    const int nI = (int)imgs.size();
    out.centre_pair0.push_back(0); out.centre_row0.push_back(0); out.row_pair0.push_back(0);
    for (int i = 0; i < n; i++) {
        const double *pi = &wrapped[3 * i];
        for (int j = 0; j < n; j++) {
            const double *pj = &wrapped[3 * j];
            int found = 0;
            for (int im = 0; im < nI; im++) {
                if (i == j && imgs[im]) continue;
                const double dx = pj[0] + shifts[3 * im] - pi[0], dy = pj[1] + shifts[3 * im + 1] - pi[1],
                             dz = pj[2] + shifts[3 * im + 2] - pi[2];
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
    *out = nb;
    return GPR_OK;
}

extern "C" int gpr_nbr_sizes(const gpr_nbr *nb, int *npairs, int *nrows) {
    GPR_REQUIRE(nb, "gpr_nbr_sizes: null handle");
    if (npairs) *npairs = (int)nb->w.size();
    if (nrows) *nrows = (int)nb->seq.size() / 2;
    return GPR_OK;
}

extern "C" int gpr_nbr_export(const gpr_nbr *nb, double *vec, double *w, int *centre_pair0, int *centre_row0,
                              int *row_pair0, int *self_row, int *seq, int *struct_row0) {
    GPR_REQUIRE(nb, "gpr_nbr_export: null handle");
    auto cp = [](auto *dst, const auto &src) { if (dst && !src.empty()) std::copy(src.begin(), src.end(), dst); };
    cp(vec, nb->vec); cp(w, nb->w); cp(centre_pair0, nb->centre_pair0); cp(centre_row0, nb->centre_row0);
    cp(row_pair0, nb->row_pair0); cp(self_row, nb->self_row); cp(seq, nb->seq); cp(struct_row0, nb->struct_row0);
    return GPR_OK;
}

extern "C" int gpr_nbr_destroy(gpr_nbr *nb) {
    delete nb;
    return GPR_OK;
}
