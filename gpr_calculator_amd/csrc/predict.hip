// predict.hip -- one native entry for the whole batched prediction of the Dot kernel at zeta = 2
// (GP.predict_structures, gaussianprocess.py:834-918 for many structures at once): neighbour search, descriptor,
// force-point regrouping, feature rows + prior variances, weight-space mean and variance, all queued on one stream
// from one host call.  It only composes the library's own entry points (nbr.hip, so3.hip, features.hip, dense.hip);
// the single synchronisation is the size download inside gpr_nbr_build_device, and none with guessed capacities.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>
#include "gpr_internal.h"

// set by the session call around gpr_predict_dot2_batch: the tail kernel then writes standard deviations (the session
// hands those to its caller; the square roots of 18 000 numbers are not the host's job)
static thread_local int t_std_out = 0;

extern "C" int gpr_predict_dot2_batch(void *stream, int nmax, int lmax, double rcut, double alpha, int nseg, int ncheb,
                                      const double *so3_table, int nstruct, const int *atom0, const double *positions,
                                      const double *cells, const int *pbc, const int *numbers, int weight_on,
                                      const unsigned char *free_mask, int ns, const int *species_host, double sigma,
                                      double sigma0, const double *w_dev, const double *M_dev, const double *Cinv_dev,
                                      int want_std, int cap_pairs, int cap_rows, double *out_dev, long long out_len) {
    hipStream_t st = (hipStream_t)stream;
    GPR_REQUIRE(nstruct >= 0 && atom0 && so3_table && species_host && w_dev && out_dev && (!want_std || M_dev || Cinv_dev),
                "gpr_predict_dot2_batch: null pointer");
    if (nstruct == 0) return GPR_OK;
    const int natoms = atom0[nstruct];
    const int d = nmax * (nmax + 1) / 2 * (lmax + 1);
    const int npair = d * (d + 1) / 2, F = ns * (npair + 1);
    int nfree = natoms;
    if (free_mask) { nfree = 0; for (int a = 0; a < natoms; a++) nfree += free_mask[a] ? 1 : 0; }
    const long long m = (long long)nstruct + 3LL * nfree;
    // species of the batch that the model does not know: no feature block, but part of the prior variance
    std::vector<int> extra;
    for (int a = 0; a < natoms; a++) {
        const int z = numbers[a];
        bool known = false;
        for (int k = 0; k < ns; k++) known = known || species_host[k] == z;
        if (!known && std::find(extra.begin(), extra.end(), z) == extra.end()) extra.push_back(z);
    }
    GPR_REQUIRE(out_len >= (want_std ? 2 : 1) * m + 4, "gpr_predict_dot2_batch: out holds %lld doubles, %lld needed",
                out_len, (want_std ? 2 : 1) * m + 4);

    // capacities guessed by the caller (from the sizes an earlier call reported): nothing below waits for the device
    gpr_nbr *nb = nullptr;
    const bool capped = cap_pairs > 0 && cap_rows > 0 && natoms > 0;
    // the list of moving atoms (fixed atoms present) travels with the geometry upload
    std::vector<int> fatoms;
    if (free_mask) {
        fatoms.reserve((size_t)nfree);
        for (int a = 0; a < natoms; a++)
            if (free_mask[a]) fatoms.push_back(a);
    }
    std::vector<void *> temps;
    auto cleanup = [&](int code) {
        for (void *p : temps) gpr_dev_free(p, st);
        if (nb) gpr_nbr_destroy(nb);
        return code;
    };
    auto dalloc = [&](size_t bytes, void **p) { int r = gpr_dev_alloc(p, bytes, st); if (!r) temps.push_back(*p); return r; };
    // one arena for everything this call needs besides the neighbour table: [psi | diag | x | dxdr | row indices | moving
    // atoms and their row offsets] -- a single stream-ordered allocation, so the launches below go out back to back
    auto up = [](size_t bytes) { return (bytes + 255) / 256 * 256; };
    char *arena = nullptr;
    size_t sz_psi = 0, sz_diag = 0, sz_x = 0, sz_dxdr = 0, sz_idx = 0;
    int frows = 0;
    auto make_arena = [&](int S) {
        frows = S;                              // bound on the force rows (the regrouping stops at the true total)
        sz_psi = up((size_t)m * F * sizeof(double)); sz_diag = up(want_std ? (size_t)m * sizeof(double) : 0);
        sz_x = up((size_t)std::max(natoms, 1) * d * sizeof(double)); sz_dxdr = up((size_t)std::max(S, 1) * d * 3 * sizeof(double));
        sz_idx = up((size_t)std::max(frows, 1) * 2 * sizeof(int));
        const size_t sz_meta = up(free_mask ? ((size_t)nfree + 2) * sizeof(int) : 0);
        void *p = nullptr;
        const int r = dalloc(sz_psi + sz_diag + sz_x + sz_dxdr + sz_idx + sz_meta, &p);
        arena = (char *)p;
        return r;
    };
    int rc, rows_done = 0;
    if (capped && !free_mask) {
        // sizes guessed and every atom moving: the arena first, then the whole neighbour search -- regrouping of the force
        // points included -- as ONE launch (gpr_nbr_build_device_rows)
        if ((rc = make_arena(cap_rows))) return cleanup(rc);
        int *ri = (int *)(arena + sz_psi + sz_diag + sz_x + sz_dxdr);
        rc = gpr_nbr_build_device_rows(&nb, stream, nstruct, atom0, positions, cells, pbc, numbers, rcut, weight_on, cap_pairs, cap_rows,
                                       ri, ri + std::max(frows, 1), &rows_done);
    } else {
        rc = gpr_nbr_build_device_extra(&nb, stream, nstruct, atom0, positions, cells, pbc, numbers, rcut, weight_on,
                                        capped ? cap_pairs : 0, capped ? cap_rows : 0, (int)fatoms.size(), fatoms.data());
    }
    if (rc) return cleanup(rc);

    int P = 0, S = 0;
    gpr_nbr_sizes(nb, &P, &S);
    const double *d_vec, *d_w; const int *d_cp0, *d_cr0, *d_rp0, *d_self, *d_seq, *d_num, *d_atom0;
    if ((rc = gpr_nbr_device_arrays(nb, &d_vec, &d_w, &d_cp0, &d_cr0, &d_rp0, &d_self, &d_seq, &d_num, &d_atom0))) return cleanup(rc);
    if (!arena && (rc = make_arena(S))) return cleanup(rc);
    void *p;
    double *d_psi = (double *)arena, *d_diag = want_std ? (double *)(arena + sz_psi) : nullptr, *d_q = nullptr;
    double *d_x = (double *)(arena + sz_psi + sz_diag), *d_dxdr = (double *)(arena + sz_psi + sz_diag + sz_x);
    int *d_rowidx = (int *)(arena + sz_psi + sz_diag + sz_x + sz_dxdr), *d_xidx = d_rowidx + std::max(frows, 1);
    int *d_meta = (int *)(arena + sz_psi + sz_diag + sz_x + sz_dxdr + sz_idx);

    // force points: every free atom; its rows are found on the device (gpr_nbr_force_rows)
    const int *d_fatoms = nullptr, *d_frow0 = nullptr;
    if (free_mask) {
        // row offsets of the moving atoms by a device scan
        d_fatoms = nfree > 0 ? gpr_nbr_extra_ptr(nb) : d_meta;
        if ((rc = gpr_nbr_point_rows(nb, nfree, d_fatoms, d_meta))) return cleanup(rc);
        d_frow0 = d_meta;
    }
    if (!rows_done && (rc = gpr_nbr_force_rows(nb, nfree, d_fatoms, d_frow0, frows, d_rowidx, d_xidx))) return cleanup(rc);

    // descriptor
    if ((rc = gpr_so3_power_spectrum(stream, nmax, lmax, rcut, alpha, nseg, ncheb, so3_table, natoms, P, S, d_vec, d_w,
                                     d_cp0, d_cr0, d_rp0, d_self, d_x, d_dxdr))) return cleanup(rc);

    // feature rows [energies ; forces] and prior variances
    if ((rc = gpr_dot2_features_ef(stream, d, nstruct, d_atom0, nfree, free_mask ? d_frow0 : d_cr0, d_xidx, d_rowidx, d_x, d_dxdr,
                                   d_num, ns, species_host, (int)extra.size(), extra.data(), sigma, sigma0, d_psi, F, d_diag)))
        return cleanup(rc);
    // mean = Psi w, var = max(diag - rowsum((Psi M) o Psi), 0); with C^-1 at hand the factored form in one launch
    if (want_std && Cinv_dev) {
        // the tail delivers the status words behind the results itself (no separate launch)
        if ((rc = gpr_ws_tail_internal(stream, (int)m, F, d_psi, F, Cinv_dev, F, w_dev, d_diag, out_dev, sigma * sigma,
                                       gpr_nbr_info_ptr(nb), out_dev + 2 * m, t_std_out))) return cleanup(rc);
        return cleanup(GPR_OK);
    } else {
        if (want_std) {
            if ((rc = dalloc((size_t)m * F * sizeof(double), &p))) return cleanup(rc);
            d_q = (double *)p;
            if ((rc = gpr_gemm(stream, 0, 0, (int)m, F, F, 1.0, d_psi, F, M_dev, F, 0.0, d_q, F))) return cleanup(rc);
        }
        if ((rc = gpr_ws_finish(stream, (int)m, F, d_psi, F, d_q, F, w_dev, d_diag, out_dev, 0.0))) return cleanup(rc);
    }
    if ((rc = gpr_nbr_info_to(nb, out_dev + (want_std ? 2 : 1) * m))) return cleanup(rc);
    return cleanup(GPR_OK);
}


// ---------------------------------------------------------------------------------------------------------------------
// Host-in / host-out form of the same prediction: the model stays in the session, the caller hands over the structures
// as host arrays and gets E, F, E_std, F_std back as host arrays -- what GP.predict_structure returns per structure
// (gaussianprocess.py:880-911), for a whole batch.  The session owns page-locked result buffers the device writes
// into, polls their status words instead of blocking on a stream, guesses the neighbour-table capacities from the
// densest batch it has seen and repeats a call whose guess was too small.
//
// Two ways to call it:
//   gpr_dot2_session_predict            one batch, synchronous (an NEB step: the optimiser needs the forces to go on);
//   gpr_dot2_session_submit / _wait     several batches in flight.  Every submission takes a *slot*: its own result
//                                       buffer and its own stream, so the device chain of batch k + 1 (eleven short
//                                       kernels, none of which fills the chip) runs beside the chain of batch k, and the
//                                       host gathers / unpacks one batch while the device works on the others.
//                                       Same kernels, same order inside a batch: results are bit-identical to _predict.
constexpr int SESSION_SLOTS = 8;
struct session_slot {
    hipStream_t st = nullptr;       // slot 0: the caller's stream (which may be the null stream)
    bool has_stream = false, own_stream = false, pooled = false;      // pooled: a stream of the device's slot pool (never destroyed)
    double *pin = nullptr;          // page-locked [2 m + 4]
    double *pin_dev = nullptr;      // its device address
    long long pin_cap = 0;
    std::vector<unsigned char> mask;
    bool busy = false;
    // the submission in flight; the input arrays stay the caller's until _wait returns
    int nstruct = 0, natoms = 0, nfree = 0;
    long long m = 0, nout = 0;
    bool masked = false;
    const int *atom0 = nullptr, *pbc = nullptr, *numbers = nullptr;
    const double *positions = nullptr, *cells = nullptr;
};
struct gpr_dot2_session {
    hipStream_t st;
    int nmax, lmax, nseg, ncheb, weight_on, ns, want_std;
    double rcut, alpha, sigma, sigma0;
    const double *table, *w, *M, *Cinv;
    std::vector<int> species;
    double pairs_per_atom = 0.0, rows_per_atom = 0.0;
    session_slot slot[SESSION_SLOTS];
    int depth = 2;                  // slots _submit rotates over
    int next = 0;
    hipEvent_t model_ready = nullptr;
};

extern "C" int gpr_dot2_session_create(gpr_dot2_session **out, void *stream, int nmax, int lmax, double rcut, double alpha,
                                       int nseg, int ncheb, const double *so3_table, int weight_on, int ns,
                                       const int *species_host, double sigma, double sigma0, const double *w_dev,
                                       const double *M_dev, const double *Cinv_dev, int want_std) {
    GPR_REQUIRE(out && so3_table && species_host && ns > 0 && w_dev && (!want_std || M_dev || Cinv_dev),
                "gpr_dot2_session_create: null pointer");
    gpr_dot2_session *s = new gpr_dot2_session();
    s->st = (hipStream_t)stream; s->nmax = nmax; s->lmax = lmax; s->rcut = rcut; s->alpha = alpha; s->nseg = nseg; s->ncheb = ncheb;
    s->table = so3_table; s->weight_on = weight_on; s->ns = ns; s->species.assign(species_host, species_host + ns);
    s->sigma = sigma; s->sigma0 = sigma0; s->w = w_dev; s->M = M_dev; s->Cinv = Cinv_dev; s->want_std = want_std;
    s->slot[0].st = s->st;          // slot 0 works on the caller's stream: the synchronous call behaves as it always did
    s->slot[0].has_stream = true;
    *out = s;
    return GPR_OK;
}

// pairs / seq rows per atom the next call sizes its neighbour tables from (normally learnt from the calls themselves)
extern "C" int gpr_dot2_session_set_hint(gpr_dot2_session *s, double pairs_per_atom, double rows_per_atom) {
    GPR_REQUIRE(s, "gpr_dot2_session_set_hint: null session");
    s->pairs_per_atom = pairs_per_atom; s->rows_per_atom = rows_per_atom;
    return GPR_OK;
}

// how many submissions may be in flight (1 .. 8; default 2)
extern "C" int gpr_dot2_session_set_depth(gpr_dot2_session *s, int depth) {
    GPR_REQUIRE(s && depth >= 1 && depth <= SESSION_SLOTS, "gpr_dot2_session_set_depth: depth must be 1 .. %d", SESSION_SLOTS);
    s->depth = depth;
    return GPR_OK;
}

extern "C" int gpr_dot2_session_destroy(gpr_dot2_session *s) {
    if (!s) return GPR_OK;
    for (session_slot &sl : s->slot) {
        if (sl.busy && sl.has_stream) (void)hipStreamSynchronize(sl.st);      // the device still writes into sl.pin
        if (sl.pin) (void)hipHostFree(sl.pin);
        if (sl.own_stream && sl.st) { (void)gpr_dev_cache_drop_stream(sl.st); (void)hipStreamSynchronize(sl.st); (void)hipStreamDestroy(sl.st); }
        else if (sl.pooled && sl.st) { (void)hipStreamSynchronize(sl.st); (void)gpr_dev_cache_drop_stream(sl.st); }
    }
    if (s->model_ready) (void)hipEventDestroy(s->model_ready);
    (void)hipGetLastError();
    delete s;
    return GPR_OK;
}

// the stream slot k works on (NULL: the slot has not been used yet, or k is out of range) -- for tests of the teardown path
extern "C" void *gpr_dot2_session_slot_stream(gpr_dot2_session *s, int k) {
    if (!s || k < 0 || k >= SESSION_SLOTS || !s->slot[k].has_stream) return nullptr;
    return (void *)s->slot[k].st;
}

// ---- the slots' streams come from a pool per device that is reserved EARLY (gpr_dot2_session_reserve_streams: the Python layer
// calls it at its first use of a device, before anything else of this library or of RCCL has created a stream).  HIP serves a
// process's streams from four hardware queues and deals them out by creation order, and how well the four chains of a depth-4
// session overlap turned out to depend on WHERE in that order the slots' streams were created, with period four: as the first three
// streams of the process (or behind four unused ones) 0.114 ms per step of the headline run whatever else the process does later
// -- a dense fit with its eight streams, an RCCL communicator --, behind one or three other streams 0.137-0.145, behind two 0.16;
// created lazily, as before, wherever the process happened to be: 0.115-0.126 alone, 0.120-0.126 after a dense fit, 0.137-0.145
// after a dense fit beside an RCCL communicator (tools/rccl_presence.py; GPR_SESSION_POOL_LEAD = unused streams created ahead of the
// pool, for that experiment; the synchronous step does not care).  Reserved first, the order no longer depends on what the process
// did before its first prediction.  Pool streams live as long as the process (a
// session hands back its cached temporaries when it ends, not the stream).  GPR_SESSION_POOL=0: a new stream per slot, as before.
namespace {
struct slot_pool { std::vector<hipStream_t> lead, slots; };
std::mutex g_pool_mu;
std::map<int, slot_pool> g_pools;

int pool_reserve_locked(slot_pool &pl, int want) {
    static const int lead_env = getenv("GPR_SESSION_POOL_LEAD") ? atoi(getenv("GPR_SESSION_POOL_LEAD")) : 0;
    while (pl.slots.empty() && (int)pl.lead.size() < lead_env) {          // (streams nobody uses: they only take the first places)
        hipStream_t c = nullptr;
        GPR_HIP_TRY(hipStreamCreateWithFlags(&c, hipStreamNonBlocking));
        pl.lead.push_back(c);
    }
    while ((int)pl.slots.size() < want) {
        hipStream_t c = nullptr;
        GPR_HIP_TRY(hipStreamCreateWithFlags(&c, hipStreamNonBlocking));
        pl.slots.push_back(c);
    }
    return GPR_OK;
}
bool pool_enabled() {
    static const bool on = !(getenv("GPR_SESSION_POOL") && atoi(getenv("GPR_SESSION_POOL")) == 0);
    return on;
}
}  // namespace

// reserve the slot streams of the current device now (idempotent; the streams of slots 1 .. 3, what the default depths use)
extern "C" int gpr_dot2_session_reserve_streams(void) {
    if (!pool_enabled()) return GPR_OK;
    int devid = 0;
    GPR_HIP_TRY(hipGetDevice(&devid));
    std::lock_guard<std::mutex> lk(g_pool_mu);
    return pool_reserve_locked(g_pools[devid], 3);
}

// a private stream for slot k >= 1: behind everything queued on the session's stream so far (the model arrays)
static int session_slot_stream(gpr_dot2_session *s, session_slot &sl) {
    if (sl.has_stream) return GPR_OK;
    if (!s->model_ready) GPR_HIP_TRY(hipEventCreateWithFlags(&s->model_ready, hipEventDisableTiming));
    if (pool_enabled()) {
        int devid = 0;
        GPR_HIP_TRY(hipGetDevice(&devid));
        const int k = (int)(&sl - s->slot);
        std::lock_guard<std::mutex> lk(g_pool_mu);
        slot_pool &pl = g_pools[devid];
        const int rc = pool_reserve_locked(pl, k);
        if (rc) return rc;
        sl.st = pl.slots[(size_t)k - 1];
        sl.own_stream = false; sl.pooled = true; sl.has_stream = true;
    } else {
        GPR_HIP_TRY(hipStreamCreateWithFlags(&sl.st, hipStreamNonBlocking));
        sl.own_stream = sl.has_stream = true;
    }
    GPR_HIP_TRY(hipEventRecord(s->model_ready, s->st));
    GPR_HIP_TRY(hipStreamWaitEvent(sl.st, s->model_ready, 0));
    return GPR_OK;
}

static int session_enqueue(gpr_dot2_session *s, session_slot &sl, int cap_pairs, int cap_rows) {
    volatile double *tail = sl.pin + sl.nout;
    for (int k = 0; k < 4; k++) tail[k] = __builtin_nan("");
    t_std_out = (s->want_std && s->Cinv) ? 1 : 0;             // the device delivers standard deviations, not variances
    const int rc = gpr_predict_dot2_batch(sl.st, s->nmax, s->lmax, s->rcut, s->alpha, s->nseg, s->ncheb, s->table, sl.nstruct, sl.atom0,
                                          sl.positions, sl.cells, sl.pbc, sl.numbers, s->weight_on, sl.masked ? sl.mask.data() : nullptr,
                                          s->ns, s->species.data(), s->sigma, s->sigma0, s->w, s->M, s->Cinv, s->want_std, cap_pairs,
                                          cap_rows, sl.pin_dev, sl.nout + 4);
    t_std_out = 0;
    return rc;
}

// fixed[natoms] (may be NULL): non-zero marks atoms held by FixAtoms -- F = F_std = 0 there, no force rows.
// Queues the whole prediction of the batch and returns; *ticket names the submission for _wait.  The input arrays must
// stay valid and unchanged until _wait has returned.  At most `depth` submissions may be waiting.
static int session_submit_on(gpr_dot2_session *s, int k, int nstruct, const int *atom0, const double *positions,
                             const double *cells, const int *pbc, const int *numbers, const unsigned char *fixed, int *ticket);

extern "C" int gpr_dot2_session_submit(gpr_dot2_session *s, int nstruct, const int *atom0, const double *positions,
                                       const double *cells, const int *pbc, const int *numbers, const unsigned char *fixed,
                                       int *ticket) {
    GPR_REQUIRE(s && nstruct >= 0 && atom0 && ticket, "gpr_dot2_session_submit: null pointer");
    int k = -1;
    for (int t = 0; t < s->depth && k < 0; t++) {
        const int c = (s->next + t) % s->depth;
        if (!s->slot[c].busy) k = c;
    }
    GPR_REQUIRE(k >= 0, "gpr_dot2_session_submit: %d submissions are waiting already (gpr_dot2_session_wait first, or raise the depth)", s->depth);
    s->next = (k + 1) % s->depth;
    return session_submit_on(s, k, nstruct, atom0, positions, cells, pbc, numbers, fixed, ticket);
}

static int session_submit_on(gpr_dot2_session *s, int k, int nstruct, const int *atom0, const double *positions,
                             const double *cells, const int *pbc, const int *numbers, const unsigned char *fixed, int *ticket) {
    session_slot &sl = s->slot[k];
    int rc = session_slot_stream(s, sl);
    if (rc) return rc;
    const int natoms = nstruct ? atom0[nstruct] : 0;
    sl.nstruct = nstruct; sl.natoms = natoms; sl.nfree = natoms; sl.masked = false;
    sl.atom0 = atom0; sl.positions = positions; sl.cells = cells; sl.pbc = pbc; sl.numbers = numbers;
    if (fixed) {
        bool any = false;
        for (int a = 0; a < natoms && !any; a++) any = fixed[a] != 0;
        if (any) {
            sl.mask.resize((size_t)natoms);
            sl.nfree = 0;
            for (int a = 0; a < natoms; a++) { sl.mask[a] = fixed[a] ? 0 : 1; sl.nfree += sl.mask[a]; }
            sl.masked = true;
        }
    }
    sl.m = (long long)nstruct + 3LL * sl.nfree;
    sl.nout = (s->want_std ? 2 : 1) * sl.m;
    *ticket = k;
    if (nstruct == 0) { sl.busy = true; return GPR_OK; }
    if (sl.pin_cap < sl.nout + 4) {
        if (sl.pin) (void)hipHostFree(sl.pin);
        sl.pin = nullptr; sl.pin_cap = 0;
        const long long want = std::max<long long>(2 * (sl.nout + 4), 4096);
        GPR_HIP_TRY(hipHostMalloc((void **)&sl.pin, (size_t)want * sizeof(double), hipHostMallocDefault));
        GPR_HIP_TRY(hipHostGetDevicePointer((void **)&sl.pin_dev, sl.pin, 0));
        sl.pin_cap = want;
    }
    int cap_pairs = 0, cap_rows = 0;
    if (s->pairs_per_atom > 0.0) {
        cap_pairs = (int)(s->pairs_per_atom * natoms * 1.125) + 64;
        cap_rows = (int)(s->rows_per_atom * natoms * 1.125) + 64;
    }
    rc = session_enqueue(s, sl, cap_pairs, cap_rows);
    if (rc) return rc;
    sl.busy = true;
    return GPR_OK;
}

// E[nstruct] total energies, F[natoms, 3], E_std[nstruct] (per atom, as upstream), F_std[natoms, 3]; the two std
// outputs may be NULL when the session was created without want_std.  info[4] (may be NULL): status words of the call.
extern "C" int gpr_dot2_session_wait(gpr_dot2_session *s, int ticket, double *E, double *F, double *E_std, double *F_std,
                                     double *info) {
    GPR_REQUIRE(s && ticket >= 0 && ticket < SESSION_SLOTS && s->slot[ticket].busy, "gpr_dot2_session_wait: no submission with ticket %d", ticket);
    session_slot &sl = s->slot[ticket];
    sl.busy = false;
    if (sl.nstruct == 0) return GPR_OK;
    GPR_REQUIRE(E && F && (!s->want_std || (E_std && F_std)), "gpr_dot2_session_wait: null pointer");
    const int nstruct = sl.nstruct, natoms = sl.natoms;
    const long long m = sl.m, nout = sl.nout;
    const int dev_std = (s->want_std && s->Cinv) ? 1 : 0;
    volatile double *tail = sl.pin + nout;
    for (int attempt = 0; attempt < 2; attempt++) {
        // the status words are the last thing the last kernel writes (system-scope fence before them): spin on them in
        // host memory; fall back to a blocking synchronisation if they do not show up within seconds
        bool seen = false;
        for (long long spin = 0; spin < (1LL << 33); spin++) {
            if (tail[0] == tail[0] && tail[1] == tail[1] && tail[2] == tail[2] && tail[3] == tail[3]) { seen = true; break; }
            if ((spin & 0xfffff) == 0xfffff && hipStreamQuery(sl.st) == hipSuccess) break;
        }
        if (!seen) GPR_HIP_TRY(hipStreamSynchronize(sl.st));
        {
            // The status words tell that the last kernel has ISSUED all its stores; that they have all LANDED in host memory
            // is something only the queue's own completion signal vouches for (writes of different XCDs reach the host on
            // different paths: a late result row behind the status words was seen once in a few hundred calls).  The words
            // end the long wait cheaply; the short rest is a busy query of the slot's stream -- no sleep, unlike a blocking
            // synchronisation.
            if (seen) {
                hipError_t q;
                while ((q = hipStreamQuery(sl.st)) == hipErrorNotReady) {}
                if (q != hipSuccess) { gpr_set_error("gpr_dot2_session_wait: %s", hipGetErrorString(q)); return GPR_ERR_HIP; }
            }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        if (tail[1] == 0.0) break;                       // no overflow
        GPR_REQUIRE(attempt == 0, "gpr_dot2_session_wait: neighbour table overflow with exact sizes");
        const int rc = session_enqueue(s, sl, 0, 0);     // the guess was too small: once more with exact sizes
        if (rc) return rc;
    }
    if (info) for (int k = 0; k < 4; k++) info[k] = tail[k];
    if (natoms > 0) {
        s->pairs_per_atom = std::max(s->pairs_per_atom, tail[2] / natoms);
        s->rows_per_atom = std::max(s->rows_per_atom, tail[3] / natoms);
    }
    GPR_REQUIRE(tail[0] == 0.0, "neighbour table is not symmetric: %d force rows missing (%d structures, %d atoms, %s, pairs %d, rows %d, slot %d)",
                (int)tail[0], nstruct, natoms, sl.masked ? "fixed atoms" : "all atoms moving", (int)tail[2], (int)tail[3], ticket);
    // unpack [mean | var]: energies per atom -> total energies, force rows -> [natoms, 3] with zeros at fixed atoms
    const double *mean = sl.pin, *var = sl.pin + m;
    const int *atom0 = sl.atom0;
    for (int k = 0; k < nstruct; k++) {
        E[k] = mean[k] * (atom0[k + 1] - atom0[k]);
        if (s->want_std) E_std[k] = dev_std ? var[k] : sqrt(var[k]);
    }
    const double *mf = mean + nstruct, *vf = var + nstruct;
    if (!sl.masked) {
        memcpy(F, mf, (size_t)3 * natoms * sizeof(double));
        if (s->want_std) {
            if (dev_std) memcpy(F_std, vf, (size_t)3 * natoms * sizeof(double));
            else for (long long i = 0; i < 3LL * natoms; i++) F_std[i] = sqrt(vf[i]);
        }
    } else {
        const unsigned char *free_mask = sl.mask.data();
        long long r = 0;
        for (int a = 0; a < natoms; a++) {
            if (free_mask[a]) {
                for (int c = 0; c < 3; c++) { F[3 * a + c] = mf[r + c]; if (s->want_std) F_std[3 * a + c] = dev_std ? vf[r + c] : sqrt(vf[r + c]); }
                r += 3;
            } else {
                for (int c = 0; c < 3; c++) { F[3 * a + c] = 0.0; if (s->want_std) F_std[3 * a + c] = 0.0; }
            }
        }
    }
    return GPR_OK;
}

// one batch, synchronous: submit on a free slot (slot 0, the caller's stream, unless a ticket holds it) and wait
extern "C" int gpr_dot2_session_predict(gpr_dot2_session *s, int nstruct, const int *atom0, const double *positions,
                                        const double *cells, const int *pbc, const int *numbers, const unsigned char *fixed,
                                        double *E, double *F, double *E_std, double *F_std, double *info) {
    GPR_REQUIRE(s && nstruct >= 0 && atom0 && E && F && (!s->want_std || (E_std && F_std)), "gpr_dot2_session_predict: null pointer");
    if (nstruct == 0) return GPR_OK;
    // slot 0 -- the caller's stream -- when it is free; with submissions in flight any free slot serves (its own stream,
    // behind the model arrays): a synchronous call never has to wait for tickets somebody else holds
    int k = -1;
    for (int c = 0; c < SESSION_SLOTS && k < 0; c++)
        if (!s->slot[c].busy) k = c;
    GPR_REQUIRE(k >= 0, "gpr_dot2_session_predict: all %d slots hold submissions that have not been waited for", SESSION_SLOTS);
    int ticket = -1;
    int rc = session_submit_on(s, k, nstruct, atom0, positions, cells, pbc, numbers, fixed, &ticket);
    if (rc) return rc;
    return gpr_dot2_session_wait(s, ticket, E, F, E_std, F_std, info);
}
