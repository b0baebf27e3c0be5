// pack.hip -- gpr_set: descriptor rows normalised, projected and laid out in
// v_mfma_f64_16x16x4_f64 operand order.
//
// What is stored per row (reference quantities: rbf_kernel.cpp:355-420, Appendix B of SURVEY.md):
//   a~  = x / n                 n = |x| (+1e-8 under GPR_SET_NUMPY_EPS)
//   A'  = (I - a~ a~^T) dxdr/n  the derivative columns with the a~ component projected out
// With these, for a row pair (a, b):  c = a~.b~ is the cosine d,  A'^T B' = dx1dr^T H dx2dr with H the
// second derivative of d (d2d_dx1dx2 in the reference),  A'^T b~ = dx1dr^T dd/dx1 and
// a~^T B' = dd/dx2^T dx2dr -- so one 4x4 product [A'^T; a~^T][B', b~] carries everything the
// reference's O(d^2) loop nest computes.
#include <algorithm>
#include <cstring>
#include <cstdarg>
#include <numeric>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>
#include "gpr_internal.h"

// ---------------------------------------------------------------- errors / misc
static thread_local std::string g_last_error;

void gpr_set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

extern "C" const char *gpr_last_error(void) { return g_last_error.c_str(); }

extern "C" const char *gpr_build_info(void) {
    return "libgprhip gfx950 (v_mfma_f64_16x16x4_f64) built " __DATE__ " " __TIME__;
}

extern "C" int gpr_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

// Stream-ordered allocations come from the device's default memory pool.  Its release threshold defaults to zero,
// i.e. every synchronisation hands the freed blocks back to the driver and the next call pays for mapping them again
// (hundreds of microseconds for the tens of MB a prediction step needs): keep them cached in the pool instead.
static void gpr_pool_keep_cached() {
    static thread_local int done_for = -1;
    int devid = 0;
    if (hipGetDevice(&devid) != hipSuccess || devid == done_for) return;
    hipMemPool_t pool;
    if (hipDeviceGetDefaultMemPool(&pool, devid) == hipSuccess) {
        unsigned long long keep = ~0ULL;
        (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
    }
    (void)hipGetLastError();
    done_for = devid;
}

// Stream-ordered temporaries.  A block freed on stream `st` may be handed to the next request on the same stream without
// a trip to the driver: everything that used it was enqueued on `st` before the free, everything that will use it is
// enqueued after the request -- the guarantee hipFreeAsync / hipMallocAsync give, minus two driver calls (2-3 us each; a
// prediction step makes three requests of the same sizes every time, between its first, short launches).  One small
// cache for the process (16 blocks, 512 MB, a mutex around it): stream order is what makes the reuse safe, whichever host
// thread asks.  Blocks that fall out go back to the driver on the stream they were freed on (plain hipFree if that
// stream is gone).
namespace {
struct cached_block { void *p; size_t bytes; hipStream_t st; int dev; };   // (device, stream): the null stream exists on every device
std::vector<cached_block> g_cache;
std::unordered_map<void *, size_t> g_sizes;               // every live block of gpr_dev_alloc -> its size
size_t g_cached_bytes = 0;
std::mutex g_cache_mu;
constexpr size_t CACHE_BLOCKS = 16, CACHE_BYTES = (size_t)512 << 20, CACHE_BLOCK_MAX = (size_t)256 << 20;

void release_to_driver(void *p, hipStream_t st) {
    if (hipFreeAsync(p, st) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(p); }
}
int current_device() { int d = 0; if (hipGetDevice(&d) != hipSuccess) { (void)hipGetLastError(); d = -1; } return d; }
}

int gpr_func_lds_once(const void *fn, int bytes) {
    static std::mutex mu;
    static std::map<std::pair<const void *, int>, int> done;       // (kernel, device) -> limit set so far
    const int devid = current_device();
    std::lock_guard<std::mutex> lk(mu);
    int &have = done[{fn, devid}];
    if (have >= bytes) return GPR_OK;
    GPR_HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    have = bytes;
    return GPR_OK;
}

int gpr_cu_count() {
    static std::mutex mu;
    static std::map<int, int> cus;
    const int devid = current_device();
    std::lock_guard<std::mutex> lk(mu);
    auto it = cus.find(devid);
    if (it != cus.end()) return it->second;
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, devid < 0 ? 0 : devid) != hipSuccess || n <= 0) { (void)hipGetLastError(); n = 256; }
    cus[devid] = n;
    return n;
}

int gpr_dev_alloc(void **p, size_t bytes, hipStream_t st) {
    *p = nullptr;
    if (bytes == 0) bytes = 16;
    const int devid = current_device();
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        for (size_t i = 0; i < g_cache.size(); i++) {
            const cached_block c = g_cache[i];
            if (c.st == st && c.dev == devid && c.bytes >= bytes && c.bytes <= 2 * bytes + 4096) {
                *p = c.p;
                g_cached_bytes -= c.bytes;
                g_cache.erase(g_cache.begin() + (long)i);
                return GPR_OK;
            }
        }
    }
    gpr_pool_keep_cached();
    GPR_HIP_TRY(hipMallocAsync(p, bytes, st));
    std::lock_guard<std::mutex> lk(g_cache_mu);
    g_sizes[*p] = bytes;
    return GPR_OK;
}

int gpr_dev_free(void *p, hipStream_t st) {
    if (!p) return GPR_OK;
    const int devid = current_device();
    std::vector<cached_block> out;                        // what leaves the cache: handed to the driver outside the lock
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        const auto it = g_sizes.find(p);
        const size_t bytes = it != g_sizes.end() ? it->second : 0;
        if (bytes == 0 || bytes > CACHE_BLOCK_MAX) {
            if (it != g_sizes.end()) g_sizes.erase(it);
            out.push_back(cached_block{p, bytes, st, devid});
        } else {
            while (!g_cache.empty() && (g_cache.size() >= CACHE_BLOCKS || g_cached_bytes + bytes > CACHE_BYTES)) {
                out.push_back(g_cache.front());               // oldest out
                g_cached_bytes -= g_cache.front().bytes;
                g_sizes.erase(g_cache.front().p);
                g_cache.erase(g_cache.begin());
            }
            g_cache.push_back(cached_block{p, bytes, st, devid});
            g_cached_bytes += bytes;
        }
    }
    for (const cached_block &c : out) release_to_driver(c.p, c.st);
    return GPR_OK;
}

// A stream is about to be destroyed: hand its cached blocks back to the driver first (a cached block remembers the
// stream it was freed on and would otherwise be released on a dangling handle when it falls out of the cache).
int gpr_dev_cache_drop_stream(hipStream_t st) {
    std::vector<cached_block> out;
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        for (size_t i = 0; i < g_cache.size();) {
            if (g_cache[i].st == st) {
                out.push_back(g_cache[i]);
                g_cached_bytes -= g_cache[i].bytes;
                g_sizes.erase(g_cache[i].p);
                g_cache.erase(g_cache.begin() + (long)i);
            } else i++;
        }
    }
    for (const cached_block &c : out) release_to_driver(c.p, c.st);
    return GPR_OK;
}

// how many cached temporaries remember `stream` as the stream they were freed on (all streams with any != 0): after a
// stream's owner has called gpr_dev_cache_drop_stream this must be zero, or a later eviction would hand a block to the
// driver on a destroyed handle (tests/test_gpu_session_teardown.py)
extern "C" int gpr_dev_cache_count(void *stream, int any) {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    int n = 0;
    for (const cached_block &c : g_cache) n += (any || c.st == (hipStream_t)stream) ? 1 : 0;
    return n;
}

extern "C" int gpr_memcpy_dtod(void *stream, void *dst, const void *src, long long bytes) {
    if (bytes <= 0) return GPR_OK;
    GPR_REQUIRE(dst && src, "gpr_memcpy_dtod: null pointer");
    GPR_HIP_TRY(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return GPR_OK;
}

// ---------------------------------------------------------------- pack kernel
struct pack_args {
    const double *x, *dxdr;
    const int *ele, *src_row;
    const int *x_index, *d_index;   // optional gather: caller row r reads x[x_index[r]], dxdr[d_index[r]], ele[x_index[r]]
    double *ops, *norms;
    int *ele_out;
    int d, width, col0, nc, role, rpad, tile_rows;
    int numpy_eps, pad_code;
    int plane, kpad;                // doubles per operand plane (64 k-steps), padded descriptor length (4 k-steps)
};

// Sixteen lanes per padded row, sixteen rows per block.  (The first form had a thread per row walking its 30 + 90 values one load
// after the other, every lane of a wavefront in a different cache line: 30-40 us for the 400 rows of one structure's force
// points.)  The lanes of a row read x and the three derivative columns together into LDS; lane 0 of the row then forms the
// norm and the three projections in the same order of summation as before (the values are unchanged); all sixteen write the
// operand planes.
__global__ void __launch_bounds__(256) pack_rows_kernel(pack_args a) {
    extern __shared__ double pk_lds[];                  // [16 rows][4 d]: x[d] | D[d][3]
    __shared__ double pk_sc[16][6];                     // per row: n, inv, proj[3], live
    const int lr = threadIdx.x >> 4, g = threadIdx.x & 15;
    const int r = blockIdx.x * 16 + lr;
    const bool inr = r < a.rpad;
    const int src = inr ? a.src_row[r] : -1;
    const int nplanes = (a.role == GPR_ROLE_RESIDENT) ? a.nc : 1;
    size_t base;
    int slot_row;
    if (a.role == GPR_ROLE_STREAM && a.nc == 4) {
        base = (size_t)(r >> 2) * a.plane;              // tile of 4 rows, comps share the plane
        slot_row = r & 3;
    } else {
        base = (size_t)(r >> 4) * nplanes * a.plane;    // 16 rows per plane, nc planes (resident)
        slot_row = r & 15;
    }
    auto slot = [&](int comp, int k) -> size_t {
        int s = k >> 2, kk = k & 3;
        if (a.role == GPR_ROLE_STREAM && a.nc == 4)
            return base + (size_t)s * GPR_WAVE + slot_row + 4 * comp + 16 * kk;
        return base + (size_t)comp * a.plane + (size_t)s * GPR_WAVE + slot_row + 16 * kk;
    };
    bool live = src >= 0;
    int xsrc = src, dsrc = src;
    if (live && a.x_index) xsrc = a.x_index[src];
    if (live && a.d_index) dsrc = a.d_index[src];
    if (xsrc < 0 || dsrc < 0) live = false;
    double *sx = pk_lds + (size_t)lr * 4 * a.d, *sD = sx + a.d;
    if (live) {
        const double *xr = a.x + (size_t)xsrc * a.d;
        for (int k = g; k < a.d; k += 16) sx[k] = xr[k];
        if (a.nc == 4) {
            const double *D = a.dxdr + (size_t)dsrc * a.d * a.width + a.col0;
            for (int e = g; e < 3 * a.d; e += 16) { const int k = e / 3, c = e - 3 * k; sD[e] = D[(size_t)k * a.width + c]; }
        }
    }
    __syncthreads();
    if (g == 0) {
        double n = 0.0, inv = 0.0, proj[3] = {0.0, 0.0, 0.0};
        bool lv = live;
        if (lv) {
            double acc = 0.0;
            for (int k = 0; k < a.d; k++) acc += sx[k] * sx[k];
            n = sqrt(acc);
            if (a.numpy_eps) n += 1e-8;                 // base.py:118-119, Dot_mb.py:188-189, :206-211
            else if (!(n > 1e-8)) lv = false;           // rbf_kernel.cpp:26,37 skip
        }
        if (lv) {
            inv = 1.0 / n;
            if (a.nc == 4)
                for (int k = 0; k < a.d; k++) {
                    const double ak = sx[k] * inv;
                    for (int c = 0; c < 3; c++) proj[c] += ak * (sD[3 * k + c] * inv);
                }
        }
        pk_sc[lr][0] = n; pk_sc[lr][1] = inv; pk_sc[lr][2] = proj[0]; pk_sc[lr][3] = proj[1]; pk_sc[lr][4] = proj[2];
        pk_sc[lr][5] = lv ? 1.0 : 0.0;
    }
    __syncthreads();
    if (!inr) return;
    live = pk_sc[lr][5] != 0.0;
    if (!live) {
        for (int e = g; e < a.nc * a.kpad; e += 16) { const int c = e / a.kpad, k = e - c * a.kpad; a.ops[slot(c, k)] = 0.0; }
        if (g == 0) { a.ele_out[r] = a.pad_code; a.norms[r] = 0.0; }
        return;
    }
    const double inv = pk_sc[lr][1];
    const int row_comp = a.nc - 1;                      // plane / comp index of the a~ row
    if (a.nc == 4) {
        const double proj[3] = {pk_sc[lr][2], pk_sc[lr][3], pk_sc[lr][4]};
        for (int e = g; e < 3 * a.kpad; e += 16) {
            const int c = e / a.kpad, k = e - c * a.kpad;
            double v = 0.0;
            if (k < a.d) { const double ak = sx[k] * inv; v = sD[3 * k + c] * inv - ak * (c == 0 ? proj[0] : (c == 1 ? proj[1] : proj[2])); }
            a.ops[slot(c, k)] = v;
        }
    }
    for (int k = g; k < a.kpad; k += 16) a.ops[slot(row_comp, k)] = k < a.d ? sx[k] * inv : 0.0;
    if (g == 0) { a.ele_out[r] = a.ele[xsrc]; a.norms[r] = pk_sc[lr][0]; }
}

// ---------------------------------------------------------------- host side
template <class T>
static int upload(T **dev, const std::vector<T> &h, hipStream_t st, long long *bytes) {
    void *p;
    int rc = gpr_dev_alloc(&p, h.size() * sizeof(T), st);
    if (rc) return rc;
    *dev = (T *)p;
    *bytes += (long long)(h.size() * sizeof(T));
    if (!h.empty()) GPR_HIP_TRY(hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, st));
    return GPR_OK;
}

extern "C" int gpr_set_create(gpr_set **out, void *stream, int role, int d, int width, int col0,
                              int nrows, int npts, const int *counts_host, const double *x_dev,
                              const double *dxdr_dev, const int *ele_dev, int flags) {
    return gpr_set_create_gather(out, stream, role, d, width, col0, nrows, npts, counts_host, x_dev, dxdr_dev, ele_dev,
                                 nullptr, nullptr, flags);
}

extern "C" int gpr_set_create_gather(gpr_set **out, void *stream, int role, int d, int width, int col0,
                                     int nrows, int npts, const int *counts_host, const double *x_dev,
                                     const double *dxdr_dev, const int *ele_dev, const int *x_index_dev,
                                     const int *dxdr_index_dev, int flags) {
    hipStream_t st = (hipStream_t)stream;
    gpr_reap_sets();
    GPR_REQUIRE(out != nullptr, "gpr_set_create: out is null");
    *out = nullptr;
    GPR_REQUIRE(role == GPR_ROLE_STREAM || role == GPR_ROLE_RESIDENT, "gpr_set_create: bad role %d", role);
    GPR_REQUIRE(width == 0 || width == 3 || width == 9, "gpr_set_create: width must be 0, 3 or 9 (got %d)", width);
    GPR_REQUIRE(width == 0 || (col0 >= 0 && col0 + 3 <= width), "gpr_set_create: col0=%d outside width=%d", col0, width);
    GPR_REQUIRE(npts >= 0 && nrows >= 0, "gpr_set_create: negative sizes");
    if (d < 1 || d > GPR_KPAD_MAX) {
        gpr_set_error("gpr_set_create: descriptor length d=%d unsupported by this build (1..%d)", d, GPR_KPAD_MAX);
        return GPR_ERR_UNSUPPORTED;
    }
    GPR_REQUIRE(nrows == 0 || (x_dev && ele_dev && (width == 0 || dxdr_dev)), "gpr_set_create: null device pointer");
    long long tot = 0;
    for (int p = 0; p < npts; p++) {
        GPR_REQUIRE(counts_host[p] >= 0, "gpr_set_create: negative count at point %d", p);
        tot += counts_host[p];
    }
    GPR_REQUIRE(tot == nrows, "gpr_set_create: counts sum to %lld but nrows=%d", tot, nrows);

    gpr_set *s = new gpr_set();
    s->role = role; s->d = d; s->width = width; s->col0 = col0; s->flags = flags;
    s->nc = (width == 0) ? 1 : 4;
    s->ks = (d <= GPR_KPAD) ? GPR_KSTEPS : (d <= 4 * GPR_KSTEPS_MAX ? GPR_KSTEPS_MAX : GPR_KSTEPS_WIDE);
    const int chunk_tiles = GPR_CHUNK_TILES * GPR_KSTEPS / s->ks;          // same bytes per staged chunk
    const size_t plane = (size_t)s->ks * GPR_WAVE;
    s->npts = npts; s->nrows = nrows;
    s->tile_rows = (role == GPR_ROLE_STREAM && s->nc == 4) ? 4 : 16;

    // packed order: points sorted by resident group count (descending, stable) so that the
    // wavefronts of one workgroup own points of similar size; both roles use the same order
    std::vector<int> order(npts);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
        return (counts_host[a] + 15) / 16 > (counts_host[b] + 15) / 16;
    });
    std::vector<long long> row0(npts + 1, 0);
    for (int p = 0; p < npts; p++) row0[p + 1] = row0[p] + counts_host[p];

    s->h_perm = order;
    s->h_cnt.resize(npts);
    std::vector<int> &src_row = s->hk_src_row;
    src_row.reserve((size_t)nrows + 16 * (size_t)npts);
    std::vector<double> &inv_cnt = s->hk_inv_cnt;
    inv_cnt.assign(npts, 0.0);
    std::vector<gpr_chunk> &chunks = s->hk_chunks;
    std::vector<gpr_rpoint> &rpts = s->hk_rpts;     // resident: one entry per wave unit
    s->h_pt_chunk0.assign(npts + 1, 0);
    s->h_pt_tiles.assign(npts, 0);
    int unit = 0;
    s->maxgroups = 0;
    std::vector<int> &rowpt = s->hk_rowpt, &wu_pt0 = s->hk_wu_pt0;
    if (role == GPR_ROLE_STREAM) {
        for (int p = 0; p < npts; p++) {
            int o = order[p], cnt = counts_host[o];
            s->h_cnt[p] = cnt;
            inv_cnt[p] = cnt > 0 ? 1.0 / cnt : 0.0;
            int units = (cnt + s->tile_rows - 1) / s->tile_rows;
            for (int r = 0; r < units * s->tile_rows; r++)
                src_row.push_back(r < cnt ? (int)(row0[o] + r) : -1);
            s->h_pt_tiles[p] = units;
            s->h_pt_chunk0[p] = (int)chunks.size();
            for (int t = 0; t < units; t += chunk_tiles) {
                int nt = std::min(chunk_tiles, units - t);
                chunks.push_back(gpr_chunk{unit + t, nt, p, (t + nt >= units) ? 1 : 0});
            }
            unit += units;
        }
    } else {
        // Resident side: the rows of consecutive packed points are laid out back to back and cut into 16-row MFMA
        // groups per *wave unit* (the points one wavefront owns: as many consecutive points as fit in 8 groups), so
        // the padding to a multiple of 16 rows is paid once per unit, not once per point (28-row force points: 4
        // points = 7 groups exactly instead of 8).  rowpt gives the packed point of every padded row (-1: padding).
        wu_pt0.assign(1, 0);
        int p = 0;
        while (p < npts) {
            int rows = 0, q = p;
            while (q < npts && (q == p || rows + counts_host[order[q]] <= GPR_WU_MAX_GROUPS * GPR_GROUP_ROWS)) {
                const int o = order[q], cnt = counts_host[o];
                s->h_cnt[q] = cnt;
                inv_cnt[q] = cnt > 0 ? 1.0 / cnt : 0.0;
                for (int r = 0; r < cnt; r++) { src_row.push_back((int)(row0[o] + r)); rowpt.push_back(q); }
                s->h_pt_tiles[q] = 0;
                rows += cnt;
                q++;
            }
            const int groups = (rows + GPR_GROUP_ROWS - 1) / GPR_GROUP_ROWS;
            for (int r = rows; r < groups * GPR_GROUP_ROWS; r++) { src_row.push_back(-1); rowpt.push_back(-1); }
            rpts.push_back(gpr_rpoint{unit, groups});
            wu_pt0.push_back(q);
            s->maxgroups = std::max(s->maxgroups, groups);
            s->max_unit_pts = std::max(s->max_unit_pts, q - p);
            unit += groups;
            p = q;
        }
        s->nwu = (int)rpts.size();
    }
    s->h_pt_chunk0[npts] = (int)chunks.size();
    s->nunits = unit;
    s->rpad = unit * s->tile_rows;
    s->nchunks = (int)chunks.size();

    int rc = GPR_OK;
    auto fail = [&](int code) { gpr_set_destroy(s, stream); return code; };
    if (role == GPR_ROLE_STREAM) {
        // launch ranges: consecutive packed points, about 128 tiles each (one task = range x resident block; a launch may take
        // them two at a time, kblock.hip); small sets are cut finer (down to 8 tiles, 2 for energy sets) so that a launch against them still has
        // a few dozen ranges to spread over the CUs -- the resident side contributes only npts / (points per wave unit) / NW blocks
        std::vector<int> &irc = s->hk_irc, &irp = s->hk_irp;
        irc.assign(1, 0); irp.assign(1, 0);
        static const int range_cap = getenv("GPR_RANGE_TILES") ? atoi(getenv("GPR_RANGE_TILES")) : 128;    // (A/B: tools/kff_traffic.sh)
        // (energy sets: 16-row tiles and a cheap epilogue -- a few tiles per task keep a small launch from being one long walk
        //  per block: 80 -> 44 us for the energy-force blocks of one structure against 30 training structures; force sets keep
        //  eight, where finer tasks only add resident reloads: 269 -> 290 us)
        static const int floor_env = getenv("GPR_RANGE_FLOOR") ? atoi(getenv("GPR_RANGE_FLOOR")) : 0;            // A/B
        const int range_floor = floor_env > 0 ? floor_env : (s->nc == 1 && unit <= 64) ? 2 : 8;         // (only the tiny ones: at 128 tiles the finer tasks cost more than they spread)
        const int range_tiles = std::max(range_floor, std::min(range_cap, unit / 64));
        int tiles = 0;
        for (int p = 0; p < npts; p++) {
            tiles += s->h_pt_tiles[p];
            if (tiles >= range_tiles && p + 1 < npts) { irc.push_back(s->h_pt_chunk0[p + 1]); irp.push_back(p + 1); tiles = 0; }
        }
        irc.push_back(s->nchunks); irp.push_back(npts);
        s->nranges = (int)irc.size() - 1;
    }
    // one allocation: [index arrays, uploaded in one copy | ele | norms | operand planes]
    std::vector<char> &hb = s->hk_blob;
    hb.clear();
    auto put = [&](const void *src, size_t nbytes) {              // 16-byte aligned pieces; returns the offset
        const size_t off = (hb.size() + 15) & ~(size_t)15;
        hb.resize(off + nbytes);
        if (nbytes) memcpy(hb.data() + off, src, nbytes);
        return off;
    };
    const size_t o_src = put(src_row.data(), src_row.size() * sizeof(int));
    const size_t o_perm = put(s->h_perm.data(), s->h_perm.size() * sizeof(int));
    const size_t o_inv = put(inv_cnt.data(), inv_cnt.size() * sizeof(double));
    size_t o_a = 0, o_b = 0, o_c = 0, o_d = 0;
    if (role == GPR_ROLE_STREAM) {
        o_a = put(chunks.data(), chunks.size() * sizeof(gpr_chunk));
        o_b = put(s->h_pt_chunk0.data(), s->h_pt_chunk0.size() * sizeof(int));
        o_c = put(s->hk_irc.data(), s->hk_irc.size() * sizeof(int));
        o_d = put(s->hk_irp.data(), s->hk_irp.size() * sizeof(int));
    } else {
        o_a = put(rpts.data(), rpts.size() * sizeof(gpr_rpoint));
        o_b = put(wu_pt0.data(), wu_pt0.size() * sizeof(int));
        o_c = put(rowpt.data(), rowpt.size() * sizeof(int));
    }
    const size_t up_bytes = hb.size();
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t planes = (size_t)s->nunits * (role == GPR_ROLE_RESIDENT ? s->nc : 1);
    const size_t ops_bytes = planes * plane * sizeof(double);
    const size_t o_ele = al(up_bytes), o_norms = al(o_ele + (size_t)s->rpad * sizeof(int));
    const size_t o_ops = al(o_norms + (size_t)s->rpad * sizeof(double)), total = o_ops + ops_bytes;
    void *p;
    if ((rc = gpr_dev_alloc(&p, total, st))) return fail(rc);
    s->blob = p;
    char *base = (char *)p;
    if (up_bytes) {
        const hipError_t ce = hipMemcpyAsync(base, hb.data(), up_bytes, hipMemcpyHostToDevice, st);
        if (ce != hipSuccess) { gpr_set_error("gpr_set_create: upload failed: %s", hipGetErrorString(ce)); return fail(GPR_ERR_HIP); }
    }
    int *d_src = (int *)(base + o_src);
    s->perm = (int *)(base + o_perm); s->inv_cnt = (double *)(base + o_inv);
    if (role == GPR_ROLE_STREAM) {
        s->chunks = (gpr_chunk *)(base + o_a); s->pt_chunk0 = (int *)(base + o_b);
        s->ir_chunk0 = (int *)(base + o_c); s->ir_pt0 = (int *)(base + o_d);
    } else {
        s->rpts = (gpr_rpoint *)(base + o_a); s->wu_pt0 = (int *)(base + o_b); s->rowpt = (int *)(base + o_c);
    }
    s->ele = (int *)(base + o_ele); s->norms = (double *)(base + o_norms); s->ops = (double *)(base + o_ops);
    s->bytes = (long long)(total - (o_perm - o_src));           // (the source-row map is only read by the pack kernel)
    if (s->rpad > 0) {
        pack_args a;
        a.x = x_dev; a.dxdr = dxdr_dev; a.ele = ele_dev; a.src_row = d_src;
        a.x_index = x_index_dev; a.d_index = dxdr_index_dev;
        a.ops = s->ops; a.norms = s->norms; a.ele_out = s->ele;
        a.d = d; a.width = width; a.col0 = col0; a.nc = s->nc; a.role = role;
        a.rpad = s->rpad; a.tile_rows = s->tile_rows;
        a.numpy_eps = (flags & GPR_SET_NUMPY_EPS) ? 1 : 0;
        a.plane = (int)plane; a.kpad = 4 * s->ks;
        a.pad_code = (role == GPR_ROLE_STREAM) ? GPR_ELE_PAD_STREAM : GPR_ELE_PAD_RESIDENT;
        const int pid = gpr_prof_begin(st, GPR_PROF_PACK);
        const size_t pk_bytes = (size_t)16 * 4 * d * sizeof(double);            // 64 KB at d = 128, next to a few hundred static bytes
        if (pk_bytes > 60 * 1024 && (rc = gpr_func_lds_once((const void *)pack_rows_kernel, 96 * 1024))) return fail(rc);
        hipLaunchKernelGGL(pack_rows_kernel, dim3((s->rpad + 15) / 16), dim3(256), pk_bytes, st, a);
        gpr_prof_end(st, pid);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) {
            gpr_set_error("pack_rows_kernel launch failed: %s", hipGetErrorString(e));
            return fail(GPR_ERR_HIP);
        }
    }
    *out = s;
    return GPR_OK;
}

// Destroyed sets whose host vectors may still feed an asynchronous upload: the device memory is released
// in stream order at once, the host object when an event recorded at destroy time has completed (polled
// from later create / destroy calls), so that destroying a set never blocks the host on the stream.
namespace {
struct dead_set { hipEvent_t ev; gpr_set *s; };
std::mutex g_dead_mu;
std::vector<dead_set> g_dead;
}

void gpr_reap_sets() {
    std::lock_guard<std::mutex> lk(g_dead_mu);
    size_t keep = 0;
    for (size_t k = 0; k < g_dead.size(); k++) {
        if (hipEventQuery(g_dead[k].ev) == hipErrorNotReady) { g_dead[keep++] = g_dead[k]; continue; }
        (void)hipEventDestroy(g_dead[k].ev);
        delete g_dead[k].s;
    }
    g_dead.resize(keep);
}

extern "C" int gpr_set_destroy(gpr_set *s, void *stream) {
    if (!s) return GPR_OK;
    hipStream_t st = (hipStream_t)stream;
    // stream-ordered frees: kernels already queued on `st` still see the arrays
    gpr_dev_free(s->blob, st);                                  // every device array of the set is a piece of it
    s->blob = nullptr; s->ops = nullptr;
    hipEvent_t ev;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess && hipEventRecord(ev, st) == hipSuccess) {
        std::lock_guard<std::mutex> lk(g_dead_mu);
        g_dead.push_back(dead_set{ev, s});
    } else {
        (void)hipStreamSynchronize(st);
        delete s;
    }
    gpr_reap_sets();
    return GPR_OK;
}

extern "C" int gpr_set_info(const gpr_set *s, int *npts, int *nrows, int *rows_padded, long long *bytes) {
    GPR_REQUIRE(s != nullptr, "gpr_set_info: null set");
    if (npts) *npts = s->npts;
    if (nrows) *nrows = s->nrows;
    if (rows_padded) *rows_padded = s->rpad;
    if (bytes) *bytes = s->bytes;
    return GPR_OK;
}
