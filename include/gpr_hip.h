/*
 * gpr_hip.h -- C-ABI of libgprhip.so: the MI355X (gfx950) implementation of the GPR hot path
 * of MaterSim/GPR_calculator.  Plain C types only; no torch / Python objects cross this boundary.
 *
 * Two levels:
 *
 *  (L0) Legacy natives.  The 13 `extern "C" void` entry points the reference binds through cffi
 *       (gpr_calc/kernels/rbf_kernel.h:4-38, gpr_calc/kernels/dot_kernel.h:4-27; cdefs in
 *       gpr_calc/kernels/librbf_builder.py:4-13 and libdot_builder.py:4-10), with identical
 *       names, argument order and semantics: HOST pointers in and out, outputs ACCUMULATED
 *       (+= / -=) into caller-zeroed buffers, rows with |x| <= 1e-8 and species mismatches
 *       skipped, void return.  They upload, run the HIP kernels on the current device and
 *       download.  There is NO CPU fallback and nothing terminates the process: on a failure
 *       (no GPU, a HIP error, an unsupported shape) the output buffers are left untouched, the
 *       reason is printed once on stderr and kept for gpr_last_error(), and the status of the
 *       most recent L0 call is what gpr_l0_last_status() returns (the void signature has no
 *       other channel).
 *
 *  (L1) Device-resident API used by the Python host layer (gpr_calculator_amd/): descriptor
 *       rows are packed once into MFMA operand order (gpr_set), covariance blocks are written
 *       straight into caller-provided device matrices, and the dense factor / solve /
 *       variance kernels operate on device pointers.  Every L1 call takes the HIP stream to
 *       launch on (hipStream_t passed as void*; NULL = default stream) and returns 0 on
 *       success or a negative gpr_status; the message is available from gpr_last_error().
 *
 * All floating point is FP64; species / counts / indices are int32.
 */
#ifndef GPR_HIP_H
#define GPR_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ status / errors */
enum gpr_status {
    GPR_OK = 0,
    GPR_ERR_HIP = -1,        /* a HIP runtime call failed (message has the HIP error string) */
    GPR_ERR_ARG = -2,        /* inconsistent shapes / null pointers / unsupported parameter   */
    GPR_ERR_UNSUPPORTED = -3,/* a configuration this build has no kernel for                  */
    GPR_ERR_NOT_SPD = -4     /* Cholesky met a non-positive pivot (info in gpr_last_error)    */
};
const char *gpr_last_error(void);
/* "gfx950 ... ; built <date>" -- lets tests prove which native library was loaded */
const char *gpr_build_info(void);
/* number of HIP devices visible (0 if none / runtime unavailable); never throws */
int gpr_device_count(void);
/* stream-ordered device-to-device copy (host code uses it to turn library-owned arrays into caller tensors) */
int gpr_memcpy_dtod(void *stream, void *dst, const void *src, long long bytes);

/* ------------------------------------------------------------------ kernel parameters */
enum gpr_kernel_kind { GPR_KERNEL_RBF = 0, GPR_KERNEL_DOT = 1 };
typedef struct gpr_kernel_params {
    int kind;        /* gpr_kernel_kind                                                        */
    double zeta;     /* power on the cosine d (RBF_mb.py:24, Dot_mb.py:10)                      */
    double sigma;    /* signal std; sigma^2 multiplies every block                              */
    double l;        /* RBF length scale (ignored for Dot)                                      */
    double sigma0;   /* Dot offset (ignored for RBF)                                            */
    double tol;      /* RBF force-force skip: pairs with K/(2 l^2) <= tol dropped               */
                     /* (rbf_kernel.cpp:395); ignored by every other block                      */
} gpr_kernel_params;

/* ------------------------------------------------------------------ packed descriptor sets */
/* A gpr_set holds the rows of a list of training / prediction POINTS (an energy point = one
 * structure: all atoms' descriptor rows; a force point = one atom: the rows of every centre
 * that feels it -- SURVEY.md App. A.1-A.2, utilities.py:340-390 `list_to_tuple`) normalised,
 * projected and laid out in v_mfma_f64_16x16x4_f64 operand order.  See DESIGN.md section 3.   */
typedef struct gpr_set gpr_set;

enum gpr_set_role {
    GPR_ROLE_STREAM = 0,   /* "set 1" side: rows streamed through LDS                          */
    GPR_ROLE_RESIDENT = 1  /* "set 2" side: rows held in registers, one point per wavefront     */
};
enum gpr_set_flags {
    GPR_SET_NUMPY_EPS = 1  /* norms get +1e-8 and small rows are NOT skipped: the convention of */
                           /* the numpy helpers behind kernel.diag (base.py:107-130,            */
                           /* Dot_mb.py:177-258)                            */
};

/* width = 0 (energy rows: x only), 3 (force rows: dxdr[n,d,3]) or 9 (stress rows, dxdr[n,d,9]);
 * col0 = first of the three derivative columns packed (0 for forces; 0, 3 or 6 select the
 * force / diagonal-stress / off-diagonal-stress triplets of a 9-wide row, see rbf_kernel.cpp:256-338).
 * counts_host[npts] = rows per point (host memory); x/dxdr/ele are DEVICE pointers with the
 * reference's row-major layouts x[nrows,d], dxdr[nrows,d,width], ele[nrows].                  */
int gpr_set_create(gpr_set **out, void *stream, int role, int d, int width, int col0, int nrows,
                   int npts, const int *counts_host, const double *x_dev, const double *dxdr_dev,
                   const int *ele_dev, int flags);
/* Same, reading the rows through index arrays (device, int32, may be NULL = identity): packed row r takes
 * x[x_index[r]], ele[x_index[r]] and dxdr[dxdr_index[r]].  This is the force-point regrouping of
 * gaussianprocess.py:852-870 (rows of every centre that feels atom a) done inside the pack kernel on the
 * descriptor kernel's own output, with the index arrays of gpr_nbr_force_rows; rows whose index is
 * negative are dropped like zero-norm rows. */
int gpr_set_create_gather(gpr_set **out, void *stream, int role, int d, int width, int col0, int nrows,
                          int npts, const int *counts_host, const double *x_dev, const double *dxdr_dev,
                          const int *ele_dev, const int *x_index_dev, const int *dxdr_index_dev, int flags);
/* Never blocks: device memory is released in stream order, host bookkeeping once the stream has passed. */
int gpr_set_destroy(gpr_set *set, void *stream);
/* introspection (tests, DESIGN.md numbers): padded rows and bytes held on the device */
int gpr_set_info(const gpr_set *set, int *npts, int *nrows, int *rows_padded, long long *bytes);

/* ------------------------------------------------------------------ covariance blocks */
/* All three write the block exactly as the reference's Python wrappers return it
 * (kee_C / kef_C / kff_C in gpr_calc/kernels/rbf_kernel.py and dot_kernel.py), i.e. with the
 * 1/n_i n_j (ee) and 1/n_i (ef) averaging and the Dot sigma^2 / zeta factors applied:
 *   ee:  out[i*ld + j]                      i < m1 energy points, j < m2 energy points
 *   ef:  out[i*ld + 3*j+b]                  (or transposed: out[(3*j+b)*ld + i])
 *   ff:  out[(3*i+a)*ld + 3*j+b]
 * `out` is a DEVICE pointer to the top-left element of the block inside a row-major matrix of
 * leading dimension ld (so blocks can be written in place into K).  If dl_out != NULL (RBF
 * only) the derivative of the block with respect to the length scale l is written there with
 * the same layout (kee/kef/kff_C(..., grad=True)'s C_l); d/dsigma is 2*block/sigma.
 * set1 must have role STREAM, set2 role RESIDENT.  `flags` (ee / ff) is a bit mask:
 *   GPR_BLOCK_SYMMETRIC  set1 and set2 pack the same points: only the upper triangle of point
 *                        pairs is computed and mirrored;
 *   GPR_BLOCK_TRANSPOSE  write the transposed block (out[j*ld + i], out[(3*j+b)*ld + 3*i+a]), which
 *                        lets predict build K(train, X) = K_trans^T directly with X streamed.   */
enum gpr_block_flags { GPR_BLOCK_SYMMETRIC = 1, GPR_BLOCK_TRANSPOSE = 2 };
int gpr_block_ee(void *stream, const gpr_kernel_params *kp, const gpr_set *set1,
                 const gpr_set *set2, int flags, double *out, long long ld,
                 double *dl_out, long long dl_ld);
int gpr_block_ef(void *stream, const gpr_kernel_params *kp, const gpr_set *eset1,
                 const gpr_set *fset2, int transpose, double *out, long long ld,
                 double *dl_out, long long dl_ld);
int gpr_block_ff(void *stream, const gpr_kernel_params *kp, const gpr_set *set1,
                 const gpr_set *set2, int flags, double *out, long long ld,
                 double *dl_out, long long dl_ld);
/* diagonal of the force-force self blocks: out[3*j+b] = kff(point j, point j)[b,b]
 * (RBF_mb.diag, RBF_mb.py:99-109; Dot_mb.diag, Dot_mb.py:72-78).  set1/set2 = the same points
 * packed in both roles. */
int gpr_diag_ff(void *stream, const gpr_kernel_params *kp, const gpr_set *set1,
                const gpr_set *set2, double *out);
/* diagonal of the energy-energy self blocks: out[j] = kee(point j, point j) */
int gpr_diag_ee(void *stream, const gpr_kernel_params *kp, const gpr_set *set1,
                const gpr_set *set2, double *out);

/* Explicit feature map of the Dot kernel at zeta = 2 (a degree-2 polynomial kernel): K = sigma^2 Psi Psi'^T with
 * per-point features Psi, so the covariance assembly is one GEMM (csrc/features.hip for the algebra; reference
 * natives dot_kee/kef/kff_many, dot_kernel.cpp:5-336).  out[npts, F] (energy set) or out[3*npts, F] (force
 * set, row 3*j+q), F = ns*(d(d+1)/2 + 1), device, caller's point order; species_host[ns] (ns <= 4) lists the
 * atomic numbers that receive a feature block. */
int gpr_dot2_features(void *stream, const gpr_set *set, int ns, const int *species_host, double sigma0,
                      double *out, long long ld);
/* Direct variant for the prediction path: the same Psi rows computed straight from the descriptor kernel's
 * x / dxdr, rows addressed through point ranges pt_row0[npts+1] (device) and the optional index arrays of
 * gpr_nbr_force_rows (NULL = identity), with no packed set; plus (diag != NULL) the prior variance
 * k(x*, x*) of every predicted quantity in the numpy-helper convention of Dot_mb.diag (Dot_mb.py:45-85,
 * 177-258).  width = 0: energy points, psi[npts, F], diag[npts]; width = 3: force points, psi[3 npts, F],
 * diag[3 npts]. */
int gpr_dot2_features_direct(void *stream, int d, int width, int npts, const int *pt_row0_dev,
                             const int *x_index_dev, const int *dxdr_index_dev, const double *x_dev,
                             const double *dxdr_dev, const int *ele_dev, int ns, const int *species_host,
                             int nsx, const int *species_extra_host,
                             double sigma, double sigma0, double *psi, long long ld, double *diag);
/* (species_extra_host[nsx]: atomic numbers that occur in the predicted points but not among the model's ns species;
 * they get no feature block -- the kernel is zero across species -- but their own row pairs are part of k(x*, x*)) */

/* Both calls of gpr_dot2_features_direct that a prediction batch needs (width 0 for its nE energy points, rows
 * [e_row0[s], e_row0[s+1]) of x; width 3 for its nF force points, rows f_row0[a] .. f_row0[a+1] of the index arrays) as
 * ONE launch: psi[nE + 3 nF, F] = [energy rows | force rows], diag[nE + 3 nF] likewise (may be NULL). */
int gpr_dot2_features_ef(void *stream, int d, int nE, const int *e_row0_dev, int nF, const int *f_row0_dev,
                         const int *x_index_dev, const int *dxdr_index_dev, const double *x_dev, const double *dxdr_dev,
                         const int *ele_dev, int ns, const int *species_host, int nsx, const int *species_extra_host,
                         double sigma, double sigma0, double *psi, long long ld, double *diag);

/* Batched prediction of the Dot kernel at zeta = 2 in one call (what GP.predict_structures does for that kernel;
 * reference: one GP.predict_structure per structure, gaussianprocess.py:834-918): host arrays of the structures in
 * (layout of gpr_nbr_build), device-resident model in (so3_table from gpr_so3_tables; w[F] and M[F, F] of the
 * weight-space fit, F = ns (d(d+1)/2 + 1); Cinv[F, F] = C^-1 lower triangular with I + Psi'^T D^-1 Psi' = C C^T may be
 * given instead of M (NULL otherwise): the variance product then skips the zero half), out_dev[(want_std ? 2 : 1) m + 4] = [mean | var | status[4]] with
 * m = nstruct + 3 nfree rows ordered energies (per atom) then (fx, fy, fz) of every free atom; free_mask[natoms]
 * (host, may be NULL = every atom free) marks the atoms that get force rows (FixAtoms, gaussianprocess.py:823-832);
 * status = [force rows the neighbour symmetry check could not find (0), overflow, neighbour pairs, seq rows].
 * cap_pairs / cap_rows > 0: sizes guessed from what an earlier call reported in status[2..3]; the
 * neighbour build then runs capped (gpr_nbr_build_device_capped) and the call queues everything without waiting for
 * the device.  overflow = 1 means the guess was too small and mean / var are meaningless: call again with 0, 0.
 * Queues neighbour search, descriptor, regrouping, features, GEMM and the tail kernel on `stream`; the caller
 * downloads out_dev. */
int gpr_predict_dot2_batch(void *stream, int nmax, int lmax, double rcut, double alpha, int nseg, int ncheb,
                           const double *so3_table, int nstruct, const int *atom0, const double *positions,
                           const double *cells, const int *pbc, const int *numbers, int weight_on,
                           const unsigned char *free_mask, int ns, const int *species_host, double sigma,
                           double sigma0, const double *w_dev, const double *M_dev, const double *Cinv_dev, int want_std,
                           int cap_pairs, int cap_rows, double *out_dev, long long out_len);

/* The same prediction with host arrays on both sides (one call per batch, what an ASE NEB loop needs per force call):
 * a session keeps the device-resident model (pointers as in gpr_predict_dot2_batch; they must stay valid), owns the
 * page-locked buffer the device writes the results into, waits for them by polling their status words, sizes the
 * neighbour tables from the densest batch seen so far and repeats a call whose guess was too small.
 *   gpr_dot2_session_predict: fixed[natoms] (may be NULL) marks FixAtoms atoms; E[nstruct] total energies, F[natoms, 3]
 *   (zero rows for fixed atoms), E_std[nstruct] per atom, F_std[natoms, 3]: the values GP.predict_structure returns
 *   (gaussianprocess.py:880-911) for every structure; info[4] (may be NULL) = the status words of the call. */
typedef struct gpr_dot2_session gpr_dot2_session;
int gpr_dot2_session_create(gpr_dot2_session **out, void *stream, int nmax, int lmax, double rcut, double alpha, int nseg,
                            int ncheb, const double *so3_table, int weight_on, int ns, const int *species_host, double sigma,
                            double sigma0, const double *w_dev, const double *M_dev, const double *Cinv_dev, int want_std);
int gpr_dot2_session_predict(gpr_dot2_session *s, int nstruct, const int *atom0, const double *positions, const double *cells,
                             const int *pbc, const int *numbers, const unsigned char *fixed, double *E, double *F,
                             double *E_std, double *F_std, double *info);
int gpr_dot2_session_set_hint(gpr_dot2_session *s, double pairs_per_atom, double rows_per_atom);   /* tests / tuning */
int gpr_dot2_session_destroy(gpr_dot2_session *s);
/* Several batches in flight (throughput; an image band evaluated batch after batch, or several walkers):
 *   gpr_dot2_session_submit   queues the whole prediction of one batch and returns at once; *ticket names it.  Every
 *                             submission in flight has its own page-locked result buffer and its own stream (the first
 *                             slot is the stream the session was created on), so the short kernels of consecutive
 *                             batches overlap on the device and the host packs / unpacks one batch while the device
 *                             works on the others.  The input arrays must stay valid until the matching _wait returns.
 *   gpr_dot2_session_wait     waits for that submission (polling its status words), repeats it with exact neighbour
 *                             table sizes if the guess was too small, and unpacks like gpr_dot2_session_predict.
 *   gpr_dot2_session_set_depth  submissions that may be in flight (1..8, default 2); one more is an error (GPR_ERR_ARG).
 * Same kernels in the same order per batch: the numbers are bit-identical to gpr_dot2_session_predict.  A session is
 * used from one host thread at a time. */
int gpr_dot2_session_submit(gpr_dot2_session *s, int nstruct, const int *atom0, const double *positions, const double *cells,
                            const int *pbc, const int *numbers, const unsigned char *fixed, int *ticket);
int gpr_dot2_session_wait(gpr_dot2_session *s, int ticket, double *E, double *F, double *E_std, double *F_std, double *info);
int gpr_dot2_session_set_depth(gpr_dot2_session *s, int depth);
/* gpr_dot2_session_predict takes slot 0 (the stream the session was created on) when no ticket holds it, otherwise any
 * free slot: synchronous calls and submissions may be mixed freely on one session.
 * Teardown checks (tests): the stream slot k works on (NULL before its first use), and the number of cached
 * stream-ordered temporaries that still remember `stream` (any != 0: all streams) -- zero for every slot stream once
 * gpr_dot2_session_destroy has returned. */
void *gpr_dot2_session_slot_stream(gpr_dot2_session *s, int k);
/* Slots 1.. take their streams from a pool per device.  HIP deals a process's streams to four hardware queues by creation order, and
 * how well the chains of a depth-4 session overlap depends on where in that order the slots' streams were created (0.115 - 0.145 ms per
 * step of the headline run): call this at the first use of a device, before anything else creates streams, and the place is fixed.
 * Idempotent.  GPR_SESSION_POOL=0: a new stream per slot at its first use instead; GPR_SESSION_POOL_LEAD: unused streams created
 * ahead of the pool (default 0; 1 - 3 are measurably worse, 4 is 0 again). */
int gpr_dot2_session_reserve_streams(void);
int gpr_dev_cache_count(void *stream, int any);

/* Kernel-only timing of the block launches with HIP events recorded on the launch stream (used by
 * bench.py's roofline leg).  gpr_profile_enable(1) starts recording for this host thread;
 * gpr_profile_fetch sums the durations (ms) and launch counts per kernel kind into arrays of 11 entries
 * (0 force-force block, 1 energy-force, 2 energy-energy, 3 diagonal-only block launches, 4 triangular product
 * gpr_trmm_lower, 5 gpr_gemm, 6 SO3 descriptor (the centre kernel), 7 Dot feature map, 8 row packing, 9 gpr_gemv_t, 10 the descriptor's radial pre-pass) and clears the
 * records. */
int gpr_profile_enable(int on);
int gpr_profile_fetch(double *ms_by_kind, int *launches_by_kind);
/* gpr_profile_enable(n) with n > 1 records every n-th launch of a kind only (events cost host time inside the region
 * they measure); gpr_profile_counts gives the launches seen since then, recorded or not */
int gpr_profile_counts(int *launches_seen_by_kind);

/* Multi-GPU sharding of the block calls issued from this host thread: with (nshards, shard) set, a
 * gpr_block_* call evaluates only the point-pair tasks t with t % nshards == shard and leaves every
 * other output element untouched (zero after the call's own zero-fill).  Summing the outputs of all
 * shards (one RCCL all-reduce) gives the full block exactly, because every element is produced by
 * exactly one shard.  (1, 0) restores the default.  Role of the reference's MPI row chunking,
 * RBF_mb.py:257-269, 471-481. */
/* The L0 natives are `void` like the reference's: one that cannot run on the device (descriptor longer than this build
 * packs, HIP error) prints the reason, leaves its output buffers untouched and returns; this reports why (GPR_OK = it ran). */
int gpr_l0_last_status(void);
int gpr_set_shard(int nshards, int shard);

/* ------------------------------------------------------------------ neighbour bookkeeping (host) */
/* Periodic neighbour search of a batch of structures with the semantics of the reference's
 * ase NeighborList(cutoffs = rcut/2, bothways, no self interaction, skin 0) use (SO3.py:348-407) and the
 * (centre, neighbour) row table `seq` (SO3.py:386-399).  Host arrays in, host arrays out (plain index
 * work; the arithmetic on the pairs happens in gpr_so3_power_spectrum).
 *   atom0[nstruct+1] atom offsets; positions[natoms,3]; cells[nstruct,9] row vectors; pbc[nstruct,3];
 *   numbers[natoms]; weight_on: neighbours of another species get weight -Z (SO3.py:381-385).         */
typedef struct gpr_nbr gpr_nbr;
int gpr_nbr_build(gpr_nbr **out, int nstruct, const int *atom0, const double *positions,
                  const double *cells, const int *pbc, const int *numbers, double rcut, int weight_on);
int gpr_nbr_sizes(const gpr_nbr *nb, int *npairs, int *nrows);
/* vec[P,3], w[P], centre_pair0[natoms+1], centre_row0[natoms+1], row_pair0[S+1], self_row[natoms],
 * seq[S,2] (structure-local atom ids), struct_row0[nstruct+1]; any pointer may be NULL */
int gpr_nbr_export(const gpr_nbr *nb, double *vec, double *w, int *centre_pair0, int *centre_row0,
                   int *row_pair0, int *self_row, int *seq, int *struct_row0);
int gpr_nbr_destroy(gpr_nbr *nb);
/* Device builder: same inputs (host arrays) and the same pairs / rows in the same order, but the search
 * runs on the GPU (one wavefront per centre atom) and the results stay in HBM.  Synchronises the stream
 * once (the pair / row totals size every later allocation).  gpr_nbr_sizes works on the handle;
 * gpr_nbr_export downloads (debug / single-structure path; centre_row0 and struct_row0 are always on the
 * host); gpr_nbr_destroy releases the device arrays in stream order. */
int gpr_nbr_build_device(gpr_nbr **out, void *stream, int nstruct, const int *atom0, const double *positions,
                         const double *cells, const int *pbc, const int *numbers, double rcut, int weight_on);
/* device pointers of a device-built handle, in the layouts of gpr_nbr_export, plus the uploaded numbers[natoms] and
 * atom0[nstruct+1] (any pointer may be NULL); they live until gpr_nbr_destroy */
int gpr_nbr_device_arrays(const gpr_nbr *nb, const double **vec, const double **w, const int **centre_pair0,
                          const int **centre_row0, const int **row_pair0, const int **self_row,
                          const int **seq, const int **numbers, const int **atom0);
/* Force points on the device (gaussianprocess.py:852-870): point p is moving atom atoms[p] (global id);
 * its rows are the seq rows (i, a) of every centre i that has a as a neighbour, ascending in i.  By the
 * symmetry of the neighbour relation their number equals the row count of centre a, so the caller knows
 * pt_row0[npts+1] (exclusive prefix of those counts, device) from the host copy of centre_row0.
 * Outputs (device, int32, nrows = pt_row0[npts]): row_index = seq row of (i, a) (index into dxdr),
 * x_index = global id of i (index into x / numbers).  atoms_dev = pt_row0_dev = NULL means every atom moves
 * (npts = natoms, nrows = S): point a is atom a and its row range is centre_row0 -- nothing to upload. */
int gpr_nbr_force_rows(const gpr_nbr *nb, int npts, const int *atoms_dev, const int *pt_row0_dev, int nrows,
                       int *row_index_dev, int *x_index_dev);
/* pt_row0_out_dev[npts + 1] for gpr_nbr_force_rows from the list of moving atoms alone (device scan of their row counts) */
int gpr_nbr_point_rows(const gpr_nbr *nb, int npts, const int *atoms_dev, int *pt_row0_out_dev);
/* number of (i, a) rows gpr_nbr_force_rows could not find (0 unless the symmetry above is broken); synchronises */
int gpr_nbr_check(const gpr_nbr *nb, int *nmissing);
/* same count written as a double to dst_dev[0] in stream order (rides along with the caller's result download) */
int gpr_nbr_status_to(const gpr_nbr *nb, double *dst_dev);
/* gpr_nbr_build_device with caller-guessed capacities for the pair and seq-row arrays: the count pass is not waited
 * for, gpr_nbr_sizes returns the capacities and gpr_nbr_export is refused.  If the true sizes exceed the guess the
 * handle describes a table that stays inside the capacities -- the structures that fitted (a prefix of the batch; none
 * of them with the five-launch builder) and empty ranges for every centre behind them -- so whatever was queued behind
 * it touches nothing out of bounds; gpr_nbr_info_to reports the overflow and the true sizes, and the caller discards
 * the results and repeats the build with exact sizes. */
int gpr_nbr_build_device_capped(gpr_nbr **out, void *stream, int nstruct, const int *atom0, const double *positions,
                                const double *cells, const int *pbc, const int *numbers, double rcut, int weight_on,
                                int cap_pairs, int cap_rows);
/* gpr_nbr_build_device(_capped) with n_extra caller ints appended to the geometry upload (e.g. the list of moving atoms
 * gpr_nbr_point_rows / gpr_nbr_force_rows want on the device): they arrive with it instead of through a copy of their
 * own; cap_pairs = cap_rows = 0 selects the exact build.  gpr_nbr_extra_ptr: where they are (device), NULL if none. */
int gpr_nbr_build_device_extra(gpr_nbr **out, void *stream, int nstruct, const int *atom0, const double *positions,
                               const double *cells, const int *pbc, const int *numbers, double rcut, int weight_on,
                               int cap_pairs, int cap_rows, int n_extra, const int *extra_host);
const int *gpr_nbr_extra_ptr(const gpr_nbr *nb);
/* Capped build of a batch in which every atom moves AND the force-point regrouping (what gpr_nbr_force_rows(nb, natoms,
 * NULL, NULL, cap_rows, row_index, x_index) writes) in one call: where the one-launch builder applies (<= 96 structures of
 * <= 128 atoms) the whole neighbour search of a batch is ONE kernel launch -- geometry upload, count, offsets across the
 * structures by look-back, fill and regrouping -- instead of five; *rows_done = 0 means the regrouping still has to be
 * asked for with gpr_nbr_force_rows. */
int gpr_nbr_build_device_rows(gpr_nbr **out, void *stream, int nstruct, const int *atom0, const double *positions,
                              const double *cells, const int *pbc, const int *numbers, double rcut, int weight_on,
                              int cap_pairs, int cap_rows, int *row_index_dev, int *x_index_dev, int *rows_done);
int gpr_nbr_set_fused(int on);      /* tests / A-B timing: 0 keeps every capped build on the five-launch path */
/* The one-launch builder's workgroups wait for each other, so it is only taken while every such workgroup in flight in
 * the process (all streams) plus the new ones fill at most half of what the device keeps resident at once
 * (hipOccupancyMaxActiveBlocksPerMultiprocessor x CU count for the build's LDS need; more than the 64 KB default of
 * dynamic LDS is requested from the runtime, and a build the device cannot hold takes the five-launch path).
 * gpr_nbr_fused_state: that capacity (largest shape admitted so far on this device) and the workgroups counted as in
 * flight right now.  gpr_nbr_debug_fused_capacity: tests pretend a capacity (< 0: back to the runtime's answer). */
int gpr_nbr_fused_state(int *capacity, int *in_flight);
int gpr_nbr_debug_fused_capacity(int workgroups);
/* dst_dev[0..3] = [missing force rows, overflow (capped builds), true pair count, true seq-row count] in stream order */
int gpr_nbr_info_to(const gpr_nbr *nb, double *dst_dev);
/* the four device ints gpr_nbr_info_to converts (for kernels that deliver them themselves, gpr_ws_tail_status) */
const int *gpr_nbr_info_ptr(const gpr_nbr *nb);

/* ------------------------------------------------------------------ SO(3) power-spectrum descriptor */
/* Tables of one descriptor configuration, built on the device once: piecewise Chebyshev coefficients (nseg
 * equal segments of [0, rcut], ncheb coefficients each) of the radial functions
 *   F0_nl(r) = 4 pi exp(-alpha r^2) sum_q G[n][q] i_l(2 alpha r rq[q]),   F1_nl = dF0_nl/dr
 * -- the reference's 10(nmax+lmax+1)-node Gauss-Chebyshev quadrature against modified spherical Bessel functions
 * (SO3.py:446-453, 563-565, 623-655), evaluated at the Chebyshev nodes -- plus the Y_lm normalisation and
 * gradient coefficient tables (SO3.py:682-707).  G[nmax, nq], rq[nq] and table[gpr_so3_table_size()] are device
 * arrays. */
long long gpr_so3_table_size(int nmax, int lmax, int nseg, int ncheb);
int gpr_so3_tables(void *stream, int nmax, int lmax, double rcut, double alpha, int nq, const double *G,
                   const double *rq, int nseg, int ncheb, double *table);
/* F0_nl, F1_nl at radii rs[nr] (device) into out[nr, 2 nmax (lmax+1)] (index 2 (n (lmax+1) + l) + {0, 1}):
 * mode 0 = the direct quadrature, mode 1 = the table.  Validation entry point (tests compare the two). */
int gpr_so3_radial(void *stream, int mode, int nmax, int lmax, double rcut, double alpha, int nq, const double *G,
                   const double *rq, int nseg, int ncheb, const double *table, int nr, const double *rs, double *out);
/* x[ncentres, d] and dxdr[nrows, d, 3] (d = nmax(nmax+1)/2 (lmax+1)) from neighbour pairs grouped by
 * centre atom and, inside a centre, by `seq` row (gpr_calc/SO3.py:186-323, 608-727).
 *   table                        from gpr_so3_tables for the same (nmax, lmax, rcut, alpha, nseg, ncheb), device
 *   vec[npairs, 3], wj[npairs]   displacement and weight (Z_j) of every neighbour pair, device
 *   centre_pair0[ncentres+1]     pair range of each centre
 *   centre_row0[ncentres+1]      seq-row range of each centre
 *   row_pair0[nrows+1]           pair range of each seq row (i, j)
 *   self_row[ncentres]           seq row (i, i), which receives minus the sum of the other rows           */
int gpr_so3_power_spectrum(void *stream, int nmax, int lmax, double rcut, double alpha, int nseg, int ncheb,
                           const double *table, int ncentres, int npairs, int nrows,
                           const double *vec, const double *wj, const int *centre_pair0,
                           const int *centre_row0, const int *row_pair0, const int *self_row,
                           double *x, double *dxdr);
/* The configuration of the reference's defaults, SO3(nmax = 3, lmax = 4), has a kernel instance with every size known at
 * compile time; every other (nmax, lmax) -- and the stress rows -- take the generic kernel.  gpr_so3_set_variant(1) sends
 * everything through the generic kernel (tests compare the two), 0 restores the default. */
int gpr_so3_set_variant(int generic_only);
/* The compile-time instance forms the radial channels of large batches in a streaming pre-pass of its own (a second, small
 * launch); batches with fewer than `ncentres` centres (default 512) keep everything in one launch.  0: always, INT_MAX: never. */
int gpr_so3_set_prepass_min(int ncentres);
/* same plus the stress rows rdxdr[nrows, d, 3, 3] = -pstress / volume of SO3.calculate(stress=True)
 * (SO3.py:240-306): positions[ncentres, 3] = centre positions as the caller gave them (not wrapped),
 * inv_vol[ncentres] = 1 / cell volume of the centre's structure, both device; rdxdr == NULL: as above */
int gpr_so3_power_spectrum_stress(void *stream, int nmax, int lmax, double rcut, double alpha, int nseg, int ncheb,
                                  const double *table, int ncentres, int npairs, int nrows,
                                  const double *vec, const double *wj, const int *centre_pair0,
                                  const int *centre_row0, const int *row_pair0, const int *self_row,
                                  double *x, double *dxdr, const double *positions, const double *inv_vol,
                                  double *rdxdr);

/* ------------------------------------------------------------------ dense FP64 linear algebra */
/* Row-major device matrices.  Lower Cholesky in place: on return the lower triangle of a holds
 * L (a = L L^T), the strict upper triangle is zeroed (scipy.linalg.cholesky(lower=True)
 * semantics, gaussianprocess.py:297).  info_dev (device int, may be NULL) receives 0 or the
 * 1-based index of the first non-positive pivot. */
int gpr_potrf_lower(void *stream, int n, double *a, long long lda, int *info_dev);
/* same, and additionally stores the inverse of every 64 x 64 diagonal block of L (zero padded, row-major,
 * block b at dinv + b*4096) into the device buffer dinv[ceil(n/64)*4096] */
int gpr_potrf_lower_dinv(void *stream, int n, double *a, long long lda, int *info_dev, double *dinv);
/* For n > 768 the factorisation runs a look-ahead schedule on two private streams of the device and, where kernels of
 * different streams overlap in time (probed, and probed again when the answer is older than a second), hands operand
 * tiles over between RUNNING kernels through polled counters.  A poll that is never answered gives up after a bounded
 * number of polls and info_dev becomes -1: NOT a statement about the matrix (which is then partly overwritten) but
 * about the schedule.  The caller restores the matrix, calls gpr_potrf_set_polling(0) -- every hand-over becomes a
 * stream event from then on -- and factors again (gpr_calculator_amd/device.py: potrf_factor).  Calls on one device
 * are serialised (host mutex + an event the next call's stream waits for); calls on different devices are independent.
 *   gpr_potrf_set_polling(0)   events only, for the rest of the process;  (-1): probe again at the next call
 *   gpr_potrf_polling()        1 when the next factorisation on the current device would poll
 *   gpr_potrf_debug_max_polls  poll budget on the current device (<= 0: default, about 10 s); tests force the time-out */
int gpr_potrf_set_polling(int mode);
int gpr_potrf_polling(void);
int gpr_potrf_debug_max_polls(int polls);
/* Pieces of the factorisation for a matrix whose rows are spread over several GPUs (1-D block-cyclic row blocks of 256;
 * gpr_calculator_amd/distributed.py drives them, RCCL moves `ops` and the panels):
 *   gpr_potrf_diag_block   factor the kb x kb (kb <= 256) diagonal block at a in place and write its operand tiles
 *                          (gpr_potrf_ops_doubles() doubles) to ops
 *   gpr_potrf_panel_rows   p[rows, kb] <- p L_kk^-T for `rows` rows of the panel below that block, from ops
 *   gpr_gemm_nt_cyclic_lower   the trailing update C -= A B^T of this rank's rows, skipping tiles above the global diagonal
 *   gpr_potrf_finish       dinv (as gpr_potrf_lower_dinv) of an assembled factor and zeroing of its strict upper triangle
 *   gpr_mirror_lower       a[i][j] = a[j][i] for j > i (covariance matrix gathered from lower-trapezoid row slabs)        */
int gpr_potrf_ops_doubles(void);
int gpr_potrf_diag_block(void *stream, int kb, double *a, long long lda, int k0, int *info_dev, double *ops);
int gpr_potrf_panel_rows(void *stream, int kb, const double *ops, double *p, long long lda, int rows);
int gpr_gemm_nt_cyclic_lower(void *stream, int m, int n, int k, double alpha, const double *a, long long lda,
                             const double *b, long long ldb, double beta, double *c, long long ldc, int nb, int nranks,
                             int rank, int lb0, int col0);
int gpr_potrf_finish(void *stream, int n, double *l, long long ldl, double *dinv);
/* C = X^T X of a lower-triangular X (zeros above its diagonal), written in full (computed as the lower triangle with the
 * zero part of every k range skipped, then mirrored): K^-1 = L^-T L^-1 from X = L^-1 (gaussianprocess.py:128-131) */
int gpr_xtx_lower(void *stream, int n, const double *x, long long ldx, double *c, long long ldc);
/* X = L^-1 of a lower-triangular L on the matrix cores (diagonal blocks of 256 rows at once, the blocks below them
 * recursively by halves).  Written: the lower triangle of X and the zeros above the diagonal inside the 256 x 256 diagonal
 * blocks; the blocks above those are left untouched -- hand over a zeroed X. */
int gpr_trtri_lower(void *stream, int n, const double *l, long long ldl, double *x, long long ldx);
int gpr_mirror_lower(void *stream, int n, double *a, long long lda);
/* Solve L X = B (trans = 0) or L^T X = B (trans = 1) in place; B is n x nrhs row-major, ldb. */
int gpr_trsm_lower(void *stream, int trans, int n, int nrhs, const double *l, long long ldl,
                   double *b, long long ldb);
/* same with the diagonal-block inverses of gpr_potrf_lower_dinv: every block solve is a small product */
int gpr_trsm_lower_dinv(void *stream, int trans, int n, int nrhs, const double *l, long long ldl,
                        const double *dinv, double *b, long long ldb);
/* With up to 8 right-hand sides and 2 .. 1024 blocks of 64 rows the solve is ONE launch whose workgroups hand their
 * solutions on by polling (csrc/dense.hip, trsv_chain_kernel).  A workgroup whose wait outlasts the poll budget gives up;
 * a one-workgroup kernel queued behind the launch then finishes the substitution from where the chain stopped, so the
 * caller always receives the solution (never NaNs for a time-out).  gpr_trsv_chain_fallbacks: how often that has happened
 * in this process (0 in a healthy run; readable at any time, meaningful once the stream has been synchronised).
 * gpr_trsv_debug_max_polls: the poll budget (tests force the fallback with 1; <= 0 restores the default). */
int gpr_trsv_chain_fallbacks(void);
int gpr_trsv_debug_max_polls(int polls);
/* C[n,nrhs] = T B with T lower triangular n x n: the strict upper triangle of T must hold zeros; whole
 * 128-column blocks to the right of a row block's diagonal block are skipped (never read) */
int gpr_trmm_lower(void *stream, int n, int nrhs, const double *t, long long ldt, const double *b,
                   long long ldb, double *c, long long ldc);

/* out[j] = | T B[:, j] |^2, T lower triangular [n x n], B [n x nrhs] row-major: gpr_trmm_lower followed by gpr_colsumsq.
 * With nrhs <= 256 (the 1 + 3 n quantities of one structure) a single kernel streams T once and never writes the product;
 * wider B goes through the two calls.  This is the subtracted term of the predictive variance, |L^-1 k*|^2
 * (reference gaussianprocess.py:903-907 evaluates k*^T K^-1 k*). */
int gpr_trmm_lower_sumsq(void *stream, int n, int nrhs, const double *t, long long ldt, const double *b,
                         long long ldb, double *out);
/* C[m,n] = alpha * op(A) op(B) + beta * C, row-major, op = transpose if the flag is set */
int gpr_gemm(void *stream, int transa, int transb, int m, int n, int k, double alpha,
             const double *a, long long lda, const double *b, long long ldb, double beta,
             double *c, long long ldc);
/* out[j] = sum_i a[i*lda+j] * x[i], j < ncols (A^T x of a row-major n x ncols matrix) */
int gpr_gemv_t(void *stream, int n, int ncols, const double *a, long long lda, const double *x, double *out);
/* a[i*lda+i] += (i < n_first ? add_first : add_rest): the noise diagonal (gaussianprocess.py:290-294) */
int gpr_add_diag(void *stream, int n, double *a, long long lda, int n_first, double add_first,
                 double add_rest);
/* out[j] = sum_i v[i*ldv + j]^2  (j < ncols): column sums of squares of an n x ncols matrix */
int gpr_colsumsq(void *stream, int n, int ncols, const double *v, long long ldv, double *out);
/* out[i] = sum_j a[i*lda+j] * b[i*ldb+j], i < m: row-wise inner products (weight-space variance of the Dot
 * kernel: psi^T M psi for every predicted quantity) */
int gpr_rowdot(void *stream, int m, int n, const double *a, long long lda, const double *b, long long ldb,
               double *out);
/* Tail of the weight-space prediction (Dot kernel, zeta = 2): out[i] = P[i,:] . w (mean) and, when Q = P M is
 * given, out[m+i] = max(diag[i] - Q[i,:] . P[i,:], 0) (variance, gaussianprocess.py:903-907), one pass */
int gpr_ws_finish(void *stream, int m, int n, const double *P, long long ldp, const double *Q, long long ldq,
                  const double *w, const double *diag, double *out, double vsq);
/* (vsq > 0 selects the factored form of the same variance: Q = P C^-T with I + Psi'^T D^-1 Psi' = C C^T, and
 * out[m+i] = max(diag[i] - vsq (|P[i,:]|^2 - |Q[i,:]|^2), 0), vsq = sigma^2; vsq = 0: the form above)
 * C[m, n] = alpha A[m, n] B^T with B n x n lower triangular (zero blocks of B^T are skipped) */
/* The same tail with the product fused in: out = [P w | max(diag - vsq (|p|^2 - |p C^-T|^2), 0)] from P[m, n], the lower
 * triangular C^-1[n, n], w[n], diag[m]; Q = P C^-T is formed tile by tile on the matrix cores and only its row sums of
 * squares are kept (no m x n intermediate). */
int gpr_ws_tail(void *stream, int m, int n, const double *P, long long ldp, const double *Cinv, long long ldc,
                const double *w, const double *diag, double *out, double vsq);
/* gpr_ws_tail followed by status_dst[0..3] = (double)status_src[0..3] (device ints, e.g. gpr_nbr_info_ptr): the words are
 * written strictly after the results -- by the kernel that writes the results last where the launch shape allows it --
 * so a host that polls them in page-locked memory may read the results as soon as it sees them */
int gpr_ws_tail_status(void *stream, int m, int n, const double *P, long long ldp, const double *Cinv, long long ldc,
                       const double *w, const double *diag, double *out, double vsq, const int *status_src, double *status_dst);
int gpr_gemm_nt_lower(void *stream, int m, int n, double alpha, const double *a, long long lda, const double *b_lower,
                      long long ldb, double *c, long long ldc);
/* out[0] = sum_ij a[i*lda+j] * b[i*ldb+j] over an m x n window (trace terms of the LML gradient,
 * gaussianprocess.py:196-198), deterministic reduction order */
int gpr_frob_dot(void *stream, int m, int n, const double *a, long long lda, const double *b,
                 long long ldb, double *out);
/* out[0] = sum_i log(a[i*lda+i]) (deterministic tree reduction) */
int gpr_sum_log_diag(void *stream, int n, const double *a, long long lda, double *out);

/* ------------------------------------------------------------------ L0: legacy natives */
void rbf_kee_many(int n1, int n2, int d, int x2i, double zeta, double sigma2, double l2,
                  double *x1, int *ele1, int *x1_inds, double *x2, int *ele2, int *x2_inds,
                  double *pout);
void rbf_kee_many_with_grad(int n1, int n2, int d, int x2i, double zeta, double sigma2, double l2,
                            double *x1, int *ele1, int *x1_inds, double *x2, int *ele2,
                            int *x2_inds, double *pout, double *dpout_dl);
void rbf_kef_many(int n1, int n2, int d, int x2i, double zeta, double sigma2, double l2,
                  double *x1, int *ele1, int *x1_inds, double *x2, double *dx2dr, int *ele2,
                  int *x2_inds, double *pout);
void rbf_kef_many_with_grad(int n1, int n2, int d, int x2i, double zeta, double sigma2, double l,
                            double *x1, int *ele1, int *x1_inds, double *x2, double *dx2dr,
                            int *ele2, int *x2_inds, double *pout);
void rbf_kef_many_stress(int n1, int n2, int d, int x2i, double zeta, double sigma2, double l2,
                         double *x1, int *ele1, int *x1_inds, double *x2, double *dx2dr,
                         int *ele2, int *x2_inds, double *pout);
void rbf_kff_many(int n1, int n2, int n2_start, int n2_end, int d, int x2i, double zeta,
                  double sigma2, double l2, double tol, double *x1, double *dx1dr, int *ele1,
                  int *x1_inds, double *x2, double *dx2dr, int *ele2, int *x2_inds, double *pout);
void rbf_kff_many_with_grad(int n1, int n2, int n2_start, int n2_end, int d, int x2i,
                            double zeta, double sigma2, double l, double *x1, double *dx1dr,
                            int *ele1, int *x1_inds, double *x2, double *dx2dr, int *ele2,
                            int *x2_inds, double *pout, double *dpout_dl);
void rbf_kff_many_stress(int n1, int n2, int n2_start, int n2_end, int d, int x2i, double zeta,
                         double sigma2, double l2, double tol, double *x1, double *dx1dr,
                         int *ele1, int *x1_inds, double *x2, double *dx2dr, int *ele2,
                         int *x2_inds, double *pout);
void dot_kee_many(int n1, int n2, int d, int x2i, double zeta, double sigma2, double sigma02,
                  double *x1, int *ele1, int *x1_inds, double *x2, int *ele2, int *x2_inds,
                  double *pout);
void dot_kef_many(int n1, int n2, int d, int x2i, double zeta, double *x1, int *ele1,
                  int *x1_inds, double *x2, double *dx2dr, int *ele2, int *x2_inds, double *pout);
void dot_kef_many_stress(int n1, int n2, int d, int x2i, double zeta, double *x1, int *ele1,
                         int *x1_inds, double *x2, double *dx2dr, int *ele2, int *x2_inds,
                         double *pout);
void dot_kff_many(int n1, int n2, int n2_start, int n2_end, int d, int x2i, double zeta,
                  double *x1, double *dx1dr, int *ele1, int *x1_inds, double *x2, double *dx2dr,
                  int *ele2, int *x2_inds, double *pout);
void dot_kff_many_stress(int n1, int n2, int n2_start, int n2_end, int d, int x2i, double zeta,
                         double *x1, double *dx1dr, int *ele1, int *x1_inds, double *x2,
                         double *dx2dr, int *ele2, int *x2_inds, double *pout);

#ifdef __cplusplus
}
#endif
#endif /* GPR_HIP_H */
