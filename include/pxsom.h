/*
 * pxsom.h -- C ABI of libpxsom.so: the MI355X (gfx950) implementation of ark-analysis' Pixie
 * pixel/cell SOM hot path.
 *
 * The reference (pure Python) reaches its native arithmetic through exactly two foreign calls
 * into the Cython package pyFlowSOM (/root/reference/src/ark/phenotyping/cluster_helpers.py:14):
 *     som(data, xdim, ydim, rlen, alpha_range, seed)            cluster_helpers.py:106-109
 *     map_data_to_nodes(weights, data)[0]                       cluster_helpers.py:152-157
 * plus numpy/scipy/pandas loops for normalisation, blur, quantiles and per-cluster means
 * (pixie_preprocessing.py:47-75, pixel_cluster_utils.py:16-142, 369-404).  Every entry point
 * below names the reference interface it replaces.  INTEGRATION.md shows the ctypes binding a
 * reference maintainer would add.
 *
 * Conventions
 *   - plain C, no exceptions cross the ABI; every function returns PXSOM_OK (0) or a negative
 *     pxsom_status; pxsom_last_error() returns a thread-local message for the last failure.
 *   - "dev" pointers are device (HBM) pointers owned by the caller (e.g. torch tensors'
 *     data_ptr()); the library allocates nothing persistent.  Scratch comes from a caller
 *     workspace sized by the matching *_workspace_bytes() query.
 *   - every launch is ordered on the caller's stream (void* = hipStream_t; NULL = default
 *     stream); calls are asynchronous unless documented otherwise; re-entrant per stream.
 *   - matrices are row-major [rows, c] with a row stride ldx given in ELEMENTS.
 *   - SOM nodes: node k = x*ydim + y on the xdim*ydim grid; labels are 1-based (k+1), as
 *     pyFlowSOM returns them (tests/phenotyping/cluster_helpers_test.py:388-391 of the reference).
 */
#ifndef PXSOM_H
#define PXSOM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PXSOM_ABI_VERSION 9

typedef enum pxsom_status {
    PXSOM_OK = 0,
    PXSOM_ERR_INVALID_ARG = -1,
    PXSOM_ERR_UNSUPPORTED = -2, /* shape / dtype outside what the kernels are built for */
    PXSOM_ERR_WORKSPACE = -3,   /* workspace NULL or too small */
    PXSOM_ERR_HIP = -4          /* a HIP runtime call failed; see pxsom_last_error() */
} pxsom_status;

typedef enum pxsom_dtype {
    PXSOM_F32 = 0, /* IEEE binary32 pixel matrix (BASELINE.json configs 2-4) */
    PXSOM_F64 = 1, /* IEEE binary64 pixel matrix (what the reference's feather tables hold) */
    PXSOM_F16 = 2  /* IEEE binary16 pixel matrix (BASELINE.json config 5); every value is used as the exact
                      real number it encodes -- results equal those for the same values held in binary64 */
} pxsom_dtype;

/* Limits of the gfx950 kernels in this build.  Rows of up to 128 channels take the MFMA filter + exact recheck; wider
 * rows (the cell SOM over the 400 cluster counts of a 20 x 20 pixel SOM: cell_cluster_utils.py:63-192) are evaluated
 * directly in the oracle's arithmetic -- same labels, a lower rate, sized for cell tables (10^5 .. 10^6 rows). */
#define PXSOM_MAX_CHANNELS 1024
#define PXSOM_MAX_NODES 1024

int pxsom_abi_version(void);
const char *pxsom_last_error(void);

/* ---- in-library kernel timer --------------------------------------------------------------
 * HIP event pairs recorded on the launch stream immediately around the dominant kernel of each
 * pxsom_assign call (the BMU filter kernel) whose row count is >= min_rows, while a profiler is
 * attached to the calling thread.  pxsom_prof_collect synchronises on the recorded events and
 * returns the summed kernel time and the number of launches, then resets.  bench.py uses it for
 * the roofline figure (the same duration rocprofv3 --kernel-trace reports for that kernel). */
int pxsom_prof_create(void **out_handle);
int pxsom_prof_destroy(void *handle);
int pxsom_prof_attach(void *handle_or_null, int64_t min_rows);
int pxsom_prof_collect(void *handle, double *total_ms, int64_t *launches);

/* ---- host helper (no GPU) ------------------------------------------------------------------
 * glibc rand() stream (TYPE_3 additive feedback), the presentation-order generator of the
 * pyFlowSOM-compatible som() front end (replaces the libc srand/rand pair inside pyFlowSOM's
 * Cython loop, cluster_helpers.py:106-109 passes `seed`). out[i] in [0, 2^31). */
int pxsom_host_glibc_rand_fill(uint32_t seed, int64_t count, int32_t *out);

/* ---- BMU assignment: replaces pyFlowSOM.map_data_to_nodes -----------------------------------
 * reference: cluster_helpers.py:150-157 (PixieSOMCluster.generate_som_clusters).
 * labels_dev[i] = 1 + argmin_k sqrt(sum_j (x_ij - w_kj)^2), evaluated as the reference does in
 * binary64 with first-minimum tie-break; rows containing NaN get label 0 (FlowSOM's minid=-1).
 *   x_dev      [n, c] row-major, dtype, row stride ldx elements
 *   w_dev      [k, c] row-major binary64 codebook
 *   labels_dev [n] int32 out
 *   dist_dev   [n] binary64 out, or NULL (the reference discards it: `[0]` at :157)
 * Mechanism: fp16-split MFMA filter + exact binary64 re-evaluation of every row whose two best
 * scores are closer than a rigorous error bound (DESIGN.md "K7").  Result is independent of
 * the filter: bit-identical to the oracle. */
size_t pxsom_assign_workspace_bytes(int64_t n, int c, int k);
int pxsom_assign(const void *x_dev, int64_t n, int c, int64_t ldx, int dtype, const double *w_dev,
                 int k, int32_t *labels_dev, double *dist_dev, void *workspace_dev,
                 size_t workspace_bytes, void *stream);
/* pxsom_assign with flags (ABI 9).  PXSOM_ASSIGN_SCREEN_ALL_LISTS: every list of rows for the exact path goes through the
 * long-list kernel (binary32 screening, binary64 for the surviving nodes), whatever its length -- by default lists shorter than
 * ~2.25e6 / c rows take the wave-per-row kernel.  Same labels either way; the flag exists so that the long-list kernel can be
 * exercised on small inputs (a per-call argument: the library reads no environment variables). */
#define PXSOM_ASSIGN_SCREEN_ALL_LISTS 1
int pxsom_assign_ex(const void *x_dev, int64_t n, int c, int64_t ldx, int dtype, const double *w_dev,
                    int k, int32_t *labels_dev, double *dist_dev, void *workspace_dev,
                    size_t workspace_bytes, int flags, void *stream);

/* Number of rows the last pxsom_assign on this workspace sent to the exact binary64 path
 * (diagnostic; synchronises the stream). */
int pxsom_assign_last_exact_rows(const void *workspace_dev, void *stream, int64_t *out_rows);

/* ---- per-cluster sums / counts: replaces the pandas groupby in compute_pixel_cluster_channel_avg
 * reference: pixel_cluster_utils.py:369-404.  ADDS into sums_dev [k, c] binary64 and
 * counts_dev [k] int64 (caller zeroes them; lets several FOVs / ranks accumulate):
 *   sums[labels[i]-1, :] += x[i, :];  counts[labels[i]-1] += 1     (labels outside 1..k skipped)
 * Also the accumulation half of the batch SOM rule (step 2 of DESIGN.md "K6b"). */
int pxsom_cluster_sums(const void *x_dev, int64_t n, int c, int64_t ldx, int dtype,
                       const int32_t *labels_dev, int k, double *sums_dev, int64_t *counts_dev,
                       void *stream);

/* ---- labels and per-cluster sums / counts in one pass over x ------------------------------------------
 * pxsom_assign followed by pxsom_cluster_sums, reading the pixel matrix ONCE: what cluster_pixels + generate_som_avg_files
 * compute together when the labelled rows are still in HBM (reference: cluster_helpers.py:150-157 +
 * pixel_cluster_utils.py:369-404).  labels_dev as pxsom_assign; sums_dev / counts_dev are ADDED into, as pxsom_cluster_sums.
 * Register-resident shapes (K = 97..100, even c <= 32, pair-aligned rows) take the fused launch; other shapes run the two
 * kernels one after the other.  Workspace: pxsom_assign_sums_workspace_bytes(n, c, k).
 * ACCURACY CONTRACT (since ABI 7): labels and counts are exact; on the fused launch the SUMS are accumulated in 64-bit
 * fixed point inside a workgroup -- every value enters rounded to a multiple of 2^-s, s = 46 + e - max(11, ceil(log2(rows a
 * workgroup meets))), 2^e the filter's power-of-two scale (|W|max * 2^e in [128, 256)) -- i.e. an absolute error per value of at
 * most 2^-28 * rows-per-workgroup / 2^11 relative to the codebook's largest magnitude (1.8e-12 for 41 K rows per workgroup),
 * then flushed as binary64.  They are therefore NOT the bit-exact binary64 sums of pxsom_assign + pxsom_cluster_sums, but
 * within 1e-6 relative of them per mean (tests: test_assign_sums_one_pass_equals_two_passes).  Values outside the format
 * (>= 2^(16 - e), non-finite) and listed rows bypass the table in binary64.  (ABI 9: the environment switch PXSOM_SUMS_F64 is
 * gone -- the library reads no environment variable; exact binary64 tables: pxsom_assign + pxsom_cluster_sums.) */
size_t pxsom_assign_sums_workspace_bytes(int64_t n, int c, int k);
/* ABI 9.  The workspace of pxsom_assign_sums / pxsom_assign_means begins with a statistics region of
 * pxsom_assign_sums_scratch_bytes(c, k) bytes (it does not move with n); the assign workspace follows it (pass
 * `workspace + scratch bytes` to pxsom_assign_last_exact_rows).  Every successful call LEAVES THAT REGION ZERO.  The _ex forms take
 * flags: PXSOM_TABLES_SCRATCH_CLEAN -- the caller vouches that the region is zero on entry (it cleared the workspace once when it
 * allocated it, and every call since returned PXSOM_OK): the call skips its clearing launch (~6 us in front of a 0.22 ms kernel).
 * After a failed call, clear the region (or drop the flag once). */
#define PXSOM_TABLES_SCRATCH_CLEAN 1
size_t pxsom_assign_sums_scratch_bytes(int c, int k);
int pxsom_assign_sums_ex(const void *x_dev, int64_t n, int c, int64_t ldx, int dtype, const double *w_dev, int k,
                         int32_t *labels_dev, double *sums_dev, int64_t *counts_dev, void *workspace_dev,
                         size_t workspace_bytes, int flags, void *stream);
int pxsom_assign_means_ex(const void *x_dev, int64_t n, int c, int64_t ldx, int dtype, const double *w_dev, int k,
                          int32_t *labels_dev, double *sums_dev, int64_t *counts_dev, double *means_dev, void *workspace_dev,
                          size_t workspace_bytes, int flags, void *stream);
int pxsom_assign_sums(const void *x_dev, int64_t n, int c, int64_t ldx, int dtype, const double *w_dev, int k,
                      int32_t *labels_dev, double *sums_dev, int64_t *counts_dev, void *workspace_dev,
                      size_t workspace_bytes, void *stream);

/* The same pass with the tables OVERWRITTEN (no clearing by the caller) and the per-cluster means formed in the same
 * final launch: means_dev [k, c] = sums / max(count, 1), or NULL.  What generate_som_avg_files needs from one process
 * (pixel_cluster_utils.py:369-404); a multi-rank job all-reduces sums / counts and divides itself. */
int pxsom_assign_means(const void *x_dev, int64_t n, int c, int64_t ldx, int dtype, const double *w_dev, int k,
                       int32_t *labels_dev, double *sums_dev, int64_t *counts_dev, double *means_dev, void *workspace_dev,
                       size_t workspace_bytes, void *stream);

/* ---- cell x pixel-cluster counts: the counting step of create_c2pc_data --------------------------
 * reference: cell_cluster_utils.py:128-141 (groupby(['label', pixel_cluster_col]).size() + pivot per
 * FOV).  hist[a_i * nb + b_i] += 1 for every i with 0 <= a_i < na and 0 <= b_i < nb (other pairs are
 * ignored); hist_dev [na, nb] int64 is accumulated into, the caller clears it.  Exact integer counts. */
int pxsom_pair_histogram(const int32_t *a_dev, const int32_t *b_dev, int64_t n, int64_t na, int nb,
                         int64_t *hist_dev, void *stream);

/* ---- exact online SOM training: replaces pyFlowSOM.som -------------------------------------
 * reference: cluster_helpers.py:106-109 (PixieSOMCluster.train_som), FlowSOM C_SOM semantics:
 * n*rlen strictly sequential steps; step t presents row order_dev[t]; Euclidean BMU (first
 * minimum); every node within Chebyshev grid distance <= threshold of the BMU moves by
 * alpha*(x - w); alpha and threshold decay linearly (a0->a1, r0->r1; threshold pinned to 0.5
 * once below 1).  All arithmetic binary64, one rounding per operation, in the oracle's order.
 *   w_dev [k=xdim*ydim, c] binary64, in: initial nodes, out: trained nodes
 *   order_dev [n*rlen] int64 presentation order (explicit input, never generated here) */
int pxsom_train_online(const void *x_dev, int64_t n, int c, int64_t ldx, int dtype, double *w_dev,
                       int xdim, int ydim, int rlen, double a0, double a1, double r0, double r1,
                       const int64_t *order_dev, void *stream);
/* The same with `flags` for the RECALLED details of the loop (pyFlowSOM 0.1.16 is absent from the build image: each
 * recollection is a named switch, oracle/pxsom_oracle.c ORC_V_*).  PXSOM_ONLINE_INT_ABS: the accumulator behind the
 * "stop at the start of a pass when change < 1" test adds C's integer abs() of each difference -- truncated towards zero
 * first, so every |x - w| < 1 counts as 0 -- instead of fabs() (oracle: ORC_V_INT_ABS).  Only observable with rlen >= 2. */
#define PXSOM_ONLINE_INT_ABS 1
int pxsom_train_online_ex(const void *x_dev, int64_t n, int c, int64_t ldx, int dtype, double *w_dev,
                          int xdim, int ydim, int rlen, double a0, double a1, double r0, double r1,
                          const int64_t *order_dev, int flags, void *stream);

/* ---- batch SOM training (throughput mode; no pyFlowSOM analogue) -----------------------------
 * One mini-batch step = pxsom_batch_accumulate [+ all-reduce of stats across ranks] + pxsom_batch_update.
 * stats_dev is [k*c sums | k counts], all binary64 (counts are exact integers), so the collective is a
 * single sum over one buffer.
 *   accumulate: zero stats; labels = BMU(x rows, w) (pxsom_assign; labels_dev [n] int32 scratch that also
 *               returns them); stats[b, :] += x_i, stats_count[b] += 1 for b = label_i - 1.
 *   update:     num[k] = sum_{b: cheb(k,b) <= thr} sums[b], den[k] = sum_{b: ...} counts[b]
 *               (summed in orc_batch_update's separable order: along y inside a grid row, then over the rows)
 *               den[k] > 0:  w[k] += (1 - (1-alpha)^den[k]) * (num[k] * (1/den[k]) - w[k])
 *               (the gain is formed as -expm1(den * log1p(-alpha)))
 *   update_prepare: the same update with sums/counts = the two halves of stats_dev; the same launch clears
 *               stats_next_dev, the buffer the next accumulate will fill (alternate two buffers; NULL or
 *               == stats_dev: cleared by a separate fill), so the next
 *               pxsom_batch_accumulate(..., PXSOM_ACC_PREPARED, ...) is a single launch for the
 *               register-resident shapes (the accumulating filter prepares the codebook itself); other
 *               shapes get workspace_dev prepared for the new codebook here (may be NULL otherwise).
 * Oracle of record: oracle/pxsom_oracle.c orc_cluster_sums / orc_batch_update. */
#define PXSOM_ACC_PREPARED 1 /* flags: stats_dev was cleared (and the workspace prepared) by update_prepare */
int pxsom_batch_accumulate(const void *x_dev, int64_t n, int c, int64_t ldx, int dtype,
                           const double *w_dev, int k, int32_t *labels_dev, double *stats_dev,
                           void *workspace_dev, size_t workspace_bytes, int flags, void *stream);
int pxsom_batch_update(double *w_dev, int xdim, int ydim, int c, const double *sums_dev,
                       const double *counts_dev, double thr, double alpha, void *stream);
int pxsom_batch_update_prepare(double *w_dev, int xdim, int ydim, int c, double *stats_dev,
                               double *stats_next_dev, double thr, double alpha, void *workspace_dev,
                               size_t workspace_bytes, void *stream);

/* ---- batch SOM training, one call per run of steps ------------------------------------------------
 * The host loop over the mini-batch steps of a pass, inside the library (no per-step Python / ctypes work).
 * Step g of total_steps (= num_passes * batch_steps) uses the rows i = (g % batch_steps) + r * batch_steps of
 * x_dev and the schedule thr/alpha(g) of orc_som_batch.  State, all caller-owned:
 *   wbuf_dev        [2][k*c] binary64: W_g, the codebook step g searches with, lives in wbuf[g % 2].
 *                   Before the first call the caller stores W_0 in wbuf[0].
 *   stats_ring_dev  [3][k*(c+1)] binary64: step g leaves its [k*c sums | k counts] in ring[g % 3] and clears
 *                   ring[(g+1) % 3]; the call with g_begin == 0 clears ring[0] first.
 * A single process runs [0, total_steps) in one call.  A multi-rank job runs ONE step per call and sum-all-reduces
 * ring[g % 3] across ranks before the next call (the only exchange of the rule).  After the last step (and its
 * all-reduce) pxsom_batch_train_finish applies the last pending update: w_out_dev [k, c] receives W_total.
 * Register-resident shapes (10 x 10 grid, even c <= 32, rows 2-element aligned) take ONE launch per step: the
 * pending update of step g-1 and the codebook preparation run at the head of step g's BMU search in every
 * workgroup; other shapes run update / prepare / search / exact / sums launches per step -- except steps of up to
 * 16 384 binary32 / binary64 rows on codebooks of up to 256 nodes x 128 channels that fit a CU's LDS, which take ONE
 * launch as well (csrc/pxsom_batch_step_wide.hip): the steps whose pending update has its threshold pinned at 0.5 on any
 * grid (a node's window is the node), the others -- up to 4 096 rows -- on grids up to 16 x 16.  Same results either way (PXSOM_TRAIN_UNFUSED forces
 * the launch-per-phase route for every step).  Oracle of record: oracle/pxsom_oracle.c orc_som_batch. */
#define PXSOM_TRAIN_UNFUSED 1
/* (flag value 2 was PXSOM_TRAIN_PERSISTENT_TAIL in ABI 6 - 8: an opt-in persistent launch for the BMU-only tail, measured slower
 * than the launches it replaced and removed in ABI 9; the bit is ignored.) */
size_t pxsom_batch_train_workspace_bytes(int64_t n, int batch_steps, int c, int k);
int pxsom_batch_train_steps(const void *x_dev, int64_t n, int c, int64_t ldx, int dtype, double *wbuf_dev,
                            double *stats_ring_dev, int xdim, int ydim, int batch_steps, int g_begin, int g_end,
                            int total_steps, double a0, double a1, double r0, double r1, void *workspace_dev,
                            size_t workspace_bytes, int flags, void *stream);
int pxsom_batch_train_finish(const double *wbuf_dev, const double *stats_ring_dev, int xdim, int ydim, int c,
                             int steps_done, int total_steps, double a0, double a1, double r0, double r1,
                             double *w_out_dev, void *stream);

/* ---- batch SOM training on a SCHEDULE (round 3): unequal mini-batch steps ----------------------------------
 * The rows are dealt into `phases` phases (row i: phase i % phases); step g of a pass takes the phases
 * [edges[g], edges[g+1]) (edges_host [steps_per_pass + 1], edges[0] == 0, edges[steps_per_pass] == phases,
 * non-decreasing; steps_per_pass <= PXSOM_MAX_SCHED_STEPS).  Its update is taken at the point of the online schedule
 * (threshold, alpha) where the rows presented before it end: (pass * phases + edges[g]) / (num_passes * phases).
 * Equal steps (edges = 0..phases) are pxsom_batch_train_steps, bit for bit.  A step is one latency-bound launch
 * whatever its size, so a pass is priced in steps: few large steps while the neighbourhood is wide, many small ones
 * in the BMU-only tail reach the quality of 64 equal steps in under half the launches (DESIGN.md "K6b").
 * State (wbuf_dev, stats_ring_dev), routes, flags and comm as pxsom_batch_train_steps[_sharded]; steps are numbered
 * over the whole run, g in [0, num_passes * steps_per_pass); the call with g_begin == 0 must come first on a workspace
 * (it clears ring[0] and, for shapes outside the fused kernel with steps wider than one phase whose kernels do not all read a
 * step's rows where they lie -- 32 channels or fewer, binary64 rows, rows not contiguous in x, a per-cluster table small enough
 * for the wave-private sums kernels --, gathers the rows into step order inside the workspace: one extra read + write of the
 * matrix per run; round 6: other shapes are read in place).  Every rank runs the same steps.
 * The WORKSPACE IS RUN STATE, like wbuf and the ring: the g_begin == 0 call also leaves the run's centring vector (the
 * mean of W_0 per channel and its norm, read by every later step's filter) in it, so the calls of one run must be given
 * the same, untouched workspace; a later call on a fresh or foreign workspace would centre on whatever bytes it finds
 * (results stay exact -- every label is settled against the binary64 codebook -- but most rows would be listed).
 * Oracle of record: oracle/pxsom_oracle.c orc_som_batch_sched.  Reference call replaced: cluster_helpers.py:98-116. */
/* Reproducible statistics for binary64 rows (sum_quantum > 0; ignored for binary32 / binary16 rows).  The per-BMU sums
 * of a step are floating-point additions in whatever order the workgroups and ranks deliver them; for binary64 rows that
 * order leaves 1e-16 noise which the degenerate first steps of a pass (near-identical nodes) can amplify into different
 * BMUs -- two runs on the same data then end in different codebooks, where the reference pins same-seed retraining
 * (tests/phenotyping/cluster_helpers_test.py:323-332 of the reference).  With sum_quantum = q (a power of two) every
 * value enters the statistics rounded to a multiple of q (round-half-even; the BMU search still sees the value itself);
 * q = pxsom_exact_sum_quantum(max |x| over the job's rows, most rows any step holds over all ranks) makes every partial
 * sum exactly representable, hence every addition exact, hence the statistics -- and the whole run -- independent of
 * order, workgroup count and rank count.  The rounding is part of the rule: orc_som_batch_sched takes the same q.
 * pxsom_absmax: max |x| over the finite entries of a matrix into out_dev[0] (0 when there is none). */
double pxsom_exact_sum_quantum(double value_bound, int64_t rows_bound);
int pxsom_absmax(const void *x_dev, int64_t n, int c, int64_t ldx, int dtype, double *out_dev, void *stream);

#define PXSOM_MAX_SCHED_STEPS 256
typedef struct pxsom_comm pxsom_comm; /* the library-owned RCCL communicator, below */
size_t pxsom_batch_train_sched_workspace_bytes(int64_t n, int c, int k, int dtype, int phases, const int32_t *edges_host,
                                               int steps_per_pass);
int pxsom_batch_train_sched(const void *x_dev, int64_t n, int c, int64_t ldx, int dtype, double *wbuf_dev,
                            double *stats_ring_dev, int xdim, int ydim, int phases, const int32_t *edges_host,
                            int steps_per_pass, int g_begin, int g_end, int num_passes, double a0, double a1, double r0,
                            double r1, double sum_quantum, void *workspace_dev, size_t workspace_bytes, int flags,
                            pxsom_comm *comm, void *stream);
/* The same with the run's first codebook W_0 handed over where the caller holds it (ABI 9): w0_dev [k, c] or NULL.  With g_begin
 * == 0 the launch that prepares the run copies it into wbuf_dev[0] itself -- no copy launch in front of a pass of 22 short steps;
 * w0_dev may be the buffer pxsom_batch_train_sched_finish later writes into.  Ignored when g_begin > 0. */
int pxsom_batch_train_sched_from(const void *x_dev, int64_t n, int c, int64_t ldx, int dtype, const double *w0_dev, double *wbuf_dev,
                                 double *stats_ring_dev, int xdim, int ydim, int phases, const int32_t *edges_host,
                                 int steps_per_pass, int g_begin, int g_end, int num_passes, double a0, double a1, double r0,
                                 double r1, double sum_quantum, void *workspace_dev, size_t workspace_bytes, int flags,
                                 pxsom_comm *comm, void *stream);
int pxsom_batch_train_sched_finish(const double *wbuf_dev, const double *stats_ring_dev, int xdim, int ydim, int c,
                                   int phases, const int32_t *edges_host, int steps_per_pass, int steps_done,
                                   int num_passes, double a0, double a1, double r0, double r1, double *w_out_dev,
                                   void *stream);
/* 1 when the steps of this matrix / shape take the one-launch fused kernel, 0 for the launch-per-phase route.  The
 * ranks of a job agree on the route before the run (MIN over ranks, PXSOM_TRAIN_UNFUSED for all otherwise): the two
 * routes round the codebook's last bits differently, and every rank must hold the same codebook. */
int pxsom_batch_train_fused_route(const void *x_dev, int c, int64_t ldx, int dtype, int xdim, int ydim, int phases);

/* ---- the exchange of a multi-rank batch run, inside the library ---------------------------------------
 * One process per GPU, rows sharded by rank: the rule's only exchange is the sum of ring[g % 3] over the ranks
 * after every step.  pxsom_batch_train_steps_sharded is pxsom_batch_train_steps with that all-reduce (RCCL,
 * in place, binary64) enqueued on `stream` right behind each step's launch, so a whole run of steps is one call
 * here as well -- no host work between a step and its exchange.  comm == NULL: no exchange (single rank).
 * RCCL is bound at run time: pxsom_comm_bind(path of the librccl.so this process uses; NULL = "librccl.so.1")
 * once per process; rank 0 draws the id with pxsom_comm_unique_id and hands it to the other ranks by whatever
 * channel launched them (the Python host uses the torch.distributed store); every rank then calls
 * pxsom_comm_create -- collective -- with its HIP device current.  Every rank must run the same steps.
 * The reference has no analogue (single-core training, cluster_helpers.py:106-109). */
#define PXSOM_COMM_ID_BYTES 128
int pxsom_comm_bind(const char *librccl_path);
int pxsom_comm_unique_id(void *id_out, size_t id_bytes);
int pxsom_comm_create(const void *id, size_t id_bytes, int nranks, int rank, pxsom_comm **out);
int pxsom_comm_destroy(pxsom_comm *comm);
int pxsom_comm_allreduce_sum_f64(pxsom_comm *comm, double *buf_dev, size_t count, void *stream);
/* One-shot peer-to-peer exchange (round 4; csrc/pxsom_comm.hip): every rank owns an exchange block in its device memory,
 * mapped by the others through HIP IPC; an exchange is ONE launch per rank (write the own contribution into every block,
 * flag, wait for the others' flags, add the slots in rank order: bit-identical sums on all ranks).  Ranks may share a
 * device (two processes on one GPU: how a one-GPU box validates the protocol) or sit on the devices of one node.
 *   create(nranks <= 16, rank, max_count binary64 values per exchange) -> handle(64 bytes; gather them from all ranks,
 *   rank order) -> connect(all handles).  The communicator then serves pxsom_comm_allreduce_sum_f64 and
 *   pxsom_batch_train_sched like an RCCL one.  pxsom_comm_p2p_error: 0, or the epoch at which a peer failed to arrive
 *   within 4 s (that exchange's buffer was set to NaN; the GPU is not left hanging; the FIRST such epoch stays on record).
 * pxsom_comm_p2p_set_fused(comm, 1) (ABI 9; EVERY rank, same value): pxsom_batch_train_sched on such a communicator runs the
 * exchange INSIDE the launches of the fused 10 x 10 step (the last workgroup of a step writes this rank's statistics into every
 * block, the next step adds the slots in rank order while it applies the update: the same bits as the one-launch exchange, one
 * launch per step); a peer that is late there turns the codebook to NaN, sets the same error word, and this rank still raises
 * its flags (over NaN slots) so that nobody waits for it.  The last step of a call and every other shape keep the one-launch
 * exchange.  The switch is a property of the communicator (no process-wide state). */
#define PXSOM_P2P_HANDLE_BYTES 64
int pxsom_comm_p2p_create(int nranks, int rank, size_t max_count, pxsom_comm **out);
int pxsom_comm_p2p_handle(pxsom_comm *comm, void *handle_out, size_t handle_bytes);
int pxsom_comm_p2p_connect(pxsom_comm *comm, const void *handles, size_t handles_bytes);
int pxsom_comm_p2p_error(pxsom_comm *comm, unsigned long long *epoch_out);
int pxsom_comm_p2p_set_fused(pxsom_comm *comm, int on);
/* pxsom_batch_train_steps with the exchange enqueued behind every step (comm: RCCL or peer-to-peer; NULL = none).  Equal
 * steps only: 1 <= batch_steps <= PXSOM_MAX_SCHED_STEPS (256) and total_steps a whole number of passes (total_steps %
 * batch_steps == 0) -- other values are PXSOM_ERR_INVALID_ARG since ABI 7; schedules: pxsom_batch_train_sched. */
int pxsom_batch_train_steps_sharded(const void *x_dev, int64_t n, int c, int64_t ldx, int dtype, double *wbuf_dev,
                                    double *stats_ring_dev, int xdim, int ydim, int batch_steps, int g_begin,
                                    int g_end, int total_steps, double a0, double a1, double r0, double r1,
                                    void *workspace_dev, size_t workspace_bytes, int flags, pxsom_comm *comm,
                                    void *stream);

/* ---- pre-processing (create_fov_pixel_data and the 99.9 % values) -----------------------------
 * reference: pixie_preprocessing.py:47-49 -> scipy.ndimage.gaussian_filter(plane, sigma) per channel:
 * separable, 'reflect' boundary, axis 0 then axis 1, binary64, scipy's symmetric-kernel summation
 * order.  weights_host [2*radius+1] is the normalised kernel exactly as scipy builds it
 * (numpy exp / sum on the host: part of the reference numerics), radius = int(4*sigma + 0.5).
 * img_dev [h, w, c] binary64 interleaved, blurred in place; tmp_dev: same-size scratch. */
#define PXSOM_BLUR_GENERIC_FORM 2 /* OR-ed into f32_semantics: the thread-per-output kernel for every shape (the pipeline's
                                    radius-8 blur otherwise runs a register-window pass down the rows and an LDS-tiled pass
                                    along the columns; same results bit for bit -- the tests compare the two) */
int pxsom_gaussian_blur_hwc(double *img_dev, double *tmp_dev, int h, int w, int c,
                            const double *weights_host, int radius, int f32_semantics, void *stream);
/* f32_semantics (here and below): the matrix holds float32 values widened to binary64 -- what the
 * pipeline feeds create_fov_pixel_data for float32 TIFFs (pixie_preprocessing.py:152-159).  scipy then
 * stores every blur pass as float32 and pandas sums / divides the float32 frame in binary32; with the flag
 * the kernels round in exactly those places, so the results are the reference's float32 values (widened). */

/* reference: pixie_preprocessing.py:67-75 + pixel_cluster_utils.normalize_rows (:126-130):
 * keep pixel i iff rowsum_i > thresh and any(x_ij != 0); out row = x_i / rowsum_i (left-to-right
 * binary64 row sum, as pandas computes it).  Kept rows are compacted in pixel order.
 *   out_rows_dev [<= n, c] binary64, out_index_dev [<= n] int64 flat pixel index of each kept row,
 *   out_count_dev [1] int64.  workspace from pxsom_rownorm_workspace_bytes(n). */
size_t pxsom_rownorm_workspace_bytes(int64_t n);
int pxsom_rowsum_filter_normalize(const double *x_dev, int64_t n, int c, double thresh,
                                  double *out_rows_dev, int64_t *out_index_dev,
                                  int64_t *out_count_dev, void *workspace_dev,
                                  size_t workspace_bytes, int f32_semantics, void *stream);

/* reference: PixelSOMCluster.normalize_data (cluster_helpers.py:242-246): x[:, j] / norm[j] in
 * binary64 (in place allowed: out_dev may equal x_dev). */
int pxsom_normalize_columns(const double *x_dev, int64_t n, int c, int64_t ldx,
                            const double *norm_dev, double *out_dev, int64_t ldo, void *stream);

/* reference: df.replace(0, nan).quantile(q) per column (pixie_preprocessing.py:406-408,
 * cluster_helpers.py:366) / np.quantile(img[img > 0], q) (pixel_cluster_utils.py:47-51):
 * type-7 (linear) quantile of the kept values (keep_mode 0: != 0 and not NaN, 1: > 0, 2: not NaN) of each
 * column.
 * out_dev [c] binary64 (NaN for a column with no kept value).  Exact: MSB-first radix select on the
 * binary64 bit patterns, then numpy's interpolation formula.  Note pandas' effective q is (q*100)/100. */
size_t pxsom_quantile_workspace_bytes(int64_t n, int c);
int pxsom_quantile_nonzero(const double *x_dev, int64_t n, int c, int64_t ldx, double q,
                           int keep_mode, double *out_dev, void *workspace_dev,
                           size_t workspace_bytes, void *stream);

/* reference: np.quantile(img[img > 0], percentile) on a float32 image (calculate_channel_percentiles,
 * pixel_cluster_utils.py:41-51) and np.quantile(summed_data, 0.05) (calculate_pixel_intensity_percentile,
 * :96-103).  Same radix select on float32 columns; index, fraction and interpolation in binary32 exactly as
 * numpy forms them for a float32 array.  keep_mode 2 keeps every non-NaN value.  out_dev [c] binary64
 * holding the float32 results. */
int pxsom_quantile_f32(const float *x_dev, int64_t n, int c, int64_t ldx, double q, int keep_mode,
                       double *out_dev, void *workspace_dev, size_t workspace_bytes, void *stream);

/* reference: np.sum(img_data / norm_vect, axis=-1) of calculate_pixel_intensity_percentile
 * (pixel_cluster_utils.py:96-101): per pixel the float32 sum over channels of img/norm, added in numpy's
 * order for a contiguous float32 axis (c <= 128).  img_dev [n, c] float32, out_dev [n] float32. */
int pxsom_scaled_rowsum_f32(const float *img_dev, int64_t n, int c, int64_t ldx, const float *norm_dev,
                            float *out_dev, void *stream);
/* The same sum in binary64: what numpy computes when the image is not float32 (uint16 / int16 exports: the
 * division by the binary64 channel percentiles promotes to float64) -- same summation order. */
int pxsom_scaled_rowsum_f64(const double *img_dev, int64_t n, int c, int64_t ldx, const double *norm_dev,
                            double *out_dev, void *stream);

/* ---- pixel cluster mask: the relabel + scatter of generate_pixel_cluster_mask ------------------------
 * reference: utils/data_utils.py:532-553 -- coordinates = row_index * W + column_index;
 * cluster_labels = [id_mapping[label] for label in ...]; img.ravel()[coordinates] = cluster_labels on an
 * int16 image of zeros.  lut_dev [lut_size] int32 holds id_mapping densely (PXSOM_LUT_UNMAPPED where the
 * mapping has no entry).  mask_dev [h, w] int16 is overwritten (0 where the table has no pixel; ids are
 * narrowed to int16 the way numpy's assignment does).  A pixel listed more than once keeps the id of its
 * LAST row, like the sequential numpy assignment.  status_dev [1] int32 comes back 0, or with
 * PXSOM_MASK_BAD_LABEL (a label outside the LUT or unmapped: the reference raises KeyError) and / or
 * PXSOM_MASK_BAD_PIXEL (a coordinate outside the image: IndexError) set; the mask is then unspecified.
 * workspace: pxsom_cluster_mask_workspace_bytes(h, w) (one int64 per pixel: index of the winning row). */
#define PXSOM_LUT_UNMAPPED (-2147483647 - 1)
#define PXSOM_MASK_BAD_LABEL 1
#define PXSOM_MASK_BAD_PIXEL 2
size_t pxsom_cluster_mask_workspace_bytes(int h, int w);
int pxsom_cluster_mask(const int64_t *row_index_dev, const int64_t *column_index_dev, const int64_t *labels_dev,
                       int64_t n, const int32_t *lut_dev, int64_t lut_size, int h, int w, int16_t *mask_dev,
                       int32_t *status_dev, void *workspace_dev, size_t workspace_bytes, void *stream);

/* ---- SOM cluster -> meta cluster: the per-pixel half of pixel_consensus_cluster -----------------------------
 * reference: cluster_helpers.py:669-682 (PixieConsensusCluster.assign_consensus_labels: Series.map through the
 * K-row mapping), applied per FOV table at pixel_meta_clustering.py:17-50.  out_dev[i] = lut_dev[labels_dev[i]] for
 * labels inside [0, lut_size), `fill` otherwise (what the map turns into NaN); in place allowed. */
int pxsom_relabel(const int32_t *labels_dev, int64_t n, const int32_t *lut_dev, int lut_size, int32_t fill,
                  int32_t *out_dev, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PXSOM_H */
