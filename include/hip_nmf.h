/*
 * hip_nmf.h -- C ABI of libhip_nmf.so, the MI355X (gfx950) NMF multiplicative-update engine.
 *
 * Drop-in boundary for the one hot path of elvis-sik/muscle_synergies:
 *
 *     model = NMF(n_components=n_components, **sklearn_kwargs)      (src/muscle_synergies/analysis.py:862)
 *     transformed_signal = model.fit_transform(matrix)              (src/muscle_synergies/analysis.py:863)
 *
 * with solver='mu', beta_loss='frobenius'.  The arithmetic replaced is scikit-learn's
 * _fit_multiplicative_update (sklearn/decomposition/_nmf.py:731-893), _multiplicative_update_w (:526-631),
 * _multiplicative_update_h (:634-728) and _beta_divergence (:85-134).  Notation is sklearn's:
 * X (T x m) ~= W (T x k) * H (k x m), one muscle per column of X; W is updated first, then H.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no exceptions, no torch/STL types.
 *  - Every array pointer is a DEVICE pointer valid on the handle's device (the Python host passes
 *    torch.Tensor.data_ptr(); torch is only the allocator / H2D copier).
 *  - The caller owns every buffer; the library keeps no caller pointer after a call returns.
 *    Scratch memory lives inside the handle (grow-only, freed by hipnmf_destroy).
 *  - A handle is bound to one device + one HIP stream and is not re-entrant; distinct handles
 *    may be driven concurrently from different host threads.
 *  - Calls return after the stream has been synchronised (results are ready on return).
 *  - Return value: 0 = ok, <0 = error class below; text via hipnmf_last_error() (thread-local).
 *  - There is NO CPU fallback: without a usable GPU every compute entry point returns HIPNMF_ERR_NO_DEVICE.
 */
#ifndef HIP_NMF_H
#define HIP_NMF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HIPNMF_VERSION 212 /* 0.2.3: round 6 added hipnmf_set_batch_hint, hipnmf_routes_describe, hipnmf_resample_weights_*; 0.2.1: round 5 added HIPNMF_W_ROW_MAJOR_PAD16 (the shard entry points no longer read plain W_ROW_MAJOR as padded); 0.2.0: round 3 grew hipnmf_envelope_params and added hipnmf_rank_sweep_stop_*, round 4 added hipnmf_sosfilt_params.mode */

#define HIPNMF_OK 0
#define HIPNMF_ERR_BAD_ARG (-1)
#define HIPNMF_ERR_HIP (-2)
#define HIPNMF_ERR_UNSUPPORTED (-3) /* shape outside the compiled kernel set (m <= 512, k <= 64) or a combination it does not hold */
#define HIPNMF_ERR_NO_DEVICE (-4)

/* memory layout of one X matrix.  Either is accepted everywhere; which one the kernels stream WITHOUT a one-off
 * conversion depends on the shape: fp32 with 7..8 or 9..16 channels (k <= 8 resp. k <= 5) and fp64 with 7..8 channels
 * (k <= 4) read ROW_MAJOR in place when n_features is exactly 8 resp. 16, ldx % 4 == 0 and X is 16-byte aligned; every other shape (and the shard
 * entry points) reads CHANNEL_MAJOR in place when ldx % 4 == 0 and X is 16-byte aligned.  The ragged entry points take
 * channel-major packed matrices and convert each distinct one once per fit where a row-major kernel is used. */
#define HIPNMF_X_ROW_MAJOR 0     /* X[t*ldx + j]  (T x m, C order; ldx >= m)                          */
#define HIPNMF_X_CHANNEL_MAJOR 1 /* X[j*ldx + t]  (m x T; = DataFrame.to_numpy() F order; ldx >= T)   */

/* memory layout of one W matrix (ldw / batch stride are implied: contiguous) */
#define HIPNMF_W_ROW_MAJOR 0       /* W[t*k + c]  (T x k, C order: what sklearn returns)               */
#define HIPNMF_W_COMPONENT_MAJOR 1 /* W[c*T + t]  (k x T: the engine's native streaming layout)        */
#define HIPNMF_W_ROW_MAJOR_PAD16 2 /* W[t*KP + c], KP = n_components rounded up to a multiple of 16, columns >= n_components zero:
                                    * the general-shape kernels' layout; hipnmf_shard_* / hipnmf_fit_tsharded_* only (version 210) */

/* objective (sklearn's beta_loss, _nmf.py:1397-1401) */
#define HIPNMF_LOSS_FROBENIUS 0 /* 'frobenius' (beta = 2): the reference's default, every entry point          */
#define HIPNMF_LOSS_KL 1        /* 'kullback-leibler' (beta = 1): hipnmf_fit_batched_* / hipnmf_fit_ragged_*;  */
                                /* err_out = sqrt(2 * KL(X || WH)) as reconstruction_err_ (_nmf.py:185-189)    */

typedef struct hipnmf_handle hipnmf_handle;

/* Problem description shared by every compute entry point (POD, passed by pointer). */
typedef struct hipnmf_problem {
  int32_t struct_size;    /* = sizeof(hipnmf_problem), ABI guard                                       */
  int32_t batch;          /* B  >= 1 independent factorisations                                        */
  int64_t n_samples;      /* T  rows of X (time samples)                                               */
  int32_t n_features;     /* m  columns of X (muscles), 1..512                                         */
  int32_t n_components;   /* k  rank, 1..64                                                            */
  int32_t x_layout;       /* HIPNMF_X_*                                                                */
  int32_t update_h;       /* 1: fit (W and H updated, _nmf.py:854); 0: transform (H fixed, :1736-1763) */
  int32_t w_layout;       /* HIPNMF_W_*                                                                */
  int32_t loss;           /* HIPNMF_LOSS_*                                                             */
  int64_t ldx;            /* leading dimension of one X matrix, in elements                            */
  int64_t x_batch_stride; /* elements between consecutive X matrices                                   */
  int32_t max_iter;       /* >= 1 (NMF max_iter, _nmf.py:831)                                          */
  int32_t check_every;    /* convergence test period; sklearn hard-codes 10 (_nmf.py:872)              */
  double tol;             /* 0 disables the test (_nmf.py:872); else stop when                         */
                          /* (prev_err - err) / err_at_init < tol (_nmf.py:883)                        */
  double l1_reg_W, l1_reg_H, l2_reg_W, l2_reg_H; /* already scaled as _compute_regularization (:1254)  */
} hipnmf_problem;

/* ---- library / device ---------------------------------------------------------------------------- */
int hipnmf_version(void);
const char* hipnmf_last_error(void);
int hipnmf_device_count(void);                            /* <0 on error (no ROCm device / driver)     */
int hipnmf_create(int device, hipnmf_handle** out);       /* own stream + workspace on `device`        */
/* Waits for the handle's own stream, frees its workspaces, events and stream.  Must not race with a call that uses `h`.
 * Process exit: the first hipnmf_create registers an exit handler (it runs BEFORE the HIP runtime's own, registered earlier);
 * a hipnmf_destroy that arrives after it -- a host freeing objects from static destructors or a garbage collector at interpreter
 * teardown -- releases the host side only and returns 0 without calling into the runtime; so does one that finds the runtime
 * deinitialised.  (The Python host closes every handle it still caches from an `atexit` hook: muscle_synergies_amd/_lib.py.) */
int hipnmf_destroy(hipnmf_handle* h);
#define HIPNMF_STREAM_NULL ((void*)1)                    /* the device's default (null) HIP stream    */
int hipnmf_set_stream(hipnmf_handle* h, void* hip_stream);/* NULL restores the handle's own stream     */
/* upper-bound estimate of the device workspace a fit of *p makes the handle allocate (informational: the library sizes and
 * grows its workspace itself) */
size_t hipnmf_workspace_bytes(const hipnmf_problem* p, int elem_size /* 4 or 8 */);
/* Device time (HIP events on the handle's stream) of the solver kernels of the last compute call. */
int hipnmf_last_kernel_ms(hipnmf_handle* h, float* ms);
/* Instance name of the solver kernel the last fit on this handle launched, e.g. "fit_rowlane_kernel<5,1,5,1>" or
 * "fit_coop_kernel<float,1,16,5>" (as rocprofv3 --kernel-trace prints it, minus the namespace); "" before the first
 * fit.  The string lives in the handle.  Lets a benchmark name the kernel it timed instead of guessing. */
const char* hipnmf_last_kernel(hipnmf_handle* h);
/* 1: compute entry points return right after enqueueing their kernels on the handle's stream (no host
 * synchronisation; hipnmf_last_kernel_ms is then unavailable).  Used with hipnmf_set_stream(torch's current
 * stream) by the time-sharded solver so that kernels and RCCL collectives are ordered by the stream alone.
 * Only the shard entry points honour it; hipnmf_fit_batched_* always returns with results ready. */
int hipnmf_set_async(hipnmf_handle* h, int enable);
/* Tuning knobs (0 = library default): threads per workgroup (256 / 512 / 768 / 1024), max row slices per matrix, and the
 * solver path: 0 chosen by the library, 1 one persistent workgroup per matrix, 2 row-sliced launches,
 * 3 cooperative multi-workgroup kernel (few long matrices; HIPNMF_ERR_UNSUPPORTED when not applicable),
 * 4 / 5 = 1 with the kernel instance pinned: 4 fit_persistent_kernel (VALU contractions), 5 fit_rowlane_kernel
 * (X H^T and W H H^T on the f32 matrix pipe; fp32, 9..16 channels, else HIPNMF_ERR_UNSUPPORTED),
 * 6 fit_small_kernel (one wave per matrix, n_samples <= 256 -- up to 512 / 768 / 1 024 for the shapes compiled with more tiles
 * in registers, which batches of at least two (float64: three) matrices per CU take by themselves --; HIPNMF_ERR_UNSUPPORTED otherwise).
 * Wide shapes (n_features > 32 or n_components > 8: fit_wide_kernel, every contraction on v_mfma_*_16x16x4; at most 8
 * components with the Frobenius loss: fit_wide4_kernel / fit_wide4d_kernel on v_mfma_f32_4x4x1 / v_mfma_f64_4x4x4; the
 * library also routes narrower shapes there where it measured faster -- DISPATCH.md): variants 0, 1 (= 4) and 2 exist -- 2 = rows sliced over
 * the chip, Frobenius loss and uniform batches only --, threads = 256 / 512 / 768 pins the instance (two workgroups per
 * CU / one with the larger W cache / three waves per SIMD, fit_wide4_kernel up to 64 channels only); 3, 5, 6 answer
 * HIPNMF_ERR_UNSUPPORTED.  Kullback-Leibler loss: a few long matrices (at most one per CU, where the library's cost model says so)
 * run row-sliced on the one-pass general-shape kernel whatever their width; variant 1 or max_slices = 1 keeps one workgroup per matrix. */
int hipnmf_set_tuning(hipnmf_handle* h, int threads, int max_slices, int variant);
/* A host that hands ONE batch to the library in several calls (chunks of a transfer pipeline, a compacted sub-batch) names the
 * size of the whole batch here: until reset with 0, every batched fit on this handle chooses its kernel family, workgroup
 * geometry and row-slice count as for a batch of max(p->batch, batch) matrices, so that a tail chunk of 52 matrices is fitted
 * by the same kernel -- bit for bit the same per-matrix arithmetic -- as the 256-matrix chunks before it (the choice between
 * the lane mappings and the matrix-pipe kernels, and between one workgroup per matrix and row slices, depends on the batch).
 * The native rank sweep uses the same mechanism internally.  No reference counterpart: sklearn's fit is one matrix per call
 * (src/muscle_synergies/analysis.py:862-863). */
int hipnmf_set_batch_hint(hipnmf_handle* h, int batch);
/* The fitted routing constants in force, as "name=value,name=value" (static string): the defaults of struct hipnmf_route_table
 * (csrc/hipnmf_internal.hpp: every measured threshold of the dispatchers in one table) with the environment variable
 * HIPNMF_ROUTES -- same syntax, read once per process -- applied.  tools/calibrate_routes.py re-derives the crossovers on the box
 * at hand and prints such a string.  Diagnostics only; no reference counterpart. */
const char* hipnmf_routes_describe(void);

/* ---- batched fit: replaces NMF(solver='mu').fit_transform / .transform for B matrices ------------- */
/*
 * X        [B] matrices, layout per p->x_layout / ldx / x_batch_stride             (in)
 * W        [B] matrices per p->w_layout, W0 on entry (init='custom', _nmf.py:1198-1208), result on return
 * H        [B][k][m] row-major, H0 on entry, components_ on return (unchanged when update_h == 0)
 * err_out  [B]    reconstruction_err_ = ||X - W H||_F after the loop (_nmf.py:1628-1630), or NULL
 * n_iter_out [B]  n_iter_ (_nmf.py:893), or NULL
 * sse_col_out [B][m]  per-column sum((X - W H)^2)  -> VAF numerators (analysis.py:660-662), or NULL
 * xsq_col_out [B][m]  per-column sum(X^2)          -> VAF denominators (analysis.py:654-656), or NULL
 *
 * Shapes: any 1 <= n_components <= 64, n_features <= 512 (the reference accepts any n <= m, analysis.py:829-846); beyond
 * 128 channels / 32 components, and for float64 with more than 16 components on more than 64 channels, the general-shape kernels
 * run (both losses; a ragged batch is fitted trial by trial there).  Layouts used in place (anything
 * else costs one conversion per fit): fp32 16-channel C-order X for the narrow row-per-lane instances, channel-major X
 * for the other narrow ones; for the wide shapes a C-order X whose rows are a whole number of 16-byte pieces
 * (n_features * sizeof % 16 == 0, ldx likewise) and a C-order W with n_components % 4 == 0.
 */
int hipnmf_fit_batched_f32(hipnmf_handle* h, const hipnmf_problem* p, const float* X, float* W, float* H,
                           float* err_out, int32_t* n_iter_out, float* sse_col_out, float* xsq_col_out);
int hipnmf_fit_batched_f64(hipnmf_handle* h, const hipnmf_problem* p, const double* X, double* W, double* H,
                           double* err_out, int32_t* n_iter_out, double* sse_col_out, double* xsq_col_out);

/* ---- ragged batch: trials of unequal length --------------------------------- */
/*
 * Same solver, but matrix b has its own number of rows.  X and W are packed in the engine-native layouts:
 * matrix b's X is channel-major [m][ld_b] starting at element x_off_b of X, its W component-major [k][ld_b]
 * starting at element w_off_b of W (ld_b >= T_b, ld_b % 4 == 0, x_off_b % 4 == 0, padding rows zero).
 * desc: HOST array [B][4] = {T_b, x_off_b, ld_b, w_off_b}.  p->n_samples = max_b T_b; p->x_layout must be
 * HIPNMF_X_CHANNEL_MAJOR and p->w_layout HIPNMF_W_COMPONENT_MAJOR; ldx / x_batch_stride are ignored.
 * H, err_out, n_iter_out, sse_col_out, xsq_col_out as in hipnmf_fit_batched_*.
 */
int hipnmf_fit_ragged_f32(hipnmf_handle* h, const hipnmf_problem* p, const int64_t* desc, const float* X, float* W,
                          float* H, float* err_out, int32_t* n_iter_out, float* sse_col_out, float* xsq_col_out);
int hipnmf_fit_ragged_f64(hipnmf_handle* h, const hipnmf_problem* p, const int64_t* desc, const double* X, double* W,
                          double* H, double* err_out, int32_t* n_iter_out, double* sse_col_out, double* xsq_col_out);

/* ---- time-sharded building blocks (one rank holds rows [t0, t1) of every X and W; H replicated) --- */
/*
 * One iteration of the sharded solver is
 *     hipnmf_shard_pass   : W_s <- W_s * (X_s H^T) / (W_s H H^T);  sums <- [W_s^T X_s | W_s^T W_s]
 *     all-reduce(sums)    : by the caller (RCCL over xGMI via torch.distributed; k*m + k*k floats)
 *     hipnmf_shard_hupdate: H <- H * (W^T X) / ((W^T W) H)
 * and hipnmf_shard_residual returns the shard's per-column SSE (and sum X^2) for the stop rule / VAF.
 * sums: [B][k*m + k*k] device buffer; sse_col/xsq_col: [B][m].  max_iter/tol in *p are ignored here.
 * Up to 32 channels and 8 components the shard entry points require p->w_layout == HIPNMF_W_COMPONENT_MAJOR and channel-major X
 * with ldx % 4 == 0 (no per-call layout conversion on the per-iteration path).  Beyond (round 4: up to 512 x 64, Frobenius) they
 * run on the general-shape kernels and require p->x_layout == HIPNMF_X_ROW_MAJOR with 16-byte aligned rows (ldx * sizeof % 16
 * == 0; the padding columns n_features..ldx-1 of X MUST be zero: they are read) and p->w_layout == HIPNMF_W_ROW_MAJOR_PAD16: W stored
 * as [n_samples][KP], KP = n_components rounded up to 16, the padding columns zero (they stay zero); sums keeps the
 * [k*m + k*k] layout: W^T X (k x m) then W^T W (k x k).  HIPNMF_W_ROW_MAJOR_PAD16 selects the general-shape kernels for narrow
 * shapes too; plain HIPNMF_W_ROW_MAJOR ([n_samples][n_components]) is HIPNMF_ERR_UNSUPPORTED on every shard entry point.
 * HIPNMF_LOSS_KL (round 4): always the general-shape kernels and their layouts, whatever the shape; sums = [W^T (X / WH) (k x m) |
 * colsum(W) in column 0 of the k x k block]; hipnmf_shard_residual returns the generalised Kullback-Leibler divergence per
 * column in sse_col (reconstruction_err_ = sqrt(2 * the sum over shards and columns)) -- call it with loss = FROBENIUS for the
 * squared-error columns of VAF; hipnmf_fit_tsharded_* does both.
 */
int hipnmf_shard_pass_f32(hipnmf_handle* h, const hipnmf_problem* p, const float* X, float* W, const float* H,
                          float* sums);
int hipnmf_shard_hupdate_f32(hipnmf_handle* h, const hipnmf_problem* p, float* H, const float* sums);
int hipnmf_shard_residual_f32(hipnmf_handle* h, const hipnmf_problem* p, const float* X, const float* W,
                              const float* H, float* sse_col, float* xsq_col);
int hipnmf_shard_pass_f64(hipnmf_handle* h, const hipnmf_problem* p, const double* X, double* W, const double* H,
                          double* sums);
int hipnmf_shard_hupdate_f64(hipnmf_handle* h, const hipnmf_problem* p, double* H, const double* sums);
int hipnmf_shard_residual_f64(hipnmf_handle* h, const hipnmf_problem* p, const double* X, const double* W,
                              const double* H, double* sse_col, double* xsq_col);

/*
 * The whole time-sharded fit on this rank's rows in one call (SURVEY.md section 8b's hipnmf_fit_tsharded; the loop
 * muscle_synergies_amd/tsharded.py drives from Python, for hosts that are not Python).  *p describes the LOCAL shard
 * (n_samples = its rows; same layout requirements as the building blocks); max_iter, tol, check_every and update_h are
 * honoured as in hipnmf_fit_batched (sklearn's stop rule on the GLOBAL residual, _nmf.py:872-884; with batch > 1 every
 * matrix must have converged).  `allreduce` sums `count` elements of `elem_size` bytes IN PLACE in a device buffer
 * over all ranks; it is called once per iteration with k*m + k*k elements per matrix and once per stop-rule check /
 * at the end with 2*m, always after the producing kernels have been enqueued on `hip_stream`: either enqueue the
 * collective on that stream (RCCL: ncclAllReduce(buf, buf, count, type, ncclSum, comm, (hipStream_t)hip_stream)) or
 * synchronise it, reduce, and return.  0 = success.  NULL = single rank.  err_out / n_iter_out [B], sse_col_out /
 * xsq_col_out [B][m] (global sums, may be NULL) are device buffers.
 */
typedef int (*hipnmf_allreduce_fn)(void* device_buf, size_t count, int elem_size, void* hip_stream, void* user);
int hipnmf_fit_tsharded_f32(hipnmf_handle* h, const hipnmf_problem* p, const float* X, float* W, float* H,
                            hipnmf_allreduce_fn allreduce, void* user, float* err_out, int32_t* n_iter_out,
                            float* sse_col_out, float* xsq_col_out);
int hipnmf_fit_tsharded_f64(hipnmf_handle* h, const hipnmf_problem* p, const double* X, double* W, double* H,
                            hipnmf_allreduce_fn allreduce, void* user, double* err_out, int32_t* n_iter_out,
                            double* sse_col_out, double* xsq_col_out);

/* ---- rank sweep: find_synergies(df, k_min, k_max) for a batch of trials (BASELINE.json config #4) -- */
/*
 * init='random' of sklearn (_nmf.py:303-314) drawn on the device: avg = sqrt(mean(X) / k), H0 = avg |N(0,1)|,
 * W0 = avg |N(0,1)| from a counter-based generator (Philox4x32-10 + Box-Muller) keyed by (seed, first_matrix + b,
 * element), so the values depend neither on the layout nor on how a batch is scattered over GPUs.  Not the stream of
 * numpy's RandomState: same distribution, other numbers.  W per p->w_layout, H [B][k][m].
 */
int hipnmf_random_init_f32(hipnmf_handle* h, const hipnmf_problem* p, uint64_t seed, int32_t first_matrix, const float* X,
                           float* W, float* H);
int hipnmf_random_init_f64(hipnmf_handle* h, const hipnmf_problem* p, uint64_t seed, int32_t first_matrix, const double* X,
                           double* W, double* H);
/*
 * For every k in [k_min, k_max]: hipnmf_random_init(seed + k) -> hipnmf_fit_batched (p->max_iter, tol, ... honoured;
 * p->n_components ignored) -> VAF over all muscles per trial (analysis.py:654-662).  The reference computes every rank
 * and leaves the choice to the user (analysis.py:753-756); selected_out is the smallest k with VAF >= vaf_threshold
 * (-1: none).  W_ws: workspace of B * T * k_max elements (holds W of rank k_max on return, per p->w_layout);
 * H_out: the components of every rank, rank after rank: [B][k][m] for k = k_min .. k_max;
 * vaf_out / err_out / n_iter_out: [B][k_max - k_min + 1]; selected_out [B].  All device pointers; err_out, n_iter_out
 * and selected_out may be NULL.
 */
int hipnmf_rank_sweep_f32(hipnmf_handle* h, const hipnmf_problem* p, int32_t k_min, int32_t k_max, double vaf_threshold,
                          uint64_t seed, int32_t first_matrix, const float* X, float* W_ws, float* H_out, float* vaf_out,
                          int32_t* selected_out, float* err_out, int32_t* n_iter_out);
int hipnmf_rank_sweep_f64(hipnmf_handle* h, const hipnmf_problem* p, int32_t k_min, int32_t k_max, double vaf_threshold,
                          uint64_t seed, int32_t first_matrix, const double* X, double* W_ws, double* H_out, double* vaf_out,
                          int32_t* selected_out, double* err_out, int32_t* n_iter_out);
/*
 * The same sweep with the "stop" of BASELINE.json config #4 ("k = 2..8 with VAF >= 0.90 stop", Rabbi et al. 2020): a
 * trial whose VAF has reached vaf_threshold at rank k is NOT fitted at the higher ranks.  After every rank the
 * still-unexplained trials are compacted and only they are launched at rank k + 1; every fit that runs is bit-identical
 * to the one hipnmf_rank_sweep_* runs for that (trial, rank) -- same starting point (keyed by the original trial index),
 * same kernel and launch geometry (chosen as for the full batch) -- so selected_out is identical by construction.
 * (trial, rank) pairs that were skipped report vaf = err = NaN, n_iter = 0 and all-zero components.  The reference
 * itself computes every rank (analysis.py:907-912) and leaves the thresholding to the user (:753-756).
 */
int hipnmf_rank_sweep_stop_f32(hipnmf_handle* h, const hipnmf_problem* p, int32_t k_min, int32_t k_max, double vaf_threshold,
                               uint64_t seed, int32_t first_matrix, const float* X, float* W_ws, float* H_out, float* vaf_out,
                               int32_t* selected_out, float* err_out, int32_t* n_iter_out);
int hipnmf_rank_sweep_stop_f64(hipnmf_handle* h, const hipnmf_problem* p, int32_t k_min, int32_t k_max, double vaf_threshold,
                               uint64_t seed, int32_t first_matrix, const double* X, double* W_ws, double* H_out, double* vaf_out,
                               int32_t* selected_out, double* err_out, int32_t* n_iter_out);

/* ---- EMG envelope preprocessing: the producer of X ------------------ */
/*
 * Batched GPU version of the tutorial pipeline that builds the matrix handed to find_synergies
 * (docs/source/tutorials/Finding muscle synergies.ipynb; src/muscle_synergies/analysis.py):
 *     zero_center (analysis.py:230-249)  ->  rms (analysis.py:435-507: sqrt(np.convolve(x^2, ones(W)/W, 'same')))
 *     ->  time_normalize (analysis.py:551-594: linear interp1d onto linspace(0, 1, n_out))
 *     ->  normalize (analysis.py:510-525: divide each channel by its max |value|)
 * Every stage is optional.  raw: [B] matrices in the layout given by x_layout / ldx / x_batch_stride
 * (same meaning as in hipnmf_problem); out: [B][n_channels][n_out ? n_out : n_samples], channel-major
 * (fed to hipnmf_fit_batched_* as HIPNMF_X_CHANNEL_MAJOR: in place for the channel-major kernels, one conversion
 * per fit for the row-major ones).
 */
typedef struct hipnmf_envelope_params {
  int32_t struct_size;    /* = sizeof(hipnmf_envelope_params)                                          */
  int32_t batch;          /* B >= 1 recordings                                                         */
  int64_t n_samples;      /* T rows (time samples) of every recording                                  */
  int32_t n_channels;     /* m columns (muscles)                                                       */
  int32_t x_layout;       /* HIPNMF_X_*                                                                */
  int64_t ldx;            /* leading dimension of one recording, in elements                           */
  int64_t x_batch_stride; /* elements between consecutive recordings                                   */
  int32_t window;         /* RMS window in samples, >= 1 (0 skips the RMS stage)                       */
  int32_t zero_center;    /* 1: subtract the per-channel mean first                                    */
  int32_t n_out;          /* 0: keep n_samples rows; > 0: time-normalise to n_out rows                 */
  int32_t normalize;      /* 1: divide every channel by its max absolute value                         */
  int32_t resample_kind;  /* HIPNMF_RESAMPLE_*: interp1d `kind` of the time normalisation (n_out > 0)  */
  int32_t reserved0;      /* must be 0                                                                 */
} hipnmf_envelope_params;
/* time_normalize forwards `kind` to scipy.interpolate.interp1d (analysis.py:551-594).  On the device: */
#define HIPNMF_RESAMPLE_LINEAR 0     /* 'linear' (the reference's default; 'slinear' is the same interpolant) */
#define HIPNMF_RESAMPLE_NEAREST 1    /* 'nearest': nearest knot, a tie goes down                              */
#define HIPNMF_RESAMPLE_NEAREST_UP 2 /* 'nearest-up': a tie goes up                                           */
#define HIPNMF_RESAMPLE_PREVIOUS 3   /* 'previous' and 'zero' (order-0 spline): last knot <= the abscissa     */
#define HIPNMF_RESAMPLE_NEXT 4       /* 'next': first knot >= the abscissa                                    */
/* 'quadratic' / 'cubic' (interp1d -> make_interp_spline: a banded collocation solve over all samples of a channel, then an
 * evaluation) are linear in the samples and depend on (n_samples, n_out, kind) only; the solve's influence decays geometrically, so
 * every output row is a short window of weights over the input.  The host builds that operator once per shape (the Python host:
 * from scipy's own B-spline design matrices, preprocess._spline_operator) and hipnmf_resample_weights_* applies it on the device:
 *   out[b][j][r] = sum_{i < taps} weights[r * taps + i] * x[b][first[r] + i][j]        (fp64 accumulation, two fixed chains)
 * x as in hipnmf_envelope_params (batch, n_samples, n_channels, x_layout, ldx, x_batch_stride; n_out = rows of the operator; the
 * other fields are ignored); first [n_out] int32 and weights [n_out][taps] doubles in DEVICE memory, first[r] + taps <= n_samples;
 * out [B][n_channels][n_out], channel-major.  Errors: HIPNMF_ERR_BAD_ARG for NULL pointers, taps outside [1, n_samples] or a bad
 * layout (the windows themselves are not range-checked: the host built them).  Replaces scipy.interpolate.interp1d(kind='quadratic' | 'cubic') inside
 * time_normalize (src/muscle_synergies/analysis.py:551-594). */
int hipnmf_resample_weights_f32(hipnmf_handle* h, const hipnmf_envelope_params* p, const float* x, const int32_t* first,
                                const double* weights, int32_t taps, float* out);
int hipnmf_resample_weights_f64(hipnmf_handle* h, const hipnmf_envelope_params* p, const double* x, const int32_t* first,
                                const double* weights, int32_t taps, double* out);

int hipnmf_emg_envelope_f32(hipnmf_handle* h, const hipnmf_envelope_params* p, const float* raw, float* out);
int hipnmf_emg_envelope_f64(hipnmf_handle* h, const hipnmf_envelope_params* p, const double* raw, double* out);


/*
 * Batched IIR filter stage of the reference: digital_filter (src/muscle_synergies/analysis.py:314-432 ->
 * scipy.signal.sosfiltfilt when zero_lag, scipy.signal.sosfilt otherwise, along the time axis) and, with
 * zero_center / rectify set, linear_envelope (analysis.py:252-311: zero_center -> abs -> low-pass).
 * The section coefficients are designed on the host exactly as the reference does (scipy.signal.butter /
 * cheby1 / cheby2(..., output="sos"), analysis.py:381-403) and handed over as `sos` = [n_sections][6] doubles
 * in HOST memory ({b0, b1, b2, 1, a1, a2} per section); `zi` = [n_sections][2] doubles in HOST memory =
 * scipy.signal.sosfilt_zi(sos) (zero_lag only; NULL lets the library compute the same steady state itself,
 * possibly different in the last bit).  x as in hipnmf_envelope_params; y: [B][n_channels][n_samples],
 * channel-major.  All arithmetic is fp64 in scipy's operation order without fused multiply-adds: the f64
 * entry point reproduces scipy bit for bit on identical input, the f32 one reads float samples and rounds the
 * fp64 result to float.  Errors: HIPNMF_ERR_BAD_ARG when n_samples <= padlen (scipy's ValueError) or
 * sos[:, 3] != 1; HIPNMF_ERR_UNSUPPORTED for more than 8 sections.
 */
typedef struct hipnmf_sosfilt_params {
  int32_t struct_size;    /* = sizeof(hipnmf_sosfilt_params)                                           */
  int32_t batch;          /* B >= 1 recordings                                                         */
  int64_t n_samples;      /* T rows (time samples) of every recording                                  */
  int32_t n_channels;     /* m columns (muscles)                                                       */
  int32_t x_layout;       /* HIPNMF_X_*                                                                */
  int64_t ldx;            /* leading dimension of one recording, in elements                           */
  int64_t x_batch_stride; /* elements between consecutive recordings                                   */
  int32_t n_sections;     /* second-order sections, 1..8                                               */
  int32_t zero_lag;       /* 1: sosfiltfilt (forward-backward, padtype='odd'); 0: sosfilt, zero state  */
  int32_t padlen;         /* zero_lag: samples of extension at each end; -1 = scipy's default 3*ntaps  */
  int32_t zero_center;    /* 1: subtract the per-channel mean before filtering (linear_envelope)       */
  int32_t rectify;        /* 1: take |x| (after centring) before filtering (linear_envelope)           */
  int32_t mode;           /* HIPNMF_SOSFILT_*: EXACT (0, scipy's recurrence bit for bit) or SCAN (time-parallel) */
} hipnmf_sosfilt_params;
/* mode (round 4; the field was `reserved0`, must-be-zero, before: old callers get the exact mode):
 *  HIPNMF_SOSFILT_EXACT  the sequential recurrence in scipy's operation order, no fused multiply-adds: the f64 entry point is
 *                        bit-identical to scipy.  Bound by the dependent fp64 chain of the recursion (0.16 of HBM at 1024 x 16 x 20 000).
 *  HIPNMF_SOSFILT_SCAN   every series cut into 256 chunks filtered at once, chunk-end states combined by a scan over the workgroup
 *                        (csrc/sosfilt_scan.hpp); fp64 with fused multiply-adds, agrees with scipy to rounding times the filter's
 *                        conditioning (1e-12 relative for the reference's 6 Hz low-pass at 2 kHz; tests bound 1e-10), not bit for
 *                        bit.  Series longer than one workgroup holds (20 224 extended samples in float, 10 496 in double) are cut
 *                        into blocks with a scan over the blocks' end states (any length up to 2^31 bytes per series). */
#define HIPNMF_SOSFILT_EXACT 0
#define HIPNMF_SOSFILT_SCAN 1

int hipnmf_sosfilt_f32(hipnmf_handle* h, const hipnmf_sosfilt_params* p, const double* sos, const double* zi,
                       const float* x, float* y);
int hipnmf_sosfilt_f64(hipnmf_handle* h, const hipnmf_sosfilt_params* p, const double* sos, const double* zi,
                       const double* x, double* y);

/* ---- on-device NNDSVD initialisation building blocks ---------------------- */
/*
 * sklearn's default init for find_synergies is NNDSVDa (_initialize_nmf, _nmf.py:221-373) on a randomized SVD.
 * For T >> m the same leading singular triplets follow from the m x m Gram matrix; the two T-long passes
 * run here, the m x m eigen-problem and the k x m algebra stay on the host (muscle_synergies_amd/init.py):
 *   hipnmf_gram          : gram[b] = X_b^T X_b (m x m, fp64) and colsum[b] = column sums of X_b (fp64)
 *   hipnmf_nndsvd_stats  : for u_j = X v_j / s_j  (j < k): sum of squares of the positive part, of the negative
 *                          part, and the signed entry of largest magnitude (svd_flip's pivot): stats[b][k][4]
 *   hipnmf_nndsvd_write  : W0[b][t][j] = coef[b][j][0] * max(+-u_j[t], 0) (sign = coef[b][j][1]), entries below
 *                          `eps` set to 0, zeros then replaced by fill[b] (NNDSVDa; pass 0 for plain NNDSVD)
 * X as in hipnmf_problem (only batch, n_samples, n_features, n_components, x_layout, ldx, x_batch_stride are
 * read); V: [B][k][m] right singular vectors, inv_s: [B][k] = 1 / s_j; W0: [B][T][k] row-major.
 * All array arguments are device pointers; gram, colsum, stats, coef, fill are fp64 for both variants.
 */
int hipnmf_gram_f32(hipnmf_handle* h, const hipnmf_problem* p, const float* X, double* gram, double* colsum);
int hipnmf_gram_f64(hipnmf_handle* h, const hipnmf_problem* p, const double* X, double* gram, double* colsum);
int hipnmf_nndsvd_stats_f32(hipnmf_handle* h, const hipnmf_problem* p, const float* X, const double* V,
                            const double* inv_s, double* stats);
int hipnmf_nndsvd_stats_f64(hipnmf_handle* h, const hipnmf_problem* p, const double* X, const double* V,
                            const double* inv_s, double* stats);
int hipnmf_nndsvd_write_f32(hipnmf_handle* h, const hipnmf_problem* p, const float* X, const double* V,
                            const double* inv_s, const double* coef, const double* fill, double eps, float* W0);
int hipnmf_nndsvd_write_f64(hipnmf_handle* h, const hipnmf_problem* p, const double* X, const double* V,
                            const double* inv_s, const double* coef, const double* fill, double eps, double* W0);

/* ---- diagnostics ------------------------------------------------------------------------------------------- */
/*
 * Rate (GB/s) at which the memory system serves the batched solver's access pattern, measured on the spot: `regions`
 * workgroups (one per matrix, one per CU at a time) each read their own `region_bytes` (a multiple of 1024; one matrix: 640 KB for
 * 16 x 10 000 fp32) `passes` times with 16-byte loads, no arithmetic.  With regions >> the number of CUs the data lives in HBM
 * and the 256 regions being read sit in the Infinity Cache -- the ceiling bench.py prices the headline kernel against
 * (roofline.memory.stream_peak_gbs), taken in the run that quotes it.  Uses the handle's workspace (regions * region_bytes).
 */
int hipnmf_diag_stream_gbs(hipnmf_handle* h, int64_t region_bytes, int32_t regions, int32_t passes, double* gbs_out);

#ifdef __cplusplus
}
#endif
#endif /* HIP_NMF_H */
