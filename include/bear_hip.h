/*
 * bear_hip.h -- C ABI of libbear_hip.so: the MI355X (gfx950) implementation of BEAR's
 * empirical-Bayes training hot path.
 *
 * The reference (debbiemarkslab/BEAR) is pure Python on TensorFlow; it has no FFI.  The
 * boundary below is what a reference maintainer binds through ctypes (INTEGRATION.md
 * shows the stub) to replace, per batch, the TensorFlow graph built by
 *   bear_model/bear_net.py:146-197  (_train_step)       -> bear_dm_prior_f64
 *   bear_model/bear_ref.py:207-259  (_train_step)       -> bear_dm_ref_f64
 *   bear_model/core.py:73-74, 138-139 (counts_log_prob) -> both (train_ar selects)
 *   bear_model/dataloader.py:35-46  (TSV -> tensors)    -> bear_parse_counts_tsv
 *
 * Conventions
 *   - every entry point returns an int status: BEAR_OK (0) or a negative bear_status;
 *     bear_strerror() names it.  Nothing throws across the boundary.
 *   - the CALLER owns every buffer.  Pointers marked [dev] are device pointers valid on
 *     the workspace's device (e.g. torch.Tensor.data_ptr()), 16-byte aligned, row-major.
 *     The library owns only the bear_ws handle.
 *   - launches are asynchronous on `stream` (a hipStream_t passed as void*; NULL = the
 *     default stream).  Outputs are valid after the stream is synchronised.
 *   - re-entrant across distinct workspaces / streams; one workspace must not be used
 *     by two streams at once.
 *   - the calling thread's current HIP device must be the workspace's device.
 *   - count rows are 5 wide: A, C, G, T (or U) and the stop symbol, in the order of
 *     bear_model/core.py:146-147; counts are uint32 (KMC's counter limit, summarize.py:66-67).
 *   - there is NO CPU fallback: without a gfx950 device the compute entry points fail
 *     with BEAR_ERR_NO_DEVICE.
 */
#ifndef BEAR_HIP_H
#define BEAR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BEAR_ABI_VERSION 6 /* 6: + bear_plan_create_auto; bear_plan_create_ref may hand back the plan's dense form (round 6) */
#define BEAR_ROW_WIDTH 5 /* alphabet_size + 1 for dna/rna */

typedef enum bear_status {
  BEAR_OK = 0,
  BEAR_ERR_INVALID_ARG = -1, /* null / misaligned pointer, bad flag */
  BEAR_ERR_NO_DEVICE = -2,   /* no usable HIP device */
  BEAR_ERR_WRONG_DEVICE = -3,/* current device != workspace device */
  BEAR_ERR_HIP = -4,         /* a HIP runtime call failed; see bear_last_hip_error() */
  BEAR_ERR_NOMEM = -5,
  BEAR_ERR_IO = -6,          /* file could not be opened / read */
  BEAR_ERR_PARSE = -7        /* malformed count-table row */
} bear_status;

typedef struct bear_ws bear_ws; /* opaque workspace: per-block partial sums + launch geometry */

int bear_abi_version(void);
const char *bear_strerror(int status);
/* hipError_t of the most recent failing HIP call on this thread (0 if none). */
int bear_last_hip_error(void);

/* Allocates the workspace on `device` (makes no change to the caller's current device).
 * A workspace serves ONE stream at a time: the block partials and the arrival counter of the kernels whose last block
 * forms the final sums are per-launch state, so launches that may overlap (two streams, two host threads) each take
 * their own workspace; launches ordered on one stream share one freely.  After a launch or synchronisation FAILED, destroy
 * the workspace and create a new one before replaying a captured graph that uses it: an arrival counter left half way by the
 * failed launch is discarded by later EAGER launches only (their stamps are newer), not by replays of an older capture. */
int bear_ws_create(int device, bear_ws **out);
int bear_ws_destroy(bear_ws *ws);

/*
 * bear_net._train_step arithmetic for one batch / shard, unscaled
 * (bear_model/bear_net.py:146-197; alpha = prior / exp(h_signed) + eps, bear_net.py:43;
 *  AR mode probs = prior + eps, bear_net.py:68).
 *   counts  [dev] uint32 [n_rows, 5]   transition counts of the training column
 *   prior   [dev] double [n_rows, 5]   ar_func(kmers) rows
 *   out     [dev] double [2]           out[0] = sum_i LL_i ; out[1] = d out[0] / d h_signed
 *                                      (0 in AR mode, bear_net.py:194-196)
 *   grad_prior [dev, nullable] double [n_rows, 5]  dLL_i / d prior_ib
 * The caller applies the -(num_kmers / batch) scale of bear_net.py:190-191.
 */
int bear_dm_prior_f64(bear_ws *ws, const uint32_t *counts, const double *prior, uint64_t n_rows,
                      double h_signed, double eps, int train_ar, double *out, double *grad_prior,
                      void *stream);

/*
 * bear_ref._train_step arithmetic with the stop net-function (bear_model/bear_ref.py:207-259,
 * prior from bear_ref.py:30-33, 63-68, reference column preprocessed as bear_ref.py:332-337,
 * net function bear_model/ar_funcs.py:121-126), unscaled.
 *   train, ref [dev] uint32 [n_rows, 5]
 *   out        [dev] double [4] = { sum LL, d/d h_signed, d/d tau_signed, d/d net_weight_signed }
 */
int bear_dm_ref_f64(bear_ws *ws, const uint32_t *train, const uint32_t *ref, uint64_t n_rows,
                    double h_signed, double tau_signed, double nu_signed, double eps, int train_ar,
                    double *out, void *stream);

/*
 * Planned variants: the per-step hot path for a count table that stays resident in HBM across
 * optimizer steps (the reference re-reads one cached table every epoch, bear_model/dataloader.py:47-48).
 * A plan holds what depends on the counts only -- per 512-context tile the (context, column) work
 * items sorted by count, as uint16 offsets, plus global lists of the rare large-count items -- so
 * the per-step kernels do no sorting.  Results are identical to the unplanned entry points.
 *   The plan is a lossless sorted sparse encoding of the counts: the planned step kernels read it INSTEAD of the
 *   count rows (the `counts` / `train` arguments of the planned entries only identify the table).
 *   bear_plan_create: synchronous; `ncol` = 5 for bear_dm_prior_plan_f64 (all columns are items),
 *     4 for bear_dm_ref_plan_f64 (the stop column has a context-independent concentration).
 *     The plan is valid for exactly the buffer contents it was built from; rebuild after any change.
 *   bear_plan_bytes: device bytes held by the plan (= extra HBM traffic per step).
 * The library owns the plan's device memory; bear_plan_destroy releases it.
 */
typedef struct bear_plan bear_plan;
int bear_plan_create(bear_ws *ws, const uint32_t *counts, uint64_t n_rows, int ncol, bear_plan **out);
/* The same for a caller that only runs the mode-N entry points (bear_dm_prior_plan_f64 / _grad_f64 / _dev_f64: any torch AR
 * function) on the plan: where more than half of the table's cells are beyond the sorted encoding's product path (counts of
 * 1e3 ... 1e5 -- a k-mer table at small k, bear_model/data/ysd1_lag_5_file_0_preshuf.tsv) the plan comes back in its DENSE form,
 * *rowwise = 1: it keeps nothing per item (a few KB of histograms), and the step streams the caller's count rows and prior rows,
 * a context per thread, every cell through the table-log form of the Stirling difference -- the sorted form of such a table is
 * ~100 B per context of overflow lists whose gathers bound the step (2e7 dense contexts: 1.30 -> 0.9 ms, with gradient rows 2.76 ->
 * 1.18 ms).  Same results to rounding.  Every other entry point that takes a plan (the fused linear / convolutional steps,
 * bear_plan_pair_contexts, ...) returns BEAR_ERR_INVALID_ARG for a dense-form plan: they walk the sorted encoding. */
int bear_plan_create_auto(bear_ws *ws, const uint32_t *counts, uint64_t n_rows, int *rowwise, bear_plan **out);
int bear_plan_destroy(bear_plan *plan);
uint64_t bear_plan_bytes(const bear_plan *plan);
/* Diagnostics: the tiles of a plan -- first context, number of contexts, number of product-path items (1 <= count <= 24) and
 * byte offset of the tile's block in the plan stream; [host] arrays of `count` entries for tiles [first, first + count). */
uint64_t bear_plan_tile_count(const bear_plan *plan);
int bear_plan_tile_info(const bear_plan *plan, uint64_t first, uint64_t count, uint64_t *row0, uint32_t *rows, uint32_t *items,
                        uint64_t *stream_offset);
/* A plan for bear_dm_ref_plan_f64 / bear_ref_train_*_f64 that also knows the REFERENCE column `ref` [dev] uint32 [n_rows, 5]
 * -- as constant as the training column while a table is resident.  Contexts without reference counts (most k-mers of a read
 * set that the reference genome does not contain) share one concentration per letter, so all their items collapse into a
 * 24-bin histogram over the count, built here once; the step then streams only the items of contexts that do have reference
 * counts, as 16-byte records {count, r_b, sum r} sorted by count.  Results are those of the streaming path (same arithmetic
 * per item; the fp64 sums in another order).  The planned entries must be called with the same `train` and `ref` buffers.
 * A table of LARGE counts (more than half of its cells beyond the sorted encoding's product path: bear_plan_create_auto's test; the
 * reference's ysd1 table) gets the plan's dense form here too: nothing kept per item, the planned mode-R entries and
 * bear_ref_train_*_f64 stream the training and reference rows, a context per thread (2e7 such contexts: 1.41 -> ~0.9 ms, 153 -> 0.002 B
 * of plan per context).  Nothing changes for the caller. */
int bear_plan_create_ref(bear_ws *ws, const uint32_t *train, const uint32_t *ref, uint64_t n_rows, bear_plan **out);
/* prior_normalized != 0: the caller asserts that every row of `prior` sums to one -- true for each
 * ar_func of the reference, all of which end in a softmax (bear_model/ar_funcs.py:44,97,121-126).
 * The concentration total A = 1/h + 5 eps is then shared by all contexts and the context terms
 * come from the plan's histogram of totals; with 0 the kernel sums each row and uses the shared A
 * only where the sum is 1 to 2 ulp. */
int bear_dm_prior_plan_f64(bear_ws *ws, const bear_plan *plan, const uint32_t *counts, const double *prior,
                           uint64_t n_rows, double h_signed, double eps, int train_ar, int prior_normalized,
                           double *out, void *stream);
/* The same with the gradient rows: grad_prior [dev] double [n_rows, 5] = d sum LL / d prior (BEAR mode), what
 * grad_tape.gradient hands back to ar_func (bear_model/bear_net.py:193).  With prior_normalized != 0 the rows are turned into
 * their gradient in place in LDS (double-buffered tiles: 84 B per context at the streaming rate); as everywhere, prior rows
 * are non-negative (a concentration f / h + eps must be positive). */
int bear_dm_prior_plan_grad_f64(bear_ws *ws, const bear_plan *plan, const uint32_t *counts, const double *prior,
                                uint64_t n_rows, double h_signed, double eps, int train_ar, int prior_normalized,
                                double *out, double *grad_prior, void *stream);
int bear_dm_ref_plan_f64(bear_ws *ws, const bear_plan *plan, const uint32_t *train, const uint32_t *ref,
                         uint64_t n_rows, double h_signed, double tau_signed, double nu_signed, double eps,
                         int train_ar, double *out, void *stream);

/* The same with h_signed read from device memory ([dev] double [1], e.g. a parameter tensor the optimizer updates in place): the
 * step is enqueued without the host reading the parameter back (one sync fewer per step; needed for steps that are followed
 * by an all-reduce on the same stream).  grad_prior [dev, nullable]: NULL = no gradient rows (parameter-free ar_func). */
int bear_dm_prior_plan_dev_f64(bear_ws *ws, const bear_plan *plan, const uint32_t *counts, const double *prior, uint64_t n_rows,
                               const double *h_signed_dev, double eps, int train_ar, int prior_normalized, double *out,
                               double *grad_prior, void *stream);

/*
 * One optimizer step with every moving quantity in device memory, in two halves:
 *
 *   bear_*_train_reduce_f64  this rank's shard, ONE launch for bear_ref and the linear head (every block derives the kernel
 *                            constants from theta in its prologue, the last block to finish sums the per-block partials in a
 *                            fixed order; the cnn step adds its forward / backward launches) into
 *                            packed [dev] double [1 + n_theta] = { sum LL, d sum LL / d theta[0..n_theta) }  (unscaled)
 *   bear_train_apply_f64     tf.keras Adam (beta 0.9 / 0.999, epsilon 1e-7; bear_ref.py:312-313, 346-350) on theta with the
 *                            gradients scale * packed[1..]; loss_buf[step] = -scale * packed[0] (the "elbo" the reference logs);
 *                            one single-block launch (update, step counter, loss record)
 *
 * Nothing synchronises with the host.  One rank enqueues them back to back (bear_*_train_step_f64 below: capturable in a
 * HIP graph and replayed -- the reference traces its step once with tf.function, bear_model/bear_ref.py:207; the bundled
 * example is 10 000 steps over 1365 contexts, launch-bound).  With the rows sharded over several ranks the caller puts ONE
 * all-reduce(sum) of `packed` between the halves -- what strategy.reduce (bear_net.py:290) and the cross-replica gradient sum
 * inside optimizer.apply_gradients (bear_net.py:278-282) do -- still without a host round trip.
 *
 *   theta    [dev] double [n_theta], updated in place by apply:
 *            bear_ref / stop net function: {h_signed, tau_signed, net_weight_signed}           (n_theta = 3)
 *            bear_net / linear:            {h_signed, mat[lag,5,5]}                            (n_theta = 1 + 25 lag)
 *            bear_net / cnn:               {h_signed, params[bear_cnn_param_count(...)]}       (n_theta = 1 + param count)
 *   adam_m, adam_v [dev] double [n_theta], adam_t [dev] double [1]: optimizer state, zero before the first step
 *   scale    the loss scale -(num_kmers / global batch) (bear_ref.py:252-253); 1 when packed already holds scaled sums
 *            (gradient accumulation: the caller adds scale_k * packed_k over acc_steps batches, bear_net.py:193-196)
 *   train_ar != 0: theta[0] = h_signed gets no gradient (bear_net.py:194-196)
 *   loss_buf [dev, nullable] double [loss_cap], indexed by the step counter adam_t
 *   the cnn step borrows per-context buffers prior_buf [n,5], t1_buf [n,16], grad_rows_buf [n,5]; call bear_cnn_reserve once
 *   before capturing (it sizes the library's block-partial buffer; nothing allocates inside a step).  The step evaluates the
 *   AR function only for the contexts that hold training counts (the plan's lists: nothing else enters the ELBO or a gradient);
 *   the other rows of prior_buf / t1_buf are left as they were -- hand over initialised (e.g. zeroed) buffers.
 */
/*
 * Optional, once per batch, after bear_plan_create (and after the k-mer order, if any): prefix levels for the convolutional step.
 * In a k-mer-sorted batch the window of position p (letters [p, p + filter_width)) is the same for all contexts that share their
 * first p + filter_width letters, and those are neighbours; with levels attached, bear_net_cnn_train_reduce_f64 / _step_f64
 * called with this plan, this kmer_code pointer, lag and filter_width evaluate a position once per DISTINCT prefix, forward and
 * backward (one launch per level; the sums are the same up to rounding) -- about 1.3 instead of 6 positions per context on a
 * dense sorted table of 13-mers; sparser tables keep the prefix lengths that pay (e.g. from lag - 2 letters down).  The levels are tied to the buffer kmer_code (bear_pack_kmers_u64 form; identity and contents,
 * like the plan's count slab).  *n_levels (nullable) = 0: nothing was attached (rows in another order, a table too sparse for its
 * prefixes to repeat, a plan whose lists skip rows, a shape outside the fused kernels): the step runs as before.
 * Holds about 45 bytes per context on a dense table.  Synchronises `stream` (set-up path).
 * A plan with levels attached carries PER-LAUNCH state (the levels' layer-1 / dT1 rows are written by every forward and backward
 * pass over them, also through a `const bear_plan *`): like a workspace it serves one stream at a time.
 */
int bear_plan_attach_cnn_levels(bear_plan *plan, const uint64_t *kmer_code, int lag, int filter_width, int *n_levels, void *stream);
/* Rows and prefix lengths (letters) of the attached levels 1 .. n (rows_out [host, nullable when capacity = 0], letters_out [host,
 * nullable]); returns their number.  A prefix length whose prefixes hardly repeat is skipped: the level below evaluates its position too. */
int bear_plan_cnn_level_rows(const bear_plan *plan, uint64_t *rows_out, int *letters_out, int capacity);
/* bear_plan_attach_cnn_levels also attaches WINDOW TABLES to every level (level 0 = the contexts) for the positions the level's rows
 * would evaluate themselves, from the last one up: what a position adds to a row's layer-1 sums depends on its filter_width-letter
 * window alone, and a batch holds at most 6^filter_width distinct windows (65 536 for eight letters of ACGT) however many rows a
 * level has.  Forward: the position is evaluated once per distinct window, a row gathers its window's row; backward: a window's dT1
 * row is the sum of its rows' dT1 rows (listed by window at attach time: fixed order, no atomics), then the position's backward pass
 * over the windows.  A table is attached while the level holds >= 8 rows per distinct window (at most 6 per level); 8 bytes per row
 * and table.  Distinct windows, positions and levels of the attached tables (nullable outputs); returns their number. */
int bear_plan_cnn_window_rows(const bear_plan *plan, uint64_t *rows_out, int *pos_out, int *level_out, int capacity);
/* bear_cnn_forward_f64 over the plan's prefix levels when they were attached for this kmer_code pointer, lag and filter width
 * (the plain forward otherwise); t1_save is required. */
int bear_cnn_forward_plan_f64(bear_ws *ws, const bear_plan *plan, const uint64_t *kmer_code, uint64_t n_rows, int lag, int filter_width,
                              int num_filters, int layer1_width, const double *params, double *prior, double *t1_save, void *stream);
int bear_train_apply_f64(double *theta, int n_theta, const double *packed, double *adam_m, double *adam_v, double *adam_t,
                         double learning_rate, double scale, int train_ar, double *loss_buf, uint64_t loss_cap, void *stream);
int bear_ref_train_reduce_f64(bear_ws *ws, const bear_plan *plan, const uint32_t *train, const uint32_t *ref, uint64_t n_rows,
                              const double *theta, double eps, int train_ar, double *packed, void *stream);
int bear_net_linear_train_reduce_f64(bear_ws *ws, const bear_plan *plan, const uint32_t *counts, const uint64_t *kmer_index, int lag,
                                     uint64_t n_rows, const double *theta, double eps, int train_ar, double *packed, void *stream);
int bear_cnn_reserve(bear_ws *ws, uint64_t n_rows, int lag, int filter_width, int num_filters, int layer1_width);
int bear_net_cnn_train_reduce_f64(bear_ws *ws, const bear_plan *plan, const uint32_t *counts, const uint64_t *kmer_code, uint64_t n_rows,
                                  int lag, int filter_width, int num_filters, int layer1_width, const double *theta, double *prior_buf,
                                  double *t1_buf, double *grad_rows_buf, double eps, int train_ar, double *packed, void *stream);
/* reduce + apply in one call (single rank); `out` / `packed` as above.  bear_ref_train_step_f64 and
 * bear_net_linear_train_step_f64 are ONE launch since round 6: the last block of the reduce kernel -- every other block has
 * arrived, i.e. is done reading theta -- runs the update of bear_train_apply_f64 behind its fixed-order sums (same source, same
 * bits as the two-launch form; BEAR_AMD_TWO_LAUNCH_STEP=1 in the environment of the call restores reduce + apply).  The
 * convolutional step stays forward / DM / backward launches + apply. */
int bear_ref_train_step_f64(bear_ws *ws, const bear_plan *plan, const uint32_t *train, const uint32_t *ref, uint64_t n_rows,
                            double *theta, double *adam_m, double *adam_v, double *adam_t, double eps, int train_ar,
                            double learning_rate, double scale, double *out, double *loss_buf, uint64_t loss_cap, void *stream);
int bear_net_linear_train_step_f64(bear_ws *ws, const bear_plan *plan, const uint32_t *counts, const uint64_t *kmer_index, int lag,
                                   uint64_t n_rows, double *theta, double *adam_m, double *adam_v, double *adam_t, double *packed,
                                   double eps, int train_ar, double learning_rate, double scale, double *loss_buf,
                                   uint64_t loss_cap, void *stream);
int bear_net_cnn_train_step_f64(bear_ws *ws, const bear_plan *plan, const uint32_t *counts, const uint64_t *kmer_code, uint64_t n_rows,
                                int lag, int filter_width, int num_filters, int layer1_width, double *theta, double *adam_m,
                                double *adam_v, double *adam_t, double *prior_buf, double *t1_buf, double *grad_rows_buf,
                                double *packed, double eps, int train_ar, double learning_rate, double scale, double *loss_buf,
                                uint64_t loss_cap, void *stream);

/*
 * The whole bear_net training step for the linear AR function, fused on a plan: replaces
 * ar_func = make_ar_func_linear(...) (bear_model/ar_funcs.py:23-46), _train_step's forward and
 * grad_tape.gradient(loss, [h_signed, mat]) (bear_model/bear_net.py:146-197).
 *   kmer_code [dev] uint64 [n_rows]   packed contexts, letter l in bits [3l, 3l+3): 0..3 letters, 4 = start
 *                                     symbol '[', 5 = any other character (all-zero one-hot row, core.py:173);
 *                                     positions >= lag hold 5.  bear_pack_kmers_u64 builds it from the int8
 *                                     code matrix [n_rows, lag] (values outside 0..4 -> 5).  lag <= 21.
 *   kmer_index [dev] uint64 [n_rows]  what the linear-head entry points read: the same contexts as row numbers of the
 *                                     kernel's letter-group tables (pairs of letters in 6-bit fields, the last three
 *                                     letters in an 8-bit field behind them), built ONCE per batch from kmer_code by
 *                                     bear_linear_index_u64 (the contexts of a batch do not change between steps).
 *                                     16-byte aligned.  Fastest when the rows of the batch are sorted by k-mer (first
 *                                     letter most significant; the sums do not depend on the order): consecutive
 *                                     contexts then share their leading groups and the backward pass adds once per wave.
 *   mat       [dev] double [lag,5,5]  the AR parameter
 *   out       [dev] double [2]        { sum LL, d sum LL / d h_signed }
 *   grad_mat  [dev] double [lag,5,5]  d sum LL / d mat (overwritten)
 * Reads 8 bytes per context plus the plan; writes nothing per context.
 */
int bear_pack_kmers_u64(const int8_t *codes, uint64_t n_rows, int lag, uint64_t *packed, void *stream);
int bear_linear_index_u64(const uint64_t *kmer_code, uint64_t n_rows, int lag, uint64_t *kmer_index, void *stream);
/* ASCII k-mer bytes as parsed from the count file [dev] uint8 [n_rows, lag] -> int8 letter codes [dev] [n_rows, lag]:
 * 0..3 = A, C, G, T (U when rna != 0), 4 = '[', -1 = anything else (the all-zero one-hot row of core.py:173). */
int bear_encode_kmers_i8(const uint8_t *ascii, uint64_t n_rows, int lag, int rna, int8_t *codes, void *stream);
int bear_dm_linear_f64(bear_ws *ws, const bear_plan *plan, const uint32_t *counts, const uint64_t *kmer_index,
                       const double *mat, int lag, uint64_t n_rows, double h_signed, double eps, int train_ar,
                       double *out, double *grad_mat, void *stream);
/*
 * Optional, once per batch, after bear_plan_create and bear_linear_index_u64 (and after the k-mer order below, if any): pairs
 * contexts of the plan's lists that share all letters but the last three, so that the fused step reads their shared table rows
 * once and reduces their gradients together, and orders the pairs of such a run so that the 16 lanes of a pass of an LDS atomic
 * meet on as few bank pairs as the run allows (copies of one k-mer share a lane and one add) -- about 15 % off a step on a
 * k-mer-sorted table; the sums are the same.  ~0.03 s per 1e8 contexts (3 ms for a batch of 5e6).  The pairing is tied to the buffer kmer_index (identity and contents: it must not change afterwards, like the
 * plan's count slab) and to lag: bear_dm_linear_f64 / bear_net_linear_train_*_f64 called with that pointer and lag take the
 * paired form, any other call the plain one.  A tile whose paired list would not fit the kernel's row threads (a stretch of
 * the table where neighbours share nothing) keeps its plain list; *paired (nullable) = 0 when that is true of more than half
 * of the tiles (a sparse table): the plan is then left as it was.  Synchronises `stream` (set-up path).
 */
int bear_plan_pair_contexts(bear_plan *plan, const uint64_t *kmer_index, int lag, int *paired, void *stream);
/* BEAR_AMD_DETERMINISTIC=1 (environment, read per call): parameter gradients that are bit-identical from run to run.
 *   bear_dm_linear_f64 / bear_net_linear_train_*: the gradient tables of the fused linear step hold 64-bit fixed-point integers
 *     scaled by 2^62 / bound, where bound >= the sum of |gradient| over everything that is added into one gradient -- derived from
 *     the table's counts (sum, non-zero cells, maximum) and the launch's h (kernels_linear.h, lin_fx_bound):
 *     d/d mat is then the exact integer sum of the contexts' rounded gradients -- independent of the order of the adds, of the
 *     cut into tiles and blocks and of the form of the plan's lists (a step that takes two launches -- paired tiles, then the
 *     tiles that kept their plain lists -- keeps the first launch's integers in the workspace's accumulator: round 6) -- turned into
 *     a double once at the end of the STEP.  Each context's gradient is rounded to bound * 2^-62 (instead of to its own last bit).
 *     Across ranks every rank converts its own integer sum before the all-reduce adds the doubles: bit-reproducible for a given
 *     number of ranks and cut of the batch, equal to rounding (<= 1e-15 of the largest entry) between different ones.
 *   bear_cnn_backward_f64 / bear_net_cnn_train_*: the backward kernel runs ONE wave per block (the block's gradient image takes its
 *     adds in program order); reproducible from run to run for a given launch geometry, at an eighth of the waves per CU.
 * The scalar sums (sum LL, d/dh, d/dtau, d/dnet_weight) are fixed-order sums in every mode.
 * bear_plan_count_total: total[3] / bound[3] [host, nullable] = {sum of all counts, cells that hold a count, largest count} of the
 *   plan's table / the values in force for the scale (default: the table's own).
 * bear_plan_set_count_bound: bound[3] >= the plan's own, each < 2^50 -- the same three of EVERYTHING that is added into one gradient
 *   with this plan's rows (ranks that share a batch: sums of the first two, maximum of the third, over the ranks). */
int bear_plan_count_total(const bear_plan *plan, double *total, double *bound);
/* 1 for libbear_hip_det.so (built with -DBEAR_DET_BUILD): everything above is always on AND the work units of a tile are dealt to
 * the waves statically instead of drawn, so that every sum of a launch -- sum LL and d/dh too -- is bit-identical from run to run
 * (in the regular library those are fixed-order across blocks, but which thread adds which item inside a block follows the draw). */
int bear_deterministic_build(void);
int bear_plan_set_count_bound(bear_plan *plan, const double *bound);
/* How the pairing went: tiles that take the paired form / tiles that keep their plain list (outputs nullable); returns 1 when the
 * plan is paired, 0 when it is not. */
int bear_plan_pair_info(const bear_plan *plan, uint64_t *paired_tiles, uint64_t *plain_tiles);

/*
 * The linear AR function as prior ROWS, forward and backward: replaces make_ar_func_linear's ar_func (bear_model/ar_funcs.py:41-45)
 * and grad_tape.gradient through it wherever the rows themselves are needed -- evaluation / h_scan (bear_model/bear_net.py:
 * 387-531), bear_ref with the linear net function (the rows are mixed with the reference prior, bear_model/bear_ref.py:63-68),
 * get_var_probs.  (bear_net's training step never forms them: bear_dm_linear_f64 above.)
 *   kmer_code  [dev] uint64 [n_rows]    packed contexts (bear_pack_kmers_u64); lag <= 21
 *   mat        [dev] double [lag,5,5]   the AR parameter
 *   forward:   prior [dev] double [n_rows,5] = softmax(sum_l mat[l, kmer[l], :]), 16-byte aligned; unknown letters add nothing
 *   backward:  grad_prior [dev] double [n_rows,5] = d L / d prior; prior = the forward rows;
 *              grad_mat [dev] double [lag,5,5] = d L / d mat (overwritten; LDS fp64 atomics: reproducible to rounding)
 * Row order: any; fastest in k-mer order (consecutive contexts share their leading letters and the backward pass adds once
 * per wave for them).  Asynchronous on `stream`; one launch each.
 */
int bear_linear_forward_f64(bear_ws *ws, const uint64_t *kmer_code, uint64_t n_rows, int lag, const double *mat, double *prior,
                            void *stream);
int bear_linear_backward_f64(bear_ws *ws, const uint64_t *kmer_code, uint64_t n_rows, int lag, const double *prior,
                             const double *grad_prior, double *grad_mat, void *stream);

/*
 * bear_ref's prior rows for a net function with parameters (linear, cnn): replaces the arithmetic of _make_ref_ar_func's ar_func
 * (bear_model/bear_ref.py:63-68 with _counts_to_probs, :9-33) and grad_tape.gradient through it.  (With the stop net function
 * the mixing lives inside bear_dm_ref_*_f64 and no rows exist.)
 *   Every [n_rows,5] array is 16-byte aligned.
 *   net_rows  [dev] double [n_rows,5]  the net function's rows g (bear_linear_forward_f64, bear_cnn_forward_f64, ...)
 *   ref_rows  [dev] double [n_rows,5]  the reference column as the driver passes it: (counts + eps) with the stop column
 *                                      zeroed (bear_model/bear_ref.py:332-337)
 *   tau_signed, net_weight_signed [dev] double [1]  the two parameters, read on the device (the optimizer's tensors)
 *   forward:  prior [dev] double [n_rows,5] = (nw g + jukes_cantor(ref, tau)) / (nw + 1), 16-byte aligned
 *   backward: grad_prior [dev] double [n_rows,5] = d L / d prior;  grad_net_rows [dev] double [n_rows,5] = d L / d g (16-byte
 *             aligned);  grad_scalars [dev] double [2] = { d L / d tau_signed, d L / d net_weight_signed }
 * Asynchronous on `stream`; one launch each.  4-letter alphabets.
 */
int bear_ref_mix_forward_f64(bear_ws *ws, const double *net_rows, const double *ref_rows, uint64_t n_rows, const double *tau_signed,
                             const double *net_weight_signed, double *prior, void *stream);
int bear_ref_mix_backward_f64(bear_ws *ws, const double *net_rows, const double *ref_rows, const double *grad_prior, uint64_t n_rows,
                              const double *tau_signed, const double *net_weight_signed, double *grad_net_rows, double *grad_scalars,
                              void *stream);

/*
 * bear_ref's training step for a net function with parameters, the reference mixing INSIDE the DM step: replaces _train_step of
 * bear_model/bear_ref.py:207-259 from the net function's rows onwards -- the mixing of bear_ref.py:63-68 (as
 * bear_ref_mix_forward_f64), sum LL and its gradients (as bear_dm_prior_plan_grad_f64 with prior_normalized), and the way back
 * through the mixing (as bear_ref_mix_backward_f64) -- in one launch: 127 B per context instead of three launches and 367 B.
 *   plan       five-column plan of the training counts (bear_plan_create(ws, counts, n_rows, 5, ...))
 *   net_rows   [dev] double [n_rows,5]  the net function's rows g: non-negative, every row sums to one (softmax output)
 *   ref_rows   [dev] double [n_rows,5]  (reference counts + eps) with the stop column zeroed (bear_model/bear_ref.py:332-337)
 *   h_signed_dev, tau_signed_dev, net_weight_signed_dev  [dev] double [1] each (the optimizer's tensors)
 *   train_ar   != 0: the multinomial of bear_model/core.py:138-139 on the mixed rows (no h: out[1] = 0)
 *   out        [dev] double [4] = { sum LL, d/d h_signed, d/d tau_signed, d/d net_weight_signed }  (unscaled)
 *   grad_net_rows [dev] double [n_rows,5] = d sum LL / d g: what the net function's backward pass takes
 * Row arrays 16-byte aligned.
 */
int bear_dm_refmix_plan_grad_f64(bear_ws *ws, const bear_plan *plan, const uint32_t *counts, const double *net_rows,
                                 const double *ref_rows, uint64_t n_rows, const double *h_signed_dev, const double *tau_signed_dev,
                                 const double *net_weight_signed_dev, double eps, int train_ar, double *out, double *grad_net_rows,
                                 void *stream);

/*
 * The convolutional AR function of bear_net, forward and backward (replaces make_ar_func_cnn's ar_func,
 * bear_model/ar_funcs.py:49-99, and grad_tape.gradient through it, bear_model/bear_net.py:193) for 4-letter alphabets,
 * num_filters = 30 and kmer_layer1_width = 16 (every reference config), lag <= 21, filter_width <= lag.
 *   kmer_code [dev] uint64 [n_rows]      packed contexts (bear_pack_kmers_u64)
 *   params    [dev] double [bear_cnn_param_count(...)]  the parameters flattened and concatenated in the reference's
 *             order (ar_funcs.py:98-99): filters [fw,5,nf], intercept0 [P,nf], weights1 [P,nf,l1], intercept1 [l1],
 *             weights2 [l1,5], intercept2 [5], scale0 [P,nf], scale1 [l1];  P = lag - fw + 1
 *   forward:  prior [dev] double [n_rows,5] = ar_func rows; t1_save [dev, nullable] double [n_rows,16] = the layer-1
 *             pre-normalisation sums, which the backward entry takes instead of redoing the first tensordot
 *   backward: grad_prior [dev] double [n_rows,5] = d L / d prior (e.g. from bear_dm_prior_plan_grad_f64);
 *             grad_params [dev] double [param_count] = d L / d params (overwritten), same layout as params.
 *   Row order: any (the sums do not depend on it).  Both kernels are fastest when the rows are sorted by k-mer (first letter
 *   most significant, as bear_net.train uploads a batch): consecutive contexts then share their leading letters, a window that
 *   a whole wave (forward: 64 contexts) / tile (backward: 32) shares is evaluated once per distinct window, and the backward
 *   pass of such a position needs only the column sums of the tile's d t1 rows (it is linear in them) -- 18 / 52 ms instead
 *   of 33 / 104 ms per 1e8 contexts at lag 13, filter width 8.
 */
int bear_cnn_param_count(int lag, int filter_width, int num_filters, int layer1_width);
int bear_cnn_forward_f64(bear_ws *ws, const uint64_t *kmer_code, uint64_t n_rows, int lag, int filter_width, int num_filters,
                         int layer1_width, const double *params, double *prior, double *t1_save, void *stream);
int bear_cnn_backward_f64(bear_ws *ws, const uint64_t *kmer_code, uint64_t n_rows, int lag, int filter_width, int num_filters,
                          int layer1_width, const double *params, const double *t1_save, const double *prior,
                          const double *grad_prior, double *grad_params, void *stream);

/*
 * Held-out evaluation, one pass over a row range: replaces _evaluation_step of bear_model/bear_net.py:323-371
 * (and its bear_ref twin, bear_ref.py:391-446, with prior = the reference-mixed AR rows) and, with n_h > 1,
 * one batch of h_scan (bear_net.py:465-531).
 *   test   [dev] uint32 [n_rows,5]  held-out transition counts
 *   train  [dev] uint32 [n_rows,5]  or NULL (use_train = False)
 *   prior  [dev] double [n_rows,5]  ar_func rows (NULL only when n_h == 0 and with_ar == 0)
 *   h      [host] n_h > 0 values (not log-transformed); van_reg [host] n_van pseudo-counts; n_h + n_van <= 64
 *   out    [dev] double [2 (n_h + n_van) + 3] =
 *          { ll_ear[n_h], ll_arm, ll_van[n_van], correct_ear[n_h], correct_arm, correct_van[n_van], total_len }
 * The arg-max noise of core.py:69-71,134-136 is a counter-based hash of (noise_seed, model, row_base + row,
 * letter) -- the same sequence for any sharding of the rows -- restated in oracle/bear_oracle.py:eval_noise.
 */
int bear_eval_f64(bear_ws *ws, const uint32_t *test, const uint32_t *train, const double *prior, uint64_t n_rows,
                  const double *h, int n_h, int with_ar, const double *van_reg, int n_van, double eps,
                  uint64_t noise_seed, uint64_t row_base, double *out, void *stream);

/*
 * The same on a sorted plan of the TEST column, for a table that stays resident (a held-out evaluation after training, the
 * train-set evaluation, every value of an h_scan): bear_eval_plan_create lists, per tile of 448 contexts, the cells and the rows
 * with a non-zero test count sorted by count, the rows whose largest TRAINING counts tie (their vanilla arg-max is decided
 * by the noise: which letters tie, most first) and the rows beyond the vanilla models' histogram bins (training + test total >=
 * 2048), and counts once what the vanilla models need of the table as a whole (histograms of the integer arguments of their
 * lgamma differences, the test counts under a unique largest training count, the total length) -- asynchronous on `stream`,
 * ~16 B of plan per context.  `train` [dev, nullable] is the conditioning column the
 * evaluations will use (NULL: none); bear_eval_plan_f64 must be given the same test / train buffers.  It streams the row slabs
 * and the plan through a three-slot LDS ring and evaluates wave-uniform units of equal counts without a workgroup barrier --
 * rows without test transitions cost nothing.  Arguments, output vector and noise stream are those of bear_eval_f64
 * (identical results up to the order of the fp64 sums; accuracies exactly).  The plan is valid for exactly the buffer contents
 * it was built from.  With vanilla models the plan's tie lists stand for the arg-max of count + van_reg + eps + noise only while
 * the noise cannot bridge a whole count: BEAR_ERR_INVALID_ARG unless 1750 eps < 0.5 and 0 <= van_reg <= 2^30 (bear_eval_f64
 * takes any values).
 *   row_ids [dev, nullable] uint32 [n_rows], 16-byte aligned: the buffers hold a COMPACTED batch -- only the contexts with
 *           held-out counts (nothing else enters any of the sums) -- and row i is row `row_base + row_ids[i]` of the table: the
 *           key of its tie-breaking noise, so the accuracies of a compacted batch equal those of the whole batch exactly.
 *           NULL: row i is row `row_base + i`.
 */
typedef struct bear_eval_plan bear_eval_plan;
int bear_eval_plan_create(bear_ws *ws, const uint32_t *test, const uint32_t *train, uint64_t n_rows, bear_eval_plan **out, void *stream);
int bear_eval_plan_destroy(bear_eval_plan *plan);
uint64_t bear_eval_plan_bytes(const bear_eval_plan *plan);
int bear_eval_plan_f64(bear_ws *ws, const bear_eval_plan *plan, const uint32_t *test, const uint32_t *train, const double *prior,
                       uint64_t n_rows, const double *h, int n_h, int with_ar, const double *van_reg, int n_van, double eps,
                       uint64_t noise_seed, uint64_t row_base, const uint32_t *row_ids, double *out, void *stream);

/*
 * BMM marginal likelihood of one dataset column: replaces _marginal_step of bear_model/dataloader.py:111-118,
 *   out[k] = sum_i lbeta(counts_i + alpha_k) - lbeta(alpha_k 1_5),   alpha [host] n_alpha <= 64, out [dev].
 */
int bear_bmm_f64(bear_ws *ws, const uint32_t *counts, uint64_t n_rows, const double *alpha, int n_alpha, double *out,
                 void *stream);

/*
 * The primitive underneath both entry points, item by item (tests / diagnostics): for x > 0 and
 * integer c >= 0,  D[i] = lgamma(x+c) - lgamma(x)  and  P[i] = digamma(x+c) - digamma(x)
 * -- the two quantities TFP's lbeta and its autodiff yield in bear_model/core.py:73-74.
 *   path 0: the code path the fused kernels choose (product + table log for c <= 31, shifted
 *           Stirling series above);  path 1 / 2: the general routine for every item, on the library
 *           log / on the table log (the form the planned kernels use for large counts).
 *   x, D, P [dev] double [n];  c [dev] uint32 [n]
 */
int bear_dm_items_f64(bear_ws *ws, const double *x, const uint32_t *c, uint64_t n, int path, double *D,
                      double *P, void *stream);

/*
 * Posterior sampling of transition probabilities (inference apps).
 *
 * bear_log_gamma_f64 replaces log_gamma.log_gamma(concs, size) (bear_model/log_gamma.py:17-76):
 *   conc [dev] double [n] > 0;  out [dev] double [n_samples, n],  out[s, i] ~ log Gamma(conc[i], 1)
 *   -- accurate for tiny concentrations, where exp(out) underflows.  The reference draws from numpy's global
 *   generator; here a draw is a pure function of (seed, s, i), restated in oracle/bear_oracle.py:log_gamma_hash.
 *
 * bear_logdir_sample_f64 replaces the concentration assembly + sampling (or MAP) of get_var_probs.get_pdf
 * (bear_model/get_var_probs.py:128-183), output='numpy' layout:
 *   counts [dev, nullable] uint32 [n_rows,5]  training-column counts (NULL = unseen k-mers, all zero, :441-444)
 *   prior  [dev, nullable] double [n_rows,5]  ar_func rows (required when n_h > 0 or with_ar)
 *   h [host] n_h values; van [host] n_van values; n_h + n_van <= 64
 *   model order: (AR if with_ar) , BEAR h_0.., vanilla van_0..   (get_var_probs.py:146-153)
 *   map != 0: log(conc / sum conc), mc_samples forced to 1; with_ar requires map
 *   out [dev] double [n_rows, 5, n_models, mc_samples]  normalised log transition probabilities
 *   a draw is a pure function of (seed, model, sample, row_base + row, letter): any sharding of the rows
 *   reproduces the same table.
 */
int bear_log_gamma_f64(const double *conc, uint64_t n, uint64_t n_samples, uint64_t seed, double *out, void *stream);
int bear_logdir_sample_f64(const uint32_t *counts, const double *prior, uint64_t n_rows, const double *h, int n_h,
                           int with_ar, const double *van, int n_van, int mc_samples, int map, uint64_t seed,
                           uint64_t row_base, double *out, void *stream);

/*
 * Synthetic "k=13 sparse" count table for measurement: rows
 * [row0, row0 + n_rows) of a table defined by a counter-based hash of (seed, row), so any
 * shard of the same table can be generated independently on any GPU.
 *   train, test, ref [dev, each nullable] uint32 [n_rows, 5]
 *   dense != 0 selects the large-count stress distribution (ysd1-like, counts 1e3..3e5).
 */
int bear_synth_counts_u32(uint64_t seed, uint64_t row0, uint64_t n_rows, int dense, uint32_t *train,
                          uint32_t *test, uint32_t *ref, void *stream);
/* Read-only pass over n_bytes of device memory (16-byte lane loads, nothing written): the measured HBM read ceiling that
 * bench.py reports next to the 8 TB/s spec peak. */
int bear_stream_read(bear_ws *ws, const void *src, uint64_t n_bytes, void *stream);
/* prior [dev] double [n_rows, 5]: positive rows summing to 1 (softmax of hashed logits). */
int bear_synth_prior_f64(uint64_t seed, uint64_t row0, uint64_t n_rows, double *prior, void *stream);

/*
 * Host-side reader of the summarize.py count-table format (bear_model/summarize.py:429-449;
 * replaces the CsvDataset + tfio decode_json path of bear_model/dataloader.py:35-46).
 * Row: kmer '\t' '[[' c,c,c,c,c '],[' ... ']]' '\n' with num_ds groups of 5.
 *   bear_count_rows: number of non-empty lines.
 *   bear_count_newlines: `wc -l` of the file (newline bytes) -- the reference's num_kmers (models/train_bear_net.py:52-55).
 *   bear_parse_counts_tsv: fills, for rows [0, max_rows):
 *     kmers  [host] char  [max_rows, lag]      (no terminator; '[' padded as in the file)
 *     counts [host] uint32 [num_ds, max_rows, 5] (planar by dataset column)
 *   *n_rows_out receives the number of rows parsed.
 */
int bear_count_rows(const char *path, uint64_t *n_rows_out);
int bear_count_newlines(const char *path, uint64_t *n_out);
/* The sparse row format of dataloader.sparse_dataloader (bear_model/dataloader.py:52-109): `kmer; [[ds, col], ...]; [value, ...]`,
 * one line per k-mer behind `skip_lines` header lines.  counts [host] uint32 [num_ds, max_rows, width] (zeroed here, then filled). */
int bear_parse_sparse_counts(const char *path, int num_ds, int width, int lag, uint64_t skip_lines, uint64_t max_rows,
                             char *kmers, uint32_t *counts, uint64_t *n_rows_out);
int bear_parse_counts_tsv(const char *path, int num_ds, int lag, uint64_t max_rows, char *kmers,
                          uint32_t *counts, uint64_t *n_rows_out);
/*
 * One rank's rows of a row-sharded table (replaces strategy.experimental_distribute_dataset,
 * bear_model/bear_net.py:273): the table of total_rows rows (this file holds global rows [row_base, row_base + file rows)) is
 * cut into batches of batch_rows rows (last one short, dataloader.py:37) and every batch into `world` contiguous pieces,
 * base = m / world rows each, the first m % world pieces one row longer; rank `rank` decodes only the lines of its pieces, in
 * file order, into kmers [host] char [max_rows, lag] and counts [host] uint32 [num_ds, max_rows, 5]; the other lines are
 * stepped over.  skip_lines header lines are ignored first (dataloader.py:7 `header`; world = 1 makes this the plain reader with
 * a header).  bear_shard_rows_count gives the number of local rows (the size to allocate) from the row counts alone.
 */
int bear_shard_rows_count(uint64_t row_base, uint64_t file_rows, uint64_t total_rows, uint64_t batch_rows, int rank, int world,
                          uint64_t *n_local_out);
int bear_parse_counts_tsv_shard(const char *path, int num_ds, int lag, uint64_t skip_lines, uint64_t row_base, uint64_t total_rows,
                                uint64_t batch_rows, int rank, int world, uint64_t max_rows, char *kmers, uint32_t *counts,
                                uint64_t *n_local_out, uint64_t *n_file_rows_out);

/*
 * Binary cache of a parsed count table: what bear_parse_counts_tsv produced, stored as it is uploaded,
 * so later runs / other ranks skip the text decode (the reference re-decodes per process and caches in RAM only,
 * bear_model/dataloader.py:47-48).  File: 64-byte header {"BEARCT01", n_rows, lag, num_ds, size and mtime of the
 * source text, offsets}, k-mer bytes [n_rows, lag], uint32 counts [num_ds, n_rows, 5].
 *   bear_stat_source: size / mtime (ns) of a text file, the staleness key stored in the header.
 *   bear_cache_write: atomic (temporary file + rename).
 *   bear_cache_read:  rows [row0, row0 + n_rows) into kmers [host, nullable] char [n_rows, lag] and
 *                     counts [host] uint32 [num_ds, n_rows, 5] -- a rank reads only its shard.
 */
int bear_stat_source(const char *path, uint64_t *size_out, int64_t *mtime_ns_out);
int bear_cache_write(const char *path, const char *kmers, const uint32_t *counts, uint64_t n_rows, int lag, int num_ds,
                     uint64_t src_size, int64_t src_mtime_ns);
int bear_cache_info(const char *path, uint64_t *n_rows, int *lag, int *num_ds, uint64_t *src_size, int64_t *src_mtime_ns);
int bear_cache_read(const char *path, uint64_t row0, uint64_t n_rows, char *kmers, uint32_t *counts);

/*
 * On-device shuffle of resident rows (replaces the `shuf` step the reference asks for before training,
 * docs/usage.rst:191-200): dst[i] = src[perm(i)], perm a keyed bijection of [0, n_rows) (4-round Feistel network with
 * cycle walking; bear_shuffle_source_row evaluates it on the host, oracle/bear_oracle.py:shuffle_perm restates it).
 *   src, dst [dev] n_rows rows of row_bytes bytes (20: a count slab; 8: packed k-mers; lag: k-mer bytes); dst != src.
 * Calling it with the same seed on every column of a table keeps the columns aligned.
 */
int bear_shuffle_rows(const void *src, void *dst, uint64_t n_rows, uint32_t row_bytes, uint64_t seed, void *stream);
uint64_t bear_shuffle_source_row(uint64_t i, uint64_t n_rows, uint64_t seed);

/*
 * k-mer order of a batch.  The sums of a training step (bear_net.py:146-197) do not depend on the order of a batch's rows, and
 * the fused AR-function entries below (bear_dm_linear_f64, bear_cnn_forward / backward_f64, bear_net_*_train_*_f64) run about
 * twice as fast when consecutive contexts share their leading letters -- 1.8 vs 2.9 ms (linear) and 54 vs 104 ms (cnn) per
 * 1e8 contexts: they are CORRECT for any order, the order is a performance precondition.  A caller establishes it once per
 * batch, before bear_plan_create (a plan is tied to the row order of the count slab it was built from):
 *   bear_kmer_order_u64: perm [dev] uint32 [n_rows] such that kmer_code[perm[0]] <= kmer_code[perm[1]] <= ... lexicographically
 *                        with the FIRST letter most significant (stable: equal contexts keep their order); kmer_code [dev] is
 *                        the bear_pack_kmers_u64 form; n_rows < 2^32.  The scratch is the CALLER's: call once with
 *                        scratch = NULL to get the size in *scratch_bytes (about 20 B per row + rocPRIM's temporary storage),
 *                        then with a 256-byte aligned device buffer of at least that size (*scratch_bytes = its size).
 *                        Asynchronous on `stream`: nothing is allocated, freed or waited for on the host.
 *   bear_gather_rows:    dst[i] = src[perm[i]] for rows of row_bytes bytes (20: a count slab; 8: packed contexts; lag: k-mer
 *                        bytes); dst != src; asynchronous on `stream`.
 * bear_amd.bear_net.train does exactly this at upload (bear_amd/_train.py: ResidentBatches(kmer_order=True)).
 */
int bear_kmer_order_u64(const uint64_t *kmer_code, uint64_t n_rows, int lag, uint32_t *perm, void *scratch, uint64_t *scratch_bytes,
                        void *stream);
int bear_gather_rows(const void *src, const uint32_t *perm, void *dst, uint64_t n_rows, uint32_t row_bytes, void *stream);

/*
 * k-mer transition counting on the device: the rows summarize.py produces through KMC and its stage-3
 * heap merge (bear_model/summarize.py:380-622), computed from the sequences as bear_model/tests/test_summarize.py:88-115
 * defines them.  Per lag: emit (context, group*5 + next letter) per transition, radix sort by context, run-length reduce.
 *   text  [dev] uint8 [n_pos]  per sequence: 5 (start), letters 0..3 (6 = any other character), 4 (stop); n_pos < 2^32
 *   group [dev] uint8 [n_pos]  group id of the sequence each position belongs to
 *   bear_kmer_sort_create: synchronous; *n_rows_out = number of distinct contexts (rows of the lag-`lag` table).
 *   bear_kmer_sort_reduce: kmers [dev, nullable] uint8 [n_rows, lag] ASCII ('[' padded as summarize.py:442);
 *                          kmer_code [dev, nullable] uint64 [n_rows] (bear_pack_kmers_u64 form);
 *                          counts [dev] uint32 [n_groups, n_rows, 5] (planar, the layout of bear_parse_counts_tsv).
 *   Rows come out sorted by packed context code; shuffle with bear_shuffle_rows before training.
 * bear_write_counts_tsv [host]: rows row_begin, row_begin + row_step, ... in the summarize.py text format
 *   (summarize.py:429-449), e.g. step = number of output bins.
 * bear_fastx_size / bear_fastx_encode [host]: a FASTA (fastq == 0) or FASTQ file as that code text (replacing the Biopython
 *   readers of summarize.py:96-100); reverse != 0 appends every sequence's reverse complement (summarize.py:202-207);
 *   group_out [nullable] receives `group` at every position.  Size first, then fill (`capacity` positions).
 */
int bear_fastx_size(const char *path, int fastq, int reverse, uint64_t *n_pos_out, uint64_t *n_seqs_out);
int bear_fastx_encode(const char *path, int fastq, int reverse, int group, uint64_t capacity, uint8_t *text,
                      uint8_t *group_out, uint64_t *n_pos_out);
typedef struct bear_kmer_sort bear_kmer_sort;
int bear_kmer_sort_create(const uint8_t *text, const uint8_t *group, uint64_t n_pos, int lag, bear_kmer_sort **out,
                          uint64_t *n_rows_out, void *stream);
int bear_kmer_sort_reduce(const bear_kmer_sort *h, int n_groups, uint8_t *kmers, uint64_t *kmer_code, uint32_t *counts,
                          void *stream);
int bear_kmer_sort_destroy(bear_kmer_sort *h);
int bear_count_last_hip_error(void);
int bear_write_counts_tsv(const char *path, const char *kmers, const uint32_t *counts, uint64_t n_rows, int lag, int num_ds,
                          uint64_t row_begin, uint64_t row_step, int append);

#ifdef __cplusplus
}
#endif
#endif /* BEAR_HIP_H */
