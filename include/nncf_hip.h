/*
 * nncf_hip.h -- C ABI of libnncf_hip.so, the MI355X (gfx950) implementation of the NNCF
 * sampling + scoring hot path.
 *
 * The reference (chentingpc/NNCF) has exactly one FFI on this path: the Cython class
 * `MultinomialSampler` (sampler/sampler.pyx:19-38) over `nodesampler::NodeSampler`
 * (sampler/nodesampler.cpp:16-77).  Everything else on the path is reached through
 * Keras `train_on_batch` / `predict_on_batch` on the dict returned by
 * `get_model` (models/model_framework.py:18,195-206).  Each entry point below names the
 * reference interface it replaces.  INTEGRATION.md shows the binding a maintainer of
 * the reference would add (ctypes stubs in place of `from sampler import
 * MultinomialSampler` and of the Keras model objects).
 *
 * Conventions
 *  - plain C types only; device buffers are raw pointers owned by the caller (e.g. the
 *    storage of a torch tensor); the library never frees caller memory;
 *  - every function returns 0 on success or a negative nncf_status; nncf_last_error()
 *    gives a thread-local message;
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream); no entry point
 *    synchronises the device unless its comment says so;
 *  - one handle = one sequential stream of state; handles are not thread-safe, distinct
 *    handles are independent.
 */
#ifndef NNCF_HIP_H_
#define NNCF_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  NNCF_OK = 0,
  NNCF_ERR_INVALID = -1,  /* bad argument (maps to ValueError) */
  NNCF_ERR_NOMEM = -2,    /* host or device allocation failed (MemoryError) */
  NNCF_ERR_HIP = -3,      /* HIP runtime error, no usable device (RuntimeError) */
  NNCF_ERR_STATE = -4     /* call order / handle state error (RuntimeError) */
} nncf_status;

const char* nncf_last_error(void);
/* Library version string, e.g. "nncf_hip 0.1 (gfx950)". */
const char* nncf_version(void);
/* Number of visible HIP devices (0 when there is none); never throws. */
int nncf_device_count(void);

/* ------------------------------------------------------------------------------------
 * B1  Negative sampler  --  replaces nodesampler::NodeSampler / MultinomialSampler
 *     (sampler/nodesampler.cpp:16-77, sampler/sampler.pyx:19-38)
 *
 * The reference materialises `int neg_table[1e8]` where id i owns a share proportional
 * to dist[i]^power (set_table, nodesampler.cpp:29-49) and draws
 * neg_table[((s = s*25214903917+11) >> 16) % 1e8]  (Rand, :23-26; sample_batch, :73-76).
 * The table is monotone in the id, so it is fully described by the int32 boundary array
 * bound[i] = #entries holding an id <= i.  The library builds bound[] on the host with
 * the reference's own double-precision predicate (bit-identical boundaries) in O(V), and
 * the device kernel maps each draw r to the id with bound[id-1] <= r < bound[id].
 * Sample streams are bit-identical to the reference for equal LCG state.
 * ---------------------------------------------------------------------------------- */
typedef struct nncf_sampler nncf_sampler;

/* NodeSampler::NodeSampler (nodesampler.cpp:55-67) + set_table (:29-49).
 * dist: host pointer, n doubles (degree / weight per id; zero => never sampled).
 * seed: LCG state before the first draw.  seed == 0 => time(NULL), as the reference
 *       (:59-60).  (The reference leaves the state uninitialised for seed != 0; here
 *       "state0 := seed" is the contract.)
 * device: HIP device ordinal that will own the boundary arrays.
 * Errors: n <= 0, n >= 1e8, NULL dist, negative/non-finite weight, all-zero dist
 *         -> NNCF_ERR_INVALID; no device -> NNCF_ERR_HIP. */
int nncf_sampler_create(const double* dist, int64_t n, double power, uint64_t seed, int device,
                        nncf_sampler** out);
int nncf_sampler_destroy(nncf_sampler* s);

/* Host-only half of nncf_sampler_create: the boundary array for `dist` (no device needed).
 * Restates set_table (nodesampler.cpp:29-49) in O(V); bit-identical to scanning the
 * reference's table. */
int nncf_sampler_build_bounds(const double* dist, int64_t n, double power, int32_t* out_host);

/* Copies bound[0..n) (int32, host) -- used to hash/compare against the reference table. */
int nncf_sampler_bounds(const nncf_sampler* s, int32_t* out_host);
int64_t nncf_sampler_size(const nncf_sampler* s);

/* NodeSampler::sample_batch (nodesampler.cpp:73-76), device output.
 * Writes n int32 ids to d_out (device memory) on `stream` and advances the LCG state by
 * n draws.  Asynchronous. */
int nncf_sampler_sample_dev(nncf_sampler* s, int64_t n, int32_t* d_out, void* stream);

/* Same draw, but only for positions [offset, offset+count) of the next n-draw batch;
 * does NOT advance the state.  Lets G ranks each produce their slice of one logical
 * stream (concatenation == the 1-GPU stream); callers then call nncf_sampler_skip(n). */
int nncf_sampler_sample_slice_dev(nncf_sampler* s, int64_t offset, int64_t count, int32_t* d_out,
                                  void* stream);
int nncf_sampler_skip(nncf_sampler* s, int64_t n);

/* MultinomialSampler.sample_batch (sampler.pyx:33-38): device draw + copy into a host
 * buffer of n int32; synchronises `stream`. */
int nncf_sampler_sample_batch(nncf_sampler* s, int64_t n, int32_t* out_host, void* stream);

/* LCG state access (the value `seed` holds in nodesampler.cpp:19). */
int nncf_sampler_get_state(const nncf_sampler* s, uint64_t* state);
int nncf_sampler_set_state(nncf_sampler* s, uint64_t state);

/* LCG jump-ahead helper: state after `n` draws from `state` (no sampler needed). */
uint64_t nncf_lcg_jump(uint64_t state, uint64_t n);


/* ------------------------------------------------------------------------------------
 * B2  Scoring / training step  --  replaces the Keras models returned by get_model
 *     (models/model_framework.py:18-206) for the id-table towers:
 *       Emb_U[uid] / Emb_C[cid] gathers      model_framework.py:59-63,85-88
 *       tf.nn.l2_normalize                   model_framework.py:62-63,109-111
 *       InteractionDot 'mul' / 'matmul'      modules/interaction/interaction_dot.py:92-107
 *       loss families                        utils/objectives.py:35-220
 *       activity_l2(u_reg)                   utils/utilities.py:122-135
 *     and the optimizer step with the north-star's sparse SGD scatter update
 *     (table[row] -= learn_rate * sum of that row's gradients in the batch).
 *
 * Tables are fp32, row-major, `row_stride` floats per row (>= dim, multiple of 4, rows
 * 16-byte aligned); ids are int32 device arrays; all work is enqueued on `stream`.
 * Duplicate rows inside a batch are reduced in a fixed order (no float atomics): results
 * are run-to-run deterministic.
 * ---------------------------------------------------------------------------------- */
typedef enum {
  NNCF_LOSS_SKIP_GRAM = 0,  /* objectives.py:59-64,99-105,202-209 */
  NNCF_LOSS_MSE = 1,        /* objectives.py:65-70,106-112,210-217 */
  NNCF_LOSS_LOG_LOSS = 2,   /* objectives.py:54-58,95-98,198-201 */
  NNCF_LOSS_MAX_MARGIN = 3  /* objectives.py:49-53,90-94,194-197 */
} nncf_loss;

typedef struct {
  int32_t batch_size_p;    /* b: positives per step (conf.batch_size_p); multiple of 32 for the shared-negative schemes */
  int32_t num_negatives;   /* k: sampled negatives per positive (train_scheme=original) */
  int32_t dim;             /* d = user_dim = item_dim; multiple of 4, <= 128 */
  int32_t row_stride;      /* floats between table rows */
  int32_t loss;            /* nncf_loss */
  int32_t normalize_user;  /* conf.emb_normalization on U_emb (model_framework.py:62-63) */
  int32_t normalize_item;  /* ... on C_emb (model_framework.py:109-111; not for 'mf') */
  float loss_gamma;        /* conf.loss_gamma */
  float neg_loss_weight;   /* conf.neg_loss_weight */
  float u_reg;             /* conf.u_reg */
  float learn_rate;        /* SGD step */
  int64_t user_count;
  int64_t item_count;
} nncf_model_cfg;

typedef struct nncf_model nncf_model;

/* Binds caller-owned device tables and allocates the per-step workspace (gradient slabs,
 * duplicate-row plans, loss partials).  max_steps = largest number of steps a single
 * nncf_epoch_* call will be asked to run (sizes the per-epoch plan buffers; >= 1). */
int nncf_model_create(const nncf_model_cfg* cfg, float* d_user_table, float* d_item_table, int device,
                      int64_t max_steps, nncf_model** out);
int nncf_model_destroy(nncf_model* m);
/* Optimiser of the model's tables.  mode 0: sparse SGD on the touched rows (the build's default,
 * BASELINE north-star).  mode 1: lazy-sparse Adam of utils/optimizer.py:108-134 (the reference
 * selects it with learn_rate < 0, basic_embedding_conf.py:61-66): only the rows of the batch move;
 * duplicate ids are summed first (TF's _apply_sparse_duplicate_indices).  mode 2: Keras Adam
 * (basic_embedding_conf.py:59-60): dense -- rows without a gradient still decay their moments and
 * move.  lr_t = learn_rate sqrt(1 - beta2^t) / (1 - beta1^t), p -= lr_t m / (sqrt(v) + epsilon).
 * d_m_*, d_v_*: zero-initialised arrays shaped like the tables (item arrays may be NULL while
 * item gradients are exported).  Resets the step counter t. */
int nncf_model_set_optimizer(nncf_model* m, int32_t mode, float beta1, float beta2, float epsilon, float* d_m_user,
                             float* d_v_user, float* d_m_item, float* d_v_item);
/* Mode 1 only -- what happens to an id that occurs several times in a batch (utils/optimizer.py:108-134).
 * as_written == 0 (default): the occurrences' gradients are summed first and the row takes ONE Adam step (TF's
 * Optimizer._apply_sparse_duplicate_indices wraps _apply_sparse that way).  as_written != 0: the file's own
 * arithmetic for an IndexedSlices that still holds duplicates -- m and v are scatter_update'd from each occurrence
 * (the last one in batch order wins: m = beta1 m_old + (1 - beta1) g_last, the same for v) and the step
 * lr_t m / (sqrt(v) + eps) is scatter_sub'ed once PER OCCURRENCE.  Id tables under neg_shared / group_neg_shared /
 * row-wise batches (a lookup per batch position); `original` batches go through nncf_step_rowwise; the one-launch
 * step (which folds duplicates into one row) is bypassed; text-tower word tables keep the summed form. */
int nncf_model_set_sparse_adam_duplicates(nncf_model* m, int32_t as_written);
int nncf_model_get_step(nncf_model* m, int64_t* out);
/* The one-launch shared-negative step (no reference counterpart: the reference runs one TF session.run per step,
 * models/train_neg_shared.py:46-51) steps table rows in place behind an in-launch hand-shake that needs the whole grid
 * resident at once.  shared != 0: this model does NOT have the device to itself (a second process or rank on the GPU,
 * an evaluator on a side stream) -> every step takes the stage + commit form, which waits for nothing inside a launch
 * (same results bit for bit).  Environment default: NNCF_SHARED_DEVICE=1. */
int nncf_model_set_shared_device(nncf_model* m, int32_t shared);
/* Clears the sticky time-out flag nncf_model_check reports.  The flag means the tables hold a PARTIAL step: restore
 * them from a checkpoint (and switch to nncf_model_set_shared_device(m, 1) if the GPU is shared) before training on. */
int nncf_model_clear_error(nncf_model* m, void* stream);
/* Synchronises `stream` and reports whether an in-launch wait ever timed out (the wait is bounded: NNCF_FUSED_SPIN_MS,
 * default 250 ms; a workgroup that times out stores nothing and every later loss of the model reads NaN):
 * NNCF_OK, or NNCF_ERR_HIP with the explanation in nncf_last_error(). */
int nncf_model_check(nncf_model* m, void* stream);

/* InteractionDot bias (modules/interaction/interaction_dot.py:62-71,96-107; conf.interaction_bias
 * 'user' | 'item' | 'both'): every score becomes U.V + d_user_bias[uid] + d_item_bias[cid] (training
 * steps and nncf_predict_scores); the biases take the model's optimiser step (call after
 * nncf_model_set_optimizer; Adam needs zero-initialised moment arrays of the same lengths).
 * d_item_bias has n_items entries indexed by CATALOGUE id also when the item table is a per-batch
 * block (text towers).  NULL pointers switch a bias off. */
int nncf_model_set_bias(nncf_model* m, float* d_user_bias, float* d_item_bias, int64_t n_items, float* d_m_user,
                        float* d_v_user, float* d_m_item, float* d_v_item);

/* learn_rate / loss hyper-parameters may change between steps (same shapes). */
int nncf_model_set_hparams(nncf_model* m, float learn_rate, float loss_gamma, float neg_loss_weight, float u_reg);

/* Measurement hook (bench.py roofline leg; no reference counterpart): between _begin and _end
 * every launch of the named kernel (e.g. "shared_tile_kernel", "original_score_kernel",
 * "tower_word_kernel", "tower_bwd_kernel") from any entry point is bracketed with HIP events
 * on its launch stream, for at most max_launches launches.  _end synchronises those events and
 * returns the mean bracket in milliseconds plus the mean of an EMPTY bracket (two events back to
 * back on the same stream) -- the marker-packet overhead to subtract when comparing with
 * rocprofv3.  Process-wide, not thread-safe; do not use while a HIP graph is being captured. */
int nncf_profile_begin(const char* kernel, int64_t max_launches);
int nncf_profile_end(double* avg_ms, int64_t* launches, double* overhead_ms);

/* Developer hook: per-workgroup s_memtime stamps of the last tile-kernel launch (only when the
 * model was created with NNCF_DEBUG_TIMING=1 in the environment).  Synchronous copy. */
int nncf_model_debug_stamps(nncf_model* m, long long* out_host, int64_t n);

/* model.train_on_batch for train_scheme=original (models/train_original.py:49-58 with the
 * 'model' head, model_framework.py:123,147-149; loss objectives.py:35-75).
 *   d_uid[b]         users of the b positives (row j of the negatives reuses user j / k)
 *   d_cid[b*(1+k)]   items: b positives, then k sampled negatives per positive, consecutively
 * Labels are implied by the layout (+1 positives; -1 / 0 negatives, train_original.py:13-14).
 * d_loss[1] receives task loss + regulariser (what train_on_batch returns).
 * update = 0 evaluates the loss only (SGD(0.) monitor models, model_framework.py:179-193). */
int nncf_step_original(nncf_model* m, const int32_t* d_uid, const int32_t* d_cid, float* d_loss, int update,
                       void* stream);

/* model.train_on_batch for ARBITRARY row-wise batches: n = b(1+k) independent (user, item, label)
 * rows in any order, as train_scheme = presample / group_sample / reverse build them
 * (models/train_presample.py:48-85, train_group_sample.py:75-88, train_reverse.py:59-68).
 * d_y[n]: +1 / -1 (skip-gram) or 1 / 0 (mse).  Pointwise losses only (the pairwise branches of
 * get_original_loss assume positives first, objectives.py:47-58) -> NNCF_ERR_INVALID otherwise.
 * nncf_epoch_rowwise: n_steps such batches back to back, d_*_all[n_steps * n]. */
int nncf_step_rowwise(nncf_model* m, const int32_t* d_uid, const int32_t* d_cid, const int32_t* d_y, float* d_loss,
                      int update, void* stream);
int nncf_epoch_rowwise(nncf_model* m, const int32_t* d_uid_all, const int32_t* d_cid_all, const int32_t* d_y_all,
                       int64_t n_steps, float* d_loss, void* stream);

/* model_neg_shared.train_on_batch (train_neg_shared.py:46-51; model_framework.py:128-129,
 * 153-155; loss objectives.py:78-117): S = U_b . V_b^T (b x b), diagonal = positives. */
int nncf_step_neg_shared(nncf_model* m, const int32_t* d_uid, const int32_t* d_cid, float* d_loss, int update,
                         void* stream);

/* Sharded variant of nncf_step_neg_shared (no reference counterpart): the item
 * "table" bound to the model is a buffer of rows FETCHED from their owner ranks and d_cid indexes
 * that buffer; user rows are updated in place, item rows are left untouched and their gradients
 * (after the normalisation backward, before the learning rate) are written to
 * d_item_grad[row][dim] for every row that occurs in d_cid. */
int nncf_step_neg_shared_export(nncf_model* m, const int32_t* d_uid, const int32_t* d_cid, float* d_loss,
                                float* d_item_grad, void* stream);

/* model_group_neg_shared.train_on_batch (train_group_neg_shared.py:53-61; model_framework.py:
 * 45-48,131-136,158-161; loss objectives.py:163-220): items are de-duplicated on device in
 * first-occurrence order (tf.unique), S = U_b . V_u^T (b x n_u), positive of row i = cid_x[i].
 * Optional outputs (may be NULL): d_cid_u[b] (first n_u valid), d_cid_x[b], d_n_u[1]. */
int nncf_step_group_neg_shared(nncf_model* m, const int32_t* d_uid, const int32_t* d_cid, float* d_loss,
                               int update, int32_t* d_cid_u, int32_t* d_cid_x, int32_t* d_n_u, void* stream);

/* Trainer hot loops (models/train_original.py:43-68, train_neg_shared.py:46-59,
 * train_group_neg_shared.py:53-68) over a device-resident, already shuffled link list:
 * step t uses links [t*b, (t+1)*b).  No host synchronisation inside; d_loss[t] = loss of
 * step t.  For `original`, d_neg holds n_steps*b*k presampled negatives (the sampler
 * stream consumed in the reference's order: step t draws [t*b*k, (t+1)*b*k)). */
int nncf_epoch_original(nncf_model* m, const int32_t* d_uid_all, const int32_t* d_cid_all, const int32_t* d_neg,
                        int64_t n_steps, float* d_loss, void* stream);
int nncf_epoch_neg_shared(nncf_model* m, const int32_t* d_uid_all, const int32_t* d_cid_all, int64_t n_steps,
                          float* d_loss, void* stream);
int nncf_epoch_group_neg_shared(nncf_model* m, const int32_t* d_uid_all, const int32_t* d_cid_all,
                                int64_t n_steps, float* d_loss, int32_t* d_n_u, void* stream);

/* model_neg_shared.predict_on_batch([uid (p,), cid (q,)]) -> (p, q) scores
 * (utils/objectives.py:310; model_framework.py:128-129): normalised-as-configured U.V^T.
 * d_out: p*q floats, row-major. */
int nncf_predict_scores(nncf_model* m, const int32_t* d_uid, int64_t p, const int32_t* d_cid, int64_t q,
                        float* d_out, void* stream);

/* ------------------------------------------------------------------------------------
 * basic_embedding item tower (modules/content/mean_pool.py:27-33,46-110; model_choice
 * basic_embedding, models/model_framework.py:90-91,98-106): word Embedding -> mean over the
 * max_content_len tokens (divides by L always, mean_pool.py:30) -> Dense(item_dim) ->
 * BatchNormalization (batch statistics over the batch's unique items; Keras-1 eps / momentum)
 * -> ReLU, with in-place sparse SGD.  The object borrows the caller's parameter arrays (device
 * memory, 16-byte aligned) and owns its scratch.
 *   content[item_count][content_len]   int32 word ids of every item (pad id included)
 *   item_ptr[item_count+1], item_word[], item_cnt[]: CSR of the same matrix, item -> distinct
 *     (word, occurrences <= 255) pairs -- the backward's transpose reads it.
 *   max_rows: most rows a train-mode block can hold (batch_size_p * (1 + num_negatives)).
 * ---------------------------------------------------------------------------------- */
typedef struct nncf_tower nncf_tower;

typedef struct nncf_tower_cfg {
  int32_t word_count;
  int32_t word_dim;     /* multiple of 4, <= 128 */
  int32_t item_dim;     /* multiple of 4, <= 128 */
  int32_t content_len;  /* L */
  int32_t max_rows;     /* multiple of 8, <= 8192 */
  int64_t item_count;
  float bn_eps;
  float bn_momentum;
  float learn_rate;
  /* cnn_embedding (modules/content/cnn_model.py:11-145, one Convolution1D layer with a single odd
   * filter length, border_mode 'same', -> BatchNorm over (items, positions) -> relu ->
   * Dropout(conv_dropout) -> average pooling over the whole length, then the Dense/BN/ReLU tail):
   * conv_filters > 0 selects it (multiple of 4, <= 128); 0 = the mean-pool tower. */
  int32_t conv_filters;
  int32_t filter_len;
  float conv_dropout;
  /* filter bank (cnn_model.py:63-81; the `default` preset's filter_lengths [[1, 3, 5]], cnn_embedding_conf.py:42-45):
   * n_filter_lens > 1 runs a Convolution1D of every listed odd length, conv_filters filters each, over the same
   * word rows and concatenates them along the filter axis (each with its own BatchNorm scale / shift); the Dense layer
   * then has n_filter_lens * conv_filters inputs.  conv_w holds the banks one after the other ([sum fl][word_dim][F]),
   * conv_b / bn1_* one 128-float pitched row per bank.  0 / 1: the single filter_len above. */
  int32_t n_filter_lens;
  int32_t filter_lens[4];
} nncf_tower_cfg;

typedef struct nncf_tower_params {
  float* words;    /* [word_count][word_dim] */
  float* dense_w;  /* [word_dim][item_dim] */
  float* dense_b;  /* [item_dim] */
  float* bn_gamma;
  float* bn_beta;
  float* bn_mean;  /* running statistics */
  float* bn_var;
  const int32_t* content;
  const int32_t* item_ptr;
  const int32_t* item_word;
  const uint8_t* item_cnt;
  /* cnn_embedding only (NULL otherwise) */
  float* conv_w;   /* [filter_len][word_dim][conv_filters] as Keras stores it; dense_w is then
                      [conv_filters][item_dim] */
  float* conv_b;   /* [conv_filters] */
  float* bn1_gamma;
  float* bn1_beta;
  float* bn1_mean;
  float* bn1_var;
  const uint8_t* item_spos;   /* [item_count][content_len]: positions of an item's tokens sorted by
                                 (word, position) */
  const uint8_t* item_start;  /* per CSR entry: index of its first position in the item's row of item_spos */
} nncf_tower_params;

/* Developer hook: per-workgroup clock stamps of the last conv forward / backward launches of a CNN tower created with
 * NNCF_CNN_DBG=64 in the environment: out[kernel (0 fwd, 1 bwd)][workgroup][16], workgroups = max_rows * filter groups.
 * Synchronous copy. */
int nncf_tower_debug_stamps(nncf_tower* t, long long* out_host, int64_t n);


int nncf_tower_create(const nncf_tower_cfg* cfg, const nncf_tower_params* params, int device, nncf_tower** out);
int nncf_tower_destroy(nncf_tower* t);
int nncf_tower_set_learn_rate(nncf_tower* t, float learn_rate);
/* cnn_embedding: call after the host changed conv_w (the tower keeps a transposed copy). */
int nncf_tower_sync_params(nncf_tower* t, void* stream);
/* conf.pooling (modules/content/mean_pool.py:68-76): 0 = 'average' (AverageEmbeddings: sum of the word rows / L, :27-33;
 * default), != 0 = 'max' (MaxPooling1D over the whole length, :70-72: per column the largest scale[w] W[w][c] of the item's
 * tokens -- rows dropped by Embedding(dropout) are zeros and take part; the gradient of a column goes to the word that
 * supplied its maximum).  CNN tower (cnn_embedding): conf.poolings[0] (modules/content/cnn_model.py:100-109) -- how the conv
 * output (BatchNorm, relu, Dropout) is pooled over the item's positions: != 0 = MaxPooling1D over the whole length (the
 * gradient of a filter goes to the position that supplied its maximum, first maximum in position order). */
int nncf_tower_set_pooling(nncf_tower* t, int32_t max_pooling);
/* Speed only (mean-pool tower, average pooling; no reference counterpart): up to 16 word ids that occur in most items of
 * any batch (the pad id, the corpus's most frequent words).  Their word-row gradient sum_s count[w][s] dL/dh[s] is
 * accumulated by the Dense backward's workgroups, which hold dL/dh, instead of by one wave of the word kernel gathering a
 * row per (word, item) cell; same sums up to fp32 association.  host_words: host array.  Call it BEFORE
 * nncf_tower_set_postings (NNCF_ERR_INVALID once postings are bound: their schedule is built around the hot set). */
int nncf_tower_set_hot_words(nncf_tower* t, const int32_t* host_words, int32_t n);
/* Speed only (mean-pool tower, average pooling, training blocks of more than 1,024 rows -- train_scheme = original and the
 * row-wise schemes; no reference counterpart): the transposed item CSR and a static schedule over it.
 *   d_postings [number of (item, word) pairs]: for every word the items that hold it (ascending), each entry = item id |
 *     (the word's count in the item) << 24 (item_count <= 2^24);
 *   d_units [n_workgroups][8][4] int32: per workgroup of the word-table backward 8 units {word (-1: empty slot), first posting,
 *     end, number of units of this word if this is its first unit else 0}; all units of a word sit in ONE workgroup, in
 *     order, every word of the table appears exactly once as a first unit (also words without postings: the optimiser visits
 *     them), a unit should hold <= 512 postings (more: the workgroup walks them in rounds of 512); a HOT word
 *     (nncf_tower_set_hot_words) keeps its first unit with an EMPTY range (first posting == end) -- its gradient row comes
 *     from the Dense backward, a non-empty range would add it twice -- which this call checks (NNCF_ERR_INVALID);
 *   d_slot_of [item_count] 64-bit words, zero-initialised, owned by the caller and written by the tower ({forward stamp,
 *     block row} of every item of the current block).
 * The word-table backward then walks its units' postings and keeps the items of the block instead of scanning a
 * (word_count x block rows) count matrix that is ~1 % full.  Device arrays, borrowed; all NULL (n_workgroups 0) switches the
 * mode off.  Same sums up to fp32 association (items ascending instead of block rows ascending). */
int nncf_tower_set_postings(nncf_tower* t, const int32_t* d_units, int32_t n_workgroups, const uint32_t* d_postings,
                            int64_t* d_slot_of);
/* conf.use_content_id (mean_pool.py:102-108): the tower's output becomes relu(BN(Dense(pool))) + Emb_Cid[item id] with
 * d_id_table [item_count][item_dim] (borrowed; NULL switches it off) and activity_l2(v_reg) on the Embedding's output
 * (v_reg * sum_d mean over the block's rows of row^2 added to the loss).  Steps of a tower with an id table materialise the
 * item block (statistics launch + BatchNorm / ReLU + the id rows), run the step kernels on it as on an id table and
 * step the id rows of the batch's unique items with the model's optimiser (d_m / d_v: zero-initialised moment arrays
 * shaped like the table for the Adam modes, d_stamp [item_count] int32 zeros for dense Adam; NULL for sparse SGD). */
int nncf_tower_set_content_id(nncf_tower* t, float* d_id_table, float v_reg, float* d_m, float* d_v, int32_t* d_stamp);
/* item_dense_transform variants (mean_pool.py:82-97, cnn_model.py:114-126): Activation(dense_hidden_actv) -- 0 relu (the
 * presets), 1 tanh, 2 sigmoid, 3 hard_sigmoid, 4 softplus, 5 softsign, 6 linear (Keras-1 names) -- and
 * Dropout(dense_hidden_dropout) behind Dense + BatchNormalization.  Anything but (relu, 0) makes the tower's steps materialise
 * the item block act(BN(z)) * mask, run the step kernels on it as on an id table and start the backward from
 * dL/dE * mask * act'(y) (as conf.use_content_id does; not the one-launch fast path).  The dropout mask is a counter-based
 * hash of (nncf_tower_set_dropout's seed, train-forward index, row, column); nncf_tower_dense_mask writes the mask
 * (including its 1/keep scale) of forward `step_index` for m rows, [m][item_dim] floats, for tests. */
int nncf_tower_set_dense_tail(nncf_tower* t, int32_t activation, float dropout);
/* conf.no_BN (modules/content/mean_pool.py:88-93: `assert conf.no_BN == True` succeeds, so BatchNormalization1 is never
 * applied): the tail is Dense -> Activation(dense_hidden_actv) -> Dropout(dense_hidden_dropout).  The Dense bias now has a
 * gradient of its own (sum of dz), bn_gamma / bn_beta / the running statistics are never read or written.  Steps run on the
 * materialised block like the other tail variants; mean-pool tower only (the CNN tower's convolution keeps its BatchNorm:
 * cnn_model.py:93 would drop that one too), no gradient export. */
int nncf_tower_set_no_bn(nncf_tower* t, int32_t no_bn);
/* A tower WITHOUT item_dense_transform (mean_pool.py:78-101, cnn_model.py:112-130: the else branch): the pooled block -- the
 * mean / max of the word rows, or the pooled conv features -- is the item row itself (its width must be item_dim); no Dense,
 * BatchNorm or Activation, no parameters behind the pooling.  Steps run on the materialised block as for the tail variants
 * (conf.use_content_id still adds its id rows); set before hot words / postings; no gradient export. */
int nncf_tower_set_no_dense(nncf_tower* t, int32_t no_dense);
int nncf_tower_dense_mask(nncf_tower* t, uint64_t step_index, int64_t m, float* d_out, void* stream);
/* conf.c_reg (mean_pool.py:56-59, cnn_model.py:45-48: activity_l2(conf.c_reg) on the word Embedding; L1L2Act,
 * utils/utilities.py:122-135): c_reg * sum over (position, column) of the mean over the block's rows of E^2 joins the loss
 * of every model step of this tower, E = the Embedding's OUTPUT after its row dropout [K-mem]; the word rows of the block
 * take its gradient 2 c_reg / n * (occurrences in the block) * scale[w]^2 * W[w] with the step's own.  0 switches it off
 * (every preset of the reference).  Costs three small launches per step (count the block's words, per-word squares, one
 * ordered sum into the loss); all tower kinds, all schemes, the data-parallel gradient export included. */
int nncf_tower_set_c_reg(nncf_tower* t, float c_reg);
/* Embedding(dropout=rate) of Keras 1 (mean_pool.py:56-59; 0.5 in the 'best' preset,
 * basic_embedding_conf.py:112): per train-mode forward a fresh mask drops whole word rows
 * (survivors scaled by 1/(1-rate)).  The mask is a counter-based hash of (seed, forward index,
 * word) evaluated inside the kernels; dropped rows are neither loaded nor updated.  Resets the
 * forward index to 0.  nncf_tower_dropout_scale writes the per-word scale (0 or 1/(1-rate)) the
 * step_index-th train-mode forward since then uses / used -- for checkers. */
int nncf_tower_set_dropout(nncf_tower* t, float rate, uint64_t seed);
/* Optimiser of the tower's parameters, modes as nncf_model_set_optimizer.  d_m_words / d_v_words:
 * shaped like the word table; d_m_dense / d_v_dense: word_dim*item_dim + 384 floats (dense_w, then
 * dense_b, bn_gamma, bn_beta at 128-float pitch), all zero-initialised.  Mode 1 steps only the
 * word rows that received a gradient; the Dense / BatchNorm parameters always take full Adam
 * steps.  CNN tower: the dense arrays continue with filter_len*word_dim*conv_filters + 384 floats
 * (conv_w, then conv_b, bn1_gamma, bn1_beta at 128-float pitch); dense_w is conv_filters*item_dim.  Inside nncf_tower_step / _epoch the model's step counter supplies t. */
int nncf_tower_set_optimizer(nncf_tower* t, int32_t mode, float beta1, float beta2, float epsilon, float* d_m_words,
                             float* d_v_words, float* d_m_dense, float* d_v_dense);
int nncf_tower_dropout_scale(nncf_tower* t, uint64_t step_index, float* d_out, void* stream);
/* CNN tower: the Dropout(conv_dropout) mask (0 or 1/(1-rate)) of the step_index-th train-mode
 * forward since nncf_tower_set_dropout, for the first m rows: d_out[m][content_len][conv_filters]. */
int nncf_tower_conv_mask(nncf_tower* t, uint64_t step_index, int64_t m, float* d_out, void* stream);

/* Data-parallel training (one process per GPU, tower replicated): while d_grad is non-NULL the
 * backward applies nothing and writes the step's gradients to d_grad -- nncf_tower_grad_floats()
 * floats: [words (word_count*word_dim) | dense_w (H*item_dim, H = conv_filters or word_dim) | dense_b,
 * bn_gamma, bn_beta (128 each)] and, for the CNN tower, [conv_w (filter_len*word_dim*conv_filters) |
 * conv_b, bn1_gamma, bn1_beta (128 each)]; every live element written, zeros for words outside the
 * batch; the pads of the 128-float vectors are left alone (hand in a zeroed buffer).  After the all-reduce,
 * nncf_tower_apply_grads takes the optimiser step with scale * d_grad (step_t: 1-based Adam step,
 * 0 = the tower's own counter).  d_grad: 16-byte aligned (NNCF_ERR_INVALID otherwise). */
int64_t nncf_tower_grad_floats(const nncf_tower* t);
int nncf_tower_set_grad_export(nncf_tower* t, float* d_grad);
int nncf_tower_apply_grads(nncf_tower* t, const float* d_grad, float scale, int64_t step_t, void* stream);

/* d_out[m][item_dim] = tower(d_item_ids).  train == 0: inference phase (running statistics, any
 * m).  train != 0: batch statistics over the first *d_n_valid rows (NULL = all m; rows past it
 * come out zero), m <= max_rows; prepare_backward != 0 also lays out what nncf_tower_backward needs, MUST then be followed
 * by it, and moves the running statistics (a forward that prepares no backward is a loss-only evaluation and leaves them
 * alone -- the rule of every step entry point: update == 0 changes nothing).  d_row_scale: an explicit per-word scale
 * overriding the built-in row-dropout mask, or NULL. */
int nncf_tower_forward(nncf_tower* t, const int32_t* d_item_ids, int64_t m, const int32_t* d_n_valid, int train,
                       int prepare_backward, const float* d_row_scale, float* d_out, void* stream);
/* d_dA[m][item_dim] = dL/d(out) of the last prepared forward: SGD step on words, dense_w,
 * dense_b, bn_gamma, bn_beta (fixed summation order, no atomics). */
int nncf_tower_backward(nncf_tower* t, const float* d_dA, void* stream);

/* One training step of basic_embedding (scheme 0 original: d_cid[b(1+k)] positives then
 * negatives; 1 neg_shared, 2 group_neg_shared: d_cid[b]): duplicate plans + tf.unique ->
 * tower forward of the unique items into the model's item table (rows = tf.unique ranks) ->
 * step kernels (user rows updated, item gradients exported: the model must have
 * nncf_model_set_item_grad_export set) -> tower backward.  update == 0: loss only.
 * nncf_tower_epoch: the trainer's inner loop, n_steps consecutive batches, plans made 64 steps
 * at a time; d_loss[n_steps]; d_n_u[n_steps] (optional) = unique items per step. */
int nncf_tower_step(nncf_tower* t, nncf_model* m, int32_t scheme, const int32_t* d_uid, const int32_t* d_cid,
                    float* d_loss, int update, const float* d_row_scale, void* stream);
int nncf_tower_epoch(nncf_tower* t, nncf_model* m, int32_t scheme, const int32_t* d_uid_all,
                     const int32_t* d_cid_all, const int32_t* d_neg_all, int64_t n_steps, float* d_loss,
                     int32_t* d_n_u, void* stream);
/* Data-parallel form of nncf_tower_epoch (no reference counterpart: /root/reference/main.py:34-35 is single-device;
 * SURVEY.md 8(e)): the tower has a gradient export buffer (nncf_tower_set_grad_export); per step the library runs the planned
 * step, all-reduces the flat gradient over `comm` (nncf_comm_*: RCCL or the callback transport; NULL = one rank) on `stream`
 * and takes the optimiser step with scale * gradient (nncf_tower_apply_grads) -- one native call per epoch instead of three
 * ctypes calls per step. */
struct nncf_comm;   /* the transport object of the multi-GPU trainers, declared with its entry points below */
int nncf_tower_epoch_dp(nncf_tower* t, nncf_model* m, int32_t scheme, struct nncf_comm* comm, const int32_t* d_uid_all,
                        const int32_t* d_cid_all, const int32_t* d_neg_all, int64_t n_steps, float* d_loss, float scale,
                        void* stream);
/* The epoch runner in two halves, for callers that put something between the steps (the
 * data-parallel path: all-reduce + nncf_tower_apply_grads after every step): plan up to 64
 * consecutive steps in one launch (d_uid[n_steps*b], d_cid[n_steps*b], d_neg[n_steps*b*k] for
 * scheme 0), then run planned step `index` (d_uid = that step's b user ids). */
int nncf_tower_plan_steps(nncf_tower* t, nncf_model* m, int32_t scheme, const int32_t* d_uid, const int32_t* d_cid,
                          const int32_t* d_neg, int64_t n_steps, void* stream);
int nncf_tower_step_at(nncf_tower* t, nncf_model* m, int32_t scheme, int64_t index, const int32_t* d_uid,
                       float* d_loss, void* stream);
/* The same for row-wise batches above a text tower -- n = b(1+k) independent (user, item, label)
 * rows in any order, what train_scheme = presample / group_sample / reverse feed to
 * model.train_on_batch with model_choice = cnn_embedding / basic_embedding
 * (models/train_presample.py:48-85, train_group_sample.py:75-88, train_reverse.py:59-68;
 * scripts/demos/run_presample_random.sh, run_group_sample.sh): d_uid[n], d_cid[n], d_y[n] as
 * nncf_step_rowwise takes them; pointwise losses only. */
int nncf_tower_step_rowwise(nncf_tower* t, nncf_model* m, const int32_t* d_uid, const int32_t* d_cid,
                            const int32_t* d_y, float* d_loss, int update, const float* d_row_scale, void* stream);
int nncf_tower_epoch_rowwise(nncf_tower* t, nncf_model* m, const int32_t* d_uid_all, const int32_t* d_cid_all,
                             const int32_t* d_y_all, int64_t n_steps, float* d_loss, void* stream);

/* Two-phase step for models whose item rows are produced per batch (text towers):
 *   nncf_plan_item_block: plans the batch (scheme 0 original: d_cid[b(1+k)]; 1 neg_shared,
 *     2 group_neg_shared: d_cid[b]); d_uniq_global[len(d_cid)] = the batch's unique item ids in
 *     tf.unique order (padding past *d_n_u points at id 0), *d_n_u = their number.  Batch
 *     positions are re-indexed to ranks, i.e. to rows of the item block.
 *   (caller fills rows [0, n_u) of the model's item table with the tower output for those ids)
 *   nncf_step_planned: the step kernels on (user table, item block) with the stored plans. */
int nncf_plan_item_block(nncf_model* m, int32_t scheme, const int32_t* d_uid, const int32_t* d_cid,
                         int32_t* d_uniq_global, int32_t* d_n_u, void* stream);
int nncf_step_planned(nncf_model* m, int32_t scheme, const int32_t* d_uid, float* d_loss, int update, void* stream);

/* Item-gradient export for towers that sit above the item "table" (text towers, sharded rows):
 * while d_item_grad is non-NULL every nncf_step_* leaves the item rows untouched and writes
 * dL/d(item row) (after the normalisation backward, before the learning rate) to
 * d_item_grad[row id][dim] for each row id that occurs in the batch.  NULL restores updates. */
int nncf_model_set_item_grad_export(nncf_model* m, float* d_item_grad);

/* Device-side stratified shuffle of a link list (group_shuffle_train, configs/data_utils.py:218-241;
 * "group_shuffling_trick"): rows grouped by item (by_user = 0) or user (1), groups in random order,
 * rows inside a group in random order; chop > 0: the first floor(n/chop)*chop rows are then cut into
 * runs of `chop` rows shuffled as units, the tail stays last.  Randomness: counter-based hash of
 * (seed, index) -- the reference's distribution, not NumPy's stream.  d_perm_out (optional):
 * source row of every output row.  d_ws: nncf_group_shuffle_ws_bytes(n) bytes of scratch.
 * n < 2^31.  Inputs and outputs must not alias. */
int64_t nncf_group_shuffle_ws_bytes(int64_t n);
int nncf_group_shuffle(const int32_t* d_uid, const int32_t* d_cid, int64_t n, int32_t by_user, int32_t chop,
                       uint64_t seed, int32_t* d_uid_out, int32_t* d_cid_out, int32_t* d_perm_out, void* d_ws,
                       int64_t ws_bytes, void* stream);

/* Device GroupSampler.sample (configs/data_utils.py:244-330; models/train_group_neg_shared.py:24-27,56 with
 * conf.group_shuffling_trick = False): d_groups[n_groups] = groups drawn from the B1 sampler over the group column
 * (nncf_sampler_sample_dev: the reference's stream), then `chop` members of each group uniformly with replacement --
 * member_ptr[groups + 1] / members[] = CSR of the positive members of every group in train order (:300-303).
 * Member picks: counter-based hash of (seed, output row) -- np.random.choice's distribution, not NumPy's stream.
 * Output row i (i < n_groups * chop) = (member, group i / chop); by_user != 0 swaps the roles. */
int nncf_group_sample(const int32_t* d_groups, int64_t n_groups, int32_t chop, const int32_t* d_member_ptr,
                      const int32_t* d_members, uint64_t seed, int32_t by_user, int32_t* d_uid_out, int32_t* d_cid_out,
                      void* stream);
/* Device GroupSampler.sample_with_negs (configs/data_utils.py:335-408, used by models/train_group_sample.py:86-88 when
 * conf.group_shuffling_trick is False), n_steps batches per call: per batch ceil(b / chop) groups (more, in the reference's
 * rounds :376-390, while the batch is short), `chop` positive members per group, random_rounding(k * chop * p_n/p_d[group])
 * negative members per group, rows = [positives ; negatives] cut to b * (1 + k), labels 1 / neg_sign.
 * d_groups [n_steps][gcap]: the group sampler's draws (its stream, gcap per batch); d_neg_members [n_steps][ncap]: the
 * member sampler's draws, consumed in batch row order; member picks and the rounding uniforms are counter-based hashes of
 * (seed, position) -- np.random.choice's / random.random()'s distributions, not their streams.  d_p_n_div_p_d: per group
 * id (double; NULL = 1, neg_dist 'unigram').  d_used_out[t] = groups batch t consumed, -1 if gcap / ncap did not suffice
 * (the reference asserts after 10 rounds).  gcap <= 4096. */
int nncf_group_sample_with_negs(const int32_t* d_groups, const int32_t* d_neg_members, int64_t n_steps, int32_t gcap,
                                int32_t ncap, int32_t batch_size_p, int32_t k, int32_t chop, const int32_t* d_member_ptr,
                                const int32_t* d_members, const double* d_p_n_div_p_d, uint64_t seed, int32_t by_user,
                                int32_t neg_sign, int32_t* d_uid_out, int32_t* d_cid_out, int32_t* d_y_out,
                                int32_t* d_used_out, void* stream);

/* Evaluator whole@k for one candidate set (utils/objectives.py:296-370,395-421 +
 * utils/metrics_ranking.py:6-35 eval_multiple).  d_scores: n_users rows of n_cand scores
 * (row_stride floats apart, e.g. from nncf_predict_scores); truth in CSR over candidate slots
 * (d_truth_ptr[n_users+1], d_truth_idx sorted ascending inside a user).  Per user with >= 1 true
 * item: d_metrics[3u..] = {map@k, recall@k, precision@k}, d_valid[u] = 1; others d_valid[u] = 0
 * (objectives.py:316).  Score ties go to the lower slot (the reference uses random noise).
 * Limits: topk <= 1024; any n_cand (the row is read twice, only candidates of the top k are held on chip). */
int nncf_eval_topk(const float* d_scores, int64_t n_users, int64_t row_stride, int32_t n_cand,
                   const int32_t* d_truth_ptr, const int32_t* d_truth_idx, int32_t topk, float* d_metrics,
                   int32_t* d_valid, void* stream);

/* The same evaluation without the score matrix (round 5): test_eval_mat + evaluate_mat (utils/objectives.py:296-370) and
 * eval_multiple (utils/metrics_ranking.py:6-35) in ONE kernel.  Users d_uid[n_users] (rows of the model's user table) are
 * scored against the candidate items d_cand[n_cand] (rows of its item table; slot j of the truth CSR = d_cand[j]) with the
 * arithmetic of nncf_predict_scores -- normalisation, InteractionDot bias, fp32 matrix products in the same order -- and every
 * user keeps its k best in on-chip lists: d_metrics / d_valid come out bit-identical to nncf_predict_scores followed by
 * nncf_eval_topk (ties to the lower slot included), nothing of size users x candidates is ever stored, any n_cand.
 * Limits: topk <= 100 (beyond: the two calls above). */
int nncf_eval_whole(nncf_model* m, const int32_t* d_uid, int64_t n_users, const int32_t* d_cand, int32_t n_cand,
                    const int32_t* d_truth_ptr, const int32_t* d_truth_idx, int32_t topk, float* d_metrics,
                    int32_t* d_valid, void* stream);

/* Row movers and routing of the sharded path (no reference counterpart).
 * nncf_shard_route: for a batch's item ids d_cid[n], with item rows owned by rank id / item_shard:
 *   d_req[n_ranks][cap] = the unique ids to request from each owner (ascending, -1 padded) and
 *   d_slot_of_pos[n] = request slot (owner * cap + k) holding the row of batch position i.
 * nncf_gather_rows: d_out[i][0..dim) = table row (d_ids[i] - id_offset) (zeros where d_ids[i] < 0).
 * nncf_scatter_apply: table[d_ids[i] - id_offset] -= lr * d_rows[i] for every i with d_ids[i] >= 0;
 *   rows of the same id are summed in list order first (deterministic, no atomics).  n <= 8192.
 * d_plan_ws: nncf_scatter_apply_ws_ints(n) int32 of 16-byte aligned scratch. */
int nncf_shard_route(const int32_t* d_cid, int32_t n, int32_t item_shard, int32_t n_ranks, int32_t cap,
                     int32_t* d_req, int32_t* d_slot_of_pos, int32_t* d_plan_ws, void* stream);
int nncf_gather_rows(const float* d_table, int32_t row_stride, int32_t dim, const int32_t* d_ids, int64_t n,
                     int32_t id_offset, float* d_out, void* stream);
int nncf_scatter_apply(float* d_table, int32_t row_stride, int32_t dim, const int32_t* d_ids, int32_t n,
                       int32_t id_offset, const float* d_rows, float lr, int32_t* d_plan_ws, void* stream);
int64_t nncf_scatter_apply_ws_ints(int32_t n);
/* nncf_scatter_merge: the same update for request blocks d_req[n_ranks][cap] (each block ascending in id, -1 padded at
 *   its tail, an id at most once per block) and their gradient rows d_rows[n_ranks * cap][dim]: rows of the same id are
 *   summed in rank order and the table row is stepped once -- no plan, no scratch, any n_ranks * cap. */
int nncf_scatter_merge(float* d_table, int32_t row_stride, int32_t dim, const int32_t* d_req, int32_t n_ranks, int32_t cap,
                       int32_t id_offset, const float* d_rows, float lr, void* stream);

/* Transport of the multi-GPU trainers (no reference counterpart: /root/reference/main.py:34-35 selects ONE device;
 * contract = SURVEY.md 8(e)): a communicator over RCCL -- librccl.so.1 is opened at run time, so the process shares the
 * copy torch.distributed already holds -- with one process per GPU.  Bootstrap: rank 0 draws the 128-byte unique id
 * (nncf_comm_unique_id), the caller broadcasts it (torch.distributed, MPI, a file), every rank calls nncf_comm_create.
 * All operations are enqueued on the caller's stream.
 *   nncf_comm_all_to_all      variable-split all-to-all: peer r is sent d_send[sum_{q<r} send_cnt[q] ..+send_cnt[r]) and
 *                             fills d_recv[sum_{q<r} recv_cnt[q] ..+recv_cnt[r]); counts in elements of elem_bytes bytes
 *                             (host arrays of n_ranks); grouped ncclSend / ncclRecv, the rank's own share by a
 *                             device-to-device copy (NNCF_COMM_SELF_VIA_RCCL=1: through RCCL too -- single-GPU tests)
 *   nncf_comm_all_reduce_f32  in-place sum
 *   nncf_comm_create_callback the same interface over a host callback (tests: gloo through Python on a CPU box or with
 *                             two ranks on one GPU); the callback returns 0 on success; elem_bytes == 0 marks an
 *                             all-reduce of send_cnt[0] floats in place
 *   nncf_comm_wire_bytes      payload bytes handed to the transport so far (own share included) */
typedef struct nncf_comm nncf_comm;
typedef int (*nncf_xfer_fn)(void* user, const void* d_send, const int64_t* send_cnt, void* d_recv, const int64_t* recv_cnt,
                            int32_t elem_bytes, int32_t n_ranks, void* stream);
int nncf_comm_unique_id(void* out_id128);
int nncf_comm_create(const void* id128, int32_t n_ranks, int32_t rank, int device, nncf_comm** out);
int nncf_comm_create_callback(nncf_xfer_fn fn, void* user, int32_t n_ranks, int32_t rank, nncf_comm** out);
int nncf_comm_destroy(nncf_comm* c);
int nncf_comm_all_to_all(nncf_comm* c, const void* d_send, const int64_t* send_cnt, void* d_recv, const int64_t* recv_cnt,
                         int32_t elem_bytes, void* stream);
int nncf_comm_all_reduce_f32(nncf_comm* c, float* d_buf, int64_t n, void* stream);
int nncf_comm_wire_bytes(nncf_comm* c, int64_t* out, int32_t reset);

/* Chunked sharded neg_shared trainer (no reference counterpart: /root/reference/main.py:34-35 selects ONE device;
 * SURVEY.md 8(e), BASELINE configs[3]).  One object per rank around a model whose "item table" is the landing buffer of
 * fetched item rows (batch_size_p rows) and this rank's shard of the real item table (rows [item_offset, item_offset +
 * shard rows), owner of an id = id / item_shard).  Two forms:
 *   weak   (d_user_shard == NULL): every rank trains its own batch of batch_size_p links per step -- users and their
 *          links live with their owner, the model's user table IS this rank's user shard, only item rows travel;
 *   strong (d_user_shard != NULL): ONE global batch of batch_size_p links per step over all ranks (the result of the
 *          one-device step on the same batch, models/train_neg_shared.py:46-51): rank g holds rows g, g + n_ranks, ... of
 *          the b x b score matrix; the model's user table is a landing buffer too (batch_size_p / n_ranks rows); the
 *          item rows of all b columns and the user rows of the local positions are fetched, both gradients go back to
 *          the owners, which add the ranks' partial rows in rank order.  Pointwise losses.
 * Ids do not depend on the training state, so for up to 64 steps at a time
 *   nncf_shard_plan        routes and duplicate plans of the chunk, batched: landing slot of every position (= rank of its
 *                          id among the step's unique ids, ascending), the COMPACT request lists req_out (per owner, per
 *                          step: exactly the unique ids that live there) and their sizes cnt_out[owner][exchange][64].
 *                          d_uid: weak [n_steps][b] ids relative to this rank's user range; strong [n_steps][b / n_ranks]
 *                          global user ids of this rank's positions.  d_cid: [n_steps][b] global item ids (strong: of
 *                          ALL positions).
 *   (host: static all-to-all cnt_out -> cnt_in; read both; variable-split all-to-all req_out -> req_in per exchange)
 *   nncf_shard_plan_owner  offsets of the received lists; gathers the rows of step 0 into rows_out.  rows_per_step
 *                          [n_exchanges][n_steps] (host): rows this rank's shard is asked for per step = sum over
 *                          requesters of cnt_in (may be NULL with one rank).
 * and per step i only
 *   (host: all-to-all rows_out -> rows_in, splits = the step's counts)
 *   nncf_shard_step        the local step on the landed rows: loss (strong: this rank's share -- the step's loss is the
 *                          sum over ranks), gradients exported per landing slot to g_out (weak: user rows stepped in place)
 *   (host: all-to-all g_out -> g_in)
 *   nncf_shard_apply       the owners' step on (gradient rows of id summed in rank order): table[id] -= lr * sum (sparse
 *                          SGD), or the model's Adam (nncf_model_set_optimizer: lazy-sparse on the touched rows / Keras
 *                          dense with the zero-gradient step on every other row of the shard) with the shard's own state
 *                          (nncf_shard_set_optimizer); advances the optimiser's step counter; then gathers the rows of
 *                          step i + 1 (after the update: synchronous steps).
 * With n_ranks == 1 both ends of every exchange are the same memory: nothing is copied (NNCF_SHARD_NOALIAS=1 in the
 * environment at nncf_shard_create keeps them apart, so that ONE rank drives the real collectives: single-GPU tests).
 * nncf_shard_counts / nncf_shard_buffers return the device pointers the host hands to its collectives (exchange 0:
 * items, 1: users in the strong form).
 *   nncf_shard_run_chunk   all of the above for up to 64 steps in ONE native call over a communicator
 *                          (nncf_shard_set_comm): plan -> counts all-to-all -> one host read of both count tables (the
 *                          chunk's only synchronisation) -> ids all-to-all -> per step: rows all-to-all, local step,
 *                          gradients all-to-all, owners' apply -- every exchange enqueued on `stream` between the kernels
 *                          it feeds.  d_uid / d_cid as nncf_shard_plan; d_losses [n_steps] (or NULL). */
typedef struct nncf_shard nncf_shard;
int nncf_shard_create(nncf_model* m, float* d_item_shard, int32_t item_stride, int32_t item_shard, int32_t item_offset,
                      float* d_user_shard, int32_t user_stride, int32_t user_shard, int32_t user_offset, int32_t n_ranks,
                      int32_t rank, int device, nncf_shard** out);
int nncf_shard_destroy(nncf_shard* h);
/* Adam on the sharded paths: moment arrays of this rank's shard of `exchange` (0 items, 1 users of the strong form), shaped
 * and strided like the shard; d_stamp [shard rows] int32 zeros for Keras (dense) Adam, NULL for the lazy-sparse form.
 * Required before the first step when the model was given an Adam optimiser (the weak form's local user rows use the
 * model's own state).  The as-written duplicate rule of the sparse Adam is refused at nncf_shard_create. */
int nncf_shard_set_optimizer(nncf_shard* h, int32_t exchange, float* d_m, float* d_v, int32_t* d_stamp);
/* InteractionDot bias (/root/reference/modules/interaction/interaction_dot.py:62-71,96-107) on the same-global-batch form: a model
 * with bias vectors (nncf_model_set_bias, indexed by landing slot: user_count / n_items >= batch_size_p) makes the session run
 * REPLICATED for every loss -- all 2 b rows AND their bias values to every rank, the whole one-device step (tile + apply +
 * bias update) on the landing buffers, the owners write rows and bias values back.  d_*_bias_shard: this rank's [user_shard] /
 * [item_shard] floats, one per bias vector the model has (NULL for the other).  nncf_shard_bias_buffers: both ends of the
 * bias exchange of `exchange` for a host-driven transport (NULL / NULL without a bias; with one rank both ends are the model's
 * vector).
 * WEAK form (round 6): the user bias vector is the model's own (local users, stepped in place: pass NULL); the item bias values
 * travel with the item rows (d_item_bias_shard: this rank's [item_shard] floats), land in the model's item-bias vector (indexed
 * by landing slot, >= batch_size_p values), the step exports d(loss)/d(cbias) per landing slot instead of stepping it, the
 * values travel back next to the gradient rows and the owner adds the ranks' values of an id in rank order and takes the
 * sparse-SGD step (Adam with a bias is refused on this form too).  nncf_shard_bias_grad_buffers: both ends of that extra exchange
 * for a host-driven transport (NULL / NULL unless the weak session has an item-bias shard). */
int nncf_shard_set_bias(nncf_shard* h, float* d_user_bias_shard, float* d_item_bias_shard);
int nncf_shard_bias_grad_buffers(nncf_shard* h, int32_t exchange, float** gb_out, float** gb_in);
/* Adam on the replicated form (pairwise losses; no bias): the moment rows of the requested ids travel with the table rows --
 * owners gather them from the shards' moment arrays (nncf_shard_set_optimizer), they land in the model's own moment arrays
 * (indexed by landing slot), the one-device step updates them, the owners write them back; Keras' dense Adam: the owned rows
 * the step touched carry its stamp in the shard, the rest of the shard takes the zero-gradient step.  Both ends of the two
 * extra exchanges of `exchange` for a host-driven transport (all NULL unless the session is replicated, the model has an Adam
 * optimiser and the shard its moment arrays; one rank: both ends are the model's arrays). */
int nncf_shard_moment_buffers(nncf_shard* h, int32_t exchange, float** m_out, float** m_in, float** v_out, float** v_in);
int nncf_shard_bias_buffers(nncf_shard* h, int32_t exchange, float** bias_out, float** bias_in);
int nncf_shard_counts(nncf_shard* h, int32_t** cnt_out, int32_t** cnt_in, int32_t* n_exchanges);
int nncf_shard_buffers(nncf_shard* h, int32_t exchange, int32_t** req_out, int32_t** req_in, float** rows_out, float** rows_in,
                       float** g_out, float** g_in, int32_t* ids_per_step);
int nncf_shard_plan(nncf_shard* h, const int32_t* d_uid, const int32_t* d_cid, int64_t n_steps, void* stream);
int nncf_shard_plan_owner(nncf_shard* h, int64_t n_steps, const int64_t* rows_per_step, void* stream);
int nncf_shard_step(nncf_shard* h, int64_t index, const int32_t* d_uid, float* d_loss, void* stream);
int nncf_shard_apply(nncf_shard* h, int64_t index, int64_t n_steps, float lr, void* stream);
int nncf_shard_set_comm(nncf_shard* h, nncf_comm* c);
int nncf_shard_run_chunk(nncf_shard* h, const int32_t* d_uid, const int32_t* d_cid, int64_t n_steps, float* d_losses, float lr,
                         void* stream);

/* Duplicate-row plan of one id list (exposed for tests): positions sorted by (id, position),
 * segment starts, and tf.unique-style first-occurrence outputs.
 *   d_order[n], d_seg_start[n+1], d_uniq[n], d_inverse[n],
 *   d_counts[4] = {n_segments, n, n_chunks, 0},
 *   d_desc[4n] = per segment {id, first position, start in d_order, count} (16-byte aligned),
 *   d_slot_seg[n] = segment of every sorted slot,
 *   d_chunk_start[n/16 + 2] = slot ranges of whole segments handed to one apply workgroup.
 * n <= 32768 (the per-step capacity of a model: batch_size_p * (1 + num_negatives)).  Up to 8192 entries everything is sorted
 * and scanned in LDS; above, the per-entry arrays are global; above 16384 the sort keys too, sorted through one 16384-key LDS
 * tile (this stand-alone entry allocates that scratch for the call and waits for the stream). */
int nncf_plan_ids(const int32_t* d_ids, int32_t n, int32_t* d_order, int32_t* d_seg_start, int32_t* d_uniq,
                  int32_t* d_inverse, int32_t* d_counts, int32_t* d_desc, int32_t* d_slot_seg,
                  int32_t* d_chunk_start, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NNCF_HIP_H_ */
