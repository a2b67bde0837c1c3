// Kernel argument blocks and workspaces of the step kernels.
#pragma once
#include <cstdint>
#include <vector>

#include "model.h"

namespace nncf {

// Adam state of one table: first / second moments (table-shaped) and, for dense Adam, the step
// at which each row last received a gradient.
struct AdamTable {
  float* m = nullptr;
  float* v = nullptr;
  int32_t* stamp = nullptr;
};

// Per-launch scalars (copied by value into the kernarg segment).
struct StepArgs {
  float* U;  // user table  (user_count x stride)
  float* V;  // item table  (item_count x stride)
  int stride, d;
  int b, k;
  int loss, norm_u, norm_v;
  float gamma, lambda, u_reg, lr;
  float* gV;  // if non-NULL the item rows are NOT updated: dL/dV rows are written to gV[row id][d]
              // (sharded mode: V is a buffer of fetched rows, gradients go back to their owners)
  float* gU = nullptr;   // the same for the user rows (same-global-batch multi-GPU step: both tables are landing buffers)
  float* gcb = nullptr;  // if non-NULL the item bias is NOT stepped: d(loss)/d(cbias) per item-table row is written here (the weak
                         // sharded form: the model's item rows AND bias values are landing slots, their owners take the step)
  // Rectangular neg_shared step (one-launch path only; no reference counterpart -- models/train_neg_shared.py:46-51 is one
  // b x b matrix on one device): bc != 0 -> this launch owns the b rows row0, row0 + row_step, ... of a bc x bc score
  // matrix.  uid[b] lists its positions' users, cid[bc] and the item plan all bc columns; the positive of local row i is
  // column row0 + i row_step; the loss is normalised by bc, negatives weigh lambda / (bc - 1).
  int bc = 0, row0 = 0, row_step = 1;
  // text tower: item row = relu(bn_z[row] * mul[c] + add[c]) -- BatchNorm + ReLU applied on load
  // from the pre-activation block, the normalised block is never materialised.  NULL = V.
  // bn_stats: five 128-float vectors mu, inv, var, mul, add.
  const float* bn_z = nullptr;
  const float* bn_stats = nullptr;
  // if non-NULL the apply kernel of the shared schemes also emits, per item chunk, the column sums
  // of dy = dL/dE * [E > 0] and dy * xhat (the BatchNorm backward's reductions): [chunk][2][128]
  float* bn_part = nullptr;
  // text tower, blocks of more than 1,024 rows (`original`): the statistics launch leaves bn_nparts partial rows
  // (shifted sums, shift = block row 0); every workgroup of the score kernel merges them itself and workgroup 0 publishes
  // mu / inv / var / mul / add (bn_stats) + the running statistics -- no one-workgroup merge launch in between
  const float* bn_parts = nullptr;   // [bn_nparts][2][128]
  int bn_nparts = 0;
  int bn_update = 0;                 // move the running statistics (an update step; not a loss-only evaluation)
  float* bn_pub = nullptr;           // = bn_stats, writable
  const float* bn_gamma = nullptr;
  const float* bn_beta = nullptr;
  float* bn_rm = nullptr;
  float* bn_rv = nullptr;
  float bn_eps = 1e-3f, bn_mom = 0.99f;
  const int32_t* bn_nv = nullptr;    // rows of the block (device scalar)
  // optimiser: 0 sparse SGD (north-star); 1 lazy-sparse Adam on the touched rows
  // (utils/optimizer.py:108-134); 2 dense Adam = Keras Adam (basic_embedding_conf.py:59-60): the
  // touched rows here, every other row by adam_idle_kernel with a zero gradient.
  int opt = 0;
  // opt == 1 only: 0 = duplicate ids of a batch summed first (TF's _apply_sparse_duplicate_indices; default),
  // 1 = utils/optimizer.py:108-134 as written for an IndexedSlices with duplicates: moments from the LAST
  // occurrence's gradient (scatter_update: last writer wins), the step subtracted once per occurrence (scatter_sub)
  int dup_as_written = 0;
  float lr_t = 0.f, beta1 = 0.9f, beta2 = 0.999f, eps = 1e-8f;  // lr_t = lr sqrt(1-b2^t)/(1-b1^t)
  int step_no = 0;
  AdamTable au, av;
  // InteractionDot bias (interaction_dot.py:62-71,96-107): score += ub[user] + cb[item]; NULL = none.
  // item_gid: global item id of item-table row i when the "table" is a per-batch block (text
  // towers); NULL = i.  abu / abc: Adam state of the two bias vectors.
  float* ub = nullptr;
  float* cb = nullptr;
  const int32_t* item_gid = nullptr;
  AdamTable abu, abc;
};

}  // namespace nncf
#ifdef __HIPCC__
#include "devutil.h"
namespace nncf {
// row `id` of the step's item table (plain table, or BatchNorm + ReLU of the tower's block)
__device__ __forceinline__ Row item_row(const StepArgs& a, int64_t id, int d4, int q) {
  return a.bn_z ? row_load_bn(a.bn_z, a.bn_stats + 384, id, a.stride, d4, q) : row_load(a.V, id, a.stride, d4, q);
}
// Text tower above the item "table": the apply kernels read the pre-activation rows, apply
// BatchNorm + ReLU themselves and -- the BatchNorm backward's reductions riding along -- emit per
// workgroup the column sums of dy = dL/dE * [E > 0] and dy * xhat over the rows they finalise.
struct BnTail {
  bool on, sums;
  Row mul, add, mu, inv, s1, s2;
};
__device__ __forceinline__ void bn_tail_init(BnTail& t, const StepArgs& a, int q) {
  t.on = a.bn_z != nullptr;
  t.sums = t.on && a.bn_part != nullptr;
  t.mul = t.add = t.mu = t.inv = t.s1 = t.s2 = row_zero();
  if (t.on) {
    t.mu = row_load(a.bn_stats, 0, 128, 32, q);
    t.inv = row_load(a.bn_stats + 128, 0, 128, 32, q);
    t.mul = row_load(a.bn_stats + 384, 0, 128, 32, q);
    t.add = row_load(a.bn_stats + 512, 0, 128, 32, q);
  }
}
__device__ __forceinline__ Row bn_tail_row(const BnTail& t, Row loaded) {
  if (!t.on) return loaded;
  Row v;
  v.a = bn_relu4(loaded.a, t.mul.a, t.add.a);
  v.b = bn_relu4(loaded.b, t.mul.b, t.add.b);
  return v;
}
__device__ __forceinline__ void bn_tail_acc4(float4 y, float4 g, float4 z, float4 mu, float4 iv, float4& t1, float4& t2) {
  const float4 dy = make_float4(y.x > 0.f ? g.x : 0.f, y.y > 0.f ? g.y : 0.f, y.z > 0.f ? g.z : 0.f, y.w > 0.f ? g.w : 0.f);
  t1 = add4(t1, dy);
  t2.x = fmaf(dy.x, (z.x - mu.x) * iv.x, t2.x);
  t2.y = fmaf(dy.y, (z.y - mu.y) * iv.y, t2.y);
  t2.z = fmaf(dy.z, (z.z - mu.z) * iv.z, t2.z);
  t2.w = fmaf(dy.w, (z.w - mu.w) * iv.w, t2.w);
}
__device__ __forceinline__ void bn_tail_acc(BnTail& t, const StepArgs&, Row vraw, Row grad, Row loaded) {
  if (!t.sums) return;
  bn_tail_acc4(vraw.a, grad.a, loaded.a, t.mu.a, t.inv.a, t.s1.a, t.s2.a);
  bn_tail_acc4(vraw.b, grad.b, loaded.b, t.mu.b, t.inv.b, t.s1.b, t.s2.b);
}
// every thread of the 256-thread workgroup calls this: the 16 groups in group order -> bn_part[chunk]
__device__ __forceinline__ void bn_tail_flush(const BnTail& t, const StepArgs& a, float* sR, int chunk) {
  if (!t.sums) return;
  const int gl = threadIdx.x >> 4, q = threadIdx.x & 15;
  float* out = a.bn_part + (int64_t)chunk * 2 * 128;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    __syncthreads();
    reinterpret_cast<float4*>(sR + gl * 128)[q] = half ? t.s2.a : t.s1.a;
    reinterpret_cast<float4*>(sR + gl * 128)[q + 16] = half ? t.s2.b : t.s1.b;
    __syncthreads();
    if (threadIdx.x < 128) {
      float v = 0.f;
      for (int g = 0; g < 16; ++g) v += sR[g * 128 + threadIdx.x];
      out[half * 128 + threadIdx.x] = v;
    }
  }
}

__device__ __forceinline__ float user_bias(const StepArgs& a, int64_t u) { return a.ub ? a.ub[u] : 0.f; }
__device__ __forceinline__ float item_bias(const StepArgs& a, int64_t i) {
  return a.cb ? a.cb[a.item_gid ? a.item_gid[i] : i] : 0.f;
}

// The optimiser step on one table row (called by one 16-lane group): raw row, complete gradient.
__device__ __forceinline__ float4 adam4(float4 g, float4& m, float4& v, float4 p, float b1, float b2, float lr_t, float eps) {
  m = make_float4(fmaf(b1, m.x, (1.f - b1) * g.x), fmaf(b1, m.y, (1.f - b1) * g.y), fmaf(b1, m.z, (1.f - b1) * g.z),
                  fmaf(b1, m.w, (1.f - b1) * g.w));
  v = make_float4(fmaf(b2, v.x, (1.f - b2) * g.x * g.x), fmaf(b2, v.y, (1.f - b2) * g.y * g.y),
                  fmaf(b2, v.z, (1.f - b2) * g.z * g.z), fmaf(b2, v.w, (1.f - b2) * g.w * g.w));
  return make_float4(p.x - lr_t * m.x / (sqrtf(v.x) + eps), p.y - lr_t * m.y / (sqrtf(v.y) + eps),
                     p.z - lr_t * m.z / (sqrtf(v.z) + eps), p.w - lr_t * m.w / (sqrtf(v.w) + eps));
}
__device__ __forceinline__ void apply_update(const StepArgs& a, const AdamTable& st, float* table, int64_t id, Row raw,
                                             Row grad, int d4, int q, float mult = 1.f) {
  if (a.opt == 0) {
    row_store(table, id, a.stride, d4, q, row_fma(-a.lr, grad, raw));
    return;
  }
  Row m = row_load(st.m, id, a.stride, d4, q), v = row_load(st.v, id, a.stride, d4, q);
  Row p;
  p.a = adam4(grad.a, m.a, v.a, raw.a, a.beta1, a.beta2, mult * a.lr_t, a.eps);   // mult: occurrences (as-written mode)
  p.b = adam4(grad.b, m.b, v.b, raw.b, a.beta1, a.beta2, mult * a.lr_t, a.eps);
  row_store(st.m, id, a.stride, d4, q, m);
  row_store(st.v, id, a.stride, d4, q, v);
  row_store(table, id, a.stride, d4, q, p);
  if (st.stamp && q == 0) st.stamp[id] = a.step_no;
}
}  // namespace nncf
#endif
namespace nncf {

// train_scheme = original
struct OriginalWs {
  float* gu;         // [b]       sum of d(loss)/d(score) over each positive's 1+k rows (user-bias gradient)
  float* g;          // [b(1+k)]  d(loss)/d(score) per row
  float* dUp;        // [b][d]    dL/dUhat summed over each positive's 1+k rows
  float* Uhat;       // [b][d]    (normalised) user row of each positive
  float* loss_part;  // [b]
  float* reg_part;   // [b]
};

// general row-wise batches (presample / group_sample / reverse)
struct RowwiseWs {
  float* g;          // [n]
  float* Urow;       // [n][d] (normalised) user row of every batch row
  float* Vrow;       // [n][d] (normalised) item row of every batch row
  float* loss_part;  // [n]
  float* reg_part;   // [n]  |U_raw|^2
};

// train_scheme = neg_shared / group_neg_shared (b x N score tile grid, 32 x 32 tiles)
struct SharedWs {
  float* dU_part;    // [NBJ][b][d]     partial dL/dUhat per column block
  float* dV_part;    // [NBI][Npad][d]  partial dL/dVhat per row block
  float* c_part;     // [NBI][Npad]     neg_shared pairwise: partial column sums of h
  float* r_part;     // [NBJ][b]        group pairwise: partial row sums of h
  float* Uhat;       // [b][d]          normalised user rows          (pairwise corrections)
  float* Vrow;       // [b][d]          normalised item row of each batch row's own item
  float* loss_part;  // [NBI*NBJ]
  float* reg_part;   // [b]             |U_raw|^2 per batch row
  float* gr_part;    // [NBJ][b]        bias: partial row sums of h (dL/dS off the pairwise corrections)
  float* gc_part;    // [NBI][Npad]     bias: partial column sums of h
  long long* dbg;    // optional [tiles][8] s_memtime stamps of the tile kernel (NNCF_DEBUG_TIMING=1)
  // text towers: with st_out set every tile workgroup derives the BatchNorm statistics of the item block itself
  // (StepArgs::bn_z, rows [0, n unique items)) and tile (0, 0) publishes mu, inv, var, mul, add there (the apply kernel and
  // the tower backward read them) + the running statistics -- no tower_stats launch in front of the tile kernel
  float* st_out = nullptr;
  const float* st_gamma = nullptr;
  const float* st_beta = nullptr;
  float* st_rm = nullptr;
  float* st_rv = nullptr;
  float st_eps = 1e-3f, st_mom = 0.99f;
  int st_update = 0;
};

int launch_original(const StepArgs& a, const int32_t* uid, const int32_t* cid_pos, const int32_t* cid_neg,
                    const OriginalWs& ws, const PlanView& pu, const PlanView& pv, float* loss_out, int update,
                    hipStream_t stream);

// group = 0: neg_shared (columns = the b batch items, positive of row i is column i)
// group = 1: group_neg_shared (columns = unique items of plan pv, positive = pv.inverse[i])
int launch_shared(const StepArgs& a, int group, const int32_t* uid, const int32_t* cid, const SharedWs& ws,
                  const PlanView& pu, const PlanView& pv, float* loss_out, int update, hipStream_t stream);

int launch_rowwise(const StepArgs& a, const int32_t* uid, const int32_t* cid, const int32_t* y, const RowwiseWs& ws,
                   const PlanView& pu, const PlanView& pv, int n, float* loss_out, int update, hipStream_t stream);

// InteractionDot bias update (bias.hip).  scheme 0 original, 1 neg_shared, 2 group_neg_shared, 3 row-wise
int launch_bias_update(const StepArgs& a, int scheme, const PlanView& pu, const PlanView& pv, const OriginalWs* ows,
                       const SharedWs* sws, const RowwiseWs* rws, hipStream_t stream);

int launch_bias_idle(const StepArgs& a, float* bias, const AdamTable& st, int64_t n, hipStream_t stream);

// dense Adam: rows of (table, state) whose stamp != a.step_no take a zero-gradient step
int launch_adam_idle(const StepArgs& a, float* table, const AdamTable& st, int64_t rows, hipStream_t stream);

int launch_predict(const StepArgs& a, const int32_t* uid, int64_t p, const int32_t* cid, int64_t q, float* out,
                   hipStream_t stream);

}  // namespace nncf

constexpr int kTileMaxBatch = 2048;   // largest batch_size_p the slab-based tile + apply path allocates for

// The model object behind the B2 entry points (include/nncf_hip.h).
struct nncf_model {
  nncf_model_cfg cfg;
  float* U = nullptr;
  float* V = nullptr;
  int device = 0;
  void* arena = nullptr;  // one device allocation, carved below
  nncf::OriginalWs ows;
  nncf::RowwiseWs rws;
  nncf::SharedWs sws;
  nncf::PlanView plan_u, plan_v;  // kPlanChunk lists each
  int64_t cap_u = 0, cap_v = 0;
  float* export_gV = nullptr;
  // optimiser (nncf_model_set_optimizer)
  int opt = 0;
  float beta1 = 0.9f, beta2 = 0.999f, eps = 1e-8f;
  int64_t adam_t = 0;  // update steps taken (Keras `iterations`)
  nncf::AdamTable au, av;
  int32_t* stamps = nullptr;  // owned: user_count + item_count ints (dense Adam)
  // InteractionDot bias (nncf_model_set_bias)
  float *ub = nullptr, *cb = nullptr;
  nncf::AdamTable abu, abc;
  int64_t bias_items = 0;     // length of cb (the catalogue, also when the item table is a per-batch block)
  int32_t* bias_stamps = nullptr;  // owned: user_count + bias_items ints (dense Adam)
  // one-launch shared-negative step (step_fused.hip): loss partials of the current / previous step and the
  // step whose loss is still to be summed (done by the next fused launch or by flush_pending_loss)
  bool tile_ok = true;        // the tile + apply workspace exists (batch_size_p <= kTileMaxBatch)
  int dup_as_written = 0;     // nncf_model_set_sparse_adam_duplicates
  int use_fused = 1;          // NNCF_SHARED_PATH=tile forces the two-launch tile + apply path
  int pair_wide = 0;          // NNCF_SHARED_PATH=pairwide: pairwise losses take the strip form (step_pair.hip) at any batch size
  float* panchor = nullptr;   // [b] positive score of every batch position (PairArgs::anchor)
  float* fpart = nullptr;     // [2][fused_max_blocks(b) * fused_split(b)]
  float* fslab = nullptr;     // [2 strips per side][split][16][128] partial output rows of split strips
  int* fcnt = nullptr;        // [2 strips per side] arrival tickets (zero between launches)
  float* fstage = nullptr;    // [2][b][128] staged gradient rows of a deferred step (FusedArgs::stage)
  int* fdone = nullptr;       // [2] workgroups that finished reading, by launch parity (FusedArgs::done)
  int* fdom = nullptr;        // [2][4] value range of the batch's rows, by launch parity (FusedArgs::dom)
  int* ferr = nullptr;        // [1] raised by a hand-shake wait that timed out (FusedArgs::err); sticky
  int shared_device = 0;      // nncf_model_set_shared_device: never wait inside a launch (stage + commit)
  long long spin_ticks = 25000000;   // bound of the in-launch wait, 100 MHz ticks (NNCF_FUSED_SPIN_MS)
  int fparity = 0;
  float* pend_out = nullptr;
  int pend_n = 0, pend_parity = 0;
};

namespace nncf {
struct FusedArgs;
// neg_shared / group_neg_shared step kernels of one batch: the one-launch path when eligible (pointwise loss,
// no bias), else tile + apply.  `tower`: statistics-fusion arguments of a text tower, or NULL.
// *fused_out (optional) tells which path ran.
int shared_step(nncf_model* m, const StepArgs& a, int group, const int32_t* uid, const int32_t* cid, const PlanView& pu,
                const PlanView& pv, float* loss_out, int update, hipStream_t st, const FusedArgs* tower = nullptr);
bool shared_step_fuses(const nncf_model* m, const StepArgs& a);
bool shared_step_pair_wide(const nncf_model* m, const StepArgs& a);   // pairwise losses: anchors -> wide strips -> commit
// sums the loss partials the last fused launch left behind (every entry point calls it before it returns)
int flush_pending_loss(nncf_model* m, hipStream_t st);
StepArgs make_args(const nncf_model* m);
// after the kernels of an update step: dense-Adam idle rows, optimiser step counter
int finish_update(nncf_model* m, const StepArgs& a, hipStream_t st);
}
