// B2 host side: model object, workspaces, per-step entry points and the epoch runners that
// replace the trainers' inner loops (models/train_original.py:43-68,
// train_neg_shared.py:46-59, train_group_neg_shared.py:53-68).
#include <cmath>
#include <cstdlib>
#include <new>
#include <string>

#include "common.h"
#include "model.h"
#include "step.h"
#include "step_fused.h"
#include "step_pair.h"

using namespace nncf;


namespace nncf {
StepArgs make_args(const nncf_model* m) {
  const nncf_model_cfg& c = m->cfg;
  StepArgs a;
  a.U = m->U;
  a.V = m->V;
  a.stride = c.row_stride;
  a.d = c.dim;
  a.b = c.batch_size_p;
  a.k = c.num_negatives;
  a.loss = c.loss;
  a.norm_u = c.normalize_user;
  a.norm_v = c.normalize_item;
  a.gamma = c.loss_gamma;
  a.lambda = c.neg_loss_weight;
  a.u_reg = c.u_reg;
  a.lr = c.learn_rate;
  a.gV = m->export_gV;
  a.opt = m->opt;
  a.dup_as_written = m->dup_as_written;
  if (m->opt) {  // the step about to be taken is number adam_t + 1
    const double t = (double)(m->adam_t + 1);
    a.beta1 = m->beta1;
    a.beta2 = m->beta2;
    a.eps = m->eps;
    a.lr_t = (float)((double)c.learn_rate * std::sqrt(1.0 - std::pow((double)m->beta2, t)) /
                     (1.0 - std::pow((double)m->beta1, t)));
    a.step_no = (int)(m->adam_t + 1);
    a.au = m->au;
    a.av = m->av;
    a.abu = m->abu;
    a.abc = m->abc;
  }
  a.ub = m->ub;
  a.cb = m->cb;
  return a;
}

// After the kernels of an update step: dense Adam moves every row that got no gradient, then the
// optimiser's step counter advances.
int finish_update(nncf_model* m, const StepArgs& a, hipStream_t st) {
  if (!m->opt) return NNCF_OK;
  if (m->opt == 2) {
    if (int rc = launch_adam_idle(a, m->U, m->au, m->cfg.user_count, st)) return rc;
    if (!m->export_gV)
      if (int rc = launch_adam_idle(a, m->V, m->av, m->cfg.item_count, st)) return rc;
    if (int rc = launch_bias_idle(a, m->ub, m->abu, m->cfg.user_count, st)) return rc;
    if (int rc = launch_bias_idle(a, m->cb, m->abc, m->bias_items, st)) return rc;
  }
  m->adam_t++;
  return NNCF_OK;
}

bool shared_step_fuses(const nncf_model* m, const StepArgs& a) {
  // the one-launch step merges the positions of a duplicated id into one row: the as-written lazy-sparse Adam
  // needs the last position's gradient on its own -> tile + apply path
  if (a.opt == 1 && a.dup_as_written) return false;
  return m->use_fused && fused_eligible(a) && (!a.bn_z || fused_max_blocks(a.b) <= 64);
}

// pairwise losses: the strip form (step_pair.hip) above the slab path's batch cap, or wherever NNCF_SHARED_PATH=pairwide asks
bool shared_step_pair_wide(const nncf_model* m, const StepArgs& a) {
  return pair_wide_eligible(a) && (!m->tile_ok || m->pair_wide);
}

int flush_pending_loss(nncf_model* m, hipStream_t st) {
  if (!m->pend_out) return NNCF_OK;
  const int mb = fused_max_blocks(m->cfg.batch_size_p) * fused_split(m->cfg.batch_size_p);
  const int rc = launch_fused_flush(m->fpart + (size_t)m->pend_parity * mb, m->pend_n, m->pend_out, m->ferr, st);
  m->pend_out = nullptr;
  return rc;
}

int shared_step(nncf_model* m, const StepArgs& a, int group, const int32_t* uid, const int32_t* cid, const PlanView& pu,
                const PlanView& pv, float* loss_out, int update, hipStream_t st, const FusedArgs* tower) {
  // the rectangular form (StepArgs::bc: a row share of a bc x bc score matrix) and the exported user gradient (gU) exist in
  // the one-launch step's 16-row strips only: the tile + apply pair would run a square b x b step over cid[0, b) and step
  // the landing buffer in place, the wide kernel ignores gU -- a silently wrong multi-GPU step
  if (a.bc || a.gU) {
    NNCF_REQUIRE(shared_step_fuses(m, a), "the rectangular / user-gradient-exporting step needs the one-launch path (pointwise loss, "
                 "no interaction bias, NNCF_SHARED_PATH unset): it has no tile + apply form");
    NNCF_REQUIRE(a.bc > 0 && a.gU != nullptr, "StepArgs::gU and StepArgs::bc go together (the strong form exports both gradients)");
    NNCF_REQUIRE(fused_row_tiles(a) == 1, "the rectangular step runs on 16-row strips only (b = %d, bc = %d)", a.b, a.bc);
  }
  if (!shared_step_fuses(m, a)) {
    if (shared_step_pair_wide(m, a)) {   // pairwise losses without slabs: anchors -> wide strips -> commit (step_pair.hip)
      if (int rc = flush_pending_loss(m, st)) return rc;
      PairArgs p;
      p.anchor = m->panchor;
      p.uhat = m->sws.Uhat;
      p.vhat = m->sws.Vrow;
      p.t_part = m->sws.c_part;      // [b / 32][b] floats carved; b / (16 MT) strips are used
      p.stage = m->fstage;
      p.loss_part = m->fpart;
      p.dom = m->fdom;
      return launch_pair_wide(a, group, uid, cid, p, pu, pv, loss_out, update, st);
    }
    if (!m->tile_ok)
      return fail(NNCF_ERR_INVALID, "batch_size_p = %d: above %d the shared-negative schemes run the one-launch step (skip-gram / "
                  "mse) or the strip form of the pairwise losses (id tables); neither takes an interaction bias, exported "
                  "gradients or the as-written sparse Adam (NNCF_SHARED_PATH unset)", a.b, kTileMaxBatch);
    if (int rc = flush_pending_loss(m, st)) return rc;
    SharedWs w = m->sws;
    if (tower && tower->stats) {   // text tower: the tile workgroups derive the BatchNorm statistics (no statistics launch)
      w.st_out = tower->stats;
      w.st_gamma = tower->gamma;
      w.st_beta = tower->beta;
      w.st_rm = tower->rm;
      w.st_rv = tower->rv;
      w.st_eps = tower->eps;
      w.st_mom = tower->momentum;
    }
    return launch_shared(a, group, uid, cid, w, pu, pv, loss_out, update, st);
  }
  const int mb = fused_max_blocks(m->cfg.batch_size_p) * fused_split(m->cfg.batch_size_p);   // as carved
  FusedArgs f = tower ? *tower : FusedArgs();
  f.split = a.bc ? fused_split_rect(a.b, a.bc) : fused_split(a.b);
  f.slab = m->fslab;
  f.cnt = m->fcnt;
  f.done = m->fdone;
  f.dom = m->fdom;
  f.err = m->ferr;
  f.spin_ticks = m->spin_ticks;
  f.shared_device = m->shared_device;
  f.stage = m->fstage;   // used only when the launcher finds the grid too large for the device
  f.parity = m->fparity;
  f.loss_part = m->fpart + (size_t)m->fparity * mb;
  f.dbg = m->sws.dbg;   // NNCF_DEBUG_TIMING=1: [2 mb + 1][32] stamps fit the tile kernel's [nb*nb][8] buffer for b >= 128
  if (m->pend_out) {
    f.prev_part = m->fpart + (size_t)m->pend_parity * mb;
    f.prev_n = m->pend_n;
    f.prev_out = m->pend_out;
  }
  if (int rc = launch_fused(a, group, uid, cid, f, pu, pv, update, st)) return rc;
  m->pend_out = loss_out;
  m->pend_n = fused_loss_parts(a);
  m->pend_parity = m->fparity;
  m->fparity ^= 1;
  return NNCF_OK;
}
}  // namespace nncf

namespace {



void carve(nncf_model* m, Carver& c) {
  const int64_t b = m->cfg.batch_size_p, k = m->cfg.num_negatives, d = m->cfg.dim;
  const int64_t n = b * (1 + k);
  const int64_t nb = (b + kTile - 1) / kTile;
  m->ows.gu = c.take<float>(b);
  m->ows.g = c.take<float>(n);
  m->ows.dUp = c.take<float>(b * d);
  m->ows.Uhat = c.take<float>(b * d);
  m->ows.loss_part = c.take<float>(b);
  m->ows.reg_part = c.take<float>(b);
  m->rws.g = c.take<float>(n);
  m->rws.Urow = c.take<float>(n * d);
  m->rws.Vrow = c.take<float>(n * d);
  m->rws.loss_part = c.take<float>(n);
  m->rws.reg_part = c.take<float>(n);
  // the tile + apply path's slabs grow with b^2 d / 32 (1 GB each at b = 8192, d = 128): large batches run the
  // one-launch step only (pointwise losses, no bias) -- see shared_step
  m->tile_ok = b <= kTileMaxBatch;
  m->sws.dU_part = c.take<float>(m->tile_ok ? nb * b * d : 4);
  m->sws.dV_part = c.take<float>(m->tile_ok ? nb * b * d : 4);
  m->sws.c_part = c.take<float>(nb * b);
  m->sws.r_part = c.take<float>(nb * b);
  m->sws.Uhat = c.take<float>(b * d);
  m->sws.Vrow = c.take<float>(b * d);
  m->sws.loss_part = c.take<float>(nb * nb);
  m->sws.reg_part = c.take<float>(b);
  m->sws.gr_part = c.take<float>(nb * b);
  m->sws.gc_part = c.take<float>(nb * b);
  m->sws.dbg = getenv("NNCF_DEBUG_TIMING") ? c.take<long long>(nb * nb * 8 + (2 * (fused_max_blocks((int)b) + 8) * kFusedMaxSplit + 1) * 32) : nullptr;
  {
    const size_t mb = (size_t)fused_max_blocks((int)b), sp = (size_t)fused_split((int)b);
    m->fpart = c.take<float>(2 * mb * sp);
    m->fslab = c.take<float>(sp > 1 ? 2 * mb * sp * 16 * 128 : 4);
    m->fcnt = c.take<int>(2 * mb);
    m->fdone = c.take<int>(2);
    m->fdom = c.take<int>(8);
    m->ferr = c.take<int>(2);
    m->fstage = c.take<float>(2 * (size_t)b * 128);
    m->panchor = c.take<float>(b);
  }
  m->cap_u = n;  // row-wise batches list a user per row
  m->cap_v = n;  // item list of `original`; the shared schemes use the first b entries
  auto plan = [&](PlanView& v, int64_t cap) {
    v.list_stride = cap;
    v.order = c.take<int32_t>(kPlanChunk * cap);
    v.seg_start = c.take<int32_t>(kPlanChunk * (cap + 1));
    v.uniq = c.take<int32_t>(kPlanChunk * cap);
    v.inverse = c.take<int32_t>(kPlanChunk * cap);
    v.counts = c.take<int32_t>(kPlanChunk * 4);
    v.desc = c.take<int32_t>(kPlanChunk * cap * 4);
    v.slot_seg = c.take<int32_t>(kPlanChunk * cap);
    v.chunk_start = c.take<int32_t>(kPlanChunk * v.chunk_stride());
    v.aux = cap > 8192 ? c.take<int32_t>(kPlanChunk * cap) : nullptr;
    v.keys = cap > 16384 ? c.take<unsigned long long>(kPlanChunk * cap) : nullptr;
  };
  plan(m->plan_u, m->cap_u);
  plan(m->plan_v, m->cap_v);
}

int check_model(const nncf_model* m) {
  if (!m) return fail(NNCF_ERR_INVALID, "model is NULL");
  return NNCF_OK;
}

int check_shared(const nncf_model* m) {
  if (m->cfg.batch_size_p % kTile != 0)
    return fail(NNCF_ERR_INVALID, "batch_size_p = %d must be a multiple of %d for the shared-negative schemes",
                m->cfg.batch_size_p, kTile);
  return NNCF_OK;
}

// Plans for steps [t0, t0+cnt) of an epoch (cnt <= kPlanChunk).
int plan_chunk(nncf_model* m, int scheme, const int32_t* uid_all, const int32_t* cid_all, const int32_t* neg,
               int64_t t0, int64_t cnt, hipStream_t stream) {
  const int b = m->cfg.batch_size_p, k = m->cfg.num_negatives;
  if (scheme == 0)
    return launch_plan_steps(uid_all + t0 * b, b, b, m->plan_u, cid_all + t0 * b, b, b, neg + t0 * b * k, (int64_t)b * k, b * k,
                             m->plan_v, nullptr, 0, nullptr, cnt, stream);
  return launch_plan_steps(uid_all + t0 * b, b, b, m->plan_u, cid_all + t0 * b, b, b, nullptr, 0, 0, m->plan_v, nullptr, 0, nullptr,
                           cnt, stream);
}

}  // namespace

extern "C" {

int nncf_model_create(const nncf_model_cfg* cfg, float* d_user_table, float* d_item_table, int device,
                      int64_t max_steps, nncf_model** out) {
  NNCF_REQUIRE(out != nullptr, "out is NULL");
  *out = nullptr;
  NNCF_REQUIRE(cfg && d_user_table && d_item_table, "NULL argument");
  NNCF_REQUIRE(cfg->batch_size_p > 0, "batch_size_p must be positive");
  NNCF_REQUIRE(cfg->num_negatives >= 0, "num_negatives must be >= 0");
  NNCF_REQUIRE(cfg->dim > 0 && cfg->dim % 4 == 0 && cfg->dim <= 128, "dim = %d must be a multiple of 4 and <= 128",
               cfg->dim);
  NNCF_REQUIRE(cfg->row_stride >= cfg->dim && cfg->row_stride % 4 == 0, "row_stride = %d must be >= dim and a multiple of 4",
               cfg->row_stride);
  NNCF_REQUIRE(cfg->loss >= 0 && cfg->loss <= 3, "[ERROR!] loss %d not specified.", cfg->loss);
  NNCF_REQUIRE((int64_t)cfg->batch_size_p * (1 + cfg->num_negatives) <= 32768,
               "batch_size_p * (1 + num_negatives) = %lld exceeds the per-step plan capacity 32768",
               (long long)cfg->batch_size_p * (1 + cfg->num_negatives));
  NNCF_REQUIRE(cfg->user_count > 0 && cfg->item_count > 0, "empty table");
  NNCF_REQUIRE(((uintptr_t)d_user_table & 15) == 0 && ((uintptr_t)d_item_table & 15) == 0, "tables must be 16-byte aligned");
  (void)max_steps;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(NNCF_ERR_HIP, "no HIP device: the step kernels have no CPU fallback");
  NNCF_REQUIRE(device >= 0 && device < ndev, "device %d out of range", device);
  nncf_model* m = new (std::nothrow) nncf_model();
  if (!m) return fail(NNCF_ERR_NOMEM, "out of host memory");
  m->cfg = *cfg;
  m->U = d_user_table;
  m->V = d_item_table;
  m->device = device;
  if (const char* e = getenv("NNCF_SHARED_PATH")) {
    m->use_fused = std::string(e) != "tile";
    m->pair_wide = std::string(e) == "pairwide";
  }
  if (const char* e = getenv("NNCF_FUSED_SPIN_MS")) {
    const double ms = atof(e);
    if (ms > 0) m->spin_ticks = (long long)(ms * 1e5);   // 100 MHz wall clock
  }
  if (const char* e = getenv("NNCF_SHARED_DEVICE")) m->shared_device = e[0] == '1';
  Carver dry;
  carve(m, dry);
  {
    DeviceGuard guard(device);
    if (guard.status != hipSuccess || hipMalloc(&m->arena, dry.off + 256) != hipSuccess) {
      delete m;
      return fail(NNCF_ERR_NOMEM, "device workspace allocation (%zu bytes) failed", dry.off);
    }
    if (hipMemset(m->arena, 0, dry.off + 256) != hipSuccess) {
      (void)hipFree(m->arena);
      delete m;
      return fail(NNCF_ERR_HIP, "clearing the device workspace failed");
    }
  }
  Carver real;
  real.base = static_cast<char*>(m->arena);
  carve(m, real);
  *out = m;
  return NNCF_OK;
}

int nncf_model_destroy(nncf_model* m) {
  if (!m) return NNCF_OK;
  if (m->arena) hipFree(m->arena);
  if (m->stamps) hipFree(m->stamps);
  if (m->bias_stamps) hipFree(m->bias_stamps);
  delete m;
  return NNCF_OK;
}

int nncf_model_debug_stamps(nncf_model* m, long long* out_host, int64_t n) {
  if (int rc = check_model(m)) return rc;
  NNCF_REQUIRE(m->sws.dbg != nullptr, "model was created without NNCF_DEBUG_TIMING=1");
  NNCF_HIP_CHECK(hipMemcpy(out_host, m->sws.dbg, sizeof(long long) * (size_t)n, hipMemcpyDeviceToHost));
  return NNCF_OK;
}

int nncf_model_set_item_grad_export(nncf_model* m, float* d_item_grad) {
  if (int rc = check_model(m)) return rc;
  m->export_gV = d_item_grad;
  return NNCF_OK;
}

int nncf_model_set_optimizer(nncf_model* m, int32_t mode, float beta1, float beta2, float epsilon, float* d_m_user,
                             float* d_v_user, float* d_m_item, float* d_v_item) {
  if (int rc = check_model(m)) return rc;
  NNCF_REQUIRE(mode >= 0 && mode <= 2, "optimizer mode must be 0 (sgd), 1 (lazy-sparse adam) or 2 (dense adam)");
  if (mode) {
    NNCF_REQUIRE(d_m_user && d_v_user, "Adam needs the user table's moment arrays");
    NNCF_REQUIRE(m->export_gV || (d_m_item && d_v_item), "Adam needs the item table's moment arrays");
    NNCF_REQUIRE(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && epsilon > 0.f, "bad Adam constants");
  }
  m->opt = mode;
  m->beta1 = beta1;
  m->beta2 = beta2;
  m->eps = epsilon;
  m->adam_t = 0;
  m->au = AdamTable{d_m_user, d_v_user, nullptr};
  m->av = AdamTable{d_m_item, d_v_item, nullptr};
  if (mode == 2) {
    const size_t n = (size_t)(m->cfg.user_count + m->cfg.item_count);
    DeviceGuard guard(m->device);
    if (!m->stamps && hipMalloc(&m->stamps, n * sizeof(int32_t)) != hipSuccess)
      return fail(NNCF_ERR_NOMEM, "stamp allocation failed");
    NNCF_HIP_CHECK(hipMemset(m->stamps, 0, n * sizeof(int32_t)));  // steps are numbered from 1
    m->au.stamp = m->stamps;
    m->av.stamp = m->stamps + m->cfg.user_count;
  }
  return NNCF_OK;
}

int nncf_model_set_bias(nncf_model* m, float* d_user_bias, float* d_item_bias, int64_t n_items, float* d_m_user,
                        float* d_v_user, float* d_m_item, float* d_v_item) {
  if (int rc = check_model(m)) return rc;
  NNCF_REQUIRE(!d_item_bias || n_items > 0, "item bias needs the catalogue size");
  if (m->opt) {
    NNCF_REQUIRE(!d_user_bias || (d_m_user && d_v_user), "Adam needs the user-bias moment arrays");
    NNCF_REQUIRE(!d_item_bias || (d_m_item && d_v_item), "Adam needs the item-bias moment arrays");
  }
  m->ub = d_user_bias;
  m->cb = d_item_bias;
  m->bias_items = d_item_bias ? n_items : 0;
  m->abu = AdamTable{d_m_user, d_v_user, nullptr};
  m->abc = AdamTable{d_m_item, d_v_item, nullptr};
  if (m->opt == 2 && (d_user_bias || d_item_bias)) {
    const size_t n = (size_t)(m->cfg.user_count + m->bias_items);
    DeviceGuard guard(m->device);
    if (m->bias_stamps) (void)hipFree(m->bias_stamps);
    m->bias_stamps = nullptr;
    if (hipMalloc(&m->bias_stamps, n * sizeof(int32_t)) != hipSuccess)
      return fail(NNCF_ERR_NOMEM, "stamp allocation failed");
    NNCF_HIP_CHECK(hipMemset(m->bias_stamps, 0, n * sizeof(int32_t)));
    m->abu.stamp = m->bias_stamps;
    m->abc.stamp = m->bias_stamps + m->cfg.user_count;
  }
  return NNCF_OK;
}

int nncf_model_set_sparse_adam_duplicates(nncf_model* m, int32_t as_written) {
  if (int rc = check_model(m)) return rc;
  m->dup_as_written = as_written ? 1 : 0;
  return NNCF_OK;
}

int nncf_model_set_shared_device(nncf_model* m, int32_t shared) {
  if (int rc = check_model(m)) return rc;
  m->shared_device = shared ? 1 : 0;
  return NNCF_OK;
}

int nncf_model_clear_error(nncf_model* m, void* stream) {
  if (int rc = check_model(m)) return rc;
  DeviceGuard guard(m->device);
  NNCF_HIP_CHECK(hipMemsetAsync(m->ferr, 0, 2 * sizeof(int), as_stream(stream)));
  return NNCF_OK;
}

int nncf_model_check(nncf_model* m, void* stream) {
  if (int rc = check_model(m)) return rc;
  int flag = 0;
  DeviceGuard guard(m->device);
  NNCF_HIP_CHECK(hipMemcpyAsync(&flag, m->ferr, sizeof(int), hipMemcpyDeviceToHost, as_stream(stream)));
  NNCF_HIP_CHECK(hipStreamSynchronize(as_stream(stream)));
  if (flag)
    return fail(NNCF_ERR_HIP, "a one-launch step timed out waiting for the other workgroups of its launch (they were not "
                "resident: another process or stream is using this GPU); the tables hold a partial step.  Create the model "
                "with nncf_model_set_shared_device(m, 1) / NNCF_SHARED_DEVICE=1 on a shared GPU");
  return NNCF_OK;
}

int nncf_model_get_step(nncf_model* m, int64_t* out) {
  if (int rc = check_model(m)) return rc;
  NNCF_REQUIRE(out != nullptr, "out is NULL");
  *out = m->adam_t;
  return NNCF_OK;
}

int nncf_model_set_hparams(nncf_model* m, float learn_rate, float loss_gamma, float neg_loss_weight, float u_reg) {
  if (int rc = check_model(m)) return rc;
  m->cfg.learn_rate = learn_rate;
  m->cfg.loss_gamma = loss_gamma;
  m->cfg.neg_loss_weight = neg_loss_weight;
  m->cfg.u_reg = u_reg;
  return NNCF_OK;
}

int nncf_step_original(nncf_model* m, const int32_t* d_uid, const int32_t* d_cid, float* d_loss, int update,
                       void* stream) {
  if (int rc = check_model(m)) return rc;
  NNCF_REQUIRE(d_uid && d_cid && d_loss, "NULL argument");
  NNCF_REQUIRE(m->cfg.num_negatives > 0, "train_scheme=original needs num_negatives > 0");
  const int b = m->cfg.batch_size_p, k = m->cfg.num_negatives;
  hipStream_t st = as_stream(stream);
  NNCF_REQUIRE(!(m->opt == 1 && m->dup_as_written),
               "sparse_adam_duplicates = 'as_written' with train_scheme = original: feed the batch through nncf_step_rowwise "
               "(every row a lookup position of its own; pointwise losses)");
  int rc = launch_plan(d_uid, 0, b, nullptr, 0, 0, m->plan_u, 1, st);
  if (rc) return rc;
  rc = launch_plan(d_cid, 0, b * (1 + k), nullptr, 0, 0, m->plan_v, 1, st);
  if (rc) return rc;
  {
    const StepArgs sa = make_args(m);
    if (int rc2 = launch_original(sa, d_uid, d_cid, d_cid + b, m->ows, m->plan_u, m->plan_v, d_loss, update, st)) return rc2;
    return update ? finish_update(m, sa, st) : NNCF_OK;
  }
}

int nncf_step_rowwise(nncf_model* m, const int32_t* d_uid, const int32_t* d_cid, const int32_t* d_y, float* d_loss,
                      int update, void* stream) {
  if (int rc = check_model(m)) return rc;
  NNCF_REQUIRE(d_uid && d_cid && d_y && d_loss, "NULL argument");
  NNCF_REQUIRE(m->cfg.num_negatives > 0, "row-wise batches need num_negatives > 0 (weights are lambda / k)");
  const int n = m->cfg.batch_size_p * (1 + m->cfg.num_negatives);
  hipStream_t st = as_stream(stream);
  int rc = launch_plan(d_uid, 0, n, nullptr, 0, 0, m->plan_u, 1, st);
  if (rc) return rc;
  rc = launch_plan(d_cid, 0, n, nullptr, 0, 0, m->plan_v, 1, st);
  if (rc) return rc;
  {
    const StepArgs sa = make_args(m);
    if (int rc2 = launch_rowwise(sa, d_uid, d_cid, d_y, m->rws, m->plan_u, m->plan_v, n, d_loss, update, st)) return rc2;
    return update ? finish_update(m, sa, st) : NNCF_OK;
  }
}

int nncf_epoch_rowwise(nncf_model* m, const int32_t* d_uid_all, const int32_t* d_cid_all, const int32_t* d_y_all,
                       int64_t n_steps, float* d_loss, void* stream) {
  if (int rc = check_model(m)) return rc;
  if (n_steps == 0) return NNCF_OK;  // nothing to do: buffers may be empty (NULL)
  NNCF_REQUIRE(d_uid_all && d_cid_all && d_y_all && d_loss, "NULL argument");
  NNCF_REQUIRE(m->cfg.num_negatives > 0 && n_steps >= 0, "bad argument");
  const int64_t n = (int64_t)m->cfg.batch_size_p * (1 + m->cfg.num_negatives);
  hipStream_t st = as_stream(stream);
  for (int64_t t0 = 0; t0 < n_steps; t0 += kPlanChunk) {
    const int64_t cnt = (n_steps - t0 < kPlanChunk) ? n_steps - t0 : kPlanChunk;
    int rc = launch_plan_steps(d_uid_all + t0 * n, n, (int)n, m->plan_u, d_cid_all + t0 * n, n, (int)n, nullptr, 0, 0, m->plan_v,
                               nullptr, 0, nullptr, cnt, st);
    if (rc) return rc;
    for (int64_t i = 0; i < cnt; ++i) {
      const int64_t t = t0 + i;
      const StepArgs a = make_args(m);
      rc = launch_rowwise(a, d_uid_all + t * n, d_cid_all + t * n, d_y_all + t * n, m->rws, m->plan_u.at(i),
                          m->plan_v.at(i), (int)n, d_loss + t, 1, st);
      if (rc) return rc;
      if ((rc = finish_update(m, a, st))) return rc;
    }
  }
  return NNCF_OK;
}

int nncf_step_neg_shared(nncf_model* m, const int32_t* d_uid, const int32_t* d_cid, float* d_loss, int update,
                         void* stream) {
  if (int rc = check_model(m)) return rc;
  if (int rc = check_shared(m)) return rc;
  NNCF_REQUIRE(d_uid && d_cid && d_loss, "NULL argument");
  const int b = m->cfg.batch_size_p;
  hipStream_t st = as_stream(stream);
  int rc = launch_plan_pair(d_uid, b, m->plan_u, d_cid, b, m->plan_v, st);
  if (rc) return rc;
  {
    const StepArgs sa = make_args(m);
    if (int rc2 = shared_step(m, sa, 0, d_uid, d_cid, m->plan_u, m->plan_v, d_loss, update, st)) return rc2;
    if (int rc2 = flush_pending_loss(m, st)) return rc2;
    return update ? finish_update(m, sa, st) : NNCF_OK;
  }
}

int nncf_step_neg_shared_export(nncf_model* m, const int32_t* d_uid, const int32_t* d_cid, float* d_loss,
                                 float* d_item_grad, void* stream) {
  if (int rc = check_model(m)) return rc;
  if (int rc = check_shared(m)) return rc;
  NNCF_REQUIRE(d_uid && d_cid && d_loss && d_item_grad, "NULL argument");
  const int b = m->cfg.batch_size_p;
  hipStream_t st = as_stream(stream);
  int rc = launch_plan_pair(d_uid, b, m->plan_u, d_cid, b, m->plan_v, st);
  if (rc) return rc;
  StepArgs a = make_args(m);
  a.gV = d_item_grad;
  rc = shared_step(m, a, 0, d_uid, d_cid, m->plan_u, m->plan_v, d_loss, 1, st);
  if (rc) return rc;
  if ((rc = flush_pending_loss(m, st))) return rc;
  if (m->opt == 2) {  // item rows belong to their owners: only the user table is stepped here
    if (int rc2 = launch_adam_idle(a, m->U, m->au, m->cfg.user_count, st)) return rc2;
  }
  if (m->opt) m->adam_t++;
  return NNCF_OK;
}

int nncf_plan_item_block(nncf_model* m, int32_t scheme, const int32_t* d_uid, const int32_t* d_cid,
                         int32_t* d_uniq_global, int32_t* d_n_u, void* stream) {
  if (int rc = check_model(m)) return rc;
  NNCF_REQUIRE(d_uid && d_cid && d_uniq_global && d_n_u, "NULL argument");
  NNCF_REQUIRE(scheme >= 0 && scheme <= 2, "scheme must be 0 (original), 1 (neg_shared) or 2 (group_neg_shared)");
  const int b = m->cfg.batch_size_p, n = scheme == 0 ? b * (1 + m->cfg.num_negatives) : b;
  hipStream_t st = as_stream(stream);
  int rc = launch_plan_pair(d_uid, b, m->plan_u, d_cid, n, m->plan_v, st);
  if (rc) return rc;
  return launch_plan_localize(m->plan_v, n, d_uniq_global, d_n_u, st);
}

int nncf_step_planned(nncf_model* m, int32_t scheme, const int32_t* d_uid, float* d_loss, int update, void* stream) {
  if (int rc = check_model(m)) return rc;
  NNCF_REQUIRE(d_uid && d_loss, "NULL argument");
  NNCF_REQUIRE(scheme >= 0 && scheme <= 2, "scheme must be 0, 1 or 2");
  const int b = m->cfg.batch_size_p;
  hipStream_t st = as_stream(stream);
  const int32_t* cid_local = m->plan_v.inverse;  // batch position -> row of the item block
  if (scheme == 0) {
    NNCF_REQUIRE(m->cfg.num_negatives > 0, "train_scheme=original needs num_negatives > 0");
    {
    const StepArgs sa = make_args(m);
    if (int rc2 = launch_original(sa, d_uid, cid_local, cid_local + b, m->ows, m->plan_u, m->plan_v, d_loss, update, st)) return rc2;
    return update ? finish_update(m, sa, st) : NNCF_OK;
  }
  }
  if (int rc = check_shared(m)) return rc;
  {
    const StepArgs sa = make_args(m);
    if (int rc2 = shared_step(m, sa, scheme == 2, d_uid, cid_local, m->plan_u, m->plan_v, d_loss, update, st)) return rc2;
    if (int rc2 = flush_pending_loss(m, st)) return rc2;
    return update ? finish_update(m, sa, st) : NNCF_OK;
  }
}

int nncf_step_group_neg_shared(nncf_model* m, const int32_t* d_uid, const int32_t* d_cid, float* d_loss, int update,
                               int32_t* d_cid_u, int32_t* d_cid_x, int32_t* d_n_u, void* stream) {
  if (int rc = check_model(m)) return rc;
  if (int rc = check_shared(m)) return rc;
  NNCF_REQUIRE(d_uid && d_cid && d_loss, "NULL argument");
  const int b = m->cfg.batch_size_p;
  hipStream_t st = as_stream(stream);
  int rc = launch_plan_pair(d_uid, b, m->plan_u, d_cid, b, m->plan_v, st);
  if (rc) return rc;
  const StepArgs sa = make_args(m);
  rc = shared_step(m, sa, 1, d_uid, d_cid, m->plan_u, m->plan_v, d_loss, update, st);
  if (rc) return rc;
  if ((rc = flush_pending_loss(m, st))) return rc;
  if (update) rc = finish_update(m, sa, st);
  if (rc) return rc;
  if (d_cid_u) NNCF_HIP_CHECK(hipMemcpyAsync(d_cid_u, m->plan_v.uniq, sizeof(int32_t) * b, hipMemcpyDeviceToDevice, st));
  if (d_cid_x) NNCF_HIP_CHECK(hipMemcpyAsync(d_cid_x, m->plan_v.inverse, sizeof(int32_t) * b, hipMemcpyDeviceToDevice, st));
  if (d_n_u) NNCF_HIP_CHECK(hipMemcpyAsync(d_n_u, m->plan_v.counts, sizeof(int32_t), hipMemcpyDeviceToDevice, st));
  return NNCF_OK;
}

int nncf_epoch_original(nncf_model* m, const int32_t* d_uid_all, const int32_t* d_cid_all, const int32_t* d_neg,
                        int64_t n_steps, float* d_loss, void* stream) {
  if (int rc = check_model(m)) return rc;
  if (n_steps == 0) return NNCF_OK;
  NNCF_REQUIRE(d_uid_all && d_cid_all && d_neg && d_loss, "NULL argument");
  NNCF_REQUIRE(m->cfg.num_negatives > 0, "train_scheme=original needs num_negatives > 0");
  NNCF_REQUIRE(n_steps >= 0, "negative step count");
  const int64_t b = m->cfg.batch_size_p, k = m->cfg.num_negatives;
  hipStream_t st = as_stream(stream);
  for (int64_t t0 = 0; t0 < n_steps; t0 += kPlanChunk) {
    const int64_t cnt = (n_steps - t0 < kPlanChunk) ? n_steps - t0 : kPlanChunk;
    int rc = plan_chunk(m, 0, d_uid_all, d_cid_all, d_neg, t0, cnt, st);
    if (rc) return rc;
    for (int64_t i = 0; i < cnt; ++i) {
      const int64_t t = t0 + i;
      const StepArgs a = make_args(m);
      rc = launch_original(a, d_uid_all + t * b, d_cid_all + t * b, d_neg + t * b * k, m->ows, m->plan_u.at(i),
                           m->plan_v.at(i), d_loss + t, 1, st);
      if (rc) return rc;
      if ((rc = finish_update(m, a, st))) return rc;
    }
  }
  return NNCF_OK;
}

static int epoch_shared(nncf_model* m, int group, const int32_t* d_uid_all, const int32_t* d_cid_all, int64_t n_steps,
                        float* d_loss, int32_t* d_n_u, void* stream) {
  if (int rc = check_model(m)) return rc;
  if (int rc = check_shared(m)) return rc;
  if (n_steps == 0) return NNCF_OK;
  NNCF_REQUIRE(d_uid_all && d_cid_all && d_loss, "NULL argument");
  NNCF_REQUIRE(n_steps >= 0, "negative step count");
  const int64_t b = m->cfg.batch_size_p;
  hipStream_t st = as_stream(stream);
  for (int64_t t0 = 0; t0 < n_steps; t0 += kPlanChunk) {
    const int64_t cnt = (n_steps - t0 < kPlanChunk) ? n_steps - t0 : kPlanChunk;
    int rc = plan_chunk(m, 1, d_uid_all, d_cid_all, nullptr, t0, cnt, st);
    if (rc) return rc;
    for (int64_t i = 0; i < cnt; ++i) {
      const int64_t t = t0 + i;
      const StepArgs a = make_args(m);
      rc = shared_step(m, a, group, d_uid_all + t * b, d_cid_all + t * b, m->plan_u.at(i), m->plan_v.at(i),
                       d_loss + t, 1, st);
      if (rc) return rc;
      if ((rc = finish_update(m, a, st))) return rc;
    }
    if (int rcf = flush_pending_loss(m, st)) return rcf;
    if (d_n_u) {
      // n_u of each planned step (counts[2*i]) -> d_n_u[t0 + i]
      NNCF_HIP_CHECK(hipMemcpy2DAsync(d_n_u + t0, sizeof(int32_t), m->plan_v.counts, 4 * sizeof(int32_t),
                                      sizeof(int32_t), (size_t)cnt, hipMemcpyDeviceToDevice, st));
    }
  }
  return NNCF_OK;
}

int nncf_epoch_neg_shared(nncf_model* m, const int32_t* d_uid_all, const int32_t* d_cid_all, int64_t n_steps,
                          float* d_loss, void* stream) {
  return epoch_shared(m, 0, d_uid_all, d_cid_all, n_steps, d_loss, nullptr, stream);
}

int nncf_epoch_group_neg_shared(nncf_model* m, const int32_t* d_uid_all, const int32_t* d_cid_all, int64_t n_steps,
                                float* d_loss, int32_t* d_n_u, void* stream) {
  return epoch_shared(m, 1, d_uid_all, d_cid_all, n_steps, d_loss, d_n_u, stream);
}

int nncf_predict_scores(nncf_model* m, const int32_t* d_uid, int64_t p, const int32_t* d_cid, int64_t q, float* d_out,
                        void* stream) {
  if (int rc = check_model(m)) return rc;
  NNCF_REQUIRE(p >= 0 && q >= 0, "negative shape");
  if (p == 0 || q == 0) return NNCF_OK;
  NNCF_REQUIRE(d_uid && d_cid && d_out, "NULL argument");
  return launch_predict(make_args(m), d_uid, p, d_cid, q, d_out, as_stream(stream));
}

}  // extern "C"
