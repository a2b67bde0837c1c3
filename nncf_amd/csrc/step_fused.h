// One-launch shared-negative step (step_fused.hip): argument block and host entry points.
#pragma once
#include "model.h"
#include "step.h"

namespace nncf {

struct FusedArgs {
  float* loss_part = nullptr;        // this step's per-workgroup loss partials [fused_max_blocks(b)]
  const float* prev_part = nullptr;  // the previous step's partials, summed into *prev_out by this launch
  int prev_n = 0;
  float* prev_out = nullptr;
  // text towers: when `stats` is set every workgroup derives the BatchNorm statistics of the item block
  // (StepArgs::bn_z, rows [0, n_u)) itself and workgroup 0 publishes mu, inv, var, mul, add (128-float
  // pitch) there + the running statistics; NULL: StepArgs::bn_stats already holds them.
  float* stats = nullptr;
  const float* gamma = nullptr;
  const float* beta = nullptr;
  float* rm = nullptr;
  float* rv = nullptr;
  float eps = 1e-3f, momentum = 0.99f;
  // opposite-range split: `split` workgroups share one strip of 16 own rows (stage st belongs to part st % split);
  // each writes its partial output rows to slab[strip][part][16][128], the last to arrive (per-strip ticket in
  // cnt[], agent-scope release / acquire) adds the parts in part order and finalises the rows
  int split = 1;
  float* slab = nullptr;
  int* cnt = nullptr;
  // read-before-write across the launch: a strip's rows are stepped IN PLACE while the opposite pass of other
  // workgroups streams the same table.  Every workgroup announces "my table reads are done" in done[parity] (one
  // relaxed agent-scope add, when its last stage's rows are in LDS); a workgroup that steps a table another pass
  // reads waits for all of them (the count follows from the plans' segment counts) before its first store.
  // done[parity ^ 1] is zeroed by this launch for the next one.  All workgroups must be resident at once: the
  // launcher checks the grid against the device's capacity.
  int* done = nullptr;
  int parity = 0;
  // the hand-shake's wait is bounded: after spin_ticks of the 100 MHz wall clock a waiting workgroup raises *err,
  // stores nothing and every later loss of the model is NaN (nncf_model_check reports it).  shared_device (host side
  // only): the model does not have the device to itself (another process, an evaluator on a side stream) -> the
  // launcher always takes the stage + commit form, which waits for nothing.
  int* err = nullptr;
  long long spin_ticks = 25000000;   // 250 ms
  int shared_device = 0;
  // set by the launcher when the grid does NOT fit the device at once: no hand-shake -- the gradient rows
  // (normalisation backward and regulariser included) go to stage[side][own slot][128] and
  // fused_commit_kernel, a second launch, steps the tables
  float* stage = nullptr;
  long long* dbg = nullptr;   // developer timing: [workgroup][32] s_memtime stamps (NNCF_DEBUG_TIMING=1)
  // Large-batch kernels, f16-plane products (round 5): dom[parity][4] = {max |x| of the batch's user rows, of its item rows
  // (float bits), largest multiplicity of a user, of an item}, written by wide_domain_kernel in front of the step.  The
  // f16-plane instantiation runs only if every staged value and every dL/dS entry fits fp16's range by a margin; otherwise
  // it returns at once and the fp32-pipe instantiation, launched right behind it with the same arguments, takes the step.
  // NULL: no guard (NNCF_WIDE_F16SPLIT=0 / =force).
  int* dom = nullptr;
};

// pointwise loss, no InteractionDot bias, b a multiple of 16
bool fused_eligible(const StepArgs& a);
int fused_max_blocks(int b);  // strips per side (= V-pass BatchNorm partials)
int fused_split(int b);       // workgroups per strip chosen for batch size b
int fused_split_rect(int b, int bc);   // the same for b rows x bc columns (StepArgs::bc)
int fused_row_tiles(const StepArgs& a);   // 16-row tiles per workgroup (1: strip kernel; 2, 4: wide kernel)
int fused_loss_parts(const StepArgs& a);  // loss partials one launch writes
constexpr int kFusedMaxSplit = 8;
int launch_fused(const StepArgs& a, int group, const int32_t* uid, const int32_t* cid, const FusedArgs& f,
                 const PlanView& pu, const PlanView& pv, int update, hipStream_t stream);
int launch_fused_flush(const float* part, int n, float* out, const int* err, hipStream_t stream);

}  // namespace nncf
