// Pairwise losses of the shared-negative schemes at large batches (step_pair.hip): argument block and host entry points.
#pragma once
#include "model.h"
#include "step.h"

namespace nncf {

struct PairArgs {
  float* anchor = nullptr;     // [b]           D[p] = Uhat[uid[p]] . Vhat[cid[p]]: the positive score of batch position p
  float* uhat = nullptr;       // [b][d]        normalised user row of every position, as it was before the step
  float* vhat = nullptr;       // [b][d]        normalised item row of every position
  float* t_part = nullptr;     // [strips][b]   per strip of the position-streaming side: sum of dL/dS over its own rows
  float* stage = nullptr;      // [2][b][128]   dL/d(normalised row) of every unique row, by rank (users, then items)
  float* loss_part = nullptr;  // [strips] (neg_shared) / [2 strips] (group): loss + regulariser partials
  // f16-plane products only where they are exact (round 5; step_fused.hip wide_in_domain): dom[4] = {max |x| of the batch's user rows,
  // of its item rows (float bits), the largest multiplicity of a row, unused}, taken by pair_anchor_kernel, cleared by the commit
  // kernel; pair_wide_kernel<H16> runs inside the domain, its fp32-pipe twin behind it outside.  NULL: no guard.
  int* dom = nullptr;
};

// log-loss / max-margin, id tables, no InteractionDot bias, no exported gradients, b a multiple of 16
bool pair_wide_eligible(const StepArgs& a);
int pair_row_tiles(const StepArgs& a);   // 16-row tiles per workgroup (2 or 4)
// anchors -> wide strips -> commit (+ the loss into *loss_out); group as launch_shared
int launch_pair_wide(const StepArgs& a, int group, const int32_t* uid, const int32_t* cid, const PairArgs& p,
                     const PlanView& pu, const PlanView& pv, float* loss_out, int update, hipStream_t stream);

}  // namespace nncf
