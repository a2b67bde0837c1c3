// Shared declarations of the item-tower kernels (tower.hip: mean-pool tower and the common
// Dense / BatchNorm / ReLU tail; cnn.hip: the convolutional front of cnn_embedding).
#pragma once
#include <cmath>

#include "common.h"
#include "devutil.h"
#include "model.h"
#include "step.h"

namespace nncf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kTP = 128;       // LDS pitch of pooled rows
constexpr int kZP = 136;       // LDS pitch of dz rows: pitch / 4 = 2 (mod 16), conflict-free ds_read_b128 fragments (step_fused.hip kFDP)
constexpr int kTowerWG = 64;   // at most this many workgroups emit statistics partials
constexpr int kRowsPerK = 256;   // rows per K-part of the weight-gradient tiles

// Convolutional front of the cnn_embedding tower (cnn.hip); F == 0: mean-pool tower.
struct CnnArgs {
  int F, fl;                 // filters, filter length (odd)
  float keep;                // Dropout(conv_dropout_rate) retain probability (1 = off)
  unsigned long long seed;   // this step's conv-dropout seed
  float* Wk;                 // [fl][wd][F]   Keras layout (fl, 1, wd, F)
  float* Wt;                 // [F][fl*wd]    transposed copy (owned by the tower)
  float *bc, *g1, *b1, *rm1, *rv1;
  float* Y;                  // [m*L][F]      conv output before BatchNorm
  float* dXw;                // [np][m*L][wd] dL/d(word row) per (slot, distinct word ordinal), dropout scale applied: np partial
  int np;                    //               rows per (slot, word) -- one per filter group of the conv backward (cnn.hip),
  long long pstride;         //               pstride floats apart; tower_word_kernel adds them in group order
  float* part1;              // [m][2][128]   per-slot sums of (y-b), (y-b)^2
  float* st1;                // 7 x 128: mu, inv, var, mul, add, sum da, sum da*xhat
  float* bpart;              // [m][2][128]
  float* mpart;              // [2][kCnnMergeWG][2][128] partial totals of the multi-workgroup cnn_stats / cnn_merge (blocks above
  unsigned* mcnt;            //   kCnnFoldRows rows) and their two arrival counters (zero between launches)
  uint8_t* amax;             // [m][128] max pooling: the position that supplied each filter's maximum
  int pool_max;              // poolings = ['max'] (cnn_model.py:107-109) over the whole length instead of the average
  float* gWpart;             // [ceil(fl*wd*F / 64)][cap][64]: per-slot dWk blocks, element-group major
  float* gbpart;             // [m][128]      per-slot column sums of dY (conv bias gradient)
  const uint8_t* item_spos;  // [item_count][L] positions of an item's tokens sorted by (word, position)
  const uint8_t* item_start; // per CSR entry: index of its first position in the item's item_spos row
  float *mC, *vC;            // Adam moments of [Wk | bc, g1, b1 (128 each)]
  int dbg;                   // NNCF_CNN_DBG developer switches: 64 = clock stamps of the conv kernels' phases -> stamps
  long long* stamps;         // [2][cap * groups][16] (nncf_tower_debug_stamps)
  // filter bank (cnn_model.py:63-81: Convolution1D of several lengths over the same input, concatenated along the
  // filter axis): every length is one CnnArgs whose F filters are columns [c_off, c_off + F) of the pooled block h /
  // dh (pitch TowerArgs::hd) and rows [c_off, ...) of the Dense weights.  Banks run one after the other:
  int c_off;                 // first column of this bank in h / dh / Wd
  int z_acc;                 // pooling pass: add this bank's h.Wd to the z the earlier banks left (0: start from bd)
  int last;                  // pooling pass, inference: this bank completes z -> BatchNorm -> ReLU -> E
  int dx_acc;                // conv backward: add to the dX of the earlier banks (0: store)
  int fold;                  // blocks of <= kCnnFoldRows rows: cnn_pool_dense merges the forward's statistics partials itself and
                             // cnn_conv_bwd the BatchNorm backward's sums (no cnn_stats / cnn_merge launch); workgroup 0 publishes st1
  long long gexp_off;        // this bank's [Wk | bc, g1, b1 (128 each)] block in the flat gradient, from the start of the conv section
  // Round 5: the backward takes the forward's index and runs over WORK UNITS (cnn.hip).
  int4* idx;                 // [cap][kCnnIdxInt4] the ItemIdx a forward workgroup built for slot s (this step's dropout mask), verbatim
  int* nck;                  // [cap] kept distinct words of slot s
  int4* units;               // [cap + capx_cap] {slot, first word | end word << 16, dWk block, second half?}, heaviest first (cnn_units_build)
  int* split;                // [0] units, [1] items split in two
  int capx;                  // this call: at most this many items are split (0: one unit per slot, in slot order, `units` unused)
  int split_words;           // an item with more kept words than this is split along its words over two workgroups
  int gcap;                  // dWk blocks per element group in gWpart (cap + capx_cap)
};
constexpr int kCnnIdxInt4 = 226;   // sizeof(ItemIdx) / 16 (cnn.hip static_asserts it)
__host__ __device__ inline int cnn_capx_cap(int cap) { return (cap < 512 ? cap : 512) / 2; }
constexpr int kCnnFoldRows = 512;
constexpr int kCnnMergeWG = 32;   // workgroups of cnn_stats / cnn_merge on such blocks: one compute unit reads 2,700 slots' partials at
                                  // ~25 GB/s (122 us); 32 read them in parallel and the last to arrive adds the 32 rows
struct CnnBanks {             // all banks of a tower (kernels that walk the whole flat gradient)
  int nb;
  CnnArgs b[4];
};

struct TowerArgs {
  // parameters
  float* W;      // [n_words][wd]
  float* Wd;     // [wd][d]
  float* bd;
  float* gamma;
  float* beta;
  float* rm;
  float* rv;
  const int32_t* content;    // [item_count][L]
  const int32_t* item_ptr;   // CSR of the content matrix: item -> (word, count) pairs
  const int32_t* item_word;
  const uint8_t* item_cnt;
  const int32_t* ent_word;   // the same pairs at a fixed stride of L per item (-1 / 0 past the item's last pair):
  const uint8_t* ent_cnt;    // one dependent load less than through item_ptr (built at create, mean-pool tower)
  const float* scale;        // per-word row-dropout scale (Embedding(dropout=...), :56-59) or NULL
  // per call
  const int32_t* ids;        // [m] item ids of the block's rows
  const int32_t* n_valid;    // rows [0, *n_valid) are real; NULL = all m
  float* h;                  // [m][wd]
  float* z;                  // [m][d]
  float* E;                  // [m][d]
  const float* dA;           // [m][d]
  float* dh;                 // [m][wd]
  float* ws;
  uint8_t* cnt_t;            // [n_words][cap] word-major token counts of the batch (all-zero between steps)
  int m, wd, d, L, cap, n_words, train, fill;
  int hd;          // width of the pooled block h = input width of the Dense layer (wd, or the conv filters)
  int skip_words;  // tower_word_kernel: only the Dense / BN parameter update blocks run
  float eps, momentum, lr;
  float keep;                    // row-dropout retain probability (1 = off)
  unsigned long long drop_seed;  // this step's mask seed
  // optimiser (StepArgs::opt): Adam state of the word table and of the packed dense parameters
  // [Wd (wd*d) | bd (128) | gamma (128) | beta (128)]
  int opt, step_no;
  float lr_t, beta1, beta2, aeps;
  AdamTable aw;
  float* mD;
  float* vD;
  // data-parallel training: if non-NULL the backward applies nothing and writes the gradients
  // [words (n_words*wd) | dense_w (wd*d) | dense_b, bn_gamma, bn_beta (128 each)] here
  float* gexp;
  // conf.pooling == 'max' (mean_pool.py:70-72: MaxPooling1D over the whole length): h[s][c] = max over the item's
  // tokens of scale[w] W[w][c]; amax[s][c] = the word that supplied it (the backward routes dh[s][c] to that word's row)
  int pool_max;
  int32_t* amax;             // [cap][wd]
  // Hot words (mean-pool tower, average pooling): the few words that sit in most items of ANY batch (the pad id in all of
  // them) made the workgroup that owns them the word kernel's long pole -- one wave gathering a dh row per (word, slot)
  // cell, thousands of them.  Their gradient sum_s count[w][s] dh[s] is taken where dh is made instead: every role-A
  // workgroup of tower_bwd_kernel (16 block rows, dh in its accumulators) writes count-weighted partial rows
  // hot_part[workgroup][h][.] and clears the cells; the word kernel adds a hot word's partials in workgroup order.
  const int32_t* hot;        // [n_hot] word ids
  int n_hot;                 // <= kHotWords
  float* hot_part;           // [cap / 16][kHotWords][128]
  int32_t* hot_cnt;          // [cap / 16][kHotWords] cells (count sum) behind each partial: does the word take a step?
  // Postings mode (mean-pool tower, average pooling, blocks of more than 1,024 rows: `original` and the row-wise schemes).
  // The dense count matrix of such a block is ~1 % full and tower_word_kernel was bound by the INSTRUCTIONS of scanning it
  // (8,001 words x 2,700 rows).  The corpus is static, so the transposed CSR -- per word the items that hold it, with the
  // count -- is built once (nncf_tower_set_postings), and with it a static SCHEDULE: every workgroup of the word kernel owns
  // 8 units, a unit = up to 512 consecutive postings of one word, all units of a word in one workgroup, units per workgroup
  // capped by their postings -- every workgroup walks its postings in ONE round of loads and gathers about the same number of
  // rows.  A step only says where each of its items sits: the pool kernel writes slot_of[item] = {stamp of this forward,
  // block row}; the word kernel keeps the postings whose item carries the current stamp.  No count cell is written or
  // cleared for any word but the hot ones (whose cells tower_bwd_kernel's role A consumes).
  const int4* pm_units;      // [pm_nwg][8] {word (-1: empty), first posting, end, units of this word if this is its first unit else 0}
  const uint32_t* post_item; // [nnz] item | count << 24, items ascending inside a word
  int2* slot_of;             // [item_count] {stamp, block row}; zero-initialised, stamps start at 1
  const uint32_t* hot_bits;  // [ceil(n_words / 32)] membership of the hot list (the cells the pool kernel still writes)
  int pm_nwg;                // workgroups of the schedule
  int pm, pm_stamp;          // this forward / backward runs in postings mode; its stamp
  // activity_l2(conf.c_reg) on the word Embedding's OUTPUT (mean_pool.py:56-59, cnn_model.py:45-48; utilities.py:122-135):
  // loss += c_reg / n * sum_w T[w] |scale[w] W[w]|^2, T[w] = occurrences of w in the block's n rows (every position counts).
  // item_dense_transform variants (mean_pool.py:82-97): Activation(dense_hidden_actv) other than relu, Dropout(dense_hidden_dropout).
  // Such a tower's steps materialise the item block (as conf.use_content_id does) and its backward starts from
  // dA * mask * act'(y) (tower_tail_grad_kernel) with the relu gates of the BatchNorm-backward kernels open (tail_pre).
  int no_dense;              // no item_dense_transform (mean_pool.py:78-101 else branch): the pooled block IS the item block
  int tail_act;              // kAct*: 0 = relu (the presets)
  int tail_pre;              // the backward's dA already carries mask * act'(y)
  int no_bn;                 // conf.no_BN (mean_pool.py:88-93): Dense -> Activation, no BatchNormalization in between -- the statistics
                             // slots hold the identity (mu 0, inv 1, mul 1, add 0), the BatchNorm-backward sums are zero, gamma reads 1
  float tail_keep;           // Dropout retain probability of this forward (1 = off)
  unsigned long long tail_seed;
  float c_reg;               // 0: off (every preset of the reference)
  int32_t* wcount;           // [n_words] T[w]: filled by tower_wcount_kernel, consumed and cleared by the word kernel's tail
  float* creg_q;             // [n_words] T[w] |scale[w] W[w]|^2 (tower_creg_kernel), summed in word order into the loss
  CnnArgs cnn;
};
constexpr int kHotWords = 16;

__device__ __forceinline__ float adam1(float g, float* m, float* v, float p, const TowerArgs& a) {
  const float mm = fmaf(a.beta1, *m, (1.f - a.beta1) * g), vv = fmaf(a.beta2, *v, (1.f - a.beta2) * g * g);
  *m = mm;
  *v = vv;
  return p - a.lr_t * mm / (sqrtf(vv) + a.aeps);
}

__host__ __device__ inline unsigned long long splitmix64(unsigned long long x) {
  x ^= x >> 30;
  x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27;
  x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return x;
}

// Embedding(dropout = 1 - keep) of Keras 1 (mean_pool.py:56-59): whole word rows are dropped
// per batch, survivors scaled by 1/keep.  The mask is a counter-based hash of (step seed, word),
// evaluated where it is needed -- no mask array, and dropped rows are never loaded.  An explicit
// mask (a.scale) overrides it.  [K-mem] the reference draws its mask from TF's RNG: the stream
// cannot match, only the distribution.
__device__ __forceinline__ float word_scale(const TowerArgs& a, int w) {
  if (a.scale) return a.scale[w];
  if (a.keep >= 1.f) return 1.f;
  const unsigned long long x = splitmix64(a.drop_seed + (unsigned long long)w * 0x9E3779B97F4A7C15ull);
  const float u = (float)(x >> 40) * (1.f / 16777216.f);
  return u < a.keep ? 1.f / a.keep : 0.f;
}

// Keras-1 activations of the Dense tail (keras/activations.py by name; hard_sigmoid = clip(0.2 y + 0.5, 0, 1) [K-mem])
// kActSoftmax (keras/activations.py softmax on a 2-D tensor: over the item_dim columns of a row) is not element-wise: tail_act /
// tail_act_grad pass it through as linear and the row kernels (tower_bn_kernel, tower_tail_grad_kernel, tower_softmax_rows_kernel)
// do the row's maximum / sum / dot themselves
enum { kActRelu = 0, kActTanh, kActSigmoid, kActHardSigmoid, kActSoftplus, kActSoftsign, kActLinear, kActSoftmax, kActCount };
__device__ __forceinline__ float row16_max(float v) {   // over the 16 lanes of a row group
  v = fmaxf(v, __shfl_xor(v, 1));
  v = fmaxf(v, __shfl_xor(v, 2));
  v = fmaxf(v, __shfl_xor(v, 4));
  v = fmaxf(v, __shfl_xor(v, 8));
  return v;
}
__device__ __forceinline__ float tail_act(int k, float y) {
  switch (k) {
    case kActRelu: return fmaxf(y, 0.f);
    case kActTanh: return tanhf(y);
    case kActSigmoid: return 1.f / (1.f + expf(-y));
    case kActHardSigmoid: return fminf(fmaxf(fmaf(0.2f, y, 0.5f), 0.f), 1.f);
    case kActSoftplus: return fmaxf(y, 0.f) + log1pf(expf(-fabsf(y)));
    case kActSoftsign: return y / (1.f + fabsf(y));
    default: return y;
  }
}
__device__ __forceinline__ float tail_act_grad(int k, float y) {
  switch (k) {
    case kActRelu: return y > 0.f ? 1.f : 0.f;
    case kActTanh: { const float t = tanhf(y); return 1.f - t * t; }
    case kActSigmoid: { const float s = 1.f / (1.f + expf(-y)); return s * (1.f - s); }
    case kActHardSigmoid: return (y > -2.5f && y < 2.5f) ? 0.2f : 0.f;
    case kActSoftplus: return 1.f / (1.f + expf(-y));
    case kActSoftsign: { const float t = 1.f + fabsf(y); return 1.f / (t * t); }
    default: return 1.f;
  }
}
// Dropout(dense_hidden_dropout) mask of element (row, c) of the block, including its 1/keep scale: counter-based hash,
// evaluated where needed (forward, backward, nncf_tower_dense_mask for the tests).  [K-mem] TF's RNG stream cannot match.
__device__ __forceinline__ float tail_mask(float keep, unsigned long long seed, int row, int c) {
  if (keep >= 1.f) return 1.f;
  const unsigned long long x = splitmix64(seed + ((unsigned long long)row * 128ull + (unsigned long long)c) * 0x9E3779B97F4A7C15ull);
  const float u = (float)(x >> 40) * (1.f / 16777216.f);
  return u < keep ? 1.f / keep : 0.f;
}

// workspace layout (floats)
__host__ __device__ inline size_t ws_stats() { return 0; }                                     // [WG][3][128]
__host__ __device__ inline size_t ws_muinv() { return (size_t)kTowerWG * 3 * 128; }             // mu, inv, var, mul, add
__host__ __device__ inline size_t ws_bsum() { return ws_muinv() + 5 * 128; }                    // [WG][2][128]
__host__ __device__ inline size_t ws_bfin() { return ws_bsum() + (size_t)kTowerWG * 2 * 128; }  // sum dy, sum dy*xhat
__host__ __device__ inline size_t ws_cnt() { return ws_bfin() + 2 * 128; }                      // 4 arrival counters
__host__ __device__ inline size_t ws_gb(int) { return ws_cnt() + 4; }                           // [ks][128]
__host__ __device__ inline size_t ws_gw(int ks) { return ws_gb(0) + (size_t)ks * 128; }         // [ks][wd*d]
inline size_t ws_total(int wd, int d, int ks) { return ws_gw(ks) + (size_t)ks * wd * d; }

__device__ __forceinline__ int valid_rows(const TowerArgs& a) { return a.n_valid ? min(*a.n_valid, a.m) : a.m; }

// column c of lane q's Row element e (e = 0..7): floats [4q,4q+4) then [64+4q, 64+4q+4)
__device__ __forceinline__ int col_of(int q, int e) { return (e < 4 ? 4 * q : 64 + 4 * q - 4) + e; }
__device__ __forceinline__ float& relem(Row& r, int e) {
  return e == 0 ? r.a.x : e == 1 ? r.a.y : e == 2 ? r.a.z : e == 3 ? r.a.w : e == 4 ? r.b.x : e == 5 ? r.b.y : e == 6 ? r.b.z : r.b.w;
}

// "Last workgroup done" reduction: every workgroup publishes its partial, the one that arrives
// last (device-scope counter) merges all partials in partial order -- the result does not depend
// on arrival order -- and re-arms the counter.  Returns true in every thread of that workgroup.
__device__ __forceinline__ bool last_block_arrives(unsigned* counter) {
  __shared__ int s_last;
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned prev = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (prev == gridDim.x - 1);
    if (s_last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (s_last) __threadfence();
  return s_last != 0;
}

// partial column sums [nparts][2][128] -> totals, in part order (threads c < d)
__device__ __forceinline__ void merge_partials(const float* part, int nparts, int c, float& t1, float& t2) {
  t1 = 0.f;
  t2 = 0.f;
  for (int p0 = 0; p0 < nparts; p0 += 16) {
    float u1[16], u2[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float* in = part + (size_t)(p0 + e) * 2 * 128;
      u1[e] = (p0 + e < nparts) ? in[c] : 0.f;
      u2[e] = (p0 + e < nparts) ? in[128 + c] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      t1 += u1[e];
      t2 += u2[e];
    }
  }
}

}  // namespace nncf

// ---- the tower object --------------------------------------------------------------------------
namespace nncf {
// Conv weight update as workgroups of the word kernel's launch (512 threads = 4 part-groups x 128
// elements): the dWk partials [nparts][nel] of the conv backward are summed in partial order --
// part-group g adds parts g, g+4, ... (16 loads in flight), the groups are combined in group order
// through LDS -- and the SGD / Adam step is taken on Wk (+ its transposed copy).  Block nb (the last)
// handles the conv bias and the BN1 scale / shift.  sG: 4 x 128 floats of LDS.
constexpr int kCnnUpdElems = 64;                      // elements of dWk per workgroup
constexpr int kCnnUpdGroups = 512 / kCnnUpdElems;     // part-groups (one wave each): 125 slots = 1 round of 16 loads per wave
__host__ __device__ inline int cnn_update_blocks(int nel) { return (nel + kCnnUpdElems - 1) / kCnnUpdElems + 1; }
__device__ __forceinline__ void cnn_update_block(const TowerArgs& a, int blk, float* sG) {
  const int F = a.cnn.F, fl = a.cnn.fl, nel = fl * a.wd * F;
  // one dWk block per slot [0, n_valid) + one per item the backward split in two (cnn_units_build), contiguous, summed in that order
  const int nv_ = valid_rows(a);
  const int nparts = nv_ + (a.cnn.capx ? a.cnn.split[1] : 0);
  const int nb = (nel + kCnnUpdElems - 1) / kCnnUpdElems;
  if (blk < nb) {
    const int c = threadIdx.x & (kCnnUpdElems - 1), g = threadIdx.x / kCnnUpdElems;
    const int el = blk * kCnnUpdElems + c;
    float acc = 0.f;
    if (el < nel) {
      for (int p0 = g; p0 < nparts; p0 += kCnnUpdGroups * 16) {
        float u[16];
#pragma unroll
        for (int e = 0; e < 16; ++e)
          u[e] = (p0 + kCnnUpdGroups * e < nparts)
                     ? a.cnn.gWpart[((int64_t)blk * a.cnn.gcap + (p0 + kCnnUpdGroups * e)) * kCnnUpdElems + c] : 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc += u[e];
      }
    }
    sG[g * kCnnUpdElems + c] = acc;
    __syncthreads();
    if (g == 0 && el < nel) {
      float gsum = 0.f;
#pragma unroll
      for (int gg = 0; gg < kCnnUpdGroups; ++gg) gsum += sG[gg * kCnnUpdElems + c];
      if (a.gexp) {  // data-parallel step: the gradient leaves, nothing is applied here
        a.gexp[(size_t)a.n_words * a.wd + (size_t)a.hd * a.d + 384 + (size_t)a.cnn.gexp_off + el] = gsum;
        return;
      }
      const float nw = a.opt ? adam1(gsum, a.cnn.mC + el, a.cnn.vC + el, a.cnn.Wk[el], a) : a.cnn.Wk[el] - a.lr * gsum;
      a.cnn.Wk[el] = nw;
      const int f = el % F, k = (el / F) % a.wd, tap = el / (F * a.wd);
      a.cnn.Wt[(int64_t)f * fl * a.wd + tap * a.wd + k] = nw;
    }
    return;
  }
  // the last block: conv bias and BN1 scale / shift (<= 128 columns): four part-groups of 128 lanes
  const int c = threadIdx.x & 127, g = threadIdx.x >> 7;
  float acc = 0.f;
  for (int p0 = g; p0 < nv_; p0 += 4 * 16) {   // 16 loads in flight: a load per iteration made this ONE workgroup the
    float u[16];                                 // launch's long pole (62 dependent trips to memory, ~19 us)
#pragma unroll
    for (int e = 0; e < 16; ++e) u[e] = (p0 + 4 * e < nv_) ? a.cnn.gbpart[(int64_t)(p0 + 4 * e) * 128 + c] : 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc += u[e];
  }
  sG[g * 128 + c] = acc;
  __syncthreads();
  if (g != 0 || c >= F) return;
  float gb = 0.f;
#pragma unroll
  for (int gg = 0; gg < 4; ++gg) gb += sG[gg * 128 + c];
  const float d1 = a.cnn.st1[640 + c], d2 = a.cnn.st1[768 + c];
  if (a.gexp) {
    float* gd = a.gexp + (size_t)a.n_words * a.wd + (size_t)a.hd * a.d + 384 + (size_t)a.cnn.gexp_off + nel;
    gd[c] = gb;
    gd[128 + c] = d2;
    gd[256 + c] = d1;
  } else if (a.opt) {
    float *mm = a.cnn.mC + nel, *vv = a.cnn.vC + nel;
    a.cnn.bc[c] = adam1(gb, mm + c, vv + c, a.cnn.bc[c], a);
    a.cnn.g1[c] = adam1(d2, mm + 128 + c, vv + 128 + c, a.cnn.g1[c], a);
    a.cnn.b1[c] = adam1(d1, mm + 256 + c, vv + 256 + c, a.cnn.b1[c], a);
  } else {
    a.cnn.bc[c] -= a.lr * gb;
    a.cnn.g1[c] -= a.lr * d2;
    a.cnn.b1[c] -= a.lr * d1;
  }
}
}  // namespace nncf

struct nncf_tower {
  nncf_tower_cfg cfg;
  nncf_tower_params prm;
  int device = 0;
  int ks = 1;
  void* arena = nullptr;
  float *h = nullptr, *z = nullptr, *dh = nullptr, *ws = nullptr;
  uint8_t* cnt_t = nullptr;
  int32_t* ent_word = nullptr;  // fixed-stride copy of the item CSR (mean-pool tower)
  uint8_t* ent_cnt = nullptr;
  float* bpart = nullptr;
  int32_t *uniq_g = nullptr, *n_u = nullptr;  // kPlanChunk blocks of max_rows ids / kPlanChunk counts
  bool fwd_cached = false;
  nncf::TowerArgs last{};
  float drop_rate = 0.f;
  uint64_t drop_seed = 0, drop_step = 0;  // train-mode forwards since nncf_tower_set_dropout
  // optimiser (nncf_tower_set_optimizer)
  int opt = 0;
  float beta1 = 0.9f, beta2 = 0.999f, aeps = 1e-8f;
  int64_t adam_t = 0;  // standalone backward calls; inside a model step the model's counter rules
  nncf::AdamTable aw;
  float *mD = nullptr, *vD = nullptr;
  int32_t* stamps = nullptr;
  float* gexp = nullptr;
  // convolutional front (cfg.conv_filters > 0)
  nncf::CnnArgs cnn{};           // bank 0 (the only one unless cfg.n_filter_lens > 1)
  int nb = 0;                    // filter banks (0: mean-pool tower)
  nncf::CnnArgs bank[3]{};       // banks 1 .. nb - 1
  const nncf::CnnArgs& bank_at(int g) const { return g == 0 ? cnn : bank[g - 1]; }
  nncf::CnnArgs& bank_at(int g) { return g == 0 ? cnn : bank[g - 1]; }
  int hd() const { return nb > 0 ? nb * cnn.F : cfg.word_dim; }   // width of the pooled block = Dense input width
  float conv_drop = 0.f;
  uint64_t conv_step = 0;  // train-mode forwards since nncf_tower_set_dropout (conv mask index)
  // conf.pooling = 'max' (nncf_tower_set_pooling)
  int pool_max = 0;
  int32_t* amax = nullptr;       // own allocation [max_rows][word_dim]
  // conf.use_content_id (mean_pool.py:102-108; nncf_tower_set_content_id): tower output + Emb_Cid[item id]
  float* cid_tab = nullptr;      // [item_count][item_dim], borrowed
  float v_reg = 0.f;             // activity_l2(v_reg) on the Emb_Cid output
  nncf::AdamTable acid;          // its Adam state (borrowed; stamp for dense Adam)
  float* cid_reg = nullptr;      // own allocation [max_rows] |Emb_Cid row|^2 of the block's rows
  // item_dense_transform variants (nncf_tower_set_dense_tail; TowerArgs::tail_act)
  int tail_act = 0;
  float tail_drop = 0.f;
  uint64_t tail_step = 0;        // train-mode forwards since nncf_tower_set_dense_tail / set_dropout (mask index)
  float* dA_tail = nullptr;      // own allocation [max_rows][item_dim]: dA * mask * act'(y)
  int no_dense = 0;              // nncf_tower_set_no_dense: no Dense / BatchNorm / Activation behind the pooling
  int no_bn = 0;                 // nncf_tower_set_no_bn: conf.no_BN (mean_pool.py:88-93)
  bool tail_variant() const { return tail_act != 0 || tail_drop > 0.f || no_bn; }
  bool tail_general() const { return tail_variant() || no_dense; }   // the step runs on the materialised item block
  // conf.c_reg (nncf_tower_set_c_reg; TowerArgs::c_reg)
  float c_reg = 0.f;
  int32_t* wcount = nullptr;     // own allocation [word_count], all-zero between steps
  float* creg_q = nullptr;       // own allocation [word_count]
  // hot words (nncf_tower_set_hot_words; TowerArgs::hot)
  int32_t* hot = nullptr;        // own allocation [kHotWords]
  int n_hot = 0;
  int32_t hot_host[nncf::kHotWords] = {};   // the same ids on the host (nncf_tower_set_postings checks the schedule against them)
  float* hot_part = nullptr;     // own allocation
  int32_t* hot_cnt = nullptr;
  // postings mode (nncf_tower_set_postings; TowerArgs::pm_units): caller-owned arrays, own hot bitmap
  const int4* pm_units = nullptr;
  int pm_nwg = 0;
  const uint32_t* post_item = nullptr;
  int2* slot_of = nullptr;
  uint32_t* hot_bits = nullptr;  // own allocation
  int pm_stamp = 0;
};

namespace nncf {
// cnn.hip
size_t cnn_carve(nncf_tower* t, Carver& c);
int cnn_sync_weights(nncf_tower* t, hipStream_t st);
int cnn_forward(nncf_tower* t, TowerArgs& a, hipStream_t st);
int cnn_backward(nncf_tower* t, TowerArgs& a, hipStream_t st);
int cnn_mask(nncf_tower* t, TowerArgs& a, int64_t m, float* out, hipStream_t st);
}  // namespace nncf
