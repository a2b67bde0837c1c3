// Mean-pool text tower (modules/content/mean_pool.py:27-33,46-110), backward of the shared-negative
// schemes in ONE launch, and the per-step word -> (slot, count) lists it reads.
//
// After the step kernels the chain of the two-kernel backward (tower.hip) was
//   tower_bwd_kernel   dz = BatchNorm backward of dL/dE;  dh = dz.Wd^T (role A);  dWd = h^T.dz (role B)
//   tower_word_kernel  W[w] -= lr/L scale_w sum_s count[w,s] dh[s];  Dense / BN parameter steps
// with dh materialised in between only because the word rows need it.  By linearity
//   sum_s count[w,s] dh[s] = (sum_s count[w,s] dz[s]) . Wd^T
// so a word workgroup can accumulate in dz space -- dz[s] is an element-wise function of dL/dE[s],
// z[s] and per-column constants -- and multiply its 8 accumulated rows by Wd^T once (fp32 MFMA, the Wd
// fragments prefetched into registers).  The dWd tiles run as further workgroups of the same launch and take
// their optimiser step themselves, into the OTHER buffer of a double-buffered Wd (the word workgroups
// of this launch still read the old weights); bd, gamma, beta have no reader in this launch and are
// stepped in place.  One launch instead of two, no dh round trip.
//
// Which (slot, count) cells a word has in a batch is known from the batch's item ids alone, so it is
// planned ahead for 64 steps at a time instead of being rediscovered inside every step: a bitmap
// [word][slot] is filled with integer atomics (order-independent), per-word offsets are its row
// popcounts scanned in word order, and every cell's place is its rank inside its row -- a deterministic
// counting sort (counts by integer atomics too), cells of a word in ascending slot order.  This replaces the 4 MB per-step byte matrix
// `cnt_t` (written by the pooling kernel, scanned and cleared by the word kernel).
#include "tower.h"

namespace nncf {

// ---- cells plan ----------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) cells_mark_kernel(TowerArgs a, CellsView cv, const int32_t* __restrict__ uniq_g,
                                                         const int32_t* __restrict__ n_u, int64_t uniq_stride) {
  const int step = blockIdx.y;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int slot = (int)(idx / a.L), e = (int)(idx % a.L);
  if (slot >= n_u[step]) return;
  const int item = uniq_g[step * uniq_stride + slot];
  const int w = a.ent_word[(int64_t)item * a.L + e];
  if (w < 0) return;
  atomicOr(cv.bits + ((int64_t)step * a.n_words + w) * cv.rw + (slot >> 5), 1u << (slot & 31));
  atomicAdd(cv.off + (int64_t)step * (a.n_words + 1) + w + 1, 1);  // cells of word w, scanned into offsets next
}

// one workgroup per step: the per-word cell counts left in off[1 .. W] by cells_mark_kernel (integer atomics:
// order-independent) -> inclusive scan in word order, so that off[w] = number of cells of words < w
__global__ void __launch_bounds__(1024) cells_offsets_kernel(TowerArgs a, CellsView cv) {
  __shared__ int sScan[1024];
  const int step = blockIdx.x, tid = threadIdx.x;
  const int W = a.n_words, per = (W + 1023) / 1024;
  int32_t* off = cv.off + (int64_t)step * (W + 1) + 1;
  int local = 0;
  for (int i = 0; i < per; ++i) {
    const int w = tid * per + i;
    if (w < W) local += off[w];
  }
  sScan[tid] = local;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const int t = tid >= o ? sScan[tid - o] : 0;
    __syncthreads();
    sScan[tid] += t;
    __syncthreads();
  }
  int run = sScan[tid] - local;
  for (int i = 0; i < per; ++i) {
    const int w = tid * per + i;
    if (w < W) {
      run += off[w];
      off[w] = run;
    }
  }
}

__global__ void __launch_bounds__(256) cells_fill_kernel(TowerArgs a, CellsView cv, const int32_t* __restrict__ uniq_g,
                                                         const int32_t* __restrict__ n_u, int64_t uniq_stride) {
  const int step = blockIdx.y;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int slot = (int)(idx / a.L), e = (int)(idx % a.L);
  if (slot >= n_u[step]) return;
  const int item = uniq_g[step * uniq_stride + slot];
  const int w = a.ent_word[(int64_t)item * a.L + e];
  if (w < 0) return;
  const uint32_t* row = cv.bits + ((int64_t)step * a.n_words + w) * cv.rw;
  int rank = __popc(row[slot >> 5] & ((1u << (slot & 31)) - 1u));
  for (int k = 0; k < (slot >> 5); ++k) rank += __popc(row[k]);
  const int pos = cv.off[(int64_t)step * (a.n_words + 1) + w] + rank;
  cv.cells[(int64_t)step * cv.cells_cap + pos] = (slot << 8) | (int)a.ent_cnt[(int64_t)item * a.L + e];
}

int launch_cells(const TowerArgs& a, const CellsView& cv, const int32_t* uniq_g, const int32_t* n_u, int64_t uniq_stride,
                 int rows, int64_t n_steps, hipStream_t st) {
  if (n_steps <= 0) return NNCF_OK;
  NNCF_HIP_CHECK(hipMemsetAsync(cv.bits, 0, sizeof(uint32_t) * (size_t)n_steps * a.n_words * cv.rw, st));
  NNCF_HIP_CHECK(hipMemsetAsync(cv.off, 0, sizeof(int32_t) * (size_t)n_steps * (a.n_words + 1), st));
  const unsigned gx = (unsigned)(((int64_t)rows * a.L + 255) / 256);
  {
    ProfScope _ps("cells_mark_kernel", st);
    hipLaunchKernelGGL(cells_mark_kernel, dim3(gx, (unsigned)n_steps), dim3(256), 0, st, a, cv, uniq_g, n_u, uniq_stride);
  }
  {
    ProfScope _ps("cells_offsets_kernel", st);
    hipLaunchKernelGGL(cells_offsets_kernel, dim3((unsigned)n_steps), dim3(1024), 0, st, a, cv);
  }
  {
    ProfScope _ps("cells_fill_kernel", st);
    hipLaunchKernelGGL(cells_fill_kernel, dim3(gx, (unsigned)n_steps), dim3(256), 0, st, a, cv, uniq_g, n_u, uniq_stride);
  }
  NNCF_HIP_CHECK(hipGetLastError());
  return NNCF_OK;
}

// ---- one-launch backward ------------------------------------------------------------------------
constexpr int kWBWords = 32;       // words per workgroup (8 waves, 4 words finished by each)
constexpr int kWBAP = 132;         // LDS pitch of the accumulated / output rows
constexpr int kWBCells = 4;        // cells (pairs of rows) in flight per wave (8: 154 VGPRs, or spills under a 128 cap)

__host__ __device__ inline size_t wordbwd_lds_bytes(int, int) {
  return sizeof(float) * ((size_t)2 * kWBWords * kWBAP + 2 * 8 * 2 * 64);
}

// Blocks [0, G), G = ceil(n_words / 32): word workgroups (32 words each: bid + j G).
// Blocks [G, G + ntile): one 16 x 16 tile of dWd = h^T.dz each, rows split over the 8 waves.
__global__ void __launch_bounds__(512) tower_wordbwd_kernel(TowerArgs a, CellsView cv, int nfused, float lr_over_L) {
  extern __shared__ __attribute__((aligned(16))) float wb_smem[];
  __shared__ float sMu[128], sInv[128], sMul[128], sAdd[128], sA[128], sBq[128];
  __shared__ f32x4 sAcc[8 * 64];
  __shared__ float sCol[8 * 16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nv = valid_rows(a);
  const int d = a.d, hd = a.hd;
  const int G = (a.n_words + kWBWords - 1) / kWBWords;
  if (tid < 128) {
    float mu = 0.f, inv = 0.f, mul = 0.f, add = 0.f, t1 = 0.f, t2 = 0.f;
    if (tid < d) {
      const float* mi = a.ws + ws_muinv();
      mu = mi[tid];
      inv = mi[128 + tid];
      mul = mi[384 + tid];
      add = mi[512 + tid];
      merge_partials(a.ws + ws_bsum(), nfused, tid, t1, t2);  // per item workgroup of the step kernel, in order
    }
    sMu[tid] = mu;
    sInv[tid] = inv;
    sMul[tid] = mul;   // gamma * inv; 0 past d: those columns of dz are exactly 0
    sAdd[tid] = add;
    sA[tid] = t1;      // sum dy       (= dbeta)
    sBq[tid] = t2;     // sum dy*xhat  (= dgamma)
  }
  const float n = (float)nv, rn = 1.f / n;
  const int x = lane & 15, kq = lane >> 4;
  // word workgroups: cell ranges of the 8 words, requested with the statistics (threads 128..135)
  int pre_c0 = 0, pre_c1 = 0;
  float pre_sc = 0.f;
  if ((int)blockIdx.x < G && tid >= 128 && tid < 128 + kWBWords) {
    const int w = (int)blockIdx.x + (tid - 128) * G;
    pre_sc = (w < a.n_words) ? word_scale(a, w) : 0.f;
    if (pre_sc != 0.f) {   // dropped words: no gradient, no update
      pre_c0 = cv.off[w];
      pre_c1 = cv.off[w + 1];
    }
  }

  if ((int)blockIdx.x >= G) {
    // ---- dWd tile (it, jt): dWd[kw][c] = sum_r h[r][kw] dz[r][c]; then the optimiser step of that tile ----
    __syncthreads();
    const int tile = (int)blockIdx.x - G;
    const int nct = (d + 15) >> 4;
    const int it = tile / nct, jt = tile % nct;
    const int kw = it * 16 + x, cn = jt * 16 + x;  // A lane: h column kw; B lane: dz column cn
    const float mu = sMu[cn], inv = sInv[cn], mul = sMul[cn], add = sAdd[cn], sa = sA[cn], sbq = sBq[cn];
    const float coef = mul * rn;
    const bool okh = kw < hd, okc = cn < d;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float colsum = 0.f;
    for (int st0 = wave; 4 * st0 < nv; st0 += 64) {  // waves take the 4-row k-steps round robin; 8 steps in flight
      float hv[8], zv[8], gv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = 4 * (st0 + 8 * u) + kq;
        const bool ok = r < nv;
        hv[u] = (ok && okh) ? a.h[(int64_t)r * hd + kw] : 0.f;
        zv[u] = (ok && okc) ? a.z[(int64_t)r * d + cn] : 0.f;
        gv[u] = (ok && okc) ? a.dA[(int64_t)r * d + cn] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = 4 * (st0 + 8 * u) + kq;
        const float xh = (zv[u] - mu) * inv;
        const float dy = (fmaf(zv[u], mul, add) > 0.f) ? gv[u] : 0.f;
        const float dzv = (r < nv) ? coef * (n * dy - sa - xh * sbq) : 0.f;
        colsum += dzv;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[u], dzv, acc, 0, 0, 0);
      }
    }
    sAcc[wave * 64 + lane] = acc;
    colsum += __shfl_xor(colsum, 16);
    colsum += __shfl_xor(colsum, 32);
    if (kq == 0) sCol[wave * 16 + x] = colsum;
    __syncthreads();
    if (wave == 0) {
      f32x4 tot = sAcc[lane];
#pragma unroll
      for (int v = 1; v < 8; ++v) tot += sAcc[v * 64 + lane];
      const size_t nW = (size_t)a.n_words * a.wd, nel = (size_t)hd * d;
#pragma unroll
      for (int r = 0; r < 4; ++r) {  // D[i = 4kq + r][j = x] = dWd[it*16 + i][cn]
        const int kwi = it * 16 + 4 * kq + r;
        if (kwi < hd && okc) {
          const size_t el = (size_t)kwi * d + cn;
          const float gw = tot[r];
          if (a.gexp) a.gexp[nW + el] = gw;
          else a.Wd_next[el] = a.opt ? adam1(gw, a.mD + el, a.vD + el, a.Wd[el], a) : a.Wd[el] - a.lr * gw;
        }
      }
      if (it == 0 && kq == 0 && okc) {  // bd, gamma, beta: nobody else in this launch reads them
        float gb = sCol[x];
#pragma unroll
        for (int v = 1; v < 8; ++v) gb += sCol[v * 16 + x];
        const float t1 = sa, t2 = sbq;
        if (a.gexp) {
          float* gd = a.gexp + nW + nel;
          gd[cn] = gb;
          gd[128 + cn] = t2;
          gd[256 + cn] = t1;
        } else if (a.opt) {
          float *mm = a.mD + nel, *vv = a.vD + nel;
          a.bd[cn] = adam1(gb, mm + cn, vv + cn, a.bd[cn], a);
          a.gamma[cn] = adam1(t2, mm + 128 + cn, vv + 128 + cn, a.gamma[cn], a);
          a.beta[cn] = adam1(t1, mm + 256 + cn, vv + 256 + cn, a.beta[cn], a);
        } else {
          a.bd[cn] -= a.lr * gb;
          a.beta[cn] -= a.lr * t1;
          a.gamma[cn] -= a.lr * t2;
        }
      }
    }
    return;
  }

  // ---- word workgroup -----------------------------------------------------------------------------
  // 32 words per workgroup (word bid + j G, j < 32: frequent words have neighbouring ids, striding spreads them),
  // ~300 workgroups: one round on the chip.  The cells of the 32 words form ONE list cut into 8 equal parts, one per
  // wave (a frequent word -- the pad id occurs in every item -- would otherwise serialise on one wave).  A wave
  // accumulates count * dz[slot] per word in registers; a word whose cells all lie inside the wave's part goes
  // straight to its row of the MFMA operand, a word cut by a part boundary leaves a partial (at most two per wave)
  // that the word's finishing wave adds in wave order.  Fixed summation order, no float atomics.
  float* sAccA = wb_smem;                                   // [32][kWBAP]  accumulated dz-space rows (MFMA A operand)
  float* sOutW = sAccA + kWBWords * kWBAP;                  // [32][kWBAP]  rows . Wd^T
  float2* sPart = reinterpret_cast<float2*>(sOutW + kWBWords * kWBAP);   // [8 waves][2][64 lanes]
  __shared__ int sC0[kWBWords], sPre[kWBWords + 1], sPartWord[16];
  __shared__ float sScale[kWBWords];
  const int bid = (int)blockIdx.x;
  const int d2 = a.wd >> 1, dz2 = d >> 1;
  constexpr int kPerWave = kWBWords / 8;   // words a wave finishes
  // Wd^T fragments of this wave's 16 output columns (kw = 16 wave + x), k-permuted like the A operand: the
  // float4 at column 16t + 4kq serves four MFMAs.  Requested first: nothing depends on them until the end.
  const int ng = (d + 15) >> 4;
  float4 bw[8];
  {
    const int kw = 16 * wave + x;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int col = 16 * t + 4 * kq;
      bw[t] = (t < ng && kw < hd && col < d) ? reinterpret_cast<const float4*>(a.Wd + (int64_t)kw * d)[col >> 2] : zero4();
    }
  }
  float2 wold[kPerWave];
#pragma unroll
  for (int i = 0; i < kPerWave; ++i) {
    const int w = bid + (kPerWave * wave + i) * G;
    wold[i] = (w < a.n_words && lane < d2) ? reinterpret_cast<const float2*>(a.W + (int64_t)w * a.wd)[lane] : make_float2(0.f, 0.f);
  }
  if (tid >= 128 && tid < 128 + kWBWords) {
    sC0[tid - 128] = pre_c0;
    sScale[tid - 128] = pre_sc;
    sPre[tid - 128 + 1] = pre_c1 - pre_c0;
  }
  if (tid < 16) sPartWord[tid] = -1;
  for (int i = tid; i < kWBWords * kWBAP; i += 512) sAccA[i] = 0.f;
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    sPre[0] = 0;
    for (int j = 0; j < kWBWords; ++j) {
      const int len = sPre[j + 1];
      sPre[j + 1] = run + len;
      run += len;
    }
  }
  __syncthreads();
  const int T = sPre[kWBWords];
  const int per = (T + 7) >> 3;
  const int f_beg = wave * per, f_end = min(T, f_beg + per);
  // column constants of this lane's two dz columns
  const int ca = 2 * lane, cb = 2 * lane + 1;
  const bool lok = lane < dz2;
  const float muA = lok ? sMu[ca] : 0.f, muB = lok ? sMu[cb] : 0.f, ivA = lok ? sInv[ca] : 0.f, ivB = lok ? sInv[cb] : 0.f;
  const float mlA = lok ? sMul[ca] : 0.f, mlB = lok ? sMul[cb] : 0.f, adA = lok ? sAdd[ca] : 0.f, adB = lok ? sAdd[cb] : 0.f;
  const float saA = lok ? sA[ca] : 0.f, saB = lok ? sA[cb] : 0.f, sqA = lok ? sBq[ca] : 0.f, sqB = lok ? sBq[cb] : 0.f;
  const float2* z2 = reinterpret_cast<const float2*>(a.z);
  const float2* g2 = reinterpret_cast<const float2*>(a.dA);
  // flat index -> (word j, address): uniform across the wave (binary search in the 33 prefix sums)
  auto locate = [&](int fidx, int& j) -> int {
    int lo = 0;
#pragma unroll
    for (int step = kWBWords / 2; step >= 1; step >>= 1)
      if (fidx >= sPre[lo + step]) lo += step;
    j = lo;
    return sC0[j] + (fidx - sPre[j]);
  };
  int ent[kWBCells], wj[kWBCells];
  auto load_ents = [&](int f0, int* e_out, int* j_out) {
#pragma unroll
    for (int u = 0; u < kWBCells; ++u) {
      int j = 0;
      const bool ok = f0 + u < f_end;
      const int addr = ok ? locate(f0 + u, j) : 0;
      e_out[u] = ok ? cv.cells[addr] : -1;
      j_out[u] = j;
    }
  };
  // a finished run of word j over flat cells [r0, r1): whole word -> its operand row; cut -> a partial slot
  auto flush = [&](int j, int r0, int r1, float2 run) {
    if (j < 0) return;
    const bool whole = r0 == sPre[j] && r1 == sPre[j + 1];
    if (whole) {
      if (lok) reinterpret_cast<float2*>(sAccA + j * kWBAP)[lane] = run;
    } else {
      const int k = (r0 != sPre[j]) ? 0 : 1;   // 0: the word began in an earlier wave's part; 1: it continues in a later one
      sPart[(wave * 2 + k) * 64 + lane] = run;
      if (lane == 0) sPartWord[wave * 2 + k] = j;
    }
  };
  load_ents(f_beg, ent, wj);
  int curj = -1, run0 = f_beg;
  float2 run = make_float2(0.f, 0.f);
  for (int f0 = f_beg; f0 < f_end; f0 += kWBCells) {
    float2 zz[kWBCells], gg[kWBCells];
#pragma unroll
    for (int u = 0; u < kWBCells; ++u) {
      const bool ok = ent[u] >= 0 && lok;
      const int64_t row = (int64_t)(ent[u] >> 8) * dz2 + lane;
      zz[u] = ok ? z2[row] : make_float2(0.f, 0.f);
      gg[u] = ok ? g2[row] : make_float2(0.f, 0.f);
    }
    int entn[kWBCells], wjn[kWBCells];
    load_ents(f0 + kWBCells, entn, wjn);   // the next round's cells travel with this round's rows
#pragma unroll
    for (int u = 0; u < kWBCells; ++u) {
      if (ent[u] >= 0) {  // uniform across the wave
        if (wj[u] != curj) {
          flush(curj, run0, f0 + u, run);
          curj = wj[u];
          run0 = f0 + u;
          run = make_float2(0.f, 0.f);
        }
        const float cf = (float)(ent[u] & 255);
        const float dyA = (fmaf(zz[u].x, mlA, adA) > 0.f) ? gg[u].x : 0.f;
        const float dyB = (fmaf(zz[u].y, mlB, adB) > 0.f) ? gg[u].y : 0.f;
        const float dzA = mlA * rn * (n * dyA - saA - (zz[u].x - muA) * ivA * sqA);
        const float dzB = mlB * rn * (n * dyB - saB - (zz[u].y - muB) * ivB * sqB);
        run.x = fmaf(cf, dzA, run.x);
        run.y = fmaf(cf, dzB, run.y);
      }
    }
#pragma unroll
    for (int u = 0; u < kWBCells; ++u) {
      ent[u] = entn[u];
      wj[u] = wjn[u];
    }
  }
  flush(curj, run0, f_end, run);
  __syncthreads();
  // words cut by part boundaries: their partials, in (wave, slot) order, by the word's finishing wave
#pragma unroll
  for (int i = 0; i < kPerWave; ++i) {
    const int j = kPerWave * wave + i;
    float2 sum = make_float2(0.f, 0.f);
    bool any = false;
    for (int p = 0; p < 16; ++p) {
      if (sPartWord[p] == j) {   // uniform across the wave
        const float2 t = sPart[p * 64 + lane];
        sum.x += t.x;
        sum.y += t.y;
        any = true;
      }
    }
    if (any && lok) reinterpret_cast<float2*>(sAccA + j * kWBAP)[lane] = sum;
  }
  __syncthreads();
  // out[word][kw] = sum_c acc[word][c] Wd[kw][c]: wave = 16-column tile of kw, both 16-row tiles of the 32 words
  if (wave * 16 < hd) {
#pragma unroll
    for (int mt = 0; mt < kWBWords / 16; ++mt) {
      const float4* pa4 = reinterpret_cast<const float4*>(sAccA + (16 * mt + x) * kWBAP + 4 * kq);
      f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        if (t < ng) {
          const float4 af = pa4[4 * t];
          o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af.x, bw[t].x, o0, 0, 0, 0);
          o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af.y, bw[t].y, o1, 0, 0, 0);
          o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af.z, bw[t].z, o0, 0, 0, 0);
          o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af.w, bw[t].w, o1, 0, 0, 0);
        }
      }
      const f32x4 o = o0 + o1;
      if (16 * wave + x < hd) {
#pragma unroll
        for (int r = 0; r < 4; ++r) sOutW[(16 * mt + 4 * kq + r) * kWBAP + 16 * wave + x] = o[r];  // D[i = word][j = kw]
      }
    }
  }
  __syncthreads();
  if (lane >= d2) return;
#pragma unroll
  for (int i = 0; i < kPerWave; ++i) {
    const int j = kPerWave * wave + i;
    const int my_w = bid + j * G;
    if (my_w >= a.n_words) continue;
    const bool hit = sPre[j + 1] > sPre[j];
    const float scale = sScale[j];
    float2 wv = wold[i];
    if (a.gexp) {
      float2 gout = make_float2(0.f, 0.f);
      if (hit) {
        const float2 sum = reinterpret_cast<const float2*>(sOutW + j * kWBAP)[lane];
        const float gs = scale / (float)a.L;
        gout = make_float2(gs * sum.x, gs * sum.y);
      }
      reinterpret_cast<float2*>(a.gexp + (int64_t)my_w * a.wd)[lane] = gout;
      continue;
    }
    if (hit) {
      const float2 sum = reinterpret_cast<const float2*>(sOutW + j * kWBAP)[lane];
      if (a.opt) {
        const float gs = scale / (float)a.L;
        float2* mp = reinterpret_cast<float2*>(a.aw.m + (int64_t)my_w * a.wd) + lane;
        float2* vp = reinterpret_cast<float2*>(a.aw.v + (int64_t)my_w * a.wd) + lane;
        float2 mm = *mp, vv = *vp;
        wv.x = adam1(gs * sum.x, &mm.x, &vv.x, wv.x, a);
        wv.y = adam1(gs * sum.y, &mm.y, &vv.y, wv.y, a);
        *mp = mm;
        *vp = vv;
      } else {
        const float sc = lr_over_L * scale;
        wv.x = fmaf(-sc, sum.x, wv.x);
        wv.y = fmaf(-sc, sum.y, wv.y);
      }
      reinterpret_cast<float2*>(a.W + (int64_t)my_w * a.wd)[lane] = wv;
    } else if (a.opt == 2) {  // dense Adam (Keras): a word without a gradient still takes the zero-gradient step
      float2* mp = reinterpret_cast<float2*>(a.aw.m + (int64_t)my_w * a.wd) + lane;
      float2* vp = reinterpret_cast<float2*>(a.aw.v + (int64_t)my_w * a.wd) + lane;
      float2 mm = *mp, vv = *vp;
      wv.x = adam1(0.f, &mm.x, &vv.x, wv.x, a);
      wv.y = adam1(0.f, &mm.y, &vv.y, wv.y, a);
      *mp = mm;
      *vp = vv;
      reinterpret_cast<float2*>(a.W + (int64_t)my_w * a.wd)[lane] = wv;
    }
  }
}

int launch_wordbwd(const TowerArgs& a, const CellsView& cv, int nfused, hipStream_t st) {
  const int G = (a.n_words + kWBWords - 1) / kWBWords;
  const int ntile = ((a.hd + 15) / 16) * ((a.d + 15) / 16);
  const size_t lds = wordbwd_lds_bytes(a.hd, a.d);
  static bool attr_set = false;
  if (!attr_set) {
    NNCF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&tower_wordbwd_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)wordbwd_lds_bytes(128, 128)));
    attr_set = true;
  }
  {
    ProfScope _ps("tower_wordbwd_kernel", st);
    hipLaunchKernelGGL(tower_wordbwd_kernel, dim3(G + ntile), dim3(512), lds, st, a, cv, nfused, a.lr / (float)a.L);
  }
  NNCF_HIP_CHECK(hipGetLastError());
  return NNCF_OK;
}

}  // namespace nncf
