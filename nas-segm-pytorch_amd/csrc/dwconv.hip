// Depthwise k x k convolution (groups == channels), fp32 NHWC, gfx950.
//
// Reference call sites: SepConv / DilConv depthwise stage
// (src/nn/layer_factory.py:198-265) and InvertedResidual's 3x3 depthwise
// (src/nn/layer_factory.py:125-158).
//
// The op is HBM-bound (2-6 FLOP/B).  The fast path is a "vertical strip"
// kernel: lanes are laid along the flattened (x, channel/4) axis of one output
// row so every wave load is one contiguous run of float4s, and each thread
// produces P output rows that are spaced so that they share input rows:
// P outputs need (P-1)*E+K row visits instead of P*K.  Weights (K*K float4)
// stay in registers.  No LDS: the K horizontal taps of neighbouring lanes
// overlap and are served by the vector L1.
//
// Packed weight layout used by all kernels here: wt[tap][C] (tap = ty*K+tx).
//
// Fusions around the conv (all optional): an input *prologue* act(x*in_scale+in_shift)
// applied as the taps are loaded - this is how a preceding BatchNorm+ReLU(6) is consumed
// without ever writing the normalised tensor (and DilConv's leading ReLU); an output
// affine+act epilogue (inference); and a *statistics* epilogue (per-workgroup sum and sum
// of squares per channel of y), i.e. the batch statistics of a following BatchNorm.
#include <math.h>
#include <atomic>

#include <vector>

#include "dw_common.h"

// output rows per thread of the forward strip kernel: 3x3 with consecutive input rows (stride 1, any
// dilation) 8 - ten input rows for eight outputs instead of six for four (144ch 256x512: 157 -> 140 us, 32ch
// 512x1024: 150 -> 128 us); at stride 2 the longer strip loses (96ch 512x1024: 210 -> 221 us) and 5x5 has no
// registers for it: 4
#ifndef NASSEG_DW_P3
#define NASSEG_DW_P3 8
#endif
// (maps of fewer than 128 rows keep 4: the launch is short of workgroups there, not of bandwidth)
// output columns per thread of the forward strip kernel (dw_fwd_strip's PX): dilated 5x5 at stride 1 on maps at least
// four column groups wide (round 4, us PX 1 -> 4: 64 channels 256x512 dilation 6 124 -> 98, 32 channels
// 128x256 dilation 6 21.5 -> 20; at dilation 1 the taps of a row share cache lines anyway: 17.8 -> 18.7)
#ifndef NASSEG_DW5_P
#define NASSEG_DW5_P 4
#endif
#ifndef NASSEG_DW5_PX
#define NASSEG_DW5_PX 4
#endif
#ifndef NASSEG_DW5_WPX  // ... of the backward-weight kernels (25 accumulators of their own: 4 columns leave one wave per SIMD)
#define NASSEG_DW5_WPX 2
#endif
static inline int dw_fwd_px(int K, int stride, int dil, int Wo, int px_max = NASSEG_DW5_PX) {
  return (K == 5 && stride == 1 && dil > 1 && Wo >= 4 * px_max * dil) ? px_max : 1;
}
// threads along x (times C/4) of the forward strip kernel
static inline int dw_fwd_xgroups(int K, int stride, int dil, int Wo, int px_max = NASSEG_DW5_PX) {
  const int px = dw_fwd_px(K, stride, dil, Wo, px_max);
  return px == 1 ? Wo : ((Wo + px * dil - 1) / (px * dil)) * dil;
}
static inline int dw_fwd_rows(int K, int e, int Ho, int px = 1) {
  if (px > 1) return NASSEG_DW5_P;
  return (K == 3 && e == 1 && Ho >= 128) ? NASSEG_DW_P3 : 4;
}

#if NASSEG_FP32_ONLY
std::atomic<int> g_dw_wgrad_lds{1};
#else
extern std::atomic<int> g_dw_wgrad_lds;
#endif

namespace {

// BatchNorm whose backward statistics a backward-data kernel gathers in its epilogue: the
// value written is g' = g * act'(scale*z + shift) and the workgroup's partial sums of g' and
// g' * (z - mean) * invstd go to stats[blk][2][C] (see nasseg_dwconv_bwd_data_bn).
struct BnBwd {
  const act_t* z;
  const float* scale;
  const float* shift;
  const float* mean;
  const float* invstd;
  int act;
};
struct BnBwdLane {
  float4 sc, sh, mu, is;
};
__device__ __forceinline__ BnBwdLane bnbwd_lane(const BnBwd& bn, int c4) {
  BnBwdLane l;
  l.sc = lda4(bn.scale + c4 * 4);
  l.sh = lda4(bn.shift + c4 * 4);
  l.mu = lda4(bn.mean + c4 * 4);
  l.is = lda4(bn.invstd + c4 * 4);
  return l;
}
// masks g in place; adds keep_if(g', ok) and its product with xhat to ssum[0], ssum[1]
__device__ __forceinline__ void bnbwd_accumulate(float4& g, float4 z, const BnBwdLane& l, int act,
                                                 bool ok, float4 (&ssum)[2]) {
  const float4 yv = fma4(z, l.sc, l.sh);
  g.x *= act_mask(yv.x, act);
  g.y *= act_mask(yv.y, act);
  g.z *= act_mask(yv.z, act);
  g.w *= act_mask(yv.w, act);
  const float4 v = keep_if(g, ok);
  const float4 xh = make_float4((z.x - l.mu.x) * l.is.x, (z.y - l.mu.y) * l.is.y,
                                (z.z - l.mu.z) * l.is.z, (z.w - l.mu.w) * l.is.w);
  ssum[0] = add4(ssum[0], v);
  ssum[1] = fma4(v, xh, ssum[1]);
}

__device__ __forceinline__ int cdiv_dev(int a, int b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------------------
// weight packing: (C,1,K,K) -> [tap][C], optional 180-degree flip
// ---------------------------------------------------------------------------
__global__ void dw_pack_kernel(const float* __restrict__ w, float* __restrict__ wt, int C, int KK, int flip) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= C * KK) return;
  int t = i / C, c = i - t * C;
  int ts = flip ? (KK - 1 - t) : t;
  wt[i] = w[c * KK + ts];
}

// ---------------------------------------------------------------------------
// forward strip kernel
// ---------------------------------------------------------------------------
// WLDS: keep the K*K per-channel weight vectors in LDS (indexed by channel group,
// needs C4 <= 64) instead of 4*K*K registers per lane - for 5x5 this is the
// difference between 1 and 4+ resident waves per SIMD.
// PRO: input prologue; STATS == 1: per-workgroup channel sums of y and y^2 to
// stats[blk][2][C]; STATS == 2: BatchNorm-backward statistics of `bn` (y is a gradient).
// PX > 1 (stride 1 only): a thread owns PX output COLUMNS as well, `dil` pixels apart, so that they share input
// columns the way its P rows share input rows: (P - 1 + K) x (PX - 1 + K) loads for P x PX outputs - 4 per output
// for 5x5 with P = PX = 4 instead of 10 - each prologue applied once per load, and one weight read from LDS serves PX
// multiplies.  The 5x5 kernels of the decoders (dilation 1 ... 12) were bound by exactly those three: 40 L1 loads,
// 120 prologue operations and 100 LDS reads per four outputs (64 channels 256x512 dilation 6: 124 us = 2.2 TB/s).
template <int K, int P, int E, bool WLDS, bool PRO, int STATS, int PX = 1>
__global__ __launch_bounds__(256) void dw_fwd_strip(
    const act_t* __restrict__ x, const float* __restrict__ wt, act_t* __restrict__ y,
    const float* __restrict__ in_scale, const float* __restrict__ in_shift, int in_act,
    const float* __restrict__ scale, const float* __restrict__ shift, int H, int W, int C4, int Ho,
    int Wo, int stride, int pad, int dil, int g, int nchunk, int act, float* __restrict__ stats,
    BnBwd bn) {
  __shared__ float4 lw[WLDS ? K * K : 1][WLDS ? 64 : 1];
  __shared__ float4 sred[STATS ? 2 : 1][STATS ? 4 : 1][STATS ? 64 : 1];
  const int C = C4 * 4;
  if (WLDS) {
    for (int i = threadIdx.x; i < K * K * C4; i += 256) {
      const int t = i / C4, c = i - t * C4;
      lw[t][c] = lda4(wt + (size_t)t * C + c * 4);
    }
    __syncthreads();
  }
  // (An XCD-aware tile order - XCD k working through the k-th eighth of the tiles so that tiles sharing halo rows and
  //  columns meet in ONE L2 - brought the fabric reads of the dilated 5x5 kernel from 2.7x to 1.05x the tensor and
  //  changed its time by nothing, rounds 3-4: the kernel is issue-bound and the Infinity Cache serves the re-reads.
  //  Removed in round 5.)
  const int bx = blockIdx.x, by = blockIdx.y, b = blockIdx.z;
  const int r = by % g, chunk = by / g;
  const int base = bx * 256;
  const int idx = base + threadIdx.x;
  const int oy0 = chunk * (P * g) + r;
  // column groups: PX == 1: one per output column; else group xg owns columns ox + i * dil, i < PX
  const int XG = PX == 1 ? Wo : ((Wo + PX * dil - 1) / (PX * dil)) * dil;
  const int idc = idx < XG * C4 ? idx : 0;
  const int xg = idc / C4;
  const int c4 = idc - xg * C4;
  const int ox = PX == 1 ? xg : (xg / dil) * (PX * dil) + xg % dil;
  const bool live = (idx < XG * C4) && (oy0 < Ho) && (ox < Wo);
  if (!STATS && !live) return;
  // (with STATS every thread stays for the workgroup reduction; dead threads work on a
  //  clamped, valid position and contribute zeros)

  float4 w[WLDS ? 1 : K * K];
  if (!WLDS) {
#pragma unroll
    for (int t = 0; t < K * K; ++t) w[t] = lda4(wt + (size_t)t * C + c4 * 4);
  }
  Prologue pro;
  if (PRO) pro = make_prologue(in_scale, in_shift, in_act, c4);

  // Out-of-range taps load from a clamped (always valid) address and are zeroed by a
  // mask afterwards: a load under a branch would be followed by its own
  // s_waitcnt vmcnt(0) and serialise the whole strip on memory latency.
  constexpr int KX = K + PX - 1;  // input columns a thread visits per row
  int xoff[KX];
  bool xok[KX];
#pragma unroll
  for (int tx = 0; tx < KX; ++tx) {
    const int ix = ox * stride - pad + tx * dil;
    xok[tx] = (ix >= 0) && (ix < W);
    xoff[tx] = (ix < 0 ? 0 : (ix >= W ? W - 1 : ix)) * C;
  }
  float4 acc[P][PX];
#pragma unroll
  for (int j = 0; j < P; ++j)
#pragma unroll
    for (int i = 0; i < PX; ++i) acc[j][i] = f4zero();

  const act_t* xb = x + (size_t)b * H * W * C + c4 * 4;
  const int iy0 = oy0 * stride - pad;
  constexpr int Q = (P - 1) * E + K;
  // One input row ahead: the loads of row q+1 are issued before the FMAs of row q and a
  // scheduling barrier keeps the compiler from hoisting every load of the (fully
  // unrolled) strip to the top, which would cost ~4*K*Q registers and all occupancy.
  auto load_row = [&](int q, float4* v) {
    const int iy = iy0 + q * dil;
    const bool yok = (iy >= 0) && (iy < H);
    const act_t* xr = xb + (size_t)(iy < 0 ? 0 : (iy >= H ? H - 1 : iy)) * W * C;
#pragma unroll
    for (int tx = 0; tx < KX; ++tx) {
      float4 t = lda4(xr + xoff[tx]);
      if (PRO) t = apply_prologue(t, pro);
      v[tx] = keep_if(t, yok && xok[tx]);  // zero padding applies to the prologue's output
    }
  };
  float4 vcur[KX], vnext[KX];
  load_row(0, vcur);
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    if (q + 1 < Q) load_row(q + 1, vnext);
#pragma unroll
    for (int j = 0; j < P; ++j) {
      const int ty = q - j * E;
      if (ty >= 0 && ty < K) {
#pragma unroll
        for (int tx = 0; tx < K; ++tx) {
          const float4 wv = WLDS ? lw[ty * K + tx][c4] : w[WLDS ? 0 : ty * K + tx];
#pragma unroll
          for (int i = 0; i < PX; ++i) acc[j][i] = fma4(wv, vcur[tx + i], acc[j][i]);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < P; ++j)
#pragma unroll
      for (int i = 0; i < PX; ++i) pin(acc[j][i]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int tx = 0; tx < KX; ++tx) vcur[tx] = vnext[tx];
  }
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = f4zero();
  if (scale) sc = lda4(scale + c4 * 4);
  if (shift) sh = lda4(shift + c4 * 4);
  float4 ssum[2] = {f4zero(), f4zero()};
  BnBwdLane bl;
  if (STATS == 2) bl = bnbwd_lane(bn, c4);
#pragma unroll
  for (int j = 0; j < P; ++j) {
    const int oy = oy0 + j * g;
#pragma unroll
    for (int i = 0; i < PX; ++i) {
      const int oxi = ox + i * dil;
      const bool ok = live && oy < Ho && (PX == 1 || oxi < Wo);
      float4 o = fma4(acc[j][i], sc, sh);
      if (act) o = act_apply4(o, act);
      if (STATS == 1) {
        const float4 m = keep_if(o, ok);
        ssum[0] = add4(ssum[0], m);
        ssum[1] = fma4(m, m, ssum[1]);
      }
      if (STATS == 2) {
        const int oyc = oy < Ho ? oy : Ho - 1;  // (unconditional load from a valid address)
        const int oxc = (PX == 1 || oxi < Wo) ? oxi : Wo - 1;
        const float4 z = lda4(bn.z + (((size_t)b * Ho + oyc) * Wo + oxc) * C + c4 * 4);
        bnbwd_accumulate(o, z, bl, bn.act, ok, ssum);
      }
      if (ok) sta4(y + (((size_t)b * Ho + oy) * Wo + oxi) * C + c4 * 4, o);
    }
  }
  if constexpr (STATS != 0) {
    const size_t blk = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    block_reduce_groups<2, 2>(ssum, sred, stats + blk * 2 * C, base, C4);
  }
}

// ---------------------------------------------------------------------------
// backward-data of a stride-2, dilation-1, pad=(K-1)/2 depthwise conv (the only
// strided form on the path).  One thread produces the 2x2 quad of input-gradient
// pixels (2a+py, 2b+px): all four read the same small patch of dy, and which
// taps contribute to which quad position is a compile-time parity pattern:
//   dx[2a+py] += w[ty] * dy[a + (py+pad-ty)/2]   for ty = (py+pad) mod 2, +2, ...
// ---------------------------------------------------------------------------
// BST: BatchNorm-backward statistics epilogue (the host makes gridDim.x * 256 a multiple of
// C4 so that a thread keeps its channel group over the grid-stride loop).
template <int K, bool BST>
__global__ __launch_bounds__(256) void dw_bwd_data_s2(const act_t* __restrict__ dy,
                                                      const float* __restrict__ wt,
                                                      act_t* __restrict__ dx, int B, int Ho, int Wo,
                                                      int C4, int H, int W, float* __restrict__ stats,
                                                      BnBwd bn) {
  __shared__ float4 sred[BST ? 2 : 1][BST ? 4 : 1][BST ? 64 : 1];
  float4 ssum[2] = {f4zero(), f4zero()};
  BnBwdLane bl;
  if (BST) bl = bnbwd_lane(bn, (int)(((int64_t)blockIdx.x * 256 + threadIdx.x) % C4));
  constexpr int PAD = (K - 1) / 2;
  // dy rows used by a quad: a + (py+PAD-ty)/2 over all valid (py,ty): [a - LO, a + HI]
  constexpr int LO = (K - 1 - PAD) / 2;  // largest ty with matching parity, py = 0 side
  constexpr int HI = (PAD + 1) / 2;
  constexpr int R = LO + HI + 1;
  const int C = C4 * 4;
  const int Hq = (H + 1) >> 1, Wq = (W + 1) >> 1;
  const int64_t total = (int64_t)B * Hq * Wq * C4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    int64_t p = i / C4;
    const int bq = (int)(p % Wq);
    p /= Wq;
    const int aq = (int)(p % Hq);
    const int b = (int)(p / Hq);
    float4 d[R][R];
#pragma unroll
    for (int ry = 0; ry < R; ++ry)
#pragma unroll
      for (int rx = 0; rx < R; ++rx) {
        const int oy = aq - LO + ry, ox = bq - LO + rx;
        const bool ok = oy >= 0 && oy < Ho && ox >= 0 && ox < Wo;
        const int oyc = oy < 0 ? 0 : (oy >= Ho ? Ho - 1 : oy), oxc = ox < 0 ? 0 : (ox >= Wo ? Wo - 1 : ox);
        d[ry][rx] = keep_if(lda4(dy + (((int64_t)b * Ho + oyc) * Wo + oxc) * C + c4 * 4), ok);
      }
    float4 o[2][2];
#pragma unroll
    for (int py = 0; py < 2; ++py)
#pragma unroll
      for (int px = 0; px < 2; ++px) o[py][px] = f4zero();
#pragma unroll
    for (int ty = 0; ty < K; ++ty)
#pragma unroll
      for (int tx = 0; tx < K; ++tx) {
        const int py = (ty + PAD) & 1, px = (tx + PAD) & 1;  // parity this tap feeds
        const int ry = (py + PAD - ty) / 2 + LO, rx = (px + PAD - tx) / 2 + LO;
        o[py][px] = fma4(lda4(wt + (size_t)(ty * K + tx) * C + c4 * 4), d[ry][rx], o[py][px]);
      }
#pragma unroll
    for (int py = 0; py < 2; ++py)
#pragma unroll
      for (int px = 0; px < 2; ++px) {
        const int iy = 2 * aq + py, ix = 2 * bq + px;
        const bool ok = iy < H && ix < W;
        if (BST) {
          const int iyc = iy < H ? iy : H - 1, ixc = ix < W ? ix : W - 1;
          const float4 z = lda4(bn.z + (((int64_t)b * H + iyc) * W + ixc) * C + c4 * 4);
          bnbwd_accumulate(o[py][px], z, bl, bn.act, ok, ssum);
        }
        if (ok) sta4(dx + (((int64_t)b * H + iy) * W + ix) * C + c4 * 4, o[py][px]);
      }
  }
  if constexpr (BST)
    block_reduce_groups<2, 2>(ssum, sred, stats + (size_t)blockIdx.x * 2 * C4 * 4, blockIdx.x * 256, C4);
}

// ---------------------------------------------------------------------------
// generic kernel: any K / stride / dilation; also the transposed (backward-
// data for stride > 1) form.  One thread per output float4.
//   forward   : y[oy,ox] = sum_t w[t] * x[oy*s - pad + ty*d, ox*s - pad + tx*d]
//   transposed: y[oy,ox] = sum_t w[t] * x[(oy + pad - ty*d)/s, (ox + pad - tx*d)/s]
//               (only taps whose numerator is a non-negative multiple of s)
// (H, W) are the dims of the tensor read, (Ho, Wo) of the tensor written.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dw_generic(
    const act_t* __restrict__ x, const float* __restrict__ wt, act_t* __restrict__ y,
    const float* __restrict__ scale, const float* __restrict__ shift, int B, int H, int W, int C4,
    int Ho, int Wo, int K, int stride, int pad, int dil, int transposed, int relu_in, int act) {  // relu_in: input ReLU
  const int C = C4 * 4;
  const size_t total = (size_t)B * Ho * Wo * C4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    size_t p = i / C4;
    const int ox = (int)(p % Wo);
    p /= Wo;
    const int oy = (int)(p % Ho);
    const int b = (int)(p / Ho);
    const act_t* xb = x + (size_t)b * H * W * C + c4 * 4;
    float4 acc = f4zero();
    for (int ty = 0; ty < K; ++ty) {
      int iy;
      bool yok;
      if (!transposed) {
        iy = oy * stride - pad + ty * dil;
        yok = iy >= 0 && iy < H;
      } else {
        int ny = oy + pad - ty * dil;
        yok = ny >= 0 && (ny % stride) == 0;
        iy = ny / stride;
        yok = yok && iy < H;
      }
      if (!yok) continue;
      for (int tx = 0; tx < K; ++tx) {
        int ix;
        bool ok;
        if (!transposed) {
          ix = ox * stride - pad + tx * dil;
          ok = ix >= 0 && ix < W;
        } else {
          int nx = ox + pad - tx * dil;
          ok = nx >= 0 && (nx % stride) == 0;
          ix = nx / stride;
          ok = ok && ix < W;
        }
        if (!ok) continue;
        float4 v = lda4(xb + ((size_t)iy * W + ix) * C);
        if (relu_in) v = relu4(v);
        acc = fma4(lda4(wt + (size_t)(ty * K + tx) * C + c4 * 4), v, acc);
      }
    }
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = f4zero();
    if (scale) sc = lda4(scale + c4 * 4);
    if (shift) sh = lda4(shift + c4 * 4);
    float4 o = act_apply4(fma4(acc, sc, sh), act);
    sta4(y + i * 4, o);
  }
}

// ---------------------------------------------------------------------------
// backward-weight strip kernel: per-thread K*K float4 accumulators over a set
// of row chunks, then a per-block reduction over lanes that share a channel
// group.  partial layout: [block][tap][C].  Deterministic (no atomics).
// ---------------------------------------------------------------------------
struct DwWgArgs {
  const act_t* x;
  const act_t* dy;
  float* partial;
  const float* in_scale;
  const float* in_shift;
  int in_act, B, H, W, C4, Ho, Wo, stride, pad, dil, g, nchunk;
  int lgx, lgy;  // PX > 1: the logical grid (lgx * lgy <= workgroups launched; the others zero their partial rows)
  // BN variant (nasseg_dwconv_wgrad_bn): dy is the masked gradient w.r.t. the BatchNorm output, z
  // the conv's raw output; dz = second half of the BatchNorm backward, computed on load and stored
  const act_t* z;
  act_t* dz;
  const float* bn_scale;
  const float* bn_shift;
  const float* bn_mean;
  const float* bn_invstd;
  const float* bn_sums;
  float invM;
  int bn_train, bn_act;
};

// (bx, by) of (gdx, gdy): the workgroup's coordinates in the layer's own grid - the grouped
// launch below runs several layers' grids side by side in one kernel
// PX > 1 (stride 1): PX columns a dilation apart per thread, as in dw_fwd_strip - (P - 1 + K) x (PX - 1 + K) loads of
// x and P x PX of dy for P x PX pixels instead of ((P - 1 + K) x K + P) x PX.
template <int K, int P, int E, bool PRO, bool BN = false, int PX = 1>
__device__ __forceinline__ void dw_wgrad_tile(const DwWgArgs& q, const int bx, const int by,
                                              const int gdx, const int gdy) {
  const act_t* __restrict__ x = q.x;
  const act_t* __restrict__ dy = q.dy;
  float* __restrict__ partial = q.partial;
  const int B = q.B, H = q.H, W = q.W, C4 = q.C4, Ho = q.Ho, Wo = q.Wo, stride = q.stride, pad = q.pad,
            dil = q.dil, g = q.g, nchunk = q.nchunk;
  __shared__ float4 red[K][4][64];
  const int tid = threadIdx.x;
  const int base = bx * 256;
  const int idx = base + tid;
  const int XG = PX == 1 ? Wo : ((Wo + PX * dil - 1) / (PX * dil)) * dil;
  const int idc = idx < XG * C4 ? idx : 0;
  const int xg = idc / C4;
  const int c4 = idc - xg * C4;
  const int ox = PX == 1 ? xg : (xg / dil) * (PX * dil) + xg % dil;
  const bool live = idx < XG * C4 && ox < Wo;
  const int C = C4 * 4;

  float4 acc[K * K];
#pragma unroll
  for (int t = 0; t < K * K; ++t) acc[t] = f4zero();
  Prologue pro;
  if (PRO) pro = make_prologue(q.in_scale, q.in_shift, q.in_act, c4);
  // dz = ca*g + cb*z + cd  ==  scale*(g - sum(g)/M - xhat*sum(g*xhat)/M)
  float4 ca = f4zero(), cb = f4zero(), cd = f4zero(), cs = f4zero();
  if (BN) {
    ca = lda4(q.bn_scale + c4 * 4);
    if (q.bn_act) cs = lda4(q.bn_shift + c4 * 4);  // g arrives without its activation mask
    if (q.bn_train) {
      const float4 is = lda4(q.bn_invstd + c4 * 4), mu = lda4(q.bn_mean + c4 * 4);
      const float4 s0 = lda4(q.bn_sums + c4 * 4), s1 = lda4(q.bn_sums + C + c4 * 4);
      const float m = q.invM;
      cb = make_float4(-ca.x * is.x * (s1.x * m), -ca.y * is.y * (s1.y * m), -ca.z * is.z * (s1.z * m),
                       -ca.w * is.w * (s1.w * m));
      cd = make_float4(ca.x * (mu.x * is.x * (s1.x * m) - s0.x * m), ca.y * (mu.y * is.y * (s1.y * m) - s0.y * m),
                       ca.z * (mu.z * is.z * (s1.z * m) - s0.z * m), ca.w * (mu.w * is.w * (s1.w * m) - s0.w * m));
    }
  }

  constexpr int KX = K + PX - 1;
  int xoff[KX];
  bool xok[KX];
#pragma unroll
  for (int tx = 0; tx < KX; ++tx) {
    const int ix = ox * stride - pad + tx * dil;
    xok[tx] = live && (ix >= 0) && (ix < W);
    xoff[tx] = (ix < 0 ? 0 : (ix >= W ? W - 1 : ix)) * C;  // clamped: loads are unconditional
  }
  constexpr int Q = (P - 1) * E + K;
  const int nwork = B * nchunk * g;
  for (int wk = by; wk < nwork; wk += gdy) {
    const int r = wk % g;
    int t2 = wk / g;
    const int chunk = t2 % nchunk;
    const int b = t2 / nchunk;
    const int oy0 = chunk * (P * g) + r;
    if (oy0 >= Ho || !live) continue;  // (wave-divergent only in the last x-block)
    float4 d[P][PX];
#pragma unroll
    for (int j = 0; j < P; ++j)
#pragma unroll
    for (int i = 0; i < PX; ++i) {
      const int oy = oy0 + j * g;
      const int oxi = ox + i * dil;
      const bool pok = oy < Ho && (PX == 1 || oxi < Wo);
      const size_t off = (((size_t)b * Ho + (oy < Ho ? oy : Ho - 1)) * Wo + ((PX == 1 || oxi < Wo) ? oxi : Wo - 1)) * C +
                         c4 * 4;
      float4 gv = lda4(dy + off);
      if (BN) {
        const float4 zv = lda4(q.z + off);
        if (q.bn_act) {
          const float4 y = fma4(zv, ca, cs);
          gv = make_float4(gv.x * act_mask(y.x, q.bn_act), gv.y * act_mask(y.y, q.bn_act),
                           gv.z * act_mask(y.z, q.bn_act), gv.w * act_mask(y.w, q.bn_act));
        }
        gv = fma4(gv, ca, fma4(zv, cb, cd));
#ifdef NASSEG_BF16
        gv = make_float4(bf16_to_f32(f32_to_bf16(gv.x)), bf16_to_f32(f32_to_bf16(gv.y)),
                         bf16_to_f32(f32_to_bf16(gv.z)), bf16_to_f32(f32_to_bf16(gv.w)));
#endif
        if (pok) sta4(q.dz + off, gv);  // (every dy element is loaded by exactly one lane)
      }
      d[j][i] = keep_if(gv, pok);
    }
    const act_t* xb = x + (size_t)b * H * W * C + c4 * 4;
    const int iy0 = oy0 * stride - pad;
    auto load_row = [&](int q, float4* v) {
      const int iy = iy0 + q * dil;
      const bool yok = (iy >= 0) && (iy < H);
      const act_t* xr = xb + (size_t)(iy < 0 ? 0 : (iy >= H ? H - 1 : iy)) * W * C;
#pragma unroll
      for (int tx = 0; tx < KX; ++tx) {
        float4 t = lda4(xr + xoff[tx]);
        if (PRO) t = apply_prologue(t, pro);
        v[tx] = keep_if(t, yok && xok[tx]);
      }
    };
    float4 vcur[KX], vnext[KX];
    load_row(0, vcur);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      if (q + 1 < Q) load_row(q + 1, vnext);
#pragma unroll
      for (int j = 0; j < P; ++j) {
        const int ty = q - j * E;
        if (ty >= 0 && ty < K) {
#pragma unroll
          for (int tx = 0; tx < K; ++tx)
#pragma unroll
            for (int i = 0; i < PX; ++i) acc[ty * K + tx] = fma4(d[j][i], vcur[tx + i], acc[ty * K + tx]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int tx = 0; tx < KX; ++tx) vcur[tx] = vnext[tx];
    }
  }
  // Block reduction without a serial tail.  Lanes l, l+C4, l+2*C4, ... of a wave hold
  // the same channel group: lanes < C4 gather them with shuffles in a fixed order; the
  // four waves then meet in a small LDS buffer indexed by channel group, K taps at a
  // time.  (C4 > 64: every lane of a wave owns a different group, the shuffle step is
  // the identity and a wave only covers 64 of the groups - see block_reduce_taps.)
  float* pout = partial + ((size_t)by * gdx + bx) * (size_t)(K * K) * C;
  block_reduce_groups<K * K, K>(acc, red, pout, base, C4);
}

template <int K, int P, int E, bool PRO, bool BN = false>
__global__ __launch_bounds__(256) void dw_wgrad_strip(DwWgArgs q) {
  dw_wgrad_tile<K, P, E, PRO, BN>(q, blockIdx.x, blockIdx.y, gridDim.x, gridDim.y);
}

// a workgroup without a tile (PX > 1: the partial buffer was sized for the one-column grid) zeroes its row
__device__ __forceinline__ void dw_wgrad_zero_row(float* partial, int row, int n) {
  float* po = partial + (size_t)row * n;
  for (int i = threadIdx.x; i < n; i += 256) po[i] = 0.f;
}
// the PX = NASSEG_DW5_PX form: a 1-D launch of as many workgroups as the partial buffer has rows
template <int K, int P, int E, bool PRO, bool BN, int PX>
__global__ __launch_bounds__(256, 2) void dw_wgrad_strip_px(DwWgArgs q) {
  const int id = blockIdx.x;
  if (id < q.lgx * q.lgy) dw_wgrad_tile<K, P, E, PRO, BN, PX>(q, id % q.lgx, id / q.lgx, q.lgx, q.lgy);
  else dw_wgrad_zero_row(q.partial, id, K * K * q.C4 * 4);
}

// several depthwise layers of one specialisation in one launch (see conv_wgrad_group_kernel)
constexpr int kDwGroup = 8;
struct DwWgGroup {
  int n;
  int start[kDwGroup + 1];
  int gx[kDwGroup], gy[kDwGroup];
  DwWgArgs a[kDwGroup];
};
template <int K, int P, int E, bool PRO, int PX = 1>
__global__ __launch_bounds__(256, 2) void dw_wgrad_group_kernel(DwWgGroup t) {
  int d = 0;
  while (d + 1 < t.n && (int)blockIdx.x >= t.start[d + 1]) ++d;
  const int local = blockIdx.x - t.start[d];
  if (PX == 1 || local < t.gx[d] * t.gy[d])
    dw_wgrad_tile<K, P, E, PRO, false, PX>(t.a[d], local % t.gx[d], local / t.gx[d], t.gx[d], t.gy[d]);
  else
    dw_wgrad_zero_row(t.a[d].partial, local, K * K * t.a[d].C4 * 4);
}

// ---------------------------------------------------------------------------
// backward-weight of a stride-1 K x K depthwise conv with the tiles staged in LDS.
// The strip kernel above keeps K*K float4 accumulators AND a (P - 1 + K) x K window of loads per thread: 200-250
// registers for 5x5, two waves per SIMD, every load normalised K times over and a dependent chain of P - 1 + K
// global-memory round trips per work item - 0.8-1.3 TB/s on the 5x5 layers of the decoders (3x3: 3.3-5.7).
// Here a workgroup stages a (TH + K - 1) x (16 + K - 1) patch of x - prologue applied ONCE per element, zero padding
// in place - and the TH x 16 tile of dy for CG channel groups in LDS, and a thread (one channel group, four
// consecutive pixels of a tile row) reads its K x (K + 3) window and four dy values from there: K + 3 + ... LDS reads
// per row of taps, no global latency inside the arithmetic.  A dilated conv is dil^2 independent dense convs on the
// pixel classes (y mod dil, x mod dil): tiles are cut in the subsampled coordinates of one class, so the patch is
// (TH + K - 1) x (16 + K - 1) pixels whatever the dilation (pad must be a multiple of it).
// Workgroup = (worker, channel chunk): a worker walks its share of the (image, class, tile) items with the
// accumulators in registers and leaves ONE partial row [K*K][C] (its chunks' workgroups each write their channels):
// the row count and layout of the strip kernel's partial buffer, so the callers' second stage is unchanged.
// ---------------------------------------------------------------------------
constexpr int kWlTW = 16;  // tile width (pixels of one class)
struct DwWlArgs {
  DwWgArgs a;       // tensors and geometry (stride 1)
  int CG, TH;       // channel groups per workgroup (<= 8, divides C4), tile rows = (256 / CG) / 4
  int nchunkC;      // C4 / CG
  int workers;      // partial rows
  int pd;           // pad / dil
  int Hs, Ws;       // rows / columns of a pixel class (upper bound over the classes)
  int ty_n, tx_n;   // tiles per class
};

template <int K, bool PRO, bool BN>
__global__ __launch_bounds__(256, 2) void dw_wgrad_lds_kernel(DwWlArgs w) {
  extern __shared__ float4 wl_lds[];
  const DwWgArgs& q = w.a;
  const int tid = threadIdx.x;
  const int CG = w.CG, TH = w.TH, dil = q.dil;
  const int C4 = q.C4, C = C4 * 4;
  const int PS = CG + 1;                   // pixel stride in LDS (float4): one group of padding against bank conflicts
  const int XW = kWlTW + K - 1;            // patch width
  float4* xs = wl_lds;                     // [(TH + K - 1) * XW][PS]
  float4* ds = wl_lds + (TH + K - 1) * XW * PS;  // [TH * kWlTW][PS]
  const int chunk = blockIdx.x % w.nchunkC;
  const int worker = blockIdx.x / w.nchunkC;
  const int c0 = chunk * CG;               // first channel group of this workgroup
  // compute role: channel group cc, tile row tr, pixels 4 * seg .. + 3
  const int cc = tid % CG;
  const int slot = tid / CG;
  const bool worker_thread = slot < TH * 4;
  const int tr = slot >> 2, seg = slot & 3;

  float4 acc[K * K];
#pragma unroll
  for (int t = 0; t < K * K; ++t) acc[t] = f4zero();

  // staging role: consecutive threads take consecutive channel groups of consecutive pixels
  Prologue pro;
  // (the prologue depends on the channel group a staging thread handles: groups repeat with period CG)
  const int sc_ = tid % CG;
  if (PRO) pro = make_prologue(q.in_scale, q.in_shift, q.in_act, c0 + sc_);
  float4 ca = f4zero(), cb = f4zero(), cd = f4zero(), cs = f4zero();
  if (BN) {
    const int c4 = c0 + sc_;
    ca = lda4(q.bn_scale + c4 * 4);
    if (q.bn_act) cs = lda4(q.bn_shift + c4 * 4);
    if (q.bn_train) {
      const float4 is = lda4(q.bn_invstd + c4 * 4), mu = lda4(q.bn_mean + c4 * 4);
      const float4 s0 = lda4(q.bn_sums + c4 * 4), s1 = lda4(q.bn_sums + C + c4 * 4);
      const float m = q.invM;
      cb = make_float4(-ca.x * is.x * (s1.x * m), -ca.y * is.y * (s1.y * m), -ca.z * is.z * (s1.z * m),
                       -ca.w * is.w * (s1.w * m));
      cd = make_float4(ca.x * (mu.x * is.x * (s1.x * m) - s0.x * m), ca.y * (mu.y * is.y * (s1.y * m) - s0.y * m),
                       ca.z * (mu.z * is.z * (s1.z * m) - s0.z * m), ca.w * (mu.w * is.w * (s1.w * m) - s0.w * m));
    }
  }
  const int npx_x = (TH + K - 1) * XW, npx_d = TH * kWlTW;
  const int stagers = (256 / CG) * CG;     // threads that stage (a whole number of pixels per pass)
  const int ppp = 256 / CG;                // pixels per staging pass
  const int spx = tid / CG;                // this thread's pixel within a pass

  const int per_img = dil * dil * w.ty_n * w.tx_n;
  const int nitems = q.B * per_img;
  for (int it = worker; it < nitems; it += w.workers) {
    const int b = it / per_img;
    int r = it - b * per_img;
    const int cls = r / (w.ty_n * w.tx_n);
    r -= cls * (w.ty_n * w.tx_n);
    const int tyi = r / w.tx_n, txi = r - tyi * w.tx_n;
    const int ry = cls / dil, rx = cls - ry * dil;
    const int Y0 = tyi * TH, X0 = txi * kWlTW;   // tile origin, class coordinates
    __syncthreads();                              // (the previous item's reads of xs / ds are done)
    if (tid < stagers) {
      // all loads of the item first (clamped addresses, no branches), then the arithmetic and the LDS stores: one
      // global-memory round trip per item instead of one per pass
      const act_t* xb = q.x + (size_t)b * q.H * q.W * C + (c0 + sc_) * 4;
      constexpr int NX = 8, ND = 4;  // passes: ceil((TH + K - 1) * XW / ppp) <= 8, TH * 16 / ppp <= 4 for every CG
      float4 xv[NX], gv[ND], zv[BN ? ND : 1];
      bool xo[NX], go[ND];
      int goff[ND];  // (element offsets inside image b: < 2^31)
      const size_t ib = (size_t)b * q.Ho * q.Wo * C + (c0 + sc_) * 4;
      auto load_x = [&]() {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
          const int p = spx + i * ppp;
          const int pc = p < npx_x ? p : 0;
          const int pj = pc / XW, pi = pc - pj * XW;
          const int iy = (Y0 + pj - w.pd) * dil + ry, ix = (X0 + pi - w.pd) * dil + rx;
          xo[i] = p < npx_x && iy >= 0 && iy < q.H && ix >= 0 && ix < q.W;
          xv[i] = lda4(xb + ((size_t)(xo[i] ? iy : 0) * q.W + (xo[i] ? ix : 0)) * C);
        }
      };
      auto load_d = [&]() {
#pragma unroll
        for (int i = 0; i < ND; ++i) {
          const int p = spx + i * ppp;
          const int pc = p < npx_d ? p : 0;
          const int pj = pc / kWlTW, pi = pc - pj * kWlTW;
          const int oy = (Y0 + pj) * dil + ry, ox = (X0 + pi) * dil + rx;
          go[i] = p < npx_d && oy < q.Ho && ox < q.Wo;
          goff[i] = ((go[i] ? oy : 0) * q.Wo + (go[i] ? ox : 0)) * C;
          gv[i] = lda4(q.dy + ib + goff[i]);
          if (BN) zv[i] = lda4(q.z + ib + goff[i]);
        }
      };
      auto store_x = [&]() {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
          const int p = spx + i * ppp;
          float4 v = xv[i];
          if (PRO) v = apply_prologue(v, pro);
          if (p < npx_x) xs[p * PS + sc_] = keep_if(v, xo[i]);
        }
      };
      auto store_d = [&]() {
#pragma unroll
        for (int i = 0; i < ND; ++i) {
          const int p = spx + i * ppp;
          float4 g = gv[i];
          if (BN) {
            if (q.bn_act) {
              const float4 y = fma4(zv[i], ca, cs);
              g = make_float4(g.x * act_mask(y.x, q.bn_act), g.y * act_mask(y.y, q.bn_act),
                              g.z * act_mask(y.z, q.bn_act), g.w * act_mask(y.w, q.bn_act));
            }
            g = fma4(g, ca, fma4(zv[i], cb, cd));
#ifdef NASSEG_BF16
            g = make_float4(bf16_to_f32(f32_to_bf16(g.x)), bf16_to_f32(f32_to_bf16(g.y)),
                            bf16_to_f32(f32_to_bf16(g.z)), bf16_to_f32(f32_to_bf16(g.w)));
#endif
            if (go[i]) sta4(q.dz + ib + goff[i], g);
          }
          if (p < npx_d) ds[p * PS + sc_] = keep_if(g, go[i]);
        }
      };
      if (BN) {  // (25 accumulators + both batches do not fit two waves per SIMD: two round trips)
        load_x();
        store_x();
        __builtin_amdgcn_sched_barrier(0);
        load_d();
        store_d();
      } else {
        load_x();
        load_d();
        store_x();
        store_d();
      }
    }
    __syncthreads();
    if (worker_thread) {
      float4 d[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) d[i] = ds[(tr * kWlTW + seg * 4 + i) * PS + cc];
#pragma unroll
      for (int ty = 0; ty < K; ++ty) {
        float4 v[K + 3];
#pragma unroll
        for (int u = 0; u < K + 3; ++u) v[u] = xs[((tr + ty) * XW + seg * 4 + u) * PS + cc];
#pragma unroll
        for (int tx = 0; tx < K; ++tx)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[ty * K + tx] = fma4(d[i], v[tx + i], acc[ty * K + tx]);
      }
    }
  }
  // the slots of a channel group meet in LDS, K taps at a time, and are added in slot order
  float* pout = q.partial + (size_t)worker * (K * K) * C;
  const int nslot = TH * 4;
#pragma unroll
  for (int r = 0; r < K; ++r) {
    __syncthreads();
    if (worker_thread) {
#pragma unroll
      for (int u = 0; u < K; ++u) wl_lds[(u * nslot + slot) * CG + cc] = acc[r * K + u];
    }
    __syncthreads();
    if (tid < K * CG) {
      const int u = tid / CG, c = tid - u * CG;
      float4 s4 = f4zero();
      for (int sl = 0; sl < nslot; ++sl) s4 = add4(s4, wl_lds[(u * nslot + sl) * CG + c]);
      sta4(pout + (size_t)(r * K + u) * C + (c0 + c) * 4, s4);
    }
  }
}

// generic backward-weight (any K): same block reduction, one tap at a time.
__global__ __launch_bounds__(256) void dw_wgrad_generic(
    const act_t* __restrict__ x, const act_t* __restrict__ dy, float* __restrict__ partial, int B,
    int H, int W, int C4, int Ho, int Wo, int K, int stride, int pad, int dil, int relu_in) {
  __shared__ float4 red[256];
  const int tid = threadIdx.x;
  const int base = blockIdx.x * 256;
  const int idx = base + tid;
  const bool live = idx < Wo * C4;
  const int ox = live ? idx / C4 : 0;
  const int c4 = live ? idx - ox * C4 : 0;
  const int C = C4 * 4;
  float* pout = partial + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (size_t)(K * K) * C;
  const int nown = C4 < 256 ? C4 : 256;
  for (int ty = 0; ty < K; ++ty)
    for (int tx = 0; tx < K; ++tx) {
      float4 a = f4zero();
      const int ix = ox * stride - pad + tx * dil;
      if (live && ix >= 0 && ix < W) {
        for (int wk = blockIdx.y; wk < B * Ho; wk += gridDim.y) {
          const int oy = wk % Ho, b = wk / Ho;
          const int iy = oy * stride - pad + ty * dil;
          if (iy < 0 || iy >= H) continue;
          float4 v = lda4(x + (((size_t)b * H + iy) * W + ix) * C + c4 * 4);
          if (relu_in) v = relu4(v);
          a = fma4(lda4(dy + (((size_t)b * Ho + oy) * Wo + ox) * C + c4 * 4), v, a);
        }
      }
      __syncthreads();
      red[tid] = a;
      __syncthreads();
      if (tid < nown) {
        float4 s = f4zero();
        for (int u = tid; u < 256; u += C4) s = add4(s, red[u]);
        const int cc = (base + tid) % C4;
        sta4(pout + (size_t)(ty * K + tx) * C + cc * 4, s);
      }
    }
}

// sum partial[blk][tap][C] over blk -> dw[(c*KK + tap)]  ((C,1,K,K) layout)
__global__ __launch_bounds__(256) void dw_wgrad_finalize(const float* __restrict__ partial,
                                                         float* __restrict__ dw, int nblk, int KK,
                                                         int C) {
  __shared__ double red[NASSEG_RP_SLICES][NASSEG_RP_ELEMS + 1];
  const int64_t per = (int64_t)KK * C;
  const int64_t i = (int64_t)blockIdx.x * NASSEG_RP_ELEMS + rp_elem();  // tap*C + c
  const bool valid = i < per;
  const double s = reduce_partials16(partial, nblk, per, i, valid, red);
  if (valid && rp_slice() == 0) {
    const int t = (int)(i / C), c = (int)(i - (int64_t)t * C);
    dw[c * KK + t] = (float)s;
  }
}

// ---------------------------------------------------------------------------
// The whole backward of a 3x3 depthwise conv that sits BETWEEN two BatchNorms of a chain
// (InvertedResidual: 1x1 -> BN -> ReLU6 -> [3x3 depthwise] -> BN -> ReLU6 -> 1x1, src/nn/
// layer_factory.py:139-152) in ONE pass over the conv's input grid:
//   dz  = second half of the BatchNorm backward behind the conv, computed on load from g and z
//         (as dw_wgrad_tile<.., BN> does) - never stored;
//   dW[ty][tx] += xa[iy][ix] * dz[oy][ox]            xa = act(in_scale*xz + in_shift), the
//   ge[iy][ix]  = act'(..) * sum_t w[ty][tx]*dz[oy][ox]   normalised input (never stored either)
//   with (oy, ox) = ((iy + pad - ty)/stride, (ix + pad - tx)/stride) where that divides,
//   plus the partial sums {sum ge, sum ge*xhat} of the BatchNorm in FRONT of the conv.
// Both products of an input pixel use the same <= 9 dz values, so a thread (one input column -
// or, stride 2, one 2x2 input quad column - and one channel group, walking down the rows) keeps a
// window of dz rows in registers and touches each of xz, g, z exactly once per pass and ge once:
// four tensor passes instead of the seven of dw_wgrad_strip<BN> + dw_fwd_strip<STATS = 2> (xz
// read twice, dz written and read back).  Weight-gradient and statistics partials per workgroup
// as in dw_wgrad_tile; deterministic.
// ---------------------------------------------------------------------------
struct DwBwdArgs {
  const act_t* xz;  // raw output of the conv in front: [B][H][W][C]
  const act_t* g;   // gradient w.r.t. this conv's BatchNorm output [B][Ho][Wo][C]
  const act_t* z;   // this conv's raw output
  const float* wt;  // [tap][C], un-flipped
  act_t* ge;        // [B][H][W][C]
  float* partial;   // [blocks][9][C]
  float* stats;     // [blocks][2][C]
  const float* in_scale; const float* in_shift; const float* in_mean; const float* in_invstd; int in_act;
  const float* bn_scale; const float* bn_shift; const float* bn_mean; const float* bn_invstd;
  const float* bn_sums; int bn_train, bn_act; float invM;
  int B, H, W, C4, Ho, Wo, rows_per_item;
  int flip;  // wt is the 180-degree rotated packing (what a stride-1 chain keeps for backward-data)
};

struct DzConst {
  float4 ca, cb, cd, cs;
};
__device__ __forceinline__ DzConst dz_const(const DwBwdArgs& q, int c4) {
  DzConst k;
  const int C = q.C4 * 4;
  k.ca = lda4(q.bn_scale + c4 * 4);
  k.cs = q.bn_act ? lda4(q.bn_shift + c4 * 4) : f4zero();
  k.cb = f4zero();
  k.cd = f4zero();
  if (q.bn_train) {
    const float4 is = lda4(q.bn_invstd + c4 * 4), mu = lda4(q.bn_mean + c4 * 4);
    const float4 s0 = lda4(q.bn_sums + c4 * 4), s1 = lda4(q.bn_sums + C + c4 * 4);
    const float m = q.invM;
    k.cb = make_float4(-k.ca.x * is.x * (s1.x * m), -k.ca.y * is.y * (s1.y * m), -k.ca.z * is.z * (s1.z * m),
                       -k.ca.w * is.w * (s1.w * m));
    k.cd = make_float4(k.ca.x * (mu.x * is.x * (s1.x * m) - s0.x * m), k.ca.y * (mu.y * is.y * (s1.y * m) - s0.y * m),
                       k.ca.z * (mu.z * is.z * (s1.z * m) - s0.z * m), k.ca.w * (mu.w * is.w * (s1.w * m) - s0.w * m));
  }
  return k;
}
// dz at output pixel (oy, ox) of image b (zero outside the map) in two halves, so that the loads of
// the next row can be issued before the arithmetic of the current one: dz_load (unconditional loads
// from a clamped address) and dz_make
struct DzRaw {
  float4 g, z;
  bool ok;
};
__device__ __forceinline__ DzRaw dz_load(const DwBwdArgs& q, int b, int oy, int ox, int c4) {
  DzRaw r;
  r.ok = oy >= 0 && oy < q.Ho && ox >= 0 && ox < q.Wo;
  const int oyc = oy < 0 ? 0 : (oy >= q.Ho ? q.Ho - 1 : oy), oxc = ox < 0 ? 0 : (ox >= q.Wo ? q.Wo - 1 : ox);
  const size_t off = (((size_t)b * q.Ho + oyc) * q.Wo + oxc) * (q.C4 * 4) + c4 * 4;
  r.g = lda4(q.g + off);
  r.z = lda4(q.z + off);
  return r;
}
__device__ __forceinline__ float4 dz_make(const DwBwdArgs& q, const DzConst& k, const DzRaw& r) {
  float4 gv = r.g;
  if (q.bn_act) {
    const float4 y = fma4(r.z, k.ca, k.cs);
    gv = make_float4(gv.x * act_mask(y.x, q.bn_act), gv.y * act_mask(y.y, q.bn_act), gv.z * act_mask(y.z, q.bn_act),
                     gv.w * act_mask(y.w, q.bn_act));
  }
  float4 v = fma4(gv, k.ca, fma4(r.z, k.cb, k.cd));
#ifdef NASSEG_BF16
  v = make_float4(bf16_to_f32(f32_to_bf16(v.x)), bf16_to_f32(f32_to_bf16(v.y)), bf16_to_f32(f32_to_bf16(v.z)),
                  bf16_to_f32(f32_to_bf16(v.w)));  // (what the two-kernel form stores and reads back)
#endif
  return keep_if(v, r.ok);
}
__device__ __forceinline__ float4 dz_at(const DwBwdArgs& q, const DzConst& k, int b, int oy, int ox, int c4) {
  return dz_make(q, k, dz_load(q, b, oy, ox, c4));
}

// STRIDE 1 or 2; 3x3, pad 1, dilation 1
#ifndef NASSEG_DWBWD_MINB
#define NASSEG_DWBWD_MINB 1
#endif
template <int STRIDE>
__global__ __launch_bounds__(256, NASSEG_DWBWD_MINB) void dw3x3_bwd_bn_kernel(DwBwdArgs q) {
  __shared__ float4 lw[9][64];
  __shared__ float4 red[3][4][64];
  const int C4 = q.C4, C = C4 * 4, H = q.H, W = q.W;
  const int tid = threadIdx.x;
  for (int i = tid; i < 9 * C4; i += 256) {
    const int t = i / C4;
    lw[t][i % C4] = lda4(q.wt + (size_t)(q.flip ? 8 - t : t) * C + (i % C4) * 4);
  }
  __syncthreads();
  const int Wq = STRIDE == 1 ? W : (W + 1) >> 1;   // columns of threads
  const int Hq = STRIDE == 1 ? H : (H + 1) >> 1;   // rows a thread walks (quad rows for stride 2)
  const int base = blockIdx.x * 256;
  const int idx = base + tid;
  const bool live = idx < Wq * C4;
  const int xq = live ? idx / C4 : 0;
  const int c4 = live ? idx - xq * C4 : 0;
  const DzConst kc = dz_const(q, c4);
  BnBwd bn = {q.xz, q.in_scale, q.in_shift, q.in_mean, q.in_invstd, q.in_act};
  const BnBwdLane bl = bnbwd_lane(bn, c4);
  Prologue pro;  // (scale / shift shared with bl)
  pro.sc = bl.sc;
  pro.sh = bl.sh;
  pro.lo = q.in_act ? 0.f : -INFINITY;
  pro.hi = q.in_act == NASSEG_ACT_RELU6 ? 6.f : INFINITY;
  float4 dwa[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) dwa[t] = f4zero();
  float4 ssum[2] = {f4zero(), f4zero()};
  const int nchunk = cdiv_dev(Hq, q.rows_per_item);
  const int nwork = q.B * nchunk;
  for (int wk = blockIdx.y; wk < nwork; wk += gridDim.y) {
    const int b = wk / nchunk;
    const int r0 = (wk - b * nchunk) * q.rows_per_item;
    int r1 = r0 + q.rows_per_item;
    if (r1 > Hq) r1 = Hq;
    if (!live) continue;
    if (STRIDE == 1) {
      // window d[ry][cx] = dz[iy - 1 + ry][ix - 1 + cx]; tap (ty, tx) pairs input (iy, ix) with
      // output (iy + 1 - ty, ix + 1 - tx) = d[2 - ty][2 - tx]
      float4 d[3][3];
#pragma unroll
      for (int cx = 0; cx < 3; ++cx) {
        d[1][cx] = dz_at(q, kc, b, r0 - 1, xq - 1 + cx, c4);
        d[2][cx] = dz_at(q, kc, b, r0, xq - 1 + cx, c4);
      }
      // the loads of row iy + 1 (three dz columns, one input pixel) are in flight while row iy is computed
      DzRaw nxt[3];
#pragma unroll
      for (int cx = 0; cx < 3; ++cx) nxt[cx] = dz_load(q, b, r0 + 1, xq - 1 + cx, c4);
      float4 zin_nxt = lda4(q.xz + (((size_t)b * H + r0) * W + xq) * C + c4 * 4);
      for (int iy = r0; iy < r1; ++iy) {
        DzRaw cur[3];
#pragma unroll
        for (int cx = 0; cx < 3; ++cx) cur[cx] = nxt[cx];
        const float4 zin = zin_nxt;
        if (iy + 1 < r1) {
          const int iyn = iy + 1 < H ? iy + 1 : H - 1;
#pragma unroll
          for (int cx = 0; cx < 3; ++cx) nxt[cx] = dz_load(q, b, iy + 2, xq - 1 + cx, c4);
          zin_nxt = lda4(q.xz + (((size_t)b * H + iyn) * W + xq) * C + c4 * 4);
        }
#pragma unroll
        for (int cx = 0; cx < 3; ++cx) {
          d[0][cx] = d[1][cx];
          d[1][cx] = d[2][cx];
          d[2][cx] = dz_make(q, kc, cur[cx]);
        }
        const size_t off = (((size_t)b * H + iy) * W + xq) * C + c4 * 4;
        const float4 xa = apply_prologue(zin, pro);
        float4 o = f4zero();
#pragma unroll
        for (int ty = 0; ty < 3; ++ty)
#pragma unroll
          for (int tx = 0; tx < 3; ++tx) {
            const float4 dv = d[2 - ty][2 - tx];
            o = fma4(lw[ty * 3 + tx][c4], dv, o);
            dwa[ty * 3 + tx] = fma4(xa, dv, dwa[ty * 3 + tx]);
          }
        bnbwd_accumulate(o, zin, bl, q.in_act, true, ssum);
        sta4(q.ge + off, o);
      }
    } else {
      // a 2x2 input quad (2a + py, 2b + px) receives from dz[a + ry][b + rx], ry, rx in {0, 1}:
      // tap ty feeds input parity py = (ty + 1) & 1 from row ry = (py + 1 - ty) / 2
      float4 d[2][2];
#pragma unroll
      for (int rx = 0; rx < 2; ++rx) d[1][rx] = dz_at(q, kc, b, r0, xq + rx, c4);
      // the loads of quad row aq + 1 (two dz columns, four input pixels) are in flight while quad row aq
      // is computed
      DzRaw nxt[2];
      float4 zin_nxt[2][2];
#pragma unroll
      for (int rx = 0; rx < 2; ++rx) nxt[rx] = dz_load(q, b, r0 + 1, xq + rx, c4);
#pragma unroll
      for (int py = 0; py < 2; ++py)
#pragma unroll
        for (int px = 0; px < 2; ++px) {
          const int iy = 2 * r0 + py, ix = 2 * xq + px;
          zin_nxt[py][px] = lda4(q.xz + (((size_t)b * H + (iy < H ? iy : H - 1)) * W + (ix < W ? ix : W - 1)) * C + c4 * 4);
        }
      for (int aq = r0; aq < r1; ++aq) {
        DzRaw cur[2];
        float4 o[2][2], xa[2][2], zin[2][2];
        bool ok[2][2];
#pragma unroll
        for (int rx = 0; rx < 2; ++rx) cur[rx] = nxt[rx];
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
          for (int px = 0; px < 2; ++px) zin[py][px] = zin_nxt[py][px];
        if (aq + 1 < r1) {
#pragma unroll
          for (int rx = 0; rx < 2; ++rx) nxt[rx] = dz_load(q, b, aq + 2, xq + rx, c4);
#pragma unroll
          for (int py = 0; py < 2; ++py)
#pragma unroll
            for (int px = 0; px < 2; ++px) {
              const int iy = 2 * (aq + 1) + py, ix = 2 * xq + px;
              zin_nxt[py][px] =
                  lda4(q.xz + (((size_t)b * H + (iy < H ? iy : H - 1)) * W + (ix < W ? ix : W - 1)) * C + c4 * 4);
            }
        }
#pragma unroll
        for (int rx = 0; rx < 2; ++rx) {
          d[0][rx] = d[1][rx];
          d[1][rx] = dz_make(q, kc, cur[rx]);
        }
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
          for (int px = 0; px < 2; ++px) {
            const int iy = 2 * aq + py, ix = 2 * xq + px;
            ok[py][px] = iy < H && ix < W;
            xa[py][px] = keep_if(apply_prologue(zin[py][px], pro), ok[py][px]);
            o[py][px] = f4zero();
          }
#pragma unroll
        for (int ty = 0; ty < 3; ++ty)
#pragma unroll
          for (int tx = 0; tx < 3; ++tx) {
            const int py = (ty + 1) & 1, px = (tx + 1) & 1;
            const int ry = (py + 1 - ty) / 2, rx = (px + 1 - tx) / 2;
            const float4 dv = d[ry][rx];
            o[py][px] = fma4(lw[ty * 3 + tx][c4], dv, o[py][px]);
            dwa[ty * 3 + tx] = fma4(xa[py][px], dv, dwa[ty * 3 + tx]);
          }
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
          for (int px = 0; px < 2; ++px) {
            bnbwd_accumulate(o[py][px], zin[py][px], bl, q.in_act, ok[py][px], ssum);
            if (ok[py][px])
              sta4(q.ge + (((size_t)b * H + 2 * aq + py) * W + 2 * xq + px) * C + c4 * 4, o[py][px]);
          }
      }
    }
  }
  const size_t blk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
  if (!live) {
#pragma unroll
    for (int t = 0; t < 9; ++t) dwa[t] = f4zero();
  }
  block_reduce_groups<9, 3>(dwa, red, q.partial + blk * 9 * C, base, C4);
  block_reduce_groups<2, 2>(ssum, reinterpret_cast<float4(*)[4][64]>(&red[0][0][0]), q.stats + blk * 2 * C, base, C4);
}

}  // namespace

extern "C" {

#if NASSEG_FP32_ONLY
// wt[tap][C] <- w (C,1,K,K); flip != 0 rotates the kernel by 180 degrees.
int nasseg_dw_pack_weight(const float* w, float* wt, int C, int K, int flip, void* stream) {
  NASSEG_REQUIRE(C > 0 && K > 0, "dw_pack_weight: bad shape C=%d K=%d", C, K);
  int n = C * K * K;
  hipLaunchKernelGGL(dw_pack_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, w, wt,
                     C, K * K, flip);
  NASSEG_LAUNCH_CHECK("dw_pack_weight");
  return NASSEG_OK;
}
#endif  // NASSEG_FP32_ONLY

int64_t nasseg_dwconv_stats_blocks(int B, int C, int Ho, int Wo, int K, int stride, int dil);
int64_t nasseg_dwconv_bwd_data_bn_blocks(int B, int C, int Ho, int Wo, int K, int stride, int pad,
                                         int dil, int transposed);

// grid of the stride-2 backward-data kernel; with the statistics epilogue it is capped (a
// grid-stride loop does the rest) and made a multiple of C4 / gcd(C4, 256)
static int s2_blocks(int B, int Hd, int Wd, int C4, bool bst) {
  const int64_t quads = (int64_t)B * ((Hd + 1) / 2) * ((Wd + 1) / 2) * C4;
  int64_t nb = (quads + 255) / 256;
  if (!bst) return (int)(nb < 65536 * 4 ? nb : 65536 * 4);
  if (nb > 4096) nb = 4096;
  int gcd = C4, t = 256;
  while (t) { const int r = gcd % t; gcd = t; t = r; }
  const int m = C4 / gcd;
  return (int)((nb + m - 1) / m * m);
}

static int dwconv_impl(const act_t* x, const float* wt, act_t* y, const float* in_scale,
                       const float* in_shift, int in_act, const float* scale, const float* shift,
                       int act, int B, int H, int W, int C, int Ho, int Wo, int K, int stride,
                       int pad, int dil, int transposed, float* stats, int stats_mode, BnBwd bn,
                       hipStream_t s) {
  NASSEG_REQUIRE(C % 4 == 0, "dwconv: C=%d must be a multiple of 4", C);
  NASSEG_REQUIRE(B > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && K > 0 && stride > 0 && dil > 0,
                 "dwconv: bad geometry");
  const int C4 = C / 4;
  StripCfg sc = strip_cfg(stride, dil);
  const bool pro = in_scale || in_shift || in_act;
  const bool strip_ok = !transposed && (K == 3 || K == 5) && (sc.e == 1 || sc.e == 2) && B <= 65535;
  if (strip_ok) {
    constexpr int P = 4, P3 = NASSEG_DW_P3;  // output rows per thread (see dw_fwd_rows)
    const int px = dw_fwd_px(K, stride, dil, Wo);
    const int nchunk = cdiv(Ho, dw_fwd_rows(K, sc.e, Ho, px) * sc.g);
    dim3 grid(cdiv(dw_fwd_xgroups(K, stride, dil, Wo) * C4, 256), nchunk * sc.g, B);
    NASSEG_REQUIRE(grid.y <= 65535, "dwconv: too many row chunks");
    NASSEG_REQUIRE(stats_mode != 2 || !pro, "dwconv_bwd_data_bn: no input prologue on this path");
    if (stats && C4 > 256) {
      hipError_t e = hipMemsetAsync(stats, 0, (size_t)grid.x * grid.y * grid.z * 2 * C * sizeof(float), s);
      if (e != hipSuccess) return nasseg_fail(NASSEG_ERR_LAUNCH, "dwconv: memset failed");
    }
#define LAUNCH_FWD3(KK, EE, WL, PR, ST)                                                             \
  hipLaunchKernelGGL((dw_fwd_strip<KK, P, EE, WL, PR, ST>), grid, dim3(256), 0, s, x, wt, y, in_scale, \
                     in_shift, in_act, scale, shift, H, W, C4, Ho, Wo, stride, pad, dil, sc.g, nchunk, \
                     act, stats, bn)
#define LAUNCH_FWD(KK, EE, WL)                                    \
  do {                                                            \
    if (stats_mode == 2) LAUNCH_FWD3(KK, EE, WL, false, 2);       \
    else if (pro && stats) LAUNCH_FWD3(KK, EE, WL, true, 1);      \
    else if (pro) LAUNCH_FWD3(KK, EE, WL, true, 0);               \
    else if (stats) LAUNCH_FWD3(KK, EE, WL, false, 1);            \
    else LAUNCH_FWD3(KK, EE, WL, false, 0);                       \
  } while (0)
    const bool wl = C4 <= 64;
    if (K == 5 && sc.e == 1 && px == 1) { if (wl) LAUNCH_FWD(5, 1, true); else LAUNCH_FWD(5, 1, false); }
    else if (K == 5 && px == 1) { if (wl) LAUNCH_FWD(5, 2, true); else LAUNCH_FWD(5, 2, false); }
#undef LAUNCH_FWD3
#define LAUNCH_FWD3(KK, EE, WL, PR, ST)                                                                   \
  hipLaunchKernelGGL((dw_fwd_strip<KK, NASSEG_DW5_P, EE, WL, PR, ST, NASSEG_DW5_PX>), grid, dim3(256), 0, s, x, wt, y, in_scale, \
                     in_shift, in_act, scale, shift, H, W, C4, Ho, Wo, stride, pad, dil, sc.g, nchunk,    \
                     act, stats, bn)
    if (K == 5 && px > 1) { if (wl) LAUNCH_FWD(5, 1, true); else LAUNCH_FWD(5, 1, false); }
#undef LAUNCH_FWD3
#define LAUNCH_FWD3(KK, EE, WL, PR, ST)                                                              \
  hipLaunchKernelGGL((dw_fwd_strip<KK, P3, EE, WL, PR, ST>), grid, dim3(256), 0, s, x, wt, y, in_scale, \
                     in_shift, in_act, scale, shift, H, W, C4, Ho, Wo, stride, pad, dil, sc.g, nchunk,  \
                     act, stats, bn)
    if (K == 3 && sc.e == 1 && dw_fwd_rows(K, sc.e, Ho) == P3) LAUNCH_FWD(3, 1, false);
#undef LAUNCH_FWD3
#define LAUNCH_FWD3(KK, EE, WL, PR, ST)                                                             \
  hipLaunchKernelGGL((dw_fwd_strip<KK, P, EE, WL, PR, ST>), grid, dim3(256), 0, s, x, wt, y, in_scale, \
                     in_shift, in_act, scale, shift, H, W, C4, Ho, Wo, stride, pad, dil, sc.g, nchunk, \
                     act, stats, bn)
    if (K == 3 && sc.e == 1 && dw_fwd_rows(K, sc.e, Ho) == P) LAUNCH_FWD(3, 1, false);
    else if (K == 3 && sc.e == 2) LAUNCH_FWD(3, 2, false);
#undef LAUNCH_FWD3
#undef LAUNCH_FWD
    NASSEG_LAUNCH_CHECK("dw_fwd_strip");
    return NASSEG_OK;
  }
  NASSEG_REQUIRE(!in_scale && !in_shift && in_act <= NASSEG_ACT_RELU,
                 "dwconv: only an input ReLU is supported as prologue on the generic path");
  if (transposed && stride == 2 && dil == 1 && (K == 3 || K == 5) && pad == (K - 1) / 2 &&
      !in_act && !scale && !shift && act == 0) {
    // (H, W) = dims of dy, (Ho, Wo) = dims of dx in the transposed call
    const bool bst = stats_mode == 2;
    const int nbq = s2_blocks(B, Ho, Wo, C4, bst);
    if (bst && C4 > 256) {
      hipError_t e = hipMemsetAsync(stats, 0, (size_t)nbq * 2 * C * sizeof(float), s);
      if (e != hipSuccess) return nasseg_fail(NASSEG_ERR_LAUNCH, "dwconv: memset failed");
    }
#define LAUNCH_S2(KK, BS) \
  hipLaunchKernelGGL((dw_bwd_data_s2<KK, BS>), dim3(nbq), dim3(256), 0, s, x, wt, y, B, H, W, C4, Ho, Wo, stats, bn)
    if (K == 3) { if (bst) LAUNCH_S2(3, true); else LAUNCH_S2(3, false); }
    else { if (bst) LAUNCH_S2(5, true); else LAUNCH_S2(5, false); }
#undef LAUNCH_S2
    NASSEG_LAUNCH_CHECK("dw_bwd_data_s2");
    return NASSEG_OK;
  }
  NASSEG_REQUIRE(!stats, "dwconv: the statistics epilogues need a 3x3 / 5x5 strip or stride-2 geometry");
  size_t total = (size_t)B * Ho * Wo * C4;
  int nb = (int)((total + 255) / 256 < 65536 * 4 ? (total + 255) / 256 : 65536 * 4);
  hipLaunchKernelGGL(dw_generic, dim3(nb), dim3(256), 0, s, x, wt, y, scale, shift, B, H, W, C4,
                     Ho, Wo, K, stride, pad, dil, transposed, in_act, act);
  NASSEG_LAUNCH_CHECK("dw_generic");
  return NASSEG_OK;
}

// y = act(scale * dwconv(in_act(in_scale*x + in_shift)) + shift); every pointer of the
// prologue / epilogue may be null (identity).  transposed != 0 computes the backward-data
// form (x = grad wrt output with dims (H,W), y = grad wrt input with dims (Ho,Wo),
// un-flipped weights).  stats != null: also writes stats[blk][2][C] (sum, sum of squares of
// y per channel) for blk < nasseg_dwconv_stats_blocks(...); forward strip geometries only.
int NASSEG_FN(dwconv)(const act_t* x, const float* wt, act_t* y, const float* in_scale,
                  const float* in_shift, int in_act, const float* scale, const float* shift, int act,
                  int B, int H, int W, int C, int Ho, int Wo, int K, int stride, int pad, int dil,
                  int transposed, float* stats, void* stream) {
  NASSEG_REQUIRE(!stats || transposed == 0, "dwconv: statistics are a forward epilogue");
  BnBwd bn = {};
  return dwconv_impl(x, wt, y, in_scale, in_shift, in_act, scale, shift, act, B, H, W, C, Ho, Wo, K,
                     stride, pad, dil, transposed, stats, stats ? 1 : 0, bn, (hipStream_t)stream);
}

// Backward-data of a depthwise conv whose forward input was the normalised activation
// a = act(scale*z + shift) of a BatchNorm, fused with the first half of that BatchNorm's
// backward (the depthwise twin of nasseg_conv_bwd_data_bn): writes
//   g = act'(scale*z + shift) * dwconv_backward_data(dy)
// and stats[blk][0][c] = sum g, stats[blk][1][c] = sum g*(z-mean)*invstd over the pixels of
// workgroup blk < nasseg_dwconv_bwd_data_bn_blocks(...).  Geometry arguments exactly as the
// nasseg_dwconv call that computes the same backward-data: either the correlation form
// (transposed == 0, stride 1, flipped weights, pad' = dil*(K-1) - pad) or the transposed
// stride-2 form; g and z have dims (Ho, Wo).
int NASSEG_FN(dwconv_bwd_data_bn)(const act_t* dy, const float* wt, act_t* g, const act_t* z,
                              const float* scale, const float* shift, const float* mean,
                              const float* invstd, int act, int B, int H, int W, int C, int Ho,
                              int Wo, int K, int stride, int pad, int dil, int transposed,
                              float* stats, void* stream) {
  NASSEG_REQUIRE(z && scale && shift && mean && invstd && stats, "dwconv_bwd_data_bn: null argument");
  NASSEG_REQUIRE(nasseg_dwconv_bwd_data_bn_blocks(B, C, Ho, Wo, K, stride, pad, dil, transposed) > 0,
                 "dwconv_bwd_data_bn: geometry has no fused path");
  BnBwd bn = {z, scale, shift, mean, invstd, act};
  return dwconv_impl(dy, wt, g, nullptr, nullptr, 0, nullptr, nullptr, 0, B, H, W, C, Ho, Wo, K,
                     stride, pad, dil, transposed, stats, 2, bn, (hipStream_t)stream);
}

#if NASSEG_FP32_ONLY
// 1 when nasseg_dwconv / nasseg_dwconv_wgrad take the fast strip path for this geometry,
// i.e. when the full input prologue and the statistics epilogue are available
int nasseg_dwconv_strip_ok(int K, int stride, int dil) {
  StripCfg sc = strip_cfg(stride, dil);
  return ((K == 3 || K == 5) && (sc.e == 1 || sc.e == 2)) ? 1 : 0;
}
#endif  // NASSEG_FP32_ONLY

#if NASSEG_FP32_ONLY
// number of statistic rows nasseg_dwconv writes (0 when the geometry has no stats support)
int64_t nasseg_dwconv_stats_blocks(int B, int C, int Ho, int Wo, int K, int stride, int dil) {
  if (!nasseg_dwconv_strip_ok(K, stride, dil)) return 0;
  StripCfg sc = strip_cfg(stride, dil);
  return (int64_t)cdiv(dw_fwd_xgroups(K, stride, dil, Wo) * (C / 4), 256) * cdiv(Ho, dw_fwd_rows(K, sc.e, Ho, dw_fwd_px(K, stride, dil, Wo)) * sc.g) *
         sc.g * B;
}
#endif  // NASSEG_FP32_ONLY

#if NASSEG_FP32_ONLY
// rows of statistics nasseg_dwconv_bwd_data_bn writes for this call (0: no fused path, use
// nasseg_dwconv + nasseg_bn_bwd_reduce)
int64_t nasseg_dwconv_bwd_data_bn_blocks(int B, int C, int Ho, int Wo, int K, int stride, int pad,
                                         int dil, int transposed) {
  if (C % 4 != 0 || B > 65535) return 0;
  if (!transposed) return nasseg_dwconv_stats_blocks(B, C, Ho, Wo, K, stride, dil);
  if (stride == 2 && dil == 1 && (K == 3 || K == 5) && pad == (K - 1) / 2)
    return s2_blocks(B, Ho, Wo, C / 4, true);
  return 0;
}
#endif  // NASSEG_FP32_ONLY

// number of workgroup rows (grid.y) of the backward-weight kernels: ~1024 workgroups,
// but at least ~8 output rows of work per workgroup so that the end-of-block reduction
// is amortised
static int64_t wgrad_rows(int B, int C, int Ho, int Wo) {
  const int64_t gx = cdiv(Wo * (C / 4), 256);
  int64_t gy = 1024 / gx;
  const int64_t rows = (int64_t)B * Ho;
  if (gy > rows / 8) gy = rows / 8;
  if (gy < 1) gy = 1;
  return gy;
}

// logical grid of the PX-column backward-weight kernel inside the gx * gy workgroups (= partial rows) of the
// one-column form: a quarter of the columns, as many row workers as fit
static void dw_wgrad_px_grid(int K, int stride, int dil, int Wo, int C4, int gx, int gy, int* lgx, int* lgy) {
  *lgx = cdiv(dw_fwd_xgroups(K, stride, dil, Wo, NASSEG_DW5_WPX) * C4, 256);
  *lgy = (gx * gy) / *lgx;
  if (*lgy < 1) *lgy = 1;  // (lgx <= gx: never reached)
}

#if NASSEG_FP32_ONLY
int64_t nasseg_dwconv_wgrad_workspace(int B, int C, int Ho, int Wo, int K) {
  const int64_t gx = cdiv(Wo * (C / 4), 256);
  return gx * wgrad_rows(B, C, Ho, Wo) * (int64_t)K * K * C;
}
#endif  // NASSEG_FP32_ONLY

// the LDS-tiled backward-weight kernel's plan for a layer (ok == 0: the strip kernel)
struct DwWlPlan {
  int ok, CG, TH, Hs, Ws, ty_n, tx_n;
  size_t lds;
};
static DwWlPlan dw_wgrad_lds_plan(int K, int stride, int pad, int dil, int C4, int Ho, int Wo) {
  DwWlPlan p = {};
  if (!g_dw_wgrad_lds.load() || K != 5 || stride != 1 || pad % dil != 0 || C4 > 256) return p;
  for (int cg = 8; cg >= 2; --cg)
    if (C4 % cg == 0) { p.CG = cg; break; }
  if (!p.CG) return p;
  p.TH = (256 / p.CG) / 4;
  p.Hs = cdiv(Ho, dil);
  p.Ws = cdiv(Wo, dil);
  if (p.Hs < 4 || p.Ws < 8) return p;  // (a class smaller than a quarter tile: the strip kernel)
  p.ty_n = cdiv(p.Hs, p.TH);
  p.tx_n = cdiv(p.Ws, kWlTW);
  p.lds = (size_t)((p.TH + K - 1) * (kWlTW + K - 1) + p.TH * kWlTW) * (p.CG + 1) * sizeof(float4);
  p.ok = 1;
  return p;
}
static int dw_wgrad_lds_launch5(const DwWgArgs& q, const DwWlPlan& p, int workers, bool pro, bool bn, hipStream_t s) {
  constexpr int K = 5;
  DwWlArgs w = {};
  w.a = q;
  w.CG = p.CG; w.TH = p.TH; w.nchunkC = q.C4 / p.CG; w.workers = workers; w.pd = q.pad / q.dil;
  w.Hs = p.Hs; w.Ws = p.Ws; w.ty_n = p.ty_n; w.tx_n = p.tx_n;
  const dim3 grid((unsigned)(workers * w.nchunkC));
#define GO_WL(PR, BB) hipLaunchKernelGGL((dw_wgrad_lds_kernel<K, PR, BB>), grid, dim3(256), p.lds, s, w)
  if (bn && pro) GO_WL(true, true);
  else if (bn) GO_WL(false, true);
  else if (pro) GO_WL(true, false);
  else GO_WL(false, false);
#undef GO_WL
  NASSEG_LAUNCH_CHECK("dw_wgrad_lds_kernel");
  return NASSEG_OK;
}

// dw (C,1,K,K) = sum over pixels of dy * in_act(in_scale*x_tap + in_shift); ws must hold
// nasseg_dwconv_wgrad_workspace() floats.  dw == null: only the partial rows [rows][K*K][C] are
// produced (rows = workspace floats / (K*K*C)) for nasseg_wgrad_finalize_many.
static int dw_wgrad_impl(const act_t* x, const act_t* dy, float* dw, float* ws, const float* in_scale,
                         const float* in_shift, int in_act, int B, int H, int W, int C, int Ho, int Wo,
                         int K, int stride, int pad, int dil, const DwWgArgs* bn, void* stream) {
  NASSEG_REQUIRE(C % 4 == 0, "dwconv_wgrad: C=%d must be a multiple of 4", C);
  hipStream_t s = (hipStream_t)stream;
  const int C4 = C / 4;
  StripCfg sc = strip_cfg(stride, dil);
  const int gx = cdiv(Wo * C4, 256);
  const int gy = (int)wgrad_rows(B, C, Ho, Wo);
  if (C4 > 256) {
    hipError_t e = hipMemsetAsync(ws, 0, (size_t)gx * gy * K * K * C * sizeof(float), s);
    if (e != hipSuccess) return nasseg_fail(NASSEG_ERR_LAUNCH, "dwconv_wgrad: memset failed");
  }
  const bool pro = in_scale || in_shift || in_act;
  const bool strip_ok = (K == 3 || K == 5) && (sc.e == 1 || sc.e == 2);
  NASSEG_REQUIRE(!bn || strip_ok, "dwconv_wgrad_bn: needs a strip geometry (nasseg_dwconv_strip_ok)");
  const DwWlPlan wl = dw_wgrad_lds_plan(K, stride, pad, dil, C4, Ho, Wo);
  if (wl.ok) {
    DwWgArgs q = {x, dy, ws, in_scale, in_shift, in_act, B, H, W, C4, Ho, Wo, stride, pad, dil, sc.g, 0};
    if (bn) {
      q.z = bn->z; q.dz = bn->dz; q.bn_scale = bn->bn_scale; q.bn_shift = bn->bn_shift; q.bn_mean = bn->bn_mean;
      q.bn_invstd = bn->bn_invstd; q.bn_sums = bn->bn_sums; q.invM = bn->invM; q.bn_train = bn->bn_train;
      q.bn_act = bn->bn_act;
    }
    int rc = dw_wgrad_lds_launch5(q, wl, gx * gy, pro, bn != nullptr, s);
    if (rc) return rc;
  } else if (strip_ok) {
    constexpr int P = 4;
    const int nchunk = cdiv(Ho, P * sc.g);
    dim3 grid(gx, gy, 1);
    DwWgArgs q = {x, dy, ws, in_scale, in_shift, in_act, B, H, W, C4, Ho, Wo, stride, pad, dil, sc.g, nchunk};
    const int px = dw_fwd_px(K, stride, dil, Wo, NASSEG_DW5_WPX);
    if (px > 1) dw_wgrad_px_grid(K, stride, dil, Wo, C4, gx, gy, &q.lgx, &q.lgy);
    if (bn) {
      q.z = bn->z; q.dz = bn->dz; q.bn_scale = bn->bn_scale; q.bn_shift = bn->bn_shift; q.bn_mean = bn->bn_mean;
      q.bn_invstd = bn->bn_invstd; q.bn_sums = bn->bn_sums; q.invM = bn->invM; q.bn_train = bn->bn_train;
      q.bn_act = bn->bn_act;
    }
#define LAUNCH_WG2(KK, EE, PR, BB) \
  hipLaunchKernelGGL((dw_wgrad_strip<KK, P, EE, PR, BB>), grid, dim3(256), 0, s, q)
#define LAUNCH_WG(KK, EE)                                   \
  do {                                                      \
    if (bn && pro) LAUNCH_WG2(KK, EE, true, true);          \
    else if (bn) LAUNCH_WG2(KK, EE, false, true);           \
    else if (pro) LAUNCH_WG2(KK, EE, true, false);          \
    else LAUNCH_WG2(KK, EE, false, false);                  \
  } while (0)
    if (K == 3 && sc.e == 1) LAUNCH_WG(3, 1);
    else if (K == 3 && sc.e == 2) LAUNCH_WG(3, 2);
    else if (K == 5 && sc.e == 1 && px == 1) LAUNCH_WG(5, 1);
    else if (K == 5 && px == 1) LAUNCH_WG(5, 2);
#undef LAUNCH_WG2
#define LAUNCH_WG2(KK, EE, PR, BB) \
  hipLaunchKernelGGL((dw_wgrad_strip_px<KK, P, EE, PR, BB, NASSEG_DW5_WPX>), dim3(gx * gy), dim3(256), 0, s, q)
    if (K == 5 && px > 1) LAUNCH_WG(5, 1);
#undef LAUNCH_WG2
#undef LAUNCH_WG
    NASSEG_LAUNCH_CHECK("dw_wgrad_strip");
  } else {
    NASSEG_REQUIRE(!in_scale && !in_shift && in_act <= NASSEG_ACT_RELU,
                   "dwconv_wgrad: only an input ReLU is supported as prologue on the generic path");
    hipLaunchKernelGGL(dw_wgrad_generic, dim3(gx, gy, 1), dim3(256), 0, s, x, dy, ws, B, H, W, C4,
                       Ho, Wo, K, stride, pad, dil, in_act);
    NASSEG_LAUNCH_CHECK("dw_wgrad_generic");
  }
  if (!dw) return NASSEG_OK;  // partial sums stay in ws for nasseg_wgrad_finalize_many
  hipLaunchKernelGGL(dw_wgrad_finalize, dim3(cdiv(K * K * C, NASSEG_RP_ELEMS)), dim3(256), 0, s, ws,
                     dw, gx * gy, K * K, C);
  NASSEG_LAUNCH_CHECK("dw_wgrad_finalize");
  return NASSEG_OK;
}

int NASSEG_FN(dwconv_wgrad)(const act_t* x, const act_t* dy, float* dw, float* ws,
                        const float* in_scale, const float* in_shift, int in_act, int B, int H, int W,
                        int C, int Ho, int Wo, int K, int stride, int pad, int dil, void* stream) {
  return dw_wgrad_impl(x, dy, dw, ws, in_scale, in_shift, in_act, B, H, W, C, Ho, Wo, K, stride, pad, dil,
                       nullptr, stream);
}

// nasseg_dwconv_wgrad of a depthwise conv whose output z went through a BatchNorm (+ activation),
// fused with the second half of that BatchNorm's backward (see nasseg_conv_wgrad_bn): g = masked
// gradient w.r.t. the BatchNorm output, sums = {sum g, sum g*xhat}; dz (written) = the gradient
// w.r.t. z, which the weight gradient is computed from.  Strip geometries only
// (nasseg_dwconv_strip_ok).
int NASSEG_FN(dwconv_wgrad_bn)(const act_t* x, const act_t* g, const act_t* z, act_t* dz, float* dw, float* ws,
                               const float* in_scale, const float* in_shift, int in_act,
                               const float* bn_scale, const float* bn_shift, const float* bn_mean,
                               const float* bn_invstd, const float* bn_sums, int bn_train, int bn_act, int B,
                               int H, int W, int C, int Ho, int Wo, int K, int stride, int pad, int dil,
                               void* stream) {
  NASSEG_REQUIRE(z && dz && bn_scale && (!bn_train || (bn_mean && bn_invstd && bn_sums)) && (!bn_act || bn_shift),
                 "dwconv_wgrad_bn: missing BatchNorm tensors");
  NASSEG_REQUIRE(B > 0 && Ho > 0 && Wo > 0, "dwconv_wgrad_bn: bad geometry");
  DwWgArgs bn = {};
  bn.z = z; bn.dz = dz; bn.bn_scale = bn_scale; bn.bn_shift = bn_shift; bn.bn_mean = bn_mean;
  bn.bn_invstd = bn_invstd; bn.bn_sums = bn_sums; bn.bn_train = bn_train; bn.bn_act = bn_act;
  bn.invM = (float)(1.0 / ((double)B * Ho * Wo));
  return dw_wgrad_impl(x, g, dw, ws, in_scale, in_shift, in_act, B, H, W, C, Ho, Wo, K, stride, pad, dil,
                       &bn, stream);
}

// First stage of `count` small depthwise layers, strip geometries of one specialisation grouped
// into launches of up to 8 layers running side by side (others are launched one by one):
// desc[16*i..] = x, dy, ws, in_scale, in_shift, in_act, B, H, W, C, Ho, Wo, K, stride, pad, dil -
// the arguments of nasseg_dwconv_wgrad without dw (pointers as integers), which this call
// equals with dw == NULL per layer; finish with nasseg_wgrad_finalize_many.
int NASSEG_FN(dwconv_wgrad_many)(int count, const int64_t* desc, void* stream) {
  NASSEG_REQUIRE(count >= 0 && (count == 0 || desc), "dwconv_wgrad_many: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  constexpr int P = 4;
  std::vector<char> done((size_t)(count > 0 ? count : 1), 0);
  auto key_of = [&](const int64_t* d, StripCfg& sc) {
    const int C = (int)d[9], K = (int)d[12];
    sc = strip_cfg((int)d[13], (int)d[15]);
    const bool strip_ok = (K == 3 || K == 5) && (sc.e == 1 || sc.e == 2) && C % 4 == 0 && C / 4 <= 256;
    if (!strip_ok) return -1;
    if (dw_wgrad_lds_plan(K, (int)d[13], (int)d[14], (int)d[15], C / 4, (int)d[10], (int)d[11]).ok)
      return -1;  // (the LDS-tiled kernel: launched by itself through the ordinary entry point)
    const bool pro = d[3] || d[4] || d[5];
    const int px = dw_fwd_px(K, (int)d[13], (int)d[15], (int)d[11], NASSEG_DW5_WPX);
    return (px > 1 ? 1000 : 0) + K * 100 + sc.e * 10 + (pro ? 1 : 0);
  };
  for (int i = 0; i < count; ++i) {
    if (done[i]) continue;
    const int64_t* di = desc + 16 * (size_t)i;
    StripCfg sci;
    const int key = key_of(di, sci);
    if (key < 0) {  // generic geometry: the ordinary entry point, first stage only
      int rc = NASSEG_FN(dwconv_wgrad)((const act_t*)di[0], (const act_t*)di[1], nullptr, (float*)di[2],
                                       (const float*)di[3], (const float*)di[4], (int)di[5], (int)di[6],
                                       (int)di[7], (int)di[8], (int)di[9], (int)di[10], (int)di[11],
                                       (int)di[12], (int)di[13], (int)di[14], (int)di[15], stream);
      if (rc) return rc;
      done[i] = 1;
      continue;
    }
    DwWgGroup t;
    t.n = 0;
    t.start[0] = 0;
    for (int j = i; j < count && t.n < kDwGroup; ++j) {
      if (done[j]) continue;
      const int64_t* d = desc + 16 * (size_t)j;
      StripCfg sc;
      if (key_of(d, sc) != key) continue;
      const int B = (int)d[6], H = (int)d[7], W = (int)d[8], C = (int)d[9], Ho = (int)d[10], Wo = (int)d[11];
      NASSEG_REQUIRE(B > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, "dwconv_wgrad_many: bad geometry");
      const int C4 = C / 4;
      const int gx = cdiv(Wo * C4, 256);
      const int gy = (int)wgrad_rows(B, C, Ho, Wo);
      DwWgArgs q = {(const act_t*)d[0], (const act_t*)d[1], (float*)d[2], (const float*)d[3],
                    (const float*)d[4], (int)d[5], B, H, W, C4, Ho, Wo, (int)d[13], (int)d[14], (int)d[15],
                    sc.g, cdiv(Ho, P * sc.g)};
      t.a[t.n] = q;
      t.gx[t.n] = gx;
      t.gy[t.n] = gy;
      if (key >= 1000) dw_wgrad_px_grid((int)d[12], (int)d[13], (int)d[15], Wo, C4, gx, gy, &t.gx[t.n], &t.gy[t.n]);
      t.start[t.n + 1] = t.start[t.n] + gx * gy;
      ++t.n;
      done[j] = 1;
    }
    const dim3 grid(t.start[t.n]);
    const int K = (key % 1000) / 100, e = (key / 10) % 10;
    const bool pro = key % 10;
    if (key >= 1000) {
      if (pro) hipLaunchKernelGGL((dw_wgrad_group_kernel<5, P, 1, true, NASSEG_DW5_WPX>), grid, dim3(256), 0, s, t);
      else hipLaunchKernelGGL((dw_wgrad_group_kernel<5, P, 1, false, NASSEG_DW5_WPX>), grid, dim3(256), 0, s, t);
      NASSEG_LAUNCH_CHECK("dw_wgrad_group_kernel");
      continue;
    }
#define GO_DW(KK, EE)                                                                              \
  do {                                                                                             \
    if (pro) hipLaunchKernelGGL((dw_wgrad_group_kernel<KK, P, EE, true>), grid, dim3(256), 0, s, t); \
    else hipLaunchKernelGGL((dw_wgrad_group_kernel<KK, P, EE, false>), grid, dim3(256), 0, s, t);   \
  } while (0)
    if (K == 3 && e == 1) GO_DW(3, 1);
    else if (K == 3 && e == 2) GO_DW(3, 2);
    else if (K == 5 && e == 1) GO_DW(5, 1);
    else GO_DW(5, 2);
#undef GO_DW
    NASSEG_LAUNCH_CHECK("dw_wgrad_group_kernel");
  }
  return NASSEG_OK;
}

#if NASSEG_FP32_ONLY
// grid of nasseg_dwconv_bwd_bn: {columns of workgroups, rows of workgroups}; 0 rows: unsupported
static void dw_bwd_grid(int B, int C, int H, int W, int stride, int* gx, int* gy, int* rpi) {
  const int C4 = C / 4;
  const int Wq = stride == 1 ? W : (W + 1) / 2, Hq = stride == 1 ? H : (H + 1) / 2;
  *gx = cdiv(Wq * C4, 256);
  *rpi = stride == 1 ? 16 : 8;  // input rows (quad rows) per work item: 2 / 16 (1 / 8) extra dz rows
  const int64_t items = (int64_t)B * cdiv(Hq, *rpi);
  int64_t y = 1024 / *gx;
  if (y > items) y = items;
  if (y < 1) y = 1;
  *gy = (int)y;
}
// workgroups (= partial rows of the weight gradient [9][C] and of the statistics [2][C]) of
// nasseg_dwconv_bwd_bn; 0: geometry not served (3x3, pad 1, dilation 1, stride 1 or 2, C % 4 == 0,
// C <= 256)
int64_t nasseg_dwconv_bwd_bn_rows(int B, int C, int H, int W, int K, int stride, int pad, int dil) {
  if (K != 3 || pad != 1 || dil != 1 || (stride != 1 && stride != 2) || C % 4 || C / 4 > 64 || B <= 0 || H <= 0 ||
      W <= 0)
    return 0;
  int gx, gy, rpi;
  dw_bwd_grid(B, C, H, W, stride, &gx, &gy, &rpi);
  return (int64_t)gx * gy;
}
#else
int64_t nasseg_dwconv_bwd_bn_rows(int B, int C, int H, int W, int K, int stride, int pad, int dil);
static void dw_bwd_grid(int B, int C, int H, int W, int stride, int* gx, int* gy, int* rpi) {
  const int C4 = C / 4;
  const int Wq = stride == 1 ? W : (W + 1) / 2, Hq = stride == 1 ? H : (H + 1) / 2;
  *gx = cdiv(Wq * C4, 256);
  *rpi = stride == 1 ? 16 : 8;
  const int64_t items = (int64_t)B * cdiv(Hq, *rpi);
  int64_t y = 1024 / *gx;
  if (y > items) y = items;
  if (y < 1) y = 1;
  *gy = (int)y;
}
#endif  // NASSEG_FP32_ONLY

// Backward of  z = dwconv3x3(act(in_scale*xz + in_shift)),  y = BatchNorm(z) (+ activation), inside a
// chain whose previous link is the BatchNorm (in_*) that produced the conv's input - all of it in one
// kernel (see dw3x3_bwd_bn_kernel):
//   wt: the weight packed [tap][C] (pack kind 3), or its rotated packing (kind 4) with wt_flipped != 0;
//   g: gradient w.r.t. y (masked already: bn_act == 0, or masked here), sums = {sum g', sum g'*xhat};
//   ge (out) [B][H][W][C]: gradient w.r.t. the previous BatchNorm's output, multiplied by in_act';
//   stats (out): rows [r][2][C] of {sum ge, sum ge*xhat_in}, r < nasseg_dwconv_bwd_bn_rows(...);
//   ws: rows [r][9][C] of weight-gradient partials; dw (C,1,3,3) when given, else partials only
//   (nasseg_wgrad_finalize_many: taps 9, N = C, K = 1).
int NASSEG_FN(dwconv_bwd_bn)(const act_t* xz, const act_t* g, const act_t* z, const float* wt, int wt_flipped,
                             act_t* ge, float* dw, float* ws, const float* in_scale, const float* in_shift,
                             const float* in_mean, const float* in_invstd, int in_act, const float* bn_scale,
                             const float* bn_shift, const float* bn_mean, const float* bn_invstd,
                             const float* bn_sums, int bn_train, int bn_act, int B, int H, int W, int C, int Ho,
                             int Wo, int K, int stride, int pad, int dil, float* stats, void* stream) {
  NASSEG_REQUIRE(xz && g && z && wt && ge && ws && stats && in_scale && in_shift && in_mean && in_invstd && bn_scale,
                 "dwconv_bwd_bn: null tensor");
  NASSEG_REQUIRE((!bn_train || (bn_mean && bn_invstd && bn_sums)) && (!bn_act || bn_shift),
                 "dwconv_bwd_bn: missing BatchNorm tensors");
  NASSEG_REQUIRE(nasseg_dwconv_bwd_bn_rows(B, C, H, W, K, stride, pad, dil) > 0,
                 "dwconv_bwd_bn: geometry not served (3x3, pad 1, stride 1 or 2)");
  NASSEG_REQUIRE(Ho == (H + 2 * pad - 3) / stride + 1 && Wo == (W + 2 * pad - 3) / stride + 1,
                 "dwconv_bwd_bn: output size does not match");
  int gx, gy, rpi;
  dw_bwd_grid(B, C, H, W, stride, &gx, &gy, &rpi);
  DwBwdArgs q = {};
  q.xz = xz; q.g = g; q.z = z; q.wt = wt; q.ge = ge; q.partial = ws; q.stats = stats;
  q.in_scale = in_scale; q.in_shift = in_shift; q.in_mean = in_mean; q.in_invstd = in_invstd; q.in_act = in_act;
  q.bn_scale = bn_scale; q.bn_shift = bn_shift; q.bn_mean = bn_mean; q.bn_invstd = bn_invstd; q.bn_sums = bn_sums;
  q.bn_train = bn_train; q.bn_act = bn_act;
  q.invM = (float)(1.0 / ((double)B * Ho * Wo));
  q.B = B; q.H = H; q.W = W; q.C4 = C / 4; q.Ho = Ho; q.Wo = Wo; q.rows_per_item = rpi;
  q.flip = wt_flipped != 0;
  hipStream_t s = (hipStream_t)stream;
  if (stride == 1) hipLaunchKernelGGL(dw3x3_bwd_bn_kernel<1>, dim3(gx, gy), dim3(256), 0, s, q);
  else hipLaunchKernelGGL(dw3x3_bwd_bn_kernel<2>, dim3(gx, gy), dim3(256), 0, s, q);
  NASSEG_LAUNCH_CHECK("dw3x3_bwd_bn_kernel");
  if (!dw) return NASSEG_OK;
  hipLaunchKernelGGL(dw_wgrad_finalize, dim3(cdiv(9 * C, NASSEG_RP_ELEMS)), dim3(256), 0, s, ws, dw, gx * gy, 9, C);
  NASSEG_LAUNCH_CHECK("dw_wgrad_finalize");
  return NASSEG_OK;
}

#if NASSEG_FP32_ONLY
// backward-weight of stride-1 5x5 depthwise layers: 1 (initial) the LDS-tiled kernel where its plan applies, 0 the
// strip kernel; v < 0 only queries.  Returns the previous setting.  Same partial-row layout either way.
int nasseg_dw_wgrad_lds(int v) {
  if (v < 0) return g_dw_wgrad_lds.load();
  return g_dw_wgrad_lds.exchange(v ? 1 : 0);
}
#endif

}  // extern "C"
