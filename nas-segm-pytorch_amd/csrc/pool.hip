// 3x3 (k x k) max / average pooling with padding k/2, fp32 NHWC, gfx950.
//
// Reference: Pool (src/nn/layer_factory.py:161-178):
//   nn.MaxPool2d(k, stride, padding=k//2)  - implicit -inf padding, the first
//     maximum in row-major window order receives the gradient;
//   nn.AvgPool2d(k, stride, padding=k//2, count_include_pad=False) - divides
//     by the number of in-bounds taps.
// Max pooling stores the winning tap index (uint8 per element) so that the
// backward pass is a gather (deterministic, no atomics).
#include <math.h>

#include <atomic>

#include "dw_common.h"

namespace {

inline int pool_grid(int64_t n) {
  int64_t b = (n + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const act_t* __restrict__ x,
                                                          act_t* __restrict__ y,
                                                          uint8_t* __restrict__ idx, int B, int H,
                                                          int W, int C4, int Ho, int Wo, int K,
                                                          int stride, int pad) {
  const int C = C4 * 4;
  const int64_t total = (int64_t)B * Ho * Wo * C4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    int64_t p = i / C4;
    const int ox = (int)(p % Wo);
    p /= Wo;
    const int oy = (int)(p % Ho);
    const int b = (int)(p / Ho);
    const act_t* xb = x + (int64_t)b * H * W * C + c4 * 4;
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    uchar4 mi = make_uchar4(0, 0, 0, 0);
    bool first = true;
    for (int ty = 0; ty < K; ++ty) {
      const int iy = oy * stride - pad + ty;
      if (iy < 0 || iy >= H) continue;
      for (int tx = 0; tx < K; ++tx) {
        const int ix = ox * stride - pad + tx;
        if (ix < 0 || ix >= W) continue;
        const float4 v = lda4(xb + ((int64_t)iy * W + ix) * C);
        const uint8_t t = (uint8_t)(ty * K + tx);
        // torch: take v when (v > max) or isnan(v); the first in-bounds tap seeds the index
        if (first || v.x > m.x || v.x != v.x) { m.x = v.x; mi.x = t; }
        if (first || v.y > m.y || v.y != v.y) { m.y = v.y; mi.y = t; }
        if (first || v.z > m.z || v.z != v.z) { m.z = v.z; mi.z = t; }
        if (first || v.w > m.w || v.w != v.w) { m.w = v.w; mi.w = t; }
        first = false;
      }
    }
    sta4(y + i * 4, m);
    if (idx) *reinterpret_cast<uchar4*>(idx + i * 4) = mi;
  }
}

// dx[iy,ix] = sum over windows (oy,ox) that contain (iy,ix) as tap t and whose idx == t
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const act_t* __restrict__ dy,
                                                          const uint8_t* __restrict__ idx,
                                                          act_t* __restrict__ dx, int B, int H,
                                                          int W, int C4, int Ho, int Wo, int K,
                                                          int stride, int pad) {
  const int C = C4 * 4;
  const int64_t total = (int64_t)B * H * W * C4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    int64_t p = i / C4;
    const int ix = (int)(p % W);
    p /= W;
    const int iy = (int)(p % H);
    const int b = (int)(p / H);
    float4 g = f4zero();
    for (int ty = 0; ty < K; ++ty) {
      const int ny = iy + pad - ty;
      if (ny < 0 || (ny % stride)) continue;
      const int oy = ny / stride;
      if (oy >= Ho) continue;
      for (int tx = 0; tx < K; ++tx) {
        const int nx = ix + pad - tx;
        if (nx < 0 || (nx % stride)) continue;
        const int ox = nx / stride;
        if (ox >= Wo) continue;
        const int64_t o = ((((int64_t)b * Ho + oy) * Wo + ox) * C4 + c4) * 4;
        const uchar4 w = *reinterpret_cast<const uchar4*>(idx + o);
        const float4 d = lda4(dy + o);
        const uint8_t t = (uint8_t)(ty * K + tx);
        if (w.x == t) g.x += d.x;
        if (w.y == t) g.y += d.y;
        if (w.z == t) g.z += d.z;
        if (w.w == t) g.w += d.w;
      }
    }
    sta4(dx + i * 4, g);
  }
}

__device__ __forceinline__ int valid_count(int o, int stride, int pad, int K, int L) {
  int lo = o * stride - pad, hi = lo + K;
  if (lo < 0) lo = 0;
  if (hi > L) hi = L;
  return hi - lo;
}

__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const act_t* __restrict__ x,
                                                          act_t* __restrict__ y, int B, int H,
                                                          int W, int C4, int Ho, int Wo, int K,
                                                          int stride, int pad) {
  const int C = C4 * 4;
  const int64_t total = (int64_t)B * Ho * Wo * C4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    int64_t p = i / C4;
    const int ox = (int)(p % Wo);
    p /= Wo;
    const int oy = (int)(p % Ho);
    const int b = (int)(p / Ho);
    const act_t* xb = x + (int64_t)b * H * W * C + c4 * 4;
    float4 s = f4zero();
    for (int ty = 0; ty < K; ++ty) {
      const int iy = oy * stride - pad + ty;
      if (iy < 0 || iy >= H) continue;
      for (int tx = 0; tx < K; ++tx) {
        const int ix = ox * stride - pad + tx;
        if (ix < 0 || ix >= W) continue;
        s = add4(s, lda4(xb + ((int64_t)iy * W + ix) * C));
      }
    }
    const float n = (float)(valid_count(oy, stride, pad, K, H) * valid_count(ox, stride, pad, K, W));
    sta4(y + i * 4, make_float4(s.x / n, s.y / n, s.z / n, s.w / n));
  }
}

__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const act_t* __restrict__ dy,
                                                          act_t* __restrict__ dx, int B, int H,
                                                          int W, int C4, int Ho, int Wo, int K,
                                                          int stride, int pad) {
  const int64_t total = (int64_t)B * H * W * C4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    int64_t p = i / C4;
    const int ix = (int)(p % W);
    p /= W;
    const int iy = (int)(p % H);
    const int b = (int)(p / H);
    float4 g = f4zero();
    for (int ty = 0; ty < K; ++ty) {
      const int ny = iy + pad - ty;
      if (ny < 0 || (ny % stride)) continue;
      const int oy = ny / stride;
      if (oy >= Ho) continue;
      const int cy = valid_count(oy, stride, pad, K, H);
      for (int tx = 0; tx < K; ++tx) {
        const int nx = ix + pad - tx;
        if (nx < 0 || (nx % stride)) continue;
        const int ox = nx / stride;
        if (ox >= Wo) continue;
        const float n = (float)(cy * valid_count(ox, stride, pad, K, W));
        const float4 d = lda4(dy + ((((int64_t)b * Ho + oy) * Wo + ox) * C4 + c4) * 4);
        g.x += d.x / n;
        g.y += d.y / n;
        g.z += d.z / n;
        g.w += d.w / n;
      }
    }
    sta4(dx + i * 4, g);
  }
}

// ---------------------------------------------------------------------------
// Pool = 1x1 conv + BatchNorm, THEN 3x3 max pooling (src/nn/layer_factory.py:161-178): the pooling reads
// the conv's RAW output z and applies the BatchNorm's affine as it loads (scale*z + shift - monotone only
// for positive scales, so the comparison runs on the transformed values), the normalised map is never
// written; backward, the gather over the <= 9 windows of an input pixel produces the gradient w.r.t. the
// BatchNorm's output together with the per-workgroup sums of that BatchNorm's backward
// {sum g, sum g*xhat} - no reduction pass over g and z.  All loads are unconditional (clamped
// coordinates, masks): the kernels above put their loads under bounds branches, which the compiler
// serialises (0.21-0.29 of the HBM peak).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool3_bn_fwd_kernel(const act_t* __restrict__ z,
                                                              const float* __restrict__ scale,
                                                              const float* __restrict__ shift,
                                                              act_t* __restrict__ y, uint8_t* __restrict__ idx,
                                                              int B, int H, int W, int C4, int Ho, int Wo,
                                                              int stride, int pad) {
  const int C = C4 * 4;
  const int64_t total = (int64_t)B * Ho * Wo * C4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    int64_t p = i / C4;
    const int ox = (int)(p % Wo);
    p /= Wo;
    const int oy = (int)(p % Ho);
    const int b = (int)(p / Ho);
    const act_t* zb = z + (int64_t)b * H * W * C + c4 * 4;
    const float4 sc = scale ? lda4(scale + c4 * 4) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 sh = shift ? lda4(shift + c4 * 4) : f4zero();
    float4 v[9];
    bool ok[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int iy = oy * stride - pad + t / 3, ix = ox * stride - pad + t % 3;
      ok[t] = iy >= 0 && iy < H && ix >= 0 && ix < W;
      const int iyc = iy < 0 ? 0 : (iy >= H ? H - 1 : iy), ixc = ix < 0 ? 0 : (ix >= W ? W - 1 : ix);
      v[t] = lda4(zb + ((int64_t)iyc * W + ixc) * C);
    }
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    int mx = 0, my = 0, mz = 0, mw = 0;
    bool seeded = false;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const float4 a = fma4(v[t], sc, sh);
      // torch: take a when (a > max) or isnan(a); the first in-bounds tap seeds the index
      const bool tx = ok[t] && (!seeded || a.x > m.x || a.x != a.x);
      const bool ty = ok[t] && (!seeded || a.y > m.y || a.y != a.y);
      const bool tz = ok[t] && (!seeded || a.z > m.z || a.z != a.z);
      const bool tw = ok[t] && (!seeded || a.w > m.w || a.w != a.w);
      m.x = tx ? a.x : m.x; mx = tx ? t : mx;
      m.y = ty ? a.y : m.y; my = ty ? t : my;
      m.z = tz ? a.z : m.z; mz = tz ? t : mz;
      m.w = tw ? a.w : m.w; mw = tw ? t : mw;
      seeded = seeded || ok[t];
    }
    sta4(y + i * 4, m);
    if (idx) *reinterpret_cast<uchar4*>(idx + i * 4) = make_uchar4((uint8_t)mx, (uint8_t)my, (uint8_t)mz, (uint8_t)mw);
  }
}

// workgroup (bx, by): lanes = 256 consecutive (x, channel-group) positions of a row of the INPUT grid,
// rows by, by + gdy, ... of the flattened (image, row) axis; g[b][iy][ix] = sum over the windows that took
// this pixel; stats[blk][0][c] = sum g, stats[blk][1][c] = sum g * (z - mean) * invstd
__global__ __launch_bounds__(256) void maxpool3_bn_bwd_kernel(const act_t* __restrict__ dy,
                                                              const uint8_t* __restrict__ idx,
                                                              const act_t* __restrict__ z,
                                                              const float* __restrict__ mean,
                                                              const float* __restrict__ invstd,
                                                              act_t* __restrict__ g, float* __restrict__ stats,
                                                              int B, int H, int W, int C4, int Ho, int Wo,
                                                              int stride, int pad) {
  __shared__ float4 sred[2][4][64];
  const int C = C4 * 4;
  const int tid = threadIdx.x;
  const int base = blockIdx.x * 256;
  const int pos = base + tid;
  const bool live = pos < W * C4;
  const int ix = live ? pos / C4 : 0;
  const int c4 = live ? pos - ix * C4 : 0;
  const float4 mu = lda4(mean + c4 * 4), is = lda4(invstd + c4 * 4);
  // the <= 3 window columns that contain ix (tap column tx): ox = (ix + pad - tx) / stride
  int oxs[3];
  bool oxok[3];
#pragma unroll
  for (int tx = 0; tx < 3; ++tx) {
    const int nx = ix + pad - tx;
    const int ox = stride == 1 ? nx : (nx >> 1);
    oxok[tx] = live && nx >= 0 && (stride == 1 || !(nx & 1)) && ox < Wo;
    oxs[tx] = ox < 0 ? 0 : (ox >= Wo ? Wo - 1 : ox);
  }
  float4 ssum[2] = {f4zero(), f4zero()};
  const int R = B * H;
  for (int r = blockIdx.y; r < R; r += gridDim.y) {
    const int b = r / H, iy = r - b * H;
    float4 acc = f4zero();
    float4 d[9];
    uchar4 w[9];
    bool ok[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int ny = iy + pad - t / 3;
      const int oy = stride == 1 ? ny : (ny >> 1);
      ok[t] = oxok[t % 3] && ny >= 0 && (stride == 1 || !(ny & 1)) && oy < Ho;
      const int oyc = oy < 0 ? 0 : (oy >= Ho ? Ho - 1 : oy);
      const int64_t o = ((((int64_t)b * Ho + oyc) * Wo + oxs[t % 3]) * C4 + c4) * 4;
      w[t] = *reinterpret_cast<const uchar4*>(idx + o);
      d[t] = lda4(dy + o);
    }
    const int64_t off = ((int64_t)r * W + ix) * C + c4 * 4;
    const float4 zv = lda4(z + (live ? off : 0));
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      acc.x += (ok[t] && w[t].x == t) ? d[t].x : 0.f;
      acc.y += (ok[t] && w[t].y == t) ? d[t].y : 0.f;
      acc.z += (ok[t] && w[t].z == t) ? d[t].z : 0.f;
      acc.w += (ok[t] && w[t].w == t) ? d[t].w : 0.f;
    }
    if (live) sta4(g + off, acc);
#ifdef NASSEG_BF16
    acc = make_float4(bf16_to_f32(f32_to_bf16(acc.x)), bf16_to_f32(f32_to_bf16(acc.y)),
                      bf16_to_f32(f32_to_bf16(acc.z)), bf16_to_f32(f32_to_bf16(acc.w)));  // (what a reduction pass would read)
#endif
    const float4 gm = keep_if(acc, live);
    ssum[0] = add4(ssum[0], gm);
    ssum[1] = fma4(gm, make_float4((zv.x - mu.x) * is.x, (zv.y - mu.y) * is.y, (zv.z - mu.z) * is.z,
                                   (zv.w - mu.w) * is.w), ssum[1]);
  }
  const int blk = blockIdx.y * gridDim.x + blockIdx.x;
  block_reduce_groups<2, 2>(ssum, sred, stats + (size_t)blk * 2 * C, base, C4);
}

// ---- stride 1: vertical strips ------------------------------------------------------------------------------
// The kernels above load 9 taps per output (forward) / 9 (gradient, winner) pairs per input pixel (backward): at
// stride 1 every element is fetched nine times, and they ran at 2.2 - 2.4 TB/s where the depthwise strips - same
// access pattern - reach 4.5.  Here a thread owns P (4 or 2) consecutive rows of one (x, channel-group) column: the
// (P + 2) x 3 values its windows touch are loaded ONCE (unconditionally, clamped coordinates + masks) and serve all
// P rows - 4.5 (6) loads per element instead of 9, the BatchNorm affine once per loaded value.  Tap order, winner rule
// and arithmetic per element are those of the kernels above: identical outputs.
template <int P>
__global__ __launch_bounds__(256) void maxpool3_bn_fwd_strip_kernel(const act_t* __restrict__ z,
                                                                    const float* __restrict__ scale,
                                                                    const float* __restrict__ shift,
                                                                    act_t* __restrict__ y, uint8_t* __restrict__ idx,
                                                                    int B, int H, int W, int C4) {
  const int C = C4 * 4;
  const int chunks = (H + P - 1) / P;
  const int64_t total = (int64_t)B * chunks * W * C4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    int64_t p = i / C4;
    const int ox = (int)(p % W);
    p /= W;
    const int ch = (int)(p % chunks);
    const int b = (int)(p / chunks);
    const int oy0 = ch * P;
    const act_t* zb = z + (int64_t)b * H * W * C + c4 * 4;
    const float4 sc = scale ? lda4(scale + c4 * 4) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 sh = shift ? lda4(shift + c4 * 4) : f4zero();
    float4 v[P + 2][3];
    bool ok[P + 2][3];
#pragma unroll
    for (int r = 0; r < P + 2; ++r) {
      const int iy = oy0 - 1 + r;
      const int iyc = iy < 0 ? 0 : (iy >= H ? H - 1 : iy);
#pragma unroll
      for (int tx = 0; tx < 3; ++tx) {
        const int ix = ox - 1 + tx;
        const int ixc = ix < 0 ? 0 : (ix >= W ? W - 1 : ix);
        ok[r][tx] = iy >= 0 && iy < H && ix >= 0 && ix < W;
        v[r][tx] = lda4(zb + ((int64_t)iyc * W + ixc) * C);
      }
    }
#pragma unroll
    for (int r = 0; r < P + 2; ++r)
#pragma unroll
      for (int tx = 0; tx < 3; ++tx) v[r][tx] = fma4(v[r][tx], sc, sh);
#pragma unroll
    for (int j = 0; j < P; ++j) {
      float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
      int mx = 0, my = 0, mz = 0, mw = 0;
      bool seeded = false;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const float4 a = v[j + t / 3][t % 3];
        const bool in = ok[j + t / 3][t % 3];
        // torch: take a when (a > max) or isnan(a); the first in-bounds tap seeds the index
        const bool tx_ = in && (!seeded || a.x > m.x || a.x != a.x);
        const bool ty_ = in && (!seeded || a.y > m.y || a.y != a.y);
        const bool tz_ = in && (!seeded || a.z > m.z || a.z != a.z);
        const bool tw_ = in && (!seeded || a.w > m.w || a.w != a.w);
        m.x = tx_ ? a.x : m.x; mx = tx_ ? t : mx;
        m.y = ty_ ? a.y : m.y; my = ty_ ? t : my;
        m.z = tz_ ? a.z : m.z; mz = tz_ ? t : mz;
        m.w = tw_ ? a.w : m.w; mw = tw_ ? t : mw;
        seeded = seeded || in;
      }
      const int oy = oy0 + j;
      if (oy < H) {
        const int64_t o = ((((int64_t)b * H + oy) * W + ox) * C4 + c4) * 4;
        sta4(y + o, m);
        if (idx) *reinterpret_cast<uchar4*>(idx + o) = make_uchar4((uint8_t)mx, (uint8_t)my, (uint8_t)mz, (uint8_t)mw);
      }
    }
  }
}

// workgroup (bx, by): lanes = 256 consecutive (x, channel-group) positions of an input row; chunks of P rows
// by, by + gdy, ... of the flattened (image, row / P) axis.  Same outputs and row layout as maxpool3_bn_bwd_kernel.
template <int P>
__global__ __launch_bounds__(256) void maxpool3_bn_bwd_strip_kernel(const act_t* __restrict__ dy,
                                                                    const uint8_t* __restrict__ idx,
                                                                    const act_t* __restrict__ z,
                                                                    const float* __restrict__ mean,
                                                                    const float* __restrict__ invstd,
                                                                    act_t* __restrict__ g, float* __restrict__ stats,
                                                                    int B, int H, int W, int C4) {
  __shared__ float4 sred[2][4][64];
  const int C = C4 * 4;
  const int tid = threadIdx.x;
  const int base = blockIdx.x * 256;
  const int pos = base + tid;
  const bool live = pos < W * C4;
  const int ix = live ? pos / C4 : 0;
  const int c4 = live ? pos - ix * C4 : 0;
  const float4 mu = lda4(mean + c4 * 4), is = lda4(invstd + c4 * 4);
  // the three window columns that contain ix: tap column tx belongs to the window at ox = ix + 1 - tx
  int oxs[3];
  bool oxok[3];
#pragma unroll
  for (int tx = 0; tx < 3; ++tx) {
    const int ox = ix + 1 - tx;
    oxok[tx] = live && ox >= 0 && ox < W;
    oxs[tx] = ox < 0 ? 0 : (ox >= W ? W - 1 : ox);
  }
  float4 ssum[2] = {f4zero(), f4zero()};
  const int chunks = (H + P - 1) / P;
  const int R = B * chunks;
  for (int r = blockIdx.y; r < R; r += gridDim.y) {
    const int b = r / chunks, iy0 = (r - b * chunks) * P;
    // output rows iy0 - 1 ... iy0 + P (the windows of input row iy are at oy = iy + 1 - ty)
    float4 d[P + 2][3];
    uchar4 w[P + 2][3];
    bool rok[P + 2];
#pragma unroll
    for (int q = 0; q < P + 2; ++q) {
      const int oy = iy0 - 1 + q;
      rok[q] = oy >= 0 && oy < H;
      const int oyc = oy < 0 ? 0 : (oy >= H ? H - 1 : oy);
#pragma unroll
      for (int tx = 0; tx < 3; ++tx) {
        const int64_t o = ((((int64_t)b * H + oyc) * W + oxs[tx]) * C4 + c4) * 4;
        w[q][tx] = *reinterpret_cast<const uchar4*>(idx + o);
        d[q][tx] = lda4(dy + o);
      }
    }
    float4 zv[P];
#pragma unroll
    for (int j = 0; j < P; ++j) {
      const int iy = iy0 + j < H ? iy0 + j : H - 1;
      zv[j] = lda4(z + (((int64_t)b * H + iy) * W + ix) * C + c4 * 4);
    }
#pragma unroll
    for (int j = 0; j < P; ++j) {
      float4 acc = f4zero();
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        // tap (ty, tx) of the window at output row iy + 1 - ty = iy0 - 1 + (j + 2 - ty)
        const int q = j + 2 - t / 3, tx = t % 3;
        const bool in = rok[q] && oxok[tx];
        acc.x += (in && w[q][tx].x == t) ? d[q][tx].x : 0.f;
        acc.y += (in && w[q][tx].y == t) ? d[q][tx].y : 0.f;
        acc.z += (in && w[q][tx].z == t) ? d[q][tx].z : 0.f;
        acc.w += (in && w[q][tx].w == t) ? d[q][tx].w : 0.f;
      }
      const bool rowok = live && iy0 + j < H;
      if (rowok) sta4(g + (((int64_t)b * H + iy0 + j) * W + ix) * C + c4 * 4, acc);
#ifdef NASSEG_BF16
      acc = make_float4(bf16_to_f32(f32_to_bf16(acc.x)), bf16_to_f32(f32_to_bf16(acc.y)),
                        bf16_to_f32(f32_to_bf16(acc.z)), bf16_to_f32(f32_to_bf16(acc.w)));  // (what a reduction pass would read)
#endif
      const float4 gm = keep_if(acc, rowok);
      ssum[0] = add4(ssum[0], gm);
      ssum[1] = fma4(gm, make_float4((zv[j].x - mu.x) * is.x, (zv[j].y - mu.y) * is.y, (zv[j].z - mu.z) * is.z,
                                     (zv[j].w - mu.w) * is.w), ssum[1]);
    }
  }
  const int blk = blockIdx.y * gridDim.x + blockIdx.x;
  block_reduce_groups<2, 2>(ssum, sred, stats + (size_t)blk * 2 * C, base, C4);
}

struct PoolBnGrid {
  int gx, gy;
};
#if NASSEG_FP32_ONLY
// stride-1 max pooling on the strip kernels: 1 (initial) four rows per thread, 2 two rows per thread; 0: one tap
// gather per element (A/B switch)
std::atomic<int> g_pool_strip{1};
#endif
extern "C" int nasseg_pool_strip(int v);

inline int pool_strip_rows(int mode) { return mode == 1 ? 4 : (mode == 2 ? 2 : 1); }
inline PoolBnGrid pool_bn_grid(int B, int H, int W, int C, int strip = 0) {
  PoolBnGrid g;
  g.gx = cdiv(W * (C / 4), 256);
  int64_t gy = 2048 / g.gx;  // ~2048 workgroups, at least two rows (strip kernel: chunks of rows) each
  const int64_t rows = (int64_t)B * cdiv(H, pool_strip_rows(strip));
  if (gy > rows / 2) gy = rows / 2;
  if (gy < 1) gy = 1;
  if (gy > 65535) gy = 65535;
  g.gy = (int)gy;
  return g;
}

}  // namespace

extern "C" {

#if NASSEG_FP32_ONLY
// statistics rows of nasseg_maxpool_bn_bwd (0: geometry not served - use the unfused ops)
int64_t nasseg_maxpool_bn_bwd_blocks(int B, int H, int W, int C, int K, int stride, int pad) {
  if (K != 3 || pad != 1 || (stride != 1 && stride != 2) || C % 4 || C / 4 > 256 || B <= 0 || H <= 0 || W <= 0) return 0;
  const PoolBnGrid g = pool_bn_grid(B, H, W, C, stride == 1 ? nasseg_pool_strip(-1) : 0);
  return (int64_t)g.gx * g.gy;
}
// stride-1 3x3 max pooling on the strip kernels (four rows per thread: 4.5 loads per element instead of 9): 1 (initial)
// / 0; v < 0 only queries.  Returns the previous setting.  Identical outputs; the statistics rows are other partitions of
// the same sums.
int nasseg_pool_strip(int v) {
  if (v < 0) return g_pool_strip.load();
  return g_pool_strip.exchange(v > 2 ? 1 : v);
}
#else
int64_t nasseg_maxpool_bn_bwd_blocks(int B, int H, int W, int C, int K, int stride, int pad);
#endif

// y = maxpool3x3(scale*z + shift) (padding 1, stride 1 or 2; scale / shift null = identity); idx: uint8 winner
// tap per output element (null when no backward will follow)
int NASSEG_FN(maxpool_bn_fwd)(const act_t* z, const float* scale, const float* shift, act_t* y, uint8_t* idx,
                              int B, int H, int W, int C, int Ho, int Wo, int stride, int pad, void* stream) {
  NASSEG_REQUIRE(C > 0 && C % 4 == 0 && stride > 0 && pad >= 0 && pad <= 2, "maxpool_bn_fwd: bad arguments");
  const int64_t n4 = (int64_t)B * Ho * Wo * (C / 4);
  const int strip = nasseg_pool_strip(-1);
  if (stride == 1 && pad == 1 && Ho == H && Wo == W && strip) {
    const dim3 grid(pool_grid((int64_t)B * cdiv(H, pool_strip_rows(strip)) * W * (C / 4)));
    if (strip == 1)
      hipLaunchKernelGGL(maxpool3_bn_fwd_strip_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, z, scale, shift, y,
                         idx, B, H, W, C / 4);
    else
      hipLaunchKernelGGL(maxpool3_bn_fwd_strip_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, z, scale, shift, y,
                         idx, B, H, W, C / 4);
    NASSEG_LAUNCH_CHECK("maxpool_bn_fwd");
    return NASSEG_OK;
  }
  hipLaunchKernelGGL(maxpool3_bn_fwd_kernel, dim3(pool_grid(n4)), dim3(256), 0, (hipStream_t)stream, z, scale,
                     shift, y, idx, B, H, W, C / 4, Ho, Wo, stride, pad);
  NASSEG_LAUNCH_CHECK("maxpool_bn_fwd");
  return NASSEG_OK;
}

// backward of the above w.r.t. the BatchNorm's output: g [B][H][W][C] and the rows
// stats[blk][2][C] = {sum g, sum g*(z-mean)*invstd} for blk < nasseg_maxpool_bn_bwd_blocks(...) (the buffer
// needs 64 rows more: nasseg_rows_sum)
int NASSEG_FN(maxpool_bn_bwd)(const act_t* dy, const uint8_t* idx, const act_t* z, const float* mean,
                              const float* invstd, act_t* g, float* stats, int B, int H, int W, int C, int Ho,
                              int Wo, int stride, int pad, void* stream) {
  NASSEG_REQUIRE(dy && idx && z && mean && invstd && g && stats, "maxpool_bn_bwd: null argument");
  NASSEG_REQUIRE(nasseg_maxpool_bn_bwd_blocks(B, H, W, C, 3, stride, pad) > 0, "maxpool_bn_bwd: geometry not served");
  const int strip = stride == 1 ? nasseg_pool_strip(-1) : 0;
  NASSEG_REQUIRE(!strip || (Ho == H && Wo == W), "maxpool_bn_bwd: stride 1 keeps the size");
  const PoolBnGrid gr = pool_bn_grid(B, H, W, C, strip);
  if (strip == 1)
    hipLaunchKernelGGL(maxpool3_bn_bwd_strip_kernel<4>, dim3(gr.gx, gr.gy), dim3(256), 0, (hipStream_t)stream, dy, idx,
                       z, mean, invstd, g, stats, B, H, W, C / 4);
  else if (strip == 2)
    hipLaunchKernelGGL(maxpool3_bn_bwd_strip_kernel<2>, dim3(gr.gx, gr.gy), dim3(256), 0, (hipStream_t)stream, dy, idx,
                       z, mean, invstd, g, stats, B, H, W, C / 4);
  else
    hipLaunchKernelGGL(maxpool3_bn_bwd_kernel, dim3(gr.gx, gr.gy), dim3(256), 0, (hipStream_t)stream, dy, idx, z,
                       mean, invstd, g, stats, B, H, W, C / 4, Ho, Wo, stride, pad);
  NASSEG_LAUNCH_CHECK("maxpool_bn_bwd");
  return NASSEG_OK;
}

// mode 0 = max (idx: uint8 [B][Ho][Wo][C] winner tap, may be null), 1 = avg
int NASSEG_FN(pool_fwd)(int mode, const act_t* x, act_t* y, uint8_t* idx, int B, int H, int W, int C,
                    int Ho, int Wo, int K, int stride, int pad, void* stream) {
  NASSEG_REQUIRE(C > 0 && C % 4 == 0, "pool_fwd: C=%d must be a multiple of 4", C);
  NASSEG_REQUIRE(K > 0 && K * K <= 255 && stride > 0, "pool_fwd: bad window");
  const int64_t n4 = (int64_t)B * Ho * Wo * (C / 4);
  hipStream_t s = (hipStream_t)stream;
  if (mode == 0)
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(pool_grid(n4)), dim3(256), 0, s, x, y, idx, B, H, W,
                       C / 4, Ho, Wo, K, stride, pad);
  else
    hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(pool_grid(n4)), dim3(256), 0, s, x, y, B, H, W,
                       C / 4, Ho, Wo, K, stride, pad);
  NASSEG_LAUNCH_CHECK("pool_fwd");
  return NASSEG_OK;
}

int NASSEG_FN(pool_bwd)(int mode, const act_t* dy, const uint8_t* idx, act_t* dx, int B, int H, int W,
                    int C, int Ho, int Wo, int K, int stride, int pad, void* stream) {
  NASSEG_REQUIRE(C > 0 && C % 4 == 0, "pool_bwd: C=%d must be a multiple of 4", C);
  NASSEG_REQUIRE(mode != 0 || idx, "pool_bwd: max pooling needs the index tensor");
  const int64_t n4 = (int64_t)B * H * W * (C / 4);
  hipStream_t s = (hipStream_t)stream;
  if (mode == 0)
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(pool_grid(n4)), dim3(256), 0, s, dy, idx, dx, B, H,
                       W, C / 4, Ho, Wo, K, stride, pad);
  else
    hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(pool_grid(n4)), dim3(256), 0, s, dy, dx, B, H, W,
                       C / 4, Ho, Wo, K, stride, pad);
  NASSEG_LAUNCH_CHECK("pool_bwd");
  return NASSEG_OK;
}

}  // extern "C"
