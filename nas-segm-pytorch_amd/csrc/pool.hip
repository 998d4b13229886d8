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

#include "common.h"

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

}  // namespace

extern "C" {

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
