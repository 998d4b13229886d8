// Per-pixel log-softmax + NLL loss with ignore index, fp32 NHWC logits, gfx950.
//
// Reference: nn.LogSoftmax() (implicit dim=1) followed by
// nn.NLLLoss2d(ignore_index=255) (src/main_search.py:435,
// src/engine/trainer.py:144-146,156-158,239-241,248-250):
//   loss = mean over pixels with target != ignore of -log_softmax(logits)[target]
// (an all-ignored batch gives 0/0 = NaN, as in torch).
// In NHWC the C class scores of one pixel are contiguous, so one lane owns one
// pixel; the forward also emits the gradient wrt the logits for a unit upstream
// gradient, so backward is a single scale.
#include <math.h>

#include "common.h"

namespace {

template <typename TL>
__global__ __launch_bounds__(256) void ce_fwd_kernel(const act_t* __restrict__ logits,
                                                     const TL* __restrict__ target, int64_t P,
                                                     int C, int ignore, float* __restrict__ partial) {
  __shared__ float red_l[256];
  __shared__ float red_n[256];
  float loss = 0.f, cnt = 0.f;
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < P; p += (int64_t)gridDim.x * 256) {
    const int64_t t = (int64_t)target[p];
    if (t == ignore || t < 0 || t >= C) continue;  // out-of-range labels are skipped, never read
    const act_t* lp = logits + p * C;
    float m = lda1(lp);
    for (int c = 1; c < C; ++c) m = fmaxf(m, lda1(lp + c));
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(lda1(lp + c) - m);
    const float lse = m + logf(s);
    loss += lse - lda1(lp + t);
    cnt += 1.f;
  }
  red_l[threadIdx.x] = loss;
  red_n[threadIdx.x] = cnt;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      red_l[threadIdx.x] += red_l[threadIdx.x + s];
      red_n[threadIdx.x] += red_n[threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    partial[blockIdx.x * 2] = red_l[0];
    partial[blockIdx.x * 2 + 1] = red_n[0];
  }
}

// out[0] = loss (mean), out[1] = number of valid pixels
__global__ void ce_finalize_kernel(const float* __restrict__ partial, int nblk,
                                   float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double l = 0.0, n = 0.0;
  for (int b = 0; b < nblk; ++b) {
    l += (double)partial[b * 2];
    n += (double)partial[b * 2 + 1];
  }
  out[0] = (float)(l / n);
  out[1] = (float)n;
}

// dlogits[p][c] = gscale[0] * (softmax(p)[c] - [c == target]) / nvalid   (0 for ignored pixels)
template <typename TL>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const act_t* __restrict__ logits,
                                                     const TL* __restrict__ target,
                                                     const float* __restrict__ stats,
                                                     const float* __restrict__ gscale, int64_t P,
                                                     int C, int ignore, act_t* __restrict__ dlogits) {
  const float g = (gscale ? gscale[0] : 1.f) / stats[1];
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < P; p += (int64_t)gridDim.x * 256) {
    const int64_t t = (int64_t)target[p];
    const act_t* lp = logits + p * C;
    act_t* dp = dlogits + p * C;
    if (t == ignore || t < 0 || t >= C) {
      for (int c = 0; c < C; ++c) sta1(dp + c, 0.f);
      continue;
    }
    float m = lda1(lp);
    for (int c = 1; c < C; ++c) m = fmaxf(m, lda1(lp + c));
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(lda1(lp + c) - m);
    const float inv = 1.f / s;
    for (int c = 0; c < C; ++c) {
      float sm = expf(lda1(lp + c) - m) * inv;
      sta1(dp + c, g * (sm - ((int64_t)c == t ? 1.f : 0.f)));
    }
  }
}

// ---------------------------------------------------------------------------
// berHu (reverse Huber) loss for the depth head (BASELINE config 5).  Not present in
// the reference ("parity unpinned"): Laina et al. 2016, eq. 2 -
//   B(d) = |d| if |d| <= c else (d^2 + c^2) / (2c),  c = 0.2 * max|d| over the batch,
// mean over all elements; c is treated as a constant in the backward pass.
// Three tiny passes: block maxima -> c, block sums -> mean.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void berhu_max_kernel(const act_t* __restrict__ pred,
                                                        const act_t* __restrict__ target, int64_t n,
                                                        float* __restrict__ partial) {
  __shared__ float red[256];
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    m = fmaxf(m, fabsf(lda1(pred + i) - lda1(target + i)));
  red[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void berhu_sum_kernel(const act_t* __restrict__ pred,
                                                        const act_t* __restrict__ target, int64_t n,
                                                        const float* __restrict__ maxpart, int nblk,
                                                        float* __restrict__ partial,
                                                        float* __restrict__ out) {
  __shared__ float red[256];
  __shared__ float cs;
  if (threadIdx.x == 0) {
    float m = 0.f;
    for (int b = 0; b < nblk; ++b) m = fmaxf(m, maxpart[b]);
    cs = 0.2f * m;
    if (blockIdx.x == 0) out[1] = cs;
  }
  __syncthreads();
  const float c = cs;
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float d = fabsf(lda1(pred + i) - lda1(target + i));
    acc += (d <= c) ? d : (d * d + c * c) / (2.f * c);
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ void berhu_finalize_kernel(const float* __restrict__ partial, int nblk, double n,
                                      float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double s = 0.0;
  for (int b = 0; b < nblk; ++b) s += (double)partial[b];
  out[0] = (float)(s / n);
}

// dpred = g/n * (sign(diff) if |diff| <= c else diff / c)
__global__ __launch_bounds__(256) void berhu_bwd_kernel(const act_t* __restrict__ pred,
                                                        const act_t* __restrict__ target,
                                                        const float* __restrict__ stats,
                                                        const float* __restrict__ gscale, int64_t n,
                                                        act_t* __restrict__ dpred) {
  const float c = stats[1];
  const float g = (gscale ? gscale[0] : 1.f) / (float)n;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float d = lda1(pred + i) - lda1(target + i);
    const float ad = fabsf(d);
    const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    sta1(dpred + i, g * ((ad <= c) ? sgn : d / c));
  }
}

inline int ce_grid(int64_t P) {
  int64_t b = (P + 255) / 256;
  if (b > 1024) b = 1024;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" {

#if NASSEG_FP32_ONLY
int64_t nasseg_ce_workspace(void) { return 2 * 1024; }
#endif

// logits [P][C] dense NHWC, target [P] (elem_size 1 = uint8, 8 = int64).
// out[0] = mean NLL over valid pixels, out[1] = valid count. ws: nasseg_ce_workspace() floats.
int NASSEG_FN(ce_fwd)(const act_t* logits, const void* target, int elem_size, int64_t P, int C,
                  int ignore, float* out, float* ws, void* stream) {
  NASSEG_REQUIRE(P > 0 && C > 0, "ce_fwd: bad shape");
  hipStream_t s = (hipStream_t)stream;
  const int grid = ce_grid(P);
  if (elem_size == 8)
    hipLaunchKernelGGL((ce_fwd_kernel<int64_t>), dim3(grid), dim3(256), 0, s, logits,
                       (const int64_t*)target, P, C, ignore, ws);
  else if (elem_size == 1)
    hipLaunchKernelGGL((ce_fwd_kernel<uint8_t>), dim3(grid), dim3(256), 0, s, logits,
                       (const uint8_t*)target, P, C, ignore, ws);
  else
    return nasseg_fail(NASSEG_ERR_ARG, "ce_fwd: elem_size %d not supported", elem_size);
  NASSEG_LAUNCH_CHECK("ce_fwd");
  hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(64), 0, s, ws, grid, out);
  NASSEG_LAUNCH_CHECK("ce_finalize");
  return NASSEG_OK;
}

// stats = out of nasseg_ce_fwd; gscale = device scalar upstream gradient (null = 1)
int NASSEG_FN(ce_bwd)(const act_t* logits, const void* target, int elem_size, const float* stats,
                  const float* gscale, int64_t P, int C, int ignore, act_t* dlogits, void* stream) {
  NASSEG_REQUIRE(P > 0 && C > 0, "ce_bwd: bad shape");
  hipStream_t s = (hipStream_t)stream;
  const int grid = ce_grid(P) * 2;
  if (elem_size == 8)
    hipLaunchKernelGGL((ce_bwd_kernel<int64_t>), dim3(grid), dim3(256), 0, s, logits,
                       (const int64_t*)target, stats, gscale, P, C, ignore, dlogits);
  else if (elem_size == 1)
    hipLaunchKernelGGL((ce_bwd_kernel<uint8_t>), dim3(grid), dim3(256), 0, s, logits,
                       (const uint8_t*)target, stats, gscale, P, C, ignore, dlogits);
  else
    return nasseg_fail(NASSEG_ERR_ARG, "ce_bwd: elem_size %d not supported", elem_size);
  NASSEG_LAUNCH_CHECK("ce_bwd");
  return NASSEG_OK;
}

// berHu loss (depth head): out[0] = mean loss, out[1] = c = 0.2 * max|pred - target|.
// pred / target: n fp32 elements, any layout (elementwise).  ws: nasseg_ce_workspace() floats.
int NASSEG_FN(berhu_fwd)(const act_t* pred, const act_t* target, int64_t n, float* out, float* ws,
                     void* stream) {
  NASSEG_REQUIRE(n > 0, "berhu_fwd: empty input");
  hipStream_t s = (hipStream_t)stream;
  const int grid = ce_grid(n);
  hipLaunchKernelGGL(berhu_max_kernel, dim3(grid), dim3(256), 0, s, pred, target, n, ws);
  NASSEG_LAUNCH_CHECK("berhu_max");
  hipLaunchKernelGGL(berhu_sum_kernel, dim3(grid), dim3(256), 0, s, pred, target, n, ws, grid,
                     ws + 1024, out);
  NASSEG_LAUNCH_CHECK("berhu_sum");
  hipLaunchKernelGGL(berhu_finalize_kernel, dim3(1), dim3(64), 0, s, ws + 1024, grid, (double)n, out);
  NASSEG_LAUNCH_CHECK("berhu_finalize");
  return NASSEG_OK;
}

int NASSEG_FN(berhu_bwd)(const act_t* pred, const act_t* target, const float* stats,
                     const float* gscale, int64_t n, act_t* dpred, void* stream) {
  NASSEG_REQUIRE(n > 0, "berhu_bwd: empty input");
  hipLaunchKernelGGL(berhu_bwd_kernel, dim3(ce_grid(n) * 2), dim3(256), 0, (hipStream_t)stream, pred,
                     target, stats, gscale, n, dpred);
  NASSEG_LAUNCH_CHECK("berhu_bwd");
  return NASSEG_OK;
}

}  // extern "C"
