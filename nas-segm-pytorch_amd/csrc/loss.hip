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

// out[0] = loss (mean), out[1] = number of valid pixels.  One workgroup of 256 threads: thread t
// adds the partials of blocks t, t+256, ... in fp64, then a fixed-order tree through LDS.
__global__ __launch_bounds__(256) void ce_finalize_kernel(const float* __restrict__ partial, int nblk,
                                                          float* __restrict__ out) {
  __shared__ double red_l[256];
  __shared__ double red_n[256];
  double l = 0.0, n = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 256) {
    l += (double)partial[b * 2];
    n += (double)partial[b * 2 + 1];
  }
  red_l[threadIdx.x] = l;
  red_n[threadIdx.x] = n;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      red_l[threadIdx.x] += red_l[threadIdx.x + s];
      red_n[threadIdx.x] += red_n[threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = (float)(red_l[0] / red_n[0]);
    out[1] = (float)red_n[0];
  }
}

// The same two computations for C <= 63 with the scores staged through LDS: a workgroup's 256
// pixels are 256*C CONTIGUOUS floats, loaded (and, backward, stored) with fully coalesced
// accesses; each lane then works on its pixel's row in LDS (row stride C|1 is odd: no bank
// conflicts).  One lane per pixel straight from HBM touches 64 rows 4*C bytes apart per load.
// Pixel -> (workgroup, thread) assignment, per-pixel arithmetic and reduction order are those
// of ce_fwd_kernel / ce_bwd_kernel: bit-identical results.
template <typename TL, bool BWD>
__global__ __launch_bounds__(256) void ce_tile_kernel(const act_t* __restrict__ logits,
                                                      const TL* __restrict__ target, int64_t P, int C,
                                                      int ignore, float* __restrict__ partial,
                                                      const float* __restrict__ stats,
                                                      const float* __restrict__ gscale,
                                                      act_t* __restrict__ dlogits) {
  extern __shared__ float tile[];  // [256][C | 1]
  __shared__ float red_l[256];
  __shared__ float red_n[256];
  const int CS = C | 1;
  const int tid = threadIdx.x;
  const int64_t ntiles = (P + 255) / 256;
  float loss = 0.f, cnt = 0.f;
  float g = 0.f;
  if (BWD) g = (gscale ? gscale[0] : 1.f) / stats[1];
  for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int64_t p0 = t * 256;
    const int np = (int)((P - p0) < 256 ? (P - p0) : 256);
    const int nel = np * C;
    const act_t* src = logits + p0 * C;
    const int nel4 = nel >> 2;  // (p0*C is a multiple of 4: vector accesses are aligned)
    for (int i = tid; i < nel4; i += 256) {
      const float4 v = lda4(src + 4 * i);
      int pix = (4 * i) / C;
      int c = 4 * i - pix * C;
      const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        tile[pix * CS + c] = e[r];
        if (++c == C) {
          c = 0;
          ++pix;
        }
      }
    }
    for (int i = 4 * nel4 + tid; i < nel; i += 256) {
      const int pix = i / C;
      tile[pix * CS + (i - pix * C)] = lda1(src + i);
    }
    __syncthreads();
    if (tid < np) {
      float* row = tile + tid * CS;
      const int64_t tg = (int64_t)target[p0 + tid];
      const bool skip = tg == ignore || tg < 0 || tg >= C;
      if (BWD && skip) {
        for (int c = 0; c < C; ++c) row[c] = 0.f;
      } else if (!skip) {
        float m = row[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, row[c]);
        float sum = 0.f;
        for (int c = 0; c < C; ++c) {
          const float e = expf(row[c] - m);
          if (BWD) row[c] = e;  // (kept: the softmax needs it again)
          sum += e;
        }
        if (BWD) {
          const float inv = 1.f / sum;
          for (int c = 0; c < C; ++c) row[c] = g * (row[c] * inv - ((int64_t)c == tg ? 1.f : 0.f));
        } else {
          loss += (m + logf(sum)) - row[tg];
          cnt += 1.f;
        }
      }
    }
    __syncthreads();
    if (BWD) {
      act_t* dst = dlogits + p0 * C;
      for (int i = tid; i < nel4; i += 256) {
        int pix = (4 * i) / C;
        int c = 4 * i - pix * C;
        float e[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          e[r] = tile[pix * CS + c];
          if (++c == C) {
            c = 0;
            ++pix;
          }
        }
        sta4(dst + 4 * i, make_float4(e[0], e[1], e[2], e[3]));
      }
      for (int i = 4 * nel4 + tid; i < nel; i += 256) {
        const int pix = i / C;
        sta1(dst + i, tile[pix * CS + (i - pix * C)]);
      }
      __syncthreads();
    }
  }
  if (!BWD) {
    red_l[tid] = loss;
    red_n[tid] = cnt;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (tid < s) {
        red_l[tid] += red_l[tid + s];
        red_n[tid] += red_n[tid + s];
      }
      __syncthreads();
    }
    if (tid == 0) {
      partial[blockIdx.x * 2] = red_l[0];
      partial[blockIdx.x * 2 + 1] = red_n[0];
    }
  }
}
constexpr int kCeTileMaxC = 63;

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
  NASSEG_REQUIRE(elem_size == 8 || elem_size == 1, "ce_fwd: elem_size %d not supported", elem_size);
  const size_t lds = (size_t)256 * (C | 1) * sizeof(float);
  const bool tiled = C <= kCeTileMaxC && ((uintptr_t)logits & 15) == 0;  // (vector staging loads)
  if (tiled && elem_size == 8)
    hipLaunchKernelGGL((ce_tile_kernel<int64_t, false>), dim3(grid), dim3(256), lds, s, logits,
                       (const int64_t*)target, P, C, ignore, ws, nullptr, nullptr, nullptr);
  else if (tiled)
    hipLaunchKernelGGL((ce_tile_kernel<uint8_t, false>), dim3(grid), dim3(256), lds, s, logits,
                       (const uint8_t*)target, P, C, ignore, ws, nullptr, nullptr, nullptr);
  else if (elem_size == 8)
    hipLaunchKernelGGL((ce_fwd_kernel<int64_t>), dim3(grid), dim3(256), 0, s, logits,
                       (const int64_t*)target, P, C, ignore, ws);
  else
    hipLaunchKernelGGL((ce_fwd_kernel<uint8_t>), dim3(grid), dim3(256), 0, s, logits,
                       (const uint8_t*)target, P, C, ignore, ws);
  NASSEG_LAUNCH_CHECK("ce_fwd");
  hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(256), 0, s, ws, grid, out);
  NASSEG_LAUNCH_CHECK("ce_finalize");
  return NASSEG_OK;
}

// stats = out of nasseg_ce_fwd; gscale = device scalar upstream gradient (null = 1)
int NASSEG_FN(ce_bwd)(const act_t* logits, const void* target, int elem_size, const float* stats,
                  const float* gscale, int64_t P, int C, int ignore, act_t* dlogits, void* stream) {
  NASSEG_REQUIRE(P > 0 && C > 0, "ce_bwd: bad shape");
  hipStream_t s = (hipStream_t)stream;
  const int grid = ce_grid(P) * 2;
  NASSEG_REQUIRE(elem_size == 8 || elem_size == 1, "ce_bwd: elem_size %d not supported", elem_size);
  const size_t lds = (size_t)256 * (C | 1) * sizeof(float);
  int64_t tiles = (P + 255) / 256;
  if (tiles > 4096) tiles = 4096;
  const bool tiled = C <= kCeTileMaxC && (((uintptr_t)logits | (uintptr_t)dlogits) & 15) == 0;
  if (tiled && elem_size == 8)
    hipLaunchKernelGGL((ce_tile_kernel<int64_t, true>), dim3((unsigned)tiles), dim3(256), lds, s, logits,
                       (const int64_t*)target, P, C, ignore, nullptr, stats, gscale, dlogits);
  else if (tiled)
    hipLaunchKernelGGL((ce_tile_kernel<uint8_t, true>), dim3((unsigned)tiles), dim3(256), lds, s, logits,
                       (const uint8_t*)target, P, C, ignore, nullptr, stats, gscale, dlogits);
  else if (elem_size == 8)
    hipLaunchKernelGGL((ce_bwd_kernel<int64_t>), dim3(grid), dim3(256), 0, s, logits,
                       (const int64_t*)target, stats, gscale, P, C, ignore, dlogits);
  else
    hipLaunchKernelGGL((ce_bwd_kernel<uint8_t>), dim3(grid), dim3(256), 0, s, logits,
                       (const uint8_t*)target, stats, gscale, P, C, ignore, dlogits);
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
