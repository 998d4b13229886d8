// Mean-IoU reward path: argmax over class logits (optionally fused with the
// bilinear up-sampling to label resolution) + confusion-matrix histogram on
// the GPU, and the IoU / accuracy arithmetic on the host.
//
// Reference: validate() (src/engine/inference.py:55-66) copies full-resolution
// logits to the host, takes numpy argmax(axis=1).astype(uint8), drops pixels
// with gt >= num_classes and calls the Cython fast_cm
// (src/helpers/miou_utils.pyx:7-30: cm[gt, pred] += 1, int64); compute_iu /
// compute_ius_accs (miou_utils.pyx:32-90) hold pi/gi/ii in C `unsigned int`
// and default absent classes to 2.0.
//
// Histogram: per-workgroup LDS-privatised int32 bins (n*n <= 4096), flushed
// with 64-bit global atomics - integer adds commute, so the result is exact
// and reproducible.  Larger n falls back to global atomics only.
#include <math.h>
#include <string.h>

#include "common.h"

namespace {

#define CM_LDS_BINS 4096

__global__ __launch_bounds__(256) void cm_u8_kernel(const uint8_t* __restrict__ preds,
                                                    const uint8_t* __restrict__ gt, int64_t P,
                                                    int n, unsigned long long* __restrict__ cm) {
  __shared__ int bins[CM_LDS_BINS];
  const int nn = n * n;
  const bool use_lds = nn <= CM_LDS_BINS;
  if (use_lds) {
    for (int i = threadIdx.x; i < nn; i += 256) bins[i] = 0;
    __syncthreads();
  }
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < P; p += (int64_t)gridDim.x * 256) {
    const int a = gt[p];
    const int q = preds[p];
    if (a >= n || q >= n) continue;  // caller contract: gt < n (inference.py:65); never index out of the matrix
    if (use_lds)
      atomicAdd(&bins[a * n + q], 1);
    else
      atomicAdd(&cm[a * n + q], 1ULL);
  }
  if (use_lds) {
    __syncthreads();
    for (int i = threadIdx.x; i < nn; i += 256)
      if (bins[i]) atomicAdd(&cm[i], (unsigned long long)bins[i]);
  }
}

struct Lin {
  int i0, i1;
  float l0, l1;
};
__device__ __forceinline__ Lin lin_coeff(int dst, float scale, int in_size, int out_size) {
  Lin r;
  if (in_size == out_size) {
    r.i0 = r.i1 = dst;
    r.l0 = 1.f;
    r.l1 = 0.f;
    return r;
  }
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  r.i0 = (int)src;
  if (r.i0 > in_size - 1) r.i0 = in_size - 1;
  r.i1 = r.i0 + ((r.i0 < in_size - 1) ? 1 : 0);
  float l1 = fminf(fmaxf(src - (float)r.i0, 0.f), 1.f);
  r.l1 = l1;
  r.l0 = 1.f - l1;
  return r;
}

// logits [B][h][w][C] -> bilinear to (H,W) -> argmax (lowest index wins ties,
// as numpy) -> uint8 prediction; pixels with gt < n are histogrammed.
__global__ __launch_bounds__(256) void argmax_cm_kernel(
    const float* __restrict__ logits, const uint8_t* __restrict__ gt, uint8_t* __restrict__ preds,
    int B, int h, int w, int C, int H, int W, float sh, float sw, int n,
    unsigned long long* __restrict__ cm) {
  __shared__ int bins[CM_LDS_BINS];
  const int nn = n * n;
  const bool use_lds = cm && nn <= CM_LDS_BINS;
  if (use_lds) {
    for (int i = threadIdx.x; i < nn; i += 256) bins[i] = 0;
    __syncthreads();
  }
  const int64_t P = (int64_t)B * H * W;
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < P; p += (int64_t)gridDim.x * 256) {
    const int ox = (int)(p % W);
    const int64_t t = p / W;
    const int oy = (int)(t % H);
    const int b = (int)(t / H);
    const Lin ly = lin_coeff(oy, sh, h, H);
    const Lin lx = lin_coeff(ox, sw, w, W);
    const float* lb = logits + (int64_t)b * h * w * C;
    const float* p00 = lb + ((int64_t)ly.i0 * w + lx.i0) * C;
    const float* p01 = lb + ((int64_t)ly.i0 * w + lx.i1) * C;
    const float* p10 = lb + ((int64_t)ly.i1 * w + lx.i0) * C;
    const float* p11 = lb + ((int64_t)ly.i1 * w + lx.i1) * C;
    float best = -INFINITY;
    int arg = 0;
    for (int c = 0; c < C; ++c) {
      // explicit rounding of every product / sum: no fma contraction, so the
      // value equals the fp32 up-sampling the oracle performs before its argmax
      const float top = __fadd_rn(__fmul_rn(lx.l0, p00[c]), __fmul_rn(lx.l1, p01[c]));
      const float bot = __fadd_rn(__fmul_rn(lx.l0, p10[c]), __fmul_rn(lx.l1, p11[c]));
      const float v = __fadd_rn(__fmul_rn(ly.l0, top), __fmul_rn(ly.l1, bot));
      if (c == 0 || v > best) {
        best = v;
        arg = c;
      }
    }
    const uint8_t q = (uint8_t)arg;
    if (preds) preds[p] = q;
    if (cm) {
      const int a = gt[p];
      if (a < n && (int)q < n) {
        if (use_lds)
          atomicAdd(&bins[a * n + q], 1);
        else
          atomicAdd(&cm[a * n + q], 1ULL);
      }
    }
  }
  if (use_lds) {
    __syncthreads();
    for (int i = threadIdx.x; i < nn; i += 256)
      if (bins[i]) atomicAdd(&cm[i], (unsigned long long)bins[i]);
  }
}

inline int cm_grid(int64_t P) {
  int64_t b = (P + 256 * 8 - 1) / (256 * 8);
  if (b > 1024) b = 1024;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" {

// cm[gt[i]*n + preds[i]] += 1 for every i with gt[i] < n; cm is int64 [n][n] on
// the device and is ACCUMULATED into (zero it first for a fresh matrix).
int nasseg_fast_cm(const uint8_t* preds, const uint8_t* gt, int64_t P, int n, int64_t* cm,
                   void* stream) {
  NASSEG_REQUIRE(n > 0 && n <= 256, "fast_cm: n_classes=%d out of range", n);
  if (P <= 0) return NASSEG_OK;
  hipLaunchKernelGGL(cm_u8_kernel, dim3(cm_grid(P)), dim3(256), 0, (hipStream_t)stream, preds, gt,
                     P, n, (unsigned long long*)cm);
  NASSEG_LAUNCH_CHECK("fast_cm");
  return NASSEG_OK;
}

// preds (uint8 [B][H][W], may be null) = argmax_c bilinear(logits)[..., c];
// cm (int64 [n][n], may be null) += histogram of (gt, pred) over pixels with gt < n.
int nasseg_argmax_cm(const float* logits, const uint8_t* gt, uint8_t* preds, int B, int h, int w,
                     int C, int H, int W, int n, int64_t* cm, void* stream) {
  NASSEG_REQUIRE(B > 0 && h > 0 && w > 0 && H > 0 && W > 0, "argmax_cm: bad shape");
  NASSEG_REQUIRE(C > 0 && C <= 256, "argmax_cm: C=%d does not fit uint8 predictions", C);
  NASSEG_REQUIRE(!cm || (gt && n > 0 && n <= 256), "argmax_cm: bad n_classes / gt");
  const float sh = (float)h / (float)H, sw = (float)w / (float)W;
  hipLaunchKernelGGL(argmax_cm_kernel, dim3(cm_grid((int64_t)B * H * W)), dim3(256), 0,
                     (hipStream_t)stream, logits, gt, preds, B, h, w, C, H, W, sh, sw, n,
                     (unsigned long long*)cm);
  NASSEG_LAUNCH_CHECK("argmax_cm");
  return NASSEG_OK;
}

// Host arithmetic of compute_iu / compute_ius_accs (miou_utils.pyx:32-90).
// pi, gi, ii and the denominator live in 32-bit unsigned ints exactly as the
// Cython `cdef unsigned int` locals do; a column/row sum or diagonal above
// 2^32-1 is reported as an error (Cython raises OverflowError there).
// iu / accs default to 2.0, n_pixels[i] = gi.  accs / n_pixels may be null.
int nasseg_compute_ius_accs(const int64_t* cm, int n, double* iu, int64_t* n_pixels, double* accs) {
  NASSEG_REQUIRE(cm && iu && n > 0, "compute_ius_accs: bad arguments");
  for (int i = 0; i < n; ++i) {
    int64_t pi64 = 0, gi64 = 0;
    for (int j = 0; j < n; ++j) {
      pi64 += cm[(int64_t)j * n + i];
      gi64 += cm[(int64_t)i * n + j];
    }
    const int64_t ii64 = cm[(int64_t)i * n + i];
    if (pi64 < 0 || gi64 < 0 || ii64 < 0 || pi64 > 4294967295LL || gi64 > 4294967295LL ||
        ii64 > 4294967295LL)
      return nasseg_fail(NASSEG_ERR_ARG,
                         "compute_ius_accs: class %d count does not fit unsigned int", i);
    const unsigned int pi = (unsigned int)pi64, gi = (unsigned int)gi64, ii = (unsigned int)ii64;
    const unsigned int denom = pi + gi - ii;  // wraps like the C original
    iu[i] = 2.0;
    if (denom > 0) iu[i] = (double)ii / (double)denom;
    if (accs) {
      accs[i] = 2.0;
      if (gi > 0) accs[i] = (double)ii / (double)gi;
    }
    if (n_pixels) n_pixels[i] = (int64_t)gi;
  }
  return NASSEG_OK;
}

}  // extern "C"
