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
#include "common.h"

namespace {

__device__ __forceinline__ float4 relu4(float4 v) {
  return make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
}

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
template <int K, int P, int E>
__global__ __launch_bounds__(256) void dw_fwd_strip(
    const float* __restrict__ x, const float* __restrict__ wt, float* __restrict__ y,
    const float* __restrict__ scale, const float* __restrict__ shift, int H, int W, int C4, int Ho,
    int Wo, int stride, int pad, int dil, int g, int nchunk, int relu_in, int act) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= Wo * C4) return;
  const int ox = idx / C4;
  const int c4 = idx - ox * C4;
  const int C = C4 * 4;
  const int b = blockIdx.z;
  const int r = blockIdx.y % g;
  const int chunk = blockIdx.y / g;
  const int oy0 = chunk * (P * g) + r;
  if (oy0 >= Ho) return;

  float4 w[K * K];
#pragma unroll
  for (int t = 0; t < K * K; ++t) w[t] = ld4(wt + (size_t)t * C + c4 * 4);

  int xoff[K];
  bool xok[K];
#pragma unroll
  for (int tx = 0; tx < K; ++tx) {
    int ix = ox * stride - pad + tx * dil;
    xok[tx] = (ix >= 0) && (ix < W);
    xoff[tx] = ix * C;
  }
  float4 acc[P];
#pragma unroll
  for (int j = 0; j < P; ++j) acc[j] = f4zero();

  const float* xb = x + (size_t)b * H * W * C + c4 * 4;
  const int iy0 = oy0 * stride - pad;
  constexpr int Q = (P - 1) * E + K;
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int iy = iy0 + q * dil;
    const bool yok = (iy >= 0) && (iy < H);
    const float* xr = xb + (size_t)iy * W * C;
    float4 v[K];
#pragma unroll
    for (int tx = 0; tx < K; ++tx) {
      v[tx] = (yok && xok[tx]) ? ld4(xr + xoff[tx]) : f4zero();
      if (relu_in) v[tx] = relu4(v[tx]);
    }
#pragma unroll
    for (int j = 0; j < P; ++j) {
      const int ty = q - j * E;
      if (ty >= 0 && ty < K) {
#pragma unroll
        for (int tx = 0; tx < K; ++tx) acc[j] = fma4(w[ty * K + tx], v[tx], acc[j]);
      }
    }
  }
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = f4zero();
  if (scale) sc = ld4(scale + c4 * 4);
  if (shift) sh = ld4(shift + c4 * 4);
#pragma unroll
  for (int j = 0; j < P; ++j) {
    const int oy = oy0 + j * g;
    if (oy < Ho) {
      float4 o = fma4(acc[j], sc, sh);
      o = act_apply4(o, act);
      st4(y + (((size_t)b * Ho + oy) * Wo + ox) * C + c4 * 4, o);
    }
  }
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
    const float* __restrict__ x, const float* __restrict__ wt, float* __restrict__ y,
    const float* __restrict__ scale, const float* __restrict__ shift, int B, int H, int W, int C4,
    int Ho, int Wo, int K, int stride, int pad, int dil, int transposed, int relu_in, int act) {
  const int C = C4 * 4;
  const size_t total = (size_t)B * Ho * Wo * C4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    size_t p = i / C4;
    const int ox = (int)(p % Wo);
    p /= Wo;
    const int oy = (int)(p % Ho);
    const int b = (int)(p / Ho);
    const float* xb = x + (size_t)b * H * W * C + c4 * 4;
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
        float4 v = ld4(xb + ((size_t)iy * W + ix) * C);
        if (relu_in) v = relu4(v);
        acc = fma4(ld4(wt + (size_t)(ty * K + tx) * C + c4 * 4), v, acc);
      }
    }
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = f4zero();
    if (scale) sc = ld4(scale + c4 * 4);
    if (shift) sh = ld4(shift + c4 * 4);
    float4 o = act_apply4(fma4(acc, sc, sh), act);
    st4(y + i * 4, o);
  }
}

// ---------------------------------------------------------------------------
// backward-weight strip kernel: per-thread K*K float4 accumulators over a set
// of row chunks, then a per-block reduction over lanes that share a channel
// group.  partial layout: [block][tap][C].  Deterministic (no atomics).
// ---------------------------------------------------------------------------
template <int K, int P, int E>
__global__ __launch_bounds__(256) void dw_wgrad_strip(
    const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ partial, int B,
    int H, int W, int C4, int Ho, int Wo, int stride, int pad, int dil, int g, int nchunk,
    int relu_in) {
  __shared__ float4 red[256];
  const int tid = threadIdx.x;
  const int base = blockIdx.x * 256;
  const int idx = base + tid;
  const bool live = idx < Wo * C4;
  const int ox = live ? idx / C4 : 0;
  const int c4 = live ? idx - ox * C4 : 0;
  const int C = C4 * 4;

  float4 acc[K * K];
#pragma unroll
  for (int t = 0; t < K * K; ++t) acc[t] = f4zero();

  int xoff[K];
  bool xok[K];
#pragma unroll
  for (int tx = 0; tx < K; ++tx) {
    int ix = ox * stride - pad + tx * dil;
    xok[tx] = live && (ix >= 0) && (ix < W);
    xoff[tx] = ix * C;
  }
  constexpr int Q = (P - 1) * E + K;
  const int nwork = B * nchunk * g;
  for (int wk = blockIdx.y; wk < nwork; wk += gridDim.y) {
    const int r = wk % g;
    int t2 = wk / g;
    const int chunk = t2 % nchunk;
    const int b = t2 / nchunk;
    const int oy0 = chunk * (P * g) + r;
    if (oy0 >= Ho || !live) continue;
    float4 d[P];
#pragma unroll
    for (int j = 0; j < P; ++j) {
      const int oy = oy0 + j * g;
      d[j] = (oy < Ho) ? ld4(dy + (((size_t)b * Ho + oy) * Wo + ox) * C + c4 * 4) : f4zero();
    }
    const float* xb = x + (size_t)b * H * W * C + c4 * 4;
    const int iy0 = oy0 * stride - pad;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int iy = iy0 + q * dil;
      const bool yok = (iy >= 0) && (iy < H);
      const float* xr = xb + (size_t)iy * W * C;
      float4 v[K];
#pragma unroll
      for (int tx = 0; tx < K; ++tx) {
        v[tx] = (yok && xok[tx]) ? ld4(xr + xoff[tx]) : f4zero();
        if (relu_in) v[tx] = relu4(v[tx]);
      }
#pragma unroll
      for (int j = 0; j < P; ++j) {
        const int ty = q - j * E;
        if (ty >= 0 && ty < K) {
#pragma unroll
          for (int tx = 0; tx < K; ++tx) acc[ty * K + tx] = fma4(d[j], v[tx], acc[ty * K + tx]);
        }
      }
    }
  }
  // block reduction: thread t (< min(C4,256)) owns channel group (base+t)%C4
  float* pout = partial + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (size_t)(K * K) * C;
  const int nown = C4 < 256 ? C4 : 256;
#pragma unroll
  for (int t = 0; t < K * K; ++t) {
    const float4 a = acc[t];
    __syncthreads();
    red[tid] = a;
    __syncthreads();
    if (tid < nown) {
      float4 s = f4zero();
      for (int u = tid; u < 256; u += C4) s = add4(s, red[u]);
      const int cc = (base + tid) % C4;
      st4(pout + (size_t)t * C + cc * 4, s);
    }
  }
}

// generic backward-weight (any K): same block reduction, one tap at a time.
__global__ __launch_bounds__(256) void dw_wgrad_generic(
    const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ partial, int B,
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
          float4 v = ld4(x + (((size_t)b * H + iy) * W + ix) * C + c4 * 4);
          if (relu_in) v = relu4(v);
          a = fma4(ld4(dy + (((size_t)b * Ho + oy) * Wo + ox) * C + c4 * 4), v, a);
        }
      }
      __syncthreads();
      red[tid] = a;
      __syncthreads();
      if (tid < nown) {
        float4 s = f4zero();
        for (int u = tid; u < 256; u += C4) s = add4(s, red[u]);
        const int cc = (base + tid) % C4;
        st4(pout + (size_t)(ty * K + tx) * C + cc * 4, s);
      }
    }
}

// sum partial[blk][tap][C] over blk -> dw[(c*KK + tap)]  ((C,1,K,K) layout)
__global__ __launch_bounds__(256) void dw_wgrad_finalize(const float* __restrict__ partial,
                                                         float* __restrict__ dw, int nblk, int KK,
                                                         int C) {
  __shared__ double red[NASSEG_RP_SLICES][NASSEG_RP_ELEMS + 1];
  const int64_t per = (int64_t)KK * C;
  const int64_t i = (int64_t)blockIdx.x * NASSEG_RP_ELEMS + (threadIdx.x & 15);  // tap*C + c
  const bool valid = i < per;
  const double s = reduce_partials16(partial, nblk, per, i, valid, red);
  if (valid && (threadIdx.x >> 4) == 0) {
    const int t = (int)(i / C), c = (int)(i - (int64_t)t * C);
    dw[c * KK + t] = (float)s;
  }
}

struct StripCfg {
  int g, e;
};
inline StripCfg strip_cfg(int stride, int dil) {
  // outputs of one strip are g rows apart; consecutive outputs are e dilated
  // input-row steps apart (see file header)
  int a = dil, b = stride;
  while (b) {
    int t = a % b;
    a = b;
    b = t;
  }
  StripCfg c;
  c.g = dil / a;
  c.e = stride / a;
  return c;
}

}  // namespace

extern "C" {

// wt[tap][C] <- w (C,1,K,K); flip != 0 rotates the kernel by 180 degrees.
int nasseg_dw_pack_weight(const float* w, float* wt, int C, int K, int flip, void* stream) {
  NASSEG_REQUIRE(C > 0 && K > 0, "dw_pack_weight: bad shape C=%d K=%d", C, K);
  int n = C * K * K;
  hipLaunchKernelGGL(dw_pack_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, w, wt,
                     C, K * K, flip);
  NASSEG_LAUNCH_CHECK("dw_pack_weight");
  return NASSEG_OK;
}

// y = act(scale * dwconv(relu_in ? relu(x) : x) + shift); scale/shift may be null.
// transposed != 0 computes the backward-data form (x = grad wrt output with
// dims (H,W), y = grad wrt input with dims (Ho,Wo), un-flipped weights).
int nasseg_dwconv(const float* x, const float* wt, float* y, const float* scale,
                  const float* shift, int B, int H, int W, int C, int Ho, int Wo, int K, int stride,
                  int pad, int dil, int transposed, int relu_in, int act, void* stream) {
  NASSEG_REQUIRE(C % 4 == 0, "dwconv: C=%d must be a multiple of 4", C);
  NASSEG_REQUIRE(B > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && K > 0 && stride > 0 && dil > 0,
                 "dwconv: bad geometry");
  hipStream_t s = (hipStream_t)stream;
  const int C4 = C / 4;
  StripCfg sc = strip_cfg(stride, dil);
  const bool strip_ok = !transposed && (K == 3 || K == 5) && (sc.e == 1 || sc.e == 2) && B <= 65535;
  if (strip_ok) {
    constexpr int P = 4;
    const int nchunk = cdiv(Ho, P * sc.g);
    dim3 grid(cdiv(Wo * C4, 256), nchunk * sc.g, B);
    NASSEG_REQUIRE(grid.y <= 65535, "dwconv: too many row chunks");
#define LAUNCH_FWD(KK, EE)                                                                       \
  hipLaunchKernelGGL((dw_fwd_strip<KK, P, EE>), grid, dim3(256), 0, s, x, wt, y, scale, shift, H, \
                     W, C4, Ho, Wo, stride, pad, dil, sc.g, nchunk, relu_in, act)
    if (K == 3 && sc.e == 1) LAUNCH_FWD(3, 1);
    else if (K == 3 && sc.e == 2) LAUNCH_FWD(3, 2);
    else if (K == 5 && sc.e == 1) LAUNCH_FWD(5, 1);
    else LAUNCH_FWD(5, 2);
#undef LAUNCH_FWD
    NASSEG_LAUNCH_CHECK("dw_fwd_strip");
    return NASSEG_OK;
  }
  size_t total = (size_t)B * Ho * Wo * C4;
  int nb = (int)((total + 255) / 256 < 65536 * 4 ? (total + 255) / 256 : 65536 * 4);
  hipLaunchKernelGGL(dw_generic, dim3(nb), dim3(256), 0, s, x, wt, y, scale, shift, B, H, W, C4,
                     Ho, Wo, K, stride, pad, dil, transposed, relu_in, act);
  NASSEG_LAUNCH_CHECK("dw_generic");
  return NASSEG_OK;
}

// workspace (floats) needed by nasseg_dwconv_wgrad
int64_t nasseg_dwconv_wgrad_workspace(int B, int C, int Ho, int Wo, int K) {
  const int C4 = C / 4;
  int64_t gx = cdiv(Wo * C4, 256);
  int64_t gy = 1024 / gx;
  if (gy < 1) gy = 1;
  if (gy > (int64_t)B * Ho) gy = (int64_t)B * Ho;
  return gx * gy * (int64_t)K * K * C;
}

// dw (C,1,K,K) = sum over pixels of dy * x_tap; ws must hold
// nasseg_dwconv_wgrad_workspace() floats.
int nasseg_dwconv_wgrad(const float* x, const float* dy, float* dw, float* ws, int B, int H, int W,
                        int C, int Ho, int Wo, int K, int stride, int pad, int dil, int relu_in,
                        void* stream) {
  NASSEG_REQUIRE(C % 4 == 0, "dwconv_wgrad: C=%d must be a multiple of 4", C);
  hipStream_t s = (hipStream_t)stream;
  const int C4 = C / 4;
  StripCfg sc = strip_cfg(stride, dil);
  const int gx = cdiv(Wo * C4, 256);
  int gy = 1024 / gx;
  if (gy < 1) gy = 1;
  if ((int64_t)gy > (int64_t)B * Ho) gy = B * Ho;
  const bool strip_ok = (K == 3 || K == 5) && (sc.e == 1 || sc.e == 2);
  if (strip_ok) {
    constexpr int P = 4;
    const int nchunk = cdiv(Ho, P * sc.g);
    dim3 grid(gx, gy, 1);
#define LAUNCH_WG(KK, EE)                                                                        \
  hipLaunchKernelGGL((dw_wgrad_strip<KK, P, EE>), grid, dim3(256), 0, s, x, dy, ws, B, H, W, C4, \
                     Ho, Wo, stride, pad, dil, sc.g, nchunk, relu_in)
    if (K == 3 && sc.e == 1) LAUNCH_WG(3, 1);
    else if (K == 3 && sc.e == 2) LAUNCH_WG(3, 2);
    else if (K == 5 && sc.e == 1) LAUNCH_WG(5, 1);
    else LAUNCH_WG(5, 2);
#undef LAUNCH_WG
    NASSEG_LAUNCH_CHECK("dw_wgrad_strip");
  } else {
    hipLaunchKernelGGL(dw_wgrad_generic, dim3(gx, gy, 1), dim3(256), 0, s, x, dy, ws, B, H, W, C4,
                       Ho, Wo, K, stride, pad, dil, relu_in);
    NASSEG_LAUNCH_CHECK("dw_wgrad_generic");
  }
  hipLaunchKernelGGL(dw_wgrad_finalize, dim3(cdiv(K * K * C, NASSEG_RP_ELEMS)), dim3(256), 0, s, ws,
                     dw, gx * gy, K * K, C);
  NASSEG_LAUNCH_CHECK("dw_wgrad_finalize");
  return NASSEG_OK;
}

}  // extern "C"
