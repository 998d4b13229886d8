// Dense convolution (1x1 pointwise and k x k, any stride / dilation) as an
// implicit GEMM on the fp32 matrix cores (v_mfma_f32_16x16x4_f32), NHWC, gfx950.
//
// Reference call sites: conv1x1 / conv3x3 / conv_bn / conv_bn_relu
// (src/nn/layer_factory.py:7-24,94-122), the pointwise stage of SepConv /
// DilConv / InvertedResidual / Pool / Adapt / ConcatReduce (:125-382) and the
// classifier heads (src/nn/micro_decoders.py:210-227,360-363).
//
// fp32 in / fp32 accumulate MFMA is bit-equivalent to an fmaf chain, which is
// what keeps logits within 1e-4 of the reference.  The op is HBM-bound for
// every channel count on the path, so the kernels avoid LDS staging entirely:
//
//  * forward / backward-data: D[n][pixel] = sum_k W[n][k] * X[pixel][k].  Each
//    lane loads one float4 along the (contiguous) reduction axis; the four
//    components feed four consecutive MFMAs (the "k" label of an MFMA slot is
//    arbitrary as long as A and B agree).  The accumulator then holds four
//    consecutive output channels of one pixel per lane -> float4 stores.
//  * backward-weight: dW[n][k] = sum_pixel dY[pixel][n] * X[pixel][k].  The
//    reduction axis (pixels) is the slow axis of both operands, so lanes load
//    float4s along n and along k and the MFMA rows/cols are a permutation
//    (row i <-> n = n0 + 4*i + comp).  Per-wave partials are reduced through
//    LDS, per-block partials by a second deterministic pass.
//
// Packed weight layouts (produced by nasseg_conv_pack_weight):
//   forward       : wp[tap][N][K]           from OIHW (N,K,kh,kw)
//   backward-data : wp[tap][K][N]           (roles of N and K swapped)
#include "common.h"

namespace {

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

struct ConvGeom {
  int B, Hs, Ws;  // source (tensor being read) dims
  int Ho, Wo;     // destination dims
  int kh, kw, stride, pad, dil, transposed;
};

// source pixel offset (in pixels) for destination pixel (b,oy,ox) and tap
// (ty,tx); returns -1 when the tap falls outside / on a stride hole.
__device__ __forceinline__ int64_t src_pixel(const ConvGeom& g, int b, int oy, int ox, int ty,
                                             int tx) {
  int iy, ix;
  if (!g.transposed) {
    iy = oy * g.stride - g.pad + ty * g.dil;
    ix = ox * g.stride - g.pad + tx * g.dil;
    if (iy < 0 || iy >= g.Hs || ix < 0 || ix >= g.Ws) return -1;
  } else {
    int ny = oy + g.pad - ty * g.dil;
    int nx = ox + g.pad - tx * g.dil;
    if (ny < 0 || nx < 0) return -1;
    if (g.stride > 1) {
      if ((ny % g.stride) || (nx % g.stride)) return -1;
      iy = ny / g.stride;
      ix = nx / g.stride;
    } else {
      iy = ny;
      ix = nx;
    }
    if (iy >= g.Hs || ix >= g.Ws) return -1;
  }
  return ((int64_t)b * g.Hs + iy) * g.Ws + ix;
}

template <bool VEC>
__device__ __forceinline__ float4 load_k4(const float* p, int k, int K) {
  // 4 values along the reduction axis starting at k, zero beyond K
  if (VEC) {
    return (k < K) ? ld4(p + k) : f4zero();
  } else {
    float4 v;
    v.x = (k + 0 < K) ? p[k + 0] : 0.f;
    v.y = (k + 1 < K) ? p[k + 1] : 0.f;
    v.z = (k + 2 < K) ? p[k + 2] : 0.f;
    v.w = (k + 3 < K) ? p[k + 3] : 0.f;
    return v;
  }
}

// ---------------------------------------------------------------------------
// forward / backward-data implicit GEMM
// ---------------------------------------------------------------------------
struct FwdArgs {
  const float* x;
  int ldx;
  const float* w;  // packed [tap][N][K]
  float* y;
  int ldy;
  const float* in_scale;
  const float* in_shift;
  int in_act;
  const float* out_scale;
  const float* out_shift;
  int out_act;
  const float* res;
  int ldres;
  int K, N;
  ConvGeom g;
};

template <int MT, int NT, bool VECK>
__global__ __launch_bounds__(256) void conv_fwd_kernel(FwdArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int j = lane & 15;   // pixel within subtile (B operand col) / n within tile (A operand row)
  const int kg = lane >> 4;  // k group
  const int64_t Mtot = (int64_t)a.g.B * a.g.Ho * a.g.Wo;
  const int64_t m_base = ((int64_t)blockIdx.x * 4 + wave) * (16 * MT);
  if (m_base >= Mtot) return;
  const int n_base = blockIdx.y * (16 * NT);

  int pb[MT], py[MT], px[MT];
  bool pok[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    int64_t m = m_base + mt * 16 + j;
    pok[mt] = m < Mtot;
    if (!pok[mt]) m = 0;
    px[mt] = (int)(m % a.g.Wo);
    int64_t t = m / a.g.Wo;
    py[mt] = (int)(t % a.g.Ho);
    pb[mt] = (int)(t / a.g.Ho);
  }
  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int ntaps = a.g.kh * a.g.kw;
  for (int tap = 0; tap < ntaps; ++tap) {
    const int ty = tap / a.g.kw, tx = tap - ty * a.g.kw;
    const float* xp[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      int64_t sp = pok[mt] ? src_pixel(a.g, pb[mt], py[mt], px[mt], ty, tx) : -1;
      xp[mt] = sp >= 0 ? a.x + sp * a.ldx : nullptr;
    }
    const float* wp[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      int n = n_base + nt * 16 + j;
      wp[nt] = n < a.N ? a.w + ((int64_t)tap * a.N + n) * a.K : nullptr;
    }
    for (int k0 = 0; k0 < a.K; k0 += 16) {
      const int k = k0 + kg * 4;
      float4 bv[MT], av[NT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        bv[mt] = xp[mt] ? load_k4<VECK>(xp[mt], k, a.K) : f4zero();
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) av[nt] = wp[nt] ? load_k4<VECK>(wp[nt], k, a.K) : f4zero();
      if (a.in_scale || a.in_shift || a.in_act) {
        float4 s = make_float4(1.f, 1.f, 1.f, 1.f), h = f4zero();
        if (a.in_scale) s = load_k4<VECK>(a.in_scale, k, a.K);
        if (a.in_shift) h = load_k4<VECK>(a.in_shift, k, a.K);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          if (xp[mt] && k < a.K) {
            float4 v = act_apply4(fma4(bv[mt], s, h), a.in_act);
            if (!VECK) {  // keep the zero padding beyond K
              if (k + 1 >= a.K) v.y = 0.f;
              if (k + 2 >= a.K) v.z = 0.f;
              if (k + 3 >= a.K) v.w = 0.f;
            }
            bv[mt] = v;
          }
        }
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          acc[mt][nt] = mfma16(av[nt].x, bv[mt].x, acc[mt][nt]);
          acc[mt][nt] = mfma16(av[nt].y, bv[mt].y, acc[mt][nt]);
          acc[mt][nt] = mfma16(av[nt].z, bv[mt].z, acc[mt][nt]);
          acc[mt][nt] = mfma16(av[nt].w, bv[mt].w, acc[mt][nt]);
        }
    }
  }
  // epilogue: lane holds pixel j of each subtile, channels n0 + 4*kg + {0..3}
  const bool vec_out = ((a.N & 3) == 0) && ((a.ldy & 3) == 0) && (!a.res || (a.ldres & 3) == 0);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    if (!pok[mt]) continue;
    const int64_t m = m_base + mt * 16 + j;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = n_base + nt * 16 + kg * 4;
      if (n >= a.N) continue;
      f32x4 c = acc[mt][nt];
      float o[4] = {c[0], c[1], c[2], c[3]};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (n + r < a.N) {
          float v = o[r];
          if (a.out_scale) v *= a.out_scale[n + r];
          if (a.out_shift) v += a.out_shift[n + r];
          v = act_apply(v, a.out_act);
          if (a.res) v += a.res[m * a.ldres + n + r];
          o[r] = v;
        }
      }
      float* yp = a.y + m * a.ldy + n;
      if (vec_out) {
        st4(yp, make_float4(o[0], o[1], o[2], o[3]));
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n + r < a.N) yp[r] = o[r];
      }
    }
  }
}

// ---------------------------------------------------------------------------
// backward-weight
// ---------------------------------------------------------------------------
struct WgArgs {
  const float* x;   // forward input  [B][Hs][Ws][ldx], K channels
  int ldx;
  const float* dy;  // grad of forward output [B][Ho][Wo][lddy], N channels
  int lddy;
  float* partial;   // [slab][tap][N][K]
  const float* in_scale;
  const float* in_shift;
  int in_act;
  int K, N;
  int kchunks;      // number of 64-wide k chunks
  int64_t pix_per_block;
  ConvGeom g;       // non-transposed forward geometry
};

template <bool VECN, bool VECK>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgArgs a) {
  __shared__ float red[3][64][65];  // waves 1..3 park their 64 accumulators here
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int li = lane & 15;  // row/col index inside a 16-wide MFMA tile
  const int pk = lane >> 4;  // pixel slot 0..3
  const int tap = blockIdx.z;
  const int ty = tap / a.g.kw, tx = tap - ty * a.g.kw;
  const int nchunk = blockIdx.y / a.kchunks;
  const int kchunk = blockIdx.y - nchunk * a.kchunks;
  const int n0 = nchunk * 64 + li * 4;
  const int k0 = kchunk * 64 + li * 4;
  const int64_t Mtot = (int64_t)a.g.B * a.g.Ho * a.g.Wo;
  const int64_t p_begin = (int64_t)blockIdx.x * a.pix_per_block;
  int64_t p_end = p_begin + a.pix_per_block;
  if (p_end > Mtot) p_end = Mtot;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) acc[i][jj] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float4 ps = make_float4(1.f, 1.f, 1.f, 1.f), ph = f4zero();
  const bool prologue = a.in_scale || a.in_shift || a.in_act;
  if (a.in_scale) ps = load_k4<VECK>(a.in_scale, k0, a.K);
  if (a.in_shift) ph = load_k4<VECK>(a.in_shift, k0, a.K);

  // 4 waves interleave groups of 4 pixels; two groups per iteration keep two
  // float4 pairs in flight per lane.  Loop bounds are wave-uniform: lanes whose
  // pixel is out of range feed zeros so that all 64 lanes reach the MFMAs.
  for (int64_t g0 = p_begin + wave * 4; g0 < p_end; g0 += 32) {
    float4 dv[2], xv[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      dv[u] = f4zero();
      xv[u] = f4zero();
      const int64_t p = g0 + u * 16 + pk;
      if (p < p_end) {
        const int ox = (int)(p % a.g.Wo);
        const int64_t t = p / a.g.Wo;
        const int oy = (int)(t % a.g.Ho);
        const int b = (int)(t / a.g.Ho);
        const int64_t sp = src_pixel(a.g, b, oy, ox, ty, tx);
        if (sp >= 0) {
          dv[u] = load_k4<VECN>(a.dy + p * a.lddy, n0, a.N);
          xv[u] = load_k4<VECK>(a.x + sp * a.ldx, k0, a.K);
          if (prologue && k0 < a.K) {
            float4 v = act_apply4(fma4(xv[u], ps, ph), a.in_act);
            if (k0 + 1 >= a.K) v.y = 0.f;
            if (k0 + 2 >= a.K) v.z = 0.f;
            if (k0 + 3 >= a.K) v.w = 0.f;
            xv[u] = v;
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const float d[4] = {dv[u].x, dv[u].y, dv[u].z, dv[u].w};
      const float xx[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
#pragma unroll
      for (int ca = 0; ca < 4; ++ca)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) acc[ca][cb] = mfma16(d[ca], xx[cb], acc[ca][cb]);
    }
  }
  // cross-wave reduction through LDS
  if (wave > 0) {
#pragma unroll
    for (int ca = 0; ca < 4; ++ca)
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave - 1][(ca * 4 + cb) * 4 + r][lane] = acc[ca][cb][r];
  }
  __syncthreads();
  if (wave == 0) {
    float* pout = a.partial + (((int64_t)blockIdx.x * gridDim.z + tap) * a.N) * a.K;
#pragma unroll
    for (int ca = 0; ca < 4; ++ca)
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int e = (ca * 4 + cb) * 4 + r;
          float v = acc[ca][cb][r] + red[0][e][lane] + red[1][e][lane] + red[2][e][lane];
          // D row i = 4*pk + r  <-> n ; D col = li <-> k
          const int n = nchunk * 64 + 4 * (4 * pk + r) + ca;
          const int k = kchunk * 64 + 4 * li + cb;
          if (n < a.N && k < a.K) pout[(int64_t)n * a.K + k] = v;
        }
  }
}

// dw (N,K,kh,kw) = sum over slabs of partial[slab][tap][N][K]
__global__ __launch_bounds__(256) void conv_wgrad_finalize(const float* __restrict__ partial,
                                                           float* __restrict__ dw, int nslab,
                                                           int ntaps, int N, int K) {
  __shared__ double red[NASSEG_RP_SLICES][NASSEG_RP_ELEMS + 1];
  const int64_t per = (int64_t)ntaps * N * K;
  const int64_t i = (int64_t)blockIdx.x * NASSEG_RP_ELEMS + (threadIdx.x & 15);  // (tap*N + n)*K + k
  const bool valid = i < per;
  const double s = reduce_partials16(partial, nslab, per, i, valid, red);
  if (valid && (threadIdx.x >> 4) == 0) {
    const int k = (int)(i % K);
    const int64_t t = i / K;
    const int n = (int)(t % N);
    const int tap = (int)(t / N);
    dw[((int64_t)n * K + k) * ntaps + tap] = (float)s;
  }
}

// OIHW (N,K,kh,kw) -> [tap][N][K]  (mode 0)  or  [tap][K][N] (mode 1)
__global__ void conv_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, int N, int K,
                                 int ntaps, int mode) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)ntaps * N * K;
  if (i >= total) return;
  int tap, n, k;
  if (mode == 0) {
    k = (int)(i % K);
    int64_t t = i / K;
    n = (int)(t % N);
    tap = (int)(t / N);
  } else {
    n = (int)(i % N);
    int64_t t = i / N;
    k = (int)(t % K);
    tap = (int)(t / K);
  }
  wp[i] = w[((int64_t)n * K + k) * ntaps + tap];
}

template <int MT, int NT>
int launch_fwd(const FwdArgs& a, hipStream_t s) {
  const int64_t Mtot = (int64_t)a.g.B * a.g.Ho * a.g.Wo;
  const int64_t gx = cdiv64(Mtot, 64 * MT);
  const int gy = cdiv(a.N, 16 * NT);
  NASSEG_REQUIRE(gx < 2147483647LL, "conv: too many pixels");
  dim3 grid((unsigned)gx, gy, 1);
  const bool veck = ((a.K & 3) == 0) && ((a.ldx & 3) == 0);
  if (veck)
    hipLaunchKernelGGL((conv_fwd_kernel<MT, NT, true>), grid, dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((conv_fwd_kernel<MT, NT, false>), grid, dim3(256), 0, s, a);
  NASSEG_LAUNCH_CHECK("conv_fwd_kernel");
  return NASSEG_OK;
}

}  // namespace

extern "C" {

// mode 0: forward layout [tap][N][K]; mode 1: backward-data layout [tap][K][N]
int nasseg_conv_pack_weight(const float* w, float* wp, int N, int K, int kh, int kw, int mode,
                            void* stream) {
  NASSEG_REQUIRE(N > 0 && K > 0 && kh > 0 && kw > 0, "conv_pack_weight: bad shape");
  const int64_t total = (int64_t)N * K * kh * kw;
  hipLaunchKernelGGL(conv_pack_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, w, wp, N, K, kh * kw, mode);
  NASSEG_LAUNCH_CHECK("conv_pack_weight");
  return NASSEG_OK;
}

// y[dst pixel][n] = out_act(out_scale[n] * sum_{tap,k} wp[tap][n][k] *
//                   in_act(in_scale[k] * x[src pixel(tap)][k] + in_shift[k]) + out_shift[n])
//                   (+ res[dst pixel][n])
// transposed == 0: (Hs,Ws) input dims, (Ho,Wo) output dims of a forward conv.
// transposed != 0: backward-data; x is the output gradient with dims (Hs,Ws),
//   y the input gradient with dims (Ho,Wo), wp packed with mode 1, K = forward
//   N, N = forward K, stride/pad/dil those of the forward conv.
int nasseg_conv_fwd(const float* x, int ldx, const float* wp, float* y, int ldy,
                    const float* in_scale, const float* in_shift, int in_act,
                    const float* out_scale, const float* out_shift, int out_act, const float* res,
                    int ldres, int B, int Hs, int Ws, int K, int Ho, int Wo, int N, int kh, int kw,
                    int stride, int pad, int dil, int transposed, void* stream) {
  NASSEG_REQUIRE(B > 0 && Hs > 0 && Ws > 0 && Ho > 0 && Wo > 0, "conv_fwd: bad geometry");
  NASSEG_REQUIRE(K > 0 && N > 0 && ldx >= K && ldy >= N, "conv_fwd: bad channels K=%d N=%d", K, N);
  FwdArgs a;
  a.x = x; a.ldx = ldx; a.w = wp; a.y = y; a.ldy = ldy;
  a.in_scale = in_scale; a.in_shift = in_shift; a.in_act = in_act;
  a.out_scale = out_scale; a.out_shift = out_shift; a.out_act = out_act;
  a.res = res; a.ldres = ldres; a.K = K; a.N = N;
  a.g.B = B; a.g.Hs = Hs; a.g.Ws = Ws; a.g.Ho = Ho; a.g.Wo = Wo;
  a.g.kh = kh; a.g.kw = kw; a.g.stride = stride; a.g.pad = pad; a.g.dil = dil;
  a.g.transposed = transposed;
  hipStream_t s = (hipStream_t)stream;
  const int tiles = cdiv(N, 16);
  if (tiles <= 1) return launch_fwd<4, 1>(a, s);
  if (tiles == 2) return launch_fwd<4, 2>(a, s);
  if (tiles == 3) return launch_fwd<4, 3>(a, s);
  if (tiles == 4) return launch_fwd<4, 4>(a, s);
  if (tiles <= 6) return launch_fwd<2, 6>(a, s);
  if (tiles <= 8) return launch_fwd<2, 8>(a, s);
  if (tiles <= 12) return launch_fwd<1, 12>(a, s);
  return launch_fwd<1, 16>(a, s);  // N > 256 is covered by grid.y
}

static int wgrad_slabs(int64_t Mtot, int N, int K, int taps) {
  // aim at ~2048 workgroups in total, at least 128 pixels per workgroup
  int64_t per = (int64_t)cdiv(N, 64) * cdiv(K, 64) * taps;
  int64_t s = 2048 / per;
  if (s < 16) s = 16;
  if (s > 1024) s = 1024;
  if (s > Mtot / 128) s = Mtot / 128;
  if (s < 1) s = 1;
  return (int)s;
}

// floats of workspace needed by nasseg_conv_wgrad
int64_t nasseg_conv_wgrad_workspace(int B, int Ho, int Wo, int N, int K, int kh, int kw) {
  return (int64_t)wgrad_slabs((int64_t)B * Ho * Wo, N, K, kh * kw) * kh * kw * N * K;
}

// dw (N,K,kh,kw) = sum_pixels dy[pixel][n] * in_act(in_scale*x[src(pixel,tap)][k]+in_shift)
int nasseg_conv_wgrad(const float* x, int ldx, const float* dy, int lddy, float* dw, float* ws,
                      const float* in_scale, const float* in_shift, int in_act, int B, int Hs,
                      int Ws, int K, int Ho, int Wo, int N, int kh, int kw, int stride, int pad,
                      int dil, void* stream) {
  NASSEG_REQUIRE(B > 0 && Hs > 0 && Ws > 0 && Ho > 0 && Wo > 0, "conv_wgrad: bad geometry");
  NASSEG_REQUIRE(K > 0 && N > 0 && ldx >= K && lddy >= N, "conv_wgrad: bad channels");
  hipStream_t s = (hipStream_t)stream;
  WgArgs a;
  a.x = x; a.ldx = ldx; a.dy = dy; a.lddy = lddy; a.partial = ws;
  a.in_scale = in_scale; a.in_shift = in_shift; a.in_act = in_act;
  a.K = K; a.N = N;
  a.kchunks = cdiv(K, 64);
  const int nchunks = cdiv(N, 64);
  const int64_t Mtot = (int64_t)B * Ho * Wo;
  const int nslab = wgrad_slabs(Mtot, N, K, kh * kw);
  a.pix_per_block = cdiv64(Mtot, nslab);
  a.g.B = B; a.g.Hs = Hs; a.g.Ws = Ws; a.g.Ho = Ho; a.g.Wo = Wo;
  a.g.kh = kh; a.g.kw = kw; a.g.stride = stride; a.g.pad = pad; a.g.dil = dil;
  a.g.transposed = 0;
  dim3 grid(nslab, nchunks * a.kchunks, kh * kw);
  const bool vecn = ((N & 3) == 0) && ((lddy & 3) == 0);
  const bool veck = ((K & 3) == 0) && ((ldx & 3) == 0);
  if (vecn && veck)
    hipLaunchKernelGGL((conv_wgrad_kernel<true, true>), grid, dim3(256), 0, s, a);
  else if (vecn)
    hipLaunchKernelGGL((conv_wgrad_kernel<true, false>), grid, dim3(256), 0, s, a);
  else if (veck)
    hipLaunchKernelGGL((conv_wgrad_kernel<false, true>), grid, dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((conv_wgrad_kernel<false, false>), grid, dim3(256), 0, s, a);
  NASSEG_LAUNCH_CHECK("conv_wgrad_kernel");
  const int64_t per = (int64_t)kh * kw * N * K;
  hipLaunchKernelGGL(conv_wgrad_finalize, dim3((unsigned)cdiv64(per, NASSEG_RP_ELEMS)), dim3(256), 0,
                     s, ws, dw, nslab, kh * kw, N, K);
  NASSEG_LAUNCH_CHECK("conv_wgrad_finalize");
  return NASSEG_OK;
}

}  // extern "C"
