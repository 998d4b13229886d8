// One stage of a separable convolution in ONE kernel: depthwise k x k -> pointwise 1x1
// (+ the batch statistics of the BatchNorm that follows), fp32 / bf16 storage, NHWC, gfx950.
//
// Reference: SepConv's [Conv2d(C, C, k, groups=C) -> Conv2d(C, N, 1) -> BatchNorm2d -> ReLU]
// stage (src/nn/layer_factory.py:241-262) and DilConv's ReLU -> depthwise -> 1x1 -> BatchNorm
// (:207-218).  There is no BatchNorm between the two convolutions, so the depthwise output
// tile never has to leave the CU: a workgroup computes it with the "vertical strip" scheme of
// dwconv.hip (lanes along the flattened (x, channel/4) axis, P = 4 output rows per thread that
// share input rows, weights in registers / LDS), parks the P x Wt pixels x C channels in LDS
// and feeds them straight to the fp32 matrix cores as the B operand of the pointwise GEMM
// (v_mfma_f32_16x16x4_f32, operand mapping and accumulation order of conv_fwd.hip, so the
// result equals the two-kernel chain bit for bit).  Only the pointwise output is written -
// plus, when a backward pass will need it, the depthwise output (the pointwise weight
// gradient's operand); the read of it by a second kernel and one launch are gone.
//
// Fusions around it, as in the separate kernels: input prologue act(in_scale*x + in_shift) on
// load (the previous stage's BatchNorm + ReLU, never materialised; DilConv's ReLU), output
// affine + activation (inference: the following BatchNorm folded in), per-workgroup sum / sum
// of squares of the output per channel (training: that BatchNorm's batch statistics).
#include "dw_common.h"

namespace {

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float4 keep4(float4 v, bool ok) { return keep_if(v, ok); }

constexpr int kSepP = 4;      // output rows per strip
constexpr int kSepMaxPix = kSepP * 48;

struct SepArgs {
  const act_t* x;
  const float* wdw;  // [tap][C]
  const float* wpw;  // [N][C]
  act_t* zdw;        // optional: the depthwise output [B][Ho][Wo][C]
  act_t* y;          // [B][Ho][Wo][N]
  const float* in_scale;
  const float* in_shift;
  int in_act;
  const float* out_scale;
  const float* out_shift;
  int out_act;
  float* stats;  // optional [blocks][2][N]
  int H, W, C4, Ho, Wo, N, stride, pad, dil, g, nchunk;
  int Wt;  // output columns per workgroup (multiple of 4, Wt * C4 <= 256)
  int KP;  // C rounded up to a multiple of 16 (reduction length of the pointwise GEMM)
};

// K: depthwise kernel size; E: dilated input-row steps between consecutive outputs of a strip;
// WLDS: depthwise weights in LDS; PRO: input prologue; NT: 16-channel output tiles (N <= 16*NT);
// MT: 16-pixel subtiles per wave (P * Wt <= 64 * MT pixels).  Whether the statistics rows are
// emitted (a.stats) and the depthwise output is stored (a.zdw) are run-time: neither sits in a loop.
// X: output columns per thread in the depthwise phase.  With one column a thread loads K input columns per input
// row for one output column - 10 loads per output float4 for a 5 x 5 - and every load carries the input prologue
// (BatchNorm + ReLU of the stage in front: fma + clamp + border mask, ~12 vector instructions) against the 100 FMAs of
// the output itself: half of what the phase issues.  X = 2 (stride 1, dilation 1): two adjacent columns share K - 1
// of their K input columns - 6 loads per output float4; the tile is twice as wide so that all lanes still work.  The
// sums run in the same order (rows, then taps, ascending) per output: the same bits.
template <int K, int E, bool WLDS, bool PRO, int NT, int MT, int X = 1>
__global__ __launch_bounds__(256) void sepconv_fwd_kernel(SepArgs a) {
  // dynamic LDS: the depthwise output tile zt[P * Wt][LS] (rows of LS = KP + 4 floats:
  // ds_read_b128 by the 16 lanes of a k-group conflict-free) and, with WLDS, the depthwise
  // weights lw[K * K][C4] (float4).  (Staging the pointwise weights here as well was measured:
  // no gain - they are L1 hits - and the LDS it takes costs a resident workgroup.)
  extern __shared__ float smem[];
  __shared__ float sred[4][2][NT * 16];
  constexpr int P = kSepP;
  const int C4 = a.C4, C = C4 * 4, H = a.H, W = a.W, Ho = a.Ho, Wo = a.Wo;
  const int LS = a.KP + 4;
  const int tid = threadIdx.x;
  const int npix = P * a.Wt;
  float* zt = smem;
  float4* lw = reinterpret_cast<float4*>(smem + npix * LS);
  const bool STATS = a.stats != nullptr;
  const bool WRZ = a.zdw != nullptr;
  if (WLDS) {
    for (int i = tid; i < K * K * C4; i += 256) lw[i] = lda4(a.wdw + (size_t)i * 4);  // [tap][C4]
  }
  if (a.KP > C) {  // zero the padding columns of the reduction axis once
    const int padc = a.KP - C;
    for (int i = tid; i < npix * padc; i += 256) zt[(i / padc) * LS + C + i % padc] = 0.f;
  }
  if (WLDS) __syncthreads();

  const int b = blockIdx.z;
  const int r = blockIdx.y % a.g;
  const int chunk = blockIdx.y / a.g;
  const int oy0 = chunk * (P * a.g) + r;
  const int ox0 = blockIdx.x * a.Wt;

  // ---- phase 1: depthwise strip -> LDS (and, optionally, HBM) -----------------------------
  if constexpr (X == 2) {
    if (tid < (a.Wt >> 1) * C4) {
      const int xp = tid / C4;
      const int c4 = tid - xp * C4;
      const int xl = xp * 2;
      const int ox = ox0 + xl;
      constexpr int KX = K + 1;
      float4 w[WLDS ? 1 : K * K];
      if (!WLDS) {
#pragma unroll
        for (int t = 0; t < K * K; ++t) w[t] = lda4(a.wdw + (size_t)t * C + c4 * 4);
      }
      Prologue pro;
      if (PRO) pro = make_prologue(a.in_scale, a.in_shift, a.in_act, c4);
      int xoff[KX];
      bool xok[KX];
#pragma unroll
      for (int t = 0; t < KX; ++t) {  // (stride 1, dilation 1: input column ox - pad + t)
        const int ix = ox - a.pad + t;
        xok[t] = (ix >= 0) && (ix < W);
        xoff[t] = (ix < 0 ? 0 : (ix >= W ? W - 1 : ix)) * C;
      }
      float4 acc[P][2];
#pragma unroll
      for (int j = 0; j < P; ++j) acc[j][0] = acc[j][1] = f4zero();
      const act_t* xb = a.x + (size_t)b * H * W * C + c4 * 4;
      const int iy0 = oy0 - a.pad;
      constexpr int Q = (P - 1) * E + K;
      auto load_row = [&](int q, float4* v) {
        const int iy = iy0 + q;
        const bool yok = (iy >= 0) && (iy < H);
        const act_t* xr = xb + (size_t)(iy < 0 ? 0 : (iy >= H ? H - 1 : iy)) * W * C;
#pragma unroll
        for (int t = 0; t < KX; ++t) {
          float4 u = lda4(xr + xoff[t]);
          if (PRO) u = apply_prologue(u, pro);
          v[t] = keep4(u, yok && xok[t]);
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
              const float4 wv = WLDS ? lw[(ty * K + tx) * C4 + c4] : w[WLDS ? 0 : ty * K + tx];
              acc[j][0] = fma4(wv, vcur[tx], acc[j][0]);
              acc[j][1] = fma4(wv, vcur[tx + 1], acc[j][1]);
            }
          }
        }
#pragma unroll
        for (int j = 0; j < P; ++j) {
          pin(acc[j][0]);
          pin(acc[j][1]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < KX; ++t) vcur[t] = vnext[t];
      }
#pragma unroll
      for (int j = 0; j < P; ++j) {
        const int oy = oy0 + j * a.g;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const bool ok = ox + c < Wo && oy < Ho;
          float4 o = acc[j][c];
#ifdef NASSEG_BF16
          o = make_float4(bf16_to_f32(f32_to_bf16(o.x)), bf16_to_f32(f32_to_bf16(o.y)),
                          bf16_to_f32(f32_to_bf16(o.z)), bf16_to_f32(f32_to_bf16(o.w)));
#endif
          if (WRZ) {
            if (ok) sta4(a.zdw + (((size_t)b * Ho + oy) * Wo + ox + c) * C + c4 * 4, o);
          }
          *reinterpret_cast<float4*>(&zt[(j * a.Wt + xl + c) * LS + c4 * 4]) = keep4(o, ok);
        }
      }
    }
  } else if (tid < a.Wt * C4) {
    const int xl = tid / C4;
    const int c4 = tid - xl * C4;
    const int ox = ox0 + xl;
    const bool colok = ox < Wo;
    const int oxc = colok ? ox : Wo - 1;
    float4 w[WLDS ? 1 : K * K];
    if (!WLDS) {
#pragma unroll
      for (int t = 0; t < K * K; ++t) w[t] = lda4(a.wdw + (size_t)t * C + c4 * 4);
    }
    Prologue pro;
    if (PRO) pro = make_prologue(a.in_scale, a.in_shift, a.in_act, c4);
    int xoff[K];
    bool xok[K];
#pragma unroll
    for (int tx = 0; tx < K; ++tx) {
      const int ix = oxc * a.stride - a.pad + tx * a.dil;
      xok[tx] = (ix >= 0) && (ix < W);
      xoff[tx] = (ix < 0 ? 0 : (ix >= W ? W - 1 : ix)) * C;
    }
    float4 acc[P];
#pragma unroll
    for (int j = 0; j < P; ++j) acc[j] = f4zero();
    const act_t* xb = a.x + (size_t)b * H * W * C + c4 * 4;
    const int iy0 = oy0 * a.stride - a.pad;
    constexpr int Q = (P - 1) * E + K;
    auto load_row = [&](int q, float4* v) {
      const int iy = iy0 + q * a.dil;
      const bool yok = (iy >= 0) && (iy < H);
      const act_t* xr = xb + (size_t)(iy < 0 ? 0 : (iy >= H ? H - 1 : iy)) * W * C;
#pragma unroll
      for (int tx = 0; tx < K; ++tx) {
        float4 t = lda4(xr + xoff[tx]);
        if (PRO) t = apply_prologue(t, pro);
        v[tx] = keep4(t, yok && xok[tx]);
      }
    };
    float4 vcur[K], vnext[K];
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
            acc[j] = fma4(WLDS ? lw[(ty * K + tx) * C4 + c4] : w[WLDS ? 0 : ty * K + tx], vcur[tx], acc[j]);
        }
      }
#pragma unroll
      for (int j = 0; j < P; ++j) pin(acc[j]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int tx = 0; tx < K; ++tx) vcur[tx] = vnext[tx];
    }
#pragma unroll
    for (int j = 0; j < P; ++j) {
      const int oy = oy0 + j * a.g;
      const bool ok = colok && oy < Ho;
      float4 o = acc[j];
#ifdef NASSEG_BF16
      // the value a separate pointwise kernel (and the backward pass) would read back
      o = make_float4(bf16_to_f32(f32_to_bf16(o.x)), bf16_to_f32(f32_to_bf16(o.y)),
                      bf16_to_f32(f32_to_bf16(o.z)), bf16_to_f32(f32_to_bf16(o.w)));
#endif
      if (WRZ) {
        if (ok) sta4(a.zdw + (((size_t)b * Ho + oy) * Wo + ox) * C + c4 * 4, o);
      }
      *reinterpret_cast<float4*>(&zt[(j * a.Wt + xl) * LS + c4 * 4]) = keep4(o, ok);
    }
  }
  __syncthreads();

  // ---- phase 2: pointwise GEMM on the tile: D[n][pixel] = sum_k Wpw[n][k] * zt[pixel][k] ----
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int j = lane & 15;
  const int kg = lane >> 4;
  const int nsub = npix >> 4;
  int wn[NT];
  bool wok[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = nt * 16 + j;
    wok[nt] = n < a.N;
    wn[nt] = wok[nt] ? n : a.N - 1;
  }
  f32x4 acc2[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc2[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int zoff[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int s = wave + mt * 4;
    zoff[mt] = ((s < nsub ? s : 0) * 16 + j) * LS;
  }
  const int nks = a.KP >> 4;
  for (int ks = 0; ks < nks; ++ks) {
    const int k = ks * 16 + kg * 4;
    const bool kok = k < C;
    float4 bv[MT], av[NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) bv[mt] = *reinterpret_cast<const float4*>(&zt[zoff[mt] + k]);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      av[nt] = keep4(ld4(a.wpw + (size_t)wn[nt] * C + (kok ? k : 0)), wok[nt] && kok);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      if (wave + mt * 4 < nsub) {  // (wave-uniform)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          acc2[mt][nt] = mfma16(av[nt].x, bv[mt].x, acc2[mt][nt]);
          acc2[mt][nt] = mfma16(av[nt].y, bv[mt].y, acc2[mt][nt]);
          acc2[mt][nt] = mfma16(av[nt].z, bv[mt].z, acc2[mt][nt]);
          acc2[mt][nt] = mfma16(av[nt].w, bv[mt].w, acc2[mt][nt]);
        }
      }
    }
  }

  // ---- epilogue: lane holds pixel (subtile, j), channels nt*16 + 4*kg .. +3 ------------------
  float sx[NT][4], sq[NT][4];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int c = 0; c < 4; ++c) sx[nt][c] = sq[nt][c] = 0.f;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int s = wave + mt * 4;
    const int p = s * 16 + j;
    const int pr = p / a.Wt, pc = p - pr * a.Wt;
    const int oy = oy0 + pr * a.g, ox = ox0 + pc;
    const bool ok = s < nsub && oy < Ho && ox < Wo;
    const size_t m = ((size_t)b * Ho + (oy < Ho ? oy : Ho - 1)) * Wo + (ox < Wo ? ox : Wo - 1);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = nt * 16 + kg * 4;
      const bool nok = n < a.N;  // N % 4 == 0
      const int nc = nok ? n : 0;
      const f32x4 c = acc2[mt][nt];
      float4 o = make_float4(c[0], c[1], c[2], c[3]);
      if (STATS) {
        const float4 v = keep4(o, ok);
        sx[nt][0] += v.x; sx[nt][1] += v.y; sx[nt][2] += v.z; sx[nt][3] += v.w;
        sq[nt][0] = fmaf(v.x, v.x, sq[nt][0]); sq[nt][1] = fmaf(v.y, v.y, sq[nt][1]);
        sq[nt][2] = fmaf(v.z, v.z, sq[nt][2]); sq[nt][3] = fmaf(v.w, v.w, sq[nt][3]);
      } else {
        if (a.out_scale) o = mul4(o, ld4(a.out_scale + nc));
        if (a.out_shift) o = add4(o, ld4(a.out_shift + nc));
        if (a.out_act) o = act_apply4(o, a.out_act);
      }
      if (ok && nok) sta4(a.y + m * a.N + n, o);
    }
  }
  if (STATS) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        sx[nt][c] = row16_allsum(sx[nt][c]);
        sq[nt][c] = row16_allsum(sq[nt][c]);
      }
      if (j == 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          sred[wave][0][nt * 16 + kg * 4 + c] = sx[nt][c];
          sred[wave][1][nt * 16 + kg * 4 + c] = sq[nt][c];
        }
      }
    }
    __syncthreads();
    const size_t blk = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    for (int t = tid; t < NT * 16; t += 256) {
      if (t < a.N) {
        float* po = a.stats + blk * 2 * a.N + t;
        po[0] = (sred[0][0][t] + sred[1][0][t]) + (sred[2][0][t] + sred[3][0][t]);
        po[a.N] = (sred[0][1][t] + sred[1][1][t]) + (sred[2][1][t] + sred[3][1][t]);
      }
    }
  }
}

// 1: two output columns per thread where the geometry allows (stride 1, dilation 1); 0: one everywhere (A/B)
#ifndef NASSEG_SEP_X2
#define NASSEG_SEP_X2 1
#endif
struct SepPlan {
  int ok, Wt, KP, g, e, nchunk, gx, gy, x;
};
inline SepPlan sep_plan(int B, int C, int Ho, int Wo, int N, int K, int stride, int dil) {
  SepPlan p = {};
  const StripCfg sc = strip_cfg(stride, dil);
  p.g = sc.g;
  p.e = sc.e;
  if (C % 4 || N % 4 || N > 64 || N <= 0 || C <= 0 || C / 4 > 64 || !(K == 3 || K == 5) ||
      !(sc.e == 1 || sc.e == 2) || B > 65535)
    return p;
  const int C4 = C / 4;
  // columns per tile: a multiple of 16 where the channel count allows it, so that the P * Wt / 16
  // pixel subtiles divide evenly among the four waves (24 / 48 channels: 192 of the 256 lanes work
  // in the depthwise phase - measured better than 240 lanes and 3,3,2,2 subtiles), else of 4
  // two columns per thread (X = 2) where the wider tiles still make 384 workgroups: tools/kbench_sepconv.py, 5 x 5
  // 32 -> 32 at 4 x 128 x 256 21.8 -> 18.2 us, 64 -> 64 45.6 -> 37.9, 48 -> 48 at 8 x 179 x 179 70.1 -> 60.9, 3 x 3
  // 32 -> 32 at 4 x 256 x 512 42.2 -> 39.1; but 64 -> 64 at 4 x 32 x 64 (64 workgroups instead of 128) 10.1 -> 12.1
  p.x = 1;
  if (NASSEG_SEP_X2 && stride == 1 && dil == 1 && C4 >= 8) {
    int w2 = 512 / C4;
    w2 = w2 >= 16 ? (w2 & ~15) : (w2 & ~3);
    const int need2 = (Wo + 3) & ~3;
    if (w2 > need2) w2 = need2;
    if (w2 >= 4 && (int64_t)B * cdiv(Ho, kSepP * sc.g) * sc.g * cdiv(Wo, w2) >= 384) p.x = 2;
  }
  int wt = 256 * p.x / C4;
  wt = wt >= 16 ? (wt & ~15) : (wt & ~3);
  if (wt > 48 * p.x) wt = 48 * p.x;
  if (wt < 4) return p;
  // narrow maps: do not spread a tile far beyond the row
  const int need = (Wo + 3) & ~3;
  if (wt > need) wt = need;
  p.Wt = wt;
  p.KP = (C + 15) & ~15;
  p.nchunk = cdiv(Ho, kSepP * sc.g);
  p.gx = cdiv(Wo, wt);
  p.gy = p.nchunk * sc.g;
  p.ok = p.gy <= 65535;
  return p;
}

template <int K, int E, bool WLDS, bool PRO, int NT>
int sep_launch2(const SepArgs& a, dim3 grid, size_t lds, int mt, int x, hipStream_t s) {
#define GO_SEP(MT_, X_) hipLaunchKernelGGL((sepconv_fwd_kernel<K, E, WLDS, PRO, NT, MT_, X_>), grid, dim3(256), lds, s, a)
  if constexpr (E == 1) {
    if (x == 2) {  // (tiles twice as wide: C >= 32 channels, at most 64 columns = four subtiles per wave, < 64 KB of LDS)
      if (mt <= 1) GO_SEP(1, 2);
      else if (mt == 2) GO_SEP(2, 2);
      else if (mt == 3) GO_SEP(3, 2);
      else GO_SEP(4, 2);
      return NASSEG_OK;
    }
  }
  if (mt <= 1) GO_SEP(1, 1);
  else if (mt == 2) GO_SEP(2, 1);
  else GO_SEP(3, 1);
#undef GO_SEP
  return NASSEG_OK;
}
template <int K, int E, bool WLDS>
int sep_launch1(const SepArgs& a, dim3 grid, size_t lds, bool pro, int nt, int mt, int x, hipStream_t s) {
#define GO_NT(PR_)                                                       \
  do {                                                                   \
    if (nt <= 1) return sep_launch2<K, E, WLDS, PR_, 1>(a, grid, lds, mt, x, s); \
    if (nt == 2) return sep_launch2<K, E, WLDS, PR_, 2>(a, grid, lds, mt, x, s); \
    if (nt == 3) return sep_launch2<K, E, WLDS, PR_, 3>(a, grid, lds, mt, x, s); \
    return sep_launch2<K, E, WLDS, PR_, 4>(a, grid, lds, mt, x, s);              \
  } while (0)
  if (pro) GO_NT(true);
  GO_NT(false);
#undef GO_NT
}

}  // namespace

extern "C" {

#if NASSEG_FP32_ONLY
// > 0: nasseg_sepconv_fwd serves this geometry, and writes that many statistics rows
// (one per workgroup) when asked for statistics; 0: run the two convolutions separately
int64_t nasseg_sepconv_blocks(int B, int C, int Ho, int Wo, int N, int K, int stride, int dil) {
  const SepPlan p = sep_plan(B, C, Ho, Wo, N, K, stride, dil);
  return p.ok ? (int64_t)p.gx * p.gy * B : 0;
}
#endif  // NASSEG_FP32_ONLY

// y = out_act(out_scale * pointwise(depthwise(in_act(in_scale*x + in_shift))) + out_shift)
//   x [B][H][W][C], wdw depthwise weights packed [tap][C] (nasseg_dw_pack_weight / pack kind 3),
//   wpw pointwise weights (N, C, 1, 1) as they are, y [B][Ho][Wo][N], C % 4 == 0, N % 4 == 0,
//   N <= 64, C <= 256, k in {3, 5}, strip geometries (nasseg_sepconv_blocks > 0).
// zdw != NULL: the depthwise output [B][Ho][Wo][C] is stored as well (the backward pass reads it).
// stats != NULL: rows [blk][2][N] of per-workgroup sums of y and y^2 (no output epilogue then),
//   blk < nasseg_sepconv_blocks(...), for nasseg_bn_finalize.
int NASSEG_FN(sepconv_fwd)(const act_t* x, const float* wdw, const float* wpw, act_t* zdw, act_t* y,
                           const float* in_scale, const float* in_shift, int in_act,
                           const float* out_scale, const float* out_shift, int out_act, int B, int H,
                           int W, int C, int Ho, int Wo, int N, int K, int stride, int pad, int dil,
                           float* stats, void* stream) {
  NASSEG_REQUIRE(x && wdw && wpw && y, "sepconv_fwd: null tensor");
  NASSEG_REQUIRE(B > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && stride > 0 && dil > 0,
                 "sepconv_fwd: bad geometry");
  const SepPlan p = sep_plan(B, C, Ho, Wo, N, K, stride, dil);
  NASSEG_REQUIRE(p.ok, "sepconv_fwd: geometry has no fused path (C=%d N=%d k=%d stride=%d dil=%d)", C, N,
                 K, stride, dil);
  NASSEG_REQUIRE(!stats || (!out_scale && !out_shift && !out_act),
                 "sepconv_fwd: statistics are taken of the raw output (no output epilogue)");
  SepArgs a = {};
  a.x = x; a.wdw = wdw; a.wpw = wpw; a.zdw = zdw; a.y = y;
  a.in_scale = in_scale; a.in_shift = in_shift; a.in_act = in_act;
  a.out_scale = out_scale; a.out_shift = out_shift; a.out_act = out_act;
  a.stats = stats;
  a.H = H; a.W = W; a.C4 = C / 4; a.Ho = Ho; a.Wo = Wo; a.N = N;
  a.stride = stride; a.pad = pad; a.dil = dil; a.g = p.g; a.nchunk = p.nchunk;
  a.Wt = p.Wt; a.KP = p.KP;
  const dim3 grid(p.gx, p.gy, B);
  const int nt = cdiv(N, 16);
  const int mt = cdiv(kSepP * p.Wt, 64);  // 16-pixel subtiles per wave
  const size_t lds = (size_t)kSepP * p.Wt * (p.KP + 4) * sizeof(float) +
                     (K == 5 ? (size_t)K * K * (C / 4) * sizeof(float4) : 0);
  const bool pro = in_scale || in_shift || in_act;
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if (K == 3 && p.e == 1) rc = sep_launch1<3, 1, false>(a, grid, lds, pro, nt, mt, p.x, s);
  else if (K == 3) rc = sep_launch1<3, 2, false>(a, grid, lds, pro, nt, mt, p.x, s);
  else if (p.e == 1) rc = sep_launch1<5, 1, true>(a, grid, lds, pro, nt, mt, p.x, s);
  else rc = sep_launch1<5, 2, true>(a, grid, lds, pro, nt, mt, p.x, s);
  if (rc) return rc;
  NASSEG_LAUNCH_CHECK("sepconv_fwd_kernel");
  return NASSEG_OK;
}

}  // extern "C"
