// Dense convolution forward / backward-data (1x1 pointwise and k x k, any stride /
// dilation) as an implicit GEMM on the fp32 matrix cores (v_mfma_f32_16x16x4_f32),
// NHWC, gfx950.
//
// Reference call sites: conv1x1 / conv3x3 / conv_bn / conv_bn_relu
// (src/nn/layer_factory.py:7-24,94-122), the pointwise stage of SepConv /
// DilConv / InvertedResidual / Pool / Adapt / ConcatReduce (:125-382) and the
// classifier heads (src/nn/micro_decoders.py:210-227,360-363).
//
// fp32 in / fp32 accumulate MFMA is bit-equivalent to an fmaf chain, which is what
// keeps logits within 1e-4 of the reference.  D[n][pixel] = sum_k W[n][k] * X[pixel][k]:
// each lane loads one float4 along the (contiguous) reduction axis and feeds its four
// components to four consecutive MFMAs (the "k" label of an MFMA slot is arbitrary as
// long as A and B agree); the accumulator then holds four consecutive output channels
// of one pixel per lane -> float4 stores.  A wave owns MT x NT tiles of 16 pixels x 16
// channels.  Up to N = 64 a wave computes all of N for its own pixels; above that (WS) the
// four waves of a workgroup share the same 16*MT pixels and split N between them, which
// keeps the per-wave weight traffic (L1) and accumulator count independent of N.  Either
// way X comes from HBM once.  No LDS.
//
// Variants (template): GATHER = needs per-tap source-pixel arithmetic (k x k, strided,
// transposed); KM = how the reduction axis is read (aligned float4 / scalar / "flat"
// im2col of a small-K conv such as the 3-channel stem); PRO = fused per-input-channel
// affine+activation prologue; VECN = N % 4 == 0 (float4 epilogue).
//
// Packed weight layouts (nasseg_conv_pack_weight):
//   mode 0 forward       : wp[tap][N][K]    from OIHW (N,K,kh,kw)
//   mode 1 backward-data : wp[tap][K][N]    (roles of N and K swapped)
//   mode 2 flat forward  : wp[N][tap*K + k]
#include "conv_args.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>

// Which pointwise calls take the persistent kernel (conv_pw_kernel).  Default (-2): where it measured
// faster (pw_fwd_plan).  v >= 0: every call it supports over at least v output pixels (0: all of them - what
// the parity tests use; a huge value: none); v == -1 only queries.  Returns the previous setting.
// Process-wide; outputs do not depend on it, BatchNorm statistics only in the rounding of their partial sums.
extern "C" int64_t nasseg_conv_pw_min_pixels(int64_t v);

#if NASSEG_FP32_ONLY
std::atomic<int> g_conv_deep_k{1};
#else
extern std::atomic<int> g_conv_deep_k;
#endif

namespace {

enum { KM_VEC = 0, KM_SCALAR = 1, KM_FLAT = 2 };

// k-steps of X fetched per iteration of the reduction loop (compile-time).  Measured: 2 and 4 are
// both slower than 1 (224->64: 205 -> 253 us, 16->96: 228 -> 378 us at 4) - the extra operand
// registers cost occupancy, and thread-level parallelism is what hides HBM latency here
#ifndef NASSEG_CONV_KU
#define NASSEG_CONV_KU 1
#endif
// initial setting of nasseg_conv_pw_min_pixels (experiments: -DNASSEG_PW_MIN_PIXELS=2000000000 turns
// conv_pw_kernel off)
#ifndef NASSEG_PW_MIN_PIXELS
#define NASSEG_PW_MIN_PIXELS -2
#endif
// 0: 17 ... 21 output channels of the LDS-tiled 3x3 kernel as two MFMA tiles (rounds 1-4; A/B, tools/gpu.sh flags)
#ifndef NASSEG_LDS3X3_VALU_TAIL
#define NASSEG_LDS3X3_VALU_TAIL 1
#endif

// 4 floats along the reduction axis starting at k (clamped, always in range); the caller
// masks what lies beyond K
template <bool VEC>
__device__ __forceinline__ float4 load4(const float* row, int k, int K) {
  if (VEC) {
    return lda4(row + (k < K ? k : 0));
  } else {
    float4 v;
    v.x = keep_if(row[k + 0 < K ? k + 0 : 0], k + 0 < K);
    v.y = keep_if(row[k + 1 < K ? k + 1 : 0], k + 1 < K);
    v.z = keep_if(row[k + 2 < K ? k + 2 : 0], k + 2 < K);
    v.w = keep_if(row[k + 3 < K ? k + 3 : 0], k + 3 < K);
    return v;
  }
}

#ifdef NASSEG_BF16
template <bool VEC>
__device__ __forceinline__ float4 load4(const bf16_t* row, int k, int K) {
  if (VEC) {
    return lda4(row + (k < K ? k : 0));
  } else {
    float4 v;
    v.x = keep_if(lda1(row + (k + 0 < K ? k + 0 : 0)), k + 0 < K);
    v.y = keep_if(lda1(row + (k + 1 < K ? k + 1 : 0)), k + 1 < K);
    v.z = keep_if(lda1(row + (k + 2 < K ? k + 2 : 0)), k + 2 < K);
    v.w = keep_if(lda1(row + (k + 3 < K ? k + 3 : 0)), k + 3 < K);
    return v;
  }
}
#endif

// EPI: any output epilogue (scale / shift(bias) / activation / residual) is present.
// STATS == 1: also emit per-workgroup partial sums of y and y^2 per output channel - the
// BatchNorm batch statistics of the layer that follows, at no extra pass over y.
// STATS == 2 (backward-data calls): y is the gradient g w.r.t. a = act(b_scale*z + b_shift),
// the normalised activation this conv read in the forward pass.  The epilogue multiplies g
// by act'(...) (so the masked gradient is what gets stored) and emits the partial sums of
// g' and g'*xhat, xhat = (z - mean)*invstd: the first half of that BatchNorm's backward at
// the price of one read of z instead of a separate pass over g and z.
// STATS == 3: only the act' mask of STATS == 2 (b_scale / b_shift may be null = identity): the
// backward of an activation that was applied on load, without a pass over dx and x.
// KU_: k-steps of X requested per iteration (see NASSEG_CONV_KU): 1 on large maps, where other waves hide a round
// trip; 4 on SMALL maps (launch_one: at most kDeepKMaxPixels = 8192 pixels), where a workgroup per CU or fewer runs and the
// reduction is a chain of round trips - 960 -> 160 at 16 x 11 x 11: 60 of them, 78 us for 0.6 GFLOP.
template <int MT, int NT, int KM, bool GATHER, bool PRO, bool VECN, bool EPI, int STATS, bool WS,
          int KU_ = NASSEG_CONV_KU>
__global__ __launch_bounds__(256) void conv_fwd_kernel(FwdArgs a) {
  constexpr bool kSums = STATS == 1 || STATS == 2;
  __shared__ float sred[(kSums && !WS) ? 4 : 1][2][(kSums && !WS) ? NT * 16 : 1];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int j = lane & 15;   // pixel within subtile (B operand col) / n within tile (A operand row)
  const int kg = lane >> 4;  // k group
  const int Mtot = a.g.B * a.g.Ho * a.g.Wo;
  const int m_base = WS ? blockIdx.x * (16 * MT) : (blockIdx.x * 4 + wave) * (16 * MT);
  const bool active = m_base < Mtot;  // wave-uniform
  if ((!kSums || WS) && !active) return;
  const int n_base = WS ? (blockIdx.y * 4 + wave) * (16 * NT) : blockIdx.y * (16 * NT);
  if (WS && n_base >= a.N) return;  // (no workgroup barrier on the WS path)

  // destination pixels of this lane (clamped: out-of-range rows compute garbage that is never stored)
  int pm[MT], pb[MT], py[MT], px[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m_base + mt * 16 + j;
    pm[mt] = m < Mtot ? m : Mtot - 1;
    if (GATHER) {
      px[mt] = pm[mt] % a.g.Wo;
      const int t = pm[mt] / a.g.Wo;
      py[mt] = t % a.g.Ho;
      pb[mt] = t / a.g.Ho;
    }
  }
  // weight rows of this lane (clamped + masked)
  int wn[NT];
  bool wok[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = n_base + nt * 16 + j;
    wok[nt] = n < a.N;
    wn[nt] = wok[nt] ? n : a.N - 1;
  }

  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int ntaps = a.g.kh * a.g.kw;
  constexpr int KU = (KM == KM_VEC) ? KU_ : 1;
  const ActSel pact = act_sel(a.in_act);
  const int Kq = (KM == KM_FLAT) ? ntaps * a.K : a.K;  // reduction length of one pass
  const int nk = (Kq + 15) >> 4;

  auto mma = [&](const float4* bv, const float4* av) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        acc[mt][nt] = mfma16(av[nt].x, bv[mt].x, acc[mt][nt]);
        acc[mt][nt] = mfma16(av[nt].y, bv[mt].y, acc[mt][nt]);
        acc[mt][nt] = mfma16(av[nt].z, bv[mt].z, acc[mt][nt]);
        acc[mt][nt] = mfma16(av[nt].w, bv[mt].w, acc[mt][nt]);
      }
  };

  if (KM == KM_FLAT) {
    // im2col on the fly: k' = tap*K + k, every element gathered separately
    for (int it = 0; it < nk; ++it) {
      const int k = it * 16 + kg * 4;
      float4 bv[MT], av[NT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        float e[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int kq = (k + c < Kq) ? k + c : 0;
          const int tap = kq / a.K, cin = kq - tap * a.K;
          const int ty = tap / a.g.kw, tx = tap - ty * a.g.kw;
          const int sp = src_pixel(a.g, pb[mt], py[mt], px[mt], ty, tx);
          const float v = lda1(a.x + (int64_t)(sp < 0 ? 0 : sp) * a.ldx + cin);
          e[c] = keep_if(v, sp >= 0 && k + c < Kq);
        }
        bv[mt] = make_float4(e[0], e[1], e[2], e[3]);
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        av[nt] = keep_if(load4<false>(a.w + (int64_t)wn[nt] * Kq, k, Kq), wok[nt]);
      mma(bv, av);
    }
  } else {
    for (int tap = 0; tap < (GATHER ? ntaps : 1); ++tap) {
      const act_t* xrow[MT];
      bool xok[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        if (GATHER) {
          const int ty = tap / a.g.kw, tx = tap - ty * a.g.kw;
          const int sp = src_pixel(a.g, pb[mt], py[mt], px[mt], ty, tx);
          xok[mt] = sp >= 0;
          xrow[mt] = a.x + (int64_t)(sp < 0 ? 0 : sp) * a.ldx;
        } else {
          xok[mt] = true;
          xrow[mt] = a.x + (int64_t)pm[mt] * a.ldx;
        }
      }
      const float* wrow[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) wrow[nt] = a.w + ((int64_t)tap * a.N + wn[nt]) * a.K;
      for (int it = 0; it < nk; it += KU) {
        // X operands of KU k-steps are requested together (bytes in flight per wave are what
        // bounds a latency-limited stream); the weights (L1 hits) follow step by step
        float4 bv[KU][MT];
#pragma unroll
        for (int u = 0; u < KU; ++u) {
          const int k = (it + u) * 16 + kg * 4;
          const bool kok = k < a.K;  // (VEC: the whole float4 is in or out)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            float4 v = load4<KM == KM_VEC>(xrow[mt], k, a.K);
            if (PRO) {
              const float4 s = a.in_scale ? load4<KM == KM_VEC>(a.in_scale, k, a.K)
                                          : make_float4(1.f, 1.f, 1.f, 1.f);
              const float4 h = a.in_shift ? load4<KM == KM_VEC>(a.in_shift, k, a.K) : f4zero();
              v = act_apply4(fma4(v, s, h), pact);
              if (KM != KM_VEC) {  // restore the zero padding beyond K
                v.y = keep_if(v.y, k + 1 < a.K);
                v.z = keep_if(v.z, k + 2 < a.K);
                v.w = keep_if(v.w, k + 3 < a.K);
              }
            }
            bv[u][mt] = keep_if(v, xok[mt] && kok);
          }
        }
        if constexpr (KU > 1) {
          // small maps: the weights of the KU steps are requested with the X operands - with a workgroup or two per
          // CU they come from L2, not L1, and a round trip per step is what the kernel would spend its time on
          float4 av[KU][NT];
#pragma unroll
          for (int u = 0; u < KU; ++u) {
            const int k = (it + u) * 16 + kg * 4;
            const bool kok = k < a.K;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              av[u][nt] = keep_if(load4<KM == KM_VEC>(wrow[nt], k, a.K), wok[nt] && kok);
          }
#pragma unroll
          for (int u = 0; u < KU; ++u) mma(bv[u], av[u]);
        } else {
#pragma unroll
          for (int u = 0; u < KU; ++u) {
            const int k = (it + u) * 16 + kg * 4;
            const bool kok = k < a.K;
            float4 av[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              av[nt] = keep_if(load4<KM == KM_VEC>(wrow[nt], k, a.K), wok[nt] && kok);
            mma(bv[u], av);
          }
        }
      }
    }
  }

  if (STATS != 0) {
    // per-channel partial sums over the MT subtiles in registers, over the 16 pixel lanes with
    // xor-shuffles (the 16-lane group of a k-group holds the same 4 channels), over the 4
    // waves through LDS (EPI is off on this path)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      float sx[4] = {0.f, 0.f, 0.f, 0.f}, sq[4] = {0.f, 0.f, 0.f, 0.f};
      float bsc[4], bsh[4], bmu[4], bis[4];
      const int nb = n_base + nt * 16 + kg * 4;
      const int nbc = nb < a.N ? nb : 0;
      if (STATS == 3) {
        const float4 t0 = a.b_scale ? lda4(a.b_scale + nbc) : make_float4(1.f, 1.f, 1.f, 1.f);
        const float4 t1 = a.b_shift ? lda4(a.b_shift + nbc) : f4zero();
        bsc[0] = t0.x; bsc[1] = t0.y; bsc[2] = t0.z; bsc[3] = t0.w;
        bsh[0] = t1.x; bsh[1] = t1.y; bsh[2] = t1.z; bsh[3] = t1.w;
      }
      if (STATS == 2) {
        const float4 t0 = lda4(a.b_scale + nbc), t1 = lda4(a.b_shift + nbc), t2 = lda4(a.b_mean + nbc),
                     t3 = lda4(a.b_invstd + nbc);
        bsc[0] = t0.x; bsc[1] = t0.y; bsc[2] = t0.z; bsc[3] = t0.w;
        bsh[0] = t1.x; bsh[1] = t1.y; bsh[2] = t1.z; bsh[3] = t1.w;
        bmu[0] = t2.x; bmu[1] = t2.y; bmu[2] = t2.z; bmu[3] = t2.w;
        bis[0] = t3.x; bis[1] = t3.y; bis[2] = t3.z; bis[3] = t3.w;
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const bool ok = active && (m_base + mt * 16 + j < Mtot);
        if (STATS == 3) {
          const float4 z4 = lda4(a.bz + (int64_t)pm[mt] * a.ldbz + nbc);
          const float zz[4] = {z4.x, z4.y, z4.z, z4.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[mt][nt][r] *= act_mask(fmaf(zz[r], bsc[r], bsh[r]), a.b_act);
        } else if (STATS == 2) {
          const float4 z4 = lda4(a.bz + (int64_t)pm[mt] * a.ldbz + nbc);
          const float zz[4] = {z4.x, z4.y, z4.z, z4.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float g = acc[mt][nt][r] * act_mask(fmaf(zz[r], bsc[r], bsh[r]), a.b_act);
            acc[mt][nt][r] = g;
            const float v = keep_if(g, ok);
            sx[r] += v;
            sq[r] = fmaf(v, (zz[r] - bmu[r]) * bis[r], sq[r]);
          }
        } else {
          const f32x4 c = acc[mt][nt];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = keep_if(c[r], ok);
            sx[r] += v;
            sq[r] = fmaf(v, v, sq[r]);
          }
        }
      }
      if (!kSums) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        sx[r] = row16_allsum(sx[r]);
        sq[r] = row16_allsum(sq[r]);
      }
      if (WS) {
        // this wave alone owns these channels of the workgroup's pixels
        if (j == 0 && nb < a.N) {
          float* po = a.stats + (int64_t)blockIdx.x * 2 * a.N + nb;
          sta4(po, make_float4(sx[0], sx[1], sx[2], sx[3]));
          sta4(po + a.N, make_float4(sq[0], sq[1], sq[2], sq[3]));
        }
      } else if constexpr (kSums) {
        if (j == 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            sred[wave][0][nt * 16 + kg * 4 + r] = sx[r];
            sred[wave][1][nt * 16 + kg * 4 + r] = sq[r];
          }
        }
      }
    }
    if constexpr (kSums && !WS) {
      __syncthreads();
      for (int t = threadIdx.x; t < NT * 16; t += 256) {
        const int n = n_base + t;
        if (n < a.N) {
          float* po = a.stats + (int64_t)blockIdx.x * 2 * a.N + n;
          po[0] = (sred[0][0][t] + sred[1][0][t]) + (sred[2][0][t] + sred[3][0][t]);
          po[a.N] = (sred[0][1][t] + sred[1][1][t]) + (sred[2][1][t] + sred[3][1][t]);
        }
      }
      if (!active) return;
    }
  }

  // epilogue: lane holds pixel j of each subtile, channels n0 + 4*kg + {0..3}
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = n_base + nt * 16 + kg * 4;
    if (VECN && !EPI) {
      const bool nok = n < a.N;  // N % 4 == 0: all four channels in or out
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int m = m_base + mt * 16 + j;
        const f32x4 c = acc[mt][nt];
        if (nok && m < Mtot) sta4(a.y + (int64_t)m * a.ldy + n, make_float4(c[0], c[1], c[2], c[3]));
      }
    } else if (VECN) {
      const bool nok = n < a.N;
      const int nc = nok ? n : 0;
      float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = f4zero();
      if (a.out_scale) sc = lda4(a.out_scale + nc);
      if (a.out_shift) sh = lda4(a.out_shift + nc);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int m = m_base + mt * 16 + j;
        const f32x4 c = acc[mt][nt];
        float4 o = fma4(make_float4(c[0], c[1], c[2], c[3]), sc, sh);
        if (a.out_act) o = act_apply4(o, a.out_act);
        if (a.res) o = add4(o, lda4(a.res + (int64_t)pm[mt] * a.ldres + nc));
        if (nok && m < Mtot) sta4(a.y + (int64_t)m * a.ldy + n, o);
      }
    } else {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int m = m_base + mt * 16 + j;
        const f32x4 c = acc[mt][nt];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool ok = (n + r < a.N) && (m < Mtot);
          const int nc = (n + r < a.N) ? n + r : 0;
          float v = c[r];
          if (a.out_scale) v *= a.out_scale[nc];
          if (a.out_shift) v += a.out_shift[nc];
          if (a.out_act) v = act_apply(v, a.out_act);
          if (a.res) v += lda1(a.res + (int64_t)pm[mt] * a.ldres + nc);
          if (ok) sta1(a.y + (int64_t)m * a.ldy + n + r, v);
        }
      }
    }
  }

}

// ---------------------------------------------------------------------------
// Pointwise (1x1, stride 1) fast path: persistent waves, the weight in LDS, X in flight.
// ---------------------------------------------------------------------------
// conv_fwd_kernel above gives every wave one tile: it loads weights (L1) and X operand by operand
// and all its latencies are hidden only by other resident waves - 2.6-3.6 TB/s on the large
// pointwise convs of the encoder.  Here a workgroup loads the whole weight [N][K] into LDS once
// and each of its waves walks a strided sequence of tiles (16*MT pixels x all N) as ONE flattened
// stream of k-blocks: the X operands of the next kPwD k-blocks (of this or the next tile) are
// always in flight while the current one is multiplied, the weight operand comes from LDS (no
// vmcnt dependency), and there is no workgroup barrier inside the loop.  Same MFMA order per
// accumulator as conv_fwd_kernel: results are bit-identical to it.  Statistics (STATS 1 / 2) are
// accumulated per wave in LDS over all its tiles: one row per workgroup of a grid that is a few
// hundred to two thousand workgroups instead of one row per 64*MT pixels.
#ifndef NASSEG_PW_D
#define NASSEG_PW_D 4
#endif
constexpr int kPwD = NASSEG_PW_D;

struct PwFwdPlan {
  int ok, nt, mt, grid;
  size_t lds;
};
// smallest pixel count that takes this path (nasseg_conv_pw_min_pixels: a tuning / testing knob)
#if NASSEG_FP32_ONLY
std::atomic<int64_t> g_pw_min_pixels{NASSEG_PW_MIN_PIXELS};
#endif
inline int pw_round_tiles(int t) {
  const int allowed[] = {1, 2, 3, 4, 6, 9, 12, 14};
  for (int v : allowed)
    if (t <= v) return v;
  return 0;
}
// a function of (pixels, N, K, kind of call) only: nasseg_conv_fwd_stats_blocks must predict the grid.
// mode 1: nasseg_conv_fwd, 2: nasseg_conv_bwd_data_bn.
// Measured against conv_fwd_kernel on the headline step (tools/gpu.sh flags, us old -> new): it wins
// where the reduction is long and the output narrow - 128->64 @128x256 51 -> 37, 192->32 46 -> 36,
// 64->64 @256x512 129 -> 106 (forward) / 121 -> 99 (backward-data), 224->64 216 -> 199, 144->24 102 -> 96,
// 32->32 @256x512 50 -> 43 - and loses on the expanding convs, where a tile is two k-blocks of
// multiplies and then 9-12 output vectors and their statistics: 24->144 98 -> 118, 32->192 36 -> 50,
// 64->128 36 -> 40, 24->24 27 -> 31; the exception is the backward-data call into 144 channels
// (24->144 @256x512: 247 -> 187), where the general kernel splits N over its waves.
inline PwFwdPlan pw_fwd_plan(int64_t M, int N, int K, int mode) {
  PwFwdPlan p = {};
  if (N <= 0 || K <= 0 || (N & 3) || (K & 3) || N > 224 || K > 256) return p;
  const int64_t knob = nasseg_conv_pw_min_pixels(-1);
  if (knob >= 0) {
    if (M < knob) return p;
  } else {
    const bool narrow = K >= 32 && N <= K && M >= 65536;
    const bool into144 = mode == 2 && N > 128 && N <= 144 && M >= 262144;
    if (!narrow && !into144) return p;
  }
  p.nt = pw_round_tiles(cdiv(N, 16));
  if (!p.nt) return p;
  p.mt = p.nt <= 6 ? 2 : 1;
  const int KP = (K + 15) & ~15;
  p.lds = ((size_t)p.nt * 16 * (KP + 4) + 2 * KP + 4 * 2 * p.nt * 16) * sizeof(float);
  if (p.lds > (size_t)(64 << 10)) return p;
  // resident workgroups per CU by registers (the smallest over the STATS variants of a tile count)
  int r = p.nt <= 2 ? 4 : (p.nt <= 4 ? 3 : (p.nt == 9 ? 3 : 2));
  const int by_lds = (int)((size_t)(160 << 10) / p.lds);
  if (r > by_lds) r = by_lds;
  const int64_t wg_tiles = cdiv64(cdiv64(M, 16 * p.mt), 4);
  p.grid = (int)(wg_tiles < 256LL * r ? wg_tiles : 256LL * r);
  p.ok = 1;
  return p;
}

template <int NT, int MT, int STATS>
__global__ __launch_bounds__(256) void conv_pw_kernel(FwdArgs a) {
  constexpr bool kSums = STATS == 1 || STATS == 2;
  constexpr int NPc = NT * 16;
  extern __shared__ float smem[];
  const int K = a.K, N = a.N;
  const int KP = (K + 15) & ~15, LSK = KP + 4, nkb = KP >> 4;
  float* wl = smem;                 // [NPc][LSK]: w[n][k], zero beyond N / K
  float* psc = wl + NPc * LSK;      // [KP] prologue scale | [KP] shift
  float* psh = psc + KP;
  float* sred = psh + KP;           // [4 waves][2][NPc]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, kg = lane >> 4;
  const int Mtot = a.g.B * a.g.Ho * a.g.Wo;
  const int ntiles = (Mtot + 16 * MT - 1) / (16 * MT);

  for (int it = tid; it < NPc * (KP >> 2); it += 256) {
    const int n = it / (KP >> 2), k = (it - n * (KP >> 2)) * 4;
    const float4 v = keep_if(lda4(a.w + (int64_t)(n < N ? n : 0) * K + (k < K ? k : 0)), n < N && k < K);
    *reinterpret_cast<float4*>(&wl[n * LSK + k]) = v;
  }
  const bool pro = a.in_scale || a.in_shift || a.in_act;
  for (int k = tid; k < KP; k += 256) {
    psc[k] = (a.in_scale && k < K) ? a.in_scale[k] : 1.f;
    psh[k] = (a.in_shift && k < K) ? a.in_shift[k] : 0.f;
  }
  for (int t = tid; t < 4 * 2 * NPc; t += 256) sred[t] = 0.f;
  __syncthreads();
  const ActSel pact = act_sel(a.in_act);
  float* my_red = sred + wave * 2 * NPc;

  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int wstride = gridDim.x * 4;
  // the loader runs kPwD k-blocks ahead of the multiplier through the same (tile, k-block) sequence
  int tl = blockIdx.x * 4 + wave, kl = 0;
  float4 ring[kPwD][MT];
  auto load_step = [&](float4* dst) {
    const int tc = tl < ntiles ? tl : ntiles - 1;  // (past the end: a valid address, never used)
    const int k = kl * 16 + kg * 4;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = tc * (16 * MT) + mt * 16 + j;
      dst[mt] = lda4(a.x + (int64_t)(m < Mtot ? m : Mtot - 1) * a.ldx + (k < K ? k : 0));
    }
    if (++kl == nkb) {
      kl = 0;
      tl += wstride;
    }
  };
#pragma unroll
  for (int d = 0; d < kPwD; ++d) load_step(ring[d]);

  int tc = blockIdx.x * 4 + wave, kc = 0;
  bool more = tc < ntiles;
  while (more) {
#pragma unroll
    for (int d = 0; d < kPwD; ++d) {
      if (tc >= ntiles) {
        more = false;
        break;
      }
      float4 bv[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) bv[mt] = ring[d][mt];
      load_step(ring[d]);
      const int k = kc * 16 + kg * 4;
      if (pro) {
        const float4 sc = *reinterpret_cast<const float4*>(&psc[k]);
        const float4 sh = *reinterpret_cast<const float4*>(&psh[k]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) bv[mt] = act_apply4(fma4(bv[mt], sc, sh), pact);
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) bv[mt] = keep_if(bv[mt], k < K);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const float4 av = *reinterpret_cast<const float4*>(&wl[(nt * 16 + j) * LSK + k]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          acc[mt][nt] = mfma16(av.x, bv[mt].x, acc[mt][nt]);
          acc[mt][nt] = mfma16(av.y, bv[mt].y, acc[mt][nt]);
          acc[mt][nt] = mfma16(av.z, bv[mt].z, acc[mt][nt]);
          acc[mt][nt] = mfma16(av.w, bv[mt].w, acc[mt][nt]);
        }
      }
      if (++kc < nkb) continue;
      // ---- the tile is complete: epilogue, lane holds pixel j of each subtile, channels 4*kg + {0..3} ----
      kc = 0;
      const int m_base = tc * (16 * MT);
      tc += wstride;
      int pm[MT];
      bool pok[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int m = m_base + mt * 16 + j;
        pok[mt] = m < Mtot;
        pm[mt] = pok[mt] ? m : Mtot - 1;
      }
      // the per-channel vectors of the epilogue are re-read (L1) per tile: hoisted out of the tile loop
      // they would occupy 8-16 registers per channel tile for the whole kernel
      const float* e_sc = STATS == 0 ? a.out_scale : a.b_scale;
      const float* e_sh = STATS == 0 ? a.out_shift : a.b_shift;
      const float* e_mu = a.b_mean;
      const float* e_is = a.b_invstd;
      asm volatile("" : "+s"(e_sc), "+s"(e_sh), "+s"(e_mu), "+s"(e_is));
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int n = nt * 16 + kg * 4;
        const bool nok = n < N;
        const int nc = nok ? n : 0;
        if (STATS == 0) {
          float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = f4zero();
          if (e_sc) sc = lda4(e_sc + nc);
          if (e_sh) sh = lda4(e_sh + nc);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const f32x4 c = acc[mt][nt];
            float4 o = make_float4(c[0], c[1], c[2], c[3]);
            if (e_sc || e_sh) o = fma4(o, sc, sh);
            if (a.out_act) o = act_apply4(o, a.out_act);
            if (a.res) o = add4(o, lda4(a.res + (int64_t)pm[mt] * a.ldres + nc));
            if (nok && pok[mt]) sta4(a.y + (int64_t)pm[mt] * a.ldy + n, o);
          }
        } else {
          float sx[4] = {0.f, 0.f, 0.f, 0.f}, sq[4] = {0.f, 0.f, 0.f, 0.f};
          float bsc[4] = {1.f, 1.f, 1.f, 1.f}, bsh[4] = {0.f, 0.f, 0.f, 0.f}, bmu[4] = {0.f, 0.f, 0.f, 0.f},
                bis[4] = {0.f, 0.f, 0.f, 0.f};
          if (STATS >= 2) {
            if (e_sc) { const float4 t = lda4(e_sc + nc); bsc[0] = t.x; bsc[1] = t.y; bsc[2] = t.z; bsc[3] = t.w; }
            if (e_sh) { const float4 t = lda4(e_sh + nc); bsh[0] = t.x; bsh[1] = t.y; bsh[2] = t.z; bsh[3] = t.w; }
          }
          if (STATS == 2) {
            const float4 t2 = lda4(e_mu + nc), t3 = lda4(e_is + nc);
            bmu[0] = t2.x; bmu[1] = t2.y; bmu[2] = t2.z; bmu[3] = t2.w;
            bis[0] = t3.x; bis[1] = t3.y; bis[2] = t3.z; bis[3] = t3.w;
          }
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            f32x4 c = acc[mt][nt];
            if (STATS >= 2) {
              const float4 z4 = lda4(a.bz + (int64_t)pm[mt] * a.ldbz + nc);
              const float zz[4] = {z4.x, z4.y, z4.z, z4.w};
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float g = c[r] * act_mask(fmaf(zz[r], bsc[r], bsh[r]), a.b_act);
                c[r] = g;
                if (STATS == 2) {
                  const float v = keep_if(g, pok[mt]);
                  sx[r] += v;
                  sq[r] = fmaf(v, (zz[r] - bmu[r]) * bis[r], sq[r]);
                }
              }
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float v = keep_if(c[r], pok[mt]);
                sx[r] += v;
                sq[r] = fmaf(v, v, sq[r]);
              }
            }
            if (nok && pok[mt]) sta4(a.y + (int64_t)pm[mt] * a.ldy + n, make_float4(c[0], c[1], c[2], c[3]));
          }
          if (kSums) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              sx[r] = row16_allsum(sx[r]);
              sq[r] = row16_allsum(sq[r]);
            }
            if (j == 0) {  // (this wave's own row of sred: no other lane touches these entries)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                my_red[n + r] += sx[r];
                my_red[NPc + n + r] += sq[r];
              }
            }
          }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // (keeps the loads of at most two channel tiles of z / res in registers at a time)
        if ((nt & 1) == 1) __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  if (kSums) {
    __syncthreads();
    for (int t = tid; t < NPc; t += 256) {
      if (t < N) {
        float* po = a.stats + (int64_t)blockIdx.x * 2 * N + t;
        po[0] = (sred[t] + sred[2 * NPc + t]) + (sred[4 * NPc + t] + sred[6 * NPc + t]);
        po[N] = (sred[NPc + t] + sred[3 * NPc + t]) + (sred[5 * NPc + t] + sred[7 * NPc + t]);
      }
    }
  }
}

// OIHW (N,K,kh,kw) -> [tap][N][K] (mode 0), [tap][K][N] (mode 1), [N][tap*K+k] (mode 2)
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
  } else if (mode == 1) {
    n = (int)(i % N);
    int64_t t = i / N;
    k = (int)(t % K);
    tap = (int)(t / K);
  } else {
    k = (int)(i % K);
    int64_t t = i / K;
    tap = (int)(t % ntaps);
    n = (int)(t / ntaps);
  }
  wp[i] = w[((int64_t)n * K + k) * ntaps + tap];
}

// ---------------------------------------------------------------------------
// 3x3 stride-1 convolution (dilation 1 ... 3) with the input tile staged in LDS.
// The gather kernel above re-reads X once per tap through L1/L2 (9x the tensor), which
// is what bounds the 64->19 class head and its backward-data; here a workgroup owns a
// th x tw patch of output pixels (at most 256: 8 x 32 unless lds3x3_tile finds a shape that
// costs the fullest CU less), stages the (th+2d) x (tw+2d) input patch of a 32-channel
// slice in LDS once (zero-filled outside the image and beyond K) and serves all nine taps
// from there: ds_read_b128 per lane, pixel stride padded to 36 floats so that the 16 pixel
// lanes of a k-group hit distinct banks.  MFMA operand mapping, weight layout (mode 0,
// [tap][N][K]) and epilogue as in conv_fwd_kernel; weights come from L1.
// Backward-data of such a conv is the same kernel on dy with flipped, role-swapped
// weights (pack kind 5) and pad' = d*(k-1) - pad.
// ---------------------------------------------------------------------------
constexpr int kLdsTH = 8, kLdsTW = 32, kLdsKC = 32, kLdsKS = kLdsKC + 4;
constexpr int kLdsMaxDil = 3;
constexpr int kLdsMaxIt = ((kLdsTH + 2 * kLdsMaxDil) * (kLdsTW + 2 * kLdsMaxDil) * (kLdsKC / 4) + 255) / 256;  // float4 of the patch per thread
constexpr int kLdsMaxPatch = kLdsMaxIt * 256 / (kLdsKC / 4);  // pixels of the largest patch a workgroup stages (544)

// The output tile of a workgroup: th x tw pixels, at most 256.  The pixel slot f of a tile is row f / tw, column f % tw
// (f < th * tw); subtile s (16 slots) belongs to wave s % 4, which multiplies its subtiles in rounds - a tile of 64
// pixels is ONE round of every wave, 256 pixels four.  512 workgroups run at once, two per CU; what a launch costs is
// (tools/tile_sweep.py on MI355X, 64 -> 64, microseconds): ~5 to launch, and per wave of up to 512 workgroups ~14
// (two slices staged and waited for, prologue, epilogue) + 8 per round where a CU holds one workgroup, 2 x 10.5 per round
// where it holds two.  Two things follow.
//   * Whole waves: 8 x 32 tiles cut a 16 x 81 x 81 map (the CVPR cells at 321 x 321) into 528 workgroups - 16 more than
//     run at once, a second wave for 3 % of the work - while 9 x 27 tiles make 432: 139 -> 107 us.
//   * Small maps want small tiles: 8 x 30 x 40 (the depth head's cells) in 8 x 32 tiles are 64 workgroups of four rounds
//     on 64 of the 256 CUs - 50 us for 0.7 GFLOP; in 64-pixel tiles 160-240 workgroups of one round - 27 us; 8 x 60 x 80
//     in 4 x 40 tiles (240 workgroups of three rounds) 54 -> 47, 16 x 41 x 41 in 6 x 21 tiles 53 -> 38.
// 8 x 32 stays wherever nothing beats it by 10 %.
struct Lds3Tile {
  int th, tw;
};
inline Lds3Tile lds3x3_tile(int B, int Ho, int Wo, int dil, int nt) {
  Lds3Tile best = {kLdsTH, kLdsTW};
#ifdef NASSEG_TUNE  // (tools/kbench_conv3x3.py: any tile by hand)
  if (const char* e = getenv("NASSEG_LDS3_TILE")) {
    int h = 0, w = 0;
    if (sscanf(e, "%d,%d", &h, &w) == 2 && h > 0 && w > 0 && h * w <= 256 && (h + 2 * dil) * (w + 2 * dil) <= kLdsMaxPatch)
      return Lds3Tile{h, w};
  }
#endif
  auto cost = [&](int th, int tw) {
    const int64_t wgs = (int64_t)B * cdiv(Ho, th) * cdiv(Wo, tw);
    const int rounds = (th * tw + 63) / 64;
    const double kf = 0.25 * nt;  // (channel tiles of MFMAs per round, relative to 64 outputs)
    const double alone = 14.0 + 8.0 * rounds * kf, shared = 14.0 + 2 * 10.5 * rounds * kf;
    const int64_t full = wgs / 512, rem = wgs % 512;
    return 5.0 + full * shared + (rem > 256 ? shared : (rem > 0 ? alone : 0.0)) +
           0.01 * (th + 2 * dil) * (tw + 2 * dil);  // (+ the patch: what decides between equals)
  };
  double bc = 0.9 * cost(best.th, best.tw);
  for (int tw = 8; tw <= 128 && tw <= Wo; ++tw) {
    for (int px = 64; px <= 256; px += 64) {
      const int th = px / tw < Ho ? px / tw : Ho;
      if (th < 1 || (th + 2 * dil) * (tw + 2 * dil) > kLdsMaxPatch) continue;
      const double c = cost(th, tw);
      if (c < bc) best = Lds3Tile{th, tw}, bc = c;
    }
  }
  return best;
}

// STATS == 1: per-workgroup sums of y and y^2 per output channel (the BatchNorm that follows a conv3x3 / conv3x3_dil3
// op of the CVPR cells, layer_factory.py:56-75) to stats[tile][2][N], tile = (b * tiles_y + ty) * tiles_x + tx.
// NV > 0 (the class heads: 19 = 16 + 3, 21 = 16 + 5 output channels): the NV channels behind the NT full tiles are
// not given a second, mostly empty MFMA tile (13 of 16 rows idle: 41 % of the kernel's MFMAs for N = 19) but are
// accumulated on the vector ALU from the operand registers the MFMAs read anyway - lane (pixel j, k-group kg) holds
// x[pixel][4 k] of every step, multiplies it with the NV weight rows' same 4 k (one address per k-group: a broadcast
// load) and keeps NV partial sums per subtile, which the four k-groups add up once at the end.  The FMAs issue in
// the shadow of the MFMAs.
// FLEX: the tile is a.th x a.tw (lds3x3_tile) instead of 8 x 32 - instantiated for the four-tile form and the two class
// heads only: with the tile a run-time value the four subtiles of a wave no longer sit at constant LDS offsets from
// each other, which costs the calls that stay on 8 x 32 tiles 3-6 % (64 -> 19 at 4 x 256 x 512: 176 -> 182 us).
template <int NT, bool VECN, bool VECK, int STATS = 0, int NV = 0, bool FLEX = false>
__global__ __launch_bounds__(256, 2) void conv3x3_lds_kernel(FwdArgs a) {
  static_assert(NV == 0 || (STATS == 0 && !VECN), "the vector-ALU channels come without statistics, stored one by one");
  extern __shared__ float tile[];
  __shared__ float sred3[STATS ? 4 : 1][2][STATS ? NT * 16 : 1];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int j = lane & 15;
  const int kg = lane >> 4;
  const int dil = a.g.dil;
  const int th = FLEX ? a.th : kLdsTH, tw = FLEX ? a.tw : kLdsTW;
  const int TR = th + 2 * dil, TC = tw + 2 * dil;
  const int b = blockIdx.z;
  const int oy0 = blockIdx.y * th, ox0 = blockIdx.x * tw;
  const int iy0 = oy0 - a.g.pad, ix0 = ox0 - a.g.pad;
  const int H = a.g.Hs, W = a.g.Ws;

  int wn[NT];
  bool wok[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = nt * 16 + j;
    wok[nt] = n < a.N;
    wn[nt] = wok[nt] ? n : a.N - 1;
  }
  // this lane's pixel in each of the wave's four 16-pixel subtiles: slot f = 64 w + 16 mt + j of the tile, row f / tw,
  // column f % tw (8 x 32: rows 2w, 2w+1; two halves); slots past the tile compute its pixel 0 and store nothing
  // (the output coordinates are worked out again where they are needed, after the main loop: eight registers held
  //  across it were what made the two-tile instantiations spill)
  auto slot_pixel = [&](int mt, int& oy, int& ox) {
    if (!FLEX) {
      oy = oy0 + 2 * wave + (mt >> 1);
      ox = ox0 + (mt & 1) * 16 + j;
      return ((2 * wave + (mt >> 1)) * TC + (mt & 1) * 16 + j) * kLdsKS;
    }
    const int f = (4 * mt + wave) * 16 + j;  // (subtile 4 mt + wave: round mt of this wave)
    const bool in_tile = f < th * tw;
    const int r = in_tile ? f / tw : 0;
    const int c = in_tile ? f - r * tw : 0;
    oy = in_tile ? oy0 + r : a.g.Ho;  // (a row beyond the map: masked like one)
    ox = ox0 + c;
    return (r * TC + c) * kLdsKS;
  };
  int toff[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    int oy, ox;
    toff[mt] = slot_pixel(mt, oy, ox);
  }
  // rounds this wave multiplies (wave-uniform): its subtiles 4 mt + wave that begin inside the tile
  int nmt = 4;
  if (FLEX) {
    const int rem = th * tw - 16 * wave;
    nmt = rem <= 0 ? 0 : (rem + 63) / 64;
    nmt = nmt > 4 ? 4 : nmt;
  }

  f32x4 acc[4][NT];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float pv[4][NV ? NV : 1];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int c = 0; c < (NV ? NV : 1); ++c) pv[mt][c] = 0.f;

  const act_t* xb = a.x + (int64_t)b * H * W * a.ldx;
  for (int kc0 = 0; kc0 < a.K; kc0 += kLdsKC) {
    if (kc0) __syncthreads();
    // The patch: (8+2d) x (32+2d) pixels x 8 float4 = 11 (d = 1) .. 14 (d = 2) float4 per thread.  ALL of a
    // thread's loads are issued before the first is stored (clamped addresses, masks instead of branches):
    // as a loop of load - store pairs this phase took ~15 us per 32-channel slice with one or two loads in
    // flight per thread, and the workgroups of a CU run it in lockstep (class head 64 -> 16: 140 us of which
    // 61 us are MFMA issue)
    {
      const int total = TR * TC * (kLdsKC / 4);
      float4 rv[kLdsMaxIt];
#pragma unroll
      for (int it = 0; it < kLdsMaxIt; ++it) {
        int idx = threadIdx.x + 256 * it;
        idx = idx < total ? idx : total - 1;
        const int q = idx % (kLdsKC / 4);
        const int p = idx / (kLdsKC / 4);
        const int pc = p % TC, pr = p / TC;
        const int iy = iy0 + pr, ix = ix0 + pc;
        const int iyc = iy < 0 ? 0 : (iy >= H ? H - 1 : iy), ixc = ix < 0 ? 0 : (ix >= W ? W - 1 : ix);
        const int k = kc0 + q * 4;
        const act_t* src = xb + ((int64_t)iyc * W + ixc) * a.ldx;
        rv[it] = load4<VECK>(src, VECK ? (k < a.K ? k : 0) : k, a.K);
      }
#pragma unroll
      for (int it = 0; it < kLdsMaxIt; ++it) {
        const int idx = threadIdx.x + 256 * it;
        const int q = idx % (kLdsKC / 4);
        const int p = idx / (kLdsKC / 4);
        const int pc = p % TC, pr = p / TC;
        const int iy = iy0 + pr, ix = ix0 + pc;
        const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
        const int k = kc0 + q * 4;
        float4 v = rv[it];
        if (!VECK) {
          v.y = keep_if(v.y, k + 1 < a.K);
          v.z = keep_if(v.z, k + 2 < a.K);
          v.w = keep_if(v.w, k + 3 < a.K);
        }
        if (idx < total) *reinterpret_cast<float4*>(&tile[p * kLdsKS + q * 4]) = keep_if(v, ok && k < a.K);
      }
    }
    __syncthreads();
    const int nks = (a.K - kc0 >= kLdsKC) ? kLdsKC / 16 : (a.K - kc0 + 15) / 16;
    const int rem_last = a.K - kc0 - (nks - 1) * 16;  // channels of the slice's last 16-wide step (>= 16: full)
    const int ntail = rem_last <= 8 ? (rem_last + 3) / 4 : 0;
    const int nvec = ntail ? nks - 1 : nks;
    // The weights of a 16-wide step are loaded one step AHEAD, into the other of two register sets: they come from
    // L1 / L2, and with the loads issued right before their MFMAs the two waves of a SIMD spent more time waiting for
    // them than issuing MFMAs (class head 64 -> 19: 61 us of MFMA issue in a 186 us kernel).  Masks are applied at
    // use - a select on a loaded value would wait for it at once.  Steps: it = tap * nvec + ks, nvec = 1 or 2.
    const int nsteps = 9 * nvec;
    auto step_of = [&](int it, int& tap, int& ks) {
      tap = nvec == 2 ? (it >> 1) : it;
      ks = nvec == 2 ? (it & 1) : 0;
    };
    auto load_w = [&](int it, float4 (&av)[NT], float4 (&xv)[NV ? NV : 1]) {
      int tap, ks;
      step_of(it, tap, ks);
      const int k = kc0 + ks * 16 + kg * 4;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) av[nt] = load4<VECK>(a.w + ((int64_t)tap * a.N + wn[nt]) * a.K, k, a.K);
#pragma unroll
      for (int c = 0; c < NV; ++c) xv[c] = load4<VECK>(a.w + ((int64_t)tap * a.N + NT * 16 + c) * a.K, k, a.K);
    };
    auto do_step = [&](int it, const float4 (&avr)[NT], const float4 (&xvr)[NV ? NV : 1]) {
      int tap, ks;
      step_of(it, tap, ks);
      const int ty = tap / 3, tx = tap - ty * 3;
      const int tsh = (ty * dil * TC + tx * dil) * kLdsKS;
      const int kl = ks * 16 + kg * 4;
      const int k = kc0 + kl;
      float4 bv[4], av[NT], xv[NV ? NV : 1];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
        if (!FLEX || mt < nmt) bv[mt] = *reinterpret_cast<const float4*>(&tile[toff[mt] + tsh + kl]);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        float4 v = avr[nt];
        if (!VECK) {
          v.y = keep_if(v.y, k + 1 < a.K);
          v.z = keep_if(v.z, k + 2 < a.K);
          v.w = keep_if(v.w, k + 3 < a.K);
        }
        av[nt] = keep_if(v, wok[nt] && k < a.K);
      }
#pragma unroll
      for (int c = 0; c < NV; ++c) xv[c] = keep_if(xvr[c], k < a.K);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        if (FLEX && mt >= nmt) continue;  // (wave-uniform: a round of subtiles beyond the tile)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          acc[mt][nt] = mfma16(av[nt].x, bv[mt].x, acc[mt][nt]);
          acc[mt][nt] = mfma16(av[nt].y, bv[mt].y, acc[mt][nt]);
          acc[mt][nt] = mfma16(av[nt].z, bv[mt].z, acc[mt][nt]);
          acc[mt][nt] = mfma16(av[nt].w, bv[mt].w, acc[mt][nt]);
        }
#pragma unroll
        for (int c = 0; c < NV; ++c) {
          pv[mt][c] = fmaf(bv[mt].x, xv[c].x, pv[mt][c]);
          pv[mt][c] = fmaf(bv[mt].y, xv[c].y, pv[mt][c]);
          pv[mt][c] = fmaf(bv[mt].z, xv[c].z, pv[mt][c]);
          pv[mt][c] = fmaf(bv[mt].w, xv[c].w, pv[mt][c]);
        }
      }
    };
    if (nsteps) {  // (uniform)
      float4 avA[NT], avB[NT], xvA[NV ? NV : 1], xvB[NV ? NV : 1];
      load_w(0, avA, xvA);
      int it = 0;
#pragma unroll 1
      for (; it + 1 < nsteps; it += 2) {
        load_w(it + 1, avB, xvB);
        __builtin_amdgcn_sched_barrier(0);
        do_step(it, avA, xvA);
        __builtin_amdgcn_sched_barrier(0);
        load_w(it + 2 < nsteps ? it + 2 : it + 1, avA, xvA);  // (the last pair re-loads step it + 1: in range, unused)
        __builtin_amdgcn_sched_barrier(0);
        do_step(it + 1, avB, xvB);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (it < nsteps) do_step(it, avA, xvA);  // (9 steps: the odd one, loaded by the last pair)
    }
    // a short tail of the slice (the class head's backward-data reduces over 19 = 16 + 3 channels per
    // tap): with the vector mapping k = 4*kg + component each of the four MFMAs of a 16-wide step would
    // carry one useful k-slot in four; here lane group kg takes channel 4*step + kg, so ceil(rem / 4)
    // MFMAs do.  (Loops of their own after the vector steps: as a branch inside them the two paths'
    // accumulators were copied through 64 v_accvgpr_mov per step.)
    for (int tap = 0; tap < (ntail ? 9 : 0); ++tap) {
      const int ty = tap / 3, tx = tap - ty * 3;
      const int tsh = (ty * dil * TC + tx * dil) * kLdsKS;
      for (int st = 0; st < ntail; ++st) {
        const int kls = nvec * 16 + st * 4 + kg;
        const int kk = kc0 + kls;
        float bs[4], as[NT];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) bs[mt] = (!FLEX || mt < nmt) ? tile[toff[mt] + tsh + kls] : 0.f;  // (zero beyond K)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          as[nt] = keep_if(a.w[((int64_t)tap * a.N + wn[nt]) * a.K + (kk < a.K ? kk : 0)], wok[nt] && kk < a.K);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          if (FLEX && mt >= nmt) continue;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma16(as[nt], bs[mt], acc[mt][nt]);
        }
#pragma unroll
        for (int c = 0; c < NV; ++c) {
          const float wv = keep_if(a.w[((int64_t)tap * a.N + NT * 16 + c) * a.K + (kk < a.K ? kk : 0)], kk < a.K);
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) pv[mt][c] = fmaf(bs[mt], wv, pv[mt][c]);
        }
      }
    }
  }
  if constexpr (NV > 0) {
    // the four k-groups' partial sums of the vector-ALU channels: lanes j, j + 16, j + 32, j + 48 -> every lane
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int c = 0; c < NV; ++c) {
        float v = pv[mt][c];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        pv[mt][c] = v;
      }
  }

  if constexpr (STATS == 1) {
    // over the wave's four subtiles in registers, its 16 pixel lanes by DPP, the four waves through LDS
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      float sx[4] = {0.f, 0.f, 0.f, 0.f}, sq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        int oy, ox;
        slot_pixel(mt, oy, ox);
        const bool pok = oy < a.g.Ho && ox < a.g.Wo;
        const f32x4 c = acc[mt][nt];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = keep_if(c[r], pok);
          sx[r] += v;
          sq[r] = fmaf(v, v, sq[r]);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        sx[r] = row16_allsum(sx[r]);
        sq[r] = row16_allsum(sq[r]);
      }
      if (j == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          sred3[wave][0][nt * 16 + kg * 4 + r] = sx[r];
          sred3[wave][1][nt * 16 + kg * 4 + r] = sq[r];
        }
      }
    }
    __syncthreads();
    const int64_t row = ((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    for (int t = threadIdx.x; t < NT * 16; t += 256) {
      if (t < a.N) {
        float* po = a.stats + row * 2 * a.N + t;
        po[0] = (sred3[0][0][t] + sred3[1][0][t]) + (sred3[2][0][t] + sred3[3][0][t]);
        po[a.N] = (sred3[0][1][t] + sred3[1][1][t]) + (sred3[2][1][t] + sred3[3][1][t]);
      }
    }
  }
  // epilogue: lane holds pixel j of each subtile, channels nt*16 + 4*kg + {0..3}
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    int oy, ox;
    slot_pixel(mt, oy, ox);
    const bool pok = oy < a.g.Ho && ox < a.g.Wo;
    const int64_t m = ((int64_t)b * a.g.Ho + (oy < a.g.Ho ? oy : a.g.Ho - 1)) * a.g.Wo +
                      (ox < a.g.Wo ? ox : a.g.Wo - 1);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = nt * 16 + kg * 4;
      const f32x4 c = acc[mt][nt];
      if (VECN) {
        const bool nok = n < a.N;
        const int nc = nok ? n : 0;
        float4 o = make_float4(c[0], c[1], c[2], c[3]);
        if (a.out_scale) o = fma4(o, lda4(a.out_scale + nc), f4zero());
        if (a.out_shift) o = add4(o, lda4(a.out_shift + nc));
        if (a.out_act) o = act_apply4(o, a.out_act);
        if (a.res) o = add4(o, lda4(a.res + m * a.ldres + nc));
        if (nok && pok) sta4(a.y + m * a.ldy + n, o);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool ok = (n + r < a.N) && pok;
          const int nc = (n + r < a.N) ? n + r : 0;
          float v = c[r];
          if (a.out_scale) v *= a.out_scale[nc];
          if (a.out_shift) v += a.out_shift[nc];
          if (a.out_act) v = act_apply(v, a.out_act);
          if (a.res) v += lda1(a.res + m * a.ldres + nc);
          if (ok) sta1(a.y + m * a.ldy + n + r, v);
        }
      }
    }
    if constexpr (NV > 0) {
      // k-group kg stores channel NT*16 + kg (and + 4 + kg: NV <= 8) of its pixel
#pragma unroll
      for (int h = 0; h < (NV + 3) / 4; ++h) {
        float v = 0.f;
#pragma unroll
        for (int c = 0; c < NV; ++c)
          if (c / 4 == h) v = (kg == (c & 3)) ? pv[mt][c] : v;
        const int n = NT * 16 + h * 4 + kg;
        const bool ok = n < a.N && h * 4 + kg < NV && pok;
        const int nc = ok ? n : 0;
        if (a.out_scale) v *= a.out_scale[nc];
        if (a.out_shift) v += a.out_shift[nc];
        if (a.out_act) v = act_apply(v, a.out_act);
        if (a.res) v += lda1(a.res + m * a.ldres + nc);
        if (ok) sta1(a.y + m * a.ldy + n, v);
      }
    }
  }
}

// FX: the tile is the planner's (lds3x3_tile) and not 8 x 32
template <int NT, bool FX>
int launch_lds3x3_t(const FwdArgs& a, bool vecn, bool veck, bool stats, hipStream_t s) {
  const int dil = a.g.dil;
  const size_t lds = (size_t)(a.th + 2 * dil) * (a.tw + 2 * dil) * kLdsKS * sizeof(float);
  dim3 grid(cdiv(a.g.Wo, a.tw), cdiv(a.g.Ho, a.th), a.g.B);
  if (lds > (size_t)(64 << 10)) {  // (dilation 3: 76.6 KB)
    static std::atomic<int> raised{0};
    if (!raised.load()) {
#define ATTR3(V_, K_, S_)                                                                                    \
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_lds_kernel<NT, V_, K_, S_, 0, FX>),       \
                            hipFuncAttributeMaxDynamicSharedMemorySize, 96 << 10)
      ATTR3(true, true, 0); ATTR3(true, false, 0); ATTR3(false, true, 0); ATTR3(false, false, 0);
      ATTR3(true, true, 1); ATTR3(true, false, 1);
#undef ATTR3
      raised.store(1);
    }
  }
#define GO3(V_, K_, S_) hipLaunchKernelGGL((conv3x3_lds_kernel<NT, V_, K_, S_, 0, FX>), grid, dim3(256), lds, s, a)
  if (stats) { if (veck) GO3(true, true, 1); else GO3(true, false, 1); }  // (statistics need N % 4 == 0)
  else if (vecn) { if (veck) GO3(true, true, 0); else GO3(true, false, 0); }
  else { if (veck) GO3(false, true, 0); else GO3(false, false, 0); }
#undef GO3
  NASSEG_LAUNCH_CHECK("conv3x3_lds_kernel");
  return NASSEG_OK;
}

inline bool lds3x3_is_default(const Lds3Tile& t) { return t.th == kLdsTH && t.tw == kLdsTW; }

template <int NT>
int launch_lds3x3(const FwdArgs& a0, bool vecn, bool veck, bool stats, hipStream_t s) {
  FwdArgs a = a0;
  // (the four-tile form picks its tile: nasseg_conv_fwd_stats_rows counts the rows under the same condition)
  const Lds3Tile t = NT == 4 ? lds3x3_tile(a.g.B, a.g.Ho, a.g.Wo, a.g.dil, 4) : Lds3Tile{kLdsTH, kLdsTW};
  a.th = t.th;
  a.tw = t.tw;
  if constexpr (NT == 4) {
    if (!lds3x3_is_default(t)) return launch_lds3x3_t<NT, true>(a, vecn, veck, stats, s);
  }
  return launch_lds3x3_t<NT, false>(a, vecn, veck, stats, s);
}

// NT full tiles on the matrix cores + NV channels on the vector ALU (N = 16 * NT + NV, K % 4 == 0, no statistics)
template <int NT, int NV, bool FX>
int launch_lds3x3_nv_t(const FwdArgs& a, hipStream_t s) {
  const int dil = a.g.dil;
  const size_t lds = (size_t)(a.th + 2 * dil) * (a.tw + 2 * dil) * kLdsKS * sizeof(float);
  dim3 grid(cdiv(a.g.Wo, a.tw), cdiv(a.g.Ho, a.th), a.g.B);
  if (lds > (size_t)(64 << 10))
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_lds_kernel<NT, false, true, 0, NV, FX>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 96 << 10);
  hipLaunchKernelGGL((conv3x3_lds_kernel<NT, false, true, 0, NV, FX>), grid, dim3(256), lds, s, a);
  NASSEG_LAUNCH_CHECK("conv3x3_lds_kernel");
  return NASSEG_OK;
}
template <int NT, int NV>
int launch_lds3x3_nv(const FwdArgs& a0, hipStream_t s) {
  FwdArgs a = a0;
  constexpr bool kPick = NV == 3 || NV == 5;  // (19 and 21 classes)
  const Lds3Tile t = kPick ? lds3x3_tile(a.g.B, a.g.Ho, a.g.Wo, a.g.dil, 1) : Lds3Tile{kLdsTH, kLdsTW};
  a.th = t.th;
  a.tw = t.tw;
  if constexpr (kPick) {
    if (!lds3x3_is_default(t)) return launch_lds3x3_nv_t<NT, NV, true>(a, s);
  }
  return launch_lds3x3_nv_t<NT, NV, false>(a, s);
}

// Several weight tensors re-packed by ONE launch (a chain of convolutions packs the
// forward and backward-data layouts of all its weights together: launches, not bytes, are
// what small layers pay for).  kind 0..2 = dense modes above; 3 = depthwise [tap][C],
// 4 = depthwise flipped by 180 degrees (N = C, K = 1); 5 = mode 1 with the taps flipped
// (backward-data of a stride-1 conv computed as a forward conv over dy).
struct PackDesc {
  const float* w;
  float* wp;
  int N, K, ntaps, kind;
  int Ksrc, koff;  // the K channels packed are [koff, koff+K) of a source with Ksrc input channels
};
constexpr int kPackMax = 96;  // (3.8 KB of kernel arguments)
struct PackTable {
  PackDesc d[kPackMax];
};

__global__ void pack_multi_kernel(PackTable t) {
  const PackDesc d = t.d[blockIdx.y];
  const int total = d.ntaps * d.N * d.K;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    int tap, n, k;
    if (d.kind == 0) {
      k = i % d.K;
      const int q = i / d.K;
      n = q % d.N;
      tap = q / d.N;
    } else if (d.kind == 1 || d.kind == 5) {
      n = i % d.N;
      const int q = i / d.N;
      k = q % d.K;
      tap = q / d.K;
      if (d.kind == 5) tap = d.ntaps - 1 - tap;
    } else if (d.kind == 2) {
      k = i % d.K;
      const int q = i / d.K;
      tap = q % d.ntaps;
      n = q / d.ntaps;
    } else {
      n = i % d.N;
      k = 0;
      tap = i / d.N;
      if (d.kind == 4) tap = d.ntaps - 1 - tap;
    }
    d.wp[i] = d.w[((int64_t)n * d.Ksrc + d.koff + k) * d.ntaps + tap];
  }
}

// maps up to this many pixels take four k-steps per round trip (conv_fwd_kernel's KU_) and the 16-pixel workgroups
// of conv_small_ws (32768 measured worse on the teacher's 256-channel 3x3 convs at 16 x 32 x 32: 64 x K x 9 weights
// per 16 pixels)
// (16384 measured on one box: config-5 shape bf16 replayed 775 -> 767, the KD teacher 1123 -> 1087, CVPR 321x321 and
//  task0 unchanged)
#ifndef NASSEG_DEEPK_MAX
#define NASSEG_DEEPK_MAX 8192
#endif
constexpr int64_t kDeepKMaxPixels = NASSEG_DEEPK_MAX;
struct Mode {
  int km;
  bool gather, pro, vecn, epi;
  int stats;  // 0 none, 1 forward BN statistics, 2 BN-backward statistics, 3 act' mask only
};

template <int MT, int NT, bool WS = false>
int launch_one(const FwdArgs& a, const Mode& md, hipStream_t s) {
  const int64_t Mtot = (int64_t)a.g.B * a.g.Ho * a.g.Wo;
  dim3 grid((unsigned)cdiv64(Mtot, (WS ? 16 : 64) * MT), cdiv(a.N, (WS ? 64 : 16) * NT), 1);
  const bool deep = Mtot <= kDeepKMaxPixels && g_conv_deep_k.load() != 0;
#define GO_(KM_, G_, P_, V_, KUX)                                                                       \
  do {                                                                                            \
    if (md.stats == 2) {                                                                          \
      if constexpr ((V_) && !(P_) && KM_ != KM_FLAT)                                              \
        hipLaunchKernelGGL((conv_fwd_kernel<MT, NT, KM_, G_, P_, V_, false, 2, WS, KUX>), grid,        \
                           dim3(256), 0, s, a);                                                   \
    } else if (md.stats == 3) {                                                                   \
      if constexpr ((V_) && !(P_) && KM_ != KM_FLAT)                                              \
        hipLaunchKernelGGL((conv_fwd_kernel<MT, NT, KM_, G_, P_, V_, false, 3, WS, KUX>), grid,        \
                           dim3(256), 0, s, a);                                                   \
    } else if (md.stats) {                                                                        \
      if constexpr (V_)                                                                           \
        hipLaunchKernelGGL((conv_fwd_kernel<MT, NT, KM_, G_, P_, V_, false, 1, WS, KUX>), grid,        \
                           dim3(256), 0, s, a);                                                   \
    } else if (md.epi || !(V_)) {                                                                 \
      hipLaunchKernelGGL((conv_fwd_kernel<MT, NT, KM_, G_, P_, V_, true, 0, WS, KUX>), grid,           \
                         dim3(256), 0, s, a);                                                     \
    } else {                                                                                      \
      hipLaunchKernelGGL((conv_fwd_kernel<MT, NT, KM_, G_, P_, V_, false, 0, WS, KUX>), grid,          \
                         dim3(256), 0, s, a);                                                     \
    }                                                                                             \
  } while (0)
#define GO(KM_, G_, P_, V_)                                   \
  do {                                                        \
    if constexpr (MT == 1 && KM_ == KM_VEC) {                 \
      if (deep) GO_(KM_, G_, P_, V_, 4);                      \
      else GO_(KM_, G_, P_, V_, NASSEG_CONV_KU);              \
    } else {                                                  \
      GO_(KM_, G_, P_, V_, NASSEG_CONV_KU);                   \
    }                                                         \
  } while (0)
  if (md.km == KM_FLAT) {
    if constexpr (NT <= 4 && !WS) {
      if (md.vecn) GO(KM_FLAT, true, false, true);
      else GO(KM_FLAT, true, false, false);
    } else {
      return nasseg_fail(NASSEG_ERR_UNSUPPORTED, "conv_fwd: flat path supports N <= 64");
    }
  } else if (!md.vecn) {
    // class-logit heads (N = 19, 21, 11, 1): always through the gather kernels
    if constexpr (NT <= 2 && !WS) {
      if (md.km == KM_VEC) GO(KM_VEC, true, false, false);
      else GO(KM_SCALAR, true, false, false);
    } else {
      return nasseg_fail(NASSEG_ERR_UNSUPPORTED, "conv_fwd: N %% 4 != 0 needs N <= 32");
    }
  } else if (md.km == KM_SCALAR) {
    GO(KM_SCALAR, true, false, true);
  } else if (md.gather) {
    GO(KM_VEC, true, false, true);
  } else if (md.pro) {
    GO(KM_VEC, false, true, true);
  } else {
    GO(KM_VEC, false, false, true);
  }
#undef GO
#undef GO_
  NASSEG_LAUNCH_CHECK("conv_fwd_kernel");
  return NASSEG_OK;
}

// pixel-tile height per wave: as many 16-pixel subtiles as still leave >= ~6
// workgroups per CU (the op is latency-bound below that)
inline int pick_mt(int64_t Mtot, int tiles) {
  if (tiles > 4) return Mtot >= 16 * 4 * 1536 ? 4 : (Mtot >= 16 * 2 * 1536 ? 2 : 1);  // wave-split N
  return Mtot >= 64 * 4 * 1536 ? 4 : (Mtot >= 64 * 2 * 1536 ? 2 : 1);
}
template <int NT, bool WS = false>
int launch_small(const FwdArgs& a, const Mode& md, hipStream_t s) {
  const int mt = pick_mt((int64_t)a.g.B * a.g.Ho * a.g.Wo, WS ? 5 : NT);
  if (mt == 4) return launch_one<4, NT, WS>(a, md, s);
  if (mt == 2) return launch_one<2, NT, WS>(a, md, s);
  return launch_one<1, NT, WS>(a, md, s);
}

template <int NT, int MT>
int launch_pw_nt(const FwdArgs& a, const PwFwdPlan& p, int stats, hipStream_t s) {
  const dim3 grid(p.grid), block(256);
  switch (stats) {
    case 0: hipLaunchKernelGGL((conv_pw_kernel<NT, MT, 0>), grid, block, p.lds, s, a); break;
    case 1: hipLaunchKernelGGL((conv_pw_kernel<NT, MT, 1>), grid, block, p.lds, s, a); break;
    case 2: hipLaunchKernelGGL((conv_pw_kernel<NT, MT, 2>), grid, block, p.lds, s, a); break;
    default: hipLaunchKernelGGL((conv_pw_kernel<NT, MT, 3>), grid, block, p.lds, s, a); break;
  }
  NASSEG_LAUNCH_CHECK("conv_pw_kernel");
  return NASSEG_OK;
}
int launch_pw(const FwdArgs& a, const PwFwdPlan& p, int stats, hipStream_t s) {
  switch (p.nt) {
    case 1: return launch_pw_nt<1, 2>(a, p, stats, s);
    case 2: return launch_pw_nt<2, 2>(a, p, stats, s);
    case 3: return launch_pw_nt<3, 2>(a, p, stats, s);
    case 4: return launch_pw_nt<4, 2>(a, p, stats, s);
    case 6: return launch_pw_nt<6, 2>(a, p, stats, s);
    case 9: return launch_pw_nt<9, 1>(a, p, stats, s);
    case 12: return launch_pw_nt<12, 1>(a, p, stats, s);
    default: return launch_pw_nt<14, 1>(a, p, stats, s);
  }
}

// SMALL maps (at most kDeepKMaxPixels pixels): a workgroup owns 16 pixels and its four waves 16 output channels each,
// whatever N - 16 x 11 x 11 pixels into 64 channels are 31 workgroups of 64 x 64 tiles otherwise, 124 waves on 1024
// SIMDs each multiplying for 8 us (3x3, 64 -> 64: 33 us; 484 waves and 11 us this way).  A function of (pixels, N, K)
// alone: nasseg_conv_fwd_stats_blocks must predict the rows.  K >= 8 keeps the flat small-K form out of it.
// N * K <= 192 K: a workgroup reads 64 x K weights for its 16 pixels - the KD teacher's 1024 -> 1024 ... 2048 -> 2048
// convs on 16 x 8 x 8 ... 16 x 16 x 16 maps re-read megabytes of weights per workgroup this way (teacher inference
// 1127 -> 956 images/s before this bound); MobileNetV2's 960 -> 160 (47 instead of 78 us) is inside.
inline bool conv_small_ws(int64_t Mtot, int N, int K) {
  return g_conv_deep_k.load() != 0 && Mtot <= kDeepKMaxPixels && (N & 3) == 0 && (K & 3) == 0 && K >= 8 &&
         (int64_t)N * K <= 196608;
}

// geometries of the LDS-tiled 3x3 kernel (conv3x3_lds_kernel); K * 9 > 64 keeps the flat small-K form out
inline bool lds3x3_geometry(int B, int Ho, int Wo, int N, int K, int kh, int kw, int stride, int pad, int dil) {
  return kh == 3 && kw == 3 && stride == 1 && dil >= 1 && dil <= kLdsMaxDil && K * 9 > 64 && cdiv(N, 16) <= 4 &&
         Wo >= kLdsTW && Ho >= kLdsTH && B <= 65535 && pad >= 0 && pad <= 2 * dil;
}

inline int fwd_pack_mode(int K, int kh, int kw) { return (kh * kw > 1 && kh * kw * K <= 64) ? 2 : 0; }

int conv_dispatch(FwdArgs& a, int stats_mode, hipStream_t s) {
  const int K = a.K, N = a.N;
  const ConvGeom& g = a.g;
  NASSEG_REQUIRE(g.B > 0 && g.Hs > 0 && g.Ws > 0 && g.Ho > 0 && g.Wo > 0, "conv_fwd: bad geometry");
  NASSEG_REQUIRE(K > 0 && N > 0 && a.ldx >= K && a.ldy >= N, "conv_fwd: bad channels K=%d N=%d", K, N);
  NASSEG_REQUIRE((int64_t)g.B * g.Hs * g.Ws < 2147483647LL && (int64_t)g.B * g.Ho * g.Wo < 2147483647LL,
                 "conv_fwd: too many pixels");
  Mode md;
  md.km = (((K & 3) == 0) && ((a.ldx & 3) == 0)) ? KM_VEC : KM_SCALAR;
  md.pro = a.in_scale || a.in_shift || a.in_act;
  md.gather = !(g.kh == 1 && g.kw == 1 && g.stride == 1 && g.pad == 0 && g.Hs == g.Ho && g.Ws == g.Wo);
  md.vecn = ((N & 3) == 0) && ((a.ldy & 3) == 0) && (!a.res || (a.ldres & 3) == 0);
  md.epi = a.out_scale || a.out_shift || a.out_act || a.res;
  md.stats = stats_mode;
  NASSEG_REQUIRE(!md.stats || (md.vecn && !md.epi),
                 "conv_fwd: statistics need N %% 4 == 0 and no output epilogue");
  NASSEG_REQUIRE(md.stats < 2 || !md.pro, "conv_bwd_data_bn: no input prologue on this path");
  if (!g.transposed && fwd_pack_mode(K, g.kh, g.kw) == 2) md.km = KM_FLAT;
  NASSEG_REQUIRE(!md.pro || (md.km == KM_VEC && !md.gather && md.vecn),
                 "conv_fwd: the input prologue needs a pointwise conv with K %% 4 == 0, N %% 4 == 0");
  NASSEG_REQUIRE(a.y || (md.stats == 1 && !md.gather && md.km != KM_FLAT),
                 "conv_fwd: y == NULL needs statistics rows and a pointwise conv");
  const int tiles = cdiv(N, 16);
  if (!md.gather && md.km != KM_FLAT) {
    // the N-split persistent kernel (conv_pwn.hip) where its plan says so
    const PwnPlan pn = nasseg_internal_pwn_plan((int64_t)g.B * g.Ho * g.Wo, N, K, stats_mode >= 2 ? 2 : 1);
    if (pn.ok) {
      const bool aligned = md.km == KM_VEC && md.vecn && (stats_mode < 2 || (a.ldbz & 3) == 0);
      NASSEG_REQUIRE(aligned || (md.stats != 1 && md.stats != 2),
                     "conv_fwd: the pointwise statistics path needs channel strides that are multiples of 4");
      if (aligned) return NASSEG_INTERNAL(pwn_launch)(a, pn, md.stats, s);
    }
    NASSEG_REQUIRE(a.y, "conv_fwd: y == NULL (statistics only) is served by the N-split pointwise kernel alone");
    const PwFwdPlan pw = pw_fwd_plan((int64_t)g.B * g.Ho * g.Wo, N, K, stats_mode >= 2 ? 2 : 1);
    if (pw.ok) {
      const bool aligned = md.km == KM_VEC && md.vecn;
      // (statistics rows were sized for this kernel's grid: with them there is no falling back)
      NASSEG_REQUIRE(aligned || (md.stats != 1 && md.stats != 2),
                     "conv_fwd: the pointwise statistics path needs channel strides that are multiples of 4");
      if (aligned) return launch_pw(a, pw, md.stats, s);
    }
  }
  // 3x3, stride 1, dilation <= 3, maps at least one tile large: input patch staged in LDS (with the forward
  // statistics rows when asked for: nasseg_conv_fwd_stats_rows counts them under the same condition)
  if (!g.transposed && lds3x3_geometry(g.B, g.Ho, g.Wo, N, K, g.kh, g.kw, g.stride, g.pad, g.dil) && md.km != KM_FLAT &&
      !md.pro && (md.stats == 0 || (md.stats == 1 && md.vecn))) {
    const bool veck = md.km == KM_VEC;
    // 17 ... 21 output channels (the class heads' 19 and 21): 16 on the matrix cores, the rest on the vector ALU
    if (NASSEG_LDS3X3_VALU_TAIL && veck && md.stats == 0 && N > 16 && N <= 21) {
      switch (N - 16) {
        case 1: return launch_lds3x3_nv<1, 1>(a, s);
        case 2: return launch_lds3x3_nv<1, 2>(a, s);
        case 3: return launch_lds3x3_nv<1, 3>(a, s);
        case 4: return launch_lds3x3_nv<1, 4>(a, s);
        default: return launch_lds3x3_nv<1, 5>(a, s);
      }
    }
    if (tiles <= 1) return launch_lds3x3<1>(a, md.vecn, veck, md.stats == 1, s);
    if (tiles == 2) return launch_lds3x3<2>(a, md.vecn, veck, md.stats == 1, s);
    // (33 ... 48 channels run on the four-tile form too: with the weights of a step loaded one step ahead the
    //  three-tile instantiation does not fit two waves per SIMD without spilling, and no layer of the reference has it)
    return launch_lds3x3<4>(a, md.vecn, veck, md.stats == 1, s);
  }
  // (without statistics rows to keep in step with the dispatch is free to look at the taps as well: a workgroup of the
  //  small-map form reads 64 x K x taps weights for 16 pixels - past 1024 x 64 of them the large-map tiling wins, the KD
  //  teacher's 256 -> 256 3x3 convs at 16 x 16 x 16)
  const bool rows_asked = md.stats == 1 || md.stats == 2;
  if (conv_small_ws((int64_t)g.B * g.Ho * g.Wo, N, K) && (rows_asked || K * g.kh * g.kw <= 1024)) {
    if (md.km == KM_VEC && md.vecn) return launch_one<1, 1, true>(a, md, s);
    // (statistics rows were counted for that kernel's 16-pixel workgroups)
    NASSEG_REQUIRE(md.stats != 1 && md.stats != 2,
                   "conv_fwd: statistics on a small map need channel strides that are multiples of 4");
  }
  if (tiles <= 1) return launch_small<1>(a, md, s);
  if (tiles == 2) return launch_small<2>(a, md, s);
  if (tiles == 3) return launch_small<3>(a, md, s);
  if (tiles == 4) return launch_small<4>(a, md, s);
  // N > 64: the four waves of a workgroup split N (2, 3 or 4 tiles each)
  if (tiles <= 8) return launch_small<2, true>(a, md, s);
  if (tiles <= 12) return launch_small<3, true>(a, md, s);
  return launch_small<4, true>(a, md, s);  // N > 256 is covered by grid.y
}

}  // namespace

extern "C" {

#if NASSEG_FP32_ONLY
// mode 0: [tap][N][K]; mode 1: backward-data [tap][K][N]; mode 2: flat [N][tap*K+k]
int nasseg_conv_pack_weight(const float* w, float* wp, int N, int K, int kh, int kw, int mode,
                            void* stream) {
  NASSEG_REQUIRE(N > 0 && K > 0 && kh > 0 && kw > 0, "conv_pack_weight: bad shape");
  NASSEG_REQUIRE(mode >= 0 && mode <= 2, "conv_pack_weight: bad mode %d", mode);
  const int64_t total = (int64_t)N * K * kh * kw;
  hipLaunchKernelGGL(conv_pack_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, w, wp, N, K, kh * kw, mode);
  NASSEG_LAUNCH_CHECK("conv_pack_weight");
  return NASSEG_OK;
}
#endif  // NASSEG_FP32_ONLY

#if NASSEG_FP32_ONLY
// count tensors in one launch.  w[i] / wp[i]: device pointers (host arrays of pointers);
// dims[7*i..] = N, K, kh, kw, kind, Ksrc, koff: the K input channels [koff, koff+K) of a source
// weight with Ksrc input channels are packed (Ksrc = 0 means the whole weight: Ksrc = K, koff = 0
// - a slice lets ConcatReduce's 1x1 over a concatenation run as two convs); kind 0/1/2 = the dense modes of
// nasseg_conv_pack_weight and 3 / 4 = depthwise (C = N, K = 1) plain / flipped, i.e.
// nasseg_dw_pack_weight(flip = 0 / 1); 5 = [K_fwd-major rows] mode 1 with flipped taps.
int nasseg_pack_weights(int count, const float* const* w, float* const* wp, const int* dims,
                        void* stream) {
  NASSEG_REQUIRE(count >= 0 && (count == 0 || (w && wp && dims)), "pack_weights: bad arguments");
  for (int base = 0; base < count; base += kPackMax) {
    const int n = count - base < kPackMax ? count - base : kPackMax;
    PackTable t;
    int64_t most = 0;
    for (int i = 0; i < n; ++i) {
      const int* d = dims + 7 * (base + i);
      NASSEG_REQUIRE(d[0] > 0 && d[1] > 0 && d[2] > 0 && d[3] > 0 && d[4] >= 0 && d[4] <= 5,
                     "pack_weights: bad descriptor %d", base + i);
      NASSEG_REQUIRE(d[4] < 3 || d[4] == 5 || d[1] == 1, "pack_weights: depthwise weights have K = 1");
      NASSEG_REQUIRE(w[base + i] && wp[base + i], "pack_weights: null tensor %d", base + i);
      t.d[i].w = w[base + i];
      t.d[i].wp = wp[base + i];
      t.d[i].N = d[0];
      t.d[i].K = d[1];
      t.d[i].ntaps = d[2] * d[3];
      t.d[i].kind = d[4];
      t.d[i].Ksrc = d[5] > 0 ? d[5] : d[1];
      t.d[i].koff = d[5] > 0 ? d[6] : 0;
      NASSEG_REQUIRE(t.d[i].koff >= 0 && t.d[i].koff + d[1] <= t.d[i].Ksrc, "pack_weights: bad slice %d",
                     base + i);
      const int64_t total = (int64_t)d[0] * d[1] * d[2] * d[3];
      NASSEG_REQUIRE(total < 2147483647LL, "pack_weights: tensor %d too large", base + i);
      if (total > most) most = total;
    }
    int64_t gx = cdiv64(most, 256);
    if (gx > 256) gx = 256;
    hipLaunchKernelGGL(pack_multi_kernel, dim3((unsigned)gx, n), dim3(256), 0, (hipStream_t)stream, t);
    NASSEG_LAUNCH_CHECK("pack_weights");
  }
  return NASSEG_OK;
}
#endif  // NASSEG_FP32_ONLY

#if NASSEG_FP32_ONLY
int64_t nasseg_conv_pw_min_pixels(int64_t v) {
  return v == -1 ? g_pw_min_pixels.load() : g_pw_min_pixels.exchange(v < 0 ? -2 : v);
}
// four k-steps per round trip in the general kernel on maps of at most 8192 pixels: 1 (initial) on, 0 off;
// v < 0 only queries.  Returns the previous setting.  Bit-identical results (the same accumulation order).
int nasseg_conv_deep_k(int v) {
  if (v < 0) return g_conv_deep_k.load();
  return g_conv_deep_k.exchange(v ? 1 : 0);
}
#endif  // NASSEG_FP32_ONLY

#if NASSEG_FP32_ONLY
// number of per-workgroup statistic rows nasseg_conv_fwd writes for this geometry
int64_t nasseg_conv_fwd_stats_blocks(int B, int Ho, int Wo, int N, int K, int pointwise) {
  const int64_t Mtot = (int64_t)B * Ho * Wo;
  const int tiles = cdiv(N, 16);
  if (pointwise) {
    // (forward statistics of conv_pwn_kernel: two rows per workgroup - value and fp32 rounding residue)
    const PwnPlan pn = nasseg_internal_pwn_plan(Mtot, N, K, pointwise);
    if (pn.ok) return pointwise == 1 ? 2 * pn.grid : pn.grid;
    const PwFwdPlan pw = pw_fwd_plan(Mtot, N, K, pointwise);
    if (pw.ok) return pw.grid;
  }
  if (conv_small_ws(Mtot, N, K)) return cdiv64(Mtot, 16);
  return cdiv64(Mtot, (tiles > 4 ? 16 : 64) * pick_mt(Mtot, tiles));
}
#endif  // NASSEG_FP32_ONLY

#if NASSEG_FP32_ONLY
// statistic rows of a FORWARD nasseg_conv_fwd call with this geometry (any kernel size): what
// nasseg_conv_fwd_stats_blocks says for 1x1 and strided forms, the tile count of the LDS-tiled kernel for the
// stride-1 3x3 forms it takes (N % 4 == 0, no input prologue)
int64_t nasseg_conv_fwd_stats_rows(int B, int Ho, int Wo, int N, int K, int kh, int kw, int stride, int pad, int dil) {
  const int pointwise = (kh == 1 && kw == 1 && stride == 1 && pad == 0) ? 1 : 0;
  if (!pointwise && (N & 3) == 0 && lds3x3_geometry(B, Ho, Wo, N, K, kh, kw, stride, pad, dil))
  {
    // (the four-tile form picks its tile; one and two channel tiles stay on 8 x 32: launch_lds3x3)
    const Lds3Tile t = cdiv(N, 16) > 2 ? lds3x3_tile(B, Ho, Wo, dil, 4) : Lds3Tile{kLdsTH, kLdsTW};
    return (int64_t)cdiv(Wo, t.tw) * cdiv(Ho, t.th) * B;
  }
  return nasseg_conv_fwd_stats_blocks(B, Ho, Wo, N, K, pointwise);
}
#endif  // NASSEG_FP32_ONLY

#if NASSEG_FP32_ONLY
// which kernel a pointwise call over B*Ho*Wo pixels takes (measurement tools name kernels by it):
// 0 conv_fwd_kernel, 1 conv_pw_kernel, 2 conv_pwn_kernel.  pointwise: 1 = nasseg_conv_fwd, 2 = nasseg_conv_bwd_data_bn.
int64_t nasseg_conv_pointwise_kernel(int B, int Ho, int Wo, int N, int K, int pointwise) {
  const int64_t Mtot = (int64_t)B * Ho * Wo;
  if (pointwise) {
    if (nasseg_internal_pwn_plan(Mtot, N, K, pointwise).ok) return 2;
    if (pw_fwd_plan(Mtot, N, K, pointwise).ok) return 1;
  }
  return 0;
}
#endif  // NASSEG_FP32_ONLY

#if NASSEG_FP32_ONLY
// does a plain forward call take conv3x3_lds_kernel?  (conv_dispatch's own condition, for measurement tools)
int64_t nasseg_conv_fwd_lds3x3(int B, int Ho, int Wo, int N, int K, int kh, int kw, int stride, int pad, int dil,
                               int with_stats) {
  return lds3x3_geometry(B, Ho, Wo, N, K, kh, kw, stride, pad, dil) && fwd_pack_mode(K, kh, kw) != 2 &&
                 (!with_stats || (N & 3) == 0)
             ? 1
             : 0;
}
#endif  // NASSEG_FP32_ONLY

#if NASSEG_FP32_ONLY
// which packing nasseg_conv_fwd expects for a forward (non-transposed) convolution
int nasseg_conv_fwd_pack_mode(int K, int kh, int kw) { return fwd_pack_mode(K, kh, kw); }
#endif  // NASSEG_FP32_ONLY

// y[dst pixel][n] = out_act(out_scale[n] * sum_{tap,k} w[tap][n][k] *
//                   in_act(in_scale[k] * x[src pixel(tap)][k] + in_shift[k]) + out_shift[n])
//                   (+ res[dst pixel][n])
// transposed == 0: (Hs,Ws) input dims, (Ho,Wo) output dims of a forward conv; wp packed
//   with mode nasseg_conv_fwd_pack_mode(K,kh,kw).
// transposed != 0: backward-data; x is the output gradient with dims (Hs,Ws),
//   y the input gradient with dims (Ho,Wo), wp packed with mode 1, K = forward
//   N, N = forward K, stride/pad/dil those of the forward conv.
// The input prologue (in_scale / in_shift / in_act) is available for pointwise
// (1x1, stride 1) convolutions with K % 4 == 0.
// stats != null (needs N % 4 == 0 and no output epilogue): also writes
//   stats[blk][0][n] = sum over the workgroup's pixels of y[.][n], stats[blk][1][n] = sum of y^2
// for blk < nasseg_conv_fwd_stats_blocks(B, Ho, Wo, N, K, pointwise) - the partials nasseg_bn_finalize consumes.
int NASSEG_FN(conv_fwd)(const act_t* x, int ldx, const float* wp, act_t* y, int ldy,
                    const float* in_scale, const float* in_shift, int in_act,
                    const float* out_scale, const float* out_shift, int out_act, const act_t* res,
                    int ldres, int B, int Hs, int Ws, int K, int Ho, int Wo, int N, int kh, int kw,
                    int stride, int pad, int dil, int transposed, float* stats, void* stream) {
  FwdArgs a = {};
  a.x = x; a.ldx = ldx; a.w = wp; a.y = y; a.ldy = ldy;
  a.in_scale = in_scale; a.in_shift = in_shift; a.in_act = in_act;
  a.out_scale = out_scale; a.out_shift = out_shift; a.out_act = out_act;
  a.res = res; a.ldres = ldres; a.stats = stats; a.K = K; a.N = N;
  a.g.B = B; a.g.Hs = Hs; a.g.Ws = Ws; a.g.Ho = Ho; a.g.Wo = Wo;
  a.g.kh = kh; a.g.kw = kw; a.g.stride = stride; a.g.pad = pad; a.g.dil = dil;
  a.g.transposed = transposed;
  return conv_dispatch(a, stats ? 1 : 0, (hipStream_t)stream);
}

// Backward-data of a dense conv whose forward input was the normalised activation
// a = act(scale*z + shift) of a BatchNorm (never materialised - see the input prologue),
// fused with the first half of that BatchNorm's backward:
//   g[p][k]  = act'(scale[k]*z[p][k] + shift[k]) * sum_{tap,n} w[tap][n][k] * dy[src(p,tap)][n]
//   stats[blk][0][k] = sum_p g[p][k],  stats[blk][1][k] = sum_p g[p][k]*(z[p][k]-mean[k])*invstd[k]
// over the pixels p of workgroup blk < nasseg_conv_fwd_stats_blocks(B, Ho, Wo, N, K, pointwise).
// Arguments as nasseg_conv_fwd with transposed != 0: dy has dims (Hs,Ws) and K channels (the
// forward conv's output channels), g and z dims (Ho,Wo) and N channels (N % 4 == 0), wp packed
// with mode 1.  Summing the stats rows (nasseg_rows_sum) gives what nasseg_bn_bwd_reduce
// returns; nasseg_bn_bwd_apply then takes g as its dy.
// stats == null: only the act' mask (scale / shift may then be null = identity, mean / invstd are
// unused) - the backward of an activation applied on load, e.g. the ReLU ahead of pre_clf.
int NASSEG_FN(conv_bwd_data_bn)(const act_t* dy, int lddy, const float* wp, act_t* g, int ldg,
                            const act_t* z, int ldz, const float* scale, const float* shift,
                            const float* mean, const float* invstd, int act, int B, int Hs, int Ws,
                            int K, int Ho, int Wo, int N, int kh, int kw, int stride, int pad,
                            int dil, float* stats, void* stream) {
  NASSEG_REQUIRE(z && (!stats || (scale && shift && mean && invstd)), "conv_bwd_data_bn: null argument");
  NASSEG_REQUIRE((ldz & 3) == 0 && ldz >= N, "conv_bwd_data_bn: bad ldz");
  FwdArgs a = {};
  a.x = dy; a.ldx = lddy; a.w = wp; a.y = g; a.ldy = ldg;
  a.stats = stats; a.K = K; a.N = N;
  a.bz = z; a.ldbz = ldz; a.b_scale = scale; a.b_shift = shift; a.b_mean = mean;
  a.b_invstd = invstd; a.b_act = act;
  a.g.B = B; a.g.Hs = Hs; a.g.Ws = Ws; a.g.Ho = Ho; a.g.Wo = Wo;
  a.g.kh = kh; a.g.kw = kw; a.g.stride = stride; a.g.pad = pad; a.g.dil = dil;
  a.g.transposed = 1;
  return conv_dispatch(a, stats ? 2 : 3, (hipStream_t)stream);
}

}  // extern "C"
