// Backward of a POINTWISE convolution that is followed by a BatchNorm, in ONE kernel:
// the second half of the BatchNorm backward (applied as g and z are loaded), the weight
// gradient AND the input gradient.  fp32 / bf16 storage, NHWC, fp32 matrix cores, gfx950.
//
// Reference: autograd of Conv2d(K, N, 1) -> BatchNorm2d at src/nn/layer_factory.py:96-98,
// 117-122 (conv_bn / conv_bn_relu / conv_1x1_bn_relu6), :139-152 (InvertedResidual's
// expansion) and :243-255 (SepConv's pointwise stage).
//
// The two-kernel form (conv_wgrad_bn + conv_fwd as backward-data) writes
//   dz = scale*(g' - sum(g')/M - xhat*sum(g'*xhat)/M)
// once and reads it back: 2*N*4 bytes per pixel, which for the expansions of the encoder
// (16 -> 96 at 512x1024, 24 -> 144 at 256x512, 32 -> 192 at 128x256: N >> K) is most of
// what the backward of the layer moves.  Here a workgroup walks a contiguous slab of
// pixels 64 at a time: the dz tile (64 x N) and the input tile (64 x K, through the
// forward's input prologue) are built in LDS, then
//   dx[p][k]  = sum_n W[n][k] * dz[p][n]          (reduction over n: B operand = LDS rows)
//   dW[n][k] += sum_p dz[p][n] * x[p][k]          (reduction over the tile's pixels)
// both on v_mfma_f32_16x16x4_f32.  dz never reaches HBM.  Each wave owns 16 of the 64
// pixels for both products and keeps the whole N x K accumulator (N*K <= 6144); waves are
// combined through LDS in a fixed order, slabs by the deterministic second stage of
// conv_wgrad.hip (nasseg_wgrad_finalize_many) - no float atomics.
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "conv_common.h"

extern "C" int nasseg_wgrad_finalize_many(int count, const float* const* partial, float* const* dw,
                                          const int* dims, void* stream);

#if NASSEG_FP32_ONLY
// pixels from which the narrow kernel rebuilds z instead of loading it (nasseg_conv_pw_bwd_rz_min_pixels)
std::atomic<int64_t> g_pw_rz_min_pixels{1 << 18};
#else
extern std::atomic<int64_t> g_pw_rz_min_pixels;
#endif

namespace {

#ifndef NASSEG_PW_CHUNK
#define NASSEG_PW_CHUNK 3
#endif
// 1: the narrow kernel RECOMPUTES the conv's raw output z = W x from the input tile it holds anyway instead of
// loading it, where that measured faster (pw_plan; round 5; 0: load z everywhere as rounds 2-4 did - A/B, tools/gpu.sh flags)
#ifndef NASSEG_PW_RECOMPUTE_Z
#define NASSEG_PW_RECOMPUTE_Z 1
#endif
constexpr int kPwTile = 64;         // pixels per tile (16 per wave)
constexpr int kPwMaxTiles = 24;     // (n, k) accumulator tiles per wave

struct PwArgs {
  const act_t* x;
  const act_t* g;
  const act_t* z;
  act_t* dx;
  float* partial;   // [slab][N][K]
  const float* wb;  // [K][N]: backward-data layout (pack mode 1)
  const float* in_scale;
  const float* in_shift;
  int in_act;
  const float* bn_scale;
  const float* bn_shift;
  const float* bn_mean;
  const float* bn_invstd;
  const float* bn_sums;  // [2][N]
  int bn_train, bn_act;
  float invM;
  int dx_act;  // != 0: dx is multiplied by act'(x) - the forward applied this activation to x on load
  // dx_stats != NULL (narrow kernel): also the first half of the backward of the BatchNorm in front, whose
  // raw output x is - per slab {sum dx, sum dx*xhat}, xhat = (x - in_mean)*in_invstd, rows [slab][2][K]
  const float* in_mean;
  const float* in_invstd;
  float* dx_stats;
  const act_t* dx_res;  // != NULL: a map of dx's shape added to dx (after the mask): the gradient of a skip that x also feeds
  int K, N, KP, NP;  // channels, rounded up to multiples of 16
  int M, pix_per_slab;
};

// act'(x) recovered from the ACTIVATED value a = act(x) held in the input tile (no affine in
// front of the activation): ReLU: a > 0; ReLU6: 0 < a < 6
__device__ __forceinline__ float4 act_mask_of(float4 v, int act) {
  const float hi = act == NASSEG_ACT_RELU6 ? 6.f : __builtin_inff();
  return make_float4((v.x > 0.f && v.x < hi) ? 1.f : 0.f, (v.y > 0.f && v.y < hi) ? 1.f : 0.f,
                     (v.z > 0.f && v.z < hi) ? 1.f : 0.f, (v.w > 0.f && v.w < hi) ? 1.f : 0.f);
}

// LDS of the narrow kernel in floats: dz tile | input tile | per-channel constants | (WL) the weight
__host__ __device__ constexpr int pw_lsn(int NT) { return NT * 16 + 4; }
__host__ __device__ constexpr int pw_lsk(int KT) { return KT * 16 + 4; }
__host__ __device__ constexpr int pw_lds_floats(int NT, int KT, bool WL) {
  return kPwTile * (pw_lsn(NT) + pw_lsk(KT)) + 4 * NT * 16 + 2 * KT * 16 + (WL ? KT * 16 * pw_lsn(NT) : 0);
}
// the weight [K][N] is kept in LDS when everything still fits twice on a CU
__host__ __device__ constexpr bool pw_weight_in_lds(int NT, int KT) {
  return pw_lds_floats(NT, KT, true) * 4 <= (72 << 10);
}
// the next tile's loads are issued ahead when their registers (8*NT + 4*KT) still leave two waves per
// SIMD next to the 4*NT*KT accumulators (32 -> 192 at 4x128x256 with one wave: 118 us, without
// prefetch and two waves: 86 us)
__host__ __device__ constexpr bool pw_prefetch(int NT, int KT) { return NT * KT <= 18; }

// NT / KT: 16-wide tiles of N / K held per wave (>= the actual counts: surplus tiles only cost
// idle MFMAs and idle lanes of the tile loads); PRO: the forward read x through
// act(in_scale*x + in_shift).  A workgroup walks its slab in tiles of 64 pixels: the tile's g, z and
// x are loaded to registers - NT + NT + KT float4 per thread - while the PREVIOUS tile is computed
// from LDS (nothing else hides the HBM latency: at most two to six workgroups are resident on a CU
// and a tile's loads were issued, waited for and used in turn - 24 -> 144 at 4x256x512: 260 us before).
// DXS: also emit the sums of the BatchNorm in front (a.dx_stats) - a separate instantiation: its 16*KT
// registers per lane would cost every other call a resident wave
template <int NT, int KT, bool PRO, bool DXS = false, bool RZ_ = false>
__global__ __launch_bounds__(256) void conv_pw_bwd_kernel(PwArgs a) {
  extern __shared__ float smem[];
  constexpr int LSN = pw_lsn(NT), LSK = pw_lsk(KT), NPc = NT * 16, KPc = KT * 16;
  constexpr bool WL = pw_weight_in_lds(NT, KT), PF = pw_prefetch(NT, KT);
  // Recompute z.  dz = ca*g' + cb*z + cd needs the conv's raw output z, N floats per pixel that the forward wrote
  // and this kernel used to read back - for the expansions of the encoder (16 -> 96, 24 -> 144: N >> K) 43 % of
  // everything it moves.  But z = W x, x is the input tile this kernel stages for the weight gradient anyway, and
  // the weight is in LDS: K / 16 MFMA steps per channel tile rebuild it, in the operand mapping and accumulation
  // order of the forward kernels (conv_fwd.hip / conv_pwn.hip: k-blocks ascending, the four components of a lane's
  // float4 in turn, zero-padded to a multiple of 16) - the same bits the forward stored.  The loader then brings g
  // alone; a wave turns its 16 pixels of g into dz in place once z is there (one more workgroup barrier per tile).
  // Where it pays (pw_plan: K <= 32, N <= 96, maps of at least 2^18 pixels): tools/kbench_pwbwd.py on MI355X, us
  // loading z -> rebuilding it: 16 -> 96 at 4x512x1024 405 -> 328, 32 -> 32 at 4x512x1024 217 -> 171, 24 -> 64 at
  // 4x256x512 104 -> 87; but 24 -> 144 173 -> 182 (nine channel tiles: the 72 extra MFMAs per tile and the registers of
  // two more k-steps), 64 -> 64 at 4x128x256 40 -> 44, 32 -> 32 at 4x128x256 15.8 -> 16.9 (small maps are not bound by
  // their bytes).  Headline on one box 275.5 -> 279.0 images/s with every supported geometry rebuilding.
  constexpr bool RZ = WL && RZ_;
  float* dzt = smem;                       // [64][LSN]
  float* xt = dzt + kPwTile * LSN;         // [64][LSK]
  float* ca = xt + kPwTile * LSK;          // ca | cb | cd | cs [NPc each], psc | psh [KPc each]
  float* cb = ca + NPc;
  float* cd = cb + NPc;
  float* cs = cd + NPc;
  float* psc = cs + NPc;
  float* psh = psc + KPc;
  float* wl = psh + KPc;                   // (WL) [KPc][LSN]: wb[k][n], zero beyond K / N
  float* xraw = wl + (WL ? KPc * LSN : 0); // (DXS) [64][LSK]: the input tile before its prologue
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, kg = lane >> 4;
  const int N = a.N, K = a.K;

  // dz = ca*g' + cb*z + cd  ==  scale*(g' - sum(g')/M - xhat*sum(g'*xhat)/M), xhat = (z - mean)*invstd,
  // g' = g * act'(scale*z + shift) when g arrives without its activation mask (bn_act != 0)
  for (int n = tid; n < NPc; n += 256) {
    float va = 0.f, vb = 0.f, vd = 0.f, vs = 0.f;
    if (n < N) {
      const float sc = a.bn_scale[n];
      va = sc;
      vs = a.bn_act ? a.bn_shift[n] : 0.f;
      if (a.bn_train) {
        const float is = a.bn_invstd[n], mu = a.bn_mean[n];
        const float s0 = a.bn_sums[n] * a.invM, s1 = a.bn_sums[N + n] * a.invM;
        vb = -sc * is * s1;
        vd = sc * (mu * is * s1 - s0);
      }
    }
    ca[n] = va; cb[n] = vb; cd[n] = vd; cs[n] = vs;
  }
  for (int k = tid; k < KPc; k += 256) {
    psc[k] = (PRO && a.in_scale && k < K) ? a.in_scale[k] : 1.f;
    psh[k] = (PRO && a.in_shift && k < K) ? a.in_shift[k] : 0.f;
  }
  if (WL) {
    for (int it = tid; it < KPc * (NPc / 4); it += 256) {
      const int k = it / (NPc / 4), n = (it - k * (NPc / 4)) * 4;
      const float4 v = keep_if(ld4(a.wb + (int64_t)(k < K ? k : 0) * N + (n < N ? n : 0)), k < K && n < N);
      *reinterpret_cast<float4*>(&wl[k * LSN + n]) = v;
    }
  }
  const ActSel pact = act_sel(a.in_act);

  f32x4 acc2[NT][KT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) acc2[nt][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int p_begin = blockIdx.x * a.pix_per_slab;
  int p_end = p_begin + a.pix_per_slab;
  if (p_end > a.M) p_end = a.M;

  // item u of a thread: float4 number tid + 256*u of the [64][NPc] (or [64][KPc]) tile
  float4 gv[PF ? NT : 1], zv[(PF && !RZ) ? NT : 1], xv[KT];
  auto issue_nz = [&](int t0, int u_lo, int u_hi, float4* G, float4* Z) {
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      if (u < u_lo || u >= u_hi) continue;
      const int it = tid + 256 * u;
      const int px = it / (NPc / 4), n = (it - px * (NPc / 4)) * 4;
      const int p = t0 + px;
      const int64_t off = (int64_t)(p < p_end ? p : p_end - 1) * N + (n < N ? n : 0);
      G[u - u_lo] = lda4(a.g + off);
      if constexpr (!RZ) Z[u - u_lo] = lda4(a.z + off);
    }
  };
  auto issue_x = [&](int t0) {
#pragma unroll
    for (int u = 0; u < KT; ++u) {
      const int it = tid + 256 * u;
      const int px = it / (KPc / 4), k = (it - px * (KPc / 4)) * 4;
      const int p = t0 + px;
      xv[u] = lda4(a.x + (int64_t)(p < p_end ? p : p_end - 1) * K + (k < K ? k : 0));
    }
  };
  auto issue = [&](int t0) {
    issue_nz(t0, 0, NT, gv, zv);
    issue_x(t0);
  };
  // registers -> the dz tile
  auto place_nz = [&](int t0, int u_lo, int u_hi, const float4* G, const float4* Z) {
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      if (u < u_lo || u >= u_hi) continue;
      const int it = tid + 256 * u;
      const int px = it / (NPc / 4), n = (it - px * (NPc / 4)) * 4;
      const bool ok = t0 + px < p_end && n < N;
      if constexpr (RZ) {
        // (g as it came: the wave that owns the pixel makes dz of it below, once z is rebuilt)
        *reinterpret_cast<float4*>(&dzt[px * LSN + n]) = keep_if(G[u - u_lo], ok);
      } else {
        const float4 va = ld4(ca + n), vb = ld4(cb + n), vd = ld4(cd + n);
        float4 g2 = G[u - u_lo];
        const float4 z2 = Z[u - u_lo];
        if (a.bn_act) {
          const float4 y = fma4(z2, va, ld4(cs + n));
          g2 = make_float4(g2.x * act_mask(y.x, a.bn_act), g2.y * act_mask(y.y, a.bn_act),
                           g2.z * act_mask(y.z, a.bn_act), g2.w * act_mask(y.w, a.bn_act));
        }
        float4 dz = fma4(g2, va, fma4(z2, vb, vd));
#ifdef NASSEG_BF16
        // (what the two-kernel form stores and reads back)
        dz = make_float4(bf16_to_f32(f32_to_bf16(dz.x)), bf16_to_f32(f32_to_bf16(dz.y)),
                         bf16_to_f32(f32_to_bf16(dz.z)), bf16_to_f32(f32_to_bf16(dz.w)));
#endif
        *reinterpret_cast<float4*>(&dzt[px * LSN + n]) = keep_if(dz, ok);
      }
    }
  };
  float4 sx[DXS ? KT : 1], sq[DXS ? KT : 1], smu[DXS ? KT : 1], sis[DXS ? KT : 1];
#pragma unroll
  for (int kt = 0; kt < (DXS ? KT : 0); ++kt) {
    const int k = kt * 16 + kg * 4;
    sx[kt] = sq[kt] = smu[kt] = sis[kt] = f4zero();
    if (k < K) {
      smu[kt] = lda4(a.in_mean + k);
      sis[kt] = lda4(a.in_invstd + k);
    }
  }
  if (PF && p_begin < p_end) issue(p_begin);
  for (int t0 = p_begin; t0 < p_end; t0 += kPwTile) {
    __syncthreads();  // (the previous tile's operands have been read; first pass: constants are in place)
    // ---- registers -> the dz tile and the input tile (through the forward's prologue) ------------
    if (PF) {
      place_nz(t0, 0, NT, gv, zv);
    } else {
      // no room for a whole tile in registers: NASSEG_PW_CHUNK items at a time (3: two waves per SIMD)
      issue_x(t0);
#pragma unroll
      for (int u0 = 0; u0 < NT; u0 += NASSEG_PW_CHUNK) {
        float4 g4[NASSEG_PW_CHUNK], z4[NASSEG_PW_CHUNK];
        issue_nz(t0, u0, u0 + NASSEG_PW_CHUNK, g4, z4);
        place_nz(t0, u0, u0 + NASSEG_PW_CHUNK, g4, z4);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int u = 0; u < KT; ++u) {
      const int it = tid + 256 * u;
      const int px = it / (KPc / 4), k = (it - px * (KPc / 4)) * 4;
      const bool ok = t0 + px < p_end && k < K;
      float4 v = xv[u];
      if (DXS) *reinterpret_cast<float4*>(&xraw[px * LSK + k]) = v;
      if (PRO) v = act_apply4(fma4(v, ld4(psc + k), ld4(psh + k)), pact);
      *reinterpret_cast<float4*>(&xt[px * LSK + k]) = keep_if(v, ok);
    }
    __syncthreads();
    if (PF && t0 + kPwTile < p_end) issue(t0 + kPwTile);  // in flight until the top of the next pass
    if constexpr (RZ) {
      // ---- z of this wave's 16 pixels rebuilt on the matrix cores, g -> dz in place ----------------
      // (channel tiles in groups of ZG: the whole of z at once - NT accumulators and 4 NT weight operands next to the
      //  prefetched tile and the N x K accumulator - cost 24 -> 144 its second wave per SIMD)
      constexpr int ZG = NT % 3 == 0 ? 3 : 2;
      const bool pok = t0 + wave * 16 + j < p_end;
      float4 bx[KT];
#pragma unroll
      for (int kb = 0; kb < KT; ++kb)
        bx[kb] = *reinterpret_cast<const float4*>(&xt[(wave * 16 + j) * LSK + kb * 16 + kg * 4]);
#pragma unroll
      for (int n0 = 0; n0 < NT; n0 += ZG) {
        f32x4 zc[ZG];
#pragma unroll
        for (int q = 0; q < ZG; ++q) zc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < KT; ++kb) {
          const float* wk = wl + (kb * 16 + kg * 4) * LSN + (n0 * 16 + j);  // W[n = 16 nt + j][k = 16 kb + 4 kg + c] = wl[k][n]
#pragma unroll
          for (int q = 0; q < ZG; ++q) {
            if (n0 + q < NT) {
              zc[q] = mfma16(wk[q * 16], bx[kb].x, zc[q]);
              zc[q] = mfma16(wk[LSN + q * 16], bx[kb].y, zc[q]);
              zc[q] = mfma16(wk[2 * LSN + q * 16], bx[kb].z, zc[q]);
              zc[q] = mfma16(wk[3 * LSN + q * 16], bx[kb].w, zc[q]);
            }
          }
        }
#pragma unroll
        for (int q = 0; q < ZG; ++q) {
          if (n0 + q >= NT) continue;
          const int n = (n0 + q) * 16 + kg * 4;  // (lane: pixel j of the wave's 16, channels n .. n + 3)
          float* slot = &dzt[(wave * 16 + j) * LSN + n];
          float4 g2 = *reinterpret_cast<const float4*>(slot);
          float4 z2 = make_float4(zc[q][0], zc[q][1], zc[q][2], zc[q][3]);
#ifdef NASSEG_BF16
          // (the value the forward stored and every other consumer of z reads)
          z2 = make_float4(bf16_to_f32(f32_to_bf16(z2.x)), bf16_to_f32(f32_to_bf16(z2.y)), bf16_to_f32(f32_to_bf16(z2.z)),
                           bf16_to_f32(f32_to_bf16(z2.w)));
#endif
          const float4 va = ld4(ca + n), vb = ld4(cb + n), vd = ld4(cd + n);
          if (a.bn_act) {
            const float4 y = fma4(z2, va, ld4(cs + n));
            g2 = make_float4(g2.x * act_mask(y.x, a.bn_act), g2.y * act_mask(y.y, a.bn_act),
                             g2.z * act_mask(y.z, a.bn_act), g2.w * act_mask(y.w, a.bn_act));
          }
          float4 dz = fma4(g2, va, fma4(z2, vb, vd));
#ifdef NASSEG_BF16
          dz = make_float4(bf16_to_f32(f32_to_bf16(dz.x)), bf16_to_f32(f32_to_bf16(dz.y)),
                           bf16_to_f32(f32_to_bf16(dz.z)), bf16_to_f32(f32_to_bf16(dz.w)));
#endif
          *reinterpret_cast<float4*>(slot) = keep_if(dz, pok && n < N);
        }
        __builtin_amdgcn_sched_barrier(0);  // (one group at a time: keeps the next group's operands out of this one's registers)
      }
      // (no barrier: both products below read only the rows of dz of THIS wave's 16 pixels, and a wave's LDS
      //  accesses execute in issue order)
    }
    // ---- input gradient of this wave's 16 pixels: dx[p][k] = sum_n wb[k][n] * dz[p][n] -----------
    {
      f32x4 acc1[KT];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) acc1[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      // (the skip's gradient for this wave's dx tile: in flight during the MFMAs below)
      float4 rres[KT];
      if (a.dx_res) {
        const int pr = t0 + wave * 16 + j;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          const int k = kt * 16 + kg * 4;
          rres[kt] = lda4(a.dx_res + ((pr < p_end && k < K) ? (int64_t)pr * K + k : 0));
        }
      }
      const float* brow = dzt + (wave * 16 + j) * LSN;
      for (int ns = 0; ns < (a.NP >> 4); ++ns) {
        const int n = ns * 16 + kg * 4;
        const float4 bv = *reinterpret_cast<const float4*>(brow + n);
        float4 av[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          const int k = kt * 16 + j;
          if (WL) av[kt] = *reinterpret_cast<const float4*>(&wl[k * LSN + n]);
          else av[kt] = keep_if(ld4(a.wb + (int64_t)(k < K ? k : 0) * N + (n < N ? n : 0)), k < K && n < N);
        }
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          acc1[kt] = mfma16(av[kt].x, bv.x, acc1[kt]);
          acc1[kt] = mfma16(av[kt].y, bv.y, acc1[kt]);
          acc1[kt] = mfma16(av[kt].z, bv.z, acc1[kt]);
          acc1[kt] = mfma16(av[kt].w, bv.w, acc1[kt]);
        }
      }
      const int p = t0 + wave * 16 + j;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        const int k = kt * 16 + kg * 4;
        float4 o = make_float4(acc1[kt][0], acc1[kt][1], acc1[kt][2], acc1[kt][3]);
        if (a.dx_act)
          o = mul4(o, act_mask_of(*reinterpret_cast<const float4*>(&xt[(wave * 16 + j) * LSK + k]), a.dx_act));
        const bool ok = p < p_end && k < K;
        if (ok) sta4(a.dx + (int64_t)p * K + k, a.dx_res ? add4(o, rres[kt]) : o);
        if (DXS) {
          // (the raw input, kept in LDS next to the activated tile: xhat cannot be taken from the activated
          //  value when the BatchNorm's weight is zero, and a global re-load here would queue up behind the
          //  next tile's loads that were just issued)
          const float4 zr = *reinterpret_cast<const float4*>(&xraw[(wave * 16 + j) * LSK + k]);
#ifdef NASSEG_BF16
          o = make_float4(bf16_to_f32(f32_to_bf16(o.x)), bf16_to_f32(f32_to_bf16(o.y)), bf16_to_f32(f32_to_bf16(o.z)),
                          bf16_to_f32(f32_to_bf16(o.w)));  // (what a separate pass would read back)
#endif
          const float4 gm = keep_if(o, ok);
          const float4 xh = make_float4((zr.x - smu[kt].x) * sis[kt].x, (zr.y - smu[kt].y) * sis[kt].y,
                                        (zr.z - smu[kt].z) * sis[kt].z, (zr.w - smu[kt].w) * sis[kt].w);
          sx[kt] = add4(sx[kt], gm);
          sq[kt] = fma4(gm, xh, sq[kt]);
        }
      }
    }
    // ---- weight gradient: dW[n][k] += sum over this wave's 16 pixels of dz[p][n] * x[p][k] -------
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int pl = wave * 16 + u * 4 + kg;  // (the MFMA's 4 reduction slots are 4 pixels)
      float av[NT], bv[KT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) av[nt] = dzt[pl * LSN + nt * 16 + j];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) bv[kt] = xt[pl * LSK + kt * 16 + j];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) acc2[nt][kt] = mfma16(av[nt], bv[kt], acc2[nt][kt]);
    }
  }

  // ---- the slab's row of BatchNorm-backward sums: 16 pixel lanes (DPP), 4 waves (LDS), fixed order ----
  if (DXS) {
    __syncthreads();
    float* sred = smem;  // [4 waves][2][KPc]
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      const float vs[4] = {row16_allsum(sx[kt].x), row16_allsum(sx[kt].y), row16_allsum(sx[kt].z), row16_allsum(sx[kt].w)};
      const float vq[4] = {row16_allsum(sq[kt].x), row16_allsum(sq[kt].y), row16_allsum(sq[kt].z), row16_allsum(sq[kt].w)};
      if (j == 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          sred[(wave * 2 + 0) * KPc + kt * 16 + kg * 4 + c] = vs[c];
          sred[(wave * 2 + 1) * KPc + kt * 16 + kg * 4 + c] = vq[c];
        }
      }
    }
    __syncthreads();
    for (int t = tid; t < KPc; t += 256) {
      if (t < K) {
        float* po = a.dx_stats + (int64_t)blockIdx.x * 2 * K + t;
        po[0] = (sred[0 * KPc + t] + sred[2 * KPc + t]) + (sred[4 * KPc + t] + sred[6 * KPc + t]);
        po[K] = (sred[1 * KPc + t] + sred[3 * KPc + t]) + (sred[5 * KPc + t] + sred[7 * KPc + t]);
      }
    }
  }
  // ---- waves -> workgroup partial, one (n, k) tile at a time, fixed order -------------------------
  float* red = smem;  // [3][64][4]
  float* pout = a.partial + (int64_t)blockIdx.x * N * K;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      __syncthreads();
      if (wave > 0) *reinterpret_cast<f32x4*>(&red[((wave - 1) * 64 + lane) * 4]) = acc2[nt][kt];
      __syncthreads();
      if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = acc2[nt][kt][r] + red[(0 * 64 + lane) * 4 + r] + red[(1 * 64 + lane) * 4 + r] +
                          red[(2 * 64 + lane) * 4 + r];
          const int n = nt * 16 + 4 * kg + r, k = kt * 16 + j;  // D row <-> n, D col <-> k
          if (n < N && k < K) pout[(int64_t)n * K + k] = v;
        }
      }
    }
}

// The same for WIDE inputs (K up to 384, N <= 64: pre_clf 224 -> 64, the CVPR decoder's adapt
// convs 320 -> 64): the N x K accumulator does not fit one wave, so the four waves split N -
// wave w owns output channels [16w, 16w+16) and walks ALL 64 pixels of a tile for the weight
// gradient (no cross-wave sum at the end); the input gradient stays 16 pixels per wave.  The
// input tile is staged 64 channels at a time (the dz tile stays).  KC: number of 64-channel chunks
// (>= ceil(K / 64)).
constexpr int kPwChunk = 64;
// The loop is software-pipelined the way the narrow kernel is.  While the MFMAs of chunk c run out of LDS, the
// loads of chunk c+1 - 64 input channels of the tile AND the matching 64 rows of the backward-data
// weight - are in flight in registers, and during a tile's last chunk the next tile's g and z as
// well.  Nothing in the MFMA phase touches global memory except the dx stores: a weight load issued
// there would queue up behind the prefetch (loads return in order) and stall the first MFMA for a
// whole HBM round trip, which is what kept the first form of this kernel (round 2: every chunk loaded,
// waited for and used in turn, the weight read from L1 inside the loop) at ~30 % of the fp32 MFMA rate
// with its two resident workgroups per CU: 224 -> 64 at 4x256x512 681 us, now 410-425 us (memory
// phase alone 160 us, dW product +90, dx product and stores +125: the MFMA phases run near their peak,
// what is left is that they do not yet overlap the memory phase).  The last chunk of K = 224 issues only the 16-wide steps it has
// channels for (2 of 4).  LDS: three [64][68] tiles (dz, x chunk, weight chunk) + constants = 55 KB.
constexpr int kW2LS = kPwChunk + 4;  // row stride of all three tiles (N <= 64)
template <int KC, bool PRO>
__global__ __launch_bounds__(256, 2) void conv_pw_bwd_wide_kernel(PwArgs a) {
  extern __shared__ float smem[];
  constexpr int LS = kW2LS;
  float* dzt = smem;                 // [64][LS]: dz[p][n]
  float* xt = dzt + kPwTile * LS;    // [64][LS]: x[p][kc + .] through the forward's prologue
  float* wt = xt + kPwTile * LS;     // [64][LS]: wb[kc + .][n] of chunks 1 .. KC-1, streamed with the x chunks
  float* wt0 = wt + kPwChunk * LS;   // [64][LS]: chunk 0 of the weight - the same for every tile: loaded once
  float* ca = wt0 + kPwChunk * LS;   // ca | cb | cd | cs [64 each], psc | psh [KC * 64 each]
  float* cb = ca + 64;
  float* cd = cb + 64;
  float* cs = cd + 64;
  float* psc = cs + 64;
  float* psh = psc + KC * kPwChunk;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, kg = lane >> 4;
  const int N = a.N, K = a.K;
  for (int n = tid; n < 64; n += 256) {
    float va = 0.f, vb = 0.f, vd = 0.f, vs = 0.f;
    if (n < N) {
      const float sc = a.bn_scale[n];
      va = sc;
      vs = a.bn_act ? a.bn_shift[n] : 0.f;
      if (a.bn_train) {
        const float is = a.bn_invstd[n], mu = a.bn_mean[n];
        const float s0 = a.bn_sums[n] * a.invM, s1 = a.bn_sums[N + n] * a.invM;
        vb = -sc * is * s1;
        vd = sc * (mu * is * s1 - s0);
      }
    }
    ca[n] = va; cb[n] = vb; cd[n] = vd; cs[n] = vs;
  }
  for (int k = tid; k < KC * kPwChunk; k += 256) {
    psc[k] = (PRO && a.in_scale && k < K) ? a.in_scale[k] : 1.f;
    psh[k] = (PRO && a.in_shift && k < K) ? a.in_shift[k] : 0.f;
  }
  const ActSel pact = act_sel(a.in_act);
  f32x4 acc2[KC][4];  // dW[16*wave + 4*kg + r][64*c + 16*q + j]
#pragma unroll
  for (int c = 0; c < KC; ++c)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc2[c][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int p_begin = blockIdx.x * a.pix_per_slab;
  int p_end = p_begin + a.pix_per_slab;
  if (p_end > a.M) p_end = a.M;
  const bool own = wave * 16 < a.NP;
  const int NS = a.NP >> 4;
  const float m_lo = a.bn_act ? 0.f : -__builtin_inff();
  const float m_hi = a.bn_act == NASSEG_ACT_RELU6 ? 6.f : __builtin_inff();
  // item u of a thread: float4 number tid + 256*u of a [64][64] tile (row = it / 16, column = 4*(it % 16))
  const int irow = tid >> 4, icol = (tid & 15) * 4;  // (+ 16 rows per u)
  float4 XN[4], WN[4], GN[4], ZN[4];
  auto issue_gz = [&](int t0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int p = t0 + irow + 16 * u;
      const int64_t off = (int64_t)(p < p_end ? p : p_end - 1) * N + (icol < N ? icol : 0);
      GN[u] = lda4(a.g + off);
      ZN[u] = lda4(a.z + off);
    }
  };
  auto issue_xw = [&](int t0, int kc) {
    const int k = kc + icol;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int p = t0 + irow + 16 * u;
      XN[u] = lda4(a.x + (int64_t)(p < p_end ? p : p_end - 1) * K + (k < K ? k : 0));
      if (kc) {  // (uniform; chunk 0 of the weight is resident)
        const int kw = kc + irow + 16 * u;
        WN[u] = ld4(a.wb + (int64_t)(kw < K ? kw : 0) * N + (icol < N ? icol : 0));
      }
    }
  };
  auto place_gz = [&](int t0) {
    const float4 va = ld4(ca + icol), vb = ld4(cb + icol), vd = ld4(cd + icol), vs = ld4(cs + icol);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int px = irow + 16 * u;
      const bool ok = t0 + px < p_end && icol < N;
      float4 gv = GN[u];
      const float4 zv = ZN[u];
      {  // g' = g * act'(scale*z + shift), without a branch: the open interval (m_lo, m_hi) is where act' = 1
        const float4 y = fma4(zv, va, vs);
        gv = make_float4((y.x > m_lo && y.x < m_hi) ? gv.x : 0.f, (y.y > m_lo && y.y < m_hi) ? gv.y : 0.f,
                         (y.z > m_lo && y.z < m_hi) ? gv.z : 0.f, (y.w > m_lo && y.w < m_hi) ? gv.w : 0.f);
      }
      float4 dz = fma4(gv, va, fma4(zv, vb, vd));
#ifdef NASSEG_BF16
      dz = make_float4(bf16_to_f32(f32_to_bf16(dz.x)), bf16_to_f32(f32_to_bf16(dz.y)),
                       bf16_to_f32(f32_to_bf16(dz.z)), bf16_to_f32(f32_to_bf16(dz.w)));
#endif
      *reinterpret_cast<float4*>(&dzt[px * LS + icol]) = keep_if(dz, ok);
    }
  };
  auto place_xw = [&](int t0, int kc) {
    const int k = kc + icol;
    const float4 sc = ld4(psc + k), sh = ld4(psh + k);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int px = irow + 16 * u;
      float4 xv = XN[u];
      if (PRO) xv = act_apply4(fma4(xv, sc, sh), pact);
      *reinterpret_cast<float4*>(&xt[px * LS + icol]) = keep_if(xv, t0 + px < p_end && k < K);
      if (kc) *reinterpret_cast<float4*>(&wt[px * LS + icol]) = keep_if(WN[u], kc + px < K && icol < N);
    }
  };
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int kw = irow + 16 * u;
    *reinterpret_cast<float4*>(&wt0[kw * LS + icol]) =
        keep_if(ld4(a.wb + (int64_t)(kw < K ? kw : 0) * N + (icol < N ? icol : 0)), kw < K && icol < N);
  }
  if (p_begin < p_end) {
    issue_gz(p_begin);
    issue_xw(p_begin, 0);
  }
  for (int t0 = p_begin; t0 < p_end; t0 += kPwTile) {
    __syncthreads();  // (the previous tile's operands have been read; first pass: the constants are in place)
    place_gz(t0);
    // (a run-time loop: unrolled, the KC copies of the body cost ~60 registers more than one copy - only the
    //  weight-gradient accumulators differ per chunk, selected by the switch below)
#pragma unroll 1
    for (int c = 0; c < KC; ++c) {
      const int kc = c * kPwChunk;
      if (kc < a.KP) {  // (uniform)
        if (c) __syncthreads();  // (the previous chunk's operands have been read)
        place_xw(t0, kc);
        __syncthreads();
        // what the NEXT pass will place: in flight during this chunk's MFMAs
        if (kc + kPwChunk < a.KP) {
          issue_xw(t0, kc + kPwChunk);
        } else if (t0 + kPwTile < p_end) {
          issue_gz(t0 + kPwTile);
          issue_xw(t0 + kPwTile, 0);
        }
        const int nq = (a.KP - kc) >> 4;  // 16-wide steps this chunk has channels for (>= 4: all)
        // The MFMA phase, specialised on the number of 16-wide steps (NQ): a run-time guard around the
        // MFMAs turns every group into "LDS read, wait, one MFMA" (measured: the first version of this loop).
        auto mfma_phase = [&](auto nq_tag) {
          constexpr int NQ = decltype(nq_tag)::value;
          // input gradient of this wave's 16 pixels, this chunk's 16*NQ input channels
          {
            const float* brow = dzt + (wave * 16 + j) * LS;
            const float* wcur = c ? wt : wt0;
            const int p = t0 + wave * 16 + j;
            f32x4 acc1[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) acc1[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
            float4 rres[NQ];  // (the skip's gradient for this chunk of the wave's dx tile: in flight during the MFMAs)
            if (a.dx_res) {
#pragma unroll
              for (int q = 0; q < NQ; ++q) {
                const int k = kc + q * 16 + kg * 4;
                rres[q] = lda4(a.dx_res + ((p < p_end && k < K) ? (int64_t)p * K + k : 0));
              }
            }
#pragma unroll 1
            for (int ns = 0; ns < NS; ++ns) {
              const int n = ns * 16 + kg * 4;
              const float4 bv = *reinterpret_cast<const float4*>(brow + n);
              float4 av[NQ];
#pragma unroll
              for (int q = 0; q < NQ; ++q) av[q] = *reinterpret_cast<const float4*>(&wcur[(q * 16 + j) * LS + n]);
#pragma unroll
              for (int q = 0; q < NQ; ++q) acc1[q] = mfma16(av[q].x, bv.x, acc1[q]);
#pragma unroll
              for (int q = 0; q < NQ; ++q) acc1[q] = mfma16(av[q].y, bv.y, acc1[q]);
#pragma unroll
              for (int q = 0; q < NQ; ++q) acc1[q] = mfma16(av[q].z, bv.z, acc1[q]);
#pragma unroll
              for (int q = 0; q < NQ; ++q) acc1[q] = mfma16(av[q].w, bv.w, acc1[q]);
            }
            const bool pok = p < p_end;
            act_t* prow = a.dx + (int64_t)(pok ? p : p_begin) * K + kc + kg * 4;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
              const int kl = q * 16 + kg * 4;
              float4 o = make_float4(acc1[q][0], acc1[q][1], acc1[q][2], acc1[q][3]);
              if (a.dx_act)
                o = mul4(o, act_mask_of(*reinterpret_cast<const float4*>(&xt[(wave * 16 + j) * LS + kl]), a.dx_act));
              if (pok && kc + kl < K) sta4(prow + q * 16, a.dx_res ? add4(o, rres[q]) : o);
            }
          }
          // weight gradient rows of this wave: all 64 pixels of the tile, 4 per MFMA
          if (own) {
            auto dw_chunk = [&](f32x4 (&acc)[4]) {
#pragma unroll 2
              for (int u = 0; u < kPwTile / 4; ++u) {
                const int pl = u * 4 + kg;
                const float av = dzt[pl * LS + wave * 16 + j];
                float bv[NQ];
#pragma unroll
                for (int q = 0; q < NQ; ++q) bv[q] = xt[pl * LS + q * 16 + j];
#pragma unroll
                for (int q = 0; q < NQ; ++q) acc[q] = mfma16(av, bv[q], acc[q]);
              }
            };
            switch (c) {
              case 0: dw_chunk(acc2[0]); break;
              case 1: dw_chunk(acc2[1]); break;
              case 2: if constexpr (KC > 2) dw_chunk(acc2[2]); break;
              case 3: if constexpr (KC > 3) dw_chunk(acc2[3]); break;
              case 4: if constexpr (KC > 4) dw_chunk(acc2[4]); break;
              default: if constexpr (KC > 5) dw_chunk(acc2[5]); break;
            }
          }
        };
        if (nq >= 4) mfma_phase(std::integral_constant<int, 4>());
        else if (nq == 3) mfma_phase(std::integral_constant<int, 3>());
        else if (nq == 2) mfma_phase(std::integral_constant<int, 2>());
        else mfma_phase(std::integral_constant<int, 1>());
      }
    }
  }
  if (own) {
    float* pout = a.partial + (int64_t)blockIdx.x * N * K;
#pragma unroll
    for (int c = 0; c < KC; ++c)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = wave * 16 + 4 * kg + r, k = c * kPwChunk + q * 16 + j;
          if (n < N && k < K) pout[(int64_t)n * K + k] = acc2[c][q][r];
        }
  }
}

struct PwPlan {
  int ok, nt, kt, nslab, pix_per_slab, wide, rz;
};
inline int pw_round_nt(int nt) {
  const int allowed[] = {2, 3, 4, 6, 9, 12};
  for (int v : allowed)
    if (nt <= v) return v;
  return 0;
}
inline PwPlan pw_plan(int64_t M, int N, int K) {
  PwPlan p = {};
  if (N % 4 || K % 4 || N <= 0 || K <= 0 || M <= 0 || M >= 2147483647LL) return p;
  const int kt0 = cdiv(K, 16);
  p.kt = kt0 <= 1 ? 1 : (kt0 <= 2 ? 2 : (kt0 <= 4 ? 4 : 0));
  p.nt = pw_round_nt(cdiv(N, 16));
  if (!p.kt || !p.nt || p.nt * p.kt > kPwMaxTiles) {
    // wide inputs: the four waves split N (conv_pw_bwd_wide_kernel), K in 64-channel chunks
    if (N > 64 || K > 384) return p;
    p.wide = 1;
    p.nt = cdiv(N, 16);
    p.kt = cdiv(K, kPwChunk);  // chunks: 2 .. 6
  }
  // Slabs: a power of two times 256 CUs - 4 workgroups per CU, 2 when the tiles take more than
  // 40 KB of LDS - each at least 4 tiles long, partials <= 16 MiB.  Measured (tools/kbench_pwbwd.py,
  // 16 -> 96 at 4x512x1024): 1024 slabs 457 us, 2048 457, 1490 532, 763 519, 512 572; 24 -> 144 at
  // 4x256x512: 512 slabs 280 us, 1024 288, 745 366 - counts that leave the CUs with unequal numbers of
  // resident workgroups cost 15-30 %.
  const int64_t lds = p.wide ? (int64_t)(32 << 10)  // (two workgroups per CU by registers: 1024 and 512 slabs time alike)
                             : (int64_t)pw_lds_floats(p.nt, p.kt, pw_weight_in_lds(p.nt, p.kt)) * 4;
  int64_t s = lds > (40 << 10) ? 512 : 1024;
  const int64_t cap = (int64_t)((p.wide ? 64 : 16) << 20) / ((int64_t)N * K * 4);
  while (s > 1 && (s > cap || s > M / (4 * kPwTile))) s >>= 1;
  // Small maps (fewer than 256 slabs of four tiles): one workgroup per CU and as many CUs as there are tiles rather than
  // long slabs on a few of them - tools/kbench_pwbwd.py, us: 32 -> 192 at 8 x 60 x 80 120 slabs 65, 200 slabs 45 (300: 65
  // again - two workgroups on some CUs); at 16 x 41 x 41 61 slabs 83, 211 slabs 34; 64 -> 64 at 8 x 60 x 80 120 slabs
  // 33, 200 slabs 24.5.
  if (!p.wide && s < 256 && M / kPwTile > s) {
    s = M / kPwTile < 256 ? M / kPwTile : 256;
    if (s > cap) s = cap;
    if (s < 1) s = 1;
  }
#ifdef NASSEG_TUNE  // (tools/kbench_pwbwd.py: any slab count by hand)
  if (const char* e = getenv("NASSEG_PW_SLABS")) {
    const int64_t v = atoll(e);
    if (v > 0 && v <= cap) s = v;
  }
#endif
  int64_t ppb = cdiv64(M, s);
  ppb = (ppb + kPwTile - 1) / kPwTile * kPwTile;
  p.pix_per_slab = (int)ppb;
  p.nslab = (int)cdiv64(M, ppb);
  // rebuild z instead of loading it (conv_pw_bwd_kernel<.., RZ_ = true>): where it measured faster
  p.rz = NASSEG_PW_RECOMPUTE_Z != 0 && !p.wide && p.kt <= 2 && p.nt <= 6 && pw_weight_in_lds(p.nt, p.kt) &&
         M >= g_pw_rz_min_pixels.load();
  p.ok = 1;
  return p;
}

template <int NT, int KT, bool RZ = false>
void pw_launch(const PwArgs& a, int nslab, bool pro, hipStream_t s) {
  constexpr size_t lds = (size_t)pw_lds_floats(NT, KT, pw_weight_in_lds(NT, KT)) * sizeof(float);
  constexpr size_t lds_dxs = lds + (size_t)kPwTile * pw_lsk(KT) * sizeof(float);  // (+ the raw input tile)
  if (lds_dxs > (64 << 10)) {  // above the default limit of dynamic LDS (per device: set on every launch)
    (void)hipFuncSetAttribute((const void*)conv_pw_bwd_kernel<NT, KT, true, true, RZ>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dxs);
    if (pro) (void)hipFuncSetAttribute((const void*)conv_pw_bwd_kernel<NT, KT, true, false, RZ>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    else (void)hipFuncSetAttribute((const void*)conv_pw_bwd_kernel<NT, KT, false, false, RZ>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  if (a.dx_stats) hipLaunchKernelGGL((conv_pw_bwd_kernel<NT, KT, true, true, RZ>), dim3(nslab), dim3(256), lds_dxs, s, a);
  else if (pro) hipLaunchKernelGGL((conv_pw_bwd_kernel<NT, KT, true, false, RZ>), dim3(nslab), dim3(256), lds, s, a);
  else hipLaunchKernelGGL((conv_pw_bwd_kernel<NT, KT, false, false, RZ>), dim3(nslab), dim3(256), lds, s, a);
}

}  // namespace

extern "C" {

#if NASSEG_FP32_ONLY
// number of partial rows [N][K] nasseg_conv_pw_bwd_bn leaves in its workspace (the workspace holds
// that many * N * K floats); 0: the geometry has no fused kernel (N*K too large, channels not
// multiples of 4) - use nasseg_conv_wgrad_bn + the backward-data call
int64_t nasseg_conv_pw_bwd_slabs(int B, int H, int W, int K, int N) {
  const PwPlan p = pw_plan((int64_t)B * H * W, N, K);
  return p.ok ? p.nslab : 0;
}
#endif  // NASSEG_FP32_ONLY

#if NASSEG_FP32_ONLY
// 1: nasseg_conv_pw_bwd_bn loads the conv's raw output z for this geometry; 0: it rebuilds z = W x from the input
// tile on the matrix cores (the narrow kernel, weight in LDS - z is then not read at all).  For measurement tools
// (bench.py charges the bytes a launch needs) and tests.
int64_t nasseg_conv_pw_bwd_reads_z(int B, int H, int W, int K, int N) {
  const PwPlan p = pw_plan((int64_t)B * H * W, N, K);
  return (p.ok && p.rz) ? 0 : 1;
}
// maps of at least this many pixels have z rebuilt (where the kernel can: K <= 32, N <= 96): 2^18 initially - below,
// the launches are not bound by their bytes and the extra MFMAs lose (32 -> 32 at 4x128x256: 15.8 -> 16.9 us); 0: every
// supported geometry (what the parity tests of the small maps use), a huge value: none.  v < 0 only queries.  Returns
// the previous setting.
int64_t nasseg_conv_pw_bwd_rz_min_pixels(int64_t v) {
  if (v < 0) return g_pw_rz_min_pixels.load();
  return g_pw_rz_min_pixels.exchange(v);
}
#endif  // NASSEG_FP32_ONLY

// Backward of y = BatchNorm(conv1x1(in_act(in_scale*x + in_shift))):
//   g [P][N]: gradient w.r.t. the BatchNorm output - masked already (bn_act == 0) or to be masked with
//   act'(scale*z + shift) here; z [P][N] the conv's raw output - it MUST be that output: where
//   nasseg_conv_pw_bwd_reads_z() is 0 the kernel rebuilds it from x and wb instead of reading it;
//   sums[2][N] = {sum g', sum g'*xhat};
//   wb: the weight packed for backward-data ([K][N], pack mode 1); P = B*H*W pixels.
// dx_act != 0 (= in_act): dx is multiplied by in_act'(in_scale*x + in_shift), i.e. it is the gradient
// w.r.t. the affine's output - w.r.t. x itself for a bare activation (the ReLU that pre_clf applies to its
// input on load), w.r.t. the output of the BatchNorm in front for a conv inside a chain.  The mask is taken
// from the activated tile (a > 0, a < 6), which is the same thing.
// Writes dx [P][K] = the gradient w.r.t. the conv's (prologue-transformed) input and the weight
// gradient: dw (N,K,1,1) when given, else only the partial rows in ws (nasseg_conv_pw_bwd_slabs rows
// of N*K floats) for nasseg_wgrad_finalize_many (taps 1, flat 0).
// dx_stats != NULL (K <= 64): x is the raw output of a BatchNorm in front (in_scale / in_shift / in_act its
// normalisation, in_mean / in_invstd its statistics, dx_act = in_act): per slab the sums {sum dx, sum dx*xhat}
// of THAT BatchNorm's backward are written to dx_stats[slab][2][K] (nasseg_conv_pw_bwd_slabs rows; add them with
// nasseg_rows_sum, the buffer needs 64 more rows) - what a nasseg_bn_bwd_reduce pass over dx and x would return.
// dx_res != NULL (K % 4 == 0, no dx_stats): a [P][K] map added to dx after the mask - the gradient of a skip
// connection that x feeds as well (InvertedResidual, src/nn/layer_factory.py:276-321), so that no separate
// accumulation of the two gradients of x is needed.
int NASSEG_FN(conv_pw_bwd_bn)(const act_t* x, const act_t* g, const act_t* z, const float* wb, act_t* dx,
                              float* dw, float* ws, const float* in_scale, const float* in_shift,
                              int in_act, int dx_act, const float* bn_scale, const float* bn_shift,
                              const float* bn_mean, const float* bn_invstd, const float* bn_sums,
                              int bn_train, int bn_act, int B, int H, int W, int K, int N, const float* in_mean,
                              const float* in_invstd, float* dx_stats, const act_t* dx_res, void* stream) {
  NASSEG_REQUIRE(x && g && wb && dx && ws && bn_scale, "conv_pw_bwd_bn: null tensor");
  NASSEG_REQUIRE(!dx_stats || (in_mean && in_invstd), "conv_pw_bwd_bn: dx_stats needs in_mean / in_invstd");
  NASSEG_REQUIRE(!dx_res || (!dx_stats && K % 4 == 0), "conv_pw_bwd_bn: dx_res needs K %% 4 == 0 and no dx_stats");
  NASSEG_REQUIRE((!bn_train || (bn_mean && bn_invstd && bn_sums)) && (!bn_act || bn_shift),
                 "conv_pw_bwd_bn: missing BatchNorm tensors");
  NASSEG_REQUIRE(B > 0 && H > 0 && W > 0, "conv_pw_bwd_bn: bad geometry");
  const int64_t M = (int64_t)B * H * W;
  PwPlan p = pw_plan(M, N, K);
  NASSEG_REQUIRE(p.ok, "conv_pw_bwd_bn: no fused kernel for K=%d N=%d", K, N);
  if (!z) {
    // the conv's output was never stored (nasseg_irdw_fwd): the kernel must rebuild it, whatever the plan prefers
    NASSEG_REQUIRE(!p.wide && p.kt <= 2 && p.nt <= 9 && pw_weight_in_lds(p.nt, p.kt),
                   "conv_pw_bwd_bn: z == NULL, but K=%d N=%d has no kernel that rebuilds z", K, N);
    p.rz = 1;
  }
  PwArgs a = {};
  a.x = x; a.g = g; a.z = z; a.dx = dx; a.partial = ws; a.wb = wb;
  a.in_scale = in_scale; a.in_shift = in_shift; a.in_act = in_act;
  a.bn_scale = bn_scale; a.bn_shift = bn_shift; a.bn_mean = bn_mean; a.bn_invstd = bn_invstd;
  a.bn_sums = bn_sums; a.bn_train = bn_train; a.bn_act = bn_act;
  a.invM = (float)(1.0 / (double)M);
  NASSEG_REQUIRE(!dx_act || dx_act == in_act,
                 "conv_pw_bwd_bn: dx can only be masked with the derivative of the input activation");
  a.dx_act = dx_act;
  a.in_mean = in_mean; a.in_invstd = in_invstd; a.dx_stats = dx_stats; a.dx_res = dx_res;
  a.K = K; a.N = N; a.KP = (K + 15) & ~15; a.NP = (N + 15) & ~15;
  a.M = (int)M; a.pix_per_slab = p.pix_per_slab;
  const bool pro = in_scale || in_shift || in_act;
  hipStream_t s = (hipStream_t)stream;
  if (p.wide) {
    NASSEG_REQUIRE(!dx_stats, "conv_pw_bwd_bn: dx_stats is not available for K > 64");
    const size_t ldsw2 = ((size_t)4 * kPwTile * kW2LS + 4 * 64 + 2 * p.kt * kPwChunk) * sizeof(float);
#define PW_WIDE(KC_)                                                                                         \
  do {                                                                                                       \
    if (ldsw2 > (64 << 10)) { /* above the default limit of dynamic LDS (per device: set on every launch) */ \
      (void)hipFuncSetAttribute((const void*)conv_pw_bwd_wide_kernel<KC_, true>,                            \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsw2);                     \
      (void)hipFuncSetAttribute((const void*)conv_pw_bwd_wide_kernel<KC_, false>,                           \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsw2);                     \
    }                                                                                                        \
    if (pro) hipLaunchKernelGGL((conv_pw_bwd_wide_kernel<KC_, true>), dim3(p.nslab), dim3(256), ldsw2, s, a); \
    else hipLaunchKernelGGL((conv_pw_bwd_wide_kernel<KC_, false>), dim3(p.nslab), dim3(256), ldsw2, s, a);    \
  } while (0)
    if (p.kt <= 2) PW_WIDE(2);
    else if (p.kt == 3) PW_WIDE(3);
    else if (p.kt == 4) PW_WIDE(4);
    else if (p.kt == 5) PW_WIDE(5);
    else PW_WIDE(6);
#undef PW_WIDE
    NASSEG_LAUNCH_CHECK("conv_pw_bwd_wide_kernel");
    if (!dw) return NASSEG_OK;
    const float* parts_w[1] = {ws};
    float* outs_w[1] = {dw};
    const int dims_w[5] = {p.nslab, 1, N, K, 0};
    return nasseg_wgrad_finalize_many(1, parts_w, outs_w, dims_w, stream);
  }
#define PW_CASE(NT_, KT_) if (p.nt == NT_ && p.kt == KT_ && p.rz) pw_launch<NT_, KT_, true>(a, p.nslab, pro, s); else
  PW_CASE(2, 1) PW_CASE(3, 1) PW_CASE(4, 1) PW_CASE(6, 1) PW_CASE(2, 2) PW_CASE(3, 2) PW_CASE(4, 2) PW_CASE(6, 2)
  PW_CASE(9, 2)  // (24 -> 144: slower than loading z by itself, faster than storing z for it - only when z == NULL)
#undef PW_CASE
#define PW_CASE(NT_, KT_) if (p.nt == NT_ && p.kt == KT_) pw_launch<NT_, KT_>(a, p.nslab, pro, s); else
  PW_CASE(2, 1) PW_CASE(3, 1) PW_CASE(4, 1) PW_CASE(6, 1) PW_CASE(9, 1) PW_CASE(12, 1)
  PW_CASE(2, 2) PW_CASE(3, 2) PW_CASE(4, 2) PW_CASE(6, 2) PW_CASE(9, 2) PW_CASE(12, 2)
  PW_CASE(2, 4) PW_CASE(3, 4) PW_CASE(4, 4) PW_CASE(6, 4)
  return nasseg_fail(NASSEG_ERR_UNSUPPORTED, "conv_pw_bwd_bn: no kernel for nt=%d kt=%d", p.nt, p.kt);
#undef PW_CASE
  NASSEG_LAUNCH_CHECK("conv_pw_bwd_kernel");
  if (!dw) return NASSEG_OK;
  const float* parts[1] = {ws};
  float* outs[1] = {dw};
  const int dims[5] = {p.nslab, 1, N, K, 0};
  return nasseg_wgrad_finalize_many(1, parts, outs, dims, stream);
}

}  // extern "C"
