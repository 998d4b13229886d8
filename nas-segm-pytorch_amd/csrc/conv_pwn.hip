// Pointwise (1x1, stride 1) convolution, forward and backward-data, as a PERSISTENT kernel whose
// workgroups stream the input through LDS and split the OUTPUT CHANNELS among their waves - NHWC, fp32
// matrix cores (v_mfma_f32_16x16x4_f32), gfx950.
//
// Reference call sites: the 1x1 convs of SepConv / DilConv / InvertedResidual / Pool / Adapt /
// ConcatReduce (src/nn/layer_factory.py:125-382) and pre_clf (src/nn/micro_decoders.py:355-358), forward
// and the gradient w.r.t. their inputs.
//
// Why a third pointwise kernel.  conv_fwd_kernel gives every wave one tile and hides every latency with
// other resident waves: its expanding calls (N > K: 16->96, 24->144, 32->192, the backward-data of every
// contracting conv) live 1-2 us per workgroup, hold 4 KB of loads in flight per workgroup and run at
// 3.1-3.7 TB/s.  conv_pw_kernel is persistent and prefetches, but every wave owns ALL output channels of
// its pixels: 9-14 accumulator tiles per wave, a cross-lane statistics reduction per tile and channel
// tile, and it only wins where the reduction is long and the output narrow.  Here:
//   * a workgroup owns tiles of 64 pixels and walks them with a stride of the grid (persistent: 256 CUs x
//     2-3 workgroups); the stream it consumes is the flattened sequence of (tile, 16-channel k-block)
//     pairs, 4 KB each: one float4 per thread, loaded with all lanes on one contiguous run of a pixel row;
//   * FOUR k-blocks are always in flight per workgroup (a ring of four float4 registers per thread),
//     whatever K is; a landed k-block is written to an LDS ring of eight slots, chunk-major
//     ([k-chunk][pixel], 66 slots per row: conflict-free ds_write_b128, near conflict-free
//     ds_read_b128) and multiplied from there by ALL waves - one workgroup barrier per four k-blocks;
//   * the waves split N (and, for N <= 64, the pixels too): wave w owns the 32-channel units w, w+4 of all
//     64 pixels (N > 64), or a 2 x 2 / 4 x 1 (pixels x units) split for narrower outputs - at most four
//     channel tiles per wave however wide the output, weights from LDS ([N][K+4]);
//   * the per-channel statistics (BatchNorm batch statistics of the layer that follows, or the
//     BatchNorm-backward sums of the layer in front, conv_fwd.hip STATS 1 / 2) stay in PER-LANE registers
//     for the whole kernel and are reduced across lanes ONCE, at its end: no shuffles per tile, one row
//     per workgroup;
//   * the input prologue (scale * x + shift, activation) is applied once per element as a k-block is
//     written to LDS, not once per wave that multiplies it;
//   * stores cover FULL 128-byte lines.  The MFMA result layout gives a lane 4 channels of one pixel, so a
//     store instruction of one 16-channel tile writes 64-byte segments (16 pixels x 64 B) - and 64-byte
//     segments are what caps the write-heavy calls: measured on this part, 2.6 TB/s of such writes against
//     5.2-5.6 TB/s for whole lines (tools/membench.hip; 16->96 ran at 3.0 TB/s with 86 % writes while its
//     backward, 46 % writes, reached 4.5).  A wave therefore owns channel tiles in adjacent PAIRS (a "unit"
//     = 32 channels = 128 B of a pixel row); lanes j and j + 8 of a 16-lane row exchange one tile each
//     (row_ror:8, four v_mov_dpp per subtile and unit) and every store instruction writes 8 pixels x 128 B.
// Per accumulator the MFMA order (k-blocks ascending, the four components of a lane's float4 in turn) is
// that of conv_fwd_kernel: convolution outputs are bit-identical to it; statistics differ in the rounding
// of their partial sums only.
#include "conv_args.h"

#include <atomic>
#include <type_traits>

extern "C" int nasseg_conv_pwn_mode(int v);

namespace {

constexpr int kTP = 64;     // pixels per tile
#ifndef NASSEG_PWN_RING
#define NASSEG_PWN_RING 4
#endif
constexpr int kRing = NASSEG_PWN_RING;  // k-blocks in flight per workgroup
constexpr int kSlots = 2 * kRing;       // LDS ring: two groups of kRing
constexpr int kRow = 66;    // float4 slots per chunk row of a k-block in LDS: [4][kRow]
constexpr int kSlotF = 4 * kRow * 4;  // floats per LDS slot

#ifdef NASSEG_BF16
typedef uint2 raw4_t;  // four bf16 as loaded
__device__ __forceinline__ raw4_t ld_raw(const bf16_t* p) { return *reinterpret_cast<const uint2*>(p); }
__device__ __forceinline__ float4 cvt_raw(raw4_t u) {
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                     __uint_as_float(u.y & 0xffff0000u));
}
#else
typedef float4 raw4_t;
__device__ __forceinline__ raw4_t ld_raw(const float* p) { return ld4(p); }
__device__ __forceinline__ float4 cvt_raw(raw4_t u) { return u; }
#endif

__device__ __forceinline__ float4 row_swap8(float4 v) {  // lanes j <-> j ^ 8 of every 16-lane row
  v.x = dpp_mov<0x128>(v.x);  // row_ror:8
  v.y = dpp_mov<0x128>(v.y);
  v.z = dpp_mov<0x128>(v.z);
  v.w = dpp_mov<0x128>(v.w);
  return v;
}
__device__ __forceinline__ float4 sel4(bool c, float4 a, float4 b) {
  return make_float4(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w);
}

// PS: how many ways the four waves split the 64 pixels of a tile (1, 2, 4); they split the 32-channel units of
// N 4 / PS ways.  MTW = 4 / PS subtiles of 16 pixels per wave, NTU units (NTW = 2 NTU channel tiles) per wave:
// tile nt of a wave is channel tile 2 * (nw + NS * (nt / 2)) + nt % 2.
template <int PS, int NTU, int STATS>
__global__ __launch_bounds__(256, 2) void conv_pwn_kernel(FwdArgs a) {
  constexpr int MTW = 4 / PS;
  constexpr int NS = 4 / PS;
  constexpr int NTW = 2 * NTU;
  constexpr bool kSums = STATS == 1 || STATS == 2;
  extern __shared__ float smem[];
  const int K = a.K, N = a.N;
  const int KP = (K + 15) & ~15, LSK = KP + 4, nkb = KP >> 4;
  const int tiles_n = (N + 15) >> 4;
  float* xs = smem;                      // [kSlots][4][kRow] float4
  float* wl = xs + kSlots * kSlotF;      // [tiles_n * 16][LSK]: w[n][k], zero beyond N / K
  float* psc = wl + tiles_n * 16 * LSK;  // [KP] prologue scale | [KP] shift
  float* psh = psc + KP;
  float* sred = psh + KP;                // [4 waves][2][NTW * 16] (PS > 1: the pixel parts meet here)
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, kg = lane >> 4;
  const int pw = PS == 1 ? 0 : (PS == 2 ? (wave >> 1) : wave);  // this wave's pixel part
  const int nw = PS == 1 ? wave : (PS == 2 ? (wave & 1) : 0);   // ... and channel part
  const int Mtot = a.g.B * a.g.Ho * a.g.Wo;
  const int ntiles = (Mtot + kTP - 1) / kTP;

  for (int it = tid; it < tiles_n * 16 * (KP >> 2); it += 256) {
    const int n = it / (KP >> 2), k = (it - n * (KP >> 2)) * 4;
    const float4 v = keep_if(ld4(a.w + (int64_t)(n < N ? n : 0) * K + (k < K ? k : 0)), n < N && k < K);
    *reinterpret_cast<float4*>(&wl[n * LSK + k]) = v;
  }
  const bool pro = a.in_scale || a.in_shift || a.in_act;
  for (int k = tid; k < KP; k += 256) {
    psc[k] = (a.in_scale && k < K) ? a.in_scale[k] : 1.f;
    psh[k] = (a.in_shift && k < K) ? a.in_shift[k] : 0.f;
  }
  __syncthreads();
  const ActSel pact = act_sel(a.in_act);
  const ActSel bact = act_sel(a.b_act);

  // ---- the loader: thread -> (pixel lp of the tile, 4-channel chunk lq of the k-block) ----
  const int lp = tid >> 2, lq = tid & 3;
  const int my_tiles = blockIdx.x < ntiles ? (ntiles - 1 - blockIdx.x) / gridDim.x + 1 : 0;
  const int total = my_tiles * nkb;  // k-blocks this workgroup consumes
  int l_tile = blockIdx.x, l_kb = 0;
  raw4_t ring[kRing];
  auto issue = [&](raw4_t& dst) {
    // (past the end of the stream: a valid address, never multiplied)
    const int tc = l_tile < ntiles ? l_tile : ntiles - 1;
    const int m = tc * kTP + lp;
    const int k = l_kb * 16 + lq * 4;
    dst = ld_raw(a.x + (int64_t)(m < Mtot ? m : Mtot - 1) * a.ldx + (k < K ? k : 0));
    if (++l_kb == nkb) {
      l_kb = 0;
      l_tile += gridDim.x;
    }
  };
#pragma unroll
  for (int u = 0; u < kRing; ++u) issue(ring[u]);

  // what the k-block written by this thread needs of the prologue: the chunk's channels k0 .. k0+3, where
  // k0 = 16 * (k-block within the tile) + 4 * lq; tracked per ring position
  int w_kb = 0;
  auto land = [&](const raw4_t& src, float* slot) {
    float4 v = cvt_raw(src);
    const int k = w_kb * 16 + lq * 4;
    if (pro) {
      const float4 sc = *reinterpret_cast<const float4*>(&psc[k]);
      const float4 sh = *reinterpret_cast<const float4*>(&psh[k]);
      v = act_apply4(fma4(v, sc, sh), pact);
    }
    v = keep_if(v, k < K);
    *reinterpret_cast<float4*>(&slot[(lq * kRow + lp) * 4]) = v;
    if (++w_kb == nkb) w_kb = 0;
  };

  f32x4 acc[MTW][NTW];
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // Forward statistics (sum y, sum y^2) run in DOUBLE: the variance comes out of E[y^2] - E[y]^2, and a channel
  // whose |mean| is hundreds of standard deviations (a 1x1 conv over a nearly constant map: seen at 471 in a
  // controller-sampled cell) loses every digit of it to fp32 sums - two fp32 evaluation orders differed by 21 % in
  // the variance there.  The adds hide behind the MFMAs (half-rate VALU, 3 per output element).  The
  // BatchNorm-backward sums (STATS 2) have no such cancellation and stay fp32.
  typedef typename std::conditional<STATS == 1, double, float>::type sum_t;
  sum_t sx[NTW][4], sq[NTW][4];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) sx[nt][r] = sq[nt][r] = (sum_t)0;

  // this wave's channel tiles and its operand offsets
  int woff[NTW];
  bool tvalid[NTW];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) {
    const int tn = 2 * (nw + NS * (nt >> 1)) + (nt & 1);
    tvalid[nt] = tn < tiles_n;  // wave-uniform
    woff[nt] = ((tvalid[nt] ? tn : 0) * 16 + j) * LSK + kg * 4;
  }
  const int xoff = (kg * kRow + pw * MTW * 16 + j) * 4;

  int c_tile = blockIdx.x, c_kb = 0;  // the multiplier's position in the stream
  int group = 0;
  for (int s0 = 0; s0 < total; s0 += kRing) {
    float* gbase = xs + group * kRing * kSlotF;
#pragma unroll
    for (int u = 0; u < kRing; ++u) {
      land(ring[u], gbase + u * kSlotF);
      issue(ring[u]);
    }
    __syncthreads();
    const int nb = total - s0 < kRing ? total - s0 : kRing;
    for (int u = 0; u < nb; ++u) {
      const float* slot = gbase + u * kSlotF;
      float4 bv[MTW];
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt) bv[mt] = *reinterpret_cast<const float4*>(&slot[xoff + mt * 64]);
      float4 av[NTW];
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) av[nt] = *reinterpret_cast<const float4*>(&wl[woff[nt] + c_kb * 16]);
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        if (!tvalid[nt]) continue;
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
          acc[mt][nt] = mfma16(av[nt].x, bv[mt].x, acc[mt][nt]);
          acc[mt][nt] = mfma16(av[nt].y, bv[mt].y, acc[mt][nt]);
          acc[mt][nt] = mfma16(av[nt].z, bv[mt].z, acc[mt][nt]);
          acc[mt][nt] = mfma16(av[nt].w, bv[mt].w, acc[mt][nt]);
        }
      }
      if (++c_kb < nkb) continue;
      // ---- the tile is complete: lane holds pixel j of each of its subtiles, channels 4*kg + {0..3} of each tile ----
      c_kb = 0;
      const int m_base = c_tile * kTP + pw * MTW * 16;
      c_tile += gridDim.x;
      int pm[MTW];
      bool pok[MTW];
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt) {
        const int m = m_base + mt * 16 + j;
        pok[mt] = m < Mtot;
        pm[mt] = pok[mt] ? m : Mtot - 1;
      }
      const float* e_sc = STATS == 0 ? a.out_scale : a.b_scale;
      const float* e_sh = STATS == 0 ? a.out_shift : a.b_shift;
      const float* e_mu = a.b_mean;
      const float* e_is = a.b_invstd;
      asm volatile("" : "+s"(e_sc), "+s"(e_sh), "+s"(e_mu), "+s"(e_is));
      const bool lo = j < 8;
#pragma unroll
      for (int u2 = 0; u2 < NTU; ++u2) {
        if (!tvalid[2 * u2]) continue;  // (wave-uniform: the unit lies beyond N)
        // per-channel vectors of the unit's two tiles (this lane's channels 4*kg .. 4*kg+3 of each)
        int nch[2];
        bool nok[2];
        float4 v_sc[2], v_sh[2], v_mu[2], v_is[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          nch[h] = (2 * (nw + NS * u2) + h) * 16 + kg * 4;
          nok[h] = nch[h] < N;
          const int nc = nok[h] ? nch[h] : 0;
          v_sc[h] = make_float4(1.f, 1.f, 1.f, 1.f);
          v_sh[h] = v_mu[h] = v_is[h] = f4zero();
          if (STATS == 0 || STATS >= 2) {
            if (e_sc) v_sc[h] = ld4(e_sc + nc);
            if (e_sh) v_sh[h] = ld4(e_sh + nc);
          }
          if (STATS == 2) {
            v_mu[h] = ld4(e_mu + nc);
            v_is[h] = ld4(e_is + nc);
          }
        }
        // lanes j < 8 keep tile 0 of their pixel and take tile 0 of pixel j + 8; lanes j >= 8 keep tile 1 of
        // their pixel and take tile 1 of pixel j - 8: every store below covers 8 pixels x 128 contiguous bytes
        const int ns = lo ? nch[0] : nch[1];
        const bool nsok = ns < N;
        // everything the unit reads besides the accumulators is requested before the first use (one round trip
        // for the unit instead of one per subtile and tile)
        float4 zv[2][MTW];
        if (STATS >= 2 || STATS == 0) {
          const act_t* zsrc = STATS == 0 ? a.res : a.bz;
          const int zld = STATS == 0 ? a.ldres : a.ldbz;
          if (STATS >= 2 || zsrc) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
              for (int mt = 0; mt < MTW; ++mt)
                zv[h][mt] = lda4(zsrc + (int64_t)pm[mt] * zld + (nok[h] ? nch[h] : 0));
          }
        }
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
          float4 fin[2];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int nt = 2 * u2 + h;
            f32x4 c = acc[mt][nt];
            acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (STATS == 0) {
              float4 o = make_float4(c[0], c[1], c[2], c[3]);
              if (e_sc || e_sh) o = fma4(o, v_sc[h], v_sh[h]);
              if (a.out_act) o = act_apply4(o, a.out_act);
              if (a.res) o = add4(o, zv[h][mt]);
              fin[h] = o;
            } else if (STATS >= 2) {
              const float zz[4] = {zv[h][mt].x, zv[h][mt].y, zv[h][mt].z, zv[h][mt].w};
              const float bsc[4] = {v_sc[h].x, v_sc[h].y, v_sc[h].z, v_sc[h].w};
              const float bsh[4] = {v_sh[h].x, v_sh[h].y, v_sh[h].z, v_sh[h].w};
              const float bmu[4] = {v_mu[h].x, v_mu[h].y, v_mu[h].z, v_mu[h].w};
              const float bis[4] = {v_is[h].x, v_is[h].y, v_is[h].z, v_is[h].w};
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float g = c[r] * act_mask(fmaf(zz[r], bsc[r], bsh[r]), bact);
                c[r] = g;
                if (STATS == 2) {
                  const float v = keep_if(g, pok[mt] && nok[h]);
                  sx[nt][r] += (sum_t)v;
                  sq[nt][r] += (sum_t)(v * ((zz[r] - bmu[r]) * bis[r]));
                }
              }
              fin[h] = make_float4(c[0], c[1], c[2], c[3]);
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const sum_t v = (sum_t)keep_if(c[r], pok[mt]);
                sx[nt][r] += v;
                sq[nt][r] += v * v;
              }
              fin[h] = make_float4(c[0], c[1], c[2], c[3]);
            }
          }
          const float4 got = row_swap8(sel4(lo, fin[1], fin[0]));
          const float4 d0 = sel4(lo, fin[0], got);  // pixel (j & 7)
          const float4 d1 = sel4(lo, got, fin[1]);  // pixel (j & 7) + 8
          const int m0 = m_base + mt * 16 + (j & 7);
          // (a.y == null: the statistics-only pass in front of nasseg_irdw_fwd - nothing is stored)
          if (a.y && nsok && m0 < Mtot) sta4(a.y + (int64_t)m0 * a.ldy + ns, d0);
          if (a.y && nsok && m0 + 8 < Mtot) sta4(a.y + (int64_t)(m0 + 8) * a.ldy + ns, d1);
        }
      }
    }
    group ^= 1;
  }

  if (kSums) {
    // one cross-lane reduction per kernel: over the 16 pixel lanes of a k-group with DPP adds, over the pixel
    // parts of the workgroup (PS > 1) through LDS in a fixed order.  STATS 1 (double sums) leaves TWO rows per
    // workgroup - the sum rounded to fp32 and what the rounding dropped - which the finaliser adds in fp64 like any
    // other rows: the fp32 row format costs the statistics nothing.
    sum_t* red = reinterpret_cast<sum_t*>(sred);
    sum_t* my_red = red + wave * 2 * NTW * 16;
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const sum_t tx = row16_allsum(sx[nt][r]);
        const sum_t tq = row16_allsum(sq[nt][r]);
        if (j == 0) {
          my_red[nt * 16 + kg * 4 + r] = tx;
          my_red[NTW * 16 + nt * 16 + kg * 4 + r] = tq;
        }
      }
    }
    __syncthreads();
    // thread t < NS * NTW * 16 writes channel c of tile nt of channel part q
    for (int t = tid; t < NS * NTW * 16; t += 256) {
      const int q = t / (NTW * 16), rem = t - q * (NTW * 16);
      const int nt = rem >> 4, c = rem & 15;
      const int n = (2 * (q + NS * (nt >> 1)) + (nt & 1)) * 16 + c;
      if (n < N) {
        sum_t vx = (sum_t)0, vq = (sum_t)0;
#pragma unroll
        for (int p = 0; p < PS; ++p) {
          const int wv = PS == 1 ? q : (PS == 2 ? (p * 2 + q) : p);
          vx += red[wv * 2 * NTW * 16 + rem];
          vq += red[wv * 2 * NTW * 16 + NTW * 16 + rem];
        }
        if (STATS == 1) {
          float* po = a.stats + (int64_t)blockIdx.x * 4 * N + n;
          const float hx = (float)vx, hq = (float)vq;
          po[0] = hx;
          po[N] = hq;
          po[2 * N] = (float)((double)vx - (double)hx);
          po[3 * N] = (float)((double)vq - (double)hq);
        } else {
          float* po = a.stats + (int64_t)blockIdx.x * 2 * N + n;
          po[0] = (float)vx;
          po[N] = (float)vq;
        }
      }
    }
  }
}

template <int PS, int NTU, int STATS>
int launch_one(const FwdArgs& a, const PwnPlan& p, hipStream_t s) {
  static std::atomic<int> raised{0};
  if (p.lds > (size_t)(64 << 10) && !raised.load()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pwn_kernel<PS, NTU, STATS>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 << 10) != hipSuccess) {
      (void)hipGetLastError();
      return nasseg_fail(NASSEG_ERR_LAUNCH, "conv_pwn_kernel: cannot raise the dynamic LDS limit");
    }
    raised.store(1);
  }
  hipLaunchKernelGGL((conv_pwn_kernel<PS, NTU, STATS>), dim3(p.grid), dim3(256), p.lds, s, a);
  NASSEG_LAUNCH_CHECK("conv_pwn_kernel");
  return NASSEG_OK;
}

template <int PS, int NTU>
int launch_stats(const FwdArgs& a, const PwnPlan& p, int stats, hipStream_t s) {
  switch (stats) {
    case 0: return launch_one<PS, NTU, 0>(a, p, s);
    case 1: return launch_one<PS, NTU, 1>(a, p, s);
    case 2: return launch_one<PS, NTU, 2>(a, p, s);
    default: return launch_one<PS, NTU, 3>(a, p, s);
  }
}

#if NASSEG_FP32_ONLY
// 0: never; 1: where it measured faster (pwn_auto); 2: every call it supports
#ifndef NASSEG_PWN_MODE
#define NASSEG_PWN_MODE 1
#endif
std::atomic<int> g_pwn_mode{NASSEG_PWN_MODE};
#endif

}  // namespace

#if NASSEG_FP32_ONLY
// which calls take this kernel when the mode is "auto": tools/kbench_pwn.py on MI355X, us today -> here.  Forward:
// 16->96 @4x512x1024 264 -> 197, 24->144 @256x512 117 -> 92, 32->192 @128x256 41 -> 36, 32->32 @512x1024 121 -> 102,
// 64->64 @256x512 70 -> 58, 64->32 49 -> 42, 128->64 122 -> 114; backward-data (with the BatchNorm-backward sums):
// 16->96 466 -> 322, 64->128 @256x512 213 -> 151, 32->64 102 -> 66, 24->96 136 -> 92, 24->144 186 -> 164, 64->224
// 426 -> 348, 32->192 @128x256 76 -> 57.  It loses where its weight leaves room for ONE workgroup per CU only and the
// call is a forward one (224->64 @256x512: 201 -> 239 against conv_pw_kernel's 192; 144->24 is a tie at 86).
static bool pwn_auto(const PwnPlan& p, int resident, int mode) { return resident >= 2 || mode == 2; }

PwnPlan nasseg_internal_pwn_plan(int64_t M, int N, int K, int mode) {
  PwnPlan p = {};
  const int md = nasseg_conv_pwn_mode(-1);
  if (md == 0 || N <= 0 || K <= 0 || (N & 3) || (K & 3) || N > 256 || K > 512) return p;
  const int tiles = cdiv(N, 16);
  const int units = cdiv(N, 32);  // pairs of channel tiles: 128 bytes of a pixel row
  if (units >= 3) {
    p.ps = 1;
    p.ntw = cdiv(units, 4);  // (units per wave)
  } else if (units == 2) {
    p.ps = 2;
    p.ntw = 1;
  } else {
    p.ps = 4;
    p.ntw = 1;
  }
  p.mtw = 4 / p.ps;
  const int KP = (K + 15) & ~15;
  p.lds = ((size_t)kSlots * kSlotF + (size_t)tiles * 16 * (KP + 4) + 2 * KP + 2 * 4 * 2 * 2 * p.ntw * 16) * sizeof(float);
  if (p.lds > (size_t)(128 << 10)) return p;
  int r = p.ntw <= 1 ? 3 : 2;  // resident workgroups per CU by registers
  const int by_lds = (int)((size_t)(160 << 10) / p.lds);
  if (r > by_lds) r = by_lds;
  if (r < 1) return p;
  if (md == 1 && !pwn_auto(p, r, mode)) return p;
  const int64_t ntiles = cdiv64(M, kTP);
  p.grid = (int)(ntiles < 256LL * r ? ntiles : 256LL * r);
  p.ok = 1;
  return p;
}

extern "C" int nasseg_conv_pwn_mode(int v) {
  if (v < 0) return g_pwn_mode.load();
  return g_pwn_mode.exchange(v > 2 ? 2 : v);
}
#endif  // NASSEG_FP32_ONLY

int NASSEG_INTERNAL(pwn_launch)(const FwdArgs& a, const PwnPlan& p, int stats, hipStream_t s) {
  if (p.ps == 1) return p.ntw == 1 ? launch_stats<1, 1>(a, p, stats, s) : launch_stats<1, 2>(a, p, stats, s);
  if (p.ps == 2) return launch_stats<2, 1>(a, p, stats, s);
  return launch_stats<4, 1>(a, p, stats, s);
}
