// Dense convolution backward-weight on the fp32 matrix cores, NHWC, gfx950:
//   dW[n][k][tap] = sum_pixel dY[pixel][n] * X[src(pixel, tap)][k]
//
// Reference: autograd of the nn.Conv2d call sites listed in conv_fwd.hip.
//
// The reduction axis (pixels) is the slow axis of both operands, so no transposition is
// needed: lanes load VN floats along n (from dY) and VK floats along k (from X) straight
// from HBM into MFMA operands, and the rows/cols of the 16x16 MFMA tile are a permutation
// of (n, k) (row i <-> n = n0 + VN*i + comp).  4 pixels per MFMA; each wave walks a
// contiguous pixel range 16 pixels at a time (8 vector loads in flight per lane).
// Per-wave accumulators are combined through LDS in a fixed order, per-workgroup partials
// [slab][tap][N][K] by a second deterministic pass (reduce_partials16).
// "Flat" mode gathers (tap, k) of a small-K conv (the 3-channel stem) as one axis.
#include <vector>

#include "conv_common.h"

namespace {

// 1: the 4 x 4 form parks its waves' accumulators one wave at a time and is held to 128 registers - four workgroups
// per CU instead of three; 0: as rounds 1-5 (A/B: tools/gpu.sh flags)
#ifndef NASSEG_WGRAD_STAGED
#define NASSEG_WGRAD_STAGED 1
#endif
constexpr int wg_min_waves(int vn, int vk) { return (NASSEG_WGRAD_STAGED && vn * vk == 16) ? 4 : 1; }
constexpr int kWgAtOnce44 = 256 * (NASSEG_WGRAD_STAGED ? 4 : 3);  // workgroups of the 4 x 4 form resident at once

struct WgArgs {
  const act_t* x;  // activations: fp32 or bf16 storage (common.h)   // forward input  [B][Hs][Ws][ldx], K channels
  int ldx;
  const act_t* dy;  // grad of forward output [B][Ho][Wo][lddy], N channels
  int lddy;
  float* partial;
  const float* in_scale;
  const float* in_shift;
  int in_act;
  int K, N;
  int kchunks;     // number of k chunks (16*VK wide)
  int pix_per_block;  // multiple of 64
  ConvGeom g;      // non-transposed forward geometry
  // BN variant (nasseg_conv_wgrad_bn): dy is the gradient w.r.t. the BatchNorm output (activation
  // mask applied), z the conv's own raw output; the second half of the BatchNorm backward runs on
  // load and its result is also written to dz for the backward-data kernel that follows
  const act_t* z;
  act_t* dz;
  int ldz, lddz, bn_train, bn_act;
  float invM;
  const float* bn_scale;
  const float* bn_shift;
  const float* bn_mean;
  const float* bn_invstd;
  const float* bn_sums;  // [2][N]: sum g, sum g*xhat
};

template <int V, bool AL>
__device__ __forceinline__ void store_vec(act_t* p, int i0, int len, const float* v) {
  if (AL && V == 4) {
    if (i0 < len) sta4(p + i0, make_float4(v[0], v[1], v[2], v[3]));
  } else {
#pragma unroll
    for (int c = 0; c < V; ++c)
      if (i0 + c < len) sta1(p + i0 + c, v[c]);
  }
}

// out[0..V) = p[i0..i0+V) from a clamped address; caller masks.  AL: V-aligned vector load.
#ifdef NASSEG_BF16
template <int V, bool AL>
__device__ __forceinline__ void load_vec(const bf16_t* p, int i0, int len, float* out) {
  if (AL && V == 4) {
    const float4 v = lda4(p + (i0 < len ? i0 : 0));
    out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
  } else if (AL && V == 2) {
    const uint32_t u = *reinterpret_cast<const uint32_t*>(p + (i0 < len ? i0 : 0));
    out[0] = __uint_as_float(u << 16);
    out[1] = __uint_as_float(u & 0xffff0000u);
  } else {
#pragma unroll
    for (int c = 0; c < V; ++c) out[c] = lda1(p + (i0 + c < len ? i0 + c : 0));
  }
}
#endif
template <int V, bool AL>
__device__ __forceinline__ void load_vec(const float* p, int i0, int len, float* out) {
  if (AL && V == 4) {
    const float4 v = lda4(p + (i0 < len ? i0 : 0));
    out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
  } else if (AL && V == 2) {
    const float2 v = *reinterpret_cast<const float2*>(p + (i0 < len ? i0 : 0));
    out[0] = v.x; out[1] = v.y;
  } else {
#pragma unroll
    for (int c = 0; c < V; ++c) out[c] = p[i0 + c < len ? i0 + c : 0];
  }
}

// VN / VK: floats per lane along n / k (chunk = 16*V); ALN / ALK: aligned vector loads;
// GATHER: per-tap source-pixel arithmetic; FLAT: (tap, k) is one gathered axis; PRO: the
// forward had an input prologue (affine + activation on x), re-applied on load.
// (bx, by, bz): the workgroup's coordinates in the layer's own grid (slab, n/k chunk, tap) - the
// grouped launch below runs several layers' grids side by side in one kernel
template <int VN, int VK, bool ALN, bool ALK, bool GATHER, bool FLAT, bool PRO, bool BN = false>
__device__ __forceinline__ void wgrad_tile(const WgArgs& a, const int bx, const int by, const int bz) {
  constexpr int NACC = VN * VK * 4;
  // waves 1..3 park their accumulators here - the 4 x 4 form one wave at a time (a third of the LDS: 16.6 instead of
  // 49.9 KB, which alone held the kernel at three workgroups per CU for the whole of its main loop)
  constexpr bool kStaged = NASSEG_WGRAD_STAGED && NACC == 64;
  __shared__ float red[kStaged ? 1 : 3][NACC][65];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int li = lane & 15;  // row/col index inside a 16-wide MFMA tile
  const int pk = lane >> 4;  // pixel slot 0..3
  const int ntaps = a.g.kh * a.g.kw;
  const int tap = FLAT ? 0 : bz;
  const int ty = tap / a.g.kw, tx = tap - ty * a.g.kw;
  const int nchunk = by / a.kchunks;
  const int kchunk = by - nchunk * a.kchunks;
  const int n0 = nchunk * 16 * VN + li * VN;
  const int k0 = kchunk * 16 * VK + li * VK;
  const int Kq = FLAT ? ntaps * a.K : a.K;
  const int Mtot = a.g.B * a.g.Ho * a.g.Wo;
  const int p_begin = bx * a.pix_per_block + wave * (a.pix_per_block >> 2);
  int p_end = p_begin + (a.pix_per_block >> 2);
  if (p_end > Mtot) p_end = Mtot;

  f32x4 acc[VN][VK];
#pragma unroll
  for (int i = 0; i < VN; ++i)
#pragma unroll
    for (int jj = 0; jj < VK; ++jj) acc[i][jj] = (f32x4){0.f, 0.f, 0.f, 0.f};

  bool nok[VN], kok[VK];
#pragma unroll
  for (int c = 0; c < VN; ++c) nok[c] = n0 + c < a.N;
#pragma unroll
  for (int c = 0; c < VK; ++c) kok[c] = k0 + c < Kq;
  // flat mode: the (tap, channel) each of this lane's k' elements stands for
  int fty[VK], ftx[VK], fcin[VK];
  if (FLAT) {
#pragma unroll
    for (int c = 0; c < VK; ++c) {
      const int kq = kok[c] ? k0 + c : 0;
      const int tp = kq / a.K;
      fcin[c] = kq - tp * a.K;
      fty[c] = tp / a.g.kw;
      ftx[c] = tp - fty[c] * a.g.kw;
    }
  }
  const ActSel pact = act_sel(a.in_act);
  float ps[VK], ph[VK];
  if (PRO) {
#pragma unroll
    for (int c = 0; c < VK; ++c) {
      const int kc = kok[c] ? k0 + c : 0;
      ps[c] = a.in_scale ? a.in_scale[kc] : 1.f;
      ph[c] = a.in_shift ? a.in_shift[kc] : 0.f;
    }
  }

  // dz = ca*g' + cb*z + cd  ==  scale*(g' - sum(g')/M - xhat*sum(g'*xhat)/M), xhat = (z - mean)*invstd,
  // g' = g * act'(scale*z + shift) (bn_act != 0: g arrives without its activation mask)
  float ca[VN], cb[VN], cd[VN], cs[VN];
  const bool wr = BN && a.dz && kchunk == 0 && tap == 0;  // every dz element is written by exactly one lane
  if (BN) {
#pragma unroll
    for (int c = 0; c < VN; ++c) {
      const int nc = nok[c] ? n0 + c : 0;
      const float sc = a.bn_scale[nc];
      ca[c] = sc;
      cb[c] = 0.f;
      cd[c] = 0.f;
      cs[c] = a.bn_act ? a.bn_shift[nc] : 0.f;
      if (a.bn_train) {
        const float is = a.bn_invstd[nc], mu = a.bn_mean[nc];
        const float s0 = a.bn_sums[nc] * a.invM, s1 = a.bn_sums[a.N + nc] * a.invM;
        cb[c] = -sc * is * s1;
        cd[c] = sc * (mu * is * s1 - s0);
      }
    }
  }

  // Loop bounds are wave-uniform; loads are unconditional from clamped addresses and
  // masked afterwards, so the 2*U vector loads of an iteration are issued back to back.
  constexpr int U = 4;
  for (int g0 = p_begin; g0 < p_end; g0 += 4 * U) {
    float dv[U][VN], xv[U][VK];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int p = g0 + u * 4 + pk;
      const bool pok = p < p_end;
      const int pc = pok ? p : p_end - 1;
      load_vec<VN, ALN>(a.dy + (int64_t)pc * a.lddy, n0, a.N, dv[u]);
      if (BN) {
        float zv[VN];
        load_vec<VN, ALN>(a.z + (int64_t)pc * a.ldz, n0, a.N, zv);
#pragma unroll
        for (int c = 0; c < VN; ++c) {
          float gm = dv[u][c];
          if (a.bn_act) gm *= act_mask(fmaf(zv[c], ca[c], cs[c]), a.bn_act);
          float v = fmaf(gm, ca[c], fmaf(zv[c], cb[c], cd[c]));
#ifdef NASSEG_BF16
          v = bf16_to_f32(f32_to_bf16(v));  // (the value the backward-data kernel will read)
#endif
          dv[u][c] = v;
        }
        if (wr && pok) store_vec<VN, ALN>(a.dz + (int64_t)pc * a.lddz, n0, a.N, dv[u]);
      }
      if (FLAT) {
        const int ox = pc % a.g.Wo;
        const int t = pc / a.g.Wo;
        const int oy = t % a.g.Ho;
        const int b = t / a.g.Ho;
#pragma unroll
        for (int c = 0; c < VK; ++c) {
          const int sp = src_pixel(a.g, b, oy, ox, fty[c], ftx[c]);
          const float v = lda1(a.x + (int64_t)(sp < 0 ? 0 : sp) * a.ldx + fcin[c]);
          xv[u][c] = keep_if(v, pok && kok[c] && sp >= 0);
        }
#pragma unroll
        for (int c = 0; c < VN; ++c) dv[u][c] = keep_if(dv[u][c], pok && nok[c]);
      } else {
        int sp = pc;
        if (GATHER) {
          const int ox = pc % a.g.Wo;
          const int t = pc / a.g.Wo;
          const int oy = t % a.g.Ho;
          const int b = t / a.g.Ho;
          sp = src_pixel(a.g, b, oy, ox, ty, tx);
        }
        const bool ok = pok && sp >= 0;
        load_vec<VK, ALK>(a.x + (int64_t)(sp < 0 ? 0 : sp) * a.ldx, k0, a.K, xv[u]);
#pragma unroll
        for (int c = 0; c < VK; ++c) {
          float v = xv[u][c];
          if (PRO) v = act_apply(fmaf(v, ps[c], ph[c]), pact);
          xv[u][c] = keep_if(v, ok && kok[c]);
        }
#pragma unroll
        for (int c = 0; c < VN; ++c) dv[u][c] = keep_if(dv[u][c], ok && nok[c]);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int ca = 0; ca < VN; ++ca)
#pragma unroll
        for (int cb = 0; cb < VK; ++cb) acc[ca][cb] = mfma16(dv[u][ca], xv[u][cb], acc[ca][cb]);
  }
  // cross-wave reduction through LDS (fixed order -> deterministic)
  if constexpr (kStaged) {
    // wave 0 adds waves 1, 2, 3 in turn: ((w0 + w1) + w2) + w3, the order of the one-pass form below
    for (int w = 1; w < 4; ++w) {
      if (wave == w) {
#pragma unroll
        for (int ca = 0; ca < VN; ++ca)
#pragma unroll
          for (int cb = 0; cb < VK; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[0][(ca * VK + cb) * 4 + r][lane] = acc[ca][cb][r];
      }
      __syncthreads();
      if (wave == 0) {
#pragma unroll
        for (int ca = 0; ca < VN; ++ca)
#pragma unroll
          for (int cb = 0; cb < VK; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[ca][cb][r] += red[0][(ca * VK + cb) * 4 + r][lane];
      }
      __syncthreads();
    }
  } else {
    if (wave > 0) {
#pragma unroll
      for (int ca = 0; ca < VN; ++ca)
#pragma unroll
        for (int cb = 0; cb < VK; ++cb)
#pragma unroll
          for (int r = 0; r < 4; ++r) red[wave - 1][(ca * VK + cb) * 4 + r][lane] = acc[ca][cb][r];
    }
    __syncthreads();
  }
  if (wave == 0) {
    float* pout = a.partial + (((int64_t)bx * (FLAT ? 1 : ntaps) + tap) * a.N) * Kq;
#pragma unroll
    for (int ca = 0; ca < VN; ++ca)
#pragma unroll
      for (int cb = 0; cb < VK; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int e = (ca * VK + cb) * 4 + r;
          const float v = kStaged ? acc[ca][cb][r]
                                  : acc[ca][cb][r] + red[0][e][lane] + red[kStaged ? 0 : 1][e][lane] +
                                        red[kStaged ? 0 : 2][e][lane];
          // D row i = 4*pk + r  <-> n ; D col = li <-> k
          const int n = nchunk * 16 * VN + VN * (4 * pk + r) + ca;
          const int k = kchunk * 16 * VK + VK * li + cb;
          if (n < a.N && k < Kq) pout[(int64_t)n * Kq + k] = v;
        }
  }
}

template <int VN, int VK, bool ALN, bool ALK, bool GATHER, bool FLAT, bool PRO>
__global__ __launch_bounds__(256, wg_min_waves(VN, VK)) void conv_wgrad_kernel(WgArgs a) {
  wgrad_tile<VN, VK, ALN, ALK, GATHER, FLAT, PRO>(a, blockIdx.x, blockIdx.y, blockIdx.z);
}
// pointwise conv followed by a BatchNorm: the BatchNorm's backward is applied to dy on load
// (its 4 x 4 form stays at three waves per SIMD: sixteen more registers of BatchNorm constants would spill under 128)
template <int VN, int VK, bool PRO>
__global__ __launch_bounds__(256) void conv_wgrad_bn_kernel(WgArgs a) {
  wgrad_tile<VN, VK, true, true, false, false, PRO, true>(a, blockIdx.x, blockIdx.y, blockIdx.z);
}

// small-K k x k conv (flat mode: the 3-channel stem) followed by a BatchNorm, when nothing needs dz:
// the BatchNorm backward is applied to dy on load and never stored
template <int VN, int VK, bool ALN>
__global__ __launch_bounds__(256) void conv_wgrad_bn_flat_kernel(WgArgs a) {
  wgrad_tile<VN, VK, ALN, false, true, true, false, true>(a, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Several layers of ONE specialisation in one launch: weight gradients of small maps are
// latency-bound launches that use a fraction of the GPU each and that nothing but the optimiser
// waits for - deferred to the end of backward and grouped they run concurrently.
constexpr int kWgGroup = 8;
struct WgGroup {
  int n;
  int start[kWgGroup + 1];  // first flattened workgroup of each layer
  int nslab[kWgGroup];
  int gy[kWgGroup];
  WgArgs a[kWgGroup];
};
template <int VN, int VK, bool ALN, bool ALK, bool GATHER, bool FLAT, bool PRO>
__global__ __launch_bounds__(256, wg_min_waves(VN, VK)) void conv_wgrad_group_kernel(WgGroup g) {
  int d = 0;
  while (d + 1 < g.n && (int)blockIdx.x >= g.start[d + 1]) ++d;
  const int local = blockIdx.x - g.start[d];
  const int bx = local % g.nslab[d];
  const int t = local / g.nslab[d];
  wgrad_tile<VN, VK, ALN, ALK, GATHER, FLAT, PRO>(g.a[d], bx, t % g.gy[d], t / g.gy[d]);
}

// dw (N,K,kh,kw) = sum over slabs of partial.  flat == 0: partial[slab][tap][N][K];
// flat != 0: partial[slab][N][tap*K + k]
__global__ __launch_bounds__(256) void conv_wgrad_finalize(const float* __restrict__ partial,
                                                           float* __restrict__ dw, int nslab,
                                                           int ntaps, int N, int K, int flat) {
  __shared__ double red[NASSEG_RP_SLICES][NASSEG_RP_ELEMS + 1];
  const int64_t per = (int64_t)ntaps * N * K;
  const int64_t i = (int64_t)blockIdx.x * NASSEG_RP_ELEMS + rp_elem();
  const bool valid = i < per;
  const double s = reduce_partials16(partial, nslab, per, i, valid, red);
  if (valid && rp_slice() == 0) {
    int n, k, tap;
    if (!flat) {  // i = (tap*N + n)*K + k
      k = (int)(i % K);
      const int64_t t = i / K;
      n = (int)(t % N);
      tap = (int)(t / N);
    } else {  // i = n*(ntaps*K) + tap*K + k
      const int Kq = ntaps * K;
      n = (int)(i / Kq);
      const int kq = (int)(i - (int64_t)n * Kq);
      tap = kq / K;
      k = kq - tap * K;
    }
    dw[((int64_t)n * K + k) * ntaps + tap] = (float)s;
  }
}

// The second stage of MANY backward-weight reductions in one launch (dense and depthwise: a
// depthwise weight is the case N = C, K = 1).  Weight gradients are consumed by the optimiser
// only, so their finalisation can leave the backward chain and be batched: ~100 launches of a
// few microseconds each become a handful.
struct FinDesc {
  const float* partial;
  float* dw;
  int nslab, ntaps, N, K, flat;
};
constexpr int kFinMax = 16;
struct FinTable {
  FinDesc d[kFinMax];
};
// elements of a layer one workgroup finalises: with at most 32 slabs (large weights: few, long
// rows) one thread adds the slabs of one element in order - coalesced rows, and exactly the sum
// reduce_partials16 forms when every slice holds at most one row; otherwise 8 elements x 32 slices
// ... otherwise 8 threads x 32 slices, a thread taking 4 consecutive elements (one 16-byte load
// per row; per % 4 == 0) or one.  Every element is summed in the order of reduce_partials16 in
// all three shapes.
__host__ __device__ inline int fin_elems_per_block(int nslab, int64_t per) {
  if (nslab <= NASSEG_RP_SLICES) return 256;
  return (per & 3) == 0 ? 4 * NASSEG_RP_ELEMS : NASSEG_RP_ELEMS;
}
// reduce_partials16 for the 4 consecutive elements e4 .. e4+3 (16-byte aligned rows)
__device__ __forceinline__ void reduce_partials16x4(const float* __restrict__ partial, int nblk, int64_t per,
                                                    int64_t e4, bool valid, double (*red)[NASSEG_RP_ELEMS + 1][4],
                                                    double* tot) {
  const int slice = rp_slice();
  const int el = rp_elem();
  double s[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) s[i][r] = 0.0;
  if (valid) {
    int b = slice;
    for (; b + 7 * NASSEG_RP_SLICES < nblk; b += 8 * NASSEG_RP_SLICES) {
      float4 v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = ld4(partial + (int64_t)(b + i * NASSEG_RP_SLICES) * per + e4);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        s[i][0] += (double)v[i].x;
        s[i][1] += (double)v[i].y;
        s[i][2] += (double)v[i].z;
        s[i][3] += (double)v[i].w;
      }
    }
    for (; b < nblk; b += NASSEG_RP_SLICES) {
      const float4 v = ld4(partial + (int64_t)b * per + e4);
      s[0][0] += (double)v.x;
      s[0][1] += (double)v.y;
      s[0][2] += (double)v.z;
      s[0][3] += (double)v.w;
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
    red[slice][el][r] = ((s[0][r] + s[1][r]) + (s[2][r] + s[3][r])) + ((s[4][r] + s[5][r]) + (s[6][r] + s[7][r]));
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) tot[r] = 0.0;
  if (slice == 0) {
#pragma unroll
    for (int i = 0; i < NASSEG_RP_SLICES; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) tot[r] += red[i][el][r];
  }
  __syncthreads();
}
__device__ __forceinline__ void fin_store(const FinDesc& d, int64_t i, double s) {
  int n, k, tap;
  if (!d.flat) {
    k = (int)(i % d.K);
    const int64_t q = i / d.K;
    n = (int)(q % d.N);
    tap = (int)(q / d.N);
  } else {
    const int Kq = d.ntaps * d.K;
    n = (int)(i / Kq);
    const int kq = (int)(i - (int64_t)n * Kq);
    tap = kq / d.K;
    k = kq - tap * d.K;
  }
  d.dw[((int64_t)n * d.K + k) * d.ntaps + tap] = (float)s;
}
__global__ __launch_bounds__(256) void wgrad_finalize_many(FinTable t) {
  __shared__ double red4[NASSEG_RP_SLICES][NASSEG_RP_ELEMS + 1][4];
  const FinDesc d = t.d[blockIdx.y];
  const int64_t per = (int64_t)d.ntaps * d.N * d.K;
  const bool direct = d.nslab <= NASSEG_RP_SLICES;
  if ((int64_t)blockIdx.x * fin_elems_per_block(d.nslab, per) >= per) return;  // (uniform over the workgroup)
  if (!direct && (per & 3) == 0) {
    const int64_t e4 = ((int64_t)blockIdx.x * NASSEG_RP_ELEMS + rp_elem()) * 4;
    double tot[4];
    reduce_partials16x4(d.partial, d.nslab, per, e4, e4 < per, red4, tot);
    if (e4 < per && rp_slice() == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) fin_store(d, e4 + r, tot[r]);
    }
    return;
  }
  int64_t i;
  bool valid;
  double s = 0.0;
  if (direct) {
    i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    valid = i < per;
    if (valid)
      for (int b = 0; b < d.nslab; ++b) s += (double)d.partial[(int64_t)b * per + i];
  } else {
    double(*red)[NASSEG_RP_ELEMS + 1] = reinterpret_cast<double(*)[NASSEG_RP_ELEMS + 1]>(&red4[0][0][0]);
    i = (int64_t)blockIdx.x * NASSEG_RP_ELEMS + rp_elem();
    valid = i < per;
    s = reduce_partials16(d.partial, d.nslab, per, i, valid, red);
    valid = valid && rp_slice() == 0;
  }
  if (valid) fin_store(d, i, s);
}
struct WgMode {
  bool aln, alk, gather, flat, pro;
};

template <int VN, int VK>
int launch_wgrad(const WgArgs& a, dim3 grid, const WgMode& m, hipStream_t s) {
#define GO(ALN_, ALK_, G_, F_, P_)                                                              \
  hipLaunchKernelGGL((conv_wgrad_kernel<VN, VK, ALN_, ALK_, G_, F_, P_>), grid, dim3(256), 0, s, a)
  if (m.flat) {
    if (m.aln) GO(true, false, true, true, false);
    else GO(false, false, true, true, false);
  } else if (m.pro) {
    // only reachable for pointwise convs with aligned channels (checked by the caller)
    GO(true, true, false, false, true);
  } else if (m.aln && m.alk) {
    if (m.gather) GO(true, true, true, false, false);
    else GO(true, true, false, false, false);
  } else if (m.aln) {
    GO(true, false, true, false, false);
  } else if (m.alk) {
    GO(false, true, true, false, false);
  } else {
    GO(false, false, true, false, false);
  }
#undef GO
  NASSEG_LAUNCH_CHECK("conv_wgrad_kernel");
  return NASSEG_OK;
}

template <int VN, int VK>
int launch_wgrad_group(const WgGroup& g, const WgMode& m, hipStream_t s) {
  const dim3 grid(g.start[g.n]);
#define GO(ALN_, ALK_, G_, F_, P_)                                                                \
  hipLaunchKernelGGL((conv_wgrad_group_kernel<VN, VK, ALN_, ALK_, G_, F_, P_>), grid, dim3(256), 0, s, g)
  if (m.flat) {
    if (m.aln) GO(true, false, true, true, false);
    else GO(false, false, true, true, false);
  } else if (m.pro) {
    GO(true, true, false, false, true);
  } else if (m.aln && m.alk) {
    if (m.gather) GO(true, true, true, false, false);
    else GO(true, true, false, false, false);
  } else if (m.aln) {
    GO(true, false, true, false, false);
  } else if (m.alk) {
    GO(false, true, true, false, false);
  } else {
    GO(false, false, true, false, false);
  }
#undef GO
  NASSEG_LAUNCH_CHECK("conv_wgrad_group_kernel");
  return NASSEG_OK;
}


// ---------------------------------------------------------------------------
// 3x3 stride-1 weight gradient with the operands staged in LDS (the class heads: 64 -> 19 / 21 at
// 256x512, src/nn/micro_decoders.py:215,226,363).  The generic kernel above runs one workgroup per
// (slab, tap) and so reads x and dy nine times - the 134 MB input of the head does not fit the L2s, the
// re-reads come from the Infinity Cache / HBM, and with N = 19 unaligned every dy load is scalar:
// 335 us at 4x256x512, 22 % of the fp32 MFMA rate.  Here a workgroup owns a 16-channel slice of K and a
// strided set of 8 x 32-pixel output tiles; per tile the (8+2d) x (32+2d) input patch of the slice and
// the dy tile (transposed to [n][pixel]) are staged ONCE, all nine taps are computed from LDS, and the
// next tile's operands are in flight in registers meanwhile (nothing but LDS is read in the MFMA phase,
// so the prefetch is not waited for - loads return in order).
//   dW[tap][n][k] += sum over the tile's pixels of dy[p][n] * x[p + tap][k]
// MFMA: rows = n (A = dy, two 16-row tiles for N <= 32), cols = the slice's 16 k (B = x), four pixels per
// instruction; each wave takes two of the tile's eight rows and keeps all 9 x 2 accumulators across
// tiles; waves meet in LDS in a fixed order at the end, slabs in the deterministic second stage.
// ---------------------------------------------------------------------------
// LDS strides: every operand read is a ds_read_b32 (lane groups of 32 = two of the MFMA's four pixel slots x 16
// rows / columns, bank = dword address mod 32): the patch's pixel stride of 16 dwords puts the two pixels on the
// two halves of the banks, dyT's row stride of 258 = 2 mod 32 puts row li, pixel slot pk on bank 2*li + pk
constexpr int kW3TH = 8, kW3TW = 32, kW3KS = 16, kW3XS = kW3KS;       // tile, k slice, pixel stride of the patch
constexpr int kW3SP = kW3TH * kW3TW + 2;                              // row stride of dyT[n][pixel]
constexpr int kW3MaxX = ((kW3TH + 4) * (kW3TW + 4) * (kW3KS / 4) + 255) / 256;  // float4 of the patch per thread (d <= 2)
constexpr int kW3MaxN = 32;
// 0: 17 ... 21 rows of n as two MFMA tiles (rounds 3-4; A/B, tools/gpu.sh flags)
#ifndef NASSEG_W3_VALU_TAIL
#define NASSEG_W3_VALU_TAIL 1
#endif

struct W3Args {
  const act_t* x;
  const act_t* dy;
  float* partial;  // [nslab][9][N][K]
  int B, H, W, K, Ho, Wo, N, pad, dil, nslab, tiles_x, tiles_y;
};

// NV > 0 (N = 16 * NT + NV, the class heads' 19 / 21): the NV rows behind the full tiles are not given a second, mostly
// empty MFMA tile but are accumulated on the vector ALU from the x operand the MFMAs read anyway: lane (k column li,
// pixel slot pk) multiplies it with dy[pixel][16 NT + c] (one LDS address per pixel slot: a broadcast read) into 9 x NV
// partial sums, which the four pixel slots add up once at the end (as conv3x3_lds_kernel does in the forward).
template <int NT, int NV = 0>  // 16-row tiles of n: 1 (N <= 16 + NV) or 2
__global__ __launch_bounds__(256, 2) void conv_wgrad3x3_lds_kernel(W3Args a) {
  constexpr int NVa = NV ? NV : 1;
  extern __shared__ float smem[];
  const int dil = a.dil;
  const int TR = kW3TH + 2 * dil, TC = kW3TW + 2 * dil;
  float* xs = smem;                                      // [TR * TC][kW3XS]
  float* dyT = xs + (kW3TH + 4) * (kW3TW + 4) * kW3XS;   // [16 * NT][kW3SP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, pk = lane >> 4;
  const int slab = blockIdx.x, k0 = blockIdx.y * kW3KS;
  const int N = a.N, K = a.K;
  const int ntiles = a.B * a.tiles_y * a.tiles_x;
  const int xtotal = TR * TC * (kW3KS / 4);

  f32x4 acc[9][NT];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float pvw[9][NVa];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int c = 0; c < NVa; ++c) pvw[t][c] = 0.f;

  // patch item it of this thread: float4 (tid & 3) of patch pixel (pr, pc) - the same for every tile; what
  // does not depend on the tile is computed once (the staging code is VALU work that the MFMAs of the
  // CU's other workgroup do not hide: the two run in lockstep)
  int xloc[kW3MaxX], xg[kW3MaxX], xl[kW3MaxX];
#pragma unroll
  for (int it = 0; it < kW3MaxX; ++it) {
    int idx = tid + 256 * it;
    idx = idx < xtotal ? idx : xtotal - 1;
    const int p = idx >> 2;
    const int pr = p / TC, pc = p - pr * TC;
    xloc[it] = pr | (pc << 8);
    xg[it] = (pr * a.W + pc) * K;              // offset from the patch's first pixel (interior tiles)
    xl[it] = p * kW3XS + (tid & 3) * 4;
  }
  // dy: thread tid owns pixel (row tid / 32, column tid % 32) of the tile and loads its N channels (76 bytes
  // for N = 19: not vectorisable, the lanes of a wave are 76 bytes apart); written transposed to dyT[n][pixel].
  // (Measured alternative: the tile as 8 contiguous runs of 32 * N floats, one float per lane - fully coalesced,
  //  but the (row, pixel, channel) bookkeeping per element costs more than the scattered loads: 260 / 224 us.)
  const int dr_r = tid >> 5, dr_c = tid & 31;
  // rows of dyT beyond N are read by the MFMAs and never written: zero them once
  for (int n = N; n < 16 * NT + NV; ++n) dyT[n * kW3SP + tid] = 0.f;
  const int tpi = a.tiles_y * a.tiles_x;
  const float inv_tpi = 1.f / (float)tpi, inv_tx = 1.f / (float)a.tiles_x;
  auto fdiv = [](int n, int d, float inv) {  // n / d for 0 <= n < 2^22 (one float multiply and a fix-up)
    int q = (int)((float)n * inv);
    q -= (q * d > n) ? 1 : 0;
    q += ((q + 1) * d <= n) ? 1 : 0;
    return q;
  };
  float4 xr[kW3MaxX];
  float dr[16 * NT + NV];
  struct TileGeo {
    int b, oy0, ox0;
    bool inner;  // the patch and the dy tile lie inside the image: no clamping, no masks
  };
  auto geo = [&](int tile) {
    TileGeo g;
    g.b = fdiv(tile, tpi, inv_tpi);
    const int t2 = tile - g.b * tpi;
    const int ty = fdiv(t2, a.tiles_x, inv_tx);
    g.oy0 = ty * kW3TH;
    g.ox0 = (t2 - ty * a.tiles_x) * kW3TW;
    const int iy0 = g.oy0 - a.pad, ix0 = g.ox0 - a.pad;
    g.inner = iy0 >= 0 && ix0 >= 0 && iy0 + TR <= a.H && ix0 + TC <= a.W && g.oy0 + kW3TH <= a.Ho &&
              g.ox0 + kW3TW <= a.Wo;
    return g;
  };
  auto issue = [&](const TileGeo& g) {
    const int iy0 = g.oy0 - a.pad, ix0 = g.ox0 - a.pad;
    const act_t* xb = a.x + (int64_t)g.b * a.H * a.W * K + k0 + (tid & 3) * 4;
    if (g.inner) {  // (uniform)
      const act_t* x0 = xb + ((int64_t)iy0 * a.W + ix0) * K;
#pragma unroll
      for (int it = 0; it < kW3MaxX; ++it) xr[it] = lda4(x0 + xg[it]);
      const act_t* dp = a.dy + (((int64_t)g.b * a.Ho + g.oy0 + dr_r) * a.Wo + g.ox0 + dr_c) * N;
#pragma unroll
      for (int n = 0; n < 16 * NT + NV; ++n)
        if (n < N) dr[n] = lda1(dp + n);  // (uniform: the 13 surplus loads of N = 19 would each touch 38 lines)
    } else {
#pragma unroll
      for (int it = 0; it < kW3MaxX; ++it) {
        int iy = iy0 + (xloc[it] & 255), ix = ix0 + (xloc[it] >> 8);
        iy = iy < 0 ? 0 : (iy >= a.H ? a.H - 1 : iy);
        ix = ix < 0 ? 0 : (ix >= a.W ? a.W - 1 : ix);
        xr[it] = lda4(xb + ((int64_t)iy * a.W + ix) * K);
      }
      int oy = g.oy0 + dr_r, ox = g.ox0 + dr_c;
      oy = oy < a.Ho ? oy : a.Ho - 1;
      ox = ox < a.Wo ? ox : a.Wo - 1;
      const act_t* dp = a.dy + (((int64_t)g.b * a.Ho + oy) * a.Wo + ox) * N;
#pragma unroll
      for (int n = 0; n < 16 * NT + NV; ++n)
        if (n < N) dr[n] = lda1(dp + n);
    }
  };
  auto place = [&](const TileGeo& g) {
    if (g.inner) {  // (uniform)
#pragma unroll
      for (int it = 0; it < kW3MaxX; ++it)
        if (tid + 256 * it < xtotal) *reinterpret_cast<float4*>(&xs[xl[it]]) = xr[it];
#pragma unroll
      for (int n = 0; n < 16 * NT + NV; ++n)
        if (n < N) dyT[n * kW3SP + tid] = dr[n];
    } else {
      const int iy0 = g.oy0 - a.pad, ix0 = g.ox0 - a.pad;
#pragma unroll
      for (int it = 0; it < kW3MaxX; ++it) {
        const int iy = iy0 + (xloc[it] & 255), ix = ix0 + (xloc[it] >> 8);
        const bool ok = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        if (tid + 256 * it < xtotal) *reinterpret_cast<float4*>(&xs[xl[it]]) = keep_if(xr[it], ok);
      }
      const bool pok = g.oy0 + dr_r < a.Ho && g.ox0 + dr_c < a.Wo;
#pragma unroll
      for (int n = 0; n < 16 * NT + NV; ++n)
        if (n < N) dyT[n * kW3SP + tid] = keep_if(dr[n], pok);
    }
  };

  int tile = slab;
  TileGeo gcur = geo(tile < ntiles ? tile : 0), gnext = gcur;
  if (tile < ntiles) issue(gcur);
  for (; tile < ntiles; tile += a.nslab) {
    __syncthreads();  // (the previous tile's operands have been read)
    place(gcur);
    __syncthreads();
    if (tile + a.nslab < ntiles) {  // in flight during the MFMAs below
      gnext = geo(tile + a.nslab);
      issue(gnext);
    }
    // this wave's two rows of the tile: 64 pixels, four per MFMA.  The operands of step u + 1 are read from
    // LDS before the MFMAs of step u are issued (explicit ping-pong, scheduling barriers in between: left to
    // itself the compiler reads one tap, waits, issues its two MFMAs, reads the next - nine exposed LDS
    // latencies per step)
    auto load_ops = [&](int u, float (&av)[NT + NVa], float (&bv)[9]) {
      const int pl = 4 * u + pk;                  // pixel within the wave's two rows
      const int r = 2 * wave + (pl >> 5), c = pl & 31;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) av[nt] = dyT[(nt * 16 + li) * kW3SP + r * kW3TW + c];
#pragma unroll
      for (int v = 0; v < NV; ++v) av[NT + v] = dyT[(NT * 16 + v) * kW3SP + r * kW3TW + c];  // (the pixel slot's dy)
      const float* xp = xs + (r * TC + c) * kW3XS + li;
#pragma unroll
      for (int t = 0; t < 9; ++t) bv[t] = xp[((t / 3) * dil * TC + (t % 3) * dil) * kW3XS];
    };
    auto mma = [&](const float (&av)[NT + NVa], const float (&bv)[9]) {
#pragma unroll
      for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[t][nt] = mfma16(av[nt], bv[t], acc[t][nt]);
#pragma unroll
        for (int v = 0; v < NV; ++v) pvw[t][v] = fmaf(av[NT + v], bv[t], pvw[t][v]);
      }
    };
    float a0[NT + NVa], b0[9], a1[NT + NVa], b1[9];
    load_ops(0, a0, b0);
#pragma unroll 1
    for (int u = 0; u < 16; u += 2) {
      load_ops(u + 1, a1, b1);
      __builtin_amdgcn_sched_barrier(0);
      mma(a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      if (u + 2 < 16) load_ops(u + 2, a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      mma(a1, b1);
      __builtin_amdgcn_sched_barrier(0);
    }
    gcur = gnext;
  }
  if constexpr (NV > 0) {
    // the four pixel slots' partial sums of the vector-ALU rows: lanes li, li + 16, li + 32, li + 48 -> every lane
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        float q = pvw[t][v];
        q += __shfl_xor(q, 16, 64);
        q += __shfl_xor(q, 32, 64);
        pvw[t][v] = q;
      }
  }

  // waves -> workgroup partial, one tap at a time, fixed order (wave 0 + 1 + 2 + 3)
  __syncthreads();
  float* red = smem;  // [3 waves][NT tiles][64 lanes][4]
  float* pout = a.partial + (int64_t)slab * 9 * N * K;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    if (t) __syncthreads();
    if (wave > 0) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        *reinterpret_cast<f32x4*>(&red[(((wave - 1) * NT + nt) * 64 + lane) * 4]) = acc[t][nt];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const float v = ((acc[t][nt][rr] + red[((0 * NT + nt) * 64 + lane) * 4 + rr]) +
                           red[((1 * NT + nt) * 64 + lane) * 4 + rr]) + red[((2 * NT + nt) * 64 + lane) * 4 + rr];
          const int n = nt * 16 + 4 * pk + rr, k = k0 + li;  // D row <-> n, D col <-> k
          if (n < N && k < K) pout[((int64_t)t * N + n) * K + k] = v;
        }
    }
  }
  if constexpr (NV > 0) {
    // the vector-ALU rows: [3 waves][9 taps][NV][16 k], all taps at once, same fixed order
    __syncthreads();
    if (wave > 0 && pk == 0) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int v = 0; v < NV; ++v) red[(((wave - 1) * 9 + t) * NV + v) * 16 + li] = pvw[t][v];
    }
    __syncthreads();
    if (wave == 0 && pk == 0) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const float q = ((pvw[t][v] + red[((0 * 9 + t) * NV + v) * 16 + li]) + red[((1 * 9 + t) * NV + v) * 16 + li]) +
                          red[((2 * 9 + t) * NV + v) * 16 + li];
          const int n = NT * 16 + v, k = k0 + li;
          if (n < N && k < K) pout[((int64_t)t * N + n) * K + k] = q;
        }
    }
  }
}

// does nasseg_conv_wgrad run this call on the LDS-tiled kernel (shape AND call details)?
inline bool wgrad3x3_call_ok(int B, int Hs, int Ws, int K, int Ho, int Wo, int N, int kh, int kw, int stride, int pad,
                             int dil);
// is the LDS-tiled kernel the one for this geometry?  (the plan below gives such layers a slab count that
// suits it; everything else about the call - partial layout, second stage - is shared with the generic kernel)
inline bool wgrad3x3_shape_ok(int N, int K, int kh, int kw, int Ho, int Wo) {
  return kh == 3 && kw == 3 && N <= kW3MaxN && K % kW3KS == 0 && K >= kW3KS && Wo >= kW3TW && Ho >= kW3TH;
}

inline bool wgrad3x3_call_ok(int B, int Hs, int Ws, int K, int Ho, int Wo, int N, int kh, int kw, int stride, int pad,
                             int dil) {
  return wgrad3x3_shape_ok(N, K, kh, kw, Ho, Wo) && stride == 1 && dil >= 1 && dil <= 2 && pad >= 0 && pad <= 2 * dil &&
         Hs == Ho + 2 * dil - 2 * pad && Ws == Wo + 2 * dil - 2 * pad &&
         (int64_t)B * Ho * Wo >= (int64_t)64 * kW3TH * kW3TW;
}

inline int pick_v(int len) { return len > 32 ? 4 : (len > 16 ? 2 : 1); }

struct WgPlan {
  int vn, vk, nchunks, kchunks, nslab, flat, pix_per_block;
};
inline WgPlan wgrad_plan(int64_t Mtot, int N, int K, int taps, int lds3x3 = 0) {
  WgPlan p;
  p.flat = (taps > 1 && taps * K <= 64) ? 1 : 0;
  const int Kq = p.flat ? taps * K : K;
  p.vn = pick_v(N);
  p.vk = pick_v(Kq);
  p.nchunks = cdiv(N, 16 * p.vn);
  p.kchunks = cdiv(Kq, 16 * p.vk);
  const int64_t per = (int64_t)p.nchunks * p.kchunks * (p.flat ? 1 : taps);
  // ~1536 workgroups in total, >= 128 pixels each (measured: 256 costs the 32x64 maps 20 %,
  // 64 gains nothing more), partial buffer <= 16 MiB
  constexpr int kBlocks = 1536, kMinPix = 128;
  int64_t s = kBlocks / per;
  if (s < 1) s = 1;
  int64_t cap_bytes = (int64_t)(16 << 20) / ((int64_t)taps * N * K * 4);
  if (cap_bytes < 8) cap_bytes = 8;
  if (s > cap_bytes) s = cap_bytes;
  if (s > Mtot / kMinPix) s = Mtot / kMinPix;
  if (s < 1) s = 1;
  // The 4 x 4 form holds 49 KB of LDS: three workgroups per CU, 768 at once, and its workgroups live for the whole
  // launch - 1024 (64 x 128 at 4 x 256 x 512) or 990 (the CVPR cells' 3 x 3 64 -> 64 at 16 x 81 x 81) of them are a
  // full round and a third of one.  Whole rounds only: longer slabs instead of a partly filled last round (the
  // workgroups of a CU share its issue slots and its bandwidth - the same work in rounds that are all full is never
  // slower and, where latency bounds the kernel, faster: tools/kbench_wgrad.py, 128 -> 64 at 4 x 256 x 512 234 ->
  // 176 us, 3 x 3 64 -> 64 at 16 x 81 x 81 154 -> 127 us).
  if (p.vn == 4 && p.vk == 4) {
    constexpr int64_t kAtOnce = kWgAtOnce44;
    const int64_t wgs = s * per, full = wgs / kAtOnce;
    if (full >= 1 && full <= 3 && full * kAtOnce / per >= 1) s = full * kAtOnce / per;
  }
  // the LDS-tiled 3x3 kernel: (slabs x K / 16) workgroups, two resident per CU, several tiles each
  if (lds3x3 && Mtot >= (int64_t)64 * kW3TH * kW3TW) {
    s = 512 / (K / kW3KS);
    if (s < 32) s = 32;
    if (s > cap_bytes) s = cap_bytes;
  }
  int64_t ppb = cdiv64(Mtot, s);
  ppb = (ppb + 63) / 64 * 64;
  p.pix_per_block = (int)ppb;
  p.nslab = (int)cdiv64(Mtot, ppb);
  return p;
}

}  // namespace

extern "C" {

#if NASSEG_FP32_ONLY
// 1: nasseg_conv_wgrad runs this (dense, no input prologue) call on its LDS-tiled 3x3 kernel - whose sums are
// ordered differently from the generic kernel's, so a caller that wants the SAME bits from its immediate and
// its grouped (nasseg_conv_wgrad_many: always the generic kernel) code paths must not group such layers
int nasseg_conv_wgrad_lds3x3(int B, int Hs, int Ws, int K, int Ho, int Wo, int N, int kh, int kw, int stride,
                             int pad, int dil) {
  return wgrad3x3_call_ok(B, Hs, Ws, K, Ho, Wo, N, kh, kw, stride, pad, dil) ? 1 : 0;
}

// floats of workspace needed by nasseg_conv_wgrad
int64_t nasseg_conv_wgrad_workspace(int B, int Ho, int Wo, int N, int K, int kh, int kw) {
  WgPlan p = wgrad_plan((int64_t)B * Ho * Wo, N, K, kh * kw, wgrad3x3_shape_ok(N, K, kh, kw, Ho, Wo));
  return (int64_t)p.nslab * kh * kw * N * K;
}
#endif  // NASSEG_FP32_ONLY

// dw (N,K,kh,kw) = sum_pixels dy[pixel][n] * in_act(in_scale*x[src(pixel,tap)][k]+in_shift)
// dw == null: only the per-slab partial sums are produced (ws); nasseg_wgrad_finalize_many turns
// the partials of many layers into their gradients with one launch.
// (the input prologue is available for pointwise convs with K % 4 == 0 and N % 4 == 0)
// everything a launch of one layer needs: kernel arguments, specialisation, grid
struct WgSetup {
  WgArgs a;
  WgMode m;
  WgPlan p;
  dim3 grid;
};
static int wgrad_setup(WgSetup& u, const act_t* x, int ldx, const act_t* dy, int lddy, float* ws,
                       const float* in_scale, const float* in_shift, int in_act, int B, int Hs,
                       int Ws, int K, int Ho, int Wo, int N, int kh, int kw, int stride, int pad,
                       int dil) {
  NASSEG_REQUIRE(B > 0 && Hs > 0 && Ws > 0 && Ho > 0 && Wo > 0, "conv_wgrad: bad geometry");
  NASSEG_REQUIRE(K > 0 && N > 0 && ldx >= K && lddy >= N, "conv_wgrad: bad channels");
  NASSEG_REQUIRE((int64_t)B * Hs * Ws < 2147483647LL && (int64_t)B * Ho * Wo < 2147483647LL,
                 "conv_wgrad: too many pixels");
  const int64_t Mtot = (int64_t)B * Ho * Wo;
  const int taps = kh * kw;
  u.p = wgrad_plan(Mtot, N, K, taps, wgrad3x3_shape_ok(N, K, kh, kw, Ho, Wo));
  const WgPlan& p = u.p;
  WgMode& m = u.m;
  m.flat = p.flat != 0;
  m.pro = in_scale || in_shift || in_act;
  m.gather = !(kh == 1 && kw == 1 && stride == 1 && pad == 0 && Hs == Ho && Ws == Wo);
  m.aln = (N % p.vn == 0) && (lddy % p.vn == 0);
  m.alk = !m.flat && (K % p.vk == 0) && (ldx % p.vk == 0);
  NASSEG_REQUIRE(!m.pro || (!m.flat && !m.gather && m.aln && m.alk),
                 "conv_wgrad: the input prologue needs a pointwise conv with aligned channels");
  WgArgs& a = u.a;
  a.x = x; a.ldx = ldx; a.dy = dy; a.lddy = lddy; a.partial = ws;
  a.in_scale = in_scale; a.in_shift = in_shift; a.in_act = in_act;
  a.K = K; a.N = N;
  a.z = nullptr; a.dz = nullptr; a.ldz = a.lddz = 0; a.bn_train = a.bn_act = 0; a.invM = 0.f;
  a.bn_scale = a.bn_shift = a.bn_mean = a.bn_invstd = a.bn_sums = nullptr;
  a.kchunks = p.kchunks;
  a.pix_per_block = p.pix_per_block;
  a.g.B = B; a.g.Hs = Hs; a.g.Ws = Ws; a.g.Ho = Ho; a.g.Wo = Wo;
  a.g.kh = kh; a.g.kw = kw; a.g.stride = stride; a.g.pad = pad; a.g.dil = dil;
  a.g.transposed = 0;
  u.grid = dim3(p.nslab, p.nchunks * p.kchunks, p.flat ? 1 : taps);
  return NASSEG_OK;
}

int NASSEG_FN(conv_wgrad)(const act_t* x, int ldx, const act_t* dy, int lddy, float* dw, float* ws,
                      const float* in_scale, const float* in_shift, int in_act, int B, int Hs,
                      int Ws, int K, int Ho, int Wo, int N, int kh, int kw, int stride, int pad,
                      int dil, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  WgSetup u;
  int rc = wgrad_setup(u, x, ldx, dy, lddy, ws, in_scale, in_shift, in_act, B, Hs, Ws, K, Ho, Wo, N,
                       kh, kw, stride, pad, dil);
  if (rc) return rc;
  const WgPlan& p = u.p;
  if (wgrad3x3_call_ok(B, Hs, Ws, K, Ho, Wo, N, kh, kw, stride, pad, dil) && ldx == K && lddy == N && !u.m.pro) {
    W3Args w = {};
    w.x = x; w.dy = dy; w.partial = ws;
    w.B = B; w.H = Hs; w.W = Ws; w.K = K; w.Ho = Ho; w.Wo = Wo; w.N = N; w.pad = pad; w.dil = dil;
    w.nslab = p.nslab; w.tiles_x = cdiv(Wo, kW3TW); w.tiles_y = cdiv(Ho, kW3TH);
    const size_t lds = ((size_t)(kW3TH + 4) * (kW3TW + 4) * kW3XS + (size_t)kW3MaxN * kW3SP) * sizeof(float);
#define W3_NV(V_)                                                                                                   \
  do {                                                                                                              \
    (void)hipFuncSetAttribute((const void*)conv_wgrad3x3_lds_kernel<1, V_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                              (int)lds);                                                                            \
    hipLaunchKernelGGL((conv_wgrad3x3_lds_kernel<1, V_>), dim3(p.nslab, K / kW3KS), dim3(256), lds, s, w);          \
  } while (0)
    if (N <= 16) {
      (void)hipFuncSetAttribute((const void*)conv_wgrad3x3_lds_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds);
      hipLaunchKernelGGL(conv_wgrad3x3_lds_kernel<1>, dim3(p.nslab, K / kW3KS), dim3(256), lds, s, w);
    } else if (NASSEG_W3_VALU_TAIL && N <= 21) {  // 17 ... 21 rows: 16 on the matrix cores, the rest on the vector ALU
      switch (N - 16) {
        case 1: W3_NV(1); break;
        case 2: W3_NV(2); break;
        case 3: W3_NV(3); break;
        case 4: W3_NV(4); break;
        default: W3_NV(5); break;
      }
#undef W3_NV
    } else {
      (void)hipFuncSetAttribute((const void*)conv_wgrad3x3_lds_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds);
      hipLaunchKernelGGL(conv_wgrad3x3_lds_kernel<2>, dim3(p.nslab, K / kW3KS), dim3(256), lds, s, w);
    }
    NASSEG_LAUNCH_CHECK("conv_wgrad3x3_lds_kernel");
  } else {
#define WG_CASE(VN_, VK_) \
  if (p.vn == VN_ && p.vk == VK_) rc = launch_wgrad<VN_, VK_>(u.a, u.grid, u.m, s); else
  WG_CASE(4, 4) WG_CASE(4, 2) WG_CASE(4, 1) WG_CASE(2, 4) WG_CASE(2, 2) WG_CASE(2, 1)
  WG_CASE(1, 4) WG_CASE(1, 2) WG_CASE(1, 1)
  rc = nasseg_fail(NASSEG_ERR_UNSUPPORTED, "conv_wgrad: no kernel for vn=%d vk=%d", p.vn, p.vk);
#undef WG_CASE
  }
  if (rc) return rc;
  if (!dw) return NASSEG_OK;  // partial sums stay in ws for nasseg_wgrad_finalize_many
  const int taps = kh * kw;
  const int64_t per = (int64_t)taps * N * K;
  hipLaunchKernelGGL(conv_wgrad_finalize, dim3((unsigned)cdiv64(per, NASSEG_RP_ELEMS)), dim3(256), 0,
                     s, ws, dw, p.nslab, taps, N, K, p.flat);
  NASSEG_LAUNCH_CHECK("conv_wgrad_finalize");
  return NASSEG_OK;
}

// nasseg_conv_wgrad of a POINTWISE conv whose output z went through a BatchNorm (+ activation),
// fused with the second half of that BatchNorm's backward: g is the gradient w.r.t. the BatchNorm
// output - bn_act == 0: with the activation mask already applied (nasseg_conv_bwd_data_bn's output,
// or no activation at all); bn_act != 0: the mask act'(scale*z + shift) is applied here, on load -,
// sums = {sum g', sum g'*xhat} of the masked gradient per channel over the M = B*H*W pixels.  On load  dz = scale*(g - sums0/M - xhat*sums1/M)  (train; eval: dz = scale*g), which is
// what the weight gradient is computed from AND is written to dz for the backward-data call -
// nasseg_bn_bwd_apply without its own pass over g and z.  K % 4 == 0, N % 4 == 0.
int NASSEG_FN(conv_wgrad_bn)(const act_t* x, int ldx, const act_t* g, int ldg, const act_t* z, int ldz,
                             act_t* dz, int lddz, float* dw, float* ws, const float* in_scale,
                             const float* in_shift, int in_act, const float* bn_scale,
                             const float* bn_shift, const float* bn_mean, const float* bn_invstd,
                             const float* bn_sums, int bn_train, int bn_act, int B, int H, int W, int K,
                             int N, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  NASSEG_REQUIRE(z && dz && bn_scale && (!bn_train || (bn_mean && bn_invstd && bn_sums)) && (!bn_act || bn_shift),
                 "conv_wgrad_bn: missing BatchNorm tensors");
  NASSEG_REQUIRE(K % 4 == 0 && N % 4 == 0 && ldx % 4 == 0 && ldg % 4 == 0 && ldz % 4 == 0 && lddz % 4 == 0 &&
                     ldz >= N && lddz >= N,
                 "conv_wgrad_bn: channels must be multiples of 4");
  WgSetup u;
  int rc = wgrad_setup(u, x, ldx, g, ldg, ws, in_scale, in_shift, in_act, B, H, W, K, H, W, N, 1, 1, 1, 0, 1);
  if (rc) return rc;
  NASSEG_REQUIRE(u.m.aln && u.m.alk && !u.m.flat && !u.m.gather, "conv_wgrad_bn: unsupported geometry");
  u.a.z = z; u.a.ldz = ldz; u.a.dz = dz; u.a.lddz = lddz;
  u.a.bn_scale = bn_scale; u.a.bn_shift = bn_shift; u.a.bn_mean = bn_mean; u.a.bn_invstd = bn_invstd;
  u.a.bn_sums = bn_sums;
  u.a.bn_train = bn_train; u.a.bn_act = bn_act;
  u.a.invM = (float)(1.0 / ((double)B * H * W));
  const WgPlan& p = u.p;
#define WG_CASE(VN_, VK_)                                                                          \
  if (p.vn == VN_ && p.vk == VK_) {                                                                \
    if (u.m.pro) hipLaunchKernelGGL((conv_wgrad_bn_kernel<VN_, VK_, true>), u.grid, dim3(256), 0, s, u.a); \
    else hipLaunchKernelGGL((conv_wgrad_bn_kernel<VN_, VK_, false>), u.grid, dim3(256), 0, s, u.a); \
  } else
  WG_CASE(4, 4) WG_CASE(4, 2) WG_CASE(4, 1) WG_CASE(2, 4) WG_CASE(2, 2) WG_CASE(2, 1)
  WG_CASE(1, 4) WG_CASE(1, 2) WG_CASE(1, 1)
  rc = nasseg_fail(NASSEG_ERR_UNSUPPORTED, "conv_wgrad_bn: no kernel for vn=%d vk=%d", p.vn, p.vk);
#undef WG_CASE
  if (rc) return rc;
  NASSEG_LAUNCH_CHECK("conv_wgrad_bn_kernel");
  if (!dw) return NASSEG_OK;
  hipLaunchKernelGGL(conv_wgrad_finalize, dim3((unsigned)cdiv64((int64_t)N * K, NASSEG_RP_ELEMS)), dim3(256), 0,
                     s, ws, dw, p.nslab, 1, N, K, 0);
  NASSEG_LAUNCH_CHECK("conv_wgrad_finalize");
  return NASSEG_OK;
}

// Weight gradient of a small-K k x k conv (kh*kw*K <= 64: nasseg_conv_fwd_pack_mode == 2, the stem)
// followed by a BatchNorm, for the case that the conv's input needs no gradient: the BatchNorm backward
// (as in nasseg_conv_wgrad_bn) is applied to g on load and dz is never written - instead of a
// nasseg_bn_bwd_apply pass that writes dz and a weight-gradient kernel that reads it back.
// Geometry arguments as nasseg_conv_wgrad; ws: nasseg_conv_wgrad_workspace floats.
int NASSEG_FN(conv_wgrad_bn_flat)(const act_t* x, int ldx, const act_t* g, int ldg, const act_t* z, int ldz,
                                  float* dw, float* ws, const float* bn_scale, const float* bn_shift,
                                  const float* bn_mean, const float* bn_invstd, const float* bn_sums,
                                  int bn_train, int bn_act, int B, int Hs, int Ws, int K, int Ho, int Wo, int N,
                                  int kh, int kw, int stride, int pad, int dil, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  NASSEG_REQUIRE(z && bn_scale && (!bn_train || (bn_mean && bn_invstd && bn_sums)) && (!bn_act || bn_shift),
                 "conv_wgrad_bn_flat: missing BatchNorm tensors");
  NASSEG_REQUIRE(ldz >= N, "conv_wgrad_bn_flat: bad ldz");
  WgSetup u;
  int rc = wgrad_setup(u, x, ldx, g, ldg, ws, nullptr, nullptr, 0, B, Hs, Ws, K, Ho, Wo, N, kh, kw, stride, pad, dil);
  if (rc) return rc;
  NASSEG_REQUIRE(u.m.flat, "conv_wgrad_bn_flat: kh*kw*K = %d is not a flat geometry", kh * kw * K);
  const WgPlan& p = u.p;
  // (the z / g vectors are read VN at a time: both need the alignment)
  const bool aln = u.m.aln && (ldz % p.vn == 0);
  u.a.z = z; u.a.ldz = ldz; u.a.dz = nullptr; u.a.lddz = 0;
  u.a.bn_scale = bn_scale; u.a.bn_shift = bn_shift; u.a.bn_mean = bn_mean; u.a.bn_invstd = bn_invstd;
  u.a.bn_sums = bn_sums;
  u.a.bn_train = bn_train; u.a.bn_act = bn_act;
  u.a.invM = (float)(1.0 / ((double)B * Ho * Wo));
#define WG_CASE(VN_, VK_)                                                                                  \
  if (p.vn == VN_ && p.vk == VK_) {                                                                        \
    if (aln) hipLaunchKernelGGL((conv_wgrad_bn_flat_kernel<VN_, VK_, true>), u.grid, dim3(256), 0, s, u.a); \
    else hipLaunchKernelGGL((conv_wgrad_bn_flat_kernel<VN_, VK_, false>), u.grid, dim3(256), 0, s, u.a);    \
  } else
  WG_CASE(4, 4) WG_CASE(4, 2) WG_CASE(4, 1) WG_CASE(2, 4) WG_CASE(2, 2) WG_CASE(2, 1)
  WG_CASE(1, 4) WG_CASE(1, 2) WG_CASE(1, 1)
  rc = nasseg_fail(NASSEG_ERR_UNSUPPORTED, "conv_wgrad_bn_flat: no kernel for vn=%d vk=%d", p.vn, p.vk);
#undef WG_CASE
  if (rc) return rc;
  NASSEG_LAUNCH_CHECK("conv_wgrad_bn_flat_kernel");
  if (!dw) return NASSEG_OK;
  hipLaunchKernelGGL(conv_wgrad_finalize, dim3((unsigned)cdiv64((int64_t)N * K * kh * kw, NASSEG_RP_ELEMS)), dim3(256),
                     0, s, ws, dw, p.nslab, kh * kw, N, K, p.flat);
  NASSEG_LAUNCH_CHECK("conv_wgrad_finalize");
  return NASSEG_OK;
}

// The first stage (per-slab partial sums into each layer's ws) of `count` layers, grouped by
// kernel specialisation into launches of up to 8 layers that run side by side - for layers whose
// maps are too small to fill the GPU.  desc[20*i..] = x, ldx, dy, lddy, ws, in_scale, in_shift,
// in_act, B, Hs, Ws, K, Ho, Wo, N, kh, kw, stride, pad, dil: the arguments of nasseg_conv_wgrad
// (pointers as integers), which this call equals with dw == NULL for every layer; finish with
// nasseg_wgrad_finalize_many.
int NASSEG_FN(conv_wgrad_many)(int count, const int64_t* desc, void* stream) {
  NASSEG_REQUIRE(count >= 0 && (count == 0 || desc), "conv_wgrad_many: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (count == 0) return NASSEG_OK;
  std::vector<WgSetup> u((size_t)count);
  std::vector<int> key((size_t)count);
  std::vector<char> done((size_t)count, 0);
  for (int i = 0; i < count; ++i) {
    const int64_t* d = desc + 20 * (size_t)i;
    int rc = wgrad_setup(u[i], (const act_t*)d[0], (int)d[1], (const act_t*)d[2], (int)d[3], (float*)d[4],
                         (const float*)d[5], (const float*)d[6], (int)d[7], (int)d[8], (int)d[9],
                         (int)d[10], (int)d[11], (int)d[12], (int)d[13], (int)d[14], (int)d[15],
                         (int)d[16], (int)d[17], (int)d[18], (int)d[19]);
    if (rc) return rc;
    const WgMode& m = u[i].m;
    key[i] = u[i].p.vn * 1000 + u[i].p.vk * 100 + (m.flat ? 16 : 0) + (m.pro ? 8 : 0) + (m.aln ? 4 : 0) +
             (m.alk ? 2 : 0) + (m.gather ? 1 : 0);
  }
  for (int i = 0; i < count; ++i) {
    if (done[i]) continue;
    WgGroup g;
    g.n = 0;
    g.start[0] = 0;
    for (int j = i; j < count && g.n < kWgGroup; ++j) {
      if (done[j] || key[j] != key[i]) continue;
      const dim3& gr = u[j].grid;
      const int64_t blocks = (int64_t)gr.x * gr.y * gr.z;
      if (g.n > 0 && g.start[g.n] + blocks > (1 << 20)) continue;
      g.a[g.n] = u[j].a;
      g.nslab[g.n] = (int)gr.x;
      g.gy[g.n] = (int)gr.y;
      g.start[g.n + 1] = g.start[g.n] + (int)blocks;
      ++g.n;
      done[j] = 1;
    }
    const WgPlan& p = u[i].p;
    int rc;
#define WG_CASE(VN_, VK_) \
  if (p.vn == VN_ && p.vk == VK_) rc = launch_wgrad_group<VN_, VK_>(g, u[i].m, s); else
    WG_CASE(4, 4) WG_CASE(4, 2) WG_CASE(4, 1) WG_CASE(2, 4) WG_CASE(2, 2) WG_CASE(2, 1)
    WG_CASE(1, 4) WG_CASE(1, 2) WG_CASE(1, 1)
    rc = nasseg_fail(NASSEG_ERR_UNSUPPORTED, "conv_wgrad_many: no kernel for vn=%d vk=%d", p.vn, p.vk);
#undef WG_CASE
    if (rc) return rc;
  }
  return NASSEG_OK;
}

#if NASSEG_FP32_ONLY
// count reductions finalised by one launch per 16: partial[i] = the workspace a
// nasseg_conv_wgrad / nasseg_dwconv_wgrad call (either dtype) left behind when its dw argument
// was null, dw[i] = that layer's gradient tensor, dims[5*i..] = number of partial rows
// (workspace floats / (taps*N*K)), taps, N, K, flat (1 when nasseg_conv_fwd_pack_mode(K,kh,kw)
// == 2, else 0); a depthwise weight (C,1,k,k) is passed as taps = k*k, N = C, K = 1.
int nasseg_wgrad_finalize_many(int count, const float* const* partial, float* const* dw,
                               const int* dims, void* stream) {
  NASSEG_REQUIRE(count >= 0 && (count == 0 || (partial && dw && dims)), "wgrad_finalize_many: bad arguments");
  for (int base = 0; base < count; base += kFinMax) {
    const int n = count - base < kFinMax ? count - base : kFinMax;
    FinTable t;
    int64_t most = 0;
    for (int i = 0; i < n; ++i) {
      const int* d = dims + 5 * (base + i);
      NASSEG_REQUIRE(partial[base + i] && dw[base + i] && d[0] > 0 && d[1] > 0 && d[2] > 0 && d[3] > 0,
                     "wgrad_finalize_many: bad descriptor %d", base + i);
      t.d[i].partial = partial[base + i];
      t.d[i].dw = dw[base + i];
      t.d[i].nslab = d[0];
      t.d[i].ntaps = d[1];
      t.d[i].N = d[2];
      t.d[i].K = d[3];
      t.d[i].flat = d[4];
      const int64_t per = (int64_t)d[1] * d[2] * d[3];
      const int64_t blocks = cdiv64(per, fin_elems_per_block(d[0], per));
      if (blocks > most) most = blocks;
    }
    hipLaunchKernelGGL(wgrad_finalize_many, dim3((unsigned)most, n), dim3(256), 0, (hipStream_t)stream, t);
    NASSEG_LAUNCH_CHECK("wgrad_finalize_many");
  }
  return NASSEG_OK;
}
#endif  // NASSEG_FP32_ONLY

}  // extern "C"
