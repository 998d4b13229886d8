// Deterministic per-channel reductions over NHWC activations, gfx950.
//
// One kernel family serves: BatchNorm batch statistics and their backward sums
// (torch.nn.BatchNorm2d as used at src/nn/layer_factory.py:56-75,94-158),
// global average pooling and its broadcast backward (GAPConv1x1,
// layer_factory.py:181-195), bias gradients of the classifier heads and the
// ParamSum coefficient gradients (layer_factory.py:353-366).
//
// Layout: input [S segments][R rows][C] with row stride ld.  Lanes run over the
// flattened (row, channel-vector) axis so every wave load is contiguous; each
// thread keeps a fixed channel vector, accumulates two fp32 sums over its rows,
// the block reduces through LDS and writes partial[seg][blk][2][C].  A second
// pass sums the per-block partials in fp64 in a fixed order: no atomics, so
// results are run-to-run reproducible.
#include "common.h"

namespace {

enum { RED_SUM = 0, RED_SUMSQ = 1, RED_BN_BWD = 2, RED_DOT2 = 3, RED_DOT1 = 4 };

struct RedArgs {
  const act_t* a;  // (activations: fp32 or bf16 storage)
  int64_t lda;
  const act_t* b;
  int64_t ldb;
  const act_t* c;
  int64_t ldc;
  const float* scale;
  const float* shift;
  const float* mean;
  const float* invstd;
  int act;
  int S;
  int64_t R;
  int C;
  int nblk;
  int64_t rows_per_blk;
  float* partial;
};

template <int VEC>
struct Vt;
template <>
struct Vt<4> {
  typedef float4 T;
  static __device__ __forceinline__ T zero() { return f4zero(); }
  static __device__ __forceinline__ T ld(const float* p) { return lda4(p); }
  static __device__ __forceinline__ T ld(const bf16_t* p) { return lda4(p); }
  static __device__ __forceinline__ void st(float* p, T v) { sta4(p, v); }
  static __device__ __forceinline__ T add(T a, T b) { return add4(a, b); }
  static __device__ __forceinline__ T mul(T a, T b) { return mul4(a, b); }
  static __device__ __forceinline__ T sub(T a, T b) {
    return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
  }
  static __device__ __forceinline__ T fma(T a, T b, T c) { return fma4(a, b, c); }
  static __device__ __forceinline__ T mask(T z, int act) {
    return make_float4(act_mask(z.x, act), act_mask(z.y, act), act_mask(z.z, act),
                       act_mask(z.w, act));
  }
};
template <>
struct Vt<1> {
  typedef float T;
  static __device__ __forceinline__ T zero() { return 0.f; }
  static __device__ __forceinline__ T ld(const float* p) { return *p; }
  static __device__ __forceinline__ T ld(const bf16_t* p) { return lda1(p); }
  static __device__ __forceinline__ void st(float* p, T v) { *p = v; }
  static __device__ __forceinline__ T add(T a, T b) { return a + b; }
  static __device__ __forceinline__ T mul(T a, T b) { return a * b; }
  static __device__ __forceinline__ T sub(T a, T b) { return a - b; }
  static __device__ __forceinline__ T fma(T a, T b, T c) { return fmaf(a, b, c); }
  static __device__ __forceinline__ T mask(T z, int act) { return act_mask(z, act); }
};

template <int MODE, int VEC>
__global__ __launch_bounds__(256) void colred_kernel(RedArgs q) {
  typedef Vt<VEC> V;
  typedef typename V::T T;
  __shared__ T red0[256];
  __shared__ T red1[256];
  const int tid = threadIdx.x;
  const int CV = q.C / VEC;
  const int rpi = 256 / CV;  // rows per iteration (CV <= 256 checked on host)
  const bool live = tid < rpi * CV;
  const int cv = tid % CV;
  const int rr = tid / CV;
  const int seg = blockIdx.y;
  const int blk = blockIdx.x;
  const int64_t r0 = (int64_t)blk * q.rows_per_blk;
  int64_t r1 = r0 + q.rows_per_blk;
  if (r1 > q.R) r1 = q.R;

  if constexpr (MODE == RED_SUMSQ) {
    // BatchNorm batch statistics: sum and sum of squares per thread and through the block in
    // DOUBLE (the kernel is bandwidth-bound: the fp64 adds are free), rounded to fp32 once per
    // block row.  With fp32 running sums the variance E[x^2] - mean^2 of a channel whose mean
    // is 50 standard deviations lost 1.6e-4 of its value (tests/test_hip_anchor.py); now the only
    // fp32 rounding is that of the <= 768 block rows, which the fp64 second pass averages out.
    __shared__ double dred[2][256][VEC];
    double s0[VEC], s1[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) s0[v] = s1[v] = 0.0;
    if (live) {
      const act_t* pa = q.a + (int64_t)seg * q.R * q.lda + cv * VEC;
#pragma unroll 8
      for (int64_t r = r0 + rr; r < r1; r += rpi) {
        const T va = V::ld(pa + r * q.lda);
        const float* f = reinterpret_cast<const float*>(&va);
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          const double x = (double)f[v];
          s0[v] += x;
          s1[v] = fma(x, x, s1[v]);
        }
      }
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      dred[0][tid][v] = s0[v];
      dred[1][tid][v] = s1[v];
    }
    __syncthreads();
    if (tid < CV) {
      double t0[VEC], t1[VEC];
#pragma unroll
      for (int v = 0; v < VEC; ++v) t0[v] = t1[v] = 0.0;
      for (int u = tid; u < rpi * CV; u += CV)
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          t0[v] += dred[0][u][v];
          t1[v] += dred[1][u][v];
        }
      float* po = q.partial + (((int64_t)seg * q.nblk + blk) * 2) * q.C + tid * VEC;
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        po[v] = (float)t0[v];
        po[q.C + v] = (float)t1[v];
      }
    }
    return;
  }
  T acc0 = V::zero(), acc1 = V::zero();
  T sc = V::zero(), sh = V::zero(), mu = V::zero(), is = V::zero();
  if (MODE == RED_BN_BWD && live) {
    sc = V::ld(q.scale + cv * VEC);
    sh = V::ld(q.shift + cv * VEC);
    mu = V::ld(q.mean + cv * VEC);
    is = V::ld(q.invstd + cv * VEC);
  }
  if (live) {
    const act_t* pa = q.a + (int64_t)seg * q.R * q.lda + cv * VEC;
    const act_t* pb = q.b ? q.b + (int64_t)seg * q.R * q.ldb + cv * VEC : nullptr;
    const act_t* pc = q.c ? q.c + (int64_t)seg * q.R * q.ldc + cv * VEC : nullptr;
#pragma unroll 8
    for (int64_t r = r0 + rr; r < r1; r += rpi) {
      T va = V::ld(pa + r * q.lda);
      if (MODE == RED_SUM) {
        acc0 = V::add(acc0, va);
      } else if (MODE == RED_SUMSQ) {
        acc0 = V::add(acc0, va);
        acc1 = V::fma(va, va, acc1);
      } else if (MODE == RED_BN_BWD) {
        T vx = V::ld(pb + r * q.ldb);
        T g = V::mul(va, V::mask(V::fma(vx, sc, sh), q.act));
        T xh = V::mul(V::sub(vx, mu), is);
        acc0 = V::add(acc0, g);
        acc1 = V::fma(g, xh, acc1);
      } else if (MODE == RED_DOT2) {
        T vx = V::ld(pb + r * q.ldb);
        T vy = V::ld(pc + r * q.ldc);
        acc0 = V::fma(va, vx, acc0);
        acc1 = V::fma(va, vy, acc1);
      } else {  // RED_DOT1
        T vx = V::ld(pb + r * q.ldb);
        acc0 = V::fma(va, vx, acc0);
      }
    }
  }
  red0[tid] = acc0;
  red1[tid] = acc1;
  __syncthreads();
  if (tid < CV) {
    T s0 = V::zero(), s1 = V::zero();
    for (int u = tid; u < rpi * CV; u += CV) {
      s0 = V::add(s0, red0[u]);
      s1 = V::add(s1, red1[u]);
    }
    float* po = q.partial + (((int64_t)seg * q.nblk + blk) * 2) * q.C + tid * VEC;
    V::st(po, s0);
    V::st(po + q.C, s1);
  }
}

// out[seg][acc][c] = mul * sum_blk partial[seg][blk][acc][c]
// grid: (ceil(2*C/16), S); 256 threads (see reduce_partials16)
__global__ __launch_bounds__(256) void colred_finalize(const float* __restrict__ partial,
                                                       float* __restrict__ out, int nblk, int nacc,
                                                       int C, float mul) {
  __shared__ double red[NASSEG_RP_SLICES][NASSEG_RP_ELEMS + 1];
  const int seg = blockIdx.y;
  const int64_t per = 2 * (int64_t)C;  // [2][C] per block, only the first nacc rows are used
  const int64_t e = (int64_t)blockIdx.x * NASSEG_RP_ELEMS + rp_elem();
  const bool valid = e < (int64_t)nacc * C;
  const double s = reduce_partials16(partial + (int64_t)seg * nblk * per, nblk, per, e, valid, red);
  if (valid && rp_slice() == 0) out[(int64_t)seg * nacc * C + e] = (float)(s * (double)mul);
}

// BatchNorm statistics finalisation (training mode).  Normalisation uses the
// biased variance, running_var the unbiased one (torch semantics).
// grid: ceil(C/8) workgroups of 8 * SLICES threads (SLICES = 128: up to 4096 partial rows in one launch).
template <int SLICES>
__global__ __launch_bounds__(8 * SLICES) void bn_stats_finalize_t(
    const float* __restrict__ partial, int nblk, int C, double M, float eps, float momentum,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ mean,
    float* __restrict__ invstd, float* __restrict__ scale, float* __restrict__ shift,
    float* __restrict__ running_mean, float* __restrict__ running_var, int64_t* nbt) {
  __shared__ double red[SLICES][NASSEG_RP_ELEMS + 1];
  const int c = blockIdx.x * NASSEG_RP_ELEMS + rp_elem();
  const bool valid = c < C;
  const int64_t per = 2 * (int64_t)C;
  // (Round 5 tried the two columns in ONE pass over the rows, every load of a round in flight - 16 rows per thread,
  //  masks instead of the scalar tail loop - and gamma / beta / running statistics requested ahead of the rows: the
  //  kernel got SLOWER, 7.9 -> 9.7 us per launch on the headline's 1024 - 1536 rows and 0.3 - 0.5 % on every workload
  //  of a same-box A/B (tools/gpu.sh flags: headline 242.0 -> 241.3, CVPR 321x321 replayed 1261 -> 1255, task0 5562 ->
  //  5538): with 1024 threads per workgroup the masked loads of rows that do not exist cost more issue slots than the
  //  round trips they save.  The two-pass form stays.)
  const double s0 = reduce_partials_n<SLICES>(partial, nblk, per, c, valid, red);
  const double s1 = reduce_partials_n<SLICES>(partial, nblk, per, (int64_t)C + c, valid, red);
  if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) *nbt += 1;
  if (!valid || rp_slice() != 0) return;
  const double mu = s0 / M;
  double var = s1 / M - mu * mu;
  if (var < 0.0) var = 0.0;
  const double is = 1.0 / sqrt(var + (double)eps);
  mean[c] = (float)mu;
  invstd[c] = (float)is;
  const double g = gamma ? (double)gamma[c] : 1.0;
  const double bt = beta ? (double)beta[c] : 0.0;
  scale[c] = (float)(g * is);
  shift[c] = (float)(bt - mu * g * is);
  if (running_mean) {
    running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * mu);
    const double unb = M > 1.0 ? var * M / (M - 1.0) : var;
    running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unb);
  }
}

// eval-mode BatchNorm folded into per-channel scale/shift (+ mean/invstd for bwd)
__global__ void bn_eval_params(int C, float eps, const float* __restrict__ gamma,
                               const float* __restrict__ beta,
                               const float* __restrict__ running_mean,
                               const float* __restrict__ running_var, float* __restrict__ mean,
                               float* __restrict__ invstd, float* __restrict__ scale,
                               float* __restrict__ shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float is = 1.0f / sqrtf(running_var[c] + eps);
  const float g = gamma ? gamma[c] : 1.f;
  const float bt = beta ? beta[c] : 0.f;
  mean[c] = running_mean[c];
  invstd[c] = is;
  scale[c] = g * is;
  shift[c] = bt - running_mean[c] * g * is;
}

// sum groups of `rows_per_group` consecutive rows of partial[nblk][per] into out[g][per]
// (fp32 out, fp64 accumulation, fixed order): first level of a two-level finalisation
template <int SLICES>
__global__ __launch_bounds__(8 * SLICES) void rows_group_sum_t(const float* __restrict__ partial,
                                                               float* __restrict__ out, int nblk, int64_t per,
                                                               int rows_per_group) {
  __shared__ double red[SLICES][NASSEG_RP_ELEMS + 1];
  const int g = blockIdx.y;
  const int r0 = g * rows_per_group;
  int nr = nblk - r0;
  if (nr > rows_per_group) nr = rows_per_group;
  const int64_t e = (int64_t)blockIdx.x * NASSEG_RP_ELEMS + rp_elem();
  const bool valid = e < per;
  const double s = reduce_partials_n<SLICES>(partial + (int64_t)r0 * per, nr, per, e, valid, red);
  if (valid && rp_slice() == 0) out[(int64_t)g * per + e] = (float)s;
}

// partial rows from which the 1024-thread second stage is used (one round of loads per thread instead of
// several): 512 until round 4; 32 measured +1 % on the replayed CVPR 321x321 step, neutral on the headline
#ifndef NASSEG_RP_WIDE_FROM
#define NASSEG_RP_WIDE_FROM 32
#endif

struct RedPlan {
  int nblk;
  int64_t rows_per_blk;
};
inline RedPlan red_plan(int S, int64_t R, int C, int vec) {
  const int CV = C / vec;
  const int rpi = 256 / CV;
  // at least 4 rounds of loads per workgroup (16 until round 4: the 16 x 11 x 11 maps of the CVPR cells ran on 7
  // workgroups - 19 us for 1 MB; one workgroup streams ~20 GB/s whatever the size of the part)
  int64_t nb = R / ((int64_t)rpi * 4);
  int64_t cap = 768 / S;
  if (cap < 1) cap = 1;
  if (nb > cap) nb = cap;
  if (nb < 1) nb = 1;
  RedPlan p;
  p.rows_per_blk = cdiv64(R, nb);
  p.nblk = (int)cdiv64(R, p.rows_per_blk);
  return p;
}

template <int MODE>
int launch_colred(RedArgs& q, hipStream_t s) {
  const int vec = (q.C % 4 == 0 && q.lda % 4 == 0 && (!q.b || q.ldb % 4 == 0) &&
                   (!q.c || q.ldc % 4 == 0))
                      ? 4
                      : 1;
  NASSEG_REQUIRE(q.C / vec <= 256, "colred: C=%d too large", q.C);
  RedPlan p = red_plan(q.S, q.R, q.C, vec);
  q.nblk = p.nblk;
  q.rows_per_blk = p.rows_per_blk;
  dim3 grid(p.nblk, q.S, 1);
  if (vec == 4)
    hipLaunchKernelGGL((colred_kernel<MODE, 4>), grid, dim3(256), 0, s, q);
  else
    hipLaunchKernelGGL((colred_kernel<MODE, 1>), grid, dim3(256), 0, s, q);
  NASSEG_LAUNCH_CHECK("colred_kernel");
  return NASSEG_OK;
}

}  // namespace

extern "C" {

#if NASSEG_FP32_ONLY
// floats of workspace for any nasseg_colred_* / nasseg_bn_* call on [S][R][C]
int64_t nasseg_colred_workspace(int S, int64_t R, int C) {
  if (C <= 0 || S <= 0 || R <= 0) return 0;
  // the scalar plan never uses more blocks than the vector plan's cap
  int64_t cap = 768 / S;
  if (cap < 1) cap = 1;
  return (int64_t)S * cap * 2 * C + 16;
}

#endif  // NASSEG_FP32_ONLY (queries and fp32-vector entry points exist once)

// mode: 0 sum(a), 1 {sum(a), sum(a^2)}, 3 {sum(a*b), sum(a*c)}, 4 sum(a*b)
// out[seg][nacc][C] = mul * sums   (nacc = 2 for modes 1 and 3, else 1)
int NASSEG_FN(colred)(int mode, const act_t* a, int64_t lda, const act_t* b, int64_t ldb,
                  const act_t* c, int64_t ldc, float* out, float* ws, int S, int64_t R, int C,
                  float mul, void* stream) {
  NASSEG_REQUIRE(S > 0 && R > 0 && C > 0, "colred: bad shape");
  hipStream_t s = (hipStream_t)stream;
  RedArgs q = {};
  q.a = a; q.lda = lda; q.b = b; q.ldb = ldb; q.c = c; q.ldc = ldc;
  q.S = S; q.R = R; q.C = C; q.partial = ws;
  int rc, nacc = 1;
  switch (mode) {
    case RED_SUM: rc = launch_colred<RED_SUM>(q, s); break;
    case RED_SUMSQ: rc = launch_colred<RED_SUMSQ>(q, s); nacc = 2; break;
    case RED_DOT2: rc = launch_colred<RED_DOT2>(q, s); nacc = 2; break;
    case RED_DOT1: rc = launch_colred<RED_DOT1>(q, s); break;
    default: return nasseg_fail(NASSEG_ERR_ARG, "colred: bad mode %d", mode);
  }
  if (rc) return rc;
  hipLaunchKernelGGL(colred_finalize, dim3(cdiv(nacc * C, NASSEG_RP_ELEMS), S), dim3(256), 0, s, ws,
                     out, q.nblk, nacc, C, mul);
  NASSEG_LAUNCH_CHECK("colred_finalize");
  return NASSEG_OK;
}

// training-mode BatchNorm statistics of x [M][C] (row stride ldx).
// Writes mean, invstd, scale = gamma*invstd, shift = beta - mean*scale and
// updates running stats / num_batches_tracked in place (any may be null).
int NASSEG_FN(bn_stats)(const act_t* x, int64_t ldx, int64_t M, int C, float eps, float momentum,
                    const float* gamma, const float* beta, float* mean, float* invstd,
                    float* scale, float* shift, float* running_mean, float* running_var,
                    int64_t* num_batches_tracked, float* ws, void* stream) {
  NASSEG_REQUIRE(M > 0 && C > 0, "bn_stats: bad shape");
  hipStream_t s = (hipStream_t)stream;
  RedArgs q = {};
  q.a = x; q.lda = ldx; q.S = 1; q.R = M; q.C = C; q.partial = ws;
  int rc = launch_colred<RED_SUMSQ>(q, s);
  if (rc) return rc;
  hipLaunchKernelGGL(bn_stats_finalize_t<NASSEG_RP_SLICES>, dim3(cdiv(C, NASSEG_RP_ELEMS)), dim3(256), 0, s, ws, q.nblk, C,
                     (double)M, eps, momentum, gamma, beta, mean, invstd, scale, shift,
                     running_mean, running_var, num_batches_tracked);
  NASSEG_LAUNCH_CHECK("bn_stats_finalize");
  return NASSEG_OK;
}

#if NASSEG_FP32_ONLY
// BatchNorm statistics from per-workgroup partials [nblk][2][C] (sum, sum of squares) that a
// producer kernel (nasseg_conv_fwd with `stats`) already wrote: same outputs as nasseg_bn_stats.
// The buffer must have room for 64 more rows ([nblk + 64][2][C]) - scratch of the first level
// of the two-level reduction used when nblk > 512; `partial` is therefore not const in effect.
int nasseg_bn_finalize(float* partial, int nblk, int64_t M, int C, float eps, float momentum,
                       const float* gamma, const float* beta, float* mean, float* invstd,
                       float* scale, float* shift, float* running_mean, float* running_var,
                       int64_t* num_batches_tracked, void* stream) {
  NASSEG_REQUIRE(M > 0 && C > 0 && nblk > 0, "bn_finalize: bad shape");
  hipStream_t s = (hipStream_t)stream;
  const float* src = partial;
  if (nblk > NASSEG_RP_WIDE_FROM && nblk <= NASSEG_RP_WIDE_MAX_ROWS) {
    // one launch of 1024-thread workgroups (1024 rows per round of loads) instead of two levels
    hipLaunchKernelGGL(bn_stats_finalize_t<NASSEG_RP_WIDE_SLICES>, dim3(cdiv(C, NASSEG_RP_ELEMS)),
                       dim3(8 * NASSEG_RP_WIDE_SLICES), 0, s, src, nblk, C, (double)M, eps, momentum, gamma, beta,
                       mean, invstd, scale, shift, running_mean, running_var, num_batches_tracked);
    NASSEG_LAUNCH_CHECK("bn_stats_finalize");
    return NASSEG_OK;
  }
  if (nblk > 512) {
    // two levels: 64 groups of rows first (rows [nblk, nblk+64) of the buffer are scratch)
    const int G = 64;
    const int rpg = cdiv(nblk, G);
    const int groups = cdiv(nblk, rpg);
    float* lvl = partial + (int64_t)nblk * 2 * C;
    hipLaunchKernelGGL(rows_group_sum_t<NASSEG_RP_SLICES>, dim3(cdiv(2 * C, NASSEG_RP_ELEMS), groups), dim3(256), 0,
                       s, partial, lvl, nblk, (int64_t)2 * C, rpg);
    NASSEG_LAUNCH_CHECK("rows_group_sum");
    src = lvl;
    nblk = groups;
  }
  hipLaunchKernelGGL(bn_stats_finalize_t<NASSEG_RP_SLICES>, dim3(cdiv(C, NASSEG_RP_ELEMS)), dim3(256), 0, s, src,
                     nblk, C, (double)M, eps, momentum, gamma, beta, mean, invstd, scale, shift, running_mean,
                     running_var, num_batches_tracked);
  NASSEG_LAUNCH_CHECK("bn_stats_finalize");
  return NASSEG_OK;
}

// out[e] = sum over rows of partial[nblk][cols] (fp64 accumulation, fixed order): turns the
// per-workgroup rows written by nasseg_conv_bwd_data_bn / nasseg_dwconv_bwd_data_bn into the
// sums[2][C] that nasseg_bn_bwd_reduce produces.  Like nasseg_bn_finalize the buffer needs room
// for 64 extra rows (first level of the two-level reduction when nblk > 512).
int nasseg_rows_sum(float* partial, int nblk, int cols, float* out, void* stream) {
  NASSEG_REQUIRE(nblk > 0 && cols > 0 && partial && out, "rows_sum: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const float* src = partial;
  if (nblk > NASSEG_RP_WIDE_FROM && nblk <= NASSEG_RP_WIDE_MAX_ROWS) {
    hipLaunchKernelGGL(rows_group_sum_t<NASSEG_RP_WIDE_SLICES>, dim3(cdiv(cols, NASSEG_RP_ELEMS), 1),
                       dim3(8 * NASSEG_RP_WIDE_SLICES), 0, s, src, out, nblk, (int64_t)cols, nblk);
    NASSEG_LAUNCH_CHECK("rows_group_sum");
    return NASSEG_OK;
  }
  if (nblk > 512) {
    const int G = 64;
    const int rpg = cdiv(nblk, G);
    const int groups = cdiv(nblk, rpg);
    float* lvl = partial + (int64_t)nblk * cols;
    hipLaunchKernelGGL(rows_group_sum_t<NASSEG_RP_SLICES>, dim3(cdiv(cols, NASSEG_RP_ELEMS), groups), dim3(256), 0, s,
                       partial, lvl, nblk, (int64_t)cols, rpg);
    NASSEG_LAUNCH_CHECK("rows_group_sum");
    src = lvl;
    nblk = groups;
  }
  hipLaunchKernelGGL(rows_group_sum_t<NASSEG_RP_SLICES>, dim3(cdiv(cols, NASSEG_RP_ELEMS), 1), dim3(256), 0, s, src,
                     out, nblk, (int64_t)cols, nblk);
  NASSEG_LAUNCH_CHECK("rows_group_sum");
  return NASSEG_OK;
}

int nasseg_bn_eval_params(int C, float eps, const float* gamma, const float* beta,
                          const float* running_mean, const float* running_var, float* mean,
                          float* invstd, float* scale, float* shift, void* stream) {
  NASSEG_REQUIRE(C > 0, "bn_eval_params: bad C");
  hipLaunchKernelGGL(bn_eval_params, dim3(cdiv(C, 128)), dim3(128), 0, (hipStream_t)stream, C, eps,
                     gamma, beta, running_mean, running_var, mean, invstd, scale, shift);
  NASSEG_LAUNCH_CHECK("bn_eval_params");
  return NASSEG_OK;
}

#endif  // NASSEG_FP32_ONLY

// BatchNorm backward sums: with g = dy * act'(x*scale+shift) and
// xhat = (x-mean)*invstd:  sums[0][c] = sum g (= dbeta), sums[1][c] = sum g*xhat (= dgamma)
int NASSEG_FN(bn_bwd_reduce)(const act_t* dy, int64_t lddy, const act_t* x, int64_t ldx, int64_t M,
                         int C, const float* scale, const float* shift, const float* mean,
                         const float* invstd, int act, float* sums, float* ws, void* stream) {
  NASSEG_REQUIRE(M > 0 && C > 0, "bn_bwd_reduce: bad shape");
  hipStream_t s = (hipStream_t)stream;
  RedArgs q = {};
  q.a = dy; q.lda = lddy; q.b = x; q.ldb = ldx;
  q.scale = scale; q.shift = shift; q.mean = mean; q.invstd = invstd; q.act = act;
  q.S = 1; q.R = M; q.C = C; q.partial = ws;
  int rc = launch_colred<RED_BN_BWD>(q, s);
  if (rc) return rc;
  hipLaunchKernelGGL(colred_finalize, dim3(cdiv(2 * C, NASSEG_RP_ELEMS), 1), dim3(256), 0, s, ws,
                     sums, q.nblk, 2, C, 1.0f);
  NASSEG_LAUNCH_CHECK("colred_finalize");
  return NASSEG_OK;
}

// First stage of nasseg_bn_bwd_reduce ONLY: rows [nasseg_colred_rows(1, M, C)][2][C] of per-workgroup {sum g, sum g*xhat}
// in `rows` (a buffer of nasseg_colred_workspace(1, M, C) floats) - for a consumer that adds them up itself
// (nasseg_bn_bwd_apply_rows).  C and both strides multiples of 4.
int NASSEG_FN(bn_bwd_reduce_rows)(const act_t* dy, int64_t lddy, const act_t* x, int64_t ldx, int64_t M, int C,
                                  const float* scale, const float* shift, const float* mean, const float* invstd,
                                  int act, float* rows, void* stream) {
  NASSEG_REQUIRE(M > 0 && C > 0 && C % 4 == 0 && lddy % 4 == 0 && ldx % 4 == 0 && rows,
                 "bn_bwd_reduce_rows: channels and strides must be multiples of 4");
  RedArgs q = {};
  q.a = dy; q.lda = lddy; q.b = x; q.ldb = ldx;
  q.scale = scale; q.shift = shift; q.mean = mean; q.invstd = invstd; q.act = act;
  q.S = 1; q.R = M; q.C = C; q.partial = rows;
  return launch_colred<RED_BN_BWD>(q, (hipStream_t)stream);
}

#if NASSEG_FP32_ONLY
// rows the first stage of a per-channel reduction over [S][R][C] leaves per segment (C % 4 == 0, aligned strides)
int64_t nasseg_colred_rows(int S, int64_t R, int C) {
  if (S <= 0 || R <= 0 || C <= 0 || (C & 3) || C / 4 > 256) return 0;
  return red_plan(S, R, C, 4).nblk;
}
#endif

}  // extern "C"
