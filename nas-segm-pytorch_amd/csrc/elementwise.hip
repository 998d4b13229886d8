// Elementwise / copy kernels over fp32 NHWC activations, gfx950.
// All are pure HBM streams: float4 per lane, grid-stride, <= 2048 workgroups.
//
// Reference call sites: BatchNorm apply + ReLU/ReLU6 (src/nn/layer_factory.py:
// 94-158), residual add (:155-158), cell sums (src/nn/micro_decoders.py:48-51,
// 110-121), ParamSum (layer_factory.py:353-366), Skip / Zero channel repeat
// (:268-297), torch.cat + F.relu in collect_all / ConcatReduce
// (micro_decoders.py:11-25,251; layer_factory.py:369-382).
#include "common.h"

namespace {

inline int ew_grid(int64_t n) {
  int64_t b = (n + 255) / 256;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (int)b;
}

// y = act(x*scale[c] + shift[c]) (+ res)
__global__ __launch_bounds__(256) void affine_act_kernel(
    const act_t* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
    const act_t* __restrict__ res, act_t* __restrict__ y, int64_t n4, int C4, int act) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    float4 v = lda4(x + i * 4);
    float4 s = scale ? lda4(scale + c4 * 4) : make_float4(1.f, 1.f, 1.f, 1.f);
    float4 h = shift ? lda4(shift + c4 * 4) : f4zero();
    v = act_apply4(fma4(v, s, h), act);
    if (res) v = add4(v, lda4(res + i * 4));
    sta4(y + i * 4, v);
  }
}

// y = ca[c] * act_a(xa*sa[c] + ha[c]) + cb[c] * act_b(xb*sb[c] + hb[c]): the sum of two op outputs whose last
// BatchNorm (+ activation) is still pending (functional.Pending), with ParamSum's per-channel coefficients; any of
// the per-channel vectors may be null (scale / coefficient 1, shift 0)
__global__ __launch_bounds__(256) void add_act2_kernel(
    const act_t* __restrict__ xa, const float* __restrict__ sa, const float* __restrict__ ha, int act_a,
    const float* __restrict__ ca, const act_t* __restrict__ xb, const float* __restrict__ sb,
    const float* __restrict__ hb, int act_b, const float* __restrict__ cb, act_t* __restrict__ y, int64_t n4,
    int C4) {
  const ActSel fa = act_sel(act_a), fb = act_sel(act_b);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    float4 u = lda4(xa + i * 4), v = lda4(xb + i * 4);
    if (sa || ha) u = fma4(u, sa ? lda4(sa + c4 * 4) : make_float4(1.f, 1.f, 1.f, 1.f), ha ? lda4(ha + c4 * 4) : f4zero());
    if (sb || hb) v = fma4(v, sb ? lda4(sb + c4 * 4) : make_float4(1.f, 1.f, 1.f, 1.f), hb ? lda4(hb + c4 * 4) : f4zero());
    u = act_apply4(u, fa);
    v = act_apply4(v, fb);
    if (ca) u = mul4(u, lda4(ca + c4 * 4));
    if (cb) v = mul4(v, lda4(cb + c4 * 4));
    sta4(y + i * 4, add4(u, v));
  }
}

// BatchNorm (+activation) backward apply:
//   g = dy * act'(x*scale+shift);  xhat = (x-mean)*invstd
//   train: dx = scale * (g - sums0/M - xhat*sums1/M)     eval: dx = scale * g
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(
    const act_t* __restrict__ dy, const act_t* __restrict__ x, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ mean,
    const float* __restrict__ invstd, const float* __restrict__ sums, float invM, int train,
    int act, act_t* __restrict__ dx, int64_t n4, int C4) {
  const int C = C4 * 4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    const float4 d = lda4(dy + i * 4);
    const float4 v = lda4(x + i * 4);
    const float4 s = lda4(scale + c4 * 4);
    const float4 h = lda4(shift + c4 * 4);
    const float4 z = fma4(v, s, h);
    float4 g = make_float4(d.x * act_mask(z.x, act), d.y * act_mask(z.y, act),
                           d.z * act_mask(z.z, act), d.w * act_mask(z.w, act));
    if (train) {
      const float4 mu = lda4(mean + c4 * 4);
      const float4 is = lda4(invstd + c4 * 4);
      const float4 s0 = lda4(sums + c4 * 4);
      const float4 s1 = lda4(sums + C + c4 * 4);
      g.x = g.x - s0.x * invM - (v.x - mu.x) * is.x * s1.x * invM;
      g.y = g.y - s0.y * invM - (v.y - mu.y) * is.y * s1.y * invM;
      g.z = g.z - s0.z * invM - (v.z - mu.z) * is.z * s1.z * invM;
      g.w = g.w - s0.w * invM - (v.w - mu.w) * is.w * s1.w * invM;
    }
    sta4(dx + i * 4, mul4(g, s));
  }
}

// bn_bwd_apply_kernel that ADDS UP the partial rows of the sums itself: rows [nrows][2][C] = per-workgroup {sum g,
// sum g*xhat} as a first-stage reduction (colred_kernel) or a fused backward-data epilogue left them.  Every workgroup
// sums the few rows it is given in fp64 in a fixed order (thread = (group of four columns, slice of the rows); the slices
// meet in LDS in slice order), workgroup 0 also writes the sums out (they are the BatchNorm's parameter gradients).  On the
// small maps of the CVPR cells (src/nn/micro_decoders.py:54-121: 16 x 11 x 11 ... 16 x 41 x 41, 8 - 128 rows) this
// replaces the row-summing launch in front of every BatchNorm backward: a dependent 5 us launch against one extra
// round trip to L2 at the head of a kernel that is launched anyway.
__global__ __launch_bounds__(256) void bn_bwd_apply_rows_kernel(
    const act_t* __restrict__ dy, const act_t* __restrict__ x, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ rows, int nrows, float* __restrict__ sums_out, float invM, int train, int act,
    act_t* __restrict__ dx, int64_t n4, int C4) {
  extern __shared__ double apply_rows_lds[];
  double* dsum = apply_rows_lds;                                     // [slices][4 * groups here] <= 1024 doubles
  float* fin = reinterpret_cast<float*>(apply_rows_lds + 1024);      // [2 C]
  const int C = C4 * 4, cols = 2 * C, ncol4 = cols >> 2;
  const int tid = threadIdx.x;
  for (int cg0 = 0; cg0 < ncol4; cg0 += 256) {
    const int nh = ncol4 - cg0 < 256 ? ncol4 - cg0 : 256;
    const int slices = 256 / nh;
    const int cg = tid % nh, sl = tid / nh;
    if (sl < slices) {
      double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
      const float* p = rows + (size_t)(cg0 + cg) * 4;
#pragma unroll 8
      for (int r = sl; r < nrows; r += slices) {
        const float4 v = ld4(p + (size_t)r * cols);
        a0 += (double)v.x;
        a1 += (double)v.y;
        a2 += (double)v.z;
        a3 += (double)v.w;
      }
      double* po = dsum + (size_t)(sl * nh + cg) * 4;
      po[0] = a0;
      po[1] = a1;
      po[2] = a2;
      po[3] = a3;
    }
    __syncthreads();
    for (int e = tid; e < nh * 4; e += 256) {
      double t = 0.0;
      for (int q = 0; q < slices; ++q) t += dsum[(size_t)(q * nh + (e >> 2)) * 4 + (e & 3)];
      fin[cg0 * 4 + e] = (float)t;
    }
    __syncthreads();
  }
  if (blockIdx.x == 0 && sums_out)
    for (int e = tid; e < cols; e += 256) sums_out[e] = fin[e];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    const float4 d = lda4(dy + i * 4);
    const float4 v = lda4(x + i * 4);
    const float4 s = lda4(scale + c4 * 4);
    const float4 h = lda4(shift + c4 * 4);
    const float4 z = fma4(v, s, h);
    float4 g = make_float4(d.x * act_mask(z.x, act), d.y * act_mask(z.y, act),
                           d.z * act_mask(z.z, act), d.w * act_mask(z.w, act));
    if (train) {
      const float4 mu = lda4(mean + c4 * 4);
      const float4 is = lda4(invstd + c4 * 4);
      const float4 s0 = *reinterpret_cast<const float4*>(fin + c4 * 4);
      const float4 s1 = *reinterpret_cast<const float4*>(fin + C + c4 * 4);
      g.x = g.x - s0.x * invM - (v.x - mu.x) * is.x * s1.x * invM;
      g.y = g.y - s0.y * invM - (v.y - mu.y) * is.y * s1.y * invM;
      g.z = g.z - s0.z * invM - (v.z - mu.z) * is.z * s1.z * invM;
      g.w = g.w - s0.w * invM - (v.w - mu.w) * is.w * s1.w * invM;
    }
    sta4(dx + i * 4, mul4(g, s));
  }
}

// y = act(alpha[c]*a + beta[c]*b); alpha / beta null = 1; b null = absent
__global__ __launch_bounds__(256) void axpby_kernel(const act_t* __restrict__ a,
                                                    const act_t* __restrict__ b,
                                                    const float* __restrict__ alpha,
                                                    const float* __restrict__ beta,
                                                    act_t* __restrict__ y, int64_t n4, int C4,
                                                    int act) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    float4 v = lda4(a + i * 4);
    if (alpha) v = mul4(v, lda4(alpha + c4 * 4));
    if (b) {
      float4 w = lda4(b + i * 4);
      if (beta) w = mul4(w, lda4(beta + c4 * 4));
      v = add4(v, w);
    }
    sta4(y + i * 4, act_apply4(v, act));
  }
}

// dx = dy * act'(y_or_z)  (ReLU: ref > 0; ReLU6: 0 < ref < 6)
__global__ __launch_bounds__(256) void act_bwd_kernel(const act_t* __restrict__ dy,
                                                      const act_t* __restrict__ ref,
                                                      act_t* __restrict__ dx, int64_t n4, int act) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 d = lda4(dy + i * 4);
    const float4 r = lda4(ref + i * 4);
    sta4(dx + i * 4, make_float4(d.x * act_mask(r.x, act), d.y * act_mask(r.y, act),
                                d.z * act_mask(r.z, act), d.w * act_mask(r.w, act)));
  }
}

__global__ __launch_bounds__(256) void fill_kernel(act_t* __restrict__ y, int64_t n, float v) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    sta1(y + i, v);
}

// strided channel-block copy: y[p][yoff + c] = act(x[p][xoff + c]) * (mask_ref ? act'(mask_ref[p][moff+c]) : 1)
// for c < C; covers torch.cat (write into a slab), its backward (slice out of a
// slab, optionally masked by the ReLU that followed the cat) and Skip's repeat.
__global__ __launch_bounds__(256) void chan_copy_kernel(
    const act_t* __restrict__ x, int64_t ldx, int xoff, act_t* __restrict__ y, int64_t ldy,
    int yoff, const act_t* __restrict__ mref, int64_t ldm, int moff, int64_t P, int C4, int act,
    int mact) {
  const int64_t n4 = P * C4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    const int64_t p = i / C4;
    float4 v = act_apply4(lda4(x + p * ldx + xoff + c4 * 4), act);
    if (mref) {
      const float4 r = lda4(mref + p * ldm + moff + c4 * 4);
      v = make_float4(v.x * act_mask(r.x, mact), v.y * act_mask(r.y, mact),
                      v.z * act_mask(r.z, mact), v.w * act_mask(r.w, mact));
    }
    sta4(y + p * ldy + yoff + c4 * 4, v);
  }
}

// dx[p][c] = sum_r dy[p][r*C + c]   (backward of the channel repeat in Skip)
__global__ __launch_bounds__(256) void chan_fold_kernel(const act_t* __restrict__ dy,
                                                        act_t* __restrict__ dx, int64_t P, int C4,
                                                        int rep) {
  const int64_t n4 = P * C4;
  const int C = C4 * 4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    const int64_t p = i / C4;
    float4 s = f4zero();
    for (int r = 0; r < rep; ++r) s = add4(s, lda4(dy + (p * rep + r) * C + c4 * 4));
    sta4(dx + i * 4, s);
  }
}

// dst[i] = src[idx[i]] for rows of `row_bytes` bytes (a multiple of V): the per-step batch of the
// task0 feature cache, gathered on the device (the reference indexes the cache with a shuffled
// index array, src/engine/trainer.py:128-137)
template <typename V>
__global__ __launch_bounds__(256) void gather_rows_kernel(const V* __restrict__ src, const int64_t* __restrict__ idx,
                                                          V* __restrict__ dst, int64_t per_row, int64_t n_src) {
  const int64_t row = blockIdx.y;
  int64_t s = idx[row];
  s = s < 0 ? 0 : (s >= n_src ? n_src - 1 : s);  // (never out of the cache, whatever the index says)
  const V* in = src + s * per_row;
  V* out = dst + row * per_row;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per_row; i += (int64_t)gridDim.x * 256) out[i] = in[i];
}

}  // namespace

extern "C" {

// y = act(x*scale[c] + shift[c]) (+ res); x, y, res dense [n/C][C]; scale/shift/res may be null
int NASSEG_FN(affine_act)(const act_t* x, const float* scale, const float* shift, const act_t* res,
                      act_t* y, int64_t n, int C, int act, void* stream) {
  NASSEG_REQUIRE(C > 0 && C % 4 == 0 && n % C == 0, "affine_act: bad shape n=%lld C=%d",
                 (long long)n, C);
  if (n == 0) return NASSEG_OK;
  hipLaunchKernelGGL(affine_act_kernel, dim3(ew_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, x,
                     scale, shift, res, y, n / 4, C / 4, act);
  NASSEG_LAUNCH_CHECK("affine_act");
  return NASSEG_OK;
}

// y = ca[c] * act_a(xa*sa[c] + ha[c]) + cb[c] * act_b(xb*sb[c] + hb[c]); xa, xb, y dense [n/C][C]; every
// per-channel vector may be null
int NASSEG_FN(add_act2)(const act_t* xa, const float* sa, const float* ha, int act_a, const float* ca,
                        const act_t* xb, const float* sb, const float* hb, int act_b, const float* cb, act_t* y,
                        int64_t n, int C, void* stream) {
  NASSEG_REQUIRE(C > 0 && C % 4 == 0 && n % C == 0 && xa && xb && y, "add_act2: bad shape n=%lld C=%d",
                 (long long)n, C);
  if (n == 0) return NASSEG_OK;
  hipLaunchKernelGGL(add_act2_kernel, dim3(ew_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, xa, sa, ha, act_a,
                     ca, xb, sb, hb, act_b, cb, y, n / 4, C / 4);
  NASSEG_LAUNCH_CHECK("add_act2");
  return NASSEG_OK;
}

int NASSEG_FN(bn_bwd_apply)(const act_t* dy, const act_t* x, const float* scale, const float* shift,
                        const float* mean, const float* invstd, const float* sums, int64_t M,
                        int C, int train, int act, act_t* dx, void* stream) {
  NASSEG_REQUIRE(C > 0 && C % 4 == 0 && M > 0, "bn_bwd_apply: bad shape");
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_grid(M * C / 4)), dim3(256), 0,
                     (hipStream_t)stream, dy, x, scale, shift, mean, invstd, sums,
                     (float)(1.0 / (double)M), train, act, dx, M * C / 4, C / 4);
  NASSEG_LAUNCH_CHECK("bn_bwd_apply");
  return NASSEG_OK;
}

// nasseg_bn_bwd_apply from the ROWS of the sums: rows [nrows][2][C] (first-stage partials of nasseg_bn_bwd_reduce_rows,
// or the statistics rows of a fused backward-data kernel) are added up by every workgroup of this launch (fp64, fixed
// order) - no nasseg_rows_sum / finalising launch in front of it; sums_out (null or [2][C]) receives the sums, i.e. the
// BatchNorm's {dbeta, dgamma}.  Meant for few rows: nrows * 2 * C * 4 bytes are read by every workgroup (the caller
// bounds them; nasseg_bn_bwd_apply_rows_max_bytes()).
int NASSEG_FN(bn_bwd_apply_rows)(const act_t* dy, const act_t* x, const float* scale, const float* shift,
                                 const float* mean, const float* invstd, const float* rows, int nrows, float* sums_out,
                                 int64_t M, int C, int train, int act, act_t* dx, void* stream) {
  NASSEG_REQUIRE(C > 0 && C % 4 == 0 && C <= 1024 && M > 0 && nrows > 0 && rows, "bn_bwd_apply_rows: bad arguments");
  const size_t lds = 1024 * sizeof(double) + (size_t)2 * C * sizeof(float);
  hipLaunchKernelGGL(bn_bwd_apply_rows_kernel, dim3(ew_grid(M * C / 4)), dim3(256), lds, (hipStream_t)stream, dy, x,
                     scale, shift, mean, invstd, rows, nrows, sums_out, (float)(1.0 / (double)M), train, act, dx,
                     M * C / 4, C / 4);
  NASSEG_LAUNCH_CHECK("bn_bwd_apply_rows");
  return NASSEG_OK;
}

#if NASSEG_FP32_ONLY
// bytes of rows (nrows * 2 * C * 4) up to which nasseg_bn_bwd_apply_rows is meant to replace a row-summing launch
int64_t nasseg_bn_bwd_apply_rows_max_bytes(void) { return 64 << 10; }
#endif

// y = act(alpha[c]*a + beta[c]*b)
int NASSEG_FN(axpby)(const act_t* a, const act_t* b, const float* alpha, const float* beta, act_t* y,
                 int64_t n, int C, int act, void* stream) {
  NASSEG_REQUIRE(C > 0 && C % 4 == 0 && n % C == 0, "axpby: bad shape");
  if (n == 0) return NASSEG_OK;
  hipLaunchKernelGGL(axpby_kernel, dim3(ew_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, a, b,
                     alpha, beta, y, n / 4, C / 4, act);
  NASSEG_LAUNCH_CHECK("axpby");
  return NASSEG_OK;
}

int NASSEG_FN(act_bwd)(const act_t* dy, const act_t* ref, act_t* dx, int64_t n, int act, void* stream) {
  NASSEG_REQUIRE(n % 4 == 0, "act_bwd: n must be a multiple of 4");
  if (n == 0) return NASSEG_OK;
  hipLaunchKernelGGL(act_bwd_kernel, dim3(ew_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, dy,
                     ref, dx, n / 4, act);
  NASSEG_LAUNCH_CHECK("act_bwd");
  return NASSEG_OK;
}

int NASSEG_FN(fill)(act_t* y, int64_t n, float v, void* stream) {
  if (n <= 0) return NASSEG_OK;
  hipLaunchKernelGGL(fill_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, y, n, v);
  NASSEG_LAUNCH_CHECK("fill");
  return NASSEG_OK;
}

#if NASSEG_FP32_ONLY
// fp32 -> bf16 (round to nearest even) and bf16 -> fp32 of n values: the (B, C, 1, 1) maps on either side of
// GAPConv1x1's fp32 island in a bf16-storage network (layer_factory.py:181-195; functional._GlobalAvgPool /
// _Broadcast).  torch's .to() does the same arithmetic - as an ATen kernel that a recorded step cannot place
// (engine/graph_dag.py treats what it does not know as a barrier).
__global__ void to_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) sta1(y + i, x[i]);
}
__global__ void from_bf16_kernel(const bf16_t* __restrict__ x, float* __restrict__ y, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = lda1(x + i);
}
extern "C" int nasseg_to_bf16(const float* x, uint16_t* y, int64_t n, void* stream) {
  if (n <= 0) return NASSEG_OK;
  NASSEG_REQUIRE(x && y, "to_bf16: null pointer");
  hipLaunchKernelGGL(to_bf16_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, (hipStream_t)stream, x,
                     reinterpret_cast<bf16_t*>(y), n);
  NASSEG_LAUNCH_CHECK("to_bf16");
  return NASSEG_OK;
}
extern "C" int nasseg_from_bf16(const uint16_t* x, float* y, int64_t n, void* stream) {
  if (n <= 0) return NASSEG_OK;
  NASSEG_REQUIRE(x && y, "from_bf16: null pointer");
  hipLaunchKernelGGL(from_bf16_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const bf16_t*>(x), y, n);
  NASSEG_LAUNCH_CHECK("from_bf16");
  return NASSEG_OK;
}
#endif

int NASSEG_FN(chan_copy)(const act_t* x, int64_t ldx, int xoff, act_t* y, int64_t ldy, int yoff,
                     const act_t* mref, int64_t ldm, int moff, int64_t P, int C, int act, int mact,
                     void* stream) {
  NASSEG_REQUIRE(C > 0 && C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && xoff % 4 == 0 &&
                     yoff % 4 == 0 && (!mref || (ldm % 4 == 0 && moff % 4 == 0)),
                 "chan_copy: channels must be multiples of 4");
  if (P <= 0) return NASSEG_OK;
  hipLaunchKernelGGL(chan_copy_kernel, dim3(ew_grid(P * C / 4)), dim3(256), 0, (hipStream_t)stream,
                     x, ldx, xoff, y, ldy, yoff, mref, ldm, moff, P, C / 4, act, mact);
  NASSEG_LAUNCH_CHECK("chan_copy");
  return NASSEG_OK;
}

int NASSEG_FN(chan_fold)(const act_t* dy, act_t* dx, int64_t P, int C, int rep, void* stream) {
  NASSEG_REQUIRE(C > 0 && C % 4 == 0 && rep > 0, "chan_fold: bad shape");
  if (P <= 0) return NASSEG_OK;
  hipLaunchKernelGGL(chan_fold_kernel, dim3(ew_grid(P * C / 4)), dim3(256), 0, (hipStream_t)stream,
                     dy, dx, P, C / 4, rep);
  NASSEG_LAUNCH_CHECK("chan_fold");
  return NASSEG_OK;
}

#if NASSEG_FP32_ONLY
// dst[i][:] = src[idx[i]][:], i < n: rows of row_bytes bytes (feature maps of one sample, fp32 or
// bf16, or its int64 label map); idx on the device, clamped to [0, n_src)
int nasseg_gather_rows(const void* src, const int64_t* idx, void* dst, int n, int64_t row_bytes, int64_t n_src,
                       void* stream) {
  NASSEG_REQUIRE(src && idx && dst && n >= 0 && row_bytes > 0 && n_src > 0 && n <= 65535,
                 "gather_rows: bad arguments");
  if (n == 0) return NASSEG_OK;
  hipStream_t s = (hipStream_t)stream;
  const bool a16 = row_bytes % 16 == 0 && ((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 16 == 0);
  const bool a4 = row_bytes % 4 == 0 && ((uintptr_t)src % 4 == 0) && ((uintptr_t)dst % 4 == 0);
  const int64_t per = a16 ? row_bytes / 16 : (a4 ? row_bytes / 4 : row_bytes);
  int64_t gx = (per + 255) / 256;
  if (gx > 1024) gx = 1024;
  const dim3 grid((unsigned)gx, (unsigned)n);
  if (a16)
    hipLaunchKernelGGL(gather_rows_kernel<float4>, grid, dim3(256), 0, s, (const float4*)src, idx, (float4*)dst, per, n_src);
  else if (a4)
    hipLaunchKernelGGL(gather_rows_kernel<float>, grid, dim3(256), 0, s, (const float*)src, idx, (float*)dst, per, n_src);
  else
    hipLaunchKernelGGL(gather_rows_kernel<uint8_t>, grid, dim3(256), 0, s, (const uint8_t*)src, idx, (uint8_t*)dst, per, n_src);
  NASSEG_LAUNCH_CHECK("gather_rows");
  return NASSEG_OK;
}
#endif  // NASSEG_FP32_ONLY

}  // extern "C"
