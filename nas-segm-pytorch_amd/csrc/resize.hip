// Bilinear resize (align_corners=False, no antialias; up- and down-sampling)
// and nearest label resize, fp32 NHWC, gfx950.
//
// Reference call sites: resize() / Adapt (src/nn/layer_factory.py:316-350),
// GAPConv1x1 broadcast (:190-194), collect_all and AggregateCell
// (src/nn/micro_decoders.py:11-25,46-51), loss / validation up-sampling
// (src/engine/trainer.py:141-143,153-155,236-238,245-247; inference.py:58-60).
//
// Index math follows torch's area_pixel_compute_source_index:
//   scale = in/out (fp32); src = max(scale*(dst+0.5)-0.5, 0); i0 = floor(src);
//   i1 = i0 + (i0 < in-1); l1 = src - i0; l0 = 1 - l1.
// Forward can write into a channel slice of a wider slab (ldy / yoff) and apply
// a ReLU, which is how collect_all's cat+relu is produced without an extra pass.
// Backward is a gather over the destination footprint of each source pixel
// (deterministic; no atomics).
#include "common.h"
#include "dw_common.h"

namespace {

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
  float l1 = src - (float)r.i0;
  l1 = fminf(fmaxf(l1, 0.f), 1.f);
  r.l1 = l1;
  r.l0 = 1.f - l1;
  return r;
}

// align_corners=True (the KD teacher's up-sampling, src/kd/rf_lw/model_lw_v2.py:258,266,274): torch's
// area_pixel_compute_source_index with align_corners: src = dst * (in-1)/(out-1) (scale 0 when out == 1)
__device__ __forceinline__ Lin lin_coeff_ac(int dst, float scale, int in_size) {
  Lin r;
  const float src = scale * (float)dst;
  r.i0 = (int)src;
  if (r.i0 > in_size - 1) r.i0 = in_size - 1;
  r.i1 = r.i0 + ((r.i0 < in_size - 1) ? 1 : 0);
  float l1 = src - (float)r.i0;
  l1 = fminf(fmaxf(l1, 0.f), 1.f);
  r.l1 = l1;
  r.l0 = 1.f - l1;
  return r;
}

inline int rs_grid(int64_t n) {
  int64_t b = (n + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

template <bool AC>
__global__ __launch_bounds__(256) void bilinear_fwd_kernel(const act_t* __restrict__ x,
                                                           act_t* __restrict__ y, int64_t ldy,
                                                           int yoff, int B, int Hi, int Wi, int C4,
                                                           int Ho, int Wo, float sh, float sw,
                                                           int act) {
  const int C = C4 * 4;
  const int64_t total = (int64_t)B * Ho * Wo * C4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    int64_t p = i / C4;
    const int ox = (int)(p % Wo);
    const int64_t prow = p / Wo;
    const int oy = (int)(prow % Ho);
    const int b = (int)(prow / Ho);
    const Lin ly = AC ? lin_coeff_ac(oy, sh, Hi) : lin_coeff(oy, sh, Hi, Ho);
    const Lin lx = AC ? lin_coeff_ac(ox, sw, Wi) : lin_coeff(ox, sw, Wi, Wo);
    const act_t* xb = x + (int64_t)b * Hi * Wi * C + c4 * 4;
    const float4 v00 = lda4(xb + ((int64_t)ly.i0 * Wi + lx.i0) * C);
    const float4 v01 = lda4(xb + ((int64_t)ly.i0 * Wi + lx.i1) * C);
    const float4 v10 = lda4(xb + ((int64_t)ly.i1 * Wi + lx.i0) * C);
    const float4 v11 = lda4(xb + ((int64_t)ly.i1 * Wi + lx.i1) * C);
    float4 o;
    o.x = ly.l0 * (lx.l0 * v00.x + lx.l1 * v01.x) + ly.l1 * (lx.l0 * v10.x + lx.l1 * v11.x);
    o.y = ly.l0 * (lx.l0 * v00.y + lx.l1 * v01.y) + ly.l1 * (lx.l0 * v10.y + lx.l1 * v11.y);
    o.z = ly.l0 * (lx.l0 * v00.z + lx.l1 * v01.z) + ly.l1 * (lx.l0 * v10.z + lx.l1 * v11.z);
    o.w = ly.l0 * (lx.l0 * v00.w + lx.l1 * v01.w) + ly.l1 * (lx.l0 * v10.w + lx.l1 * v11.w);
    sta4(y + p * ldy + yoff + c4 * 4, act_apply4(o, act));
  }
}

// scalar-channel variant (C not a multiple of 4: class logits)
__global__ __launch_bounds__(256) void bilinear_fwd_scalar_kernel(const act_t* __restrict__ x,
                                                                  act_t* __restrict__ y, int B,
                                                                  int Hi, int Wi, int C, int Ho,
                                                                  int Wo, float sh, float sw) {
  const int64_t total = (int64_t)B * Ho * Wo * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    int64_t p = i / C;
    const int ox = (int)(p % Wo);
    const int64_t prow = p / Wo;
    const int oy = (int)(prow % Ho);
    const int b = (int)(prow / Ho);
    const Lin ly = lin_coeff(oy, sh, Hi, Ho);
    const Lin lx = lin_coeff(ox, sw, Wi, Wo);
    const act_t* xb = x + (int64_t)b * Hi * Wi * C + c;
    const float v00 = lda1(xb + ((int64_t)ly.i0 * Wi + lx.i0) * C);
    const float v01 = lda1(xb + ((int64_t)ly.i0 * Wi + lx.i1) * C);
    const float v10 = lda1(xb + ((int64_t)ly.i1 * Wi + lx.i0) * C);
    const float v11 = lda1(xb + ((int64_t)ly.i1 * Wi + lx.i1) * C);
    sta1(y + i, ly.l0 * (lx.l0 * v00 + lx.l1 * v01) + ly.l1 * (lx.l0 * v10 + lx.l1 * v11));
  }
}

__device__ __forceinline__ float4 keep_if4(float4 v, bool ok) {
  const unsigned m = 0u - (unsigned)ok;
  v.x = __uint_as_float(__float_as_uint(v.x) & m);
  v.y = __uint_as_float(__float_as_uint(v.y) & m);
  v.z = __uint_as_float(__float_as_uint(v.z) & m);
  v.w = __uint_as_float(__float_as_uint(v.w) & m);
  return v;
}

// One input of a channel concatenation, written into its slice [yoff, yoff + C) of the slab y (row stride
// ldy): bilinearly resized (align_corners=False) when RESIZE, with the producer's pending BatchNorm +
// activation - act(scale*v + shift), scale / shift null = none - applied to the SOURCE values as they are
// loaded (the producer never wrote its normalised output), and the per-workgroup sums of the slab's own
// BatchNorm statistics: stats[blk][0][c] = sum y, stats[blk][1][c] = sum y*y over what this workgroup wrote
// (rows of 2*ldy floats, columns yoff + c; null = none).
// workgroup (bx, by): lanes = 256 consecutive (x, channel-group) positions of an output row, rows by,
// by + gdy, ... of the flattened (image, row) axis - the layout block_reduce_groups sums over.
template <bool RESIZE>
__global__ __launch_bounds__(256) void cat_src_fwd_kernel(const act_t* __restrict__ x, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, int act,
                                                          act_t* __restrict__ y, int64_t ldy, int yoff,
                                                          float* __restrict__ stats, int B, int Hi, int Wi, int C4,
                                                          int Ho, int Wo, float sh, float sw) {
  __shared__ float4 sred[2][4][64];
  const int C = C4 * 4;
  const int tid = threadIdx.x;
  const int base = blockIdx.x * 256;
  const int pos = base + tid;
  const bool live = pos < Wo * C4;
  const int ox = live ? pos / C4 : 0;
  const int c4 = live ? pos - ox * C4 : 0;
  const float4 sc = scale ? lda4(scale + c4 * 4) : make_float4(1.f, 1.f, 1.f, 1.f);
  const float4 sf = shift ? lda4(shift + c4 * 4) : f4zero();
  const Lin lx = lin_coeff(ox, sw, Wi, Wo);
  float4 ssum[2] = {f4zero(), f4zero()};
  const int R = B * Ho;
  for (int r = blockIdx.y; r < R; r += gridDim.y) {
    const int b = r / Ho, oy = r - b * Ho;
    float4 o;
    if (RESIZE) {
      const Lin ly = lin_coeff(oy, sh, Hi, Ho);
      const act_t* xb = x + (int64_t)b * Hi * Wi * C + c4 * 4;
      const float4 v00 = act_apply4(fma4(lda4(xb + ((int64_t)ly.i0 * Wi + lx.i0) * C), sc, sf), act);
      const float4 v01 = act_apply4(fma4(lda4(xb + ((int64_t)ly.i0 * Wi + lx.i1) * C), sc, sf), act);
      const float4 v10 = act_apply4(fma4(lda4(xb + ((int64_t)ly.i1 * Wi + lx.i0) * C), sc, sf), act);
      const float4 v11 = act_apply4(fma4(lda4(xb + ((int64_t)ly.i1 * Wi + lx.i1) * C), sc, sf), act);
      o.x = ly.l0 * (lx.l0 * v00.x + lx.l1 * v01.x) + ly.l1 * (lx.l0 * v10.x + lx.l1 * v11.x);
      o.y = ly.l0 * (lx.l0 * v00.y + lx.l1 * v01.y) + ly.l1 * (lx.l0 * v10.y + lx.l1 * v11.y);
      o.z = ly.l0 * (lx.l0 * v00.z + lx.l1 * v01.z) + ly.l1 * (lx.l0 * v10.z + lx.l1 * v11.z);
      o.w = ly.l0 * (lx.l0 * v00.w + lx.l1 * v01.w) + ly.l1 * (lx.l0 * v10.w + lx.l1 * v11.w);
    } else {
      o = act_apply4(fma4(lda4(x + ((int64_t)r * Wo + ox) * C + c4 * 4), sc, sf), act);
    }
    if (live) sta4(y + ((int64_t)r * Wo + ox) * ldy + yoff + c4 * 4, o);
#ifdef NASSEG_BF16
    o = make_float4(bf16_to_f32(f32_to_bf16(o.x)), bf16_to_f32(f32_to_bf16(o.y)), bf16_to_f32(f32_to_bf16(o.z)),
                    bf16_to_f32(f32_to_bf16(o.w)));  // (the statistics of what a pass over the slab would read)
#endif
    o = keep_if4(o, live);
    ssum[0] = add4(ssum[0], o);
    ssum[1] = fma4(o, o, ssum[1]);
  }
  if (stats) {
    const int blk = blockIdx.y * gridDim.x + blockIdx.x;
    block_reduce_groups<2, 2>(ssum, sred, stats + (size_t)blk * 2 * ldy + yoff, base, C4, (int)ldy);
  }
}

// Backward of one input of the concatenation: the slab BatchNorm's backward applied to this input's channel
// slice - v = scale*(du - s0/M - xhat*s1/M) with du the ReLU-masked gradient nasseg_conv_bwd_data_bn left and
// xhat from the slab itself (eval: v = scale*du) - written densely [rows][C] (the gradient w.r.t. this input at
// the slab's size: final when the input had that size, the operand of nasseg_bilinear_bwd otherwise).  When the
// input came in pending (z + the producer's BatchNorm statistics), v is multiplied by act'(tscale*z + tshift)
// and the producer's BatchNorm-backward sums {sum g, sum g*(z - tmean)*tinvstd} are emitted per workgroup
// (rows of 2*C floats): the producer's backward then needs no reduction pass over g and z.  RESIZE: the pending
// producer has ANOTHER size (z [B][Hi][Wi][C]): v is written unmasked (nasseg_bilinear_bwd_act masks what it
// transposes), the sums are formed against the interpolated mask / mask * xhat.
// Same workgroup layout as cat_src_fwd_kernel.
// ZSAME: the input came in pending and has the slab's size - its slice of the slab is act(tscale*z + tshift), what
// nasseg_cat_src_fwd wrote there: it is rebuilt from z (which is read for the mask anyway) instead of being loaded
// (round 5: three tensor passes instead of four).
template <bool RESIZE, bool ZSAME = false>
__global__ __launch_bounds__(256) void cat_src_bwd_kernel(
    const act_t* __restrict__ du, const act_t* __restrict__ slab, int64_t ld, int off,
    const float* __restrict__ sscale, const float* __restrict__ smean, const float* __restrict__ sinvstd,
    const float* __restrict__ sums, float invM, int train, const act_t* __restrict__ z,
    const float* __restrict__ tstats, int act, act_t* __restrict__ g, float* __restrict__ part, int R, int Wo,
    int C4, int Ho, int Hi, int Wi, float sh, float sw) {
  __shared__ float4 sred[2][4][64];
  const int C = C4 * 4;
  const int tid = threadIdx.x;
  const int base = blockIdx.x * 256;
  const int pos = base + tid;
  const bool live = pos < Wo * C4;
  const int ox = live ? pos / C4 : 0;
  const int c4 = live ? pos - ox * C4 : 0;
  const int sc = off + c4 * 4;  // channel of the slab
  const float4 ssc = lda4(sscale + sc), smu = lda4(smean + sc), sis = lda4(sinvstd + sc);
  const float4 s0 = lda4(sums + sc), s1 = lda4(sums + ld + sc);
  float4 tmu = f4zero(), tis = f4zero(), tsc = f4zero(), tsh = f4zero();
  if (z) {
    tmu = lda4(tstats + c4 * 4);
    tis = lda4(tstats + C + c4 * 4);
    tsc = lda4(tstats + 2 * C + c4 * 4);
    tsh = lda4(tstats + 3 * C + c4 * 4);
  }
  float4 ssum[2] = {f4zero(), f4zero()};
  const Lin lxz = RESIZE ? lin_coeff(ox, sw, Wi, Wo) : Lin{0, 0, 1.f, 0.f};
  for (int r = blockIdx.y; r < R; r += gridDim.y) {
    const int64_t pix = (int64_t)r * Wo + ox;
    const float4 d = lda4(du + pix * ld + sc);
    float4 v = d;
    float4 zs = f4zero(), ts = f4zero();
    if (ZSAME) {
      zs = lda4(z + pix * C + c4 * 4);
      ts = fma4(zs, tsc, tsh);
    }
    if (train) {
      float4 x;
      if (ZSAME) {
        x = act_apply4(ts, act);  // (the expression nasseg_cat_src_fwd stored)
#ifdef NASSEG_BF16
        x = make_float4(bf16_to_f32(f32_to_bf16(x.x)), bf16_to_f32(f32_to_bf16(x.y)), bf16_to_f32(f32_to_bf16(x.z)),
                        bf16_to_f32(f32_to_bf16(x.w)));
#endif
      } else {
        x = lda4(slab + pix * ld + sc);
      }
      v.x = d.x - s0.x * invM - (x.x - smu.x) * sis.x * s1.x * invM;
      v.y = d.y - s0.y * invM - (x.y - smu.y) * sis.y * s1.y * invM;
      v.z = d.z - s0.z * invM - (x.z - smu.z) * sis.z * s1.z * invM;
      v.w = d.w - s0.w * invM - (x.w - smu.w) * sis.w * s1.w * invM;
    }
    v = mul4(v, ssc);
    float4 xh = f4zero();
    float4 ms = make_float4(1.f, 1.f, 1.f, 1.f);  // RESIZE: interpolated mask, xh: interpolated mask * xhat
    if (ZSAME || (z && !RESIZE)) {
      const float4 zv = ZSAME ? zs : lda4(z + pix * C + c4 * 4);
      const float4 t = ZSAME ? ts : fma4(zv, tsc, tsh);
      v = make_float4(v.x * act_mask(t.x, act), v.y * act_mask(t.y, act), v.z * act_mask(t.z, act),
                      v.w * act_mask(t.w, act));
      xh = make_float4((zv.x - tmu.x) * tis.x, (zv.y - tmu.y) * tis.y, (zv.z - tmu.z) * tis.z,
                       (zv.w - tmu.w) * tis.w);
    }
    if (z && RESIZE) {
      // the producer's gradient is g(p) = m(p) * sum_o w(o, p) v(o) (nasseg_bilinear_bwd_act forms it): its sums
      // over p are sum_o v(o) * interp[m](o) and sum_o v(o) * interp[m * xhat](o) - formed here, at the slab's size,
      // from the 4 taps the forward read
      const int b = r / Ho, oy = r - b * Ho;
      const Lin ly = lin_coeff(oy, sh, Hi, Ho);
      const act_t* zb = z + (int64_t)b * Hi * Wi * C + c4 * 4;
      ms = f4zero();
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int yy = (k & 2) ? ly.i1 : ly.i0, xx = (k & 1) ? lxz.i1 : lxz.i0;
        const float w = ((k & 2) ? ly.l1 : ly.l0) * ((k & 1) ? lxz.l1 : lxz.l0);
        const float4 zv = lda4(zb + ((int64_t)yy * Wi + xx) * C);
        const float4 t = fma4(zv, tsc, tsh);
        const float4 m = make_float4(w * act_mask(t.x, act), w * act_mask(t.y, act), w * act_mask(t.z, act),
                                     w * act_mask(t.w, act));
        ms = add4(ms, m);
        xh.x = fmaf(m.x, (zv.x - tmu.x) * tis.x, xh.x);
        xh.y = fmaf(m.y, (zv.y - tmu.y) * tis.y, xh.y);
        xh.z = fmaf(m.z, (zv.z - tmu.z) * tis.z, xh.z);
        xh.w = fmaf(m.w, (zv.w - tmu.w) * tis.w, xh.w);
      }
    }
    if (live) sta4(g + pix * C + c4 * 4, v);
#ifdef NASSEG_BF16
    v = make_float4(bf16_to_f32(f32_to_bf16(v.x)), bf16_to_f32(f32_to_bf16(v.y)), bf16_to_f32(f32_to_bf16(v.z)),
                    bf16_to_f32(f32_to_bf16(v.w)));  // (what a reduction pass over g would read)
#endif
    v = keep_if4(v, live);
    ssum[0] = RESIZE ? fma4(v, ms, ssum[0]) : add4(ssum[0], v);
    ssum[1] = fma4(v, xh, ssum[1]);
  }
  if (part) {
    const int blk = blockIdx.y * gridDim.x + blockIdx.x;
    block_reduce_groups<2, 2>(ssum, sred, part + (size_t)blk * 2 * C, base, C4);
  }
}

// Backward of ParamSum, out = ca[c] * ya + cb[c] * yb (src/nn/layer_factory.py:353-366), for BOTH operands from one
// read of the gradient: y_s = act_s(scale_s * z_s + shift_s) where operand s is a conv chain's raw output with its
// BatchNorm + activation pending (tstats_s: mean | invstd | scale | shift), else y_s = z_s.  Writes
// g_s = c_s * dy * act_s'(...) (the gradient the producer's chain takes: masked, with its BatchNorm-backward sums
// {sum g, sum g * xhat} as per-workgroup rows part_s) and the coefficient gradients' rows cpart [blk][2][C] =
// {sum dy * ya, sum dy * yb}.  Same workgroup layout as cat_src_fwd_kernel.  Replaces two scaling passes, a two-dot
// reduction and the producers' two mask-and-reduce passes: 5 tensor passes instead of 11, 1 launch instead of 8.
__global__ __launch_bounds__(256) void psum_bwd_kernel(
    const act_t* __restrict__ dy, const act_t* __restrict__ za, const float* __restrict__ tsa, int act_a,
    const float* __restrict__ ca, act_t* __restrict__ ga, float* __restrict__ part_a, const act_t* __restrict__ zb,
    const float* __restrict__ tsb, int act_b, const float* __restrict__ cb, act_t* __restrict__ gb,
    float* __restrict__ part_b, float* __restrict__ cpart, int R, int Wo, int C4) {
  __shared__ float4 sred[2][4][64];
  const int C = C4 * 4;
  const int tid = threadIdx.x;
  const int base = blockIdx.x * 256;
  const int pos = base + tid;
  const bool live = pos < Wo * C4;
  const int ox = live ? pos / C4 : 0;
  const int c4 = live ? pos - ox * C4 : 0;
  const float4 one = make_float4(1.f, 1.f, 1.f, 1.f);
  float4 mu[2] = {f4zero(), f4zero()}, is[2] = {f4zero(), f4zero()}, sc[2] = {one, one}, sh[2] = {f4zero(), f4zero()};
  const float* ts[2] = {tsa, tsb};
#pragma unroll
  for (int s = 0; s < 2; ++s)
    if (ts[s]) {
      mu[s] = lda4(ts[s] + c4 * 4);
      is[s] = lda4(ts[s] + C + c4 * 4);
      sc[s] = lda4(ts[s] + 2 * C + c4 * 4);
      sh[s] = lda4(ts[s] + 3 * C + c4 * 4);
    }
  const float4 cf[2] = {ca ? lda4(ca + c4 * 4) : one, cb ? lda4(cb + c4 * 4) : one};
  const ActSel as[2] = {act_sel(tsa ? act_a : NASSEG_ACT_NONE), act_sel(tsb ? act_b : NASSEG_ACT_NONE)};
  const act_t* zz[2] = {za, zb};
  act_t* gg[2] = {ga, gb};
  float4 sa[2] = {f4zero(), f4zero()}, sb[2] = {f4zero(), f4zero()}, scf[2] = {f4zero(), f4zero()};
  for (int r = blockIdx.y; r < R; r += gridDim.y) {
    const int64_t e = ((int64_t)r * Wo + ox) * C + c4 * 4;
    const float4 d = keep_if4(lda4(dy + e), live);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const float4 zv = lda4(zz[s] + e);
      const float4 t = fma4(zv, sc[s], sh[s]);
      scf[s] = fma4(d, act_apply4(t, as[s]), scf[s]);
      float4 g = mul4(d, cf[s]);
      g = make_float4(g.x * act_mask(t.x, as[s]), g.y * act_mask(t.y, as[s]), g.z * act_mask(t.z, as[s]),
                      g.w * act_mask(t.w, as[s]));
      if (live && gg[s]) sta4(gg[s] + e, g);
#ifdef NASSEG_BF16
      g = make_float4(bf16_to_f32(f32_to_bf16(g.x)), bf16_to_f32(f32_to_bf16(g.y)), bf16_to_f32(f32_to_bf16(g.z)),
                      bf16_to_f32(f32_to_bf16(g.w)));  // (what a reduction pass over g would read)
#endif
      const float4 xh = make_float4((zv.x - mu[s].x) * is[s].x, (zv.y - mu[s].y) * is[s].y, (zv.z - mu[s].z) * is[s].z,
                                    (zv.w - mu[s].w) * is[s].w);
      float4(&acc)[2] = s == 0 ? sa : sb;
      acc[0] = add4(acc[0], g);
      acc[1] = fma4(g, xh, acc[1]);
    }
  }
  const int blk = blockIdx.y * gridDim.x + blockIdx.x;
  if (part_a) block_reduce_groups<2, 2>(sa, sred, part_a + (size_t)blk * 2 * C, base, C4);
  if (part_b) {
    __syncthreads();
    block_reduce_groups<2, 2>(sb, sred, part_b + (size_t)blk * 2 * C, base, C4);
  }
  __syncthreads();
  block_reduce_groups<2, 2>(scf, sred, cpart + (size_t)blk * 2 * C, base, C4);
}

// Gradient junction of a node with several consumers (a cell's node read by several ops, a block's output read by the
// next repeats / blocks and collect_all: src/nn/micro_decoders.py:95-121,380-398): the NG gradients its consumers
// returned are summed in index order - one launch and NG + 1 tensor passes where autograd's accumulation took NG - 1
// launches of three passes each - and, when the node is a conv chain's raw output whose BatchNorm + activation is
// pending (ts: mean | invstd | scale | shift), the sum is multiplied by act'(scale*z + shift) and comes with that
// BatchNorm's backward sums {sum g, sum g * xhat} as per-workgroup rows `part` (the chain's own mask-and-reduce pass
// over gradient and z is gone).  Workgroup layout of cat_src_fwd_kernel / psum_bwd_kernel.
struct JunctionSrc {
  const act_t* g[8];
};
template <int NG>
__global__ __launch_bounds__(256) void grad_junction_kernel(JunctionSrc src, const act_t* __restrict__ z,
                                                            const float* __restrict__ ts, int act,
                                                            act_t* __restrict__ out, float* __restrict__ part, int R,
                                                            int Wo, int C4) {
  __shared__ float4 sred[2][4][64];
  const int C = C4 * 4;
  const int tid = threadIdx.x;
  const int base = blockIdx.x * 256;
  const int pos = base + tid;
  const bool live = pos < Wo * C4;
  const int ox = live ? pos / C4 : 0;
  const int c4 = live ? pos - ox * C4 : 0;
  float4 mu = f4zero(), is = f4zero(), sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = f4zero();
  if (ts) {
    mu = lda4(ts + c4 * 4);
    is = lda4(ts + C + c4 * 4);
    sc = lda4(ts + 2 * C + c4 * 4);
    sh = lda4(ts + 3 * C + c4 * 4);
  }
  const ActSel as = act_sel(ts ? act : NASSEG_ACT_NONE);
  float4 acc[2] = {f4zero(), f4zero()};
  for (int r = blockIdx.y; r < R; r += gridDim.y) {
    const int64_t e = ((int64_t)r * Wo + ox) * C + c4 * 4;
    float4 gv[NG];
#pragma unroll
    for (int i = 0; i < NG; ++i) gv[i] = lda4(src.g[i] + e);
    float4 zv = f4zero();
    if (ts) zv = lda4(z + e);
    float4 g = gv[0];
#pragma unroll
    for (int i = 1; i < NG; ++i) g = add4(g, gv[i]);
    if (ts) {
      const float4 t = fma4(zv, sc, sh);
      g = make_float4(g.x * act_mask(t.x, as), g.y * act_mask(t.y, as), g.z * act_mask(t.z, as), g.w * act_mask(t.w, as));
    }
    g = keep_if4(g, live);
    if (live) sta4(out + e, g);
    if (part) {
#ifdef NASSEG_BF16
      g = make_float4(bf16_to_f32(f32_to_bf16(g.x)), bf16_to_f32(f32_to_bf16(g.y)), bf16_to_f32(f32_to_bf16(g.z)),
                      bf16_to_f32(f32_to_bf16(g.w)));  // (what a reduction pass over the stored gradient would read)
#endif
      const float4 xh = make_float4((zv.x - mu.x) * is.x, (zv.y - mu.y) * is.y, (zv.z - mu.z) * is.z,
                                    (zv.w - mu.w) * is.w);
      acc[0] = add4(acc[0], g);
      acc[1] = fma4(g, xh, acc[1]);
    }
  }
  if (part) {
    const int blk = blockIdx.y * gridDim.x + blockIdx.x;
    block_reduce_groups<2, 2>(acc, sred, part + (size_t)blk * 2 * C, base, C4);
  }
}

struct CatGrid {
  int gx, gy;
};
inline CatGrid cat_grid(int B, int Ho, int Wo, int C) {
  CatGrid g;
  g.gx = cdiv(Wo * (C / 4), 256);
  const int64_t rows = (int64_t)B * Ho;
  // ~2048 workgroups on large maps, <= 512 (one-level finalisation of the statistics rows) on small ones
  int64_t gy = ((int64_t)g.gx * rows >= 8192 ? 2048 : 512) / g.gx;
  if (gy > rows / 2) gy = rows / 2;
  if (gy < 1) gy = 1;
  if (gy > 65535) gy = 65535;
  g.gy = (int)gy;
  return g;
}

// conservative destination range [lo, hi] whose source footprint can touch index i
__device__ __forceinline__ void dst_range(int i, float scale, int out_size, int& lo, int& hi) {
  // src(o) in [i-1, i+1)  <=>  o in [(i-0.5)/scale - 0.5, (i+1.5)/scale - 0.5)
  const float inv = 1.0f / scale;
  lo = (int)floorf(((float)i - 0.5f) * inv - 0.5f) - 1;
  hi = (int)ceilf(((float)i + 1.5f) * inv - 0.5f) + 1;
  if (lo < 0) lo = 0;
  if (hi > out_size - 1) hi = out_size - 1;
}
__device__ __forceinline__ float lin_weight(int o, int i, float scale, int in_size, int out_size) {
  const Lin l = lin_coeff(o, scale, in_size, out_size);
  float w = 0.f;
  if (l.i0 == i) w += l.l0;
  if (l.i1 == i) w += l.l1;
  return w;
}

// dx * act'(scale*z + shift) when a mask tensor is given (the producer's pending activation: nasseg_bilinear_bwd_act)
__device__ __forceinline__ float4 mask_by(float4 g, const act_t* __restrict__ mz, const float* __restrict__ msc,
                                          const float* __restrict__ msh, int mact, int64_t elem, int c) {
  if (!mz) return g;
  const float4 t = fma4(lda4(mz + elem), lda4(msc + c), lda4(msh + c));
  return make_float4(g.x * act_mask(t.x, mact), g.y * act_mask(t.y, mact), g.z * act_mask(t.z, mact),
                     g.w * act_mask(t.w, mact));
}

template <int VEC>
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const act_t* __restrict__ dy,
                                                           int64_t lddy, int dyoff,
                                                           act_t* __restrict__ dx, int B, int Hi,
                                                           int Wi, int CV, int Ho, int Wo, float sh,
                                                           float sw, const act_t* __restrict__ mz,
                                                           const float* __restrict__ msc,
                                                           const float* __restrict__ msh, int mact) {
  const int64_t total = (int64_t)B * Hi * Wi * CV;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cv = (int)(i % CV);
    int64_t p = i / CV;
    const int ix = (int)(p % Wi);
    p /= Wi;
    const int iy = (int)(p % Hi);
    const int b = (int)(p / Hi);
    int ylo, yhi, xlo, xhi;
    if (Hi == Ho) { ylo = yhi = iy; } else dst_range(iy, sh, Ho, ylo, yhi);
    if (Wi == Wo) { xlo = xhi = ix; } else dst_range(ix, sw, Wo, xlo, xhi);
    float4 g = f4zero();
    for (int oy = ylo; oy <= yhi; ++oy) {
      const float wy = lin_weight(oy, iy, sh, Hi, Ho);
      if (wy == 0.f) continue;
      const act_t* drow = dy + (((int64_t)b * Ho + oy) * Wo) * lddy + dyoff + cv * VEC;
      for (int ox = xlo; ox <= xhi; ++ox) {
        const float wx = lin_weight(ox, ix, sw, Wi, Wo);
        if (wx == 0.f) continue;
        const float w = wy * wx;
        if (VEC == 4) {
          const float4 d = lda4(drow + (int64_t)ox * lddy);
          g.x = fmaf(w, d.x, g.x);
          g.y = fmaf(w, d.y, g.y);
          g.z = fmaf(w, d.z, g.z);
          g.w = fmaf(w, d.w, g.w);
        } else {
          g.x = fmaf(w, lda1(drow + (int64_t)ox * lddy), g.x);
        }
      }
    }
    if (VEC == 4)
      sta4(dx + i * 4, mask_by(g, mz, msc, msh, mact, i * 4, cv * 4));
    else
      sta1(dx + i, g.x);
  }
}

// The same gather with the destination window of a source pixel made tight and walked without branches:
// dst o contributes to source i iff src(o) in (i-1, i+1), i.e. o in ((i-0.5)/scale - 0.5, (i+1.5)/scale - 0.5)
// - at most 4 per axis when up-sampling by <= 2 (NW = 4), at most 2 when down-sampling by >= 2 (NW = 2).
// The window is widened by 0.01 (never narrowed: a destination dropped by rounding would be one whose weight
// is rounding noise; one added has weight exactly 0 from lin_weight); all NW x NW loads are issued
// unconditionally (clamped coordinates, masked values), anything beyond NW (never, for the factors the
// launcher sends here) is walked by the tail loops.
__device__ __forceinline__ void dst_range_tight(int i, float scale, int in_size, int out_size, int& lo, int& hi) {
  if (in_size == out_size) {
    lo = hi = i;
    return;
  }
  const float inv = 1.0f / scale;
  lo = (int)ceilf(((float)i - 0.5f) * inv - 0.5f - 0.01f);
  hi = (int)floorf(((float)i + 1.5f) * inv - 0.5f + 0.01f);
  if (lo < 0) lo = 0;
  if (hi > out_size - 1) hi = out_size - 1;
}

template <int NW>
__global__ __launch_bounds__(256) void bilinear_bwd_win_kernel(const act_t* __restrict__ dy, int64_t lddy, int dyoff,
                                                               act_t* __restrict__ dx, int B, int Hi, int Wi, int C4,
                                                               int Ho, int Wo, float sh, float sw,
                                                               const act_t* __restrict__ mz,
                                                               const float* __restrict__ msc,
                                                               const float* __restrict__ msh, int mact) {
  const int64_t total = (int64_t)B * Hi * Wi * C4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    int64_t p = i / C4;
    const int ix = (int)(p % Wi);
    p /= Wi;
    const int iy = (int)(p % Hi);
    const int b = (int)(p / Hi);
    int ylo, yhi, xlo, xhi;
    dst_range_tight(iy, sh, Hi, Ho, ylo, yhi);
    dst_range_tight(ix, sw, Wi, Wo, xlo, xhi);
    float wy[NW], wx[NW];
    int oy[NW], ox[NW];
#pragma unroll
    for (int t = 0; t < NW; ++t) {
      // (a source pixel no destination samples has an EMPTY window, hi < lo - down-sampling leaves most of
      //  them so: the loads stay unconditional, at a clamped coordinate, with weight 0)
      const int o = ylo + t;
      const bool ok = o <= yhi;
      oy[t] = ok ? o : (ylo < Ho ? ylo : Ho - 1);
      wy[t] = ok ? lin_weight(oy[t], iy, sh, Hi, Ho) : 0.f;
      const int q = xlo + t;
      const bool okx = q <= xhi;
      ox[t] = okx ? q : (xlo < Wo ? xlo : Wo - 1);
      wx[t] = okx ? lin_weight(ox[t], ix, sw, Wi, Wo) : 0.f;
    }
    const act_t* db = dy + (int64_t)b * Ho * Wo * lddy + dyoff + c4 * 4;
    float4 d[NW][NW];
#pragma unroll
    for (int ty = 0; ty < NW; ++ty)
#pragma unroll
      for (int tx = 0; tx < NW; ++tx) d[ty][tx] = lda4(db + ((int64_t)oy[ty] * Wo + ox[tx]) * lddy);
    float4 g = f4zero();
#pragma unroll
    for (int ty = 0; ty < NW; ++ty)
#pragma unroll
      for (int tx = 0; tx < NW; ++tx) {
        const float w = wy[ty] * wx[tx];
        const float4 v = keep_if4(d[ty][tx], w != 0.f);
        g.x = fmaf(w, v.x, g.x);
        g.y = fmaf(w, v.y, g.y);
        g.z = fmaf(w, v.z, g.z);
        g.w = fmaf(w, v.w, g.w);
      }
    // (windows wider than NW: not reached for the scale factors this kernel is launched for)
    for (int o = ylo; o <= yhi; ++o)
      for (int q = xlo; q <= xhi; ++q) {
        if (o < ylo + NW && q < xlo + NW) continue;
        const float w = lin_weight(o, iy, sh, Hi, Ho) * lin_weight(q, ix, sw, Wi, Wo);
        if (w == 0.f) continue;
        const float4 v = lda4(db + ((int64_t)o * Wo + q) * lddy);
        g.x = fmaf(w, v.x, g.x);
        g.y = fmaf(w, v.y, g.y);
        g.z = fmaf(w, v.z, g.z);
        g.w = fmaf(w, v.w, g.w);
      }
    sta4(dx + i * 4, mask_by(g, mz, msc, msh, mact, i * 4, c4 * 4));
  }
}

// Separable form of the backward for large up-sampling factors (the footprint of one source
// pixel is ~(2f+2)^2 destination pixels: 324 taps at f = 8 in the joint gather, 2 x 18 here):
//   pass 1  tmp[b][oy][ix][c] = sum_ox wx(ox, ix) * dy[b][oy][ox][c]
//   pass 2  dx[b][iy][ix][c]  = sum_oy wy(oy, iy) * tmp[b][oy][ix][c]
// One axis per kernel (AXIS 1 = x, 0 = y); the destination window is walked without branches
// (clamped loads, zero weights drop out).
// (TS / TD: element types of src / dst - the intermediate of the two passes stays fp32)
template <int AXIS, typename TS, typename TD>
__global__ __launch_bounds__(256) void bilinear_bwd_axis_kernel(const TS* __restrict__ src,
                                                                int64_t lds, int soff,
                                                                TD* __restrict__ dst, int B,
                                                                int H, int Wsrc, int Wdst, int C4,
                                                                int Hdst, float scale,
                                                                const act_t* __restrict__ mz,
                                                                const float* __restrict__ msc,
                                                                const float* __restrict__ msh, int mact) {
  // AXIS 1: src [B][H][Wsrc][lds], dst [B][H][Wdst][C] (reduce along x: Wsrc = Wo, Wdst = Wi)
  // AXIS 0: src [B][H][Wsrc][C] with H = Ho, dst [B][Hdst][Wsrc][C]   (reduce along y)
  const int rows_out = AXIS == 1 ? H : Hdst;
  const int cols_out = AXIS == 1 ? Wdst : Wsrc;
  const int64_t total = (int64_t)B * rows_out * cols_out * C4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    int64_t p = i / C4;
    const int x = (int)(p % cols_out);
    p /= cols_out;
    const int y = (int)(p % rows_out);
    const int b = (int)(p / rows_out);
    const int idx = AXIS == 1 ? x : y;             // source-grid index this thread gathers for
    const int n_in = AXIS == 1 ? Wdst : Hdst;      // size of the low-resolution axis
    const int n_out = AXIS == 1 ? Wsrc : H;        // size of the high-resolution axis
    int lo, hi;
    dst_range(idx, scale, n_out, lo, hi);
    float4 g = f4zero();
    for (int o = lo; o <= hi; ++o) {
      const float w = lin_weight(o, idx, scale, n_in, n_out);
      const int64_t pix = AXIS == 1 ? ((int64_t)b * H + y) * Wsrc + o : ((int64_t)b * H + o) * Wsrc + x;
      const float4 d = keep_if4(lda4(src + pix * lds + soff + c4 * 4), w != 0.f);
      g.x = fmaf(w, d.x, g.x);
      g.y = fmaf(w, d.y, g.y);
      g.z = fmaf(w, d.z, g.z);
      g.w = fmaf(w, d.w, g.w);
    }
    sta4(dst + i * 4, mask_by(g, mz, msc, msh, mact, i * 4, c4 * 4));  // (mz: second pass only)
  }
}

// nearest resize of integer label maps (torch 'nearest': src = min(floor(dst*in/out), in-1))
template <typename TI>
__global__ __launch_bounds__(256) void nearest_label_kernel(const TI* __restrict__ x,
                                                            int64_t* __restrict__ y, int B, int Hi,
                                                            int Wi, int Ho, int Wo, float sh,
                                                            float sw) {
  const int64_t total = (int64_t)B * Ho * Wo;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ox = (int)(i % Wo);
    const int64_t t = i / Wo;
    const int oy = (int)(t % Ho);
    const int b = (int)(t / Ho);
    int iy = (int)floorf((float)oy * sh);
    int ix = (int)floorf((float)ox * sw);
    if (iy > Hi - 1) iy = Hi - 1;
    if (ix > Wi - 1) ix = Wi - 1;
    y[i] = (int64_t)x[((int64_t)b * Hi + iy) * Wi + ix];
  }
}

}  // namespace

extern "C" {

// y[b,oy,ox, yoff:yoff+C] = act(bilinear(x)[b,oy,ox,:]); x dense [B][Hi][Wi][C]
int NASSEG_FN(bilinear_fwd)(const act_t* x, act_t* y, int64_t ldy, int yoff, int B, int Hi, int Wi,
                        int C, int Ho, int Wo, int act, void* stream) {
  NASSEG_REQUIRE(B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0, "bilinear_fwd: bad shape");
  const float sh = (float)Hi / (float)Ho, sw = (float)Wi / (float)Wo;
  hipStream_t s = (hipStream_t)stream;
  if (C % 4 == 0 && ldy % 4 == 0 && yoff % 4 == 0) {
    hipLaunchKernelGGL(bilinear_fwd_kernel<false>, dim3(rs_grid((int64_t)B * Ho * Wo * (C / 4))),
                       dim3(256), 0, s, x, y, ldy, yoff, B, Hi, Wi, C / 4, Ho, Wo, sh, sw, act);
  } else {
    NASSEG_REQUIRE(ldy == C && yoff == 0 && act == 0,
                   "bilinear_fwd: slab output / activation need C %% 4 == 0");
    hipLaunchKernelGGL(bilinear_fwd_scalar_kernel, dim3(rs_grid((int64_t)B * Ho * Wo * C)),
                       dim3(256), 0, s, x, y, B, Hi, Wi, C, Ho, Wo, sh, sw);
  }
  NASSEG_LAUNCH_CHECK("bilinear_fwd");
  return NASSEG_OK;
}

// the same with align_corners=True (nn.Upsample(size, mode="bilinear", align_corners=True)): forward only -
// its one caller, the distillation teacher, runs under no_grad.  C %% 4 == 0.
int NASSEG_FN(bilinear_ac_fwd)(const act_t* x, act_t* y, int B, int Hi, int Wi, int C, int Ho, int Wo,
                               void* stream) {
  NASSEG_REQUIRE(B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0 && C % 4 == 0, "bilinear_ac_fwd: bad shape");
  const float sh = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f;
  const float sw = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
  hipLaunchKernelGGL(bilinear_fwd_kernel<true>, dim3(rs_grid((int64_t)B * Ho * Wo * (C / 4))), dim3(256), 0,
                     (hipStream_t)stream, x, y, (int64_t)C, 0, B, Hi, Wi, C / 4, Ho, Wo, sh, sw, 0);
  NASSEG_LAUNCH_CHECK("bilinear_ac_fwd");
  return NASSEG_OK;
}

#if NASSEG_FP32_ONLY
// rows of statistics nasseg_cat_src_fwd writes for a slab of this geometry (C: channels of ONE input; every
// input of the slab must have the same C for the rows to line up); 0: geometry not served
int64_t nasseg_cat_src_blocks(int B, int Ho, int Wo, int C) {
  if (B <= 0 || Ho <= 0 || Wo <= 0 || C <= 0 || C % 4 || C / 4 > 256) return 0;
  const CatGrid g = cat_grid(B, Ho, Wo, C);
  return (int64_t)g.gx * g.gy;
}
#else
int64_t nasseg_cat_src_blocks(int B, int Ho, int Wo, int C);
#endif

// one input x [B][Hi][Wi][C] of a concatenation -> y[b,oy,ox, yoff:yoff+C] (row stride ldy), resized when
// (Hi, Wi) != (Ho, Wo); scale / shift / act: the producer's pending BatchNorm + activation (null / 0 = none);
// stats: null or [nasseg_cat_src_blocks(B, Ho, Wo, C) + 64][2][ldy] floats - columns yoff.. of every row are
// written ({sum, sum of squares} of the slice, for nasseg_bn_finalize over the whole slab)
int NASSEG_FN(cat_src_fwd)(const act_t* x, const float* scale, const float* shift, int act, act_t* y, int64_t ldy,
                           int yoff, float* stats, int B, int Hi, int Wi, int C, int Ho, int Wo, void* stream) {
  NASSEG_REQUIRE(x && y && nasseg_cat_src_blocks(B, Ho, Wo, C) > 0 && Hi > 0 && Wi > 0 && ldy % 4 == 0 &&
                     yoff % 4 == 0 && yoff + C <= ldy && (!scale == !shift),
                 "cat_src_fwd: bad arguments");
  const float sh = (float)Hi / (float)Ho, sw = (float)Wi / (float)Wo;
  const CatGrid g = cat_grid(B, Ho, Wo, C);
  hipStream_t s = (hipStream_t)stream;
  if (Hi == Ho && Wi == Wo)
    hipLaunchKernelGGL(cat_src_fwd_kernel<false>, dim3(g.gx, g.gy), dim3(256), 0, s, x, scale, shift, act, y, ldy, yoff,
                       stats, B, Hi, Wi, C / 4, Ho, Wo, sh, sw);
  else
    hipLaunchKernelGGL(cat_src_fwd_kernel<true>, dim3(g.gx, g.gy), dim3(256), 0, s, x, scale, shift, act, y, ldy, yoff,
                       stats, B, Hi, Wi, C / 4, Ho, Wo, sh, sw);
  NASSEG_LAUNCH_CHECK("cat_src_fwd");
  return NASSEG_OK;
}

// backward of nasseg_cat_src_fwd's input w.r.t. the slab-sized tensor: see cat_src_bwd_kernel.  du / slab:
// [B*Ho*Wo][ld], this input's channels at off; sscale / smean / sinvstd [ld], sums [2][ld]: the slab BatchNorm
// and its backward sums; z / tstats (mean | invstd | scale | shift, C each) / act: the pending producer (null:
// none), z of size (Hi, Wi) - when that is not the slab's size g is left UNMASKED for nasseg_bilinear_bwd_act and
// only the sums see the mask; g [B*Ho*Wo][C]; part: null or [nasseg_cat_src_blocks(B, Ho, Wo, C) + 64][2][C]
// (needs z)
int NASSEG_FN(cat_src_bwd)(const act_t* du, const act_t* slab, int64_t ld, int off, const float* sscale,
                           const float* smean, const float* sinvstd, const float* sums, int train, const act_t* z,
                           const float* tstats, int act, act_t* g, float* part, int B, int Ho, int Wo, int C, int Hi,
                           int Wi, void* stream) {
  NASSEG_REQUIRE(du && slab && sscale && smean && sinvstd && sums && g && nasseg_cat_src_blocks(B, Ho, Wo, C) > 0 &&
                     ld % 4 == 0 && off % 4 == 0 && off + C <= ld && (!z == !tstats) && (!part || z) && Hi > 0 &&
                     Wi > 0,
                 "cat_src_bwd: bad arguments");
  const CatGrid gr = cat_grid(B, Ho, Wo, C);
  const double M = (double)B * Ho * Wo;
  const float sh = (float)Hi / (float)Ho, sw = (float)Wi / (float)Wo;
#ifndef NASSEG_CAT_ZSAME  // (0: load the slab slice as rounds 3-4 did - A/B, tools/gpu.sh flags)
#define NASSEG_CAT_ZSAME 1
#endif
  if (NASSEG_CAT_ZSAME && Hi == Ho && Wi == Wo && z)  // (a pending input of the slab's size: its slice of the slab is rebuilt from z)
    hipLaunchKernelGGL((cat_src_bwd_kernel<false, true>), dim3(gr.gx, gr.gy), dim3(256), 0, (hipStream_t)stream, du, slab,
                       ld, off, sscale, smean, sinvstd, sums, (float)(1.0 / M), train, z, tstats, act, g, part, B * Ho,
                       Wo, C / 4, Ho, Hi, Wi, sh, sw);
  else if (Hi == Ho && Wi == Wo)
    hipLaunchKernelGGL(cat_src_bwd_kernel<false>, dim3(gr.gx, gr.gy), dim3(256), 0, (hipStream_t)stream, du, slab, ld,
                       off, sscale, smean, sinvstd, sums, (float)(1.0 / M), train, z, tstats, act, g, part, B * Ho, Wo,
                       C / 4, Ho, Hi, Wi, sh, sw);
  else
    hipLaunchKernelGGL(cat_src_bwd_kernel<true>, dim3(gr.gx, gr.gy), dim3(256), 0, (hipStream_t)stream, du, slab, ld,
                       off, sscale, smean, sinvstd, sums, (float)(1.0 / M), train, z, tstats, act, g, part, B * Ho, Wo,
                       C / 4, Ho, Hi, Wi, sh, sw);
  NASSEG_LAUNCH_CHECK("cat_src_bwd");
  return NASSEG_OK;
}

// ParamSum backward (psum_bwd_kernel): dy, za, zb, ga, gb dense [B*H*W][C]; tsa / tsb: null (operand is a finished
// map) or its pending BatchNorm's mean | invstd | scale | shift; ca / cb: the coefficients (null: 1); ga / gb: null
// when that operand needs no gradient; part_a / part_b: null or [nasseg_cat_src_blocks(B, H, W, C) + 64][2][C]
// (pending operands only); cpart: [blocks + 64][2][C] rows of {sum dy * ya, sum dy * yb}.
int NASSEG_FN(psum_bwd)(const act_t* dy, const act_t* za, const float* tsa, int act_a, const float* ca, act_t* ga,
                        float* part_a, const act_t* zb, const float* tsb, int act_b, const float* cb, act_t* gb,
                        float* part_b, float* cpart, int B, int H, int W, int C, void* stream) {
  NASSEG_REQUIRE(dy && za && zb && cpart && nasseg_cat_src_blocks(B, H, W, C) > 0 && (!part_a || tsa) &&
                     (!part_b || tsb),
                 "psum_bwd: bad arguments");
  const CatGrid gr = cat_grid(B, H, W, C);
  hipLaunchKernelGGL(psum_bwd_kernel, dim3(gr.gx, gr.gy), dim3(256), 0, (hipStream_t)stream, dy, za, tsa, act_a, ca, ga,
                     part_a, zb, tsb, act_b, cb, gb, part_b, cpart, B * H, W, C / 4);
  NASSEG_LAUNCH_CHECK("psum_bwd");
  return NASSEG_OK;
}

// out = g0 + g1 + ... (n of them, 1 <= n <= 8, added in that order; dense [B*H*W][C] like out) - and, with a pending
// BatchNorm + activation of the node (z, tstats: mean | invstd | scale | shift), out *= act'(scale*z + shift) with the
// rows part [nasseg_cat_src_blocks(B, H, W, C) + 64][2][C] of {sum out, sum out * xhat} (null: no rows).
int NASSEG_FN(grad_junction)(const act_t* g0, const act_t* g1, const act_t* g2, const act_t* g3, const act_t* g4,
                             const act_t* g5, const act_t* g6, const act_t* g7, int n, const act_t* z,
                             const float* tstats, int act, act_t* out, float* part, int B, int H, int W, int C,
                             void* stream) {
  JunctionSrc src = {{g0, g1, g2, g3, g4, g5, g6, g7}};
  NASSEG_REQUIRE(n >= 1 && n <= 8 && out && nasseg_cat_src_blocks(B, H, W, C) > 0 && (!z == !tstats) && (!part || z),
                 "grad_junction: bad arguments");
  for (int i = 0; i < n; ++i) NASSEG_REQUIRE(src.g[i], "grad_junction: gradient %d is null", i);
  const CatGrid gr = cat_grid(B, H, W, C);
  const dim3 grid(gr.gx, gr.gy), block(256);
  hipStream_t s = (hipStream_t)stream;
  switch (n) {
    case 1: hipLaunchKernelGGL(grad_junction_kernel<1>, grid, block, 0, s, src, z, tstats, act, out, part, B * H, W, C / 4); break;
    case 2: hipLaunchKernelGGL(grad_junction_kernel<2>, grid, block, 0, s, src, z, tstats, act, out, part, B * H, W, C / 4); break;
    case 3: hipLaunchKernelGGL(grad_junction_kernel<3>, grid, block, 0, s, src, z, tstats, act, out, part, B * H, W, C / 4); break;
    case 4: hipLaunchKernelGGL(grad_junction_kernel<4>, grid, block, 0, s, src, z, tstats, act, out, part, B * H, W, C / 4); break;
    case 5: hipLaunchKernelGGL(grad_junction_kernel<5>, grid, block, 0, s, src, z, tstats, act, out, part, B * H, W, C / 4); break;
    case 6: hipLaunchKernelGGL(grad_junction_kernel<6>, grid, block, 0, s, src, z, tstats, act, out, part, B * H, W, C / 4); break;
    case 7: hipLaunchKernelGGL(grad_junction_kernel<7>, grid, block, 0, s, src, z, tstats, act, out, part, B * H, W, C / 4); break;
    default: hipLaunchKernelGGL(grad_junction_kernel<8>, grid, block, 0, s, src, z, tstats, act, out, part, B * H, W, C / 4); break;
  }
  NASSEG_LAUNCH_CHECK("grad_junction");
  return NASSEG_OK;
}

static int64_t bilinear_bwd_ws(int B, int Hi, int Wi, int C, int Ho, int Wo) {
  if (C % 4 != 0 || Ho < 3 * Hi || Wo < 3 * Wi) return 0;
  return (int64_t)B * Ho * Wi * C;
}

#if NASSEG_FP32_ONLY
// floats of workspace nasseg_bilinear_bwd wants for this geometry (0: single-pass gather)
int64_t nasseg_bilinear_bwd_workspace(int B, int Hi, int Wi, int C, int Ho, int Wo) {
  return bilinear_bwd_ws(B, Hi, Wi, C, Ho, Wo);
}
#endif

// dx [B][Hi][Wi][C] = transpose of the forward map applied to dy[..., dyoff:dyoff+C].
// ws: nasseg_bilinear_bwd_workspace() floats, or null (then always the single-pass gather).
static int bilinear_bwd_launch(const act_t* dy, int64_t lddy, int dyoff, act_t* dx, int B, int Hi, int Wi, int C,
                               int Ho, int Wo, float* ws, const act_t* mz, const float* msc, const float* msh,
                               int mact, void* stream) {
  NASSEG_REQUIRE(B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0, "bilinear_bwd: bad shape");
  const float sh = (float)Hi / (float)Ho, sw = (float)Wi / (float)Wo;
  hipStream_t s = (hipStream_t)stream;
  const bool vec = C % 4 == 0 && lddy % 4 == 0 && dyoff % 4 == 0;
  NASSEG_REQUIRE(vec || !mz, "bilinear_bwd_act: channels must be multiples of 4");
  if (ws && vec && bilinear_bwd_ws(B, Hi, Wi, C, Ho, Wo) > 0) {
    // up-sampling by >= 3 in both directions: separable two-pass form
    hipLaunchKernelGGL((bilinear_bwd_axis_kernel<1, act_t, float>), dim3(rs_grid((int64_t)B * Ho * Wi * (C / 4))),
                       dim3(256), 0, s, dy, lddy, dyoff, ws, B, Ho, Wo, Wi, C / 4, 0, sw, nullptr, nullptr, nullptr, 0);
    hipLaunchKernelGGL((bilinear_bwd_axis_kernel<0, float, act_t>), dim3(rs_grid((int64_t)B * Hi * Wi * (C / 4))),
                       dim3(256), 0, s, ws, (int64_t)C, 0, dx, B, Ho, Wi, Wi, C / 4, Hi, sh, mz, msc, msh, mact);
    NASSEG_LAUNCH_CHECK("bilinear_bwd_axis");
    return NASSEG_OK;
  }
  // 1/scale = destinations per source step: <= 0.5 (down-sampling by >= 2) or <= 2 (up to 2x up-sampling) on both
  // axes - the tight-window gathers; anything else the general one
  const float ih = (float)Ho / (float)Hi, iw = (float)Wo / (float)Wi;
  if (vec && ih <= 0.5f && iw <= 0.5f)
    hipLaunchKernelGGL((bilinear_bwd_win_kernel<2>), dim3(rs_grid((int64_t)B * Hi * Wi * (C / 4))), dim3(256), 0, s,
                       dy, lddy, dyoff, dx, B, Hi, Wi, C / 4, Ho, Wo, sh, sw, mz, msc, msh, mact);
  else if (vec && ih <= 2.f && iw <= 2.f)
    hipLaunchKernelGGL((bilinear_bwd_win_kernel<4>), dim3(rs_grid((int64_t)B * Hi * Wi * (C / 4))), dim3(256), 0, s,
                       dy, lddy, dyoff, dx, B, Hi, Wi, C / 4, Ho, Wo, sh, sw, mz, msc, msh, mact);
  else if (vec)
    hipLaunchKernelGGL((bilinear_bwd_kernel<4>), dim3(rs_grid((int64_t)B * Hi * Wi * (C / 4))),
                       dim3(256), 0, s, dy, lddy, dyoff, dx, B, Hi, Wi, C / 4, Ho, Wo, sh, sw, mz, msc, msh, mact);
  else
    hipLaunchKernelGGL((bilinear_bwd_kernel<1>), dim3(rs_grid((int64_t)B * Hi * Wi * C)), dim3(256),
                       0, s, dy, lddy, dyoff, dx, B, Hi, Wi, C, Ho, Wo, sh, sw, nullptr, nullptr, nullptr, 0);
  NASSEG_LAUNCH_CHECK("bilinear_bwd");
  return NASSEG_OK;
}

int NASSEG_FN(bilinear_bwd)(const act_t* dy, int64_t lddy, int dyoff, act_t* dx, int B, int Hi, int Wi,
                        int C, int Ho, int Wo, float* ws, void* stream) {
  return bilinear_bwd_launch(dy, lddy, dyoff, dx, B, Hi, Wi, C, Ho, Wo, ws, nullptr, nullptr, nullptr, 0, stream);
}

// the same, the result multiplied by act'(scale*z + shift) of a pending activation at the SOURCE's size
// (z [B][Hi][Wi][C]; C %% 4 == 0): the gradient w.r.t. a resized Pending input of ConcatReduce, masked for its
// producer (whose BatchNorm-backward sums nasseg_cat_src_bwd has formed at the slab's size)
int NASSEG_FN(bilinear_bwd_act)(const act_t* dy, int64_t lddy, int dyoff, const act_t* z, const float* scale,
                                const float* shift, int act, act_t* dx, int B, int Hi, int Wi, int C, int Ho, int Wo,
                                float* ws, void* stream) {
  NASSEG_REQUIRE(z && scale && shift, "bilinear_bwd_act: null argument");
  return bilinear_bwd_launch(dy, lddy, dyoff, dx, B, Hi, Wi, C, Ho, Wo, ws, z, scale, shift, act, stream);
}

#if NASSEG_FP32_ONLY
// labels: elem_size 1 (uint8) or 8 (int64) -> int64 [B][Ho][Wo]
int nasseg_nearest_label(const void* x, int elem_size, int64_t* y, int B, int Hi, int Wi, int Ho,
                         int Wo, void* stream) {
  NASSEG_REQUIRE(B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, "nearest_label: bad shape");
  const float sh = (float)Hi / (float)Ho, sw = (float)Wi / (float)Wo;
  hipStream_t s = (hipStream_t)stream;
  const int grid = rs_grid((int64_t)B * Ho * Wo);
  if (elem_size == 1)
    hipLaunchKernelGGL((nearest_label_kernel<uint8_t>), dim3(grid), dim3(256), 0, s,
                       (const uint8_t*)x, y, B, Hi, Wi, Ho, Wo, sh, sw);
  else if (elem_size == 8)
    hipLaunchKernelGGL((nearest_label_kernel<int64_t>), dim3(grid), dim3(256), 0, s,
                       (const int64_t*)x, y, B, Hi, Wi, Ho, Wo, sh, sw);
  else
    return nasseg_fail(NASSEG_ERR_ARG, "nearest_label: elem_size %d not supported", elem_size);
  NASSEG_LAUNCH_CHECK("nearest_label");
  return NASSEG_OK;
}

#endif  // NASSEG_FP32_ONLY

}  // extern "C"
