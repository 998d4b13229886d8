// Pieces shared by the depthwise kernels (dwconv.hip) and the fused depthwise -> pointwise
// SepConv stage (sepconv.hip): masked loads, the input prologue and the geometry of a
// "vertical strip" (see dwconv.hip's header).
#pragma once
#include <math.h>

#include "common.h"

namespace {

// keep v where ok, +0.0 elsewhere, without a select the compiler could turn back into
// a branch around the producing load (mask = all ones / all zeros)
__device__ __forceinline__ float4 keep_if(float4 v, bool ok) {
  const unsigned m = 0u - (unsigned)ok;
  v.x = __uint_as_float(__float_as_uint(v.x) & m);
  v.y = __uint_as_float(__float_as_uint(v.y) & m);
  v.z = __uint_as_float(__float_as_uint(v.z) & m);
  v.w = __uint_as_float(__float_as_uint(v.w) & m);
  return v;
}

// Pin a value at this program point: LLVM otherwise sinks the whole FMA chain of an
// accumulator into the (conditional) block that finally stores it, which keeps every
// loaded operand alive until the end of the kernel.
__device__ __forceinline__ void pin(float4& v) {
  asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
}

__device__ __forceinline__ float4 relu4(float4 v) {
  return make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
}

// input prologue of one channel group: clamp(v*sc + sh, lo, hi); (lo, hi) encode the
// activation (none: -inf..inf, ReLU: 0..inf, ReLU6: 0..6) so there is no branch per tap
struct Prologue {
  float4 sc, sh;
  float lo, hi;
};
__device__ __forceinline__ Prologue make_prologue(const float* scale, const float* shift, int act,
                                                  int c4) {
  Prologue p;
  p.sc = scale ? lda4(scale + c4 * 4) : make_float4(1.f, 1.f, 1.f, 1.f);
  p.sh = shift ? lda4(shift + c4 * 4) : f4zero();
  p.lo = act ? 0.f : -INFINITY;
  p.hi = act == NASSEG_ACT_RELU6 ? 6.f : INFINITY;
  return p;
}
__device__ __forceinline__ float4 apply_prologue(float4 v, const Prologue& p) {
  v = fma4(v, p.sc, p.sh);
  v.x = fminf(fmaxf(v.x, p.lo), p.hi);
  v.y = fminf(fmaxf(v.y, p.lo), p.hi);
  v.z = fminf(fmaxf(v.z, p.lo), p.hi);
  v.w = fminf(fmaxf(v.w, p.lo), p.hi);
  return v;
}

struct StripCfg {
  int g, e;
};
inline StripCfg strip_cfg(int stride, int dil) {
  // outputs of one strip are g rows apart; consecutive outputs are e dilated
  // input-row steps apart (see dwconv.hip's header)
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
