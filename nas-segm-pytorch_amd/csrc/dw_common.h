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
  // v_med3_f32 = the clamp in one instruction (a NaN comes out as lo, as fminf(fmaxf(v, lo), hi) gives)
  v.x = __builtin_amdgcn_fmed3f(v.x, p.lo, p.hi);
  v.y = __builtin_amdgcn_fmed3f(v.y, p.lo, p.hi);
  v.z = __builtin_amdgcn_fmed3f(v.z, p.lo, p.hi);
  v.w = __builtin_amdgcn_fmed3f(v.w, p.lo, p.hi);
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

// Sum acc[t] (t < NV) over all threads of the workgroup that share a channel group and
// write out[t][C], RB values per LDS round (NV % RB == 0).  Thread `tid` of a workgroup whose
// first flattened (x, c/4) index is `base` owns channel group (base + tid) % C4.  With
// C4 > 256 only the groups present in the workgroup are written (caller zero-fills).
// Lanes l, l+C4, l+2*C4, ... of a wave hold the same group: lanes < C4 gather them with
// shuffles in a fixed order; the four waves then meet in LDS indexed by channel group.
// ld: distance between out[t] and out[t + 1] (0 = C: the dense [NV][C] row; wider when the C channels are a
// slice of a row over more channels - a concatenation slab's statistics).
template <int NV, int RB>
__device__ __forceinline__ void block_reduce_groups(float4 (&acc)[NV], float4 (*red)[4][64],
                                                    float* __restrict__ out, int base, int C4, int ld = 0) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int C = ld ? ld : C4 * 4;
  const int nown = C4 < 64 ? C4 : 64;
  if (C4 < 64) {
#pragma unroll
    for (int t = 0; t < NV; ++t) {
      const float4 v = acc[t];
      float4 s = v;
      for (int off = C4; off < 64; off += C4) {
        const int src = lane + off;
        const bool ok = src < 64;
        const int sl = ok ? src : lane;
        const float ax = __shfl(v.x, sl), ay = __shfl(v.y, sl), az = __shfl(v.z, sl),
                    aw = __shfl(v.w, sl);
        if (ok) {
          s.x += ax;
          s.y += ay;
          s.z += az;
          s.w += aw;
        }
      }
      acc[t] = s;
    }
  }
  const int cc = (base + tid) % C4;  // channel group of this lane
  if (C4 <= 64) {
#pragma unroll
    for (int r = 0; r < NV / RB; ++r) {
      __syncthreads();
      if (lane < nown) {
#pragma unroll
        for (int u = 0; u < RB; ++u) red[u][wave][cc] = acc[r * RB + u];
      }
      __syncthreads();
      if (tid < C4) {
#pragma unroll
        for (int u = 0; u < RB; ++u) {
          const float4 s = add4(add4(red[u][0][tid], red[u][1][tid]),
                                add4(red[u][2][tid], red[u][3][tid]));
          sta4(out + (size_t)(r * RB + u) * C + tid * 4, s);
        }
      }
    }
  } else {
    // C4 > 64: at most ceil(256 / C4) <= 3 threads of the workgroup share a group, so a
    // direct strided sum by the first min(C4, 256) threads has no serial tail to speak of
    float4* flat = &red[0][0][0];  // needs RB * 256 >= 256 float4
    const int nsum = C4 < 256 ? C4 : 256;
#pragma unroll
    for (int t = 0; t < NV; ++t) {
      __syncthreads();
      flat[tid] = acc[t];
      __syncthreads();
      if (tid < nsum) {
        float4 s = f4zero();
        for (int u = tid; u < 256; u += C4) s = add4(s, flat[u]);
        sta4(out + (size_t)t * C + cc * 4, s);
      }
    }
  }
}


}  // namespace
