// Shared pieces of the dense-convolution kernels (conv_fwd.hip, conv_wgrad.hip):
// implicit-GEMM geometry and the fp32 MFMA wrapper.
//
// Code-generation rules learned on gfx950 / hipcc 7.2 and followed throughout:
//  * never put a global load under a data-dependent branch: the compiler closes every
//    such block with s_waitcnt vmcnt(0), which serialises the loop on HBM latency.
//    Load from a clamped (always valid) address and zero the value with a bit mask.
//  * run-time mode flags inside hot loops become branches; make them template
//    parameters or hoist them out of the loop.
#pragma once
#include "common.h"

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

struct ConvGeom {
  int B, Hs, Ws;  // source (tensor being read) dims
  int Ho, Wo;     // destination dims
  int kh, kw, stride, pad, dil, transposed;
};

// source pixel index for destination pixel (b,oy,ox) and tap (ty,tx);
// -1 when the tap falls outside / on a stride hole.
__device__ __forceinline__ int src_pixel(const ConvGeom& g, int b, int oy, int ox, int ty, int tx) {
  int iy, ix;
  if (!g.transposed) {
    iy = oy * g.stride - g.pad + ty * g.dil;
    ix = ox * g.stride - g.pad + tx * g.dil;
    if (iy < 0 || iy >= g.Hs || ix < 0 || ix >= g.Ws) return -1;
  } else {
    const int ny = oy + g.pad - ty * g.dil;
    const int nx = ox + g.pad - tx * g.dil;
    if (ny < 0 || nx < 0) return -1;
    if (g.stride > 1) {
      if ((ny % g.stride) || (nx % g.stride)) return -1;
      iy = ny / g.stride;
      ix = nx / g.stride;
    } else {
      iy = ny;
      ix = nx;
    }
    if (iy >= g.Hs || ix >= g.Ws) return -1;
  }
  return (b * g.Hs + iy) * g.Ws + ix;
}

__device__ __forceinline__ float keep_if(float v, bool ok) {
  return __uint_as_float(__float_as_uint(v) & (0u - (unsigned)ok));
}
__device__ __forceinline__ float4 keep_if(float4 v, bool ok) {
  const unsigned m = 0u - (unsigned)ok;
  v.x = __uint_as_float(__float_as_uint(v.x) & m);
  v.y = __uint_as_float(__float_as_uint(v.y) & m);
  v.z = __uint_as_float(__float_as_uint(v.z) & m);
  v.w = __uint_as_float(__float_as_uint(v.w) & m);
  return v;
}
