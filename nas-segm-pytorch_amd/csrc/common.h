// Shared device/host helpers for the nasseg gfx950 kernels.
// All activations are NHWC ("channels_last"), fp32 or bf16 storage (act_t below): element
// (b, y, x, c) of a tensor with pixel stride ld lives at ((b*H + y)*W + x)*ld + c.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define NASSEG_OK 0
#define NASSEG_ERR_ARG (-1)
#define NASSEG_ERR_LAUNCH (-2)
#define NASSEG_ERR_UNSUPPORTED (-3)

// activation codes shared by every fused prologue / epilogue
#define NASSEG_ACT_NONE 0
#define NASSEG_ACT_RELU 1
#define NASSEG_ACT_RELU6 2

int nasseg_fail(int code, const char* fmt, ...);

#define NASSEG_REQUIRE(cond, ...)                                  \
  do {                                                             \
    if (!(cond)) return nasseg_fail(NASSEG_ERR_ARG, __VA_ARGS__);  \
  } while (0)

#define NASSEG_LAUNCH_CHECK(name)                                                  \
  do {                                                                             \
    hipError_t e_ = hipGetLastError();                                             \
    if (e_ != hipSuccess)                                                          \
      return nasseg_fail(NASSEG_ERR_LAUNCH, "%s: %s", name, hipGetErrorString(e_)); \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

__device__ __forceinline__ float act_apply(float v, int act) {
  if (act == NASSEG_ACT_RELU) return fmaxf(v, 0.f);
  if (act == NASSEG_ACT_RELU6) return fminf(fmaxf(v, 0.f), 6.f);
  return v;
}
__device__ __forceinline__ float4 act_apply4(float4 v, int act) {
  v.x = act_apply(v.x, act);
  v.y = act_apply(v.y, act);
  v.z = act_apply(v.z, act);
  v.w = act_apply(v.w, act);
  return v;
}
// Activation of a fused prologue inside a hot loop: the run-time `act` becomes two loop-invariant
// scalars (is there a clamp, and its upper bound) and a select, instead of a branch per element.
struct ActSel {
  bool on;
  float hi;
};
__device__ __forceinline__ ActSel act_sel(int act) {
  ActSel s;
  s.on = act != NASSEG_ACT_NONE;
  s.hi = act == NASSEG_ACT_RELU6 ? 6.f : __builtin_inff();
  return s;
}
__device__ __forceinline__ float act_apply(float v, const ActSel& s) {
  const float r = fminf(fmaxf(v, 0.f), s.hi);
  return s.on ? r : v;
}
__device__ __forceinline__ float4 act_apply4(float4 v, const ActSel& s) {
  v.x = act_apply(v.x, s);
  v.y = act_apply(v.y, s);
  v.z = act_apply(v.z, s);
  v.w = act_apply(v.w, s);
  return v;
}
// derivative mask of the activation evaluated at pre-activation value z
__device__ __forceinline__ float act_mask(float z, int act) {
  if (act == NASSEG_ACT_RELU) return z > 0.f ? 1.f : 0.f;
  if (act == NASSEG_ACT_RELU6) return (z > 0.f && z < 6.f) ? 1.f : 0.f;
  return 1.f;
}
// ... the same with the activation as loop-invariant scalars (no branch per element)
__device__ __forceinline__ float act_mask(float z, const ActSel& s) {
  const bool inside = z > 0.f && z < s.hi;
  return (s.on && !inside) ? 0.f : 1.f;
}
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// ---------------------------------------------------------------------------
// Storage type of ACTIVATIONS (feature maps and their gradients).  Every source file is
// compiled twice: act_t = float exports nasseg_<op>, act_t = bfloat16 (-DNASSEG_BF16) exports
// the twin nasseg_bf16_<op> with identical arguments.  Only storage changes: values are
// widened to fp32 on load and rounded to nearest-even on store; all arithmetic, the MFMA
// accumulation, statistics, parameters, parameter gradients and workspaces stay fp32.
// bf16_t is a struct so that an activation can never be read as a number by accident.
// ---------------------------------------------------------------------------
struct bf16_t {
  uint16_t v;
};
#ifdef NASSEG_BF16
typedef bf16_t act_t;
#define NASSEG_FN(name) nasseg_bf16_##name
#define NASSEG_FP32_ONLY 0
#else
typedef float act_t;
#define NASSEG_FN(name) nasseg_##name
#define NASSEG_FP32_ONLY 1  // entry points that never touch activations exist once, in this build

#endif

// Sum over the 16 lanes of a DPP row (lanes 16*g .. 16*g+15), result in every lane of the row: the same
// pairing as an xor butterfly over offsets 1, 2, 4, 8 (bit-identical), but four full-rate v_add_f32_dpp
// instead of four ds_bpermute round trips through the LDS crossbar.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float row16_allsum(float v) {
  v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);  // row_half_mirror
  v += dpp_mov<0x140>(v);  // row_mirror
  return v;
}

template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row16_allsum(double v) {  // (same pairing as the float form)
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x141>(v);
  v += dpp_mov<0x140>(v);
  return v;
}

__device__ __forceinline__ float bf16_to_f32(uint32_t h) { return __uint_as_float(h << 16); }
__device__ __forceinline__ uint32_t f32_to_bf16(float f) {
  const uint32_t u = __float_as_uint(f);
  const uint32_t r = (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;  // round to nearest even
  return (f != f) ? 0x7fc0u : r;                                // (NaN stays NaN)
}
__device__ __forceinline__ float lda1(const float* p) { return *p; }
__device__ __forceinline__ float lda1(const bf16_t* p) { return bf16_to_f32(p->v); }
__device__ __forceinline__ void sta1(float* p, float v) { *p = v; }
__device__ __forceinline__ void sta1(bf16_t* p, float v) { p->v = (uint16_t)f32_to_bf16(v); }
__device__ __forceinline__ float4 lda4(const float* p) { return ld4(p); }
__device__ __forceinline__ float4 lda4(const bf16_t* p) {
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
                     __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}
__device__ __forceinline__ void sta4(float* p, float4 v) { st4(p, v); }
__device__ __forceinline__ void sta4(bf16_t* p, float4 v) {
  uint2 u;
  u.x = f32_to_bf16(v.x) | (f32_to_bf16(v.y) << 16);
  u.y = f32_to_bf16(v.z) | (f32_to_bf16(v.w) << 16);
  *reinterpret_cast<uint2*>(p) = u;
}
__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
typedef float f32x2 __attribute__((ext_vector_type(2)));
// two v_pk_fma_f32 (two fp32 fused multiply-adds per lane and instruction) instead of four v_fma_f32: the depthwise
// and elementwise kernels issue a third fewer vector instructions (dilated 5x5 depthwise: 3440 -> 2000 per wave,
// 99.8 -> 91.9 us; every other kernel within 1 %).  Same fused operation per element: results are bit-identical.
__device__ __forceinline__ float4 fma4(float4 a, float4 b, float4 c) {
  const f32x2 lo = __builtin_elementwise_fma((f32x2){a.x, a.y}, (f32x2){b.x, b.y}, (f32x2){c.x, c.y});
  const f32x2 hi = __builtin_elementwise_fma((f32x2){a.z, a.w}, (f32x2){b.z, b.w}, (f32x2){c.z, c.w});
  return make_float4(lo.x, lo.y, hi.x, hi.y);
}
__device__ __forceinline__ float4 add4(float4 a, float4 b) {
  return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}
__device__ __forceinline__ float4 mul4(float4 a, float4 b) {
  return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w);
}

// ---------------------------------------------------------------------------
// Second stage of every deterministic two-stage reduction: sum partial[b][e]
// over b = 0..nblk-1 for NASSEG_RP_ELEMS consecutive elements per workgroup.
// 256 threads = 8 elements x 32 slices of the partial index; each thread adds
// its slice in a fixed order (fp64, 8 loads in flight), the 32 slices are then
// combined in a fixed order through LDS.  The total is returned in the threads
// with slice == 0 (rp_slice() == 0); `valid` == (e < per).
// ---------------------------------------------------------------------------
#define NASSEG_RP_ELEMS 8
#define NASSEG_RP_SLICES 32
__device__ __forceinline__ int rp_elem() { return threadIdx.x % NASSEG_RP_ELEMS; }
__device__ __forceinline__ int rp_slice() { return threadIdx.x / NASSEG_RP_ELEMS; }
// SLICES = 32: 256 threads (the default); SLICES = 128: 1024 threads, for 512 < nblk <= 4096 rows in ONE launch
// (1024 rows per round of 8 loads in flight) instead of a two-level finalisation in two.
template <int SLICES>
__device__ __forceinline__ double reduce_partials_n(const float* __restrict__ partial, int nblk,
                                                    int64_t per, int64_t e, bool valid,
                                                    double (*red)[NASSEG_RP_ELEMS + 1]) {
  const int slice = rp_slice();
  const int el = rp_elem();
  double s[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) s[i] = 0.0;
  if (valid) {
    int b = slice;
    for (; b + 7 * SLICES < nblk; b += 8 * SLICES) {
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = partial[(int64_t)(b + i * SLICES) * per + e];
#pragma unroll
      for (int i = 0; i < 8; ++i) s[i] += (double)v[i];
    }
    for (; b < nblk; b += SLICES) s[0] += (double)partial[(int64_t)b * per + e];
  }
  red[slice][el] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  __syncthreads();
  if (SLICES > 32) {
    // fixed-order tree: 32 threads per element add SLICES / 32 slices each, then slice 0 adds the 32
    double part = 0.0;
    if (slice < 32) {
#pragma unroll
      for (int i = 0; i < SLICES / 32; ++i) part += red[slice * (SLICES / 32) + i][el];
    }
    __syncthreads();
    if (slice < 32) red[slice][el] = part;
    __syncthreads();
  }
  double tot = 0.0;
  if (slice == 0) {
#pragma unroll
    for (int i = 0; i < 32; ++i) tot += red[i][el];
  }
  __syncthreads();
  return tot;
}
__device__ __forceinline__ double reduce_partials16(const float* __restrict__ partial, int nblk,
                                                    int64_t per, int64_t e, bool valid,
                                                    double (*red)[NASSEG_RP_ELEMS + 1]) {
  return reduce_partials_n<NASSEG_RP_SLICES>(partial, nblk, per, e, valid, red);
}
#define NASSEG_RP_WIDE_SLICES 128  // (1024 threads)
#define NASSEG_RP_WIDE_MAX_ROWS 4096
