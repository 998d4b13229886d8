// Gradient-norm clipping + optimiser steps of one training step as TWO launches.
//
// The reference ends every step with (src/engine/trainer.py:163-166,258-268)
//     clip_grad_norm_(encoder.parameters(), enc_grad_clip); clip_grad_norm_(decoder.parameters(), dec_grad_clip)
//     optim_enc.step(); optim_dec.step()
// on torch.optim.SGD(momentum, weight_decay) / torch.optim.Adam(weight_decay) objects (src/utils/solvers.py:6-52).
// torch's multi-tensor implementations need ~190 launches for that on a MobileNetV2 + searched decoder (741 when
// Adam is capturable, rocprofv3 of the replayed CVPR 321x321 step: 2.8 ms of 14.5) - elementwise work over 2-5 M
// floats.  Here:
//
//   optim_norm_kernel   one workgroup per CHUNK of a parameter tensor: sum of g^2 in fp64 -> partial[chunk];
//                       workgroup 0 also advances the per-tensor step counters (device memory: the step can be
//                       replayed from a hipGraph with plain, non-"capturable" optimisers).
//   optim_apply_kernel  every workgroup adds the partials of its clip set in a fixed order (a few hundred doubles
//                       out of L2), forms clip_coef = min(1, max_norm / (total_norm + 1e-6)) as clip_grad_norm_ does,
//                       and updates its chunk: g <- g * clip_coef (written back, as torch does in place), then the
//                       SGD or Adam arithmetic of torch.optim in the order torch applies it.
//
// The tensor table (pointers, sizes, clip set, hyper-parameter group) and the chunk list are DEVICE arrays the host
// side (engine/optim_native.py) builds once per parameter set and re-uploads only when a gradient moved.
#include "common.h"

#define NASSEG_OPTIM_CHUNK 4096
#define NASSEG_OPTIM_MAX_GROUPS 8

namespace {

// int64 columns of one row of the tensor table
enum { T_P = 0, T_G, T_S1, T_S2, T_NUMEL, T_CLIP, T_HYPER, T_FLAGS, T_COLS };

struct Hyper {
  int kind;  // 0 SGD, 1 Adam
  double lr, wd, a, b, eps;  // a: momentum | beta1, b: beta2
};

struct OptimArgs {
  const int64_t* tensors;  // [n_tensors][T_COLS]
  const int* chunks;       // [n_chunks][2] = {tensor, first element}; sorted by clip set
  int n_tensors, n_chunks;
  float* dstep;            // [n_tensors] steps taken so far (Adam's bias correction)
  double* partial;         // [n_chunks]
  float* norms;            // [n_clip] total_norm of each clip set (what clip_grad_norm_ returns)
  Hyper h[NASSEG_OPTIM_MAX_GROUPS];
  float max_norm[NASSEG_OPTIM_MAX_GROUPS];
  int clip_first[NASSEG_OPTIM_MAX_GROUPS], clip_count[NASSEG_OPTIM_MAX_GROUPS];
  int n_clip;
};

// sum over the 256 threads of a workgroup, same value in every thread, fixed order
__device__ __forceinline__ double block_sum(double v, double* red) {
  v = row16_allsum(v);
  const int tid = threadIdx.x;
  if ((tid & 15) == 0) red[tid >> 4] = v;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int i = 0; i < 16; ++i) t += red[i];
  __syncthreads();
  return t;
}

__global__ __launch_bounds__(256) void optim_norm_kernel(OptimArgs a) {
  __shared__ double red[16];
  const int tid = threadIdx.x;
  // (only rows that are stepped: a table row without a gradient this step - T_G == 0 - is not stepped by torch
  //  either, and its bias correction must not run ahead of the optimiser's own state["step"])
  if (blockIdx.x == 0)
    for (int i = tid; i < a.n_tensors; i += 256)
      if (a.tensors[(int64_t)i * T_COLS + T_G] != 0) a.dstep[i] += 1.f;
  if (a.n_chunks == 0) return;
  const int t = a.chunks[2 * blockIdx.x], off = a.chunks[2 * blockIdx.x + 1];
  const int64_t* row = a.tensors + (int64_t)t * T_COLS;
  if (row[T_CLIP] < 0) {  // (uniform per workgroup)
    if (tid == 0) a.partial[blockIdx.x] = 0.0;
    return;
  }
  const float* g = reinterpret_cast<const float*>(row[T_G]) + off;
  int n = (int)(row[T_NUMEL] - off);
  if (n > NASSEG_OPTIM_CHUNK) n = NASSEG_OPTIM_CHUNK;
  double s = 0.0;
  if (row[T_FLAGS] & 1) {
#pragma unroll
    for (int j = 0; j < NASSEG_OPTIM_CHUNK / 1024; ++j) {
      const int i = (j * 256 + tid) * 4;
      if (i + 3 < n) {
        const float4 v = ld4(g + i);
        s += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
      } else {
        for (int k = i; k < n; ++k) s += (double)g[k] * g[k];
      }
    }
  } else {
    for (int i = tid; i < n; i += 256) s += (double)g[i] * g[i];
  }
  s = block_sum(s, red);
  if (tid == 0) a.partial[blockIdx.x] = s;
}

struct Upd {
  int kind;
  float coef;        // clip coefficient (1 when the tensor is in no clip set)
  bool clip;         // write g * coef back
  float lr, wd, mom; // SGD
  float w1, b2, w2, bc2s, eps, step_size;  // Adam: 1 - beta1, beta2, 1 - beta2, sqrt(1 - beta2^t), eps, lr / (1 - beta1^t)
  bool has_buf;
};

// (where torch's own kernels evaluate a + alpha * b in one expression the compiler fuses it: the same fused forms
//  here make an SGD step bit-identical to torch's and an Adam step equal to it to the last bit or two)
__device__ __forceinline__ void update1(const Upd& u, float& p, float& g, float& s1, float& s2) {
  g = g * u.coef;
  float d = u.wd != 0.f ? __fadd_rn(g, __fmul_rn(u.wd, p)) : g;  // grad.add(param, alpha=weight_decay)
  if (u.kind == 0) {
    if (u.has_buf) {
      s1 = __fadd_rn(__fmul_rn(s1, u.mom), d);  // buf.mul_(momentum).add_(d_p)
      d = s1;
    }
    p = fmaf(-u.lr, d, p);  // param.add_(d_p, alpha=-lr)
  } else {
    s1 = fmaf(u.w1, __fsub_rn(d, s1), s1);                     // exp_avg.lerp_(grad, 1 - beta1)
    s2 = fmaf(__fmul_rn(u.w2, d), d, __fmul_rn(s2, u.b2));     // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(s2), u.bc2s), u.eps);
    p = fmaf(-u.step_size, __fdiv_rn(s1, denom), p);           // param.addcdiv_(exp_avg, denom, value=-step_size)
  }
}

__global__ __launch_bounds__(256) void optim_apply_kernel(OptimArgs a) {
  __shared__ double red[16];
  const int tid = threadIdx.x;
  const int t = a.chunks[2 * blockIdx.x], off = a.chunks[2 * blockIdx.x + 1];
  const int64_t* row = a.tensors + (int64_t)t * T_COLS;
  const int c = (int)row[T_CLIP];
  Upd u;
  u.coef = 1.f;
  u.clip = c >= 0;
  if (c >= 0) {
    const double* part = a.partial + a.clip_first[c];
    const int np = a.clip_count[c];
    double s = 0.0;
    for (int i = tid; i < np; i += 256) s += part[i];
    s = block_sum(s, red);
    const float tn = (float)sqrt(s);
    float coef = a.max_norm[c] / (tn + 1e-6f);
    u.coef = coef > 1.f ? 1.f : coef;  // (a NaN norm stays NaN, as torch.clamp keeps it)
    if ((int)blockIdx.x == a.clip_first[c] && tid == 0) a.norms[c] = tn;
  }
  const Hyper& h = a.h[row[T_HYPER]];
  u.kind = h.kind;
  u.lr = (float)h.lr;
  u.wd = (float)h.wd;
  u.mom = (float)h.a;
  u.has_buf = row[T_S1] != 0;
  u.w1 = u.b2 = u.w2 = u.bc2s = u.eps = u.step_size = 0.f;
  if (h.kind == 1) {
    const double step = (double)a.dstep[t];  // (already advanced by optim_norm_kernel)
    const double bc1 = 1.0 - pow(h.a, step), bc2 = 1.0 - pow(h.b, step);
    u.w1 = (float)(1.0 - h.a);
    u.b2 = (float)h.b;
    u.w2 = (float)(1.0 - h.b);
    u.bc2s = (float)sqrt(bc2);
    u.eps = (float)h.eps;
    u.step_size = (float)(h.lr / bc1);
  }
  float* p = reinterpret_cast<float*>(row[T_P]) + off;
  float* g = reinterpret_cast<float*>(row[T_G]) + off;
  float* s1 = u.has_buf ? reinterpret_cast<float*>(row[T_S1]) + off : nullptr;
  float* s2 = h.kind == 1 ? reinterpret_cast<float*>(row[T_S2]) + off : nullptr;
  int n = (int)(row[T_NUMEL] - off);
  if (n > NASSEG_OPTIM_CHUNK) n = NASSEG_OPTIM_CHUNK;
  if (row[T_FLAGS] & 1) {
#pragma unroll
    for (int j = 0; j < NASSEG_OPTIM_CHUNK / 1024; ++j) {
      const int i = (j * 256 + tid) * 4;
      if (i + 3 < n) {
        float4 pv = ld4(p + i), gv = ld4(g + i);
        float4 av = s1 ? ld4(s1 + i) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 bv = s2 ? ld4(s2 + i) : make_float4(0.f, 0.f, 0.f, 0.f);
        update1(u, pv.x, gv.x, av.x, bv.x);
        update1(u, pv.y, gv.y, av.y, bv.y);
        update1(u, pv.z, gv.z, av.z, bv.z);
        update1(u, pv.w, gv.w, av.w, bv.w);
        st4(p + i, pv);
        if (u.clip) st4(g + i, gv);
        if (s1) st4(s1 + i, av);
        if (s2) st4(s2 + i, bv);
      } else {
        for (int k = i; k < n; ++k) {
          float pv = p[k], gv = g[k], av = s1 ? s1[k] : 0.f, bv = s2 ? s2[k] : 0.f;
          update1(u, pv, gv, av, bv);
          p[k] = pv;
          if (u.clip) g[k] = gv;
          if (s1) s1[k] = av;
          if (s2) s2[k] = bv;
        }
      }
    }
  } else {
    for (int k = tid; k < n; k += 256) {
      float pv = p[k], gv = g[k], av = s1 ? s1[k] : 0.f, bv = s2 ? s2[k] : 0.f;
      update1(u, pv, gv, av, bv);
      p[k] = pv;
      if (u.clip) g[k] = gv;
      if (s1) s1[k] = av;
      if (s2) s2[k] = bv;
    }
  }
}

}  // namespace

extern "C" {

int64_t nasseg_optim_chunk(void) { return NASSEG_OPTIM_CHUNK; }

// One clip + optimiser step over every tensor of the table (see include/nasseg.h for the table formats).
int nasseg_optim_step(const int64_t* tensors, int n_tensors, const int* chunks, int n_chunks, const double* hyper,
                      int n_hyper, const double* clips, int n_clip, float* dstep, double* partial, float* norms,
                      void* stream) {
  NASSEG_REQUIRE(n_tensors >= 0 && n_chunks >= 0, "optim_step: bad counts");
  if (n_tensors == 0) return NASSEG_OK;
  NASSEG_REQUIRE(tensors && chunks && hyper && dstep && partial, "optim_step: null table");
  NASSEG_REQUIRE(n_chunks > 0, "optim_step: tensors without chunks");
  NASSEG_REQUIRE(n_hyper > 0 && n_hyper <= NASSEG_OPTIM_MAX_GROUPS, "optim_step: 1..%d hyper-parameter groups",
                 NASSEG_OPTIM_MAX_GROUPS);
  NASSEG_REQUIRE(n_clip >= 0 && n_clip <= NASSEG_OPTIM_MAX_GROUPS && (n_clip == 0 || (clips && norms)),
                 "optim_step: 0..%d clip sets", NASSEG_OPTIM_MAX_GROUPS);
  OptimArgs a = {};
  a.tensors = tensors;
  a.chunks = chunks;
  a.n_tensors = n_tensors;
  a.n_chunks = n_chunks;
  a.dstep = dstep;
  a.partial = partial;
  a.norms = norms;
  for (int i = 0; i < n_hyper; ++i) {
    const double* h = hyper + 6 * i;
    NASSEG_REQUIRE(h[0] == 0.0 || h[0] == 1.0, "optim_step: hyper group %d: kind must be 0 (SGD) or 1 (Adam)", i);
    a.h[i].kind = (int)h[0];
    a.h[i].lr = h[1];
    a.h[i].wd = h[2];
    a.h[i].a = h[3];
    a.h[i].b = h[4];
    a.h[i].eps = h[5];
  }
  a.n_clip = n_clip;
  for (int i = 0; i < n_clip; ++i) {
    const double* c = clips + 3 * i;
    a.max_norm[i] = (float)c[0];
    a.clip_first[i] = (int)c[1];
    a.clip_count[i] = (int)c[2];
    NASSEG_REQUIRE(a.clip_first[i] >= 0 && a.clip_count[i] > 0 && a.clip_first[i] + a.clip_count[i] <= n_chunks,
                   "optim_step: clip set %d outside the chunk list", i);
  }
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(optim_norm_kernel, dim3(n_clip > 0 ? n_chunks : 1), dim3(256), 0, s, a);
  NASSEG_LAUNCH_CHECK("optim_norm");
  hipLaunchKernelGGL(optim_apply_kernel, dim3(n_chunks), dim3(256), 0, s, a);
  NASSEG_LAUNCH_CHECK("optim_apply");
  return NASSEG_OK;
}

}  // extern "C"
