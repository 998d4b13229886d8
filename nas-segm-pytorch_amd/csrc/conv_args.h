// Argument block of the dense-convolution kernels (conv_fwd.hip, conv_pwn.hip) and the hand-over between
// the two translation units.  Compiled twice like every kernel source (act_t = float / bfloat16): internal,
// non-exported functions carry the build in their name (NASSEG_INTERNAL).
#pragma once
#include "conv_common.h"

#ifdef NASSEG_BF16
#define NASSEG_INTERNAL(name) nasseg_internal_bf16_##name
#else
#define NASSEG_INTERNAL(name) nasseg_internal_##name
#endif
#define NASSEG_HIDDEN __attribute__((visibility("hidden")))

struct FwdArgs {
  const act_t* x;  // activations: fp32 or bf16 storage (common.h)
  int ldx;
  const float* w;
  act_t* y;
  int ldy;
  const float* in_scale;
  const float* in_shift;
  int in_act;
  const float* out_scale;
  const float* out_shift;
  int out_act;
  const act_t* res;
  int ldres;
  float* stats;  // optional [gridDim.x][2][N]: per-workgroup partial sums (see STATS)
  // STATS == 2: the BatchNorm whose backward statistics are gathered (y is the gradient
  // w.r.t. act(b_scale*bz + b_shift))
  const act_t* bz;
  int ldbz;
  const float* b_scale;
  const float* b_shift;
  const float* b_mean;
  const float* b_invstd;
  int b_act;
  int K, N;
  ConvGeom g;
  int th, tw;  // conv3x3_lds_kernel: the workgroup's output tile, rows x columns (lds3x3_tile)
};


// conv_pwn.hip: the N-split persistent pointwise kernel.  plan: is this (pixels, N, K, call kind) served, and
// with how many workgroups (= statistics rows); launch: stats as conv_dispatch's stats_mode (0..3).
struct PwnPlan {
  int ok, ps, mtw, ntw, grid;
  size_t lds;
};
NASSEG_HIDDEN PwnPlan nasseg_internal_pwn_plan(int64_t M, int N, int K, int mode);
NASSEG_HIDDEN int NASSEG_INTERNAL(pwn_launch)(const FwdArgs& a, const PwnPlan& p, int stats, hipStream_t s);
