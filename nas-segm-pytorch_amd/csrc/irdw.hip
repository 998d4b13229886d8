// InvertedResidual's expansion never stored: the 3x3 depthwise kernels rebuild their input on the matrix cores.
//
// Reference: MobileNetV2's InvertedResidual (src/nn/layer_factory.py:125-158): 1x1 conv K -> C = 6 K + BatchNorm +
// ReLU6, 3x3 depthwise (stride 1 or 2) + BatchNorm + ReLU6, 1x1 conv C -> N + BatchNorm.  The expanded map z1 has six
// times the channels of the block's input: at 4x3x1024x2048 the 96-channel map at 512x1024 alone is 805 MB, written
// by the expansion, read by the depthwise forward and read again by the depthwise backward (the pointwise backward
// already rebuilds it, conv_pwbwd.hip) - 2.4 GB of the step's 50 GB for the first block, ~4.8 GB for the encoder.
// But z1 = W1 x is K / 4 MFMA steps per 16 x 16 (channel x pixel) tile from an input that is six times smaller:
//   * statistics of z1 come from a pass of the pointwise kernel that stores nothing (nasseg_conv_fwd with y == NULL);
//   * nasseg_irdw_fwd:  depthwise forward over act(BN1(W1 x)), z1 rebuilt row by row;
//   * nasseg_irdw_bwd:  the one-kernel depthwise backward (dwconv.hip: dw3x3_bwd_bn_kernel - BatchNorm backward of
//     the depthwise conv's BatchNorm on load, weight gradient, input gradient masked with ReLU6' of the expansion's
//     BatchNorm, that BatchNorm's backward sums) with z1 rebuilt instead of read.
//
// Mapping.  A wave owns ONE 16-channel tile and a strip of 16 pixel columns and walks rows.  v_mfma_f32_16x16x4_f32
// with A = W1 (lane (j, kg): row n = 16 t + j, k = 16 kb + 4 kg + c) and B = x (lane (j, kg): pixel column j,
// k = 16 kb + 4 kg + c) leaves lane (j, kg) with channels 16 t + 4 kg .. + 3 of pixel column j: the float4 a
// depthwise thread works on - z1 lands in the registers of the lane that consumes it, nothing goes through LDS.
// The k-blocks ascend and the four components of a lane's float4 go in turn: the accumulation order of the forward
// kernels (conv_pwn.hip / conv_fwd.hip) - the rebuilt z1 has the bits the statistics pass saw.
// Forward needs the 3x3 neighbourhood of act(BN1(z1)): rows come from the walk (a window of three), columns j - 1
// and j + 1 from the neighbouring lanes of the 16-lane DPP row (row_shr:1 / row_shl:1); lanes 0 and 15 of a strip
// are halo (stride 1: 14 output columns per strip; stride 2: lane j holds input column 14 s - 1 + j, odd lanes up
// to 13 produce the 7 output columns).  Backward needs z1 only AT the pixel it differentiates; its 3x3 window is over
// the gradient dz2, of which a lane loads and differentiates its own column and takes the others from its neighbours
// the same way (stride 1: 14 input columns per strip; stride 2: 15 quad columns, lane 15 supplies lane 14's right).
// A workgroup is three or four waves - neighbouring channel tiles of one (image, row chunk, strip) item, reading the
// same x rows (L1); blockIdx.y picks the group of tiles - and is persistent: it takes items with the stride of
// gridDim.x and leaves its channels of ONE row of partial sums (more waves per workgroup would cap the registers of
// the backward kernel below what it needs: 168 VGPRs and 143 spilled at twelve waves).
#include "dw_common.h"

extern "C" int nasseg_wgrad_finalize_many(int count, const float* const* partial, float* const* dw, const int* dims,
                                           void* stream);
extern "C" int nasseg_rows_sum(float* partial, int nblk, int cols, float* out, void* stream);

namespace {

#ifndef NASSEG_IR_BWD_MINB
#define NASSEG_IR_BWD_MINB 1
#endif
#ifndef NASSEG_IR_FWD_MINB
#define NASSEG_IR_FWD_MINB 1
#endif
constexpr int kIrMaxTiles = 12;  // C <= 192
constexpr int kIrMaxWaves = 4;   // waves (channel tiles) per workgroup

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {  // (conv_common.h's wrapper)
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

template <int CTRL>
__device__ __forceinline__ float dpp_shift0(float v) {  // (bound_ctrl: a lane without a source reads 0 - no "old" move)
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float4 from_left(float4 v) {  // lane j <- lane j - 1 of its 16-lane row (0 at j = 0)
  return make_float4(dpp_shift0<0x111>(v.x), dpp_shift0<0x111>(v.y), dpp_shift0<0x111>(v.z), dpp_shift0<0x111>(v.w));
}
__device__ __forceinline__ float4 from_right(float4 v) {  // lane j <- lane j + 1 (0 at j = 15)
  return make_float4(dpp_shift0<0x101>(v.x), dpp_shift0<0x101>(v.y), dpp_shift0<0x101>(v.z), dpp_shift0<0x101>(v.w));
}
__device__ __forceinline__ float4 round_like_storage(float4 v) {
#ifdef NASSEG_BF16
  // (the value the expansion would have stored and every consumer of z1 would have read)
  v = make_float4(bf16_to_f32(f32_to_bf16(v.x)), bf16_to_f32(f32_to_bf16(v.y)), bf16_to_f32(f32_to_bf16(v.z)),
                  bf16_to_f32(f32_to_bf16(v.w)));
#endif
  return v;
}

struct IrCommon {
  const act_t* x;         // [B][H][W][K]: the block's input (raw output of the BatchNorm in front, or a finished map)
  const float* w1;        // [C][K]: the expansion's weight as PyTorch stores it
  const float* in_scale;  // prologue of x (null: none)
  const float* in_shift;
  int in_act;
  const float* sc1;       // BatchNorm of the expansion: scale, shift (+ act1)
  const float* sh1;
  int act1;
  int B, H, W, K, C, Ho, Wo;
  int strips, chunks, rows_per_chunk, items;
};

// what a lane needs to rebuild z1 for its (tile, k-quads): W1 operands, offsets and prologue vectors of x
template <int KT, bool PRO>
struct IrLane {
  float aw[KT][4];
  int koff[KT];
  float4 psc[KT], psh[KT];
  float plo, phi;
};
template <int KT, bool PRO>
__device__ __forceinline__ IrLane<KT, PRO> ir_lane(const IrCommon& q, int t, int j, int kg) {
  IrLane<KT, PRO> l;
#pragma unroll
  for (int kb = 0; kb < KT; ++kb) {
    const int k0 = 16 * kb + 4 * kg;
    const bool kok = k0 < q.K;  // (K % 4 == 0: a quad is inside or outside)
    l.koff[kb] = kok ? k0 : 0;  // (outside: a valid address; its products meet zero weights)
#pragma unroll
    for (int c = 0; c < 4; ++c) l.aw[kb][c] = kok ? q.w1[(size_t)(16 * t + j) * q.K + k0 + c] : 0.f;
    if (PRO) {
      l.psc[kb] = (q.in_scale && kok) ? ld4(q.in_scale + k0) : make_float4(1.f, 1.f, 1.f, 1.f);
      l.psh[kb] = (q.in_shift && kok) ? ld4(q.in_shift + k0) : f4zero();
    }
  }
  l.plo = q.in_act ? 0.f : -INFINITY;
  l.phi = q.in_act == NASSEG_ACT_RELU6 ? 6.f : INFINITY;
  return l;
}
// z1 (4 channels of this lane's tile) at a pixel, in two halves so that the loads of the NEXT row are in flight while
// this one is multiplied: ir_load (the pixel's x, from a valid address) and ir_mma
template <int KT, bool PRO>
struct IrRawX {
  float4 v[KT];
};
template <int KT, bool PRO>
__device__ __forceinline__ IrRawX<KT, PRO> ir_load(const IrLane<KT, PRO>& l, const act_t* xp) {
  IrRawX<KT, PRO> r;
#pragma unroll
  for (int kb = 0; kb < KT; ++kb) r.v[kb] = lda4(xp + l.koff[kb]);
  return r;
}
template <int KT, bool PRO>
__device__ __forceinline__ float4 ir_mma(const IrLane<KT, PRO>& l, const IrRawX<KT, PRO>& r) {
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kb = 0; kb < KT; ++kb) {
    float4 v = r.v[kb];
    if (PRO) {
      v = fma4(v, l.psc[kb], l.psh[kb]);
      v.x = __builtin_amdgcn_fmed3f(v.x, l.plo, l.phi);
      v.y = __builtin_amdgcn_fmed3f(v.y, l.plo, l.phi);
      v.z = __builtin_amdgcn_fmed3f(v.z, l.plo, l.phi);
      v.w = __builtin_amdgcn_fmed3f(v.w, l.plo, l.phi);
    }
    acc = mfma16(l.aw[kb][0], v.x, acc);
    acc = mfma16(l.aw[kb][1], v.y, acc);
    acc = mfma16(l.aw[kb][2], v.z, acc);
    acc = mfma16(l.aw[kb][3], v.w, acc);
  }
  return round_like_storage(make_float4(acc[0], acc[1], acc[2], acc[3]));
}

// ---------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------
struct IrFwdArgs {
  IrCommon c;
  const float* wdw;  // [9][C]
  act_t* z2;         // [B][Ho][Wo][C]
  float* stats;      // [gridDim.x][2][C]: sum, sum of squares of z2
};

template <int STRIDE, int KT, bool PRO>
__global__ __launch_bounds__(64 * kIrMaxWaves, NASSEG_IR_FWD_MINB) void irdw_fwd_kernel(IrFwdArgs a) {
  const IrCommon& q = a.c;
  const int lane = threadIdx.x & 63, t = blockIdx.y * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int j = lane & 15, kg = lane >> 4;
  const int C = q.C, K = q.K, H = q.H, W = q.W, Ho = q.Ho, Wo = q.Wo;
  const int c4 = 4 * t + kg;
  const IrLane<KT, PRO> ln = ir_lane<KT, PRO>(q, t, j, kg);
  const float4 sc1 = ld4(q.sc1 + 4 * c4), sh1 = ld4(q.sh1 + 4 * c4);
  const float lo1 = q.act1 ? 0.f : -INFINITY, hi1 = q.act1 == NASSEG_ACT_RELU6 ? 6.f : INFINITY;
  float4 wd[9];
#pragma unroll
  for (int tp = 0; tp < 9; ++tp) wd[tp] = ld4(a.wdw + (size_t)tp * C + 4 * c4);
  double dsum[4] = {0.0, 0.0, 0.0, 0.0}, dsq[4] = {0.0, 0.0, 0.0, 0.0};
  // output column of this lane: stride 1: lanes 1 .. 14 <-> ox = 14 s + j - 1 (input column = output column);
  // stride 2: odd lanes 1 .. 13 <-> ox = 7 s + (j - 1) / 2, centred on input column 2 ox = 14 s - 1 + j
  const bool out_lane = STRIDE == 1 ? (j >= 1 && j <= 14) : ((j & 1) && j <= 13);

  for (int item = blockIdx.x; item < q.items; item += gridDim.x) {
    const int s = item % q.strips;
    const int rest = item / q.strips;
    const int ch = rest % q.chunks, b = rest / q.chunks;
    const int oy0 = ch * q.rows_per_chunk;
    const int oy1 = oy0 + q.rows_per_chunk < Ho ? oy0 + q.rows_per_chunk : Ho;
    const int col = 14 * s - 1 + j;
    const bool colok = col >= 0 && col < W;
    const int colc = col < 0 ? 0 : (col >= W ? W - 1 : col);
    const int ox = STRIDE == 1 ? col : 7 * s + (j >> 1);
    const bool outok = out_lane && ox < Wo;
    const act_t* xb = q.x + ((size_t)b * H * W + colc) * K;
    act_t* zb = a.z2 + ((size_t)b * Ho * Wo + (outok ? ox : 0)) * C + 4 * c4;
    float4 fs = f4zero(), fq = f4zero();
    // a row of act(BN1(z1)) for this lane's column, with what the lanes to the left and right hold (zero padding
    // of the depthwise conv: rows and columns outside the map are 0, not act(shift)).  The x of row iy + 1 is
    // requested (from a clamped, valid row) before row iy is multiplied.
    auto fetch = [&](int iy) {
      return ir_load<KT, PRO>(ln, xb + (size_t)(iy < 0 ? 0 : (iy >= H ? H - 1 : iy)) * W * K);
    };
    auto make_row = [&](const IrRawX<KT, PRO>& raw, int iy, float4& L, float4& M, float4& R) {
      const float4 z = ir_mma<KT, PRO>(ln, raw);
      float4 a1 = fma4(z, sc1, sh1);
      a1.x = __builtin_amdgcn_fmed3f(a1.x, lo1, hi1);
      a1.y = __builtin_amdgcn_fmed3f(a1.y, lo1, hi1);
      a1.z = __builtin_amdgcn_fmed3f(a1.z, lo1, hi1);
      a1.w = __builtin_amdgcn_fmed3f(a1.w, lo1, hi1);
      a1 = keep_if(a1, colok && iy >= 0 && iy < H);
      M = a1;
      L = from_left(a1);
      R = from_right(a1);
    };
    auto emit = [&](int oy, const float4& A0, const float4& A1, const float4& A2, const float4& B0, const float4& B1,
                    const float4& B2, const float4& C0, const float4& C1, const float4& C2) {
      float4 o = mul4(wd[0], A0);
      o = fma4(wd[1], A1, o);
      o = fma4(wd[2], A2, o);
      o = fma4(wd[3], B0, o);
      o = fma4(wd[4], B1, o);
      o = fma4(wd[5], B2, o);
      o = fma4(wd[6], C0, o);
      o = fma4(wd[7], C1, o);
      o = fma4(wd[8], C2, o);
      if (outok) sta4(zb + (size_t)oy * Wo * C, o);
      const float4 m = keep_if(o, outok);
      fs = add4(fs, m);
      fq = fma4(m, m, fq);
    };
    float4 L0, M0, R0, L1, M1, R1, L2, M2, R2;
    if (STRIDE == 1) {
      // rows oy0 - 1, oy0 prime the window; then one new row per output row, the window's roles rotating over an
      // unrolled triple (no register copies)
      IrRawX<KT, PRO> ra = fetch(oy0 - 1), rb = fetch(oy0);
      make_row(ra, oy0 - 1, L0, M0, R0);
      ra = fetch(oy0 + 1);
      make_row(rb, oy0, L1, M1, R1);
      int oy = oy0;
      for (; oy + 2 < oy1; oy += 3) {
        rb = fetch(oy + 2);
        make_row(ra, oy + 1, L2, M2, R2);
        emit(oy, L0, M0, R0, L1, M1, R1, L2, M2, R2);
        ra = fetch(oy + 3);
        make_row(rb, oy + 2, L0, M0, R0);
        emit(oy + 1, L1, M1, R1, L2, M2, R2, L0, M0, R0);
        rb = fetch(oy + 4);
        make_row(ra, oy + 3, L1, M1, R1);
        emit(oy + 2, L2, M2, R2, L0, M0, R0, L1, M1, R1);
        ra = rb;
      }
      if (oy < oy1) {
        rb = fetch(oy + 2);
        make_row(ra, oy + 1, L2, M2, R2);
        emit(oy, L0, M0, R0, L1, M1, R1, L2, M2, R2);
        if (oy + 1 < oy1) {
          make_row(rb, oy + 2, L0, M0, R0);
          emit(oy + 1, L1, M1, R1, L2, M2, R2, L0, M0, R0);
        }
      }
    } else {
      IrRawX<KT, PRO> ra = fetch(2 * oy0 - 1), rb = fetch(2 * oy0);
      make_row(ra, 2 * oy0 - 1, L0, M0, R0);
      ra = fetch(2 * oy0 + 1);
      int oy = oy0;
      // (the window's odd row changes roles every output row: an unrolled pair)
      for (; oy + 1 < oy1; oy += 2) {
        make_row(rb, 2 * oy, L1, M1, R1);
        rb = fetch(2 * oy + 2);
        make_row(ra, 2 * oy + 1, L2, M2, R2);
        ra = fetch(2 * oy + 3);
        emit(oy, L0, M0, R0, L1, M1, R1, L2, M2, R2);
        make_row(rb, 2 * oy + 2, L1, M1, R1);
        rb = fetch(2 * oy + 4);
        make_row(ra, 2 * oy + 3, L0, M0, R0);
        ra = fetch(2 * oy + 5);
        emit(oy + 1, L2, M2, R2, L1, M1, R1, L0, M0, R0);
      }
      if (oy < oy1) {
        make_row(rb, 2 * oy, L1, M1, R1);
        make_row(ra, 2 * oy + 1, L2, M2, R2);
        emit(oy, L0, M0, R0, L1, M1, R1, L2, M2, R2);
      }
    }
    dsum[0] += (double)fs.x; dsum[1] += (double)fs.y; dsum[2] += (double)fs.z; dsum[3] += (double)fs.w;
    dsq[0] += (double)fq.x; dsq[1] += (double)fq.y; dsq[2] += (double)fq.z; dsq[3] += (double)fq.w;
  }
  // one row per workgroup: the 16 pixel lanes of a k-group meet with DPP adds; lane j == 0 writes its four channels
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    dsum[r] = row16_allsum(dsum[r]);
    dsq[r] = row16_allsum(dsq[r]);
  }
  if (j == 0) {
    float* row = a.stats + (size_t)blockIdx.x * 2 * C + 4 * c4;
    *reinterpret_cast<float4*>(row) = make_float4((float)dsum[0], (float)dsum[1], (float)dsum[2], (float)dsum[3]);
    *reinterpret_cast<float4*>(row + C) = make_float4((float)dsq[0], (float)dsq[1], (float)dsq[2], (float)dsq[3]);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------------------
struct IrBwdArgs {
  IrCommon c;
  const float* mu1;   // BatchNorm of the expansion: mean, invstd (its backward sums are gathered here)
  const float* is1;
  const act_t* g;     // gradient w.r.t. the depthwise conv's BatchNorm output [B][Ho][Wo][C]
  const act_t* z2;    // the depthwise conv's raw output
  const float* wdw;   // [9][C] (flip: the 180-degree rotated packing)
  int flip;
  act_t* ge;          // [B][H][W][C]: gradient w.r.t. the expansion's BatchNorm output, masked with act1'
  float* partial;     // [gridDim.x][9][C]
  float* stats;       // [gridDim.x][2][C]: {sum ge, sum ge * xhat1}
  const float* sc2; const float* sh2; const float* mu2; const float* is2; const float* sums2;
  int train2, act2;
  float invM;
};

struct IrDz {
  float4 ca, cb, cd, cs;
};
__device__ __forceinline__ IrDz ir_dz_const(const IrBwdArgs& a, int c4) {
  // dz = ca*g' + cb*z + cd  ==  scale*(g' - sum(g')/M - xhat*sum(g'*xhat)/M); g' = g * act2'(scale*z + shift)
  IrDz k;
  const int C = a.c.C;
  k.ca = ld4(a.sc2 + c4 * 4);
  k.cs = a.act2 ? ld4(a.sh2 + c4 * 4) : f4zero();
  k.cb = f4zero();
  k.cd = f4zero();
  if (a.train2) {
    const float4 is = ld4(a.is2 + c4 * 4), mu = ld4(a.mu2 + c4 * 4);
    const float4 s0 = ld4(a.sums2 + c4 * 4), s1 = ld4(a.sums2 + C + c4 * 4);
    const float m = a.invM;
    k.cb = make_float4(-k.ca.x * is.x * (s1.x * m), -k.ca.y * is.y * (s1.y * m), -k.ca.z * is.z * (s1.z * m),
                       -k.ca.w * is.w * (s1.w * m));
    k.cd = make_float4(k.ca.x * (mu.x * is.x * (s1.x * m) - s0.x * m), k.ca.y * (mu.y * is.y * (s1.y * m) - s0.y * m),
                       k.ca.z * (mu.z * is.z * (s1.z * m) - s0.z * m), k.ca.w * (mu.w * is.w * (s1.w * m) - s0.w * m));
  }
  return k;
}
struct IrRaw {
  float4 g, z;
  bool ok;
};
__device__ __forceinline__ IrRaw ir_dz_load(const IrBwdArgs& a, int b, int oy, int ox, int c4) {
  IrRaw r;
  const int Ho = a.c.Ho, Wo = a.c.Wo;
  r.ok = oy >= 0 && oy < Ho && ox >= 0 && ox < Wo;
  const int oyc = oy < 0 ? 0 : (oy >= Ho ? Ho - 1 : oy), oxc = ox < 0 ? 0 : (ox >= Wo ? Wo - 1 : ox);
  const size_t off = (((size_t)b * Ho + oyc) * Wo + oxc) * a.c.C + c4 * 4;
  r.g = lda4(a.g + off);
  r.z = lda4(a.z2 + off);
  return r;
}
// (the activations as loop-invariant scalars: a run-time `act` inside the row loop is a branch per component)
__device__ __forceinline__ float4 ir_dz_make(const ActSel& act2, const IrDz& k, const IrRaw& r) {
  const float4 y = fma4(r.z, k.ca, k.cs);
  const float4 gv = make_float4(r.g.x * act_mask(y.x, act2), r.g.y * act_mask(y.y, act2), r.g.z * act_mask(y.z, act2),
                                r.g.w * act_mask(y.w, act2));
  return keep_if(round_like_storage(fma4(gv, k.ca, fma4(r.z, k.cb, k.cd))), r.ok);
}

template <int STRIDE, int KT, bool PRO>
__global__ __launch_bounds__(64 * kIrMaxWaves, NASSEG_IR_BWD_MINB) void irdw_bwd_kernel(IrBwdArgs a) {
  const IrCommon& q = a.c;
  const int lane = threadIdx.x & 63, t = blockIdx.y * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int j = lane & 15, kg = lane >> 4;
  const int C = q.C, K = q.K, H = q.H, W = q.W;
  const int c4 = 4 * t + kg;
  const IrLane<KT, PRO> ln = ir_lane<KT, PRO>(q, t, j, kg);
  const float4 sc1 = ld4(q.sc1 + 4 * c4), sh1 = ld4(q.sh1 + 4 * c4);
  const float4 mu1 = ld4(a.mu1 + 4 * c4), is1 = ld4(a.is1 + 4 * c4);
  const float lo1 = q.act1 ? 0.f : -INFINITY, hi1 = q.act1 == NASSEG_ACT_RELU6 ? 6.f : INFINITY;
  const IrDz kc = ir_dz_const(a, c4);
  const ActSel sel1 = act_sel(q.act1), sel2 = act_sel(a.act2);
  float4 lw[9];
#pragma unroll
  for (int tp = 0; tp < 9; ++tp) lw[tp] = ld4(a.wdw + (size_t)(a.flip ? 8 - tp : tp) * C + 4 * c4);
  float4 dwa[9];
#pragma unroll
  for (int tp = 0; tp < 9; ++tp) dwa[tp] = f4zero();
  float4 ssum[2] = {f4zero(), f4zero()};
  // ge of one input pixel: masked with act1'(sc1 z1 + sh1); its sums against xhat1
  auto finish = [&](float4& o, float4 z1, bool ok) {
    const float4 y = fma4(z1, sc1, sh1);
    o.x *= act_mask(y.x, sel1);
    o.y *= act_mask(y.y, sel1);
    o.z *= act_mask(y.z, sel1);
    o.w *= act_mask(y.w, sel1);
    const float4 v = keep_if(o, ok);
    const float4 xh = make_float4((z1.x - mu1.x) * is1.x, (z1.y - mu1.y) * is1.y, (z1.z - mu1.z) * is1.z,
                                  (z1.w - mu1.w) * is1.w);
    ssum[0] = add4(ssum[0], v);
    ssum[1] = fma4(v, xh, ssum[1]);
  };
  auto activated = [&](float4 z1) {
    float4 v = fma4(z1, sc1, sh1);
    v.x = __builtin_amdgcn_fmed3f(v.x, lo1, hi1);
    v.y = __builtin_amdgcn_fmed3f(v.y, lo1, hi1);
    v.z = __builtin_amdgcn_fmed3f(v.z, lo1, hi1);
    v.w = __builtin_amdgcn_fmed3f(v.w, lo1, hi1);
    return v;
  };

  for (int item = blockIdx.x; item < q.items; item += gridDim.x) {
    const int s = item % q.strips;
    const int rest = item / q.strips;
    const int ch = rest % q.chunks, b = rest / q.chunks;
    const int r0 = ch * q.rows_per_chunk;
    // A lane loads and differentiates ONE column of the gradient; the 3x3 window's other columns come from the
    // neighbouring lanes of its 16-lane row (DPP), as in the forward kernel: a third of the loads and of the
    // BatchNorm-backward arithmetic of the window, and registers for a prefetch two rows deep.
    if (STRIDE == 1) {
      // lane j <-> column 14 s - 1 + j; lanes 0 and 15 are halo
      const int r1 = r0 + q.rows_per_chunk < H ? r0 + q.rows_per_chunk : H;
      const int ix = 14 * s - 1 + j;
      const bool live = j >= 1 && j <= 14 && ix < W;
      const int xc = ix < 0 ? 0 : (ix >= W ? W - 1 : ix);
      const bool colok = ix >= 0 && ix < W;  // (dz outside the map is 0: ir_dz_load masks by the clamped column itself)
      const act_t* xcol = q.x + ((size_t)b * H * W + xc) * K;
      // window rows: dz[iy - 1], dz[iy], dz[iy + 1] with their left / right neighbours
      float4 dl[3], dm[3], dr[3];
      auto make_dz = [&](const IrRaw& raw, float4& L, float4& M, float4& R) {
        IrRaw r = raw;
        r.ok = r.ok && colok;
        M = ir_dz_make(sel2, kc, r);
        L = from_left(M);
        R = from_right(M);
      };
      make_dz(ir_dz_load(a, b, r0 - 1, xc, c4), dl[1], dm[1], dr[1]);
      make_dz(ir_dz_load(a, b, r0, xc, c4), dl[2], dm[2], dr[2]);
      IrRaw nxt = ir_dz_load(a, b, r0 + 1, xc, c4), nxt2 = ir_dz_load(a, b, r0 + 2, xc, c4);
      IrRawX<KT, PRO> xnxt = ir_load<KT, PRO>(ln, xcol + (size_t)r0 * W * K);
      IrRawX<KT, PRO> xnxt2 = ir_load<KT, PRO>(ln, xcol + (size_t)(r0 + 1 < H ? r0 + 1 : H - 1) * W * K);
      for (int iy = r0; iy < r1; ++iy) {
        const IrRaw cur = nxt;
        const IrRawX<KT, PRO> xcur = xnxt;
        nxt = nxt2;
        xnxt = xnxt2;
        // (rows iy + 1 and iy + 2 are in flight while row iy is computed)
        nxt2 = ir_dz_load(a, b, iy + 3, xc, c4);
        xnxt2 = ir_load<KT, PRO>(ln, xcol + (size_t)(iy + 2 < H ? iy + 2 : H - 1) * W * K);
        const float4 z1 = ir_mma<KT, PRO>(ln, xcur);
        dl[0] = dl[1]; dm[0] = dm[1]; dr[0] = dr[1];
        dl[1] = dl[2]; dm[1] = dm[2]; dr[1] = dr[2];
        make_dz(cur, dl[2], dm[2], dr[2]);
        const float4 xa = keep_if(activated(z1), live);
        // tap (ty, tx) pairs input (iy, ix) with output (iy + 1 - ty, ix + 1 - tx): row 2 - ty of the window, column
        // ix + 1 (tx = 0: the lane to the right), ix (tx = 1), ix - 1 (tx = 2: the lane to the left)
        float4 o = f4zero();
#pragma unroll
        for (int ty = 0; ty < 3; ++ty) {
          o = fma4(lw[ty * 3 + 0], dr[2 - ty], o);
          dwa[ty * 3 + 0] = fma4(xa, dr[2 - ty], dwa[ty * 3 + 0]);
          o = fma4(lw[ty * 3 + 1], dm[2 - ty], o);
          dwa[ty * 3 + 1] = fma4(xa, dm[2 - ty], dwa[ty * 3 + 1]);
          o = fma4(lw[ty * 3 + 2], dl[2 - ty], o);
          dwa[ty * 3 + 2] = fma4(xa, dl[2 - ty], dwa[ty * 3 + 2]);
        }
        finish(o, z1, live);
        if (live) sta4(a.ge + (((size_t)b * H + iy) * W + ix) * C + c4 * 4, o);
      }
    } else {
      // lane j <-> quad column 15 s + j (input columns 2 xq, 2 xq + 1); lane 15 only supplies the gradient column to
      // the right of lane 14.  A 2x2 input quad (2a + py, 2b + px) receives from dz[a + ry][b + rx], ry, rx in {0, 1}:
      // tap ty feeds input parity py = (ty + 1) & 1 from row ry = (py + 1 - ty) / 2
      const int Hq = (H + 1) >> 1, Wq = (W + 1) >> 1;
      const int r1 = r0 + q.rows_per_chunk < Hq ? r0 + q.rows_per_chunk : Hq;
      const int xq = 15 * s + j;
      const bool live = j <= 14 && xq < Wq;
      const int xqc = xq < Wq ? xq : Wq - 1;
      float4 d[2][2];  // [ry][rx]
      auto make_dz = [&](const IrRaw& raw, float4& M, float4& R) {
        M = ir_dz_make(sel2, kc, raw);
        R = from_right(M);
      };
      auto fetch_quad = [&](int aq, IrRawX<KT, PRO> (&dst)[2][2]) {
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
          for (int px = 0; px < 2; ++px) {
            const int iy = 2 * aq + py, ix = 2 * xqc + px;
            dst[py][px] = ir_load<KT, PRO>(ln, q.x + (((size_t)b * H + (iy < H ? iy : H - 1)) * W + (ix < W ? ix : W - 1)) * K);
          }
      };
      // (a column beyond the map's last output column reads as 0: ir_dz_load checks the UNclamped column)
      make_dz(ir_dz_load(a, b, r0, xq, c4), d[1][0], d[1][1]);
      IrRaw nxt = ir_dz_load(a, b, r0 + 1, xq, c4);
      IrRawX<KT, PRO> xnxt[2][2];
      fetch_quad(r0, xnxt);
      for (int aq = r0; aq < r1; ++aq) {
        const IrRaw cur = nxt;
        IrRawX<KT, PRO> xcur[2][2];
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
          for (int px = 0; px < 2; ++px) xcur[py][px] = xnxt[py][px];
        // (the loads of quad row aq + 1 - one dz column, four x pixels - are in flight while quad row aq is computed)
        nxt = ir_dz_load(a, b, aq + 2, xq, c4);
        fetch_quad(aq + 1 < Hq ? aq + 1 : aq, xnxt);
        d[0][0] = d[1][0];
        d[0][1] = d[1][1];
        make_dz(cur, d[1][0], d[1][1]);
        // one input pixel of the quad at a time (z1, its activation and its gradient live for one pixel only)
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
          for (int px = 0; px < 2; ++px) {
            const int iy = 2 * aq + py, ix = 2 * xqc + px;
            const bool ok = live && iy < H && ix < W;
            const float4 z1 = ir_mma<KT, PRO>(ln, xcur[py][px]);
            const float4 xa = keep_if(activated(z1), ok);
            float4 o = f4zero();
#pragma unroll
            for (int ty = 0; ty < 3; ++ty)
#pragma unroll
              for (int tx = 0; tx < 3; ++tx) {
                if (((ty + 1) & 1) != py || ((tx + 1) & 1) != px) continue;
                const int ry = (py + 1 - ty) / 2, rx = (px + 1 - tx) / 2;
                const float4 dv = d[ry][rx];
                o = fma4(lw[ty * 3 + tx], dv, o);
                dwa[ty * 3 + tx] = fma4(xa, dv, dwa[ty * 3 + tx]);
              }
            finish(o, z1, ok);
            if (ok) sta4(a.ge + (((size_t)b * H + iy) * W + 2 * xq + px) * C + c4 * 4, o);
          }
      }
    }
  }
  // one row per workgroup: the 16 column lanes of a k-group meet with DPP adds; lane j == 0 writes
#pragma unroll
  for (int tp = 0; tp < 9; ++tp) {
    float4 v = dwa[tp];
    v = make_float4(row16_allsum(v.x), row16_allsum(v.y), row16_allsum(v.z), row16_allsum(v.w));
    if (j == 0) *reinterpret_cast<float4*>(a.partial + ((size_t)blockIdx.x * 9 + tp) * C + 4 * c4) = v;
  }
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    float4 v = ssum[r];
    v = make_float4(row16_allsum(v.x), row16_allsum(v.y), row16_allsum(v.z), row16_allsum(v.w));
    if (j == 0) *reinterpret_cast<float4*>(a.stats + ((size_t)blockIdx.x * 2 + r) * C + 4 * c4) = v;
  }
}


// ---------------------------------------------------------------------------------------------------------------
// statistics of the expansion's output without computing it: z1 = W1 x~ is linear in x~ = pro(x), so
//   sum_p z1[p][n] = w_n . s,   sum_p z1[p][n]^2 = w_n^T S w_n,   s = sum_p x~[p],  S = sum_p x~[p] x~[p]^T
// - a K x K matrix and a K-vector from ONE pass over the block's input (K = 16 ... 32 channels where z1 has 96 ... 192),
// then K^2 multiply-adds per output channel.  S = X^T X is itself a matrix product with the pixels as the reduction
// axis: v_mfma_f32_16x16x4_f32 with A[m][k] = B[k][m] = x~[pixel k][channel m], four pixels per instruction.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kMomTile = 64;  // pixels per LDS tile
constexpr int kMomGrid = 512;

template <int KT, bool PRO>
__global__ __launch_bounds__(256) void ir_moments_kernel(const act_t* __restrict__ x, const float* __restrict__ in_scale,
                                                         const float* __restrict__ in_shift, int in_act, int K, int M,
                                                         float* __restrict__ rows) {
  constexpr int KP = 16 * KT, Q = KP / 4, LSK = KP + 4, IT = (kMomTile * Q) / 256;  // float4 items per thread and tile
  __shared__ float xs[kMomTile * LSK];
  __shared__ float red[KP * KP > 256 * 4 ? KP * KP : 256 * 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, kg = lane >> 4;
  const int quad = tid % Q, k0 = 4 * quad;  // (256 % Q == 0: a thread's items all belong to one channel quad)
  const bool kok = k0 < K;
  float4 psc = make_float4(1.f, 1.f, 1.f, 1.f), psh = f4zero();
  if (PRO && kok) {
    if (in_scale) psc = ld4(in_scale + k0);
    if (in_shift) psh = ld4(in_shift + k0);
  }
  const float plo = in_act ? 0.f : -INFINITY, phi = in_act == NASSEG_ACT_RELU6 ? 6.f : INFINITY;
  f32x4 acc[KT][KT];
#pragma unroll
  for (int ta = 0; ta < KT; ++ta)
#pragma unroll
    for (int tb = 0; tb < KT; ++tb) acc[ta][tb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 csum = f4zero();
  const int ntiles = (M + kMomTile - 1) / kMomTile;
  float4 raw[IT];
  auto issue = [&](int tile) {
#pragma unroll
    for (int u = 0; u < IT; ++u) {
      const int p = tile * kMomTile + (tid + 256 * u) / Q;
      raw[u] = lda4(x + (size_t)(p < M ? p : M - 1) * K + (kok ? k0 : 0));
    }
  };
  if ((int)blockIdx.x < ntiles) issue(blockIdx.x);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
#pragma unroll
    for (int u = 0; u < IT; ++u) {
      const int px = (tid + 256 * u) / Q;
      float4 v = raw[u];
      if (PRO) {
        v = fma4(v, psc, psh);
        v.x = __builtin_amdgcn_fmed3f(v.x, plo, phi);
        v.y = __builtin_amdgcn_fmed3f(v.y, plo, phi);
        v.z = __builtin_amdgcn_fmed3f(v.z, plo, phi);
        v.w = __builtin_amdgcn_fmed3f(v.w, plo, phi);
      }
      v = keep_if(v, kok && tile * kMomTile + px < M);
      *reinterpret_cast<float4*>(&xs[px * LSK + k0]) = v;
      csum = add4(csum, v);
    }
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) issue(tile + gridDim.x);  // (in flight while this tile is multiplied)
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {  // wave w takes pixel groups 4 w .. 4 w + 3 of the tile's 16
      const float* row = &xs[(4 * (4 * wave + q4) + kg) * LSK + j];
      float a[KT];
#pragma unroll
      for (int ta = 0; ta < KT; ++ta) a[ta] = row[16 * ta];
#pragma unroll
      for (int ta = 0; ta < KT; ++ta)
#pragma unroll
        for (int tb = 0; tb < KT; ++tb) acc[ta][tb] = mfma16(a[ta], a[tb], acc[ta][tb]);
    }
    __syncthreads();
  }
  // the four waves' S in a fixed order, then the column sums
  float* out = rows + (size_t)blockIdx.x * (KP * KP + KP);
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int ta = 0; ta < KT; ++ta)
#pragma unroll
        for (int tb = 0; tb < KT; ++tb)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float* slot = &red[(16 * ta + 4 * kg + i) * KP + 16 * tb + j];
            *slot = (w == 0 ? 0.f : *slot) + acc[ta][tb][i];
          }
    }
    __syncthreads();
  }
  for (int e = tid; e < KP * KP; e += 256) out[e] = red[e];
  __syncthreads();
  *reinterpret_cast<float4*>(&red[4 * tid]) = csum;
  __syncthreads();
  if (tid < KP) {
    const int qd = tid >> 2, c = tid & 3;
    float sacc = 0.f;
    for (int u = qd; u < 256; u += Q) sacc += red[4 * u + c];
    out[KP * KP + tid] = sacc;
  }
}

// mean / variance of z1 per output channel from the summed moments, and everything nasseg_bn_finalize writes.
// 256 threads = 8 output channels x 32 lanes: lane k of a channel n forms w[n][k] * (S[k][.] . w[n][.]) and
// w[n][k] * s[k]; the 32 lanes meet with shuffles (fp64 throughout).
__global__ __launch_bounds__(256) void ir_moments_finalize(const float* __restrict__ mom, const float* __restrict__ w1,
                                                           int K, int KP, int C, double M, float eps, float momentum,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ mean, float* __restrict__ invstd,
                                                           float* __restrict__ scale, float* __restrict__ shift,
                                                           float* __restrict__ running_mean,
                                                           float* __restrict__ running_var, int64_t* nbt) {
  __shared__ float S[32 * 32 + 32];
  __shared__ float wn[8][32];
  for (int e = threadIdx.x; e < KP * KP + KP; e += 256) S[e] = mom[e];
  const int nl = threadIdx.x >> 5, k = threadIdx.x & 31;
  const int n = blockIdx.x * 8 + nl;
  wn[nl][k] = (n < C && k < K) ? w1[(size_t)n * K + k] : 0.f;
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) *nbt += 1;
  const double wk = (double)wn[nl][k];
  double r = 0.0;
  if (k < K) {
    for (int l = 0; l < K; ++l) r += (double)S[k * KP + l] * (double)wn[nl][l];
  }
  double wsw = wk * r, ws = k < K ? wk * (double)S[KP * KP + k] : 0.0;
#pragma unroll
  for (int off = 16; off >= 1; off >>= 1) {  // (lanes 32 nl' .. 32 nl' + 31 of a wave: a fixed tree)
    wsw += __shfl_xor(wsw, off);
    ws += __shfl_xor(ws, off);
  }
  if (n >= C || k != 0) return;
  const double mu = ws / M;
  double var = wsw / M - mu * mu;
  if (var < 0.0) var = 0.0;
  const double is = 1.0 / sqrt(var + (double)eps);
  mean[n] = (float)mu;
  invstd[n] = (float)is;
  const double g = gamma ? (double)gamma[n] : 1.0;
  const double bt = beta ? (double)beta[n] : 0.0;
  scale[n] = (float)(g * is);
  shift[n] = (float)(bt - mu * g * is);
  if (running_mean) {
    running_mean[n] = (float)((1.0 - momentum) * (double)running_mean[n] + momentum * mu);
    const double unb = M > 1.0 ? var * M / (M - 1.0) : var;
    running_var[n] = (float)((1.0 - momentum) * (double)running_var[n] + momentum * unb);
  }
}

// work decomposition shared by the queries and the launches
struct IrPlan {
  int ok, kt, waves, groups, strips, chunks, rows_per_chunk, items, grid;
};
inline IrPlan ir_plan(int B, int H, int W, int K, int C, int stride, bool backward) {
  IrPlan p = {};
  if (B <= 0 || H <= 0 || W <= 0 || K % 4 || K < 4 || K > 32 || C % 16 || C < 16 || C > 16 * kIrMaxTiles ||
      (stride != 1 && stride != 2))
    return p;
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  p.kt = K <= 16 ? 1 : 2;
  const int tiles = C / 16;
  p.waves = tiles % 4 == 0 ? 4 : (tiles % 3 == 0 ? 3 : (tiles % 2 == 0 ? 2 : 1));
  p.groups = tiles / p.waves;
  int rows;  // rows the chunks divide: output rows (forward), input rows / quad rows (backward)
  if (!backward) {
    p.strips = stride == 1 ? cdiv(W, 14) : cdiv(Wo, 7);
    rows = Ho;
  } else {
    p.strips = stride == 1 ? cdiv(W, 14) : cdiv((W + 1) / 2, 15);
    rows = stride == 1 ? H : (H + 1) / 2;
  }
  // workgroups: what is resident at once (2048 threads per CU); items: about four per workgroup, chunks of at least
  // 8 rows (forward recomputes 2 / 1 halo rows per chunk)
  // (two workgroups per CU and group of tiles: the kernels hold 150 - 250 registers per lane)
  const int resident = 256 * 2;
  int rpc = 64;
  while (rpc > 8 && (int64_t)B * cdiv(rows, rpc) * p.strips < 4LL * resident) rpc >>= 1;
  p.rows_per_chunk = rpc;
  p.chunks = cdiv(rows, rpc);
  const int64_t items = (int64_t)B * p.chunks * p.strips;
  if (items >= 2147483647LL) return p;
  p.items = (int)items;
  p.grid = p.items < resident ? p.items : resident;
  p.ok = 1;
  return p;
}

}  // namespace

extern "C" {

#if NASSEG_FP32_ONLY
// partial rows (= workgroups) of nasseg_irdw_fwd / nasseg_irdw_bwd; 0: geometry not served (K % 4 == 0, K <= 32,
// C % 16 == 0, C <= 192, stride 1 or 2 - MobileNetV2's 16 -> 96, 24 -> 144, 32 -> 192)
int64_t nasseg_irdw_rows(int B, int H, int W, int K, int C, int stride, int backward) {
  const IrPlan p = ir_plan(B, H, W, K, C, stride, backward != 0);
  return p.ok ? p.grid : 0;
}
#else
int64_t nasseg_irdw_rows(int B, int H, int W, int K, int C, int stride, int backward);
#endif

#if NASSEG_FP32_ONLY
// floats of workspace nasseg_irdw_stats needs (partial moment rows + their sum)
int64_t nasseg_irdw_stats_workspace(int K) {
  const int KP = K <= 16 ? 16 : 32;
  return (int64_t)(kMomGrid + 64 + 1) * (KP * KP + KP);
}
#endif

// Training-mode BatchNorm statistics of z1 = W1 pro(x) [B*H*W][C] WITHOUT computing z1 (see ir_moments_kernel): what
// nasseg_bn_finalize writes for the stored map - mean, invstd, scale = gamma * invstd, shift, running statistics,
// num_batches_tracked - to the rounding of the sums (moments in fp32 per workgroup, everything after that in fp64).
// K % 4 == 0, K <= 32, C <= 16384; ws: nasseg_irdw_stats_workspace(K) floats.
int NASSEG_FN(irdw_stats)(const act_t* x, const float* w1, const float* in_scale, const float* in_shift, int in_act,
                          int B, int H, int W, int K, int C, float eps, float momentum, const float* gamma,
                          const float* beta, float* mean, float* invstd, float* scale, float* shift,
                          float* running_mean, float* running_var, int64_t* num_batches_tracked, float* ws,
                          void* stream) {
  NASSEG_REQUIRE(x && w1 && mean && invstd && scale && shift && ws, "irdw_stats: null tensor");
  NASSEG_REQUIRE(B > 0 && H > 0 && W > 0 && K >= 4 && K <= 32 && K % 4 == 0 && C > 0, "irdw_stats: bad shape");
  const int64_t M64 = (int64_t)B * H * W;
  NASSEG_REQUIRE(M64 < 2147483647LL, "irdw_stats: too many pixels");
  const int M = (int)M64;
  const int KP = K <= 16 ? 16 : 32, per = KP * KP + KP;
  const int ntiles = (M + kMomTile - 1) / kMomTile;
  const int grid = ntiles < kMomGrid ? ntiles : kMomGrid;
  const bool pro = in_scale || in_shift || in_act;
  hipStream_t s = (hipStream_t)stream;
  if (KP == 16 && pro) hipLaunchKernelGGL((ir_moments_kernel<1, true>), dim3(grid), dim3(256), 0, s, x, in_scale, in_shift, in_act, K, M, ws);
  else if (KP == 16) hipLaunchKernelGGL((ir_moments_kernel<1, false>), dim3(grid), dim3(256), 0, s, x, in_scale, in_shift, in_act, K, M, ws);
  else if (pro) hipLaunchKernelGGL((ir_moments_kernel<2, true>), dim3(grid), dim3(256), 0, s, x, in_scale, in_shift, in_act, K, M, ws);
  else hipLaunchKernelGGL((ir_moments_kernel<2, false>), dim3(grid), dim3(256), 0, s, x, in_scale, in_shift, in_act, K, M, ws);
  NASSEG_LAUNCH_CHECK("ir_moments_kernel");
  float* total = ws + (size_t)(kMomGrid + 64) * per;
  const int rc = nasseg_rows_sum(ws, grid, per, total, stream);
  if (rc != NASSEG_OK) return rc;
  hipLaunchKernelGGL(ir_moments_finalize, dim3(cdiv(C, 8)), dim3(256), 0, s, total, w1, K, KP, C, (double)M, eps, momentum,
                     gamma, beta, mean, invstd, scale, shift, running_mean, running_var, num_batches_tracked);
  NASSEG_LAUNCH_CHECK("ir_moments_finalize");
  return NASSEG_OK;
}

#define IR_LAUNCH(KERNEL, ARGS)                                                                              \
  do {                                                                                                       \
    const dim3 grid(p.grid, p.groups), block(64 * p.waves);                                                            \
    if (stride == 1 && p.kt == 1 && pro) hipLaunchKernelGGL((KERNEL<1, 1, true>), grid, block, 0, s, ARGS);  \
    else if (stride == 1 && p.kt == 1) hipLaunchKernelGGL((KERNEL<1, 1, false>), grid, block, 0, s, ARGS);   \
    else if (stride == 1 && pro) hipLaunchKernelGGL((KERNEL<1, 2, true>), grid, block, 0, s, ARGS);          \
    else if (stride == 1) hipLaunchKernelGGL((KERNEL<1, 2, false>), grid, block, 0, s, ARGS);                \
    else if (p.kt == 1 && pro) hipLaunchKernelGGL((KERNEL<2, 1, true>), grid, block, 0, s, ARGS);            \
    else if (p.kt == 1) hipLaunchKernelGGL((KERNEL<2, 1, false>), grid, block, 0, s, ARGS);                  \
    else if (pro) hipLaunchKernelGGL((KERNEL<2, 2, true>), grid, block, 0, s, ARGS);                         \
    else hipLaunchKernelGGL((KERNEL<2, 2, false>), grid, block, 0, s, ARGS);                                 \
  } while (0)

static void ir_fill_common(IrCommon& c, const act_t* x, const float* w1, const float* in_scale, const float* in_shift,
                           int in_act, const float* sc1, const float* sh1, int act1, int B, int H, int W, int K, int C,
                           int Ho, int Wo, const IrPlan& p) {
  c.x = x; c.w1 = w1; c.in_scale = in_scale; c.in_shift = in_shift; c.in_act = in_act;
  c.sc1 = sc1; c.sh1 = sh1; c.act1 = act1;
  c.B = B; c.H = H; c.W = W; c.K = K; c.C = C; c.Ho = Ho; c.Wo = Wo;
  c.strips = p.strips; c.chunks = p.chunks; c.rows_per_chunk = p.rows_per_chunk; c.items = p.items;
}

// z2 = dwconv3x3(act1(sc1 * (W1 * pro(x)) + sh1)), pad 1, stride 1 or 2, with the expansion's output rebuilt on the
// matrix cores instead of read; stats: rows [nasseg_irdw_rows(.., 0)][2][C] of {sum z2, sum z2^2} for
// nasseg_bn_finalize.  w1 (C, K, 1, 1) as PyTorch stores it; wdw packed [9][C] (nasseg_dw_pack_weight, not flipped).
int NASSEG_FN(irdw_fwd)(const act_t* x, const float* w1, const float* wdw, act_t* z2, const float* in_scale,
                        const float* in_shift, int in_act, const float* bn1_scale, const float* bn1_shift, int act1,
                        int B, int H, int W, int K, int C, int Ho, int Wo, int stride, float* stats, void* stream) {
  NASSEG_REQUIRE(x && w1 && wdw && z2 && bn1_scale && bn1_shift && stats, "irdw_fwd: null tensor");
  const IrPlan p = ir_plan(B, H, W, K, C, stride, false);
  NASSEG_REQUIRE(p.ok, "irdw_fwd: geometry not served (K=%d C=%d stride %d)", K, C, stride);
  NASSEG_REQUIRE(Ho == (H - 1) / stride + 1 && Wo == (W - 1) / stride + 1, "irdw_fwd: output size does not match");
  IrFwdArgs a = {};
  ir_fill_common(a.c, x, w1, in_scale, in_shift, in_act, bn1_scale, bn1_shift, act1, B, H, W, K, C, Ho, Wo, p);
  a.wdw = wdw; a.z2 = z2; a.stats = stats;
  const bool pro = in_scale || in_shift || in_act;
  hipStream_t s = (hipStream_t)stream;
  IR_LAUNCH(irdw_fwd_kernel, a);
  NASSEG_LAUNCH_CHECK("irdw_fwd_kernel");
  return NASSEG_OK;
}

// The one-kernel backward of that depthwise conv (nasseg_dwconv_bwd_bn) with the expansion's output rebuilt from x:
// ge (out) [B][H][W][C] = act1'(.) * dwconv_backward_data(dz2), dz2 the BatchNorm-backward of g on load (bn2_*);
// stats rows [r][2][C] of {sum ge, sum ge * xhat1}, ws rows [r][9][C] of weight-gradient partials, r <
// nasseg_irdw_rows(.., 1); dw (C, 1, 3, 3) when given (else nasseg_wgrad_finalize_many: taps 9, N = C, K = 1).
int NASSEG_FN(irdw_bwd)(const act_t* x, const float* w1, const act_t* g, const act_t* z2, const float* wdw,
                        int wdw_flipped, act_t* ge, float* dw, float* ws, const float* in_scale,
                        const float* in_shift, int in_act, const float* bn1_scale, const float* bn1_shift,
                        const float* bn1_mean, const float* bn1_invstd, int act1, const float* bn2_scale,
                        const float* bn2_shift, const float* bn2_mean, const float* bn2_invstd,
                        const float* bn2_sums, int bn2_train, int bn2_act, int B, int H, int W, int K, int C, int Ho,
                        int Wo, int stride, float* stats, void* stream) {
  NASSEG_REQUIRE(x && w1 && g && z2 && wdw && ge && ws && stats && bn1_scale && bn1_shift && bn1_mean && bn1_invstd &&
                     bn2_scale,
                 "irdw_bwd: null tensor");
  NASSEG_REQUIRE((!bn2_train || (bn2_mean && bn2_invstd && bn2_sums)) && (!bn2_act || bn2_shift),
                 "irdw_bwd: missing BatchNorm tensors");
  const IrPlan p = ir_plan(B, H, W, K, C, stride, true);
  NASSEG_REQUIRE(p.ok, "irdw_bwd: geometry not served (K=%d C=%d stride %d)", K, C, stride);
  NASSEG_REQUIRE(Ho == (H - 1) / stride + 1 && Wo == (W - 1) / stride + 1, "irdw_bwd: output size does not match");
  IrBwdArgs a = {};
  ir_fill_common(a.c, x, w1, in_scale, in_shift, in_act, bn1_scale, bn1_shift, act1, B, H, W, K, C, Ho, Wo, p);
  a.mu1 = bn1_mean; a.is1 = bn1_invstd; a.g = g; a.z2 = z2; a.wdw = wdw; a.flip = wdw_flipped != 0;
  a.ge = ge; a.partial = ws; a.stats = stats;
  a.sc2 = bn2_scale; a.sh2 = bn2_shift; a.mu2 = bn2_mean; a.is2 = bn2_invstd; a.sums2 = bn2_sums;
  a.train2 = bn2_train; a.act2 = bn2_act;
  a.invM = (float)(1.0 / ((double)B * Ho * Wo));
  const bool pro = in_scale || in_shift || in_act;
  hipStream_t s = (hipStream_t)stream;
  IR_LAUNCH(irdw_bwd_kernel, a);
  NASSEG_LAUNCH_CHECK("irdw_bwd_kernel");
  if (!dw) return NASSEG_OK;
  const float* parts[1] = {ws};
  float* outs[1] = {dw};
  const int dims[5] = {p.grid, 9, C, 1, 0};
  return nasseg_wgrad_finalize_many(1, parts, outs, dims, stream);
}

}  // extern "C"
