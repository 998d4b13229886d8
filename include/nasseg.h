/* nasseg.h - C ABI of libnasseg_hip.so: the MI355X (gfx950) kernels behind the
 * NAS inner loop of DrSleep/nas-segm-pytorch.
 *
 * The reference has no FFI for this path: its boundary is the Python op
 * registry OPS / AGG_OPS (src/nn/layer_factory.py:27-91), the decoder classes
 * (src/nn/micro_decoders.py:142,257) and one Cython module
 * (src/helpers/miou_utils.pyx).  Underneath, every op is an ATen call; the
 * entry points below are the from-scratch gfx950 replacements of exactly those
 * ATen calls, one group per reference call site.  The host side
 * (nas-segm-pytorch_amd/) binds them with ctypes and re-creates the reference's
 * registry / decoder / engine API on top.
 *
 * Conventions
 *  - plain C types only; every pointer except `const int64_t* cm` in
 *    nasseg_compute_ius_accs is a DEVICE pointer owned by the caller (PyTorch's
 *    caching allocator); the library allocates nothing and keeps no state apart
 *    from a per-thread error string;
 *  - activations are fp32 NHWC ("channels_last"): element (b,y,x,c) of a tensor
 *    with pixel stride ld is at ((b*H + y)*W + x)*ld + c; weights arrive in the
 *    PyTorch layouts ((C,1,k,k) depthwise, (N,K,kh,kw) dense) and are re-packed
 *    by the pack entry points;
 *  - `stream` is a hipStream_t passed as void* (0 = default stream); calls are
 *    asynchronous on that stream, re-entrant and thread-safe;
 *  - return value 0 = ok, negative = error; nasseg_last_error() then returns a
 *    message valid until the same thread's next failing call.  The Python
 *    binding raises RuntimeError, which the reference's try_except decorator
 *    (src/helpers/utils.py:172-187) turns into reward 0 for that candidate;
 *  - act codes: 0 none, 1 ReLU, 2 ReLU6.
 */
#ifndef NASSEG_H
#define NASSEG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* nasseg_last_error(void);
int nasseg_abi_version(void);
int nasseg_device_count(void);

/* ---- depthwise convolution ------------------------------------------------
 * replaces nn.Conv2d(C, C, k, stride, padding, dilation, groups=C, bias=False)
 * in SepConv / DilConv (layer_factory.py:198-265) and InvertedResidual
 * (layer_factory.py:125-158). */
int nasseg_dw_pack_weight(const float* w, float* wt, int C, int K, int flip, void* stream);
int nasseg_dwconv(const float* x, const float* wt, float* y, const float* in_scale,
                  const float* in_shift, int in_act, const float* scale, const float* shift, int act,
                  int B, int H, int W, int C, int Ho, int Wo, int K, int stride, int pad, int dil,
                  int transposed, float* stats, void* stream);
int nasseg_dwconv_strip_ok(int K, int stride, int dil);
int64_t nasseg_dwconv_stats_blocks(int B, int C, int Ho, int Wo, int K, int stride, int dil);
/* backward-data fused with the first half of the backward of the BatchNorm whose
 * normalised output act(scale*z+shift) the conv read (normalise-on-read chains): writes
 * g = act'(..) * dwconv_backward_data(dy) and per-workgroup rows of {sum g, sum g*xhat} */
int nasseg_dwconv_bwd_data_bn(const float* dy, const float* wt, float* g, const float* z,
                              const float* scale, const float* shift, const float* mean,
                              const float* invstd, int act, int B, int H, int W, int C, int Ho,
                              int Wo, int K, int stride, int pad, int dil, int transposed,
                              float* stats, void* stream);
int64_t nasseg_dwconv_bwd_data_bn_blocks(int B, int C, int Ho, int Wo, int K, int stride, int pad,
                                         int dil, int transposed);
/* the whole backward of a 3x3 depthwise conv between two BatchNorms of a chain (InvertedResidual, layer_factory.py:
 * 139-152) in one kernel: BatchNorm backward behind the conv on load, weight gradient, input gradient masked with
 * the activation of the BatchNorm in front (in_*) and that BatchNorm's backward partial sums; xz, g, z read once,
 * ge written once.  rows = workgroups = partial rows of ws [rows][9][C] and stats [rows][2][C]; 0 = not served */
int64_t nasseg_dwconv_bwd_bn_rows(int B, int C, int H, int W, int K, int stride, int pad, int dil);
int nasseg_dwconv_bwd_bn(const float* xz, const float* g, const float* z, const float* wt, int wt_flipped, float* ge,
                         float* dw, float* ws, const float* in_scale, const float* in_shift, const float* in_mean,
                         const float* in_invstd, int in_act, const float* bn_scale, const float* bn_shift,
                         const float* bn_mean, const float* bn_invstd, const float* bn_sums, int bn_train, int bn_act,
                         int B, int H, int W, int C, int Ho, int Wo, int K, int stride, int pad, int dil, float* stats,
                         void* stream);
int64_t nasseg_dwconv_wgrad_workspace(int B, int C, int Ho, int Wo, int K);
int nasseg_dwconv_wgrad(const float* x, const float* dy, float* dw, float* ws,
                        const float* in_scale, const float* in_shift, int in_act, int B, int H, int W,
                        int C, int Ho, int Wo, int K, int stride, int pad, int dil, void* stream);

/* backward-weight of stride-1 5x5 depthwise layers (any dilation that divides the padding): 1 (initial) a kernel that
 * stages its tiles in LDS - the prologue applied once per element, the 5x8 window of a thread read from LDS, a dilated
 * conv cut into its dil^2 pixel classes; 0: the strip kernel of the other geometries.  v < 0 only queries.  Returns
 * the previous setting.  Partial rows (nasseg_dwconv_wgrad_workspace) have the same count and layout either way. */
int nasseg_dw_wgrad_lds(int v);

/* ---- one SepConv stage in one kernel: depthwise k x k -> pointwise 1x1 (+ BN statistics) -----
 * replaces the Conv2d(C, C, k, groups=C) -> Conv2d(C, N, 1) pair of SepConv / DilConv
 * (layer_factory.py:207-218,241-262): the depthwise output tile stays in LDS and feeds the
 * matrix cores directly; zdw (optional) receives the depthwise output for the backward pass. */
int64_t nasseg_sepconv_blocks(int B, int C, int Ho, int Wo, int N, int K, int stride, int dil);
int nasseg_sepconv_fwd(const float* x, const float* wdw, const float* wpw, float* zdw, float* y,
                       const float* in_scale, const float* in_shift, int in_act,
                       const float* out_scale, const float* out_shift, int out_act, int B, int H,
                       int W, int C, int Ho, int Wo, int N, int K, int stride, int pad, int dil,
                       float* stats, void* stream);

/* ---- InvertedResidual with its expansion never stored (layer_factory.py:125-158) ---------------------------
 * The 1x1 expansion K -> C = 6 K + BatchNorm + ReLU6 in front of the 3x3 depthwise conv is six times wider than the
 * block's input; its raw output z1 = W1 x is K / 4 MFMA steps per 16 x 16 tile.  Statistics of z1: a nasseg_conv_fwd
 * call with y == NULL (nothing stored).  nasseg_irdw_fwd: z2 = dwconv3x3(act1(bn1_scale * (W1 pro(x)) + bn1_shift)),
 * pad 1, stride 1 | 2, with statistics rows of z2; nasseg_irdw_bwd: nasseg_dwconv_bwd_bn with z1 rebuilt from x.
 * w1: (C, K, 1, 1) as PyTorch stores it; wdw: packed [9][C] (nasseg_dw_pack_weight; wdw_flipped != 0: its rotated
 * packing); pro(x) = in_act(in_scale * x + in_shift) (null: none).  Rows (workgroups) of statistics [r][2][C] and of
 * weight-gradient partials [r][9][C]: nasseg_irdw_rows(.., backward); 0 = geometry not served (K % 4 == 0, K <= 32,
 * C % 16 == 0, C <= 192: MobileNetV2's 16 -> 96, 24 -> 144, 32 -> 192). */
int64_t nasseg_irdw_rows(int B, int H, int W, int K, int C, int stride, int backward);
/* Training-mode BatchNorm statistics of z1 = W1 pro(x) WITHOUT computing z1: z1 is linear in pro(x), so its first and
 * second moments per output channel follow from the K-vector and K x K matrix of moments of pro(x) (one pass over the
 * block's input).  Writes what nasseg_bn_finalize writes for the stored map (to the rounding of the sums).
 * ws: nasseg_irdw_stats_workspace(K) floats. */
int64_t nasseg_irdw_stats_workspace(int K);
int nasseg_irdw_stats(const float* x, const float* w1, const float* in_scale, const float* in_shift, int in_act, int B,
                      int H, int W, int K, int C, float eps, float momentum, const float* gamma, const float* beta,
                      float* mean, float* invstd, float* scale, float* shift, float* running_mean, float* running_var,
                      int64_t* num_batches_tracked, float* ws, void* stream);
int nasseg_irdw_fwd(const float* x, const float* w1, const float* wdw, float* z2, const float* in_scale,
                    const float* in_shift, int in_act, const float* bn1_scale, const float* bn1_shift, int act1,
                    int B, int H, int W, int K, int C, int Ho, int Wo, int stride, float* stats, void* stream);
int nasseg_irdw_bwd(const float* x, const float* w1, const float* g, const float* z2, const float* wdw,
                    int wdw_flipped, float* ge, float* dw, float* ws, const float* in_scale, const float* in_shift,
                    int in_act, const float* bn1_scale, const float* bn1_shift, const float* bn1_mean,
                    const float* bn1_invstd, int act1, const float* bn2_scale, const float* bn2_shift,
                    const float* bn2_mean, const float* bn2_invstd, const float* bn2_sums, int bn2_train, int bn2_act,
                    int B, int H, int W, int K, int C, int Ho, int Wo, int stride, float* stats, void* stream);

/* ---- dense convolution on the fp32 matrix cores ----------------------------
 * replaces conv1x1 / conv3x3 / conv_bn / conv_bn_relu (layer_factory.py:7-24,
 * 94-122), every pointwise stage (:125-382) and the classifier heads
 * (micro_decoders.py:210-227,360-363); fused input affine+act prologue and
 * output affine/bias+act(+residual) epilogue. */
int nasseg_conv_pack_weight(const float* w, float* wp, int N, int K, int kh, int kw, int mode,
                            void* stream);
/* several weights (dense and depthwise) re-packed by one launch: dims[7*i..] =
 * N, K, kh, kw, kind, Ksrc, koff; kind 0/1/2 as mode above, 3 / 4 = depthwise plain / flipped,
 * 5 = mode 1 with flipped taps (backward-data of a stride-1 conv as a forward conv over dy);
 * Ksrc > 0 packs only input channels [koff, koff+K) of a weight with Ksrc input channels */
int nasseg_pack_weights(int count, const float* const* w, float* const* wp, const int* dims,
                        void* stream);
int nasseg_conv_fwd_pack_mode(int K, int kh, int kw);
/* y == NULL with stats != NULL: only the statistics rows are produced, nothing is stored - for a pointwise conv the
 * N-split persistent kernel serves (nasseg_conv_pointwise_kernel == 2), else an error. */
int nasseg_conv_fwd(const float* x, int ldx, const float* wp, float* y, int ldy,
                    const float* in_scale, const float* in_shift, int in_act,
                    const float* out_scale, const float* out_shift, int out_act, const float* res,
                    int ldres, int B, int Hs, int Ws, int K, int Ho, int Wo, int N, int kh, int kw,
                    int stride, int pad, int dil, int transposed, float* stats, void* stream);
/* rows of statistics a call with N output channels over B*Ho*Wo output pixels writes; K = its reduction
 * channels; pointwise: 0 = any other geometry, 1 = a 1x1, stride-1, unpadded nasseg_conv_fwd call, 2 = such a
 * nasseg_conv_bwd_data_bn call - those may take the persistent pointwise kernel (weight in LDS, inputs
 * prefetched, one statistics row per workgroup of a grid that depends on K) */
int64_t nasseg_conv_fwd_stats_blocks(int B, int Ho, int Wo, int N, int K, int pointwise);
/* ... of a FORWARD call with any kernel size: nasseg_conv_fwd_stats_blocks for 1x1 and strided forms, the tile count of
 * the LDS-tiled kernel for the stride-1 3x3 forms it takes (dilation 1 ... 3, N <= 64, maps of at least 8 x 32 pixels:
 * conv3x3 / conv3x3_dil3 of the CVPR cells, layer_factory.py:56-75).  The tile is picked per map (tiles of 64 to 256
 * pixels: whole waves of the 512 workgroups that run at once, small tiles on small maps), so the count is NOT
 * B * ceil(Ho / 8) * ceil(Wo / 32) in general.  A forward call that passes `stats` for a 3x3 geometry must size them
 * with THIS query. */
int64_t nasseg_conv_fwd_stats_rows(int B, int Ho, int Wo, int N, int K, int kh, int kw, int stride, int pad, int dil);
/* tuning / testing knob: which pointwise calls take the persistent kernel.  -2 (initial): where it measured
 * faster; v >= 0: every call it supports over at least v output pixels (0 = all, a huge value = none);
 * v == -1 only queries.  Returns the previous setting.  Outputs are bit-identical either way; BatchNorm
 * statistics differ in the rounding of their partial sums.  Row counts from nasseg_conv_fwd_stats_blocks are
 * valid for the setting they were asked under. */
int64_t nasseg_conv_pw_min_pixels(int64_t v);
/* tuning / testing knob: which pointwise calls take the N-split persistent kernel (csrc/conv_pwn.hip: input streamed
 * through an LDS ring with four 16-channel blocks in flight per workgroup, the waves of a workgroup split the output
 * channels, statistics reduced across lanes once per kernel).  0: none; 1 (initial): where it measured faster;
 * 2: every call it supports (N, K multiples of 4, N <= 256, K <= 512); v < 0 only queries.  Returns the previous
 * setting.  It is asked BEFORE nasseg_conv_pw_min_pixels' kernel.  Outputs are bit-identical either way; BatchNorm
 * statistics differ in the rounding of their partial sums; row counts from nasseg_conv_fwd_stats_blocks are valid
 * for the setting they were asked under. */
int nasseg_conv_pwn_mode(int v);
/* The general MFMA kernel on maps of at most 8192 pixels (the 11 x 11 ... 21 x 21 maps of the CVPR decoders and of the
 * MobileNetV2 tail at 321 x 321): 1 (initial) four 16-channel steps of the reduction per memory round trip instead of
 * one - with a workgroup per CU or fewer there is no other wave to hide it (960 -> 160 at 16 x 11 x 11: 60 round trips,
 * 78 us for 0.6 GFLOP); 0: one step, as on large maps.  v < 0 only queries.  Returns the previous setting.
 * Bit-identical results. */
int nasseg_conv_deep_k(int v);
/* stride-1 3x3 max pooling (nasseg_maxpool_bn_fwd / _bwd) on the strip kernels - a thread owns four rows of a column
 * and loads the 6 x 3 values their windows touch once: 1 (initial) / 0.  v < 0 only queries.  Returns the previous
 * setting.  Identical outputs. */
int nasseg_pool_strip(int v);
/* which kernel a 1x1, stride-1, unpadded call with these sizes takes under the current settings: 0 the general MFMA
 * kernel, 1 the persistent kernel whose waves own all output channels, 2 the N-split persistent kernel.  pointwise
 * as for nasseg_conv_fwd_stats_blocks (1 forward, 2 backward-data).  For measurement tools (bench.py names kernel
 * families by it). */
int64_t nasseg_conv_pointwise_kernel(int B, int Ho, int Wo, int N, int K, int pointwise);
/* 1 when a plain (not transposed, no input prologue) nasseg_conv_fwd call of this geometry takes the LDS-tiled 3x3
 * kernel (conv3x3_lds_kernel): stride 1, dilation <= 3, at least one 8 x 32 tile, N <= 64; with_stats: the call asks for
 * statistics rows (served there when N % 4 == 0).  For measurement tools, like nasseg_conv_pointwise_kernel. */
int64_t nasseg_conv_fwd_lds3x3(int B, int Ho, int Wo, int N, int K, int kh, int kw, int stride, int pad, int dil,
                               int with_stats);
/* dense twin of nasseg_dwconv_bwd_data_bn (arguments as nasseg_conv_fwd, transposed) */
int nasseg_conv_bwd_data_bn(const float* dy, int lddy, const float* wp, float* g, int ldg,
                            const float* z, int ldz, const float* scale, const float* shift,
                            const float* mean, const float* invstd, int act, int B, int Hs, int Ws,
                            int K, int Ho, int Wo, int N, int kh, int kw, int stride, int pad,
                            int dil, float* stats, void* stream);
/* 1: nasseg_conv_wgrad runs this call on its LDS-tiled 3x3 kernel (stride 1, dilation <= 2, N <= 32,
 * K % 16 == 0, maps of at least 64 tiles of 8 x 32 pixels: the class heads, src/nn/micro_decoders.py:215,226,363),
 * whose summation order differs from the generic kernel that nasseg_conv_wgrad_many always uses */
int nasseg_conv_wgrad_lds3x3(int B, int Hs, int Ws, int K, int Ho, int Wo, int N, int kh, int kw, int stride,
                             int pad, int dil);
int64_t nasseg_conv_wgrad_workspace(int B, int Ho, int Wo, int N, int K, int kh, int kw);
/* weight gradients are read by the optimiser only: nasseg_conv_wgrad / nasseg_dwconv_wgrad called
 * with dw == NULL leave their per-slab partial sums in ws, and this call finalises many layers
 * with one launch per 16: dims[5*i..] = partial rows, taps, N, K, flat (depthwise: N = C, K = 1) */
int nasseg_wgrad_finalize_many(int count, const float* const* partial, float* const* dw,
                               const int* dims, void* stream);
/* first stage of `count` small layers in launches of up to 8 layers of one kernel specialisation
 * running side by side: desc[20*i..] = the arguments of nasseg_conv_wgrad from x to dil (without
 * dw), pointers as integers; equals nasseg_conv_wgrad(..., dw = NULL, ...) per layer */
int nasseg_conv_wgrad_many(int count, const int64_t* desc, void* stream);
/* the depthwise twin: desc[16*i..] = the arguments of nasseg_dwconv_wgrad from x to dil, without dw */
int nasseg_dwconv_wgrad_many(int count, const int64_t* desc, void* stream);

/* Weight gradient fused with the second half of a BatchNorm backward (replaces one nasseg_bn_bwd_apply pass
 * per layer; autograd of BatchNorm2d + Conv2d at src/nn/layer_factory.py:96-98,117-122,125-158,243-255).
 * The conv's output z went through BatchNorm (+ activation); g is the gradient w.r.t. that output with the
 * activation mask applied (bn_act == 0: as nasseg_conv_bwd_data_bn / nasseg_dwconv_bwd_data_bn leave it, or no
 * activation) or still to be applied here (bn_act != 0: g' = g * act'(scale*z + shift), g' for g below);
 * sums[2][N] = {sum g, sum g*xhat} over the M pixels (nasseg_bn_bwd_reduce / nasseg_rows_sum).  On load
 *   dz = scale*(g - sums0/M - xhat*sums1/M)   (bn_train; else dz = scale*g),  xhat = (z - mean)*invstd,
 * from which the weight gradient is computed AND which is written to dz for the backward-data call.
 * nasseg_conv_wgrad_bn: pointwise convs (1x1, stride 1), K % 4 == 0, N % 4 == 0, workspace of
 * nasseg_conv_wgrad_workspace(B,H,W,N,K,1,1).  nasseg_dwconv_wgrad_bn: nasseg_dwconv_strip_ok geometries.
 * dw == NULL: first stage only (nasseg_wgrad_finalize_many). */
int nasseg_conv_wgrad_bn(const float* x, int ldx, const float* g, int ldg, const float* z, int ldz, float* dz,
                         int lddz, float* dw, float* ws, const float* in_scale, const float* in_shift, int in_act,
                         const float* bn_scale, const float* bn_shift, const float* bn_mean, const float* bn_invstd,
                         const float* bn_sums, int bn_train, int bn_act, int B, int H, int W, int K, int N, void* stream);
/* nasseg_conv_wgrad_bn for a small-K k x k conv (kh*kw*K <= 64: nasseg_conv_fwd_pack_mode == 2 - MobileNetV2's
 * stem, the first op of its chain) whose input needs no gradient: the BatchNorm backward is applied to g on load and dz is
 * never written (instead of nasseg_bn_bwd_apply + nasseg_conv_wgrad).  Geometry as nasseg_conv_wgrad. */
int nasseg_conv_wgrad_bn_flat(const float* x, int ldx, const float* g, int ldg, const float* z, int ldz, float* dw,
                              float* ws, const float* bn_scale, const float* bn_shift, const float* bn_mean,
                              const float* bn_invstd, const float* bn_sums, int bn_train, int bn_act, int B, int Hs,
                              int Ws, int K, int Ho, int Wo, int N, int kh, int kw, int stride, int pad, int dil,
                              void* stream);
int nasseg_dwconv_wgrad_bn(const float* x, const float* g, const float* z, float* dz, float* dw, float* ws,
                           const float* in_scale, const float* in_shift, int in_act, const float* bn_scale, const float* bn_shift,
                           const float* bn_mean, const float* bn_invstd, const float* bn_sums, int bn_train, int bn_act,
                           int B, int H, int W, int C, int Ho, int Wo, int K, int stride, int pad, int dil,
                           void* stream);
/* The whole backward of a pointwise conv followed by BatchNorm in one kernel: BatchNorm backward on load (as
 * nasseg_conv_wgrad_bn), weight gradient AND input gradient; dz never reaches HBM.  wb = the weight packed for
 * backward-data ([K][N], pack mode 1); dx [B][H][W][K]; ws: nasseg_conv_pw_bwd_slabs(...) * N * K floats (0 slabs:
 * no fused kernel for these channel counts - N*K <= 6144 with K <= 64, or N <= 64 with K <= 384); dx_act != 0
 * (= in_act): dx additionally multiplied by in_act'(in_scale*x + in_shift), i.e. the gradient w.r.t. the affine's
 * output (w.r.t. x itself for a bare activation); dw == NULL: partial rows only.  dx_stats != NULL (K <= 64; x is the
 * raw output of a BatchNorm with statistics in_mean / in_invstd, dx_act = in_act): also that BatchNorm's backward
 * sums {sum dx, sum dx*xhat} per slab, rows [slabs][2][K] for nasseg_rows_sum (buffer: slabs + 64 rows).
 * dx_res != NULL (K % 4 == 0, no dx_stats): a [P][K] map added to dx after the mask - the gradient of a skip connection
 * that x feeds as well (InvertedResidual, src/nn/layer_factory.py:276-321: autograd then has nothing to accumulate). */
int64_t nasseg_conv_pw_bwd_slabs(int B, int H, int W, int K, int N);
/* 1: nasseg_conv_pw_bwd_bn loads z for this geometry; 0: it rebuilds z = W x from the input tile it stages anyway (the
 * narrow kernel with its weight in LDS: same operand mapping and accumulation order as the forward kernels, the same
 * bits) and z is not read.  A z that is passed must BE the conv's raw output; z == NULL says it was never stored
 * (nasseg_irdw_fwd) and forces the rebuild - an error where no kernel can (K > 32, N > 144).  For measurement tools
 * and tests. */
int64_t nasseg_conv_pw_bwd_reads_z(int B, int H, int W, int K, int N);
/* pixels from which nasseg_conv_pw_bwd_bn rebuilds z (where it can: K <= 32, N <= 96): 2^18 initially; 0: every
 * supported geometry, a huge value: none.  v < 0 only queries.  Returns the previous setting. */
int64_t nasseg_conv_pw_bwd_rz_min_pixels(int64_t v);
int nasseg_conv_pw_bwd_bn(const float* x, const float* g, const float* z, const float* wb, float* dx, float* dw,
                          float* ws, const float* in_scale, const float* in_shift, int in_act, int dx_act,
                          const float* bn_scale, const float* bn_shift, const float* bn_mean, const float* bn_invstd,
                          const float* bn_sums, int bn_train, int bn_act, int B, int H, int W, int K, int N,
                          const float* in_mean, const float* in_invstd, float* dx_stats, const float* dx_res,
                          void* stream);
int nasseg_conv_wgrad(const float* x, int ldx, const float* dy, int lddy, float* dw, float* ws,
                      const float* in_scale, const float* in_shift, int in_act, int B, int Hs,
                      int Ws, int K, int Ho, int Wo, int N, int kh, int kw, int stride, int pad,
                      int dil, void* stream);

/* ---- per-channel reductions / BatchNorm ------------------------------------
 * replaces nn.BatchNorm2d (layer_factory.py:56-75,94-158,369-382), x.mean(2).mean(3)
 * in GAPConv1x1 (:181-195), bias / ParamSum coefficient gradients (:353-366). */
int64_t nasseg_colred_workspace(int S, int64_t R, int C);
int nasseg_colred(int mode, const float* a, int64_t lda, const float* b, int64_t ldb,
                  const float* c, int64_t ldc, float* out, float* ws, int S, int64_t R, int C,
                  float mul, void* stream);
int nasseg_bn_stats(const float* x, int64_t ldx, int64_t M, int C, float eps, float momentum,
                    const float* gamma, const float* beta, float* mean, float* invstd,
                    float* scale, float* shift, float* running_mean, float* running_var,
                    int64_t* num_batches_tracked, float* ws, void* stream);
/* partial: rows [nblk, nblk + 64) of the buffer are scratch of the two-level reduction (hence not const) */
int nasseg_bn_finalize(float* partial, int nblk, int64_t M, int C, float eps, float momentum,
                       const float* gamma, const float* beta, float* mean, float* invstd,
                       float* scale, float* shift, float* running_mean, float* running_var,
                       int64_t* num_batches_tracked, void* stream);
int nasseg_bn_eval_params(int C, float eps, const float* gamma, const float* beta,
                          const float* running_mean, const float* running_var, float* mean,
                          float* invstd, float* scale, float* shift, void* stream);
int nasseg_bn_bwd_reduce(const float* dy, int64_t lddy, const float* x, int64_t ldx, int64_t M,
                         int C, const float* scale, const float* shift, const float* mean,
                         const float* invstd, int act, float* sums, float* ws, void* stream);
/* sums[cols] = sum of the rows of partial[nblk][cols] (buffer needs nblk + 64 rows: scratch of the two-level
 * reduction, written - hence not const) */
int nasseg_rows_sum(float* partial, int nblk, int cols, float* out, void* stream);
int nasseg_bn_bwd_apply(const float* dy, const float* x, const float* scale, const float* shift,
                        const float* mean, const float* invstd, const float* sums, int64_t M,
                        int C, int train, int act, float* dx, void* stream);
/* BatchNorm backward whose consumer adds up the partial rows of the sums itself (the small maps of the CVPR cells,
 * micro_decoders.py:54-121: a row-summing launch in front of every BatchNorm backward costs a dependent ~5 us).
 * nasseg_bn_bwd_reduce_rows: the first stage of nasseg_bn_bwd_reduce only - rows [nasseg_colred_rows(1, M, C)][2][C] in
 * a buffer of nasseg_colred_workspace(1, M, C) floats.  nasseg_bn_bwd_apply_rows: nasseg_bn_bwd_apply from such rows
 * (or the statistics rows of a fused backward-data kernel): every workgroup adds them in fp64 in a fixed order; sums_out
 * (null or [2][C]) receives {sum g, sum g*xhat}.  Meant for nrows * 2 * C * 4 <= nasseg_bn_bwd_apply_rows_max_bytes(). */
int64_t nasseg_colred_rows(int S, int64_t R, int C);
int64_t nasseg_bn_bwd_apply_rows_max_bytes(void);
int nasseg_bn_bwd_reduce_rows(const float* dy, int64_t lddy, const float* x, int64_t ldx, int64_t M, int C,
                              const float* scale, const float* shift, const float* mean, const float* invstd, int act,
                              float* rows, void* stream);
int nasseg_bn_bwd_apply_rows(const float* dy, const float* x, const float* scale, const float* shift, const float* mean,
                             const float* invstd, const float* rows, int nrows, float* sums_out, int64_t M, int C,
                             int train, int act, float* dx, void* stream);

/* ---- elementwise / copies ----------------------------------------------------
 * BN apply + ReLU/ReLU6 + residual (layer_factory.py:94-158), cell sums
 * (micro_decoders.py:48-51,110-121), ParamSum (:353-366), Skip / Zero (:268-297),
 * torch.cat (+F.relu) (micro_decoders.py:11-25,251; layer_factory.py:369-382). */
int nasseg_affine_act(const float* x, const float* scale, const float* shift, const float* res,
                      float* y, int64_t n, int C, int act, void* stream);
int nasseg_axpby(const float* a, const float* b, const float* alpha, const float* beta, float* y,
                 int64_t n, int C, int act, void* stream);
/* y = ca[c] * act_a(xa*sa[c] + ha[c]) + cb[c] * act_b(xb*sb[c] + hb[c]): the sum of two op outputs (cell sums,
 * micro_decoders.py:48-51,110-121; ParamSum's coefficients, layer_factory.py:353-366) whose last BatchNorm +
 * activation is applied as they are loaded - the normalised maps are never written.  Null vectors: scale /
 * coefficient 1, shift 0. */
int nasseg_add_act2(const float* xa, const float* sa, const float* ha, int act_a, const float* ca, const float* xb,
                    const float* sb, const float* hb, int act_b, const float* cb, float* y, int64_t n, int C,
                    void* stream);
int nasseg_act_bwd(const float* dy, const float* ref, float* dx, int64_t n, int act, void* stream);
int nasseg_fill(float* y, int64_t n, float v, void* stream);
int nasseg_chan_copy(const float* x, int64_t ldx, int xoff, float* y, int64_t ldy, int yoff,
                     const float* mref, int64_t ldm, int moff, int64_t P, int C, int act, int mact,
                     void* stream);
int nasseg_chan_fold(const float* dy, float* dx, int64_t P, int C, int rep, void* stream);
/* dst[i] = src[idx[i]], rows of row_bytes bytes, idx an int64 DEVICE array clamped to [0, n_src):
 * the batch of the task0 feature cache (Xy_train[k][indices], src/engine/trainer.py:128-137) */
int nasseg_gather_rows(const void* src, const int64_t* idx, void* dst, int n, int64_t row_bytes,
                       int64_t n_src, void* stream);

/* ---- pooling: Pool (layer_factory.py:161-178); mode 0 max, 1 avg ------------- */
int nasseg_pool_fwd(int mode, const float* x, float* y, uint8_t* idx, int B, int H, int W, int C,
                    int Ho, int Wo, int K, int stride, int pad, void* stream);
int nasseg_pool_bwd(int mode, const float* dy, const uint8_t* idx, float* dx, int B, int H, int W,
                    int C, int Ho, int Wo, int K, int stride, int pad, void* stream);
/* Pool's 1x1 conv + BatchNorm followed by 3x3 max pooling (src/nn/layer_factory.py:161-178) without the
 * normalised map: the pooling applies scale*z + shift to the conv's raw output as it loads; backward gives
 * the gradient w.r.t. the BatchNorm's output together with the per-workgroup sums of its backward. */
int64_t nasseg_maxpool_bn_bwd_blocks(int B, int H, int W, int C, int K, int stride, int pad);
int nasseg_maxpool_bn_fwd(const float* z, const float* scale, const float* shift, float* y, uint8_t* idx,
                          int B, int H, int W, int C, int Ho, int Wo, int stride, int pad, void* stream);
int nasseg_maxpool_bn_bwd(const float* dy, const uint8_t* idx, const float* z, const float* mean,
                          const float* invstd, float* g, float* stats, int B, int H, int W, int C, int Ho,
                          int Wo, int stride, int pad, void* stream);

/* ---- resize: nn.Upsample / F.interpolate bilinear, align_corners=False
 * (layer_factory.py:190-194,338-350; micro_decoders.py:11-25,46-51; trainer.py:141-143,
 * 236-238,245-247; inference.py:58-60) and nearest label resize (trainer.py:43-49,236-238). */
/* One input of ConcatReduce's torch.cat (src/nn/layer_factory.py:369-382, after Adapt's resize :316-350)
 * written into its channel slice of the slab: resized when its size differs, the producer's still pending
 * BatchNorm + activation applied as the source is loaded, and the slab's own BatchNorm statistics emitted
 * as per-workgroup rows for nasseg_bn_finalize (the slab is not read again for them).  Backward
 * (nasseg_cat_src_bwd): the slab BatchNorm's backward applied to one input's slice, masked with the pending
 * activation's derivative, with the producer's BatchNorm-backward sums as per-workgroup rows. */
int64_t nasseg_cat_src_blocks(int B, int Ho, int Wo, int C);
/* Backward of ParamSum (a[c]*x + b[c]*y, layer_factory.py:353-366) for both operands from one read of the gradient:
 * an operand may be a conv chain's raw output with its BatchNorm + activation pending (ts*: mean | invstd | scale |
 * shift; null: a finished map).  g* = c* dy act'(...) - masked, with the producer's BatchNorm-backward sums as rows
 * part* [nasseg_cat_src_blocks(B, H, W, C) + 64][2][C] - and cpart [blocks + 64][2][C]: rows of the coefficient
 * gradients {sum dy*x, sum dy*y} (nasseg_rows_sum finishes them). */
int nasseg_psum_bwd(const float* dy, const float* za, const float* tsa, int act_a, const float* ca, float* ga,
                    float* part_a, const float* zb, const float* tsb, int act_b, const float* cb, float* gb,
                    float* part_b, float* cpart, int B, int H, int W, int C, void* stream);
/* Gradient junction of a node with several consumers (a cell's node read by several ops, a block's output read by
 * later blocks and collect_all: micro_decoders.py:95-121,380-398; torch's autograd adds such gradients pairwise, one
 * launch and three tensor passes per add): out = g0 + g1 + ... + g(n-1), 1 <= n <= 8, added in that order, all dense
 * [B*H*W][C]; unused pointers null.  With z / tstats (the node is a conv chain's raw output whose BatchNorm + activation
 * is pending; tstats: mean | invstd | scale | shift) out is also multiplied by act'(scale*z + shift) and part
 * [nasseg_cat_src_blocks(B, H, W, C) + 64][2][C] (null: none) receives that BatchNorm's backward sums {sum out,
 * sum out * xhat} as rows. */
int nasseg_grad_junction(const float* g0, const float* g1, const float* g2, const float* g3, const float* g4,
                         const float* g5, const float* g6, const float* g7, int n, const float* z, const float* tstats,
                         int act, float* out, float* part, int B, int H, int W, int C, void* stream);
int nasseg_cat_src_fwd(const float* x, const float* scale, const float* shift, int act, float* y, int64_t ldy,
                       int yoff, float* stats, int B, int Hi, int Wi, int C, int Ho, int Wo, void* stream);
int nasseg_cat_src_bwd(const float* du, const float* slab, int64_t ld, int off, const float* sscale,
                       const float* smean, const float* sinvstd, const float* sums, int train, const float* z,
                       const float* tstats, int act, float* g, float* part, int B, int Ho, int Wo, int C, int Hi,
                       int Wi, void* stream);
/* nasseg_bilinear_bwd with the result multiplied by act'(scale*z + shift) at the source's size: the gradient of a
 * resized pending input, masked for its producer (its sums come from nasseg_cat_src_bwd). */
int nasseg_bilinear_bwd_act(const float* dy, int64_t lddy, int dyoff, const float* z, const float* scale,
                            const float* shift, int act, float* dx, int B, int Hi, int Wi, int C, int Ho, int Wo,
                            float* ws, void* stream);
int nasseg_bilinear_fwd(const float* x, float* y, int64_t ldy, int yoff, int B, int Hi, int Wi,
                        int C, int Ho, int Wo, int act, void* stream);
/* nn.Upsample(size, mode="bilinear", align_corners=True) - src/kd/rf_lw/model_lw_v2.py:258,266,274 (the
 * distillation teacher's decoder); forward only, C % 4 == 0, dense output [B][Ho][Wo][C] */
int nasseg_bilinear_ac_fwd(const float* x, float* y, int B, int Hi, int Wi, int C, int Ho, int Wo,
                           void* stream);
int64_t nasseg_bilinear_bwd_workspace(int B, int Hi, int Wi, int C, int Ho, int Wo);
int nasseg_bilinear_bwd(const float* dy, int64_t lddy, int dyoff, float* dx, int B, int Hi, int Wi,
                        int C, int Ho, int Wo, float* ws, void* stream);
int nasseg_nearest_label(const void* x, int elem_size, int64_t* y, int B, int Hi, int Wi, int Ho,
                         int Wo, void* stream);

/* ---- loss: nn.LogSoftmax() + nn.NLLLoss2d(ignore_index=255)
 * (main_search.py:435; trainer.py:144-146,239-241) -------------------------------- */
int64_t nasseg_ce_workspace(void);
int nasseg_ce_fwd(const float* logits, const void* target, int elem_size, int64_t P, int C,
                  int ignore, float* out, float* ws, void* stream);
int nasseg_ce_bwd(const float* logits, const void* target, int elem_size, const float* stats,
                  const float* gscale, int64_t P, int C, int ignore, float* dlogits, void* stream);

/* berHu loss of the depth head (BASELINE config 5; absent from the reference - Laina et al.
 * 2016 eq. 2, "parity unpinned") */
int nasseg_berhu_fwd(const float* pred, const float* target, int64_t n, float* out, float* ws,
                     void* stream);
int nasseg_berhu_bwd(const float* pred, const float* target, const float* stats,
                     const float* gscale, int64_t n, float* dpred, void* stream);

/* ---- mean-IoU reward: helpers/miou_utils.pyx fast_cm :7-30, compute_iu :32-57,
 * compute_ius_accs :59-90; argmax + up-sampling of engine/inference.py:58-66 ------ */
int nasseg_fast_cm(const uint8_t* preds, const uint8_t* gt, int64_t P, int n, int64_t* cm,
                   void* stream);
int nasseg_argmax_cm(const float* logits, const uint8_t* gt, uint8_t* preds, int B, int h, int w,
                     int C, int H, int W, int n, int64_t* cm, void* stream);
/* host-side (cm, iu, n_pixels, accs are HOST pointers) */
int nasseg_compute_ius_accs(const int64_t* cm, int n, double* iu, int64_t* n_pixels, double* accs);


/* ---- clip_grad_norm_ + optimiser steps: src/engine/trainer.py:163-166,258-268 on the torch.optim.SGD /
 * torch.optim.Adam objects of src/utils/solvers.py:6-52 - two launches for every parameter of the step.
 *   tensors  DEVICE int64 [n_tensors][8]: parameter, gradient, state 1 (SGD momentum_buffer | Adam exp_avg; 0: SGD
 *            without momentum), state 2 (Adam exp_avg_sq; else 0) addresses, numel, clip set (-1: none), hyper
 *            group, flags (bit 0: all four addresses are 16-byte aligned).  fp32, dense, gradient laid out like
 *            the parameter.
 *   chunks   DEVICE int32 [n_chunks][2] = {tensor, first element}: nasseg_optim_chunk() elements each; the chunks of
 *            one clip set are consecutive.
 *   hyper    HOST double [n_hyper][6] = {kind (0 SGD, 1 Adam), lr, weight_decay, momentum | beta1, beta2, eps};
 *            SGD: no nesterov, no dampening; Adam: no amsgrad; neither maximises.  n_hyper <= 8.
 *   clips    HOST double [n_clip][3] = {max_norm, first chunk, chunk count}, n_clip <= 8: L2 norm over the set,
 *            gradients scaled in place by min(1, max_norm / (norm + 1e-6)); norms[n_clip] receives the norms.
 *   dstep    DEVICE float [n_tensors]: steps taken per tensor, advanced by the call (Adam's bias correction reads
 *            it on the device: the call can be replayed from a hipGraph).
 *   partial  DEVICE double [n_chunks] workspace. */
int64_t nasseg_optim_chunk(void);
int nasseg_optim_step(const int64_t* tensors, int n_tensors, const int* chunks, int n_chunks, const double* hyper,
                      int n_hyper, const double* clips, int n_clip, float* dstep, double* partial, float* norms,
                      void* stream);

/* ---- hipGraph scheduling of a captured step (reference src/nn/micro_decoders.py:54-139: the five
 * ops of a ContextualCell read one input, the two cells of a MergeCell share nothing) ---------------------------------
 * A step recorded from one stream is a line of nodes.  The host side (engine/graph_dag.py) knows from this header
 * which pointers every entry point reads (const) and writes (non-const) and with which address ranges it was called;
 * these host-only calls let it attribute the recorded nodes to calls and replace the line's edges by the real
 * dependencies (kept in a few lanes), so that independent branches replay side by side.
 *   nasseg_graph_capture_nodes: nodes recorded so far by the capture `stream` belongs to; -1: not capturing.
 *   nasseg_graph_node_kinds:    kinds[i] of the i-th recorded node: 0 kernel, 1 memcpy, 2 memset, 3 other.
 *   nasseg_graph_rewire:        graph = hipGraph_t recorded from ONE stream with n_nodes nodes; its edges are replaced
 *                               by edges[2*e] -> edges[2*e+1] (indices in recording order, pointing forward).  The
 *                               caller guarantees they cover every read/write hazard of the recorded launches. */
int nasseg_graph_capture_nodes(void* stream);
int nasseg_graph_node_kinds(void* graph, int n_nodes, int* kinds);
int nasseg_graph_rewire(void* graph, int n_nodes, int n_edges, const int* edges);
/* Stages and lanes.  A dependency between two branches INSIDE one hipGraph costs this runtime about as much as a small
 * kernel, and a graph with branches is no longer replayed from pre-built packets; so the host side cuts the recorded
 * line into STAGES whose LANES share nothing, every (stage, lane) a line graph of its own, and orders them with events:
 *   nasseg_graph_split:  part[i] = the part of the i-th recorded node (kernel and memset nodes only, else
 *                        NASSEG_ERR_UNSUPPORTED); execs[p] receives a hipGraphExec_t holding part p's nodes in
 *                        recording order (0: empty part).  The recorded graph is not modified.
 *   nasseg_graph_run:    one replay - ops[3*i..] = {kind, a, b}: 0 launch executable graph a on stream b, 1 record
 *                        event a on stream b, 2 stream b waits for event a; b == 0 means `stream`.  Asynchronous.
 *   nasseg_lane_*:       the non-blocking streams / timing-free events those ops name (handles as void*). */
int nasseg_graph_split(void* graph, int n_nodes, const int* part, int n_parts, void** execs);
int nasseg_graph_exec_destroy(void* exec);
int nasseg_lane_stream_create(void** stream);
int nasseg_lane_event_create(void** event);
int nasseg_lane_destroy(void* stream, void* event);
int nasseg_graph_run(int n_ops, const int64_t* ops, void* stream);

/* ---- bfloat16 activation storage --------------------------------------------
 * Every entry point above that reads or writes ACTIVATIONS (feature maps and their gradients)
 * has a twin nasseg_bf16_<op> with the same arguments in which those tensors are stored as
 * bfloat16 (BASELINE config 5).  Only storage changes: values are widened to fp32 on load and
 * rounded to nearest-even on store; arithmetic, MFMA accumulation, BatchNorm statistics,
 * parameters, parameter gradients, workspaces and every per-channel vector stay fp32, and the
 * size / workspace queries are shared with the fp32 entry points. */
typedef uint16_t nasseg_bf16_t;
/* fp32 -> bf16 (round to nearest even) / bf16 -> fp32 of n values: the (B, C, 1, 1) maps on either side of
 * GAPConv1x1's fp32 island (layer_factory.py:181-195) in a bf16-storage network */
int nasseg_to_bf16(const float* x, nasseg_bf16_t* y, int64_t n, void* stream);
int nasseg_from_bf16(const nasseg_bf16_t* x, float* y, int64_t n, void* stream);
int nasseg_bf16_irdw_stats(const nasseg_bf16_t* x, const float* w1, const float* in_scale, const float* in_shift,
                           int in_act, int B, int H, int W, int K, int C, float eps, float momentum, const float* gamma,
                           const float* beta, float* mean, float* invstd, float* scale, float* shift,
                           float* running_mean, float* running_var, int64_t* num_batches_tracked, float* ws,
                           void* stream);
int nasseg_bf16_irdw_fwd(const nasseg_bf16_t* x, const float* w1, const float* wdw, nasseg_bf16_t* z2,
                         const float* in_scale, const float* in_shift, int in_act, const float* bn1_scale,
                         const float* bn1_shift, int act1, int B, int H, int W, int K, int C, int Ho, int Wo, int stride,
                         float* stats, void* stream);
int nasseg_bf16_irdw_bwd(const nasseg_bf16_t* x, const float* w1, const nasseg_bf16_t* g, const nasseg_bf16_t* z2,
                         const float* wdw, int wdw_flipped, nasseg_bf16_t* ge, float* dw, float* ws,
                         const float* in_scale, const float* in_shift, int in_act, const float* bn1_scale,
                         const float* bn1_shift, const float* bn1_mean, const float* bn1_invstd, int act1,
                         const float* bn2_scale, const float* bn2_shift, const float* bn2_mean,
                         const float* bn2_invstd, const float* bn2_sums, int bn2_train, int bn2_act, int B, int H, int W,
                         int K, int C, int Ho, int Wo, int stride, float* stats, void* stream);
int nasseg_bf16_dwconv_bwd_bn(const nasseg_bf16_t* xz, const nasseg_bf16_t* g, const nasseg_bf16_t* z, const float* wt,
                              int wt_flipped, nasseg_bf16_t* ge, float* dw, float* ws, const float* in_scale, const float* in_shift,
                              const float* in_mean, const float* in_invstd, int in_act, const float* bn_scale,
                              const float* bn_shift, const float* bn_mean, const float* bn_invstd, const float* bn_sums,
                              int bn_train, int bn_act, int B, int H, int W, int C, int Ho, int Wo, int K, int stride,
                              int pad, int dil, float* stats, void* stream);
int nasseg_bf16_conv_pw_bwd_bn(const nasseg_bf16_t* x, const nasseg_bf16_t* g, const nasseg_bf16_t* z, const float* wb,
                               nasseg_bf16_t* dx, float* dw, float* ws, const float* in_scale, const float* in_shift,
                               int in_act, int dx_act, const float* bn_scale, const float* bn_shift, const float* bn_mean,
                               const float* bn_invstd, const float* bn_sums, int bn_train, int bn_act, int B, int H,
                               int W, int K, int N, const float* in_mean, const float* in_invstd, float* dx_stats,
                               const nasseg_bf16_t* dx_res, void* stream);
int nasseg_bf16_sepconv_fwd(const nasseg_bf16_t* x, const float* wdw, const float* wpw, nasseg_bf16_t* zdw,
                            nasseg_bf16_t* y, const float* in_scale, const float* in_shift, int in_act,
                            const float* out_scale, const float* out_shift, int out_act, int B, int H, int W, int C,
                            int Ho, int Wo, int N, int K, int stride, int pad, int dil, float* stats, void* stream);
int nasseg_bf16_affine_act(const nasseg_bf16_t* x, const float* scale, const float* shift, const nasseg_bf16_t* res,
                      nasseg_bf16_t* y, int64_t n, int C, int act, void* stream);
int nasseg_bf16_psum_bwd(const nasseg_bf16_t* dy, const nasseg_bf16_t* za, const float* tsa, int act_a, const float* ca,
                         nasseg_bf16_t* ga, float* part_a, const nasseg_bf16_t* zb, const float* tsb, int act_b,
                         const float* cb, nasseg_bf16_t* gb, float* part_b, float* cpart, int B, int H, int W, int C,
                         void* stream);
int nasseg_bf16_add_act2(const nasseg_bf16_t* xa, const float* sa, const float* ha, int act_a, const float* ca,
                         const nasseg_bf16_t* xb, const float* sb, const float* hb, int act_b, const float* cb,
                         nasseg_bf16_t* y, int64_t n, int C, void* stream);
int nasseg_bf16_bn_bwd_apply(const nasseg_bf16_t* dy, const nasseg_bf16_t* x, const float* scale, const float* shift,
                        const float* mean, const float* invstd, const float* sums, int64_t M,
                        int C, int train, int act, nasseg_bf16_t* dx, void* stream);
int nasseg_bf16_bn_bwd_reduce_rows(const nasseg_bf16_t* dy, int64_t lddy, const nasseg_bf16_t* x, int64_t ldx, int64_t M,
                                   int C, const float* scale, const float* shift, const float* mean, const float* invstd,
                                   int act, float* rows, void* stream);
int nasseg_bf16_bn_bwd_apply_rows(const nasseg_bf16_t* dy, const nasseg_bf16_t* x, const float* scale, const float* shift,
                                  const float* mean, const float* invstd, const float* rows, int nrows, float* sums_out,
                                  int64_t M, int C, int train, int act, nasseg_bf16_t* dx, void* stream);
int nasseg_bf16_grad_junction(const nasseg_bf16_t* g0, const nasseg_bf16_t* g1, const nasseg_bf16_t* g2,
                              const nasseg_bf16_t* g3, const nasseg_bf16_t* g4, const nasseg_bf16_t* g5,
                              const nasseg_bf16_t* g6, const nasseg_bf16_t* g7, int n, const nasseg_bf16_t* z,
                              const float* tstats, int act, nasseg_bf16_t* out, float* part, int B, int H, int W, int C,
                              void* stream);
int nasseg_bf16_axpby(const nasseg_bf16_t* a, const nasseg_bf16_t* b, const float* alpha, const float* beta, nasseg_bf16_t* y,
                 int64_t n, int C, int act, void* stream);
int nasseg_bf16_act_bwd(const nasseg_bf16_t* dy, const nasseg_bf16_t* ref, nasseg_bf16_t* dx, int64_t n, int act, void* stream);
int nasseg_bf16_fill(nasseg_bf16_t* y, int64_t n, float v, void* stream);
int nasseg_bf16_chan_copy(const nasseg_bf16_t* x, int64_t ldx, int xoff, nasseg_bf16_t* y, int64_t ldy, int yoff,
                     const nasseg_bf16_t* mref, int64_t ldm, int moff, int64_t P, int C, int act, int mact,
                     void* stream);
int nasseg_bf16_chan_fold(const nasseg_bf16_t* dy, nasseg_bf16_t* dx, int64_t P, int C, int rep, void* stream);
int nasseg_bf16_pool_fwd(int mode, const nasseg_bf16_t* x, nasseg_bf16_t* y, uint8_t* idx, int B, int H, int W, int C,
                    int Ho, int Wo, int K, int stride, int pad, void* stream);
int nasseg_bf16_pool_bwd(int mode, const nasseg_bf16_t* dy, const uint8_t* idx, nasseg_bf16_t* dx, int B, int H, int W,
                    int C, int Ho, int Wo, int K, int stride, int pad, void* stream);
int nasseg_bf16_bilinear_fwd(const nasseg_bf16_t* x, nasseg_bf16_t* y, int64_t ldy, int yoff, int B, int Hi, int Wi,
                        int C, int Ho, int Wo, int act, void* stream);
int nasseg_bf16_maxpool_bn_fwd(const nasseg_bf16_t* z, const float* scale, const float* shift, nasseg_bf16_t* y, uint8_t* idx,
                               int B, int H, int W, int C, int Ho, int Wo, int stride, int pad, void* stream);
int nasseg_bf16_maxpool_bn_bwd(const nasseg_bf16_t* dy, const uint8_t* idx, const nasseg_bf16_t* z, const float* mean,
                               const float* invstd, nasseg_bf16_t* g, float* stats, int B, int H, int W, int C, int Ho,
                               int Wo, int stride, int pad, void* stream);
int nasseg_bf16_cat_src_fwd(const nasseg_bf16_t* x, const float* scale, const float* shift, int act, nasseg_bf16_t* y,
                            int64_t ldy, int yoff, float* stats, int B, int Hi, int Wi, int C, int Ho, int Wo,
                            void* stream);
int nasseg_bf16_cat_src_bwd(const nasseg_bf16_t* du, const nasseg_bf16_t* slab, int64_t ld, int off, const float* sscale,
                            const float* smean, const float* sinvstd, const float* sums, int train,
                            const nasseg_bf16_t* z, const float* tstats, int act, nasseg_bf16_t* g, float* part, int B,
                            int Ho, int Wo, int C, int Hi, int Wi, void* stream);
int nasseg_bf16_bilinear_bwd_act(const nasseg_bf16_t* dy, int64_t lddy, int dyoff, const nasseg_bf16_t* z,
                                 const float* scale, const float* shift, int act, nasseg_bf16_t* dx, int B, int Hi, int Wi,
                                 int C, int Ho, int Wo, float* ws, void* stream);
int nasseg_bf16_bilinear_ac_fwd(const nasseg_bf16_t* x, nasseg_bf16_t* y, int B, int Hi, int Wi, int C, int Ho, int Wo,
                                void* stream);
int nasseg_bf16_bilinear_bwd(const nasseg_bf16_t* dy, int64_t lddy, int dyoff, nasseg_bf16_t* dx, int B, int Hi, int Wi,
                        int C, int Ho, int Wo, float* ws, void* stream);
int nasseg_bf16_ce_fwd(const nasseg_bf16_t* logits, const void* target, int elem_size, int64_t P, int C,
                  int ignore, float* out, float* ws, void* stream);
int nasseg_bf16_ce_bwd(const nasseg_bf16_t* logits, const void* target, int elem_size, const float* stats,
                  const float* gscale, int64_t P, int C, int ignore, nasseg_bf16_t* dlogits, void* stream);
int nasseg_bf16_berhu_fwd(const nasseg_bf16_t* pred, const nasseg_bf16_t* target, int64_t n, float* out, float* ws,
                     void* stream);
int nasseg_bf16_berhu_bwd(const nasseg_bf16_t* pred, const nasseg_bf16_t* target, const float* stats,
                     const float* gscale, int64_t n, nasseg_bf16_t* dpred, void* stream);
int nasseg_bf16_colred(int mode, const nasseg_bf16_t* a, int64_t lda, const nasseg_bf16_t* b, int64_t ldb,
                  const nasseg_bf16_t* c, int64_t ldc, float* out, float* ws, int S, int64_t R, int C,
                  float mul, void* stream);
int nasseg_bf16_bn_stats(const nasseg_bf16_t* x, int64_t ldx, int64_t M, int C, float eps, float momentum,
                    const float* gamma, const float* beta, float* mean, float* invstd,
                    float* scale, float* shift, float* running_mean, float* running_var,
                    int64_t* num_batches_tracked, float* ws, void* stream);
int nasseg_bf16_bn_bwd_reduce(const nasseg_bf16_t* dy, int64_t lddy, const nasseg_bf16_t* x, int64_t ldx, int64_t M,
                         int C, const float* scale, const float* shift, const float* mean,
                         const float* invstd, int act, float* sums, float* ws, void* stream);
int nasseg_bf16_dwconv(const nasseg_bf16_t* x, const float* wt, nasseg_bf16_t* y, const float* in_scale,
                  const float* in_shift, int in_act, const float* scale, const float* shift, int act,
                  int B, int H, int W, int C, int Ho, int Wo, int K, int stride, int pad, int dil,
                  int transposed, float* stats, void* stream);
int nasseg_bf16_dwconv_bwd_data_bn(const nasseg_bf16_t* dy, const float* wt, nasseg_bf16_t* g, const nasseg_bf16_t* z,
                              const float* scale, const float* shift, const float* mean,
                              const float* invstd, int act, int B, int H, int W, int C, int Ho,
                              int Wo, int K, int stride, int pad, int dil, int transposed,
                              float* stats, void* stream);
int nasseg_bf16_dwconv_wgrad(const nasseg_bf16_t* x, const nasseg_bf16_t* dy, float* dw, float* ws,
                        const float* in_scale, const float* in_shift, int in_act, int B, int H, int W,
                        int C, int Ho, int Wo, int K, int stride, int pad, int dil, void* stream);
int nasseg_bf16_conv_fwd(const nasseg_bf16_t* x, int ldx, const float* wp, nasseg_bf16_t* y, int ldy,
                    const float* in_scale, const float* in_shift, int in_act,
                    const float* out_scale, const float* out_shift, int out_act, const nasseg_bf16_t* res,
                    int ldres, int B, int Hs, int Ws, int K, int Ho, int Wo, int N, int kh, int kw,
                    int stride, int pad, int dil, int transposed, float* stats, void* stream);
int nasseg_bf16_conv_bwd_data_bn(const nasseg_bf16_t* dy, int lddy, const float* wp, nasseg_bf16_t* g, int ldg,
                            const nasseg_bf16_t* z, int ldz, const float* scale, const float* shift,
                            const float* mean, const float* invstd, int act, int B, int Hs, int Ws,
                            int K, int Ho, int Wo, int N, int kh, int kw, int stride, int pad,
                            int dil, float* stats, void* stream);
int nasseg_bf16_conv_wgrad_many(int count, const int64_t* desc, void* stream);
int nasseg_bf16_dwconv_wgrad_many(int count, const int64_t* desc, void* stream);
int nasseg_bf16_conv_wgrad_bn(const nasseg_bf16_t* x, int ldx, const nasseg_bf16_t* g, int ldg,
                              const nasseg_bf16_t* z, int ldz, nasseg_bf16_t* dz, int lddz, float* dw, float* ws,
                              const float* in_scale, const float* in_shift, int in_act, const float* bn_scale, const float* bn_shift,
                              const float* bn_mean, const float* bn_invstd, const float* bn_sums, int bn_train, int bn_act,
                              int B, int H, int W, int K, int N, void* stream);
int nasseg_bf16_conv_wgrad_bn_flat(const nasseg_bf16_t* x, int ldx, const nasseg_bf16_t* g, int ldg,
                                   const nasseg_bf16_t* z, int ldz, float* dw, float* ws, const float* bn_scale,
                                   const float* bn_shift, const float* bn_mean, const float* bn_invstd,
                                   const float* bn_sums, int bn_train, int bn_act, int B, int Hs, int Ws, int K, int Ho,
                                   int Wo, int N, int kh, int kw, int stride, int pad, int dil, void* stream);
int nasseg_bf16_dwconv_wgrad_bn(const nasseg_bf16_t* x, const nasseg_bf16_t* g, const nasseg_bf16_t* z,
                                nasseg_bf16_t* dz, float* dw, float* ws, const float* in_scale,
                                const float* in_shift, int in_act, const float* bn_scale, const float* bn_shift, const float* bn_mean,
                                const float* bn_invstd, const float* bn_sums, int bn_train, int bn_act, int B, int H, int W,
                                int C, int Ho, int Wo, int K, int stride, int pad, int dil, void* stream);
int nasseg_bf16_conv_wgrad(const nasseg_bf16_t* x, int ldx, const nasseg_bf16_t* dy, int lddy, float* dw, float* ws,
                      const float* in_scale, const float* in_shift, int in_act, int B, int Hs,
                      int Ws, int K, int Ho, int Wo, int N, int kh, int kw, int stride, int pad,
                      int dil, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NASSEG_H */
