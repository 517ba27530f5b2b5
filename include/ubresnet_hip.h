/*
 * ubresnet_hip.h -- C ABI of libubresnet_hip.so (hand-written HIP kernels, gfx950 / MI355X).
 *
 * The reference (NuTufts/ubresnet) has no native code and no FFI: every FLOP of its hot path is
 * a torch.nn layer call.  This header is the boundary a binding would
 * target instead of those layer calls; each entry point cites the reference call it replaces.
 * The Python host side (ubresnet_amd/_lib.py, ctypes) binds exactly these symbols.
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless stated; `stream` is a hipStream_t passed as void*.
 *   - activations are NHWC with explicit element strides (ubr_tensor); element type selected by
 *     `dtype` (UBR_F32 / UBR_BF16 / UBR_F16); accumulation is always fp32; statistics are fp64.
 *   - every function validates its arguments on the host and returns 0 on success or a
 *     negative UBR_E* code; ubr_last_error() gives a message.  No function allocates,
 *     frees or synchronises (safe under hipGraph capture), no global mutable state except
 *     the per-thread error string.
 */
#ifndef UBRESNET_HIP_H
#define UBRESNET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UBR_F32 0
#define UBR_BF16 1
#define UBR_F16 2

#define UBR_OK 0
#define UBR_EINVAL (-1)   /* bad argument / unsupported shape */
#define UBR_ELAUNCH (-2)  /* hip launch error */

#define UBR_MAX_TAPS 64
/* Every fp64 accumulator that many workgroups add into (`stats`, `red`, `acc`) is an array of
 * UBR_STAT_SLOTS copies laid out [slot][n]; kernels add into slot blockIdx.x % UBR_STAT_SLOTS and
 * the finalize / cast entry points sum the slots.  Callers allocate UBR_STAT_SLOTS * n doubles
 * and zero them (ubr_zero) before the first accumulating launch. */
#define UBR_STAT_SLOTS 32
/* The BatchNorm-backward / block-tail reduce passes launch at most 1024 workgroups and use only the first UBR_RED_SLOTS
 * stripes of their buffer (the others stay zero, so ubr_bn_bwd_finalize still sums all of them): the apply passes with the
 * finalize fused (ubr_*_apply_fin) re-sum the stripes in every workgroup, and 8 stripes keep that to 128 B per channel. */
#define UBR_RED_SLOTS 8

/* Strided NHWC view: element (n,y,x,c) lives at p + n*sn + y*sy + x*sx + c (strides in elements). */
typedef struct {
  void* p;
  int64_t sn, sy, sx;
} ubr_tensor;

/* Per-input-channel affine + clamp applied while the operand is loaded:
 *   v = max((v - sub[c])*scale[c] + shift[c], lo[c]).
 * This is how BatchNorm2d + ReLU are folded into the consumer: sub = mean, scale = gamma*invstd,
 * shift = beta (the (x-mean)*invstd*gamma+beta form of ATen, which keeps precision when a
 * channel's variance is tiny, as on mostly-empty LArTPC crops); lo = 0 for ReLU, -FLT_MAX for
 * none.  All four NULL = identity.  Zero padding is applied AFTER the transform, as nn.Conv2d
 * pads the BN/ReLU output. */
typedef struct {
  const float* sub;
  const float* scale;
  const float* shift;
  const float* lo;
} ubr_chan_affine;

/* ------------------------------------------------------------------------------------------
 * Implicit-GEMM direct convolution on MFMA.  One descriptor covers:
 *   nn.Conv2d k in {1,3,7}, stride 1/2, dilation 1/3/5  (models/common_layers.py:13-15,33;
 *       models/ub_uresnet.py:41,60,64; models/ASPP_ResNet.py:199-220,275) forward,
 *   nn.ConvTranspose2d k4 s2 p1 forward (models/common_layers.py:125) as 4 output phases,
 *   and the data gradients of both (what autograd's ConvolutionBackward computes).
 * out(oy,ox,co) = bias[co] + addend(oy,ox,co)
 *               + sum_t sum_ci  xform(x)(oy*S + iy0 + dy[t], ox*S + ix0 + dx[t], ci) * w[wt[t]][ci][co]
 * (optionally with ReLUs around the addend, see `act`)
 * `w` is the packed image written by ubr_pack_weights.  `y` is a view of the output grid
 * (OH x OW); phase decompositions pass a view with doubled strides.
 * stats (optional): stats[co] += sum over the grid of out, stats[Cout+co] += sum of out^2
 * (BatchNorm2d batch statistics, accumulated in fp64).
 * epilogue = 1: fused nn.LogSoftmax(dim=1) (models/ub_uresnet.py:143): y.p is then a float*
 * NCHW [N, Cout, OH, OW] buffer and Cout (= num_classes) <= 16.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int32_t dtype;
  int32_t N, H, W, Cin;      /* input tensor extent */
  ubr_tensor x;
  ubr_chan_affine xf;
  const void* w;             /* packed weights */
  int32_t Cout, Cout_pad;    /* Cout_pad = Cout rounded up to 16 (packed image width) */
  int32_t ntaps;
  int8_t dy[UBR_MAX_TAPS], dx[UBR_MAX_TAPS];
  uint8_t wt[UBR_MAX_TAPS];  /* packed tap index used by tap t */
  int32_t S, iy0, ix0;
  int32_t OH, OW;            /* output grid */
  ubr_tensor y;
  ubr_tensor addend;         /* p == NULL: none */
  const float* bias;         /* NULL: none */
  double* stats;             /* NULL: none */
  int32_t epilogue;          /* 0 store T NHWC, 1 log-softmax to fp32 NCHW */
  int32_t tile_hint;         /* 0 auto */
  int32_t act;               /* inference epilogue (BatchNorm folded into w/bias): bit 0 = ReLU before the addend,
                              * bit 1 = ReLU after it:  out = relu?( relu?(conv + bias) + addend )  -- with both bits
                              * this is the whole BasicBlock tail (models/common_layers.py:47-56) */
  int32_t pad_;
  /* Training, data-gradient convs (both optional; zero-initialise the descriptor):
   * addend_mask -- the addend is gated by the ReLU bit mask of a block tail (ubr_block_tail_fwd_masked: one byte per output
   *   pixel and 16-byte channel unit, [N*OH*OW][Cout / channels-per-unit], bit e = channel e of the unit):
   *   out = conv + addend * bit.  This re-forms the skip gradient g_out*[out>0] of an identity block from g_out and the mask, so
   *   the tail's backward need not write it (ubr_block_tail_bwd_apply_fin with g_sc == NULL).
   * bnb_c -- BatchNorm-backward sums in the epilogue: `stats` then receives, per output channel, sum g_y and sum g_y*xhat of
   *   a = max(bn(c), 0)  (g = this conv's output rounded to the storage type, exactly what a separate ubr_bn_bwd_reduce over
   *   the stored tensor reads; g_y = g*[bn(c) > 0]; xhat = (c - mean)*invstd) instead of the output's own statistics: the
   *   reduce pass of ubr_bn_bwd_* for one extra read of c.  c is a view of the OUTPUT grid.  Stripes: UBR_RED_SLOTS. */
  const uint8_t* addend_mask;
  ubr_tensor bnb_c;
  const float *bnb_mean, *bnb_scale, *bnb_shift, *bnb_invstd;
  /* Several output phases in ONE launch (nphase 2..4; 0 or 1 = the single launch described above): the phases of a stride-2
   * transposed conv, or of the data gradient of a stride-2 conv, share x and w but have their own taps and their own strided view of
   * the output (and of the addend).  dy/dx/wt hold the phases' taps back to back: phase p uses taps [phase_tap0[p], +phase_ntaps[p]);
   * its output view is `y` moved by phase_yoff[p] elements (same strides, same OH x OW), its addend `addend` moved by
   * phase_aoff[p].  Same arithmetic per output element as one launch per phase; no statistics / training epilogues. */
  int32_t nphase;
  int32_t phase_tap0[4], phase_ntaps[4];
  int32_t pad3_;
  int64_t phase_yoff[4], phase_aoff[4];
  int32_t stats_slots;       /* stripes of `stats` to use: 0 = UBR_STAT_SLOTS; UBR_RED_SLOTS when the consumer sums them itself
                              * (ubr_block_tail_fwd_fin).  bnb_c always uses UBR_RED_SLOTS. */
  int32_t pad2_;
} ubr_conv_desc;

int ubr_conv(const ubr_conv_desc* d, void* stream);
/* tile configuration (FW, NT, TWF, PIPE = template arguments of conv_igemm_kernel) chosen by this thread's last
 * ubr_conv call: lets host-side timing be keyed by kernel symbol, as rocprofv3 reports it */
void ubr_conv_last_config(int* fw, int* nt, int* twf, int* pipe);
/* kernel symbol (as rocprofv3 --kernel-trace names it) of this thread's last ubr_conv launch, for host-side timing tables */
int ubr_conv_last_kernel(char* buf, int n);

/* Weight repack: fp32 master weights (PyTorch layouts) -> packed image for ubr_conv.
 *   dst[t][ku][m][e] = (T) src[m*sm + (ku*CPU+e)*sk + tapidx[t]],  m < M (zero for M <= m < Mpad)
 * forward orientation of Conv2d [Cout][Cin][kh][kw]: M=Cout, K=Cin, sm=Cin*kh*kw, sk=kh*kw;
 * data-gradient orientation: M=Cin, K=Cout, sm=kh*kw, sk=Cin*kh*kw; ConvTranspose2d
 * [Cin][Cout][4][4] likewise.  K beyond Kvalid is zero-filled up to Kpad (multiple of CPU). */
int ubr_pack_weights(int dtype, const float* src, void* dst, int M, int Mpad, int Kvalid, int Kpad,
                     int64_t sm, int64_t sk, int ntaps, const int32_t* tapidx_host, void* stream);

/* All weight images of a network in ONE launch (a train step repacks ~115 tensors; the fused optimizers do
 * not bump tensor version counters, so the executor repacks every pass instead of caching).  `items_dev` is
 * a DEVICE array; each item is one ubr_pack_weights call with tapidx[t] = t * tap_stride, KU = Kpad / CPU. */
typedef struct {
  const float* src;
  void* dst;
  int64_t sm, sk, tap_stride;
  const float* oscale;       /* optional per-row (m) multiplier: BatchNorm folded into the weights for inference; NULL = 1 */
  int32_t M, Mpad, Kvalid, KU, ntaps, pad_;
} ubr_pack_item;
int ubr_pack_weights_batched(int dtype, const ubr_pack_item* items_dev, int nitems, void* stream);

/* Inference: eval-mode BatchNorm2d folded into the convolution in front of it (deploy/run_ubresnet_precropped.py:88-89
 * runs model.eval()).  For every site:  scale = gamma/sqrt(running_var+eps)  (goes into ubr_pack_item.oscale) and
 * bias = (conv_bias - running_mean)*scale + beta  (goes into ubr_conv_desc.bias), computed in fp64.  items_dev is a
 * DEVICE array; one launch for all sites of a network. */
typedef struct {
  const float *gamma, *beta, *running_mean, *running_var;
  const float* conv_bias;    /* NULL: the convolution has no bias */
  float *scale, *bias;       /* outputs, C floats each */
  int32_t C;
  float eps;
} ubr_bn_fold_item;
int ubr_bn_fold_batched(const ubr_bn_fold_item* items_dev, int nitems, void* stream);

/* ------------------------------------------------------------------------------------------
 * Weight gradient (autograd ConvolutionBackward, weight part) on MFMA with pixels as K.
 *   dW[t][co][ci] = sum_{n,oy,ox} g(n,oy,ox,co) * xform(x)(n, oy*S+iy0+dy[t], ox*S+ix0+dx[t], ci)
 * Each workgroup writes an fp32 partial slab; ubr_wgrad_reduce sums slabs in a fixed order
 * (bitwise reproducible) and scatters into the PyTorch weight-gradient layout:
 *   dst[co*sm + ci*sk + tapidx[t]] (+)= sum_slabs
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int32_t dtype;
  int32_t N, H, W, Cin;
  ubr_tensor x;
  ubr_chan_affine xf;
  int32_t GH, GW, Cout;      /* gradient grid and channels (Cout multiple of 16, padded) */
  ubr_tensor g;
  int32_t ntaps;
  int8_t dy[UBR_MAX_TAPS], dx[UBR_MAX_TAPS];
  int32_t S, iy0, ix0;
  float* slabs;              /* workspace, >= ubr_wgrad_workspace() bytes */
  int32_t nsplit;            /* number of pixel splits (slabs); from ubr_wgrad_plan */
  int32_t exclusive;         /* nonzero: nothing else runs beside this launch (the last weight gradient of a backward pass): fill
                                the GPU (three workgroups per CU, narrow staging) instead of leaving room for the compute stream */
} ubr_wgrad_desc;

/* returns the number of slabs the launch will use for this shape (>=1) and the workspace bytes */
int ubr_wgrad_plan(const ubr_wgrad_desc* d, int32_t* nsplit, int64_t* workspace_bytes);
int ubr_wgrad(const ubr_wgrad_desc* d, void* stream);
void ubr_wgrad_last_config(int* ma, int* nb, int* tpg, int* nsplit_mode, int* bigx);   /* template arguments of wgrad_kernel */
int ubr_wgrad_last_pc(void);   /* 1: that launch was the producer/consumer variant (wgrad_kernel's 7th template argument) */
int ubr_wgrad_reduce(float* slabs /* clobbered */, int nsplit, int ntaps, int Cout_pad, int Cin,
                     int Cout_valid, int Cin_valid, float* dst, int64_t sm, int64_t sk,
                     const int32_t* tapidx_host, int accumulate, void* stream);
/* Several weight gradients' slab sums in ONE launch (the 73 sums of a U-ResNet backward pass were 73 launches of a few
 * microseconds each on the weight-gradient stream): item i is exactly ubr_wgrad_reduce's argument list; per element the same
 * additions in the same order, so the results are bit-identical to nitems single calls.  At most UBR_REDUCE_BATCH items per call. */
#define UBR_REDUCE_BATCH 16
typedef struct {
  float* slabs;
  float* dst;
  int32_t nsplit, ntaps, Cout_pad, Cin, Cout_valid, Cin_valid, accumulate, pad_;
  int64_t sm, sk;
  int32_t tapidx[UBR_MAX_TAPS];
} ubr_wgrad_reduce_item;
int ubr_wgrad_reduce_batched(const ubr_wgrad_reduce_item* items_host, int nitems, void* stream);

/* ------------------------------------------------------------------------------------------
 * Stem: conv1 7x7 s1 p3 + bias on the caller's NCHW fp32 image (models/ub_uresnet.py:41,94;
 * Cin in {1..4}), writing raw NHWC output and BatchNorm statistics; and its weight/bias grads.
 * ---------------------------------------------------------------------------------------- */
int ubr_stem_forward(int dtype, const float* x_nchw, int N, int Cin, int H, int W,
                     const float* weight /*[Cout][Cin][7][7]*/, const float* bias, int Cout,
                     ubr_tensor y, double* stats, void* stream);
int ubr_stem_wgrad(int dtype, const float* x_nchw, int N, int Cin, int H, int W, ubr_tensor g, int Cout,
                   float* partial /*workspace*/, int64_t partial_bytes, float* dweight, float* dbias,
                   int accumulate, void* stream);
int64_t ubr_stem_wgrad_workspace(int N, int Cin, int H, int W, int Cout);
/* Stem on the matrix cores: expand the NCHW fp32 image into NHWC with 16 channels per plane,
 * channel kx (0..6) = the plane shifted by kx-3 columns (7..15 zero).  conv1 then is a 7-tap
 * vertical convolution over 16*Cin channels run by ubr_conv / ubr_wgrad. */
int ubr_stem_expand(int dtype, const float* x_nchw, int N, int Cin, int H, int W, void* out, int64_t out_ps, void* stream);

/* ------------------------------------------------------------------------------------------
 * BatchNorm2d (eps, momentum from the module; e.g. models/common_layers.py:24)
 * ---------------------------------------------------------------------------------------- */
/* train: stats (fp64 sum, sumsq over `count` elements per channel) -> mean, invstd,
 * scale = gamma*invstd, shift = beta (so bn(x) = (x-mean)*scale + shift); running stats updated
 * with the unbiased variance. */
int ubr_bn_finalize(const double* stats, double count, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, int64_t* num_batches_tracked,
                    float momentum, float eps, int C,
                    float* scale, float* shift, float* mean, float* invstd, void* stream);
/* eval: the same four vectors from the running statistics */
int ubr_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, float eps, int C, float* scale, float* shift,
                       float* mean, float* invstd, void* stream);

/* Backward of  a = max(bn(c), lo)  given g_a (sum of up to two tensors):
 *   pass 1 (reduce): red[c] += sum g_y, red[C+c] += sum g_y*xhat      (g_y = g_a * [bn(c) > lo])
 *   finalize       : dgamma, dbeta, and the per-channel constants of pass 2
 *   pass 2 (apply) : g_c = scale * (g_y - k1 - xhat*k2)
 * relu = 0 drops the mask (BatchNorm with no ReLU, e.g. bnpass, models/common_layers.py:50-51). */
int ubr_bn_bwd_reduce(int dtype, int64_t npix, int C, const void* ga, int64_t ga_ps, const void* ga2, int64_t ga2_ps,
                      const void* c, int64_t c_ps, const float* scale, const float* shift, const float* mean,
                      const float* invstd, int relu, double* red, void* stream);
int ubr_bn_bwd_finalize(const double* red, double count, const float* scale /*gamma*invstd*/, const float* invstd,
                        int C, float* dgamma, float* dbeta, int accumulate, float* k1, float* k2, void* stream);
int ubr_bn_bwd_apply(int dtype, int64_t npix, int C, const void* ga, int64_t ga_ps, const void* ga2, int64_t ga2_ps,
                     const void* c, int64_t c_ps, const float* scale, const float* shift, const float* mean,
                     const float* invstd, int relu, const float* k1, const float* k2,
                     void* gc, int64_t gc_ps, void* stream);

/* pass 2 with the finalize fused: every workgroup forms k1 = sum(g_y)/count, k2 = sum(g_y*xhat)/count from the reduce pass's
 * stripes itself and workgroup 0 writes dgamma / dbeta (either may be NULL) -- one launch less on the dependent chain.
 * Needs 8*C bytes of LDS (C <= 8192). */
int ubr_bn_bwd_apply_fin(int dtype, int64_t npix, int C, const void* ga, int64_t ga_ps, const void* ga2, int64_t ga2_ps,
                         const void* c, int64_t c_ps, const float* scale, const float* shift, const float* mean,
                         const float* invstd, int relu, const double* red, double count, float* dgamma, float* dbeta,
                         void* gc, int64_t gc_ps, void* stream);

/* ------------------------------------------------------------------------------------------
 * BasicBlock tail (models/common_layers.py:47-56):
 *   out = relu( relu(bn2(c2)) + shortcut ),  shortcut = bnpass(cb)  or  x
 * forward; backward pass 1 (all per-channel reductions of both BatchNorms) and pass 2
 * (g_c2, and g_cb or the identity-skip gradient g_skip = g_out*[out>0]).
 * ---------------------------------------------------------------------------------------- */
int ubr_block_tail_fwd(int dtype, int64_t npix, int C, const void* c2, int64_t c2_ps, const float* mean2,
                       const float* scale2, const float* shift2, const void* sc, int64_t sc_ps, const float* mean_b,
                       const float* scale_b, const float* shift_b /*NULL => identity shortcut*/,
                       void* out, int64_t out_ps, void* stream);
int ubr_block_tail_bwd_reduce(int dtype, int64_t npix, int C, const void* go, int64_t go_ps, const void* go2, int64_t go2_ps,
                              const void* out, int64_t out_ps, const void* c2, int64_t c2_ps,
                              const float* scale2, const float* shift2, const float* mean2, const float* invstd2,
                              const void* cb, int64_t cb_ps, const float* mean_b, const float* invstd_b,
                              double* red2, double* red_b, void* stream);
int ubr_block_tail_bwd_apply(int dtype, int64_t npix, int C, const void* go, int64_t go_ps, const void* go2, int64_t go2_ps,
                             const void* out, int64_t out_ps, const void* c2, int64_t c2_ps,
                             const float* scale2, const float* shift2, const float* mean2, const float* invstd2,
                             const float* k1_2, const float* k2_2,
                             const void* cb, int64_t cb_ps, const float* scale_b, const float* mean_b, const float* invstd_b,
                             const float* k1_b, const float* k2_b,
                             void* g_c2, int64_t g_c2_ps, void* g_sc, int64_t g_sc_ps, void* stream);

/* Forward with the train-mode BatchNorm finalize(s) fused: the convs that produced c2 (and cb) accumulated their statistics with
 * UBR_RED_SLOTS stripes (ubr_conv_desc.stats_slots); every workgroup forms mean / scale from them, workgroup 0 writes the
 * site's vectors (for the backward pass), the running statistics and the batch counter -- ubr_bn_finalize's arithmetic, without
 * its launches on the dependent chain.  momentum < 0 = cumulative averaging (nn.BatchNorm2d(momentum=None)).  bn_b NULL =
 * identity shortcut.  relu_mask may be NULL.  Needs 12*C (24*C) bytes of LDS. */
typedef struct {
  const double* stats;             /* [UBR_STAT_SLOTS][2*C] of which the first UBR_RED_SLOTS stripes are used */
  const float *gamma, *beta;
  float *running_mean, *running_var; int64_t* num_batches_tracked;     /* all NULL: not tracked */
  float momentum, eps;
  float *scale, *shift, *mean, *invstd;                                 /* outputs, C floats each */
} ubr_bn_fwd_fin;
int ubr_block_tail_fwd_fin(int dtype, int64_t npix, int C, const void* c2, int64_t c2_ps, const ubr_bn_fwd_fin* bn2,
                           const void* sc, int64_t sc_ps, const ubr_bn_fwd_fin* bn_b, double count,
                           void* out, int64_t out_ps, uint8_t* relu_mask, void* stream);

/* The same three with the final ReLU's mask kept as bits: `relu_mask` holds one byte per pixel and 16-byte channel unit
 * ([npix][C / channels-per-unit], bit e = "stored output of channel e of the unit is > 0").  The forward writes it beside
 * `out`; the two backward passes read it instead of `out` -- 1 byte per unit instead of 16, i.e. two of the eight tensor passes
 * of a block tail's backward (autograd keeps the whole output for threshold_backward; reference models/common_layers.py:56). */
int ubr_block_tail_fwd_masked(int dtype, int64_t npix, int C, const void* c2, int64_t c2_ps, const float* mean2,
                              const float* scale2, const float* shift2, const void* sc, int64_t sc_ps, const float* mean_b,
                              const float* scale_b, const float* shift_b, void* out, int64_t out_ps, uint8_t* relu_mask, void* stream);
int ubr_block_tail_bwd_reduce_masked(int dtype, int64_t npix, int C, const void* go, int64_t go_ps, const void* go2, int64_t go2_ps,
                                     const uint8_t* relu_mask, const void* c2, int64_t c2_ps,
                                     const float* scale2, const float* shift2, const float* mean2, const float* invstd2,
                                     const void* cb, int64_t cb_ps, const float* mean_b, const float* invstd_b,
                                     double* red2, double* red_b, void* stream);
int ubr_block_tail_bwd_apply_masked(int dtype, int64_t npix, int C, const void* go, int64_t go_ps, const void* go2, int64_t go2_ps,
                                    const uint8_t* relu_mask, const void* c2, int64_t c2_ps,
                                    const float* scale2, const float* shift2, const float* mean2, const float* invstd2,
                                    const float* k1_2, const float* k2_2,
                                    const void* cb, int64_t cb_ps, const float* scale_b, const float* mean_b, const float* invstd_b,
                                    const float* k1_b, const float* k2_b,
                                    void* g_c2, int64_t g_c2_ps, void* g_sc, int64_t g_sc_ps, void* stream);

/* masked apply pass with both finalizes fused (see ubr_bn_bwd_apply_fin); red2 / red_b are the reduce pass's buffers.
 * On an identity block (cb == NULL) g_sc may be NULL: the skip gradient g_out*[out>0] is then not written, and the
 * consumer re-forms it from g_out and the mask (ubr_conv_desc.addend_mask). */
int ubr_block_tail_bwd_apply_fin(int dtype, int64_t npix, int C, const void* go, int64_t go_ps, const void* go2, int64_t go2_ps,
                                 const uint8_t* relu_mask, const void* c2, int64_t c2_ps,
                                 const float* scale2, const float* shift2, const float* mean2, const float* invstd2,
                                 const double* red2, float* dgamma2, float* dbeta2,
                                 const void* cb, int64_t cb_ps, const float* scale_b, const float* mean_b, const float* invstd_b,
                                 const double* red_b, float* dgamma_b, float* dbeta_b, double count,
                                 void* g_c2, int64_t g_c2_ps, void* g_sc, int64_t g_sc_ps, void* stream);

/* ------------------------------------------------------------------------------------------
 * nn.MaxPool2d(3, stride, padding=1)  (models/ub_uresnet.py:44 stride 2; ASPP_ResNet.py:222 stride 1)
 * forward reads the (optionally transformed) input, writes the pooled map and optionally the
 * transformed input itself (`xcopy`, the skip tensor x0 of models/ub_uresnet.py:96);
 * backward gathers g_pooled through the arg-max (first maximum in scan order, as ATen) and adds
 * `g_extra` (gradient arriving through the skip connection).  `argmax` (optional, uint8
 * [N][OH][OW][C], tap index ky*3+kx): forward records it, and a stride-2 backward given it reads
 * four gradient units and four index words per 2x2 input block instead of re-scanning windows;
 * without it backward recomputes the arg-max from x.
 * ---------------------------------------------------------------------------------------- */
int ubr_maxpool_fwd(int dtype, int N, int H, int W, int C, int stride, const void* x, int64_t x_ps,
                    ubr_chan_affine xf, void* pooled, int64_t p_ps, void* xcopy, int64_t xc_ps, uint8_t* argmax, void* stream);
int ubr_maxpool_bwd(int dtype, int N, int H, int W, int C, int stride, const void* x, int64_t x_ps,
                    ubr_chan_affine xf, const void* g_pooled, int64_t gp_ps, const void* g_extra, int64_t ge_ps,
                    void* gx, int64_t gx_ps, const uint8_t* argmax, void* stream);

/* ------------------------------------------------------------------------------------------
 * Head / loss
 * ---------------------------------------------------------------------------------------- */
/* nn.LogSoftmax backward fused with the NCHW->NHWC re-layout: g_logits (T, NHWC, Cpad=16 channels,
 * zero padded) = g_logp - exp(logp) * sum_c g_logp */
int ubr_logsoftmax_bwd(int dtype, int N, int C, int H, int W, const float* g_logp_nchw, const float* logp_nchw,
                       void* g_logits, int64_t gl_ps, void* stream);
/* PixelWiseNLLLoss.forward (training/pixelwise_nllloss.py:41-61): acc[0] += sum over pixels of
 * -predict[b,target,h,w]*classw[target]*pixelweights ; loss = acc/(B*H*W) is formed by the caller.
 * bad_labels (optional device counter, caller-zeroed): += number of targets outside [0,C) that are not ignore_index;
 * F.nll_loss raises a device assert for those, the host side of this package raises RuntimeError from the count. */
int ubr_pixelwise_nll_fwd(const float* predict_nchw, const int64_t* target, const float* pixelweights,
                          const float* classw /*NULL*/, int N, int C, int H, int W, int64_t ignore_index,
                          double* acc, unsigned long long* bad_labels /*NULL*/, void* stream);
int ubr_pixelwise_nll_bwd(const float* g_loss /*device scalar*/, const int64_t* target, const float* pixelweights,
                          const float* classw, int N, int C, int H, int W, int64_t ignore_index,
                          float* g_predict_nchw, void* stream);
/* accuracy() (training/train_ubresnet2018_wlarcv2.py:509-566) in one pass: cm[true*C+pred] += 1
 * with pred = first arg-max over channels (Tensor.max(1) tie-break). */
int ubr_confusion(const float* logp_nchw, const int64_t* target, int N, int C, int H, int W,
                  unsigned long long* cm, void* stream);

/* per-channel sum over pixels of an NHWC tensor (conv bias gradients): out[c] (+)= sum_p g[p][c] */
int ubr_channel_sum(int dtype, int64_t npix, int C, const void* g, int64_t g_ps, double* red, void* stream);
/* dst[i] (+)= scale * sum_slots src[slot*stride + i] */
int ubr_cast_f64_to_f32(const double* src, int stride, int slots, float* dst, int n, double scale, int accumulate, void* stream);
int ubr_zero(void* p, int64_t bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Flat optimizer steps over one contiguous fp32 parameter buffer (laid out like the flat gradient
 * buffer of the backward pass): ONE launch per step instead of one per tensor list.
 * Formulas are torch.optim's (L2 weight decay added to the gradient, not decoupled):
 *   Adam  (training/train_ubresnet2018_wlarcv2.py:155-157: lr 1e-5, weight_decay 1e-4)
 *   SGD   (training/train_ubresnet2018_wlarcv1.py:127-129: momentum 0.9, weight_decay 1e-4)
 * `step` is the 1-based step count (bias corrections); `grad_scale` multiplies the gradient first
 * (1/world for a summed all-reduce, 1 otherwise); n must be a multiple of 4, buffers 16-byte aligned.
 * ---------------------------------------------------------------------------------------- */
int ubr_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int64_t step, float grad_scale, void* stream);
int ubr_sgd_step(float* param, const float* grad, float* momentum_buf /* NULL iff momentum == 0 */, int64_t n, float lr,
                 float momentum, float dampening, float weight_decay, int nesterov, int first_step, float grad_scale, void* stream);

/* ------------------------------------------------------------------------------------------
 * Whole-view tiling (deploy/run_ubresnet_wholeview.py:191-277 slices (bs,1,512,832) crops out of
 * [3,1,rows,cols] plane images and stitches the network output back).  tile_desc_host is an
 * int32 [ntiles][7] HOST array {plane, row0, col0, keep_r0, keep_r1, keep_c0, keep_c1}: the tile
 * covers view[plane][row0:row0+th][col0:col0+tw]; when stitching, tile pixels inside the keep
 * window (tile coordinates) are written to out[plane][c][row0+y][col0+x].  Keep windows of a
 * tiling partition the view, so every output pixel has exactly one writer.
 * ---------------------------------------------------------------------------------------- */
#define UBR_MAX_TILES 64
int ubr_crop_tiles(const float* view /*[P][rows][cols]*/, int P, int rows, int cols, const int32_t* tile_desc_host, int ntiles,
                   int th, int tw, float* out /*[ntiles][1][th][tw]*/, void* stream);
int ubr_stitch_tiles(const float* scores /*[ntiles][C][th][tw]*/, int C, int th, int tw, const int32_t* tile_desc_host, int ntiles,
                     float* out /*[P][C][rows][cols]*/, int P, int rows, int cols, void* stream);

/* ------------------------------------------------------------------------------------------
 * Launch plans ("tapes").  The reference has no scheduler of its own: each layer is a Python-level torch.nn call
 * (models/ub_uresnet.py:88-147) and autograd replays them in reverse.  Here the host records the launch sequence of a
 * pass ONCE per (network, shape, dtype) and replays it from C++:
 *   ubr_tape_begin(t, n, streams): from now on every entry point of this header called on this thread with one of the
 *       `streams` (slot i = streams[i]) still runs AND is appended to the tape with all its arguments resolved
 *       (descriptor validation, tile selection and LDS planning are not repeated at replay).
 *   ubr_tape_fork(t, a, b): slot b waits for everything recorded so far on slot a (event record + stream wait);
 *       this is how the weight-gradient side stream forks from and joins the dgrad chain.
 *   ubr_tape_mark(t, s) -> id: an event recorded on slot s at this point of every replay; ubr_tape_wait_mark makes any
 *       other stream (e.g. the RCCL exchange stream of the data-parallel reducer) wait for it.
 *   ubr_tape_pause(t, 1/0): launches in between run but are not recorded (ops whose operands change per step).
 *   ubr_tape_replay(t, n, streams): re-issue the recording on the given streams.  All device addresses are baked in,
 *       so the caller keeps every buffer of the pass alive and at the same address for the life of the tape.
 * The optimizer steps are never recorded (their scalars change every step).  Not thread-safe per tape; recording is
 * per thread.
 * ---------------------------------------------------------------------------------------- */
#define UBR_TAPE_MAX_STREAMS 4
typedef struct ubr_tape ubr_tape;
ubr_tape* ubr_tape_create(void);
void ubr_tape_destroy(ubr_tape* t);
int ubr_tape_begin(ubr_tape* t, int nstreams, void* const* streams);
int ubr_tape_end(ubr_tape* t);
int ubr_tape_pause(ubr_tape* t, int on);
int ubr_tape_fork(ubr_tape* t, int from_slot, int to_slot);
int ubr_tape_mark(ubr_tape* t, int slot);                 /* >= 0: mark id; < 0: error */
int ubr_tape_wait_mark(const ubr_tape* t, int mark, void* stream);
int ubr_tape_size(const ubr_tape* t);
int ubr_tape_replay(const ubr_tape* t, int nstreams, void* const* streams);
/* tag the launches recorded from now on (label >= 0; -1 = none): the host's index of the operator call they belong to */
int ubr_tape_set_label(ubr_tape* t, int label);
/* replay with timing events around every launch on its own stream; synchronises the tape's streams.  ms[i], label[i] per
   node (label -2: fork / mark node); cap = room in both arrays (>= ubr_tape_size) */
int ubr_tape_replay_timed(const ubr_tape* t, int nstreams, void* const* streams, float* ms, int32_t* label, int cap);

const char* ubr_last_error(void);
int ubr_version(void);

#ifdef __cplusplus
}
#endif
#endif /* UBRESNET_HIP_H */
