/*
 * ocrs_hip.h -- C ABI of libocrs_hip.so, the MI355X (gfx950) implementation of the ocrs-models
 * detection / recognition train-step hot path.
 *
 * The reference (robertknight/ocrs-models) is pure PyTorch: its "FFI" for this path is the set of
 * torch.nn / torch.nn.functional calls made from ocrs_models/models.py, train_detection.py and
 * train_rec.py.  Each entry point below names the reference call site(s) it replaces.
 *
 * Conventions
 *   - plain pointers to DEVICE memory, sizes as int/long, a hipStream_t last; no torch types.
 *   - return 0 on success, 1 = bad argument, 2 = HIP launch/runtime error.  Nothing is allocated.
 *   - activations are NHWC ("[P][C]", P = N*H*W pixels), dtype: 0 = fp32, 1 = bf16 (raw uint16 bits).
 *     All arithmetic/accumulation is fp32; parameters, statistics and gradients of parameters are fp32.
 *   - "tr" arrays are per-channel load transforms [3][C] = scale | shift | lo, applied by every
 *     consumer as  x~ = max(x*scale + shift, lo)  (a producer's BatchNorm+ReLU, or identity 1|0|-inf).
 *   - functions that "accumulate" use atomics into a buffer the caller has zeroed.
 */
#ifndef OCRS_HIP_H
#define OCRS_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#ifndef __HIP_PLATFORM_AMD__
typedef struct ihipStream_t* hipStream_t;
#endif

/* ------------------------------------------------------------------ weight packing ---------- */
/* W[k][m] -> MFMA A-operand fragments.  mode 0: element at src[(k/K2)*s1 + (k%K2)*s2 + m*sm];
 * mode 1: ConvTranspose2d forward effective weight (src = W[Cup][Cout][3][3], K = 4*Cup, M = 4*Cout, K2 = Cup). */
int ocrs_pack_frags(const float* src, int mode, int K, int M, int K2, long s1, long s2, long sm, void* out, int dtype, hipStream_t st);
long ocrs_pack_frags_bytes(int K, int M, int dtype);
/* All weight packs of a step in one launch: table = device int64 [n][9] = { src, out, mode, K, M, K2, s1, s2, sm }. */
int ocrs_pack_frags_multi(const long long* table, int n, long max_frag_threads, int dtype, hipStream_t st);

/* ------------------------------------------------------------------ detection forward ------- */
/* DepthwiseConv block up to its pre-BatchNorm output: conv2d(groups=C, 3x3, pad 1) -> conv2d(1x1)
 * (ocrs_models/models.py:11-22) with the channel concat of models.py:89 folded in (xa|xb).
 * gstat [2][Cout] fp64 (sum z | sum z^2) is ACCUMULATED: the caller zeroes it (one memset for all layers of a step);
 * the same holds for gsum of ocrs_bn_bwd_reduce.
 * gamma / pooled (nullable; need ocrs_dwpw_fwd_pool_supported): also write nn.MaxPool2d(2) (models.py:54) of the block output in its
 * pre-BatchNorm form -- the z of each window's selected element (max z for gamma >= 0, min z for gamma < 0: relu(bn(.)) is monotone in z
 * with the sign of the BatchNorm weight) -- to pooled [N][H/2][W/2][Cout]; consumers read it through the block's load transform.
 */
int ocrs_dwpw_fwd(const void* xa, const void* xb, int Ca, int Cb, const float* tra, const float* trb, const float* wdw, const void* wpk,
                  void* z, double* gstat, const float* gamma, void* pooled, int Cout, int N, int H, int W, int dtype, hipStream_t st);
/* ocrs_dwpw_fwd + ocrs_bn_finalize in one launch (models.py:11-23 for the deep levels, 32..256 channels in bf16): the last workgroup done finalises
   the BatchNorm statistics.  counter: a zeroed device word (left zeroed); count .. lo as ocrs_bn_finalize.  No fused pooling. */
long ocrs_dwpw_fwd_fin_supported(int Cin, int Cout, int dtype);
int ocrs_dwpw_fwd_fin(const void* xa, const void* xb, int Ca, int Cb, const float* tra, const float* trb, const float* wdw, const void* wpk, void* z,
                      double* gstat, unsigned* counter, long count, const float* bn_w, const float* bn_b, float eps, float momentum, float* tr, float* saved,
                      float* run_mean, float* run_var, long long* nbt, float lo, int Cout, int N, int H, int W, int dtype, hipStream_t st);
long ocrs_dwpw_fwd_pool_supported(int Cin, int Cout); /* 1 / 0 */
/* The same block forward in fp32 -- the reference's own arithmetic (train_detection.py:92-97 runs models.py:11-23 without autocast) -- as
 * register-resident row-streaming waves (csrc/det_rs32.hip, round 6): Cin = Ca + Cb in {8, 16, 32} (concat only as 8 | 8 or 16 | 16), Cout in
 * {8, 16, 32}; depthwise on the VALU with DPP neighbours, pointwise on the exact-fp32 matrix cores (v_mfma_f32_16x16x4_f32), no LDS tile.
 * wdw [Cin][9], wpw [Cout][Cin]: the fp32 masters in the reference layout.  gstat / gamma / pooled: as ocrs_dwpw_fwd.  counter (nullable: then
 * count .. lo are ignored and the caller runs ocrs_bn_finalize): a zeroed device word (left zeroed) -- the launch's last workgroup finalises the
 * BatchNorm statistics exactly as ocrs_bn_finalize does. */
long ocrs_rs32_fwd_supported(int Ca, int Cb, int Cout, int dtype); /* 1 / 0 */
int ocrs_rs32_fwd(const float* xa, const float* xb, int Ca, int Cb, const float* tra, const float* trb, const float* wdw, const float* wpw, float* z,
                  double* gstat, const float* gamma, float* pooled, unsigned* counter, long count, const float* bn_w, const float* bn_b, float eps,
                  float momentum, float* tr, float* saved, float* run_mean, float* run_var, long long* nbt, float lo, int Cout, int N, int H, int W,
                  hipStream_t st);
/* Backward of the same block in fp32 as ONE row-streaming pass (csrc/det_rs32.hip; the autograd of models.py:11-23 as train_detection.py:96 runs it):
 * replaces ocrs_bn_bwd_finalize + ocrs_pw_bwd + ocrs_dw_bwd -- per pixel g (+ g2), z and x are read once and dL/dx~ is written once, the depthwise-input
 * gradient `du` never goes to memory.  Cin = Ca + Cb in {8, 16} (concat 8 | 8), Cout in {8, 16}; of level 1 also 16 | 16 -> 16 (two single-source passes)
 * and 16 -> 32.
 * pooled = 1 (single source only): g1 / g2 are at half resolution and routed through MaxPool2d(2) (models.py:54) to each window's first maximum.
 * gsum [2][Cout] fp64: the block's COMPLETE BatchNorm-backward sums (the dz coefficients are derived in the prologue; dgamma / dbeta are written);
 * bn: the block's load transform [3][Cout]; dwpw / dwdw ACCUMULATED through ws (ocrs_rs32_bwd_ws_floats() floats; alive until ocrs_bwd_defer_flush
 * when a deferral window is open); saved_a / gsum_a, saved_b / gsum_b (nullable): as ocrs_dw_bwd. */
long ocrs_rs32_bwd_supported(int Ca, int Cb, int Cout, int pooled, int dtype); /* 1 / 0 */
long ocrs_rs32_bwd_ws_floats(int Ca, int Cb, int Cout, int N, int H, int W);
int ocrs_rs32_bwd(const float* xa, const float* xb, int Ca, int Cb, const float* tra, const float* trb, const float* wdw, const float* wpw, const float* g1,
                  const float* g2, const float* z, const float* bn, const double* gsum, const float* gamma, const float* saved, float* dgamma, float* dbeta,
                  float* gxa, float* gxb, float* dwpw, float* dwdw, float* ws, const float* saved_a, double* gsum_a, const float* saved_b, double* gsum_b,
                  int pooled, int Cout, int N, int H, int W, hipStream_t st);
/* ocrs_rs32_bwd for the block in front of out_conv (models.py:125-129; 8 -> 8, single source): its output gradient is formed on the fly,
   g[p][c] = gl[p] * whead[c], from out_conv's dL/dlogit gl [P] fp32 (what ocrs_head_bwd_gl / ocrs_head_bwd_loss write: 4 instead of 32 bytes per pixel) -- the same
   fp32 product ocrs_head_bwd stores, so every output is bit-identical to ocrs_head_bwd + ocrs_rs32_bwd. */
long ocrs_rs32_bwd_head_supported(int Ca, int Cb, int Cout, int dtype); /* 1 / 0 */
int ocrs_rs32_bwd_head(const float* xa, int Ca, const float* tra, const float* wdw, const float* wpw, const float* gl, const float* whead, const float* z,
                       const float* bn, const double* gsum, const float* gamma, const float* saved, float* dgamma, float* dbeta, float* gxa, float* dwpw,
                       float* dwdw, float* ws, const float* saved_a, double* gsum_a, int Cout, int N, int H, int W, hipStream_t st);
/* The same block forward on the matrix cores (csrc/det_mm.hip; bf16, Cin and Cout in {8, 16, 32} and the 32 | 32 concat): depthwise and
 * pointwise conv composed into one 3x3 implicit GEMM (effective weight Wpw[o][c] * Wdw[c][tap] built from the fp32 masters wdw [Cin][9],
 * wpw [Cout][Cin]).  The batch statistics go to ws as ocrs_mm_fwd_nparts() per-block partials [Cout][sum z | sum z^2] (fp32) that
 * ocrs_bn_finalize_parts reduces in a fixed order (bit-reproducible; no atomics).  gamma / pooled: as ocrs_dwpw_fwd. */
long ocrs_mm_fwd_supported(int Ca, int Cb, int Cout, int dtype); /* 1 / 0 */
long ocrs_mm_fwd_nparts(int Ca, int Cb, int Cout, int N, int H, int W);
int ocrs_mm_fwd(const void* xa, const void* xb, int Ca, int Cb, const float* tra, const float* trb, const float* wdw, const float* wpw, void* z,
                float* ws, const float* gamma, void* pooled, int Cout, int N, int H, int W, int dtype, hipStream_t st);
/* ocrs_mm_fwd with ocrs_bn_finalize_parts folded into the same launch (models.py:11-24: conv + BatchNorm2d training statistics; the last workgroup to
   finish reduces the per-block partials in the association order of ocrs_bn_finalize_parts: bit-identical tr / saved / running statistics).
   counter: one zeroed 32-bit word (left zero); bn_w / bn_b: the block's BatchNorm weight / bias; count, eps, momentum, tr, saved, run_mean,
   run_var, nbt, lo: as ocrs_bn_finalize_parts. */
int ocrs_mm_fwd_fin(const void* xa, const void* xb, int Ca, int Cb, const float* tra, const float* trb, const float* wdw, const float* wpw, void* z, float* ws,
                    const float* gamma, void* pooled, unsigned* counter, long count, const float* bn_w, const float* bn_b, float eps, float momentum, float* tr,
                    float* saved, float* run_mean, float* run_var, long long* nbt, float lo, int Cout, int N, int H, int W, int dtype, hipStream_t st);
/* ocrs_mm_fwd_fin for the block behind the first block (models.py:115, in_conv's second DepthwiseConv): the input is given as the first block's u plane
   (ocrs_dwpw_c1_fwd_u) and pointwise weight wexp [8]; tra = the first block's load transform [3][8].  No pooling. */
int ocrs_mm_fwd_fin_xu(const void* xu, const float* wexp, const float* tra, const float* wdw, const float* wpw, void* z, float* ws, unsigned* counter, long count,
                       const float* bn_w, const float* bn_b, float eps, float momentum, float* tr, float* saved, float* run_mean, float* run_var, long long* nbt,
                       float lo, int Cout, int N, int H, int W, int dtype, hipStream_t st);
int ocrs_bn_finalize_parts(const float* parts, int nparts, long count, int C, const float* gamma, const float* beta, float eps, float momentum,
                           float* tr, float* saved, float* run_mean, float* run_var, long long* nbt, float lo, hipStream_t st);
/* Same for the first block (1 -> 8 channels, models.py:115) reading the fp32 image (N,1,H,W). */
int ocrs_dwpw_c1_fwd(const float* img, const float* wdw, const float* wpw, void* z, double* gstat, int N, int H, int W, int dtype,
                     hipStream_t st);
/* The same first block (models.py:115: in_conv's DepthwiseConv(1, 8)), additionally writing uplane [N][H][W] bf16 = its rounded depthwise output u: the block
   output is rank one over the channels, z[p][c] = round(wpw[c] * u[p]), so consumers that take the u plane (ocrs_mm_bwd_fin_xu) read 2 instead of 16
   bytes per pixel.  ocrs_dwpw_c1_u_supported: 1 / 0. */
long ocrs_dwpw_c1_u_supported(int N, int H, int W, int dtype);
int ocrs_dwpw_c1_fwd_u(const float* img, const float* wdw, const float* wpw, void* z, void* uplane, double* gstat, int N, int H, int W, int dtype,
                       hipStream_t st);
/* nn.BatchNorm2d training statistics (models.py:23): sums -> tr [3][C], saved mean|rstd [2][C], running stats, num_batches_tracked. */
int ocrs_bn_finalize(const double* gstat, long count, int C, const float* gamma, const float* beta, float eps, float momentum, float* tr,
                     float* saved, float* run_mean, float* run_var, long long* nbt, float lo, hipStream_t st);
/* nn.MaxPool2d(2) (models.py:54) over relu(bn(z)).  raw = 0: out = the window maximum; raw = 1: out = the pre-BatchNorm z of the selected
 * element (first maximum), to be consumed through the producer's load transform `tr` like any block output. */
int ocrs_maxpool_fwd(const void* z, const float* tr, void* out, int C, int N, int H, int W, int raw, int dtype, hipStream_t st);
/* nn.ConvTranspose2d(k=3, s=2) + crop (models.py:76-78, 82-87). */
int ocrs_convt_fwd(const void* x, const float* tr, const void* wpk, const float* bias, void* out, int Cup, int Cout, int N, int h, int w,
                   int H, int W, int dtype, hipStream_t st);
/* The same ConvTranspose2d forward in fp32 as row-streaming waves over the input grid (csrc/det_rs32.hip, round 6): (Cup, Cout) in {(16, 8), (32, 16),
   (32, 32)}; wt = the fp32 MASTER weight [Cup][Cout][3][3] (no packed fragments: the effective per-parity fragments are built in LDS per workgroup). */
long ocrs_rs32_convt_fwd_supported(int Cup, int Cout, int dtype); /* 1 / 0 */
int ocrs_rs32_convt_fwd(const float* x, const float* tr, const float* wt, const float* bias, float* out, int Cup, int Cout, int N, int h, int w, int H, int W,
                        hipStream_t st);
/* ... and its input gradient (the dx half of ocrs_convt_bwd_parts; autograd of models.py:76-78) in fp32 as row-streaming waves over the input grid, (Cup, Cout)
   in {(16, 8), (32, 16)}, from the MASTER weight; x / tr / saved / gsum (nullable together): also the BatchNorm-backward sums of the block that produced x when
   this ConvTranspose is its only consumer (as ocrs_convt_bwd's saved / gsum: [2][Cup] fp64, ACCUMULATED). */
long ocrs_rs32_convt_dgrad_supported(int Cup, int Cout, int dtype); /* 1 / 0 */
int ocrs_rs32_convt_dgrad(const float* g, const float* wt, float* dx, const float* x, const float* tr, const float* saved, double* gsum, int Cup, int Cout, int N,
                          int h, int w, int H, int W, hipStream_t st);
/* out_conv: nn.Conv2d(8, 1, 1) + nn.Sigmoid (models.py:125-129). */
int ocrs_head_fwd(const void* z, const float* tr, const float* w, const float* b, float* pred, long P, int dtype, hipStream_t st);

/* ------------------------------------------------------------------ detection backward ------ */
/* autograd of BatchNorm2d+ReLU (+MaxPool2d when pooled=1): reductions, then per-channel dz coefficients + dgamma/dbeta. */
int ocrs_bn_bwd_reduce(const void* g1, const void* g2, int pooled, const void* z, const float* bn, const float* saved, double* gsum, int C,
                       int N, int H, int W, int dtype, hipStream_t st);
int ocrs_bn_bwd_finalize(const double* gsum, long count, int C, const float* gamma, const float* saved, float* coef, float* dgamma,
                         float* dbeta, hipStream_t st);
/* autograd of the 1x1 conv (dgrad written to du, wgrad accumulated into dwpw [Cout][Cin]). */
/*   ws: ocrs_pw_bwd_ws_floats() floats of workspace (per-block partials, deterministic two-stage reduction) or NULL (float atomics). */
long ocrs_pw_bwd_ws_floats(int Cin, int Cout, int N, int H, int W);
int ocrs_pw_bwd(const void* xa, const void* xb, int Ca, int Cb, const float* tra, const float* trb, const float* wdw, const void* g1,
                const void* g2, int pooled, const void* z, const float* bn, const float* coef, const void* wpk_d, void* du, float* dwpw,
                float* ws, int Cout, int N, int H, int W, int dtype, hipStream_t st);
/* ocrs_bn_bwd_finalize + ocrs_pw_bwd in one call (the deep-level bf16 kernel folds the finalisation into its prologue): gsum [2][Cout] fp64 complete
   sums of this block, gamma, saved [mean | rstd]; dgamma / dbeta written; coef [3][Cout] is scratch. */
int ocrs_pw_bwd_fin(const void* xa, const void* xb, int Ca, int Cb, const float* tra, const float* trb, const float* wdw, const void* g1, const void* g2,
                    int pooled, const void* z, const float* bn, float* coef, const double* gsum, const float* gamma, const float* saved, float* dgamma,
                    float* dbeta, const void* wpk_d, void* du, float* dwpw, float* ws, int Cout, int N, int H, int W, int dtype, hipStream_t st);
/* autograd of the depthwise 3x3 conv (dL/dx~ split at channel Ca into gxa|gxb; dwdw [C][1][3][3] accumulated). */
/*   ws: ocrs_dw_bwd_ws_floats() floats of workspace (per-block partials, two-stage reduction) or NULL (float atomics). */
long ocrs_dw_bwd_ws_floats(int C, int N, int H, int W);
/*   gsum_a / gsum_b (nullable, need ws): ALSO accumulate the BatchNorm-backward sums [sum ghat | sum ghat*zhat] ([2][Ca] / [2][Cb] fp64,
 *   zeroed by the caller before the first consumer) of the blocks that produced source a / b -- this replaces their ocrs_bn_bwd_reduce
 *   when every consumer of that block output is a depthwise conv; saved_a / saved_b = those blocks' saved [mean | rstd]. */
int ocrs_dw_bwd(const void* xa, const void* xb, int Ca, int Cb, const float* tra, const float* trb, const float* wdw, const void* du,
                void* gxa, void* gxb, float* dwdw, float* ws, const float* saved_a, double* gsum_a, const float* saved_b, double* gsum_b,
                int N, int H, int W, int dtype, hipStream_t st);
/* The same block backward on the matrix cores (csrc/det_mm.hip; bf16, Cin and Cout in {8, 16, 32}, a 32 | 32 concat as two launches):
 * depthwise and pointwise conv composed into one 3x3 implicit GEMM with the effective weight Wpw[o][c] * Wdw[c][tap]; replaces
 * ocrs_pw_bwd + ocrs_dw_bwd (autograd of models.py:12-22 through BatchNorm2d + ReLU [+ MaxPool2d(2) when pooled], models.py:23-24, 54) -- the
 * pointwise input gradient du is never formed.  wdw [Cin][9] / wpw [Cout][Cin]: fp32 master weights; g1 (+ g2) at half resolution when
 * pooled = 1; dwpw / dwdw are ACCUMULATED by a single writer per element (deterministic, no atomics); ws = ocrs_mm_bwd_ws_floats() floats;
 * saved_a / gsum_a, saved_b / gsum_b: as ocrs_dw_bwd (nullable). */
long ocrs_mm_bwd_supported(int Ca, int Cb, int Cout, int dtype); /* 1 / 0 */
long ocrs_mm_bwd_ws_floats(int Ca, int Cb, int Cout, int N, int H, int W);
int ocrs_mm_bwd(const void* xa, const void* xb, int Ca, int Cb, const float* tra, const float* trb, const float* wdw, const float* wpw,
                const void* g1, const void* g2, int pooled, const void* z, const float* bn, const float* coef, void* gxa, void* gxb, float* dwpw,
                float* dwdw, float* ws, const float* saved_a, double* gsum_a, const float* saved_b, double* gsum_b, int Cout, int N, int H,
                int W, int dtype, hipStream_t st);
/* ocrs_mm_bwd with ocrs_bn_bwd_finalize folded into the kernel prologue: instead of `coef`, this block's complete BatchNorm-backward sums gsum [2][Cout]
   (fp64), gamma and saved [mean | rstd]; dgamma / dbeta [Cout] are written (models.py:23 backward). */
int ocrs_mm_bwd_fin(const void* xa, const void* xb, int Ca, int Cb, const float* tra, const float* trb, const float* wdw, const float* wpw, const void* g1,
                    const void* g2, int pooled, const void* z, const float* bn, const double* gsum, const float* gamma, const float* saved, float* dgamma,
                    float* dbeta, void* gxa, void* gxb, float* dwpw, float* dwdw, float* ws, const float* saved_a, double* gsum_a, const float* saved_b,
                    double* gsum_b, int Cout, int N, int H, int W, int dtype, hipStream_t st);
/* ocrs_mm_bwd_fin for the block in front of out_conv (models.py:143 reads up.0.contract's output): the gradient w.r.t. the block output is
   not read from memory but formed as round(gl[p] * whead[c]) from ocrs_head_bwd_gl's 4-byte-per-pixel gl.  ocrs_mm_bwd_head_supported: 1 if this
   launch is covered (the row-streaming backward: bf16, 8 -> 8 channels, one source). */
long ocrs_mm_bwd_head_supported(int Ca, int Cb, int Cout, int N, int H, int W, int dtype);
int ocrs_mm_bwd_fin_head(const void* xa, int Ca, const float* tra, const float* wdw, const float* wpw, const float* gl, const float* whead, const void* z,
                         const float* bn, const double* gsum, const float* gamma, const float* saved, float* dgamma, float* dbeta, void* gxa, float* dwpw,
                         float* dwdw, float* ws, const float* saved_a, double* gsum_a, int Cout, int N, int H, int W, int dtype, hipStream_t st);
/* ocrs_mm_bwd_fin for the block behind the first block (models.py:115: in_conv's second DepthwiseConv): its input is given as the first block's u plane
   (ocrs_dwpw_c1_fwd_u) and pointwise weight wexp [8] -- x[p][c] = round(wexp[c] * u[p]), the stored values, rebuilt from 2 instead of 16 bytes per pixel.
   Needs ocrs_mm_bwd_head_supported(8, 0, Cout, ...). */
int ocrs_mm_bwd_fin_xu(const void* xu, const float* wexp, const float* tra, const float* wdw, const float* wpw, const void* g1, const void* g2, const void* z,
                       const float* bn, const double* gsum, const float* gamma, const float* saved, float* dgamma, float* dbeta, void* gxa, float* dwpw,
                       float* dwdw, float* ws, const float* saved_a, double* gsum_a, int Cout, int N, int H, int W, int dtype, hipStream_t st);
/* acc64 [17] fp64 = dWpw [8] | dWdw [9], ACCUMULATED (caller-zeroed; the caller adds it to the fp32 gradients): fp64 sums of the per-block
 * fp32 partials are exact, hence independent of the order the blocks finish in (float atomics into the fp32 gradients were not). */
int ocrs_dwpw_c1_bwd(const float* img, const float* wdw, const float* wpw, const void* g1, const void* g2, int pooled, const void* z,
                     const float* bn, const float* coef, double* acc64, int N, int H, int W, int dtype, hipStream_t st);
/* autograd of ConvTranspose2d + crop. */
long ocrs_convt_bwd_ws_floats(int Cup, int Cout, int N, int h, int w, int dtype);
/* saved / gsum (nullable; need ocrs_convt_bwd_stats_supported): x is the raw output of a block consumed ONLY by this ConvTranspose -> its
 * BatchNorm-backward sums [2][Cup] (fp64, ACCUMULATED; saved = the block's [mean | rstd]) come from this pass instead of ocrs_bn_bwd_reduce. */
/* dbias64 [Cout] fp64 (caller-zeroed): where the generic (deep-level) path accumulates the bias gradient; the caller adds it to dbias. */
int ocrs_convt_bwd(const void* x, const float* tr, const void* g, const void* wpk_d, void* dx, float* dW, float* dbias, double* dbias64, float* ws,
                   const float* saved, double* gsum, int Cup, int Cout, int N, int h, int w, int H, int W, int dtype, hipStream_t st);
long ocrs_convt_bwd_stats_supported(int Cup, int Cout, int dtype); /* 1 / 0 */
/* The same pass in two calls -- parts bit 0: input gradient dx; bit 1: weight + bias gradients (off the backward's critical path: may run on another
 * stream) -- where ocrs_convt_bwd_splittable() (the generic deep-level path; the tiled path of levels 0-2 needs parts == 3). */
long ocrs_convt_bwd_splittable(int Cup, int Cout, int dtype);
int ocrs_convt_bwd_parts(const void* x, const float* tr, const void* g, const void* wpk_d, void* dx, float* dW, float* dbias, double* dbias64, float* ws,
                         const float* saved, double* gsum, int Cup, int Cout, int N, int h, int w, int H, int W, int parts, int dtype, hipStream_t st);
/* Fused first-block backward (round 5; models.py:115 in_conv = DoubleConv(1, 8): the backward of its first DepthwiseConv block).  The block's weight gradient is
   linear in dL/dx~ of its only consumer, so ocrs_mm_bwd_fin_xu_c1 -- ocrs_mm_bwd_fin_xu for in_conv.seq.1 -- accumulates, instead of storing dL/dx~ (16 B / pixel),
   the sums c1acc [8][32] fp64 (caller-zeroed: R[c] = sum ghat1[c] u, T[tap] = sum (sum_c wexp[c] A[c] ghat1[c]) img(tap), 8 replicas) from the network input
   img [N H W] fp32; ocrs_dwpw_c1_fwd_us is ocrs_dwpw_c1_fwd_u that also accumulates the forward-only sums fsum [20] fp64 (caller-zeroed: sum u | sum u^2 |
   sum u img(tap) [9] | sum img(tap) [9]); ocrs_c1_bwd_fin combines both with the block's BatchNorm-backward coefficients coef [3][8] (ocrs_bn_bwd_finalize)
   into acc64 [17] += dWpw [8] | dWdw [9] -- the output of ocrs_dwpw_c1_bwd, which is then not needed. */
int ocrs_dwpw_c1_fwd_us(const float* img, const float* wdw, const float* wpw, void* z, void* uplane, double* gstat, double* fsum, int N, int H, int W, int dtype,
                        hipStream_t st);
int ocrs_mm_bwd_fin_xu_c1(const void* xu, const float* wexp, const float* tra, const float* wdw, const float* wpw, const void* g1, const void* g2, const void* z,
                          const float* bn, const double* gsum, const float* gamma, const float* saved, float* dgamma, float* dbeta, float* dwpw, float* dwdw,
                          float* ws, const float* saved_a, double* gsum_a, const float* img, double* c1acc, int Cout, int N, int H, int W, int dtype,
                          hipStream_t st);
int ocrs_c1_bwd_fin(const double* c1acc, const double* fsum, const float* coef, const float* wexp, double* acc64, hipStream_t st);
/* Deferred second stage of the block backward (models.py:7-28 backward; no reference counterpart -- it is launch scheduling): between _begin and _flush
   the block-backward entry points (ocrs_mm_bwd*, ocrs_pw_bwd*, ocrs_dw_bwd) (1) finalise the BatchNorm-backward sums they produce for their input's
   producers (gsum_a / gsum_b) in the last workgroup of the block kernel, using state carved from `scratch` (ndoubles ZEROED fp64 values, left zeroed),
   and (2) queue the single-writer reductions of their weight-gradient partials instead of launching them; _flush launches all queued reductions as
   ONE kernel on `st` and ends the mode.  `scratch` and every workspace `ws` passed meanwhile must stay valid until _flush has been queued; dwpw /
   dwdw are complete only after it.  Results are bit-identical to the undeferred launches except for the gsum sums (exact fp64 sums of the fp32
   per-block partials instead of fp32 chain sums).  Per-process state: one backward at a time. */
int ocrs_bwd_defer_begin(double* scratch, long ndoubles);
int ocrs_bwd_defer_flush(hipStream_t st);
/* autograd of out_conv + sigmoid. */
/* acc64 [9] fp64 = dw [8] | db, ACCUMULATED (caller-zeroed, caller adds it to the fp32 gradients; see ocrs_dwpw_c1_bwd). */
int ocrs_head_bwd(const void* z, const float* tr, const float* w, const float* pred, const float* gpred, void* gy, double* acc64,
                  const float* saved, double* gsum, long P, int dtype, hipStream_t st);
/* The same (models.py:127-130, 143: out_conv + Sigmoid backward) with a compact output: gl [P] fp32 = dL/dlogit instead of the 8-channel gradient gy
   (gy[p][c] = gl[p] * w[c] is formed by the consumer, ocrs_mm_bwd_fin_head). */
int ocrs_head_bwd_gl(const void* z, const float* tr, const float* w, const float* pred, const float* gpred, float* gl, double* acc64,
                     const float* saved, double* gsum, long P, int dtype, hipStream_t st);
/* ocrs_balanced_bce_bwd + ocrs_head_bwd_gl in one pass (the autograd of train_detection.py:225-263's loss followed by models.py:143's out_conv + Sigmoid
   backward): dL/dpred is formed on the fly from what ocrs_balanced_bce_fwd saved (pred, target, lpx, cls, state) and gout [1], the upstream gradient
   of the scalar loss; it is never written to memory.  P % 4 == 0; saved / gsum as ocrs_head_bwd (required). */
int ocrs_head_bwd_loss(const void* z, const float* tr, const float* w, const float* pred, const float* target, const float* lpx, const unsigned char* cls,
                       const void* state, const float* gout, float* gl, double* acc64, const float* saved, double* gsum, long P, int dtype,
                       hipStream_t st);

/* ------------------------------------------------------------------ detection loss ---------- */
/* balanced_cross_entropy_loss (ocrs_models/train_detection.py:225-263), forward and backward. */
long ocrs_loss_state_bytes(void);
long ocrs_loss_hist_bytes(void);
int ocrs_balanced_bce_fwd(const float* pred, const float* target, float* lpx, unsigned char* cls, void* state, void* hist, float* loss_out,
                          long P, hipStream_t st);
int ocrs_balanced_bce_bwd(const float* pred, const float* target, const float* lpx, const unsigned char* cls, const void* state,
                          const float* gout, float* gpred, long P, hipStream_t st);
/* dst[table[r][0] + i] += src[table[r][1] + i], i < table[r][2], r < nrows (table: int32 [nrows][3], device): the ~10 fp64 accumulator folds of a
   detection backward (out_conv / first-block weights, ConvTranspose biases: ocrs_models/models.py:93-143 autograd) in one launch. */
int ocrs_fold64_multi(const int* table, int nrows, float* dst, const double* src, hipStream_t st);

/* ------------------------------------------------------------------ recognition (CRNN) ------ */
/* nn.Conv2d forward / dgrad, GRU input projections, nn.Linear as one implicit-GEMM kernel (ocrs_models/models.py:189-240, 245, 248).
 * gstat (nullable) [2][M] fp64 (sum z | sum z^2 of the stored outputs) is ACCUMULATED: the caller zeroes it (one fill for all layers of a
 * step); the same holds for gsum of ocrs_rec_bn_reduce / ocrs_avgpool_bn_reduce. */
int ocrs_conv_igemm(const void* x, int ldx, const void* wpk, void* out, int ldo, const float* bias, int relu, double* gstat, int Cin, int M, int N,
                    int Hi, int Wi, int Ho, int Wo, int KH, int KW, int padh, int padw, int dtype, hipStream_t st);
/* weight gradients of Conv2d / Linear / GRU (autograd of the calls above). */
/* Split-bf16 (bf16x3) weight-gradient GEMM for fp32 operands: dW [CA][CB] += A^T B over P rows (throughput mode of the GRU / Linear
 * weight gradients, train_rec.py:140 backward of models.py:264-268); <= ~1.1e-5 relative error per product, fp32 accumulation.
 * CB % 4 == 0; CA may be ragged (the class count) when ldA >= round_up(CA, 4). */
long ocrs_wgrad_gemm_x3_ws_floats(int CA, int CB, long P);
int ocrs_wgrad_gemm_x3(const float* A, int ldA, int CA, const float* B, int ldB, int CB, float* dW, float* ws, long P, hipStream_t st);
/* Split-bf16 GEMM for fp32 operands (GRU input projections and their input gradients in throughput mode, models.py:264-266):
 * out [P][ldo] = X [P][ldx] (K columns) * W (+ bias), W[m][k] = Wm[m * ldw + k] (km = 0) or Wm[k * ldw + m] (km = 1).
 * Kw (0: K): the k extent Wm really has; X columns [Kw, K) meet zero weights.  km = 0 also takes M % 4 != 0 (nn.Linear(512, n_classes) and
 * its input gradient, models.py:245-248): output columns [M, round_up(M, 4)) are written as 0. */
int ocrs_gemm_x3(const float* X, int ldx, int K, const float* Wm, int ldw, int km, const float* bias, float* out, int ldo, int M, long P, int Kw,
                 hipStream_t st);
/* The pipelined form of the same GEMM (csrc/rec_gemm.hip, round 4: one workgroup per CU = 8 MFMA waves + 4 producer waves that move both
 * operands by LDS-DMA through a three-stage ring, persistent over XCD-ordered tiles).  Weights pre-split and pre-packed once per step:
 * wpk = ocrs_pack_frags(mode 2, dtype 1) of W as A[m][k], 2 * ocrs_pack_frags_bytes(K, M, 1) bytes.  K % 32 == 0, M % 128 == 0, M <= 2048,
 * P * ldx * 4 < 2^31 -- ocrs_gemm_x3p_supported() returns 1 for shapes it takes.  Bit-identical to ocrs_gemm_x3. */
long ocrs_gemm_x3p_supported(int ldx, int K, int ldo, int M, long P);
int ocrs_gemm_x3p(const float* X, int ldx, int K, const void* wpk, const float* bias, float* out, int ldo, int M, long P, hipStream_t st);
/* ... with the workgroup tile height chosen by the caller: ntw = 4 (256 rows), 2 (128 rows) or 0 (automatic, what ocrs_gemm_x3p does). */
int ocrs_gemm_x3p_tiles(const float* X, int ldx, int K, const void* wpk, const float* bias, float* out, int ldo, int M, long P, int ntw, hipStream_t st);
long ocrs_wgrad_gather_ws_floats(int CA, int CB, int ntaps, long P, int dtype);
int ocrs_wgrad_gather(const void* A, int ldA, int CA, const float* trA, const void* B, int ldB, int CB, float* dW, float* ws, int N, int hA,
                      int wA, int HB, int WB, int stride, int padh, int padw, int KH, int KW, int dtype, hipStream_t st);
/* weight gradient of the 3x3 / pad 1 Conv2d layers (models.py:189-231), all nine taps per staged tile. */
long ocrs_conv3x3_wgrad_ws_floats(int Cout, int Cin, int N, int H, int W);
int ocrs_conv3x3_wgrad(const void* dz, int Cout, const void* x, int Cin, float* dW, float* ws, int N, int H, int W, int dtype, hipStream_t st);
/* Conv2d(1,32,3,p1) + ReLU + MaxPool2d(2) (models.py:180-187) fused, forward and backward. */
int ocrs_conv0_fwd(const float* img, const float* w, const float* bias, void* out, int N, int H, int W, int dtype, hipStream_t st);
int ocrs_conv0_bwd(const float* img, const float* w, const float* bias, const void* g, float* dW, float* db, int N, int H, int W, int dtype,
                   hipStream_t st);
/* BatchNorm2d + ReLU + MaxPool2d((2,2)|(2,1)) (models.py:197-199, 214-216, 231-233) forward and the pieces of its backward. */
int ocrs_act_pool_fwd(const void* z, const float* tr, void* out, int C, int N, int H, int W, int PH, int PW, int dtype, hipStream_t st);
int ocrs_rec_bn_reduce(const void* g, const void* z, const float* bn, const float* saved, double* gsum, int C, int N, int H, int W, int PH, int PW,
                       int dtype, hipStream_t st);
int ocrs_dz_apply(const void* g, const void* z, const float* bn, const float* coef, void* dz, int C, int N, int H, int W, int PH, int PW, int dtype,
                  float* dsum /* nullable [C]: += column sums of dz (the bias gradient of a biased conv, models.py:201, 218) */, hipStream_t st);
/* BatchNorm2d + AvgPool2d((4,1)) + permute/reshape to (W, N, C*H) (models.py:241-242, 259-262). */
int ocrs_avgpool_fwd(const void* z, const float* tr, float* seq, int C, int N, int H, int W, int dtype, hipStream_t st);
int ocrs_avgpool_bn_reduce(const float* gseq, const void* z, const float* saved, double* gsum, int C, int N, int H, int W, int dtype, hipStream_t st);
int ocrs_avgpool_dz(const float* gseq, const void* z, const float* coef, void* dz, int C, int N, int H, int W, int dtype, hipStream_t st);
int ocrs_col_sum(const void* a, int ld, int C, float* out, long rows, int dtype, hipStream_t st);
/* nn.GRU(128, 256, bidirectional, 2 layers) recurrence, one layer at a time (models.py:245, 264-266). */
int ocrs_gru_layer_fwd(const float* gi, const float* whh_pk, const float* bhh, float* out, float* saved, int T, int N, hipStream_t st);
int ocrs_gru_layer_bwd(const float* dout, const float* saved, const float* out, const float* whhT_pk, float* dgi, float* dgh, float* dhz, int T,
                       int N, hipStream_t st);
/* The same recurrence as ONE persistent launch per layer and pass (csrc/rec_gru_seq.hip): groups of 16 workgroups own (direction, 32 batch
   columns) and exchange h_t / dgh_t per step with agent-scope 8-byte accesses and an arrival counter.  whh: the fp32 master [2][768][256]
   (weight_hh_l*, weight_hh_l*_reverse stacked; no fragment packing);  sync: ocrs_gru_seq_sync_words(N) 32-bit words (zeroed by the call);
   xws: ocrs_gru_seq_ws_floats(N) floats, the per-step exchange buffer (MFMA-fragment order; initialised by the call);
   err: ONE caller-owned 32-bit word, zeroed once by the caller and sticky -- set if a wait inside a launch timed out (outputs incomplete);
   exact != 0: fp32 MFMA (reference arithmetic), 0: split-bf16 x3 (fp32-class).  ocrs_gru_seq_supported: 1 when every workgroup of the launch
   can be resident on the current device (otherwise use the per-step entry points above; OCRS_GRU_SEQ=0 forces that). */
long ocrs_gru_seq_supported(int N);
long ocrs_gru_seq_sync_words(int N);
long ocrs_gru_seq_ws_floats(int N);
int ocrs_gru_seq_fwd(const float* gi, const float* whh, const float* bhh, float* out, float* saved, int T, int N, unsigned* sync, unsigned* err, float* xws,
                     int exact, hipStream_t st);
int ocrs_gru_seq_bwd(const float* dout, const float* saved, const float* out, const float* whh, float* dgi, float* dgh, int T, int N, unsigned* sync,
                     unsigned* err, float* xws, int exact, float* dbih, float* dbhh, hipStream_t st);
/* dbih / dbhh (nullable) [2 * 768]: the bias gradients (column sums of dgi / dgh over time and batch) are ACCUMULATED there by the same launch. */
int ocrs_gru_seq_status(const unsigned* err, hipStream_t st);
/* nn.LogSoftmax(dim=2) (models.py:250). */
int ocrs_log_softmax_fwd(const void* logits, float* out, long rows, int C, int ld, int dtype, hipStream_t st);
int ocrs_log_softmax_bwd(const float* lp, const float* g, void* dlogits, long rows, int C, int ld, int dtype, hipStream_t st);
/* torch.nn.CTCLoss() (ocrs_models/train_rec.py:104,121): forward (alpha) and backward (beta + gradient). */
int ocrs_ctc_fwd(const float* lp, const int* targets, const long long* in_len, const long long* tg_len, float* alpha, float* nll, float* loss,
                 int T, int N, int C, int Lpad, int Smax, hipStream_t st);
int ocrs_ctc_bwd(const float* lp, const int* targets, const long long* in_len, const long long* tg_len, const float* alpha, const float* nll,
                 const float* gout, float* grad, int T, int N, int C, int Lpad, int Smax, hipStream_t st);
/* The same loss with the beta lattice computed by the same launch (alpha and beta recursions side by side, two workgroups per sample) and a
 * backward that is parallel over (sample, time step) -- round 4; loss and gradient bit-identical to ocrs_ctc_fwd + ocrs_ctc_bwd.
 * alpha, beta: workspaces [N][T][Smax] fp32. */
int ocrs_ctc_fwd_ab(const float* lp, const int* targets, const long long* in_len, const long long* tg_len, float* alpha, float* beta, float* nll,
                    float* loss, int T, int N, int C, int Lpad, int Smax, hipStream_t st);
int ocrs_ctc_grad_ab(const float* lp, const int* targets, const long long* in_len, const long long* tg_len, const float* alpha, const float* beta,
                     const float* nll, const float* gout, float* grad, int T, int N, int C, int Lpad, int Smax, hipStream_t st);
/* torch.nn.CTCLoss() forward AND gradient in one launch (train_rec.py:104,121 + the backward of train_rec.py:140), lattice and log-probabilities
 * in LDS.  ocrs_ctc_fused_lds_bytes: LDS bytes a sample needs, 0 = shape not covered (use ocrs_ctc_fwd / ocrs_ctc_bwd).
 * grad_pre [T][N][C] (nullable: loss only) = dloss/dlog_probs for an upstream gradient of 1. */
long ocrs_ctc_fused_lds_bytes(int T, int C, int Smax);
int ocrs_ctc_fused(const float* lp, const int* targets, const long long* in_len, const long long* tg_len, float* nll, float* loss, float* grad_pre, int T,
                   int N, int C, int Lpad, int Smax, hipStream_t st);
/* out[i] = in[i] * g[0], g a device scalar: the upstream gradient of loss.backward() applied to grad_pre. */
int ocrs_scale_by_dev(const float* in, const float* g, float* out, long n, hipStream_t st);
/* The same with the alpha lattice kept for the backward in fp16 (BASELINE configs[4] "fp16 CTC alpha/beta"; SURVEY D5: a separately-toleranced
 * variant): alpha16 [N][T][Smax] fp16 = alpha - rowmax, rowmax [N][T] fp32 (row maximum per time step).  The recursion and the loss stay fp32
 * (identical loss bits); the gradient sees the fp16 rounding of the lattice (~1e-3 relative). */
int ocrs_ctc_fwd_h16(const float* lp, const int* targets, const long long* in_len, const long long* tg_len, void* alpha16, float* rowmax, float* nll,
                     float* loss, int T, int N, int C, int Lpad, int Smax, hipStream_t st);
int ocrs_ctc_bwd_h16(const float* lp, const int* targets, const long long* in_len, const long long* tg_len, const void* alpha16, const float* rowmax,
                     const float* nll, const float* gout, float* grad, int T, int N, int C, int Lpad, int Smax, hipStream_t st);
/* preds.argmax(-1) + ctc_greedy_decode_text's collapse (train_rec.py:52; datasets/util.py:147-177). */
int ocrs_ctc_greedy_decode(const float* lp, const long long* in_len, int* amax, int* labels, int* lens, int T, int N, int C, hipStream_t st);

/* ------------------------------------------------------------------ input pipeline ----------- */
/* transform_image (ocrs_models/datasets/util.py:27-35): out[i] = float(img_u8[i]) / 255 - 0.5; both pointers 16-byte aligned. */
int ocrs_transform_image_u8(const void* img_u8, void* out, long n, int dtype, hipStream_t st);
/* collate_samples, image part (ocrs_models/train_rec.py:285-299): B crops of H rows, crop b = (H, widths[b]) row-major starting at element
 * offs[b] of `packed` (kind 0: uint8, transform_image fused; kind 1: fp32 already transformed) -> out (B,1,H,Wpad), right-padded with 0.0. */
int ocrs_collate_pad(const void* packed, const long long* offs, const int* widths, void* out, int B, int H, int Wpad, int kind, int dtype,
                     hipStream_t st);
/* torchvision resize(img, [oh, ow], antialias=True) on a float tensor (ocrs_models/datasets/hiertext.py:288-294) =
 * F.interpolate(mode="bilinear", antialias=True, align_corners=False): in [planes][h][w] -> out [planes][oh][ow]; ws = planes*h*ow floats. */
int ocrs_resize_aa(const float* in, float* ws, float* out, int planes, int h, int w, int oh, int ow, hipStream_t st);
long ocrs_resize_aa_ws_floats(int planes, int h, int ow);

/* ------------------------------------------------------------------ optimiser ---------------- */
/* table [nt][5] int64 {param, grad, exp_avg, exp_avg_sq, numel}; chunks [nchunks][2] int32 {tensor, chunk of ocrs_opt_chunk()}. */
int ocrs_opt_chunk(void);
/* torch.nn.utils.clip_grad_norm_ (ocrs_models/train_rec.py:148). */
int ocrs_clip_grad_norm(const long long* table, const int* chunks, int nchunks, float max_norm, double* sumsq, float* norm_out,
                        float* coef_out, int scale_in_place, hipStream_t st);
/* torch.optim.Adam.step (ocrs_models/train_detection.py:97,378; train_rec.py:151,381). */
int ocrs_adam_step(const long long* table, const int* chunks, int nchunks, float b1, float b2, float eps, float step_size, float bc2_sqrt,
                   const float* gscale, hipStream_t st);
/* The same step in capturable form (a train step recorded into a hipGraph, ocrs_models_amd/graph.py): the step count is a device fp32 [1]
   (incremented by this call before use), bias corrections are derived on the device; lr is the only per-step host scalar. */
int ocrs_adam_step_dev(const long long* table, const int* chunks, int nchunks, double b1, double b2, float eps, double lr, float* step,
                       const float* gscale, hipStream_t st);
int ocrs_fill_f32(float* p, float v, long n, hipStream_t st);

/* Measurement support (csrc/prof.hip): while enabled, every launch of the DepthwiseConv block-backward families records its start / stop timestamps
   from the dispatch packet itself (hipExtLaunchKernelGGL) -- no event barrier packets between the kernels.  ocrs_prof_count: launches recorded
   so far;  ocrs_prof_read: durations in ms of launches [first, first + n) into a HOST array (synchronises the stream). */
int ocrs_prof_enable(int on);
long ocrs_prof_count(void);
int ocrs_prof_read(float* ms, long first, long n, hipStream_t st);
/* Test support for the data-parallel path (reference: none -- train_detection.py:375-376 is single-process): `blocks` 256-thread
   workgroups that stay resident for `micros` microseconds on stream st, the CU footprint of a collective's channel kernels.  The persistent
   GRU launches must survive next to it (tests/test_train_loop_gpu.py::test_persistent_gru_under_rccl_allreduce_load). */
int ocrs_cu_hog(int blocks, int micros, hipStream_t st);

#ifdef __cplusplus
}
#endif
#endif /* OCRS_HIP_H */
