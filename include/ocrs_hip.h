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

/* ------------------------------------------------------------------ detection forward ------- */
/* DepthwiseConv block up to its pre-BatchNorm output: conv2d(groups=C, 3x3, pad 1) -> conv2d(1x1)
 * (ocrs_models/models.py:11-22) with the channel concat of models.py:89 folded in (xa|xb). */
int ocrs_dwpw_fwd(const void* xa, const void* xb, int Ca, int Cb, const float* tra, const float* trb, const float* wdw, const void* wpk,
                  void* z, double* gstat, int Cout, int N, int H, int W, int dtype, hipStream_t st);
/* Same for the first block (1 -> 8 channels, models.py:115) reading the fp32 image (N,1,H,W). */
int ocrs_dwpw_c1_fwd(const float* img, const float* wdw, const float* wpw, void* z, double* gstat, int N, int H, int W, int dtype,
                     hipStream_t st);
/* nn.BatchNorm2d training statistics (models.py:23): sums -> tr [3][C], saved mean|rstd [2][C], running stats, num_batches_tracked. */
int ocrs_bn_finalize(const double* gstat, long count, int C, const float* gamma, const float* beta, float eps, float momentum, float* tr,
                     float* saved, float* run_mean, float* run_var, long long* nbt, float lo, hipStream_t st);
/* nn.MaxPool2d(2) (models.py:54) over relu(bn(z)). */
int ocrs_maxpool_fwd(const void* z, const float* tr, void* out, int C, int N, int H, int W, int dtype, hipStream_t st);
/* nn.ConvTranspose2d(k=3, s=2) + crop (models.py:76-78, 82-87). */
int ocrs_convt_fwd(const void* x, const float* tr, const void* wpk, const float* bias, void* out, int Cup, int Cout, int N, int h, int w,
                   int H, int W, int dtype, hipStream_t st);
/* out_conv: nn.Conv2d(8, 1, 1) + nn.Sigmoid (models.py:125-129). */
int ocrs_head_fwd(const void* z, const float* tr, const float* w, const float* b, float* pred, long P, int dtype, hipStream_t st);

/* ------------------------------------------------------------------ detection backward ------ */
/* autograd of BatchNorm2d+ReLU (+MaxPool2d when pooled=1): reductions, then per-channel dz coefficients + dgamma/dbeta. */
int ocrs_bn_bwd_reduce(const void* g1, const void* g2, int pooled, const void* z, const float* bn, const float* saved, double* gsum, int C,
                       int N, int H, int W, int dtype, hipStream_t st);
int ocrs_bn_bwd_finalize(const double* gsum, long count, int C, const float* gamma, const float* saved, float* coef, float* dgamma,
                         float* dbeta, hipStream_t st);
/* autograd of the 1x1 conv (dgrad written to du, wgrad accumulated into dwpw [Cout][Cin]). */
int ocrs_pw_bwd(const void* xa, const void* xb, int Ca, int Cb, const float* tra, const float* trb, const float* wdw, const void* g1,
                const void* g2, int pooled, const void* z, const float* bn, const float* coef, const void* wpk_d, void* du, float* dwpw,
                int Cout, int N, int H, int W, int dtype, hipStream_t st);
/* autograd of the depthwise 3x3 conv (dL/dx~ split at channel Ca into gxa|gxb; dwdw [C][1][3][3] accumulated). */
int ocrs_dw_bwd(const void* xa, const void* xb, int Ca, int Cb, const float* tra, const float* trb, const float* wdw, const void* du,
                void* gxa, void* gxb, float* dwdw, int N, int H, int W, int dtype, hipStream_t st);
int ocrs_dwpw_c1_bwd(const float* img, const float* wdw, const float* wpw, const void* g1, const void* g2, int pooled, const void* z,
                     const float* bn, const float* coef, float* du_ws, float* dwpw, float* dwdw, int N, int H, int W, int dtype,
                     hipStream_t st);
/* autograd of ConvTranspose2d + crop. */
int ocrs_convt_bwd(const void* x, const float* tr, const void* g, const void* wpk_d, void* dx, float* dW, float* dbias, int Cup, int Cout,
                   int N, int h, int w, int H, int W, int dtype, hipStream_t st);
/* autograd of out_conv + sigmoid. */
int ocrs_head_bwd(const void* z, const float* tr, const float* w, const float* pred, const float* gpred, void* gy, float* dw, float* db,
                  long P, int dtype, hipStream_t st);

/* ------------------------------------------------------------------ detection loss ---------- */
/* balanced_cross_entropy_loss (ocrs_models/train_detection.py:225-263), forward and backward. */
long ocrs_loss_state_bytes(void);
long ocrs_loss_hist_bytes(void);
int ocrs_balanced_bce_fwd(const float* pred, const float* target, float* lpx, unsigned char* cls, void* state, void* hist, float* loss_out,
                          long P, hipStream_t st);
int ocrs_balanced_bce_bwd(const float* pred, const float* target, const float* lpx, const unsigned char* cls, const void* state,
                          const float* gout, float* gpred, long P, hipStream_t st);

/* ------------------------------------------------------------------ optimiser ---------------- */
/* table [nt][5] int64 {param, grad, exp_avg, exp_avg_sq, numel}; chunks [nchunks][2] int32 {tensor, chunk of ocrs_opt_chunk()}. */
int ocrs_opt_chunk(void);
/* torch.nn.utils.clip_grad_norm_ (ocrs_models/train_rec.py:148). */
int ocrs_clip_grad_norm(const long long* table, const int* chunks, int nchunks, float max_norm, double* sumsq, float* norm_out,
                        float* coef_out, int scale_in_place, hipStream_t st);
/* torch.optim.Adam.step (ocrs_models/train_detection.py:97,378; train_rec.py:151,381). */
int ocrs_adam_step(const long long* table, const int* chunks, int nchunks, float b1, float b2, float eps, float step_size, float bc2_sqrt,
                   const float* gscale, hipStream_t st);
int ocrs_fill_f32(float* p, float v, long n, hipStream_t st);

#ifdef __cplusplus
}
#endif
#endif /* OCRS_HIP_H */
