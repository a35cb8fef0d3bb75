// Row-streaming forms of the matrix-core DepthwiseConv block kernels (det_rs.hip): interface towards det_mm.hip, which owns the C entry points
// (ocrs_mm_bwd / ocrs_mm_fwd) and routes a launch here when rs_*_supported() says so.
#pragma once
#include "det_common.h"

// backward (direct gradient, Cin / Cout in {8, 16}): 1 if the row-streaming kernel covers the launch
bool rs_bwd_supported(int Ca, int Cb, int Cout, int pooled, int N, int H, int W);
// number of per-block partials the launch writes to ws (each PART = Cout * Cin + 11 * Cin floats, the layout k_mm_bwd_reduce sums)
int rs_bwd_blocks(int Cin, int Cout, int N, int H, int W, int g2);
void rs_bwd_launch(const Src2<bf16>& x, const float* tra, const float* trb, const float* wdw, const float* wpw, int ldw, const bf16* g1, const bf16* g2,
                   const bf16* z, const float* bn, const float* coef, bf16* gxa, bf16* gxb, float* ws, bool stats, int Cout, int N, int H, int W,
                   const BnFin& fin, hipStream_t st, const float* gl = nullptr, const float* whead = nullptr,
                   const bf16* xu = nullptr, const float* wexp = nullptr,  // gl / whead: k_rs_bwd<..., HEAD>; xu / wexp: k_rs_bwd<..., XU>
                   const BwdLast& bl = BwdLast{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0},  // bl.raw: the producers' sums finalised by the last workgroup
                   const float* img = nullptr, double* c1acc = nullptr);  // img / c1acc (with xu): k_rs_bwd<..., C1> -- the first block's sums, no dx~ store

// forward (Cin = 8 from one source or from the first block's u plane, or Cin = 16 from one source or the 8 | 8 concat; Cout = 8; no fused pooling): k_rs_fwd
bool rs_fwd_supported(int Ca, int Cb, int Cout, int N, int H, int W);
int rs_fwd_blocks(int N, int H, int W);  // = the number of per-block statistics partials [Cout][2] the launch writes
void rs_fwd_launch(const Src2<bf16>& x, const bf16* xu, const float* wexp, const float* tra, const float* trb, const float* wdw, const float* wpw, bf16* z, float* ws,
                   int Cout, int N, int H, int W, int nb, const FwdFin& fin, hipStream_t st);
