// Device-side input pipeline of the two train loops: the bytes a DataLoader worker produces go over PCIe as
// uint8 and are expanded on the GPU, instead of being expanded to fp32 on the host and copied at 4x the size.
//
//   k_transform_u8   transform_image             ocrs_models/datasets/util.py:27-35      img.float() / 255.0 - 0.5
//   k_collate_pad    collate_samples, image part ocrs_models/train_rec.py:285-299        right-pad every crop to the bucket width with 0.0
//   k_resize_aa_*    resize(.., antialias=True)  ocrs_models/datasets/hiertext.py:288-294 separable triangle filter (ATen
//                                                                                         _upsample_bilinear2d_aa, align_corners=False)
//
// All three are HBM/byte-bound streaming kernels: no LDS, no MFMA, coalesced 4..16-byte accesses, grids sized to the data.
// C = 1 at these module edges, so NCHW == NHWC and the outputs feed the models without a transpose.
#include "common.h"

namespace {

__device__ __forceinline__ float px_u8(unsigned v) { return (float)v / 255.0f - 0.5f; }  // IEEE fp32 divide, same value as ATen's

template <typename T>
__device__ __forceinline__ void store4(T* p, const float (&v)[4]);
template <>
__device__ __forceinline__ void store4<float>(float* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <>
__device__ __forceinline__ void store4<bf16>(bf16* p, const float (&v)[4]) {
    *reinterpret_cast<uint2*>(p) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
}

// n8 = bytes; each thread expands 16 input bytes per iteration (one b128 load, 4 vector stores), scalar tail.
template <typename T>
__global__ __launch_bounds__(256) void k_transform_u8(const uint8_t* __restrict__ in, T* __restrict__ out, long n) {
    const long nvec = n / 16;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
        const uint4 raw = reinterpret_cast<const uint4*>(in)[i];
        const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float v[4] = {px_u8(w[q] & 0xff), px_u8((w[q] >> 8) & 0xff), px_u8((w[q] >> 16) & 0xff), px_u8(w[q] >> 24)};
            store4(out + i * 16 + q * 4, v);
        }
    }
    for (long i = nvec * 16 + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) Elem<T>::st(out + i, px_u8(in[i]));
}

// One thread = 4 consecutive output columns of one row of one crop.  Crops are packed back to back ((H, w_b) row-major each) so a
// row starts at an arbitrary byte: scalar loads (L2/TA coalesces the 256-byte span of a wave), one 8/16-byte store.
// KIND 0: uint8 crops, transform fused.  KIND 1: fp32 crops that are already in [-0.5, 0.5] (the reference's sample format).
template <typename T, int KIND>
__global__ __launch_bounds__(256) void k_collate_pad(const void* __restrict__ packed, const long long* __restrict__ offs, const int* __restrict__ widths,
                                                     T* __restrict__ out, int H, int Wpad) {
    const int b = blockIdx.z, y = blockIdx.y;
    const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (x0 >= Wpad) return;
    const int w = widths[b];
    const long long row = offs[b] + (long long)y * w;
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int x = x0 + i;
        float f = 0.0f;  // pad value 0.0 = mid grey (train_rec.py:295)
        if (x < w) f = KIND == 0 ? px_u8(static_cast<const uint8_t*>(packed)[row + x]) : static_cast<const float*>(packed)[row + x];
        v[i] = f;
    }
    store4(out + ((size_t)b * H + y) * Wpad + x0, v);
}

// ---- antialiased bilinear resize: one pass per axis, one thread per output element ------------------------------------------
// ATen's weights for output index i along an axis of input size n, output size m (align_corners = False):
//   scale = n / m; support = max(scale, 1); center = scale * (i + 0.5); lo = max(int(center - support + 0.5), 0);
//   cnt = min(int(center + support + 0.5), n) - lo; w_j = tri((j + lo - center + 0.5) / max(scale, 1)), normalised to sum 1.
struct AaSpan {
    int lo, cnt;
    float center, inv, total;
};
__device__ __forceinline__ AaSpan aa_span(int i, int n, float scale) {
    AaSpan s;
    const float support = scale >= 1.0f ? scale : 1.0f;
    s.inv = scale >= 1.0f ? 1.0f / scale : 1.0f;
    s.center = scale * ((float)i + 0.5f);
    s.lo = max((int)(s.center - support + 0.5f), 0);
    s.cnt = min((int)(s.center + support + 0.5f), n) - s.lo;
    s.total = 0.0f;
    for (int j = 0; j < s.cnt; ++j) s.total += fmaxf(0.0f, 1.0f - fabsf(((float)(j + s.lo) - s.center + 0.5f) * s.inv));
    return s;
}
__device__ __forceinline__ float aa_w(const AaSpan& s, int j) {
    const float w = fmaxf(0.0f, 1.0f - fabsf(((float)(j + s.lo) - s.center + 0.5f) * s.inv));
    return s.total != 0.0f ? w / s.total : w;
}

// in [planes][h][w] -> out [planes][h][ow]
__global__ __launch_bounds__(256) void k_resize_aa_h(const float* __restrict__ in, float* __restrict__ out, int h, int w, int ow, float scale) {
    const int ox = blockIdx.x * 256 + threadIdx.x;
    if (ox >= ow) return;
    const size_t rowi = ((size_t)blockIdx.z * h + blockIdx.y);
    const AaSpan s = aa_span(ox, w, scale);
    const float* src = in + rowi * w + s.lo;
    float acc = 0.0f;
    for (int j = 0; j < s.cnt; ++j) acc += aa_w(s, j) * src[j];
    out[rowi * ow + ox] = acc;
}
// in [planes][h][ow] -> out [planes][oh][ow]
__global__ __launch_bounds__(256) void k_resize_aa_v(const float* __restrict__ in, float* __restrict__ out, int h, int oh, int ow, float scale) {
    const int ox = blockIdx.x * 256 + threadIdx.x;
    if (ox >= ow) return;
    const int oy = blockIdx.y;
    const AaSpan s = aa_span(oy, h, scale);  // uniform over the block: scalar registers
    const float* src = in + ((size_t)blockIdx.z * h + s.lo) * ow + ox;
    float acc = 0.0f;
    for (int j = 0; j < s.cnt; ++j) acc += aa_w(s, j) * src[(size_t)j * ow];
    out[((size_t)blockIdx.z * oh + oy) * ow + ox] = acc;
}

}  // namespace

extern "C" {

int ocrs_transform_image_u8(const void* img_u8, void* out, long n, int dtype, hipStream_t st) {
    OCRS_CHECK_ARG(n >= 0 && (dtype == 0 || dtype == 1));
    if (n == 0) return OCRS_OK;
    OCRS_CHECK_ARG(img_u8 && out && (reinterpret_cast<uintptr_t>(img_u8) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0);
    long g = (n / 16 + 255) / 256;
    g = g < 1 ? 1 : (g > kNumCU * 16 ? kNumCU * 16 : g);
    if (dtype == 0)
        hipLaunchKernelGGL(k_transform_u8<float>, dim3((unsigned)g), dim3(256), 0, st, static_cast<const uint8_t*>(img_u8), static_cast<float*>(out), n);
    else
        hipLaunchKernelGGL(k_transform_u8<bf16>, dim3((unsigned)g), dim3(256), 0, st, static_cast<const uint8_t*>(img_u8), static_cast<bf16*>(out), n);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

int ocrs_collate_pad(const void* packed, const long long* offs, const int* widths, void* out, int B, int H, int Wpad, int kind, int dtype,
                     hipStream_t st) {
    OCRS_CHECK_ARG(packed && offs && widths && out && B >= 0 && H > 0 && Wpad > 0 && Wpad % 4 == 0);
    OCRS_CHECK_ARG((kind == 0 || kind == 1) && (dtype == 0 || dtype == 1) && B <= 65535 && H <= 65535);
    OCRS_CHECK_ARG((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    if (B == 0) return OCRS_OK;
    const dim3 grid((Wpad / 4 + 255) / 256, H, B);
#define OCRS_COLLATE(T_, K_) \
    hipLaunchKernelGGL((k_collate_pad<T_, K_>), grid, dim3(256), 0, st, packed, offs, widths, static_cast<T_*>(out), H, Wpad)
    if (dtype == 0 && kind == 0) OCRS_COLLATE(float, 0);
    else if (dtype == 0) OCRS_COLLATE(float, 1);
    else if (kind == 0) OCRS_COLLATE(bf16, 0);
    else OCRS_COLLATE(bf16, 1);
#undef OCRS_COLLATE
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

long ocrs_resize_aa_ws_floats(int planes, int h, int ow) { return (long)planes * h * ow; }

int ocrs_resize_aa(const float* in, float* ws, float* out, int planes, int h, int w, int oh, int ow, hipStream_t st) {
    OCRS_CHECK_ARG(in && ws && out && planes > 0 && h > 0 && w > 0 && oh > 0 && ow > 0 && planes <= 65535 && h <= 65535 && oh <= 65535);
    const float sx = (float)w / (float)ow, sy = (float)h / (float)oh;
    hipLaunchKernelGGL(k_resize_aa_h, dim3((ow + 255) / 256, h, planes), dim3(256), 0, st, in, ws, h, w, ow, sx);
    hipLaunchKernelGGL(k_resize_aa_v, dim3((ow + 255) / 256, oh, planes), dim3(256), 0, st, (const float*)ws, out, h, oh, ow, sy);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

}  // extern "C"
