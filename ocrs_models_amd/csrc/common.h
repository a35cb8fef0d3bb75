// Common device helpers for the gfx950 (MI355X / CDNA4) kernels.
// Wave = 64 lanes, everything here hard-codes that.  No CUDA compatibility paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define OCRS_OK 0
#include <hip/hip_ext.h>
#define OCRS_ERR_ARG 1
#define OCRS_ERR_HIP 2

#define OCRS_CHECK_ARG(cond)            \
    do {                                \
        if (!(cond)) return OCRS_ERR_ARG; \
    } while (0)

#define OCRS_LAUNCH_CHECK()                              \
    do {                                                 \
        if (hipGetLastError() != hipSuccess) return OCRS_ERR_HIP; \
    } while (0)

static constexpr int kNumCU = 256;

// ----------------------------------------------------------------------------------------------
// storage types: activations live in HBM as NHWC, either fp32 or bf16; all arithmetic is fp32.
// ----------------------------------------------------------------------------------------------
struct bf16 {
    unsigned short v;
};

__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
// fp32 -> bf16 (round to nearest even): plain casts, which hipcc lowers to v_cvt_pk_bf16_f32 on gfx950
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned short f2bf(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }

template <class T>
struct Elem;
template <>
struct Elem<float> {
    static constexpr bool is_bf16 = false;
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
    __device__ static __forceinline__ float round(float v) { return v; }
};
template <>
struct Elem<bf16> {
    static constexpr bool is_bf16 = true;
    __device__ static __forceinline__ float ld(const bf16* p) { return bf2f(p->v); }
    __device__ static __forceinline__ void st(bf16* p, float v) { p->v = f2bf(v); }
    __device__ static __forceinline__ float round(float v) { return bf2f(f2bf(v)); }
};

// 8 consecutive channels (the NHWC access quantum): 16 B for bf16, 32 B for fp32.
__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    const float4 b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void load8(const bf16* p, float (&v)[8]) {
    const uint4 a = *reinterpret_cast<const uint4*>(p);
    v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
    v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
    v[4] = __uint_as_float(a.z << 16); v[5] = __uint_as_float(a.z & 0xffff0000u);
    v[6] = __uint_as_float(a.w << 16); v[7] = __uint_as_float(a.w & 0xffff0000u);
}
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
    const bf16x2_t v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ void store8(bf16* p, const float (&v)[8]) {
    *reinterpret_cast<uint4*>(p) = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
}
// store8 with the bf16 packing hidden from the optimiser (inline v_cvt_pk_bf16_f32).  With the plain (__bf16) casts hipcc SLP-vectorises
// the whole producing loop (e.g. the 9-tap depthwise sum) around the packed converts and live ranges explode: k_pw_bwd<16,16> went from
// 129 to 181 VGPRs (occupancy 3 -> 2) with store8() at the end of the tap loop.
__device__ __forceinline__ void store8_opaque(float* p, const float (&v)[8]) { store8(p, v); }
__device__ __forceinline__ void store8_opaque(bf16* p, const float (&v)[8]) {
    unsigned pk[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[i]) : "v"(v[2 * i]), "v"(v[2 * i + 1]));
    *reinterpret_cast<uint4*>(p) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
}
// raw (still packed) 8-channel vector: lets a kernel issue global loads early and convert at the point of use
template <class T>
struct Raw8;
template <>
struct Raw8<float> {
    float4 a, b;
};
template <>
struct Raw8<bf16> {
    uint4 a;
};
__device__ __forceinline__ Raw8<float> load8_raw(const float* p) {
    Raw8<float> r;
    r.a = *reinterpret_cast<const float4*>(p);
    r.b = *reinterpret_cast<const float4*>(p + 4);
    return r;
}
__device__ __forceinline__ Raw8<bf16> load8_raw(const bf16* p) {
    Raw8<bf16> r;
    r.a = *reinterpret_cast<const uint4*>(p);
    return r;
}
__device__ __forceinline__ void unpack8(const Raw8<float>& r, float (&v)[8]) {
    v[0] = r.a.x; v[1] = r.a.y; v[2] = r.a.z; v[3] = r.a.w;
    v[4] = r.b.x; v[5] = r.b.y; v[6] = r.b.z; v[7] = r.b.w;
}
__device__ __forceinline__ void unpack8(const Raw8<bf16>& r, float (&v)[8]) {
    v[0] = __uint_as_float(r.a.x << 16); v[1] = __uint_as_float(r.a.x & 0xffff0000u);
    v[2] = __uint_as_float(r.a.y << 16); v[3] = __uint_as_float(r.a.y & 0xffff0000u);
    v[4] = __uint_as_float(r.a.z << 16); v[5] = __uint_as_float(r.a.z & 0xffff0000u);
    v[6] = __uint_as_float(r.a.w << 16); v[7] = __uint_as_float(r.a.w & 0xffff0000u);
}

// 4 consecutive channels (MFMA accumulator quad): 8 B / 16 B
__device__ __forceinline__ void store4(float* p, float a, float b, float c, float d) {
    *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
}
__device__ __forceinline__ void store4(bf16* p, float a, float b, float c, float d) {
    *reinterpret_cast<uint2*>(p) = make_uint2(pack2bf(a, b), pack2bf(c, d));
}
__device__ __forceinline__ void load4(const float* p, float (&v)[4]) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
}
__device__ __forceinline__ void load4(const bf16* p, float (&v)[4]) {
    const uint2 a = *reinterpret_cast<const uint2*>(p);
    v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
    v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
}

// raw 4-channel vector (register prefetch of one MFMA-quad / depthwise slab element)
template <class T>
struct Raw4;
template <>
struct Raw4<float> {
    float4 a;
};
template <>
struct Raw4<bf16> {
    uint2 a;
};
__device__ __forceinline__ Raw4<float> load4_raw(const float* p) {
    Raw4<float> r;
    r.a = *reinterpret_cast<const float4*>(p);
    return r;
}
__device__ __forceinline__ Raw4<bf16> load4_raw(const bf16* p) {
    Raw4<bf16> r;
    r.a = *reinterpret_cast<const uint2*>(p);
    return r;
}
__device__ __forceinline__ void unpack4(const Raw4<float>& r, float (&v)[4]) {
    v[0] = r.a.x; v[1] = r.a.y; v[2] = r.a.z; v[3] = r.a.w;
}
__device__ __forceinline__ void unpack4(const Raw4<bf16>& r, float (&v)[4]) {
    v[0] = __uint_as_float(r.a.x << 16); v[1] = __uint_as_float(r.a.x & 0xffff0000u);
    v[2] = __uint_as_float(r.a.y << 16); v[3] = __uint_as_float(r.a.y & 0xffff0000u);
}

// ----------------------------------------------------------------------------------------------
// wave64 helpers
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// Deterministic block-level sum of N per-thread values (all threads of a 256-thread block contribute to the same N outputs): fixed wave
// shuffle tree -> one LDS slot per (wave, value) -> thread i < N adds the four waves in order.  No atomics: float LDS atomics from several
// waves onto one address complete in arrival order, which made every statistic that went through them differ in the last bits from run to
// run.  s_slots: [4][N] floats.  Returns the block total to thread i (i < N); other threads get 0.
template <int N>
__device__ __forceinline__ float block_sum_det(const float (&v)[N], float* s_slots) {
    const int wave = threadIdx.x >> 6;
    __syncthreads();  // s_slots may still be read from an earlier use
#pragma unroll
    for (int i = 0; i < N; ++i) {
        float a = v[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
        if ((threadIdx.x & 63) == 0) s_slots[wave * N + i] = a;
    }
    __syncthreads();
    const int i = threadIdx.x;
    return i < N ? (s_slots[i] + s_slots[N + i]) + (s_slots[2 * N + i] + s_slots[3 * N + i]) : 0.f;
}

// Deterministic column sum of per-block partials ws[nb][nelem] for element e: 256-thread blocks = 32 columns x 8 interleaved chains, eight
// loads in flight per chain, fixed association order, ONE writer per element (no atomics).  Returns true (with the total) in the writer.
__device__ __forceinline__ bool det_column_sum(const float* __restrict__ ws, int nb, long nelem, long e, float& total) {
    __shared__ float red[8][32];
    const int col = threadIdx.x & 31, chain = threadIdx.x >> 5;
    float s = 0.f;
    if (e < nelem) {
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int b = chain;
        for (; b + 56 < nb; b += 64) {
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] += ws[(long)(b + 8 * u) * nelem + e];
        }
        for (int u = 0; b < nb; b += 8, ++u) a[u & 7] += ws[(long)b * nelem + e];
        s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
    red[chain][col] = s;
    __syncthreads();
    total = ((red[0][col] + red[1][col]) + (red[2][col] + red[3][col])) + ((red[4][col] + red[5][col]) + (red[6][col] + red[7][col]));
    return chain == 0 && e < nelem;
}

// sum over the 16 lanes sharing (lane >> 4) = one DPP row; every lane of the row gets the sum (4 DPP adds, no LDS traffic)
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float quad16_sum(float v) {
    v += dpp_f<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_f<0x141>(v);  // row_half_mirror
    v += dpp_f<0x140>(v);  // row_mirror
    return v;
}

// ----------------------------------------------------------------------------------------------
// MFMA (matrix core) traits.  K is always consumed in chunks of KCH = 32.
//   D[16 x 16] += A[16 x K] * B[K x 16]
//   bf16: v_mfma_f32_16x16x32_bf16   A: lane l holds A[l&15][(l>>4)*8 + 0..7]; B: B[(l>>4)*8+0..7][l&15]
//   fp32: v_mfma_f32_16x16x4_f32 x8  A: lane l holds A[l&15][l>>4];            B: B[l>>4][l&15]  (exact fp32)
//   D  : lane l holds D[(l>>4)*4 + r][l&15], r = 0..3                       (both)
// ----------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));  // operand of the packed fp32 VALU ops (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

static constexpr int KCH = 32;

template <class T>
struct Mma;

template <>
struct Mma<bf16> {
    // LDS tile row: 32 bf16 + 8 pad = 80 B (16-B aligned, conflict-free ds_read_b128 over 16 rows)
    static constexpr int LDS_PITCH = 40;
    // packed weight fragment: 8 bf16 per lane per (chunk, tile) = one 16-B load
    struct Frag {
        uint4 q;
    };
    // fragment from an LDS tile [row][pitch] (row = M or N index, 32 consecutive k along the row);
    // kvalid = number of valid k in this chunk (8,16,32), the rest reads as zero.
    __device__ static __forceinline__ Frag load_p(const bf16* tile, int pitch, int row0, int lane, int kvalid) {
        Frag f;
        const int kg = (lane >> 4) * 8;
        if (kg < kvalid)
            f.q = *reinterpret_cast<const uint4*>(tile + (row0 + (lane & 15)) * pitch + kg);
        else
            f.q = make_uint4(0, 0, 0, 0);
        return f;
    }
    // pre-packed weight fragment (see k_pack_frags): one 16-B load per lane
    __device__ static __forceinline__ Frag load_w(const void* wpk, long frag_idx, int lane) {
        Frag f;
        f.q = reinterpret_cast<const uint4*>(wpk)[frag_idx * 64 + lane];
        return f;
    }
    template <int KS_UNUSED>
    __device__ static __forceinline__ f32x4 mma(const Frag& a, const Frag& b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a.q), __builtin_bit_cast(bf16x8, b.q), c, 0, 0, 0);
    }
};

template <>
struct Mma<float> {
    static constexpr int LDS_PITCH = 36;  // 32 f32 + 4 pad = 144 B (16-B aligned)
    struct Frag {
        float v[8];  // k-steps 0..7 of the chunk
    };
    __device__ static __forceinline__ Frag load_p(const float* tile, int pitch, int row0, int lane, int kvalid) {
        Frag f;
        const float* r = tile + (row0 + (lane & 15)) * pitch + (lane >> 4);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) f.v[ks] = (ks * 4 < kvalid) ? r[ks * 4] : 0.f;
        return f;
    }
    __device__ static __forceinline__ Frag load_w(const void* wpk, long frag_idx, int lane) {
        Frag f;
        const float4* p = reinterpret_cast<const float4*>(wpk) + (frag_idx * 64 + lane) * 2;
        const float4 a = p[0], b = p[1];
        f.v[0] = a.x; f.v[1] = a.y; f.v[2] = a.z; f.v[3] = a.w;
        f.v[4] = b.x; f.v[5] = b.y; f.v[6] = b.z; f.v[7] = b.w;
        return f;
    }
    template <int KS>
    __device__ static __forceinline__ f32x4 mma(const Frag& a, const Frag& b, f32x4 c) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[ks], b.v[ks], c, 0, 0, 0);
        return c;
    }
};

// gfx950 LDS transpose read (ds_read_b64_tr_b16).  Within every group of 16 lanes, lane i supplies the 8-byte-aligned LDS address of 4
// consecutive 16-bit elements = row (i >> 2), columns (i & 3)*4.. of a 4 x 16 block (the row stride is free: it is whatever the lanes'
// addresses say); lane c of the group RECEIVES column c of that block (rows 0..3).  With an LDS tile in the natural NHWC order
// [position][channel] this yields, without any transposing store, the MFMA operand "row = channel, 4 consecutive K = positions" that the
// weight-gradient GEMMs (K = pixels) need.  Semantics verified on hardware by tools/probes/tr_probe.hip.
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ s16x4 lds_tr4(const bf16* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
}
// one 16x16x32 bf16 operand (8 K per lane) from two transpose reads: K = (first 4 rows | second 4 rows)
__device__ __forceinline__ bf16x8 lds_tr8(const bf16* p0, const bf16* p1) {
    const s16x4 a = lds_tr4(p0), b = lds_tr4(p1);
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

// sum over all lanes of the wave with the same (lane % CG), CG in {1,2,4}; every lane gets its class' sum
template <int CG>
__device__ __forceinline__ float lane_class_sum(float v) {
    static_assert(CG == 1 || CG == 2 || CG == 4 || CG == 8, "lane classes");
    if (CG <= 1) v += dpp_f<0xB1>(v);  // xor 1
    if (CG <= 2) v += dpp_f<0x4E>(v);  // xor 2
    if (CG <= 4) v += dpp_f<0x124>(v); // row_ror:4
    v += dpp_f<0x128>(v);              // row_ror:8
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// Workgroup barrier that orders LDS traffic only (s_waitcnt lgkmcnt(0) + s_barrier).  __syncthreads() also drains vmcnt(0), which
// would wait for register-prefetched global loads of the NEXT tile; use this one inside software-pipelined tile loops.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
template <bool LDS_ONLY>
__device__ __forceinline__ void tile_barrier() {
    if constexpr (LDS_ONLY)
        lds_barrier();
    else
        __syncthreads();
}


// XCD-aware persistent tile schedule: block b runs on XCD (b % 8); give each XCD a contiguous
// range of tiles so that neighbouring tiles (which share 3x3 halo rows) hit the same L2.
struct TileSched {
    long first, step, end;
    __device__ TileSched(long ntiles) {
        const long nb = gridDim.x;
        if ((nb & 7) == 0 && ntiles >= nb) {
            const long per = (ntiles + 7) / 8;
            const long xcd = blockIdx.x & 7;
            first = xcd * per + (blockIdx.x >> 3);
            step = nb >> 3;
            end = (xcd + 1) * per < ntiles ? (xcd + 1) * per : ntiles;
        } else {
            first = blockIdx.x;
            step = nb;
            end = ntiles;
        }
    }
};

#include <stdlib.h>
#include <atomic>
// "done once per device" flag for hipFuncSetAttribute(MaxDynamicSharedMemorySize): the attribute is per device, launchers run on the forward
// and on the autograd thread (setting it twice is harmless, skipping it on a second device is not)
struct DevOnce {
    std::atomic<unsigned long long> mask{0};
    static unsigned long long bit() {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
        return 1ull << (dev & 63);
    }
    bool need() const { return !(mask.load(std::memory_order_relaxed) & bit()); }
    void done() { mask.fetch_or(bit(), std::memory_order_relaxed); }
};
// Per-launch timestamps from the dispatch packet (csrc/prof.hip): OCRS_LAUNCH_T == hipLaunchKernelGGL unless ocrs_prof_enable(1)
struct OcrsProf {
    int on, used, cap;
    hipEvent_t* ev;  // [2 * cap]: start / stop pairs
};
OcrsProf& ocrs_prof();
#define OCRS_LAUNCH_T(kernel, grid, block, smem, st, ...)                                                   \
    do {                                                                                                    \
        OcrsProf& pf_ = ocrs_prof();                                                                        \
        if (pf_.on && pf_.used < pf_.cap) {                                                                 \
            hipEvent_t e0_ = pf_.ev[2 * pf_.used], e1_ = pf_.ev[2 * pf_.used + 1];                          \
            ++pf_.used;                                                                                     \
            hipExtLaunchKernelGGL(kernel, grid, block, smem, st, e0_, e1_, 0, __VA_ARGS__);                 \
        } else {                                                                                            \
            hipLaunchKernelGGL(kernel, grid, block, smem, st, __VA_ARGS__);                                 \
        }                                                                                                   \
    } while (0)

static inline int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}
static inline int persistent_grid(long ntiles, int blocks_per_cu) {
    static const int bpc_env = env_int("OCRS_BPC", 0);
    if (bpc_env > 0) blocks_per_cu = bpc_env;
    long cap = (long)kNumCU * blocks_per_cu;
    long g = ntiles < cap ? ntiles : cap;
    if (g >= 8) g &= ~7L;
    return (int)(g < 1 ? 1 : g);
}
