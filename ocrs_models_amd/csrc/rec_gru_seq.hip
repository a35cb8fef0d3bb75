// Bidirectional GRU recurrence as ONE persistent launch per layer and pass (gfx950) -- same arithmetic contract as rec_gru.hip (PyTorch gate
// order r, z, n; reference ocrs_models/models.py:245, 264-266), without the T kernel boundaries.
//
// rec_gru.hip runs one launch per time step: 404 launches per train step of 6.9 (forward) / 8.6 us (backward) each -- 3.1 ms, 36 % of the
// CRNN step -- of which the recurrent GEMM itself is ~1.3 us: the rest is the kernel boundary (L2 write-back + invalidate, ~1.5 us), the
// re-load of W_hh and of the step's operands behind it, and the launch ramp.  Here the T steps run inside one launch:
//   * the recurrence is independent per batch column, so a GROUP of 16 workgroups owns (direction, 32 batch columns): workgroup jt of the
//     group computes hidden units 16 jt .. 16 jt + 15 (3 gates = 3 MFMA M tiles, 2 N tiles of 16 columns) for every step;
//   * its slice of W_hh lives in registers for the whole sequence (K = 256 split over the 8 waves as in rec_gru.hip: 24 registers per wave);
//   * per step the group exchanges h_t (forward) / dgh_t (backward) through global memory with the hand-off recipe of the CDNA4 guide
//     (Guideline 16, form "8-byte agent-scope atomics on both sides"): every value is stored once with an 8-byte agent-scope store (write
//     through), each storing wave drains vmcnt, the workgroup barriers, ONE lane bumps the group's arrival counter; consumers poll that one
//     word relaxed (one lane), barrier, and read the operands with 8-byte agent-scope loads (L1 bypass).  No fences, no dependence on
//     dispatch order or workgroup -> XCD placement (the group -> XCD mapping below is for speed only).  Arrival counters are monotonic
//     (16 * step) and zeroed by a memset node before every launch; a bounded spin raises a caller-owned, sticky error word instead of
//     hanging the GPU (the host side checks it: ocrs_models_amd/recognition.py).
//   * SAME-XCD FAST PATH.  Write-through stores, memory-side atomics and loads that miss the L2 make a hand-off ~4 us -- no better than the
//     kernel boundary it replaces.  The launch places the 16 workgroups of a group on ONE XCD (block b runs on XCD b % 8) and VERIFIES it at
//     run time: every workgroup publishes its HW_REG_XCC_ID with the agent-scope recipe above and reads its 15 peers'.  Only if all 16 agree
//     does the group switch to plain stores and an L2-executed (workgroup-scope) arrival counter: the XCD's L2 is the coherence point of its
//     CUs (L1 is write-through; the consumers' agent-scope loads bypass their L1), so the exchange never leaves the L2.  Any other placement
//     keeps the agent-scope path: results never depend on where the dispatcher puts a workgroup, only the speed does.
//   * the step's own operands (gi; dout, saved gates) are prefetched one step ahead, h_{t-1} of the thread's own (unit, column) pair and the
//     z * dh carry of the backward stay in registers.
// EXACT = true: v_mfma_f32_16x16x4_f32 (the reference's fp32 arithmetic); false: split-bf16 x3 (hi*hi + hi*lo + lo*hi on the bf16 MFMA, fp32
// accumulation: fp32-class, ~2^-17 relative per product, 5x fewer MFMA cycles) -- the same switch as the projection GEMMs (OCRS_GRU_X3).
#include "common.h"
float* rec_defer_partials(int nb, int n, float* out);  // rec_conv.hip (deferred, fixed-order second stage: det_common.h)

#ifndef OCRS_GRU_FAST_ACT
#define OCRS_GRU_FAST_ACT 1  // throughput mode: hardware exp2 / rcp in the gate activations (0: libm, as the exact-fp32 mode always uses)
#endif
#ifndef OCRS_GRU_NT_OUT
#define OCRS_GRU_NT_OUT 1  // forward recurrence: non-temporal stores for out / saved
#endif
#ifndef OCRS_GRU_WAIT_SLEEP
#define OCRS_GRU_WAIT_SLEEP 1  // s_sleep between two polls of a group's arrival counter (x 64 clocks)
#endif
#ifndef OCRS_GRU_POLL_SLEEP
#define OCRS_GRU_POLL_SLEEP 2  // ... between two polls of the tagged exchange words
#endif
namespace {
constexpr int SH = 256, S3 = 768;  // hidden size, 3 gates
constexpr int SNB = 32;            // batch columns per group
constexpr int SNW = 8;             // waves per workgroup
constexpr unsigned SPIN_LIMIT = 1u << 24;

typedef unsigned long long u64;
__device__ __forceinline__ u64 ld_agent(const void* p) { return __hip_atomic_load(reinterpret_cast<const u64*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(void* p, float a, float b) {
    const u64 v = (u64)__float_as_uint(a) | ((u64)__float_as_uint(b) << 32);
    __hip_atomic_store(reinterpret_cast<u64*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// 8 consecutive floats by four 8-byte agent-scope loads
__device__ __forceinline__ void ld8_agent(const float* p, float (&v)[8]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const u64 x = ld_agent(p + 2 * q);
        v[2 * q] = __uint_as_float((unsigned)x);
        v[2 * q + 1] = __uint_as_float((unsigned)(x >> 32));
    }
}
// gate activations.  FAST (throughput mode, split-bf16 products): hardware exp2 / rcp (~1 ulp each, |error| ~ 2e-7 -- two orders below the
// 2^-17 per-product error of that mode); the libm forms cost ~0.2 us of the 2.7 us a forward step takes (the whole step is one dependent
// chain: CRNN step 4.93 -> 4.89 ms).  The exact-fp32 parity mode keeps libm.
template <bool FAST>
__device__ __forceinline__ float sigm(float x) {
    if constexpr (FAST) return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504088896f * x));
    else return 1.f / (1.f + expf(-x));
}
template <bool FAST>
__device__ __forceinline__ float tanh_(float x) {
    if constexpr (FAST) return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(2.88539008177793f * x));
    else return tanhf(x);
}

// hi / lo bf16 split of 8 floats (one 16x16x32 operand each)
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 h = (__bf16)v[i];
        hi[i] = h;
        lo[i] = (__bf16)(v[i] - (float)h);
    }
}

constexpr int SYNC_STRIDE = 32;  // 32-bit words per group: [0] arrival counter, [16 .. 31] XCC ids (+1) of its workgroups; one 128-byte line

// workgroup -> (group, jt): group g on XCD g % 8 (block b is observed to run on XCD b % 8; the grid is padded to 8 x ceil(ngroups / 8) groups
// and workgroups of the padding exit at once).  Speed only: see seq_same_xcd().
__device__ __forceinline__ bool seq_role(int ngroups, int& group, int& jt) {
    const int b = blockIdx.x, slot = b >> 3;
    group = (b & 7) + 8 * (slot >> 4);
    jt = slot & 15;
    return group < ngroups;
}

// one lane, once per launch: publish this workgroup's XCD id, read the 15 peers'; true iff the whole group shares one XCD (= one L2)
__device__ __forceinline__ bool seq_same_xcd(unsigned* gs, int jt, unsigned* err, bool& ok) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc = (xcc & 0xfu) + 1u;  // (0 = not published yet)
    __hip_atomic_store(gs + 16 + jt, xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool same = true;
    ok = true;
    for (int k = 0; k < 16 && ok; ++k) {
        unsigned v = 0;
        for (unsigned spins = 0;; ++spins) {
            v = __hip_atomic_load(gs + 16 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v != 0u) break;
            if ((spins & 1023u) == 1023u) {
                if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = false; break; }
                if (spins >= SPIN_LIMIT) {
                    __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = false;
                    break;
                }
            }
            __builtin_amdgcn_s_sleep(1);
        }
        same = same && v == xcc;
    }
    return same;
}
// the group's exchange stores / arrival signal: agent scope (write-through, memory-side atomic) or, on the verified same-XCD path, L2-local
__device__ __forceinline__ void st_x(bool fast, float* p, float a, float b) {
    if (fast)
        *reinterpret_cast<float2*>(p) = make_float2(a, b);
    else
        st_agent(p, a, b);
}
__device__ __forceinline__ void seq_signal(bool fast, unsigned* cnt) {
    if (fast)
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // (executes in the XCD's L2)
    else
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// one lane: wait until the group's arrival counter reaches `need`; false on timeout / error raised elsewhere
__device__ __forceinline__ bool seq_wait(unsigned* cnt, unsigned need, unsigned* err) {
    for (unsigned spins = 0;; ++spins) {
        if (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need) return true;
        if ((spins & 1023u) == 1023u) {
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
            if (spins >= SPIN_LIMIT) {
                __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
        __builtin_amdgcn_s_sleep(OCRS_GRU_WAIT_SLEEP);
    }
}
}  // namespace

// Exchange workspace (xws): the operand each step hands to the next, in MFMA B-fragment order so that every load instruction of a wave covers
// whole cache lines:  X[group][parity = step & 1][kc = K chunk of 32][nt = 16-column tile][q = 0..3][lane = column (lane & 15) + 16 * kq][2]
// holds element k = 32 kc + 8 kq + 2 q + {0, 1} of column 16 nt + (lane & 15).  Two parities: a workgroup can only run one step ahead of the
// slowest of its group (it waits for all 16 arrivals of step s before it reads), so the buffer it overwrites at step s + 1 was read at step s.
template <int NKC>
__device__ __forceinline__ float* xslot(float* xws, int group, int par, int kc, int nt, int q, int lane) {
    return xws + ((((((long)group * 2 + par) * NKC + kc) * 2 + nt) * 4 + q) * 64 + lane) * 2;
}
// B fragment (8 consecutive k of this lane's column) of chunk kc: four 8-byte agent-scope loads, each instruction 512 contiguous bytes
template <int NKC>
__device__ __forceinline__ void xload(float* xws, int group, int par, int kc, int nt, int lane, float (&v)[8]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const u64 x = ld_agent(xslot<NKC>(xws, group, par, kc, nt, q, lane));
        v[2 * q] = __uint_as_float((unsigned)x);
        v[2 * q + 1] = __uint_as_float((unsigned)(x >> 32));
    }
}

// ---- split-bf16 exchange (EXACT = false): the data is the flag.  A value travels as ONE 32-bit word (hi bf16 << 16 | lo bf16) whose lowest bit
// -- the last mantissa bit of lo, 2^-17 of the value -- carries a tag that alternates between successive writes of the same slot
// (tag = ~(step >> 1) & 1; the workspace is preset to zero and the first two writes carry 1).  Consumers simply load their fragment and
// re-load until every word shows the tag of the step they need: no drain, no barrier, no arrival counter, no second round trip for a flag
// (Guideline 16, form R2: single aligned word per granule, so a reader can never pair a new tag with an old value).  The split is done once
// by the producer instead of by each of the 16 consumers.
__device__ __forceinline__ unsigned pack_hilo(float v, unsigned tag) {
    const __bf16 h = (__bf16)v;
    const __bf16 l = (__bf16)(v - (float)h);
    return ((unsigned)__builtin_bit_cast(unsigned short, h) << 16) | ((unsigned)__builtin_bit_cast(unsigned short, l) & 0xfffeu) | tag;
}
// 8 packed words -> the hi and lo MFMA operands
__device__ __forceinline__ void unpack_hilo(const unsigned (&w)[8], bf16x8& hi, bf16x8& lo) {
    unsigned h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h[i] = (w[2 * i] >> 16) | (w[2 * i + 1] & 0xffff0000u);
        l[i] = (w[2 * i] & 0xffffu) | (w[2 * i + 1] << 16);
    }
    hi = __builtin_bit_cast(bf16x8, make_uint4(h[0], h[1], h[2], h[3]));
    lo = __builtin_bit_cast(bf16x8, make_uint4(l[0], l[1], l[2], l[3]));
}
// this wave's fragments of NC consecutive chunks, polled until every word carries `tag`; false on timeout / error elsewhere
template <int NKC, int NC>
__device__ __forceinline__ bool xpoll(float* xws, int group, int par, int kc0, int nt, int lane, unsigned tag, unsigned* err, unsigned (&w)[NC][8]) {
    for (unsigned spins = 0;; ++spins) {
        unsigned bad = 0;
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const u64 x = ld_agent(xslot<NKC>(xws, group, par, kc0 + c, nt, q, lane));
                w[c][2 * q] = (unsigned)x;
                w[c][2 * q + 1] = (unsigned)(x >> 32);
                bad |= (w[c][2 * q] ^ tag) | (w[c][2 * q + 1] ^ tag);
            }
        if (!__any((int)(bad & 1u))) return true;
        if ((spins & 255u) == 255u) {
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
            if (spins >= (SPIN_LIMIT >> 2)) {
                __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
        __builtin_amdgcn_s_sleep(OCRS_GRU_POLL_SLEEP);
    }
}
// The forward's bulk outputs of a step (out, saved: 265 MB per launch) as NON-TEMPORAL stores: plain stores stream them through the L2 the
// exchange workspace lives in -- 2.50 -> 2.00 us per step.  (Measured and not used: the same for the backward's dgi / dgh stores (4.14 -> 4.17-4.29),
// non-temporal loads of gi (2.01 -> 2.55) and of the backward's operands (no change).)
__device__ __forceinline__ void nt_store(float* p, float v) {
#if OCRS_GRU_NT_OUT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
__device__ __forceinline__ void nt_store2(float* p, float a, float b) {
    typedef float f2v __attribute__((ext_vector_type(2)));
#if OCRS_GRU_NT_OUT
    __builtin_nontemporal_store((f2v){a, b}, reinterpret_cast<f2v*>(p));
#else
    *reinterpret_cast<float2*>(p) = make_float2(a, b);
#endif
}
__device__ __forceinline__ void st_xw(bool fast, float* p, unsigned a, unsigned b) { st_x(fast, p, __uint_as_float(a), __uint_as_float(b)); }
// the same split without a tag (counter hand-offs: the backward kernel): hi = bf16(v), lo = bf16(v - hi), exactly what split8 computes --
// done ONCE by the producer instead of by each of its 16 consumers (48 values x ~4 VALU per lane and step, on the step's critical chain)
__device__ __forceinline__ unsigned pack_hilo_full(float v) {
    const __bf16 h = (__bf16)v;
    const __bf16 l = (__bf16)(v - (float)h);
    return ((unsigned)__builtin_bit_cast(unsigned short, h) << 16) | (unsigned)__builtin_bit_cast(unsigned short, l);
}
// 8 packed words -> the hi and lo MFMA operands: one v_perm_b32 per pair and plane
__device__ __forceinline__ void unpack_hilo_perm(const float (&w)[8], bf16x8& hi, bf16x8& lo) {
    unsigned h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned a = __float_as_uint(w[2 * i]), b = __float_as_uint(w[2 * i + 1]);
        h[i] = __builtin_amdgcn_perm(b, a, 0x07060302u);  // (a >> 16) | (b & 0xffff0000)
        l[i] = __builtin_amdgcn_perm(b, a, 0x05040100u);  // (a & 0xffff) | (b << 16)
    }
    hi = __builtin_bit_cast(bf16x8, make_uint4(h[0], h[1], h[2], h[3]));
    lo = __builtin_bit_cast(bf16x8, make_uint4(l[0], l[1], l[2], l[3]));
}

// gi [T][N][2*768] (b_ih added), whh [2][768][256] fp32 master, bhh [2][768], out [T][N][512], saved [T][N][2][4][256] (nullable)
// sync: per group SYNC_STRIDE words (zeroed before the launch); err: sticky error word (set when a wait times out); xws: exchange workspace.
// Waves: wave = 4 nt + kk owns the 16-column tile nt and the K quarter kk (chunks 2 kk, 2 kk + 1) for all three gates.
template <bool EXACT>
__global__ __launch_bounds__(512, 2) void k_gru_seq_fwd(const float* __restrict__ gi, const float* __restrict__ whh, const float* __restrict__ bhh,
                                                        float* __restrict__ out, float* __restrict__ saved, int T, int N, unsigned* sync, unsigned* err,
                                                        float* xws, int ngroups, int try_fast) {
    __shared__ float red[EXACT ? 1 : 2][4][3][2][16][17];  // (tagged exchange: one barrier per step, so the partials alternate between two buffers)
    __shared__ int s_ok, s_fast, s_fail;
    int group, jt;
    if (!seq_role(ngroups, group, jt)) return;
    const int d = group & 1, b0 = (group >> 1) * SNB;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kq = lane >> 4, wnt = wave >> 2, kk = wave & 3;
    unsigned* cnt = sync + group * SYNC_STRIDE;
    if (tid == 0) {
        bool ok;
        const bool same = seq_same_xcd(cnt, jt, err, ok);
        s_ok = ok ? 1 : 0;
        s_fast = (same && try_fast) ? 1 : 0;
        s_fail = 0;
    }
    __syncthreads();
    if (!s_ok) return;
    const bool fast = s_fast != 0;

    // this wave's slice of W_hh: rows (gate g, unit 16 jt + l15), K = 64 kk + 32 c + 8 kq .. + 7
    float wf[EXACT ? 3 : 1][EXACT ? 2 : 1][8];
    bf16x8 whi[EXACT ? 1 : 3][EXACT ? 1 : 2], wlo[EXACT ? 1 : 3][EXACT ? 1 : 2];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float* wr = whh + ((long)d * S3 + g * SH + jt * 16 + l15) * SH + kk * 64 + c * 32 + kq * 8;
            float v[8];
            const float4 a = *reinterpret_cast<const float4*>(wr), b = *reinterpret_cast<const float4*>(wr + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
            if constexpr (EXACT) {
#pragma unroll
                for (int i = 0; i < 8; ++i) wf[g][c][i] = v[i];
            } else {
                split8(v, whi[g][c], wlo[g][c]);
            }
        }
    // epilogue role: one (unit, column) pair per thread; its exchange slot: k = j -> chunk jt >> 1, position 16 (jt & 1) + jl
    const int jl = tid & 15, bl = tid >> 4;
    const int b = b0 + bl, j = jt * 16 + jl;
    const bool bv = b < N;
    const int xkc = jt >> 1, xpos = (jt & 1) * 16 + jl, xlane = (bl & 15) + 16 * (xpos >> 3), xq = (xpos & 7) >> 1, xnt = bl >> 4;
    const float bh_r = bhh[d * S3 + j], bh_z = bhh[d * S3 + SH + j], bh_n = bhh[d * S3 + 2 * SH + j];
    float gi_r = 0.f, gi_z = 0.f, gi_n = 0.f, hp = 0.f;
    float gn_r = 0.f, gn_z = 0.f, gn_n = 0.f;  // the NEXT step's gi (loaded a whole step ahead)
    auto load_gi = [&](int t, float& r, float& z, float& n) {
        if (bv) {
            const float* gir = gi + ((long)t * N + b) * (2 * S3) + d * S3 + j;
            r = gir[0];
            z = gir[SH];
            n = gir[2 * SH];
        }
    };
    load_gi(d == 0 ? 0 : T - 1, gi_r, gi_z, gi_n);
    // Everything a step stores or loads besides the exchange -- out / saved of the step, gi of the next -- is issued right BEHIND the wait for the
    // peers' h (deferred by one step), not in front of it: the price of a hand-off sits in the consumer CU's own memory queue
    // (MI355X_MICROARCH.md, handoff-1to1: unloaded -> unloaded 1.1 us, -> loaded 2.5 us), and the polls used to queue behind these:
    // 2.76 -> 2.50 us per step standalone.  (Issued behind the step's MFMAs instead: 2.58.  The same deferral in the backward kernel, whose
    // publish drains vmcnt(0) in front of its arrival signal, makes every step wait for the deferred stores' acknowledgements: 4.0 -> 8 us.)
    float p_rv = 0.f, p_zv = 0.f, p_nv = 0.f, p_hn = 0.f, p_hv = 0.f, p_hv1 = 0.f;
    auto store_prev = [&](int s) {  // out / saved of step s - 1
        if (s > 0 && bv) {
            const int tp = d == 0 ? s - 1 : T - s;
            if ((jl & 1) == 0) nt_store2(out + ((long)tp * N + b) * 512 + d * SH + j, p_hv, p_hv1);
            if (saved) {
                float* sv = saved + (((long)tp * N + b) * 2 + d) * 4 * SH + j;
                nt_store(sv, p_rv);
                nt_store(sv + SH, p_zv);
                nt_store(sv + 2 * SH, p_nv);
                nt_store(sv + 3 * SH, p_hn);
            }
        }
    };
    auto after_wait = [&](int s) {  // tagged hand-off, s = the step that is starting: the previous step's stores, the next step's gi
        store_prev(s);
        if (s + 1 < T) load_gi(d == 0 ? s + 1 : T - 2 - s, gn_r, gn_z, gn_n);
    };
    constexpr bool FASTACT = !EXACT && OCRS_GRU_FAST_ACT != 0;

#ifdef OCRS_GRU_SEQ_PROF
    unsigned long long pt[6] = {0, 0, 0, 0, 0, 0}, pc = __builtin_readcyclecounter();
#define PROF_MARK(i) { const unsigned long long now = __builtin_readcyclecounter(); pt[i] += now - pc; pc = now; }
#else
#define PROF_MARK(i)
#endif
    for (int s = 0; s < T; ++s) {
        const int t = d == 0 ? s : T - 1 - s;
        float gh[3] = {0.f, 0.f, 0.f};
        if (!EXACT && s == 0) after_wait(0);
        if (s > 0) {
            f32x4 acc[3];
#pragma unroll
            for (int g = 0; g < 3; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
            // recurrent GEMM: gh[gate][unit][column] += W_hh[., K quarter] h_{t-1}[column][K quarter]
            if constexpr (EXACT) {
                if (tid == 0) s_ok = seq_wait(cnt, 16u * (unsigned)s, err) ? 1 : 0;
                __syncthreads();
                PROF_MARK(0)
                if (!s_ok) return;  // (uniform)
                float hb[2][8];
#pragma unroll
                for (int c = 0; c < 2; ++c) xload<8>(xws, group, (s - 1) & 1, 2 * kk + c, wnt, lane, hb[c]);
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int i = 0; i < 8; ++i)
#pragma unroll
                        for (int g = 0; g < 3; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[g][c][i], hb[c][i], acc[g], 0, 0, 0);
            } else {
                unsigned w[2][8];
                if (!xpoll<8, 2>(xws, group, (s - 1) & 1, 2 * kk, wnt, lane, (unsigned)((((s - 1) >> 1) & 1) ^ 1), err, w)) s_fail = 1;
                PROF_MARK(0)
                after_wait(s);
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    bf16x8 hhi, hlo;
                    unpack_hilo(w[c], hhi, hlo);
#pragma unroll
                    for (int g = 0; g < 3; ++g) {
                        f32x4 cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo[g][c], hhi, acc[g], 0, 0, 0);
                        cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi[g][c], hlo, cc, 0, 0, 0);
                        acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi[g][c], hhi, cc, 0, 0, 0);
                    }
                }
            }
            const int rb = EXACT ? 0 : (s & 1);
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[rb][kk][g][wnt][kq * 4 + r][l15] = acc[g][r];
            __syncthreads();
            PROF_MARK(2)
            if constexpr (!EXACT) {
                if (s_fail) return;  // (uniform: written before the barrier)
            }
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const float* rp = &red[rb][0][g][bl >> 4][jl][bl & 15];
                constexpr int WS = 3 * 2 * 16 * 17;
                gh[g] = (rp[0] + rp[WS]) + (rp[2 * WS] + rp[3 * WS]);
            }
        }
        const float rv = sigm<FASTACT>(gi_r + gh[0] + bh_r);
        const float zv = sigm<FASTACT>(gi_z + gh[1] + bh_z);
        const float hn = gh[2] + bh_n;
        const float nv = tanh_<FASTACT>(gi_n + rv * hn);
        const float hv = bv ? (1.f - zv) * nv + zv * hp : 0.f;
        hp = hv;
        const float hv1 = __shfl_down(hv, 1);  // unit j + 1 of the same column (adjacent lane)
        PROF_MARK(3)
        if (s + 1 < T) {
            if constexpr (EXACT) {
                // publish h_t first (the only store the peers wait for): exchange store, drain, barrier, one lane signals
                if ((jl & 1) == 0) st_x(fast, xslot<8>(xws, group, s & 1, xkc, xnt, xq, xlane), hv, hv1);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) seq_signal(fast, cnt);
            } else {
                const unsigned w0 = pack_hilo(hv, (unsigned)(((s >> 1) & 1) ^ 1));
                const unsigned w1 = __shfl_down(w0, 1);
                if ((jl & 1) == 0) st_xw(fast, xslot<8>(xws, group, s & 1, xkc, xnt, xq, xlane), w0, w1);
            }
        }
        PROF_MARK(4)
        p_rv = rv; p_zv = zv; p_nv = nv; p_hn = hn; p_hv = hv; p_hv1 = hv1;
        if constexpr (EXACT) {  // counter hand-off: its publish drains vmcnt(0) -- nothing may be pending in front of it, so no deferral
            store_prev(s + 1);
            if (s + 1 < T) load_gi(d == 0 ? s + 1 : T - 2 - s, gi_r, gi_z, gi_n);  // next step's operands: in flight during the wait
        } else {
            gi_r = gn_r; gi_z = gn_z; gi_n = gn_n;
        }
        PROF_MARK(5)
    }
    if constexpr (!EXACT) after_wait(T);  // (the last step's out / saved)
#ifdef OCRS_GRU_SEQ_PROF
    if (tid == 0 && jt == 0) {
        cnt[1] = fast ? 1u : 0u;
        for (int i = 0; i < 6; ++i) cnt[2 + i] = (unsigned)(pt[i] / (unsigned long long)T);
    }
#endif
}

// BPTT.  dout [T][N][512], saved / out from the forward, whh [2][768][256] master; dgi, dgh [T][N][1536] (gradients w.r.t. gi and gh).
// Step s: direction 0 processes t = T-1-s, direction 1 processes t = s;  dh = dout[t] + z * dh (carry) + W_hh^T dgh[previous step].
// Waves: wave = 4 nt + kk owns the column tile nt and K chunks 6 kk .. 6 kk + 5 of the 24 (K = 768 gate rows).
// (Both arithmetic modes use the arrival-counter hand-off here: with 3x the forward's exchange volume a poll pass over the wave's 24 fragment
// loads is so long that the tagged-word form measured SLOWER -- 455 vs 407 us per launch at T = 101, N = 256.)
template <bool EXACT>
__global__ __launch_bounds__(512, 2) void k_gru_seq_bwd(const float* __restrict__ dout, const float* __restrict__ saved, const float* __restrict__ out,
                                                        const float* __restrict__ whh, float* __restrict__ dgi, float* __restrict__ dgh, int T, int N,
                                                        unsigned* sync, unsigned* err, float* xws, int ngroups, int try_fast,
                                                        float* __restrict__ dbih, float* __restrict__ dbhh, float* __restrict__ ws_ih, float* __restrict__ ws_hh) {
    // ws_ih / ws_hh (nullable, round 5): per-batch-group partials [ngroups / 2][2 * S3] of the bias gradients instead of the cross-group float atomics,
    // summed in a fixed order by the deferred reduce launch (rec_defer_partials)
    __shared__ float red[4][2][16][17];
    __shared__ int s_ok, s_fast;
    int group, jt;
    if (!seq_role(ngroups, group, jt)) return;
    const int d = group & 1, b0 = (group >> 1) * SNB;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kq = lane >> 4, wnt = wave >> 2, kk = wave & 3;
    unsigned* cnt = sync + group * SYNC_STRIDE;
    if (tid == 0) {
        bool ok;
        const bool same = seq_same_xcd(cnt, jt, err, ok);
        s_ok = ok ? 1 : 0;
        s_fast = (same && try_fast) ? 1 : 0;
    }
    __syncthreads();
    if (!s_ok) return;
    const bool fast = s_fast != 0;

    // this wave's slice of W_hh^T: rows = unit 16 jt + l15, K = gate rows 192 kk + 32 c + 8 kq .. + 7
    float wf[EXACT ? 6 : 1][8];
    bf16x8 whi[EXACT ? 1 : 6], wlo[EXACT ? 1 : 6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = whh[((long)d * S3 + kk * 192 + c * 32 + kq * 8 + i) * SH + jt * 16 + l15];
        if constexpr (EXACT) {
#pragma unroll
            for (int i = 0; i < 8; ++i) wf[c][i] = v[i];
        } else {
            split8(v, whi[c], wlo[c]);
        }
    }
    const int jl = tid & 15, bl = tid >> 4;
    const int b = b0 + bl, j = jt * 16 + jl;
    const bool bv = b < N;
    // exchange slot of gate g: k = 256 g + j -> chunk 8 g + (jt >> 1)
    const int xkc = jt >> 1, xpos = (jt & 1) * 16 + jl, xlane = (bl & 15) + 16 * (xpos >> 3), xq = (xpos & 7) >> 1, xnt = bl >> 4;
    float e_dout = 0.f, e_r = 0.f, e_z = 0.f, e_n = 0.f, e_hn = 0.f, e_hp = 0.f, carry = 0.f;
    float sb_r = 0.f, sb_z = 0.f, sb_n = 0.f, sb_nr = 0.f;  // this (unit, column)'s share of the bias gradients: the gate gradients summed over time
    auto load_ep = [&](int t) {
        if (bv) {
            e_dout = dout[((long)t * N + b) * 512 + d * SH + j];
            const float* sv = saved + (((long)t * N + b) * 2 + d) * 4 * SH + j;
            e_r = sv[0];
            e_z = sv[SH];
            e_n = sv[2 * SH];
            e_hn = sv[3 * SH];
            const int tp = d == 0 ? t - 1 : t + 1;
            e_hp = (tp >= 0 && tp < T) ? out[((long)tp * N + b) * 512 + d * SH + j] : 0.f;
        }
    };
    load_ep(d == 0 ? T - 1 : 0);

#ifdef OCRS_GRU_SEQ_PROF
    unsigned long long pt[6] = {0, 0, 0, 0, 0, 0}, pc = __builtin_readcyclecounter();
#endif
    for (int s = 0; s < T; ++s) {
        const int t = d == 0 ? T - 1 - s : s;
        float dh = e_dout;
        if (s > 0) {
            if (tid == 0) s_ok = seq_wait(cnt, 16u * (unsigned)s, err) ? 1 : 0;
            __syncthreads();
            PROF_MARK(0)
            if (!s_ok) return;
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                float gb[3][8];
#pragma unroll
                for (int c = 0; c < 3; ++c) xload<24>(xws, group, (s - 1) & 1, 6 * kk + 3 * half + c, wnt, lane, gb[c]);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int cw = 3 * half + c;
                    if constexpr (EXACT) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[cw][i], gb[c][i], acc, 0, 0, 0);
                    } else {
                        bf16x8 ghi, glo;
                        unpack_hilo_perm(gb[c], ghi, glo);  // (the producers published hi | lo words)
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo[cw], ghi, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi[cw], glo, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi[cw], ghi, acc, 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) red[kk][wnt][kq * 4 + r][l15] = acc[r];
            PROF_MARK(1)
            __syncthreads();
            PROF_MARK(2)
            const float* rp = &red[0][bl >> 4][jl][bl & 15];
            constexpr int WS = 2 * 16 * 17;
            dh += carry + ((rp[0] + rp[WS]) + (rp[2 * WS] + rp[3 * WS]));
        }
        float dn_pre = dh * (1.f - e_z) * (1.f - e_n * e_n);
        float dz = dh * (e_hp - e_n) * e_z * (1.f - e_z);
        float dr = dn_pre * e_hn * e_r * (1.f - e_r);
        float dnr = dn_pre * e_r;
        if (!bv) dn_pre = dz = dr = dnr = 0.f;
        sb_r += dr;
        sb_z += dz;
        sb_n += dn_pre;
        sb_nr += dnr;
        carry = dh * e_z;
        const float dr1 = __shfl_down(dr, 1), dz1 = __shfl_down(dz, 1), dnr1 = __shfl_down(dnr, 1);
        PROF_MARK(3)
        if (s + 1 < T) {
            if constexpr (EXACT) {
                if ((jl & 1) == 0) {
                    st_x(fast, xslot<24>(xws, group, s & 1, xkc, xnt, xq, xlane), dr, dr1);
                    st_x(fast, xslot<24>(xws, group, s & 1, 8 + xkc, xnt, xq, xlane), dz, dz1);
                    st_x(fast, xslot<24>(xws, group, s & 1, 16 + xkc, xnt, xq, xlane), dnr, dnr1);
                }
            } else {
                const unsigned wr = pack_hilo_full(dr), wz = pack_hilo_full(dz), wn = pack_hilo_full(dnr);
                const unsigned wr1 = __shfl_down(wr, 1), wz1 = __shfl_down(wz, 1), wn1 = __shfl_down(wn, 1);
                if ((jl & 1) == 0) {
                    st_xw(fast, xslot<24>(xws, group, s & 1, xkc, xnt, xq, xlane), wr, wr1);
                    st_xw(fast, xslot<24>(xws, group, s & 1, 8 + xkc, xnt, xq, xlane), wz, wz1);
                    st_xw(fast, xslot<24>(xws, group, s & 1, 16 + xkc, xnt, xq, xlane), wn, wn1);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) seq_signal(fast, cnt);
        }
        PROF_MARK(4)
        if (bv) {
            float* gi_ = dgi + ((long)t * N + b) * (2 * S3) + d * S3 + j;
            gi_[0] = dr;
            gi_[SH] = dz;
            gi_[2 * SH] = dn_pre;
            if ((jl & 1) == 0) {
                float* gh_ = dgh + ((long)t * N + b) * (2 * S3) + d * S3 + j;
                *reinterpret_cast<float2*>(gh_) = make_float2(dr, dr1);
                *reinterpret_cast<float2*>(gh_ + SH) = make_float2(dz, dz1);
                *reinterpret_cast<float2*>(gh_ + 2 * SH) = make_float2(dnr, dnr1);
            }
        }
        if (s + 1 < T) load_ep(d == 0 ? T - 2 - s : s + 1);
        PROF_MARK(5)
    }
#ifdef OCRS_GRU_SEQ_PROF
    if (tid == 0 && jt == 0) {
        cnt[1] = fast ? 1u : 0u;
        for (int i = 0; i < 6; ++i) cnt[2 + i] = (unsigned)(pt[i] / (unsigned long long)T);
    }
#endif
    // bias gradients (nullable): db_ih = column sums of dgi, db_hh = column sums of dgh over (t, n) -- the two 159-MB re-reads of dgi / dgh by
    // k_col_sum4 (0.17 ms per CRNN step) are not needed.  Lanes -> the wave's four batch columns (lanes 16 apart), waves through LDS in a fixed
    // order, then one fp32 atomic per (group, unit, gate) -- 8 groups per direction at N = 256 (like k_col_sum4's per-block atomics).
    if (dbih || dbhh) {
        float v[4] = {sb_r, sb_z, sb_n, sb_nr};
        __syncthreads();  // (red is free: every wave is past its last read)
        float* rs = &red[0][0][0][0];  // [8 waves][4 sums][16 units]
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            v[q] += __shfl_xor(v[q], 16, 64);
            v[q] += __shfl_xor(v[q], 32, 64);
            if (lane < 16) rs[(wave * 4 + q) * 16 + lane] = v[q];
        }
        __syncthreads();
        if (tid < 64) {
            const int q = tid >> 4, u = tid & 15;
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) a += rs[(w * 4 + q) * 16 + u];
            const int col = d * S3 + jt * 16 + u;
            const long prow = (long)(group >> 1) * (2 * S3);
            auto acc = [&](float* out, float* ws, int c) {
                if (ws) ws[prow + c] = a;
                else atomicAdd(&out[c], a);
            };
            if (q < 3) {
                if (dbih) acc(dbih, ws_ih, col + q * SH);
                if (dbhh && q < 2) acc(dbhh, ws_hh, col + q * SH);
            } else if (dbhh) {
                acc(dbhh, ws_hh, col + 2 * SH);
            }
        }
    }
}

static int seq_groups(int N) { return 2 * ((N + SNB - 1) / SNB); }
static int seq_grid(int N) { return 8 * ((seq_groups(N) + 7) / 8) * 16; }  // padded to whole rounds of the 8 XCDs
static int seq_try_fast() { return env_int("OCRS_GRU_SEQ_FAST", 1); }  // (read per call: the tests compare both paths in one process)

template <class K>
static bool seq_fits(K kernel, int nblocks) {
    int dev = 0, ncu = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 512, 0) != hipSuccess) return false;
    return (long)ncu * per_cu >= nblocks;  // every workgroup must be resident: they wait for each other
}

extern "C" {

// 1 if the persistent recurrence can run for N batch columns on the current device (all 32 * ceil(N / 32) workgroups co-resident)
long ocrs_gru_seq_supported(int N) {
    static const int on = env_int("OCRS_GRU_SEQ", 1);
    if (!on || N <= 0) return 0;
    static int cap = -1;  // resident workgroups the device holds of the most demanding of the four kernels
    if (cap < 0) {
        cap = 0;
        for (int nb = 16; nb <= 4096; nb += 16) {
            if (!(seq_fits(k_gru_seq_fwd<true>, nb) && seq_fits(k_gru_seq_bwd<true>, nb) && seq_fits(k_gru_seq_fwd<false>, nb) && seq_fits(k_gru_seq_bwd<false>, nb))) break;
            cap = nb;
        }
    }
    // Head-room: the launch only deadlocks-until-timeout if some workgroup cannot become resident while its peers spin.  Other streams'
    // kernels may hold workgroup slots at that moment -- the DDP bucketer's RCCL all-reduce (<= 32 channels = workgroups by default, 64 with
    // NCCL_MAX_NCHANNELS raised) starts right before the backward recurrence -- so the grid must fit with that many slots to spare
    // (OCRS_GRU_SEQ_HEADROOM, default 64).  N = 256 needs 256 of the 512 slots an MI355X offers these kernels.
    static const int headroom = env_int("OCRS_GRU_SEQ_HEADROOM", 64);
    return seq_grid(N) + headroom <= cap;
}
long ocrs_gru_seq_sync_words(int N) { return (long)seq_groups(N) * SYNC_STRIDE; }
// floats of the exchange workspace of one launch (the backward needs 3x the forward: size for the backward, both passes take it)
long ocrs_gru_seq_ws_floats(int N) { return (long)seq_groups(N) * 2 * 24 * 2 * 4 * 64 * 2; }

// Recurrent part of one bidirectional GRU layer, all T steps in one launch.  whh: the fp32 master [2][768][256] (no fragment packing);
// sync: ocrs_gru_seq_sync_words(N) 32-bit words (zeroed here);  xws: ocrs_gru_seq_ws_floats(N) floats (exchange workspace, initialised here);
// err: ONE caller-owned 32-bit word, zeroed by the caller once and sticky: set
// when a wait inside the launch timed out (outputs incomplete) -- check it with ocrs_gru_seq_status or from the host side at a convenient
// point;  exact != 0: fp32 MFMA, 0: split-bf16 x3.  Other arguments as ocrs_gru_layer_fwd.  Returns OCRS_ERR_ARG when the grid cannot be
// co-resident (use the per-step entry point then).
int ocrs_gru_seq_fwd(const float* gi, const float* whh, const float* bhh, float* out, float* saved, int T, int N, unsigned* sync, unsigned* err, float* xws,
                     int exact, hipStream_t st) {
    OCRS_CHECK_ARG(gi && whh && bhh && out && sync && err && xws && T > 0 && N > 0 && ocrs_gru_seq_supported(N));
    const int ng = seq_groups(N);
    const size_t sync_bytes = (size_t)ng * SYNC_STRIDE * sizeof(unsigned);
    const size_t tag_bytes = exact ? 0 : (size_t)ng * 2 * 8 * 2 * 4 * 64 * 2 * sizeof(float);  // split-bf16 form: the exchange words carry tags
    if (reinterpret_cast<char*>(xws) == reinterpret_cast<char*>(sync) + sync_bytes) {            // one fill when the two are one allocation
        if (hipMemsetAsync(sync, 0, sync_bytes + tag_bytes, st) != hipSuccess) return OCRS_ERR_HIP;
    } else {
        if (hipMemsetAsync(sync, 0, sync_bytes, st) != hipSuccess) return OCRS_ERR_HIP;
        if (tag_bytes && hipMemsetAsync(xws, 0, tag_bytes, st) != hipSuccess) return OCRS_ERR_HIP;
    }
    if (exact)
        hipLaunchKernelGGL(k_gru_seq_fwd<true>, dim3(seq_grid(N)), dim3(512), 0, st, gi, whh, bhh, out, saved, T, N, sync, err, xws, ng, seq_try_fast());
    else
        hipLaunchKernelGGL(k_gru_seq_fwd<false>, dim3(seq_grid(N)), dim3(512), 0, st, gi, whh, bhh, out, saved, T, N, sync, err, xws, ng, seq_try_fast());
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

int ocrs_gru_seq_bwd(const float* dout, const float* saved, const float* out, const float* whh, float* dgi, float* dgh, int T, int N, unsigned* sync,
                     unsigned* err, float* xws, int exact, float* dbih, float* dbhh, hipStream_t st) {
    OCRS_CHECK_ARG(dout && saved && out && whh && dgi && dgh && sync && err && xws && T > 0 && N > 0 && ocrs_gru_seq_supported(N));
    const int ng = seq_groups(N);
    if (hipMemsetAsync(sync, 0, (size_t)ng * SYNC_STRIDE * sizeof(unsigned), st) != hipSuccess) return OCRS_ERR_HIP;
    // deferring (ocrs_bwd_defer_begin): the bias gradients as per-batch-group partials + a queued fixed-order sum instead of float atomics
    float* ws_ih = dbih ? rec_defer_partials(ng / 2, 2 * S3, dbih) : nullptr;
    float* ws_hh = dbhh ? rec_defer_partials(ng / 2, 2 * S3, dbhh) : nullptr;
    if (exact)
        hipLaunchKernelGGL(k_gru_seq_bwd<true>, dim3(seq_grid(N)), dim3(512), 0, st, dout, saved, out, whh, dgi, dgh, T, N, sync, err, xws, ng, seq_try_fast(), dbih, dbhh, ws_ih, ws_hh);
    else
        hipLaunchKernelGGL(k_gru_seq_bwd<false>, dim3(seq_grid(N)), dim3(512), 0, st, dout, saved, out, whh, dgi, dgh, T, N, sync, err, xws, ng, seq_try_fast(), dbih, dbhh, ws_ih, ws_hh);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

// the sticky error word (non-zero: a wait timed out in some launch since it was zeroed -- outputs incomplete); synchronises the stream
int ocrs_gru_seq_status(const unsigned* err, hipStream_t st) {
    unsigned e = 0;
    if (hipMemcpyAsync(&e, err, sizeof(unsigned), hipMemcpyDeviceToHost, st) != hipSuccess) return OCRS_ERR_HIP;
    if (hipStreamSynchronize(st) != hipSuccess) return OCRS_ERR_HIP;
    return e ? OCRS_ERR_HIP : OCRS_OK;
}

}  // extern "C"
