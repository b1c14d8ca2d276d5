// Shared device/host helpers for libdesenet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>

#include "../../include/desenet_hip.h"

typedef __bf16 bf16_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// ---- error plumbing (thread-local message, never abort) -------------------------------------------------------
void dsn_set_error(const char* fmt, ...);
#define DSN_FAIL(code, ...)      \
    do {                         \
        dsn_set_error(__VA_ARGS__); \
        return (code);           \
    } while (0)
#define DSN_CHECK_ARG(cond, ...) \
    do {                         \
        if (!(cond)) DSN_FAIL(DSN_EINVAL, __VA_ARGS__); \
    } while (0)
#define DSN_LAUNCH_CHECK(what)                                                             \
    do {                                                                                   \
        hipError_t e_ = hipGetLastError();                                                 \
        if (e_ != hipSuccess) DSN_FAIL((int)e_, "%s: %s", what, hipGetErrorString(e_));    \
    } while (0)

// Opt a kernel into more than 64 KB of dynamic LDS: once per (kernel, device ordinal), checked.  Returns DSN_OK or the HIP error
// (message set): a launch site returns it instead of failing later with "invalid argument" at launch time.
int dsn_lds_attr(const void* kern, int bytes);
// Cache policy of every LDS-DMA load (last operand of __builtin_amdgcn_raw_ptr_buffer_load_lds; gfx950: 1 = sc0, 2 = nt, 16 = sc1).
// The counted s_waitcnt vmcnt schedules need DMAs to complete in issue order, which holds for loads served from L2 / HBM but not for
// hits in the CU's vector L1 (tools/exp/oob_order.hip).
#ifndef DSN_DMA_AUX
#define DSN_DMA_AUX 0
#endif
#define DSN_LDS_ATTR(kern, bytes)                                         \
    do {                                                                  \
        const int a_ = dsn_lds_attr((const void*)(kern), (bytes));        \
        if (a_ != DSN_OK) return a_;                                      \
    } while (0)

// ---- tensor view as passed to kernels -------------------------------------------------------------------------
struct TV {
    void*   p;
    int32_t n, h, w, c;
    int64_t ldc;
};
static inline TV tv(const dsn_tensor* t) { return TV{t->ptr, t->n, t->h, t->w, t->c, t->ldc}; }
static inline int64_t npix(const dsn_tensor* t) { return (int64_t)t->n * t->h * t->w; }
static inline bool tensor_ok(const dsn_tensor* t) {
    return t && t->ptr && (t->dtype == DSN_F32 || t->dtype == DSN_BF16) && t->n > 0 && t->h > 0 && t->w > 0 &&
           t->c > 0 && t->ldc >= t->c;
}

// ---- element load/store in fp32 domain ------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

#if defined(__HIPCC__)
// ---- 16-byte (or scalar) channel vectors ---------------------------------------------------------------------------------
template <typename T, int V> struct VecIO {   // generic / scalar
    __device__ static __forceinline__ void load(const T* p, float (&o)[V]) {
#pragma unroll
        for (int i = 0; i < V; ++i) o[i] = to_f32<T>(p[i]);
    }
    __device__ static __forceinline__ void store(T* p, const float (&o)[V]) {
#pragma unroll
        for (int i = 0; i < V; ++i) p[i] = from_f32<T>(o[i]);
    }
};
// store(): non-temporal by default (resampling / copy kernels); a translation unit that defines DSN_VECIO_PLAIN before including
// this header gets ordinary stores (bn_act.hip does: its outputs are re-read by the next kernel, and on the 50-200 MB tensors of
// the 1280^2 configuration that is worth 0.9 % of the step).  DSN_VECIO_NTLOAD likewise turns load() into a non-temporal load (inputs
// that are read exactly once).
template <> struct VecIO<float, 4> {
    __device__ static __forceinline__ void load(const float* p, float (&o)[4]) {
#ifdef DSN_VECIO_NTLOAD
        const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
#else
        const f32x4 v = *reinterpret_cast<const f32x4*>(p);
#endif
        o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
    }
    __device__ static __forceinline__ void store(float* p, const float (&o)[4]) {
#ifdef DSN_VECIO_PLAIN
        *reinterpret_cast<f32x4*>(p) = f32x4{o[0], o[1], o[2], o[3]};
#else
        __builtin_nontemporal_store(f32x4{o[0], o[1], o[2], o[3]}, reinterpret_cast<f32x4*>(p));
#endif
    }
};
template <> struct VecIO<bf16_t, 8> {
    __device__ static __forceinline__ void load(const bf16_t* p, float (&o)[8]) {
#ifdef DSN_VECIO_NTLOAD
        const bf16x8 v = __builtin_bit_cast(bf16x8, __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)));
#else
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#endif
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (float)v[i];
    }
    __device__ static __forceinline__ void store(bf16_t* p, const float (&o)[8]) {
        bf16x8 v;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (bf16_t)o[i];
#ifdef DSN_VECIO_PLAIN
        *reinterpret_cast<u32x4*>(p) = __builtin_bit_cast(u32x4, v);
#else
        __builtin_nontemporal_store(__builtin_bit_cast(u32x4, v), reinterpret_cast<u32x4*>(p));
#endif
    }
};
template <typename T> struct VW { static constexpr int N = 16 / sizeof(T); };

#endif

// v_exp_f32 + v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE division: these kernels are ALU-latency bound on the
// small maps (one wave per SIMD), and 1 ulp is far inside the fp32 parity tolerance (1e-3 relative).
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == DSN_ACT_SILU) return v * sigmoidf_(v);
    if (act == DSN_ACT_SIGMOID) return sigmoidf_(v);
    return v;
}
// d act(u) / du
__device__ __forceinline__ float act_grad(float u, int act) {
    if (act == DSN_ACT_SILU) {
        float s = sigmoidf_(u);
        return s * (1.0f + u * (1.0f - s));
    }
    if (act == DSN_ACT_SIGMOID) {
        float s = sigmoidf_(u);
        return s * (1.0f - s);
    }
    return 1.0f;
}
// The same two functions over a register vector with the (wave-uniform) activation code tested ONCE.  Inside a per-element loop the
// test is a scalar branch per element: it cuts the loop into basic blocks, so every element's v_exp -> v_rcp chain runs alone with
// its latency exposed (measured: the backward apply pass was ALU-bound at 2.3 TB/s on 13-26 MB tensors).  Same arithmetic per
// element, bit-identical results.
template <int V> __device__ __forceinline__ void apply_act_vec(float (&v)[V], int act) {
    if (act == DSN_ACT_SILU) {
#pragma unroll
        for (int k = 0; k < V; ++k) v[k] = v[k] * sigmoidf_(v[k]);
    } else if (act == DSN_ACT_SIGMOID) {
#pragma unroll
        for (int k = 0; k < V; ++k) v[k] = sigmoidf_(v[k]);
    }
}
template <int V> __device__ __forceinline__ void act_grad_vec(const float (&u)[V], int act, float (&g)[V]) {
    if (act == DSN_ACT_SILU) {
#pragma unroll
        for (int k = 0; k < V; ++k) { const float s = sigmoidf_(u[k]); g[k] = s * (1.0f + u[k] * (1.0f - s)); }
    } else if (act == DSN_ACT_SIGMOID) {
#pragma unroll
        for (int k = 0; k < V; ++k) { const float s = sigmoidf_(u[k]); g[k] = s * (1.0f - s); }
    } else {
#pragma unroll
        for (int k = 0; k < V; ++k) g[k] = 1.0f;
    }
}


// ---- per-channel reductions that cross blocks ---------------------------------------------------------------------------
// BatchNorm needs per-channel sums over ALL pixels before anything downstream can run.  The producing kernel (conv epilogue
// or a reduction sweep) adds each block's per-channel partials into fp64 accumulators with hardware atomics
// (global_atomic_add_f64, no return value: measured free next to the kernel's own traffic), BN_NREP replicas picked by
// block index keep same-address contention low.  The CONSUMING kernel folds the replicas in its prologue -- every block
// redundantly, 2 channels per thread out of L2 -- so there is no finalize launch and no cross-block handshake at all.
// (Measured alternatives on MI355X: a finalize launch costs ~6 us per layer; "last block finalizes" costs ~7 us of serial
// device-scope round trips -- ticket, fold, store -- at the kernel's tail, and an agent-scope __threadfence() per block
// ~19 us because it writes the XCD's L2 back.)
// Contract: the accumulators are ZERO when the producer starts; nobody restores them (callers hand out slices of an arena
// that is cleared once per step).  fp64 sums of fp32 partials: the addition order can only show up below 1e-13 relative.
constexpr int BN_NREP = 8;
struct BnAcc {
    double* acc;          // [BN_NREP][2][C]
    int32_t C;
    double  count;        // pixels per channel
};
static inline int64_t bn_acc_bytes(int c) { return (int64_t)BN_NREP * 2 * c * sizeof(double); }

#if defined(__HIPCC__)
__device__ __forceinline__ void bn_acc_add(const BnAcc& f, int rep, int c, float s0, float s1) {
    double* a = f.acc + (size_t)(rep & (BN_NREP - 1)) * 2 * f.C;
    unsafeAtomicAdd(a + c, (double)s0);
    unsafeAtomicAdd(a + f.C + c, (double)s1);
}
// fold the replicas of channel c (the producing kernel has completed: plain loads)
__device__ __forceinline__ void bn_acc_fold(const BnAcc& f, int c, double& s, double& ss) {
    s = 0.0; ss = 0.0;
#pragma unroll
    for (int r = 0; r < BN_NREP; ++r) {
        const double* a = f.acc + (size_t)r * 2 * f.C;
        s += a[c];
        ss += a[f.C + c];
    }
}
#endif

// ---- BatchNorm + activation from accumulators (dsn_lazy_materialize's operand descriptor) -----------------------------------------
// A training-mode Conv block leaves its RAW convolution result y plus the per-channel fp64 sums in its BnAcc; ONE elementwise launch
// folds the accumulators in its prologue and writes z = act(y * scale + shift) (+ shortcut).  A tensor may be a concat of several
// producers: up to DSN_LAZY_MAXSEG channel segments.  (Rounds 2-3 also applied the transform inside the CONSUMING convolution's
// operand loader; measured slower and deleted in round 4 -- DESIGN.md 8.)
//   acc != NULL : forward, before the producer's statistics have been finalised -- the consumer folds the accumulators itself
//                 (every block, redundantly, exactly as ew_prologue does: identical scale / shift bits)
//   acc == NULL : scale / shift arrays (written by dsn_bn_finalize_multi at the end of the forward pass) -- backward (wgrad)
//   both NULL   : identity segment
// Channel c of the consumer's input inside [c0, c1) is accumulator channel ch0 + (c - c0) and parameter index p0 + (c - c0).
#if defined(__HIPCC__)
typedef dsn_lazy_seg LazySeg;
typedef dsn_lazy_in LazyIn;
// per-channel (scale, shift) of segment s for its local channel k (0-based inside the segment)
__device__ __forceinline__ void lazy_fold(const LazySeg& s, int k, float& sc, float& sh) {
    if (s.acc) {
        const double* a0 = (const double*)s.acc;
        double sum = 0.0, ssq = 0.0;
#pragma unroll
        for (int r = 0; r < BN_NREP; ++r) {
            const double* a = a0 + (size_t)r * 2 * s.acc_c;
            sum += a[s.ch0 + k];
            ssq += a[s.acc_c + s.ch0 + k];
        }
        const double mean = sum / s.count;
        double var = ssq / s.count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)s.eps));
        const float g = s.gamma ? s.gamma[s.p0 + k] : 1.f, b = s.beta ? s.beta[s.p0 + k] : 0.f;
        sc = g * rstd;
        sh = b - (float)mean * sc;
    } else if (s.scale) {
        sc = s.scale[s.p0 + k];
        sh = s.shift[s.p0 + k];
    } else {
        sc = 1.f;
        sh = 0.f;
    }
}
// fill sc[0..C), sh[0..C) and act8[0..C/8) (activation code per 8-channel group) in LDS; all threads of the block call it
__device__ __forceinline__ void lazy_table(const LazyIn& lz, int C, float* sc, float* sh, unsigned char* act8, int nthreads) {
    for (int c = threadIdx.x; c < C; c += nthreads) {
        float a = 1.f, b = 0.f;
        int act = DSN_ACT_NONE;
        for (int s = 0; s < lz.nseg; ++s)
            if (c >= lz.seg[s].c0 && c < lz.seg[s].c1) {
                lazy_fold(lz.seg[s], c - lz.seg[s].c0, a, b);
                act = lz.seg[s].act;
            }
        sc[c] = a;
        sh[c] = b;
        if ((c & 7) == 0) act8[c >> 3] = (unsigned char)act;
    }
}
#endif

// ---- BatchNorm backward sums in a dgrad epilogue (include/desenet_hip.h: dsn_bnred) -------------------------------------------------
// Every thread of the vectorised tile store owns ONE channel vector (256 % vectors-per-row == 0) over several rows: it keeps the two
// partial sums of its VEC channels in registers while it stores, then the block folds them (lanes of equal channel vector by
// shuffles, the four waves through LDS) and adds them to the producer's fp64 accumulators -- the same accumulators, replica
// scheme and consumer (ew_prologue mode 1) as the stand-alone reduction.
typedef dsn_bnred BnRed;
// channels of y a launch with backward sums reads per output pixel (algorithmic bytes of the launch)
static inline double bnred_channels(const BnRed* br) {
    double c = 0.0;
    if (br) for (int i = 0; i < br->nseg; ++i) c += br->seg[i].c1 - br->seg[i].c0;
    return c;
}
#if defined(__HIPCC__)
template <typename T, int VEC, int NIT> struct BnRedLane {
    const T* yp;
    int64_t yld;
    float sc[VEC], sh[VEC], mu[VEC], rs[VEC], q0[VEC], q1[VEC];
    u32x4 yv[NIT];          // y at the NIT vectors this thread stores: fetched BEFORE the tile is staged, consumed after its stores
    int act;
    bool on;
    __device__ __forceinline__ void init(const BnRed& br, int ch) {       // ch: first channel of this thread's vector
        on = false;
#pragma unroll
        for (int k = 0; k < VEC; ++k) q0[k] = q1[k] = 0.f;
        for (int s = 0; s < br.nseg; ++s) {
            const dsn_bnred_seg& g = br.seg[s];
            if (ch >= g.c0 && ch < g.c1) {
                on = true;
                const int k0 = ch - g.c0;
                yp = (const T*)g.y + k0;
                yld = g.yld;
                act = g.act;
#pragma unroll
                for (int k = 0; k < VEC; k += 4) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(g.scale + k0 + k), b = *reinterpret_cast<const f32x4*>(g.shift + k0 + k);
                    const f32x4 c = *reinterpret_cast<const f32x4*>(g.mean + k0 + k), d = *reinterpret_cast<const f32x4*>(g.rstd + k0 + k);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { sc[k + e] = a[e]; sh[k + e] = b[e]; mu[k + e] = c[e]; rs[k + e] = d[e]; }
                }
            }
        }
    }
    __device__ __forceinline__ void prefetch(int it, int64_t row) {      // row < 0: this vector is not stored
        if (on && row >= 0) yv[it] = *reinterpret_cast<const u32x4*>(yp + row * yld);
    }
    // outv: the values just stored (rounded to T -- what the apply pass will read as dz)
    __device__ __forceinline__ void add(int it, const T (&outv)[VEC]) {
        if (!on) return;
        T yl[VEC];
        *reinterpret_cast<u32x4*>(yl) = yv[it];
        float u[VEC], gr[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) u[k] = to_f32<T>(yl[k]) * sc[k] + sh[k];
        act_grad_vec<VEC>(u, act, gr);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const float y = to_f32<T>(yl[k]);
            const float gk = to_f32<T>(outv[k]) * gr[k];
            q0[k] += gk;
            q1[k] += gk * ((y - mu[k]) * rs[k]);
        }
    }
    // red: >= 8 * VPR * VEC floats of LDS nobody else is using; n0: first GEMM column of the tile; ncol: GEMM columns;
    // chmod > 0: GEMM column -> channel is col % chmod (depth-to-space store of the stride-2 dgrad)
    template <int VPR> __device__ __forceinline__ void finish(const BnRed& br, float* red, int rep, int n0, int ncol, int chmod) {
        constexpr int BNW = VPR * VEC;
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
        for (int o = VPR; o < 64; o <<= 1) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                q0[k] += __shfl_xor(q0[k], o);
                q1[k] += __shfl_xor(q1[k], o);
            }
        }
        if (lane < VPR) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                red[(wave * BNW + lane * VEC + k) * 2] = q0[k];
                red[(wave * BNW + lane * VEC + k) * 2 + 1] = q1[k];
            }
        }
        __syncthreads();
        if (tid < BNW && n0 + tid < ncol) {
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                s0 += red[(w * BNW + tid) * 2];
                s1 += red[(w * BNW + tid) * 2 + 1];
            }
            const int ch = chmod > 0 ? (n0 + tid) % chmod : n0 + tid;
            for (int s = 0; s < br.nseg; ++s) {
                const dsn_bnred_seg& g = br.seg[s];
                if (ch >= g.c0 && ch < g.c1) bn_acc_add(BnAcc{(double*)g.acc, g.acc_c, 0.0}, rep, g.ch0 + ch - g.c0, s0, s1);
            }
        }
    }
};
#endif

#if defined(__HIPCC__)
// 32-bit fill as a KERNEL.  hipMemsetAsync nodes captured into a hipGraph were observed (ROCm 7.2, gfx950) not to take effect
// reliably on replay for small, 4-byte-aligned ranges inside a larger workspace (stale contents -> garbage indices), so the
// library never relies on memset nodes.
static __global__ void dsn_fill_u32_kernel(uint32_t* __restrict__ p, uint32_t v, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}
static __global__ void dsn_fill_u32x4_kernel(u32x4* __restrict__ p, uint32_t v, int64_t nv, uint32_t* __restrict__ tail, int64_t ntail) {
    const u32x4 vv = {v, v, v, v};
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) p[i] = vv;
    if (blockIdx.x == 0 && (int64_t)threadIdx.x < ntail) tail[threadIdx.x] = v;
}
static inline void dsn_fill_u32(void* p, uint32_t v, int64_t n_words, hipStream_t st) {
    if (n_words <= 0) return;
    int64_t b = (n_words + 255) / 256;
    hipLaunchKernelGGL(dsn_fill_u32_kernel, dim3((unsigned)(b > 1024 ? 1024 : b)), dim3(256), 0, st, (uint32_t*)p, v, n_words);
}
#endif

// conv3x3.hip: the halo-tile kernel for 3x3 / stride-1 convolutions (forward and data gradient).  Returns 1 when the layer is not
// one it takes (nothing launched), 0 when it ran, another status on failure.  finp: BatchNorm accumulators for the epilogue or NULL.
struct BnAcc;
int dsn_conv3x3_halo_try(const dsn_tensor* s, const void* w, const float* bias, const dsn_tensor* r, const dsn_tensor* d,
                         const dsn_conv_params* p, int is_dgrad, const BnAcc* finp, void* stream, const dsn_bnred* br = nullptr);

int dsn_conv1x1_dma_try(const dsn_tensor* s, const void* w, const float* bias, const dsn_tensor* r, const dsn_tensor* d,
                        const dsn_conv_params* p, int is_dgrad, const BnAcc* finp, void* stream, const dsn_bnred* br = nullptr);

// conv_ws.hip: weights-stationary persistent kernels, tried before the one-trip kernels above (same return convention)
int dsn_conv1x1_ws_try(const dsn_tensor* s, const void* w, const float* bias, const dsn_tensor* r, const dsn_tensor* d,
                       const dsn_conv_params* p, int is_dgrad, const BnAcc* finp, void* stream, const dsn_bnred* br = nullptr);

int dsn_conv3x3_ws_try(const dsn_tensor* s, const void* w, const float* bias, const dsn_tensor* r, const dsn_tensor* d,
                       const dsn_conv_params* p, int is_dgrad, const BnAcc* finp, void* stream, const dsn_bnred* br = nullptr);

// conv_pp.hip: 256-pixel x 128 / 256-channel block tiles, two wave groups in ping-pong (bf16, 3x3 / stride 1 / dilation 1); tried
// first (same return convention)
int dsn_conv3x3_pp_try(const dsn_tensor* s, const void* w, const float* bias, const dsn_tensor* r, const dsn_tensor* d,
                       const dsn_conv_params* p, int is_dgrad, const BnAcc* finp, void* stream, const dsn_bnred* br = nullptr);

// conv_pp.hip: the same schedule for 1x1 / stride 1 (256 consecutive pixels x 128 / 256 channels, both operands through the ring)
int dsn_conv1x1_pp_try(const dsn_tensor* s, const void* w, const float* bias, const dsn_tensor* r, const dsn_tensor* d,
                       const dsn_conv_params* p, int is_dgrad, const BnAcc* finp, void* stream, const dsn_bnred* br = nullptr);

// conv_pp.hip: the 3x3 / stride-2 data gradient (2x2 form, depth-to-space store) on the same kernel
int dsn_dgrad_s2_pp_try(const dsn_tensor* dy, const void* w_s2, const dsn_tensor* dx, const dsn_conv_params* p, const dsn_bnred* br,
                        void* stream);

// the gather forms of the 1x1 kernel: 3x3 / stride 2 / pad 1 forward, and its data gradient (2x2 form, depth-to-space store)
int dsn_conv3x3s2_ws_try(const dsn_tensor* s, const void* w, const float* bias, const dsn_tensor* r, const dsn_tensor* d,
                         const dsn_conv_params* p, const BnAcc* finp, void* stream);
int dsn_dgrad_s2_ws_try(const dsn_tensor* dy, const void* w_s2, const dsn_tensor* dx, const dsn_conv_params* p, const dsn_bnred* br,
                        void* stream);

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

#define DSN_DISPATCH_DTYPE(dt, T, ...)                 \
    do {                                               \
        if ((dt) == DSN_F32) {                         \
            typedef float T;                           \
            __VA_ARGS__;                               \
        } else {                                       \
            typedef bf16_t T;                          \
            __VA_ARGS__;                               \
        }                                              \
    } while (0)

// ---- optional live profiler: HIP events recorded on the launch stream around the hot kernels (bench.py roofline) ----
enum {
    KID_IGEMM = 0,            // + dtype*20 + cfg*2 + is_dgrad   (cfg 0..4 -> 128x128, 128x64, 64x64, 128x32, 64x16)
    KID_WGRAD = 40,           // + dtype
    KID_WGRAD_REDUCE = 42,
    KID_BN_STATS = 43,
    KID_BN_ACT_FWD = 44,
    KID_BN_BWD_REDUCE = 45,
    KID_BN_BWD_APPLY = 46,
    KID_COUNT = 47
};
struct ProfScope {
    int slot;
    hipStream_t st;
    ProfScope(int kid, double flops, double bytes, hipStream_t stream);
    // labelled form: `label` = "<rocprofv3 symbol family>/<dtype>/<tile>/<fwd|dgrad>", `layer` = shape key of the launch
    // (both copied; built by the launch site only while the profiler is on: dsn_prof_on())
    ProfScope(const char* label, const char* layer, double flops, double bytes, hipStream_t stream);
    ~ProfScope();
};
bool dsn_prof_on();
// label / layer strings of a convolution launch (sym: kernel symbol family as rocprofv3 --kernel-trace prints it)
struct ProfConv {
    char label[64], layer[64];
    ProfConv(const char* sym, bool bf16, int bm, int bn, bool dgrad, int k, int stride, int dil, int cs, int cd, int n, int h, int w) {
        label[0] = layer[0] = 0;
        if (!dsn_prof_on()) return;
        snprintf(label, sizeof(label), "%s/%s/%dx%d/%s", sym, bf16 ? "bf16" : "f32", bm, bn, dgrad ? "dgrad" : "fwd");
        snprintf(layer, sizeof(layer), "k%ds%dd%d %d->%d @%dx%dx%d", k, stride, dil, cs, cd, n, h, w);
    }
};
