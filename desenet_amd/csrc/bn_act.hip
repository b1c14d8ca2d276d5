// BatchNorm (training statistics, apply, backward) + activation kernels over NHWC [P][C] views.  All HBM-bound streams.
//
// Layout of work: a 256-thread block owns a strip of pixels; thread (ty, tx) owns ONE 16-byte channel vector (4 fp32 /
// 8 bf16 channels) -- so its per-channel constants (scale, shift, mean, rstd, ...) are loaded once into registers -- and
// walks the strip's pixels ty, ty+TY, ... with FOUR independent 16-byte loads in flight per operand (memory-level
// parallelism: these tensors are 1-13 MB, the kernels live or die by latency).  Consecutive lanes read consecutive channel
// vectors of one pixel, then the next pixel: fully coalesced whenever ldc == C, 16-byte segments otherwise (concat slices).
//
// Per-channel reductions fold the block's TY partials through LDS and write ONE row of per-block partials; a finalize
// kernel (one wave per channel) reduces the rows in fp64 in a fixed order -> deterministic, no float atomics.
#include "common.h"

namespace {

constexpr int THREADS = 256;
constexpr int UNROLL = 4;
constexpr int MAX_RED_BLOCKS = 1024;

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
template <> struct VecIO<float, 4> {
    __device__ static __forceinline__ void load(const float* p, float (&o)[4]) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p);
        o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
    }
    __device__ static __forceinline__ void store(float* p, const float (&o)[4]) {
        *reinterpret_cast<f32x4*>(p) = f32x4{o[0], o[1], o[2], o[3]};
    }
};
template <> struct VecIO<bf16_t, 8> {
    __device__ static __forceinline__ void load(const bf16_t* p, float (&o)[8]) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (float)v[i];
    }
    __device__ static __forceinline__ void store(bf16_t* p, const float (&o)[8]) {
        bf16x8 v;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (bf16_t)o[i];
        *reinterpret_cast<bf16x8*>(p) = v;
    }
};
template <typename T> struct VW { static constexpr int N = 16 / sizeof(T); };

// ---- functors ----------------------------------------------------------------------------------------------------------
// Reductions: accumulate two per-channel quantities from (a = in0, b = in1).
template <int V> struct StatsF {            // sum y, sum y^2
    __device__ __forceinline__ void prepare(int) {}
    __device__ __forceinline__ void acc(const float (&a)[V], const float (&)[V], float (&q0)[V], float (&q1)[V]) const {
#pragma unroll
        for (int k = 0; k < V; ++k) { q0[k] += a[k]; q1[k] += a[k] * a[k]; }
    }
};
struct BnParams { const float *scale, *shift, *mean, *rstd, *means; int act, C; };
template <int V> struct BwdRedF {           // sum g, sum g*yhat with g = dz * act'(y*scale+shift)
    BnParams p;
    float sc[V], sh[V], mu[V], rs[V];
    __device__ __forceinline__ void prepare(int c0) {
#pragma unroll
        for (int k = 0; k < V; ++k) { sc[k] = p.scale[c0 + k]; sh[k] = p.shift[c0 + k]; mu[k] = p.mean[c0 + k]; rs[k] = p.rstd[c0 + k]; }
    }
    __device__ __forceinline__ void acc(const float (&y)[V], const float (&dz)[V], float (&q0)[V], float (&q1)[V]) const {
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float g = dz[k] * act_grad(y[k] * sc[k] + sh[k], p.act);
            q0[k] += g;
            q1[k] += g * ((y[k] - mu[k]) * rs[k]);
        }
    }
};
// Elementwise: o = f(a, b)
template <int V> struct FwdF {              // act(y*scale + shift) [+ res]
    BnParams p; bool has_res;
    float sc[V], sh[V];
    __device__ __forceinline__ void prepare(int c0) {
#pragma unroll
        for (int k = 0; k < V; ++k) { sc[k] = p.scale ? p.scale[c0 + k] : 1.f; sh[k] = p.scale ? p.shift[c0 + k] : 0.f; }
    }
    __device__ __forceinline__ void apply(const float (&y)[V], const float (&r)[V], float (&o)[V]) const {
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float u = apply_act(y[k] * sc[k] + sh[k], p.act);
            o[k] = has_res ? u + r[k] : u;
        }
    }
};
template <int V> struct BwdApplyF {         // BN: scale*(g - mean_g - yhat*mean_gyhat);  no BN: dz*act'(y)
    BnParams p;
    float sc[V], sh[V], mu[V], rs[V], m1[V], m2[V];
    __device__ __forceinline__ void prepare(int c0) {
        if (!p.scale) return;
#pragma unroll
        for (int k = 0; k < V; ++k) {
            sc[k] = p.scale[c0 + k]; sh[k] = p.shift[c0 + k]; mu[k] = p.mean[c0 + k]; rs[k] = p.rstd[c0 + k];
            m1[k] = p.means[c0 + k]; m2[k] = p.means[p.C + c0 + k];
        }
    }
    __device__ __forceinline__ void apply(const float (&y)[V], const float (&dz)[V], float (&o)[V]) const {
#pragma unroll
        for (int k = 0; k < V; ++k) {
            if (p.scale) {
                const float g = dz[k] * act_grad(y[k] * sc[k] + sh[k], p.act);
                o[k] = sc[k] * (g - m1[k] - ((y[k] - mu[k]) * rs[k]) * m2[k]);
            } else {
                o[k] = dz[k] * act_grad(y[k], p.act);
            }
        }
    }
};

struct Strip { int64_t P; int C; int64_t per_block; };

// ---- reduction kernel ---------------------------------------------------------------------------------------------------
template <typename T, int V, bool HAS_B, typename F>
__global__ __launch_bounds__(THREADS) void reduce2_kernel(const T* __restrict__ a, int64_t ald, const T* __restrict__ b,
                                                          int64_t bld, Strip s, float* __restrict__ partial, F f) {
    __shared__ float red[2 * THREADS * V];
    const int ncv = s.C / V;
    const int TX = ncv < THREADS ? ncv : THREADS;
    const int TY = THREADS / TX;
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int64_t p0 = blockIdx.x * s.per_block;
    const int64_t p1 = (p0 + s.per_block < s.P) ? p0 + s.per_block : s.P;
    float* out = partial + (int64_t)blockIdx.x * 2 * s.C;
    for (int cv0 = 0; cv0 < ncv; cv0 += TX) {
        const int cv = cv0 + tx;
        float q0[V], q1[V];
#pragma unroll
        for (int k = 0; k < V; ++k) q0[k] = q1[k] = 0.f;
        if (ty < TY && cv < ncv) {
            f.prepare(cv * V);
            const T* ap = a + cv * V;
            const T* bp = HAS_B ? b + cv * V : nullptr;
            int64_t p = p0 + ty;
            for (; p + (UNROLL - 1) * TY < p1; p += UNROLL * TY) {
                float va[UNROLL][V], vb[UNROLL][V];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    VecIO<T, V>::load(ap + (p + u * TY) * ald, va[u]);
                    if (HAS_B) VecIO<T, V>::load(bp + (p + u * TY) * bld, vb[u]);
                }
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) f.acc(va[u], vb[u], q0, q1);
            }
            for (; p < p1; p += TY) {
                float va[V], vb[V];
                VecIO<T, V>::load(ap + p * ald, va);
                if (HAS_B) VecIO<T, V>::load(bp + p * bld, vb);
                f.acc(va, vb, q0, q1);
            }
        }
        __syncthreads();
        if (ty < TY) {
#pragma unroll
            for (int k = 0; k < V; ++k) {
                red[(ty * TX + tx) * V + k] = q0[k];
                red[THREADS * V + (ty * TX + tx) * V + k] = q1[k];
            }
        }
        __syncthreads();
        const int cbase = cv0 * V, cnum = ((ncv - cv0 < TX) ? (ncv - cv0) : TX) * V;
        for (int c = threadIdx.x; c < cnum; c += THREADS) {
            const int ctx = c / V, ck = c % V;
            float s0 = 0.f, s1 = 0.f;
            for (int t = 0; t < TY; ++t) {
                s0 += red[(t * TX + ctx) * V + ck];
                s1 += red[THREADS * V + (t * TX + ctx) * V + ck];
            }
            out[cbase + c] = s0;
            out[s.C + cbase + c] = s1;
        }
    }
}

// ---- elementwise kernel -------------------------------------------------------------------------------------------------
template <typename T, int V, bool HAS_B, typename F>
__global__ __launch_bounds__(THREADS) void ew2_kernel(const T* __restrict__ a, int64_t ald, const T* __restrict__ b,
                                                      int64_t bld, T* __restrict__ o, int64_t old_, Strip s, F f) {
    const int ncv = s.C / V;
    const int TX = ncv < THREADS ? ncv : THREADS;
    const int TY = THREADS / TX;
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    if (ty >= TY) return;
    const int64_t p0 = blockIdx.x * s.per_block;
    const int64_t p1 = (p0 + s.per_block < s.P) ? p0 + s.per_block : s.P;
    for (int cv = tx; cv < ncv; cv += TX) {
        f.prepare(cv * V);
        const T* ap = a + cv * V;
        const T* bp = HAS_B ? b + cv * V : nullptr;
        T* op = o + cv * V;
        int64_t p = p0 + ty;
        for (; p + (UNROLL - 1) * TY < p1; p += UNROLL * TY) {
            float va[UNROLL][V], vb[UNROLL][V], vo[V];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                VecIO<T, V>::load(ap + (p + u * TY) * ald, va[u]);
                if (HAS_B) VecIO<T, V>::load(bp + (p + u * TY) * bld, vb[u]);
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                f.apply(va[u], vb[u], vo);
                VecIO<T, V>::store(op + (p + u * TY) * old_, vo);
            }
        }
        for (; p < p1; p += TY) {
            float va[V], vb[V], vo[V];
            VecIO<T, V>::load(ap + p * ald, va);
            if (HAS_B) VecIO<T, V>::load(bp + p * bld, vb);
            f.apply(va, vb, vo);
            VecIO<T, V>::store(op + p * old_, vo);
        }
    }
}

// ---- finalize kernels: one wave per channel folds the per-block rows in fp64 ------------------------------------------------
__device__ __forceinline__ void fold_rows(const float* __restrict__ partial, int nblocks, int C, int c, int lane,
                                          double& s, double& ss) {
    s = 0.0; ss = 0.0;
    for (int b = lane; b < nblocks; b += 64) {
        s += (double)partial[(int64_t)b * 2 * C + c];
        ss += (double)partial[(int64_t)b * 2 * C + C + c];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_down(s, off);
        ss += __shfl_down(ss, off);
    }
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ partial, int nblocks, int C,
                                                          double count, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta,
                                                          float* __restrict__ running_mean,
                                                          float* __restrict__ running_var, float momentum, float eps,
                                                          float* __restrict__ scale, float* __restrict__ shift,
                                                          float* __restrict__ mean_o, float* __restrict__ rstd_o) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    double s, ss;
    fold_rows(partial, nblocks, C, c, lane, s, ss);
    if (lane != 0) return;
    const double mean = s / count;
    double var = ss / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float sc = g * rstd;
    scale[c] = sc;
    shift[c] = b - (float)mean * sc;
    mean_o[c] = (float)mean;
    rstd_o[c] = rstd;
    if (running_mean) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int nblocks, int C,
                                                              double count, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, int accumulate,
                                                              float* __restrict__ means) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    double s, ss;
    fold_rows(partial, nblocks, C, c, lane, s, ss);
    if (lane != 0) return;
    if (dbeta) dbeta[c] = accumulate ? dbeta[c] + (float)s : (float)s;
    if (dgamma) dgamma[c] = accumulate ? dgamma[c] + (float)ss : (float)ss;
    means[c] = (float)(s / count);
    means[C + c] = (float)(ss / count);
}

// ---- host helpers ---------------------------------------------------------------------------------------------------------
inline bool vec_ok(const dsn_tensor* t) {
    const int vw = t->dtype == DSN_F32 ? 4 : 8;
    return t->c % vw == 0 && t->ldc % vw == 0 && ((uintptr_t)t->ptr % 16) == 0;
}
inline bool same_shape(const dsn_tensor* a, const dsn_tensor* b) {
    return a->n == b->n && a->h == b->h && a->w == b->w && a->c == b->c && a->dtype == b->dtype;
}
// strips: enough blocks to fill the chip, each thread one UNROLL-deep sweep when the tensor is large enough
inline Strip make_strip(int64_t P, int C, int V, int max_blocks, int* nblocks) {
    const int ncv = C / V;
    const int TX = ncv < THREADS ? ncv : THREADS, TY = THREADS / TX;
    int64_t per = (int64_t)TY * UNROLL;
    int64_t nb = (P + per - 1) / per;
    if (nb > max_blocks) {
        per = (P + max_blocks - 1) / max_blocks;
        per = (per + TY - 1) / TY * TY;
        nb = (P + per - 1) / per;
    }
    *nblocks = (int)(nb < 1 ? 1 : nb);
    return Strip{P, C, per};
}

template <typename T, bool HAS_B, template <int> class F, typename... A>
void launch_reduce(bool vec, const dsn_tensor* a, const dsn_tensor* b, float* partial, int* nblocks, hipStream_t st,
                   A... args) {
    constexpr int VV = VW<T>::N;
    const int64_t P = npix(a);
    if (vec) {
        Strip s = make_strip(P, a->c, VV, MAX_RED_BLOCKS, nblocks);
        hipLaunchKernelGGL((reduce2_kernel<T, VV, HAS_B, F<VV>>), dim3(*nblocks), dim3(THREADS), 0, st, (const T*)a->ptr,
                           a->ldc, b ? (const T*)b->ptr : nullptr, b ? b->ldc : 0, s, partial, F<VV>{args...});
    } else {
        Strip s = make_strip(P, a->c, 1, MAX_RED_BLOCKS, nblocks);
        hipLaunchKernelGGL((reduce2_kernel<T, 1, HAS_B, F<1>>), dim3(*nblocks), dim3(THREADS), 0, st, (const T*)a->ptr,
                           a->ldc, b ? (const T*)b->ptr : nullptr, b ? b->ldc : 0, s, partial, F<1>{args...});
    }
}

template <typename T, bool HAS_B, template <int> class F, typename... A>
void launch_ew(bool vec, const dsn_tensor* a, const dsn_tensor* b, const dsn_tensor* o, hipStream_t st, A... args) {
    constexpr int VV = VW<T>::N;
    const int64_t P = npix(a);
    int nb;
    if (vec) {
        Strip s = make_strip(P, a->c, VV, 16384, &nb);
        hipLaunchKernelGGL((ew2_kernel<T, VV, HAS_B, F<VV>>), dim3(nb), dim3(THREADS), 0, st, (const T*)a->ptr, a->ldc,
                           b ? (const T*)b->ptr : nullptr, b ? b->ldc : 0, (T*)o->ptr, o->ldc, s, F<VV>{args...});
    } else {
        Strip s = make_strip(P, a->c, 1, 16384, &nb);
        hipLaunchKernelGGL((ew2_kernel<T, 1, HAS_B, F<1>>), dim3(nb), dim3(THREADS), 0, st, (const T*)a->ptr, a->ldc,
                           b ? (const T*)b->ptr : nullptr, b ? b->ldc : 0, (T*)o->ptr, o->ldc, s, F<1>{args...});
    }
}

}  // namespace

extern "C" int64_t dsn_bn_workspace_bytes(int32_t c) { return (int64_t)(MAX_RED_BLOCKS + 1) * 2 * c * sizeof(float); }

extern "C" int dsn_bn_stats(const dsn_tensor* y, const float* gamma, const float* beta, float* running_mean,
                            float* running_var, float momentum, float eps, float* scale, float* shift, float* mean,
                            float* rstd, void* workspace, int64_t workspace_bytes, void* stream) {
    DSN_CHECK_ARG(tensor_ok(y) && scale && shift && mean && rstd && workspace, "bn_stats: null argument");
    DSN_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "bn_stats: running stats must come in pairs");
    if (workspace_bytes < dsn_bn_workspace_bytes(y->c)) DSN_FAIL(DSN_EWORKSPACE, "bn_stats: workspace too small");
    const int64_t P = npix(y);
    float* partial = (float*)workspace;
    hipStream_t st = (hipStream_t)stream;
    int nb = 1;
    {
        ProfScope prof(KID_BN_STATS, 0.0, (double)P * y->c * (y->dtype == DSN_F32 ? 4.0 : 2.0), st);
        DSN_DISPATCH_DTYPE(y->dtype, T, (launch_reduce<T, false, StatsF>(vec_ok(y), y, nullptr, partial, &nb, st)));
    }
    DSN_LAUNCH_CHECK("bn_stats reduce");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(y->c, 4)), dim3(256), 0, st, partial, nb, y->c, (double)P, gamma, beta,
                       running_mean, running_var, momentum, eps, scale, shift, mean, rstd);
    DSN_LAUNCH_CHECK("bn_stats finalize");
    return DSN_OK;
}

// Finalize from partial rows produced elsewhere (the convolution epilogue: dsn_conv2d_fwd_stats).
extern "C" int dsn_bn_finalize(const float* partial, int32_t rows, int32_t c, int64_t count, const float* gamma,
                               const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                               float* scale, float* shift, float* mean, float* rstd, void* stream) {
    DSN_CHECK_ARG(partial && rows > 0 && c > 0 && count > 0 && scale && shift && mean && rstd, "bn_finalize: bad args");
    DSN_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "bn_finalize: running stats must come in pairs");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(c, 4)), dim3(256), 0, (hipStream_t)stream, partial, rows, c,
                       (double)count, gamma, beta, running_mean, running_var, momentum, eps, scale, shift, mean, rstd);
    DSN_LAUNCH_CHECK("bn_finalize");
    return DSN_OK;
}

extern "C" int dsn_bn_act_fwd(const dsn_tensor* y, const float* scale, const float* shift, int32_t act,
                              const dsn_tensor* residual, const dsn_tensor* z, void* stream) {
    DSN_CHECK_ARG(tensor_ok(y) && tensor_ok(z) && same_shape(y, z), "bn_act_fwd: invalid tensors");
    DSN_CHECK_ARG((scale == nullptr) == (shift == nullptr), "bn_act_fwd: scale/shift must come in pairs");
    if (residual) DSN_CHECK_ARG(tensor_ok(residual) && same_shape(y, residual), "bn_act_fwd: residual mismatch");
    const int64_t P = npix(y);
    const bool v = vec_ok(y) && vec_ok(z) && (!residual || vec_ok(residual));
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(KID_BN_ACT_FWD, 0.0, (double)P * y->c * (y->dtype == DSN_F32 ? 4.0 : 2.0) * (residual ? 3 : 2), st);
    BnParams bp{scale, shift, nullptr, nullptr, nullptr, act, y->c};
    DSN_DISPATCH_DTYPE(y->dtype, T, {
        if (residual)
            launch_ew<T, true, FwdF>(v, y, residual, z, st, bp, true);
        else
            launch_ew<T, false, FwdF>(v, y, nullptr, z, st, bp, false);
    });
    DSN_LAUNCH_CHECK("bn_act_fwd");
    return DSN_OK;
}

extern "C" int dsn_bn_act_bwd(const dsn_tensor* dz, const dsn_tensor* y, const float* scale, const float* shift,
                              const float* mean, const float* rstd, int32_t act, const dsn_tensor* dy, float* dgamma,
                              float* dbeta, int32_t accumulate, void* workspace, int64_t workspace_bytes,
                              void* stream) {
    DSN_CHECK_ARG(tensor_ok(dz) && tensor_ok(y) && tensor_ok(dy) && same_shape(dz, y) && same_shape(dy, y),
                  "bn_act_bwd: invalid tensors");
    DSN_CHECK_ARG(scale && shift && mean && rstd && workspace, "bn_act_bwd: null argument");
    if (workspace_bytes < dsn_bn_workspace_bytes(y->c)) DSN_FAIL(DSN_EWORKSPACE, "bn_act_bwd: workspace too small");
    const int64_t P = npix(y);
    float* partial = (float*)workspace;
    float* means = partial + (int64_t)MAX_RED_BLOCKS * 2 * y->c;
    hipStream_t st = (hipStream_t)stream;
    const bool v = vec_ok(y) && vec_ok(dz) && vec_ok(dy);
    const double esz = y->dtype == DSN_F32 ? 4.0 : 2.0;
    BnParams bp{scale, shift, mean, rstd, means, act, y->c};
    int nb = 1;
    {
        ProfScope prof(KID_BN_BWD_REDUCE, 0.0, 2.0 * P * y->c * esz, st);
        DSN_DISPATCH_DTYPE(y->dtype, T, (launch_reduce<T, true, BwdRedF>(v, y, dz, partial, &nb, st, bp)));
    }
    DSN_LAUNCH_CHECK("bn_act_bwd reduce");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(y->c, 4)), dim3(256), 0, st, partial, nb, y->c, (double)P, dgamma,
                       dbeta, accumulate, means);
    DSN_LAUNCH_CHECK("bn_act_bwd finalize");
    {
        ProfScope prof(KID_BN_BWD_APPLY, 0.0, 3.0 * P * y->c * esz, st);
        DSN_DISPATCH_DTYPE(y->dtype, T, (launch_ew<T, true, BwdApplyF>(v, y, dz, dy, st, bp)));
    }
    DSN_LAUNCH_CHECK("bn_act_bwd apply");
    return DSN_OK;
}

extern "C" int dsn_act_bwd(const dsn_tensor* dz, const dsn_tensor* y, int32_t act, const dsn_tensor* dy, void* stream) {
    DSN_CHECK_ARG(tensor_ok(dz) && tensor_ok(y) && tensor_ok(dy) && same_shape(dz, y) && same_shape(dy, y),
                  "act_bwd: invalid tensors");
    const bool v = vec_ok(y) && vec_ok(dz) && vec_ok(dy);
    BnParams bp{nullptr, nullptr, nullptr, nullptr, nullptr, act, y->c};
    DSN_DISPATCH_DTYPE(y->dtype, T, (launch_ew<T, true, BwdApplyF>(v, y, dz, dy, (hipStream_t)stream, bp)));
    DSN_LAUNCH_CHECK("act_bwd");
    return DSN_OK;
}
