// BatchNorm (training statistics, apply, backward) + activation kernels over NHWC [P][C] views.  HBM-bound.
//
// Per-channel reductions: a 256-thread block owns a strip of pixels; thread (ty, tx) accumulates channel vector tx over
// pixels ty, ty+TY, ... (consecutive lanes read consecutive channels of one pixel -> coalesced 16-byte loads), the
// block folds its TY partials through LDS and writes ONE row of per-block partials; a finalize kernel reduces the rows
// in fp64 in a fixed order (deterministic: no float atomics).
#include "common.h"

namespace {

constexpr int RED_THREADS = 256;
constexpr int MAX_RED_BLOCKS = 1024;

template <typename T, int V> struct VecIO {
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
template <> struct VecIO<bf16_t, 4> {
    __device__ static __forceinline__ void load(const bf16_t* p, float (&o)[4]) {
        const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (float)v[i];
    }
    __device__ static __forceinline__ void store(bf16_t* p, const float (&o)[4]) {
        bf16x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (bf16_t)o[i];
        *reinterpret_cast<bf16x4*>(p) = v;
    }
};

// ---- generic two-quantity per-channel reduction -------------------------------------------------------------
// F::eval(c, y, dz) -> (q0, q1) accumulated per channel.
struct StatsF {   // sum y, sum y^2
    __device__ __forceinline__ void operator()(int, float y, float, float& q0, float& q1) const { q0 += y; q1 += y * y; }
};
struct BwdF {     // sum g, sum g*yhat with g = dz * act'(u)
    const float *scale, *shift, *mean, *rstd;
    int act;
    __device__ __forceinline__ void operator()(int c, float y, float dz, float& q0, float& q1) const {
        const float u = y * scale[c] + shift[c];
        const float g = dz * act_grad(u, act);
        q0 += g;
        q1 += g * ((y - mean[c]) * rstd[c]);
    }
};

template <typename T, int V, bool HAS_DZ, typename F>
__global__ __launch_bounds__(RED_THREADS) void reduce2_kernel(const T* __restrict__ y, int64_t yld,
                                                              const T* __restrict__ dz, int64_t zld, int64_t P, int C,
                                                              float* __restrict__ partial, F f) {
    __shared__ float red[2 * RED_THREADS * 4];
    const int ncv = C / V;
    const int TX = ncv < RED_THREADS ? ncv : RED_THREADS;
    const int TY = RED_THREADS / TX;
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int64_t per = (P + gridDim.x - 1) / gridDim.x;
    const int64_t p0 = blockIdx.x * per, p1 = (p0 + per < P) ? p0 + per : P;
    float* out = partial + (int64_t)blockIdx.x * 2 * C;
    for (int cv0 = 0; cv0 < ncv; cv0 += TX) {
        const int cv = cv0 + tx;
        float q0[V], q1[V];
#pragma unroll
        for (int i = 0; i < V; ++i) q0[i] = q1[i] = 0.f;
        if (ty < TY && cv < ncv) {
            for (int64_t p = p0 + ty; p < p1; p += TY) {
                float a[V], b[V];
                VecIO<T, V>::load(y + p * yld + cv * V, a);
                if (HAS_DZ) VecIO<T, V>::load(dz + p * zld + cv * V, b);
#pragma unroll
                for (int i = 0; i < V; ++i) f(cv * V + i, a[i], HAS_DZ ? b[i] : 0.f, q0[i], q1[i]);
            }
        }
        // fold the TY partials of each channel
        __syncthreads();
        if (ty < TY) {
#pragma unroll
            for (int i = 0; i < V; ++i) {
                red[(ty * TX + tx) * V + i] = q0[i];
                red[RED_THREADS * 4 + (ty * TX + tx) * V + i] = q1[i];
            }
        }
        __syncthreads();
        const int cbase = cv0 * V, cnum = ((ncv - cv0 < TX) ? (ncv - cv0) : TX) * V;
        for (int c = threadIdx.x; c < cnum; c += RED_THREADS) {
            const int ctx = c / V, ci = c % V;
            float s0 = 0.f, s1 = 0.f;
            for (int t = 0; t < TY; ++t) {
                s0 += red[(t * TX + ctx) * V + ci];
                s1 += red[RED_THREADS * 4 + (t * TX + ctx) * V + ci];
            }
            out[cbase + c] = s0;
            out[C + cbase + c] = s1;
        }
    }
}

// One wave per channel: the 64 lanes stride over the per-block partial rows and fold in fp64 with shuffles (fixed order ->
// deterministic).  256-thread blocks = 4 channels per block.
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

// sums -> dgamma/dbeta (+=) and the two means the apply pass needs (written after the partial rows)
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

template <typename T, int V>
__global__ void bn_act_fwd_kernel(const T* __restrict__ y, int64_t yld, const float* __restrict__ scale,
                                  const float* __restrict__ shift, int act, const T* __restrict__ res, int64_t rld,
                                  T* __restrict__ z, int64_t zld, int64_t P, int C) {
    const int ncv = C / V;
    const int64_t total = P * ncv;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = i / ncv;
        const int c0 = (int)(i - p * ncv) * V;
        float a[V], r[V];
        VecIO<T, V>::load(y + p * yld + c0, a);
        if (res) VecIO<T, V>::load(res + p * rld + c0, r);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float u = a[k];
            if (scale) u = u * scale[c0 + k] + shift[c0 + k];
            u = apply_act(u, act);
            a[k] = res ? u + r[k] : u;
        }
        VecIO<T, V>::store(z + p * zld + c0, a);
    }
}

// dy = scale * (g - mean_g - yhat * mean_gyhat),  g = dz * act'(y*scale+shift)          (BN present)
// dy = dz * act'(y)                                                                     (scale == nullptr)
template <typename T, int V>
__global__ void bn_act_bwd_apply_kernel(const T* __restrict__ dz, int64_t zld, const T* __restrict__ y, int64_t yld,
                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                        const float* __restrict__ means, int act, T* __restrict__ dy, int64_t dld,
                                        int64_t P, int C) {
    const int ncv = C / V;
    const int64_t total = P * ncv;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = i / ncv;
        const int c0 = (int)(i - p * ncv) * V;
        float a[V], g[V];
        VecIO<T, V>::load(y + p * yld + c0, a);
        VecIO<T, V>::load(dz + p * zld + c0, g);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const int c = c0 + k;
            if (scale) {
                const float u = a[k] * scale[c] + shift[c];
                const float gg = g[k] * act_grad(u, act);
                const float yh = (a[k] - mean[c]) * rstd[c];
                g[k] = scale[c] * (gg - means[c] - yh * means[C + c]);
            } else {
                g[k] = g[k] * act_grad(a[k], act);
            }
        }
        VecIO<T, V>::store(dy + p * dld + c0, g);
    }
}

inline int ew_grid(int64_t total) {
    int64_t b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}
inline int red_blocks(int64_t P) {
    int64_t b = (P + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 512 ? 512 : b));
}
inline bool vec4_ok(const dsn_tensor* t) {
    const int es = t->dtype == DSN_F32 ? 4 : 2;
    return t->c % 4 == 0 && t->ldc % 4 == 0 && ((uintptr_t)t->ptr % (4 * es)) == 0;
}
inline bool same_shape(const dsn_tensor* a, const dsn_tensor* b) {
    return a->n == b->n && a->h == b->h && a->w == b->w && a->c == b->c && a->dtype == b->dtype;
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
    const int nb = red_blocks(P);
    float* partial = (float*)workspace;
    hipStream_t st = (hipStream_t)stream;
    const bool v4 = vec4_ok(y);
    DSN_CHECK_ARG(v4 || y->c <= RED_THREADS, "bn_stats: C=%d needs C%%4==0 or C<=256", y->c);
    const double esz = y->dtype == DSN_F32 ? 4.0 : 2.0;
    {
    ProfScope prof(KID_BN_STATS, 0.0, (double)P * y->c * esz, st);
    DSN_DISPATCH_DTYPE(y->dtype, T, {
        if (v4)
            hipLaunchKernelGGL((reduce2_kernel<T, 4, false, StatsF>), dim3(nb), dim3(RED_THREADS), 0, st,
                               (const T*)y->ptr, y->ldc, (const T*)nullptr, (int64_t)0, P, y->c, partial, StatsF{});
        else
            hipLaunchKernelGGL((reduce2_kernel<T, 1, false, StatsF>), dim3(nb), dim3(RED_THREADS), 0, st,
                               (const T*)y->ptr, y->ldc, (const T*)nullptr, (int64_t)0, P, y->c, partial, StatsF{});
    });
    }
    DSN_LAUNCH_CHECK("bn_stats reduce");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(y->c, 4)), dim3(256), 0, st, partial, nb, y->c, (double)P, gamma,
                       beta, running_mean, running_var, momentum, eps, scale, shift, mean, rstd);
    DSN_LAUNCH_CHECK("bn_stats finalize");
    return DSN_OK;
}

extern "C" int dsn_bn_act_fwd(const dsn_tensor* y, const float* scale, const float* shift, int32_t act,
                              const dsn_tensor* residual, const dsn_tensor* z, void* stream) {
    DSN_CHECK_ARG(tensor_ok(y) && tensor_ok(z) && same_shape(y, z), "bn_act_fwd: invalid tensors");
    DSN_CHECK_ARG((scale == nullptr) == (shift == nullptr), "bn_act_fwd: scale/shift must come in pairs");
    if (residual) DSN_CHECK_ARG(tensor_ok(residual) && same_shape(y, residual), "bn_act_fwd: residual mismatch");
    const int64_t P = npix(y);
    const bool v4 = vec4_ok(y) && vec4_ok(z) && (!residual || vec4_ok(residual));
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(KID_BN_ACT_FWD, 0.0, (double)P * y->c * (y->dtype == DSN_F32 ? 4.0 : 2.0) * (residual ? 3 : 2), st);
    DSN_DISPATCH_DTYPE(y->dtype, T, {
        const T* r = residual ? (const T*)residual->ptr : nullptr;
        const int64_t rld = residual ? residual->ldc : 0;
        if (v4)
            hipLaunchKernelGGL((bn_act_fwd_kernel<T, 4>), dim3(ew_grid(P * (y->c / 4))), dim3(256), 0, st,
                               (const T*)y->ptr, y->ldc, scale, shift, act, r, rld, (T*)z->ptr, z->ldc, P, y->c);
        else
            hipLaunchKernelGGL((bn_act_fwd_kernel<T, 1>), dim3(ew_grid(P * y->c)), dim3(256), 0, st, (const T*)y->ptr,
                               y->ldc, scale, shift, act, r, rld, (T*)z->ptr, z->ldc, P, y->c);
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
    const int nb = red_blocks(P);
    float* partial = (float*)workspace;
    float* means = partial + (int64_t)MAX_RED_BLOCKS * 2 * y->c;
    hipStream_t st = (hipStream_t)stream;
    const bool v4 = vec4_ok(y) && vec4_ok(dz) && vec4_ok(dy);
    DSN_CHECK_ARG(v4 || y->c <= RED_THREADS, "bn_act_bwd: C=%d needs C%%4==0 or C<=256", y->c);
    BwdF f{scale, shift, mean, rstd, act};
    const double esz = y->dtype == DSN_F32 ? 4.0 : 2.0;
    {
    ProfScope prof(KID_BN_BWD_REDUCE, 0.0, 2.0 * P * y->c * esz, st);
    DSN_DISPATCH_DTYPE(y->dtype, T, {
        if (v4)
            hipLaunchKernelGGL((reduce2_kernel<T, 4, true, BwdF>), dim3(nb), dim3(RED_THREADS), 0, st, (const T*)y->ptr,
                               y->ldc, (const T*)dz->ptr, dz->ldc, P, y->c, partial, f);
        else
            hipLaunchKernelGGL((reduce2_kernel<T, 1, true, BwdF>), dim3(nb), dim3(RED_THREADS), 0, st, (const T*)y->ptr,
                               y->ldc, (const T*)dz->ptr, dz->ldc, P, y->c, partial, f);
    });
    }
    DSN_LAUNCH_CHECK("bn_act_bwd reduce");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(y->c, 4)), dim3(256), 0, st, partial, nb, y->c, (double)P,
                       dgamma, dbeta, accumulate, means);
    DSN_LAUNCH_CHECK("bn_act_bwd finalize");
    ProfScope prof2(KID_BN_BWD_APPLY, 0.0, 3.0 * P * y->c * esz, st);
    DSN_DISPATCH_DTYPE(y->dtype, T, {
        if (v4)
            hipLaunchKernelGGL((bn_act_bwd_apply_kernel<T, 4>), dim3(ew_grid(P * (y->c / 4))), dim3(256), 0, st,
                               (const T*)dz->ptr, dz->ldc, (const T*)y->ptr, y->ldc, scale, shift, mean, rstd, means,
                               act, (T*)dy->ptr, dy->ldc, P, y->c);
        else
            hipLaunchKernelGGL((bn_act_bwd_apply_kernel<T, 1>), dim3(ew_grid(P * y->c)), dim3(256), 0, st,
                               (const T*)dz->ptr, dz->ldc, (const T*)y->ptr, y->ldc, scale, shift, mean, rstd, means,
                               act, (T*)dy->ptr, dy->ldc, P, y->c);
    });
    DSN_LAUNCH_CHECK("bn_act_bwd apply");
    return DSN_OK;
}

extern "C" int dsn_act_bwd(const dsn_tensor* dz, const dsn_tensor* y, int32_t act, const dsn_tensor* dy, void* stream) {
    DSN_CHECK_ARG(tensor_ok(dz) && tensor_ok(y) && tensor_ok(dy) && same_shape(dz, y) && same_shape(dy, y),
                  "act_bwd: invalid tensors");
    const int64_t P = npix(y);
    const bool v4 = vec4_ok(y) && vec4_ok(dz) && vec4_ok(dy);
    hipStream_t st = (hipStream_t)stream;
    DSN_DISPATCH_DTYPE(y->dtype, T, {
        if (v4)
            hipLaunchKernelGGL((bn_act_bwd_apply_kernel<T, 4>), dim3(ew_grid(P * (y->c / 4))), dim3(256), 0, st,
                               (const T*)dz->ptr, dz->ldc, (const T*)y->ptr, y->ldc, (const float*)nullptr,
                               (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                               (const float*)nullptr, act, (T*)dy->ptr, dy->ldc, P, y->c);
        else
            hipLaunchKernelGGL((bn_act_bwd_apply_kernel<T, 1>), dim3(ew_grid(P * y->c)), dim3(256), 0, st,
                               (const T*)dz->ptr, dz->ldc, (const T*)y->ptr, y->ldc, (const float*)nullptr,
                               (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                               (const float*)nullptr, act, (T*)dy->ptr, dy->ldc, P, y->c);
    });
    DSN_LAUNCH_CHECK("act_bwd");
    return DSN_OK;
}
