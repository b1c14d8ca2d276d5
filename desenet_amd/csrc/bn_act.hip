// BatchNorm (training statistics, apply, backward) + activation kernels over NHWC [P][C] views.  All HBM-bound streams.
//
// Layout of work: a 256-thread block owns a strip of pixels; thread (ty, tx) owns ONE 16-byte channel vector (4 fp32 /
// 8 bf16 channels) -- so its per-channel constants (scale, shift, mean, rstd, ...) are loaded once into registers -- and
// walks the strip's pixels ty, ty+TY, ... with FOUR independent 16-byte loads in flight per operand (memory-level
// parallelism: these tensors are 1-13 MB, the kernels live or die by latency).  Consecutive lanes read consecutive channel
// vectors of one pixel, then the next pixel: fully coalesced whenever ldc == C, 16-byte segments otherwise (concat slices).
//
// Per-channel reductions fold the block's TY partials through LDS and add them into fp64 accumulators (common.h: BnAcc);
// the elementwise kernel that consumes the statistics folds the accumulators in its prologue (EwPro) -- no finalize launch.
// Ordinary (not non-temporal) output stores in this file's kernels: measured with two libraries alternated in one call, DeSeNet-m at
// 1280^2 19.38-19.41 vs 19.57-19.58 ms per step, config 3 3.942 vs 3.940 ms (common.h: VecIO::store)
#define DSN_VECIO_PLAIN 1
// ... and non-temporal LOADS: every input of these kernels is read exactly once (five alternated pairs on config 3: 3.930 vs 3.940 ms;
// DeSeNet-m at 1280^2 19.24-19.28 vs 19.27-19.31 ms)
#define DSN_VECIO_NTLOAD 1
#include "common.h"

namespace {

constexpr int THREADS = 256;
// vectors per thread and batch.  4 cost the backward apply 136 VGPRs = 3 blocks per CU = 768 resident blocks, fewer than the
// 800-1023 a batch-8 layer launches (two rounds of a latency-bound pass); 2 -> 96 VGPRs, 5 blocks per CU: config 3 +3 %.
// (Issuing the first batch above the prologue -- to overlap the accumulator fold with the operands' latency -- measured 2.6 %
// SLOWER at equal occupancy, and 1 vector per batch 1.2 % slower than 2.)
constexpr int UNROLL = 2;
constexpr int MAX_RED_BLOCKS = 512;   // (round 3: 256 / 512 / 1024 / 2048 -> 1985 / 1990 / 1984 / 1980 img/s on config 3)

// V consecutive per-channel fp32 constants (c0 is a multiple of V, arrays are 32-byte aligned): 16-byte loads instead of V
// scalar ones -- a thread used to issue up to 48 scalar loads of constants before touching its 2-8 data vectors.
template <int V> __device__ __forceinline__ void ldvec(const float* __restrict__ p, float (&o)[V]) {
    if constexpr (V % 4 == 0) {
#pragma unroll
        for (int i = 0; i < V / 4; ++i) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(p + 4 * i);
            o[4 * i] = v[0]; o[4 * i + 1] = v[1]; o[4 * i + 2] = v[2]; o[4 * i + 3] = v[3];
        }
    } else {
#pragma unroll
        for (int i = 0; i < V; ++i) o[i] = p[i];
    }
}

// ---- functors ----------------------------------------------------------------------------------------------------------
// Reductions: accumulate two per-channel quantities from (a = in0, b = in1).
template <int V> struct StatsF {            // sum y, sum y^2
    __device__ __forceinline__ void prepare(int, const float*) {}
    __device__ __forceinline__ void acc(const float (&a)[V], const float (&)[V], float (&q0)[V], float (&q1)[V]) const {
#pragma unroll
        for (int k = 0; k < V; ++k) { q0[k] += a[k]; q1[k] += a[k] * a[k]; }
    }
};
struct BnParams { const float *scale, *shift, *mean, *rstd, *means; int act, C; };
template <int V> struct BwdRedF {           // sum g, sum g*yhat with g = dz * act'(y*scale+shift)
    BnParams p;
    float sc[V], sh[V], mu[V], rs[V];
    __device__ __forceinline__ void prepare(int c0, const float*) {
        ldvec<V>(p.scale + c0, sc); ldvec<V>(p.shift + c0, sh); ldvec<V>(p.mean + c0, mu); ldvec<V>(p.rstd + c0, rs);
    }
    __device__ __forceinline__ void acc(const float (&y)[V], const float (&dz)[V], float (&q0)[V], float (&q1)[V]) const {
        float u[V], gr[V];
#pragma unroll
        for (int k = 0; k < V; ++k) u[k] = y[k] * sc[k] + sh[k];
        act_grad_vec<V>(u, p.act, gr);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float g = dz[k] * gr[k];
            q0[k] += g;
            q1[k] += g * ((y[k] - mu[k]) * rs[k]);
        }
    }
};
// Elementwise: o = f(a, b)
template <int V> struct FwdF {              // act(y*scale + shift) [+ res]
    BnParams p; bool has_res;
    float sc[V], sh[V];
    __device__ __forceinline__ void prepare(int c0, const float* pro) {     // pro: (scale, shift) from the kernel prologue
        if (pro) {
            ldvec<V>(pro + c0, sc); ldvec<V>(pro + p.C + c0, sh);
        } else if (p.scale) {
            ldvec<V>(p.scale + c0, sc); ldvec<V>(p.shift + c0, sh);
        } else {
#pragma unroll
            for (int k = 0; k < V; ++k) { sc[k] = 1.f; sh[k] = 0.f; }
        }
    }
    __device__ __forceinline__ void apply(const float (&y)[V], const float (&r)[V], float (&o)[V]) const {
        float u[V];
#pragma unroll
        for (int k = 0; k < V; ++k) u[k] = y[k] * sc[k] + sh[k];
        apply_act_vec<V>(u, p.act);
#pragma unroll
        for (int k = 0; k < V; ++k) o[k] = has_res ? u[k] + r[k] : u[k];
    }
};
template <int V> struct BwdApplyF {         // BN: scale*(g - mean_g - yhat*mean_gyhat);  no BN: dz*act'(y)
    BnParams p;
    float sc[V], sh[V], mu[V], ka[V], kb[V];     // dy = sc*g + ka*(y - mu) + kb,  ka = -sc*rstd*mean_gyhat, kb = -sc*mean_g
    __device__ __forceinline__ void prepare(int c0, const float* pro) {     // pro: [sc, sh, mu, ka, kb] from the prologue
        if (!p.scale) return;
        if (pro) {
            ldvec<V>(pro + c0, sc); ldvec<V>(pro + p.C + c0, sh); ldvec<V>(pro + 2 * p.C + c0, mu);
            ldvec<V>(pro + 3 * p.C + c0, ka); ldvec<V>(pro + 4 * p.C + c0, kb);
        } else {
            float rs[V], m1[V], m2[V];
            ldvec<V>(p.scale + c0, sc); ldvec<V>(p.shift + c0, sh); ldvec<V>(p.mean + c0, mu); ldvec<V>(p.rstd + c0, rs);
            ldvec<V>(p.means + c0, m1); ldvec<V>(p.means + p.C + c0, m2);
#pragma unroll
            for (int k = 0; k < V; ++k) { ka[k] = -sc[k] * rs[k] * m2[k]; kb[k] = -sc[k] * m1[k]; }
        }
    }
    __device__ __forceinline__ void apply(const float (&y)[V], const float (&dz)[V], float (&o)[V]) const {
        float u[V], gr[V];
        if (p.scale) {
#pragma unroll
            for (int k = 0; k < V; ++k) u[k] = y[k] * sc[k] + sh[k];
            act_grad_vec<V>(u, p.act, gr);
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const float g = dz[k] * gr[k];
                o[k] = sc[k] * g + (ka[k] * (y[k] - mu[k]) + kb[k]);
            }
        } else {
            act_grad_vec<V>(y, p.act, gr);
#pragma unroll
            for (int k = 0; k < V; ++k) o[k] = dz[k] * gr[k];
        }
    }
};

struct Strip { int64_t P; int C; int64_t per_block; };

// ---- reduction kernel ---------------------------------------------------------------------------------------------------
template <typename T, int V, bool HAS_B, typename F>
__global__ __launch_bounds__(THREADS) void reduce2_kernel(const T* __restrict__ a, int64_t ald, const T* __restrict__ b,
                                                          int64_t bld, Strip s, const BnAcc fin, F f) {
    __shared__ float red[2 * THREADS * V];
    const int ncv = s.C / V;
    const int TX = ncv < THREADS ? ncv : THREADS;
    const int TY = THREADS / TX;
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int64_t p0 = blockIdx.x * s.per_block;
    const int64_t p1 = (p0 + s.per_block < s.P) ? p0 + s.per_block : s.P;
    for (int cv0 = 0; cv0 < ncv; cv0 += TX) {
        const int cv = cv0 + tx;
        float q0[V], q1[V];
#pragma unroll
        for (int k = 0; k < V; ++k) q0[k] = q1[k] = 0.f;
        if (ty < TY && cv < ncv) {
            f.prepare(cv * V, nullptr);
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
            bn_acc_add(fin, blockIdx.x, cbase + c, s0, s1);
        }
    }
}

// ---- elementwise kernel -------------------------------------------------------------------------------------------------
// Optional prologue: fold the fp64 accumulators the PREVIOUS kernel filled into the two per-channel constants this kernel
// needs -- (scale, shift) of a training-mode BatchNorm forward, or (mean g, mean g*yhat) of its backward -- into LDS.  Every
// block does it redundantly (2 channels per thread, 16 L2 hits each); block 0 also writes what later passes need.
constexpr int PRO_MAXC = 1024;
struct EwPro {
    BnAcc a;                  // a.acc == nullptr: no prologue
    int32_t mode;             // 0 forward statistics, 1 backward sums
    const float *gamma, *beta;
    float *running_mean, *running_var;
    float momentum, eps;
    float *scale, *shift, *mean, *rstd;     // forward: written (saved for the backward pass); backward: read
    float *dgamma, *dbeta;                  // backward: parameter gradients
    int32_t accumulate;
    float pgrad_scale;                      // backward: factor on the dgamma / dbeta contribution (SyncBN: 1 / world size)
    // two BatchNorm modules over one merged tensor (C3's cv2 | cv1 as ONE convolution): channels >= split belong to the second
    // module, whose parameter vectors are indexed from 0.  split == 0: one module.
    int32_t split;
    const float *gamma2, *beta2;
    float *running_mean2, *running_var2, *dgamma2, *dbeta2;
};
__device__ __forceinline__ void ew_prologue(const EwPro& q, float* lds) {
    const int C = q.a.C;
    const bool writer = blockIdx.x == 0;
    for (int c = threadIdx.x; c < C; c += THREADS) {
        double s, ss;
        bn_acc_fold(q.a, c, s, ss);
        const bool second = q.split > 0 && c >= q.split;
        const int pc = second ? c - q.split : c;                       // index into the owning module's parameter vectors
        if (q.mode == 0) {
            const double mean = s / q.a.count;
            double var = ss / q.a.count - mean * mean;
            if (var < 0.0) var = 0.0;
            const float rstd = (float)(1.0 / sqrt(var + (double)q.eps));
            const float* gam = second ? q.gamma2 : q.gamma;
            const float* bet = second ? q.beta2 : q.beta;
            const float g = gam ? gam[pc] : 1.f, b = bet ? bet[pc] : 0.f;
            const float sc = g * rstd, sh = b - (float)mean * sc;
            lds[c] = sc;
            lds[C + c] = sh;
            if (writer) {
                q.scale[c] = sc; q.shift[c] = sh; q.mean[c] = (float)mean; q.rstd[c] = rstd;
                float* rm = second ? q.running_mean2 : q.running_mean;
                float* rv = second ? q.running_var2 : q.running_var;
                if (rm) {
                    const double unbiased = q.a.count > 1.0 ? var * q.a.count / (q.a.count - 1.0) : var;
                    rm[pc] = (1.f - q.momentum) * rm[pc] + q.momentum * (float)mean;
                    rv[pc] = (1.f - q.momentum) * rv[pc] + q.momentum * (float)unbiased;
                }
            }
        } else {
            const float sc = q.scale[c], m1 = (float)(s / q.a.count), m2 = (float)(ss / q.a.count);
            lds[c] = sc;
            lds[C + c] = q.shift[c];
            lds[2 * C + c] = q.mean[c];
            lds[3 * C + c] = -sc * q.rstd[c] * m2;
            lds[4 * C + c] = -sc * m1;
            if (writer) {
                const float gs = (float)s * q.pgrad_scale, gss = (float)ss * q.pgrad_scale;
                float* db = second ? q.dbeta2 : q.dbeta;
                float* dg = second ? q.dgamma2 : q.dgamma;
                if (db) db[pc] = q.accumulate ? db[pc] + gs : gs;
                if (dg) dg[pc] = q.accumulate ? dg[pc] + gss : gss;
            }
        }
    }
    __syncthreads();
}

template <typename T, int V, bool HAS_B, bool PRO, typename F>
__global__ __launch_bounds__(THREADS) void ew2_kernel(const T* __restrict__ a, int64_t ald, const T* __restrict__ b,
                                                      int64_t bld, T* __restrict__ o, int64_t old_, Strip s, const EwPro pro, F f) {
    __shared__ __attribute__((aligned(16))) float s_pro[PRO ? 5 * PRO_MAXC : 4];
    if (PRO) ew_prologue(pro, s_pro);
    const float* lds = PRO ? s_pro : nullptr;
    const int ncv = s.C / V;
    const int TX = ncv < THREADS ? ncv : THREADS;
    const int TY = THREADS / TX;
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    if (ty >= TY) return;
    const int64_t p0 = blockIdx.x * s.per_block;
    const int64_t p1 = (p0 + s.per_block < s.P) ? p0 + s.per_block : s.P;
    for (int cv = tx; cv < ncv; cv += TX) {
        f.prepare(cv * V, lds);
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

// ---- deferred BatchNorm: materialise z = act(y*scale + shift) [+ residual] with constants folded from the accumulators -----
// The same elementwise pass as ew2_kernel<FwdF>, but driven by dsn_lazy_in descriptors (common.h): the input -- and the optional
// residual, a Bottleneck shortcut that is itself a deferred tensor -- may be concats of raw pre-BN segments and ordinary ones.
// No side effects: saved statistics and running averages are written once per step by bn_finalize_multi_kernel.
template <typename T, int V, bool HAS_R>
__global__ __launch_bounds__(THREADS) void lazy_ew_kernel(const T* __restrict__ a, int64_t ald, const T* __restrict__ r, int64_t rld,
                                                          T* __restrict__ o, int64_t old_, Strip s, const LazyIn la, const LazyIn lr) {
    __shared__ __attribute__((aligned(16))) float sc[HAS_R ? 2 : 1][PRO_MAXC], sh[HAS_R ? 2 : 1][PRO_MAXC];
    __shared__ unsigned char act8[HAS_R ? 2 : 1][PRO_MAXC / 8];
    lazy_table(la, s.C, sc[0], sh[0], act8[0], THREADS);
    if (HAS_R) lazy_table(lr, s.C, sc[HAS_R ? 1 : 0], sh[HAS_R ? 1 : 0], act8[HAS_R ? 1 : 0], THREADS);
    __syncthreads();
    const int ncv = s.C / V;
    const int TX = ncv < THREADS ? ncv : THREADS;
    const int TY = THREADS / TX;
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    if (ty >= TY) return;
    const int64_t p0 = blockIdx.x * s.per_block;
    const int64_t p1 = (p0 + s.per_block < s.P) ? p0 + s.per_block : s.P;
    for (int cv = tx; cv < ncv; cv += TX) {
        float asc[V], ash[V], rsc[V], rsh[V];
        ldvec<V>(sc[0] + cv * V, asc); ldvec<V>(sh[0] + cv * V, ash);
        const int aact = act8[0][(cv * V) >> 3];
        int ract = 0;
        if (HAS_R) { ldvec<V>(sc[HAS_R ? 1 : 0] + cv * V, rsc); ldvec<V>(sh[HAS_R ? 1 : 0] + cv * V, rsh); ract = act8[HAS_R ? 1 : 0][(cv * V) >> 3]; }
        const T* ap = a + cv * V;
        const T* rp = HAS_R ? r + cv * V : nullptr;
        T* op = o + cv * V;
        auto one = [&](const float (&va)[V], const float (&vr)[V], int64_t p) {
            float vo[V];
#pragma unroll
            for (int k = 0; k < V; ++k) vo[k] = va[k] * asc[k] + ash[k];
            apply_act_vec<V>(vo, aact);
            if (HAS_R) {
                float ur[V];
#pragma unroll
                for (int k = 0; k < V; ++k) ur[k] = vr[k] * rsc[k] + rsh[k];
                apply_act_vec<V>(ur, ract);
                // a deferred residual is what its own materialisation would have stored: rounded to T before the add
#pragma unroll
                for (int k = 0; k < V; ++k) vo[k] += to_f32<T>(from_f32<T>(ur[k]));
            }
            VecIO<T, V>::store(op + p * old_, vo);
        };
        int64_t p = p0 + ty;
        for (; p + (UNROLL - 1) * TY < p1; p += UNROLL * TY) {
            float va[UNROLL][V], vr[UNROLL][V];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                VecIO<T, V>::load(ap + (p + u * TY) * ald, va[u]);
                if (HAS_R) VecIO<T, V>::load(rp + (p + u * TY) * rld, vr[u]);
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) one(va[u], vr[u], p + u * TY);
        }
        for (; p < p1; p += TY) {
            float va[V], vr[V];
            VecIO<T, V>::load(ap + p * ald, va);
            if (HAS_R) VecIO<T, V>::load(rp + p * rld, vr);
            one(va, vr, p);
        }
    }
}

// saved statistics + running averages of up to FINAL_MAX BatchNorm modules per launch (end of the forward pass)
constexpr int FINAL_MAX = 38;      // (kernel arguments: 8 + 39 * 2 (+ padding) + 38 * 104 bytes <= 4 KB, asserted below)
struct FinalTable {
    int32_t n, pad;
    int16_t first[FINAL_MAX + 1];     // first block of each entry (256 channels per block: a few hundred blocks at most)
    dsn_bn_final e[FINAL_MAX];
};
static_assert(sizeof(FinalTable) <= 4096, "FinalTable travels by value: keep it inside the 4 KB kernel-argument budget");
__global__ __launch_bounds__(256) void bn_finalize_multi_kernel(const FinalTable t) {
    int l = 0;
    while (l + 1 < t.n && t.first[l + 1] <= (int)blockIdx.x) ++l;
    const dsn_bn_final& e = t.e[l];
    const int c = ((int)blockIdx.x - t.first[l]) * 256 + threadIdx.x;
    if (c >= e.n) return;
    const double* a0 = (const double*)e.acc;
    double s = 0.0, ss = 0.0;
#pragma unroll
    for (int r = 0; r < BN_NREP; ++r) {
        const double* a = a0 + (size_t)r * 2 * e.acc_c;
        s += a[e.ch0 + c];
        ss += a[e.acc_c + e.ch0 + c];
    }
    const double mean = s / e.count;
    double var = ss / e.count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)e.eps));
    const float g = e.gamma ? e.gamma[c] : 1.f, b = e.beta ? e.beta[c] : 0.f;
    const float sc = g * rstd;
    e.scale[c] = sc;
    e.shift[c] = b - (float)mean * sc;
    e.mean[c] = (float)mean;
    e.rstd[c] = rstd;
    if (e.running_mean) {
        const double unbiased = e.count > 1.0 ? var * e.count / (e.count - 1.0) : var;
        e.running_mean[c] = (1.f - e.momentum) * e.running_mean[c] + e.momentum * (float)mean;
        e.running_var[c] = (1.f - e.momentum) * e.running_var[c] + e.momentum * (float)unbiased;
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

// Standalone finalize of forward statistics from the accumulators (dsn_bn_stats: BN after a biased conv, channel sums);
// restores the zeros, so that entry point keeps a "zero once" workspace.
__global__ __launch_bounds__(256) void bn_acc_finalize_kernel(const BnAcc a, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float* __restrict__ running_mean,
                                                              float* __restrict__ running_var, float momentum, float eps,
                                                              float* __restrict__ scale, float* __restrict__ shift,
                                                              float* __restrict__ mean_o, float* __restrict__ rstd_o) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= a.C) return;
    double s, ss;
    bn_acc_fold(a, c, s, ss);
    for (int r = 0; r < BN_NREP; ++r) {
        a.acc[(size_t)r * 2 * a.C + c] = 0.0;
        a.acc[(size_t)r * 2 * a.C + a.C + c] = 0.0;
    }
    const double mean = s / a.count;
    double var = ss / a.count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float sc = g * rstd;
    scale[c] = sc;
    shift[c] = b - (float)mean * sc;
    mean_o[c] = (float)mean;
    rstd_o[c] = rstd;
    if (running_mean) {
        const double unbiased = a.count > 1.0 ? var * a.count / (a.count - 1.0) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

// Per-channel sums (conv bias gradients): out[c] (+)= sum over pixels; restores the zeros like bn_acc_finalize_kernel.
__global__ __launch_bounds__(256) void channel_sum_finalize_kernel(const BnAcc a, float* __restrict__ out, int accumulate) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= a.C) return;
    double s, ss;
    bn_acc_fold(a, c, s, ss);
    for (int r = 0; r < BN_NREP; ++r) {
        a.acc[(size_t)r * 2 * a.C + c] = 0.0;
        a.acc[(size_t)r * 2 * a.C + a.C + c] = 0.0;
    }
    out[c] = (accumulate ? out[c] : 0.f) + (float)s;
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
inline Strip make_strip(int64_t P, int C, int V, int max_blocks, int* nblocks, bool adaptive = true) {
    const int ncv = C / V;
    const int TX = ncv < THREADS ? ncv : THREADS, TY = THREADS / TX;
    // vectors per thread: UNROLL when the tensor can still give every SIMD a few waves that way, fewer on small maps (a
    // thread's 8*UNROLL elements are serial ALU work -- exp, rcp, converts -- that nothing hides at one wave per SIMD)
    const int64_t vectors = P * ncv;
    static const int ufix = [] { const char* e = getenv("DSN_EW_UNROLL"); return e ? atoi(e) : 0; }();   // tuning knob
    // (round 3, after the block cap below: one full batch per thread everywhere measured +0.2 % over 1 vector on the small maps;
    //  2 / 4 batches per thread on them -1 % / -3.4 %: the small tensors want many blocks, the large ones few)
    int u = UNROLL;
    (void)vectors;
    if (ufix > 0) u = ufix > 16 ? 16 : ufix;      // (values above UNROLL: several batches per thread)
    if (!adaptive) u = UNROLL;      // reductions: every block ends with an LDS fold + 2C atomics -> fewer, fatter blocks
    int64_t per = (int64_t)TY * u;
    int64_t nb = (P + per - 1) / per;
    // every block of the elementwise kernels folds the accumulators of ALL C channels in its prologue (16 loads + fp64 math per channel):
    // past two blocks per CU more blocks only repeat that.  Measured (DSN_EW_MAXB, 0 = the launch's own cap): 256 -> 1933, 384 -> 1944,
    // 512 -> 1972, 768 -> 1967, 1024 -> 1961, uncapped 1951 img/s on config 3; config 5 197.7 -> 200.9.
    static const int maxb = [] { const char* e = getenv("DSN_EW_MAXB"); return e ? atoi(e) : 512; }();
    if (adaptive && maxb > 0 && max_blocks > maxb) max_blocks = maxb;
    if (nb > max_blocks) {
        per = (P + max_blocks - 1) / max_blocks;
        per = (per + TY - 1) / TY * TY;
        nb = (P + per - 1) / per;
    }
    *nblocks = (int)(nb < 1 ? 1 : nb);
    return Strip{P, C, per};
}

template <typename T, bool HAS_B, template <int> class F, typename... A>
void launch_reduce(bool vec, const dsn_tensor* a, const dsn_tensor* b, const BnAcc& fin, hipStream_t st, A... args) {
    int nb = 1;
    int* nblocks = &nb;
    constexpr int VV = VW<T>::N;
    const int64_t P = npix(a);
    if (vec) {
        Strip s = make_strip(P, a->c, VV, MAX_RED_BLOCKS, nblocks, false);
        hipLaunchKernelGGL((reduce2_kernel<T, VV, HAS_B, F<VV>>), dim3(*nblocks), dim3(THREADS), 0, st, (const T*)a->ptr,
                           a->ldc, b ? (const T*)b->ptr : nullptr, b ? b->ldc : 0, s, fin, F<VV>{args...});
    } else {
        Strip s = make_strip(P, a->c, 1, MAX_RED_BLOCKS, nblocks, false);
        hipLaunchKernelGGL((reduce2_kernel<T, 1, HAS_B, F<1>>), dim3(*nblocks), dim3(THREADS), 0, st, (const T*)a->ptr,
                           a->ldc, b ? (const T*)b->ptr : nullptr, b ? b->ldc : 0, s, fin, F<1>{args...});
    }
}

template <typename T, bool HAS_B, template <int> class F, typename... A>
void launch_ew(bool vec, const dsn_tensor* a, const dsn_tensor* b, const dsn_tensor* o, const EwPro* pro, hipStream_t st,
               A... args) {
    constexpr int VV = VW<T>::N;
    const int64_t P = npix(a);
    const T* bp = b ? (const T*)b->ptr : nullptr;
    const int64_t bld = b ? b->ldc : 0;
    int nb;
    if (vec) {
        Strip s = make_strip(P, a->c, VV, 16384, &nb);
        if (pro)
            hipLaunchKernelGGL((ew2_kernel<T, VV, HAS_B, true, F<VV>>), dim3(nb), dim3(THREADS), 0, st, (const T*)a->ptr, a->ldc,
                               bp, bld, (T*)o->ptr, o->ldc, s, *pro, F<VV>{args...});
        else
            hipLaunchKernelGGL((ew2_kernel<T, VV, HAS_B, false, F<VV>>), dim3(nb), dim3(THREADS), 0, st, (const T*)a->ptr, a->ldc,
                               bp, bld, (T*)o->ptr, o->ldc, s, EwPro{}, F<VV>{args...});
    } else {
        Strip s = make_strip(P, a->c, 1, 16384, &nb);
        if (pro)
            hipLaunchKernelGGL((ew2_kernel<T, 1, HAS_B, true, F<1>>), dim3(nb), dim3(THREADS), 0, st, (const T*)a->ptr, a->ldc,
                               bp, bld, (T*)o->ptr, o->ldc, s, *pro, F<1>{args...});
        else
            hipLaunchKernelGGL((ew2_kernel<T, 1, HAS_B, false, F<1>>), dim3(nb), dim3(THREADS), 0, st, (const T*)a->ptr, a->ldc,
                               bp, bld, (T*)o->ptr, o->ldc, s, EwPro{}, F<1>{args...});
    }
}

}  // namespace

// fp64 accumulators [BN_NREP][2][c] (common.h).  dsn_bn_stats wants them zero-filled ONCE (it restores the zeros);
// dsn_conv2d_fwd_bnacc + dsn_bn_act_fwd_acc and dsn_bn_act_bwd want them ZERO ON ENTRY and leave them dirty.
extern "C" int64_t dsn_bn_workspace_bytes(int32_t c) { return bn_acc_bytes(c); }

extern "C" int dsn_bn_stats(const dsn_tensor* y, const float* gamma, const float* beta, float* running_mean,
                            float* running_var, float momentum, float eps, float* scale, float* shift, float* mean,
                            float* rstd, void* workspace, int64_t workspace_bytes, void* stream) {
    DSN_CHECK_ARG(tensor_ok(y) && scale && shift && mean && rstd && workspace, "bn_stats: null argument");
    DSN_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "bn_stats: running stats must come in pairs");
    if (workspace_bytes < dsn_bn_workspace_bytes(y->c)) DSN_FAIL(DSN_EWORKSPACE, "bn_stats: workspace too small");
    const int64_t P = npix(y);
    hipStream_t st = (hipStream_t)stream;
    BnAcc f{(double*)workspace, y->c, (double)P};
    {
        ProfScope prof(KID_BN_STATS, 0.0, (double)P * y->c * (y->dtype == DSN_F32 ? 4.0 : 2.0), st);
        DSN_DISPATCH_DTYPE(y->dtype, T, (launch_reduce<T, false, StatsF>(vec_ok(y), y, nullptr, f, st)));
    }
    DSN_LAUNCH_CHECK("bn_stats reduce");
    hipLaunchKernelGGL(bn_acc_finalize_kernel, dim3(cdiv(y->c, 256)), dim3(256), 0, st, f, gamma, beta, running_mean,
                       running_var, momentum, eps, scale, shift, mean, rstd);
    DSN_LAUNCH_CHECK("bn_stats finalize");
    return DSN_OK;
}

// out[c] (+)= sum_{n,h,w} t[n,h,w,c]: the bias gradient of a biased convolution (Detect heads, seg classifier), two launches
// and no host-framework kernels.  workspace as dsn_bn_stats.
extern "C" int dsn_channel_sum(const dsn_tensor* t, float* out, int32_t accumulate, void* workspace, int64_t workspace_bytes,
                               void* stream) {
    DSN_CHECK_ARG(tensor_ok(t) && out && workspace, "channel_sum: null argument");
    if (workspace_bytes < dsn_bn_workspace_bytes(t->c)) DSN_FAIL(DSN_EWORKSPACE, "channel_sum: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    BnAcc f{(double*)workspace, t->c, (double)npix(t)};
    DSN_DISPATCH_DTYPE(t->dtype, T, (launch_reduce<T, false, StatsF>(vec_ok(t), t, nullptr, f, st)));
    DSN_LAUNCH_CHECK("channel_sum reduce");
    hipLaunchKernelGGL(channel_sum_finalize_kernel, dim3(cdiv(t->c, 256)), dim3(256), 0, st, f, out, accumulate);
    DSN_LAUNCH_CHECK("channel_sum finalize");
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
            launch_ew<T, true, FwdF>(v, y, residual, z, nullptr, st, bp, true);
        else
            launch_ew<T, false, FwdF>(v, y, nullptr, z, nullptr, st, bp, false);
    });
    DSN_LAUNCH_CHECK("bn_act_fwd");
    return DSN_OK;
}

// Training-mode BatchNorm + activation (+ shortcut) straight from the accumulators dsn_conv2d_fwd_bnacc filled: the kernel's
// prologue turns the sums into scale/shift; block 0 also writes scale/shift/mean/rstd (saved for dsn_bn_act_bwd) and
// updates the running statistics.
extern "C" int dsn_bn_act_fwd_acc(const dsn_tensor* y, const void* acc, int64_t acc_bytes, double count, const float* gamma,
                                  const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                                  float* scale, float* shift, float* mean, float* rstd, int32_t act,
                                  const dsn_tensor* residual, const dsn_tensor* z, const dsn_bn_split* second,
                                  void* stream) {
    DSN_CHECK_ARG(tensor_ok(y) && tensor_ok(z) && same_shape(y, z), "bn_act_fwd_acc: invalid tensors");
    DSN_CHECK_ARG(acc && scale && shift && mean && rstd, "bn_act_fwd_acc: null argument");
    DSN_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "bn_act_fwd_acc: running stats must come in pairs");
    DSN_CHECK_ARG(y->c <= PRO_MAXC, "bn_act_fwd_acc: at most %d channels", PRO_MAXC);
    if (acc_bytes < bn_acc_bytes(y->c)) DSN_FAIL(DSN_EWORKSPACE, "bn_act_fwd_acc: accumulator buffer too small");
    if (residual) DSN_CHECK_ARG(tensor_ok(residual) && same_shape(y, residual), "bn_act_fwd_acc: residual mismatch");
    const int64_t P = npix(y);
    const bool v = vec_ok(y) && vec_ok(z) && (!residual || vec_ok(residual));
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(KID_BN_ACT_FWD, 0.0, (double)P * y->c * (y->dtype == DSN_F32 ? 4.0 : 2.0) * (residual ? 3 : 2), st);
    BnParams bp{scale, shift, nullptr, nullptr, nullptr, act, y->c};
    EwPro pro{};
    pro.a = BnAcc{(double*)acc, y->c, count > 0.0 ? count : (double)P};      // count: SyncBN passes the global pixel count
    pro.mode = 0;
    pro.gamma = gamma; pro.beta = beta; pro.running_mean = running_mean; pro.running_var = running_var;
    pro.momentum = momentum; pro.eps = eps;
    pro.scale = scale; pro.shift = shift; pro.mean = mean; pro.rstd = rstd;
    if (second && second->split_c > 0) {
        DSN_CHECK_ARG(second->split_c < y->c && (second->running_mean == nullptr) == (running_mean == nullptr),
                      "bn_act_fwd_acc: bad split");
        pro.split = second->split_c;
        pro.gamma2 = second->gamma; pro.beta2 = second->beta;
        pro.running_mean2 = second->running_mean; pro.running_var2 = second->running_var;
    }
    DSN_DISPATCH_DTYPE(y->dtype, T, {
        if (residual)
            launch_ew<T, true, FwdF>(v, y, residual, z, &pro, st, bp, true);
        else
            launch_ew<T, false, FwdF>(v, y, nullptr, z, &pro, st, bp, false);
    });
    DSN_LAUNCH_CHECK("bn_act_fwd_acc");
    return DSN_OK;
}

// Backward of BN + act in its two halves (dsn_bn_act_bwd = both).  SyncBatchNorm all-reduces `workspace` between them and
// passes the global pixel count and 1 / world_size for the parameter-gradient contribution (dgamma / dbeta stay per-rank sums
// after the gradient all-reduce, as torch.nn.SyncBatchNorm + DDP produce).
extern "C" int dsn_bn_act_bwd_reduce(const dsn_tensor* dz, const dsn_tensor* y, const float* scale, const float* shift,
                                     const float* mean, const float* rstd, int32_t act, void* workspace,
                                     int64_t workspace_bytes, void* stream) {
    DSN_CHECK_ARG(tensor_ok(dz) && tensor_ok(y) && same_shape(dz, y), "bn_act_bwd_reduce: invalid tensors");
    DSN_CHECK_ARG(scale && shift && mean && rstd && workspace, "bn_act_bwd_reduce: null argument");
    if (workspace_bytes < dsn_bn_workspace_bytes(y->c)) DSN_FAIL(DSN_EWORKSPACE, "bn_act_bwd_reduce: workspace too small");
    DSN_CHECK_ARG(y->c <= PRO_MAXC, "bn_act_bwd_reduce: at most %d channels", PRO_MAXC);
    const int64_t P = npix(y);
    hipStream_t st = (hipStream_t)stream;
    const bool v = vec_ok(y) && vec_ok(dz);
    const double esz = y->dtype == DSN_F32 ? 4.0 : 2.0;
    BnParams bp{scale, shift, mean, rstd, nullptr, act, y->c};
    BnAcc f{(double*)workspace, y->c, (double)P};
    ProfScope prof(KID_BN_BWD_REDUCE, 0.0, 2.0 * P * y->c * esz, st);
    DSN_DISPATCH_DTYPE(y->dtype, T, (launch_reduce<T, true, BwdRedF>(v, y, dz, f, st, bp)));
    DSN_LAUNCH_CHECK("bn_act_bwd reduce");
    return DSN_OK;
}

// The same sums for a channel SLICE of a block (dz, y: the slice's views; scale .. rstd: the slice's parameters), added at channel
// ch0 of a wider accumulator ([DSN_BN_NREP][2][acc_c] doubles): the rest of the block's channels got theirs in the epilogue of
// the input-gradient convolution that completed their dz (dsn_conv2d_dgrad_bnred).
extern "C" int dsn_bn_act_bwd_reduce_into(const dsn_tensor* dz, const dsn_tensor* y, const float* scale, const float* shift,
                                          const float* mean, const float* rstd, int32_t act, void* acc, int32_t acc_c, int32_t ch0,
                                          void* stream) {
    DSN_CHECK_ARG(tensor_ok(dz) && tensor_ok(y) && same_shape(dz, y), "bn_act_bwd_reduce_into: invalid tensors");
    DSN_CHECK_ARG(scale && shift && mean && rstd && acc && ch0 >= 0 && acc_c >= ch0 + y->c, "bn_act_bwd_reduce_into: bad argument");
    DSN_CHECK_ARG(y->c <= PRO_MAXC, "bn_act_bwd_reduce_into: at most %d channels", PRO_MAXC);
    const int64_t P = npix(y);
    hipStream_t st = (hipStream_t)stream;
    const bool v = vec_ok(y) && vec_ok(dz);
    const double esz = y->dtype == DSN_F32 ? 4.0 : 2.0;
    BnParams bp{scale, shift, mean, rstd, nullptr, act, y->c};
    BnAcc f{(double*)acc + ch0, acc_c, (double)P};      // (replica stride and the second sum's offset follow acc_c)
    ProfScope prof(KID_BN_BWD_REDUCE, 0.0, 2.0 * P * y->c * esz, st);
    DSN_DISPATCH_DTYPE(y->dtype, T, (launch_reduce<T, true, BwdRedF>(v, y, dz, f, st, bp)));
    DSN_LAUNCH_CHECK("bn_act_bwd reduce (slice)");
    return DSN_OK;
}

extern "C" int dsn_bn_act_bwd_apply(const dsn_tensor* dz, const dsn_tensor* y, const float* scale, const float* shift,
                                    const float* mean, const float* rstd, int32_t act, const dsn_tensor* dy, float* dgamma,
                                    float* dbeta, int32_t accumulate, const void* workspace, int64_t workspace_bytes,
                                    double count, float pgrad_scale, const dsn_bn_split* second, void* stream) {
    DSN_CHECK_ARG(tensor_ok(dz) && tensor_ok(y) && tensor_ok(dy) && same_shape(dz, y) && same_shape(dy, y),
                  "bn_act_bwd_apply: invalid tensors");
    DSN_CHECK_ARG(scale && shift && mean && rstd && workspace, "bn_act_bwd_apply: null argument");
    if (workspace_bytes < dsn_bn_workspace_bytes(y->c)) DSN_FAIL(DSN_EWORKSPACE, "bn_act_bwd_apply: workspace too small");
    DSN_CHECK_ARG(y->c <= PRO_MAXC, "bn_act_bwd_apply: at most %d channels", PRO_MAXC);
    const int64_t P = npix(y);
    hipStream_t st = (hipStream_t)stream;
    const bool v = vec_ok(y) && vec_ok(dz) && vec_ok(dy);
    const double esz = y->dtype == DSN_F32 ? 4.0 : 2.0;
    BnParams bp{scale, shift, mean, rstd, nullptr, act, y->c};
    EwPro pro{};
    pro.a = BnAcc{(double*)workspace, y->c, count > 0.0 ? count : (double)P};
    pro.mode = 1;
    pro.scale = (float*)scale; pro.shift = (float*)shift; pro.mean = (float*)mean; pro.rstd = (float*)rstd;   // read only
    pro.dgamma = dgamma; pro.dbeta = dbeta; pro.accumulate = accumulate;
    pro.pgrad_scale = pgrad_scale;
    if (second && second->split_c > 0) {
        DSN_CHECK_ARG(second->split_c < y->c, "bn_act_bwd_apply: bad split");
        pro.split = second->split_c;
        pro.dgamma2 = second->dgamma; pro.dbeta2 = second->dbeta;
    }
    ProfScope prof(KID_BN_BWD_APPLY, 0.0, 3.0 * P * y->c * esz, st);
    DSN_DISPATCH_DTYPE(y->dtype, T, (launch_ew<T, true, BwdApplyF>(v, y, dz, dy, &pro, st, bp)));
    DSN_LAUNCH_CHECK("bn_act_bwd apply");
    return DSN_OK;
}

extern "C" int dsn_bn_act_bwd(const dsn_tensor* dz, const dsn_tensor* y, const float* scale, const float* shift,
                              const float* mean, const float* rstd, int32_t act, const dsn_tensor* dy, float* dgamma,
                              float* dbeta, int32_t accumulate, void* workspace, int64_t workspace_bytes,
                              void* stream) {
    DSN_CHECK_ARG(tensor_ok(dy), "bn_act_bwd: invalid tensors");
    int rc = dsn_bn_act_bwd_reduce(dz, y, scale, shift, mean, rstd, act, workspace, workspace_bytes, stream);
    if (rc) return rc;
    return dsn_bn_act_bwd_apply(dz, y, scale, shift, mean, rstd, act, dy, dgamma, dbeta, accumulate, workspace,
                                workspace_bytes, 0.0, 1.f, nullptr, stream);
}

static int lazy_check(const dsn_lazy_in* l, const dsn_tensor* t, const char* what) {
    if (!l) return DSN_OK;
    DSN_CHECK_ARG(l->nseg >= 0 && l->nseg <= DSN_LAZY_MAXSEG, "%s: %d segments", what, l->nseg);
    for (int i = 0; i < l->nseg; ++i) {
        const dsn_lazy_seg& s = l->seg[i];
        DSN_CHECK_ARG(s.c0 >= 0 && s.c1 > s.c0 && s.c1 <= t->c, "%s: segment %d covers [%d, %d) of %d channels", what, i, s.c0, s.c1, t->c);
        if (s.c0 % 8 || s.c1 % 8) DSN_FAIL(DSN_EUNSUPPORTED, "%s: segment bounds must be multiples of 8", what);
        DSN_CHECK_ARG(!s.acc || (s.count > 0 && s.acc_c >= s.ch0 + (s.c1 - s.c0)), "%s: bad accumulator in segment %d", what, i);
        DSN_CHECK_ARG((s.scale == nullptr) == (s.shift == nullptr), "%s: scale/shift must come in pairs", what);
    }
    return DSN_OK;
}

// z = lazy(x) [+ lazy(residual)]: materialises a deferred-BatchNorm tensor (or concat) -- see lazy_ew_kernel.
extern "C" int dsn_lazy_materialize(const dsn_tensor* x, const dsn_lazy_in* lx, const dsn_tensor* residual,
                                    const dsn_lazy_in* lres, const dsn_tensor* z, void* stream) {
    DSN_CHECK_ARG(tensor_ok(x) && tensor_ok(z) && same_shape(x, z), "lazy_materialize: invalid tensors");
    if (residual) DSN_CHECK_ARG(tensor_ok(residual) && same_shape(x, residual), "lazy_materialize: residual mismatch");
    int rc = lazy_check(lx, x, "lazy_materialize");
    if (rc) return rc;
    if (residual && (rc = lazy_check(lres, residual, "lazy_materialize (residual)"))) return rc;
    if (x->c > PRO_MAXC || !vec_ok(x) || !vec_ok(z) || (residual && !vec_ok(residual)))
        DSN_FAIL(DSN_EUNSUPPORTED, "lazy_materialize: needs 16-byte channel vectors and at most %d channels", PRO_MAXC);
    const LazyIn la = lx ? *lx : LazyIn{}, lr = (residual && lres) ? *lres : LazyIn{};
    const int64_t P = npix(x);
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(KID_BN_ACT_FWD, 0.0, (double)P * x->c * (x->dtype == DSN_F32 ? 4.0 : 2.0) * (residual ? 3 : 2), st);
    DSN_DISPATCH_DTYPE(x->dtype, T, {
        constexpr int VV = VW<T>::N;
        int nb;
        Strip s = make_strip(P, x->c, VV, 16384, &nb);
        if (residual)
            hipLaunchKernelGGL((lazy_ew_kernel<T, VV, true>), dim3(nb), dim3(THREADS), 0, st, (const T*)x->ptr, x->ldc,
                               (const T*)residual->ptr, residual->ldc, (T*)z->ptr, z->ldc, s, la, lr);
        else
            hipLaunchKernelGGL((lazy_ew_kernel<T, VV, false>), dim3(nb), dim3(THREADS), 0, st, (const T*)x->ptr, x->ldc,
                               (const T*)nullptr, (int64_t)0, (T*)z->ptr, z->ldc, s, la, lr);
    });
    DSN_LAUNCH_CHECK("lazy_materialize");
    return DSN_OK;
}

extern "C" int dsn_bn_finalize_multi(const dsn_bn_final* entries, int32_t n, void* stream) {
    DSN_CHECK_ARG(entries && n > 0, "bn_finalize_multi: bad arguments");
    for (int base = 0; base < n; base += FINAL_MAX) {
        FinalTable t{};
        t.n = n - base < FINAL_MAX ? n - base : FINAL_MAX;
        int blocks = 0;
        for (int i = 0; i < t.n; ++i) {
            const dsn_bn_final& e = entries[base + i];
            DSN_CHECK_ARG(e.acc && e.n > 0 && e.acc_c >= e.ch0 + e.n && e.count > 0 && e.scale && e.shift && e.mean && e.rstd &&
                              (e.running_mean == nullptr) == (e.running_var == nullptr),
                          "bn_finalize_multi: bad entry %d", base + i);
            t.e[i] = e;
            t.first[i] = blocks;
            blocks += (e.n + 255) / 256;
            DSN_CHECK_ARG(blocks <= INT16_MAX, "bn_finalize_multi: %d blocks do not fit the table's 16-bit block index", blocks);
        }
        t.first[t.n] = (int16_t)blocks;
        hipLaunchKernelGGL(bn_finalize_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t);
        DSN_LAUNCH_CHECK("bn_finalize_multi");
    }
    return DSN_OK;
}

extern "C" int dsn_act_bwd(const dsn_tensor* dz, const dsn_tensor* y, int32_t act, const dsn_tensor* dy, void* stream) {
    DSN_CHECK_ARG(tensor_ok(dz) && tensor_ok(y) && tensor_ok(dy) && same_shape(dz, y) && same_shape(dy, y),
                  "act_bwd: invalid tensors");
    const bool v = vec_ok(y) && vec_ok(dz) && vec_ok(dy);
    BnParams bp{nullptr, nullptr, nullptr, nullptr, nullptr, act, y->c};
    DSN_DISPATCH_DTYPE(y->dtype, T, (launch_ew<T, true, BwdApplyF>(v, y, dz, dy, nullptr, (hipStream_t)stream, bp)));
    DSN_LAUNCH_CHECK("act_bwd");
    return DSN_OK;
}
