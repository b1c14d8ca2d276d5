// Data-movement kernels of the hot path: Focus space-to-depth, SPP max-pools, nearest / bilinear(align_corners) resampling,
// adaptive average pools, FFM channel scaling -- forward and backward.  All HBM/L2-bound; backward passes are written as
// GATHERS (each destination element sums its own contributions) so results are deterministic and need no atomics.
#include "common.h"

namespace {

inline int ew_grid(int64_t total) {
    int64_t b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}
#define GRID_STRIDE(i, total) \
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (total); i += (int64_t)gridDim.x * blockDim.x)

// (pixel p = i / ncv, cv, w, h, n) of the flat index i over [N][H][W][ncv].  32-bit divisions whenever the index fits: a 64-bit
// division is ~4x the instructions, and with four of them per 16-byte vector the upsampling kernels spent more on index
// arithmetic than on memory.
__device__ __forceinline__ void split_nhwc(int64_t i, int ncv, int W, int H, int64_t& p, int& cv, int& w, int& h, int& n) {
    if (i < (1ll << 32)) {
        const unsigned u = (unsigned)i, q = u / (unsigned)ncv, r = q / (unsigned)W, m = r / (unsigned)H;
        p = q; cv = (int)(u - q * (unsigned)ncv); w = (int)(q - r * (unsigned)W); h = (int)(r - m * (unsigned)H); n = (int)m;
    } else {
        p = i / ncv; cv = (int)(i - p * ncv);
        const int64_t r = p / W, m = r / H;
        w = (int)(p - r * W); h = (int)(r - m * H); n = (int)m;
    }
}

// ---- Focus: y[n][h][w][g*C + c] = x[n][c][2h + (g&1)][2w + (g>>1)]                       (common.py:626) -------------
// Input element: fp32 (already normalised), or uint8 pixels -- then the loader's `imgs.float() / 255.0` (train.py:329,
// val.py:213, detect.py:129) happens here: a correctly rounded fp32 division, bit-identical to ATen's.
__device__ __forceinline__ float focus_in(float v) { return v; }
__device__ __forceinline__ float focus_in(uint8_t v) { return (float)v / 255.0f; }

template <typename T, typename IN>
__global__ void focus_s2d_kernel(const IN* __restrict__ x, T* __restrict__ y, int N, int C, int H, int W, int cy,
                                 int64_t yld) {
    const int Ho = H / 2, Wo = W / 2;
    const int64_t total = (int64_t)N * Ho * Wo * cy;
    GRID_STRIDE(i, total) {
        const int ch = (int)(i % cy);
        int64_t p = i / cy;
        const int w = (int)(p % Wo);
        int64_t t = p / Wo;
        const int h = (int)(t % Ho);
        const int n = (int)(t / Ho);
        float v = 0.f;
        if (ch < 4 * C) {
            const int g = ch / C, c = ch - g * C;
            v = focus_in(x[(((int64_t)n * C + c) * H + 2 * h + (g & 1)) * W + 2 * w + (g >> 1)]);
        }
        y[p * yld + ch] = from_f32<T>(v);
    }
}

// One thread per OUTPUT pixel: 2 x C float2 loads (lanes walk w: fully coalesced), the cy channels leave as 16-byte vectors.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef uint8_t u8x2_t __attribute__((ext_vector_type(2)));
template <typename IN> struct Pair;
template <> struct Pair<float> { using type = f32x2_t; };
template <> struct Pair<uint8_t> { using type = u8x2_t; };
template <typename T, int V, int MAXCY, typename IN>
__global__ void focus_s2d_px_kernel(const IN* __restrict__ x, T* __restrict__ y, int N, int C, int H, int W, int cy,
                                    int64_t yld) {
    using P2 = typename Pair<IN>::type;
    const int Ho = H / 2, Wo = W / 2;
    const int64_t total = (int64_t)N * Ho * Wo;
    GRID_STRIDE(p, total) {
        int64_t pp;
        int cv0, w, h, n;
        split_nhwc(p, 1, Wo, Ho, pp, cv0, w, h, n);
        float o[MAXCY];
#pragma unroll
        for (int k = 0; k < MAXCY; ++k) o[k] = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c >= C) break;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const P2 v = *reinterpret_cast<const P2*>(x + (((int64_t)n * C + c) * H + 2 * h + r) * W + 2 * w);
                o[r * C + c] = focus_in(v[0]);              // g = r      (column 2w)
                o[(r + 2) * C + c] = focus_in(v[1]);        // g = r + 2  (column 2w + 1)
            }
        }
#pragma unroll
        for (int k = 0; k < MAXCY; k += V) {
            if (k >= cy) break;
            float part[V];
#pragma unroll
            for (int e = 0; e < V; ++e) part[e] = o[k + e];
            VecIO<T, V>::store(y + p * yld + k, part);
        }
    }
}

// ---- stride-1 max pool with -inf padding; first maximum in row-major window order (ATen max_pool2d) ----------------------
template <typename T>
__global__ void maxpool_kernel(const T* __restrict__ x, int64_t xld, T* __restrict__ y, int64_t yld,
                               int32_t* __restrict__ idx, int N, int H, int W, int C, int k) {
    const int r = k / 2;
    const int64_t total = (int64_t)N * H * W * C;
    GRID_STRIDE(i, total) {
        const int c = (int)(i % C);
        const int64_t p = i / C;
        const int w = (int)(p % W);
        const int64_t t = p / W;
        const int h = (int)(t % H);
        const int n = (int)(t / H);
        float best = -INFINITY;
        int bi = -1;
        const int h0 = h - r < 0 ? 0 : h - r, h1 = h + r >= H ? H - 1 : h + r;
        const int w0 = w - r < 0 ? 0 : w - r, w1 = w + r >= W ? W - 1 : w + r;
        for (int hh = h0; hh <= h1; ++hh)
            for (int ww = w0; ww <= w1; ++ww) {
                const int q = hh * W + ww;
                const float v = to_f32<T>(x[((int64_t)n * H * W + q) * xld + c]);
                if (v > best || bi < 0 || v != v) { best = v; bi = q; }
            }
        y[p * yld + c] = from_f32<T>(best);
        if (idx) idx[i] = bi;
    }
}

// Stride-1 max pools of ONE input at up to three window sizes (SPP: 5 / 9 / 13) in one launch.  A block owns one image and
// one 16-byte channel vector: the [H*W][V] tile is loaded into LDS once, and each pool is computed separably -- horizontal
// (max, first column) per pixel, then vertical over rows -- 2k LDS reads per output instead of k*k global ones.  "First row
// holding the maximum, first column inside that row" is exactly ATen's first maximum in row-major window order.
struct MaxFwd {
    void* y[3];
    int64_t ld[3];
    int32_t* idx[3];
    int32_t k[3];
    int32_t n;
};
template <typename T, int V>
__global__ __launch_bounds__(256) void maxpool_multi_kernel(const T* __restrict__ x, int64_t xld, const MaxFwd out, int H, int W,
                                                            int C) {
    extern __shared__ float smp[];
    const int HW = H * W;
    float* sx = smp;                                   // [V][HW] input planes (lane = pixel: conflict-free LDS)
    float* hv = smp + (size_t)HW * V;                  // [V][HW] horizontal max
    int* hc = reinterpret_cast<int*>(hv + (size_t)HW * V);   // [V][HW] its column
    const int ncv = C / V;
    const int tile = blockIdx.x / out.n, b0 = blockIdx.x % out.n;      // one pool size per block: 3x the blocks, the tile
    const int n = tile / ncv, cv = tile % ncv;                          // (a few KB out of L2) is simply loaded again
    for (int p = threadIdx.x; p < HW; p += 256) {
        float v[V];
        VecIO<T, V>::load(x + ((int64_t)n * HW + p) * xld + cv * V, v);
#pragma unroll
        for (int k = 0; k < V; ++k) sx[k * HW + p] = v[k];
    }
    __syncthreads();
    for (int b = b0; b <= b0; ++b) {
        const int r = out.k[b] / 2;
        for (int p = threadIdx.x; p < HW; p += 256) {
            const int row = p / W, w = p - row * W;
            const int w0 = w - r < 0 ? 0 : w - r, w1 = w + r >= W ? W - 1 : w + r;
#pragma unroll
            for (int k = 0; k < V; ++k) {
                float best = -INFINITY;
                int bc = -1;
                for (int col = w0; col <= w1; ++col) {
                    const float val = sx[k * HW + row * W + col];
                    if (val > best || bc < 0 || val != val) { best = val; bc = col; }
                }
                hv[k * HW + p] = best;
                hc[k * HW + p] = bc;
            }
        }
        __syncthreads();
        T* y = (T*)out.y[b];
        for (int p = threadIdx.x; p < HW; p += 256) {
            const int h = p / W, w = p - h * W;
            const int h0 = h - r < 0 ? 0 : h - r, h1 = h + r >= H ? H - 1 : h + r;
            float res[V];
            int q[V];
#pragma unroll
            for (int k = 0; k < V; ++k) {
                float best = -INFINITY;
                int bi = -1;
                for (int row = h0; row <= h1; ++row) {
                    const float val = hv[k * HW + row * W + w];
                    if (val > best || bi < 0 || val != val) { best = val; bi = row * W + hc[k * HW + row * W + w]; }
                }
                res[k] = best;
                q[k] = bi;
            }
            const int64_t op = (int64_t)n * HW + p;
            VecIO<T, V>::store(y + op * out.ld[b] + cv * V, res);
            if (out.idx[b]) {
#pragma unroll
                for (int k = 0; k < V; k += 4)
                    *reinterpret_cast<u32x4*>(out.idx[b] + op * C + cv * V + k) =
                        u32x4{(uint32_t)q[k], (uint32_t)q[k + 1], (uint32_t)q[k + 2], (uint32_t)q[k + 3]};
            }
        }
        __syncthreads();
    }
}

// The three SPP pools as a CASCADE (round 3): window sizes k0, 2 k0 - 1, 3 k0 - 2 (5 / 9 / 13, common.py:172-185) are pool_k0
// applied once, twice, three times -- with the clipped (-inf padded) windows of MaxPool2d the reachable set of the composition is
// exactly the larger clipped window.  ATen's "first maximum in row-major window order" is the maximum of (value, flat index) pairs
// under (value descending, index ascending), which IS associative, so the cascade carries pairs and reproduces the arg-max bit for
// bit: 6 separable passes of k0 pair reads per element instead of 2 (5 + 9 + 13).  A block owns one image x one 16-byte channel
// vector for ALL three outputs (the old kernel: one block per output, 256 threads, 42 us on 8 x 256 x 20 x 20).
template <typename T, int V, int NT>
__global__ __launch_bounds__(NT) void maxpool_cascade_kernel(const T* __restrict__ x, int64_t xld, const MaxFwd out, int H, int W,
                                                             int C, int r) {
    extern __shared__ float smp[];
    const int HW = H * W, PL = HW * V;
    float* val[2] = {smp, smp + PL};
    int* idx[2] = {reinterpret_cast<int*>(smp + 2 * PL), reinterpret_cast<int*>(smp + 3 * PL)};
    const int ncv = C / V;
    const int n = blockIdx.x / ncv, cv = blockIdx.x % ncv;
    for (int p = threadIdx.x; p < HW; p += NT) {
        float v[V];
        VecIO<T, V>::load(x + ((int64_t)n * HW + p) * xld + cv * V, v);
#pragma unroll
        for (int k = 0; k < V; ++k) { val[0][k * HW + p] = v[k]; idx[0][k * HW + p] = p; }
    }
    __syncthreads();
    // (value, index) a beats b: larger value, or equal value at a smaller flat index; a NaN always wins (as ATen's scan keeps it)
    auto better = [](float v, int i, float bv, int bi) { return v > bv || (v == bv && i < bi) || v != v; };
    for (int stage = 0; stage < out.n; ++stage) {
        // row pass: buffer 0 -> 1
        for (int it = threadIdx.x; it < PL; it += NT) {
            const int k = it / HW, p = it - k * HW;
            const int row = p / W, w = p - row * W;
            const int w0 = w - r < 0 ? 0 : w - r, w1 = w + r >= W ? W - 1 : w + r;
            const float* sv = val[0] + k * HW + row * W;
            const int* si = idx[0] + k * HW + row * W;
            float bv = sv[w0];
            int bi = si[w0];
            for (int c = w0 + 1; c <= w1; ++c) {
                const float v = sv[c];
                const int i = si[c];
                if (better(v, i, bv, bi)) { bv = v; bi = i; }
            }
            val[1][it] = bv;
            idx[1][it] = bi;
        }
        __syncthreads();
        // column pass: buffer 1 -> 0
        for (int it = threadIdx.x; it < PL; it += NT) {
            const int k = it / HW, p = it - k * HW;
            const int h = p / W, w = p - h * W;
            const int h0 = h - r < 0 ? 0 : h - r, h1 = h + r >= H ? H - 1 : h + r;
            const float* sv = val[1] + k * HW + w;
            const int* si = idx[1] + k * HW + w;
            float bv = sv[h0 * W];
            int bi = si[h0 * W];
            for (int rr = h0 + 1; rr <= h1; ++rr) {
                const float v = sv[rr * W];
                const int i = si[rr * W];
                if (better(v, i, bv, bi)) { bv = v; bi = i; }
            }
            val[0][it] = bv;
            idx[0][it] = bi;
        }
        __syncthreads();
        T* y = (T*)out.y[stage];
        for (int p = threadIdx.x; p < HW; p += NT) {
            float res[V];
            int q[V];
#pragma unroll
            for (int k = 0; k < V; ++k) { res[k] = val[0][k * HW + p]; q[k] = idx[0][k * HW + p]; }
            const int64_t op = (int64_t)n * HW + p;
            VecIO<T, V>::store(y + op * out.ld[stage] + cv * V, res);
            if (out.idx[stage]) {
#pragma unroll
                for (int k = 0; k < V; k += 4)
                    *reinterpret_cast<u32x4*>(out.idx[stage] + op * C + cv * V + k) =
                        u32x4{(uint32_t)q[k], (uint32_t)q[k + 1], (uint32_t)q[k + 2], (uint32_t)q[k + 3]};
            }
        }
        // (the next stage's row pass only reads buffer 0 and writes buffer 1: no barrier needed between this store loop and it)
    }
}

// dx[n][q][c] (+)= sum over outputs p whose window contains q and whose arg-max is q
template <typename T>
__global__ void maxpool_bwd_kernel(const T* __restrict__ dy, int64_t yld, const int32_t* __restrict__ idx,
                                   T* __restrict__ dx, int64_t xld, int N, int H, int W, int C, int k, int accumulate) {
    const int r = k / 2;
    const int64_t total = (int64_t)N * H * W * C;
    GRID_STRIDE(i, total) {
        const int c = (int)(i % C);
        const int64_t p = i / C;
        const int w = (int)(p % W);
        const int64_t t = p / W;
        const int h = (int)(t % H);
        const int n = (int)(t / H);
        const int q = h * W + w;
        float s = 0.f;
        const int h0 = h - r < 0 ? 0 : h - r, h1 = h + r >= H ? H - 1 : h + r;
        const int w0 = w - r < 0 ? 0 : w - r, w1 = w + r >= W ? W - 1 : w + r;
        for (int hh = h0; hh <= h1; ++hh)
            for (int ww = w0; ww <= w1; ++ww) {
                const int64_t op = (int64_t)n * H * W + hh * W + ww;
                if (idx[op * C + c] == q) s += to_f32<T>(dy[op * yld + c]);
            }
        T* o = dx + p * xld + c;
        if (accumulate) s += to_f32<T>(*o);
        *o = from_f32<T>(s);
    }
}

// ---- nearest x2 ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ void up2_kernel(const T* __restrict__ x, int64_t xld, T* __restrict__ y, int64_t yld, int N, int H, int W,
                           int C) {
    const int Ho = 2 * H, Wo = 2 * W;
    const int64_t total = (int64_t)N * Ho * Wo * C;
    GRID_STRIDE(i, total) {
        const int c = (int)(i % C);
        const int64_t p = i / C;
        const int w = (int)(p % Wo);
        const int64_t t = p / Wo;
        const int h = (int)(t % Ho);
        const int n = (int)(t / Ho);
        y[p * yld + c] = x[(((int64_t)n * H + (h >> 1)) * W + (w >> 1)) * xld + c];
    }
}
template <typename T, int V>
__global__ void up2_vec_kernel(const T* __restrict__ x, int64_t xld, T* __restrict__ y, int64_t yld, int N, int H, int W, int C) {
    const int Ho = 2 * H, Wo = 2 * W, ncv = C / V;
    const int64_t total = (int64_t)N * Ho * Wo * ncv;
    GRID_STRIDE(i, total) {
        int64_t p;
        int cv, w, h, n;
        split_nhwc(i, ncv, Wo, Ho, p, cv, w, h, n);
        *reinterpret_cast<u32x4*>(y + p * yld + cv * V) =
            *reinterpret_cast<const u32x4*>(x + (((int64_t)n * H + (h >> 1)) * W + (w >> 1)) * xld + cv * V);
    }
}
template <typename T, int V>
__global__ void up2_bwd_vec_kernel(const T* __restrict__ dy, int64_t yld, T* __restrict__ dx, int64_t xld, int N, int H, int W,
                                   int C, int accumulate) {
    const int Wo = 2 * W, ncv = C / V;
    const int64_t total = (int64_t)N * H * W * ncv;
    GRID_STRIDE(i, total) {
        int64_t p;
        int cv, w, h, n;
        split_nhwc(i, ncv, W, H, p, cv, w, h, n);
        const int64_t q = ((int64_t)n * 2 * H + 2 * h) * Wo + 2 * w;
        float a[V], b[V], c[V], d[V];
        VecIO<T, V>::load(dy + q * yld + cv * V, a);
        VecIO<T, V>::load(dy + (q + 1) * yld + cv * V, b);
        VecIO<T, V>::load(dy + (q + Wo) * yld + cv * V, c);
        VecIO<T, V>::load(dy + (q + Wo + 1) * yld + cv * V, d);
        T* o = dx + p * xld + cv * V;
        if (accumulate) {
            float old[V];
            VecIO<T, V>::load(o, old);
#pragma unroll
            for (int k = 0; k < V; ++k) a[k] += old[k];
        }
#pragma unroll
        for (int k = 0; k < V; ++k) a[k] += b[k] + c[k] + d[k];
        VecIO<T, V>::store(o, a);
    }
}
template <typename T>
__global__ void up2_bwd_kernel(const T* __restrict__ dy, int64_t yld, T* __restrict__ dx, int64_t xld, int N, int H,
                               int W, int C, int accumulate) {
    const int Wo = 2 * W;
    const int64_t total = (int64_t)N * H * W * C;
    GRID_STRIDE(i, total) {
        const int c = (int)(i % C);
        const int64_t p = i / C;
        const int w = (int)(p % W);
        const int64_t t = p / W;
        const int h = (int)(t % H);
        const int n = (int)(t / H);
        const int64_t q = ((int64_t)n * 2 * H + 2 * h) * Wo + 2 * w;
        float s = to_f32<T>(dy[q * yld + c]) + to_f32<T>(dy[(q + 1) * yld + c]) + to_f32<T>(dy[(q + Wo) * yld + c]) +
                  to_f32<T>(dy[(q + Wo + 1) * yld + c]);
        T* o = dx + p * xld + c;
        if (accumulate) s += to_f32<T>(*o);
        *o = from_f32<T>(s);
    }
}

// ---- bilinear, align_corners=True (ATen upsample_bilinear2d: scale = (in-1)/(out-1), src = scale*dst) ------------------------
struct Lerp { int i0, i1; float l0, l1; };
__device__ __forceinline__ Lerp lerp_coord(int o, float scale, int in) {
    const float s = scale * (float)o;
    Lerp r;
    r.i0 = (int)s;
    if (r.i0 > in - 1) r.i0 = in - 1;
    r.i1 = r.i0 + (r.i0 < in - 1 ? 1 : 0);
    r.l1 = s - (float)r.i0;
    r.l0 = 1.f - r.l1;
    return r;
}
__host__ __device__ inline float ac_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }

template <typename T, bool NCHW_OUT>
__global__ void bilinear_kernel(const T* __restrict__ x, int64_t xld, void* __restrict__ yv, int64_t yld, int N, int Hi,
                                int Wi, int Ho, int Wo, int C, float sh, float sw) {
    const int64_t total = (int64_t)N * Ho * Wo * C;
    GRID_STRIDE(i, total) {
        int c, w, h, n;
        int64_t pq;
        if (NCHW_OUT) split_nhwc(i, Wo, Ho, C, pq, w, h, c, n);     // i enumerates NCHW so the fp32 stores coalesce along w
        else split_nhwc(i, C, Wo, Ho, pq, c, w, h, n);
        const Lerp a = lerp_coord(h, sh, Hi), b = lerp_coord(w, sw, Wi);
        const T* base = x + (int64_t)n * Hi * Wi * xld + c;
        const float v00 = to_f32<T>(base[((int64_t)a.i0 * Wi + b.i0) * xld]);
        const float v01 = to_f32<T>(base[((int64_t)a.i0 * Wi + b.i1) * xld]);
        const float v10 = to_f32<T>(base[((int64_t)a.i1 * Wi + b.i0) * xld]);
        const float v11 = to_f32<T>(base[((int64_t)a.i1 * Wi + b.i1) * xld]);
        const float v = a.l0 * (b.l0 * v00 + b.l1 * v01) + a.l1 * (b.l0 * v10 + b.l1 * v11);
        if (NCHW_OUT)
            ((float*)yv)[i] = v;
        else
            ((T*)yv)[(((int64_t)n * Ho + h) * Wo + w) * yld + c] = from_f32<T>(v);
    }
}

// gather form: dx[n][hi][wi][c] (+)= sum_{ho,wo} wh(ho->hi) * ww(wo->wi) * dy[n][ho][wo][c]
template <typename T, bool NCHW_DY>
__global__ void bilinear_bwd_kernel(const void* __restrict__ dyv, int64_t yld, T* __restrict__ dx, int64_t xld, int N,
                                    int Hi, int Wi, int Ho, int Wo, int C, float sh, float sw, int accumulate) {
    const int64_t total = (int64_t)N * Hi * Wi * C;
    GRID_STRIDE(i, total) {
        const int c = (int)(i % C);
        int64_t t = i / C;
        const int wi = (int)(t % Wi); t /= Wi;
        const int hi = (int)(t % Hi);
        const int n = (int)(t / Hi);
        int h_lo = 0, h_hi = Ho - 1, w_lo = 0, w_hi = Wo - 1;
        if (sh > 0.f) {
            h_lo = (int)floorf((float)(hi - 1) / sh) - 1;
            h_hi = (int)ceilf((float)(hi + 1) / sh) + 1;
            h_lo = h_lo < 0 ? 0 : h_lo;
            h_hi = h_hi > Ho - 1 ? Ho - 1 : h_hi;
        }
        if (sw > 0.f) {
            w_lo = (int)floorf((float)(wi - 1) / sw) - 1;
            w_hi = (int)ceilf((float)(wi + 1) / sw) + 1;
            w_lo = w_lo < 0 ? 0 : w_lo;
            w_hi = w_hi > Wo - 1 ? Wo - 1 : w_hi;
        }
        float s = 0.f;
        for (int ho = h_lo; ho <= h_hi; ++ho) {
            const Lerp a = lerp_coord(ho, sh, Hi);
            const float wh = (a.i0 == hi ? a.l0 : 0.f) + (a.i1 == hi ? a.l1 : 0.f);
            if (wh == 0.f) continue;
            for (int wo = w_lo; wo <= w_hi; ++wo) {
                const Lerp b = lerp_coord(wo, sw, Wi);
                const float ww = (b.i0 == wi ? b.l0 : 0.f) + (b.i1 == wi ? b.l1 : 0.f);
                if (ww == 0.f) continue;
                float g;
                if (NCHW_DY)
                    g = ((const float*)dyv)[(((int64_t)n * C + c) * Ho + ho) * Wo + wo];
                else
                    g = to_f32<T>(((const T*)dyv)[(((int64_t)n * Ho + ho) * Wo + wo) * yld + c]);
                s += wh * ww * g;
            }
        }
        T* o = dx + (((int64_t)n * Hi + hi) * Wi + wi) * xld + c;
        if (accumulate) s += to_f32<T>(*o);
        *o = from_f32<T>(s);
    }
}

// 16-byte channel vectors (NHWC in, NHWC out): one thread = one output pixel x V channels, four 16-byte taps
template <typename T, int V>
__global__ void bilinear_vec_kernel(const T* __restrict__ x, int64_t xld, T* __restrict__ y, int64_t yld, int N, int Hi, int Wi,
                                    int Ho, int Wo, int C, float sh, float sw) {
    const int ncv = C / V;
    const int64_t total = (int64_t)N * Ho * Wo * ncv;
    GRID_STRIDE(i, total) {
        int64_t p;
        int cv, w, h, n;
        split_nhwc(i, ncv, Wo, Ho, p, cv, w, h, n);
        const Lerp a = lerp_coord(h, sh, Hi), b = lerp_coord(w, sw, Wi);
        const T* base = x + (int64_t)n * Hi * Wi * xld + cv * V;
        float v00[V], v01[V], v10[V], v11[V], o[V];
        VecIO<T, V>::load(base + ((int64_t)a.i0 * Wi + b.i0) * xld, v00);
        VecIO<T, V>::load(base + ((int64_t)a.i0 * Wi + b.i1) * xld, v01);
        VecIO<T, V>::load(base + ((int64_t)a.i1 * Wi + b.i0) * xld, v10);
        VecIO<T, V>::load(base + ((int64_t)a.i1 * Wi + b.i1) * xld, v11);
#pragma unroll
        for (int k = 0; k < V; ++k) o[k] = a.l0 * (b.l0 * v00[k] + b.l1 * v01[k]) + a.l1 * (b.l0 * v10[k] + b.l1 * v11[k]);
        VecIO<T, V>::store(y + (((int64_t)n * Ho + h) * Wo + w) * yld + cv * V, o);
    }
}

// up to four (source, destination) pairs in ONE launch: job j owns blocks [blk0[j], blk0[j+1]) (PyramidPooling's branches)
struct BilinearMulti {
    const void* x[4];
    void* y[4];
    int64_t xld[4], yld[4];
    int32_t hi[4], wi[4], c[4];
    float sh[4], sw[4];
    int32_t blk0[5];
    int32_t n, N, Ho, Wo;
};
template <typename T, int V>
__global__ __launch_bounds__(256) void bilinear_vec_multi_kernel(const BilinearMulti m) {
    int j = 0;
    while (j + 1 < m.n && (int)blockIdx.x >= m.blk0[j + 1]) ++j;
    const T* __restrict__ x = (const T*)m.x[j];
    T* __restrict__ y = (T*)m.y[j];
    const int Hi = m.hi[j], Wi = m.wi[j], ncv = m.c[j] / V;
    const int64_t xld = m.xld[j], yld = m.yld[j];
    const float sh = m.sh[j], sw = m.sw[j];
    const int64_t total = (int64_t)m.N * m.Ho * m.Wo * ncv;
    const int nblk = m.blk0[j + 1] - m.blk0[j];
    for (int64_t i = ((int64_t)blockIdx.x - m.blk0[j]) * 256 + threadIdx.x; i < total; i += (int64_t)nblk * 256) {
        int64_t p;
        int cv, w, h, n;
        split_nhwc(i, ncv, m.Wo, m.Ho, p, cv, w, h, n);
        const Lerp a = lerp_coord(h, sh, Hi), b = lerp_coord(w, sw, Wi);
        const T* base = x + (int64_t)n * Hi * Wi * xld + cv * V;
        float v00[V], v01[V], v10[V], v11[V], o[V];
        VecIO<T, V>::load(base + ((int64_t)a.i0 * Wi + b.i0) * xld, v00);
        VecIO<T, V>::load(base + ((int64_t)a.i0 * Wi + b.i1) * xld, v01);
        VecIO<T, V>::load(base + ((int64_t)a.i1 * Wi + b.i0) * xld, v10);
        VecIO<T, V>::load(base + ((int64_t)a.i1 * Wi + b.i1) * xld, v11);
#pragma unroll
        for (int k = 0; k < V; ++k) o[k] = a.l0 * (b.l0 * v00[k] + b.l1 * v01[k]) + a.l1 * (b.l0 * v10[k] + b.l1 * v11[k]);
        VecIO<T, V>::store(y + (((int64_t)n * m.Ho + h) * m.Wo + w) * yld + cv * V, o);
    }
}

// gather form with 16-byte channel vectors (NHWC dy)
template <typename T, int V>
__global__ void bilinear_bwd_vec_kernel(const T* __restrict__ dy, int64_t yld, T* __restrict__ dx, int64_t xld, int N, int Hi,
                                        int Wi, int Ho, int Wo, int C, float sh, float sw, int accumulate) {
    const int ncv = C / V;
    const int64_t total = (int64_t)N * Hi * Wi * ncv;
    GRID_STRIDE(i, total) {
        const int cv = (int)(i % ncv);
        int64_t t = i / ncv;
        const int wi = (int)(t % Wi); t /= Wi;
        const int hi = (int)(t % Hi);
        const int n = (int)(t / Hi);
        int h_lo = 0, h_hi = Ho - 1, w_lo = 0, w_hi = Wo - 1;
        if (sh > 0.f) {
            h_lo = (int)floorf((float)(hi - 1) / sh) - 1;
            h_hi = (int)ceilf((float)(hi + 1) / sh) + 1;
            h_lo = h_lo < 0 ? 0 : h_lo;
            h_hi = h_hi > Ho - 1 ? Ho - 1 : h_hi;
        }
        if (sw > 0.f) {
            w_lo = (int)floorf((float)(wi - 1) / sw) - 1;
            w_hi = (int)ceilf((float)(wi + 1) / sw) + 1;
            w_lo = w_lo < 0 ? 0 : w_lo;
            w_hi = w_hi > Wo - 1 ? Wo - 1 : w_hi;
        }
        float s[V];
#pragma unroll
        for (int k = 0; k < V; ++k) s[k] = 0.f;
        for (int ho = h_lo; ho <= h_hi; ++ho) {
            const Lerp a = lerp_coord(ho, sh, Hi);
            const float wh = (a.i0 == hi ? a.l0 : 0.f) + (a.i1 == hi ? a.l1 : 0.f);
            if (wh == 0.f) continue;
            for (int wo = w_lo; wo <= w_hi; ++wo) {
                const Lerp b = lerp_coord(wo, sw, Wi);
                const float ww = (b.i0 == wi ? b.l0 : 0.f) + (b.i1 == wi ? b.l1 : 0.f);
                if (ww == 0.f) continue;
                float g[V];
                VecIO<T, V>::load(dy + (((int64_t)n * Ho + ho) * Wo + wo) * yld + cv * V, g);
#pragma unroll
                for (int k = 0; k < V; ++k) s[k] += wh * ww * g[k];
            }
        }
        T* o = dx + (((int64_t)n * Hi + hi) * Wi + wi) * xld + cv * V;
        if (accumulate) {
            float old[V];
            VecIO<T, V>::load(o, old);
#pragma unroll
            for (int k = 0; k < V; ++k) s[k] += old[k];
        }
        VecIO<T, V>::store(o, s);
    }
}

// Separable form of the same (round 3): a block owns one dx ROW (n, hi) and a range of channel vectors.  Pass 1 folds the few dy
// rows that touch hi into an fp32 row t[wo][c] in LDS (dy rows are read as whole contiguous NHWC lines); pass 2 folds, for every dx
// pixel of the row, the few columns that touch it.  The gather form above walks a (2 scale + 3)^2 window per dx vector with a lerp
// evaluation per candidate (121 candidates at x4: 38.9 us for the seg head's m32 branch, 13 MB of dy); here every dy row is read
// about twice in total and the lerp arithmetic is per row / per column.
template <typename T, int V>
__global__ __launch_bounds__(256) void bilinear_bwd_rows_kernel(const T* __restrict__ dy, int64_t yld, T* __restrict__ dx, int64_t xld,
                                                                int Hi, int Wi, int Ho, int Wo, int C, int cvb, float sh, float sw,
                                                                int accumulate) {
    extern __shared__ float trow[];                    // [Wo][cvb * V]
    const int nsplit = (C / V + cvb - 1) / cvb;
    const int part = blockIdx.x % nsplit;
    const int row = blockIdx.x / nsplit;               // n * Hi + hi
    const int hi = row % Hi, n = row / Hi;
    const int cv0 = part * cvb;
    const int ncv = (C / V - cv0) < cvb ? (C / V - cv0) : cvb;
    int h_lo = 0, h_hi = Ho - 1;
    if (sh > 0.f) {
        h_lo = (int)floorf((float)(hi - 1) / sh) - 1;
        h_hi = (int)ceilf((float)(hi + 1) / sh) + 1;
        h_lo = h_lo < 0 ? 0 : h_lo;
        h_hi = h_hi > Ho - 1 ? Ho - 1 : h_hi;
    }
    for (int it = threadIdx.x; it < Wo * ncv; it += 256) {
        const int wo = it / ncv, cv = it - wo * ncv;
        float sacc[V];
#pragma unroll
        for (int k = 0; k < V; ++k) sacc[k] = 0.f;
        for (int ho = h_lo; ho <= h_hi; ++ho) {
            const Lerp a = lerp_coord(ho, sh, Hi);
            const float wh = (a.i0 == hi ? a.l0 : 0.f) + (a.i1 == hi ? a.l1 : 0.f);
            if (wh == 0.f) continue;
            float g[V];
            VecIO<T, V>::load(dy + (((int64_t)n * Ho + ho) * Wo + wo) * yld + (cv0 + cv) * V, g);
#pragma unroll
            for (int k = 0; k < V; ++k) sacc[k] += wh * g[k];
        }
#pragma unroll
        for (int k = 0; k < V; ++k) trow[(wo * cvb + cv) * V + k] = sacc[k];
    }
    __syncthreads();
    for (int it = threadIdx.x; it < Wi * ncv; it += 256) {
        const int wi = it / ncv, cv = it - wi * ncv;
        int w_lo = 0, w_hi = Wo - 1;
        if (sw > 0.f) {
            w_lo = (int)floorf((float)(wi - 1) / sw) - 1;
            w_hi = (int)ceilf((float)(wi + 1) / sw) + 1;
            w_lo = w_lo < 0 ? 0 : w_lo;
            w_hi = w_hi > Wo - 1 ? Wo - 1 : w_hi;
        }
        float sacc[V];
#pragma unroll
        for (int k = 0; k < V; ++k) sacc[k] = 0.f;
        for (int wo = w_lo; wo <= w_hi; ++wo) {
            const Lerp b = lerp_coord(wo, sw, Wi);
            const float ww = (b.i0 == wi ? b.l0 : 0.f) + (b.i1 == wi ? b.l1 : 0.f);
            if (ww == 0.f) continue;
#pragma unroll
            for (int k = 0; k < V; ++k) sacc[k] += ww * trow[(wo * cvb + cv) * V + k];
        }
        T* o = dx + (((int64_t)n * Hi + hi) * Wi + wi) * xld + (cv0 + cv) * V;
        if (accumulate) {
            float old[V];
            VecIO<T, V>::load(o, old);
#pragma unroll
            for (int k = 0; k < V; ++k) sacc[k] += old[k];
        }
        VecIO<T, V>::store(o, sacc);
    }
}

// NCHW fp32 dy (the full-resolution seg-logit gradient) -> NHWC dx, separable: a block owns one (n, c, hi) row of dx.
// Pass 1: every thread folds its dy COLUMNS over the few dy rows that touch hi (coalesced along w) into LDS; pass 2: every
// dx pixel of the row folds the few columns that touch it.  The gather form above reads a ~17x17 window per dx element
// (324 strided loads each, 78 us for the 26 MB logit gradient); this reads each needed dy row once per dx row (~2x in total).
template <typename T>
__global__ __launch_bounds__(256) void bilinear_bwd_nchw_rows_kernel(const float* __restrict__ dy, T* __restrict__ dx,
                                                                     int64_t xld, int N, int Hi, int Wi, int Ho, int Wo, int C,
                                                                     float sh, float sw, int accumulate) {
    extern __shared__ float srow[];                 // [Wo]
    int b = blockIdx.x;
    const int hi = b % Hi; b /= Hi;
    const int c = b % C;
    const int n = b / C;
    int h_lo = 0, h_hi = Ho - 1;
    if (sh > 0.f) {
        h_lo = (int)floorf((float)(hi - 1) / sh) - 1;
        h_hi = (int)ceilf((float)(hi + 1) / sh) + 1;
        h_lo = h_lo < 0 ? 0 : h_lo;
        h_hi = h_hi > Ho - 1 ? Ho - 1 : h_hi;
    }
    const float* plane = dy + ((int64_t)n * C + c) * Ho * Wo;
    for (int wo = threadIdx.x; wo < Wo; wo += 256) {
        float s = 0.f;
        for (int ho = h_lo; ho <= h_hi; ++ho) {
            const Lerp a = lerp_coord(ho, sh, Hi);
            const float wh = (a.i0 == hi ? a.l0 : 0.f) + (a.i1 == hi ? a.l1 : 0.f);
            if (wh != 0.f) s += wh * plane[(int64_t)ho * Wo + wo];
        }
        srow[wo] = s;
    }
    __syncthreads();
    for (int wi = threadIdx.x; wi < Wi; wi += 256) {
        int w_lo = 0, w_hi = Wo - 1;
        if (sw > 0.f) {
            w_lo = (int)floorf((float)(wi - 1) / sw) - 1;
            w_hi = (int)ceilf((float)(wi + 1) / sw) + 1;
            w_lo = w_lo < 0 ? 0 : w_lo;
            w_hi = w_hi > Wo - 1 ? Wo - 1 : w_hi;
        }
        float s = 0.f;
        for (int wo = w_lo; wo <= w_hi; ++wo) {
            const Lerp bb = lerp_coord(wo, sw, Wi);
            const float ww = (bb.i0 == wi ? bb.l0 : 0.f) + (bb.i1 == wi ? bb.l1 : 0.f);
            if (ww != 0.f) s += ww * srow[wo];
        }
        T* o = dx + (((int64_t)n * Hi + hi) * Wi + wi) * xld + c;
        if (accumulate) s += to_f32<T>(*o);
        *o = from_f32<T>(s);
    }
}

// ---- adaptive average pool: bin o covers [floor(o*H/k), ceil((o+1)*H/k)) ------------------------------------------------
// a / b for small non-negative operands through the float reciprocal (+-1 correction: exact below 2^22): ~8 instructions against
// ~30 for the integer division sequence -- the bin arithmetic below runs a few dozen of these per wave before any memory access
__device__ __forceinline__ int div_small(int a, int b) {
    if ((unsigned)(a | b) >> 22) return a / b;
    int q = (int)((float)a * __builtin_amdgcn_rcpf((float)b));
    q -= (q * b > a) ? 1 : 0;
    q += ((q + 1) * b <= a) ? 1 : 0;
    return q;
}
__device__ __forceinline__ int bin_lo(int o, int in, int k) { return div_small(o * in, k); }
__device__ __forceinline__ int bin_hi(int o, int in, int k) { return div_small((o + 1) * in + k - 1, k); }

template <typename T>
__global__ void adaptive_avgpool_bwd_kernel(const T* __restrict__ dy, int64_t yld, T* __restrict__ dx, int64_t xld,
                                            int N, int H, int W, int C, int KH, int KW, int accumulate) {
    const int64_t total = (int64_t)N * H * W * C;
    GRID_STRIDE(i, total) {
        const int c = (int)(i % C);
        int64_t t = i / C;
        const int w = (int)(t % W); t /= W;
        const int h = (int)(t % H);
        const int n = (int)(t / H);
        // bins overlap by at most one pixel: only the bins around floor(h*K/H) can contain h
        const int oh0 = (h * KH) / H, ow0 = (w * KW) / W;
        float s = 0.f;
        // (maps smaller than the pool grid have bins overlapping further: scan them all)
        const int oh_lo = H < KH ? 0 : (oh0 > 0 ? oh0 - 1 : 0), oh_hi = H < KH ? KH - 1 : (oh0 + 1 < KH ? oh0 + 1 : KH - 1);
        const int ow_lo = W < KW ? 0 : (ow0 > 0 ? ow0 - 1 : 0), ow_hi = W < KW ? KW - 1 : (ow0 + 1 < KW ? ow0 + 1 : KW - 1);
        for (int oh = oh_lo; oh <= oh_hi; ++oh) {
            const int h0 = bin_lo(oh, H, KH), h1 = bin_hi(oh, H, KH);
            if (h < h0 || h >= h1) continue;
            for (int ow = ow_lo; ow <= ow_hi; ++ow) {
                const int w0 = bin_lo(ow, W, KW), w1 = bin_hi(ow, W, KW);
                if (w < w0 || w >= w1) continue;
                s += to_f32<T>(dy[(((int64_t)n * KH + oh) * KW + ow) * yld + c]) / (float)((h1 - h0) * (w1 - w0));
            }
        }
        T* o = dx + (((int64_t)n * H + h) * W + w) * xld + c;
        if (accumulate) s += to_f32<T>(*o);
        *o = from_f32<T>(s);
    }
}

// Vectorised multi-source form: dx (+)= sum_i avgpool_bwd(dy_i) for up to 4 pooled gradients with different grids, ONE
// pass over dx with 16-byte accesses (PyramidPooling backward: four pools accumulate into the same dx).
struct PoolSrcs {
    const void* dy[4];
    int64_t ld[4];
    int32_t KH[4], KW[4];
    int32_t n;
};
template <typename T, int V>
__global__ void adaptive_avgpool_bwd_multi_kernel(const PoolSrcs srcs, T* __restrict__ dx, int64_t xld, int N, int H, int W,
                                                  int C, int accumulate) {
    const int ncv = C / V;
    const int64_t total = (int64_t)N * H * W * ncv;
    GRID_STRIDE(i, total) {
        const int cv = (int)(i % ncv);
        int64_t t = i / ncv;
        const int w = (int)(t % W); t /= W;
        const int h = (int)(t % H);
        const int n = (int)(t / H);
        float s[V];
#pragma unroll
        for (int k = 0; k < V; ++k) s[k] = 0.f;
        for (int b = 0; b < srcs.n; ++b) {
            const int KH = srcs.KH[b], KW = srcs.KW[b];
            const T* dy = (const T*)srcs.dy[b];
            const int oh0 = (h * KH) / H, ow0 = (w * KW) / W;
            const int oh_lo = H < KH ? 0 : (oh0 > 0 ? oh0 - 1 : 0), oh_hi = H < KH ? KH - 1 : (oh0 + 1 < KH ? oh0 + 1 : KH - 1);
            const int ow_lo = W < KW ? 0 : (ow0 > 0 ? ow0 - 1 : 0), ow_hi = W < KW ? KW - 1 : (ow0 + 1 < KW ? ow0 + 1 : KW - 1);
            for (int oh = oh_lo; oh <= oh_hi; ++oh) {
                const int h0 = bin_lo(oh, H, KH), h1 = bin_hi(oh, H, KH);
                if (h < h0 || h >= h1) continue;
                for (int ow = ow_lo; ow <= ow_hi; ++ow) {
                    const int w0 = bin_lo(ow, W, KW), w1 = bin_hi(ow, W, KW);
                    if (w < w0 || w >= w1) continue;
                    float v[V];
                    VecIO<T, V>::load(dy + (((int64_t)n * KH + oh) * KW + ow) * srcs.ld[b] + cv * V, v);
                    const float area = (float)((h1 - h0) * (w1 - w0));
#pragma unroll
                    for (int k = 0; k < V; ++k) s[k] += v[k] / area;
                }
            }
        }
        T* o = dx + (((int64_t)n * H + h) * W + w) * xld + cv * V;
        if (accumulate) {
            float old[V];
            VecIO<T, V>::load(o, old);
#pragma unroll
            for (int k = 0; k < V; ++k) s[k] += old[k];
        }
        VecIO<T, V>::store(o, s);
    }
}

// Row form of the same (round 3): a block owns one dx row (n, h).  The pooled rows that contain h (at most two per source:
// adaptive bins overlap by one row when H is not a multiple of the grid) are block-uniform, and the bins of every column are
// tabulated once per block in LDS -- the element loop is left with loads, multiplies by a reciprocal bin area and adds.  (The form
// above spends ~100 integer divisions per 16-byte vector on bin arithmetic: 37 us for PyramidPooling's 8 x 128 x 80 x 80 dx.)
// NS: number of sources (compile time).  Averaging is separable, so the block first folds, per source, the (at most two) pooled
// rows that contain h into ONE fp32 row  rv[b][ow][c] = sum_a dy_b[n][oh_a][ow][c] / height_a  in LDS (a few KB: the pooled maps
// are 1 .. 6 wide); the element loop then touches global memory for dx only and reads its (at most two per source) column bins
// from LDS.  Reading the bins from global memory per element -- up to 16 loads of the same few KB per 16-byte vector, from every
// block of the launch -- took 26 us for PyramidPooling's 13 MB dx whichever way the loads were arranged.
// blockIdx.y: segment of wseg columns.
template <typename T, int V, int NS>
__global__ __launch_bounds__(256) void adaptive_avgpool_bwd_rows_kernel(const PoolSrcs srcs, T* __restrict__ dx, int64_t xld, int H,
                                                                        int W, int C, int accumulate, int wseg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pool_smem[];
    int koff[NS + 1];
    koff[0] = 0;
#pragma unroll
    for (int b = 0; b < NS; ++b) koff[b + 1] = koff[b] + srcs.KW[b];
    float* rv = reinterpret_cast<float*>(pool_smem);                                   // [sum KW][C]
    int* ctab = reinterpret_cast<int*>(pool_smem + (size_t)koff[NS] * C * 4);          // [NS][wseg][4]: ow0, ow1 (or -1), 1/width bits
    const int h = blockIdx.x % H, n = blockIdx.x / H;
    const int ncv = C / V;
    const int w_lo = blockIdx.y * wseg, w_n = (W - w_lo) < wseg ? (W - w_lo) : wseg;
    for (int it = threadIdx.x; it < NS * w_n; it += 256) {
        const int b = div_small(it, w_n), wl = it - b * w_n, w = w_lo + wl;
        const int KW = srcs.KW[b];
        const int ow0 = div_small(w * KW, W);
        int cand[3] = {ow0 - 1, ow0, ow0 + 1}, got = 0;
        int o[2] = {-1, -1};
        float inv[2] = {0.f, 0.f};
        for (int j = 0; j < 3; ++j) {
            const int ow = cand[j];
            if (ow < 0 || ow >= KW || got == 2) continue;
            const int w0 = bin_lo(ow, W, KW), w1 = bin_hi(ow, W, KW);
            if (w < w0 || w >= w1) continue;
            o[got] = ow;
            inv[got] = 1.f / (float)(w1 - w0);
            ++got;
        }
        int* e = ctab + (b * wseg + wl) * 4;
        e[0] = o[0]; e[1] = o[1]; e[2] = __float_as_int(inv[0]); e[3] = __float_as_int(inv[1]);
    }
    // rows: block-uniform (scalar) arithmetic
    int oh[NS][2];
    float ih[NS][2];
#pragma unroll
    for (int b = 0; b < NS; ++b) {
        oh[b][0] = oh[b][1] = -1;
        ih[b][0] = ih[b][1] = 0.f;
        const int KH = srcs.KH[b];
        const int oh0 = div_small(h * KH, H);
        bool first = true;
#pragma unroll
        for (int j = -1; j <= 1; ++j) {
            const int c = oh0 + j;
            if (c < 0 || c >= KH) continue;
            const int h0 = bin_lo(c, H, KH), h1 = bin_hi(c, H, KH);
            if (h < h0 || h >= h1) continue;
            if (first) { oh[b][0] = c; ih[b][0] = 1.f / (float)(h1 - h0); first = false; }
            else { oh[b][1] = c; ih[b][1] = 1.f / (float)(h1 - h0); }
        }
    }
    // the row-combined pooled vectors (both candidate rows loaded unconditionally: a missing one reads row 0 with weight 0)
    for (int it = threadIdx.x; it < koff[NS] * ncv; it += 256) {
        const int col = div_small(it, ncv), cv = it - col * ncv;
        int b = 0;
#pragma unroll
        for (int j = 1; j < NS; ++j) b += col >= koff[j] ? 1 : 0;
        int r0 = 0, r1 = 0, k0 = 0, KH = 1, KW = 1;
        float f0 = 0.f, f1 = 0.f;
        const T* dy = nullptr;
        int64_t ld = 0;
#pragma unroll
        for (int j = 0; j < NS; ++j)
            if (b == j) {
                r0 = oh[j][0] >= 0 ? oh[j][0] : 0; f0 = oh[j][0] >= 0 ? ih[j][0] : 0.f;
                r1 = oh[j][1] >= 0 ? oh[j][1] : 0; f1 = oh[j][1] >= 0 ? ih[j][1] : 0.f;
                k0 = koff[j]; KH = srcs.KH[j]; KW = srcs.KW[j]; dy = (const T*)srcs.dy[j]; ld = srcs.ld[j];
            }
        const int ow = col - k0;
        float v0[V], v1[V];
        VecIO<T, V>::load(dy + (((int64_t)n * KH + r0) * KW + ow) * ld + cv * V, v0);
        VecIO<T, V>::load(dy + (((int64_t)n * KH + r1) * KW + ow) * ld + cv * V, v1);
#pragma unroll
        for (int k = 0; k < V; ++k) rv[col * C + cv * V + k] = v0[k] * f0 + v1[k] * f1;
    }
    __syncthreads();
    for (int it = threadIdx.x; it < w_n * ncv; it += 256) {
        const int wl = it / ncv, cv = it - wl * ncv, w = w_lo + wl;
        T* o = dx + (((int64_t)n * H + h) * W + w) * xld + cv * V;
        float s[V];
#pragma unroll
        for (int k = 0; k < V; ++k) s[k] = 0.f;
        if (accumulate) VecIO<T, V>::load(o, s);
#pragma unroll
        for (int b = 0; b < NS; ++b) {
            const int4 ev = *reinterpret_cast<const int4*>(ctab + (b * wseg + wl) * 4);
            const float f0 = ev.x >= 0 ? __int_as_float(ev.z) : 0.f, f1 = ev.y >= 0 ? __int_as_float(ev.w) : 0.f;
            const float* p0 = rv + (koff[b] + (ev.x >= 0 ? ev.x : 0)) * C + cv * V;
            const float* p1 = rv + (koff[b] + (ev.y >= 0 ? ev.y : 0)) * C + cv * V;
#pragma unroll
            for (int k = 0; k < V; k += 4) {
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(p0 + k), a1 = *reinterpret_cast<const f32x4*>(p1 + k);
#pragma unroll
                for (int e = 0; e < 4; ++e) s[k + e] += a0[e] * f0 + a1[e] * f1;
            }
        }
        VecIO<T, V>::store(o, s);
    }
}

// Max-pool backward as a scatter through LDS: a block owns one image and one 16-byte channel vector, zeroes an [H*W][V]
// fp32 image of dx in LDS, walks every output pixel of every pool ONCE (16-byte dy load + V indices) adding into the
// arg-max position with LDS atomics, then writes dx out.  The gather form above reads k*k (25 / 81 / 169) index + gradient
// pairs per input pixel.  fp32 LDS atomics: the summation order inside a pixel is not fixed (ATen's CUDA max_pool2d
// backward uses atomicAdd the same way).
struct MaxSrcs {
    const void* dy[3];
    int64_t ld[3];
    const int32_t* idx[3];
    int32_t n;
};
template <typename T, int V>
__global__ __launch_bounds__(256) void maxpool_bwd_scatter_kernel(const MaxSrcs srcs, T* __restrict__ dx, int64_t xld, int HW,
                                                                  int C, int accumulate) {
    extern __shared__ float sacc[];          // [V][HW] planes
    const int ncv = C / V;
    const int n = blockIdx.x / ncv, cv = blockIdx.x % ncv;
    for (int i = threadIdx.x; i < HW * V; i += 256) sacc[i] = 0.f;
    __syncthreads();
    for (int b = 0; b < srcs.n; ++b) {
        const T* dy = (const T*)srcs.dy[b];
        const int32_t* idx = srcs.idx[b];
        for (int p = threadIdx.x; p < HW; p += 256) {
            const int64_t op = (int64_t)n * HW + p;
            float g[V];
            VecIO<T, V>::load(dy + op * srcs.ld[b] + cv * V, g);
            int q[V];
#pragma unroll
            for (int k = 0; k < V; k += 4) {
                const u32x4 qi = *reinterpret_cast<const u32x4*>(idx + op * C + cv * V + k);
                q[k] = (int)qi[0]; q[k + 1] = (int)qi[1]; q[k + 2] = (int)qi[2]; q[k + 3] = (int)qi[3];
            }
#pragma unroll
            for (int k = 0; k < V; ++k)
                __hip_atomic_fetch_add(&sacc[k * HW + q[k]], g[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    for (int p = threadIdx.x; p < HW; p += 256) {
        float s[V];
#pragma unroll
        for (int k = 0; k < V; ++k) s[k] = sacc[k * HW + p];
        T* o = dx + ((int64_t)n * HW + p) * xld + cv * V;
        if (accumulate) {
            float old[V];
            VecIO<T, V>::load(o, old);
#pragma unroll
            for (int k = 0; k < V; ++k) s[k] += old[k];
        }
        VecIO<T, V>::store(o, s);
    }
}

// ---- FFM: out = feat*att + feat                                                          (common.py:240-241) --------
// V channels per thread (16-byte accesses when the tensors allow), 32-bit index arithmetic (the launchers check P * C < 2^31):
// the scalar form with its 64-bit i % C, i / C, p / HW per ELEMENT ran at 1.5 TB/s (18 / 15 us on DeSeNet-s' 8 x 80 x 80 x 128 map)
template <typename T, int V>
__global__ __launch_bounds__(256) void ffm_scale_kernel(const T* __restrict__ f, int64_t fld, const T* __restrict__ att, int64_t ald,
                                                        T* __restrict__ out, int64_t old_, unsigned HW, unsigned P, int C) {
    const unsigned ncv = (unsigned)(C / V), total = P * ncv;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const unsigned p = i / ncv, c = (i - p * ncv) * V, n = p / HW;
        float v[V], a[V];
        VecIO<T, V>::load(f + (int64_t)p * fld + c, v);
        VecIO<T, V>::load(att + (int64_t)n * ald + c, a);
#pragma unroll
        for (int k = 0; k < V; ++k) v[k] = v[k] * a[k] + v[k];
        VecIO<T, V>::store(out + (int64_t)p * old_ + c, v);
    }
}
template <typename T, int V>
__global__ __launch_bounds__(256) void ffm_scale_bwd_feat_kernel(const T* __restrict__ dout, int64_t dld, const T* __restrict__ att,
                                                                 int64_t ald, T* __restrict__ dfeat, int64_t fld, unsigned HW, unsigned P,
                                                                 int C, int accumulate) {
    const unsigned ncv = (unsigned)(C / V), total = P * ncv;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const unsigned p = i / ncv, c = (i - p * ncv) * V, n = p / HW;
        float g[V], a[V];
        VecIO<T, V>::load(dout + (int64_t)p * dld + c, g);
        VecIO<T, V>::load(att + (int64_t)n * ald + c, a);
#pragma unroll
        for (int k = 0; k < V; ++k) g[k] = g[k] * a[k] + g[k];
        T* o = dfeat + (int64_t)p * fld + c;
        if (accumulate) {
            float old[V];
            VecIO<T, V>::load(o, old);
#pragma unroll
            for (int k = 0; k < V; ++k) g[k] += old[k];
        }
        VecIO<T, V>::store(o, g);
    }
}
// ---- split window reductions ----------------------------------------------------------------------------------------------
// out[seg][c] = sum over a 2-D pixel window of  w(pixel) * x[pixel][c] (* y[pixel][c])  for a handful of segments (pyramid
// bins, whole images, the few source pixels of a tiny bilinear input).  A segment is far too much work for one block and
// there are far too few segments to fill 256 CUs, so every segment is split over S blocks by window rows; each block folds
// its share through LDS ((ty, tx) layout: tx = channel, coalesced) and writes one fp32 partial row; a second tiny kernel
// adds the S rows in order (deterministic, no atomics).
enum { WR_AVGPOOL = 0, WR_FFM_ATT = 1, WR_BILINEAR_BWD = 2 };
struct WRGeom {
    int32_t N, H, W, C;       // the big map (pool input / feat / dy of the bilinear)
    int32_t KH, KW;           // the small map (pool output / 1x1 / dx of the bilinear)
    int32_t S;
    float sh, sw;             // bilinear scales
    int64_t xld, yld;
};

template <typename T, int MODE, int V>
__device__ __forceinline__ void window_reduce_body(const T* __restrict__ x, const T* __restrict__ y, float* __restrict__ partial,
                                                   const WRGeom& g, const int bid, float* red) {
    const int s = bid % g.S;
    const int seg = bid / g.S;
    const int kw = seg % g.KW, kh = (seg / g.KW) % g.KH, n = seg / (g.KW * g.KH);
    int h0 = 0, h1 = g.H, w0 = 0, w1 = g.W;
    float inv = 1.f;
    if (MODE == WR_AVGPOOL) {
        h0 = bin_lo(kh, g.H, g.KH); h1 = bin_hi(kh, g.H, g.KH);
        w0 = bin_lo(kw, g.W, g.KW); w1 = bin_hi(kw, g.W, g.KW);
        inv = 1.f / (float)((h1 - h0) * (w1 - w0));
    } else if (MODE == WR_BILINEAR_BWD) {
        if (g.sh > 0.f) {
            h0 = (int)floorf((float)(kh - 1) / g.sh) - 1; h1 = (int)ceilf((float)(kh + 1) / g.sh) + 2;
            h0 = h0 < 0 ? 0 : h0; h1 = h1 > g.H ? g.H : h1;
        }
        if (g.sw > 0.f) {
            w0 = (int)floorf((float)(kw - 1) / g.sw) - 1; w1 = (int)ceilf((float)(kw + 1) / g.sw) + 2;
            w0 = w0 < 0 ? 0 : w0; w1 = w1 > g.W ? g.W : w1;
        }
    }
    const int ww = w1 - w0;
    const int ncv = g.C / V;                      // V channels per thread (16-byte loads when V > 1)
    const int TX = ncv < 256 ? ncv : 256, TY = 256 / TX;
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    float* out = partial + (int64_t)bid * g.C;
    for (int cv0 = 0; cv0 < ncv; cv0 += TX) {
        const int cv = cv0 + tx;
        float acc[V];
#pragma unroll
        for (int k = 0; k < V; ++k) acc[k] = 0.f;
        if (ty < TY && cv < ncv) {
            for (int hh = h0 + s; hh < h1; hh += g.S) {
                float wh = 1.f;
                if (MODE == WR_BILINEAR_BWD) {
                    const Lerp a = lerp_coord(hh, g.sh, g.KH);
                    wh = (a.i0 == kh ? a.l0 : 0.f) + (a.i1 == kh ? a.l1 : 0.f);
                    if (wh == 0.f) continue;
                }
                for (int q = ty; q < ww; q += TY) {
                    const int wq = w0 + q;
                    float wgt = wh;
                    if (MODE == WR_BILINEAR_BWD) {
                        const Lerp b = lerp_coord(wq, g.sw, g.KW);
                        wgt *= (b.i0 == kw ? b.l0 : 0.f) + (b.i1 == kw ? b.l1 : 0.f);
                    }
                    const int64_t p = ((int64_t)n * g.H + hh) * g.W + wq;
                    float v[V];
                    VecIO<T, V>::load(x + p * g.xld + cv * V, v);
                    if (MODE == WR_FFM_ATT) {
                        float u[V];
                        VecIO<T, V>::load(y + p * g.yld + cv * V, u);
#pragma unroll
                        for (int k = 0; k < V; ++k) v[k] *= u[k];
                    }
#pragma unroll
                    for (int k = 0; k < V; ++k) acc[k] += wgt * v[k];
                }
            }
        }
        __syncthreads();
        if (ty < TY) {
#pragma unroll
            for (int k = 0; k < V; ++k) red[(ty * TX + tx) * V + k] = acc[k];
        }
        __syncthreads();
        const int cnum = ((ncv - cv0 < TX) ? (ncv - cv0) : TX) * V;
        for (int c = threadIdx.x; c < cnum; c += 256) {
            float tot = 0.f;
            for (int t = 0; t < TY; ++t) tot += red[t * TX * V + c];
            out[cv0 * V + c] = tot * inv;
        }
    }
}

template <typename T, int MODE, int V>
__global__ __launch_bounds__(256) void window_reduce_kernel(const T* __restrict__ x, const T* __restrict__ y,
                                                            float* __restrict__ partial, const WRGeom g) {
    __shared__ float red[256 * V];
    window_reduce_body<T, MODE, V>(x, y, partial, g, (int)blockIdx.x, red);
}

// Up to four window reductions in ONE launch (PyramidPooling: the four adaptive pools of one map, and the four small-source
// bilinear backward passes of its branches): job j owns blocks [blk0[j], blk0[j+1]).
constexpr int WR_MAXJOBS = 4;
struct WRMulti {
    WRGeom g[WR_MAXJOBS];
    const void* x[WR_MAXJOBS];
    float* partial[WR_MAXJOBS];
    void* out[WR_MAXJOBS];
    int64_t old_[WR_MAXJOBS];
    int32_t blk0[WR_MAXJOBS + 1];      // reduce kernel
    int32_t fin0[WR_MAXJOBS + 1];      // finalize kernel
    int32_t nseg[WR_MAXJOBS];
    int32_t n, accumulate;
};
template <typename T, int MODE, int V>
__global__ __launch_bounds__(256) void window_reduce_multi_kernel(const WRMulti m) {
    __shared__ float red[256 * V];
    int j = 0;
    while (j + 1 < m.n && (int)blockIdx.x >= m.blk0[j + 1]) ++j;
    window_reduce_body<T, MODE, V>((const T*)m.x[j], (const T*)nullptr, m.partial[j], m.g[j], (int)blockIdx.x - m.blk0[j], red);
}

// out[seg][c] (+)= sum_k partial[seg*S + k][c].  A block owns one segment and 32 channels; its 8 thread groups each add
// every 8th partial row (independent loads in flight), LDS folds the 8 sums in a fixed order.  (One thread per output
// walking all S rows serially took 13 us for a few KB.)
template <typename T>
__device__ __forceinline__ void window_finalize_body(const float* __restrict__ partial, int S, int C, T* __restrict__ out,
                                                     int64_t old_, int accumulate, const int bid, float (*red)[33]) {
    const int cgroups = (C + 31) / 32;
    const int64_t seg = bid / cgroups;
    const int c = (bid % cgroups) * 32 + (threadIdx.x & 31), ty = threadIdx.x >> 5;
    float v0 = 0.f, v1 = 0.f;
    if (c < C) {
        const float* src = partial + seg * S * C + c;
        int k = ty;
        for (; k + 8 < S; k += 16) { v0 += src[(int64_t)k * C]; v1 += src[(int64_t)(k + 8) * C]; }
        if (k < S) v0 += src[(int64_t)k * C];
    }
    red[ty][threadIdx.x & 31] = v0 + v1;
    __syncthreads();
    if (ty == 0 && c < C) {
        float v = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t) v += red[t][threadIdx.x];
        T* o = out + seg * old_ + c;
        if (accumulate) v += to_f32<T>(*o);
        *o = from_f32<T>(v);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void window_finalize_kernel(const float* __restrict__ partial, int S, int64_t nseg, int C,
                                                              T* __restrict__ out, int64_t old_, int accumulate) {
    __shared__ float red[8][33];
    window_finalize_body<T>(partial, S, C, out, old_, accumulate, (int)blockIdx.x, red);
}

template <typename T>
__global__ __launch_bounds__(256) void window_finalize_multi_kernel(const WRMulti m) {
    __shared__ float red[8][33];
    int j = 0;
    while (j + 1 < m.n && (int)blockIdx.x >= m.fin0[j + 1]) ++j;
    window_finalize_body<T>(m.partial[j], m.g[j].S, m.g[j].C, (T*)m.out[j], m.old_[j], m.accumulate, (int)blockIdx.x - m.fin0[j],
                            red);
}

inline int wr_splits(int64_t nseg, int rows) {
    int64_t s = (768 + nseg - 1) / nseg;
    if (s > rows) s = rows;
    return (int)(s < 1 ? 1 : s);
}

inline bool same_nhwc(const dsn_tensor* a, const dsn_tensor* b) {
    return a->n == b->n && a->h == b->h && a->w == b->w && a->c == b->c && a->dtype == b->dtype;
}

}  // namespace

namespace {
inline bool vec16(const dsn_tensor* t) {
    const int vw = t->dtype == DSN_F32 ? 4 : 8;
    return t->c % vw == 0 && t->ldc % vw == 0 && ((uintptr_t)t->ptr % 16) == 0;
}
}  // namespace

namespace {
// ---- letterbox (mixed_datasets.py:722-752) + `img.transpose(2, 0, 1)[::-1]` (:576) in ONE launch ---------------------------
// src: uint8 HWC (BGR as cv2 loads it), h0 x w0.  dst: uint8 H x W, either HWC in the source's channel order (what
// letterbox() returns) or CHW with the channel order reversed (what the loader hands to the network).  Inside the window
// [top, top + nh) x [left, left + nw) the pixel is the INTER_LINEAR resize of src to nh x nw, outside it the border colour.
// The resize restates OpenCV's 8-bit INTER_LINEAR (cv2.resize is a third-party dependency that is not in the image): source
// coordinate f = (d + 0.5) * (in / out) - 0.5, taps (floor f, +1) clamped to the image, 11-bit fixed-point weights
// cvRound(w * 2048) (round half to even), horizontal pass exact in int32, vertical pass
// (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2.
__device__ __forceinline__ void lb_tap(int d, int in, double scale, int& i0, int& i1, int& a0, int& a1) {
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= s;
    if (s < 0) { s = 0; f = 0.f; }
    if (s >= in - 1) { s = in - 1; f = 0.f; }
    i0 = s;
    i1 = s + 1 < in ? s + 1 : in - 1;
    a0 = (int)rintf((1.f - f) * 2048.f);
    a1 = (int)rintf(f * 2048.f);
}
__global__ __launch_bounds__(256) void letterbox_u8_kernel(const uint8_t* __restrict__ src, int h0, int w0,
                                                           uint8_t* __restrict__ dst, int H, int W, int nh, int nw, int top,
                                                           int left, int pad0, int pad1, int pad2, int chw_reversed,
                                                           double sy, double sx) {
    const int64_t total = (int64_t)H * W;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int x = (int)(i % W), y = (int)(i / W);
        int v[3] = {pad0, pad1, pad2};
        const int ry = y - top, rx = x - left;
        if (ry >= 0 && ry < nh && rx >= 0 && rx < nw) {
            if (nh == h0 && nw == w0) {
#pragma unroll
                for (int c = 0; c < 3; ++c) v[c] = src[((int64_t)ry * w0 + rx) * 3 + c];
            } else {
                int y0, y1, b0, b1, x0, x1, a0, a1;
                lb_tap(ry, h0, sy, y0, y1, b0, b1);
                lb_tap(rx, w0, sx, x0, x1, a0, a1);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int r0 = src[((int64_t)y0 * w0 + x0) * 3 + c] * a0 + src[((int64_t)y0 * w0 + x1) * 3 + c] * a1;
                    const int r1 = src[((int64_t)y1 * w0 + x0) * 3 + c] * a0 + src[((int64_t)y1 * w0 + x1) * 3 + c] * a1;
                    const int o = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
                    v[c] = o < 0 ? 0 : (o > 255 ? 255 : o);
                }
            }
        }
        if (chw_reversed) {
#pragma unroll
            for (int c = 0; c < 3; ++c) dst[(int64_t)(2 - c) * total + i] = (uint8_t)v[c];
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) dst[i * 3 + c] = (uint8_t)v[c];
        }
    }
}

template <typename IN>
int focus_impl(const IN* x, int32_t n, int32_t c, int32_t h, int32_t w, const dsn_tensor* y, void* stream) {
    DSN_CHECK_ARG(x && tensor_ok(y) && n > 0 && c > 0, "focus_s2d: invalid arguments");
    DSN_CHECK_ARG(h % 2 == 0 && w % 2 == 0, "focus_s2d: H and W must be even (got %dx%d)", h, w);
    DSN_CHECK_ARG(y->n == n && y->h == h / 2 && y->w == w / 2 && y->c >= 4 * c, "focus_s2d: output shape mismatch");
    if (c <= 4 && w % 2 == 0 && ((uintptr_t)x % (2 * sizeof(IN))) == 0 && vec16(y) && y->c <= 16) {
        const int64_t px = npix(y);
        if (y->dtype == DSN_F32)
            hipLaunchKernelGGL((focus_s2d_px_kernel<float, 4, 16, IN>), dim3(ew_grid(px)), dim3(256), 0, (hipStream_t)stream,
                               x, (float*)y->ptr, n, c, h, w, y->c, y->ldc);
        else
            hipLaunchKernelGGL((focus_s2d_px_kernel<bf16_t, 8, 16, IN>), dim3(ew_grid(px)), dim3(256), 0, (hipStream_t)stream,
                               x, (bf16_t*)y->ptr, n, c, h, w, y->c, y->ldc);
        DSN_LAUNCH_CHECK("focus_s2d");
        return DSN_OK;
    }
    const int64_t total = npix(y) * y->c;
    DSN_DISPATCH_DTYPE(y->dtype, T,
                       hipLaunchKernelGGL((focus_s2d_kernel<T, IN>), dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x,
                                          (T*)y->ptr, n, c, h, w, y->c, y->ldc));
    DSN_LAUNCH_CHECK("focus_s2d");
    return DSN_OK;
}
}  // namespace

extern "C" int dsn_focus_s2d(const float* x, int32_t n, int32_t c, int32_t h, int32_t w, const dsn_tensor* y,
                             void* stream) {
    return focus_impl<float>(x, n, c, h, w, y, stream);
}

extern "C" int dsn_letterbox_u8(const uint8_t* src_hwc, int32_t h0, int32_t w0, uint8_t* dst, int32_t h, int32_t w,
                                int32_t new_h, int32_t new_w, int32_t top, int32_t left, int32_t pad0, int32_t pad1,
                                int32_t pad2, int32_t chw_reversed, void* stream) {
    DSN_CHECK_ARG(src_hwc && dst && h0 > 0 && w0 > 0 && h > 0 && w > 0 && new_h > 0 && new_w > 0, "letterbox_u8: bad sizes");
    DSN_CHECK_ARG(top >= 0 && left >= 0 && top + new_h <= h && left + new_w <= w, "letterbox_u8: window outside the output");
    hipLaunchKernelGGL(letterbox_u8_kernel, dim3(ew_grid((int64_t)h * w)), dim3(256), 0, (hipStream_t)stream, src_hwc, h0, w0,
                       dst, h, w, new_h, new_w, top, left, pad0, pad1, pad2, chw_reversed, (double)h0 / new_h, (double)w0 / new_w);
    DSN_LAUNCH_CHECK("letterbox_u8");
    return DSN_OK;
}

// the same from the loader's uint8 NCHW batch, with its `.float() / 255.0` folded in (4x fewer input bytes)
extern "C" int dsn_focus_s2d_u8(const uint8_t* x, int32_t n, int32_t c, int32_t h, int32_t w, const dsn_tensor* y,
                                void* stream) {
    return focus_impl<uint8_t>(x, n, c, h, w, y, stream);
}

extern "C" int dsn_maxpool_s1_multi(const dsn_tensor* x, const dsn_tensor* ys, void* const* idxs, const int32_t* ks,
                                    int32_t n_out, void* stream) {
    DSN_CHECK_ARG(tensor_ok(x) && ys && ks && n_out >= 1 && n_out <= 3, "maxpool_s1_multi: invalid arguments");
    DSN_CHECK_ARG((int64_t)x->h * x->w < (1ll << 31), "maxpool_s1_multi: map too large");
    bool vec = vec16(x);
    for (int i = 0; i < n_out; ++i) {
        DSN_CHECK_ARG(tensor_ok(&ys[i]) && same_nhwc(x, &ys[i]) && ks[i] >= 1 && (ks[i] & 1),
                      "maxpool_s1_multi: output %d does not match x", i);
        vec = vec && vec16(&ys[i]) && (!idxs || !idxs[i] || (uintptr_t)idxs[i] % 16 == 0);
    }
    hipStream_t st = (hipStream_t)stream;
    const int HW = x->h * x->w;
    const int V = x->dtype == DSN_F32 ? 4 : 8;
    const size_t lds = (size_t)HW * V * 3 * sizeof(float);
    // cascade form: outputs k0, 2 k0 - 1, 3 k0 - 2 in this order (SPP's 5 / 9 / 13)
    // (whole map x channel vector x two (value, index) planes in LDS: 16-byte vectors up to 32 x 32 maps, 8-byte bf16 vectors up
    //  to 48 x 48 -- config 5's SPP sits at 40 x 40, where the one-thread-per-output kernel below took 3 x 112 us)
    const size_t l_full = (size_t)HW * V * 4 * sizeof(float), l_half = l_full / 2;
    const bool narrow = x->dtype == DSN_BF16 && l_full > 64 * 1024 && l_half <= 150 * 1024 && x->c % 4 == 0;
    bool cascade = vec && (l_full <= 64 * 1024 || narrow);
    for (int i = 0; i < n_out; ++i) cascade = cascade && ks[i] == (i + 1) * (ks[0] - 1) + 1;
    static const bool no_cascade = getenv("DSN_MAXPOOL_CASCADE") && atoi(getenv("DSN_MAXPOOL_CASCADE")) == 0;
    if (cascade && !no_cascade) {
        MaxFwd out{};
        out.n = n_out;
        for (int i = 0; i < n_out; ++i) {
            out.y[i] = ys[i].ptr; out.ld[i] = ys[i].ldc; out.idx[i] = idxs ? (int32_t*)idxs[i] : nullptr; out.k[i] = ks[i];
        }
        if (x->dtype == DSN_F32) {
            hipLaunchKernelGGL((maxpool_cascade_kernel<float, 4, 512>), dim3(x->n * (x->c / 4)), dim3(512), l_full, st,
                               (const float*)x->ptr, x->ldc, out, x->h, x->w, x->c, ks[0] / 2);
        } else if (narrow) {
            DSN_LDS_ATTR((maxpool_cascade_kernel<bf16_t, 4, 1024>), 150 * 1024);
            hipLaunchKernelGGL((maxpool_cascade_kernel<bf16_t, 4, 1024>), dim3(x->n * (x->c / 4)), dim3(1024), l_half, st,
                               (const bf16_t*)x->ptr, x->ldc, out, x->h, x->w, x->c, ks[0] / 2);
        } else {
            hipLaunchKernelGGL((maxpool_cascade_kernel<bf16_t, 8, 1024>), dim3(x->n * (x->c / 8)), dim3(1024), l_full, st,
                               (const bf16_t*)x->ptr, x->ldc, out, x->h, x->w, x->c, ks[0] / 2);
        }
    } else if (vec && lds <= 60 * 1024) {
        MaxFwd out{};
        out.n = n_out;
        for (int i = 0; i < n_out; ++i) {
            out.y[i] = ys[i].ptr; out.ld[i] = ys[i].ldc; out.idx[i] = idxs ? (int32_t*)idxs[i] : nullptr; out.k[i] = ks[i];
        }
        const dim3 grid(x->n * (x->c / V) * n_out);
        if (x->dtype == DSN_F32)
            hipLaunchKernelGGL((maxpool_multi_kernel<float, 4>), grid, dim3(256), lds, st, (const float*)x->ptr, x->ldc, out,
                               x->h, x->w, x->c);
        else
            hipLaunchKernelGGL((maxpool_multi_kernel<bf16_t, 8>), grid, dim3(256), lds, st, (const bf16_t*)x->ptr, x->ldc, out,
                               x->h, x->w, x->c);
    } else {
        const int64_t total = npix(x) * x->c;
        for (int i = 0; i < n_out; ++i)
            DSN_DISPATCH_DTYPE(x->dtype, T,
                               hipLaunchKernelGGL(maxpool_kernel<T>, dim3(ew_grid(total)), dim3(256), 0, st, (const T*)x->ptr,
                                                  x->ldc, (T*)ys[i].ptr, ys[i].ldc, idxs ? (int32_t*)idxs[i] : nullptr, x->n,
                                                  x->h, x->w, x->c, ks[i]));
    }
    DSN_LAUNCH_CHECK("maxpool_s1");
    return DSN_OK;
}

extern "C" int dsn_maxpool_s1(const dsn_tensor* x, const dsn_tensor* y, int32_t* idx, int32_t k, void* stream) {
    DSN_CHECK_ARG(y, "maxpool_s1: invalid arguments");
    void* idxs[1] = {idx};
    return dsn_maxpool_s1_multi(x, y, idxs, &k, 1, stream);
}

extern "C" int dsn_maxpool_s1_bwd_multi(const dsn_tensor* dys, const void* const* idxs, const int32_t* ks, int32_t n_src,
                                        const dsn_tensor* dx, int32_t accumulate, void* stream) {
    DSN_CHECK_ARG(dys && idxs && ks && n_src >= 1 && n_src <= 3 && tensor_ok(dx), "maxpool_s1_bwd_multi: invalid arguments");
    bool vec = vec16(dx);
    for (int i = 0; i < n_src; ++i) {
        DSN_CHECK_ARG(tensor_ok(&dys[i]) && same_nhwc(&dys[i], dx) && idxs[i] && ks[i] >= 1 && (ks[i] & 1),
                      "maxpool_s1_bwd_multi: source %d does not match dx", i);
        vec = vec && vec16(&dys[i]) && ((uintptr_t)idxs[i] % 16 == 0);
    }
    hipStream_t st = (hipStream_t)stream;
    const int HW = dx->h * dx->w;
    const int V = dx->dtype == DSN_F32 ? 4 : 8;
    const size_t lds = (size_t)HW * V * sizeof(float);
    if (vec && lds <= 60 * 1024) {
        MaxSrcs srcs{};
        srcs.n = n_src;
        for (int i = 0; i < n_src; ++i) { srcs.dy[i] = dys[i].ptr; srcs.ld[i] = dys[i].ldc; srcs.idx[i] = (const int32_t*)idxs[i]; }
        const dim3 grid(dx->n * (dx->c / V));
        if (dx->dtype == DSN_F32)
            hipLaunchKernelGGL((maxpool_bwd_scatter_kernel<float, 4>), grid, dim3(256), lds, st, srcs, (float*)dx->ptr, dx->ldc,
                               HW, dx->c, accumulate);
        else
            hipLaunchKernelGGL((maxpool_bwd_scatter_kernel<bf16_t, 8>), grid, dim3(256), lds, st, srcs, (bf16_t*)dx->ptr,
                               dx->ldc, HW, dx->c, accumulate);
    } else {
        const int64_t total = npix(dx) * dx->c;
        for (int i = 0; i < n_src; ++i)
            DSN_DISPATCH_DTYPE(dx->dtype, T,
                               hipLaunchKernelGGL(maxpool_bwd_kernel<T>, dim3(ew_grid(total)), dim3(256), 0, st,
                                                  (const T*)dys[i].ptr, dys[i].ldc, (const int32_t*)idxs[i], (T*)dx->ptr,
                                                  dx->ldc, dx->n, dx->h, dx->w, dx->c, ks[i], (accumulate || i > 0) ? 1 : 0));
    }
    DSN_LAUNCH_CHECK("maxpool_s1_bwd");
    return DSN_OK;
}

extern "C" int dsn_maxpool_s1_bwd(const dsn_tensor* dy, const int32_t* idx, const dsn_tensor* dx, int32_t k,
                                  int32_t accumulate, void* stream) {
    DSN_CHECK_ARG(dy && idx, "maxpool_s1_bwd: invalid arguments");
    const void* idxs[1] = {idx};
    return dsn_maxpool_s1_bwd_multi(dy, idxs, &k, 1, dx, accumulate, stream);
}

extern "C" int dsn_upsample_nearest2x(const dsn_tensor* x, const dsn_tensor* y, void* stream) {
    DSN_CHECK_ARG(tensor_ok(x) && tensor_ok(y) && x->dtype == y->dtype && y->n == x->n && y->h == 2 * x->h &&
                      y->w == 2 * x->w && y->c == x->c,
                  "upsample_nearest2x: shape mismatch");
    const int64_t total = npix(y) * y->c;
    if (vec16(x) && vec16(y)) {
        if (x->dtype == DSN_F32)
            hipLaunchKernelGGL((up2_vec_kernel<float, 4>), dim3(ew_grid(total / 4)), dim3(256), 0, (hipStream_t)stream,
                               (const float*)x->ptr, x->ldc, (float*)y->ptr, y->ldc, x->n, x->h, x->w, x->c);
        else
            hipLaunchKernelGGL((up2_vec_kernel<bf16_t, 8>), dim3(ew_grid(total / 8)), dim3(256), 0, (hipStream_t)stream,
                               (const bf16_t*)x->ptr, x->ldc, (bf16_t*)y->ptr, y->ldc, x->n, x->h, x->w, x->c);
        DSN_LAUNCH_CHECK("upsample_nearest2x");
        return DSN_OK;
    }
    DSN_DISPATCH_DTYPE(x->dtype, T,
                       hipLaunchKernelGGL(up2_kernel<T>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream,
                                          (const T*)x->ptr, x->ldc, (T*)y->ptr, y->ldc, x->n, x->h, x->w, x->c));
    DSN_LAUNCH_CHECK("upsample_nearest2x");
    return DSN_OK;
}

extern "C" int dsn_upsample_nearest2x_bwd(const dsn_tensor* dy, const dsn_tensor* dx, int32_t accumulate, void* stream) {
    DSN_CHECK_ARG(tensor_ok(dx) && tensor_ok(dy) && dx->dtype == dy->dtype && dy->n == dx->n && dy->h == 2 * dx->h &&
                      dy->w == 2 * dx->w && dy->c == dx->c,
                  "upsample_nearest2x_bwd: shape mismatch");
    const int64_t total = npix(dx) * dx->c;
    if (vec16(dx) && vec16(dy)) {
        if (dx->dtype == DSN_F32)
            hipLaunchKernelGGL((up2_bwd_vec_kernel<float, 4>), dim3(ew_grid(total / 4)), dim3(256), 0, (hipStream_t)stream,
                               (const float*)dy->ptr, dy->ldc, (float*)dx->ptr, dx->ldc, dx->n, dx->h, dx->w, dx->c, accumulate);
        else
            hipLaunchKernelGGL((up2_bwd_vec_kernel<bf16_t, 8>), dim3(ew_grid(total / 8)), dim3(256), 0, (hipStream_t)stream,
                               (const bf16_t*)dy->ptr, dy->ldc, (bf16_t*)dx->ptr, dx->ldc, dx->n, dx->h, dx->w, dx->c,
                               accumulate);
        DSN_LAUNCH_CHECK("upsample_nearest2x_bwd");
        return DSN_OK;
    }
    DSN_DISPATCH_DTYPE(dx->dtype, T,
                       hipLaunchKernelGGL(up2_bwd_kernel<T>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream,
                                          (const T*)dy->ptr, dy->ldc, (T*)dx->ptr, dx->ldc, dx->n, dx->h, dx->w, dx->c,
                                          accumulate));
    DSN_LAUNCH_CHECK("upsample_nearest2x_bwd");
    return DSN_OK;
}

extern "C" int dsn_bilinear_ac(const dsn_tensor* x, const dsn_tensor* y, int32_t out_nchw, void* stream) {
    DSN_CHECK_ARG(tensor_ok(x) && y && y->ptr && y->n == x->n && y->c == x->c && y->h > 0 && y->w > 0,
                  "bilinear_ac: invalid arguments");
    DSN_CHECK_ARG(out_nchw ? (y->dtype == DSN_F32) : (y->dtype == x->dtype && y->ldc >= y->c),
                  "bilinear_ac: NCHW output must be fp32; NHWC output must match the input dtype");
    const float sh = ac_scale(x->h, y->h), sw = ac_scale(x->w, y->w);
    const int64_t total = (int64_t)y->n * y->h * y->w * y->c;
    hipStream_t st = (hipStream_t)stream;
    if (!out_nchw && vec16(x) && vec16(y)) {
        const int V = x->dtype == DSN_F32 ? 4 : 8;
        const int64_t tv = total / V;
        if (x->dtype == DSN_F32)
            hipLaunchKernelGGL((bilinear_vec_kernel<float, 4>), dim3(ew_grid(tv)), dim3(256), 0, st, (const float*)x->ptr, x->ldc,
                               (float*)y->ptr, y->ldc, x->n, x->h, x->w, y->h, y->w, x->c, sh, sw);
        else
            hipLaunchKernelGGL((bilinear_vec_kernel<bf16_t, 8>), dim3(ew_grid(tv)), dim3(256), 0, st, (const bf16_t*)x->ptr, x->ldc,
                               (bf16_t*)y->ptr, y->ldc, x->n, x->h, x->w, y->h, y->w, x->c, sh, sw);
        DSN_LAUNCH_CHECK("bilinear_ac");
        return DSN_OK;
    }
    DSN_DISPATCH_DTYPE(x->dtype, T, {
        if (out_nchw)
            hipLaunchKernelGGL((bilinear_kernel<T, true>), dim3(ew_grid(total)), dim3(256), 0, st, (const T*)x->ptr,
                               x->ldc, y->ptr, y->ldc, x->n, x->h, x->w, y->h, y->w, x->c, sh, sw);
        else
            hipLaunchKernelGGL((bilinear_kernel<T, false>), dim3(ew_grid(total)), dim3(256), 0, st, (const T*)x->ptr,
                               x->ldc, y->ptr, y->ldc, x->n, x->h, x->w, y->h, y->w, x->c, sh, sw);
    });
    DSN_LAUNCH_CHECK("bilinear_ac");
    return DSN_OK;
}

extern "C" int dsn_bilinear_ac_bwd(const dsn_tensor* dy, int32_t dy_nchw, const dsn_tensor* dx, int32_t accumulate,
                                   void* workspace, int64_t workspace_bytes, void* stream) {
    DSN_CHECK_ARG(tensor_ok(dx) && dy && dy->ptr && dy->n == dx->n && dy->c == dx->c && dy->h > 0 && dy->w > 0,
                  "bilinear_ac_bwd: invalid arguments");
    DSN_CHECK_ARG(dy_nchw ? (dy->dtype == DSN_F32) : (dy->dtype == dx->dtype),
                  "bilinear_ac_bwd: NCHW dy must be fp32; NHWC dy must match dx dtype");
    const float sh = ac_scale(dx->h, dy->h), sw = ac_scale(dx->w, dy->w);
    const int64_t total = npix(dx) * dx->c;
    hipStream_t st = (hipStream_t)stream;
    // tiny source maps (PyramidPooling: 1..6 pixels a side feeding 80x80): every source pixel gathers a large window ->
    // split reduction instead of one thread per element
    const int64_t nseg = npix(dx);
    if (!dy_nchw && (int64_t)dx->h * dx->w <= 64 && workspace) {
        WRGeom g{dy->n, dy->h, dy->w, dy->c, dx->h, dx->w, wr_splits(nseg, dy->h), sh, sw, dy->ldc, 0};
        if (workspace_bytes < nseg * g.S * dx->c * (int64_t)sizeof(float))
            DSN_FAIL(DSN_EWORKSPACE, "bilinear_ac_bwd: workspace too small");
        DSN_DISPATCH_DTYPE(dx->dtype, T, {
            if (vec16(dy))
                hipLaunchKernelGGL((window_reduce_kernel<T, WR_BILINEAR_BWD, VW<T>::N>), dim3((unsigned)(nseg * g.S)), dim3(256), 0,
                                   st, (const T*)dy->ptr, (const T*)nullptr, (float*)workspace, g);
            else
                hipLaunchKernelGGL((window_reduce_kernel<T, WR_BILINEAR_BWD, 1>), dim3((unsigned)(nseg * g.S)), dim3(256), 0, st,
                                   (const T*)dy->ptr, (const T*)nullptr, (float*)workspace, g);
            hipLaunchKernelGGL(window_finalize_kernel<T>, dim3((unsigned)(nseg * ((dx->c + 31) / 32))), dim3(256), 0, st,
                               (const float*)workspace, g.S, nseg, dx->c, (T*)dx->ptr, dx->ldc, accumulate);
        });
        DSN_LAUNCH_CHECK("bilinear_ac_bwd (split)");
        return DSN_OK;
    }
    if (!dy_nchw && vec16(dx) && vec16(dy) && dy->h >= dx->h && dy->w >= dx->w) {
        // separable rows form: one block per dx row and per group of channel vectors (<= 48 KB of fp32 row in LDS, enough blocks
        // to cover the chip)
        const int V = dx->dtype == DSN_F32 ? 4 : 8;
        const int ncv = dx->c / V;
        int cvb = ncv;
        while (cvb > 1 && ((int64_t)dy->w * cvb * V * 4 > 48 * 1024 || (int64_t)dx->n * dx->h * ((ncv + cvb - 1) / cvb) < 512)) cvb = (cvb + 1) / 2;
        if ((int64_t)dy->w * cvb * V * 4 <= 48 * 1024) {
            const int nsplit = (ncv + cvb - 1) / cvb;
            const size_t lds = (size_t)dy->w * cvb * V * 4;
            const dim3 grid((unsigned)(dx->n * dx->h * nsplit));
            if (dx->dtype == DSN_F32)
                hipLaunchKernelGGL((bilinear_bwd_rows_kernel<float, 4>), grid, dim3(256), lds, st, (const float*)dy->ptr, dy->ldc,
                                   (float*)dx->ptr, dx->ldc, dx->h, dx->w, dy->h, dy->w, dx->c, cvb, sh, sw, accumulate);
            else
                hipLaunchKernelGGL((bilinear_bwd_rows_kernel<bf16_t, 8>), grid, dim3(256), lds, st, (const bf16_t*)dy->ptr, dy->ldc,
                                   (bf16_t*)dx->ptr, dx->ldc, dx->h, dx->w, dy->h, dy->w, dx->c, cvb, sh, sw, accumulate);
            DSN_LAUNCH_CHECK("bilinear_ac_bwd (separable rows)");
            return DSN_OK;
        }
    }
    if (!dy_nchw && vec16(dx) && vec16(dy)) {
        const int V = dx->dtype == DSN_F32 ? 4 : 8;
        if (dx->dtype == DSN_F32)
            hipLaunchKernelGGL((bilinear_bwd_vec_kernel<float, 4>), dim3(ew_grid(total / V)), dim3(256), 0, st,
                               (const float*)dy->ptr, dy->ldc, (float*)dx->ptr, dx->ldc, dx->n, dx->h, dx->w, dy->h, dy->w, dx->c,
                               sh, sw, accumulate);
        else
            hipLaunchKernelGGL((bilinear_bwd_vec_kernel<bf16_t, 8>), dim3(ew_grid(total / V)), dim3(256), 0, st,
                               (const bf16_t*)dy->ptr, dy->ldc, (bf16_t*)dx->ptr, dx->ldc, dx->n, dx->h, dx->w, dy->h, dy->w,
                               dx->c, sh, sw, accumulate);
        DSN_LAUNCH_CHECK("bilinear_ac_bwd");
        return DSN_OK;
    }
    if (dy_nchw && (int64_t)dy->w * 4 <= 48 * 1024 && (int64_t)dx->n * dx->c * dx->h < (1ll << 30)) {
        DSN_DISPATCH_DTYPE(dx->dtype, T,
                           hipLaunchKernelGGL(bilinear_bwd_nchw_rows_kernel<T>, dim3((unsigned)(dx->n * dx->c * dx->h)), dim3(256),
                                              (size_t)dy->w * 4, st, (const float*)dy->ptr, (T*)dx->ptr, dx->ldc, dx->n, dx->h,
                                              dx->w, dy->h, dy->w, dx->c, sh, sw, accumulate));
        DSN_LAUNCH_CHECK("bilinear_ac_bwd (rows)");
        return DSN_OK;
    }
    DSN_DISPATCH_DTYPE(dx->dtype, T, {
        if (dy_nchw)
            hipLaunchKernelGGL((bilinear_bwd_kernel<T, true>), dim3(ew_grid(total)), dim3(256), 0, st, dy->ptr, dy->ldc,
                               (T*)dx->ptr, dx->ldc, dx->n, dx->h, dx->w, dy->h, dy->w, dx->c, sh, sw, accumulate);
        else
            hipLaunchKernelGGL((bilinear_bwd_kernel<T, false>), dim3(ew_grid(total)), dim3(256), 0, st, dy->ptr,
                               dy->ldc, (T*)dx->ptr, dx->ldc, dx->n, dx->h, dx->w, dy->h, dy->w, dx->c, sh, sw,
                               accumulate);
    });
    DSN_LAUNCH_CHECK("bilinear_ac_bwd");
    return DSN_OK;
}

// ---- PyramidPooling's four branches in one launch per stage (common.py:597-615) ------------------------------------------
namespace {
template <int MODE>
int window_multi(const dsn_tensor* const* big, const dsn_tensor* const* small_, int n, int accumulate, void* workspace,
                 int64_t workspace_bytes, hipStream_t st, const char* what) {
    WRMulti m{};
    m.n = n;
    m.accumulate = accumulate;
    int64_t off = 0;
    int blk = 0, fin = 0;
    for (int j = 0; j < n; ++j) {
        const dsn_tensor* B = big[j];
        const dsn_tensor* S = small_[j];
        const int64_t nseg = (int64_t)S->n * S->h * S->w;
        const int splits = wr_splits(nseg, B->h);
        float sh = 0.f, sw = 0.f;
        if (MODE == WR_BILINEAR_BWD) { sh = ac_scale(S->h, B->h); sw = ac_scale(S->w, B->w); }
        m.g[j] = WRGeom{B->n, B->h, B->w, B->c, S->h, S->w, splits, sh, sw, B->ldc, 0};
        m.x[j] = B->ptr;
        m.partial[j] = (float*)((char*)workspace + off);
        m.out[j] = S->ptr;
        m.old_[j] = S->ldc;
        m.nseg[j] = (int32_t)nseg;
        m.blk0[j] = blk;
        m.fin0[j] = fin;
        blk += (int)(nseg * splits);
        fin += (int)(nseg * ((S->c + 31) / 32));
        off += nseg * splits * B->c * (int64_t)sizeof(float);
    }
    m.blk0[n] = blk;
    m.fin0[n] = fin;
    if (!workspace || workspace_bytes < off) DSN_FAIL(DSN_EWORKSPACE, "%s: workspace too small", what);
    DSN_DISPATCH_DTYPE(big[0]->dtype, T, {
        hipLaunchKernelGGL((window_reduce_multi_kernel<T, MODE, VW<T>::N>), dim3((unsigned)blk), dim3(256), 0, st, m);
        hipLaunchKernelGGL(window_finalize_multi_kernel<T>, dim3((unsigned)fin), dim3(256), 0, st, m);
    });
    return DSN_OK;
}
}  // namespace

// ys[j] = AdaptiveAvgPool2d(ys[j].h)(x) for up to four output sizes: x is read by ONE reduction launch, one finalize launch
extern "C" int dsn_adaptive_avgpool_multi(const dsn_tensor* x, const dsn_tensor* ys, int32_t n_out, void* workspace,
                                          int64_t workspace_bytes, void* stream) {
    DSN_CHECK_ARG(tensor_ok(x) && ys && n_out >= 1 && n_out <= WR_MAXJOBS && vec16(x), "adaptive_avgpool_multi: invalid arguments");
    const dsn_tensor* big[WR_MAXJOBS];
    const dsn_tensor* sm[WR_MAXJOBS];
    for (int j = 0; j < n_out; ++j) {
        DSN_CHECK_ARG(tensor_ok(&ys[j]) && ys[j].dtype == x->dtype && ys[j].n == x->n && ys[j].c == x->c,
                      "adaptive_avgpool_multi: output %d does not match the input", j);
        big[j] = x;
        sm[j] = &ys[j];
    }
    int rc = window_multi<WR_AVGPOOL>(big, sm, n_out, 0, workspace, workspace_bytes, (hipStream_t)stream, "adaptive_avgpool_multi");
    if (rc) return rc;
    DSN_LAUNCH_CHECK("adaptive_avgpool_multi");
    return DSN_OK;
}

// dxs[j] (+)= bilinear(align_corners=True) backward of dys[j], small sources (h*w <= 64 each): one reduction + one finalize launch
extern "C" int dsn_bilinear_ac_bwd_multi(const dsn_tensor* dys, const dsn_tensor* dxs, int32_t n, int32_t accumulate,
                                         void* workspace, int64_t workspace_bytes, void* stream) {
    DSN_CHECK_ARG(dys && dxs && n >= 1 && n <= WR_MAXJOBS, "bilinear_ac_bwd_multi: invalid arguments");
    const dsn_tensor* big[WR_MAXJOBS];
    const dsn_tensor* sm[WR_MAXJOBS];
    for (int j = 0; j < n; ++j) {
        DSN_CHECK_ARG(tensor_ok(&dys[j]) && tensor_ok(&dxs[j]) && dys[j].dtype == dxs[0].dtype && dxs[j].dtype == dxs[0].dtype &&
                          dys[j].n == dxs[j].n && dys[j].c == dxs[j].c && vec16(&dys[j]) && dxs[j].h * dxs[j].w <= 64,
                      "bilinear_ac_bwd_multi: job %d needs matching NHWC tensors, 16-byte channel vectors and a source of <= 64 pixels", j);
        big[j] = &dys[j];
        sm[j] = &dxs[j];
    }
    int rc = window_multi<WR_BILINEAR_BWD>(big, sm, n, accumulate, workspace, workspace_bytes, (hipStream_t)stream,
                                           "bilinear_ac_bwd_multi");
    if (rc) return rc;
    DSN_LAUNCH_CHECK("bilinear_ac_bwd_multi");
    return DSN_OK;
}

// ys[j] = bilinear(align_corners=True)(xs[j]) for up to four sources onto destinations of ONE common size (channel slices of a
// concat buffer): one launch
extern "C" int dsn_bilinear_ac_multi(const dsn_tensor* xs, const dsn_tensor* ys, int32_t n, void* stream) {
    DSN_CHECK_ARG(xs && ys && n >= 1 && n <= 4, "bilinear_ac_multi: invalid arguments");
    BilinearMulti m{};
    m.n = n; m.N = ys[0].n; m.Ho = ys[0].h; m.Wo = ys[0].w;
    int blk = 0;
    for (int j = 0; j < n; ++j) {
        DSN_CHECK_ARG(tensor_ok(&xs[j]) && tensor_ok(&ys[j]) && xs[j].dtype == ys[0].dtype && ys[j].dtype == ys[0].dtype &&
                          xs[j].n == ys[j].n && xs[j].c == ys[j].c && ys[j].n == m.N && ys[j].h == m.Ho && ys[j].w == m.Wo &&
                          vec16(&xs[j]) && vec16(&ys[j]),
                      "bilinear_ac_multi: job %d needs matching NHWC tensors with 16-byte channel vectors and a common output size", j);
        m.x[j] = xs[j].ptr; m.y[j] = ys[j].ptr; m.xld[j] = xs[j].ldc; m.yld[j] = ys[j].ldc;
        m.hi[j] = xs[j].h; m.wi[j] = xs[j].w; m.c[j] = xs[j].c;
        m.sh[j] = ac_scale(xs[j].h, m.Ho); m.sw[j] = ac_scale(xs[j].w, m.Wo);
        m.blk0[j] = blk;
        const int vw = ys[0].dtype == DSN_F32 ? 4 : 8;
        blk += ew_grid((int64_t)m.N * m.Ho * m.Wo * (xs[j].c / vw));
    }
    m.blk0[n] = blk;
    if (ys[0].dtype == DSN_F32)
        hipLaunchKernelGGL((bilinear_vec_multi_kernel<float, 4>), dim3((unsigned)blk), dim3(256), 0, (hipStream_t)stream, m);
    else
        hipLaunchKernelGGL((bilinear_vec_multi_kernel<bf16_t, 8>), dim3((unsigned)blk), dim3(256), 0, (hipStream_t)stream, m);
    DSN_LAUNCH_CHECK("bilinear_ac_multi");
    return DSN_OK;
}

extern "C" int64_t dsn_window_reduce_workspace_bytes(int32_t n_segments, int32_t rows, int32_t c) {
    return (int64_t)n_segments * wr_splits(n_segments, rows) * c * (int64_t)sizeof(float);
}

extern "C" int dsn_adaptive_avgpool(const dsn_tensor* x, const dsn_tensor* y, void* workspace, int64_t workspace_bytes,
                                    void* stream) {
    DSN_CHECK_ARG(tensor_ok(x) && tensor_ok(y) && x->dtype == y->dtype && x->n == y->n && x->c == y->c,
                  "adaptive_avgpool: invalid arguments");
    const int64_t nseg = (int64_t)y->n * y->h * y->w;
    WRGeom g{x->n, x->h, x->w, x->c, y->h, y->w, wr_splits(nseg, x->h), 0.f, 0.f, x->ldc, 0};
    if (!workspace || workspace_bytes < nseg * g.S * x->c * (int64_t)sizeof(float))
        DSN_FAIL(DSN_EWORKSPACE, "adaptive_avgpool: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    DSN_DISPATCH_DTYPE(x->dtype, T, {
        if (vec16(x))
            hipLaunchKernelGGL((window_reduce_kernel<T, WR_AVGPOOL, VW<T>::N>), dim3((unsigned)(nseg * g.S)), dim3(256), 0, st,
                               (const T*)x->ptr, (const T*)nullptr, (float*)workspace, g);
        else
            hipLaunchKernelGGL((window_reduce_kernel<T, WR_AVGPOOL, 1>), dim3((unsigned)(nseg * g.S)), dim3(256), 0, st,
                               (const T*)x->ptr, (const T*)nullptr, (float*)workspace, g);
        hipLaunchKernelGGL(window_finalize_kernel<T>, dim3((unsigned)(nseg * ((x->c + 31) / 32))), dim3(256), 0, st,
                           (const float*)workspace, g.S, nseg, x->c, (T*)y->ptr, y->ldc, 0);
    });
    DSN_LAUNCH_CHECK("adaptive_avgpool");
    return DSN_OK;
}

extern "C" int dsn_adaptive_avgpool_bwd_multi(const dsn_tensor* dys, int32_t n_src, const dsn_tensor* dx, int32_t accumulate,
                                              void* stream) {
    DSN_CHECK_ARG(dys && n_src >= 1 && n_src <= 4 && tensor_ok(dx), "adaptive_avgpool_bwd_multi: invalid arguments");
    bool vec = vec16(dx);
    for (int i = 0; i < n_src; ++i) {
        DSN_CHECK_ARG(tensor_ok(&dys[i]) && dx->dtype == dys[i].dtype && dx->n == dys[i].n && dx->c == dys[i].c,
                      "adaptive_avgpool_bwd_multi: source %d does not match dx", i);
        vec = vec && vec16(&dys[i]);
    }
    hipStream_t st = (hipStream_t)stream;
    if (vec) {
        PoolSrcs srcs{};
        srcs.n = n_src;
        for (int i = 0; i < n_src; ++i) {
            srcs.dy[i] = dys[i].ptr; srcs.ld[i] = dys[i].ldc; srcs.KH[i] = dys[i].h; srcs.KW[i] = dys[i].w;
        }
        const int V = dx->dtype == DSN_F32 ? 4 : 8;
        int64_t sum_kw = 0;
        for (int i = 0; i < n_src; ++i) sum_kw += dys[i].w;
        // LDS: the row-combined pooled vectors (fp32 [sum KW][C]) + the column table
        bool rows = sum_kw * dx->c * 4 + (int64_t)n_src * dx->w * 16 <= 60 * 1024 && dx->n * dx->h >= 64;
        for (int i = 0; i < n_src; ++i) rows = rows && dys[i].h <= dx->h && dys[i].w <= dx->w;     // (bins: at most two per row / column)
        if (rows) {
            static const int seg_items = [] { const char* e = getenv("DSN_POOLBWD_ITEMS"); return e ? atoi(e) : 512; }();
            int wseg = seg_items / (dx->c / V);               // columns per block
            wseg = wseg < 1 ? 1 : (wseg > dx->w ? dx->w : wseg);
            const size_t lds = (size_t)sum_kw * dx->c * 4 + (size_t)n_src * wseg * 16;
            const dim3 grid((unsigned)(dx->n * dx->h), (unsigned)((dx->w + wseg - 1) / wseg));
#define DSN_POOL_ROWS(T_, V_, NS_)                                                                                                \
    hipLaunchKernelGGL((adaptive_avgpool_bwd_rows_kernel<T_, V_, NS_>), grid, dim3(256), lds, st, srcs, (T_*)dx->ptr, dx->ldc, dx->h, \
                       dx->w, dx->c, accumulate, wseg)
            if (dx->dtype == DSN_F32) {
                switch (n_src) {
                    case 1: DSN_POOL_ROWS(float, 4, 1); break;
                    case 2: DSN_POOL_ROWS(float, 4, 2); break;
                    case 3: DSN_POOL_ROWS(float, 4, 3); break;
                    default: DSN_POOL_ROWS(float, 4, 4); break;
                }
            } else {
                switch (n_src) {
                    case 1: DSN_POOL_ROWS(bf16_t, 8, 1); break;
                    case 2: DSN_POOL_ROWS(bf16_t, 8, 2); break;
                    case 3: DSN_POOL_ROWS(bf16_t, 8, 3); break;
                    default: DSN_POOL_ROWS(bf16_t, 8, 4); break;
                }
            }
#undef DSN_POOL_ROWS
            DSN_LAUNCH_CHECK("adaptive_avgpool_bwd (rows)");
            return DSN_OK;
        }
        const int64_t total = npix(dx) * (dx->c / V);
        if (dx->dtype == DSN_F32)
            hipLaunchKernelGGL((adaptive_avgpool_bwd_multi_kernel<float, 4>), dim3(ew_grid(total)), dim3(256), 0, st, srcs,
                               (float*)dx->ptr, dx->ldc, dx->n, dx->h, dx->w, dx->c, accumulate);
        else
            hipLaunchKernelGGL((adaptive_avgpool_bwd_multi_kernel<bf16_t, 8>), dim3(ew_grid(total)), dim3(256), 0, st, srcs,
                               (bf16_t*)dx->ptr, dx->ldc, dx->n, dx->h, dx->w, dx->c, accumulate);
    } else {
        const int64_t total = npix(dx) * dx->c;
        for (int i = 0; i < n_src; ++i)
            DSN_DISPATCH_DTYPE(dx->dtype, T,
                               hipLaunchKernelGGL(adaptive_avgpool_bwd_kernel<T>, dim3(ew_grid(total)), dim3(256), 0, st,
                                                  (const T*)dys[i].ptr, dys[i].ldc, (T*)dx->ptr, dx->ldc, dx->n, dx->h, dx->w,
                                                  dx->c, dys[i].h, dys[i].w, (accumulate || i > 0) ? 1 : 0));
    }
    DSN_LAUNCH_CHECK("adaptive_avgpool_bwd");
    return DSN_OK;
}

extern "C" int dsn_adaptive_avgpool_bwd(const dsn_tensor* dy, const dsn_tensor* dx, int32_t accumulate, void* stream) {
    return dsn_adaptive_avgpool_bwd_multi(dy, 1, dx, accumulate, stream);
}

extern "C" int dsn_ffm_scale(const dsn_tensor* feat, const dsn_tensor* att, const dsn_tensor* out, void* stream) {
    DSN_CHECK_ARG(tensor_ok(feat) && tensor_ok(att) && tensor_ok(out) && same_nhwc(feat, out) &&
                      att->dtype == feat->dtype && att->n == feat->n && att->h == 1 && att->w == 1 && att->c == feat->c,
                  "ffm_scale: invalid arguments");
    const int64_t P = npix(feat);
    DSN_CHECK_ARG(P * feat->c < (1ll << 31), "ffm_scale: 2^31 or more elements");
    const unsigned HW = (unsigned)(feat->h * feat->w);
    DSN_DISPATCH_DTYPE(feat->dtype, T, {
        if (vec16(feat) && vec16(att) && vec16(out))
            hipLaunchKernelGGL((ffm_scale_kernel<T, VW<T>::N>), dim3(ew_grid(P * feat->c / VW<T>::N)), dim3(256), 0, (hipStream_t)stream,
                               (const T*)feat->ptr, feat->ldc, (const T*)att->ptr, att->ldc, (T*)out->ptr, out->ldc, HW, (unsigned)P,
                               feat->c);
        else
            hipLaunchKernelGGL((ffm_scale_kernel<T, 1>), dim3(ew_grid(P * feat->c)), dim3(256), 0, (hipStream_t)stream,
                               (const T*)feat->ptr, feat->ldc, (const T*)att->ptr, att->ldc, (T*)out->ptr, out->ldc, HW, (unsigned)P,
                               feat->c);
    });
    DSN_LAUNCH_CHECK("ffm_scale");
    return DSN_OK;
}

extern "C" int dsn_ffm_scale_bwd(const dsn_tensor* dout, const dsn_tensor* feat, const dsn_tensor* att,
                                 const dsn_tensor* dfeat, const dsn_tensor* datt, int32_t accumulate, void* workspace,
                                 int64_t workspace_bytes, void* stream) {
    DSN_CHECK_ARG(tensor_ok(dout) && tensor_ok(feat) && tensor_ok(att) && tensor_ok(dfeat) && tensor_ok(datt) &&
                      same_nhwc(dout, feat) && same_nhwc(dfeat, feat) && same_nhwc(att, datt) && att->n == feat->n &&
                      att->h == 1 && att->w == 1 && att->c == feat->c && att->dtype == feat->dtype,
                  "ffm_scale_bwd: invalid arguments");
    const int64_t P = npix(feat), HW = (int64_t)feat->h * feat->w;
    DSN_CHECK_ARG(P * feat->c < (1ll << 31), "ffm_scale_bwd: 2^31 or more elements");
    hipStream_t st = (hipStream_t)stream;
    const int64_t nseg = feat->n;
    WRGeom g{feat->n, feat->h, feat->w, feat->c, 1, 1, wr_splits(nseg, feat->h), 0.f, 0.f, dout->ldc, feat->ldc};
    if (!workspace || workspace_bytes < nseg * g.S * feat->c * (int64_t)sizeof(float))
        DSN_FAIL(DSN_EWORKSPACE, "ffm_scale_bwd: workspace too small");
    DSN_DISPATCH_DTYPE(feat->dtype, T, {
        if (vec16(dout) && vec16(feat))
            hipLaunchKernelGGL((window_reduce_kernel<T, WR_FFM_ATT, VW<T>::N>), dim3((unsigned)(nseg * g.S)), dim3(256), 0, st,
                               (const T*)dout->ptr, (const T*)feat->ptr, (float*)workspace, g);
        else
            hipLaunchKernelGGL((window_reduce_kernel<T, WR_FFM_ATT, 1>), dim3((unsigned)(nseg * g.S)), dim3(256), 0, st,
                               (const T*)dout->ptr, (const T*)feat->ptr, (float*)workspace, g);
        hipLaunchKernelGGL(window_finalize_kernel<T>, dim3((unsigned)(nseg * ((feat->c + 31) / 32))), dim3(256), 0, st,
                           (const float*)workspace, g.S, nseg, feat->c, (T*)datt->ptr, datt->ldc, 0);
        if (vec16(dout) && vec16(att) && vec16(dfeat))
            hipLaunchKernelGGL((ffm_scale_bwd_feat_kernel<T, VW<T>::N>), dim3(ew_grid(P * feat->c / VW<T>::N)), dim3(256), 0, st,
                               (const T*)dout->ptr, dout->ldc, (const T*)att->ptr, att->ldc, (T*)dfeat->ptr, dfeat->ldc, (unsigned)HW,
                               (unsigned)P, feat->c, accumulate);
        else
            hipLaunchKernelGGL((ffm_scale_bwd_feat_kernel<T, 1>), dim3(ew_grid(P * feat->c)), dim3(256), 0, st,
                               (const T*)dout->ptr, dout->ldc, (const T*)att->ptr, att->ldc, (T*)dfeat->ptr, dfeat->ldc, (unsigned)HW,
                               (unsigned)P, feat->c, accumulate);
    });
    DSN_LAUNCH_CHECK("ffm_scale_bwd");
    return DSN_OK;
}
