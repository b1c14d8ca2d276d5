// Error state, version, weight packing, casts and strided copies.
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void dsn_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- dynamic-LDS opt-in, once per (kernel, device) ------------------------------------------------------------------------
#include <mutex>
#include <unordered_map>
int dsn_lds_attr(const void* kern, int bytes) {
    static std::mutex mu;
    static std::unordered_map<const void*, uint64_t> done;        // kernel -> bit per device ordinal
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    const uint64_t bit = 1ull << (dev & 63);
    std::lock_guard<std::mutex> lk(mu);
    uint64_t& m = done[kern];
    if (m & bit) return DSN_OK;
    const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        dsn_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize = %d) on device %d: %s", bytes, dev, hipGetErrorString(e));
        return (int)e;
    }
    m |= bit;
    return DSN_OK;
}

// ---- live profiler -------------------------------------------------------------------------------------------------
#include <map>
#include <string>
#include <vector>
extern "C" const char* dsn_profile_kernel_name(int32_t kid);
namespace {
struct ProfEntry { hipEvent_t a, b; int kid; double flops, bytes; char label[64], layer[64]; hipStream_t st; };
bool g_prof_on = false;
std::vector<ProfEntry> g_prof;
size_t g_prof_used = 0;
int prof_slot(hipStream_t st) {
    if (!g_prof_on) return -1;
    if (g_prof_used == g_prof.size()) {
        ProfEntry e{};
        if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) return -1;
        g_prof.push_back(e);
    }
    return (int)g_prof_used++;
}
}
bool dsn_prof_on() { return g_prof_on; }
ProfScope::ProfScope(int kid, double flops, double bytes, hipStream_t stream) : slot(prof_slot(stream)), st(stream) {
    if (slot < 0) return;
    ProfEntry& e = g_prof[slot];
    e.kid = kid; e.flops = flops; e.bytes = bytes; e.st = stream;
    snprintf(e.label, sizeof(e.label), "%s", dsn_profile_kernel_name(kid));
    e.layer[0] = 0;
    (void)hipEventRecord(e.a, st);
}
ProfScope::ProfScope(const char* label, const char* layer, double flops, double bytes, hipStream_t stream)
    : slot(prof_slot(stream)), st(stream) {
    if (slot < 0) return;
    ProfEntry& e = g_prof[slot];
    e.kid = -1; e.flops = flops; e.bytes = bytes; e.st = stream;
    snprintf(e.label, sizeof(e.label), "%s", label);
    snprintf(e.layer, sizeof(e.layer), "%s", layer ? layer : "");
    (void)hipEventRecord(e.a, st);
}
ProfScope::~ProfScope() {
    if (slot >= 0) (void)hipEventRecord(g_prof[slot].b, st);
}

// An event pair with NOTHING between its two records measures two queue markers back to back (4.4 us on MI355X / ROCm 7.2); around a
// kernel the first marker lies outside the interval and the second inside, so HALF of the empty pair is subtracted from every record --
// a record then approximates the kernel's duration as rocprofv3 --kernel-trace reports it (checked on the step's BatchNorm passes:
// 10.7 us raw, 8.5 corrected, 8.3 under rocprofv3).  For the 5-10 us launches of a batch-8 step the marker was 20-30 % of the figure.
// Minimum of 16 empty pairs on the records' stream.
namespace {
float prof_pair_overhead_ms(hipStream_t st) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess) return 0.f;
    if (hipEventCreate(&b) != hipSuccess) { (void)hipEventDestroy(a); return 0.f; }
    float best = 1e9f;
    for (int i = 0; i < 16; ++i) {
        float ms = 0.f;
        if (hipEventRecord(a, st) != hipSuccess || hipEventRecord(b, st) != hipSuccess || hipEventSynchronize(b) != hipSuccess ||
            hipEventElapsedTime(&ms, a, b) != hipSuccess) { best = 0.f; break; }
        best = ms < best ? ms : best;
    }
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return best > 1e8f ? 0.f : 0.5f * best;
}
float prof_adjust(float ms, float ov) { const float v = ms - ov; return v > 0.2f * ms ? v : 0.2f * ms; }
}  // namespace

extern "C" int dsn_profile_enable(int32_t on) {
    g_prof_on = on != 0;
    g_prof_used = 0;
    return DSN_OK;
}

// Text dump of everything recorded since dsn_profile_enable(1), aggregated per (label, layer): one line per pair,
// "label\tlayer\tlaunches\ttotal_ms\ttotal_flops\ttotal_bytes\n".  Returns the number of bytes the full dump needs (call
// again with a larger buffer if it exceeds `cap`; the records are kept until the dump fitted), < 0 on error.
extern "C" int64_t dsn_profile_dump(char* out, int64_t cap) {
    struct Agg { double n = 0, ms = 0, fl = 0, by = 0; };
    std::map<std::pair<std::string, std::string>, Agg> agg;
    static float ov = 0.f;                    // (measured when a dump starts, kept for the second call that fetches the text)
    if (!out && g_prof_used) {
        (void)hipEventSynchronize(g_prof[g_prof_used - 1].b);
        ov = prof_pair_overhead_ms(g_prof[0].st);
    }
    for (size_t i = 0; i < g_prof_used; ++i) {
        float ms = 0.f;
        hipError_t e = hipEventSynchronize(g_prof[i].b);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, g_prof[i].a, g_prof[i].b);
        if (e != hipSuccess) { dsn_set_error("profile_dump: %s", hipGetErrorString(e)); return -(int64_t)e; }
        Agg& a = agg[{g_prof[i].label, g_prof[i].layer}];
        a.n += 1.0; a.ms += prof_adjust(ms, ov); a.fl += g_prof[i].flops; a.by += g_prof[i].bytes;
    }
    std::string s;
    char line[256];
    snprintf(line, sizeof(line), "__event_pair_overhead\t\t1\t%.6f\t0\t0\n", ov);      // (what was subtracted from every record: one marker)
    s += line;
    for (auto& kv : agg) {
        snprintf(line, sizeof(line), "%s\t%s\t%.0f\t%.6f\t%.6e\t%.6e\n", kv.first.first.c_str(), kv.first.second.c_str(),
                 kv.second.n, kv.second.ms, kv.second.fl, kv.second.by);
        s += line;
    }
    const int64_t need = (int64_t)s.size() + 1;
    if (out && cap >= need) {
        memcpy(out, s.c_str(), (size_t)need);
        g_prof_used = 0;
    }
    return need;
}

// Aggregates per legacy kernel id since dsn_profile_enable(1): out[kid] = {launches, total_ms, total_flops, total_bytes}
// (labelled records -- the convolution launches -- are not in this view: dsn_profile_dump).
extern "C" int dsn_profile_collect(double* out /* [KID_COUNT][4] */, int32_t n_kids) {
    DSN_CHECK_ARG(out && n_kids >= KID_COUNT, "profile_collect: need room for %d kernel ids", KID_COUNT);
    for (int i = 0; i < n_kids * 4; ++i) out[i] = 0.0;
    for (size_t i = 0; i < g_prof_used; ++i) {
        if (g_prof[i].kid < 0) continue;
        float ms = 0.f;
        hipError_t e = hipEventSynchronize(g_prof[i].b);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, g_prof[i].a, g_prof[i].b);
        if (e != hipSuccess) DSN_FAIL((int)e, "profile_collect: %s", hipGetErrorString(e));
        double* o = out + 4 * g_prof[i].kid;
        o[0] += 1.0; o[1] += ms; o[2] += g_prof[i].flops; o[3] += g_prof[i].bytes;
    }
    g_prof_used = 0;
    return DSN_OK;
}

extern "C" int32_t dsn_profile_kernel_count(void) { return KID_COUNT; }

extern "C" const char* dsn_profile_kernel_name(int32_t kid) {
    static thread_local char buf[64];
    static const char* cfg[7] = {"128x128", "128x64", "64x64", "128x32", "64x16", "32x64", "64x32"};
    if (kid >= KID_IGEMM && kid < KID_WGRAD) {
        const int dt = kid / 20, c = (kid % 20) / 2, dg = kid & 1;
        snprintf(buf, sizeof(buf), "igemm_kernel/%s/%s/%s", dt ? "bf16" : "f32", c < 7 ? cfg[c] : "?", dg ? "dgrad" : "fwd");
        return buf;
    }
    switch (kid) {
        case KID_WGRAD: return "wgrad_grouped/f32";
        case KID_WGRAD + 1: return "wgrad_grouped/bf16";
        case KID_WGRAD_REDUCE: return "wgrad_reduce_grouped_kernel";
        case KID_BN_STATS: return "reduce2_kernel/StatsF";
        case KID_BN_ACT_FWD: return "lazy_ew_kernel+ew2_kernel/FwdF";
        case KID_BN_BWD_REDUCE: return "reduce2_kernel/BwdRedF";
        case KID_BN_BWD_APPLY: return "ew2_kernel/BwdApplyF";
    }
    return "?";
}

extern "C" int dsn_version(void) { return DSN_VERSION; }
extern "C" const char* dsn_last_error(void) { return g_err; }

namespace {

// out[co][kh][kw][ci_pad] = scale[co] * w[co][ci][kh][kw]
template <typename T>
__global__ void pack_fwd_kernel(const float* __restrict__ w, const float* __restrict__ scale, T* __restrict__ out,
                                int co, int ci, int kh, int kw, int ci_pad) {
    const int64_t total = (int64_t)co * kh * kw * ci_pad;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % ci_pad);
        int64_t t = i / ci_pad;
        const int x = (int)(t % kw); t /= kw;
        const int y = (int)(t % kh);
        const int o = (int)(t / kh);
        float v = 0.f;
        if (c < ci) {
            v = w[(((int64_t)o * ci + c) * kh + y) * kw + x];
            if (scale) v *= scale[o];
        }
        out[i] = from_f32<T>(v);
    }
}

// out[ci][kh][kw][co] = w[co][ci][kh][kw]
template <typename T>
__global__ void pack_dgrad_kernel(const float* __restrict__ w, T* __restrict__ out, int co, int ci, int kh, int kw,
                                  int co_pad) {
    const int64_t total = (int64_t)co_pad * kh * kw * ci;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int o = (int)(i % co_pad);
        int64_t t = i / co_pad;
        const int x = (int)(t % kw); t /= kw;
        const int y = (int)(t % kh);
        const int c = (int)(t / kh);
        out[i] = from_f32<T>(o < co ? w[(((int64_t)o * ci + c) * kh + y) * kw + x] : 0.f);
    }
}

__global__ void unpack_wgrad_kernel(const float* __restrict__ dw, float* __restrict__ grad, int co, int ci, int kh,
                                    int kw, int ci_pad, int accumulate) {
    const int64_t total = (int64_t)co * ci * kh * kw;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % kw);
        int64_t t = i / kw;
        const int y = (int)(t % kh); t /= kh;
        const int c = (int)(t % ci);
        const int o = (int)(t / ci);
        const float v = dw[(((int64_t)o * kh + y) * kw + x) * ci_pad + c];
        grad[i] = accumulate ? grad[i] + v : v;
    }
}

template <typename T>
__global__ void cast_kernel(const float* __restrict__ src, T* __restrict__ dst, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = from_f32<T>(src[i]);
}

template <typename T>
__global__ void copy_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t npx, int c, int64_t xld, int64_t yld,
                            int accumulate) {
    const int64_t total = npx * c;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = i / c;
        const int ch = (int)(i - p * c);
        float v = to_f32<T>(x[p * xld + ch]);
        T* o = y + p * yld + ch;
        if (accumulate) v += to_f32<T>(*o);
        *o = from_f32<T>(v);
    }
}

template <typename T, int V>
__global__ void copy_vec_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t npx, int c, int64_t xld, int64_t yld,
                                int accumulate) {
    const int ncv = c / V;
    const int64_t total = npx * ncv;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = i / ncv;
        const int cv = (int)(i - p * ncv);
        float v[V];
        VecIO<T, V>::load(x + p * xld + cv * V, v);
        T* o = y + p * yld + cv * V;
        if (accumulate) {
            float old[V];
            VecIO<T, V>::load(o, old);
#pragma unroll
            for (int k = 0; k < V; ++k) v[k] += old[k];
        }
        VecIO<T, V>::store(o, v);
    }
}

// ---- multi-tensor weight pack: ONE launch packs every conv weight of the model (forward + dgrad layouts) -------------------
// work[b] = {tensor id, tile index}.  OIHW is a [co][J] matrix with J = ci*taps contiguous; a block moves one 32(o) x 64(j)
// tile through LDS: coalesced 256-byte fp32 row reads, the dgrad layout [j][o] (= [ci][tap][co]) leaves as 64-byte segments
// and the forward layout [o][tap][ci_pad] as contiguous channels for 1x1 convs (stride ci_pad for 3x3).  The previous
// element-per-thread version gathered with stride `taps` and scattered 2-byte dgrad writes: 110 us for 29 MB.
constexpr int PACK_TO = 32, PACK_TJ = 64;
template <typename T>
__global__ __launch_bounds__(256) void pack_multi_kernel(const dsn_pack_desc* __restrict__ descs,
                                                         const int2* __restrict__ work) {
    __shared__ float tile[PACK_TO][PACK_TJ + 1];
    const int2 wk = work[blockIdx.x];
    const dsn_pack_desc d = descs[wk.x];
    const float* __restrict__ w = (const float*)d.w_oihw;
    T* __restrict__ of = (T*)d.out_fwd;
    T* __restrict__ od = (T*)d.out_dgrad;
    const int taps = d.kh * d.kw;
    const int J = d.ci * taps;
    const int tiles_j = (J + PACK_TJ - 1) / PACK_TJ;
    const int o0 = (wk.y / tiles_j) * PACK_TO, j0 = (wk.y % tiles_j) * PACK_TJ;
    for (int idx = threadIdx.x; idx < PACK_TO * PACK_TJ; idx += 256) {
        const int r = idx / PACK_TJ, cj = idx % PACK_TJ;
        const int o = o0 + r, j = j0 + cj;
        tile[r][cj] = (o < d.co && j < J) ? w[(int64_t)o * J + j] : 0.f;
    }
    __syncthreads();
    if (of) {
        for (int idx = threadIdx.x; idx < PACK_TO * PACK_TJ; idx += 256) {
            const int r = idx / PACK_TJ, cj = idx % PACK_TJ;
            const int o = o0 + r, j = j0 + cj;
            if (o < d.co && j < J) {
                const int c = j / taps, tap = j - c * taps;
                of[((int64_t)o * taps + tap) * d.ci_pad + c] = from_f32<T>(tile[r][cj]);
            }
        }
    }
    if (od) {
        for (int idx = threadIdx.x; idx < PACK_TO * PACK_TJ; idx += 256) {
            const int cj = idx / PACK_TO, r = idx % PACK_TO;
            const int o = o0 + r, j = j0 + cj;
            if (o < d.co && j < J) od[(int64_t)j * (d.co_pad > d.co ? d.co_pad : d.co) + o] = from_f32<T>(tile[r][cj]);
        }
    }
    T* __restrict__ o2 = (T*)d.out_dgrad_s2;
    if (o2) {     // 3x3 stride-2 convs: W2[(py,px,ci)][ty][tx][co], (ky -> py,ty): 1 -> (0,0), 2 -> (1,0), 0 -> (1,1); same for kx
        for (int idx = threadIdx.x; idx < PACK_TO * PACK_TJ; idx += 256) {
            const int cj = idx / PACK_TO, r = idx % PACK_TO;
            const int o = o0 + r, j = j0 + cj;
            if (o < d.co && j < J) {
                const int c = j / 9, tap = j - c * 9, ky = tap / 3, kx = tap - ky * 3;
                const int py = ky == 1 ? 0 : 1, ty = ky == 0 ? 1 : 0, px = kx == 1 ? 0 : 1, tx = kx == 0 ? 1 : 0;
                o2[((((int64_t)(py * 2 + px) * d.ci + c) * 2 + ty) * 2 + tx) * d.co + o] = from_f32<T>(tile[r][cj]);
            }
        }
    }
}

// ---- multi-tensor SGD (momentum, Nesterov, weight decay): torch.optim.SGD's update for every parameter in ONE launch -------
// p, g, buf are fp32.  hyper[group] = {lr, momentum, dampening, weight_decay, nesterov, first_step, 0, 0} lives in DEVICE
// memory so that a captured graph picks up learning-rate changes (the scheduler writes it between replays).
//   g' = g + wd*p ; buf = first ? g' : mom*buf + (1-damp)*g' ; step = nesterov ? g' + mom*buf : buf ; p -= lr*step
constexpr int SGD_CHUNK = 1024;       // elements per block: 256 threads x float4
__global__ __launch_bounds__(256) void sgd_multi_kernel(const dsn_sgd_desc* __restrict__ descs, int n,
                                                        const float* __restrict__ hyper) {
    int lo = 0, hi = n - 1;           // largest tensor whose first chunk is <= blockIdx.x (uniform: scalar loads)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].first_chunk <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const dsn_sgd_desc d = descs[lo];
    const float* h = hyper + 8 * d.group;
    const float lr = h[0], mom = h[1], damp = h[2], wd = h[3];
    const bool nesterov = h[4] != 0.f, first = h[5] != 0.f;
    const int64_t i0 = ((int64_t)blockIdx.x - d.first_chunk) * SGD_CHUNK + threadIdx.x * 4;
    float* __restrict__ p = (float*)d.param;
    const float* __restrict__ g = (const float*)d.grad;
    float* __restrict__ b = (float*)d.momentum_buf;
    if (i0 + 4 <= d.numel && (((uintptr_t)p | (uintptr_t)g | (uintptr_t)b) & 15) == 0) {
        f32x4 pv = *reinterpret_cast<const f32x4*>(p + i0);
        const f32x4 gv = *reinterpret_cast<const f32x4*>(g + i0);
        f32x4 bv = (mom != 0.f && !first) ? *reinterpret_cast<const f32x4*>(b + i0) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float gg = gv[k] + wd * pv[k];
            if (mom != 0.f) {
                bv[k] = first ? gg : mom * bv[k] + (1.f - damp) * gg;
                gg = nesterov ? gg + mom * bv[k] : bv[k];
            }
            pv[k] -= lr * gg;
        }
        *reinterpret_cast<f32x4*>(p + i0) = pv;
        if (mom != 0.f) *reinterpret_cast<f32x4*>(b + i0) = bv;
    } else {
        for (int64_t i = i0; i < i0 + 4 && i < d.numel; ++i) {
            float gg = g[i] + wd * p[i];
            if (mom != 0.f) {
                const float nb = first ? gg : mom * b[i] + (1.f - damp) * gg;
                b[i] = nb;
                gg = nesterov ? gg + mom * nb : nb;
            }
            p[i] -= lr * gg;
        }
    }
}

// ---- multi-tensor EMA: ModelEMA.update (torch_utils.py:330-342) for every floating-point state_dict entry in ONE launch -----
// The reference runs `v *= d; v += (1. - d) * m` per tensor: three separately rounded fp32 operations (no FMA), with d and
// (1 - d) computed in double on the host and cast to fp32 -- reproduced exactly (__fmul_rn / __fadd_rn never contract).
// coef (DEVICE memory) = {(float)d, (float)(1 - d)} so a captured graph follows the decay ramp.
__global__ __launch_bounds__(256) void ema_multi_kernel(const dsn_ema_desc* __restrict__ descs, int n,
                                                        const float* __restrict__ coef) {
#pragma clang fp contract(off)      // plain operators below, never fused (HIP's __fmul_rn / __fadd_rn inline as contractable ops)
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].first_chunk <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const dsn_ema_desc d = descs[lo];
    const float dk = coef[0], omd = coef[1];
    const int64_t i0 = ((int64_t)blockIdx.x - d.first_chunk) * SGD_CHUNK + threadIdx.x * 4;
    float* __restrict__ e = (float*)d.ema;
    const float* __restrict__ m = (const float*)d.model;
    if (i0 + 4 <= d.numel && (((uintptr_t)e | (uintptr_t)m) & 15) == 0) {
        f32x4 ev = *reinterpret_cast<const f32x4*>(e + i0);
        const f32x4 mv = *reinterpret_cast<const f32x4*>(m + i0);
#pragma unroll
        for (int k = 0; k < 4; ++k) ev[k] = ev[k] * dk + omd * mv[k];
        *reinterpret_cast<f32x4*>(e + i0) = ev;
    } else {
        for (int64_t i = i0; i < i0 + 4 && i < d.numel; ++i) e[i] = e[i] * dk + omd * m[i];
    }
}

inline int grid_for(int64_t total, int threads = 256) {
    int64_t b = (total + threads - 1) / threads;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace

extern "C" int dsn_pack_weight_fwd(const float* w, const float* scale, void* out, int32_t dtype, int32_t co, int32_t ci,
                                   int32_t kh, int32_t kw, int32_t ci_pad, void* stream) {
    DSN_CHECK_ARG(w && out && co > 0 && ci > 0 && kh > 0 && kw > 0 && ci_pad >= ci, "pack_weight_fwd: bad args");
    const int64_t total = (int64_t)co * kh * kw * ci_pad;
    DSN_DISPATCH_DTYPE(dtype, T,
                       hipLaunchKernelGGL(pack_fwd_kernel<T>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                                          w, scale, (T*)out, co, ci, kh, kw, ci_pad));
    DSN_LAUNCH_CHECK("pack_weight_fwd");
    return DSN_OK;
}

extern "C" int dsn_pack_weight_dgrad(const float* w, void* out, int32_t dtype, int32_t co, int32_t ci, int32_t kh,
                                     int32_t kw, int32_t co_pad, void* stream) {
    DSN_CHECK_ARG(w && out && co > 0 && ci > 0 && kh > 0 && kw > 0 && co_pad >= co, "pack_weight_dgrad: bad args");
    const int64_t total = (int64_t)co_pad * kh * kw * ci;
    DSN_DISPATCH_DTYPE(dtype, T,
                       hipLaunchKernelGGL(pack_dgrad_kernel<T>, dim3(grid_for(total)), dim3(256), 0,
                                          (hipStream_t)stream, w, (T*)out, co, ci, kh, kw, co_pad));
    DSN_LAUNCH_CHECK("pack_weight_dgrad");
    return DSN_OK;
}

extern "C" int32_t dsn_pack_tiles(int32_t co, int32_t ci, int32_t kh, int32_t kw) {
    return ((co + PACK_TO - 1) / PACK_TO) * ((ci * kh * kw + PACK_TJ - 1) / PACK_TJ);
}

extern "C" int dsn_pack_weights_multi(const dsn_pack_desc* descs_dev, const int32_t* work_dev, int32_t n_work,
                                      int32_t dtype, void* stream) {
    DSN_CHECK_ARG(descs_dev && work_dev && n_work > 0, "pack_weights_multi: bad args");
    DSN_DISPATCH_DTYPE(dtype, T,
                       hipLaunchKernelGGL(pack_multi_kernel<T>, dim3(n_work), dim3(256), 0, (hipStream_t)stream, descs_dev,
                                          (const int2*)work_dev));
    DSN_LAUNCH_CHECK("pack_weights_multi");
    return DSN_OK;
}

extern "C" int32_t dsn_sgd_chunk(void) { return SGD_CHUNK; }

extern "C" int dsn_sgd_step(const dsn_sgd_desc* descs_dev, int32_t n_tensors, int32_t n_chunks, const float* hyper_dev,
                            void* stream) {
    DSN_CHECK_ARG(descs_dev && hyper_dev && n_tensors > 0 && n_chunks > 0, "sgd_step: bad args");
    hipLaunchKernelGGL(sgd_multi_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, descs_dev, n_tensors, hyper_dev);
    DSN_LAUNCH_CHECK("sgd_step");
    return DSN_OK;
}

extern "C" int dsn_unpack_wgrad(const float* dw, float* grad, int32_t co, int32_t ci, int32_t kh, int32_t kw,
                                int32_t ci_pad, int32_t accumulate, void* stream) {
    DSN_CHECK_ARG(dw && grad && co > 0 && ci > 0 && kh > 0 && kw > 0 && ci_pad >= ci, "unpack_wgrad: bad args");
    const int64_t total = (int64_t)co * kh * kw * ci;
    hipLaunchKernelGGL(unpack_wgrad_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dw, grad, co, ci,
                       kh, kw, ci_pad, accumulate);
    DSN_LAUNCH_CHECK("unpack_wgrad");
    return DSN_OK;
}

extern "C" int dsn_cast(const float* src, void* dst, int32_t dtype, int64_t n, void* stream) {
    DSN_CHECK_ARG(src && dst && n >= 0, "cast: bad args");
    if (n == 0) return DSN_OK;
    DSN_DISPATCH_DTYPE(dtype, T,
                       hipLaunchKernelGGL(cast_kernel<T>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, src,
                                          (T*)dst, n));
    DSN_LAUNCH_CHECK("cast");
    return DSN_OK;
}

extern "C" int dsn_copy(const dsn_tensor* x, const dsn_tensor* y, int32_t accumulate, void* stream) {
    DSN_CHECK_ARG(tensor_ok(x) && tensor_ok(y) && x->dtype == y->dtype, "copy: invalid tensors");
    DSN_CHECK_ARG(x->n == y->n && x->h == y->h && x->w == y->w && x->c == y->c, "copy: shape mismatch");
    const int64_t total = npix(x) * x->c;
    const int vw = x->dtype == DSN_F32 ? 4 : 8;
    if (x->c % vw == 0 && x->ldc % vw == 0 && y->ldc % vw == 0 && ((uintptr_t)x->ptr % 16) == 0 && ((uintptr_t)y->ptr % 16) == 0) {
        if (x->dtype == DSN_F32)
            hipLaunchKernelGGL((copy_vec_kernel<float, 4>), dim3(grid_for(total / 4)), dim3(256), 0, (hipStream_t)stream,
                               (const float*)x->ptr, (float*)y->ptr, npix(x), x->c, x->ldc, y->ldc, accumulate);
        else
            hipLaunchKernelGGL((copy_vec_kernel<bf16_t, 8>), dim3(grid_for(total / 8)), dim3(256), 0, (hipStream_t)stream,
                               (const bf16_t*)x->ptr, (bf16_t*)y->ptr, npix(x), x->c, x->ldc, y->ldc, accumulate);
        DSN_LAUNCH_CHECK("copy");
        return DSN_OK;
    }
    DSN_DISPATCH_DTYPE(x->dtype, T,
                       hipLaunchKernelGGL(copy_kernel<T>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                                          (const T*)x->ptr, (T*)y->ptr, npix(x), x->c, x->ldc, y->ldc, accumulate));
    DSN_LAUNCH_CHECK("copy");
    return DSN_OK;
}

// ---- small utilities that keep host-framework kernels out of the captured step ---------------------------------------------
// dsn_fill32: p[0..n_words) = value (gradient buffer / accumulator arena clears; hipMemsetAsync graph nodes are avoided, see
// common.h).  dsn_add_i64: p[i] += v (BatchNorm num_batches_tracked: every counter is a view of one int64 vector).
extern "C" int dsn_fill32(void* p, uint32_t value, int64_t n_words, void* stream) {
    DSN_CHECK_ARG(p && n_words >= 0 && ((uintptr_t)p % 4) == 0, "fill32: bad args");
    if ((((uintptr_t)p) % 16) == 0 && n_words >= 4) {
        const int64_t nv = n_words / 4;
        int64_t b = (nv + 255) / 256;
        hipLaunchKernelGGL(dsn_fill_u32x4_kernel, dim3((unsigned)(b > 2048 ? 2048 : b)), dim3(256), 0, (hipStream_t)stream,
                           (u32x4*)p, value, nv, (uint32_t*)p + nv * 4, n_words - nv * 4);
    } else {
        dsn_fill_u32(p, value, n_words, (hipStream_t)stream);
    }
    DSN_LAUNCH_CHECK("fill32");
    return DSN_OK;
}

static __global__ void add_i64_kernel(int64_t* __restrict__ p, int64_t n, int64_t v) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) p[i] += v;
}
// dsn_copy_multi: up to DSN_COPY_MAXSEG flat copies in ONE launch, each zero-filled from copy_bytes to total_bytes -- the staging of
// a training batch into the static buffers a captured step reads (image batch, label rows padded to the buffer's capacity, masks:
// train.py:329 `imgs.to(device)`, :352-354 targets) instead of three copies and a fill.
namespace {
struct CopyPlan {
    int32_t n;
    int32_t first[DSN_COPY_MAXSEG + 1];     // first block of each segment
    dsn_copy_seg s[DSN_COPY_MAXSEG];
    int32_t wide[DSN_COPY_MAXSEG];          // 1: 16-byte units, 0: 4-byte units
};
constexpr int COPY_PER_THREAD = 4;
__global__ __launch_bounds__(256) void copy_multi_kernel(const CopyPlan pl) {
    int j = 0;
    while (j + 1 < pl.n && (int)blockIdx.x >= pl.first[j + 1]) ++j;
    const dsn_copy_seg sg = pl.s[j];
    const int64_t base = ((int64_t)blockIdx.x - pl.first[j]) * 256 * COPY_PER_THREAD + threadIdx.x;
    if (pl.wide[j]) {
        const int64_t nc = sg.copy_bytes / 16, nt = sg.total_bytes / 16;
        u32x4 v[COPY_PER_THREAD];
#pragma unroll
        for (int u = 0; u < COPY_PER_THREAD; ++u) {
            const int64_t i = base + u * 256;
            v[u] = i < nc ? ((const u32x4*)sg.src)[i] : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int u = 0; u < COPY_PER_THREAD; ++u) {
            const int64_t i = base + u * 256;
            if (i < nt) ((u32x4*)sg.dst)[i] = v[u];
        }
    } else {
        const int64_t nc = sg.copy_bytes / 4, nt = sg.total_bytes / 4;
#pragma unroll
        for (int u = 0; u < COPY_PER_THREAD; ++u) {
            const int64_t i = base + u * 256;
            if (i < nt) ((uint32_t*)sg.dst)[i] = i < nc ? ((const uint32_t*)sg.src)[i] : 0u;
        }
    }
}
}  // namespace

extern "C" int dsn_copy_multi(const dsn_copy_seg* segs, int32_t n, void* stream) {
    DSN_CHECK_ARG(segs && n > 0 && n <= DSN_COPY_MAXSEG, "copy_multi: 1..%d segments", DSN_COPY_MAXSEG);
    CopyPlan pl{};
    pl.n = n;
    int64_t blocks = 0;
    for (int i = 0; i < n; ++i) {
        const dsn_copy_seg& g = segs[i];
        DSN_CHECK_ARG(g.dst && g.copy_bytes >= 0 && g.total_bytes >= g.copy_bytes && (g.src || g.copy_bytes == 0),
                      "copy_multi: bad segment %d", i);
        DSN_CHECK_ARG(g.copy_bytes % 4 == 0 && g.total_bytes % 4 == 0 && (uintptr_t)g.dst % 4 == 0 && (uintptr_t)g.src % 4 == 0,
                      "copy_multi: segment %d is not made of aligned 32-bit words", i);
        pl.s[i] = g;
        pl.wide[i] = (g.copy_bytes % 16 == 0 && g.total_bytes % 16 == 0 && (uintptr_t)g.dst % 16 == 0 && (uintptr_t)g.src % 16 == 0) ? 1 : 0;
        const int64_t units = g.total_bytes / (pl.wide[i] ? 16 : 4);
        pl.first[i] = (int32_t)blocks;
        blocks += (units + 256 * COPY_PER_THREAD - 1) / (256 * COPY_PER_THREAD);
        DSN_CHECK_ARG(blocks < (1ll << 30), "copy_multi: too large");
    }
    pl.first[n] = (int32_t)blocks;
    if (blocks == 0) return DSN_OK;
    hipLaunchKernelGGL(copy_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, pl);
    DSN_LAUNCH_CHECK("copy_multi");
    return DSN_OK;
}

extern "C" int dsn_add_i64(void* p, int64_t n, int64_t value, void* stream) {
    DSN_CHECK_ARG(p && n >= 0, "add_i64: bad args");
    if (n == 0) return DSN_OK;
    hipLaunchKernelGGL(add_i64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (int64_t*)p, n, value);
    DSN_LAUNCH_CHECK("add_i64");
    return DSN_OK;
}

extern "C" int dsn_ema_step(const dsn_ema_desc* descs_dev, int32_t n_tensors, int32_t n_chunks, const float* coef_dev,
                            void* stream) {
    DSN_CHECK_ARG(descs_dev && coef_dev && n_tensors > 0 && n_chunks > 0, "ema_step: bad args");
    hipLaunchKernelGGL(ema_multi_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, descs_dev, n_tensors, coef_dev);
    DSN_LAUNCH_CHECK("ema_step");
    return DSN_OK;
}
