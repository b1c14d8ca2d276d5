// Error state, version, weight packing, casts and strided copies.
#include "common.h"

static thread_local char g_err[512] = "";

void dsn_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- live profiler -------------------------------------------------------------------------------------------------
#include <vector>
namespace {
struct ProfEntry { hipEvent_t a, b; int kid; double flops, bytes; };
bool g_prof_on = false;
std::vector<ProfEntry> g_prof;
size_t g_prof_used = 0;
}
ProfScope::ProfScope(int kid, double flops, double bytes, hipStream_t stream) : slot(-1), st(stream) {
    if (!g_prof_on) return;
    if (g_prof_used == g_prof.size()) {
        ProfEntry e{};
        if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) return;
        g_prof.push_back(e);
    }
    slot = (int)g_prof_used++;
    g_prof[slot].kid = kid; g_prof[slot].flops = flops; g_prof[slot].bytes = bytes;
    (void)hipEventRecord(g_prof[slot].a, st);
}
ProfScope::~ProfScope() {
    if (slot >= 0) (void)hipEventRecord(g_prof[slot].b, st);
}

extern "C" int dsn_profile_enable(int32_t on) {
    g_prof_on = on != 0;
    g_prof_used = 0;
    return DSN_OK;
}

// Aggregates per kernel id since dsn_profile_enable(1): out[kid] = {launches, total_ms, total_flops, total_bytes}.
extern "C" int dsn_profile_collect(double* out /* [KID_COUNT][4] */, int32_t n_kids) {
    DSN_CHECK_ARG(out && n_kids >= KID_COUNT, "profile_collect: need room for %d kernel ids", KID_COUNT);
    for (int i = 0; i < n_kids * 4; ++i) out[i] = 0.0;
    for (size_t i = 0; i < g_prof_used; ++i) {
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
        snprintf(buf, sizeof(buf), "igemm_%s_%s_%s", dt ? "bf16" : "f32", c < 7 ? cfg[c] : "?", dg ? "dgrad" : "fwd");
        return buf;
    }
    switch (kid) {
        case KID_WGRAD: return "wgrad_f32";
        case KID_WGRAD + 1: return "wgrad_bf16";
        case KID_WGRAD_REDUCE: return "wgrad_reduce";
        case KID_BN_STATS: return "bn_stats_reduce";
        case KID_BN_ACT_FWD: return "bn_act_fwd";
        case KID_BN_BWD_REDUCE: return "bn_act_bwd_reduce";
        case KID_BN_BWD_APPLY: return "bn_act_bwd_apply";
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
__global__ void pack_dgrad_kernel(const float* __restrict__ w, T* __restrict__ out, int co, int ci, int kh, int kw) {
    const int64_t total = (int64_t)co * kh * kw * ci;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int o = (int)(i % co);
        int64_t t = i / co;
        const int x = (int)(t % kw); t /= kw;
        const int y = (int)(t % kh);
        const int c = (int)(t / kh);
        out[i] = from_f32<T>(w[(((int64_t)o * ci + c) * kh + y) * kw + x]);
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

// ---- multi-tensor weight pack: ONE launch packs every conv weight of the model (forward + dgrad layouts) -------------------
// work[b] = {tensor id, first element}; each block converts up to PACK_CHUNK elements of one tensor.
constexpr int PACK_CHUNK = 2048;
template <typename T>
__global__ __launch_bounds__(256) void pack_multi_kernel(const dsn_pack_desc* __restrict__ descs,
                                                         const int2* __restrict__ work) {
    const int2 wk = work[blockIdx.x];
    const dsn_pack_desc d = descs[wk.x];
    const float* __restrict__ w = (const float*)d.w_oihw;
    T* __restrict__ of = (T*)d.out_fwd;
    T* __restrict__ od = (T*)d.out_dgrad;
    const int taps = d.kh * d.kw;
    const int64_t n_fwd = (int64_t)d.co * taps * d.ci_pad;
    const int64_t end = (wk.y + PACK_CHUNK < n_fwd) ? wk.y + PACK_CHUNK : n_fwd;
    for (int64_t i = wk.y + threadIdx.x; i < end; i += 256) {
        // i indexes the forward layout [co][tap][ci_pad]
        const int c = (int)(i % d.ci_pad);
        const int64_t t = i / d.ci_pad;
        const int tap = (int)(t % taps);
        const int o = (int)(t / taps);
        float v = 0.f;
        if (c < d.ci) v = w[((int64_t)o * d.ci + c) * taps + tap];
        if (of) of[i] = from_f32<T>(v);
        if (od && c < d.ci) od[((int64_t)c * taps + tap) * d.co + o] = from_f32<T>(v);
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
                                     int32_t kw, void* stream) {
    DSN_CHECK_ARG(w && out && co > 0 && ci > 0 && kh > 0 && kw > 0, "pack_weight_dgrad: bad args");
    const int64_t total = (int64_t)co * kh * kw * ci;
    DSN_DISPATCH_DTYPE(dtype, T,
                       hipLaunchKernelGGL(pack_dgrad_kernel<T>, dim3(grid_for(total)), dim3(256), 0,
                                          (hipStream_t)stream, w, (T*)out, co, ci, kh, kw));
    DSN_LAUNCH_CHECK("pack_weight_dgrad");
    return DSN_OK;
}

extern "C" int32_t dsn_pack_chunk(void) { return PACK_CHUNK; }

extern "C" int dsn_pack_weights_multi(const dsn_pack_desc* descs_dev, const int32_t* work_dev, int32_t n_work,
                                      int32_t dtype, void* stream) {
    DSN_CHECK_ARG(descs_dev && work_dev && n_work > 0, "pack_weights_multi: bad args");
    DSN_DISPATCH_DTYPE(dtype, T,
                       hipLaunchKernelGGL(pack_multi_kernel<T>, dim3(n_work), dim3(256), 0, (hipStream_t)stream, descs_dev,
                                          (const int2*)work_dev));
    DSN_LAUNCH_CHECK("pack_weights_multi");
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
    DSN_DISPATCH_DTYPE(x->dtype, T,
                       hipLaunchKernelGGL(copy_kernel<T>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                                          (const T*)x->ptr, (T*)y->ptr, npix(x), x->c, x->ldc, y->ldc, accumulate));
    DSN_LAUNCH_CHECK("copy");
    return DSN_OK;
}
