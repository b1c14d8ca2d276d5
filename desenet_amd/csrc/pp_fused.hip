// PyramidPooling's four branches (common.py:588-615: AdaptiveAvgPool2d(k) -> Conv(1x1) + BatchNorm + SiLU, k = 1, 2, 3, 6) as ONE launch
// each way (round 3).  The pooled maps are 8 .. 288 pixels: the generic path spends 8 launches forward (conv, BN + act per branch)
// and 11 backward (BN sums, BN apply, dgrad per branch; the weight gradients ride in the grouped launch) on ~2 MFLOP -- ~19 kernels at
// the 4.6 us floor of the step.  Here block j owns branch j end to end:
//   forward : z = x W^T (bf16 MFMA, fp32 accumulators) -> batch statistics of the fp32 accumulators (fp64 sums, fixed order) ->
//             scale / shift / running averages -> y = act(z * scale + shift) from the ROUNDED z, as the elementwise pass reads it;
//   backward: g = dy * act'(z * scale + shift), the two BatchNorm sums (fp64, fixed order), dgamma / dbeta, dz rounded to bf16 as the
//             generic pass stores it, then dx = dz W (MFMA) and dW += dz^T x (MFMA, K = pixels) straight into the fp32 gradient.
// Branches whose map is 1 x 1 have no BatchNorm (quirk Q1, common.py:53): z -> act -> y, dz = dy * act'(z).
// bf16 only, and only when a branch (pixels x channels) fits LDS -- otherwise DSN_EUNSUPPORTED and the caller keeps the per-branch path.
#include "common.h"

namespace {

constexpr int PP_THREADS = 1024, PP_WAVES = PP_THREADS / 64;
// LDS rows are padded by 16 bytes: an MFMA fragment read takes 16 rows at one column position, and with 64- / 256-byte rows those
// land in the same four banks (16-way conflicts: 12 of the forward kernel's 15.5 us at 288 pixels were this); + 16 bytes rotates
// every row by four banks.  The fp32 accumulator tile gets + 4 floats for the per-pixel-strided reads of the statistics pass.
constexpr int PADE = 8, PADF = 4;

__device__ __forceinline__ f32x4 mma(const bf16_t* a_row, const bf16_t* b_row, f32x4 acc) {
    // D[a-row 4 fg + e][b-row fr] += sum_k A[.][k] B[.][k], k = 8 fg .. 8 fg + 7 of this 32-wide step
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(a_row);
    const bf16x8 b = *reinterpret_cast<const bf16x8*>(b_row);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
}

// Copy a [rows][cols] bf16 matrix (row stride lds elements, 16-byte aligned rows) into a zero-padded [rows_p][cols_p] LDS tile, four
// 16-byte loads in flight per thread (a plain copy loop waits for every load before it issues the next: ~1.5 us per trip here).
__device__ __forceinline__ void stage_tile(bf16_t* dst, int cols_p, int rows_p, const bf16_t* src, int64_t lds, int rows, int cols) {
    const int vpr = cols_p / 8, total = rows_p * vpr, dld = cols_p + PADE;      // (padded destination rows: see PADE)
    for (int base = threadIdx.x; base < total; base += 4 * PP_THREADS) {
        u32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = base + u * PP_THREADS;
            const int r = i / vpr, c8 = (i - r * vpr) * 8;
            v[u] = u32x4{0u, 0u, 0u, 0u};
            if (i < total && r < rows && c8 < cols) v[u] = *reinterpret_cast<const u32x4*>(src + (int64_t)r * lds + c8);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = base + u * PP_THREADS;
            const int r = i / vpr, c8 = (i - r * vpr) * 8;
            if (i < total) *reinterpret_cast<u32x4*>(dst + (size_t)r * dld + c8) = v[u];
        }
    }
}

// LDS layout helpers (all row lengths are multiples of 32 elements -> 64-byte rows, 16-byte aligned fragments)
__device__ __forceinline__ int up(int v, int m) { return (v + m - 1) / m * m; }

__global__ __launch_bounds__(PP_THREADS) void pp_stages_fwd_kernel(const dsn_pp_args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const dsn_pp_stage& s = a.s[blockIdx.x];
    const int P = s.P, C = a.C, Co = a.Co;
    const int Pp = up(P, 16), Cp = up(C, 32), Cop = up(Co, 16);
    const int XL = Cp + PADE, ZL = Cop + PADF;                          // row strides of the bf16 tiles / the fp32 tile
    bf16_t* xs = reinterpret_cast<bf16_t*>(smem);                       // [Pp][XL]
    bf16_t* ws = xs + (size_t)Pp * XL;                                  // [Cop][XL]
    float* zf = reinterpret_cast<float*>(ws + (size_t)Cop * XL);        // [Pp][ZL] fp32 accumulators
    float* sc = zf + (size_t)Pp * ZL;                                   // [Cop] scale, [Cop] shift
    float* sh = sc + Cop;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const bf16_t* x = (const bf16_t*)s.x;
    const bf16_t* w = (const bf16_t*)s.w;
    // ---- stage x and W (zero padding for rows / columns beyond the real sizes)
    stage_tile(xs, Cp, Pp, x, C, P, C);
    stage_tile(ws, Cp, Cop, w, C, Co, C);
    __syncthreads();
    // ---- z = x W^T: tile (mt, nt) = 16 pixels x 16 channels; A = weight rows, B = pixel rows: a lane holds 4 channels of one pixel
    const int tiles_m = Pp / 16, tiles_n = Cop / 16;
    for (int t = wave; t < tiles_m * tiles_n; t += PP_WAVES) {
        const int mt = t / tiles_n, nt = t - mt * tiles_n;
        f32x4 acc{0.f, 0.f, 0.f, 0.f};
        for (int k0 = 0; k0 < Cp; k0 += 32)
            acc = mma(ws + (size_t)(nt * 16 + fr) * XL + k0 + fg * 8, xs + (size_t)(mt * 16 + fr) * XL + k0 + fg * 8, acc);
        *reinterpret_cast<f32x4*>(zf + (size_t)(mt * 16 + fr) * ZL + nt * 16 + fg * 4) = acc;
    }
    __syncthreads();
    // ---- batch statistics (fp64, fixed order): 32 partial sums per channel, then one thread per channel
    if (s.has_bn) {
        // 32 lanes per channel: lane `part` sums pixels part, part + 32, ..; the halves of a wave fold with a fixed shuffle tree
        for (int base = 0; base < Cop; base += PP_THREADS / 32) {
            const int co = base + (tid >> 5), part = tid & 31;
            double s0 = 0.0, s1 = 0.0;
            if (co < Co)
                for (int p = part; p < P; p += 32) {
                    const double v = (double)zf[(size_t)p * ZL + co];
                    s0 += v; s1 += v * v;
                }
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) { s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); }
            if (part == 0 && co < Co) {
                const double count = (double)P;
                const double mean = s0 / count;
                double var = s1 / count - mean * mean;
                if (var < 0.0) var = 0.0;
                const float rstd = (float)(1.0 / sqrt(var + (double)a.eps));
                const float g = s.gamma ? s.gamma[co] : 1.f, b = s.beta ? s.beta[co] : 0.f;
                const float scale = g * rstd, shift = b - (float)mean * scale;
                sc[co] = scale; sh[co] = shift;
                s.stats[co] = scale; s.stats[Co + co] = shift; s.stats[2 * Co + co] = (float)mean; s.stats[3 * Co + co] = rstd;
                if (s.running_mean) {
                    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
                    s.running_mean[co] = (1.f - a.momentum) * s.running_mean[co] + a.momentum * (float)mean;
                    s.running_var[co] = (1.f - a.momentum) * s.running_var[co] + a.momentum * (float)unbiased;
                }
            }
        }
    } else if (tid < Co) {
        sc[tid] = 1.f; sh[tid] = 0.f;
    }
    __syncthreads();
    // ---- z (rounded) and y = act(z * scale + shift)
    bf16_t* z = (bf16_t*)s.z;
    bf16_t* y = (bf16_t*)s.y;
    for (int i = tid; i < P * (Co / 8); i += PP_THREADS) {
        const int p = i / (Co / 8), c8 = (i - p * (Co / 8)) * 8;
        bf16x8 zv;
        float u[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            zv[k] = (bf16_t)zf[(size_t)p * ZL + c8 + k];
            u[k] = (float)zv[k] * sc[c8 + k] + sh[c8 + k];
        }
        apply_act_vec<8>(u, a.act);
        bf16x8 yv;
#pragma unroll
        for (int k = 0; k < 8; ++k) yv[k] = (bf16_t)u[k];
        *reinterpret_cast<bf16x8*>(z + (int64_t)p * s.zld + c8) = zv;
        *reinterpret_cast<bf16x8*>(y + (int64_t)p * s.yld + c8) = yv;
    }
}

// Fragment of a "transposed" MFMA operand: row (lane & 15) of the operand is COLUMN col0 + (lane & 15) of a row-major LDS tile, its 8
// K-values run down the tile's rows k0 + 8 fg .. + 7.  ds_read_b64_tr_b16 (wgrad.hip's pattern): lane 4q + p of a 16-lane group
// addresses row q, columns 4p .. 4p + 3 of a 4 x 16 block and receives column (lane & 15), rows 0 .. 3; two reads = 8 rows.
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 frag_tr(const bf16_t* tile, int ld, int k0, int col0, int fr, int fg) {
    const bf16_t* base = tile + (size_t)(k0 + 8 * fg + (fr >> 2)) * ld + col0 + 4 * (fr & 3);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)base);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 4 * (size_t)ld));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

__global__ __launch_bounds__(PP_THREADS) void pp_stages_bwd_kernel(const dsn_pp_args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const dsn_pp_stage& s = a.s[blockIdx.x];
    const int P = s.P, C = a.C, Co = a.Co;
    const int Pp = up(P, 32), Cp = up(C, 32), Cop = up(Co, 32);
    const int XL = Cp + PADE, OL = Cop + PADE;                          // padded row strides
    bf16_t* xs = reinterpret_cast<bf16_t*>(smem);                       // [Pp][XL]   x, pixel rows
    bf16_t* ws = xs + (size_t)Pp * XL;                                  // [Cop][XL]  W, output-channel rows
    bf16_t* zb = ws + (size_t)Cop * XL;                                 // [Pp][OL]   z
    bf16_t* dyb = zb + (size_t)Pp * OL;                                 // [Pp][OL]   dy
    bf16_t* dzs = dyb + (size_t)Pp * OL;                                // [Pp][OL]   dz
    float* cst = reinterpret_cast<float*>(dzs + (size_t)Pp * OL);       // [4][Cop]: scale, shift, mean, rstd
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const bf16_t* x = (const bf16_t*)s.x;
    const bf16_t* w = (const bf16_t*)s.w;
    const bf16_t* z = (const bf16_t*)s.z;
    const bf16_t* dy = (const bf16_t*)s.dy;
    const float *scale = cst, *shift = cst + Cop, *mean = cst + 2 * Cop, *rstd = cst + 3 * Cop;
    if (s.has_bn)
        for (int i = tid; i < 4 * Co; i += PP_THREADS) cst[(i / Co) * Cop + (i % Co)] = s.stats[i];
    // ---- stage x, W, z, dy with 16-byte copies (zero padding beyond the real rows / columns)
    stage_tile(xs, Cp, Pp, x, C, P, C);
    stage_tile(ws, Cp, Cop, w, C, Co, C);
    stage_tile(zb, Cop, Pp, z, s.zld, P, Co);
    stage_tile(dyb, Cop, Pp, dy, s.dyld, P, Co);
    __syncthreads();
    // ---- one pass per element, values in registers: half-wave h owns channel cbase + h, lane `part` its pixels part, part + 32, ..
    //      g = dy * act'(u) once, the two sums by a fixed shuffle tree, their means broadcast from lane 0 of the half, dz written
    constexpr int MAXK = 16;                                            // P <= 512 (pp_check)
    for (int cbase = 0; cbase < Cop; cbase += PP_THREADS / 32) {
        const int co = cbase + (tid >> 5), part = tid & 31;
        const bool live = co < Co;
        const float scv = live && s.has_bn ? scale[co] : 1.f, shv = live && s.has_bn ? shift[co] : 0.f;
        const float mu = live && s.has_bn ? mean[co] : 0.f, rs = live && s.has_bn ? rstd[co] : 0.f;
        float gk[MAXK], zk[MAXK];
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int k = 0; k < MAXK; ++k) {
            const int p = part + 32 * k;
            gk[k] = 0.f; zk[k] = 0.f;
            if (live && p < P) {
                zk[k] = (float)zb[(size_t)p * OL + co];
                gk[k] = (float)dyb[(size_t)p * OL + co] * act_grad(zk[k] * scv + shv, a.act);
                s0 += (double)gk[k];
                s1 += (double)(gk[k] * ((zk[k] - mu) * rs));
            }
        }
        float kav = 0.f, kbv = 0.f;
        if (s.has_bn) {
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) { s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); }      // (every lane of the half ends with the totals)
            const double count = (double)P;
            const float m1 = (float)(s0 / count), m2 = (float)(s1 / count);
            kav = -scv * rs * m2;
            kbv = -scv * m1;
            if (part == 0 && live) {
                if (s.dbeta) s.dbeta[co] = a.accumulate ? s.dbeta[co] + (float)s0 : (float)s0;
                if (s.dgamma) s.dgamma[co] = a.accumulate ? s.dgamma[co] + (float)s1 : (float)s1;
            }
        }
#pragma unroll
        for (int k = 0; k < MAXK; ++k) {
            const int p = part + 32 * k;
            if (co < Cop && p < Pp) {            // padding rows / columns are zeros (gk = 0 there, and kb must not leak into them)
                float dzv = gk[k];
                if (s.has_bn && live && p < P) dzv = scv * dzv + (kav * (zk[k] - mu) + kbv);
                dzs[(size_t)p * OL + co] = (bf16_t)dzv;
            }
        }
    }
    __syncthreads();
    // ---- dx[p][c] = sum_co dz[p][co] W[co][c]: A = W "transposed" (rows = input channels, K = output channels down the rows of ws),
    //      B = dz rows (pixels): a lane holds 4 consecutive input channels of one pixel
    bf16_t* dx = (bf16_t*)s.dx;
    {
        const int tiles_m = up(P, 16) / 16, tiles_n = up(C, 16) / 16;
        for (int t = wave; t < tiles_m * tiles_n; t += PP_WAVES) {
            const int mt = t / tiles_n, nt = t - mt * tiles_n;
            f32x4 acc{0.f, 0.f, 0.f, 0.f};
            for (int k0 = 0; k0 < Cop; k0 += 32) {
                const bf16x8 av = frag_tr(ws, XL, k0, nt * 16, fr, fg);
                const bf16x8 bv = *reinterpret_cast<const bf16x8*>(dzs + (size_t)(mt * 16 + fr) * OL + k0 + fg * 8);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc, 0, 0, 0);
            }
            const int p = mt * 16 + fr, c = nt * 16 + fg * 4;
            if (p < P && c < C) {
                bf16_t o[4] = {(bf16_t)acc[0], (bf16_t)acc[1], (bf16_t)acc[2], (bf16_t)acc[3]};
                *reinterpret_cast<uint64_t*>(dx + (int64_t)p * C + c) = *reinterpret_cast<uint64_t*>(o);
            }
        }
    }
    // ---- dW[co][c] (+)= sum_p dz[p][co] x[p][c]: K = pixels, both operands gathered down the rows of their pixel-major tiles
    if (s.dw) {
        const int tiles_m = up(C, 16) / 16, tiles_n = up(Co, 16) / 16;         // (m: input-channel tiles = B rows, n: output-channel tiles = A rows)
        for (int t = wave; t < tiles_m * tiles_n; t += PP_WAVES) {
            const int mt = t / tiles_n, nt = t - mt * tiles_n;
            f32x4 acc{0.f, 0.f, 0.f, 0.f};
            for (int k0 = 0; k0 < Pp; k0 += 32) {
                const bf16x8 av = frag_tr(dzs, OL, k0, nt * 16, fr, fg);
                const bf16x8 bv = frag_tr(xs, XL, k0, mt * 16, fr, fg);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc, 0, 0, 0);
            }
            const int c = mt * 16 + fr;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int co = nt * 16 + fg * 4 + e;
                if (co < Co && c < C) {
                    float* d = s.dw + (int64_t)co * C + c;
                    *d = a.accumulate ? *d + acc[e] : acc[e];
                }
            }
        }
    }
}

size_t pp_fwd_lds(int P, int C, int Co) {
    const int Pp = (P + 15) / 16 * 16, Cp = (C + 31) / 32 * 32, Cop = (Co + 15) / 16 * 16;
    return ((size_t)Pp + Cop) * (Cp + PADE) * 2 + (size_t)Pp * (Cop + PADF) * 4 + 2 * Cop * 4;
}
size_t pp_bwd_lds(int P, int C, int Co) {
    const int Pp = (P + 31) / 32 * 32, Cp = (C + 31) / 32 * 32, Cop = (Co + 31) / 32 * 32;
    return (((size_t)Pp + Cop) * (Cp + PADE) + 3 * (size_t)Pp * (Cop + PADE)) * 2 + 4 * Cop * 4;
}

int pp_check(const dsn_pp_args* a, bool bwd, size_t* lds_out) {
    DSN_CHECK_ARG(a && a->nstage >= 1 && a->nstage <= DSN_PP_MAXSTAGE && a->C > 0 && a->Co > 0, "pp_stages: bad arguments");
    if (a->dtype != DSN_BF16 || a->C % 8 != 0 || a->Co % 8 != 0) DSN_FAIL(DSN_EUNSUPPORTED, "pp_stages: bf16 with 16-byte channel vectors only");
    size_t lds = 0;
    for (int j = 0; j < a->nstage; ++j) {
        const dsn_pp_stage& s = a->s[j];
        DSN_CHECK_ARG(s.P > 0 && s.x && s.w && s.z, "pp_stages: null operand in branch %d", j);
        if (s.P > 512) DSN_FAIL(DSN_EUNSUPPORTED, "pp_stages: more than 512 pixels in a branch");
        DSN_CHECK_ARG(!s.has_bn || s.stats, "pp_stages: branch %d has BatchNorm but no statistics buffer", j);
        if (((uintptr_t)s.x | (uintptr_t)s.w | (uintptr_t)s.z) % 16 != 0 || s.zld % 8 != 0) DSN_FAIL(DSN_EUNSUPPORTED, "pp_stages: unaligned operand");
        if (bwd) {
            DSN_CHECK_ARG(s.dy && s.dx, "pp_stages_bwd: null gradient operand in branch %d", j);
            if (((uintptr_t)s.dx) % 8 != 0) DSN_FAIL(DSN_EUNSUPPORTED, "pp_stages_bwd: unaligned dx");
        } else {
            DSN_CHECK_ARG(s.y, "pp_stages_fwd: null output in branch %d", j);
            if (((uintptr_t)s.y) % 16 != 0 || s.yld % 8 != 0) DSN_FAIL(DSN_EUNSUPPORTED, "pp_stages_fwd: unaligned y");
        }
        const size_t l = bwd ? pp_bwd_lds(s.P, a->C, a->Co) : pp_fwd_lds(s.P, a->C, a->Co);
        lds = l > lds ? l : lds;
    }
    if (lds > 160 * 1024) DSN_FAIL(DSN_EUNSUPPORTED, "pp_stages: a branch needs %zu bytes of LDS", lds);
    *lds_out = lds;
    return DSN_OK;
}

}  // namespace

extern "C" int dsn_pp_stages_supported(int32_t max_pixels, int32_t c, int32_t co, int32_t dtype) {
    if (dtype != DSN_BF16 || c % 8 != 0 || co % 8 != 0 || max_pixels <= 0 || max_pixels > 512) return 0;
    return pp_fwd_lds(max_pixels, c, co) <= 160 * 1024 && pp_bwd_lds(max_pixels, c, co) <= 160 * 1024;
}

extern "C" int dsn_pp_stages_fwd(const dsn_pp_args* a, void* stream) {
    size_t lds = 0;
    const int rc = pp_check(a, false, &lds);
    if (rc) return rc;
    DSN_LDS_ATTR(pp_stages_fwd_kernel, 160 * 1024);
    hipLaunchKernelGGL(pp_stages_fwd_kernel, dim3(a->nstage), dim3(PP_THREADS), lds, (hipStream_t)stream, *a);
    DSN_LAUNCH_CHECK("pp_stages_fwd");
    return DSN_OK;
}

extern "C" int dsn_pp_stages_bwd(const dsn_pp_args* a, void* stream) {
    size_t lds = 0;
    const int rc = pp_check(a, true, &lds);
    if (rc) return rc;
    DSN_LDS_ATTR(pp_stages_bwd_kernel, 160 * 1024);
    hipLaunchKernelGGL(pp_stages_bwd_kernel, dim3(a->nstage), dim3(PP_THREADS), lds, (hipStream_t)stream, *a);
    DSN_LAUNCH_CHECK("pp_stages_bwd");
    return DSN_OK;
}
