// Weight gradient of a convolution on MFMA (gfx950):  dW[co][tap][ci] = sum_p dy[p][co] * x[src(p, tap)][ci]
//
// Per tap this is a GEMM whose REDUCTION dimension is the pixel index p (up to N*160*160 = 204,800 at batch 8) and whose
// output is tiny (Co x Ci).  One workgroup owns a 64(co) x 64(ci) tile of ONE tap over ONE contiguous pixel range
// (split-K); partial tiles go to fp32 slabs that a second kernel adds up in a fixed order -- deterministic, no float
// atomics.  Both operands are pixel-major in HBM (NHWC), i.e. "K-strided" for MFMA; the 32-pixel chunks are staged
// row-major [pixel][channel] in LDS with coalesced 16-byte loads and transposed on the way OUT of LDS:
//   fp32 : v_mfma_f32_16x16x4_f32 takes one scalar per lane  -> plain ds_read_b32 from a padded image
//   bf16 : v_mfma_f32_16x16x32_bf16 takes 8 K-contiguous bf16 -> two ds_read_b64_tr_b16 (hardware transpose) per fragment
#include <stdlib.h>

#include <algorithm>

#include "common.h"

namespace {

struct WGeom {
    int32_t P, Ho, Wo, Hi, Wi, Co, Ci, Cip, KH, KW, stride, pad, dil;
    int64_t yld, xld;
    int32_t S, ppb, tiles_co, tiles_ci;
    int32_t accumulate;
    int32_t CiLoad;       // channels x actually holds (>= Ci: zero-padded tail may be loaded, never stored)
    int32_t oihw;         // final output layout: 0 = packed [Co][tap][Cip], 1 = OIHW [Co][Ci][tap]
    uint32_t x_bytes, dy_bytes;   // buffer-descriptor ranges of the vector-load paths (both tensors < 2 GiB there)
    int32_t npx, npy;             // halo-tile kernel: 4 x 8-pixel patches per image row / column; P and ppb then count PATCHES
};

constexpr int TB = 64;    // tile edge (co and ci)

template <typename T> struct WTraits;
template <> struct WTraits<float> {
    static constexpr int VEC = 4, ROW = TB + 16;   // padded row (floats): lanes l and l+16 land 16 banks apart
};
template <> struct WTraits<bf16_t> {
    static constexpr int VEC = 8, ROW = TB;        // 128-byte rows for the transposing read
};

// bf16 tiles are [pixel row][64 channels] = 128-byte rows.  A transposing read (ds_read_b64_tr_b16) touches, per 32-lane
// half, 8 rows x 32 bytes; linear rows would put the 4 rows of equal parity on the same 8 banks (4-way conflict).  XOR-ing
// the 32-byte column group with swz(row) spreads them over the four groups of the 256-byte bank row: conflict-free.
__device__ __forceinline__ int swz(int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) << 1); }

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

template <typename T, bool VECLOAD, int PK>
__device__ __forceinline__ void wgrad_body(const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ out,
                                           const WGeom& g, int bid) {
    constexpr int VEC = WTraits<T>::VEC, ROW = WTraits<T>::ROW;
    constexpr int VPR = TB / VEC;          // vectors per tile row
    constexpr int NV = VPR / 8;            // vectors per thread per row (8 threads per row)
    constexpr int RG = PK / 32;            // row groups: thread (srow, sv) stages rows srow + 32*rg
    __shared__ __attribute__((aligned(16))) T sA[2][PK * ROW];
    __shared__ __attribute__((aligned(16))) T sB[2][PK * ROW];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fg = lane >> 4;
    const int taps = g.KH * g.KW;
    const int tci = bid % g.tiles_ci; bid /= g.tiles_ci;
    const int tco = bid % g.tiles_co; bid /= g.tiles_co;
    const int tap = bid % taps;
    const int split = bid / taps;
    const int ky = tap / g.KW, kx = tap - ky * g.KW;
    const int co0 = tco * TB, ci0 = tci * TB;
    const int p_begin = split * g.ppb;
    const int p_end = (p_begin + g.ppb < g.P) ? p_begin + g.ppb : g.P;
    const int nchunks = (p_end - p_begin + PK - 1) / PK;

    const int srow = tid >> 3;      // staged pixel row 0..31
    const int sv = tid & 7;         // first vector of the row
    u32x4 ra[RG][NV], rb[RG][NV];

    // Vector path: raw buffer loads with 32-bit byte offsets (masked lanes get an out-of-range offset and read 0) and a
    // pixel cursor (p, ox, oy, n) per staged row that ADVANCES by PK per chunk instead of being re-derived with two integer
    // divisions -- this kernel was VALU-bound on address arithmetic (SQ_INSTS_VALU: ~140 per chunk per wave for 4 MFMAs).
    constexpr uint32_t OOB = 0xFFFFFFF0u;
    constexpr int ES = (int)sizeof(T);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, VECLOAD ? g.x_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, VECLOAD ? g.dy_bytes : 0, 0x00020000);
    // (32-bit integer multiplies are quarter-rate: the offsets below ADVANCE by a constant per chunk.  For the same-size
    // stride-1 convs -- all but six layers -- the input pixel of tap (ky,kx) is output pixel + a constant, so the x offset
    // advances the same way and only the validity mask needs the cursor.)
    const bool lin = g.stride == 1 && g.Hi == g.Ho && g.Wi == g.Wo;
    int cp[RG], cox[RG], coy[RG], cn[RG];
    uint32_t aoff[RG], boff[RG];
    const int tap_dy = ky * g.dil - g.pad, tap_dx = kx * g.dil - g.pad;
    const uint32_t astep = (uint32_t)(PK * (int)g.yld) * ES, bstep = (uint32_t)(PK * (int)g.xld) * ES;
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) {
        cp[rg] = p_begin + srow + 32 * rg;
        const int pc = cp[rg] < g.P ? cp[rg] : g.P - 1;
        cox[rg] = pc % g.Wo;
        const int t = pc / g.Wo;
        coy[rg] = t % g.Ho;
        cn[rg] = t / g.Ho;
        aoff[rg] = (uint32_t)(cp[rg] * (int)g.yld + co0) * ES;
        boff[rg] = (uint32_t)((cp[rg] + tap_dy * g.Wi + tap_dx) * (int)g.xld + ci0) * ES;     // used when lin
    }
    auto load_vec = [&](const T* p, int c, int climit) -> u32x4 {
        T tmp[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) tmp[e] = (c + e < climit) ? p[e] : from_f32<T>(0.f);
        return *reinterpret_cast<u32x4*>(tmp);
    };
    auto load_chunk = [&](int ch) {      // called with ch = 0, 1, 2, ... in order (the cursor advances)
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) {
            if constexpr (VECLOAD) {
                const int p = cp[rg];
                const bool pok = p < p_end;
                const int iy = coy[rg] * g.stride + tap_dy, ix = cox[rg] * g.stride + tap_dx;
                const bool xok = pok && (unsigned)iy < (unsigned)g.Hi && (unsigned)ix < (unsigned)g.Wi;
                const uint32_t abase = aoff[rg];
                const uint32_t bbase = lin ? boff[rg] : (uint32_t)(((cn[rg] * g.Hi + iy) * g.Wi + ix) * (int)g.xld + ci0) * ES;
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const int vc = (sv + 8 * i) * VEC;
                    ra[rg][i] = __builtin_amdgcn_raw_buffer_load_b128(yr, (pok && co0 + vc < g.Co) ? abase + vc * ES : OOB, 0, 0);
                    rb[rg][i] = __builtin_amdgcn_raw_buffer_load_b128(xr, (xok && ci0 + vc < g.CiLoad) ? bbase + vc * ES : OOB, 0, 0);
                }
                aoff[rg] += astep;
                boff[rg] += bstep;
                cp[rg] = p + PK;
                cox[rg] += PK;
                while (cox[rg] >= g.Wo) {
                    cox[rg] -= g.Wo;
                    if (++coy[rg] >= g.Ho) { coy[rg] = 0; ++cn[rg]; }
                }
            } else {
                const int p = p_begin + ch * PK + srow + 32 * rg;
                const bool pok = p < p_end;
                int ox = 0, oy = 0, n = 0;
                if (pok) {
                    ox = p % g.Wo;
                    const int t = p / g.Wo;
                    oy = t % g.Ho;
                    n = t / g.Ho;
                }
                const int iy = oy * g.stride - g.pad + ky * g.dil, ix = ox * g.stride - g.pad + kx * g.dil;
                const bool xok = pok && iy >= 0 && iy < g.Hi && ix >= 0 && ix < g.Wi;
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const int vc = (sv + 8 * i) * VEC;
                    ra[rg][i] = u32x4{0u, 0u, 0u, 0u};
                    rb[rg][i] = u32x4{0u, 0u, 0u, 0u};
                    if (pok) ra[rg][i] = load_vec(dy + (int64_t)p * g.yld + co0 + vc, co0 + vc, g.Co);
                    if (xok)
                        rb[rg][i] = load_vec(x + (((int64_t)n * g.Hi + iy) * g.Wi + ix) * g.xld + ci0 + vc, ci0 + vc, g.CiLoad);
                }
            }
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int rg = 0; rg < RG; ++rg)
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                int vc = (sv + 8 * i) * VEC;
                if constexpr (sizeof(T) == 2) vc ^= swz(srow) << 4;
                *reinterpret_cast<u32x4*>(&sA[buf][(srow + 32 * rg) * ROW + vc]) = ra[rg][i];
                *reinterpret_cast<u32x4*>(&sB[buf][(srow + 32 * rg) * ROW + vc]) = rb[rg][i];
            }
    };

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int buf) {
        if constexpr (sizeof(T) == 4) {
            const float* A = reinterpret_cast<const float*>(sA[buf]);
            const float* B = reinterpret_cast<const float*>(sB[buf]);
#pragma unroll
            for (int s = 0; s < PK / 4; ++s) {
                float a[2], b[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) a[i] = A[(4 * s + fg) * ROW + (wm * 2 + i) * 16 + fr];
#pragma unroll
                for (int j = 0; j < 2; ++j) b[j] = B[(4 * s + fg) * ROW + (wn * 2 + j) * 16 + fr];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        } else {
            // lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3 of a 4x16 block; it receives column (lane&15),
            // rows 0..3.  Block rows = pixels 8*fg + {0..3} then {4..7}; block columns = the fragment's 16 channels.
            const int q = fr >> 2, pp = fr & 3;
            const int fz = swz(8 * fg + q);
#pragma unroll
            for (int ks = 0; ks < PK / 32; ++ks) {
                bf16x8 a[2], b[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const bf16_t* base = &sA[buf][(32 * ks + 8 * fg + q) * ROW + (((wm * 2 + i) ^ fz) << 4) + 4 * pp];
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)base);
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(base + 4 * ROW));
                    a[i] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const bf16_t* base = &sB[buf][(32 * ks + 8 * fg + q) * ROW + (((wn * 2 + j) ^ fz) << 4) + 4 * pp];
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)base);
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(base + 4 * ROW));
                    b[j] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
    };

    if (nchunks > 0) {
        load_chunk(0);
        store_chunk(0);
    }
    __syncthreads();
    for (int it = 0; it < nchunks; ++it) {
        const int buf = it & 1;
        const bool more = it + 1 < nchunks;
        if (more) load_chunk(it + 1);
        compute(buf);
        if (more) store_chunk(buf ^ 1);
        __syncthreads();
    }

    // D[row = co][col = ci]
    float* o = out + (g.S > 1 ? (int64_t)split * g.Co * taps * g.Cip : 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int co = co0 + (wm * 2 + i) * 16 + fg * 4 + e;
                const int ci = ci0 + (wn * 2 + j) * 16 + fr;
                if (co < g.Co && ci < g.Ci) {
                    float* d = (g.S == 1 && g.oihw) ? o + ((int64_t)co * g.Ci + ci) * taps + tap
                                                    : o + ((int64_t)co * taps + tap) * g.Cip + ci;
                    float v = acc[i][j][e];
                    if (g.S == 1 && g.accumulate) v += *d;
                    *d = v;
                }
            }
}


// ---- 128 x 128 tile (bf16, 16-byte paths): the per-tap kernel for layers whose weight matrix is at least that large -------------
// The 64 x 64 tile runs 4 MFMAs per wave between two barriers and reads 8 transposed fragments halves for them -- it is bound by
// LDS and barriers, not by MFMA (7-12 % of the bf16 rate on config 5's layers), and a Co x Ci = 256 x 256 layer reads each operand
// four times.  Here a wave owns 64(co) x 64(ci): 16 MFMAs per 32-pixel chunk for 16 fragment reads, each operand is read half as
// often, and a block is 4x fewer work items for the same layer.  Rows are [pixel][128 channels] = 256 bytes: every row starts at
// bank 0, so the 32-byte column group is XOR-ed with swz128(row) -- a bijection of the 8 rows a transposing read touches per
// 32-lane half (rows 8*fg + q: bits 0, 1, 3) onto the 8 groups, with bit 0 of the row in the TOP bit so that the two rows of a
// 16-lane store pass land in different halves of the bank row.
constexpr int TB2 = 128;
__device__ __forceinline__ int swz128(int row) { return ((row & 1) << 2) | ((row >> 1) & 1) | (((row >> 3) & 1) << 1); }

template <int PK>
__device__ __forceinline__ void wgrad_body128(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, float* __restrict__ out,
                                              const WGeom& g, int bid) {
    constexpr int ROW = TB2, RG = PK / 32, NV = 2;
    __shared__ __attribute__((aligned(16))) bf16_t sA[2][PK * ROW];
    __shared__ __attribute__((aligned(16))) bf16_t sB[2][PK * ROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fg = lane >> 4;
    const int taps = g.KH * g.KW;
    const int tci = bid % g.tiles_ci; bid /= g.tiles_ci;
    const int tco = bid % g.tiles_co; bid /= g.tiles_co;
    const int tap = bid % taps;
    const int split = bid / taps;
    const int ky = tap / g.KW, kx = tap - ky * g.KW;
    const int co0 = tco * TB2, ci0 = tci * TB2;
    const int p_begin = split * g.ppb;
    const int p_end = (p_begin + g.ppb < g.P) ? p_begin + g.ppb : g.P;
    const int nchunks = (p_end - p_begin + PK - 1) / PK;
    const int srow = tid >> 3, sv = tid & 7;        // staged pixel row, first of the thread's two 16-byte vectors (sv, sv + 8)
    u32x4 ra[RG][NV], rb[RG][NV];
    constexpr uint32_t OOB = 0xFFFFFFF0u;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, g.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, g.dy_bytes, 0x00020000);
    const bool lin = g.stride == 1 && g.Hi == g.Ho && g.Wi == g.Wo;
    int cp[RG], cox[RG], coy[RG], cn[RG];
    uint32_t aoff[RG], boff[RG];
    const int tap_dy = ky * g.dil - g.pad, tap_dx = kx * g.dil - g.pad;
    const uint32_t astep = (uint32_t)(PK * (int)g.yld) * 2, bstep = (uint32_t)(PK * (int)g.xld) * 2;
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) {
        cp[rg] = p_begin + srow + 32 * rg;
        const int pc = cp[rg] < g.P ? cp[rg] : g.P - 1;
        cox[rg] = pc % g.Wo;
        const int t = pc / g.Wo;
        coy[rg] = t % g.Ho;
        cn[rg] = t / g.Ho;
        aoff[rg] = (uint32_t)(cp[rg] * (int)g.yld + co0) * 2;
        boff[rg] = (uint32_t)((cp[rg] + tap_dy * g.Wi + tap_dx) * (int)g.xld + ci0) * 2;     // used when lin
    }
    auto load_chunk = [&]() {            // (the cursor advances: called once per chunk, in order)
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) {
            const int p = cp[rg];
            const bool pok = p < p_end;
            const int iy = coy[rg] * g.stride + tap_dy, ix = cox[rg] * g.stride + tap_dx;
            const bool xok = pok && (unsigned)iy < (unsigned)g.Hi && (unsigned)ix < (unsigned)g.Wi;
            const uint32_t abase = aoff[rg];
            const uint32_t bbase = lin ? boff[rg] : (uint32_t)(((cn[rg] * g.Hi + iy) * g.Wi + ix) * (int)g.xld + ci0) * 2;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int vc = (sv + 8 * i) * 8;
                ra[rg][i] = __builtin_amdgcn_raw_buffer_load_b128(yr, (pok && co0 + vc < g.Co) ? abase + vc * 2 : OOB, 0, 0);
                rb[rg][i] = __builtin_amdgcn_raw_buffer_load_b128(xr, (xok && ci0 + vc < g.CiLoad) ? bbase + vc * 2 : OOB, 0, 0);
            }
            aoff[rg] += astep;
            boff[rg] += bstep;
            cp[rg] = p + PK;
            cox[rg] += PK;
            while (cox[rg] >= g.Wo) {
                cox[rg] -= g.Wo;
                if (++coy[rg] >= g.Ho) { coy[rg] = 0; ++cn[rg]; }
            }
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int rg = 0; rg < RG; ++rg)
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int v = sv + 8 * i;                                   // vector of the row: 32-byte group v >> 1, half v & 1
                const int col = ((((v >> 1) ^ swz128(srow)) << 1) | (v & 1)) << 3;
                *reinterpret_cast<u32x4*>(&sA[buf][(srow + 32 * rg) * ROW + col]) = ra[rg][i];
                *reinterpret_cast<u32x4*>(&sB[buf][(srow + 32 * rg) * ROW + col]) = rb[rg][i];
            }
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto compute = [&](int buf) {
        const int q = fr >> 2, pp = fr & 3;
        const int fz = swz128(8 * fg + q);
#pragma unroll
        for (int ks = 0; ks < PK / 32; ++ks) {
            bf16x8 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bf16_t* base = &sA[buf][(32 * ks + 8 * fg + q) * ROW + (((wm * 4 + i) ^ fz) << 4) + 4 * pp];
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)base);
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 4 * ROW));
                a[i] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bf16_t* base = &sB[buf][(32 * ks + 8 * fg + q) * ROW + (((wn * 4 + j) ^ fz) << 4) + 4 * pp];
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)base);
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 4 * ROW));
                b[j] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };
    if (nchunks > 0) {
        load_chunk();
        store_chunk(0);
    }
    __syncthreads();
    for (int it = 0; it < nchunks; ++it) {
        const int buf = it & 1;
        const bool more = it + 1 < nchunks;
        if (more) load_chunk();
        compute(buf);
        if (more) store_chunk(buf ^ 1);
        __syncthreads();
    }
    float* o = out + (g.S > 1 ? (int64_t)split * g.Co * taps * g.Cip : 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int co = co0 + (wm * 4 + i) * 16 + fg * 4 + e;
                const int ci = ci0 + (wn * 4 + j) * 16 + fr;
                if (co < g.Co && ci < g.Ci) {
                    float* d = (g.S == 1 && g.oihw) ? o + ((int64_t)co * g.Ci + ci) * taps + tap
                                                    : o + ((int64_t)co * taps + tap) * g.Cip + ci;
                    float v = acc[i][j][e];
                    if (g.S == 1 && g.accumulate) v += *d;
                    *d = v;
                }
            }
}

// (Measured and dropped: the same tile staged with `buffer_load ... lds` into a three-stage ring, two chunks ahead, counted waits --
//  bit-identical, and 0.3-0.5 % SLOWER on both configurations.  These kernels are not bound by the latency of their own loads: with
//  ~770 resident blocks each pulling 16 KB per ~1 us chunk they sit at the aggregate L2 bandwidth, which only fewer re-reads --
//  larger tiles -- relieve.)

// ---- all-taps variant (bf16, 3x3): one block owns a 64(co) x 64(ci) tile of ALL nine taps over its pixel range ------------
// The per-tap kernel above re-reads dy once per tap (9x: 472 MB instead of 52 MB for the Focus conv -- it ran at the HBM
// rate of the re-reads).  Here dy is staged once per chunk and shared by the nine taps, the nine shifted x gathers hit the
// same cache lines, 10 16-byte loads are in flight per thread, and every barrier is followed by 36 MFMAs per wave.
template <int NT>
__device__ __forceinline__ void wgrad_alltaps_body(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                   float* __restrict__ out, const WGeom& g, int bid) {
    constexpr int PKA = 32, ROW = TB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* sA = reinterpret_cast<bf16_t*>(smem_raw);                    // [2][PKA*ROW]
    bf16_t* sB = sA + 2 * PKA * ROW;                                     // [2][NT][PKA*ROW]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fg = lane >> 4;
    const int tci = bid % g.tiles_ci; bid /= g.tiles_ci;
    const int tco = bid % g.tiles_co;
    const int split = bid / g.tiles_co;
    const int co0 = tco * TB, ci0 = tci * TB;
    const int p_begin = split * g.ppb;
    const int p_end = (p_begin + g.ppb < g.P) ? p_begin + g.ppb : g.P;
    const int nchunks = (p_end - p_begin + PKA - 1) / PKA;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, g.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, g.dy_bytes, 0x00020000);
    constexpr uint32_t OOB = 0xFFFFFFF0u;

    const int srow = tid >> 3, sv = tid & 7;     // staged pixel row, 16-byte vector (8 channels) of the row
    const int vc = sv * 8;
    u32x4 ra, rb[NT];

    // pixel cursor + offsets that advance by a constant per chunk (no integer divisions or multiplies in the loop; see the
    // per-tap kernel).  Same-size stride-1 convs: input pixel of tap (ky,kx) = output pixel + (ky*dil-pad)*Wi + (kx*dil-pad).
    const bool lin = g.stride == 1 && g.Hi == g.Ho && g.Wi == g.Wo;
    int cp = p_begin + srow, cox, coy, cn;
    {
        const int pc = cp < g.P ? cp : g.P - 1;
        cox = pc % g.Wo;
        const int t = pc / g.Wo;
        coy = t % g.Ho;
        cn = t / g.Ho;
    }
    uint32_t aoff = (uint32_t)(cp * (int)g.yld + co0 + vc) * 2u;
    uint32_t boff = (uint32_t)((cp - g.pad * g.Wi - g.pad) * (int)g.xld + ci0 + vc) * 2u;     // tap (0,0), lin only
    const uint32_t astep = (uint32_t)(PKA * (int)g.yld) * 2u, bstep = (uint32_t)(PKA * (int)g.xld) * 2u;
    const uint32_t brow = (uint32_t)(g.dil * g.Wi * (int)g.xld) * 2u, bcol = (uint32_t)(g.dil * (int)g.xld) * 2u;
    const bool aok = co0 + vc < g.Co, bok = ci0 + vc < g.CiLoad;

    auto load_chunk = [&](int) {           // called for chunk 0, 1, 2, ... in order
        const bool pok = cp < p_end;
        ra = __builtin_amdgcn_raw_buffer_load_b128(yr, (pok && aok) ? aoff : OOB, 0, 0);
        const int iy0 = coy * g.stride - g.pad, ix0 = cox * g.stride - g.pad;
        const int nb = cn * g.Hi;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int ky = t / 3, kx = t - ky * 3;        // NT == KH*KW with KW == 3 (checked on the host)
            const int iy = iy0 + ky * g.dil, ix = ix0 + kx * g.dil;
            const bool ok = pok && bok && (unsigned)iy < (unsigned)g.Hi && (unsigned)ix < (unsigned)g.Wi;
            const uint32_t o = lin ? boff + (uint32_t)ky * brow + (uint32_t)kx * bcol
                                   : (uint32_t)(((nb + iy) * g.Wi + ix) * (int)g.xld + ci0 + vc) * 2u;
            rb[t] = __builtin_amdgcn_raw_buffer_load_b128(xr, ok ? o : OOB, 0, 0);
        }
        cp += PKA;
        aoff += astep;
        boff += bstep;
        cox += PKA;
        while (cox >= g.Wo) {
            cox -= g.Wo;
            if (++coy >= g.Ho) { coy = 0; ++cn; }
        }
    };
    auto store_chunk = [&](int buf) {
        const int vs = vc ^ (swz(srow) << 4);
        *reinterpret_cast<u32x4*>(&sA[(buf * PKA + srow) * ROW + vs]) = ra;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            *reinterpret_cast<u32x4*>(&sB[((buf * NT + t) * PKA + srow) * ROW + vs]) = rb[t];
        }
    };

    f32x4 acc[NT][2][2];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto tr_frag = [&](const bf16_t* base) -> bf16x8 {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)base);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 4 * ROW));
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto compute = [&](int buf) {
        const int q = fr >> 2, pp = fr & 3;
        const int fz = swz(8 * fg + q);
        bf16x8 a[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = tr_frag(&sA[(buf * PKA + 8 * fg + q) * ROW + (((wm * 2 + i) ^ fz) << 4) + 4 * pp]);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            bf16x8 b[2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
                b[j] = tr_frag(&sB[((buf * NT + t) * PKA + 8 * fg + q) * ROW + (((wn * 2 + j) ^ fz) << 4) + 4 * pp]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[t][i][j], 0, 0, 0);
        }
    };

    if (nchunks > 0) {
        load_chunk(0);
        store_chunk(0);
    }
    __syncthreads();
    for (int it = 0; it < nchunks; ++it) {
        const int buf = it & 1;
        const bool more = it + 1 < nchunks;
        if (more) load_chunk(it + 1);
        compute(buf);
        if (more) store_chunk(buf ^ 1);
        __syncthreads();
    }

    float* o = out + (g.S > 1 ? (int64_t)split * g.Co * NT * g.Cip : 0);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int co = co0 + (wm * 2 + i) * 16 + fg * 4 + e;
                    const int ci = ci0 + (wn * 2 + j) * 16 + fr;
                    if (co < g.Co && ci < g.Ci) {
                        float* d = (g.S == 1 && g.oihw) ? o + ((int64_t)co * g.Ci + ci) * NT + t
                                                        : o + ((int64_t)co * NT + t) * g.Cip + ci;
                        float v = acc[t][i][j][e];
                        if (g.S == 1 && g.accumulate) v += *d;
                        *d = v;
                    }
                }
}

// ---- halo-tile all-taps variant (bf16, same-size 3x3 / stride 1 / pad = dilation = 1..3) ------------------------------------------
// The all-taps kernel above stages NINE shifted copies of the x tile per 32-pixel chunk (10 16-byte loads per thread, 80 KB of LDS:
// two blocks per CU).  Here a chunk is a 4 x 8 PATCH of output pixels and x is staged once with its halo ((4 + 2d) x (8 + 2d)
// pixels): tap (ky, kx) of patch pixel (r, c) is halo pixel (r + ky d, c + kx d), and because every lane of a transposing read
// supplies the address of its own row, the nine taps are nine address offsets into the same tile.  3 loads per thread per chunk at
// d = 1 (6 at d = 3), 48 KB of LDS.  Halo rows are laid out with a pitch of 16 LDS rows (at most 14 used) so that the 8 rows a
// 32-lane half reads -- (fg + ky d) * 16 + kx d + q for two values of fg and q = 0..3 -- differ in (row parity, swzh(row)):
// conflict-free for every tap.
constexpr int HPITCH = 16, HROWS = 10, HLOADS = 5;      // d <= 3: 10 x 14 halo pixels x 8 vectors <= 5 x 256
__device__ __forceinline__ int swzh(int row) { return ((row >> 1) & 1) | (((row >> 4) & 1) << 1); }

// S = 2 (3x3 / stride 2 / pad 1, the stem and downsampling layers): the patch's taps reach (2 * 3 + 3) x (2 * 7 + 3) = 9 x 17 input
// pixels.  They are stored with a pitch of 18 rows plus ONE extra row for every second pair of halo rows, so that the rows a
// 32-lane half reads -- hr = 2 fg + ky, hc = 2 q + kx: four same-parity rows per fg -- fall on opposite halves of the bank row for
// the two values of fg, and (row >> 1) & 3 spreads the four of one fg over the four 32-byte groups.
template <int S> __device__ __forceinline__ int halo_row(int hr, int hc) {
    return S == 1 ? hr * HPITCH + hc : hr * 18 + hc + ((hr >> 1) & 1);
}
template <int S> __device__ __forceinline__ int halo_swz(int row) { return S == 1 ? swzh(row) : (row >> 1) & 3; }

constexpr int HALO_A = 32 * TB, HALO_B = (HROWS * HPITCH + 8) * TB;      // elements per stage of the dy tile / the halo tile (stride 2: 8 * 18 + 16 + 1 rows)
template <int S>
__device__ __forceinline__ void wgrad_halo_body(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, float* __restrict__ out,
                                                const WGeom& g, int bid, bf16_t (*sA)[HALO_A], bf16_t (*sB)[HALO_B]) {
    constexpr int ROW = TB, NT = 9;        // (the LDS stages belong to the kernel: both strides share one allocation)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fg = lane >> 4;
    const int tci = bid % g.tiles_ci; bid /= g.tiles_ci;
    const int tco = bid % g.tiles_co;
    const int split = bid / g.tiles_co;
    const int co0 = tco * TB, ci0 = tci * TB;
    const int p_begin = split * g.ppb;                                  // (patches)
    const int p_end = (p_begin + g.ppb < g.P) ? p_begin + g.ppb : g.P;
    const int nchunks = p_end - p_begin;
    const int D = g.dil, HWd = S == 1 ? 8 + 2 * D : 17, NHP = (S == 1 ? 4 + 2 * D : 9) * HWd;      // halo width, halo pixels
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, g.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, g.dy_bytes, 0x00020000);
    constexpr uint32_t OOB = 0xFFFFFFF0u;
    // staging plan: dy vector (pixel k = tid >> 3 of the patch, channels 8 * (tid & 7) ..), x vectors idx = tid + 256 i of the halo
    // (idx >> 3 = halo pixel, idx & 7 = vector)
    const int v8 = (tid & 7) * 8;
    const int ak = tid >> 3, ar = ak >> 3, ac = ak & 7;
    int hr[HLOADS], hc[HLOADS], hrow[HLOADS];
    bool hon[HLOADS];
#pragma unroll
    for (int i = 0; i < HLOADS; ++i) {
        const int hp = (tid + 256 * i) >> 3;
        hon[i] = hp < NHP;
        hr[i] = hp / HWd;
        hc[i] = hp - hr[i] * HWd;
        hrow[i] = halo_row<S>(hr[i], hc[i]);
    }
    const bool aok = co0 + v8 < g.Co, bok = ci0 + v8 < g.CiLoad;
    // patch cursor
    int pn, py, px;
    {
        const int per = g.npx * g.npy;
        pn = p_begin / per;
        const int t = p_begin - pn * per;
        py = t / g.npx;
        px = t - py * g.npx;
    }
    u32x4 ra, rb[HLOADS];
    auto load_chunk = [&]() {
        const int y0 = py * 4, x0 = px * 8;
        const int oy = y0 + ar, ox = x0 + ac;
        const bool pok = aok && oy < g.Ho && ox < g.Wo;
        ra = __builtin_amdgcn_raw_buffer_load_b128(yr, pok ? (uint32_t)(((pn * g.Ho + oy) * g.Wo + ox) * (int)g.yld + co0 + v8) * 2u : OOB, 0, 0);
#pragma unroll
        for (int i = 0; i < HLOADS; ++i) {
            if (i >= 2 && !hon[i]) continue;             // (d = 1 needs two of the five)
            const int iy = y0 * S - D + hr[i], ix = x0 * S - D + hc[i];
            const bool ok = hon[i] && bok && (unsigned)iy < (unsigned)g.Hi && (unsigned)ix < (unsigned)g.Wi;
            rb[i] = __builtin_amdgcn_raw_buffer_load_b128(xr, ok ? (uint32_t)(((pn * g.Hi + iy) * g.Wi + ix) * (int)g.xld + ci0 + v8) * 2u : OOB, 0, 0);
        }
        if (++px >= g.npx) {
            px = 0;
            if (++py >= g.npy) { py = 0; ++pn; }
        }
    };
    auto store_chunk = [&](int buf) {
        *reinterpret_cast<u32x4*>(&sA[buf][ak * ROW + (v8 ^ (swz(ak) << 4))]) = ra;
#pragma unroll
        for (int i = 0; i < HLOADS; ++i)
            if (hon[i]) *reinterpret_cast<u32x4*>(&sB[buf][hrow[i] * ROW + (v8 ^ (halo_swz<S>(hrow[i]) << 4))]) = rb[i];
    };
    f32x4 acc[NT][2][2];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto tr_frag = [&](const bf16_t* base, int hi_rows) -> bf16x8 {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)base);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + hi_rows * ROW));
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto compute = [&](int buf) {
        const int q = fr >> 2, pp = fr & 3;
        const int fz = swz(8 * fg + q);
        bf16x8 a[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = tr_frag(&sA[buf][(8 * fg + q) * ROW + (((wm * 2 + i) ^ fz) << 4) + 4 * pp], 4);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int ky = t / 3, kx = t - ky * 3;
            // halo pixel of patch pixel (fg, q) at this tap [and of (fg, q + 4): 4 S rows further, same halo row, same swizzle]
            const int R = halo_row<S>(fg * S + ky * D, q * S + kx * D);
            const int hz = halo_swz<S>(R);
            bf16x8 b[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = tr_frag(&sB[buf][R * ROW + (((wn * 2 + j) ^ hz) << 4) + 4 * pp], 4 * S);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[t][i][j], 0, 0, 0);
        }
    };
    if (nchunks > 0) {
        load_chunk();
        store_chunk(0);
    }
    __syncthreads();
    for (int it = 0; it < nchunks; ++it) {
        const int buf = it & 1;
        const bool more = it + 1 < nchunks;
        if (more) load_chunk();
        compute(buf);
        if (more) store_chunk(buf ^ 1);
        __syncthreads();
    }
    float* o = out + (g.S > 1 ? (int64_t)split * g.Co * NT * g.Cip : 0);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int co = co0 + (wm * 2 + i) * 16 + fg * 4 + e;
                    const int ci = ci0 + (wn * 2 + j) * 16 + fr;
                    if (co < g.Co && ci < g.Ci) {
                        float* d = (g.S == 1 && g.oihw) ? o + ((int64_t)co * g.Ci + ci) * NT + t
                                                        : o + ((int64_t)co * NT + t) * g.Cip + ci;
                        float v = acc[t][i][j][e];
                        if (g.S == 1 && g.accumulate) v += *d;
                        *d = v;
                    }
                }
}

// dw (+)= sum_s slab[s].  A 256-thread block owns 64 consecutive packed elements: 16 lanes x float4 cover them, and the 16
// lane-groups each add every 16th slab with four 16-byte loads in flight (the old one-float-per-lane loop was pure load
// latency: 64 dependent 256-byte reads per wave).  The 16 partial sums are combined in a fixed order -> deterministic.
// A block folds RED_W = 256 consecutive outputs (1 KB per slab: whole DRAM bursts; 64 outputs per block ran at 2.6 TB/s): lane q of
// every wave owns outputs 4q .. 4q+3, the four waves take the slabs k = wave (mod 4), four loads in flight per lane.
constexpr int RED_W = 256;
template <bool VEC4>
__device__ __forceinline__ void wgrad_reduce_group(const float* __restrict__ slabs, float* __restrict__ dw, int64_t n, int S,
                                                   int accumulate, int oihw, int Ci, int Cip, int taps, int64_t base,
                                                   float (*part)[RED_W + 4]) {
    const int q = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int64_t j = base + q * 4;
    f32x4 s0{0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    if (VEC4) {
        if (j < n) {
            const float* src = slabs + j;
            int k = grp;
            for (; k + 12 < S; k += 16) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(src + (int64_t)k * n);
                const f32x4 b = *reinterpret_cast<const f32x4*>(src + (int64_t)(k + 4) * n);
                const f32x4 c = *reinterpret_cast<const f32x4*>(src + (int64_t)(k + 8) * n);
                const f32x4 d = *reinterpret_cast<const f32x4*>(src + (int64_t)(k + 12) * n);
                s0 += a; s1 += b; s2 += c; s3 += d;
            }
            for (; k < S; k += 4) s0 += *reinterpret_cast<const f32x4*>(src + (int64_t)k * n);
        }
    } else {
        for (int e = 0; e < 4; ++e)
            if (j + e < n)
                for (int k = grp; k < S; k += 4) s0[e] += slabs[(int64_t)k * n + j + e];
    }
    const f32x4 sum = (s0 + s1) + (s2 + s3);
    *reinterpret_cast<f32x4*>(&part[grp][q * 4]) = sum;
    __syncthreads();
    {
        const int t = threadIdx.x;
        const float v0 = (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
        float v = v0;
        const int64_t jj = base + t;
        if (jj < n) {
            int64_t dst = jj;
            bool ok = true;
            if (oihw) {
                const int ci = (int)(jj % Cip);
                const int64_t tt = jj / Cip;
                const int tap = (int)(tt % taps);
                const int64_t co = tt / taps;
                ok = ci < Ci;
                dst = (co * Ci + ci) * taps + tap;
            }
            if (ok) {
                if (accumulate) v += dw[dst];
                dw[dst] = v;
            }
        }
    }
    __syncthreads();
}

template <bool VEC4>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dw,
                                                           int64_t n, int S, int accumulate, int oihw, int Ci, int Cip,
                                                           int taps) {
    __shared__ __attribute__((aligned(16))) float part[4][RED_W + 4];
    for (int64_t base = (int64_t)blockIdx.x * RED_W; base < n; base += (int64_t)gridDim.x * RED_W)
        wgrad_reduce_group<VEC4>(slabs, dw, n, S, accumulate, oihw, Ci, Cip, taps, base, part);
}

// ---- single-layer kernels ---------------------------------------------------------------------------------------------------
template <typename T, bool VECLOAD, int PK>
__global__ __launch_bounds__(256) void wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                    float* __restrict__ out, const WGeom g) {
    wgrad_body<T, VECLOAD, PK>(x, dy, out, g, blockIdx.x);
}
template <int PK>
__global__ __launch_bounds__(256, 2) void wgrad128_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                          float* __restrict__ out, const WGeom g) {
    wgrad_body128<PK>(x, dy, out, g, blockIdx.x);
}

// ---- ping-pong "row" kernel (kind 5; bf16, 3x3 / stride 1 / pad 1 / dilation 1, same-size maps) --------------------------------
// The per-tap 128 x 128 kernel above stages 16 KB of operands per 64 MFMAs of the block: at the MFMA rate that is 64 B per clock
// and CU out of L2 -- it runs at the aggregate L2 bandwidth (645 TFLOP/s = 26 % of the bf16 rate on config 5), not on the matrix
// cores.  Here a block owns ONE KERNEL ROW (ky; the three taps kx = 0, 1, 2) of a 128 (co) x 128 (ci) tile over a range of
// 16 x 16-pixel patches: per k-step (2 patch rows x 16 columns = 32 pixels) it stages the dy rows once (8 KB) and the x rows of
// that kernel row WITH their two halo columns (2 x 18 pixels, 9 KB) and the three taps read the same x tile at pixel offsets
// kx = 0, 1, 2 -- 17 KB per 384 MFMAs, 22 B per clock and CU at the full MFMA rate.  Structure as conv_pp.hip: 512 threads, two
// wave groups staggered by one barrier (one group reads fragments and issues LDS-DMA while the other runs its 24 MFMAs), a ring of
// six 20 KB stages filled by LDS-DMA four k-steps ahead, counted s_waitcnt vmcnt (never 0 in the loop), raw s_barrier.
//   wave (group g, wq): co half hc = wq & 1 (64 channels), ci quarter qc = 2 g + (wq >> 1) (32 channels), all three taps:
//   3 x 4 x 2 accumulator tiles = 96 registers; per k-step 4 A fragments (dy, shared by the taps) + 3 x 2 B fragments (x).
//   Operands are pixel-major in LDS ([pixel][128 channels] = 256-byte rows) and transposed on the way out (ds_read_b64_tr_b16: per
//   16-lane group 4 pixels x 16 channels); the 32-byte channel group g of pixel slot s sits at g ^ f(s), f(s) = (s & 3) | ((s >> 3)
//   & 1) << 2: the eight pixel rows a 32-lane half reads ({b .. b+3} and {b+8 .. b+11} for every base b, i.e. for every tap offset)
//   get eight different groups -- conflict-free.  LDS-DMA writes lane-linearly, so the swizzle is applied to the SOURCE slot.
//   Borders: dy / x pixels outside the image (ragged patches, the halo columns and rows of the zero padding) and channels beyond
//   Co / Ci are out-of-range DMA lanes = zeros.
constexpr int WPP_STAGE = 32 * 256 + 48 * 256;       // dy tile + x tile (48 pixel slots, 36 used) = 20480 B
#ifndef DSN_WPP_D
#define DSN_WPP_D 4                                   // (5 measured: config 5 18.84 vs 18.79 ms -- the latency is covered at 4)
#endif
constexpr int WPP_D = DSN_WPP_D, WPP_R = WPP_D + 2;   // prefetch distance, ring stages (a stage is refilled >= 2 phases after its read)
constexpr int WPP_LDS = WPP_R * WPP_STAGE;            // 122880 B (D = 4)
static_assert(WPP_D >= 4 && WPP_D <= 5 && WPP_LDS <= 160 * 1024, "ring depth");

__device__ __forceinline__ int wpp_f(int s) { return (s & 3) | (((s >> 3) & 1) << 2); }
template <int N> __device__ __forceinline__ void wpp_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ void wgrad_pp_body(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, float* __restrict__ out,
                                              const WGeom& g, int bid, unsigned char* smem) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = wave >> 2, wq = wave & 3;
    const int hc = wq & 1, qc = 2 * grp + (wq >> 1);
    const int fr = lane & 15, fg = lane >> 4;
    const int taps = 9;
    const int tci = bid % g.tiles_ci; bid /= g.tiles_ci;
    const int tco = bid % g.tiles_co; bid /= g.tiles_co;
    const int ky = bid % 3;
    const int split = bid / 3;
    const int co0 = tco * 128, ci0 = tci * 128;
    const int pb = split * g.ppb;                                   // patches [pb, pe) of the N * npy * npx list
    const int pe = (pb + g.ppb < g.P) ? pb + g.ppb : g.P;
    const int T = 8 * (pe - pb);                                    // k-steps

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, g.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, g.dy_bytes, 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr;
    constexpr uint32_t OOB = 0xFFFFFFF0u;

    // ---- DMA cursor: the patch the prefetch is in (it runs WPP_D k-steps ahead of the compute) -----------------------------------
    // lane -> (pixel slot, 16-byte slot v): dy instruction `wave` covers pixels 4 wave .. + 4; x instructions wave (and wave + 8 for
    // group 0) cover pixel slots 4 t .. + 4 of the 2 x 18 tile (slots >= 36: padding, always out of range)
    const int v = lane & 15;
    const int ka = 4 * wave + (lane >> 4);                           // dy pixel of the k-step: row ka >> 4, column ka & 15
    const int lsa = ((((v >> 1) ^ wpp_f(ka)) << 1) | (v & 1)) * 8;   // logical channel offset this lane fetches
    int sb[2], lsb[2], xrw[2], xcl[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        sb[u] = 4 * (wave + 8 * u) + (lane >> 4);
        lsb[u] = ((((v >> 1) ^ wpp_f(sb[u])) << 1) | (v & 1)) * 8;
        xrw[u] = sb[u] / 18;
        xcl[u] = sb[u] - xrw[u] * 18;
    }
    const uint32_t arow = (uint32_t)(2 * g.Wo * (int)g.yld) * 2u, brow = (uint32_t)(2 * g.Wi * (int)g.xld) * 2u;   // two image rows
    // (validity is a flag of its own: the x offset of image 0's row -1 is a NEGATIVE number wrapped to 32 bits -- it becomes valid
    //  once hk * brow is added -- and one such value coincides with the out-of-range sentinel)
    uint32_t a_off = 0, b_off[2] = {0, 0};
    bool a_ok = false, b_ok[2] = {false, false};
    int a_ylim = 0, b_y[2] = {0, 0};
    // Round 4 (tools/exp/oob_order.hip): a DMA whose lanes are ALL out of range retires at once and must not stand for an operation in
    // flight in the counted waits.  (a) Only wave 0 issues the second x instruction (pixel slots 32 .. 35): for waves 1 .. 3 it
    // covered the never-read padding slots 36 .. 47 and was dead in every k-step -- their waits were short of two operations.
    // (b) c_edge: the cursor's patch can have dead DMAs at all (ragged patches, the zero-padding rows above / below the image, partial
    // channel tiles) -- only then are the k-step's DMAs checked (one ballot each) and a wave with dead ones among the two k-steps a
    // wait counts subtracts them (at most five operations are ever allowed in flight: a six-way chain).
    const int npk = wave == 0 ? 3 : 2;                              // DMA instructions of this wave per k-step
    bool c_edge = false;
    int dq0 = 0, dq1 = 0, dq2 = 0;                                  // dead DMAs of the youngest .. third youngest k-step requested
    auto cursor = [&](int patch) {                                  // per-lane source offsets of k-step 0 of `patch`
        a_ok = false; b_ok[0] = false; b_ok[1] = false;
        c_edge = false;
        if (patch >= pe) return;
        const int per = g.npy * g.npx;
        const int n = patch / per, rem = patch - n * per;
        const int y0 = (rem / g.npx) * 16, x0 = (rem % g.npx) * 16;
        c_edge = y0 + ky - 1 < 0 || y0 + ky + 15 > g.Hi || x0 + 17 > g.Wi || y0 + 16 > g.Ho || x0 + 16 > g.Wo || co0 + 128 > g.Co ||
                 ci0 + 128 > g.CiLoad;
        const int ya = y0 + (ka >> 4), xa = x0 + (ka & 15);
        a_ylim = g.Ho - ya;
        a_ok = xa < g.Wo && co0 + lsa < g.Co;
        a_off = (uint32_t)(((n * g.Ho + ya) * g.Wo + xa) * (int)g.yld + co0 + lsa) * 2u;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int yb = y0 + xrw[u] + ky - 1, xb = x0 + xcl[u] - 1;
            b_y[u] = yb;
            b_ok[u] = sb[u] < 36 && (unsigned)xb < (unsigned)g.Wi && ci0 + lsb[u] < g.CiLoad;
            b_off[u] = (uint32_t)(((n * g.Hi + yb) * g.Wi + xb) * (int)g.xld + ci0 + lsb[u]) * 2u;          // (yb = -1: wraps; rows are masked below)
        }
    };
    auto issue = [&](int hk, int slot) {                            // LDS-DMA of k-step hk of the cursor's patch into ring stage `slot`
        unsigned char* st = smem + slot * WPP_STAGE;
        const uint32_t oa = (a_ok && 2 * hk < a_ylim) ? a_off + (uint32_t)hk * arow : OOB;
        const uint32_t o0 = (b_ok[0] && (unsigned)(b_y[0] + 2 * hk) < (unsigned)g.Hi) ? b_off[0] + (uint32_t)hk * brow : OOB;
        const uint32_t o1 = (b_ok[1] && (unsigned)(b_y[1] + 2 * hk) < (unsigned)g.Hi) ? b_off[1] + (uint32_t)hk * brow : OOB;
        int nd = 0;
        if (c_edge) {
            nd = (__ballot(oa != OOB) == 0ull ? 1 : 0) + (__ballot(o0 != OOB) == 0ull ? 1 : 0) + ((wave == 0 && __ballot(o1 != OOB) == 0ull) ? 1 : 0);
            nd = __builtin_amdgcn_readfirstlane(nd);
        }
        dq2 = dq1; dq1 = dq0; dq0 = nd;                                      // (k-steps past the block's range: c_edge is off, the tail waits leave them out)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(yr, (lds_ptr)(st + wave * 1024), 16, oa, 0, 0, DSN_DMA_AUX);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr)(st + 8192 + wave * 1024), 16, o0, 0, 0, DSN_DMA_AUX);
        if (wave == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr)(st + 8192 + (wave + 8) * 1024), 16, o1, 0, 0, DSN_DMA_AUX);
    };

    // ---- fragment addresses (stage-relative) -----------------------------------------------------------------------------------
    const int q = fr >> 2, pp = fr & 3;
    int a_addr[4], b_addr[3][2][2];                                 // A: co block i; B: tap kx, k half (pixels +0 / +4), ci block jn
    {
        const int k = 8 * fg + q;                                   // pixel of the lower half of the fragment (upper: + 4, same f)
        const int f = wpp_f(k);
#pragma unroll
        for (int i = 0; i < 4; ++i) a_addr[i] = k * 256 + (((hc * 4 + i) ^ f) << 5) + pp * 8;
        const int s0 = (fg >> 1) * 18 + 8 * (fg & 1) + q;          // x pixel slot of pixel k at tap 0
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int s = s0 + kx + 4 * h;
#pragma unroll
                for (int jn = 0; jn < 2; ++jn) b_addr[kx][h][jn] = 8192 + s * 256 + (((qc * 2 + jn) ^ wpp_f(s)) << 5) + pp * 8;
            }
    }
    f32x4 acc[3][4][2];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) acc[kx][i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto tr2 = [&](const unsigned char* p, bool second_a) -> bf16x8 {     // two transposing reads: pixels +0..3 and +4..7
        (void)second_a;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 4 * 256));
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };

    // ---- prologue: k-steps 0 .. D-1 in flight, k-step 0 landed -----------------------------------------------------------------------
    cursor(pb);
#pragma unroll
    for (int j = 0; j < WPP_D; ++j) issue(j, j);
    // (k-step 0 landed; younger: k-steps 1 .. D - 1 -- a wave with dead DMAs anywhere in the prologue drains)
    {
        if (c_edge) wpp_wait<0>();                                  // (conservative: the first patch touches an edge)
        else if (npk == 3) wpp_wait<3 * (WPP_D - 1)>(); else wpp_wait<2 * (WPP_D - 1)>();
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (grp == 1) {
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    int sc = 0;                                                     // ring stage of the k-step being computed
    for (int pi = pb; pi < pe; ++pi) {
#pragma unroll
        for (int hk = 0; hk < 8; ++hk) {
            const unsigned char* st = smem + sc * WPP_STAGE;
            // ---- load segment: fragments of this k-step, then the prefetch of k-step + D ---------------------------------------------
            bf16x8 a[4], b[3][2];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = tr2(st + a_addr[i], true);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int jn = 0; jn < 2; ++jn) {
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(st + b_addr[kx][0][jn]));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(st + b_addr[kx][1][jn]));
                    b[kx][jn] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                }
            // k-step + 1 landed: the DMAs of k-steps + 2, + 3 (issued in the two phases before this one) may stay in flight.
            // In the block's LAST patch the k-steps past its end are all-out-of-range padding DMAs: they retire at once
            // (tools/exp/oob_order.hip) and must not be counted -- only the real k-steps behind k-step + 1 are.
            const int dd = dq0 + dq1 + (WPP_D >= 5 ? dq2 : 0);      // dead DMAs among k-steps + 2 .. + D - 1 (patches at an edge)
            if (dd != 0) {                                          // exact count
                const int real = (pi == pe - 1 && hk + WPP_D > 8) ? (6 - hk < 0 ? 0 : (6 - hk < WPP_D - 2 ? 6 - hk : WPP_D - 2)) : WPP_D - 2;
                const int al = real * npk - dd;
                if (al <= 0) wpp_wait<0>();
                else if (al == 1) wpp_wait<1>();
                else if (al == 2) wpp_wait<2>();
                else if (al == 3) wpp_wait<3>();
                else if (al == 4) wpp_wait<4>();
                else if (al == 5) wpp_wait<5>();
                else if (al == 6) wpp_wait<6>();
                else if (al == 7) wpp_wait<7>();
                else wpp_wait<8>();
            } else if (pi == pe - 1 && hk + WPP_D > 8) {
                const int real = 6 - hk < 0 ? 0 : (6 - hk < WPP_D - 2 ? 6 - hk : WPP_D - 2);       // of k-steps hk + 2 .. hk + D - 1, those < 8
                if (real == 0) wpp_wait<0>();
                else if (real == 1) { if (npk == 3) wpp_wait<3>(); else wpp_wait<2>(); }
                else if (real == 2) { if (npk == 3) wpp_wait<6>(); else wpp_wait<4>(); }
                else { if (npk == 3) wpp_wait<3 * (WPP_D - 2)>(); else wpp_wait<2 * (WPP_D - 2)>(); }
            } else {
                if (npk == 3) wpp_wait<3 * (WPP_D - 2)>(); else wpp_wait<2 * (WPP_D - 2)>();
            }
            if (hk == 8 - WPP_D) cursor(pi + 1);                     // the prefetch enters the next patch (all lanes OOB past the range)
            int sd = sc + WPP_D;
            if (sd >= WPP_R) sd -= WPP_R;
            issue((hk + WPP_D) & 7, sd);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ---- MFMA segment ----------------------------------------------------------------------------------------------------
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int jn = 0; jn < 2; ++jn)
                        acc[kx][i][jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[kx][jn], acc[kx][i][jn], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            sc = sc + 1 == WPP_R ? 0 : sc + 1;
        }
    }
    if (grp == 0) {
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    (void)T;
    // ---- epilogue: the three taps of this kernel row, straight from the accumulators (row = co, column = ci of each 16 x 16 tile)
    float* o = out + (g.S > 1 ? (int64_t)split * g.Co * taps * g.Cip : 0);
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int tap = ky * 3 + kx;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int co = co0 + hc * 64 + i * 16 + fg * 4 + e;
                    const int ci = ci0 + qc * 32 + jn * 16 + fr;
                    if (co < g.Co && ci < g.Ci) {
                        float* d = (g.S == 1 && g.oihw) ? o + ((int64_t)co * g.Ci + ci) * taps + tap
                                                        : o + ((int64_t)co * taps + tap) * g.Cip + ci;
                        float val = acc[kx][i][jn][e];
                        if (g.S == 1 && g.accumulate) val += *d;
                        *d = val;
                    }
                }
    }
}
__global__ __launch_bounds__(512, 2) void wgrad_pp_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                          float* __restrict__ out, const WGeom g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wpp_smem[];
    wgrad_pp_body(x, dy, out, g, blockIdx.x, wpp_smem);
}

__global__ __launch_bounds__(256, 2) void wgrad_halo_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                            float* __restrict__ out, const WGeom g) {
    __shared__ __attribute__((aligned(16))) bf16_t sA[2][HALO_A];
    __shared__ __attribute__((aligned(16))) bf16_t sB[2][HALO_B];
    if (g.stride == 2) wgrad_halo_body<2>(x, dy, out, g, blockIdx.x, sA, sB);
    else wgrad_halo_body<1>(x, dy, out, g, blockIdx.x, sA, sB);
}
template <int NT>
__global__ __launch_bounds__(256, 2) void wgrad_alltaps_bf16_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                                 float* __restrict__ out, const WGeom g) {
    wgrad_alltaps_body<NT>(x, dy, out, g, blockIdx.x);
}

// ---- grouped kernels: the weight gradients of ALL layers of a backward pass in three launches ---------------------------------
// dW feeds nothing but the optimizer, so nothing in the backward pass waits for it.  Layer by layer each wgrad is a 10-30 us
// launch that cannot fill 256 CUs (a 64x64 tile per tap, split-K capped by the pixel count) followed by a 6-10 us slab
// reduction: ~150 launches and ~2.2 ms per step at batch 8.  Queued instead (dsn_conv2d_wgrad_plan) and run together
// (dsn_conv2d_wgrad_run), the blocks of every layer share one grid: a block finds its job by binary search over the
// per-job first-block index (uniform per block: scalar loads) and then runs the same body as the single-layer kernel.
struct WJob {
    const void* x;
    const void* dy;
    float* out;            // slabs (S > 1) or dw
    float* dw;
    WGeom g;
    int32_t kind;          // 0: per-tap blocks (64 x 64 tiles), 1: all-taps blocks, 3: per-tap blocks with 128 x 128 tiles, 4: halo-tile all-taps,
                           // 5: ping-pong kernel-row blocks (128 x 128 x 3 taps)
    int32_t dtype;
    int32_t blocks[6];     // blocks of this job in the per-tap / all-taps / reduce / 128-tile / halo-tile / ping-pong launch
    int32_t start[6];      // first block of this job in each launch
    int64_t n_out;
    double flops, bytes;
};
// Blocks of one job that the hardware puts on the same XCD (consecutive block ids go round-robin over the eight XCDs) get
// CONSECUTIVE work items: the (co, ci, tap) tiles of one pixel range then run on one XCD at about the same time and share its L2
// -- every tile of a range reads the same dy rows (and, per ci tile, the same x rows).  Dealt round-robin, each of up to eight
// XCDs fetched its own copy (PMC: 2.56 GB per step against 1.25 GB of operands).
__device__ __forceinline__ int xcd_local(int local, int nblocks) {
    const int q = nblocks >> 3, r = nblocks & 7, x = local & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (local >> 3);
}
__device__ __forceinline__ int find_job(const WJob* __restrict__ jobs, int n, int bid, int k) {
    int lo = 0, hi = n - 1;            // largest job index whose first block is <= bid
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].start[k] <= bid) lo = mid; else hi = mid - 1;
    }
    return lo;
}
template <typename T, int PK>
__global__ __launch_bounds__(256) void wgrad_grouped_kernel(const WJob* __restrict__ jobs, int n) {
    const int l = find_job(jobs, n, blockIdx.x, 0);
    const WGeom g = jobs[l].g;
    wgrad_body<T, true, PK>((const T*)jobs[l].x, (const T*)jobs[l].dy, jobs[l].out, g,
                            xcd_local(blockIdx.x - jobs[l].start[0], jobs[l].blocks[0]));
}
__global__ __launch_bounds__(256, 2) void wgrad_alltaps_grouped_kernel(const WJob* __restrict__ jobs, int n) {
    const int l = find_job(jobs, n, blockIdx.x, 1);
    const WGeom g = jobs[l].g;
    wgrad_alltaps_body<9>((const bf16_t*)jobs[l].x, (const bf16_t*)jobs[l].dy, jobs[l].out, g,
                          xcd_local(blockIdx.x - jobs[l].start[1], jobs[l].blocks[1]));
}
template <int PK>
__global__ __launch_bounds__(256, 2) void wgrad128_grouped_kernel(const WJob* __restrict__ jobs, int n) {
    const int l = find_job(jobs, n, blockIdx.x, 3);
    const WGeom g = jobs[l].g;
    wgrad_body128<PK>((const bf16_t*)jobs[l].x, (const bf16_t*)jobs[l].dy, jobs[l].out, g,
                      xcd_local(blockIdx.x - jobs[l].start[3], jobs[l].blocks[3]));
}
__global__ __launch_bounds__(256, 2) void wgrad_halo_grouped_kernel(const WJob* __restrict__ jobs, int n) {
    const int l = find_job(jobs, n, blockIdx.x, 4);
    const WGeom g = jobs[l].g;
    __shared__ __attribute__((aligned(16))) bf16_t sA[2][HALO_A];
    __shared__ __attribute__((aligned(16))) bf16_t sB[2][HALO_B];
    const int b = xcd_local(blockIdx.x - jobs[l].start[4], jobs[l].blocks[4]);
    if (g.stride == 2) wgrad_halo_body<2>((const bf16_t*)jobs[l].x, (const bf16_t*)jobs[l].dy, jobs[l].out, g, b, sA, sB);
    else wgrad_halo_body<1>((const bf16_t*)jobs[l].x, (const bf16_t*)jobs[l].dy, jobs[l].out, g, b, sA, sB);
}
__global__ __launch_bounds__(512, 2) void wgrad_pp_grouped_kernel(const WJob* __restrict__ jobs, int n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wpp_smem[];
    const int l = find_job(jobs, n, blockIdx.x, 5);
    const WGeom g = jobs[l].g;
    wgrad_pp_body((const bf16_t*)jobs[l].x, (const bf16_t*)jobs[l].dy, jobs[l].out, g,
                  xcd_local(blockIdx.x - jobs[l].start[5], jobs[l].blocks[5]), wpp_smem);
}
__global__ __launch_bounds__(256) void wgrad_reduce_grouped_kernel(const WJob* __restrict__ jobs, int n) {
    __shared__ __attribute__((aligned(16))) float part[4][RED_W + 4];
    const int l = find_job(jobs, n, blockIdx.x, 2);
    const WJob& j = jobs[l];
    const int64_t base = (int64_t)(blockIdx.x - j.start[2]) * RED_W;
    const int taps = j.g.KH * j.g.KW;
    if (j.n_out % 4 == 0)    // slabs come 256-byte aligned out of the queue's arena
        wgrad_reduce_group<true>(j.out, j.dw, j.n_out, j.g.S, j.g.accumulate, j.g.oihw, j.g.Ci, j.g.Cip, taps, base, part);
    else
        wgrad_reduce_group<false>(j.out, j.dw, j.n_out, j.g.S, j.g.accumulate, j.g.oihw, j.g.Ci, j.g.Cip, taps, base, part);
}

inline bool use_alltaps(const dsn_tensor* x, const dsn_tensor* dy, const dsn_conv_params* p) {
    if (x->dtype != DSN_BF16 || p->kh != 3 || p->kw != 3) return false;
    static const int mode = [] { const char* e = getenv("DSN_WGRAD_ALLTAPS"); return e ? atoi(e) : -1; }();   // tuning knob
    if (mode == 0) return false;
    // measured (tools/sweep_wgrad.sh): sharing dy across the taps wins on the large maps (P >= 32k pixels: Focus, the stride-2
    // stem, 160x160 bottlenecks, FFM); on the small maps the per-tap blocks' 9x higher block count matters more
    if (mode < 0 && npix(dy) < 32768) return false;
    const bool vl = (dy->c % 8 == 0) && (x->c % 8 == 0) && (dy->ldc % 8 == 0) && (x->ldc % 8 == 0) &&
                    ((uintptr_t)x->ptr % 16 == 0) && ((uintptr_t)dy->ptr % 16 == 0);
    const bool fits = (npix(x) * x->ldc < (1ll << 30)) && (npix(dy) * dy->ldc < (1ll << 30));
    return vl && fits;
}

// halo-tile all-taps kernel: same-size 3x3 / stride 1 / pad = dilation = 1..3, and 3x3 / stride 2 / pad 1
inline bool use_halo(const dsn_tensor* x, const dsn_tensor* dy, const dsn_conv_params* p) {
    // Measured (MI355X): a block of this kernel is 20-25 % faster than an all-taps block on every 3x3 shape of both configurations
    // (tools/bench_wgrad_kinds.py: 155-165 -> 118-130 us for a 4096-pixel range).  While it could not take the stride-2 layers it
    // only SPLIT the all-taps launch in two under-filled ones (config 3 4.70 -> 4.82 ms); with them it replaces that launch for
    // every DeSeNet shape: config 3 4.690 -> 4.678 ms, config 5 23.31 -> 23.06 ms from the all-taps threshold up (mode 2, default);
    // on every eligible layer (mode 1) the small maps lose their 9x block count: 4.727 ms.  DSN_WGRAD_HALO = 0 / 1 / 2.
    static const int mode = [] { const char* e = getenv("DSN_WGRAD_HALO"); return e ? atoi(e) : 2; }();
    static const int minpx = [] { const char* e = getenv("DSN_WGRAD_HALO_MINPX"); return e ? atoi(e) : 32768; }();
    if (mode == 0 || x->dtype != DSN_BF16) return false;
    const bool s1 = p->stride == 1 && p->dil >= 1 && p->dil <= 3 && p->pad == p->dil && x->h == dy->h && x->w == dy->w;
    const bool s2 = p->stride == 2 && p->dil == 1 && p->pad == 1;
    if (p->kh != 3 || p->kw != 3 || !(s1 || s2)) return false;
    if (mode == 2 && npix(dy) < minpx) return false;
    const bool vl = (dy->c % 8 == 0) && (x->c % 8 == 0) && (dy->ldc % 8 == 0) && (x->ldc % 8 == 0) &&
                    ((uintptr_t)x->ptr % 16 == 0) && ((uintptr_t)dy->ptr % 16 == 0);
    const bool fits = (npix(x) * x->ldc < (1ll << 30)) && (npix(dy) * dy->ldc < (1ll << 30));
    return vl && fits;
}

// 128 x 128 tiles: bf16, whole 128-channel blocks on both sides, enough pixels that the 4x fewer tiles still fill the chip through
// the split (DSN_WGRAD_T128 = 0: never, 1: whenever the shape allows, default: P >= DSN_WGRAD_T128_MINPX pixels)
inline bool use_tile128(const dsn_tensor* x, const dsn_tensor* dy, const dsn_conv_params* p, bool alltaps) {
    static const int mode = [] { const char* e = getenv("DSN_WGRAD_T128"); return e ? atoi(e) : -1; }();
    // (measured, MI355X: config 5 24.57 ms without the 128-wide tiles, 23.83 with them from 16384 pixels up, 23.69 on every eligible
    //  layer; config 3 4.728 / 4.79 / 4.712 ms -- a partial move leaves two under-filled grouped launches, so it is all or nothing)
    static const int minpx = [] { const char* e = getenv("DSN_WGRAD_T128_MINPX"); return e ? atoi(e) : 0; }();
    if (mode == 0 || alltaps || x->dtype != DSN_BF16) return false;
    if (dy->c % TB2 != 0 || x->c % TB2 != 0) return false;
    return mode == 1 || npix(dy) >= minpx;
}

// ping-pong kernel-row kernel (kind 5): bf16, same-size 3x3 / stride 1 / pad 1 / dilation 1, whole 128-channel tiles on both sides,
// 16 x 16 patches that cover >= 80 % of the map (DSN_WGRAD_PP: 0 never, 1 default, 2 every eligible layer)
int g_wgrad_pp_mode = getenv("DSN_WGRAD_PP") ? atoi(getenv("DSN_WGRAD_PP")) : 1;
inline bool use_pp(const dsn_tensor* x, const dsn_tensor* dy, const dsn_conv_params* p, int32_t oihw, int32_t ci_pad) {
    const int mode = g_wgrad_pp_mode;
    static const int minpx = [] { const char* e = getenv("DSN_WGRAD_PP_MINPX"); return e ? atoi(e) : 12800; }();
    if (mode == 0 || x->dtype != DSN_BF16) return false;
    if (p->kh != 3 || p->kw != 3 || p->stride != 1 || p->pad != 1 || p->dil != 1 || x->h != dy->h || x->w != dy->w) return false;
    (void)oihw;
    if (dy->c % 128 != 0 || x->c % 128 != 0 || ci_pad != x->c) return false;
    const bool vl = (dy->ldc % 8 == 0) && (x->ldc % 8 == 0) && ((uintptr_t)x->ptr % 16 == 0) && ((uintptr_t)dy->ptr % 16 == 0) &&
                    ((npix(x) - 1) * x->ldc + x->c) * 2 < (1ll << 31) && ((npix(dy) - 1) * dy->ldc + dy->c) * 2 < (1ll << 31);
    if (!vl) return false;
    if (mode == 2) return true;
    const int ty = (dy->h + 15) / 16, tx = (dy->w + 15) / 16;
    if ((double)dy->h * dy->w / ((double)ty * tx * 256.0) < 0.8 || npix(dy) < minpx) return false;
    // Measured (MI355X, whole step, alternated runs): config 5 (thirteen 128 -> 128 @160, fifteen 256 -> 256 @80, 512 <-> 256 @160
    // layers share the launch) 18.60 -> 18.47 ms; config 3, where FFM's 256 -> 128 @80 would be the ONLY job of this launch (96
    // blocks on 256 CUs while the launch it left loses its largest job): 3.91 -> 3.98 ms.  So: layers with >= 256 output channels or
    // >= 100k output pixels -- config 5's, not config 3's.
    static const int big_px = [] { const char* e = getenv("DSN_WGRAD_PP_BIGPX"); return e ? atoi(e) : 100000; }();
    return dy->c >= 256 || npix(dy) >= big_px;
}
// patches per block of the ping-pong kernel: about DSN_WGRAD_BLOCKSPP blocks per job (3 kernel rows x tiles x splits)
inline void pp_split(const dsn_tensor* x, const dsn_tensor* dy, int* S, int* ppb, int* P) {
    static const int target = [] { const char* e = getenv("DSN_WGRAD_BLOCKSPP"); return e ? atoi(e) : 96; }();
    const int np = dy->n * ((dy->h + 15) / 16) * ((dy->w + 15) / 16);
    const int base = 3 * (dy->c / 128) * (x->c / 128);
    int s = (target + base - 1) / base;
    if (s > np) s = np;
    if (s < 1) s = 1;
    *ppb = (np + s - 1) / s;
    *S = (np + *ppb - 1) / *ppb;
    *P = np;
}

inline int choose_split(const WGeom& g, bool alltaps = false, bool t128 = false) {
    const int64_t base = (int64_t)g.tiles_co * g.tiles_ci * (alltaps ? 1 : g.KH * g.KW);
    static const int target = [] { const char* e = getenv("DSN_WGRAD_BLOCKS"); return e ? atoi(e) : 320; }();
    static const int scap = [] { const char* e = getenv("DSN_WGRAD_SCAP"); return e ? atoi(e) : 256; }();
    static const int minpx = [] { const char* e = getenv("DSN_WGRAD_MINPX"); return e ? atoi(e) : 2048; }();
    // (re-measured with the grouped launches, MI355X: blocks 288..352 x minpx 4096..5120 is a plateau -- 0.54 ms gather + 0.06 ms
    //  slab reduce; 256 / 1024 was 0.56 + 0.14: four times the slab traffic for parallelism the shared grids no longer need.
    //  Round 3, with the 256-output slab reduction: minpx 2048 + 96 blocks per 128-wide job -1.2 % on the config-3 step (4.115 ->
    //  4.065 ms), 1024 / 3072 / 8192 worse, config 5 flat)
    static const int target128 = [] { const char* e = getenv("DSN_WGRAD_BLOCKS128"); return e ? atoi(e) : 96; }();     // (sweep: 64..128 best on config 5)
    const int tgt = alltaps ? target / 2 : (t128 ? target128 : target), cap = alltaps ? scap * 2 : scap;   // all-taps blocks are 9x heavier
    int64_t s = (tgt + base - 1) / base;
    const int64_t smax = (g.P + minpx - 1) / minpx;
    if (s > smax) s = smax;
    if (s > cap) s = cap;
    return (int)(s < 1 ? 1 : s);
}

}  // namespace

extern "C" int64_t dsn_conv2d_wgrad_workspace_bytes(const dsn_tensor* x, const dsn_tensor* dy,
                                                    const dsn_conv_params* p, int32_t ci_pad) {
    if (!x || !dy || !p) return 0;
    if (use_pp(x, dy, p, 1, x->c) && ci_pad == x->c) {
        int S, ppb, P;
        pp_split(x, dy, &S, &ppb, &P);
        return S > 1 ? (int64_t)S * dy->c * 9 * x->c * sizeof(float) : 0;
    }
    WGeom g{};
    g.P = (int32_t)npix(dy);
    const bool hl = use_halo(x, dy, p);
    const bool at = hl || use_alltaps(x, dy, p);
    const int tb = use_tile128(x, dy, p, at) ? TB2 : TB;
    g.tiles_co = (dy->c + tb - 1) / tb;
    g.tiles_ci = (x->c + tb - 1) / tb;
    g.KH = p->kh; g.KW = p->kw;
    int S = choose_split(g, at, tb == TB2);
    if (hl) {       // (the split is re-derived on whole patches: same arithmetic as make_job)
        const int64_t np = (int64_t)x->n * ((dy->h + 3) / 4) * ((dy->w + 7) / 8);
        const int64_t ppb = (np + S - 1) / S;
        S = (int)((np + ppb - 1) / ppb);
    }
    const int64_t row = ci_pad > x->c ? ci_pad : x->c;
    return S > 1 ? (int64_t)S * dy->c * p->kh * p->kw * row * sizeof(float) : 0;
}

namespace {

// Validate one layer's weight-gradient problem and describe it (geometry, split, block counts, slab placement).
int make_job(const dsn_tensor* x, const dsn_tensor* dy, float* dw, int32_t ci_pad, int32_t oihw, const dsn_conv_params* p,
             void* workspace, int64_t workspace_bytes, WJob* job, bool* vec_out) {
    DSN_CHECK_ARG(tensor_ok(x) && tensor_ok(dy) && dw && p, "conv wgrad: null/invalid argument");
    DSN_CHECK_ARG(x->dtype == dy->dtype && x->n == dy->n && ci_pad > 0 && (oihw ? ci_pad <= x->c : ci_pad >= x->c),
                  "conv wgrad: dtype/batch/channel mismatch");
    const int ho = (x->h + 2 * p->pad - p->dil * (p->kh - 1) - 1) / p->stride + 1;
    const int wo = (x->w + 2 * p->pad - p->dil * (p->kw - 1) - 1) / p->stride + 1;
    DSN_CHECK_ARG(ho == dy->h && wo == dy->w, "conv wgrad: dy is %dx%d, expected %dx%d", dy->h, dy->w, ho, wo);
    DSN_CHECK_ARG(npix(dy) < (1ll << 31) && npix(x) < (1ll << 31), "conv wgrad: too many pixels");
    WGeom g{};
    g.P = (int32_t)npix(dy); g.Ho = dy->h; g.Wo = dy->w; g.Hi = x->h; g.Wi = x->w;
    g.Co = dy->c; g.Ci = x->c; g.Cip = ci_pad;
    g.KH = p->kh; g.KW = p->kw; g.stride = p->stride; g.pad = p->pad; g.dil = p->dil;
    g.yld = dy->ldc; g.xld = x->ldc;
    const bool pp = use_pp(x, dy, p, oihw, ci_pad);
    const bool halo = !pp && use_halo(x, dy, p) && (!oihw || ci_pad >= 0);
    const bool alltaps = !pp && !halo && use_alltaps(x, dy, p);
    bool t128 = !pp && use_tile128(x, dy, p, alltaps || halo);
    {   // (the 128-tile kernel exists for the 16-byte paths only: the same conditions as *vec_out below)
        const int es0 = 2;
        t128 = t128 && (dy->ldc % 8 == 0) && (x->ldc % 8 == 0) && ((uintptr_t)x->ptr % 16 == 0) && ((uintptr_t)dy->ptr % 16 == 0) &&
               ((npix(x) - 1) * x->ldc + x->c) * es0 < (1ll << 31) && ((npix(dy) - 1) * dy->ldc + dy->c) * es0 < (1ll << 31) &&
               (!oihw || ci_pad >= x->c);
    }
    const int tb = t128 ? TB2 : TB;
    g.tiles_co = (g.Co + tb - 1) / tb; g.tiles_ci = (g.Ci + tb - 1) / tb;
    g.S = choose_split(g, alltaps || halo, t128);
    static const int gpk = [] {
        const char *e = getenv("DSN_WGRAD_GPK"), *f = getenv("DSN_WGRAD_T128_PK");
        return ((e && atoi(e) == 64) || (f && atoi(f) == 64)) ? 64 : 32;
    }();
    const int PK = gpk;       // pixel ranges are multiples of the largest chunk any kernel variant may use
    g.ppb = (((g.P + g.S - 1) / g.S) + PK - 1) / PK * PK;
    g.S = (g.P + g.ppb - 1) / g.ppb;
    if (pp) {       // the unit of work is a 16 x 16 patch; blocks = kernel rows x 128 x 128 tiles x splits
        g.tiles_co = g.Co / 128; g.tiles_ci = g.Ci / 128;
        g.npx = (g.Wo + 15) / 16; g.npy = (g.Ho + 15) / 16;
        int S, ppb, P;
        pp_split(x, dy, &S, &ppb, &P);
        g.S = S; g.ppb = ppb; g.P = P;
    }
    if (halo) {     // the unit of work is a 4 x 8 patch: P and ppb count patches from here on (flops / bytes below use npix)
        g.npx = (g.Wo + 7) / 8; g.npy = (g.Ho + 3) / 4;
        const int S0 = choose_split(g, true, false);
        g.P = dy->n * g.npy * g.npx;
        g.ppb = (g.P + S0 - 1) / S0;
        g.S = (g.P + g.ppb - 1) / g.ppb;
    }
    g.accumulate = p->accumulate;
    g.oihw = oihw ? 1 : 0;
    g.CiLoad = x->c;
    if (oihw) g.Ci = (x->c < ci_pad) ? x->c : ci_pad, g.Cip = x->c;   // OIHW: ci_pad = REAL Ci, x may carry zero-padded channels
    const int64_t n_out = (int64_t)g.Co * g.KH * g.KW * g.Cip;
    float* out = dw;
    if (g.S > 1) {
        if (!workspace || workspace_bytes < (int64_t)g.S * n_out * (int64_t)sizeof(float))
            DSN_FAIL(DSN_EWORKSPACE, "conv wgrad: workspace too small (%lld bytes needed)",
                     (long long)((int64_t)g.S * n_out * sizeof(float)));
        out = (float*)workspace;
    }
    const int es = x->dtype == DSN_F32 ? 4 : 2, vec = 16 / es;
    // dy may carry a channel count that is not a multiple of the vector width when its rows are padded (ldc rounded up,
    // Detect: 33 -> 40): the last vector of a row then reads padding lanes, which only feed gradient rows >= Co that are
    // never stored (any value, even NaN, is harmless there)
    const int co_up = (g.Co + vec - 1) / vec * vec;
    const bool dy_padded = g.Co % vec != 0 && co_up <= g.yld;
    const int64_t xb = ((npix(x) - 1) * x->ldc + x->c) * es;
    const int64_t yb = dy_padded ? npix(dy) * dy->ldc * es : ((npix(dy) - 1) * dy->ldc + dy->c) * es;
    *vec_out = (g.Co % vec == 0 || dy_padded) && (g.CiLoad % vec == 0) && (g.yld % vec == 0) && (g.xld % vec == 0) &&
               ((uintptr_t)x->ptr % 16 == 0) && ((uintptr_t)dy->ptr % 16 == 0) && xb < (1ll << 31) && yb < (1ll << 31);
    g.x_bytes = (uint32_t)(xb < (1ll << 31) ? xb : 0);
    g.dy_bytes = (uint32_t)(yb < (1ll << 31) ? yb : 0);
    *job = WJob{};
    job->x = x->ptr; job->dy = dy->ptr; job->out = out; job->dw = dw; job->g = g;
    job->kind = pp ? 5 : halo ? 4 : alltaps ? 1 : (t128 ? 3 : 0);
    job->dtype = x->dtype;
    job->blocks[job->kind] = g.tiles_ci * g.tiles_co * (pp ? 3 : (alltaps || halo) ? 1 : g.KH * g.KW) * g.S;
    job->blocks[2] = g.S > 1 ? (int32_t)((n_out + RED_W - 1) / RED_W) : 0;
    job->n_out = n_out;
    job->flops = 2.0 * (double)npix(dy) * g.Co * g.Ci * g.KH * g.KW;
    job->bytes = ((double)npix(x) * g.Ci + (double)npix(dy) * g.Co) * es + (double)n_out * 4;
    return DSN_OK;
}

constexpr size_t ALLTAPS_LDS = (size_t)2 * 32 * TB * 2 * (1 + 9);   // A + 9 shifted B tiles, double buffered: 80 KiB
int alltaps_attr_once() {
    DSN_LDS_ATTR(wgrad_alltaps_bf16_kernel<9>, (int)ALLTAPS_LDS);
    DSN_LDS_ATTR(wgrad_alltaps_grouped_kernel, (int)ALLTAPS_LDS);
    return DSN_OK;
}

}  // namespace

extern "C" int dsn_wgrad_pp_mode(int32_t mode) {
    if (mode >= 0) g_wgrad_pp_mode = mode;
    return g_wgrad_pp_mode;
}

extern "C" int dsn_conv2d_wgrad(const dsn_tensor* x, const dsn_tensor* dy, float* dw, int32_t ci_pad, int32_t oihw,
                                const dsn_conv_params* p, void* workspace, int64_t workspace_bytes, void* stream) {
    WJob job;
    bool vl = false;
    int rc = make_job(x, dy, dw, ci_pad, oihw, p, workspace, workspace_bytes, &job, &vl);
    if (rc) return rc;
    const WGeom& g = job.g;
    float* out = job.out;
    hipStream_t st = (hipStream_t)stream;
    // packed layout with a padded channel axis: the reduction also visits the padding lanes, which no tile writes
    if (g.S > 1 && !g.oihw && g.Cip != g.Ci) {
        dsn_fill_u32(out, 0u, (int64_t)g.S * job.n_out, st);
    }
    static int pk_bf16 = [] { const char* e = getenv("DSN_WGRAD_PK"); int v = e ? atoi(e) : 32; return (v == 64 || v == 128) ? v : 32; }();
    const int PK = (x->dtype == DSN_F32 || job.kind == 1) ? 32 : pk_bf16;     // (ppb is a multiple of 32; 64/128 only for tuning runs
    if (PK != 32 && g.ppb % PK != 0) DSN_FAIL(DSN_EINVAL, "DSN_WGRAD_PK: pixel range per block is not a multiple of %d", PK);
    dim3 grid(job.blocks[job.kind]), block(256);
    {
        ProfScope prof(KID_WGRAD + (x->dtype == DSN_BF16 ? 1 : 0), job.flops, job.bytes, st);
        if (job.kind == 5) {
            DSN_LDS_ATTR(wgrad_pp_kernel, WPP_LDS);
            hipLaunchKernelGGL(wgrad_pp_kernel, grid, dim3(512), WPP_LDS, st, (const bf16_t*)x->ptr, (const bf16_t*)dy->ptr, out, g);
        } else if (job.kind == 4) {
            hipLaunchKernelGGL(wgrad_halo_kernel, grid, block, 0, st, (const bf16_t*)x->ptr, (const bf16_t*)dy->ptr, out, g);
        } else if (job.kind == 3) {
            hipLaunchKernelGGL(wgrad128_kernel<32>, grid, block, 0, st, (const bf16_t*)x->ptr, (const bf16_t*)dy->ptr, out, g);
        } else if (job.kind == 1) {
            { const int a_ = alltaps_attr_once(); if (a_ != DSN_OK) return a_; }
            hipLaunchKernelGGL(wgrad_alltaps_bf16_kernel<9>, grid, block, ALLTAPS_LDS, st, (const bf16_t*)x->ptr,
                               (const bf16_t*)dy->ptr, out, g);
        } else if (x->dtype == DSN_F32) {
            if (vl)
                hipLaunchKernelGGL((wgrad_kernel<float, true, 32>), grid, block, 0, st, (const float*)x->ptr, (const float*)dy->ptr, out, g);
            else
                hipLaunchKernelGGL((wgrad_kernel<float, false, 32>), grid, block, 0, st, (const float*)x->ptr, (const float*)dy->ptr, out, g);
        } else {
            const bf16_t *xp = (const bf16_t*)x->ptr, *yp = (const bf16_t*)dy->ptr;
            if (!vl) hipLaunchKernelGGL((wgrad_kernel<bf16_t, false, 32>), grid, block, 0, st, xp, yp, out, g);
            else if (PK == 128) hipLaunchKernelGGL((wgrad_kernel<bf16_t, true, 128>), grid, block, 0, st, xp, yp, out, g);
            else if (PK == 64) hipLaunchKernelGGL((wgrad_kernel<bf16_t, true, 64>), grid, block, 0, st, xp, yp, out, g);
            else hipLaunchKernelGGL((wgrad_kernel<bf16_t, true, 32>), grid, block, 0, st, xp, yp, out, g);
        }
    }
    DSN_LAUNCH_CHECK("conv wgrad");
    if (g.S > 1) {
        ProfScope prof(KID_WGRAD_REDUCE, 0.0, (double)(g.S + 1) * job.n_out * 4, st);
        const int64_t b = job.blocks[2];
        const dim3 rgrid((int)(b > 8192 ? 8192 : b));
        if (job.n_out % 4 == 0 && (uintptr_t)out % 16 == 0)
            hipLaunchKernelGGL(wgrad_reduce_kernel<true>, rgrid, dim3(256), 0, st, out, dw, job.n_out, g.S, g.accumulate, g.oihw,
                               g.Ci, g.Cip, g.KH * g.KW);
        else
            hipLaunchKernelGGL(wgrad_reduce_kernel<false>, rgrid, dim3(256), 0, st, out, dw, job.n_out, g.S, g.accumulate, g.oihw,
                               g.Ci, g.Cip, g.KH * g.KW);
        DSN_LAUNCH_CHECK("conv wgrad reduce");
    }
    return DSN_OK;
}

// ---- queued form: plan every layer's job during the backward pass, run them all in three launches at its end -----------------
extern "C" int64_t dsn_wgrad_job_bytes(void) { return (int64_t)sizeof(WJob); }

// Fills *job_out (HOST memory, dsn_wgrad_job_bytes() bytes); launches nothing.  DSN_EUNSUPPORTED when the layer cannot take
// the grouped kernels (channel counts / pointers that rule out 16-byte loads, packed-layout padding): run dsn_conv2d_wgrad.
// workspace must be 16-byte aligned and stay untouched until dsn_conv2d_wgrad_run has executed.
extern "C" int dsn_conv2d_wgrad_plan(const dsn_tensor* x, const dsn_tensor* dy, float* dw, int32_t ci_pad, int32_t oihw,
                                     const dsn_conv_params* p, void* workspace, int64_t workspace_bytes, void* job_out) {
    DSN_CHECK_ARG(job_out, "conv wgrad plan: null job");
    WJob job;
    bool vl = false;
    int rc = make_job(x, dy, dw, ci_pad, oihw, p, workspace, workspace_bytes, &job, &vl);
    if (rc) return rc;
    if (!vl || (job.g.S > 1 && !job.g.oihw && job.g.Cip != job.g.Ci) || ((uintptr_t)job.out % 16) != 0)
        DSN_FAIL(DSN_EUNSUPPORTED, "conv wgrad plan: layer needs the single-layer path");
    *(WJob*)job_out = job;
    return DSN_OK;
}

// jobs_host: n planned jobs, contiguous.  Assigns every job its first block in each of the three launches (in place) and
// returns the grid sizes + totals in launch_out[10] = {grid per-tap, grid all-taps, grid reduce, dtype, flops, bytes,
// reduce bytes, any deferred x, grid 128-tile, grid halo-tile} (doubles).  Upload the array AFTER this call.
extern "C" int dsn_conv2d_wgrad_plan_finish(void* jobs_host, int32_t n, double* launch_out) {
    DSN_CHECK_ARG(jobs_host && n > 0 && launch_out, "conv wgrad plan_finish: bad arguments");
    WJob* jobs = (WJob*)jobs_host;
    // longest blocks first: a block's life is its pixel range, and the hardware dispatches blocks in index order -- heavy
    // jobs at the end of the grid would leave a tail of a few long blocks on an otherwise drained chip
    std::stable_sort(jobs, jobs + n, [](const WJob& a, const WJob& b) { return a.g.ppb > b.g.ppb; });
    int64_t start[6] = {0, 0, 0, 0, 0, 0};
    double flops = 0, bytes = 0, rbytes = 0;
    for (int i = 0; i < n; ++i) {
        DSN_CHECK_ARG(jobs[i].dtype == jobs[0].dtype, "conv wgrad plan_finish: mixed dtypes in one queue");
        for (int k = 0; k < 6; ++k) {
            jobs[i].start[k] = (int32_t)start[k];
            start[k] += jobs[i].blocks[k];
        }
        flops += jobs[i].flops;
        bytes += jobs[i].bytes;
        if (jobs[i].g.S > 1) rbytes += (double)(jobs[i].g.S + 1) * jobs[i].n_out * 4;
    }
    for (int k = 0; k < 6; ++k)
        DSN_CHECK_ARG(start[k] < (1ll << 31), "conv wgrad plan_finish: too many blocks");
    launch_out[0] = (double)start[0]; launch_out[1] = (double)start[1]; launch_out[2] = (double)start[2];
    launch_out[3] = (double)jobs[0].dtype; launch_out[4] = flops; launch_out[5] = bytes; launch_out[6] = rbytes;
    launch_out[7] = (double)start[5];       // grid of the ping-pong kernel-row launch
    launch_out[8] = (double)start[3];       // grid of the 128 x 128-tile launch
    launch_out[9] = (double)start[4];       // grid of the halo-tile all-taps launch
    static const bool dump = getenv("DSN_WGRAD_DUMP") && atoi(getenv("DSN_WGRAD_DUMP"));     // the plan, one line per job (stderr)
    if (dump) {
        static const char* kinds[6] = {"tap64", "alltaps", "-", "tap128", "halo", "pprow"};
        double kb[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < n; ++i) {
            const WGeom& g = jobs[i].g;
            fprintf(stderr, "wgrad job %2d %-7s %4d->%4d k%dx%d s%d out %3dx%3d  S %3d ppb %6d tiles %dx%d blocks %4d  operands %6.1f MB slabs %6.1f MB\n",
                    i, kinds[jobs[i].kind], g.Ci, g.Co, g.KH, g.KW, g.stride, g.Ho, g.Wo, g.S, g.ppb, g.tiles_co, g.tiles_ci,
                    jobs[i].blocks[jobs[i].kind], jobs[i].bytes / 1e6, g.S > 1 ? (double)g.S * jobs[i].n_out * 4 / 1e6 : 0.0);
            kb[jobs[i].kind] += jobs[i].bytes;
        }
        fprintf(stderr, "wgrad plan: operands by kind: tap64 %.1f MB, tap128 %.1f MB, halo %.1f MB, alltaps %.1f MB, pprow %.1f MB\n", kb[0] / 1e6,
                kb[3] / 1e6, kb[4] / 1e6, kb[1] / 1e6, kb[5] / 1e6);
    }
    return DSN_OK;
}

// jobs_dev: the device copy of the finished plan.  Three launches: per-tap blocks, all-taps blocks, slab reductions.
// (The gather launches -- per-tap 64- / 128-wide tiles, halo-tile, all-taps -- are independent of each other; forking them onto side
//  streams so that their tails overlap was measured: 4.81 -> 6.57 ms per config-3 step, 23.1 -> 25.6 ms on config 5.  Concurrent
//  queues do not interleave blocks the way one grid does on this part; they stay on one stream.)
extern "C" int dsn_conv2d_wgrad_run(const void* jobs_dev, int32_t n, const double* launch, void* stream) {
    DSN_CHECK_ARG(jobs_dev && n > 0 && launch, "conv wgrad run: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const WJob* jobs = (const WJob*)jobs_dev;
    const int g0 = (int)launch[0], g1 = (int)launch[1], g2 = (int)launch[2], dtype = (int)launch[3];
    const int g3 = (int)launch[8], g4 = (int)launch[9], g5 = (int)launch[7];
    {
        ProfScope prof(KID_WGRAD + (dtype == DSN_BF16 ? 1 : 0), launch[4], launch[5], st);
        if (g5 > 0) {      // (first: its blocks are the longest)
            DSN_CHECK_ARG(dtype == DSN_BF16, "conv wgrad run: ping-pong jobs are bf16 only");
            DSN_LDS_ATTR(wgrad_pp_grouped_kernel, WPP_LDS);
            hipLaunchKernelGGL(wgrad_pp_grouped_kernel, dim3(g5), dim3(512), WPP_LDS, st, jobs, n);
        }
        if (g0 > 0) {
            static const int gpk = [] { const char* e = getenv("DSN_WGRAD_GPK"); return (e && atoi(e) == 64) ? 64 : 32; }();
            if (dtype == DSN_F32)
                hipLaunchKernelGGL((wgrad_grouped_kernel<float, 32>), dim3(g0), dim3(256), 0, st, jobs, n);
            else if (gpk == 64)
                hipLaunchKernelGGL((wgrad_grouped_kernel<bf16_t, 64>), dim3(g0), dim3(256), 0, st, jobs, n);
            else
                hipLaunchKernelGGL((wgrad_grouped_kernel<bf16_t, 32>), dim3(g0), dim3(256), 0, st, jobs, n);
        }
        if (g3 > 0) {
            DSN_CHECK_ARG(dtype == DSN_BF16, "conv wgrad run: 128-tile jobs are bf16 only");
            static const int pk128 = [] { const char* e = getenv("DSN_WGRAD_T128_PK"); return (e && atoi(e) == 64) ? 64 : 32; }();
            if (pk128 == 64) hipLaunchKernelGGL(wgrad128_grouped_kernel<64>, dim3(g3), dim3(256), 0, st, jobs, n);
            else hipLaunchKernelGGL(wgrad128_grouped_kernel<32>, dim3(g3), dim3(256), 0, st, jobs, n);
        }
        if (g4 > 0) {
            DSN_CHECK_ARG(dtype == DSN_BF16, "conv wgrad run: halo-tile jobs are bf16 only");
            hipLaunchKernelGGL(wgrad_halo_grouped_kernel, dim3(g4), dim3(256), 0, st, jobs, n);
        }
        if (g1 > 0) {
            DSN_CHECK_ARG(dtype == DSN_BF16, "conv wgrad run: all-taps jobs are bf16 only");
            { const int a_ = alltaps_attr_once(); if (a_ != DSN_OK) return a_; }
            hipLaunchKernelGGL(wgrad_alltaps_grouped_kernel, dim3(g1), dim3(256), ALLTAPS_LDS, st, jobs, n);
        }
    }
    DSN_LAUNCH_CHECK("conv wgrad grouped");
    if (g2 > 0) {
        ProfScope prof(KID_WGRAD_REDUCE, 0.0, launch[6], st);
        hipLaunchKernelGGL(wgrad_reduce_grouped_kernel, dim3(g2), dim3(256), 0, st, jobs, n);
        DSN_LAUNCH_CHECK("conv wgrad grouped reduce");
    }
    return DSN_OK;
}
