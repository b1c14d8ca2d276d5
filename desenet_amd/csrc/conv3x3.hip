// 3x3 / stride-1 convolution (any dilation, 'same' padding) on MFMA with the input operand held in an LDS HALO TILE -- forward and
// data gradient of Bottleneck.cv2 (common.py:107), RFB2's branch convolutions (common.py:515-530, dilation 1 / 2 / 3) and FFM's
// 3x3 (common.py:225).
//
// Why a second kernel next to igemm.hip: the implicit-GEMM kernel re-fetches every input pixel once per filter tap (nine 128-byte
// gathers through L2 per pixel and 64-channel slab) with only two K chunks of prefetch, so a block's life is 9 x nslab dependent
// round trips to L2 / MALL -- measured 14.4 us for 3x3 64 -> 64 @ 8x80x80 (3.8 GFLOP, 13 MB: 10 % of the MFMA rate, 0.9 TB/s).
// Here a block owns a TH x TW patch of output pixels of ONE image and BN output channels:
//   * per 64-channel slab (128 B per pixel) the (TH + 2d) x (TW + 2d) input patch is brought in ONCE by LDS-DMA
//     (buffer_load_dwordx4 ... lds, 8 pixels x 128 B per wave-instruction; out-of-image pixels are out-of-range lanes = zeros),
//     next slab's patch prefetched while the current one is used;
//   * the nine taps are nine MFMA passes over SHIFTED reads of that patch -- row r of the A fragment is halo pixel
//     (ty + ky*d, tx + kx*d): no global address arithmetic, no A traffic inside the tap loop;
//   * the weights of (slab, tap) stream through a 3-stage LDS-DMA ring (they are shared by every block: L2 hits);
//   * one raw s_barrier per tap, counted s_waitcnt vmcnt, no ordinary global load inside the loop (cdna_hip_programming.md 5).
// LDS rows are 128 B = 8 slots of 16 B, slot s of weight row r at s ^ ((r >> 1) & 7) and of halo pixel (hy, hx) at hslot(s, hx);
// LDS-DMA writes lane-linearly, so the swizzle is
// applied on the SOURCE side (the lane that owns physical slot v fetches logical slot v ^ swz(p)).
// Data gradient = the same kernel over dy with the taps mirrored (tap k reads halo offset (2 - k) * d) and the weights in the
// [Ci][KH][KW][Co] layout.  Epilogue as igemm.hip: bias, activation, residual, accumulate, BatchNorm partial sums, 16-byte stores.
#include <stdlib.h>

#include "common.h"

namespace {

struct HGeom {
    int32_t N, H, W;          // images, map (source and destination have the same size)
    int32_t Cs, Cd;           // source / destination channels
    int32_t d;                // dilation (= padding)
    int32_t flip;             // 1: data gradient (mirrored taps)
    int32_t tiles_y, tiles_x, tiles_n;
    int32_t act, accumulate;
    int64_t sld, dld, rld;
    uint32_t src_bytes, w_bytes;
    int32_t nslab;
    int32_t HH, HW, NP;       // halo rows / columns / pixels
    int32_t halo_stride;      // bytes between the two halo buffers (0: single buffer)
};

template <typename T> struct HMma;
template <> struct HMma<float> {
    static constexpr int VEC = 4;
    __device__ static __forceinline__ void run(f32x4& acc, const u32x4& a, const u32x4& b) {
        const f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0], bf[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1], bf[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[2], bf[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[3], bf[3], acc, 0, 0, 0);
    }
};
template <> struct HMma<bf16_t> {
    static constexpr int VEC = 8;
    __device__ static __forceinline__ void run(f32x4& acc, const u32x4& a, const u32x4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    }
};

constexpr int ROWB = 128;
constexpr int CPAD = 4;
constexpr int NBR = 3;               // stages of the weight ring

// Halo-patch swizzle.  A ds_read_b128 is served in four 16-lane groups that are NOT contiguous ({0-3, 12-15, 20-27}, {4-11, 16-19,
// 28-31}, ...: MI355X_MICROARCH.md, LDS): a group holds 8 lanes with slot s (fg even) and 8 with slot s ^ 1, and of the halo pixels
// they address, the two that share a column (patch rows ry and ry + 1) always sit in lanes of DIFFERENT slot parity.  A halo row is an
// even number of pixels, so a pixel's half of the 256-byte bank row is its column parity.  Keeping bit 0 of the slot and XOR-ing its
// upper two bits with (column >> 1) & 3 therefore gives the 8 lanes of either half 4 distinct column pairs x 2 slot parities = 8
// distinct positions, for every tap offset and both patch shapes.  (Versions one and two used the linear pixel index and
// (column >> 1) ^ (row parity << 2): SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS 1.5 and 1.0 on this kernel, 0.06 in the implicit GEMM.)
__device__ __forceinline__ int hslot(int slot, int hx) { return (((slot >> 1) ^ ((hx >> 1) & 3)) << 1) | (slot & 1); }

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// IH: LDS-DMA instructions per wave that bring in one halo slab (32 pixels each over the 4 waves): covers up to IH * 32 pixels
// MULTI: more than one channel slab -- the next slab's patch is prefetched into a second halo buffer at tap 0 (with a single slab
// there is no second buffer and nothing is prefetched; the counted waits differ accordingly).
template <typename T, int TH, int TW, int MI, int NI, int WGM, int WGN, int IH, bool MULTI>
__global__ __launch_bounds__(256) void conv3x3_halo_kernel(const T* __restrict__ src, const T* __restrict__ wpk,
                                                           const float* __restrict__ bias, const T* __restrict__ res,
                                                           T* __restrict__ dst, const BnAcc fin, const HGeom g, const BnRed br) {
    static_assert(WGM * WGN == 4, "4 waves per block");
    constexpr int BM = WGM * MI * 16, BN = WGN * NI * 16;
    static_assert(BM == TH * TW, "the M tile is the TH x TW patch");
    static_assert(BN % 32 == 0, "weight stages are filled 32 rows per pass");
    constexpr int VEC = HMma<T>::VEC;
    constexpr int KC = ROWB / (int)sizeof(T);
    constexpr int BR = BN / 32;
    constexpr int LDC = BN + CPAD;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // [weight ring: NBR x BN rows][halo buffer 0][halo buffer 1]; the epilogue's fp32 staging tile reuses the front
    unsigned char* sB = smem;
    unsigned char* sH = smem + NBR * BN * ROWB;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int fr = lane & 15, fg = lane >> 4;
    const int tiles_img = g.tiles_y * g.tiles_x;
    const int tile = xcd_remap(blockIdx.x, g.N * tiles_img * g.tiles_n);
    const int tn = tile % g.tiles_n, tmi = tile / g.tiles_n;
    const int n = tmi / tiles_img, trem = tmi - n * tiles_img;
    const int y0 = (trem / g.tiles_x) * TH, x0 = (trem % g.tiles_x) * TW;
    const int n0 = tn * BN;

    const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, g.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wpk, 0, g.w_bytes, 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr;
    constexpr uint32_t OOB = 0xFFFFFFF0u;

    // ---- halo fetch plan: instruction j of this wave covers halo pixels 32*j + 8*wave .. +8, lane -> pixel p, physical slot lane & 7
    uint32_t hoff[IH];
#pragma unroll
    for (int j = 0; j < IH; ++j) {
        const int p = 32 * j + 8 * wave + (lane >> 3);
        const int hy = p / g.HW, hx = p - hy * g.HW;
        const int gy = y0 - g.d + hy, gx = x0 - g.d + hx;
        const bool ok = p < g.NP && (unsigned)gy < (unsigned)g.H && (unsigned)gx < (unsigned)g.W;
        const int ls = hslot(lane & 7, hx);                                 // logical slot this lane fetches (hslot is an involution)
        hoff[j] = ok ? (uint32_t)((((int64_t)n * g.H + gy) * g.W + gx) * g.sld + ls * VEC) * (uint32_t)sizeof(T) : OOB;
    }
    auto load_halo = [&](int slab, int hbuf) {
        const uint32_t add = slab < g.nslab ? (uint32_t)slab * ROWB : OOB;    // (past the last slab: every lane out of range)
#pragma unroll
        for (int j = 0; j < IH; ++j) {
            const uint32_t off = (hoff[j] == OOB || add == OOB) ? OOB : hoff[j] + add;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srsrc, (lds_ptr)(sH + hbuf * g.halo_stride + (32 * j + 8 * wave) * ROWB), 16, off, 0, 0, DSN_DMA_AUX);
        }
    };
    // ---- weight fetch plan: thread (row r0 + 32 i, physical slot tid & 7) of stage `buf`
    const int r0 = tid >> 3;
    const int lsb = (tid & 7) ^ ((r0 >> 1) & 7);
    uint32_t woff[BR];
#pragma unroll
    for (int i = 0; i < BR; ++i) {
        const int nn = n0 + r0 + 32 * i;
        woff[i] = nn < g.Cd ? (uint32_t)((int64_t)nn * 9 * g.Cs + lsb * VEC) * (uint32_t)sizeof(T) : OOB;
    }
    // Per-wave facts about DMA instructions whose lanes are ALL out of range: they retire at once (tools/exp/oob_order.hip) and must not
    // stand for operations in flight in the counted waits below.  hdead: such halo DMAs per slab (border patches: rows / columns of
    // the halo outside the image, the padding pixels past the patch); wfull: every weight DMA of this wave is a real one (false:
    // rows past Cd of a partial channel tile -- that wave drains at every wait).
    int hdead = 0;
#pragma unroll
    for (int j = 0; j < IH; ++j) hdead += (__ballot(hoff[j] != OOB) == 0ull) ? 1 : 0;
    hdead = __builtin_amdgcn_readfirstlane(hdead);
    bool wfull = true;
#pragma unroll
    for (int i = 0; i < BR; ++i) wfull = wfull && __ballot(woff[i] != OOB) != 0ull;
    const int F = g.nslab * 9;
    auto load_w = [&](int f, int buf) {           // flat index f = slab * 9 + tap: K offset = tap * Cs + slab * KC
        const int s = f / 9, t = f - s * 9;
        const uint32_t add = f < F ? (uint32_t)(t * g.Cs + s * KC) * (uint32_t)sizeof(T) : OOB;
#pragma unroll
        for (int i = 0; i < BR; ++i) {
            const uint32_t off = (woff[i] == OOB || add == OOB) ? OOB : woff[i] + add;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_ptr)(sB + (buf * BN + 32 * i + 8 * wave) * ROWB), 16, off, 0, 0, DSN_DMA_AUX);
        }
    };

    // ---- fragment addressing ------------------------------------------------------------------------------------------------
    int hp0[MI], hy0[MI], hx0[MI];                 // halo pixel (and its row / column) of A row (wm*MI + i)*16 + fr at tap offset (0, 0)
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int r = (wm * MI + i) * 16 + fr;
        hy0[i] = r / TW;
        hx0[i] = r % TW;
        hp0[i] = hy0[i] * g.HW + hx0[i];
    }
    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](const unsigned char* hb, int oy, int ox, int bbuf) {
        const int toff = oy * g.HW + ox;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            u32x4 fa[MI], fb[NI];
            const int slot = 4 * h + fg;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int p = hp0[i] + toff;
                fa[i] = *reinterpret_cast<const u32x4*>(hb + p * ROWB + (hslot(slot, hx0[i] + ox) << 4));
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int r = (wn * NI + j) * 16 + fr;
                fb[j] = *reinterpret_cast<const u32x4*>(sB + (bbuf * BN + r) * ROWB + ((slot ^ ((r >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) HMma<T>::run(acc[i][j], fa[i], fb[j]);
        }
    };

    // ---- main loop: slabs x 9 taps, one barrier per tap -------------------------------------------------------------------------
    // Issue order: prologue halo(0), W(0), W(1); iteration f = (s, t) after its barrier: W(f + 2), and at t == 0 halo(s + 1).
    // Real LDS-DMAs complete in issue order (tools/exp/oob_order.hip), so before the barrier of (s, t) a wave may leave outstanding: W(f + 1) always, plus halo(s + 1) while it
    // is younger than W(f) (t == 1) or sits between W(f) and W(f + 1) (t == 2).
    load_halo(0, 0);
    load_w(0, 0);
    load_w(1, 1);
    int wbuf = 0, lbuf = 2;
    for (int s = 0; s < g.nslab; ++s) {
        const unsigned char* hb = sH + (s & 1) * g.halo_stride;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            // outstanding vector-memory operations a wave may leave behind the ones it needs (issue order of iteration (s, 0):
            // W(f + 2), [MULTI: halo(s + 1)]; W(f + 2) alone in the other iterations)
            constexpr int EX = MULTI ? IH : 0;
            // lgkmcnt(0): the compiler may sink the MFMAs of the previous tap -- and with them the wait for their fragment reads --
            // BELOW this barrier (registers only: the "memory" clobbers do not hold them), which would leave ds_reads of ring stage
            // `lbuf` merely issued when another wave's LDS-DMA starts overwriting it after the barrier.  Observed: sporadic wrong
            // patches on >= 800-block bf16 launches (tests/test_kernels_gpu.py: the 8 x 70 x 67 case).
            // Round 4 (tools/exp/oob_order.hip): an LDS-DMA whose lanes are ALL out of range retires at once, so the padding operations
            // of the LAST slab -- halo(s + 1) and, at the last tap, W(f + 1) -- must not be counted as operations still in flight.
            const bool lasts = s == g.nslab - 1;
            if ((lasts && t == 8) || !wfull) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            else if (EX > 0 && (t == 1 || t == 2) && !lasts) {
                // W(f + 1) and the REAL DMAs of halo(s + 1): BR + IH - hdead (a per-wave constant: a short compare chain)
#define DSN_HWAIT(k) else if (EX >= k && hdead == k) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(BR + (EX >= k ? EX - k : 0)) : "memory");
                if (hdead == 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(BR + EX) : "memory");
                DSN_HWAIT(1) DSN_HWAIT(2) DSN_HWAIT(3) DSN_HWAIT(4) DSN_HWAIT(5) DSN_HWAIT(6) DSN_HWAIT(7) DSN_HWAIT(8)
                else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(BR) : "memory");
#undef DSN_HWAIT
            } else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(BR) : "memory");
            // (the "memory" clobber already keeps every ds_read of the previous tap above the wait; sched_barrier pins the whole
            //  issue order at this point -- cdna_hip_programming.md 5: place reads by the count, not by clean runs)
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            load_w(s * 9 + t + 2, lbuf);
            if (MULTI && t == 0) load_halo(s + 1, (s + 1) & 1);      // (past the last slab: zeros into the buffer slab s - 1 used)
            const int ky = t / 3, kx = t - ky * 3;
            const int oy = (g.flip ? 2 - ky : ky) * g.d, ox = (g.flip ? 2 - kx : kx) * g.d;
            compute(hb, oy, ox, wbuf);
            wbuf = wbuf + 1 == NBR ? 0 : wbuf + 1;
            lbuf = lbuf + 1 == NBR ? 0 : lbuf + 1;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- epilogue: act(acc + bias) staged as fp32 [BM][BN + CPAD] (row r = patch pixel (r / TW, r % TW)) ------------------------
    constexpr int VPR = BN / VEC;
    constexpr int NIT = (BM * VPR + 255) / 256;
    const int vh = (g.H - y0 < TH) ? g.H - y0 : TH, vw = (g.W - x0 < TW) ? g.W - x0 : TW;     // valid part of the patch
    BnRedLane<T, VEC, NIT> bl;                      // (dgrad that completes dz of a BatchNorm block: common.h)
    if (br.nseg) {
        bl.init(br, n0 + (tid % VPR) * VEC);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * 256, rl = idx / VPR, ty = rl / TW, tx = rl - ty * TW;
            const bool ok = idx < BM * VPR && ty < vh && tx < vw && n0 + (idx - rl * VPR) * VEC < g.Cd;
            bl.prefetch(it, ok ? ((int64_t)n * g.H + y0 + ty) * g.W + x0 + tx : -1);
        }
    }
    float* sC = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int cl = (wn * NI + j) * 16 + fr;
        const int col = n0 + cl;
        const float bv = (bias && col < g.Cd) ? bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i)
        {
            float v4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v4[e] = acc[i][j][e] + bv;
            apply_act_vec<4>(v4, g.act);          // (the activation code is tested per 4 values, not per value)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int rl = (wm * MI + i) * 16 + fg * 4 + e;
                sC[rl * LDC + cl] = v4[e];
            }
        }
    }
    __syncthreads();
    if (fin.acc) {
        float* red = sC + BM * LDC;
        constexpr int TYS = 256 / BN > 0 ? 256 / BN : 1;
        const int tx = tid % BN, ty = tid / BN;
        float s = 0.f, ss = 0.f;
        if (ty < TYS) {
            for (int r = ty; r < BM; r += TYS) {
                if (r / TW < vh && r % TW < vw) {
                    const float val = sC[r * LDC + tx];
                    s += val;
                    ss += val * val;
                }
            }
            red[ty * BN + tx] = s;
            red[256 + ty * BN + tx] = ss;
        }
        __syncthreads();
        if (tid < BN && n0 + tid < g.Cd) {
            float t0 = 0.f, t1 = 0.f;
#pragma unroll
            for (int t = 0; t < TYS; ++t) {
                t0 += red[t * BN + tid];
                t1 += red[256 + t * BN + tid];
            }
            bn_acc_add(fin, tmi, n0 + tid, t0, t1);
        }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = tid + it * 256;
        if (idx >= BM * VPR) break;
        const int rl = idx / VPR, cv = idx - rl * VPR;
        const int col = n0 + cv * VEC;
        const int ty = rl / TW, tx = rl - ty * TW;
        if (ty >= vh || tx >= vw || col >= g.Cd) continue;
        const int64_t row = ((int64_t)n * g.H + y0 + ty) * g.W + x0 + tx;
        float vals[VEC];
#pragma unroll
        for (int e = 0; e < VEC; e += 4) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(sC + rl * LDC + cv * VEC + e);
            vals[e] = t[0]; vals[e + 1] = t[1]; vals[e + 2] = t[2]; vals[e + 3] = t[3];
        }
        T* o = dst + row * g.dld + col;
        if (res) {
            T rv[VEC];
            *reinterpret_cast<u32x4*>(rv) = *reinterpret_cast<const u32x4*>(res + row * g.rld + col);
#pragma unroll
            for (int e = 0; e < VEC; ++e) vals[e] += to_f32<T>(rv[e]);
        }
        if (g.accumulate) {
            T ov[VEC];
            *reinterpret_cast<u32x4*>(ov) = *reinterpret_cast<const u32x4*>(o);
#pragma unroll
            for (int e = 0; e < VEC; ++e) vals[e] += to_f32<T>(ov[e]);
        }
        T outv[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) outv[e] = from_f32<T>(vals[e]);
        *reinterpret_cast<u32x4*>(o) = *reinterpret_cast<u32x4*>(outv);
        if (br.nseg) bl.add(it, outv);
    }
    if (br.nseg) {
        __syncthreads();
        bl.template finish<VPR>(br, sC, tmi, n0, g.Cd, 0);
    }
}

template <typename T, int TH, int TW, int MI, int NI, int WGM, int WGN, int IH, bool MULTI>
int launch_halo1(const dsn_tensor* s, const void* w, const float* bias, const dsn_tensor* r, const dsn_tensor* d, HGeom g,
                const BnAcc& fin, hipStream_t st, const BnRed* brp) {
    constexpr int BM = WGM * MI * 16, BN = WGN * NI * 16;
    g.tiles_y = (g.H + TH - 1) / TH;
    g.tiles_x = (g.W + TW - 1) / TW;
    g.tiles_n = (g.Cd + BN - 1) / BN;
    g.HH = TH + 2 * g.d; g.HW = TW + 2 * g.d; g.NP = g.HH * g.HW;
    if (g.NP > IH * 32) return 1;                                              // dilation too large for this instantiation
    const size_t halo_bytes = (size_t)IH * 32 * ROWB;
    g.halo_stride = g.nslab > 1 ? (int32_t)halo_bytes : 0;
    const size_t loop = (size_t)NBR * BN * ROWB + halo_bytes * (g.nslab > 1 ? 2 : 1);
    const size_t epi = (size_t)BM * (BN + CPAD) * 4 + 2 * 256 * 4;
    const size_t lds = loop > epi ? loop : epi;
    auto kern = conv3x3_halo_kernel<T, TH, TW, MI, NI, WGM, WGN, IH, MULTI>;
    DSN_LDS_ATTR(kern, 150 * 1024);
    const int blocks = g.N * g.tiles_y * g.tiles_x * g.tiles_n;
    const double elems = (double)g.N * g.H * g.W * (g.Cs + (double)g.Cd * (1 + (r ? 1 : 0) + (g.accumulate ? 1 : 0)) + bnred_channels(brp)) +
                         9.0 * g.Cs * g.Cd;
    const ProfConv pc("conv3x3_halo_kernel", sizeof(T) == 2, BM, BN, g.flip != 0, 3, 1, g.d, g.Cs, g.Cd, g.N, g.H, g.W);
    ProfScope prof(pc.label, pc.layer, 2.0 * g.N * g.H * g.W * g.Cd * 9.0 * g.Cs, elems * sizeof(T), st);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, st, (const T*)s->ptr, (const T*)w, bias, r ? (const T*)r->ptr : nullptr,
                       (T*)d->ptr, fin, g, brp ? *brp : BnRed{});
    DSN_LAUNCH_CHECK("conv3x3 (halo tile)");
    return DSN_OK;
}

// ---- 1x1 / stride-1 convolution (forward and data gradient: the same GEMM) with BOTH operands fetched in one round trip ---------
// C3's cv1 / cv2 / cv3, Bottleneck.cv1, SPP, the seg head's 1x1s (common.py:103-145,172-185, yolo.py:161-181) have K = 64 .. 256
// input channels: the whole [64 pixels x K] input tile and [64 channels x K] weight tile of a block fit LDS (16 KB per 64-channel
// slab), so every LDS-DMA of the block is issued in its first instructions and the block's life is ONE trip to L2 / MALL, then
// NS x 8 MFMAs per wave, then the epilogue -- against prologue address arithmetic, three register stages and a barrier per chunk in
// the implicit-GEMM kernel.  NS = number of 128-byte channel slabs (1 .. 4), a template parameter so that the counted waits are
// immediates.
template <typename T, int MI, int NI, int WGM, int WGN, int NS>
__global__ __launch_bounds__(256) void conv1x1_dma_kernel(const T* __restrict__ src, const T* __restrict__ wpk,
                                                          const float* __restrict__ bias, const T* __restrict__ res,
                                                          T* __restrict__ dst, const BnAcc fin, const HGeom g, const BnRed br) {
    static_assert(WGM * WGN == 4, "4 waves per block");
    constexpr int BM = WGM * MI * 16, BN = WGN * NI * 16;
    static_assert(BM % 32 == 0 && BN % 32 == 0, "stages are filled 32 rows per pass");
    constexpr int VEC = HMma<T>::VEC;
    constexpr int AR = BM / 32, BR = BN / 32, LPC = AR + BR;
    constexpr int LDC = BN + CPAD;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sA = smem;                          // [NS][BM rows][128 B]
    unsigned char* sB = smem + NS * BM * ROWB;         // [NS][BN rows][128 B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int fr = lane & 15, fg = lane >> 4;
    const int64_t M = (int64_t)g.N * g.H * g.W;
    const int tiles_m = (int)((M + BM - 1) / BM);
    const int tile = xcd_remap(blockIdx.x, tiles_m * g.tiles_n);
    const int tn = tile % g.tiles_n, tm = tile / g.tiles_n;
    const int64_t m0 = (int64_t)tm * BM;
    const int n0 = tn * BN;
    const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, g.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wpk, 0, g.w_bytes, 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr;
    constexpr uint32_t OOB = 0xFFFFFFF0u;
    const int r0 = tid >> 3;
    const int ls = (tid & 7) ^ ((r0 >> 1) & 7);        // logical slot fetched by the lane that owns physical slot tid & 7
    uint32_t aoff[AR], boff[BR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const int64_t m = m0 + r0 + 32 * i;
        aoff[i] = m < M ? (uint32_t)(m * g.sld + ls * VEC) * (uint32_t)sizeof(T) : OOB;
    }
#pragma unroll
    for (int i = 0; i < BR; ++i) {
        const int nn = n0 + r0 + 32 * i;
        boff[i] = nn < g.Cd ? (uint32_t)((int64_t)nn * g.Cs + ls * VEC) * (uint32_t)sizeof(T) : OOB;
    }
    // (the LDS-DMA builtin lives in a lambda: it does not exist for the host pass, and a kernel body that names it directly is
    //  silently dropped there -- no device stub, an undefined kernel handle at load time)
    auto fetch_all = [&]() {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                const uint32_t off = aoff[i] == OOB ? OOB : aoff[i] + (uint32_t)(s * ROWB);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srsrc, (lds_ptr)(sA + ((s * BM) + 32 * i + 8 * wave) * ROWB), 16, off, 0, 0, DSN_DMA_AUX);
            }
#pragma unroll
            for (int i = 0; i < BR; ++i) {
                const uint32_t off = boff[i] == OOB ? OOB : boff[i] + (uint32_t)(s * ROWB);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_ptr)(sB + ((s * BN) + 32 * i + 8 * wave) * ROWB), 16, off, 0, 0, DSN_DMA_AUX);
            }
        }
    };
    fetch_all();
    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto slab = [&](int s) {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            u32x4 fa[MI], fb[NI];
            const int slot = 4 * h + fg;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int r = (wm * MI + i) * 16 + fr;
                fa[i] = *reinterpret_cast<const u32x4*>(sA + (s * BM + r) * ROWB + ((slot ^ ((r >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int r = (wn * NI + j) * 16 + fr;
                fb[j] = *reinterpret_cast<const u32x4*>(sB + (s * BN + r) * ROWB + ((slot ^ ((r >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) HMma<T>::run(acc[i][j], fa[i], fb[j]);
        }
    };
    // A wave one of whose DMA instructions has ALL lanes out of range (rows past M in the last pixel tile, channels past Cd in a
    // partial tile) drains instead: such a DMA retires at once (round 4, tools/exp/oob_order.hip), so the counts below would let
    // the wave go with real DMAs of the slab it needs still in flight -- and its share of the weight rows is read by every wave.
    {
        bool full = true;
#pragma unroll
        for (int i = 0; i < AR; ++i) full = full && __ballot(aoff[i] != OOB) != 0ull;
#pragma unroll
        for (int i = 0; i < BR; ++i) full = full && __ballot(boff[i] != OOB) != 0ull;
        if (!full) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // (the waits are immediates: one statement per slab; slab s may leave the LDS-DMA of the NS - 1 - s younger slabs in flight)
    if constexpr (NS > 0) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 1) * LPC) : "memory"); slab(0); }
    if constexpr (NS > 1) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * LPC) : "memory"); slab(1); }
    if constexpr (NS > 2) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 3) * LPC) : "memory"); slab(2); }
    if constexpr (NS > 3) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 4) * LPC) : "memory"); slab(3); }
    __syncthreads();
    // ---- epilogue (as igemm.hip): act(acc + bias) staged as fp32, BatchNorm partial sums, 16-byte stores ------------------------
    constexpr int VPR = BN / VEC;
    constexpr int NIT = (BM * VPR + 255) / 256;
    const int rows = (M - m0 < BM) ? (int)(M - m0) : BM;
    BnRedLane<T, VEC, NIT> bl;                      // (dgrad that completes dz of a BatchNorm block: common.h)
    if (br.nseg) {
        bl.init(br, n0 + (tid % VPR) * VEC);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * 256, rl = idx / VPR;
            const bool ok = idx < BM * VPR && rl < rows && n0 + (idx - rl * VPR) * VEC < g.Cd;
            bl.prefetch(it, ok ? m0 + rl : -1);
        }
    }
    float* sC = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int cl = (wn * NI + j) * 16 + fr;
        const int col = n0 + cl;
        const float bv = (bias && col < g.Cd) ? bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i)
        {
            float v4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v4[e] = acc[i][j][e] + bv;
            apply_act_vec<4>(v4, g.act);          // (the activation code is tested per 4 values, not per value)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int rl = (wm * MI + i) * 16 + fg * 4 + e;
                sC[rl * LDC + cl] = v4[e];
            }
        }
    }
    __syncthreads();
    if (fin.acc) {
        float* red = sC + BM * LDC;
        constexpr int TYS = 256 / BN > 0 ? 256 / BN : 1;
        const int tx = tid % BN, ty = tid / BN;
        float s = 0.f, ss = 0.f;
        if (ty < TYS) {
            for (int r = ty; r < rows; r += TYS) {
                const float val = sC[r * LDC + tx];
                s += val;
                ss += val * val;
            }
            red[ty * BN + tx] = s;
            red[256 + ty * BN + tx] = ss;
        }
        __syncthreads();
        if (tid < BN && n0 + tid < g.Cd) {
            float t0 = 0.f, t1 = 0.f;
#pragma unroll
            for (int t = 0; t < TYS; ++t) {
                t0 += red[t * BN + tid];
                t1 += red[256 + t * BN + tid];
            }
            bn_acc_add(fin, tm, n0 + tid, t0, t1);
        }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = tid + it * 256;
        if (idx >= BM * VPR) break;
        const int rl = idx / VPR, cv = idx - rl * VPR;
        const int col = n0 + cv * VEC;
        if (rl >= rows || col >= g.Cd) continue;
        const int64_t row = m0 + rl;
        float vals[VEC];
#pragma unroll
        for (int e = 0; e < VEC; e += 4) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(sC + rl * LDC + cv * VEC + e);
            vals[e] = t[0]; vals[e + 1] = t[1]; vals[e + 2] = t[2]; vals[e + 3] = t[3];
        }
        T* o = dst + row * g.dld + col;
        if (res) {
            T rv[VEC];
            *reinterpret_cast<u32x4*>(rv) = *reinterpret_cast<const u32x4*>(res + row * g.rld + col);
#pragma unroll
            for (int e = 0; e < VEC; ++e) vals[e] += to_f32<T>(rv[e]);
        }
        if (g.accumulate) {
            T ov[VEC];
            *reinterpret_cast<u32x4*>(ov) = *reinterpret_cast<const u32x4*>(o);
#pragma unroll
            for (int e = 0; e < VEC; ++e) vals[e] += to_f32<T>(ov[e]);
        }
        T outv[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) outv[e] = from_f32<T>(vals[e]);
        *reinterpret_cast<u32x4*>(o) = *reinterpret_cast<u32x4*>(outv);
        if (br.nseg) bl.add(it, outv);
    }
    if (br.nseg) {
        __syncthreads();
        bl.template finish<VPR>(br, sC, tm, n0, g.Cd, 0);
    }
}

template <typename T, int MI, int NI, int WGM, int WGN, int NS>
int launch_1x1(const dsn_tensor* s, const void* w, const float* bias, const dsn_tensor* r, const dsn_tensor* d, HGeom g,
               const BnAcc& fin, int is_dgrad, hipStream_t st, const BnRed* brp) {
    constexpr int BM = WGM * MI * 16, BN = WGN * NI * 16;
    const int64_t M = (int64_t)g.N * g.H * g.W;
    g.tiles_n = (g.Cd + BN - 1) / BN;
    const size_t loop = (size_t)NS * (BM + BN) * ROWB;
    const size_t epi = (size_t)BM * (BN + CPAD) * 4 + 2 * 256 * 4;
    const size_t lds = loop > epi ? loop : epi;
    auto kern = conv1x1_dma_kernel<T, MI, NI, WGM, WGN, NS>;
    DSN_LDS_ATTR(kern, 150 * 1024);
    const int blocks = (int)((M + BM - 1) / BM) * g.tiles_n;
    const double elems = (double)M * (g.Cs + (double)g.Cd * (1 + (r ? 1 : 0) + (g.accumulate ? 1 : 0)) + bnred_channels(brp)) + (double)g.Cs * g.Cd;
    const ProfConv pc("conv1x1_dma_kernel", sizeof(T) == 2, BM, BN, is_dgrad != 0, 1, 1, 1, g.Cs, g.Cd, g.N, g.H, g.W);
    ProfScope prof(pc.label, pc.layer, 2.0 * M * g.Cd * g.Cs, elems * sizeof(T), st);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, st, (const T*)s->ptr, (const T*)w, bias, r ? (const T*)r->ptr : nullptr,
                       (T*)d->ptr, fin, g, brp ? *brp : BnRed{});
    DSN_LAUNCH_CHECK("conv1x1 (one-trip LDS-DMA)");
    return DSN_OK;
}

template <typename T, int MI, int NI, int WGM, int WGN>
int launch_1x1_ns(const dsn_tensor* s, const void* w, const float* bias, const dsn_tensor* r, const dsn_tensor* d, const HGeom& g,
                  const BnAcc& fin, int is_dgrad, hipStream_t st, const BnRed* brp) {
    switch (g.nslab) {
        case 1: return launch_1x1<T, MI, NI, WGM, WGN, 1>(s, w, bias, r, d, g, fin, is_dgrad, st, brp);
        case 2: return launch_1x1<T, MI, NI, WGM, WGN, 2>(s, w, bias, r, d, g, fin, is_dgrad, st, brp);
        case 3: return launch_1x1<T, MI, NI, WGM, WGN, 3>(s, w, bias, r, d, g, fin, is_dgrad, st, brp);
        case 4: return launch_1x1<T, MI, NI, WGM, WGN, 4>(s, w, bias, r, d, g, fin, is_dgrad, st, brp);
        default: return 1;
    }
}

template <typename T, int TH, int TW, int MI, int NI, int WGM, int WGN, int IH>
int launch_halo(const dsn_tensor* s, const void* w, const float* bias, const dsn_tensor* r, const dsn_tensor* d, HGeom g,
                const BnAcc& fin, hipStream_t st, const BnRed* brp) {
    if (g.nslab > 1) return launch_halo1<T, TH, TW, MI, NI, WGM, WGN, IH, true>(s, w, bias, r, d, g, fin, st, brp);
    return launch_halo1<T, TH, TW, MI, NI, WGM, WGN, IH, false>(s, w, bias, r, d, g, fin, st, brp);
}

}  // namespace

// Called by the convolution entry points of igemm.hip before they fall back to the implicit-GEMM kernel.  Returns 1 when the layer
// is not one this kernel takes (then nothing was launched), 0 when it ran, < 0 / hipError on failure.
int dsn_conv3x3_halo_try(const dsn_tensor* s, const void* w, const float* bias, const dsn_tensor* r, const dsn_tensor* d,
                         const dsn_conv_params* p, int is_dgrad, const BnAcc* finp, void* stream, const BnRed* br) {
    static const int mode = [] { const char* e = getenv("DSN_HALO"); return e ? atoi(e) : 1; }();       // 0: never
    if (!mode) return 1;
    if (p->kh != 3 || p->kw != 3 || p->stride != 1 || p->pad != p->dil || p->dil < 1 || p->dil > 3) return 1;
    if (s->h != d->h || s->w != d->w || s->n != d->n || s->dtype != d->dtype) return 1;
    const int es = s->dtype == DSN_F32 ? 4 : 2, vec = 16 / es, kc = ROWB / es;
    if (s->c % kc != 0 || d->c % vec != 0 || s->ldc % vec != 0 || d->ldc % vec != 0) return 1;
    if (((uintptr_t)s->ptr | (uintptr_t)d->ptr | (uintptr_t)w) % 16 != 0) return 1;
    if (r && (r->ldc % vec != 0 || (uintptr_t)r->ptr % 16 != 0)) return 1;
    const int64_t sb = ((npix(s) - 1) * s->ldc + s->c) * es, wb = (int64_t)d->c * 9 * s->c * es;
    if (sb >= (1ll << 31) || wb >= (1ll << 31) || npix(d) * d->ldc * es >= (1ll << 40)) return 1;
    // patches of 8 x 8 (or 8 x 16) output pixels; ragged border patches are masked (DSN_HALO_RAGGED=0: leave maps that are not
    // multiples of 8 to the implicit-GEMM kernel)
    static const int ragged = [] { const char* e = getenv("DSN_HALO_RAGGED"); return e ? atoi(e) : 1; }();
    if (!ragged && (s->h % 8 != 0 || s->w % 8 != 0)) return 1;
    HGeom g{};
    g.N = s->n; g.H = s->h; g.W = s->w; g.Cs = s->c; g.Cd = d->c; g.d = p->dil; g.flip = is_dgrad ? 1 : 0;
    g.act = p->act; g.accumulate = p->accumulate;
    g.sld = s->ldc; g.dld = d->ldc; g.rld = r ? r->ldc : 0;
    g.src_bytes = (uint32_t)sb; g.w_bytes = (uint32_t)wb;
    g.nslab = s->c / kc;
    const BnRed* la = (br && br->nseg > 0) ? br : nullptr;
    BnAcc fin{};
    if (finp) fin = *finp;
    hipStream_t st = (hipStream_t)stream;
    const int64_t px = npix(d);
    // tile choice: 8 x 16 x 128 where the layer is large enough to fill the chip with such blocks (config 5's maps), else 8 x 8 x 64
    // (measured: back to back on one layer the large patch wins from ~200 blocks up -- config 5's 512 -> 512 @40: 49.6 -> 43.1 us -- but
    //  inside the training step, with cold weights, only the very large grids keep the gain: 24.72 ms per config-5 step with the
    //  threshold at 192 blocks, 24.31 at 1024, 24.34 with the large patch off.  DSN_HALO_BIG_MIN moves the threshold.)
    static const int big_min = [] { const char* e = getenv("DSN_HALO_BIG_MIN"); return e ? atoi(e) : 1024; }();
    const bool big = mode != 3 && d->c >= 128 && s->w % 16 == 0 && s->h % 8 == 0 &&
                     (mode == 2 || (px / 128) * ((d->c + 127) / 128) >= big_min);
    // halo pixels per instantiation = IH * 32: (8+2)^2 = 100 <= 128, (8+6)^2 = 196 <= 224, (8+2) x (16+2) = 180 <= 192
    if (s->dtype == DSN_F32) {
        if (g.d == 1) return launch_halo<float, 8, 8, 2, 2, 2, 2, 4>(s, w, bias, r, d, g, fin, st, la);
        return launch_halo<float, 8, 8, 2, 2, 2, 2, 7>(s, w, bias, r, d, g, fin, st, la);
    }
    if (big && g.d == 1) return launch_halo<bf16_t, 8, 16, 4, 4, 2, 2, 6>(s, w, bias, r, d, g, fin, st, la);
    // 8 x 16 x 64 (round 3): 16 MFMAs per wave between two barriers instead of 8, half as many weight fetches per pixel, still two
    // blocks per CU.  Per layer (config 5, HIP events) 17-18 % faster than 8 x 8 x 64: 128 -> 128 @160 dgrad 78 -> 64 us, 256 -> 256
    // @80 64 -> 52 / 59 -> 50 us.  Under replay the config-5 STEP gains only 0.3-0.6 % (rocprofv3: these launches -0.45 ms, every
    // other kernel of the step 2-4 % slower -- at ~1150 W the part runs at 2.31 GHz, not 2.4: the step is energy-bound); config 3 flat.
    // DSN_HALO_MID_MIN=0 turns it off, larger values set a block-count threshold.
    static const int mid_min = [] { const char* e = getenv("DSN_HALO_MID_MIN"); return e ? atoi(e) : 1; }();
    if (mid_min > 0 && mode != 3 && g.d == 1 && d->c >= 64 && s->w % 16 == 0 && s->h % 8 == 0 && (px / 128) * ((d->c + 63) / 64) >= mid_min)
        return launch_halo<bf16_t, 8, 16, 4, 2, 2, 2, 6>(s, w, bias, r, d, g, fin, st, la);
    if (g.d == 1) return launch_halo<bf16_t, 8, 8, 2, 2, 2, 2, 4>(s, w, bias, r, d, g, fin, st, la);
    return launch_halo<bf16_t, 8, 8, 2, 2, 2, 2, 7>(s, w, bias, r, d, g, fin, st, la);
}

// The same for 1x1 / stride-1 layers with at most four 128-byte channel slabs (conv1x1_dma_kernel).
int dsn_conv1x1_dma_try(const dsn_tensor* s, const void* w, const float* bias, const dsn_tensor* r, const dsn_tensor* d,
                        const dsn_conv_params* p, int is_dgrad, const BnAcc* finp, void* stream, const BnRed* br) {
    static const int mode = [] { const char* e = getenv("DSN_DMA1X1"); return e ? atoi(e) : 1; }();       // 0: never
    if (!mode) return 1;
    if (p->kh != 1 || p->kw != 1 || p->stride != 1 || p->pad != 0) return 1;
    if (s->h != d->h || s->w != d->w || s->n != d->n || s->dtype != d->dtype) return 1;
    const int es = s->dtype == DSN_F32 ? 4 : 2, vec = 16 / es, kc = ROWB / es;
    if (s->c % kc != 0 || s->c / kc > 4 || d->c % vec != 0 || s->ldc % vec != 0 || d->ldc % vec != 0) return 1;
    if (((uintptr_t)s->ptr | (uintptr_t)d->ptr | (uintptr_t)w) % 16 != 0) return 1;
    if (r && (r->ldc % vec != 0 || (uintptr_t)r->ptr % 16 != 0)) return 1;
    const int64_t sb = ((npix(s) - 1) * s->ldc + s->c) * es, wb = (int64_t)d->c * s->c * es;
    if (sb >= (1ll << 31) || wb >= (1ll << 31)) return 1;
    HGeom g{};
    g.N = s->n; g.H = s->h; g.W = s->w; g.Cs = s->c; g.Cd = d->c;
    g.act = p->act; g.accumulate = p->accumulate;
    g.sld = s->ldc; g.dld = d->ldc; g.rld = r ? r->ldc : 0;
    g.src_bytes = (uint32_t)sb; g.w_bytes = (uint32_t)wb;
    g.nslab = s->c / kc;
    const BnRed* la = (br && br->nseg > 0) ? br : nullptr;
    BnAcc fin{};
    if (finp) fin = *finp;
    hipStream_t st = (hipStream_t)stream;
    // tile choice as igemm.hip's for these layers: 32 x 64 where 64 x 64 tiles would leave the chip under-filled
    const int64_t tiles64 = ((npix(d) + 63) / 64) * ((d->c + 63) / 64);
    if (d->c < 64 || tiles64 < 256) {
        if (s->dtype == DSN_F32) return launch_1x1_ns<float, 1, 2, 2, 2>(s, w, bias, r, d, g, fin, is_dgrad, st, la);
        return launch_1x1_ns<bf16_t, 1, 2, 2, 2>(s, w, bias, r, d, g, fin, is_dgrad, st, la);
    }
    if (s->dtype == DSN_F32) return launch_1x1_ns<float, 2, 2, 2, 2>(s, w, bias, r, d, g, fin, is_dgrad, st, la);
    return launch_1x1_ns<bf16_t, 2, 2, 2, 2>(s, w, bias, r, d, g, fin, is_dgrad, st, la);
}
