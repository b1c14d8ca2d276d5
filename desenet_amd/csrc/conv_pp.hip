// 3x3 / stride-1 / dilation-1 convolution on MFMA for bf16, BIG block tiles: 16 x 16 output pixels (BM = 256) x BN = 128 or 256
// output channels per block, 512 threads = 8 waves in TWO GROUPS that alternate between a "load" segment (LDS fragment reads +
// LDS-DMA issue) and an MFMA segment -- forward and data gradient of Bottleneck.cv2 (common.py:101-111), FFM's 3x3 (common.py:222-226,
// via yolo.py:181) and every other k3 / s1 / d1 layer whose input channels are whole 64-channel slabs.
//
// Why a third 3x3 kernel: the halo-tile kernel (conv3x3.hip) and the implicit GEMM run four waves with 64 x 64 tiles and ONE
// barrier per tap; every wave reads its fragments and then issues its MFMAs, so the matrix pipe of a SIMD idles while its wave(s)
// wait for LDS and for the barrier (best launch 818 TFLOP/s = 33 % of the dense bf16 rate; config 5's C3 layers 500-650).  Here a
// SIMD always holds one wave of each group, and the two groups are staggered by one barrier:
//
//     interval   2P            2P+1          2P+2          2P+3
//     group 0    load(P)       MFMA(P)       load(P+1)     MFMA(P+1)
//     group 1    MFMA(P-1)     load(P)       MFMA(P)       load(P+1)
//
// (cdna_hip_programming.md 5, "256^2 8-phase template": per phase {fragment reads + prefetch issue -> s_barrier -> 16 MFMAs ->
// s_barrier}, the second wave row entering one barrier late.)  A phase is 16 x v_mfma_f32_16x16x32_bf16 per wave = a 64 x 64 x 32
// product; a load segment is at most 8 ds_read_b128 + 2 LDS-DMA issues per wave, i.e. <= 128 of the 256 LDS cycles the other group's
// MFMAs take.  K runs over (64-channel slab, tap, 32-channel half): "k-half" j.
//
//   A operand (pixels): per slab the 18 x 18 HALO patch of the block is brought in once by LDS-DMA (6 x 8 pixels x 128 B per wave;
//     pixels outside the image are out-of-range lanes = zeros) into one of two 48 KB buffers -- the next slab's patch is prefetched
//     during taps 1..6 -- and the nine taps read it at shifted addresses (slot swizzle `hslot` of conv3x3.hip: conflict-free for
//     every tap offset).  Group g owns patch rows 8g .. 8g+7.
//   B operand (weights [Cd][3][3][Cs], or the data gradient's [Cs][3][3][Cd] with the halo offsets mirrored): per k-half one PIECE
//     = BN rows x 64 B, through a ring of R pieces (BN 128: 6 x 8 KB, prefetch distance 4; BN 256: 3 x 16 KB, distance 2), rows
//     of 64 B = four 16-byte slots, slot s of row r stored at s ^ ((-(r >> 2)) & 3): the 16 lanes of every ds_read_b128 group hit
//     16 different slots of the 256-byte bank row.  LDS-DMA writes lane-linearly, so the swizzle is applied to the SOURCE slot.
//   Hazards (cdna_hip_programming.md "Read a staged buffer one phase AFTER the wait that retires it").  WAR: the fragment reads of
//     a load segment return while the wave waits at its barrier and are complete (the compiler's counted lgkmcnt waits in front of
//     the MFMAs) before the wave arrives at the barrier that ENDS its MFMA segment; group 1 reads in interval 2P+1, so a stage read
//     in phase P is refilled by LDS-DMAs issued in phase P+2 or later (interval >= 2P+4): ring distance R - 2 (one phase per k-half)
//     or R - 1 (two), the next halo patch from tap 1 on.  RAW:
//     piece j+1 (and the next halo patch) is waited for with a counted `s_waitcnt vmcnt(N)` in the LAST load segment of k-half j
//     by every wave; both groups pass a barrier after that before either reads it.  N is a compile-time function of the unrolled
//     position (static issue schedule: one weight DMA per phase, one halo DMA in the first phase of taps 1..6); DMAs past the end
//     of K are issued with all lanes out of range (they write zeros into dead stages) so that the counts never change.
//   TWO = two blocks per CU (BN = 128 only; launches of more than ~256 blocks): 80 KB of LDS and <= 128 VGPRs per block.  ONE halo
//     buffer (the next slab's patch is fetched at the slab boundary: both groups line up on a barrier, fetch, wait, re-stagger --
//     that bubble, like the prologue's cold fetch and the epilogue's store burst, is filled by the OTHER block's MFMAs), ring of
//     4 x 8 KB with distance 3, and here the load segment does end with `s_waitcnt lgkmcnt(0)` so that a stage may be refilled one
//     phase after it was read.  With one block per CU all blocks of a launch run in lock step: every CU fetches its first patch at
//     the same time (2-3 us of HBM burst), nobody computes during the store burst at the end, and a 400-block launch pays both
//     twice; measured a + b K fit on 200-block grids: a = 4-7 us, b = 0.37 us per k-half (~350 cycles per interval).
//   Epilogue: accumulators staged as fp32 [256][128 + 4] in LDS (two passes for BN = 256), then the epilogue of conv3x3.hip over
//     512 threads: bias, activation, residual, accumulate, BatchNorm partial sums (forward) and BatchNorm backward sums of the
//     block whose dz this data gradient completes (common.h: BnRedLane), 16-byte stores.
#include <stdlib.h>

#include "common.h"

namespace {

struct PGeom {
    int32_t N, H, W;
    int32_t Cs, Cd;
    int32_t flip;             // 1: data gradient (mirrored halo offsets)
    int32_t tiles_y, tiles_x, tiles_n;
    int32_t act, accumulate;
    int64_t sld, dld, rld;
    uint32_t src_bytes, w_bytes;
    int32_t nslab;
    int32_t tail;                // 1: exact wait counts where the younger DMAs are all-out-of-range padding (DSN_PP_TAIL=0: the counts of the steady state, A/B only)
    int32_t d2s_c, Hout, Wout;   // 2x2 form of the stride-2 data gradient: GEMM column c = parity (c / d2s_c) of channel c % d2s_c
};

constexpr int PT = 16;                       // patch edge (output pixels)
constexpr int PHW = PT + 2;                  // halo edge
constexpr int PNP = PHW * PHW;               // 324 halo pixels
constexpr int PIH = 6;                       // LDS-DMA instructions per wave and halo patch: 6 x 8 waves x 8 pixels = 384 slots
constexpr int HALO_BYTES = PIH * 64 * 128;   // 49152
constexpr int RING_BYTES = 49152;            // 6 x 8 KB (BN 128) = 3 x 16 KB (BN 256)
constexpr int PP_LDS = 2 * HALO_BYTES + RING_BYTES;     // 147456
constexpr int PP_LDS_TWO = HALO_BYTES + 4 * 8192;       // 81920: two blocks fill the 160 KB of a CU exactly
constexpr int PCPAD = 4;

__device__ __forceinline__ int pp_hslot(int slot, int hx) { return (((slot >> 1) ^ ((hx >> 1) & 3)) << 1) | (slot & 1); }

__device__ __forceinline__ int pp_xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// counted wait with a run-time-looking argument: every call site sits in a fully unrolled loop, the switch folds to one instruction
__device__ __forceinline__ void pp_wait_vm(int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// static issue schedule: does the first phase of k-half jj (position inside a slab, any integer: periodic with 18) carry a halo DMA?
__host__ __device__ constexpr bool pp_halo_at(int jj) {
    const int m = ((jj % 18) + 18) % 18;
    return (m & 1) == 0 && (m >> 1) >= 1 && (m >> 1) <= PIH;          // taps 1 .. 6 (tap 0 is too early: see the hazard note)
}
// vector-memory operations a wave may leave outstanding when, in the last load segment of k-half jj, it needs piece jj + 1 landed.
// MH phases per k-half, GW weight DMAs per piece (one per phase), prefetch distance D (piece j + D is issued during k-half j).
__host__ __device__ constexpr int pp_wait_count(int jj, int MH, int GW, int D) {
    int n = 0;
    if (MH == 1) n += pp_halo_at(jj + 1 - D) ? 1 : 0;          // the halo DMA issued behind piece j+1's weight DMA in the same phase
    for (int i = jj + 2 - D; i <= jj - 1; ++i) n += GW + (pp_halo_at(i) ? 1 : 0);
    if (MH == 2) n += 1 + (pp_halo_at(jj) ? 1 : 0);             // phase (jj, 0) of this k-half: its weight DMA (+ halo DMA)
    return n;
}

// The same by simulation, for both tap sets (NT = 9: 3x3; NT = 4: the 2x2 form of the stride-2 data gradient, whose slab has
// 8 k-halves and issues its 5 halo DMAs in k-halves 1..5, IN FRONT of that phase's weight DMA): walk the phases backwards from the
// wait, in reverse program order, and count the DMAs that are younger than the youngest one the wait needs -- a part of piece
// jj + 1, or, in the last k-half of a slab, a DMA of the next halo patch.
__host__ __device__ constexpr bool pp_halo_at_g(int jj, int NT) {
    const int KH = 2 * NT, m = ((jj % KH) + KH) % KH;
    return NT == 9 ? pp_halo_at(jj) : (m >= 1 && m <= 5);
}
__host__ __device__ constexpr int pp_wait_count_g(int jj, int MH, int GW, int D, int NT) {
    const int KH = 2 * NT;
    const bool need_halo = jj == KH - 1;
    int n = 0;
    for (int ph = jj * MH + MH - 2; ph >= (jj - D - 2) * MH; --ph) {
        const int i = ph >= 0 ? ph / MH : -((-ph + MH - 1) / MH), mh = ph - i * MH;
        const bool h = mh == 0 && pp_halo_at_g(i, NT), hn = h && need_halo && i >= 0;
        const bool wn = i + D == jj + 1;
        if (NT == 9) {          // program order: weight, halo
            if (h) { if (hn) return n; ++n; }
            if (wn) return n;
            ++n;
        } else {                // program order: halo, weight
            if (wn) return n;
            ++n;
            if (h) { if (hn) return n; ++n; }
        }
    }
    return n;
}
// The LAST slab of a block: the halo DMAs (next slab) and the weight DMAs of pieces past the end of K are all-out-of-range padding --
// they retire IMMEDIATELY (tools/exp/oob_order.hip), so they must not be counted as operations that may still be in flight: only
// the real pieces behind piece jj + 1 are.
__host__ __device__ constexpr int pp_wait_count_last(int jj, int MH, int GW, int D, int NT) {
    const int KH = 2 * NT;
    if (jj + 1 >= KH) return 0;
    int n = 0;
    for (int ph = jj * MH + MH - 2; ph >= (jj - D - 2) * MH; --ph) {
        const int i = ph >= 0 ? ph / MH : -((-ph + MH - 1) / MH);
        if (i + D == jj + 1) return n;
        if (i + D < KH) ++n;
    }
    return n;
}
// (run-time form for the 1x1 kernel, whose K is not unrolled by slab: real pieces behind piece j + 1, at most `cap`)
__device__ __forceinline__ void pp_wait_vm_rt(int n) {
    if (n <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else pp_wait_vm(n);
}
// Which halo DMAs (bit q = DMA q of the patch) pp_wait_count_g(jj, ...) counts as possibly in flight.  A wave whose halo DMA q has
// ALL lanes out of range (border patches: its eight pixels lie outside the image) must subtract it: that DMA retires at once.
__host__ __device__ constexpr unsigned pp_wait_halo_mask(int jj, int MH, int GW, int D, int NT) {
    const int KH = 2 * NT;
    const bool need_halo = jj == KH - 1;
    unsigned mask = 0u;
    for (int ph = jj * MH + MH - 2; ph >= (jj - D - 2) * MH; --ph) {
        const int i = ph >= 0 ? ph / MH : -((-ph + MH - 1) / MH), mh = ph - i * MH;
        const bool h = mh == 0 && pp_halo_at_g(i, NT), hn = h && need_halo && i >= 0;
        const bool wn = i + D == jj + 1;
        const int m = ((i % KH) + KH) % KH;
        const unsigned bit = 1u << (NT == 9 ? (m >> 1) - 1 : m - 1);
        if (NT == 9) {
            if (h) { if (hn) return mask; mask |= bit; }
            if (wn) return mask;
        } else {
            if (wn) return mask;
            if (h) { if (hn) return mask; mask |= bit; }
        }
    }
    return mask;
}
constexpr bool pp_wait_count_same(int MH, int GW, int D) {
    for (int jj = 0; jj < 18; ++jj)
        if (pp_wait_count_g(jj, MH, GW, D, 9) != pp_wait_count(jj, MH, GW, D)) return false;
    return true;
}
static_assert(pp_wait_count_same(1, 1, 4) && pp_wait_count_same(2, 2, 2), "the simulated wait counts reproduce the closed form of the 3x3 schedule");

// BatchNorm backward sums of the block(s) whose dz this data gradient completes -- BnRedLane of common.h with the per-channel
// constants (scale, shift, mean, rstd) in an LDS table [4][columns] instead of 32 registers per thread (the two-blocks-per-CU
// variant has 128 VGPRs for everything).  A thread owns one channel vector (8 channels) over NIT rows.
template <int NIT, bool PREF> struct PpRed {
    const bf16_t* yp;
    int64_t yld;
    float q0[8], q1[8];
    u32x4 yv[PREF ? NIT : 1];      // PREF: y at the NIT vectors this thread stores, fetched before the tile is staged
    int act;
    bool on;
    __device__ __forceinline__ void init(const BnRed& br, int ch) {
        on = false;
#pragma unroll
        for (int k = 0; k < 8; ++k) q0[k] = q1[k] = 0.f;
        for (int s = 0; s < br.nseg; ++s) {
            const dsn_bnred_seg& sg = br.seg[s];
            if (ch >= sg.c0 && ch < sg.c1) {
                on = true;
                yp = (const bf16_t*)sg.y + (ch - sg.c0);
                yld = sg.yld;
                act = sg.act;
            }
        }
    }
    // one table column: thread `c` of the first `cols` threads (channel ch); rows scale | shift | mean | rstd
    static __device__ __forceinline__ void fill(const BnRed& br, float* tab, int cols, int c, int ch) {
        float a = 0.f, b = 0.f, m = 0.f, r = 0.f;
        for (int s = 0; s < br.nseg; ++s) {
            const dsn_bnred_seg& sg = br.seg[s];
            if (ch >= sg.c0 && ch < sg.c1) {
                const int k = ch - sg.c0;
                a = sg.scale[k]; b = sg.shift[k]; m = sg.mean[k]; r = sg.rstd[k];
            }
        }
        tab[c] = a; tab[cols + c] = b; tab[2 * cols + c] = m; tab[3 * cols + c] = r;
    }
    __device__ __forceinline__ void prefetch(int it, int64_t row) {
        if (on && row >= 0) yv[PREF ? it : 0] = *reinterpret_cast<const u32x4*>(yp + row * yld);
    }
    // outv: the values just stored (rounded to bf16 -- what the apply pass will read as dz); tab + c0: this thread's columns
    __device__ __forceinline__ void add(int it, const bf16_t (&outv)[8], const float* tab, int cols, int c0) {
        if (!on) return;
        bf16_t yl[8];
        *reinterpret_cast<u32x4*>(yl) = yv[PREF ? it : 0];
        // four channels at a time: the constants are transient registers (the two-blocks-per-CU variant has none to spare)
#pragma unroll
        for (int k = 0; k < 8; k += 4) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(tab + c0 + k), sh = *reinterpret_cast<const f32x4*>(tab + cols + c0 + k);
            float u[4], gr[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) u[e] = (float)yl[k + e] * sc[e] + sh[e];
            act_grad_vec<4>(u, act, gr);
            const f32x4 mu = *reinterpret_cast<const f32x4*>(tab + 2 * cols + c0 + k), rs = *reinterpret_cast<const f32x4*>(tab + 3 * cols + c0 + k);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float y = (float)yl[k + e];
                const float gk = (float)outv[k + e] * gr[e];
                q0[k + e] += gk;
                q1[k + e] += gk * ((y - mu[e]) * rs[e]);
            }
        }
    }
};

template <int BN, bool TWO, int MH, bool LIN, bool D2S>
__device__ __forceinline__ void conv3x3_pp_epilogue_impl(f32x4 (&acc)[MH][4][4], unsigned char* smem, const float* __restrict__ bias,
                                                        const bf16_t* __restrict__ res, bf16_t* __restrict__ dst, const BnAcc& fin,
                                                        const PGeom& g, const BnRed& br, int n, int y0, int x0, int n0, int tmi);
#ifdef DSN_PP_STAMP
__device__ unsigned long long pp_stamp_buf[6 * 4096];
#endif

// DIR variants: MFMA operands swapped (A = weights, B = pixels), so that a lane's accumulators are 4 consecutive channels of ONE pixel;
// the weight rows of each 32-channel group are permuted in LDS (by the DMA's source addressing) so that the accumulators of the
// MFMA pair (2q, 2q + 1) are 8 consecutive channels: physical row r of a piece holds channel pp_perm_row(r) of the tile.
__device__ __forceinline__ int pp_perm_row(int r) { return (r & ~31) + (((r & 15) >> 2) << 3) + (((r >> 4) & 1) << 2) + (r & 3); }

template <int BN, bool TWO, int MH, bool LIN, bool D2S>
__device__ __forceinline__ void pp_epilogue_direct(f32x4 (&acc)[MH][4][4], unsigned char* smem, const float* __restrict__ bias,
                                                  const bf16_t* __restrict__ res, bf16_t* __restrict__ dst, const BnAcc& fin,
                                                  const PGeom& g, const BnRed& br, int n, int y0, int x0, int n0, int tmi);

template <int BN, bool TWO, int MH, bool LIN = false, bool D2S = false>
__device__ __forceinline__ void conv3x3_pp_epilogue(f32x4 (&acc)[MH][4][4], unsigned char* smem, const float* __restrict__ bias,
                                                   const bf16_t* __restrict__ res, bf16_t* __restrict__ dst, const BnAcc& fin,
                                                   const PGeom& g, const BnRed& br, int n, int y0, int x0, int n0, int tmi) {
    conv3x3_pp_epilogue_impl<BN, TWO, MH, LIN, D2S>(acc, smem, bias, res, dst, fin, g, br, n, y0, x0, n0, tmi);
}

// NT = 4: the 2x2 / stride-1 form of the 3x3 / stride-2 data gradient (igemm.hip, dsn_conv2d_dgrad_s2): taps (ty, tx) in {0, 1}^2
// read dy at (y + ty, x + tx) -- a 17 x 17 halo with the patch in its top-left corner --, weights [4 Ci][2][2][Co], 8 k-halves per
// slab, depth-to-space store.  Ring: 8 x 8 KB, distance 6 (BN 128) / 4 x 16 KB, distance 3 (BN 256); halo buffers 2 x 40 KB.
template <int BN, bool FLIP, bool TWO, int NT = 9, bool DIR = false>
__global__ __launch_bounds__(512, TWO ? 4 : 2) void conv3x3_pp_kernel(const bf16_t* __restrict__ src, const bf16_t* __restrict__ wpk,
                                                         const float* __restrict__ bias, const bf16_t* __restrict__ res,
                                                         bf16_t* __restrict__ dst, const BnAcc fin, const PGeom g, const BnRed br) {
    static_assert(BN == 128 || BN == 256, "block tiles of 128 or 256 output channels");
    static_assert(!TWO || BN == 128, "two blocks per CU: 128-channel tiles");
    static_assert(NT == 9 || (NT == 4 && !FLIP && !TWO), "tap sets: 3x3, or the 2x2 form of the stride-2 data gradient");
    constexpr int KH = 2 * NT;                   // k-halves per slab
    constexpr int PADO = NT == 9 ? 1 : 0;        // halo rows / columns in front of the patch
    constexpr int PIHN = NT == 9 ? PIH : 5;      // halo DMAs per wave and patch (NT 4: 17 rows x 18 slots = 306 <= 320)
    constexpr int PNPN = NT == 9 ? PNP : 17 * PHW;
    constexpr int HB = PIHN * 64 * 128;          // bytes of a halo buffer
    typedef bf16_t T;
    constexpr int MH = BN / 128;                 // 64-pixel halves of a wave's pixel rows = phases per k-half
    constexpr int GW = BN / 128;                 // weight DMAs per wave and piece
    constexpr int R = NT == 4 ? (BN == 128 ? 8 : 4) : TWO ? 4 : (BN == 128 ? 6 : 3);   // ring stages (KH % R == 0)
    // prefetch distance in k-halves.  A stage is refilled no earlier than TWO phases after the phase that read it (the reads of a
    // load segment are only known complete once their wave has run its MFMA segment, i.e. after the NEXT barrier pair): with one
    // phase per k-half that is distance R - 2, with two phases per k-half R - 1.
    // (TWO: lgkmcnt(0) closes the load segment, so the distance is R - 1 there as well)
    constexpr int D = (MH == 1 && !TWO) ? R - 2 : R - 1;
    constexpr int PIECE = BN * 64;
    constexpr int WGN = BN == 128 ? 2 : 4;       // waves of a group across the output channels (64 each)
    constexpr int VEC = 8;
    static_assert(R * PIECE == (NT == 4 ? 65536 : TWO ? 4 * 8192 : RING_BYTES) && (TWO || KH % R == 0), "ring size");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sH = smem;
    unsigned char* sB = smem + (TWO ? 1 : 2) * HB;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = wave >> 2, wq = wave & 3;
    const int wm = wq / WGN, wn = wq % WGN;      // (BN 256: wm == 0)
    const int fr = lane & 15, fg = lane >> 4;
    const int tiles_img = g.tiles_y * g.tiles_x;
    const int tile = pp_xcd_remap(blockIdx.x, g.N * tiles_img * g.tiles_n);
    const int tn = tile % g.tiles_n, tmi = tile / g.tiles_n;
    const int n = tmi / tiles_img, trem = tmi - n * tiles_img;
    const int y0 = (trem / g.tiles_x) * PT, x0 = (trem % g.tiles_x) * PT;
    const int n0 = tn * BN;
#ifdef DSN_PP_STAMP
    const unsigned long long st_re = __builtin_amdgcn_s_memrealtime();
#endif

    const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, g.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wpk, 0, g.w_bytes, 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr;
    constexpr uint32_t OOB = 0xFFFFFFF0u;

    // ---- halo fetch plan: DMA q of this wave covers halo pixel slots 64 q + 8 wave .. + 8; lane -> pixel, physical slot lane & 7
    // (tq: threadIdx.x, passed in so that the TWO variant can hand over an OPAQUE copy at each slab boundary -- otherwise the six
    //  loop-invariant offsets are hoisted out of the slab loop and live in registers the 128-VGPR budget does not have)
    auto halo_off = [&](int q, int tq) -> uint32_t {
        const int lane = tq & 63, wave = tq >> 6;
        const int p = 64 * q + 8 * wave + (lane >> 3);
        const int hy = p / PHW, hx = p - hy * PHW;
        const int gy = y0 - PADO + hy, gx = x0 - PADO + hx;
        const bool ok = p < PNPN && (unsigned)gy < (unsigned)g.H && (unsigned)gx < (unsigned)g.W;
        const int ls = pp_hslot(lane & 7, hx);                       // logical slot fetched (pp_hslot is an involution in the slot)
        return ok ? (uint32_t)((((int64_t)n * g.H + gy) * g.W + gx) * g.sld + ls * VEC) * 2u : OOB;
    };
    uint32_t hoff[TWO ? 1 : PIHN];              // (TWO: recomputed at each slab boundary -- six registers fewer in the loop)
    if constexpr (!TWO) {
#pragma unroll
        for (int q = 0; q < PIHN; ++q) hoff[q] = halo_off(q, tid);
    }
    auto load_halo1 = [&](int slab, int hbuf, int q, int tq = 0) {
        const uint32_t add = slab < g.nslab ? (uint32_t)slab * 128u : OOB;
        const uint32_t ho = TWO ? halo_off(q, tq) : hoff[TWO ? 0 : q];
        const uint32_t off = (ho == OOB || add == OOB) ? OOB : ho + add;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srsrc, (lds_ptr)(sH + hbuf * HB + (64 * q + 8 * wave) * 128), 16, off, 0, 0, DSN_DMA_AUX);
    };
    // ---- weight fetch plan: DMA `part` of this wave covers rows 16 (wave + 8 part) .. + 16 of the piece; lane -> row, physical slot
    uint32_t woff[GW];
#pragma unroll
    for (int part = 0; part < GW; ++part) {
        const int prow = 16 * (wave + 8 * part) + (lane >> 2);         // physical row of the piece
        const int row = DIR ? pp_perm_row(prow) : prow;               // channel of the tile it holds
        const int ls = (lane & 3) ^ ((-(lane >> 4)) & 3);             // ((prow >> 2) & 3) == lane >> 4: logical slot of this lane
        woff[part] = n0 + row < g.Cd ? (uint32_t)((int64_t)(n0 + row) * NT * g.Cs + ls * VEC) * 2u : OOB;
    }
    // Per-wave facts about DMA instructions whose lanes are ALL out of range (they retire at once, tools/exp/oob_order.hip, so
    // the wait counts must not include them): hoob = such halo DMAs (border patches), wall = every weight DMA of this wave is real
    // (false: rows past Cd in a partial channel tile -- that wave simply drains).
    unsigned hoob = 0u;
    if constexpr (!TWO) {
#pragma unroll
        for (int q = 0; q < PIHN; ++q) hoob |= (__ballot(hoff[q] != OOB) == 0ull) ? 1u << q : 0u;
    }
    bool wall = true;
#pragma unroll
    for (int part = 0; part < GW; ++part) wall = wall && __ballot(woff[part] != OOB) != 0ull;
    const int T_ALL = g.nslab * KH;                                   // k-halves
    // piece jp (global k-half index): slab jp / KH, weight tap (jp % KH) / 2, channel half jp & 1
    auto load_w1 = [&](int jp, int slot, int part) {
        const int s = jp / KH, jj = jp - s * KH, t = jj >> 1, kk = jj & 1;
        const uint32_t add = jp < T_ALL ? (uint32_t)(t * g.Cs + s * 64 + kk * 32) * 2u : OOB;
        const uint32_t off = (woff[part] == OOB || add == OOB) ? OOB : woff[part] + add;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_ptr)(sB + slot * PIECE + (wave + 8 * part) * 1024), 16, off, 0, 0, DSN_DMA_AUX);
    };

    // ---- fragment addressing ------------------------------------------------------------------------------------------------
    // A: fragment i of m-half mh = patch row 8 grp + 4 (MH == 2 ? mh : wm) + i, pixel column fr; halo pixel (row + oy, fr + ox)
    int a_base[3][2];
    {
        const int yb = 8 * grp + (MH == 2 ? 0 : 4 * wm);
#pragma unroll
        for (int ox = 0; ox < 3; ++ox)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) a_base[ox][kk] = (yb * PHW + fr) * 128 + (pp_hslot(4 * kk + fg, fr + ox) << 4);   // (+ ox pixels in the read offset)
    }
    const int b_base = (wn * 64 + fr) * 64 + ((fg ^ ((-(fr >> 2)) & 3)) << 4);

    f32x4 acc[MH][4][4];
#pragma unroll
    for (int m = 0; m < MH; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[m][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: halo(0), pieces 0 .. D-1; halo(0) and piece 0 landed before the first barrier ------------------------------------
#pragma unroll
    for (int q = 0; q < PIHN; ++q) load_halo1(0, 0, q, tid);
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int part = 0; part < GW; ++part) load_w1(i, i % R, part);
    if (!wall) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // (dead weight DMAs must not stand for the halo DMAs in front of them)
    else pp_wait_vm((D - 1) * GW);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (grp == 1) {                                 // the stagger: group 1 runs one barrier interval behind group 0
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }

#ifdef DSN_PP_STAMP      // diagnostic build only (tools/exp/pp_clock.sh): shader clock held inside the main loop (MI355X_MICROARCH.md, DVFS item 6)
    const unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    u32x4 fa[4], fb[4];
    for (int s = 0; s < g.nslab; ++s) {
        const unsigned char* hb = sH + (TWO ? 0 : (s & 1) * HB);
        const bool last = s == g.nslab - 1 && g.tail;   // (block-uniform) the padding DMAs of this slab retire at once: exact counts
        const int so = TWO ? (s & 1) * 2 : 0;        // TWO: 18 k-halves per slab over 4 stages: the ring position shifts by 2 per slab
#pragma unroll
        for (int jj = 0; jj < KH; ++jj) {
            const int t = jj >> 1, kk = jj & 1;
            // halo offset of weight tap t (data gradient: mirrored -- the same tap order, hence the same fp32 summation order, as
            // the kernels of conv3x3.hip / conv_ws.hip / igemm.hip)
            const int oy = NT == 4 ? t >> 1 : FLIP ? 2 - t / 3 : t / 3, ox = NT == 4 ? t & 1 : FLIP ? 2 - t % 3 : t % 3;
            const int slot = TWO ? ((jj + so) & 3) : jj % R;       // (!TWO: 18 % R == 0, a compile-time function of jj)
            const int slot_w = TWO ? ((jj + D + so) & 3) : (jj + D) % R;
#pragma unroll
            for (int mh = 0; mh < MH; ++mh) {
                // ---- load segment ----------------------------------------------------------------------------------------------
                if (mh == 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        fb[j] = *reinterpret_cast<const u32x4*>(sB + slot * PIECE + b_base + j * 1024);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    fa[i] = *reinterpret_cast<const u32x4*>(hb + a_base[ox][kk] + ((4 * mh + i + oy) * PHW + ox) * 128);
                // piece j+1 (and, at jj == 17, the next halo patch) landed
                if (mh == MH - 1) {
                    if (last) pp_wait_vm(TWO ? (jj + 1 >= KH ? 0 : ((D - 2 < KH - 2 - jj ? D - 2 : KH - 2 - jj) * GW)) : pp_wait_count_last(jj, MH, GW, D, NT));
                    else if (TWO) pp_wait_vm((D - 2) * GW);
                    else if (hoob) {                                        // (wave-uniform; border patches only)
                        const unsigned HM = pp_wait_halo_mask(jj, MH, GW, D, NT);    // (folds: jj is an unrolled index)
                        const int CNT = pp_wait_count_g(jj, MH, GW, D, NT), k = (int)__popc(hoob & HM);
                        if (k == 0) pp_wait_vm(CNT);                       // (short compare chain: k is mostly 0 .. 2)
                        else if (k == 1) pp_wait_vm(CNT - 1);
                        else if (k == 2) pp_wait_vm(CNT - 2);
                        else pp_wait_vm_rt(CNT - k);
                    } else pp_wait_vm(pp_wait_count_g(jj, MH, GW, D, NT));
                    if (!wall) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                if (NT == 4 && mh == 0 && pp_halo_at_g(jj, NT)) load_halo1(s + 1, (s + 1) & 1, jj - 1);
                load_w1(s * KH + jj + D, slot_w, GW == 1 ? 0 : mh);
                if (NT == 9 && !TWO && mh == 0 && pp_halo_at(jj)) load_halo1(s + 1, (s + 1) & 1, t - 1);
                // !TWO: no lgkmcnt wait here -- the fragment reads return while the wave waits at the barrier; the compiler's own
                // counted waits in front of the MFMAs order the registers
                if (TWO) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // ---- MFMA segment ----------------------------------------------------------------------------------------------
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[mh][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, DIR ? fb[j] : fa[i]),
                                                                               __builtin_bit_cast(bf16x8, DIR ? fa[i] : fb[j]), acc[mh][i][j], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (TWO && s + 1 < g.nslab) {
            // slab boundary with ONE halo buffer: line the groups up (every wave has run the MFMA segment behind its last fragment
            // reads), fetch the next patch, publish it, re-stagger.  The other block of this CU computes meanwhile.
            if (grp == 0) {
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
            int tq = threadIdx.x;
            asm volatile("" : "+v"(tq));
#pragma unroll
            for (int q = 0; q < PIH; ++q) {
                load_halo1(s + 1, 0, q, tq);
                __builtin_amdgcn_sched_barrier(0);           // (one offset computation at a time: no six-deep register set)
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (grp == 1) {
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if (grp == 0) {                                 // group 0 joins the barrier group 1 is one interval late for
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
#ifdef DSN_PP_STAMP
    if (tid == 0 && blockIdx.x < 4096) {        // (a buffer of its own: no output value depends on the stamps)
        pp_stamp_buf[6 * blockIdx.x] = __builtin_amdgcn_s_memtime() - st_t0;
        pp_stamp_buf[6 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - st_r0;
        pp_stamp_buf[6 * blockIdx.x + 2] = st_re;
        pp_stamp_buf[6 * blockIdx.x + 3] = st_r0 - st_re;
    }
    const unsigned long long st_r1 = __builtin_amdgcn_s_memrealtime();
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the out-of-range DMAs past the end of K: zeros into dead stages)
    __syncthreads();
    // (opaque moves: they end the accumulators' main-loop live ranges here, so that the register pressure of the epilogue cannot
    //  make the allocator park accumulators in scratch across the slab loop's back edge -- it did: 18 scratch operations per slab)
#pragma unroll
    for (int m = 0; m < MH; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(acc[m][i][j]));
    if constexpr (DIR) pp_epilogue_direct<BN, TWO, MH, false, NT == 4>(acc, smem, bias, res, dst, fin, g, br, n, y0, x0, n0, tmi);
    else conv3x3_pp_epilogue<BN, TWO, MH, false, NT == 4>(acc, smem, bias, res, dst, fin, g, br, n, y0, x0, n0, tmi);
#ifdef DSN_PP_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0 && blockIdx.x < 4096) pp_stamp_buf[6 * blockIdx.x + 4] = __builtin_amdgcn_s_memrealtime() - st_r1;
#endif
}

// The epilogue re-derives its lane coordinates from threadIdx.x behind an opaque move: nothing of them stays live across the main
// loop (TWO runs at 128 VGPRs: 64 accumulators + 32 fragment registers leave 32 for everything else).
// LIN (the 1x1 kernel below): tile row r is pixel 256 tmi + r of the [N H W] pixel list instead of patch pixel (r >> 4, r & 15).
// D2S (2x2 form of the stride-2 data gradient): GEMM column c is channel c % d2s_c of destination pixel (2y + py, 2x + px),
// (py, px) = parity c / d2s_c, of a destination Hout x Wout.
template <int BN, bool TWO, int MH, bool LIN, bool D2S>
__device__ __forceinline__ void conv3x3_pp_epilogue_impl(f32x4 (&acc)[MH][4][4], unsigned char* smem, const float* __restrict__ bias,
                                                        const bf16_t* __restrict__ res, bf16_t* __restrict__ dst, const BnAcc& fin,
                                                        const PGeom& g, const BnRed& br, int n, int y0, int x0, int n0, int tmi) {
    typedef bf16_t T;
    constexpr int VEC = 8;
    constexpr int WGN = BN == 128 ? 2 : 4;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6;
    const int grp = wave >> 2, wq = wave & 3;
    const int wm = wq / WGN, wn = wq % WGN;
    const int fr = lane & 15, fg = lane >> 4;
    // ---- epilogue: per 128-channel pass, act(acc + bias) staged as fp32 [256][128 + 4] (row = patch pixel y * 16 + x) -------------
    // Passes: the one-block 128-channel tile stages all its accumulators at once (135 KB).  The others (80 KB of LDS / 256 channels)
    // run TWO passes, and in pass p EVERY wave stages the channel blocks j = 2p, 2p+1 of its 64 channels -- half of its accumulators,
    // so that the store work of pass 0 (BatchNorm-backward operands: ~80 registers) runs beside 32 / 64 live accumulator registers
    // instead of 64 / 128; no scratch (a kernel that uses scratch pays for it at every dispatch: +15 us measured here).  Staged column
    // cl of pass p is channel n0 + (cl >> 5) * 64 + 32 p + (cl & 31): 32-channel segments, a 16-byte vector never straddles one.
    constexpr int NPASS = (BN == 128 && !TWO) ? 1 : 2;
    constexpr int EPW = BN / NPASS;                 // staged columns per pass
    constexpr int PLDC = EPW + PCPAD;
    constexpr int VPR = EPW / VEC;                  // channel vectors per staged row
    constexpr int NIT = (256 * VPR) / 512;          // vectors per thread
    const int vh = (g.H - y0 < PT) ? g.H - y0 : PT, vw = (g.W - x0 < PT) ? g.W - x0 : PT;      // valid part of the patch
    const int64_t m0 = (int64_t)tmi * 256;
    const int vrows = LIN ? (int)((int64_t)g.N * g.H * g.W - m0 < 256 ? (int64_t)g.N * g.H * g.W - m0 : 256) : 256;
    auto valid = [&](int rl) { return LIN ? rl < vrows : ((rl >> 4) < vh && (rl & 15) < vw); };
    auto grow = [&](int rl) -> int64_t { return LIN ? m0 + rl : ((int64_t)n * g.H + y0 + (rl >> 4)) * g.W + x0 + (rl & 15); };
    // D2S: destination channel / pixel row of GEMM column `col` at tile row rl (-1: outside an odd-sized destination)
    auto dch = [&](int col) { return D2S ? col % g.d2s_c : col; };
    auto drow = [&](int rl, int col) -> int64_t {
        if (!D2S) return grow(rl);
        const int par = col / g.d2s_c, yy = 2 * (y0 + (rl >> 4)) + (par >> 1), xx = 2 * (x0 + (rl & 15)) + (par & 1);
        return (yy < g.Hout && xx < g.Wout) ? ((int64_t)n * g.Hout + yy) * g.Wout + xx : -1;
    };
    float* sC = reinterpret_cast<float*>(smem);
    float* red = sC + 256 * PLDC;                   // scratch for the per-channel folds (<= 8 KB)
    float* tab = red + 2048;                        // [4][EPW] BatchNorm constants of this pass's columns (<= 2 KB)
    auto chan = [&](int pass, int cl) { return NPASS == 1 ? n0 + cl : n0 + (cl >> 5) * 64 + 32 * pass + (cl & 31); };
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
        // (TWO, and the 1x1 kernel's 256-channel tile -- 128 live accumulators: y is loaded inside the store loop, no register set
        //  held across the staging)
        constexpr bool PREF = !TWO && !(LIN && MH == 2);
        PpRed<NIT, PREF> bl;
        if (br.nseg) {
            bl.init(br, dch(chan(pass, (tid % VPR) * VEC)));
            if (tid < EPW) PpRed<NIT, PREF>::fill(br, tab, EPW, tid, dch(chan(pass, tid)));
#pragma unroll
            for (int it = 0; it < (PREF ? NIT : 0); ++it) {
                const int idx = tid + it * 512, rl = idx / VPR;
                const int pc = chan(pass, (idx - rl * VPR) * VEC);
                const bool ok = valid(rl) && pc < g.Cd;
                bl.prefetch(it, ok ? drow(rl, pc) : -1);
            }
        }
        // BatchNorm partial sums (forward): taken from the accumulator registers while they are staged -- a lane owns one column of
        // 16 MH rows per channel block, the four row groups of a wave fold by two shuffles, the waves of a column through LDS
        float cs0[4 / NPASS], cs1[4 / NPASS];
#pragma unroll
        for (int jl = 0; jl < 4 / NPASS; ++jl) cs0[jl] = cs1[jl] = 0.f;
        const bool full = LIN ? vrows == 256 : (vh == PT && vw == PT);
#pragma unroll
        for (int jl = 0; jl < 4 / NPASS; ++jl) {
            const int j = NPASS == 1 ? jl : 2 * pass + jl;
            const int cl = NPASS == 1 ? wn * 64 + j * 16 + fr : wn * 32 + jl * 16 + fr;
            const int col = chan(pass, cl);
            const float bv = (bias && col < g.Cd) ? bias[col] : 0.f;
#pragma unroll
            for (int m = 0; m < MH; ++m)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v4[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v4[e] = acc[m][i][j][e] + bv;
                    apply_act_vec<4>(v4, g.act);
                    const int py = 8 * grp + 4 * (MH == 2 ? m : wm) + i;
                    if (fin.acc) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float t = (full || valid(py * 16 + fg * 4 + e)) ? v4[e] : 0.f;
                            cs0[jl] += t;
                            cs1[jl] += t * t;
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) sC[(py * 16 + fg * 4 + e) * PLDC + cl] = v4[e];
                }
        }
        if (fin.acc) {
#pragma unroll
            for (int jl = 0; jl < 4 / NPASS; ++jl) {
                cs0[jl] += __shfl_xor(cs0[jl], 16); cs1[jl] += __shfl_xor(cs1[jl], 16);
                cs0[jl] += __shfl_xor(cs0[jl], 32); cs1[jl] += __shfl_xor(cs1[jl], 32);
                if (fg == 0) {
                    red[(wave * 64 + jl * 16 + fr) * 2] = cs0[jl];
                    red[(wave * 64 + jl * 16 + fr) * 2 + 1] = cs1[jl];
                }
            }
        }
        __syncthreads();
        if (fin.acc) {
            if (tid < EPW && chan(pass, tid) < g.Cd) {
                const int wn_ = NPASS == 1 ? tid >> 6 : tid >> 5, ci = NPASS == 1 ? tid & 63 : tid & 31;
                float t0 = 0.f, t1 = 0.f;
#pragma unroll
                for (int gq = 0; gq < 2; ++gq)
#pragma unroll
                    for (int wm_ = 0; wm_ < 4 / WGN; ++wm_) {
                        const int w = gq * 4 + wm_ * WGN + wn_;
                        t0 += red[(w * 64 + ci) * 2];
                        t1 += red[(w * 64 + ci) * 2 + 1];
                    }
                bn_acc_add(fin, tmi, chan(pass, tid), t0, t1);
            }
        }
#pragma unroll(TWO ? 1 : NIT)
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * 512;
            const int rl = idx / VPR, cv = idx - rl * VPR;
            const int col = chan(pass, cv * VEC);
            if (!valid(rl) || col >= g.Cd) continue;
            const int64_t row = drow(rl, col);
            if (D2S && row < 0) continue;
            if (!PREF && br.nseg) bl.prefetch(it, row);
            float vals[VEC];
#pragma unroll
            for (int e = 0; e < VEC; e += 4) {
                const f32x4 t4 = *reinterpret_cast<const f32x4*>(sC + rl * PLDC + cv * VEC + e);
                vals[e] = t4[0]; vals[e + 1] = t4[1]; vals[e + 2] = t4[2]; vals[e + 3] = t4[3];
            }
            T* o = dst + row * g.dld + dch(col);
            if (res) {
                T rv[VEC];
                *reinterpret_cast<u32x4*>(rv) = *reinterpret_cast<const u32x4*>(res + row * g.rld + dch(col));
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
            if (br.nseg) bl.add(it, outv, tab, EPW, cv * VEC);
        }
        if (br.nseg) {
            // fold of BnRedLane::finish over EIGHT waves: lanes of equal channel vector by shuffles, the waves through LDS
            __syncthreads();
#pragma unroll
            for (int o = VPR; o < 64; o <<= 1) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    bl.q0[k] += __shfl_xor(bl.q0[k], o);
                    bl.q1[k] += __shfl_xor(bl.q1[k], o);
                }
            }
            if (lane < VPR) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    red[(wave * EPW + lane * VEC + k) * 2] = bl.on ? bl.q0[k] : 0.f;
                    red[(wave * EPW + lane * VEC + k) * 2 + 1] = bl.on ? bl.q1[k] : 0.f;
                }
            }
            __syncthreads();
            if (tid < EPW && chan(pass, tid) < g.Cd) {
                float s0 = 0.f, s1 = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) {
                    s0 += red[(w * EPW + tid) * 2];
                    s1 += red[(w * EPW + tid) * 2 + 1];
                }
                const int ch = dch(chan(pass, tid));
                for (int sg = 0; sg < br.nseg; ++sg) {
                    const dsn_bnred_seg& q = br.seg[sg];
                    if (ch >= q.c0 && ch < q.c1) bn_acc_add(BnAcc{(double*)q.acc, q.acc_c, 0.0}, tmi, q.ch0 + ch - q.c0, s0, s1);
                }
            }
        }
        __syncthreads();
    }
}


// ---- epilogue straight from the accumulator registers (DIR) -----------------------------------------------------------------------------
// Lane (fr, fg) of wave (grp, wm, wn) owns pixel row 16 py + fr of the tile for py = 8 grp + 4 (MH == 2 ? mh : wm) + i and the two channel
// vectors wn 64 + q 32 + fg 8 .. + 8 (q = 0, 1): no LDS staging tile, no second pass; 16-byte stores, 64 contiguous bytes per pixel
// and wave instruction.  The operands the epilogue reads (shortcut, old value, y of the BatchNorm-backward sums) are fetched one
// (mh, i) step ahead.  BatchNorm sums (forward) / BatchNorm-backward sums stay in 32 registers per lane and are folded over the 16
// pixel lanes by shuffles, over the waves of a column through LDS ([8][64][2] floats at the start of the dead LDS; the per-channel
// constants of the backward sums in a table behind it).
template <int BN, bool TWO, int MH, bool LIN, bool D2S>
__device__ __forceinline__ void pp_epilogue_direct(f32x4 (&acc)[MH][4][4], unsigned char* smem, const float* __restrict__ bias,
                                                  const bf16_t* __restrict__ res, bf16_t* __restrict__ dst, const BnAcc& fin,
                                                  const PGeom& g, const BnRed& br, int n, int y0, int x0, int n0, int tmi) {
    typedef bf16_t T;
    constexpr int WGN = BN == 128 ? 2 : 4;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6;
    const int grp = wave >> 2, wq = wave & 3;
    const int wm = wq / WGN, wn = wq % WGN;
    const int fr = lane & 15, fg = lane >> 4;
    float* red = reinterpret_cast<float*>(smem);          // [8 waves][64 columns][2]
    float* tab = red + 1024;                              // [4][BN]: scale | shift | mean | rstd of the block's columns
    const int vh = (g.H - y0 < PT) ? g.H - y0 : PT, vw = (g.W - x0 < PT) ? g.W - x0 : PT;
    const int64_t m0 = (int64_t)tmi * 256;
    const int vrows = LIN ? (int)((int64_t)g.N * g.H * g.W - m0 < 256 ? (int64_t)g.N * g.H * g.W - m0 : 256) : 256;
    auto valid = [&](int rl) { return LIN ? rl < vrows : ((rl >> 4) < vh && (rl & 15) < vw); };
    int col[2], cch[2], par[2];
    bool okc[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        col[q] = n0 + wn * 64 + q * 32 + fg * 8;
        okc[q] = col[q] < g.Cd;
        par[q] = D2S ? col[q] / g.d2s_c : 0;
        cch[q] = D2S ? col[q] - par[q] * g.d2s_c : col[q];
    }
    // destination pixel row of tile row rl for vector q (-1: nothing stored)
    auto rowof = [&](int rl, int q) -> int64_t {
        if (!valid(rl) || !okc[q]) return -1;
        if (LIN) return m0 + rl;
        if (!D2S) return ((int64_t)n * g.H + y0 + (rl >> 4)) * g.W + x0 + (rl & 15);
        const int yy = 2 * (y0 + (rl >> 4)) + (par[q] >> 1), xx = 2 * (x0 + (rl & 15)) + (par[q] & 1);
        return (yy < g.Hout && xx < g.Wout) ? ((int64_t)n * g.Hout + yy) * g.Wout + xx : -1;
    };
    // BatchNorm-backward sums: the segment of each of the lane's vectors, the block's table of constants
    const T* yp[2] = {nullptr, nullptr};
    int64_t yld[2] = {0, 0};
    int yact[2] = {0, 0};
    if (br.nseg) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
            for (int sg = 0; sg < br.nseg; ++sg) {
                const dsn_bnred_seg& z = br.seg[sg];
                if (okc[q] && cch[q] >= z.c0 && cch[q] < z.c1) {
                    yp[q] = (const T*)z.y + (cch[q] - z.c0);
                    yld[q] = z.yld;
                    yact[q] = z.act;
                }
            }
        if (tid < BN) {
            const int c = n0 + tid;
            PpRed<1, false>::fill(br, tab, BN, tid, D2S ? c % g.d2s_c : c);
        }
        __syncthreads();
    }
    float bq[2][8];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int k = 0; k < 8; ++k) bq[q][k] = (bias && okc[q]) ? bias[col[q] + k] : 0.f;
    float s0[2][8], s1[2][8];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int k = 0; k < 8; ++k) s0[q][k] = s1[q][k] = 0.f;

    u32x4 rv[2][2], ov[2][2], yv[2][2];          // [stage][q]: shortcut, old value, y -- fetched one step ahead
    auto tile_row = [&](int mi) { return (8 * grp + 4 * (MH == 2 ? (mi >> 2) : wm) + (mi & 3)) * 16 + fr; };
    auto fetch = [&](int mi, int st) {
        const int rl = tile_row(mi);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int64_t row = rowof(rl, q);
            if (row < 0) continue;
            if (res) rv[st][q] = *reinterpret_cast<const u32x4*>(res + row * g.rld + cch[q]);
            if (g.accumulate) ov[st][q] = *reinterpret_cast<const u32x4*>(dst + row * g.dld + cch[q]);
            if (yp[q]) yv[st][q] = *reinterpret_cast<const u32x4*>(yp[q] + row * yld[q]);
        }
    };
    fetch(0, 0);
#pragma unroll
    for (int mi = 0; mi < MH * 4; ++mi) {
        const int st = mi & 1, mh = mi >> 2, i = mi & 3;
        if (mi + 1 < MH * 4) fetch(mi + 1, st ^ 1);
        const int rl = tile_row(mi);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int64_t row = rowof(rl, q);
            float vals[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) vals[k] = acc[mh][i][2 * q + (k >> 2)][k & 3] + bq[q][k];
            apply_act_vec<8>(vals, g.act);
            if (fin.acc) {          // (forward: no shortcut / accumulate / activation in front of the statistics)
                const bool in = okc[q] && valid(rl);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float t = in ? vals[k] : 0.f;
                    s0[q][k] += t;
                    s1[q][k] += t * t;
                }
            }
            if (row < 0) continue;
            if (res) {
                T r8[8];
                *reinterpret_cast<u32x4*>(r8) = rv[st][q];
#pragma unroll
                for (int k = 0; k < 8; ++k) vals[k] += to_f32<T>(r8[k]);
            }
            if (g.accumulate) {
                T o8[8];
                *reinterpret_cast<u32x4*>(o8) = ov[st][q];
#pragma unroll
                for (int k = 0; k < 8; ++k) vals[k] += to_f32<T>(o8[k]);
            }
            T outv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) outv[k] = from_f32<T>(vals[k]);
            *reinterpret_cast<u32x4*>(dst + row * g.dld + cch[q]) = *reinterpret_cast<u32x4*>(outv);
            if (yp[q]) {
                T yl[8];
                *reinterpret_cast<u32x4*>(yl) = yv[st][q];
                const float* tc = tab + wn * 64 + q * 32 + fg * 8;
#pragma unroll
                for (int k = 0; k < 8; k += 4) {
                    const f32x4 sc = *reinterpret_cast<const f32x4*>(tc + k), sh = *reinterpret_cast<const f32x4*>(tc + BN + k);
                    float u[4], gr[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) u[e] = (float)yl[k + e] * sc[e] + sh[e];
                    act_grad_vec<4>(u, yact[q], gr);
                    const f32x4 mu = *reinterpret_cast<const f32x4*>(tc + 2 * BN + k), rs = *reinterpret_cast<const f32x4*>(tc + 3 * BN + k);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float y = (float)yl[k + e];
                        const float gk = (float)outv[k + e] * gr[e];
                        s0[q][k + e] += gk;
                        s1[q][k + e] += gk * ((y - mu[e]) * rs[e]);
                    }
                }
            }
        }
    }
    if (fin.acc || br.nseg) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
                    s0[q][k] += __shfl_xor(s0[q][k], o);
                    s1[q][k] += __shfl_xor(s1[q][k], o);
                }
                if (fr == 0) {
                    red[(wave * 64 + q * 32 + fg * 8 + k) * 2] = s0[q][k];
                    red[(wave * 64 + q * 32 + fg * 8 + k) * 2 + 1] = s1[q][k];
                }
            }
        __syncthreads();
        if (tid < BN && n0 + tid < g.Cd) {
            const int wn_ = tid >> 6, ci = tid & 63;
            float t0 = 0.f, t1 = 0.f;
#pragma unroll
            for (int gq = 0; gq < 2; ++gq)
#pragma unroll
                for (int wm_ = 0; wm_ < 4 / WGN; ++wm_) {
                    const int w = gq * 4 + wm_ * WGN + wn_;
                    t0 += red[(w * 64 + ci) * 2];
                    t1 += red[(w * 64 + ci) * 2 + 1];
                }
            if (fin.acc) {
                bn_acc_add(fin, tmi, n0 + tid, t0, t1);
            } else {
                const int ch = D2S ? (n0 + tid) % g.d2s_c : n0 + tid;
                for (int sg = 0; sg < br.nseg; ++sg) {
                    const dsn_bnred_seg& z = br.seg[sg];
                    if (ch >= z.c0 && ch < z.c1) bn_acc_add(BnAcc{(double*)z.acc, z.acc_c, 0.0}, tmi, z.ch0 + ch - z.c0, t0, t1);
                }
            }
        }
    }
}

// epilogue form: 1 = straight from the accumulator registers (pp_epilogue_direct), 0 = staged through LDS (DSN_PP_DIR, dsn_pp_dir())
int g_pp_dir = getenv("DSN_PP_DIR") ? atoi(getenv("DSN_PP_DIR")) : 0;
inline bool pp_dir() { return g_pp_dir != 0; }

template <int BN, bool TWO>
int launch_pp(const dsn_tensor* s, const void* w, const float* bias, const dsn_tensor* r, const dsn_tensor* d, PGeom g, const BnAcc& fin,
              const BnRed* br, hipStream_t st) {
    g.tiles_y = (g.H + PT - 1) / PT;
    g.tiles_x = (g.W + PT - 1) / PT;
    g.tiles_n = (g.Cd + BN - 1) / BN;
    // (the register epilogue exists for the 128-channel one-block-per-CU tile: with 128 live accumulators or a 128-VGPR budget it spills)
    constexpr bool CAN_DIR = BN == 128 && !TWO;
    auto kern = (CAN_DIR && pp_dir()) ? (g.flip ? conv3x3_pp_kernel<BN, true, TWO, 9, CAN_DIR> : conv3x3_pp_kernel<BN, false, TWO, 9, CAN_DIR>)
                                      : (g.flip ? conv3x3_pp_kernel<BN, true, TWO> : conv3x3_pp_kernel<BN, false, TWO>);
    constexpr int LDS = TWO ? PP_LDS_TWO : PP_LDS;
    DSN_LDS_ATTR(kern, LDS);
    const int blocks = g.N * g.tiles_y * g.tiles_x * g.tiles_n;
    const double elems = (double)g.N * g.H * g.W * (g.Cs + (double)g.Cd * (1 + (r ? 1 : 0) + (g.accumulate ? 1 : 0)) + bnred_channels(br)) +
                         9.0 * g.Cs * g.Cd;
    const ProfConv pc(TWO ? "conv3x3_pp2_kernel" : "conv3x3_pp_kernel", true, 256, BN, g.flip != 0, 3, 1, 1, g.Cs, g.Cd, g.N, g.H, g.W);
    ProfScope prof(pc.label, pc.layer, 2.0 * g.N * g.H * g.W * g.Cd * 9.0 * g.Cs, elems * 2, st);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), LDS, st, (const bf16_t*)s->ptr, (const bf16_t*)w, bias,
                       r ? (const bf16_t*)r->ptr : nullptr, (bf16_t*)d->ptr, fin, g, br ? *br : BnRed{});
    DSN_LAUNCH_CHECK("conv3x3 (ping-pong big tile)");
    return DSN_OK;
}


template <int BN>
int launch_pp_s2(const dsn_tensor* dy, const void* w, const dsn_tensor* dx, PGeom g, const BnRed* br, hipStream_t st) {
    g.tiles_y = (g.H + PT - 1) / PT;
    g.tiles_x = (g.W + PT - 1) / PT;
    g.tiles_n = (g.Cd + BN - 1) / BN;
    constexpr bool CAN_DIR = BN == 128;
    auto kern = (CAN_DIR && pp_dir()) ? conv3x3_pp_kernel<BN, false, false, 4, CAN_DIR> : conv3x3_pp_kernel<BN, false, false, 4>;
    constexpr int LDS = PP_LDS;                  // 2 x 40 KB halo + 64 KB ring = 144 KB; the epilogue's staging needs 142 KB
    DSN_LDS_ATTR(kern, LDS);
    const int blocks = g.N * g.tiles_y * g.tiles_x * g.tiles_n;
    const double elems = (double)g.N * g.H * g.W * g.Cs + (double)g.N * g.Hout * g.Wout * (g.d2s_c * (1 + (g.accumulate ? 1 : 0)) + bnred_channels(br)) +
                         4.0 * g.Cs * g.Cd;
    const ProfConv pc("conv3x3_pp_kernel", true, 256, BN, true, 2, 2, 1, g.Cs, g.d2s_c, g.N, g.Hout, g.Wout);
    ProfScope prof(pc.label, pc.layer, 2.0 * g.N * g.H * g.W * g.d2s_c * 9.0 * g.Cs, elems * 2, st);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), LDS, st, (const bf16_t*)dy->ptr, (const bf16_t*)w, (const float*)nullptr,
                       (const bf16_t*)nullptr, (bf16_t*)dx->ptr, BnAcc{}, g, br ? *br : BnRed{});
    DSN_LAUNCH_CHECK("conv3x3 stride-2 data gradient (ping-pong big tile)");
    return DSN_OK;
}

// ---- 1x1 / stride-1 convolution (forward and data gradient) with the same two-group schedule ----------------------------------------
// C[M = N H W pixels][Cd] = A[M][Cs] * W[Cd][Cs]^T, block tile 256 consecutive pixels x BN channels.  Both operands are K-contiguous
// rows, so BOTH stream through the ring: piece j = k-half j (32 channels) = A part [256 rows][64 B] + B part [BN rows][64 B], rows
// swizzled like the weight pieces above.  BN 128: 6 stages x 24 KB, prefetch distance 4 (96 KB in flight per CU -- these layers
// are HBM-bound below ~512 channels, the prefetch depth is what matters); BN 256: 4 x 32 KB, distance 3, the piece issued in two
// halves -- the first phase of a k-half refills the B part and the A rows of pixel half 0 (last read two phases earlier), the
// second phase the A rows of pixel half 1 (the same rule as above: refill no earlier than two phases after the read).
// Per wave and piece: 2 A DMAs (rows 16 py .. + 16 for the wave's py of each pixel half) + BN / 128 B DMAs, a static schedule.
template <int BN, bool DIR = false>
__global__ __launch_bounds__(512, 2) void conv1x1_pp_kernel(const bf16_t* __restrict__ src, const bf16_t* __restrict__ wpk,
                                                            const float* __restrict__ bias, const bf16_t* __restrict__ res,
                                                            bf16_t* __restrict__ dst, const BnAcc fin, const PGeom g, const BnRed br) {
    static_assert(BN == 128 || BN == 256, "block tiles of 128 or 256 output channels");
    constexpr int MH = BN / 128, GW = BN / 128;
    constexpr int APIECE = 256 * 64;
    constexpr int PIECE = APIECE + BN * 64;
    constexpr int R = BN == 128 ? 6 : 4;
    constexpr int D = MH == 1 ? R - 2 : R - 1;
    constexpr int NPK = 2 + GW;                  // DMAs per wave and piece
    constexpr int WGN = BN == 128 ? 2 : 4;
    constexpr int VEC = 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = wave >> 2, wq = wave & 3;
    const int wm = wq / WGN, wn = wq % WGN;
    const int fr = lane & 15, fg = lane >> 4;
    const int tile = pp_xcd_remap(blockIdx.x, g.tiles_y * g.tiles_n);        // (tiles_y: 256-pixel tiles of the pixel list)
    const int tn = tile % g.tiles_n, tmi = tile / g.tiles_n;
    const int n0 = tn * BN;
    const int64_t M = (int64_t)g.N * g.H * g.W, m0 = (int64_t)tmi * 256;
#ifdef DSN_PP_STAMP
    const unsigned long long st_re = __builtin_amdgcn_s_memrealtime();
#endif

    const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, g.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wpk, 0, g.w_bytes, 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr;
    constexpr uint32_t OOB = 0xFFFFFFF0u;

    // lane -> (row lane >> 2 of the 16 rows of a DMA, physical slot lane & 3); logical slot fetched: the swizzle's inverse (an involution)
    const int dls = (lane & 3) ^ ((-(lane >> 4)) & 3);
    uint32_t aoff[2], woff[GW];
    int apy[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        apy[q] = (wave >> 2) * 8 + 4 * q + (wave & 3);                       // 16-row group: pixel half q of its group's rows
        const int64_t m = m0 + apy[q] * 16 + (lane >> 2);
        aoff[q] = m < M ? (uint32_t)((m * g.sld + dls * VEC) * 2) : OOB;
    }
#pragma unroll
    for (int part = 0; part < GW; ++part) {
        const int prow = 16 * (wave + 8 * part) + (lane >> 2);
        const int row = DIR ? pp_perm_row(prow) : prow;
        woff[part] = n0 + row < g.Cd ? (uint32_t)((int64_t)(n0 + row) * g.Cs + dls * VEC) * 2u : OOB;
    }
    const int T = g.Cs >> 5;                                                 // k-halves
    // (a wave one of whose DMA instructions has all lanes out of range -- rows past M in the last pixel tile, channels past Cd in a
    //  partial tile -- drains at every wait: such DMAs retire at once and the counts would stand for nothing)
    bool wall = true;
#pragma unroll
    for (int q = 0; q < 2; ++q) wall = wall && __ballot(aoff[q] != OOB) != 0ull;
#pragma unroll
    for (int part = 0; part < GW; ++part) wall = wall && __ballot(woff[part] != OOB) != 0ull;
    auto load_a = [&](int jp, int slot, int q) {
        const uint32_t off = (aoff[q] == OOB || jp >= T) ? OOB : aoff[q] + (uint32_t)jp * 64u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srsrc, (lds_ptr)(smem + slot * PIECE + apy[q] * 1024), 16, off, 0, 0, DSN_DMA_AUX);
    };
    auto load_b = [&](int jp, int slot, int part) {
        const uint32_t off = (woff[part] == OOB || jp >= T) ? OOB : woff[part] + (uint32_t)jp * 64u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_ptr)(smem + slot * PIECE + APIECE + (wave + 8 * part) * 1024), 16, off, 0, 0, DSN_DMA_AUX);
    };

    const int swz = (fg ^ ((-(fr >> 2)) & 3)) << 4;
    const int a_base = ((8 * grp + (MH == 2 ? 0 : 4 * wm)) * 16 + fr) * 64 + swz;
    const int b_base = APIECE + (wn * 64 + fr) * 64 + swz;

    f32x4 acc[MH][4][4];
#pragma unroll
    for (int m = 0; m < MH; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[m][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: pieces 0 .. D-1; piece 0 landed before the first barrier ---------------------------------------------------------
#pragma unroll
    for (int i = 0; i < D; ++i) {
        load_a(i, i % R, 0);
        load_a(i, i % R, 1);
#pragma unroll
        for (int part = 0; part < GW; ++part) load_b(i, i % R, part);
    }
    // (pieces past the end of K are all-out-of-range padding DMAs: they retire at once and must not be counted as in flight)
    if (!wall) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (T >= D || !g.tail) pp_wait_vm((D - 1) * NPK); else pp_wait_vm_rt((T - 1) * NPK);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (grp == 1) {
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }

#ifdef DSN_PP_STAMP
    const unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    u32x4 fa[4], fb[4];
    for (int jb = 0; jb < T; jb += R) {
#pragma unroll
        for (int jj = 0; jj < R; ++jj) {
            if (jb + jj >= T) break;                                         // (block-uniform)
            const unsigned char* st = smem + jj * PIECE;
            const int slot_w = (jj + D) % R;
#pragma unroll
            for (int mh = 0; mh < MH; ++mh) {
                // ---- load segment ----------------------------------------------------------------------------------------------
                if (mh == 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const u32x4*>(st + b_base + j * 1024);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const u32x4*>(st + a_base + (4 * mh + i) * 1024);
                const int rem = g.tail ? T - 2 - (jb + jj) : D;             // real pieces behind piece j + 1
                if (MH == 1) {
                    if (!wall) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    else if (rem >= D - 2) pp_wait_vm((D - 2) * NPK);       // piece j+1 landed (this wave's share)
                    else pp_wait_vm_rt(rem * NPK);                          // (tail: the padding DMAs retire at once)
                    load_a(jb + jj + D, slot_w, 0);
                    load_a(jb + jj + D, slot_w, 1);
                    load_b(jb + jj + D, slot_w, 0);
                } else if (mh == 0) {
                    load_b(jb + jj + D, slot_w, 0);
                    load_b(jb + jj + D, slot_w, 1);
                    load_a(jb + jj + D, slot_w, 0);
                } else {
                    if (!wall) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    else if (rem >= D - 1) pp_wait_vm((D - 2) * NPK + 3);   // (+ 3: the first-phase DMAs of piece j + D)
                    else pp_wait_vm_rt(rem * NPK);
                    load_a(jb + jj + D, slot_w, 1);
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // ---- MFMA segment ----------------------------------------------------------------------------------------------
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[mh][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, DIR ? fb[j] : fa[i]),
                                                                               __builtin_bit_cast(bf16x8, DIR ? fa[i] : fb[j]), acc[mh][i][j], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if (grp == 0) {
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
#ifdef DSN_PP_STAMP
    if (tid == 0 && blockIdx.x < 4096) {
        pp_stamp_buf[6 * blockIdx.x] = __builtin_amdgcn_s_memtime() - st_t0;
        pp_stamp_buf[6 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - st_r0;
        pp_stamp_buf[6 * blockIdx.x + 2] = st_re;
        pp_stamp_buf[6 * blockIdx.x + 3] = st_r0 - st_re;
    }
    const unsigned long long st_r1 = __builtin_amdgcn_s_memrealtime();
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MH; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(acc[m][i][j]));
    if constexpr (DIR) pp_epilogue_direct<BN, false, MH, true, false>(acc, smem, bias, res, dst, fin, g, br, 0, 0, 0, n0, tmi);
    else conv3x3_pp_epilogue<BN, false, MH, true>(acc, smem, bias, res, dst, fin, g, br, 0, 0, 0, n0, tmi);
#ifdef DSN_PP_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0 && blockIdx.x < 4096) pp_stamp_buf[6 * blockIdx.x + 4] = __builtin_amdgcn_s_memrealtime() - st_r1;
#endif
}

template <int BN>
int launch_pp1(const dsn_tensor* s, const void* w, const float* bias, const dsn_tensor* r, const dsn_tensor* d, PGeom g, const BnAcc& fin,
               const BnRed* br, hipStream_t st) {
    const int64_t M = (int64_t)g.N * g.H * g.W;
    g.tiles_y = (int32_t)((M + 255) / 256);
    g.tiles_x = 1;
    g.tiles_n = (g.Cd + BN - 1) / BN;
    constexpr bool CAN_DIR = BN == 128;
    auto kern = (CAN_DIR && pp_dir()) ? conv1x1_pp_kernel<BN, CAN_DIR> : conv1x1_pp_kernel<BN>;
    constexpr int LDS = (BN == 128 ? 6 * 24576 : 4 * 32768) > PP_LDS ? (BN == 128 ? 6 * 24576 : 4 * 32768) : PP_LDS;   // (ring; epilogue staging)
    DSN_LDS_ATTR(kern, LDS);
    const int blocks = g.tiles_y * g.tiles_n;
    const double elems = (double)M * (g.Cs + (double)g.Cd * (1 + (r ? 1 : 0) + (g.accumulate ? 1 : 0)) + bnred_channels(br)) + (double)g.Cs * g.Cd;
    const ProfConv pc("conv1x1_pp_kernel", true, 256, BN, g.flip != 0, 1, 1, 1, g.Cs, g.Cd, g.N, g.H, g.W);
    ProfScope prof(pc.label, pc.layer, 2.0 * M * g.Cd * g.Cs, elems * 2, st);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), LDS, st, (const bf16_t*)s->ptr, (const bf16_t*)w, bias,
                       r ? (const bf16_t*)r->ptr : nullptr, (bf16_t*)d->ptr, fin, g, br ? *br : BnRed{});
    DSN_LAUNCH_CHECK("conv1x1 (ping-pong big tile)");
    return DSN_OK;
}

}  // namespace

#ifdef DSN_PP_STAMP
// diagnostic build: (shader cycles, 100 MHz ticks) of the main loop of the first n blocks of the last launch
extern "C" int dsn_pp_stamp_read(unsigned long long* out, int32_t n) {
    if (n > 4096) n = 4096;
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pp_stamp_buf), (size_t)n * 6 * sizeof(unsigned long long));
}
#endif

extern "C" int dsn_pp_dir(int32_t on) {
    if (on >= 0) g_pp_dir = on;
    return g_pp_dir;
}

// selection mode (environment DSN_PP at load time, dsn_pp_mode() at run time: tests, A/B runs)
static int g_pp_mode = getenv("DSN_PP") ? atoi(getenv("DSN_PP")) : 1;
extern "C" int dsn_pp_mode(int32_t mode) {
    if (mode >= 0) g_pp_mode = mode;
    return g_pp_mode;
}

// Tried first by the 3x3 entry points of igemm.hip (same convention as dsn_conv3x3_halo_try: 1 = not this kernel's layer, nothing
// launched).  Mode: 0 never, 1 (default) layers with enough 256-pixel patches to fill the chip, 2 every eligible layer, 3 the same with
// 256-channel tiles wherever the channel count allows, 4 / 5 every eligible layer on 128-channel tiles with two / one block(s) per
// CU (tests, A/B runs).
int dsn_conv3x3_pp_try(const dsn_tensor* s, const void* w, const float* bias, const dsn_tensor* r, const dsn_tensor* d,
                       const dsn_conv_params* p, int is_dgrad, const BnAcc* finp, void* stream, const BnRed* br) {
    const int md = g_pp_mode;
    if (!md) return 1;
    if (s->dtype != DSN_BF16 || d->dtype != DSN_BF16) return 1;
    if (p->kh != 3 || p->kw != 3 || p->stride != 1 || p->pad != 1 || p->dil != 1) return 1;
    if (s->h != d->h || s->w != d->w || s->n != d->n) return 1;
    if (s->c % 64 != 0 || d->c % 8 != 0 || s->ldc % 8 != 0 || d->ldc % 8 != 0) return 1;
    if (((uintptr_t)s->ptr | (uintptr_t)d->ptr | (uintptr_t)w) % 16 != 0) return 1;
    if (r && (r->ldc % 8 != 0 || (uintptr_t)r->ptr % 16 != 0)) return 1;
    const int64_t sb = ((npix(s) - 1) * s->ldc + s->c) * 2, wb = (int64_t)d->c * 9 * s->c * 2;
    if (sb >= (1ll << 31) || wb >= (1ll << 31) || npix(d) * d->ldc * 2 >= (1ll << 40)) return 1;
    const int ty = (s->h + PT - 1) / PT, tx = (s->w + PT - 1) / PT;
    const int64_t patches = (int64_t)s->n * ty * tx;
    // Tile choice (measured on MI355X, tools/bench_ops.py, DSN_BENCH_SET=pp / m):
    //   256-channel tiles (per-wave 128 x 64: 12 fragment reads per 32 MFMAs instead of 16) from 200 patches up -- 512 -> 256 @4x160x160:
    //   198 us against 205 (128-channel tiles); 256 -> 256 @2x160x160: 54.8 against 57.1; with 100 patches (256 -> 256 @4x80x80) the
    //   100-block grid loses, 48.9 against 30.5 us;
    //   two blocks per CU where the 128-channel grid has more blocks than CUs and K is short (<= 2 slabs: prologue / epilogue bursts
    //   are a third of a block's life): 128 -> 128 @4x160x160 35.0 against 38.3 us, 64 -> 128 22.6 against 26.0; at 4 slabs it loses
    //   (62.2 against 59.4 us).
    static const int bn256_min = [] { const char* e = getenv("DSN_PP_BN256_MIN"); return e ? atoi(e) : 200; }();
    const bool wide = d->c % 256 == 0 && (md == 3 || (md < 4 && patches >= bn256_min));
    if (md == 1) {
        // worth it where the patches waste little of the map and the grid fills most of the chip
        static const int min_blocks = [] { const char* e = getenv("DSN_PP_MIN_BLOCKS"); return e ? atoi(e) : 160; }();
        const double fill = (double)s->h * s->w / ((double)ty * tx * PT * PT);
        const int64_t blocks = patches * ((d->c + 127) / 128);
        static const double min_fill = [] { const char* e = getenv("DSN_PP_MIN_FILL"); return e ? atof(e) : 0.8; }();
        if (d->c < 128 || fill < min_fill || blocks < min_blocks) return 1;
    }
    PGeom g{};
    g.N = s->n; g.H = s->h; g.W = s->w; g.Cs = s->c; g.Cd = d->c; g.flip = is_dgrad ? 1 : 0;
    g.act = p->act; g.accumulate = p->accumulate;
    g.sld = s->ldc; g.dld = d->ldc; g.rld = r ? r->ldc : 0;
    g.src_bytes = (uint32_t)sb; g.w_bytes = (uint32_t)wb;
    static const int tail_exact = [] { const char* e = getenv("DSN_PP_TAIL"); return e ? atoi(e) : 1; }();
    g.tail = tail_exact;
    g.nslab = s->c / 64;
    BnAcc fin{};
    if (finp) fin = *finp;
    const BnRed* brp = (br && br->nseg > 0) ? br : nullptr;
    if (wide) return launch_pp<256, false>(s, w, bias, r, d, g, fin, brp, (hipStream_t)stream);
    // two blocks per CU where the grid has more blocks than CUs (mode 4: always; mode 5: never)
    static const int two_min = [] { const char* e = getenv("DSN_PP_TWO_MIN"); return e ? atoi(e) : 257; }();
    static const int two_maxslab = [] { const char* e = getenv("DSN_PP_TWO_MAXSLAB"); return e ? atoi(e) : 2; }();
    const int64_t blocks128 = patches * ((d->c + 127) / 128);
    if (md != 5 && (md == 4 || (blocks128 >= two_min && g.nslab <= two_maxslab))) return launch_pp<128, true>(s, w, bias, r, d, g, fin, brp, (hipStream_t)stream);
    return launch_pp<128, false>(s, w, bias, r, d, g, fin, brp, (hipStream_t)stream);
}

// 1x1 / stride 1: tried first by the 1x1 chains of igemm.hip.  Mode (DSN_PP1 / dsn_pp1_mode): 0 never, 1 default (see below), 2 every
// eligible layer on 128-channel tiles, 3 the same with 256-channel tiles wherever the channel count allows.
static int g_pp1_mode = getenv("DSN_PP1") ? atoi(getenv("DSN_PP1")) : 1;
extern "C" int dsn_pp1_mode(int32_t mode) {
    if (mode >= 0) g_pp1_mode = mode;
    return g_pp1_mode;
}

int dsn_conv1x1_pp_try(const dsn_tensor* s, const void* w, const float* bias, const dsn_tensor* r, const dsn_tensor* d,
                       const dsn_conv_params* p, int is_dgrad, const BnAcc* finp, void* stream, const BnRed* br) {
    const int md = g_pp1_mode;
    if (!md) return 1;
    if (s->dtype != DSN_BF16 || d->dtype != DSN_BF16) return 1;
    if (p->kh != 1 || p->kw != 1 || p->stride != 1 || p->pad != 0) return 1;
    if (s->h != d->h || s->w != d->w || s->n != d->n) return 1;
    if (s->c % 32 != 0 || d->c % 8 != 0 || s->ldc % 8 != 0 || d->ldc % 8 != 0) return 1;
    if (((uintptr_t)s->ptr | (uintptr_t)d->ptr | (uintptr_t)w) % 16 != 0) return 1;
    if (r && (r->ldc % 8 != 0 || (uintptr_t)r->ptr % 16 != 0)) return 1;
    const int64_t sb = ((npix(s) - 1) * s->ldc + s->c) * 2, wb = (int64_t)d->c * s->c * 2;
    if (sb >= (1ll << 31) || wb >= (1ll << 31) || npix(d) * d->ldc * 2 >= (1ll << 40)) return 1;
    const int64_t tiles_m = (npix(s) + 255) / 256;
    bool wide = d->c % 256 == 0 && md == 3;
    if (md == 1) {
        // Chosen from per-layer times INSIDE config 5's training step (profiles/r04c_pp1_layers_in_step.txt; the stand-alone
        // microbenchmark flatters this kernel: in the step the forward carries the BatchNorm sums and the data gradient the
        // BatchNorm-backward sums in the epilogue).  us per launch, weights-stationary / implicit-GEMM kernels -> this one:
        //   768 -> 256 @4x160x160 fwd 117.5 -> 75.9 (256-channel tiles), 512 -> 512 @4x80x80 fwd 37.0 -> 32.1 / dgrad 51.1 -> 43.2 (256),
        //   512 -> 256 @4x160x160 fwd 70.0 -> 61.9 (256), 1024 -> 512 @4x80x80 fwd 63.5 -> 43.0 (256), 1024 -> 1024 @4x40x40 fwd 32.5 -> 26.0 /
        //   dgrad 36.8 -> 33.2 (128), 2048 -> 1024 @4x40x40 fwd 50.4 -> 42.1 / its twin dgrad 53.1 -> 42.8 (128), 512 -> 256 @4x80x80 fwd 22.6 -> 19.0 (128);
        //   it loses below 512 input channels (256 -> 256 @4x80x80 fwd 13.8 -> 16.4, 128 -> 128 @4x160x160 16.5 -> 22.8: pure streams),
        //   on grids of ~100 blocks (512 -> 512 @4x40x40 14.1 -> 16.6), and -- with the BatchNorm-backward sums -- for 256 output
        //   channels at K = 512 (256 -> 512 @4x160x160 dgrad 91.7 -> 101.6).  256-channel tiles from 200 blocks up; at 100
        //   blocks they lose (1024 -> 1024 @4x40x40 fwd 32.5 -> 39.5).
        static const int min_blocks = [] { const char* e = getenv("DSN_PP1_MIN_BLOCKS"); return e ? atoi(e) : 160; }();
        static const int min_k = [] { const char* e = getenv("DSN_PP1_MIN_K"); return e ? atoi(e) : 512; }();
        if (d->c < 256 || s->c < min_k) return 1;
        if (br && br->nseg > 0 && d->c < 512 && s->c < 1024) return 1;
        wide = d->c % 256 == 0 && tiles_m * (d->c / 256) >= 200;
        if (!wide && tiles_m * ((d->c + 127) / 128) < min_blocks) return 1;
    }
    PGeom g{};
    g.N = s->n; g.H = s->h; g.W = s->w; g.Cs = s->c; g.Cd = d->c; g.flip = is_dgrad ? 1 : 0;
    g.act = p->act; g.accumulate = p->accumulate;
    g.sld = s->ldc; g.dld = d->ldc; g.rld = r ? r->ldc : 0;
    g.src_bytes = (uint32_t)sb; g.w_bytes = (uint32_t)wb;
    static const int tail_exact = [] { const char* e = getenv("DSN_PP_TAIL"); return e ? atoi(e) : 1; }();
    g.tail = tail_exact;
    g.nslab = 0;
    BnAcc fin{};
    if (finp) fin = *finp;
    const BnRed* brp = (br && br->nseg > 0) ? br : nullptr;
    if (wide) return launch_pp1<256>(s, w, bias, r, d, g, fin, brp, (hipStream_t)stream);
    return launch_pp1<128>(s, w, bias, r, d, g, fin, brp, (hipStream_t)stream);
}

// 3x3 / stride-2 / pad-1 data gradient in its 2x2 form (weights dsn_pack_desc.out_dgrad_s2): tried first by dsn_conv2d_dgrad_s2.
// Same mode switch as the 3x3 kernel (DSN_PP / dsn_pp_mode): 0 never, 1 default, >= 2 every eligible layer (3: 256-column tiles
// wherever 4 Ci allows).
int dsn_dgrad_s2_pp_try(const dsn_tensor* dy, const void* w_s2, const dsn_tensor* dx, const dsn_conv_params* p, const BnRed* br,
                        void* stream) {
    const int md = g_pp_mode;
    if (!md) return 1;
    if (dy->dtype != DSN_BF16 || dx->dtype != DSN_BF16) return 1;
    if (dy->c % 64 != 0 || dx->c % 8 != 0 || dy->ldc % 8 != 0 || dx->ldc % 8 != 0) return 1;
    if (((uintptr_t)dy->ptr | (uintptr_t)dx->ptr | (uintptr_t)w_s2) % 16 != 0) return 1;
    const int64_t sb = ((npix(dy) - 1) * dy->ldc + dy->c) * 2, wb = (int64_t)4 * dx->c * 4 * dy->c * 2;
    if (sb >= (1ll << 31) || wb >= (1ll << 31) || npix(dx) * dx->ldc * 2 >= (1ll << 40)) return 1;
    const int ty = (dy->h + PT - 1) / PT, tx = (dy->w + PT - 1) / PT;
    const int64_t patches = (int64_t)dy->n * ty * tx;
    const int cd = 4 * dx->c;
    bool wide = cd % 256 == 0 && (md == 3 || md == 1);
    if (md == 1) {
        static const int min_blocks = [] { const char* e = getenv("DSN_PP_S2_MIN_BLOCKS"); return e ? atoi(e) : 64; }();
        // (in-step, config 5, us: implicit GEMM -> this kernel: 128 -> 4 x 64 @4x320x320 380 -> 309, 512 -> 4 x 256 @4x80x80 205 -> 152,
        //  256 -> 4 x 256 @4x80x80 145 -> 115, 256 -> 4 x 128 @4x160x160 228 -> 212; 40 x 40 and 20 x 20 dy maps -- 16 x 16 patches
        //  cover 69 % / 39 % of them -- lose: 190 -> 199, 120 -> 137; config 3: 128 -> 4 x 64 @8x80x80 59 -> 54, 64 -> 4 x 32 @8x160x160 89 -> 89)
        static const double min_fill = [] { const char* e = getenv("DSN_PP_S2_MIN_FILL"); return e ? atof(e) : 0.8; }();
        const double fill = (double)dy->h * dy->w / ((double)ty * tx * PT * PT);
        wide = cd % 256 == 0 && patches * (cd / 256) >= 200;
        // one-slab layers (64 dy channels: config 3's layer 1) are all prologue + epilogue here: 89 us against 62-65 on the gather form
        if (cd < 128 || dy->c < 128 || fill < min_fill || patches * ((cd + 127) / 128) < min_blocks) return 1;
    }
    PGeom g{};
    g.N = dy->n; g.H = dy->h; g.W = dy->w; g.Cs = dy->c; g.Cd = cd; g.flip = 0;
    g.act = DSN_ACT_NONE; g.accumulate = p->accumulate;
    g.sld = dy->ldc; g.dld = dx->ldc; g.rld = 0;
    g.src_bytes = (uint32_t)sb; g.w_bytes = (uint32_t)wb;
    static const int tail_exact = [] { const char* e = getenv("DSN_PP_TAIL"); return e ? atoi(e) : 1; }();
    g.tail = tail_exact;
    g.nslab = dy->c / 64;
    g.d2s_c = dx->c; g.Hout = dx->h; g.Wout = dx->w;
    const BnRed* brp = (br && br->nseg > 0) ? br : nullptr;
    if (wide) return launch_pp_s2<256>(dy, w_s2, dx, g, brp, (hipStream_t)stream);
    return launch_pp_s2<128>(dy, w_s2, dx, g, brp, (hipStream_t)stream);
}
