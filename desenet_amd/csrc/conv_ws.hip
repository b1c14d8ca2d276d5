// Weights-stationary, persistent convolution kernels (round 3): 1x1 / stride 1 and 3x3 / stride 1 (dilation 1..3), forward and
// data gradient -- Conv, Bottleneck, C3, SPP, RFB2, FFM and the seg head's 1x1s (common.py:42-56,101-145,172-185,222-242,504-545;
// yolo.py:161-181) and ATen's convolution_backward (input) behind them.
//
// Why: per-layer rocprofv3 / HIP-event tables (profiles/r03*) showed the one-trip kernels of conv3x3.hip bound by what every block
// pulls through L2, not by HBM or MFMA: a 64-pixel block re-fetches its whole weight tile (64 x K x 2 B = 16..300 KB) for 8..16 KB of
// input -- a 3x3 64 -> 64 @ 8x80x80 layer moves 59 MB of weights for 6.5 MB of activations, the same for every C3 3x3 of the net
// (the three have equal MACs by design), and each block's life is one exposed trip to L2 + a short MFMA burst + an LDS-staged store.
// Here a block is PERSISTENT: it loads the weights of its output-channel tile into LDS once and then walks its share of the pixel
// tiles, with the next tile's input arriving by LDS-DMA (buffer_load ... lds) while the current one is multiplied and stored:
//   * weights: one trip per block (<= 256 x bpc blocks) instead of one per 64 pixels;
//   * input: double-buffered tiles, one counted s_waitcnt vmcnt per tile (stores of the previous tile stay in flight), one raw
//     s_barrier per tile, no barrier inside a tile (the weights are resident: nothing streams but the pixels);
//   * MFMA with the operands swapped (A = weight rows, B = pixels): the accumulator then holds 4 consecutive CHANNELS of one
//     pixel per lane, and with the weight rows of two 16-channel tiles interleaved in LDS (row order c0 + 8*(i>>2) + 4*t + (i&3))
//     a lane owns 8 consecutive bf16 channels = one 16-byte store straight from registers: no LDS staging tile, no second barrier;
//   * BatchNorm partial sums (training forward) stay in registers across all tiles of the block and reach the fp64 accumulators
//     once per block.
// Layout contracts are those of conv3x3.hip (NHWC activations with a pixel stride, weights [Co][KH][KW][Ci] / [Ci][KH][KW][Co] for
// the data gradient, 128-byte LDS rows with the slot XOR swizzle applied on the source side of the DMA).
#include <stdlib.h>

#include "common.h"

namespace {

struct WGeom {
    int32_t N, H, W;          // images, map (source and destination have the same size)
    int32_t Cs, Cd;           // source / destination channels
    int32_t d, flip;          // 3x3: dilation (= padding); 1: data gradient (mirrored taps)
    int32_t act;
    int64_t sld, dld;
    uint32_t src_bytes, w_bytes, dst_bytes;
    int32_t tiles_m, tiles_n; // pixel tiles (3x3: N * tiles_y * tiles_x patches), channel tiles
    int32_t tiles_y, tiles_x;
    int32_t HW, NP;           // 3x3: halo columns / pixels
    int32_t wrow;             // elements per weight row (taps * Cs)
    // gather forms of the 1x1 kernel (G > 0): rows are pixels of the (H, W) grid above, K = taps x gCt channels fetched from
    // source pixel (y * gS + dy, x * gS + dx) of a (gHs, gWs) map; Cs = taps * gCt.  d2s_c > 0: GEMM column (cls, ci) of row
    // (n, y, x) is stored at pixel (n, 2y + cls / 2, 2x + cls % 2), channel ci, of a (Hout, Wout) map (stride-2 data gradient).
    int32_t gS, gHs, gWs, gCt;
    int32_t d2s_c, Hout, Wout;
};

template <typename T> struct WMma;
template <> struct WMma<float> {
    static constexpr int VEC = 4;
    // D[channel][pixel] += W[channel][k] * X[pixel][k]
    __device__ static __forceinline__ void run(f32x4& acc, const u32x4& w, const u32x4& x) {
        const f32x4 wf = __builtin_bit_cast(f32x4, w), xf = __builtin_bit_cast(f32x4, x);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[0], xf[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[1], xf[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[2], xf[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[3], xf[3], acc, 0, 0, 0);
    }
};
template <> struct WMma<bf16_t> {
    static constexpr int VEC = 8;
    __device__ static __forceinline__ void run(f32x4& acc, const u32x4& w, const u32x4& x) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), acc, 0, 0, 0);
    }
};

constexpr int ROWB = 128;
constexpr uint32_t OOB = 0xFFFFFFF0u;
typedef __attribute__((address_space(3))) void* lds_ptr;

// Hand-issued LDS fragment reads.  hipcc schedules ds_reads as late as it can (registers are what it minimises), i.e. right in
// front of the MFMA that consumes them; with one wave per SIMD that exposes the full LDS latency at every k-step.  Issued from
// inline asm the reads keep their place -- PF steps ahead of their MFMAs -- and the wave waits with a COUNTED lgkmcnt that leaves
// the younger steps in flight (cdna_hip_programming.md 5.7 form (iii): "=v" loads, a wait-only statement, sched_barrier(0) so
// that no MFMA is hoisted above the wait).  addr: 32-bit LDS byte address.
#define DS_READ_B128(dst, addr) asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr))
// s_waitcnt vmcnt(BASE + n * STEP) for a wave-uniform n in 0 .. 3: the DMA ring waits leave (D - 2) younger tiles and the n store
// groups issued since the awaited tile's DMA in flight.  (The counts rest on what tools/exp/oob_order.hip measured on this part: cold
// LDS-DMAs of distinct lines and stores complete in issue order; register loads overtake DMAs and DMAs whose lanes are all out of
// range retire at once -- neither is ever counted: see wait_ring / DeadOps below.)
template <int BASE, int STEP> __device__ __forceinline__ void wait_vm(int n) {
    static_assert(BASE + 3 * STEP <= 63, "vmcnt is a 6-bit counter");
    if (n <= 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BASE) : "memory");
    else if (n == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BASE + STEP) : "memory");
    else if (n == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BASE + 2 * STEP) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BASE + 3 * STEP) : "memory");
}
// The same with `real` (0 .. D - 2) of the younger DMA groups being REAL ones: the groups requested for tiles past a block's share are
// all-out-of-range padding DMAs, which retire at once (round 4, tools/exp/oob_order.hip) and must not stand for operations in flight.
template <int D, int GSZ, int STEP> __device__ __forceinline__ void wait_ring(int real, int n) {
    static_assert(D >= 2 && D <= 4, "ring depth");
    if constexpr (D == 2) { wait_vm<0, STEP>(n); }
    else if constexpr (D == 3) { if (real >= 1) wait_vm<GSZ, STEP>(n); else wait_vm<0, STEP>(n); }
    else { if (real >= 2) wait_vm<2 * GSZ, STEP>(n); else if (real == 1) wait_vm<GSZ, STEP>(n); else wait_vm<0, STEP>(n); }
}
// prologue: weights + the first group landed; of the D - 2 younger groups only the real ones may be counted (as above)
template <int D, int GSZ> __device__ __forceinline__ void wait_prologue(int real) {
    static_assert(D >= 2 && D <= 4, "ring depth");
    if (D >= 4 && real >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * GSZ) : "memory");
    else if (D >= 3 && real >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GSZ) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// (A wave with dead operations among the counted ones -- see DeadOps below -- DRAINS: an exact runtime count needs a compare tree
//  over the 6-bit immediate at every wait, measured +11 % on the short 3x3 kernels of the 20x20 / 40x40 maps for its code size alone.)
__device__ __forceinline__ void wait_vm_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// Per-wave ledger of the vector-memory operations that retire AT ONCE because every lane is out of range (tools/exp/oob_order.hip):
// border rows of a halo, pixel rows past M, channel tiles past Cd, padding pixels of a halo stage -- DMAs and stores alike (a dead
// store is taken for instant too: subtracting it can only lengthen a wait).  The static counts of the ring waits take every counted
// operation for one in flight; a wave subtracts its dead ones among them.  g0 / g1: dead DMAs of the youngest / second youngest DMA
// group (0 for the padding groups past a block's share: the static counts leave those out already), s0 .. s2: of the youngest store groups.
struct DeadOps {
    int g0, g1, s0, s1, s2;
    __device__ __forceinline__ void init() { g0 = g1 = s0 = s1 = s2 = 0; }
    __device__ __forceinline__ void dma(int d) { g1 = g0; g0 = d; }
    __device__ __forceinline__ void st(int d) { s2 = s1; s1 = s0; s0 = d; }
    template <int D> __device__ __forceinline__ int among(int nst) const {
        return (D >= 3 ? g0 : 0) + (D >= 4 ? g1 : 0) + (nst >= 1 ? s0 : 0) + (nst >= 2 ? s1 : 0) + (nst >= 3 ? s2 : 0);
    }
};
__device__ __forceinline__ int all_out(bool lane_in_range) { return __builtin_amdgcn_readfirstlane(__ballot(lane_in_range) == 0ull ? 1 : 0); }
// (the ballots are skipped where a wave-uniform test on the tile's position rules dead operations out: interior tiles)
__device__ __forceinline__ int all_out(bool check, bool lane_in_range) { return check ? all_out(lane_in_range) : 0; }
template <int D, int GSZ, int STEP> __device__ __forceinline__ void wait_ring(int real, int n, int dead) {
    if (dead == 0) wait_ring<D, GSZ, STEP>(real, n);
    else wait_vm_drain();
}
template <int D, int GSZ> __device__ __forceinline__ void wait_prologue(int real, int dead) {
    if (dead == 0) wait_prologue<D, GSZ>(real);
    else wait_vm_drain();
}
// The same for a wave `pad` of whose GSZ DMAs per group are dead in EVERY group (the padding pixels past the end of a halo stage):
// those are not kept in the ledger, the wave's groups simply are GSZ - pad operations long (one more static form).
template <int D, int GSZ, int STEP> __device__ __forceinline__ void wait_ring(int real, int n, int dead, int pad) {
    if (pad == 0) wait_ring<D, GSZ, STEP>(real, n, dead);
    else if (dead == 0 && pad == 1) wait_ring<D, GSZ - 1, STEP>(real, n);
    else wait_vm_drain();
}
template <int D, int GSZ> __device__ __forceinline__ void wait_prologue(int real, int dead, int pad) {
    if (pad == 0) wait_prologue<D, GSZ>(real, dead);
    else if (dead == 0 && pad == 1) wait_prologue<D, GSZ - 1>(real);
    else wait_vm_drain();
}
template <int N> __device__ __forceinline__ void wait_lgkm() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// LDS row q of a block's weight tile holds output channel wrow_channel(q): identity for fp32 (a lane's 4 accumulator rows are 4
// consecutive channels = 16 bytes), and for bf16 the rows of two 16-channel MFMA tiles interleaved in groups of 4 so that tile
// 2p gives a lane channels 8*fg + 0..3 and tile 2p + 1 channels 8*fg + 4..7 of the 32-channel pair p.
template <typename T> __device__ __forceinline__ int wrow_channel(int q) {
    if constexpr (sizeof(T) == 4) return q;
    const int p = q >> 5, t = (q >> 4) & 1, i = q & 15;
    return p * 32 + (i >> 2) * 8 + t * 4 + (i & 3);
}

// ---- epilogue of one (pixel tile i, channel vector v) of a lane: bias, activation, BatchNorm sums, one 16-byte store ------------
template <typename T, int NI> struct OutVec;
template <int NI> struct OutVec<bf16_t, NI> {
    static constexpr int NV = NI / 2, CPV = 8;
    // channel offset (inside the block's BN channels) of vector v of the lane's wave-column wn, lane group fg
    __device__ static __forceinline__ int ch(int wn, int v, int fg) { return (wn * NI + 2 * v) * 16 + fg * 8; }
    __device__ static __forceinline__ void get(const f32x4 (&acc)[NI], int v, float (&o)[8]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e] = acc[2 * v][e]; o[4 + e] = acc[2 * v + 1][e]; }
    }
};
template <int NI> struct OutVec<float, NI> {
    static constexpr int NV = NI, CPV = 4;
    __device__ static __forceinline__ int ch(int wn, int v, int fg) { return (wn * NI + v) * 16 + fg * 4; }
    __device__ static __forceinline__ void get(const f32x4 (&acc)[NI], int v, float (&o)[4]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = acc[v][e];
    }
};

template <typename T, int CPV> __device__ __forceinline__ u32x4 pack_out(const float (&o)[CPV]) {
    if constexpr (sizeof(T) == 4) {
        return __builtin_bit_cast(u32x4, f32x4{o[0], o[1], o[2], o[3]});
    } else {
        bf16x8 v;
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (bf16_t)o[k];
        return __builtin_bit_cast(u32x4, v);
    }
}

// Per-lane BatchNorm partial sums of the block's output (training forward: the conv is plain, the sums are of the fp32 accumulators
// exactly as conv3x3.hip's epilogue forms them), kept across tiles; finish() folds lanes -> waves -> fp64 accumulators.
template <int NV, int CPV, bool ON> struct StatRegs {
    float s[ON ? NV : 1][ON ? CPV : 1], ss[ON ? NV : 1][ON ? CPV : 1];
    __device__ __forceinline__ void zero() {
        if constexpr (ON) {
#pragma unroll
            for (int v = 0; v < NV; ++v)
#pragma unroll
                for (int k = 0; k < CPV; ++k) s[v][k] = ss[v][k] = 0.f;
        }
    }
    __device__ __forceinline__ void add(int v, const float (&o)[CPV], bool valid) {
        if constexpr (ON) {
#pragma unroll
            for (int k = 0; k < CPV; ++k) {
                const float x = valid ? o[k] : 0.f;
                s[v][k] += x;
                ss[v][k] += x * x;
            }
        }
    }
};

// ---- data-gradient extras: shortcut gradient, accumulation into dx, BatchNorm backward sums (include/desenet_hip.h: dsn_bnred) ----
// A dgrad launch may add a residual tensor (a Bottleneck shortcut's gradient), add to what dx already holds (fan-in), and -- when it
// writes the FINAL dz of one or two BatchNorm blocks -- form their backward sums from the value it stores and the producer's y.
// These operands are per-lane 16-byte vectors at exactly the (pixel, channel vector) positions the lane stores; they are fetched by
// hand-issued buffer loads ONE TILE AHEAD into a ping-pong register set (an ordinary load next to in-flight LDS-DMA makes hipcc
// drain the whole vector-memory queue at its first use: cdna_hip_programming.md 5, trap (b)) and waited for with counted vmcnt.
struct WsX {
    const void* res;
    int64_t rld;
    uint32_t res_bytes;
    int32_t accumulate;
    int32_t nseg, _pad;
    struct Seg {
        int32_t c0, c1, ch0, acc_c, act, _pad;
        const void* y;
        int64_t yld;
        uint32_t y_bytes, _pad2;
        const float *scale, *shift, *mean, *rstd;
        double* acc;
    } seg[2];
};

__device__ __forceinline__ u32x4 make_rsrc4(const void* p, uint32_t bytes) {
    const uint64_t a = (uint64_t)p;
    return u32x4{(uint32_t)a, (uint32_t)(a >> 32) & 0xFFFFu, bytes, 0x00020000u};
}
// (descriptor words come from kernel arguments: wave-uniform; s_nop 4: SGPR written by a VALU/SALU just before a VMEM reads it)
#define BUF_LOAD_B128(dst, off, rsrc) asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(dst) : "v"(off), "s"(rsrc) : "memory")

template <typename T, int MI, int NV, int CPV> struct WsXRegs { u32x4 res[MI][NV], old[MI][NV], y0[MI][NV], y1[MI][NV]; };

template <typename T, int MI, int NV, int CPV> struct WsXState {
    static constexpr int XL = 4 * MI * NV;               // buffer loads per wave and tile (always issued; absent operands: out of range)
    u32x4 r_res, r_old, r_y0, r_y1;
    int segv[NV];                                        // segment of the lane's vector v (-1: none)
    int actv[NV];
    float sc[NV][CPV], sh[NV][CPV], mu[NV][CPV], rs[NV][CPV], q0[NV][CPV], q1[NV][CPV];
    __device__ __forceinline__ void init(const WsX& ex, const void* dst, uint32_t dst_bytes, const int (&cabs)[NV], int Cd) {
        r_res = make_rsrc4(ex.res, ex.res ? ex.res_bytes : 0);
        r_old = make_rsrc4(dst, ex.accumulate ? dst_bytes : 0);
        r_y0 = make_rsrc4(ex.seg[0].y, ex.nseg > 0 ? ex.seg[0].y_bytes : 0);
        r_y1 = make_rsrc4(ex.seg[1].y, ex.nseg > 1 ? ex.seg[1].y_bytes : 0);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            segv[v] = -1;
            actv[v] = 0;
#pragma unroll
            for (int k = 0; k < CPV; ++k) { sc[v][k] = sh[v][k] = mu[v][k] = rs[v][k] = 0.f; q0[v][k] = q1[v][k] = 0.f; }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if (s < ex.nseg && cabs[v] >= ex.seg[s].c0 && cabs[v] < ex.seg[s].c1 && cabs[v] < Cd) {
                    segv[v] = s;
                    actv[v] = ex.seg[s].act;
                    const int k0 = cabs[v] - ex.seg[s].c0;
#pragma unroll
                    for (int k = 0; k < CPV; ++k) {
                        sc[v][k] = ex.seg[s].scale[k0 + k]; sh[v][k] = ex.seg[s].shift[k0 + k];
                        mu[v][k] = ex.seg[s].mean[k0 + k]; rs[v][k] = ex.seg[s].rstd[k0 + k];
                    }
                }
            }
        }
    }
    // request the operands of one tile: pix[i] = destination pixel index of the lane's pixel in tile row i (or -1: not stored)
    __device__ __forceinline__ void issue(WsXRegs<T, MI, NV, CPV>& x, const WsX& ex, const int64_t (&pix)[MI], const int (&cabs)[NV],
                                          int64_t dld, int Cd) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const bool ok = pix[i] >= 0 && cabs[v] < Cd;
                const uint32_t o_res = ok ? (uint32_t)(pix[i] * ex.rld + cabs[v]) * (uint32_t)sizeof(T) : OOB;
                const uint32_t o_old = ok ? (uint32_t)(pix[i] * dld + cabs[v]) * (uint32_t)sizeof(T) : OOB;
                const uint32_t o_y0 = (ok && segv[v] == 0) ? (uint32_t)(pix[i] * ex.seg[0].yld + (cabs[v] - ex.seg[0].c0)) * (uint32_t)sizeof(T) : OOB;
                const uint32_t o_y1 = (ok && segv[v] == 1) ? (uint32_t)(pix[i] * ex.seg[1].yld + (cabs[v] - ex.seg[1].c0)) * (uint32_t)sizeof(T) : OOB;
                BUF_LOAD_B128(x.res[i][v], o_res, r_res);
                BUF_LOAD_B128(x.old[i][v], o_old, r_old);
                BUF_LOAD_B128(x.y0[i][v], o_y0, r_y0);
                BUF_LOAD_B128(x.y1[i][v], o_y1, r_y1);
            }
    }
    // the same with a destination pixel per (tile row, vector) and the vectors' channels given (depth-to-space store)
    __device__ __forceinline__ void issue_pv(WsXRegs<T, MI, NV, CPV>& x, const WsX& ex, const int64_t (&pix)[MI][NV], const int (&cv)[NV],
                                             int64_t dld) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const bool ok = pix[i][v] >= 0;
                const uint32_t o_res = ok ? (uint32_t)(pix[i][v] * ex.rld + cv[v]) * (uint32_t)sizeof(T) : OOB;
                const uint32_t o_old = ok ? (uint32_t)(pix[i][v] * dld + cv[v]) * (uint32_t)sizeof(T) : OOB;
                const uint32_t o_y0 = (ok && segv[v] == 0) ? (uint32_t)(pix[i][v] * ex.seg[0].yld + (cv[v] - ex.seg[0].c0)) * (uint32_t)sizeof(T) : OOB;
                const uint32_t o_y1 = (ok && segv[v] == 1) ? (uint32_t)(pix[i][v] * ex.seg[1].yld + (cv[v] - ex.seg[1].c0)) * (uint32_t)sizeof(T) : OOB;
                BUF_LOAD_B128(x.res[i][v], o_res, r_res);
                BUF_LOAD_B128(x.old[i][v], o_old, r_old);
                BUF_LOAD_B128(x.y0[i][v], o_y0, r_y0);
                BUF_LOAD_B128(x.y1[i][v], o_y1, r_y1);
            }
    }
    // o: accumulator values of (tile row i, vector v) -> + residual + old; returns the packed stored value and adds the BN sums
    __device__ __forceinline__ u32x4 apply(const WsXRegs<T, MI, NV, CPV>& x, int i, int v, float (&o)[CPV], bool ok) {
        T rv[CPV], ov[CPV], yv[CPV];
        *reinterpret_cast<u32x4*>(rv) = x.res[i][v];
        *reinterpret_cast<u32x4*>(ov) = x.old[i][v];
        *reinterpret_cast<u32x4*>(yv) = segv[v] == 1 ? x.y1[i][v] : x.y0[i][v];
#pragma unroll
        for (int k = 0; k < CPV; ++k) {       // (absent operands read as zeros; the shortcut first, then the old value: the order of
            o[k] += to_f32<T>(rv[k]);         //  every other epilogue -- with both present `o += rv + ov` rounds differently, round 4's
            o[k] += to_f32<T>(ov[k]);         //  randomised sweep found 43 of 2.4 M elements one bf16 ulp apart)
        }
        const u32x4 packed = pack_out<T, CPV>(o);
        if (segv[v] >= 0 && ok) {
            T outv[CPV];
            *reinterpret_cast<u32x4*>(outv) = packed;
            float u[CPV], gr[CPV];
#pragma unroll
            for (int k = 0; k < CPV; ++k) u[k] = to_f32<T>(yv[k]) * sc[v][k] + sh[v][k];
            act_grad_vec<CPV>(u, actv[v], gr);
#pragma unroll
            for (int k = 0; k < CPV; ++k) {
                const float y = to_f32<T>(yv[k]);
                const float gk = to_f32<T>(outv[k]) * gr[k];
                q0[v][k] += gk;
                q1[v][k] += gk * ((y - mu[v][k]) * rs[v][k]);
            }
        }
        return packed;
    }
};

// ================================================================================================================================
// 1x1 / stride 1.  Block tile BM pixels x BN channels, K = NS slabs of 128 bytes (any channel count that is a multiple of the
// 16-byte vector: lanes past the last channel fetch nothing and leave zeros).  LDS: [weights NS x BN rows][2 stages x NS x BM rows].
// ================================================================================================================================
// G: 0 = 1x1 / stride 1.  Gather forms (round 3) -- the same GEMM with the K axis assembled from several source pixels per row:
//    1 = 3x3 / stride 2 / pad 1 forward (K = 9 taps x gCt channels, tap (ky, kx) of output (y, x) at source (2y + ky - 1, 2x + kx - 1));
//    2 = its data gradient as the 2x2 / stride-1 convolution over dy of igemm.hip (K = 4 taps x gCt, tap (ty, tx) at (y + ty, x + tx),
//        4 * Ci columns, depth-to-space store).  A 16-byte slot never straddles a tap (gCt % 8 == 0); slots past the last tap, and
//        taps that fall outside the map, are out-of-range lanes of the DMA = zeros.  The implicit-GEMM kernel ran these layers at
//        145-220 TFLOP/s with 18 VALU instructions per MFMA and the weights re-fetched per 64-pixel block.
template <typename T, int MI, int NI, int WGM, int WGN, int NS, bool STATS, int D, bool EX, int G = 0>
__global__ __launch_bounds__(256) void conv1x1_ws_kernel(const T* __restrict__ src, const T* __restrict__ wpk,
                                                         const float* __restrict__ bias, T* __restrict__ dst, const BnAcc fin,
                                                         const WGeom g, const WsX ex) {
    static_assert(!EX || (D == 3 && !STATS), "the extras variant (data gradient) runs a 3-stage ring without BatchNorm forward sums");
    static_assert(WGM * WGN == 4, "4 waves per block");
    constexpr int BM = WGM * MI * 16, BN = WGN * NI * 16;
    static_assert(BM % 32 == 0 && BN % 32 == 0, "tiles are filled 32 rows per DMA pass");
    constexpr int VEC = WMma<T>::VEC;
    constexpr int KC = ROWB / (int)sizeof(T);
    constexpr int AR = BM / 32, BR = BN / 32;
    typedef OutVec<T, NI> OV;
    constexpr int NV = OV::NV, CPV = OV::CPV;
    constexpr int ST = MI * NV;                      // store instructions per wave and tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sW = smem;                                  // [NS][BN][128 B]
    static_assert(D >= 2 && D <= 4, "ring of 2 .. 4 pixel-tile stages");
    constexpr int LPT = NS * AR;                      // LDS-DMA instructions per wave and pixel tile
    unsigned char* sA = smem + NS * BN * ROWB;                 // [D][NS][BM][128 B]
    // [WGM][BN][2] fold buffer of the sums (STATS / extras): behind the ring; the gather forms reuse the weight area (dead after the
    // last tile, and no DMA targets it) so that weights + ring may fill the LDS exactly
    float* sRed = G > 0 ? reinterpret_cast<float*>(sW) : reinterpret_cast<float*>(sA + D * NS * BM * ROWB);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int fr = lane & 15, fg = lane >> 4;
    const int64_t M = (int64_t)g.N * g.H * g.W;
    const int tn = blockIdx.x % g.tiles_n, grp = blockIdx.x / g.tiles_n, ngrp = gridDim.x / g.tiles_n;
    const int n0 = tn * BN;

    const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, g.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wpk, 0, g.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t drsrc = __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, g.dst_bytes, 0x00020000);

    // ---- DMA plan: pass i of a tile covers rows 32 i + (tid >> 3); the lane that owns physical slot tid & 7 fetches logical slot ls
    const int r0 = tid >> 3;
    const int ls = (tid & 7) ^ ((r0 >> 1) & 7);
    bool kin[NS];
    int gdy[G > 0 ? NS : 1], gdx[G > 0 ? NS : 1];
    uint32_t gco[G > 0 ? NS : 1];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int e = s * KC + ls * VEC;                 // first K index of the slot this lane fetches in slab s
        kin[s] = e < g.Cs;
        if constexpr (G > 0) {
            const int tap = e / g.gCt;
            gco[s] = (uint32_t)(e - tap * g.gCt) * (uint32_t)sizeof(T);
            if constexpr (G == 1) { const int ky = tap / 3; gdy[s] = ky - 1; gdx[s] = tap - ky * 3 - 1; }
            else { gdy[s] = tap >> 1; gdx[s] = tap & 1; }
        }
    }
    auto load_w = [&]() {
#pragma unroll
        for (int i = 0; i < BR; ++i) {
            const int ch = n0 + wrow_channel<T>(32 * i + r0);
            const uint32_t base = ch < g.Cd ? (uint32_t)((int64_t)ch * g.wrow + ls * VEC) * (uint32_t)sizeof(T) : OOB;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const uint32_t off = (base == OOB || !kin[s]) ? OOB : base + (uint32_t)(s * ROWB);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_ptr)(sW + ((s * BN) + 32 * i + 8 * wave) * ROWB), 16, off, 0, 0, DSN_DMA_AUX);
            }
        }
    };
    DeadOps dead;
    dead.init();
    // The plain 1x1 form needs no ledger: its only dead operations are those of the last pixel tile when M % BM != 0 (that tile is
    // simply not counted as a real one: tiles_cnt), a short K (kin) and a partial channel tile -- blocks with either of those drain.
    bool kall = true;
#pragma unroll
    for (int s = 0; s < NS; ++s) kall = kall && kin[s];
    const int tiles_cnt = G == 0 ? (int)(M / BM) : g.tiles_m;
    const bool blk_drain = G == 0 && (!kall || n0 + BN > g.Cd);
    auto load_a = [&](int tm, int stage) {
        int nd = 0;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int64_t m = (int64_t)tm * BM + 32 * i + r0;
            const bool live = tm < g.tiles_m && m < M;
            if constexpr (G == 0) {
                const uint32_t base = live ? (uint32_t)(m * g.sld + ls * VEC) * (uint32_t)sizeof(T) : OOB;
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const uint32_t off = (base == OOB || !kin[s]) ? OOB : base + (uint32_t)(s * ROWB);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(srsrc, (lds_ptr)(sA + (((stage * NS + s) * BM) + 32 * i + 8 * wave) * ROWB), 16, off, 0, 0, DSN_DMA_AUX);
                }
            } else {
                const int mi = live ? (int)m : 0;
                const int x = mi % g.W, t = mi / g.W;
                const int y = t % g.H, n = t / g.H;
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const int iy = y * g.gS + gdy[s], ix = x * g.gS + gdx[s];
                    const bool ok = live && kin[s] && (unsigned)iy < (unsigned)g.gHs && (unsigned)ix < (unsigned)g.gWs;
                    const uint32_t off = ok ? (uint32_t)(((int64_t)(n * g.gHs + iy) * g.gWs + ix) * g.sld) * (uint32_t)sizeof(T) + gco[s] : OOB;
                    nd += all_out(ok);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(srsrc, (lds_ptr)(sA + (((stage * NS + s) * BM) + 32 * i + 8 * wave) * ROWB), 16, off, 0, 0, DSN_DMA_AUX);
                }
            }
        }
        if constexpr (G != 0) dead.dma(tm < g.tiles_m ? nd : 0); // (a padding group is not counted by the waits at all)
    };

    // per-lane epilogue constants: bias of the lane's channel vectors
    float bv[NV][CPV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int c = n0 + OV::ch(wn, v, fg);
#pragma unroll
        for (int k = 0; k < CPV; ++k) bv[v][k] = (bias && c + k < g.Cd) ? bias[c + k] : 0.f;
    }
    StatRegs<NV, CPV, STATS> stat;
    stat.zero();
    typedef WsXState<T, MI, NV, CPV> XS;
    typedef WsXRegs<T, MI, NV, CPV> XR;
    XS xs;
    XR xa, xb;
    int cabs[NV];
    int ccls[G == 2 ? NV : 1];                           // depth-to-space: sub-pixel class of the lane's vector v; cabs = its channel
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        cabs[v] = n0 + OV::ch(wn, v, fg);
        if constexpr (G == 2) {
            const int c = cabs[v];
            ccls[v] = c < g.Cd ? c / g.d2s_c : -1;
            cabs[v] = c < g.Cd ? c - ccls[v] * g.d2s_c : 0;
        }
    }
    if constexpr (EX) xs.init(ex, dst, g.dst_bytes, cabs, G == 2 ? g.d2s_c : g.Cd);
    // destination pixel of GEMM row m for vector v (-1: not stored)
    auto d2s_pixel = [&](int64_t m, bool live, int v) -> int64_t {
        const int mi = live ? (int)m : 0;
        const int x = mi % g.W, t = mi / g.W;
        const int y = t % g.H, n = t / g.H;
        const int oy = 2 * y + (ccls[G == 2 ? v : 0] >> 1), ox = 2 * x + (ccls[G == 2 ? v : 0] & 1);
        return (live && ccls[G == 2 ? v : 0] >= 0 && oy < g.Hout && ox < g.Wout) ? ((int64_t)n * g.Hout + oy) * g.Wout + ox : -1;
    };
    auto issue_x = [&](int tm, XR& x) {
        if constexpr (G == 2) {
            int64_t pixv[MI][NV];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int64_t m = (int64_t)tm * BM + (wm * MI + i) * 16 + fr;
#pragma unroll
                for (int v = 0; v < NV; ++v) pixv[i][v] = d2s_pixel(m, tm < g.tiles_m && m < M, v);
            }
            xs.issue_pv(x, ex, pixv, cabs, g.dld);
        } else {
            int64_t pix[MI];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int64_t m = (int64_t)tm * BM + (wm * MI + i) * 16 + fr;
                pix[i] = (tm < g.tiles_m && m < M) ? m : -1;
            }
            xs.issue(x, ex, pix, cabs, g.dld, g.Cd);
        }
    };

    // LDS byte offsets of the fragment reads (slab 0): pixel row / weight row of MFMA index fr, logical slot 4 h + fg swizzled by row
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr)smem;
    uint32_t xpre[2][MI], wpre[2][NI];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int r = (wm * MI + i) * 16 + fr;
            xpre[h][i] = (uint32_t)(r * ROWB + (((4 * h + fg) ^ ((r >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int r = (wn * NI + j) * 16 + fr;
            wpre[h][j] = lds0 + (uint32_t)(r * ROWB + (((4 * h + fg) ^ ((r >> 1) & 7)) << 4));
        }
    }

    // ring: tiles grp + j ngrp, j = 0 .. D - 2 are requested up front (tiles past the block's share: all lanes out of range, so
    // that every wave issues the same number of DMA instructions and the counted waits below stay exact)
    load_w();
#pragma unroll
    for (int j = 0; j < D - 1; ++j) load_a(grp + j * ngrp, j);
    {   // weights + the first tile (the extras just requested are register loads: they overtake DMAs and are not counted)
        if constexpr (EX) issue_x(grp, xa);
        int real = 0;
#pragma unroll
        for (int j = 1; j <= D - 2; ++j) real += (grp + j * ngrp < tiles_cnt) ? 1 : 0;
        if (blk_drain) wait_vm_drain();
        else if constexpr (G == 0) wait_prologue<D, LPT>(real);
        else wait_prologue<D, LPT>(real, dead.template among<D>(0));
    }
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int k = 0; k < CPV; ++k) asm volatile("" : "+v"(bv[v][k]));       // (bias loads retired with the wait above)

    int stage = 0;
    auto tile_step = [&](int tm, int it, XR& cur, XR& nxt) {
        // every wave has waited for its own DMA of this tile and finished reading the stage the next DMA overwrites
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if constexpr (EX) issue_x(tm + ngrp, nxt);                 // issue order per tile: extras(t + 1), DMA(t + D - 1), stores(t)
        load_a(tm + (D - 1) * ngrp, stage == 0 ? D - 1 : stage - 1);

        f32x4 acc[MI][NI];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // 2 NS k-steps (slab, 64-byte half); fragments requested PF steps ahead of their MFMAs (see DS_READ_B128)
        constexpr int NSTEP = 2 * NS, PF = 2;
        constexpr int RPS = MI + NI;
        static_assert(RPS * PF <= 15, "lgkmcnt is a 4-bit counter");
        const uint32_t a0 = lds0 + (uint32_t)(sA - smem) + (uint32_t)(stage * NS * BM * ROWB);
        u32x4 fx[PF + 1][MI], fw[PF + 1][NI];
        auto frags = [&](int k, int b) {
            const int sl = k >> 1, h = k & 1;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const uint32_t a = a0 + xpre[h][i] + (uint32_t)(sl * BM * ROWB);
                DS_READ_B128(fx[b][i], a);
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const uint32_t a = wpre[h][j] + (uint32_t)(sl * BN * ROWB);
                DS_READ_B128(fw[b][j], a);
            }
        };
#pragma unroll
        for (int k = 0; k < PF && k < NSTEP; ++k) frags(k, k);
#pragma unroll
        for (int k = 0; k < NSTEP; ++k) {
            if (k + PF < NSTEP) {
                frags(k + PF, (k + PF) % (PF + 1));
                wait_lgkm<RPS * PF>();
            } else if (k + 1 < NSTEP) {
                wait_lgkm<RPS>();
            } else {
                wait_lgkm<0>();
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) WMma<T>::run(acc[i][j], fw[k % (PF + 1)][j], fx[k % (PF + 1)][i]);
        }
        // ---- epilogue straight from the accumulators: lane (fg, fr) owns pixel fr of each 16-pixel tile, channels fg * CPV ..
        int sd = 0;                                                 // dead stores of this tile (see DeadOps)
        if constexpr (G == 1) {
            if ((int64_t)(tm + 1) * BM > M || n0 + BN > g.Cd) {
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int v = 0; v < NV; ++v)
                        sd += all_out((int64_t)tm * BM + (wm * MI + i) * 16 + fr < M && n0 + OV::ch(wn, v, fg) < g.Cd);
            }
        }
        if constexpr (EX) {
            // extras(t) were requested one tile ago; younger in the queue: DMA(t + 1), stores(t - 1), extras(t + 1), DMA(t + 2).
            // Rounds 2-3 waited with the full count of those (2 LPT + ST + XL).  That is UNSOUND on this part (round 4,
            // tools/exp/oob_order.hip, profiles/r04g_vmcnt_order_probe.txt): vmcnt counts completions, and (a) an LDS-DMA whose lanes
            // are ALL out of range retires immediately, (b) register loads and LDS-DMAs are not ordered with respect to each other
            // -- only register loads among themselves return in issue order.  On blocks with 1-4 tiles DMA(t + 1) / DMA(t + 2)
            // are the all-out-of-range padding DMAs, the counter fell below the count with extras(t) still in flight and the
            // BatchNorm-backward sums were formed from stale registers (2-20 % off on the stride-2 gather form, once in ~700 runs
            // on the 1x1 form).  Sound: allow only the YOUNGER REGISTER LOADS, extras(t + 1) -- then extras(t), older loads of the
            // same kind, are back.
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XS::XL) : "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int64_t m = (int64_t)tm * BM + (wm * MI + i) * 16 + fr;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int c = n0 + OV::ch(wn, v, fg);
                float o[CPV];
                OV::get(acc[i], v, o);
                bool ok = m < M && c < g.Cd;
                uint32_t off;
                if constexpr (G == 2) {
                    const int64_t dp = d2s_pixel(m, ok, v);
                    ok = dp >= 0;
                    off = ok ? (uint32_t)(dp * g.dld + cabs[v]) * (uint32_t)sizeof(T) : OOB;
                } else {
                    off = ok ? (uint32_t)(m * g.dld + c) * (uint32_t)sizeof(T) : OOB;
                }
                if constexpr (G == 2) sd += all_out(ok);
                if constexpr (EX) {
                    __builtin_amdgcn_raw_buffer_store_b128(xs.apply(cur, i, v, o, ok), drsrc, off, 0, 0);
                } else {
                    stat.add(v, o, ok);
#pragma unroll
                    for (int k = 0; k < CPV; ++k) o[k] += bv[v][k];
                    if (g.act == DSN_ACT_SILU) {               // (one uniform branch per vector, not per element)
#pragma unroll
                        for (int k = 0; k < CPV; ++k) o[k] *= sigmoidf_(o[k]);
                    } else if (g.act == DSN_ACT_SIGMOID) {
#pragma unroll
                        for (int k = 0; k < CPV; ++k) o[k] = sigmoidf_(o[k]);
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(pack_out<T, CPV>(o), drsrc, off, 0, 0);
                }
            }
        }
        if constexpr (EX) {
            // DMA(t + 1) is older than stores(t - 1), extras(t + 1), DMA(t + 2), stores(t)  (first tile: extras(t), extras(t + 1),
            // DMA(t + 2), stores(t)).  Of those only the STORES and a REAL DMA(t + 2) may be counted as still in flight (round 4,
            // tools/exp/oob_order.hip): register loads overtake an older LDS-DMA, and the padding DMA of a tile past the block's
            // share retires at once -- counting them let the wait go with DMA(t + 1) itself outstanding.  (Extras still in flight
            // now hold the wait a little longer: they were requested a whole tile ago.)
            const bool dma2 = tm + 2 * ngrp < tiles_cnt;
            if constexpr (G != 0) dead.st(sd);
            const int dd = G == 0 ? (blk_drain ? 1 : 0) : dead.template among<3>(it == 0 ? 1 : 2);
            if (dd) wait_vm_drain();
            else if (it == 0) { if (dma2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT + ST) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ST) : "memory"); }
            else { if (dma2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * ST + LPT) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * ST) : "memory"); }
        } else {
            // the next tile's DMA is older than (D - 2) younger tiles and the store groups issued since: min(it + 1, D - 1) of them
            int real = 0;
#pragma unroll
            for (int k = 2; k <= D - 1; ++k) real += (tm + k * ngrp < tiles_cnt) ? 1 : 0;
            const int nst = it + 1 < D - 1 ? it + 1 : D - 1;
            if constexpr (G == 0) {
                if (blk_drain) wait_vm_drain();
                else wait_ring<D, LPT, ST>(real, nst);
            } else {
                dead.st(sd);
                wait_ring<D, LPT, ST>(real, nst, dead.template among<D>(nst));
            }
        }
        stage = stage + 1 == D ? 0 : stage + 1;
    };
    {
        int tm = grp, it = 0;
        while (tm < g.tiles_m) {                 // (two tiles per trip: the extras registers ping-pong with static indices)
            tile_step(tm, it, xa, xb);
            tm += ngrp; ++it;
            if (tm >= g.tiles_m) break;
            tile_step(tm, it, xb, xa);
            tm += ngrp; ++it;
        }
    }

    if constexpr (EX) {
        // BatchNorm backward sums: fold the 16 pixel lanes of a channel group, then the WGM waves of a column, then one atomic pair
        if (ex.nseg > 0) {
#pragma unroll
            for (int v = 0; v < NV; ++v)
#pragma unroll
                for (int k = 0; k < CPV; ++k) {
#pragma unroll
                    for (int o = 1; o < 16; o <<= 1) {
                        xs.q0[v][k] += __shfl_xor(xs.q0[v][k], o);
                        xs.q1[v][k] += __shfl_xor(xs.q1[v][k], o);
                    }
                }
            __syncthreads();
            if (fr == 0) {
#pragma unroll
                for (int v = 0; v < NV; ++v)
#pragma unroll
                    for (int k = 0; k < CPV; ++k) {
                        const int cl = OV::ch(wn, v, fg) + k;
                        sRed[(wm * BN + cl) * 2] = xs.q0[v][k];
                        sRed[(wm * BN + cl) * 2 + 1] = xs.q1[v][k];
                    }
            }
            __syncthreads();
            if (tid < BN && n0 + tid < g.Cd) {
                float t0 = 0.f, t1 = 0.f;
#pragma unroll
                for (int w = 0; w < WGM; ++w) {
                    t0 += sRed[(w * BN + tid) * 2];
                    t1 += sRed[(w * BN + tid) * 2 + 1];
                }
                const int ch = G == 2 ? (n0 + tid) % g.d2s_c : n0 + tid;      // (depth-to-space: the four sub-pixel columns of a channel)
                for (int sg = 0; sg < ex.nseg; ++sg)
                    if (ch >= ex.seg[sg].c0 && ch < ex.seg[sg].c1)
                        bn_acc_add(BnAcc{ex.seg[sg].acc, ex.seg[sg].acc_c, 0.0}, blockIdx.x, ex.seg[sg].ch0 + ch - ex.seg[sg].c0, t0, t1);
            }
        }
    }

    if constexpr (STATS) {
        // lanes of one fg group hold different pixels of the same channels: fold the 16 fr lanes, then the WGM waves of a column
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int k = 0; k < CPV; ++k) {
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
                    stat.s[v][k] += __shfl_xor(stat.s[v][k], o);
                    stat.ss[v][k] += __shfl_xor(stat.ss[v][k], o);
                }
            }
        __syncthreads();
        if (fr == 0) {
#pragma unroll
            for (int v = 0; v < NV; ++v)
#pragma unroll
                for (int k = 0; k < CPV; ++k) {
                    const int cl = OV::ch(wn, v, fg) + k;
                    sRed[(wm * BN + cl) * 2] = stat.s[v][k];
                    sRed[(wm * BN + cl) * 2 + 1] = stat.ss[v][k];
                }
        }
        __syncthreads();
        if (tid < BN && n0 + tid < g.Cd) {
            float t0 = 0.f, t1 = 0.f;
#pragma unroll
            for (int w = 0; w < WGM; ++w) {
                t0 += sRed[(w * BN + tid) * 2];
                t1 += sRed[(w * BN + tid) * 2 + 1];
            }
            bn_acc_add(fin, blockIdx.x, n0 + tid, t0, t1);
        }
    }
}

struct WsPlan {
    int grid;
    size_t lds;
};

// grid of persistent blocks: as many as fit the chip at this LDS footprint, every block the same number of tiles where possible
static WsPlan ws_plan(int tiles_m, int tiles_n, size_t lds, int max_bpc) {
    int bpc = (int)((160 * 1024) / lds);
    if (bpc < 1) bpc = 1;
    if (bpc > max_bpc) bpc = max_bpc;
    int groups = (256 * bpc) / tiles_n;
    if (groups < 1) groups = 1;
    if (groups > tiles_m) groups = tiles_m;
    const int per = (tiles_m + groups - 1) / groups;
    groups = (tiles_m + per - 1) / per;
    return WsPlan{groups * tiles_n, lds};
}

// MODE: 0 = forward epilogue (bias, activation), 1 = forward + BatchNorm sums, 2 = data gradient with extras (WsX)
template <typename T, int MI, int NI, int WGM, int WGN, int NS, int MODE>
int launch_1x1_ws(const dsn_tensor* s, const void* w, const float* bias, const dsn_tensor* d, WGeom g, const BnAcc& fin,
                  int is_dgrad, hipStream_t st, const WsX& ex) {
    constexpr int BM = WGM * MI * 16, BN = WGN * NI * 16;
    constexpr bool STATS = MODE == 1, EX = MODE == 2;
    // pixel-tile stages: as deep as keeps two blocks per CU (weights + ring <= ~78 KB), 2 .. 4; the extras variant runs 3
    constexpr int TILEB = NS * BM * ROWB, WB1 = NS * BN * ROWB;
    constexpr int DFIT = (78 * 1024 - WB1) / TILEB;
    constexpr int D = EX ? 3 : (DFIT >= 4 ? 4 : (DFIT >= 3 ? 3 : 2));
    const int64_t M = (int64_t)g.N * g.H * g.W;
    g.tiles_m = (int)((M + BM - 1) / BM);
    g.tiles_n = (g.Cd + BN - 1) / BN;
    const size_t lds = (size_t)NS * (BN + D * BM) * ROWB + ((STATS || EX) ? (size_t)WGM * BN * 2 * 4 : 0);
    static const int max_bpc = [] { const char* e = getenv("DSN_WS_BPC"); return e ? atoi(e) : 4; }();
    const WsPlan pl = ws_plan(g.tiles_m, g.tiles_n, lds, max_bpc);
    // (the extras variant pins D = 3: with the 128 x 32 tile and four slabs that is 214 KB -- not this kernel's launch; the caller
    // falls through to the one-trip / implicit-GEMM kernels)
    if (lds > 160 * 1024) return 1;
    auto kern = conv1x1_ws_kernel<T, MI, NI, WGM, WGN, NS, STATS, D, EX>;
    DSN_LDS_ATTR(kern, 160 * 1024);
    double xch = (ex.res ? 1.0 : 0.0) + (ex.accumulate ? 1.0 : 0.0);
    for (int i = 0; i < ex.nseg; ++i) xch += (double)(ex.seg[i].c1 - ex.seg[i].c0) / g.Cd;
    const double elems = (double)M * (g.Cs + (double)g.Cd * (1.0 + xch)) + (double)g.Cs * g.Cd;
    const ProfConv pc("conv1x1_ws_kernel", sizeof(T) == 2, BM, BN, is_dgrad != 0, 1, 1, 1, g.Cs, g.Cd, g.N, g.H, g.W);
    ProfScope prof(pc.label, pc.layer, 2.0 * M * g.Cd * g.Cs, elems * sizeof(T), st);
    hipLaunchKernelGGL(kern, dim3(pl.grid), dim3(256), pl.lds, st, (const T*)s->ptr, (const T*)w, bias, (T*)d->ptr, fin, g, ex);
    DSN_LAUNCH_CHECK("conv1x1 (weights-stationary)");
    return DSN_OK;
}

template <typename T, int MI, int NI, int WGM, int WGN, int MODE>
int launch_1x1_ws_ns(const dsn_tensor* s, const void* w, const float* bias, const dsn_tensor* d, const WGeom& g, const BnAcc& fin,
                     int is_dgrad, hipStream_t st, int ns, const WsX& ex) {
    switch (ns) {
        case 1: return launch_1x1_ws<T, MI, NI, WGM, WGN, 1, MODE>(s, w, bias, d, g, fin, is_dgrad, st, ex);
        case 2: return launch_1x1_ws<T, MI, NI, WGM, WGN, 2, MODE>(s, w, bias, d, g, fin, is_dgrad, st, ex);
        case 3: return launch_1x1_ws<T, MI, NI, WGM, WGN, 3, MODE>(s, w, bias, d, g, fin, is_dgrad, st, ex);
        case 4: return launch_1x1_ws<T, MI, NI, WGM, WGN, 4, MODE>(s, w, bias, d, g, fin, is_dgrad, st, ex);
        default: return 1;
    }
}

template <typename T, int MODE>
int launch_1x1_ws_cfg(const dsn_tensor* s, const void* w, const float* bias, const dsn_tensor* d, const WGeom& g, const BnAcc& fin,
                      int is_dgrad, hipStream_t st, int ns, const WsX& ex) {
    if (ns > 4) {      // 5 .. 8 slabs (K <= 512 bf16): 32 x 64 tiles, 64 KB of weights + a 2-stage ring, one block per CU; no extras
        if constexpr (MODE != 2) {
            switch (ns) {
                case 5: return launch_1x1_ws<T, 1, 2, 2, 2, 5, MODE>(s, w, bias, d, g, fin, is_dgrad, st, ex);
                case 6: return launch_1x1_ws<T, 1, 2, 2, 2, 6, MODE>(s, w, bias, d, g, fin, is_dgrad, st, ex);
                case 7: return launch_1x1_ws<T, 1, 2, 2, 2, 7, MODE>(s, w, bias, d, g, fin, is_dgrad, st, ex);
                case 8: return launch_1x1_ws<T, 1, 2, 2, 2, 8, MODE>(s, w, bias, d, g, fin, is_dgrad, st, ex);
                default: return 1;
            }
        }
        return 1;
    }
    static const int force = [] { const char* e = getenv("DSN_WS_CFG"); return e ? atoi(e) : -1; }();     // tuning knob
    int cfg = force;
    if (cfg < 0 || cfg > 2) cfg = g.Cd <= 32 ? 2 : (ns >= 3 ? 1 : 0);
    switch (cfg) {
        case 1: return launch_1x1_ws_ns<T, 1, 2, 2, 2, MODE>(s, w, bias, d, g, fin, is_dgrad, st, ns, ex);     // 32 x 64
        case 2: return launch_1x1_ws_ns<T, 2, 2, 4, 1, MODE>(s, w, bias, d, g, fin, is_dgrad, st, ns, ex);     // 128 x 32
        default: return launch_1x1_ws_ns<T, 2, 2, 2, 2, MODE>(s, w, bias, d, g, fin, is_dgrad, st, ns, ex);    // 64 x 64
    }
}

// ---- gather forms (G = 1: 3x3 / stride 2 forward, G = 2: its data gradient with depth-to-space store) ---------------------------
// MODE as above.  The ring depth is what fits 160 KB next to the resident weights (the extras variant needs exactly three stages).
template <typename T, int MI, int NI, int WGM, int WGN, int NS, int MODE, int G>
int launch_gather_ws(const dsn_tensor* s, const void* w, const dsn_tensor* d, WGeom g, const BnAcc& fin, hipStream_t st, const WsX& ex,
                     const float* bias = nullptr) {
    constexpr int BM = WGM * MI * 16, BN = WGN * NI * 16;
    constexpr bool STATS = MODE == 1, EX = MODE == 2;
    constexpr int TILEB = NS * BM * ROWB, WB1 = NS * BN * ROWB, RED = 0;      // (the fold buffer reuses the weight area)
    static_assert(WGM * BN * 2 * 4 <= WB1, "fold buffer inside the weight area");
    // two blocks per CU with a 2-stage ring where that fits, else the deepest ring one block can have
    constexpr int DFIT = (160 * 1024 - WB1) / TILEB;
    constexpr int D = EX ? 3 : (WB1 + 2 * TILEB <= 80 * 1024 ? 2 : (DFIT >= 3 ? 3 : 2));
    static_assert(WB1 + D * TILEB + RED <= 160 * 1024, "weights + ring exceed the LDS of a CU");
    const int64_t M = (int64_t)g.N * g.H * g.W;
    g.tiles_m = (int)((M + BM - 1) / BM);
    g.tiles_n = (g.Cd + BN - 1) / BN;
    const size_t lds = (size_t)WB1 + (size_t)D * TILEB + RED;
    const WsPlan pl = ws_plan(g.tiles_m, g.tiles_n, lds, 4);
    auto kern = conv1x1_ws_kernel<T, MI, NI, WGM, WGN, NS, STATS, D, EX, G>;
    DSN_LDS_ATTR(kern, 160 * 1024);
    double xch = 0.0;
    for (int i = 0; i < ex.nseg; ++i) xch += (double)(ex.seg[i].c1 - ex.seg[i].c0) / (G == 2 ? g.d2s_c : g.Cd);
    const double src_el = (double)g.N * g.gHs * g.gWs * g.gCt;
    const double elems = src_el + (double)M * g.Cd * (1.0 + xch) + (double)g.Cs * g.Cd;
    // flops as the implicit-GEMM launch counts them (the 2x2 form of the data gradient: 16 / 9 of the convolution's)
    const ProfConv pc("conv1x1_ws_kernel", sizeof(T) == 2, BM, BN, G == 2, G == 1 ? 3 : 2, 2, 1, g.gCt, G == 1 ? g.Cd : g.d2s_c,
                      g.N, g.H, g.W);
    ProfScope prof(pc.label, pc.layer, 2.0 * M * g.Cd * g.Cs, elems * sizeof(T), st);
    hipLaunchKernelGGL(kern, dim3(pl.grid), dim3(256), pl.lds, st, (const T*)s->ptr, (const T*)w, bias, (T*)d->ptr, fin, g, ex);
    DSN_LAUNCH_CHECK("conv (weights-stationary, gathered K)");
    return DSN_OK;
}

// ================================================================================================================================
// 3x3 / stride 1 / dilation d (padding d).  A block owns BN output channels -- ALL nine taps of their weights stay in LDS
// ([tap][slab][BN rows][128 B]: 72 KB for 64 channels x one 64-channel slab) -- and walks a contiguous range of TH x TW output
// patches; per (patch, slab) the (TH + 2d) x (TW + 2d) halo of the slab arrives by LDS-DMA into one of two stages while the previous
// one is multiplied: nine taps = nine shifted reads of the halo, 18 k-steps without a barrier (conv3x3.hip streams the weights of
// every tap through a ring behind a barrier per tap, once per 64 pixels).  Halo swizzle as conv3x3.hip's hslot().
// ================================================================================================================================
__device__ __forceinline__ int hslot(int slot, int hx) { return (((slot >> 1) ^ ((hx >> 1) & 3)) << 1) | (slot & 1); }

template <typename T, int TH, int TW, int MI, int NI, int WGM, int WGN, int IH, int NS, bool STATS, int D, bool HP, bool EX>
__global__ __launch_bounds__(256) void conv3x3_ws_kernel(const T* __restrict__ src, const T* __restrict__ wpk,
                                                         const float* __restrict__ bias, T* __restrict__ dst, const BnAcc fin,
                                                         const WGeom g, const WsX ex) {
    static_assert(!EX || (D == 3 && NS == 1 && !STATS), "extras variant (data gradient): one slab, 3-stage ring");
    static_assert(WGM * WGN == 4, "4 waves per block");
    constexpr int BM = WGM * MI * 16, BN = WGN * NI * 16;
    static_assert(BM == TH * TW, "the pixel tile is the TH x TW patch");
    static_assert(BN % 32 == 0, "weight rows are filled 32 per DMA pass");
    constexpr int VEC = WMma<T>::VEC;
    constexpr int KC = ROWB / (int)sizeof(T);
    constexpr int BR = BN / 32;
    typedef OutVec<T, NI> OV;
    constexpr int NV = OV::NV, CPV = OV::CPV;
    constexpr int ST = MI * NV;
    constexpr int HSTAGE = IH * 32 * ROWB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sW = smem;                                  // [9][NS][BN][128 B]
    static_assert(D >= 2 && D <= 4 && (NS == 1 || NS == 2), "ring of 2 .. 4 halo stages; one or two slabs");
    unsigned char* sH = smem + 9 * NS * BN * ROWB;             // [D][IH * 32][128 B]
    float* sRed = reinterpret_cast<float*>(sH + D * HSTAGE);   // [WGM][BN][2] (STATS)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int fr = lane & 15, fg = lane >> 4;
    const int tn = blockIdx.x % g.tiles_n, grp = blockIdx.x / g.tiles_n, ngrp = gridDim.x / g.tiles_n;
    const int n0 = tn * BN;
    const int per = (g.tiles_m + ngrp - 1) / ngrp;              // contiguous patch range of this block: neighbours share halo columns
    const int t_begin = grp * per, t_end = (t_begin + per < g.tiles_m) ? t_begin + per : g.tiles_m;
    const int tiles_img = g.tiles_y * g.tiles_x;

    const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, g.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wpk, 0, g.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t drsrc = __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, g.dst_bytes, 0x00020000);

    // ---- weights: row q = 32 i + (tid >> 3) of (tap, slab); the lane that owns physical slot tid & 7 fetches logical slot lsw
    const int r0 = tid >> 3;
    const int lsw = (tid & 7) ^ ((r0 >> 1) & 7);
    auto load_w = [&]() {
#pragma unroll
        for (int i = 0; i < BR; ++i) {
            const int ch = n0 + wrow_channel<T>(32 * i + r0);
            const uint32_t base = ch < g.Cd ? (uint32_t)((int64_t)ch * g.wrow + lsw * VEC) * (uint32_t)sizeof(T) : OOB;
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const bool kin = s * KC + lsw * VEC < g.Cs;
                    const uint32_t off = (base == OOB || !kin) ? OOB : base + (uint32_t)((t * g.Cs + s * KC) * (int)sizeof(T));
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_ptr)(sW + (((t * NS + s) * BN) + 32 * i + 8 * wave) * ROWB), 16, off, 0, 0, DSN_DMA_AUX);
                }
        }
    };
    // ---- halo: instruction j of this wave covers halo pixels 32 j + 8 wave .. + 8; lane -> pixel p, physical slot lane & 7
    // chunk c of this block = (patch t_begin + c / NS, slab c % NS); chunks past the block's range: every lane out of range
    DeadOps dead;
    dead.init();
    auto load_halo = [&](int c, int stage) {
        const int tile = t_begin + c / NS, slab = c % NS;
        const bool live = tile < t_end;
        const int n = tile / tiles_img, trem = tile - n * tiles_img;
        const int y0 = (trem / g.tiles_x) * TH, x0 = (trem % g.tiles_x) * TW;
        int nd = 0;
        static_assert(IH <= 8, "halo offsets are kept in eight registers");
        uint32_t off[8];           // (an array bound naming IH here makes hipcc drop the kernel's host stub)
#pragma unroll
        for (int j = 0; j < IH; ++j) {
            const int p = 32 * j + 8 * wave + (lane >> 3);
            const int hy = p / g.HW, hx = p - hy * g.HW;
            const int gy = y0 - g.d + hy, gx = x0 - g.d + hx;
            const int ls = hslot(lane & 7, hx);
            const bool ok = live && p < g.NP && (unsigned)gy < (unsigned)g.H && (unsigned)gx < (unsigned)g.W && slab * KC + ls * VEC < g.Cs;
            off[j] = ok ? (uint32_t)((((int64_t)n * g.H + gy) * g.W + gx) * g.sld + slab * KC + ls * VEC) * (uint32_t)sizeof(T) : OOB;
        }
        if (live && (y0 < g.d || x0 < g.d || y0 + TH + g.d > g.H || x0 + TW + g.d > g.W)) {       // a patch at the image border
#pragma unroll
            for (int j = 0; j < IH; ++j)
                nd += (32 * j + 8 * wave < g.NP) ? all_out(off[j] != OOB) : 0;     // (groups wholly past the stage's pixels: `pad` below)
        }
#pragma unroll
        for (int j = 0; j < IH; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srsrc, (lds_ptr)(sH + stage * HSTAGE + (32 * j + 8 * wave) * ROWB), 16, off[j], 0, 0, DSN_DMA_AUX);
        dead.dma(nd);                                             // (0 for a padding group: the waits do not count those at all)
    };
    int pad = 0;                                                  // this wave's DMAs that are dead in every group
#pragma unroll
    for (int j = 0; j < IH; ++j) pad += (32 * j + 8 * wave < g.NP) ? 0 : 1;

    float bv[NV][CPV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int c = n0 + OV::ch(wn, v, fg);
#pragma unroll
        for (int k = 0; k < CPV; ++k) bv[v][k] = (bias && c + k < g.Cd) ? bias[c + k] : 0.f;
    }
    StatRegs<NV, CPV, STATS> stat;
    stat.zero();
    typedef WsXState<T, MI, NV, CPV> XS;
    typedef WsXRegs<T, MI, NV, CPV> XR;
    XS xs;
    XR xa, xb;
    int cabs[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) cabs[v] = n0 + OV::ch(wn, v, fg);
    if constexpr (EX) xs.init(ex, dst, g.dst_bytes, cabs, g.Cd);
    auto issue_x = [&](int tile, XR& x) {
        int64_t pix[MI];
        const int n = tile / tiles_img, trem = tile - n * tiles_img;
        const int y0 = (trem / g.tiles_x) * TH, x0 = (trem % g.tiles_x) * TW;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int r = (wm * MI + i) * 16 + fr;
            const int y = y0 + r / TW, xx = x0 + r % TW;
            pix[i] = (tile < t_end && y < g.H && xx < g.W) ? ((int64_t)n * g.H + y) * g.W + xx : -1;
        }
        xs.issue(x, ex, pix, cabs, g.dld, g.Cd);
    };

    // fragment addressing: MFMA column fr of pixel tile i is patch pixel r = (wm MI + i) 16 + fr = (r / TW, r % TW)
    int hp0[MI], hx0[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int r = (wm * MI + i) * 16 + fr;
        hx0[i] = r % TW;
        hp0[i] = (r / TW) * g.HW + hx0[i];
    }

    // LDS byte offsets of the fragment reads: xpre[column tap rx][half h][i] = halo pixel (row of r, column of r + rx d), logical
    // slot 4 h + fg through hslot(); wpre[h][j] = weight row of MFMA row fr in tile j (tap 0, slab 0), slot swizzled by row
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr)smem;
    uint32_t xpre[3][2][MI], wpre[2][NI];
#pragma unroll
    for (int rx = 0; rx < 3; ++rx)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < MI; ++i)
                xpre[rx][h][i] = (uint32_t)((hp0[i] + rx * g.d) * ROWB + (hslot(4 * h + fg, hx0[i] + rx * g.d) << 4));
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int r = (wn * NI + j) * 16 + fr;
            wpre[h][j] = lds0 + (uint32_t)(r * ROWB + (((4 * h + fg) ^ ((r >> 1) & 7)) << 4));
        }

    load_w();
#pragma unroll
    for (int j = 0; j < D - 1; ++j) load_halo(j, j);
    {   // weights + the first chunk (chunk j belongs to patch t_begin + j / NS; register-load extras are not counted)
        if constexpr (EX) issue_x(t_begin, xa);
        int real = 0;
#pragma unroll
        for (int j = 1; j <= D - 2; ++j) real += (t_begin + j / NS < t_end) ? 1 : 0;
        wait_prologue<D, IH>(real, dead.template among<D>(0), pad);
    }
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int k = 0; k < CPV; ++k) asm volatile("" : "+v"(bv[v][k]));

    int stage = 0, c = 0;                    // c: chunk counter of this block
    auto patch_step = [&](int tile, XR& cur, XR& nxt) {
        f32x4 acc[MI][NI];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if constexpr (EX) issue_x(tile + 1, nxt);                   // issue order per patch: extras(t + 1), halo(t + D - 1), stores(t)
            load_halo(c + D - 1, stage == 0 ? D - 1 : stage - 1);      // into the stage chunk c - 1 has just released
            if constexpr (HP) {
            // 18 k-steps (tap, 64-byte half); the fragments of step k + PF are requested before the MFMAs of step k
            constexpr int NSTEP = 18, PF = 2;
            const uint32_t hb = lds0 + (uint32_t)(sH - smem) + (uint32_t)(stage * HSTAGE);
            u32x4 fx[PF + 1][MI], fw[PF + 1][NI];
            auto frags = [&](int k, int b) {
                const int t = k >> 1, h = k & 1;
                const int ky = t / 3, kx = t - ky * 3;
                const int ry = g.flip ? 2 - ky : ky, rx = g.flip ? 2 - kx : kx;
                const uint32_t rowoff = (uint32_t)(ry * g.d * g.HW) * ROWB;           // wave-uniform
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const uint32_t a = hb + rowoff + xpre[rx][h][i];
                    DS_READ_B128(fx[b][i], a);
                }
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const uint32_t a = wpre[h][j] + (uint32_t)((t * NS + s) * BN * ROWB);
                    DS_READ_B128(fw[b][j], a);
                }
            };
            auto mma = [&](int b) {
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) WMma<T>::run(acc[i][j], fw[b][j], fx[b][i]);
            };
            constexpr int RPS = MI + NI;                         // reads per step
            static_assert(PF == 2 && RPS * PF <= 15, "two steps ahead; lgkmcnt is a 4-bit counter");
#pragma unroll
            for (int k = 0; k < PF; ++k) frags(k, k);
#pragma unroll
            for (int k = 0; k < NSTEP; ++k) {
                if (k + PF < NSTEP) {
                    frags(k + PF, (k + PF) % (PF + 1));
                    wait_lgkm<RPS * PF>();
                } else if (k + 1 < NSTEP) {
                    wait_lgkm<RPS>();                               // k == NSTEP - 2: one younger step still in flight
                } else {
                    wait_lgkm<0>();
                }
                mma(k % (PF + 1));
            }
            } else {
                // compiler-scheduled fragment reads (measured faster than the hand-issued form on the 8 x 16 patch configurations)
                const unsigned char* hbp = sH + stage * HSTAGE;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int ky = t / 3, kx = t - ky * 3;
                    const int oy = (g.flip ? 2 - ky : ky) * g.d, ox = (g.flip ? 2 - kx : kx) * g.d;
                    const int toff = oy * g.HW + ox;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        u32x4 gx[MI], gw[NI];
                        const int slot = 4 * h + fg;
#pragma unroll
                        for (int i = 0; i < MI; ++i)
                            gx[i] = *reinterpret_cast<const u32x4*>(hbp + (hp0[i] + toff) * ROWB + (hslot(slot, hx0[i] + ox) << 4));
#pragma unroll
                        for (int j = 0; j < NI; ++j) {
                            const int r = (wn * NI + j) * 16 + fr;
                            gw[j] = *reinterpret_cast<const u32x4*>(sW + ((t * NS + s) * BN + r) * ROWB + ((slot ^ ((r >> 1) & 7)) << 4));
                        }
#pragma unroll
                        for (int i = 0; i < MI; ++i)
#pragma unroll
                            for (int j = 0; j < NI; ++j) WMma<T>::run(acc[i][j], gw[j], gx[i]);
                    }
                }
            }
            if (s + 1 < NS) {
                // chunk c + 1 (the next slab of this patch) is older than D - 2 younger halos and the store groups issued since its
                // DMA went out at the start of chunk c - D + 2: chunks j in [c - D + 2, c] with j % NS == NS - 1 (and j >= 0)
                int nst = 0;
#pragma unroll
                for (int b = 0; b <= D - 2; ++b) nst += (c - b >= 0 && (c - b) % NS == NS - 1) ? 1 : 0;
                int real = 0;
#pragma unroll
                for (int k = 2; k <= D - 1; ++k) real += (t_begin + (c + k) / NS < t_end) ? 1 : 0;
                wait_ring<D, IH, ST>(real, nst, dead.template among<D>(nst), pad);
                ++c;
            }
            stage = stage + 1 == D ? 0 : stage + 1;
        }
        int sd = 0;                                                 // dead stores of this patch (see DeadOps)
        // ---- epilogue: lane (fg, fr) owns patch pixel (r / TW, r % TW) of each 16-pixel tile, channels fg * CPV ..
        const int n = tile / tiles_img, trem = tile - n * tiles_img;
        const int y0 = (trem / g.tiles_x) * TH, x0 = (trem % g.tiles_x) * TW;
        if (y0 + TH > g.H || x0 + TW > g.W || n0 + BN > g.Cd) {
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int r = (wm * MI + i) * 16 + fr;
#pragma unroll
                for (int v = 0; v < NV; ++v) sd += all_out(y0 + r / TW < g.H && x0 + r % TW < g.W && n0 + OV::ch(wn, v, fg) < g.Cd);
            }
        }
        if constexpr (EX) {     // (counts as in conv1x1_ws_kernel; c == patches done by this block)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XS::XL) : "memory");      // only the younger REGISTER loads: see conv1x1_ws_kernel
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int r = (wm * MI + i) * 16 + fr;
            const int y = y0 + r / TW, x = x0 + r % TW;
            const int64_t m = ((int64_t)n * g.H + y) * g.W + x;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int cc = n0 + OV::ch(wn, v, fg);
                float o[CPV];
                OV::get(acc[i], v, o);
                const bool ok = y < g.H && x < g.W && cc < g.Cd;
                const uint32_t off = ok ? (uint32_t)(m * g.dld + cc) * (uint32_t)sizeof(T) : OOB;
                if constexpr (EX) {
                    __builtin_amdgcn_raw_buffer_store_b128(xs.apply(cur, i, v, o, ok), drsrc, off, 0, 0);
                } else {
                    stat.add(v, o, ok);
#pragma unroll
                    for (int k = 0; k < CPV; ++k) o[k] += bv[v][k];
                    if (g.act == DSN_ACT_SILU) {
#pragma unroll
                        for (int k = 0; k < CPV; ++k) o[k] *= sigmoidf_(o[k]);
                    } else if (g.act == DSN_ACT_SIGMOID) {
#pragma unroll
                        for (int k = 0; k < CPV; ++k) o[k] = sigmoidf_(o[k]);
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(pack_out<T, CPV>(o), drsrc, off, 0, 0);
                }
            }
        }
        if constexpr (EX) {
            // (as in conv1x1_ws_kernel: only the stores and a REAL halo DMA of patch + 2 may be counted as in flight -- register loads
            //  overtake an older LDS-DMA, padding DMAs retire at once)
            const bool dma2 = t_begin + (c + 2) / NS < t_end;          // the halo DMA issued in this chunk (chunk c + 2) is a real one
            dead.st(sd);
            const int dd = dead.template among<3>(c == 0 ? 1 : 2) + (dma2 ? pad : 0);
            if (dd) wait_vm_drain();
            else if (c == 0) { if (dma2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IH + ST) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ST) : "memory"); }
            else { if (dma2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * ST + IH) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * ST) : "memory"); }
            ++c;
        } else {   // as above, after this patch's ST stores (chunk c is a last-slab chunk)
            int nst = 0;
#pragma unroll
            for (int b = 0; b <= D - 2; ++b) nst += (c - b >= 0 && (c - b) % NS == NS - 1) ? 1 : 0;
            int real = 0;
#pragma unroll
            for (int k = 2; k <= D - 1; ++k) real += (t_begin + (c + k) / NS < t_end) ? 1 : 0;
            dead.st(sd);
            wait_ring<D, IH, ST>(real, nst, dead.template among<D>(nst), pad);
            ++c;
        }
    };
    {
        int tile = t_begin;
        while (tile < t_end) {
            patch_step(tile, xa, xb);
            if (++tile >= t_end) break;
            patch_step(tile, xb, xa);
            ++tile;
        }
    }

    if constexpr (EX) {
        if (ex.nseg > 0) {
#pragma unroll
            for (int v = 0; v < NV; ++v)
#pragma unroll
                for (int k = 0; k < CPV; ++k) {
#pragma unroll
                    for (int o = 1; o < 16; o <<= 1) {
                        xs.q0[v][k] += __shfl_xor(xs.q0[v][k], o);
                        xs.q1[v][k] += __shfl_xor(xs.q1[v][k], o);
                    }
                }
            __syncthreads();
            if (fr == 0) {
#pragma unroll
                for (int v = 0; v < NV; ++v)
#pragma unroll
                    for (int k = 0; k < CPV; ++k) {
                        const int cl = OV::ch(wn, v, fg) + k;
                        sRed[(wm * BN + cl) * 2] = xs.q0[v][k];
                        sRed[(wm * BN + cl) * 2 + 1] = xs.q1[v][k];
                    }
            }
            __syncthreads();
            if (tid < BN && n0 + tid < g.Cd) {
                float t0 = 0.f, t1 = 0.f;
#pragma unroll
                for (int w = 0; w < WGM; ++w) {
                    t0 += sRed[(w * BN + tid) * 2];
                    t1 += sRed[(w * BN + tid) * 2 + 1];
                }
                const int ch = n0 + tid;
                for (int sg = 0; sg < ex.nseg; ++sg)
                    if (ch >= ex.seg[sg].c0 && ch < ex.seg[sg].c1)
                        bn_acc_add(BnAcc{ex.seg[sg].acc, ex.seg[sg].acc_c, 0.0}, blockIdx.x, ex.seg[sg].ch0 + ch - ex.seg[sg].c0, t0, t1);
            }
        }
    }

    if constexpr (STATS) {
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int k = 0; k < CPV; ++k) {
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
                    stat.s[v][k] += __shfl_xor(stat.s[v][k], o);
                    stat.ss[v][k] += __shfl_xor(stat.ss[v][k], o);
                }
            }
        __syncthreads();
        if (fr == 0) {
#pragma unroll
            for (int v = 0; v < NV; ++v)
#pragma unroll
                for (int k = 0; k < CPV; ++k) {
                    const int cl = OV::ch(wn, v, fg) + k;
                    sRed[(wm * BN + cl) * 2] = stat.s[v][k];
                    sRed[(wm * BN + cl) * 2 + 1] = stat.ss[v][k];
                }
        }
        __syncthreads();
        if (tid < BN && n0 + tid < g.Cd) {
            float t0 = 0.f, t1 = 0.f;
#pragma unroll
            for (int w = 0; w < WGM; ++w) {
                t0 += sRed[(w * BN + tid) * 2];
                t1 += sRed[(w * BN + tid) * 2 + 1];
            }
            bn_acc_add(fin, blockIdx.x, n0 + tid, t0, t1);
        }
    }
}

template <typename T, int TH, int TW, int MI, int NI, int WGM, int WGN, int IH, int NS, int MODE>
int launch_3x3_ws(const dsn_tensor* s, const void* w, const float* bias, const dsn_tensor* d, WGeom g, const BnAcc& fin,
                  hipStream_t st, const WsX& ex) {
    constexpr int BM = WGM * MI * 16, BN = WGN * NI * 16;
    constexpr bool STATS = MODE == 1, EX = MODE == 2;
    if constexpr (EX && NS != 1) {
        return 1;
    } else {
        // halo stages: as many (<= 4) as fit next to the weights in 160 KB; the extras variant runs 3
        constexpr int WB = 9 * NS * BN * ROWB, HS = IH * 32 * ROWB;
        constexpr int RED = (STATS || EX) ? WGM * BN * 2 * 4 : 0, CAP = 160 * 1024;
        constexpr int D = EX ? 3 : ((WB + 4 * HS + RED <= CAP) ? 4 : (WB + 3 * HS + RED <= CAP) ? 3 : 2);
        g.tiles_y = (g.H + TH - 1) / TH;
        g.tiles_x = (g.W + TW - 1) / TW;
        g.tiles_m = g.N * g.tiles_y * g.tiles_x;
        g.tiles_n = (g.Cd + BN - 1) / BN;
        g.HW = TW + 2 * g.d;
        g.NP = (TH + 2 * g.d) * g.HW;
        if (g.NP > IH * 32) return 1;
        const size_t lds = (size_t)WB + (size_t)D * HS + RED;
        if (lds > 160 * 1024) return 1;
        const WsPlan pl = ws_plan(g.tiles_m, g.tiles_n, lds, 2);
        constexpr bool HP = TW == 8;          // hand-issued fragment reads: measured +10 % on 8 x 8 patches, -30 % on 8 x 16 ones
        auto kern = conv3x3_ws_kernel<T, TH, TW, MI, NI, WGM, WGN, IH, NS, STATS, D, HP, EX>;
        DSN_LDS_ATTR(kern, 160 * 1024);
        double xch = (ex.res ? 1.0 : 0.0) + (ex.accumulate ? 1.0 : 0.0);
        for (int i = 0; i < ex.nseg; ++i) xch += (double)(ex.seg[i].c1 - ex.seg[i].c0) / g.Cd;
        const double elems = (double)g.N * g.H * g.W * (g.Cs + (double)g.Cd * (1.0 + xch)) + 9.0 * g.Cs * g.Cd;
        const ProfConv pc("conv3x3_ws_kernel", sizeof(T) == 2, BM, BN, g.flip != 0, 3, 1, g.d, g.Cs, g.Cd, g.N, g.H, g.W);
        ProfScope prof(pc.label, pc.layer, 2.0 * g.N * g.H * g.W * g.Cd * 9.0 * g.Cs, elems * sizeof(T), st);
        hipLaunchKernelGGL(kern, dim3(pl.grid), dim3(256), pl.lds, st, (const T*)s->ptr, (const T*)w, bias, (T*)d->ptr, fin, g, ex);
        DSN_LAUNCH_CHECK("conv3x3 (weights-stationary)");
        return DSN_OK;
    }
}

// shape dispatch: dilation picks the halo size (IH), channels the (slabs, BN) pair whose weights fit LDS
template <typename T, int MODE>
int launch_3x3_ws_cfg(const dsn_tensor* s, const void* w, const float* bias, const dsn_tensor* d, const WGeom& g, const BnAcc& fin,
                      hipStream_t st, int ns, const WsX& ex) {
    // 8 x 8 patches: (8 + 2d)^2 halo pixels = 100 / 144 / 196 -> IH 4 / 5 / 7
    if (ns == 1 && g.Cd > 32) {        // 64 output channels per block, one slab: 72 KB of weights
        if (g.d == 1) return launch_3x3_ws<T, 8, 8, 2, 2, 2, 2, 4, 1, MODE>(s, w, bias, d, g, fin, st, ex);
        if (g.d == 2) return launch_3x3_ws<T, 8, 8, 2, 2, 2, 2, 5, 1, MODE>(s, w, bias, d, g, fin, st, ex);
        return launch_3x3_ws<T, 8, 8, 2, 2, 2, 2, 7, 1, MODE>(s, w, bias, d, g, fin, st, ex);
    }
    if (ns == 1) {                     // <= 32 output channels: 8 x 16 patches x 32 channels (36 KB of weights)
        if (g.d == 1) return launch_3x3_ws<T, 8, 16, 2, 2, 4, 1, 6, 1, MODE>(s, w, bias, d, g, fin, st, ex);
        return 1;
    }
    if (ns == 2) {                     // two slabs: 32 output channels per block
        if (g.d == 1) return launch_3x3_ws<T, 8, 16, 2, 2, 4, 1, 6, 2, MODE>(s, w, bias, d, g, fin, st, ex);
        return 1;
    }
    return 1;
}


// ================================================================================================================================
// 3x3 / stride 1 / pad 1 over a THIN input: 16 bf16 channels per pixel (32 bytes) -- Focus' convolution on the space-to-depth image
// (common.py:618-627: 12 channels, zero-padded to 16; 819,200 output pixels per batch of 8, the largest map of the network).
// With whole 128-byte slabs per tap (the kernels above, igemm.hip) three quarters of every LDS row and MFMA k-step are zeros.  Here
// the three taps of one kernel ROW are one contiguous run: pixels x-1, x, x+1 of an NHWC row are 3 x 32 = 96 adjacent bytes, and the
// packed weights [Co][ky][kx][16] hold the matching 48 coefficients contiguously -- K = 3 rows x 64 (48 + 16 zero-weight) channels,
// 6 k-steps instead of 18, and the halo patch is stored as it lies in memory (no swizzle: 32-byte pixels at stride 32 are
// conflict-free for the fragment reads by themselves).  Same persistent / ring / epilogue structure as conv3x3_ws_kernel.
// ================================================================================================================================
template <int MI, int NI, int WGN, bool STATS, int D>
__global__ __launch_bounds__(256) void conv3x3_thin_ws_kernel(const bf16_t* __restrict__ src, const bf16_t* __restrict__ wpk,
                                                              const float* __restrict__ bias, bf16_t* __restrict__ dst,
                                                              const BnAcc fin, const WGeom g) {
    typedef bf16_t T;
    constexpr int WGM = 4 / WGN;
    constexpr int TW = 16, TH = WGM * MI, BN = WGN * NI * 16;
    constexpr int PXB = 32;                              // bytes per input pixel
    constexpr int HWP = TW + 2, HROWB = HWP * PXB;       // halo row: 18 pixels = 576 bytes
    constexpr int NPIECE = (TH + 2) * HROWB / 16;        // 16-byte pieces of a halo patch
    constexpr int IH = (NPIECE + 255) / 256;             // DMA instructions per wave and patch
    // a stage is what the IH DMA instructions of a patch COVER (lanes past the patch write zeros: they must stay inside the stage);
    // the over-reading last k-step of the last row lands in that zero tail
    constexpr int HSTAGE = IH * 256 * 16;
    static_assert(HSTAGE >= (TH + 2) * HROWB + 64, "zero tail behind the patch");
    typedef OutVec<T, NI> OV;
    constexpr int NV = OV::NV, CPV = OV::CPV;
    constexpr int ST = MI * NV;
    static_assert(BN % 32 == 0, "weight rows are filled 32 per DMA pass");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sW = smem;                                  // [3 ky][BN rows][128 B]: 48 coefficients + 16 zeros
    unsigned char* sH = smem + 3 * BN * ROWB;                  // [D][HSTAGE]
    float* sRed = reinterpret_cast<float*>(sH + D * HSTAGE);   // [WGM][BN][2] (STATS)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int fr = lane & 15, fg = lane >> 4;
    const int tn = blockIdx.x % g.tiles_n, grp = blockIdx.x / g.tiles_n, ngrp = gridDim.x / g.tiles_n;
    const int n0 = tn * BN;
    const int per = (g.tiles_m + ngrp - 1) / ngrp;
    const int t_begin = grp * per, t_end = (t_begin + per < g.tiles_m) ? t_begin + per : g.tiles_m;
    const int tiles_img = g.tiles_y * g.tiles_x;

    const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, g.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wpk, 0, g.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t drsrc = __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, g.dst_bytes, 0x00020000);

    // weights: row q = 32 i + (tid >> 3) of kernel row ky; physical slot tid & 7 <- logical slot lsw (slots 6, 7: zeros)
    const int r0 = tid >> 3;
    const int lsw = (tid & 7) ^ ((r0 >> 1) & 7);
    auto load_w = [&]() {
#pragma unroll
        for (int i = 0; i < BN / 32; ++i) {
            const int ch = n0 + wrow_channel<T>(32 * i + r0);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const uint32_t off = (ch < g.Cd && lsw < 6) ? (uint32_t)(((int64_t)ch * 9 + ky * 3) * 16 + lsw * 8) * 2u : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_ptr)(sW + ((ky * BN) + 32 * i + 8 * wave) * ROWB), 16, off, 0, 0, DSN_DMA_AUX);
            }
        }
    };
    // halo patch c of this block: rows y0 - 1 .. y0 + TH, columns x0 - 1 .. x0 + TW, stored row after row as in memory
    DeadOps dead;
    dead.init();
    auto load_halo = [&](int c, int stage) {
        const int tile = t_begin + c;
        const bool live = tile < t_end;
        const int n = tile / tiles_img, trem = tile - n * tiles_img;
        const int y0 = (trem / g.tiles_x) * TH, x0 = (trem % g.tiles_x) * TW;
        int nd = 0;
        const bool chk = y0 < 1 || x0 < 1 || y0 + TH + 1 > g.H || x0 + TW + 1 > g.W || IH * 256 > NPIECE;
#pragma unroll
        for (int j = 0; j < IH; ++j) {
            const int q = j * 256 + tid;
            const int hy = q / (HROWB / 16), piece = q - hy * (HROWB / 16);
            const int gy = y0 - 1 + hy, gx = x0 - 1 + (piece >> 1);
            const bool ok = live && q < NPIECE && (unsigned)gy < (unsigned)g.H && (unsigned)gx < (unsigned)g.W;
            const uint32_t off = ok ? (uint32_t)((((int64_t)n * g.H + gy) * g.W + gx) * g.sld + (piece & 1) * 8) * 2u : OOB;
            nd += all_out(chk, ok);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srsrc, (lds_ptr)(sH + stage * HSTAGE + (j * 256 + 64 * wave) * 16), 16, off, 0, 0, DSN_DMA_AUX);
        }
        dead.dma(live ? nd : 0);                                  // (a padding group is not counted by the waits at all)
    };

    float bv[NV][CPV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int c = n0 + OV::ch(wn, v, fg);
#pragma unroll
        for (int k = 0; k < CPV; ++k) bv[v][k] = (bias && c + k < g.Cd) ? bias[c + k] : 0.f;
    }
    StatRegs<NV, CPV, STATS> stat;
    stat.zero();

    load_w();
#pragma unroll
    for (int j = 0; j < D - 1; ++j) load_halo(j, j);
    {
        int real = 0;
#pragma unroll
        for (int j = 1; j <= D - 2; ++j) real += (t_begin + j < t_end) ? 1 : 0;
        wait_prologue<D, IH>(real, dead.template among<D>(0));
    }
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int k = 0; k < CPV; ++k) asm volatile("" : "+v"(bv[v][k]));

    int stage = 0, c = 0;
    for (int tile = t_begin; tile < t_end; ++tile, ++c) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        load_halo(c + D - 1, stage == 0 ? D - 1 : stage - 1);
        f32x4 acc[MI][NI];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const unsigned char* hb = sH + stage * HSTAGE;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                u32x4 fx[MI], fw[NI];
                const int slot = 4 * h + fg;
#pragma unroll
                for (int i = 0; i < MI; ++i)       // pixel (row wm MI + i, column fr): its run starts at halo column fr (= x - 1)
                    fx[i] = *reinterpret_cast<const u32x4*>(hb + ((wm * MI + i + ky) * HWP + fr) * PXB + slot * 16);
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const int r = (wn * NI + j) * 16 + fr;
                    fw[j] = *reinterpret_cast<const u32x4*>(sW + (ky * BN + r) * ROWB + ((slot ^ ((r >> 1) & 7)) << 4));
                }
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) WMma<T>::run(acc[i][j], fw[j], fx[i]);
            }
        }
        const int n = tile / tiles_img, trem = tile - n * tiles_img;
        const int y0 = (trem / g.tiles_x) * TH, x0 = (trem % g.tiles_x) * TW;
        int sd = 0;                                                 // dead stores of this patch (see DeadOps)
        const bool schk = y0 + TH > g.H || x0 + TW > g.W || n0 + BN > g.Cd;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int y = y0 + wm * MI + i, x = x0 + fr;
            const int64_t m = ((int64_t)n * g.H + y) * g.W + x;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int cc = n0 + OV::ch(wn, v, fg);
                float o[CPV];
                OV::get(acc[i], v, o);
                const bool ok = y < g.H && x < g.W && cc < g.Cd;
                stat.add(v, o, ok);
#pragma unroll
                for (int k = 0; k < CPV; ++k) o[k] += bv[v][k];
                if (g.act == DSN_ACT_SILU) {
#pragma unroll
                    for (int k = 0; k < CPV; ++k) o[k] *= sigmoidf_(o[k]);
                } else if (g.act == DSN_ACT_SIGMOID) {
#pragma unroll
                    for (int k = 0; k < CPV; ++k) o[k] = sigmoidf_(o[k]);
                }
                const uint32_t off = ok ? (uint32_t)(m * g.dld + cc) * 2u : OOB;
                sd += all_out(schk, ok);
                __builtin_amdgcn_raw_buffer_store_b128(pack_out<T, CPV>(o), drsrc, off, 0, 0);
            }
        }
        int real = 0;
#pragma unroll
        for (int k = 2; k <= D - 1; ++k) real += (tile + k < t_end) ? 1 : 0;
        dead.st(sd);
        const int nst = c + 1 < D - 1 ? c + 1 : D - 1;
        wait_ring<D, IH, ST>(real, nst, dead.template among<D>(nst));
        stage = stage + 1 == D ? 0 : stage + 1;
    }

    if constexpr (STATS) {
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int k = 0; k < CPV; ++k) {
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
                    stat.s[v][k] += __shfl_xor(stat.s[v][k], o);
                    stat.ss[v][k] += __shfl_xor(stat.ss[v][k], o);
                }
            }
        __syncthreads();
        if (fr == 0) {
#pragma unroll
            for (int v = 0; v < NV; ++v)
#pragma unroll
                for (int k = 0; k < CPV; ++k) {
                    const int cl = OV::ch(wn, v, fg) + k;
                    sRed[(wm * BN + cl) * 2] = stat.s[v][k];
                    sRed[(wm * BN + cl) * 2 + 1] = stat.ss[v][k];
                }
        }
        __syncthreads();
        if (tid < BN && n0 + tid < g.Cd) {
            float t0 = 0.f, t1 = 0.f;
#pragma unroll
            for (int w = 0; w < WGM; ++w) {
                t0 += sRed[(w * BN + tid) * 2];
                t1 += sRed[(w * BN + tid) * 2 + 1];
            }
            bn_acc_add(fin, blockIdx.x, n0 + tid, t0, t1);
        }
    }
}

template <bool STATS>
int launch_3x3_thin(const dsn_tensor* s, const void* w, const float* bias, const dsn_tensor* d, WGeom g, const BnAcc& fin, hipStream_t st) {
    constexpr int MI = 2, NI = 2, WGN = 1, WGM = 4, D = 4;
    constexpr int TH = WGM * MI, TW = 16, BM = TH * TW, BN = WGN * NI * 16;
    constexpr int HSTAGE = (((TH + 2) * 18 * 2 + 255) / 256) * 256 * 16;
    static_assert(BM == 128, "8 x 16 patches");
    g.tiles_y = (g.H + TH - 1) / TH;
    g.tiles_x = (g.W + TW - 1) / TW;
    g.tiles_m = g.N * g.tiles_y * g.tiles_x;
    g.tiles_n = (g.Cd + BN - 1) / BN;
    const size_t lds = (size_t)3 * BN * ROWB + (size_t)D * HSTAGE + (STATS ? (size_t)WGM * BN * 2 * 4 : 0);
    const WsPlan pl = ws_plan(g.tiles_m, g.tiles_n, lds, 4);
    auto kern = conv3x3_thin_ws_kernel<MI, NI, WGN, STATS, D>;
    const double elems = (double)g.N * g.H * g.W * (g.Cs + (double)g.Cd) + 9.0 * g.Cs * g.Cd;
    const ProfConv pc("conv3x3_thin_ws_kernel", true, BM, BN, false, 3, 1, 1, g.Cs, g.Cd, g.N, g.H, g.W);
    ProfScope prof(pc.label, pc.layer, 2.0 * g.N * g.H * g.W * g.Cd * 9.0 * g.Cs, elems * 2, st);
    hipLaunchKernelGGL(kern, dim3(pl.grid), dim3(256), pl.lds, st, (const bf16_t*)s->ptr, (const bf16_t*)w, bias, (bf16_t*)d->ptr, fin, g);
    DSN_LAUNCH_CHECK("conv3x3 (thin input, weights-stationary)");
    return DSN_OK;
}

// ---- the same for the fp32 path: 12 channels per pixel (48 bytes: Focus.fwd pads to the 16-byte vector only) ----------------------
// A kernel row's three taps are 36 contiguous floats; K per row = 3 steps of 16 floats (64 bytes per lane group of four), the last
// one reading 12 floats past the run -- the next pixel's finite values (or the zero tail of the stage) against zero weights.  The
// general kernel spends a whole 32-float slab per tap (18 k-steps, 62 % zeros, MFMA-bound at 264 us for 16 x 320 x 320 pixels).
// LDS: weights [3 ky x 3 steps][BN rows][64 B] (a fragment read covers one contiguous KB: no swizzle), halo patches as in memory.
template <int MI, int NI, int WGN, bool STATS, int D>
__global__ __launch_bounds__(256) void conv3x3_thin_f32_ws_kernel(const float* __restrict__ src, const float* __restrict__ wpk,
                                                                  const float* __restrict__ bias, float* __restrict__ dst,
                                                                  const BnAcc fin, const WGeom g) {
    typedef float T;
    constexpr int WGM = 4 / WGN;
    constexpr int TW = 16, TH = WGM * MI, BN = WGN * NI * 16;
    constexpr int PXB = 48;                              // bytes per input pixel
    constexpr int HWP = TW + 2, HROWB = HWP * PXB;       // halo row: 18 pixels = 864 bytes = 54 pieces
    constexpr int RP = HROWB / 16;
    constexpr int NPIECE = (TH + 2) * RP;
    constexpr int IH = (NPIECE + 255) / 256;
    constexpr int HSTAGE = IH * 256 * 16;                // (lanes past the patch write zeros: the over-read of the last run lands there)
    static_assert(HSTAGE >= (TH + 2) * HROWB + 64, "zero tail behind the patch");
    constexpr int WPIECE = 9 * BN * 4, IW = (WPIECE + 255) / 256;
    constexpr int WBYTES = IW * 256 * 16;
    typedef OutVec<T, NI> OV;
    constexpr int NV = OV::NV, CPV = OV::CPV;
    constexpr int ST = MI * NV;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sW = smem;                                  // [9][BN][64 B]
    unsigned char* sH = smem + WBYTES;                         // [D][HSTAGE]
    float* sRed = reinterpret_cast<float*>(sH + D * HSTAGE);   // [WGM][BN][2] (STATS)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int fr = lane & 15, fg = lane >> 4;
    const int tn = blockIdx.x % g.tiles_n, grp = blockIdx.x / g.tiles_n, ngrp = gridDim.x / g.tiles_n;
    const int n0 = tn * BN;
    const int per = (g.tiles_m + ngrp - 1) / ngrp;
    const int t_begin = grp * per, t_end = (t_begin + per < g.tiles_m) ? t_begin + per : g.tiles_m;
    const int tiles_img = g.tiles_y * g.tiles_x;

    const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, g.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wpk, 0, g.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t drsrc = __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, g.dst_bytes, 0x00020000);

    auto load_w = [&]() {
#pragma unroll
        for (int j = 0; j < IW; ++j) {
            const int q = j * 256 + tid;
            const int kys = q / (BN * 4), rem = q - kys * (BN * 4);
            const int row = rem >> 2, pc = rem & 3;
            const int ky = kys / 3, st = kys - ky * 3;
            const int pp = st * 4 + pc;                       // 16-byte piece of the 144-byte coefficient run
            const int ch = n0 + row;
            const uint32_t off = (q < WPIECE && ch < g.Cd && pp < 9) ? (uint32_t)(((int64_t)ch * 9 + ky * 3) * 12 * 4 + pp * 16) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_ptr)(sW + (j * 256 + 64 * wave) * 16), 16, off, 0, 0, DSN_DMA_AUX);
        }
    };
    DeadOps dead;
    dead.init();
    auto load_halo = [&](int c, int stage) {
        const int tile = t_begin + c;
        const bool live = tile < t_end;
        const int n = tile / tiles_img, trem = tile - n * tiles_img;
        const int y0 = (trem / g.tiles_x) * TH, x0 = (trem % g.tiles_x) * TW;
        int nd = 0;
        const bool chk = y0 < 1 || x0 < 1 || y0 + TH + 1 > g.H || x0 + TW + 1 > g.W || IH * 256 > NPIECE;
#pragma unroll
        for (int j = 0; j < IH; ++j) {
            const int q = j * 256 + tid;
            const int hy = q / RP, piece = q - hy * RP;
            const int px = piece / 3, sub = piece - px * 3;
            const int gy = y0 - 1 + hy, gx = x0 - 1 + px;
            const bool ok = live && q < NPIECE && (unsigned)gy < (unsigned)g.H && (unsigned)gx < (unsigned)g.W;
            const uint32_t off = ok ? (uint32_t)((((int64_t)n * g.H + gy) * g.W + gx) * g.sld + sub * 4) * 4u : OOB;
            nd += all_out(chk, ok);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srsrc, (lds_ptr)(sH + stage * HSTAGE + (j * 256 + 64 * wave) * 16), 16, off, 0, 0, DSN_DMA_AUX);
        }
        dead.dma(live ? nd : 0);                                  // (a padding group is not counted by the waits at all)
    };

    float bv[NV][CPV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int c = n0 + OV::ch(wn, v, fg);
#pragma unroll
        for (int k = 0; k < CPV; ++k) bv[v][k] = (bias && c + k < g.Cd) ? bias[c + k] : 0.f;
    }
    StatRegs<NV, CPV, STATS> stat;
    stat.zero();

    load_w();
#pragma unroll
    for (int j = 0; j < D - 1; ++j) load_halo(j, j);
    {
        int real = 0;
#pragma unroll
        for (int j = 1; j <= D - 2; ++j) real += (t_begin + j < t_end) ? 1 : 0;
        wait_prologue<D, IH>(real, dead.template among<D>(0));
    }
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int k = 0; k < CPV; ++k) asm volatile("" : "+v"(bv[v][k]));

    int stage = 0, c = 0;
    for (int tile = t_begin; tile < t_end; ++tile, ++c) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        load_halo(c + D - 1, stage == 0 ? D - 1 : stage - 1);
        f32x4 acc[MI][NI];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const unsigned char* hb = sH + stage * HSTAGE;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
            for (int h = 0; h < 3; ++h) {
                u32x4 fx[MI], fw[NI];
#pragma unroll
                for (int i = 0; i < MI; ++i)       // pixel (row wm MI + i, column fr): its run starts at halo column fr (= x - 1)
                    fx[i] = *reinterpret_cast<const u32x4*>(hb + ((wm * MI + i + ky) * HWP + fr) * PXB + h * 64 + fg * 16);
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const int r = (wn * NI + j) * 16 + fr;
                    fw[j] = *reinterpret_cast<const u32x4*>(sW + ((ky * 3 + h) * BN + r) * 64 + fg * 16);
                }
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) WMma<T>::run(acc[i][j], fw[j], fx[i]);
            }
        }
        const int n = tile / tiles_img, trem = tile - n * tiles_img;
        const int y0 = (trem / g.tiles_x) * TH, x0 = (trem % g.tiles_x) * TW;
        int sd = 0;                                                 // dead stores of this patch (see DeadOps)
        const bool schk = y0 + TH > g.H || x0 + TW > g.W || n0 + BN > g.Cd;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int y = y0 + wm * MI + i, x = x0 + fr;
            const int64_t m = ((int64_t)n * g.H + y) * g.W + x;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int cc = n0 + OV::ch(wn, v, fg);
                float o[CPV];
                OV::get(acc[i], v, o);
                const bool ok = y < g.H && x < g.W && cc < g.Cd;
                stat.add(v, o, ok);
#pragma unroll
                for (int k = 0; k < CPV; ++k) o[k] += bv[v][k];
                if (g.act == DSN_ACT_SILU) {
#pragma unroll
                    for (int k = 0; k < CPV; ++k) o[k] *= sigmoidf_(o[k]);
                } else if (g.act == DSN_ACT_SIGMOID) {
#pragma unroll
                    for (int k = 0; k < CPV; ++k) o[k] = sigmoidf_(o[k]);
                }
                const uint32_t off = ok ? (uint32_t)(m * g.dld + cc) * 4u : OOB;
                sd += all_out(schk, ok);
                __builtin_amdgcn_raw_buffer_store_b128(pack_out<T, CPV>(o), drsrc, off, 0, 0);
            }
        }
        int real = 0;
#pragma unroll
        for (int k = 2; k <= D - 1; ++k) real += (tile + k < t_end) ? 1 : 0;
        dead.st(sd);
        const int nst = c + 1 < D - 1 ? c + 1 : D - 1;
        wait_ring<D, IH, ST>(real, nst, dead.template among<D>(nst));
        stage = stage + 1 == D ? 0 : stage + 1;
    }

    if constexpr (STATS) {
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int k = 0; k < CPV; ++k) {
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
                    stat.s[v][k] += __shfl_xor(stat.s[v][k], o);
                    stat.ss[v][k] += __shfl_xor(stat.ss[v][k], o);
                }
            }
        __syncthreads();
        if (fr == 0) {
#pragma unroll
            for (int v = 0; v < NV; ++v)
#pragma unroll
                for (int k = 0; k < CPV; ++k) {
                    const int cl = OV::ch(wn, v, fg) + k;
                    sRed[(wm * BN + cl) * 2] = stat.s[v][k];
                    sRed[(wm * BN + cl) * 2 + 1] = stat.ss[v][k];
                }
        }
        __syncthreads();
        if (tid < BN && n0 + tid < g.Cd) {
            float t0 = 0.f, t1 = 0.f;
#pragma unroll
            for (int w = 0; w < WGM; ++w) {
                t0 += sRed[(w * BN + tid) * 2];
                t1 += sRed[(w * BN + tid) * 2 + 1];
            }
            bn_acc_add(fin, blockIdx.x, n0 + tid, t0, t1);
        }
    }
}

template <bool STATS>
int launch_3x3_thin_f32(const dsn_tensor* s, const void* w, const float* bias, const dsn_tensor* d, WGeom g, const BnAcc& fin, hipStream_t st) {
    constexpr int MI = 2, NI = 2, WGN = 1, WGM = 4, D = 4;
    constexpr int TH = WGM * MI, TW = 16, BM = TH * TW, BN = WGN * NI * 16;
    constexpr int HSTAGE = (((TH + 2) * 18 * 3 + 255) / 256) * 256 * 16;
    constexpr int WBYTES = ((9 * BN * 4 + 255) / 256) * 256 * 16;
    g.tiles_y = (g.H + TH - 1) / TH;
    g.tiles_x = (g.W + TW - 1) / TW;
    g.tiles_m = g.N * g.tiles_y * g.tiles_x;
    g.tiles_n = (g.Cd + BN - 1) / BN;
    const size_t lds = (size_t)WBYTES + (size_t)D * HSTAGE + (STATS ? (size_t)WGM * BN * 2 * 4 : 0);
    const WsPlan pl = ws_plan(g.tiles_m, g.tiles_n, lds, 4);
    auto kern = conv3x3_thin_f32_ws_kernel<MI, NI, WGN, STATS, D>;
    DSN_LDS_ATTR(kern, 150 * 1024);
    const double elems = (double)g.N * g.H * g.W * (g.Cs + (double)g.Cd) + 9.0 * g.Cs * g.Cd;
    const ProfConv pc("conv3x3_thin_f32_ws_kernel", false, BM, BN, false, 3, 1, 1, g.Cs, g.Cd, g.N, g.H, g.W);
    ProfScope prof(pc.label, pc.layer, 2.0 * g.N * g.H * g.W * g.Cd * 9.0 * g.Cs, elems * 4, st);
    hipLaunchKernelGGL(kern, dim3(pl.grid), dim3(256), pl.lds, st, (const float*)s->ptr, (const float*)w, bias, (float*)d->ptr, fin, g);
    DSN_LAUNCH_CHECK("conv3x3 (thin fp32 input, weights-stationary)");
    return DSN_OK;
}

}  // namespace

// selection mode of the two kernels (environment DSN_WS / DSN_WS3 at load time; dsn_ws_mode() at run time: tests, A/B runs)
static int g_ws_mode[2] = {getenv("DSN_WS") ? atoi(getenv("DSN_WS")) : 1, getenv("DSN_WS3") ? atoi(getenv("DSN_WS3")) : 1};
extern "C" int dsn_ws_mode(int32_t mode_1x1, int32_t mode_3x3) {
    if (mode_1x1 >= 0) g_ws_mode[0] = mode_1x1;
    if (mode_3x3 >= 0) g_ws_mode[1] = mode_3x3;
    return g_ws_mode[0] * 16 + g_ws_mode[1];
}

// extras of a data-gradient launch -> WsX (false: an operand the register-prefetch path does not take)
static bool ws_extras(WsX& ex, const dsn_tensor* r, const dsn_tensor* d, const dsn_conv_params* p, const dsn_bnred* br, int es) {
    ex = WsX{};
    const int vec = 16 / es;
    if (r) {
        if (r->ldc % vec != 0 || (uintptr_t)r->ptr % 16 != 0 || r->dtype != d->dtype) return false;
        const int64_t rb = ((npix(r) - 1) * r->ldc + r->c) * es;
        if (rb >= (1ll << 32) - 64) return false;
        ex.res = r->ptr; ex.rld = r->ldc; ex.res_bytes = (uint32_t)rb;
    }
    ex.accumulate = p->accumulate ? 1 : 0;
    if (br && br->nseg > 0) {
        if (br->nseg > 2) return false;
        ex.nseg = br->nseg;
        for (int i = 0; i < br->nseg; ++i) {
            const dsn_bnred_seg& b = br->seg[i];
            if (b.c0 % vec || b.c1 % vec || b.yld % vec || (uintptr_t)b.y % 16) return false;
            const int64_t yb = ((npix(d) - 1) * b.yld + (b.c1 - b.c0)) * es;
            if (yb >= (1ll << 32) - 64) return false;
            WsX::Seg& sg = ex.seg[i];
            sg.c0 = b.c0; sg.c1 = b.c1; sg.ch0 = b.ch0; sg.acc_c = b.acc_c; sg.act = b.act;
            sg.y = b.y; sg.yld = b.yld; sg.y_bytes = (uint32_t)yb;
            sg.scale = b.scale; sg.shift = b.shift; sg.mean = b.mean; sg.rstd = b.rstd; sg.acc = (double*)b.acc;
        }
    }
    return true;
}

// Tried by the convolution entry points of igemm.hip BEFORE the one-trip kernels of conv3x3.hip.  Returns 1 when the launch is not
// one this kernel takes (nothing launched), 0 when it ran, < 0 / hipError on failure.
int dsn_conv1x1_ws_try(const dsn_tensor* s, const void* w, const float* bias, const dsn_tensor* r, const dsn_tensor* d,
                       const dsn_conv_params* p, int is_dgrad, const BnAcc* finp, void* stream, const dsn_bnred* br) {
    const int mode = g_ws_mode[0];             // 0: never, 1: default, 2: no extras variants, 3: extras on every eligible launch
    if (!mode) return 1;
    if (p->kh != 1 || p->kw != 1 || p->stride != 1 || p->pad != 0) return 1;
    const bool extras = r || p->accumulate || (br && br->nseg > 0);
    if (extras && (mode == 2 || bias || p->act != DSN_ACT_NONE || (finp && finp->acc))) return 1;
    if (s->h != d->h || s->w != d->w || s->n != d->n || s->dtype != d->dtype) return 1;
    const int es = s->dtype == DSN_F32 ? 4 : 2, vec = 16 / es, kc = ROWB / es;
    const int ns = (s->c + kc - 1) / kc;
    // 5 .. 8 slabs (K <= 512 bf16, no extras): 64 KB of resident weights per 64 output channels, one block per CU -- the implicit
    // GEMM fetched those 64 KB once per 32 / 64 pixels (DSN_WS_LONGK=0: off; DSN_WS_LONGK_MINPX: smallest map taken)
    static const int longk = [] { const char* e = getenv("DSN_WS_LONGK"); return e ? atoi(e) : 1; }();
    static const int longk_minpx = [] { const char* e = getenv("DSN_WS_LONGK_MINPX"); return e ? atoi(e) : 2048; }();
    static const int longk_f32 = [] { const char* e = getenv("DSN_WS_LONGK_F32"); return e ? atoi(e) : 1; }();
    const int ns_max = (longk && !extras && (s->dtype == DSN_BF16 || longk_f32) && npix(d) >= longk_minpx) ? 8 : 4;
    if (s->c % vec != 0 || ns > ns_max || d->c % vec != 0 || s->ldc % vec != 0 || d->ldc % vec != 0) return 1;
    if (((uintptr_t)s->ptr | (uintptr_t)d->ptr | (uintptr_t)w) % 16 != 0) return 1;
    const int64_t sb = ((npix(s) - 1) * s->ldc + s->c) * es, wb = (int64_t)d->c * s->c * es;
    const int64_t db = ((npix(d) - 1) * d->ldc + d->c) * es;
    if (sb >= (1ll << 31) || wb >= (1ll << 31) || db >= (1ll << 32) - 64) return 1;
    if (npix(d) < 2048) return 1;              // tiny maps (pyramid pooling, FFM attention): one-trip kernels
    WGeom g{};
    g.N = s->n; g.H = s->h; g.W = s->w; g.Cs = s->c; g.Cd = d->c;
    g.act = p->act;
    g.sld = s->ldc; g.dld = d->ldc;
    g.src_bytes = (uint32_t)sb; g.w_bytes = (uint32_t)wb; g.dst_bytes = (uint32_t)db;
    g.wrow = s->c;
    BnAcc fin{};
    if (finp) fin = *finp;
    hipStream_t st = (hipStream_t)stream;
    WsX ex{};
    if (extras) {
        // bf16 only: the fp32 form (twice the accumulator and operand registers) does not fit the 256-VGPR budget next to the two
        // register sets of prefetched operands.  And only the long-K launches (>= 3 slabs: 32 x 64 tiles): measured inside the
        // training step (profiles/r03f_*), the extras variant beats the one-trip kernel there (16.6 vs 19.1, 10.0 vs 11.7 us) and
        // loses on the one- and two-slab layers (17.3 vs 12.7 us) -- DSN_WS=3 takes every eligible launch (tests do).
        if (s->dtype == DSN_F32 || (ns < 3 && mode != 3) || !ws_extras(ex, r, d, p, br, es)) return 1;
        return launch_1x1_ws_cfg<bf16_t, 2>(s, w, nullptr, d, g, fin, is_dgrad, st, ns, ex);
    }
    if (fin.acc) {
        if (bias || p->act != DSN_ACT_NONE) return 1;
        if (s->dtype == DSN_F32) return launch_1x1_ws_cfg<float, 1>(s, w, bias, d, g, fin, is_dgrad, st, ns, ex);
        return launch_1x1_ws_cfg<bf16_t, 1>(s, w, bias, d, g, fin, is_dgrad, st, ns, ex);
    }
    if (s->dtype == DSN_F32) return launch_1x1_ws_cfg<float, 0>(s, w, bias, d, g, fin, is_dgrad, st, ns, ex);
    return launch_1x1_ws_cfg<bf16_t, 0>(s, w, bias, d, g, fin, is_dgrad, st, ns, ex);
}

int dsn_conv3x3_ws_try(const dsn_tensor* s, const void* w, const float* bias, const dsn_tensor* r, const dsn_tensor* d,
                       const dsn_conv_params* p, int is_dgrad, const BnAcc* finp, void* stream, const dsn_bnred* br) {
    const int mode = g_ws_mode[1];             // as in dsn_conv1x1_ws_try
    if (!mode) return 1;
    if (p->kh != 3 || p->kw != 3 || p->stride != 1 || p->pad != p->dil || p->dil < 1 || p->dil > 3) return 1;
    const bool extras = r || p->accumulate || (br && br->nseg > 0);
    if (extras && (mode == 2 || bias || p->act != DSN_ACT_NONE || (finp && finp->acc))) return 1;
    if (s->h != d->h || s->w != d->w || s->n != d->n || s->dtype != d->dtype) return 1;
    const int es = s->dtype == DSN_F32 ? 4 : 2, vec = 16 / es, kc = ROWB / es;
    const int ns = (s->c + kc - 1) / kc;
    if (s->c % vec != 0 || ns > 2 || d->c % vec != 0 || s->ldc % vec != 0 || d->ldc % vec != 0) return 1;
    if (((uintptr_t)s->ptr | (uintptr_t)d->ptr | (uintptr_t)w) % 16 != 0) return 1;
    const int64_t sb = ((npix(s) - 1) * s->ldc + s->c) * es, wb = (int64_t)d->c * 9 * s->c * es;
    const int64_t db = ((npix(d) - 1) * d->ldc + d->c) * es;
    if (sb >= (1ll << 31) || wb >= (1ll << 31) || db >= (1ll << 32) - 64) return 1;
    if (npix(d) < 2048) return 1;
    if (s->dtype == DSN_BF16 && s->c == 16 && s->ldc == 16 && p->dil == 1 && !is_dgrad && !extras && d->c % 8 == 0) {
        // thin input (Focus): three taps of a kernel row are one contiguous 96-byte run
        WGeom t{};
        t.N = s->n; t.H = s->h; t.W = s->w; t.Cs = 16; t.Cd = d->c; t.d = 1; t.act = p->act;
        t.sld = s->ldc; t.dld = d->ldc;
        t.src_bytes = (uint32_t)sb; t.w_bytes = (uint32_t)wb; t.dst_bytes = (uint32_t)db;
        t.wrow = 9 * 16;
        BnAcc f{};
        if (finp) f = *finp;
        if (f.acc) {
            if (bias || p->act != DSN_ACT_NONE) return 1;
            return launch_3x3_thin<true>(s, w, bias, d, t, f, (hipStream_t)stream);
        }
        return launch_3x3_thin<false>(s, w, bias, d, t, f, (hipStream_t)stream);
    }
    if (s->dtype == DSN_F32 && s->c == 12 && s->ldc == 12 && p->dil == 1 && !is_dgrad && !extras) {
        // the fp32 Focus shape: 12-channel pixels, a kernel row = 36 contiguous floats
        WGeom t{};
        t.N = s->n; t.H = s->h; t.W = s->w; t.Cs = 12; t.Cd = d->c; t.d = 1; t.act = p->act;
        t.sld = s->ldc; t.dld = d->ldc;
        t.src_bytes = (uint32_t)sb; t.w_bytes = (uint32_t)wb; t.dst_bytes = (uint32_t)db;
        t.wrow = 9 * 12;
        BnAcc f{};
        if (finp) f = *finp;
        if (f.acc) {
            if (bias || p->act != DSN_ACT_NONE) return 1;
            return launch_3x3_thin_f32<true>(s, w, bias, d, t, f, (hipStream_t)stream);
        }
        return launch_3x3_thin_f32<false>(s, w, bias, d, t, f, (hipStream_t)stream);
    }
    WGeom g{};
    g.N = s->n; g.H = s->h; g.W = s->w; g.Cs = s->c; g.Cd = d->c; g.d = p->dil; g.flip = is_dgrad ? 1 : 0;
    g.act = p->act;
    g.sld = s->ldc; g.dld = d->ldc;
    g.src_bytes = (uint32_t)sb; g.w_bytes = (uint32_t)wb; g.dst_bytes = (uint32_t)db;
    g.wrow = 9 * s->c;
    BnAcc fin{};
    if (finp) fin = *finp;
    hipStream_t st = (hipStream_t)stream;
    WsX ex{};
    if (extras) {
        // (bf16, one slab: see conv1x1.)  Measured in the step: 34.4 vs 40.9 us on the 32 -> 32 @160 layer (implicit GEMM before),
        // but 21 vs 16 us against the halo-tile ring kernel on the 64-channel layers -- those stay there unless DSN_WS3=3
        if (s->dtype == DSN_F32 || ns != 1 || (d->c > 32 && mode != 3) || !ws_extras(ex, r, d, p, br, es)) return 1;
        return launch_3x3_ws_cfg<bf16_t, 2>(s, w, nullptr, d, g, fin, st, ns, ex);
    }
    if (fin.acc) {
        if (bias || p->act != DSN_ACT_NONE) return 1;
        if (s->dtype == DSN_F32) return launch_3x3_ws_cfg<float, 1>(s, w, bias, d, g, fin, st, ns, ex);
        return launch_3x3_ws_cfg<bf16_t, 1>(s, w, bias, d, g, fin, st, ns, ex);
    }
    if (s->dtype == DSN_F32) return launch_3x3_ws_cfg<float, 0>(s, w, bias, d, g, fin, st, ns, ex);
    return launch_3x3_ws_cfg<bf16_t, 0>(s, w, bias, d, g, fin, st, ns, ex);
}

// 3x3 / stride 2 / pad 1 forward through the gather form (bf16; K = 9 * Ci <= 9 slabs).  Returns 1 when the layer is not taken.
int dsn_conv3x3s2_ws_try(const dsn_tensor* s, const void* w, const float* bias, const dsn_tensor* r, const dsn_tensor* d,
                         const dsn_conv_params* p, const BnAcc* finp, void* stream) {
    static const int mode = [] { const char* e = getenv("DSN_WS_S2"); return e ? atoi(e) : 1; }();
    if (!mode || !g_ws_mode[0]) return 1;
    if (p->kh != 3 || p->kw != 3 || p->stride != 2 || p->pad != 1 || p->dil != 1 || p->accumulate || r) return 1;
    if (s->dtype != d->dtype) return 1;
    const int es = s->dtype == DSN_F32 ? 4 : 2, vec = 16 / es, kc = ROWB / es;
    if (s->n != d->n || d->h != (s->h - 1) / 2 + 1 || d->w != (s->w - 1) / 2 + 1) return 1;
    if (s->c % vec != 0 || d->c % vec != 0 || s->ldc % vec != 0 || d->ldc % vec != 0) return 1;
    if (((uintptr_t)s->ptr | (uintptr_t)d->ptr | (uintptr_t)w) % 16 != 0) return 1;
    const int K = 9 * s->c, ns = (K + kc - 1) / kc;
    const int64_t sb = ((npix(s) - 1) * s->ldc + s->c) * es, wb = (int64_t)d->c * K * es, db = ((npix(d) - 1) * d->ldc + d->c) * es;
    if (sb >= (1ll << 31) || wb >= (1ll << 31) || db >= (1ll << 32) - 64 || npix(d) < 16384) return 1;
    WGeom g{};
    g.N = d->n; g.H = d->h; g.W = d->w; g.Cs = K; g.Cd = d->c; g.act = p->act;
    g.sld = s->ldc; g.dld = d->ldc;
    g.src_bytes = (uint32_t)sb; g.w_bytes = (uint32_t)wb; g.dst_bytes = (uint32_t)db;
    g.wrow = K; g.gS = 2; g.gHs = s->h; g.gWs = s->w; g.gCt = s->c;
    BnAcc fin{};
    if (finp) fin = *finp;
    if (fin.acc && (bias || p->act != DSN_ACT_NONE)) return 1;       // (training forward: plain conv + BatchNorm sums)
    hipStream_t st = (hipStream_t)stream;
    const WsX ex{};
    if (s->dtype == DSN_F32) {
        // 32 -> 64 @ 320 -> 160 in fp32: K = 288 floats = 9 slabs, 32 x 64 tiles (config 2's second layer)
        if (ns == 9 && !fin.acc) return launch_gather_ws<float, 1, 2, 2, 2, 9, 0, 1>(s, w, d, g, fin, st, ex, bias);
        return 1;
    }
    if (ns == 5) {             // 32 -> 64 @ 320 -> 160: 32 x 64 tiles, 40 KB of weights + 2 x 20 KB: two blocks per CU
        if (fin.acc) return launch_gather_ws<bf16_t, 1, 2, 2, 2, 5, 1, 1>(s, w, d, g, fin, st, ex);
        return launch_gather_ws<bf16_t, 1, 2, 2, 2, 5, 0, 1>(s, w, d, g, fin, st, ex, bias);
    }
    if (ns == 9) {             // 64 -> 128 @ 160 -> 80: 32 x 64 tiles, 72 KB of weights per 64 output channels
        if (fin.acc) return launch_gather_ws<bf16_t, 1, 2, 2, 2, 9, 1, 1>(s, w, d, g, fin, st, ex);
        return launch_gather_ws<bf16_t, 1, 2, 2, 2, 9, 0, 1>(s, w, d, g, fin, st, ex, bias);
    }
    return 1;
}

// its data gradient (igemm.hip: 2x2 / stride-1 convolution over dy, [4 Ci][2][2][Co] weights, depth-to-space store), with the
// BatchNorm backward sums of the block(s) whose dz it completes.
int dsn_dgrad_s2_ws_try(const dsn_tensor* dy, const void* w_s2, const dsn_tensor* dx, const dsn_conv_params* p, const dsn_bnred* br,
                        void* stream) {
    static const int mode = [] { const char* e = getenv("DSN_WS_S2"); return e ? atoi(e) : 1; }();
    if (!mode || !g_ws_mode[0]) return 1;
    // (Round 4: with the BatchNorm-backward sums the COUNTED wait in front of the extras gave wrong sums on grids of 1-4 tiles per
    //  block -- 8x32->64 @160, 2x32->64 @320, 16x32->64 @128: 2-20 % off the stand-alone reduction, stores bit-identical; the kernel
    //  now drains there, see tile_step; tests/test_pp_gpu.py::test_stride2_data_gradient_2x2_form holds every stride-2 dgrad kernel
    //  to the reduction on those grids.)
    if (dy->dtype != DSN_BF16 || dx->dtype != DSN_BF16) return 1;
    if (dy->c % 8 != 0 || dx->c % 8 != 0 || dy->ldc % 8 != 0 || dx->ldc % 8 != 0) return 1;
    if (((uintptr_t)dy->ptr | (uintptr_t)dx->ptr | (uintptr_t)w_s2) % 16 != 0) return 1;
    const int K = 4 * dy->c, ns = (K + 63) / 64;
    if (K % 64 != 0) return 1;
    const int64_t sb = ((npix(dy) - 1) * dy->ldc + dy->c) * 2, wb = (int64_t)4 * dx->c * K * 2, db = ((npix(dx) - 1) * dx->ldc + dx->c) * 2;
    if (sb >= (1ll << 31) || wb >= (1ll << 31) || db >= (1ll << 32) - 64 || npix(dy) < 16384) return 1;
    WGeom g{};
    g.N = dy->n; g.H = dy->h; g.W = dy->w; g.Cs = K; g.Cd = 4 * dx->c; g.act = DSN_ACT_NONE;
    g.sld = dy->ldc; g.dld = dx->ldc;
    g.src_bytes = (uint32_t)sb; g.w_bytes = (uint32_t)wb; g.dst_bytes = (uint32_t)db;
    g.wrow = K; g.gS = 1; g.gHs = dy->h; g.gWs = dy->w; g.gCt = dy->c;
    g.d2s_c = dx->c; g.Hout = dx->h; g.Wout = dx->w;
    WsX ex{};
    if (!ws_extras(ex, nullptr, dx, p, br, 2)) return 1;
    hipStream_t st = (hipStream_t)stream;
    static const int cfg = [] { const char* e = getenv("DSN_WS_S2_CFG"); return e ? atoi(e) : 0; }();     // tuning knob
    if (ns == 4) {             // 64 -> (4 x 32) @160
        if (cfg == 1) return launch_gather_ws<bf16_t, 1, 4, 2, 2, 4, 2, 2>(dy, w_s2, dx, g, BnAcc{}, st, ex);   // 32 x 128: dy fetched once
        if (cfg == 2) return launch_gather_ws<bf16_t, 2, 2, 2, 2, 4, 2, 2>(dy, w_s2, dx, g, BnAcc{}, st, ex);   // 64 x 64
        return launch_gather_ws<bf16_t, 1, 2, 2, 2, 4, 2, 2>(dy, w_s2, dx, g, BnAcc{}, st, ex);                 // 32 x 64, two blocks per CU
    }
    if (ns == 8 && cfg != 3)   // 128 -> (4 x 64) @80: 64 KB of weights + 3 x 32 KB = the whole LDS
        return launch_gather_ws<bf16_t, 1, 2, 2, 2, 8, 2, 2>(dy, w_s2, dx, g, BnAcc{}, st, ex);
    return 1;
}
