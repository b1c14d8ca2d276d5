// Implicit-GEMM convolution on MFMA for gfx950 (CDNA4): forward and data-gradient share one gather-GEMM kernel.
//
//   C[m][n] = sum_k  A[m][k] * B[n][k],   k = (tap, c) linearised as tap*Cs + c
//     m   : destination pixel (n, y, x) of an NHWC tensor (dense in N,H,W; channel stride ldc)
//     n   : destination channel
//     A   : gathered source pixel vectors, src = (dst*a + b + tap*d) / q per axis, zero outside the map
//           forward : a = stride, b = -pad, d = dil,  q = 1       (src = input x,  dst = output y)
//           dgrad   : a = 1,      b = +pad, d = -dil, q = stride  (src = dy,       dst = dx; needs divisibility)
//     B   : packed weights [n][tap][c]   (K-contiguous per destination channel)
//
// Tiling: 256 threads = 4 waves (64-wide); block tile BM x BN = (WGM*MI*16) x (WGN*NI*16).  K advances in 128-byte chunks
// (32 fp32 / 64 bf16 consecutive k: a chunk may span several taps, so thin layers -- Focus' 12 channels, the 32-channel
// stem -- waste nothing), double-buffered in LDS with register-staged prefetch: the 16-byte global loads of chunk i+1
// (6-8 per thread) are in flight while chunk i is on the matrix cores, one barrier per chunk.
// MFMA: v_mfma_f32_16x16x4_f32 (exact fp32, config 2) / v_mfma_f32_16x16x32_bf16 (configs 3-5), fp32 accumulate.
// One 16-byte LDS fragment per lane feeds 4 fp32 MFMAs or 1 bf16 MFMA: lane l holds row (l & 15), 16-byte K-group
// (l >> 4) of a 64-byte half-chunk -- for fp32 the k order inside a 16-k group is permuted identically for A and B.
// LDS rows are 128 B = 8 slots of 16 B; slot s of row r lives at s ^ ((r >> 1) & 7): the 16 lanes of every ds_read_b128
// service group then hit 16 distinct slots of the 256-byte bank row, and an 8-lane ds_write_b128 group writes one whole row.
// Epilogue: accumulators (+bias, activation) are staged through LDS as fp32, optional BatchNorm partial sums (per-channel
// sum / sum of squares over the tile's rows, from the un-rounded fp32 values) are written for the finalize kernel, and the
// tile leaves with 16-byte stores (+ residual, += destination) -- instead of 2-byte stores per lane.
#include <stdlib.h>

#include "common.h"

namespace {

struct Geom {
    int32_t M;            // destination pixels
    int32_t Hd, Wd;       // destination map
    int32_t Hs, Ws, Cs;   // source map, source channels (K per tap)
    int32_t Cd;           // destination channels
    int32_t KH, KW, Ktot;
    int32_t a, b, d, q;
    int32_t act, accumulate;
    int64_t sld, dld, rld;
    int32_t tiles_m, tiles_n;
    int32_t is_dgrad;
    int32_t fits32;       // every source element offset fits a signed 32-bit integer
    // depth-to-space store (stride-2 dgrad as a 2x2 stride-1 conv over dy): GEMM column (cls, ci), cls = py*2 + px, of GEMM
    // row (n, y, x) lands at destination pixel (n, 2y + py, 2x + px), channel ci.  d2s_c = channels per sub-pixel (0 = off).
    int32_t d2s_c, Hout, Wout;
    int32_t longk;        // K chunks from which the branch-free main loop is used (tuning knob DSN_IGEMM_LONGK)
    uint32_t src_bytes, w_bytes;   // buffer-descriptor ranges: out-of-range lanes of a buffer_load return 0 (free zero padding)
    // stride-2 dgrad by destination-pixel parity class (py, px): only the taps that can reach a class are visited
    // (1 + 2 + 2 + 4 of the 9 taps of a 3x3 instead of 9 masked ones for every pixel).  Class c = py*2 + px.
    int32_t cls_tile0[5];      // first M-tile of each class (cls_tile0[4] = tiles_m)
    int32_t cls_h[4], cls_w[4];
    int32_t cls_ntaps[4];
    uint8_t cls_taps[4][12];   // tap index ky*KW + kx
};

template <typename T> struct Mma;
template <> struct Mma<float> {
    static constexpr int VEC = 4;   // elements per 16-byte vector
    __device__ static __forceinline__ void run(f32x4& acc, const u32x4& a, const u32x4& b) {
        // bit-cast the whole vector first: indexing the u32 vector inside __builtin_bit_cast(float, a[s]) made hipcc
        // (ROCm 7.2) feed element 0 to all four MFMAs.
        const f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0], bf[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1], bf[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[2], bf[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[3], bf[3], acc, 0, 0, 0);
    }
};
template <> struct Mma<bf16_t> {
    static constexpr int VEC = 8;
    __device__ static __forceinline__ void run(f32x4& acc, const u32x4& a, const u32x4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b),
                                                      acc, 0, 0, 0);
    }
};

constexpr int ROWB = 128;           // bytes per LDS row = one K chunk
constexpr int CPAD = 4;             // fp32 staging row = BN + 4 floats (half-waves land 16 banks apart)

// bijective XCD-aware remap: blocks that share an XCD (bid % 8 equal) get a contiguous range of tiles
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

template <int BM, int BN, int NBUF = 2> constexpr int smem_bytes() {
    constexpr int loop = NBUF * (BM + BN) * ROWB;
    constexpr int epi = BM * (BN + CPAD) * 4 + 2 * 256 * 4;
    return loop > epi ? loop : epi;
}

// LD: how a thread finds the 16 bytes it stages.  0 = per-element tap decode (channel counts that rule out vectors),
// 1 = 16-byte vectors with a per-thread incremental (tap, channel) decode (a chunk may span taps: Cs = 16 / 32),
// 2 = uniform tap: Cs is a multiple of the chunk, so the tap of a chunk is the same for the whole block -- tap stepping runs
//     on the scalar unit, border handling is one bit test against a per-row tap mask built once, and an address is
//     base + (uniform offset): ~4 VALU per load instead of ~35 (at 1.5 blocks per CU nothing hides the address arithmetic).
// GL > 0: LDS-DMA staging (buffer_load_dwordx4 ... lds) into a ring of GL LDS stages instead of register stages feeding two LDS
// buffers: the loads of chunks i+1 .. i+GL-1 are in flight while chunk i is on the matrix cores at NO register cost (the
// register-staged loop waits for the loads of the same or the next iteration: ~0.6 us per K chunk of pure load latency on
// layers that live out of L2 / MALL).  An LDS-DMA wave-instruction writes 64 x 16 B LINEARLY (8 rows x 8 slots): the XOR swizzle
// of the tile is applied on the SOURCE side -- thread (row, physical slot v) fetches logical slot v ^ swz(row).  One raw
// s_barrier per chunk, counted s_waitcnt vmcnt (never 0 inside the loop), no ordinary global load inside the loop.
template <typename T, int MI, int NI, int WGM, int WGN, int LD, bool PAR, int NST = 3, int GL = 0>
__global__ __launch_bounds__(256, (GL ? 2 : MI * NI <= 4 ? (WGM == 4 ? 3 : 4) : MI * NI <= 8 ? 3 : 2)) void igemm_kernel(const T* __restrict__ src, const T* __restrict__ wpk,
                                                    const float* __restrict__ bias, const T* __restrict__ res,
                                                    T* __restrict__ dst, float* __restrict__ stats, const BnAcc fin, const Geom g,
                                                    const BnRed br) {
    static_assert(WGM * WGN == 4, "4 waves per block");
    static_assert(GL == 0 || (LD >= 1 && !PAR && GL >= 3), "LDS-DMA staging: vector paths, >= 3 stages");
    constexpr int BM = WGM * MI * 16, BN = WGN * NI * 16;
    constexpr int VEC = Mma<T>::VEC;
    constexpr bool VECLOAD = LD >= 1, UNI = LD == 2;
    static_assert(!(UNI && PAR), "uniform-tap staging has no parity-class form");
    constexpr int KC = ROWB / (int)sizeof(T);                  // k per chunk: 32 fp32 / 64 bf16
    constexpr int AR = (BM + 31) / 32, BR = (BN + 31) / 32;    // staged rows per thread (8 threads per row)
    constexpr int NBUF = GL ? GL : 2;
    static_assert(GL == 0 || (BM % 32 == 0 && BN % 32 == 0), "LDS-DMA pieces are 8 whole rows per wave");
    __shared__ __attribute__((aligned(16))) unsigned char smem[smem_bytes<BM, BN, NBUF>()];
    unsigned char* sA = smem;
    unsigned char* sB = smem + NBUF * BM * ROWB;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int tile = xcd_remap(blockIdx.x, g.tiles_m * g.tiles_n);
    const int tm = tile / g.tiles_n, tn = tile % g.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- destination rows of this tile -----------------------------------------------------------------------------
    // plain: row r of tile tm is destination pixel m0 + r.  PAR: tile tm belongs to parity class cls and row r is the
    // r-th pixel (n, yh, xh) of that class, i.e. destination pixel (n, 2*yh + py, 2*xh + px).
    int cls = 0, cm0 = m0, cH = g.Hd, cW = g.Wd, cpy = 0, cpx = 0, cM = g.M;
    if (PAR) {
        while (cls < 3 && tm >= g.cls_tile0[cls + 1]) ++cls;
        cm0 = (tm - g.cls_tile0[cls]) * BM;
        cH = g.cls_h[cls]; cW = g.cls_w[cls];
        cpy = cls >> 1; cpx = cls & 1;
        cM = (g.M / (g.Hd * g.Wd)) * cH * cW;
    }
    auto dst_pixel = [&](int r, int& y, int& x, int& n) -> bool {     // r: row within the tile
        const int m = cm0 + r;
        if (r >= BM || m >= cM) return false;
        x = m % cW;
        const int t = m / cW;
        y = t % cH;
        n = t / cH;
        if (PAR) { y = 2 * y + cpy; x = 2 * x + cpx; }
        return true;
    };

    // ---- per-thread staging rows -----------------------------------------------------------------------------
    const int r0 = tid >> 3;        // row 0..31 (+32*i)
    // 16-byte vector within the 128-byte chunk this thread FETCHES.  Register staging: the physical LDS slot is chosen at the
    // ds_write (v ^ swz(row)).  LDS-DMA: the physical slot is the lane's (tid & 7), so the thread fetches the vector that belongs
    // there, (tid & 7) ^ swz(row) -- swz(row) = (row >> 1) & 7 is the same for rows r0, r0 + 32, ...
    const int v = GL ? ((tid & 7) ^ ((r0 >> 1) & 7)) : (tid & 7);
    int32_t py[AR], px[AR];
    int64_t nbase[AR];
    bool rowok[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        int y = 0, x = 0, n = 0;
        rowok[i] = dst_pixel(r0 + 32 * i, y, x, n);
        py[i] = y * g.a + g.b;
        px[i] = x * g.a + g.b;
        nbase[i] = (int64_t)n * g.Hs * g.Ws;
    }
    const int Kc = PAR ? g.cls_ntaps[cls] * g.Cs : g.Ktot;      // K extent of this tile
    const int nchunks = (Kc + KC - 1) / KC;
    const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, g.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wpk, 0, g.w_bytes, 0x00020000);

    // three register stages: chunks i+1, i+2, i+3 are in flight while chunk i is on the matrix cores (the loads of a
    // chunk get two full iterations to land; one stage left every iteration waiting ~0.5 us for L2)
    u32x4 rga[NST][AR], rgb[NST][BR];      // NST register stages (6 was measured: slower everywhere -- the registers cost occupancy)

    // source address of destination row i for tap (ky, kx); nullptr when the tap falls outside the map
    auto src_row = [&](int i, int ky, int kx) -> const T* {
        int sy = py[i] + ky * g.d, sx = px[i] + kx * g.d;
        bool ok = rowok[i] && sy >= 0 && sx >= 0;
        if (g.q > 1) {
            ok = ok && (sy % g.q == 0) && (sx % g.q == 0);
            sy /= g.q;
            sx /= g.q;
        }
        ok = ok && sy < g.Hs && sx < g.Ws;
        return ok ? src + (nbase[i] + (int64_t)sy * g.Ws + sx) * g.sld : nullptr;
    };

    // incremental (tap, channel) decode of this thread's 16-byte vector: chunks are visited in order, so k advances by
    // KC per call -- no integer division in the loop (the address arithmetic, not memory, was the bottleneck: ~200 VALU
    // instructions per chunk per thread before)
    int kc_c = v * VEC, kc_tap = 0, kc_ky = 0, kc_kx = 0;
    auto kc_norm = [&]() {
        while (kc_c >= g.Cs) {
            kc_c -= g.Cs;
            ++kc_tap;
            if (++kc_kx == g.KW) { kc_kx = 0; ++kc_ky; }
        }
    };
    kc_norm();
    // stride-1 gathers (every forward conv with q == 1): per-row element offset of (n, py, px), 32-bit when the tensor fits
    const bool fast = !PAR && g.q == 1 && g.fits32;
    int32_t base_off[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) base_off[i] = (int32_t)((nbase[i] + (int64_t)py[i] * g.Ws + px[i]) * g.sld);

    // uniform-tap staging state (LD == 2): chunks are requested in order, so (tap, chunk-in-tap) advance on the scalar unit
    uint32_t tapmask[AR], voffA[AR], voffB[BR];
    bool bok[BR];
    int u_cc = 0, u_tap = 0, u_ky = 0, u_kx = 0;
    uint32_t u_offA = 0, u_offB = 0;
    const int u_cpt = g.Cs / KC;                                // chunks per tap
    if (UNI) {
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            uint32_t mk = 0;
            if (rowok[i]) {
                int t = 0;
                for (int ky = 0; ky < g.KH; ++ky)
                    for (int kx = 0; kx < g.KW; ++kx, ++t)
                        if ((unsigned)(py[i] + ky * g.d) < (unsigned)g.Hs && (unsigned)(px[i] + kx * g.d) < (unsigned)g.Ws)
                            mk |= 1u << t;
            }
            tapmask[i] = mk;
            voffA[i] = (uint32_t)(base_off[i] + v * VEC) * (uint32_t)sizeof(T);
        }
#pragma unroll
        for (int i = 0; i < BR; ++i) {
            const int r = r0 + 32 * i, n = n0 + r;
            bok[i] = r < BN && n < g.Cd;
            voffB[i] = (uint32_t)(n * g.Ktot + v * VEC) * (uint32_t)sizeof(T);
        }
    }
    // LDS-DMA ring, uniform taps: the chunks of which taps carry a DMA instruction of this wave with ALL lanes out of range (bit t;
    // bit 31: every chunk -- a weight DMA past Cd).  Such DMAs retire at once (tools/exp/oob_order.hip): see the ring below.
    uint32_t u_deadtaps = 0;
    if constexpr (GL > 0) {
        if (UNI) {
            for (int t = 0; t < g.KH * g.KW && t < 31; ++t) {
                bool any = false;
#pragma unroll
                for (int i = 0; i < AR; ++i) any = any || __ballot((tapmask[i] >> t) & 1u) == 0ull;
                u_deadtaps |= any ? 1u << t : 0u;
            }
#pragma unroll
            for (int i = 0; i < BR; ++i) u_deadtaps |= __ballot(bok[i]) == 0ull ? 0x80000000u : 0u;
            u_deadtaps = __builtin_amdgcn_readfirstlane(u_deadtaps);
        }
    }

    typedef __attribute__((address_space(3))) void* lds_ptr;
    auto fetchA = [&](u32x4& dstv, int i, uint32_t off, int gbuf) {
        if constexpr (GL > 0)     // 8 rows x 128 B per wave-instruction, lane-linear: rows 32*i + 8*wave .. +8 of stage gbuf
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srsrc, (lds_ptr)(sA + (gbuf * BM + 32 * i + 8 * wave) * ROWB), 16, off, 0, 0, DSN_DMA_AUX);
        else
            dstv = __builtin_amdgcn_raw_buffer_load_b128(srsrc, off, 0, 0);
    };
    auto fetchB = [&](u32x4& dstv, int i, uint32_t off, int gbuf) {
        if constexpr (GL > 0)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_ptr)(sB + (gbuf * BN + 32 * i + 8 * wave) * ROWB), 16, off, 0, 0, DSN_DMA_AUX);
        else
            dstv = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, off, 0, 0);
    };
    // LDS-DMA ring only: DMA instructions of the chunk just requested whose lanes are ALL out of range (a tap outside the image for
    // the wave's eight rows, rows past M, channels past Cd).  They retire at once (tools/exp/oob_order.hip), so a wave with such
    // operations among the chunks a ring wait counts as in flight drains instead (see the ring below).
    int gl_dead = 0;
    auto all_out = [&](bool lane_in_range) -> int { return __builtin_amdgcn_readfirstlane(__ballot(lane_in_range) == 0ull ? 1 : 0); };
    auto load_chunk = [&](int ch, u32x4 (&ra)[AR], u32x4 (&rb)[BR], int gbuf = 0) {
        const int k0 = ch * KC + v * VEC;
        int nd = 0;
        if (UNI) {
            constexpr uint32_t OOB = 0xFFFFFFF0u;
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                const bool ok = u_tap < 32 && ((tapmask[i] >> (u_tap & 31)) & 1u);
                const uint32_t off = ok ? voffA[i] + u_offA : OOB;
                fetchA(ra[i], i, off, gbuf);
            }
#pragma unroll
            for (int i = 0; i < BR; ++i) {
                const uint32_t off = bok[i] ? voffB[i] + u_offB : OOB;
                fetchB(rb[i], i, off, gbuf);
            }
            if constexpr (GL > 0) gl_dead = (u_tap >= 31 || (u_deadtaps & (0x80000000u | (1u << (u_tap & 31))))) ? 1 : 0;
            u_offB += ROWB;
            if (++u_cc == u_cpt) {
                u_cc = 0;
                ++u_tap;
                if (++u_kx == g.KW) { u_kx = 0; ++u_ky; }
                u_offA = (uint32_t)((u_ky * g.d * g.Ws + u_kx * g.d) * (int32_t)g.sld) * (uint32_t)sizeof(T);
            } else {
                u_offA += ROWB;
            }
        } else if (VECLOAD) {
            const bool kok = k0 < Kc;
            int tap, c, ky, kx;
            if (PAR) {
                tap = k0 / g.Cs;
                c = k0 - tap * g.Cs;
                tap = kok ? g.cls_taps[cls][tap] : 0;
                ky = tap / g.KW; kx = tap - ky * g.KW;
            } else {
                tap = kc_tap; c = kc_c; ky = kc_ky; kx = kc_kx;
                kc_c += KC;
                kc_norm();
            }
            const int kb = PAR ? tap * g.Cs + c : k0;           // position of this vector in the packed weight row
            // buffer loads: an offset beyond the descriptor's range returns zeros, so padding / masked taps / tile
            // overhang cost one v_cndmask on the OFFSET instead of a branch + four on the data
            constexpr uint32_t OOB = 0xFFFFFFF0u;
            if (fast) {
                const int dyo = ky * g.d, dxo = kx * g.d;
                const int32_t off_tap = (dyo * g.Ws + dxo) * (int32_t)g.sld + c;
#pragma unroll
                for (int i = 0; i < AR; ++i) {
                    const bool ok = kok && rowok[i] && (unsigned)(py[i] + dyo) < (unsigned)g.Hs &&
                                    (unsigned)(px[i] + dxo) < (unsigned)g.Ws;
                    const uint32_t off = ok ? (uint32_t)(base_off[i] + off_tap) * (uint32_t)sizeof(T) : OOB;
                    if constexpr (GL > 0) nd += all_out(ok);
                    fetchA(ra[i], i, off, gbuf);
                }
            } else {
#pragma unroll
                for (int i = 0; i < AR; ++i) {
                    const T* p = kok ? src_row(i, ky, kx) : nullptr;
                    const uint32_t off = p ? (uint32_t)((p + c) - src) * (uint32_t)sizeof(T) : OOB;
                    if constexpr (GL > 0) nd += all_out(p != nullptr);
                    fetchA(ra[i], i, off, gbuf);
                }
            }
#pragma unroll
            for (int i = 0; i < BR; ++i) {
                const int r = r0 + 32 * i, n = n0 + r;
                const bool ok = kok && r < BN && n < g.Cd;
                const uint32_t off = ok ? (uint32_t)(n * g.Ktot + kb) * (uint32_t)sizeof(T) : OOB;
                if constexpr (GL > 0) nd += all_out(ok);
                fetchB(rb[i], i, off, gbuf);
            }
            if constexpr (GL > 0) gl_dead = nd;
        } else {   // generic path: per-element tap decode (channel counts that are not a multiple of the vector width)
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                T tmp[VEC];
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const int k = k0 + e;
                    T val = from_f32<T>(0.f);
                    if (k < g.Ktot) {
                        const int tap = k / g.Cs, c = k - tap * g.Cs;
                        const int ky = tap / g.KW, kx = tap - ky * g.KW;
                        const T* p = src_row(i, ky, kx);
                        if (p) val = p[c];
                    }
                    tmp[e] = val;
                }
                ra[i] = *reinterpret_cast<u32x4*>(tmp);
            }
#pragma unroll
            for (int i = 0; i < BR; ++i) {
                T tmp[VEC];
                const int r = r0 + 32 * i, n = n0 + r;
#pragma unroll
                for (int e = 0; e < VEC; ++e)
                    tmp[e] = (r < BN && n < g.Cd && k0 + e < g.Ktot) ? wpk[(int64_t)n * g.Ktot + k0 + e] : from_f32<T>(0.f);
                rb[i] = *reinterpret_cast<u32x4*>(tmp);
            }
        }
    };
    auto store_chunk = [&](int buf, const u32x4 (&ra)[AR], const u32x4 (&rb)[BR]) {
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int r = r0 + 32 * i;
            if (BM % 32 == 0 || r < BM) *reinterpret_cast<u32x4*>(sA + (buf * BM + r) * ROWB + ((v ^ ((r >> 1) & 7)) << 4)) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < BR; ++i) {
            const int r = r0 + 32 * i;
            if (BN % 32 == 0 || r < BN) *reinterpret_cast<u32x4*>(sB + (buf * BN + r) * ROWB + ((v ^ ((r >> 1) & 7)) << 4)) = rb[i];
        }
    };

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fg = lane >> 4;
    auto compute = [&](int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {      // two 64-byte half-chunks
            u32x4 fa[MI], fb[NI];
            const int slot = 4 * h + fg;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int r = (wm * MI + i) * 16 + fr;
                fa[i] = *reinterpret_cast<const u32x4*>(sA + (buf * BM + r) * ROWB + ((slot ^ ((r >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int r = (wn * NI + j) * 16 + fr;
                fb[j] = *reinterpret_cast<const u32x4*>(sB + (buf * BN + r) * ROWB + ((slot ^ ((r >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) Mma<T>::run(acc[i][j], fa[i], fb[j]);
        }
    };

    // ---- main loop: NST register stages feeding 2 LDS buffers, one barrier per chunk ------------------------------------
    // step i: loads of chunk i+NST go into the stage that held chunk i (already in LDS), chunk i runs on the matrix cores,
    // chunk i+1 moves from its stage into the other LDS buffer.
    // Two forms.  Long K (>= g.longk chunks, the 3x3 layers): BRANCH-FREE body -- loads past the last chunk are issued anyway
    // (offsets out of range: zeros, no traffic) and the only control flow is the uniform loop exit, so the compiler's
    // s_waitcnt bookkeeping keeps two chunks in flight (vmcnt(8..11) in the ISA).  With the guards of the short form it joins
    // the paths pessimistically and waits for the loads of the SAME iteration (vmcnt(0..3)): harmless when operands sit in
    // L2, a full HBM round trip per chunk when they are cold, as weights are right after the per-step re-pack.  Short K keeps
    // the guarded form: no wasted out-of-range loads / LDS stores when a block's life is two or three chunks.
    if constexpr (GL > 0) {
        // LDS-DMA ring: chunks i+1 .. i+GL-1 in flight while chunk i computes.  Loads past the last chunk are issued anyway
        // (out-of-range offsets: zeros, no traffic) so that the counted wait is the same constant in every iteration.
        constexpr int LPC = AR + BR;                       // LDS-DMA instructions per chunk per wave
        static_assert((GL - 2) * LPC < 64, "vmcnt is a 6-bit counter");
        static_assert(GL <= 4, "the dead-operation window below holds GL - 2 <= 2 chunks");
        int dq0 = 0, dq1 = 0;                              // dead DMAs of the youngest / second youngest chunk requested
#pragma unroll
        for (int s = 0; s < GL - 1; ++s) { load_chunk(s, rga[0], rgb[0], s); dq1 = dq0; dq0 = s < nchunks ? gl_dead : 0; }
        int lbuf = GL - 1, cbuf = 0;                       // stage the next load goes to / stage of the chunk to compute
        for (int i = 0; i < nchunks; ++i) {
            // (lgkmcnt(0): this wave's fragment reads of chunk i - 1 have COMPLETED, not merely issued, before anyone may overwrite
            //  their stage -- the compiler is free to sink the MFMAs that consume them below the barrier; see conv3x3.hip)
            // this wave's pieces of chunk i have landed.  Behind chunk i only the REAL chunks may be counted as in flight: the loads
            // past the last chunk are all-out-of-range padding DMAs, which retire at once (round 4, tools/exp/oob_order.hip) -- with
            // the constant count the last GL - 2 chunks of a block were read on trust.
            {
                const int rem = nchunks - 1 - i;
                // (the GL - 2 chunks younger than chunk i are the last GL - 2 requested; padding chunks enter the window as 0: the
                //  tail counts leave them out already)
                if (dq0 + (GL >= 4 ? dq1 : 0) != 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                else if (rem >= GL - 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((GL - 2) * LPC) : "memory");
                else if (GL >= 4 && rem == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LPC) : "memory");
                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);             // (issue order pinned at the wait: nothing of chunk i - 1 is placed below it)
            __builtin_amdgcn_s_barrier();                  // everyone's have; and everyone is done reading stage lbuf (chunk i-1)
            asm volatile("" ::: "memory");
            load_chunk(i + GL - 1, rga[0], rgb[0], lbuf);
            dq1 = dq0; dq0 = i + GL - 1 < nchunks ? gl_dead : 0;
            compute(cbuf);
            lbuf = lbuf + 1 == GL ? 0 : lbuf + 1;
            cbuf = cbuf + 1 == GL ? 0 : cbuf + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the (empty) tail loads must not land in the epilogue's staging area
        __syncthreads();
    } else if (nchunks >= g.longk) {
#pragma unroll
        for (int s = 0; s < NST; ++s) load_chunk(s, rga[s], rgb[s]);
        store_chunk(0, rga[0], rgb[0]);
        __syncthreads();
        for (int it = 0; it < nchunks; it += NST) {
#pragma unroll
            for (int s = 0; s < NST; ++s) {
                const int i = it + s;
                if (s > 0 && i >= nchunks) break;
                load_chunk(i + NST, rga[s], rgb[s]);
                compute(i & 1);
                store_chunk((i + 1) & 1, rga[(s + 1) % NST], rgb[(s + 1) % NST]);
                __syncthreads();
            }
        }
    } else {
#pragma unroll
        for (int s = 0; s < NST; ++s)
            if (nchunks > s) load_chunk(s, rga[s], rgb[s]);
        if (nchunks > 0) store_chunk(0, rga[0], rgb[0]);
        __syncthreads();
        for (int it = 0; it < nchunks; it += NST) {
#pragma unroll
            for (int s = 0; s < NST; ++s) {
                const int i = it + s;
                if (i < nchunks) {
                    if (i + NST < nchunks) load_chunk(i + NST, rga[s], rgb[s]);
                    compute(i & 1);
                    if (i + 1 < nchunks) store_chunk((i + 1) & 1, rga[(s + 1) % NST], rgb[(s + 1) % NST]);
                    __syncthreads();
                }
            }
        }
    }

    // ---- epilogue: stage act(acc + bias) as fp32 [BM][BN + CPAD] ------------------------------------------------
    constexpr int LDC = BN + CPAD;
    constexpr int VPR = BN / VEC;                   // vectors per tile row
    constexpr int NIT = (BM * VPR + 255) / 256;     // vectors per thread in the vectorised tile store
    // vector idx of the tile -> destination (row, first channel); false: nothing is stored there
    auto dest = [&](int idx, int& row, int& col) -> bool {
        const int rl = idx / VPR, cv = idx - rl * VPR;
        col = n0 + cv * VEC;
        row = m0 + rl;
        if (PAR) {
            int y, x, n;
            if (!dst_pixel(rl, y, x, n)) return false;
            row = (n * g.Hd + y) * g.Wd + x;
        }
        if (row >= g.M || col >= g.Cd) return false;
        if (g.d2s_c > 0) {
            const int cls = col / g.d2s_c;
            col -= cls * g.d2s_c;
            const int x = row % g.Wd, t = row / g.Wd;
            const int Y = 2 * (t % g.Hd) + (cls >> 1), X = 2 * x + (cls & 1);
            if (Y >= g.Hout || X >= g.Wout) return false;
            row = ((t / g.Hd) * g.Hout + Y) * g.Wout + X;
        }
        return true;
    };
    BnRedLane<T, VEC, NIT> bl;                      // (dgrad that completes dz of a BatchNorm block: its backward sums ride here)
    if (br.nseg) {                                  // (host: only with the vectorised store)
        const int c = n0 + (tid % VPR) * VEC;
        bl.init(br, g.d2s_c > 0 ? c % g.d2s_c : c);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            int row, col;
            const bool ok = tid + it * 256 < BM * VPR && dest(tid + it * 256, row, col);
            bl.prefetch(it, ok ? row : -1);
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

    // optional BatchNorm partial statistics of this tile (rows m0 .. m0+BM) -> stats[tm][0: sum, 1: sumsq][Cd]
    if (stats || fin.acc) {
        float* red = sC + BM * LDC;                 // 2 * 256 floats of scratch behind the staged tile
        constexpr int TYS = 256 / BN > 0 ? 256 / BN : 1;
        const int tx = tid % BN, ty = tid / BN;
        float s = 0.f, ss = 0.f;
        if (ty < TYS) {
            const int rows = (cM - cm0 < BM) ? cM - cm0 : BM;
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
            if (fin.acc) {
                bn_acc_add(fin, tm, n0 + tid, t0, t1);
            } else {
                stats[((int64_t)tm * 2) * g.Cd + n0 + tid] = t0;
                stats[((int64_t)tm * 2 + 1) * g.Cd + n0 + tid] = t1;
            }
        }
    }

    // ---- vectorised tile store (+ residual, += destination) -----------------------------------------------------
    const bool vst = (g.Cd % VEC == 0) && (g.dld % VEC == 0) && (((uintptr_t)dst) % 16 == 0) &&
                     (!res || ((g.rld % VEC == 0) && (((uintptr_t)res) % 16 == 0)));
    if (vst) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * 256;
            if (idx >= BM * VPR) break;
            const int rl = idx / VPR, cv = idx - rl * VPR;
            int row, col;
            if (!dest(idx, row, col)) continue;
            float vals[VEC];
#pragma unroll
            for (int e = 0; e < VEC; e += 4) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(sC + rl * LDC + cv * VEC + e);
                vals[e] = t[0]; vals[e + 1] = t[1]; vals[e + 2] = t[2]; vals[e + 3] = t[3];
            }
            T* o = dst + (int64_t)row * g.dld + col;
            if (res) {
                T rv[VEC];
                *reinterpret_cast<u32x4*>(rv) = *reinterpret_cast<const u32x4*>(res + (int64_t)row * g.rld + col);
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
            __syncthreads();                        // (the staged tile is no longer needed: its LDS takes the partial sums)
            bl.template finish<VPR>(br, sC, tm, n0, g.Cd, g.d2s_c);
        }
    } else {
        for (int idx = tid; idx < BM * BN; idx += 256) {
            const int rl = idx / BN, cl = idx - rl * BN;
            const int col = n0 + cl;
            int row = m0 + rl;
            if (PAR) {
                int y, x, n;
                if (!dst_pixel(rl, y, x, n)) continue;
                row = (n * g.Hd + y) * g.Wd + x;
            }
            if (row >= g.M || col >= g.Cd) continue;
            float val = sC[rl * LDC + cl];
            if (res) val += to_f32<T>(res[(int64_t)row * g.rld + col]);
            T* o = dst + (int64_t)row * g.dld + col;
            if (g.accumulate) val += to_f32<T>(*o);
            *o = from_f32<T>(val);
        }
    }
}

template <typename T, int MI, int NI, int WGM, int WGN>
int launch_cfg(const T* src, const T* w, const float* bias, const T* res, T* dst, float* stats, const BnAcc& fin, Geom g,
               bool vec, hipStream_t st, int* tiles_m_out, const BnRed* brp = nullptr) {
    const BnRed br = brp ? *brp : BnRed{};
    if (br.nseg > 0 && !(vec && g.Cd % Mma<T>::VEC == 0 && g.dld % Mma<T>::VEC == 0 && ((uintptr_t)dst) % 16 == 0 &&
                         (!res || (g.rld % Mma<T>::VEC == 0 && ((uintptr_t)res) % 16 == 0))))
        DSN_FAIL(DSN_EUNSUPPORTED, "conv dgrad with BatchNorm sums: the layer cannot take the vectorised store");
    constexpr int BM = WGM * MI * 16, BN = WGN * NI * 16;
    constexpr int CFG = (BM == 128 && BN == 128) ? 0 : (BM == 128 && BN == 64) ? 1 : (BM == 64 && BN == 64) ? 2
                        : (BM == 128 && BN == 32) ? 3 : (BM == 64 && BN == 16) ? 4 : (BM == 32 && BN == 64) ? 5 : 6;
    const int64_t src_elems_ld = ((int64_t)(g.M / (g.Hd * g.Wd)) * g.Hs * g.Ws - 1) * g.sld + g.Cs;
    g.fits32 = (src_elems_ld < ((1ll << 30) - (1ll << 22))) && ((int64_t)g.Cd * g.Ktot < (1ll << 30));
    g.src_bytes = (uint32_t)(src_elems_ld * sizeof(T));
    g.w_bytes = (uint32_t)((int64_t)g.Cd * g.Ktot * sizeof(T));
    vec = vec && g.fits32;     // the vector path addresses through 32-bit buffer offsets
    const bool par = g.is_dgrad && g.q == 2 && vec;
    if (par) {
        const int nimg = g.M / (g.Hd * g.Wd);
        int t = 0;
        for (int c = 0; c < 4; ++c) {
            const int py = c >> 1, px = c & 1;
            g.cls_h[c] = (g.Hd - py + 1) / 2;
            g.cls_w[c] = (g.Wd - px + 1) / 2;
            g.cls_tile0[c] = t;
            t += (int)(((int64_t)nimg * g.cls_h[c] * g.cls_w[c] + BM - 1) / BM);
            int nt = 0;
            for (int ky = 0; ky < g.KH; ++ky)
                for (int kx = 0; kx < g.KW; ++kx) {
                    const int vy = py + g.b + ky * g.d, vx = px + g.b + kx * g.d;     // b = +pad, d = -dil
                    if (((vy % 2) + 2) % 2 == 0 && ((vx % 2) + 2) % 2 == 0 && nt < 12) g.cls_taps[c][nt++] = (uint8_t)(ky * g.KW + kx);
                }
            g.cls_ntaps[c] = nt;
        }
        g.cls_tile0[4] = t;
        g.tiles_m = t;
    } else {
        g.tiles_m = (g.M + BM - 1) / BM;
    }
    g.tiles_n = (g.Cd + BN - 1) / BN;
    if (tiles_m_out) *tiles_m_out = g.tiles_m;
    dim3 grid(g.tiles_m * g.tiles_n), block(256);
    // algorithmic work of this launch: every source/destination element and every weight touched once
    const double K = (double)g.Ktot;
    const double src_elems = (double)(g.M / (g.Hd * g.Wd)) * g.Hs * g.Ws * g.Cs;
    const double elems = src_elems + (double)g.M * g.Cd * (1 + (res ? 1 : 0) + (g.accumulate ? 1 : 0)) + K * g.Cd +
                         (double)g.M * (g.d2s_c > 0 ? 4.0 : 1.0) * bnred_channels(brp);
    (void)CFG;
    // (a, q): forward stride a, dgrad source stride q; d = +-dilation; depth-to-space = the stride-2 dgrad in its 2x2 form
    const ProfConv pc("igemm_kernel", sizeof(T) == 2, BM, BN, g.is_dgrad != 0, g.KH, g.d2s_c > 0 ? 2 : (g.is_dgrad ? g.q : g.a),
                      g.d < 0 ? -g.d : g.d, g.Cs, g.Cd, g.M / (g.Hd * g.Wd), g.Hd, g.Wd);
    ProfScope prof(pc.label, pc.layer, 2.0 * g.M * g.Cd * K, elems * sizeof(T), st);
    static const bool no_uni = getenv("DSN_IGEMM_NOUNI") != nullptr;                                          // tuning knob
    static const int longk = [] { const char* e = getenv("DSN_IGEMM_LONGK"); return e ? atoi(e) : 9; }();
    g.longk = longk;
    const bool uni = vec && !par && g.q == 1 && g.Cs % (ROWB / (int)sizeof(T)) == 0 && g.KH * g.KW <= 32 && !no_uni;
    // LDS-DMA ring (3 stages) instead of register staging.  Measured per layer (tools/bench_ops.py, MI355X): it wins where the
    // grid is small -- at most ~1.75 blocks of 64x64 per CU, so the larger LDS footprint costs no occupancy and the deeper
    // prefetch is all gain (3x3 on 20x20 / 40x40 maps: 16.8 -> 13.9, 23.7 -> 19.8, 14.8 -> 12.9 us; 1x1 512 -> 256 @20: 7.4 ->
    // 5.7) -- and loses on the large-grid layers (1x1 128 -> 64 @80: 7.3 -> 9.3 us: three resident blocks instead of four).
    // The 128x128 tile (96 KB: one block per CU) lost with the ring on every config-5 layer (65.7 -> 87.9 us on 3x3 128 -> 128 @160)
    // and a 4-stage ring lost to the 3-stage one everywhere (a third fewer resident blocks): neither is built.
    // DSN_IGEMM_GL = 0 (never) | 3 (always) overrides.
    static const int gl_env = [] { const char* e = getenv("DSN_IGEMM_GL"); return e ? atoi(e) : -1; }();
    if constexpr (BM % 32 == 0 && BN % 32 == 0 && BM * BN <= 128 * 64) {
        const int64_t blocks = (int64_t)g.tiles_m * g.tiles_n;
        const bool want = gl_env >= 0 ? gl_env == 3 : (blocks <= 448 && g.Ktot * (int)sizeof(T) >= 512);
        if (!par && vec && want) {
            if (uni) hipLaunchKernelGGL((igemm_kernel<T, MI, NI, WGM, WGN, 2, false, 3, 3>), grid, block, 0, st, src, w, bias, res, dst, stats, fin, g, br);
            else hipLaunchKernelGGL((igemm_kernel<T, MI, NI, WGM, WGN, 1, false, 3, 3>), grid, block, 0, st, src, w, bias, res, dst, stats, fin, g, br);
            DSN_LAUNCH_CHECK("igemm (LDS-DMA)");
            return DSN_OK;
        }
    }
    if (par)
        hipLaunchKernelGGL((igemm_kernel<T, MI, NI, WGM, WGN, 1, true>), grid, block, 0, st, src, w, bias, res, dst, stats, fin, g, br);
    else if (uni)
        hipLaunchKernelGGL((igemm_kernel<T, MI, NI, WGM, WGN, 2, false>), grid, block, 0, st, src, w, bias, res, dst, stats, fin, g, br);
    else if (vec)
        hipLaunchKernelGGL((igemm_kernel<T, MI, NI, WGM, WGN, 1, false>), grid, block, 0, st, src, w, bias, res, dst, stats, fin, g, br);
    else
        hipLaunchKernelGGL((igemm_kernel<T, MI, NI, WGM, WGN, 0, false>), grid, block, 0, st, src, w, bias, res, dst, stats, fin, g, br);
    DSN_LAUNCH_CHECK("igemm");
    return DSN_OK;
}

template <typename T>
int launch(const dsn_tensor* s, const void* w, const float* bias, const dsn_tensor* r, const dsn_tensor* d, float* stats,
           const BnAcc& fin, Geom g, hipStream_t st, int* tiles_m_out, const BnRed* br = nullptr) {
    constexpr int VEC = Mma<T>::VEC;
    const T* src = (const T*)s->ptr;
    const T* res = r ? (const T*)r->ptr : nullptr;
    T* dst = (T*)d->ptr;
    const bool vec = (g.Cs % VEC == 0) && (g.sld % VEC == 0) && (((uintptr_t)src) % 16 == 0) &&
                     (((uintptr_t)w) % 16 == 0);
    // tile choice: wide-N tiles for wide layers, tall-skinny for narrow ones; small M prefers smaller tiles so the
    // grid still covers the 256 CUs.
    static const int force = [] { const char* e = getenv("DSN_IGEMM_CFG"); return e ? atoi(e) : -1; }();    // tuning knob
    static const int force_s2 = [] { const char* e = getenv("DSN_IGEMM_S2_CFG"); return e ? atoi(e) : -1; }();    // tuning knob: stride-2 layers only
    const bool is_s2 = g.d2s_c > 0 || (g.a == 2 && g.KH == 3);
    switch (is_s2 && force_s2 >= 0 ? force_s2 : force) {
        case 0: return launch_cfg<T, 4, 4, 2, 2>(src, (const T*)w, bias, res, dst, stats, fin, g, vec, st, tiles_m_out, br);
        case 1: return launch_cfg<T, 4, 2, 2, 2>(src, (const T*)w, bias, res, dst, stats, fin, g, vec, st, tiles_m_out, br);
        case 2: return launch_cfg<T, 2, 2, 2, 2>(src, (const T*)w, bias, res, dst, stats, fin, g, vec, st, tiles_m_out, br);
        case 3: return launch_cfg<T, 2, 2, 4, 1>(src, (const T*)w, bias, res, dst, stats, fin, g, vec, st, tiles_m_out, br);
        case 4: return launch_cfg<T, 1, 1, 4, 1>(src, (const T*)w, bias, res, dst, stats, fin, g, vec, st, tiles_m_out, br);
        case 5: return launch_cfg<T, 1, 2, 2, 2>(src, (const T*)w, bias, res, dst, stats, fin, g, vec, st, tiles_m_out, br);
        case 6: return launch_cfg<T, 2, 1, 2, 2>(src, (const T*)w, bias, res, dst, stats, fin, g, vec, st, tiles_m_out, br);
        default: break;
    }
    // Tile choice, from tools/sweep_igemm.sh on the DeSeNet-s layer shapes (batch 8): these layers are latency-bound, not
    // MFMA-bound, so SMALL tiles win almost everywhere (more blocks per CU hide the load -> LDS -> MFMA round trips);
    // only the few large GEMMs (FFM 3x3: 1600 tiles x K 2304) amortise a 128x128 tile.
    const int64_t tiles64 = (int64_t)((g.M + 63) / 64) * ((g.Cd + 63) / 64);
    if (g.Cd <= 16) return launch_cfg<T, 1, 1, 4, 1>(src, (const T*)w, bias, res, dst, stats, fin, g, vec, st, tiles_m_out, br);
    if (g.Cd <= 32) {
        if (g.M >= 400000)      // Focus conv / stem dgrad: 128 pixels x 32 channels
            return launch_cfg<T, 2, 2, 4, 1>(src, (const T*)w, bias, res, dst, stats, fin, g, vec, st, tiles_m_out, br);
        return launch_cfg<T, 2, 1, 2, 2>(src, (const T*)w, bias, res, dst, stats, fin, g, vec, st, tiles_m_out, br);      // 64x32
    }
    if (tiles64 >= 1536 && g.Ktot >= 1024 && g.Cd >= 128)
        return launch_cfg<T, 4, 4, 2, 2>(src, (const T*)w, bias, res, dst, stats, fin, g, vec, st, tiles_m_out, br);      // 128x128
    // (256 .. 400 tiles: 32x64 measured 5.0 vs 5.6 us per launch on short-K layers standalone, but inside the step the 64x64 tile is
    // ahead -- 3.938 vs 3.947 ms, DSN_IGEMM_SMALLK = 0 vs 512 = largest K that still takes the small tile there)
    static const int smallk = [] { const char* e = getenv("DSN_IGEMM_SMALLK"); return e ? atoi(e) : 0; }();
    if (tiles64 < 256 || (tiles64 <= 400 && g.Ktot <= smallk))
        return launch_cfg<T, 1, 2, 2, 2>(src, (const T*)w, bias, res, dst, stats, fin, g, vec, st, tiles_m_out, br);      // 32x64
    return launch_cfg<T, 2, 2, 2, 2>(src, (const T*)w, bias, res, dst, stats, fin, g, vec, st, tiles_m_out, br);          // 64x64
}

int check_common(const dsn_tensor* s, const void* w, const dsn_tensor* d, const dsn_conv_params* p) {
    DSN_CHECK_ARG(tensor_ok(s) && tensor_ok(d) && w && p, "conv: null/invalid tensor");
    DSN_CHECK_ARG(s->dtype == d->dtype, "conv: dtype mismatch");
    DSN_CHECK_ARG(s->n == d->n, "conv: batch mismatch");
    DSN_CHECK_ARG(p->kh > 0 && p->kw > 0 && p->stride > 0 && p->dil > 0 && p->pad >= 0, "conv: bad params");
    DSN_CHECK_ARG(npix(s) < (1ll << 31) && npix(d) < (1ll << 31), "conv: too many pixels for int32 indexing");
    return DSN_OK;
}

int conv_fwd_impl(const dsn_tensor* x, const void* w, const float* bias, const dsn_tensor* residual, const dsn_tensor* y,
                  const dsn_conv_params* p, float* stats, int* tiles_m_out, void* stream, const BnAcc* finp = nullptr) {
    int rc = check_common(x, w, y, p);
    if (rc) return rc;
    const int ho = (x->h + 2 * p->pad - p->dil * (p->kh - 1) - 1) / p->stride + 1;
    const int wo = (x->w + 2 * p->pad - p->dil * (p->kw - 1) - 1) / p->stride + 1;
    DSN_CHECK_ARG(ho == y->h && wo == y->w, "conv fwd: output is %dx%d, expected %dx%d", y->h, y->w, ho, wo);
    if (residual)
        DSN_CHECK_ARG(tensor_ok(residual) && residual->dtype == y->dtype && residual->n == y->n &&
                          residual->h == y->h && residual->w == y->w && residual->c == y->c,
                      "conv fwd: residual shape mismatch");
    if (!stats) {      // 3x3 / stride 1 with whole 64-channel slabs: the halo-tile kernel (conv3x3.hip)
        rc = dsn_conv1x1_pp_try(x, w, bias, residual, y, p, 0, finp, stream);       // long-K 1x1 on big tiles (conv_pp.hip)
        if (rc != 1) return rc;
        rc = dsn_conv1x1_ws_try(x, w, bias, residual, y, p, 0, finp, stream);       // 1x1, weights-stationary persistent blocks
        if (rc != 1) return rc;
        rc = dsn_conv3x3_pp_try(x, w, bias, residual, y, p, 0, finp, stream);       // 3x3 / stride 1 on big tiles (conv_pp.hip)
        if (rc != 1) return rc;
        rc = dsn_conv3x3_ws_try(x, w, bias, residual, y, p, 0, finp, stream);       // 3x3 / stride 1, the same
        if (rc != 1) return rc;
        rc = dsn_conv3x3s2_ws_try(x, w, bias, residual, y, p, finp, stream);        // 3x3 / stride 2 stems: gathered K
        if (rc != 1) return rc;
        rc = dsn_conv3x3_halo_try(x, w, bias, residual, y, p, 0, finp, stream);
        if (rc != 1) return rc;
        rc = dsn_conv1x1_dma_try(x, w, bias, residual, y, p, 0, finp, stream);      // 1x1 with <= 4 channel slabs
        if (rc != 1) return rc;
    }
    Geom g{};
    g.M = (int32_t)npix(y); g.Hd = y->h; g.Wd = y->w;
    g.Hs = x->h; g.Ws = x->w; g.Cs = x->c; g.Cd = y->c;
    g.KH = p->kh; g.KW = p->kw; g.Ktot = p->kh * p->kw * x->c;
    g.a = p->stride; g.b = -p->pad; g.d = p->dil; g.q = 1;
    g.act = p->act; g.accumulate = p->accumulate;
    g.sld = x->ldc; g.dld = y->ldc; g.rld = residual ? residual->ldc : 0;
    BnAcc fin{};
    if (finp) fin = *finp;
    if (x->dtype == DSN_F32) return launch<float>(x, w, bias, residual, y, stats, fin, g, (hipStream_t)stream, tiles_m_out);
    return launch<bf16_t>(x, w, bias, residual, y, stats, fin, g, (hipStream_t)stream, tiles_m_out);
}

}  // namespace

extern "C" int dsn_conv2d_fwd(const dsn_tensor* x, const void* w, const float* bias, const dsn_tensor* residual,
                              const dsn_tensor* y, const dsn_conv_params* p, void* stream) {
    return conv_fwd_impl(x, w, bias, residual, y, p, nullptr, nullptr, stream);
}

// conv (no bias / activation / residual) whose epilogue also emits per-tile BatchNorm partial sums: stats must hold
// dsn_conv2d_stats_rows(M) * 2 * Co floats; *rows_out receives the number of rows actually written.
extern "C" int32_t dsn_conv2d_stats_rows(int64_t out_pixels) { return (int32_t)((out_pixels + 63) / 64); }

extern "C" int dsn_conv2d_fwd_stats(const dsn_tensor* x, const void* w, const dsn_tensor* y, const dsn_conv_params* p,
                                    float* stats, int32_t* rows_out, void* stream) {
    DSN_CHECK_ARG(stats && rows_out && p && p->act == DSN_ACT_NONE && !p->accumulate,
                  "conv2d_fwd_stats: needs a stats buffer and a plain convolution");
    int rows = 0;
    int rc = conv_fwd_impl(x, w, nullptr, nullptr, y, p, stats, &rows, stream);
    *rows_out = rows;
    return rc;
}

// Training forward of conv + BatchNorm statistics in ONE launch: the epilogue reduces the fp32 accumulators per channel and
// adds them into `acc` (dsn_bn_acc_bytes(Co) bytes, ZERO on entry); dsn_bn_act_fwd_acc folds them in its prologue.
extern "C" int dsn_conv2d_fwd_bnacc(const dsn_tensor* x, const void* w, const dsn_tensor* y, const dsn_conv_params* p,
                                    void* acc, int64_t acc_bytes, void* stream) {
    DSN_CHECK_ARG(p && p->act == DSN_ACT_NONE && !p->accumulate, "conv2d_fwd_bnacc: needs a plain convolution");
    DSN_CHECK_ARG(y && acc, "conv2d_fwd_bnacc: null argument");
    if (acc_bytes < bn_acc_bytes(y->c)) DSN_FAIL(DSN_EWORKSPACE, "conv2d_fwd_bnacc: accumulator buffer too small");
    BnAcc f{(double*)acc, y->c, (double)npix(y)};
    return conv_fwd_impl(x, w, nullptr, nullptr, y, p, nullptr, nullptr, stream, &f);
}

// Stride-2 3x3 (pad 1) input gradient as ONE stride-1 2x2 convolution over dy with 4*Ci output columns + depth-to-space store:
//   dx[n, 2y+py, 2x+px, ci] = sum_{ty,tx in {0,1}} sum_co dy[n, y+ty, x+tx, co] * W2[(py,px,ci)][ty][tx][co]
// with W2 = w[co][ci][ky][kx] for (py,ty) -> ky in {(0,0)->1, (1,0)->2, (1,1)->0} (same for x), zero otherwise
// (dsn_pack_desc.out_dgrad_s2).  Versus the parity-class form: every block runs the full 4-tap K loop (no 1-chunk tiles),
// M is the dy grid (4x fewer, 4x wider tiles) and each (n, y) writes two FULL destination rows (2x+px adjacent) instead of
// every other pixel.  16/9 of the MACs -- irrelevant, these layers are latency/traffic-bound.
namespace {
int bnred_check(const dsn_bnred* br, const dsn_tensor* dx) {
    if (!br) return DSN_OK;
    const int vec = dx->dtype == DSN_F32 ? 4 : 8;
    DSN_CHECK_ARG(br->nseg >= 0 && br->nseg <= DSN_BNRED_MAXSEG, "conv dgrad with BatchNorm sums: %d segments", br->nseg);
    for (int i = 0; i < br->nseg; ++i) {
        const dsn_bnred_seg& s = br->seg[i];
        DSN_CHECK_ARG(s.c0 >= 0 && s.c1 > s.c0 && s.c1 <= dx->c, "conv dgrad with BatchNorm sums: segment %d covers [%d, %d) of %d channels",
                      i, s.c0, s.c1, dx->c);
        DSN_CHECK_ARG(s.y && s.scale && s.shift && s.mean && s.rstd && s.acc && s.acc_c >= s.ch0 + (s.c1 - s.c0) && s.ch0 >= 0,
                      "conv dgrad with BatchNorm sums: null / short operand in segment %d", i);
        if (s.c0 % vec || s.c1 % vec || s.yld % vec || (uintptr_t)s.y % 16 || (uintptr_t)s.scale % 16 || (uintptr_t)s.shift % 16 ||
            (uintptr_t)s.mean % 16 || (uintptr_t)s.rstd % 16)
            DSN_FAIL(DSN_EUNSUPPORTED, "conv dgrad with BatchNorm sums: segment %d is not made of 16-byte channel vectors", i);
        for (int j = 0; j < i; ++j)
            DSN_CHECK_ARG(s.c0 >= br->seg[j].c1 || s.c1 <= br->seg[j].c0, "conv dgrad with BatchNorm sums: segments %d and %d overlap", j, i);
    }
    return DSN_OK;
}
}  // namespace

static int conv_dgrad_s2_impl(const dsn_tensor* dy, const void* w_s2, const dsn_tensor* dx, const dsn_conv_params* p,
                              const dsn_bnred* br, void* stream) {
    int rc = check_common(dy, w_s2, dx, p);
    if (rc) return rc;
    if ((rc = bnred_check(br, dx))) return rc;
    DSN_CHECK_ARG(p->kh == 3 && p->kw == 3 && p->stride == 2 && p->pad == 1 && p->dil == 1,
                  "conv dgrad_s2: 3x3 / stride 2 / pad 1 only");
    const int ho = (dx->h + 2 - 3) / 2 + 1, wo = (dx->w + 2 - 3) / 2 + 1;
    DSN_CHECK_ARG(ho == dy->h && wo == dy->w, "conv dgrad_s2: dy is %dx%d, expected %dx%d", dy->h, dy->w, ho, wo);
    const int vecw = dy->dtype == DSN_F32 ? 4 : 8;
    DSN_CHECK_ARG(dx->c % vecw == 0 && dx->ldc % vecw == 0 && ((uintptr_t)dx->ptr % 16) == 0 && dy->c % vecw == 0,
                  "conv dgrad_s2: channel counts / alignment need the 16-byte paths");
    Geom g{};
    g.M = (int32_t)npix(dy); g.Hd = dy->h; g.Wd = dy->w;           // GEMM rows = dy pixels
    g.Hs = dy->h; g.Ws = dy->w; g.Cs = dy->c; g.Cd = 4 * dx->c;
    g.KH = 2; g.KW = 2; g.Ktot = 4 * dy->c;
    g.a = 1; g.b = 0; g.d = 1; g.q = 1;
    g.act = DSN_ACT_NONE; g.accumulate = p->accumulate; g.is_dgrad = 1;
    g.sld = dy->ldc; g.dld = dx->ldc; g.rld = 0;
    g.d2s_c = dx->c; g.Hout = dx->h; g.Wout = dx->w;
    rc = dsn_dgrad_s2_pp_try(dy, w_s2, dx, p, br, stream);                          // big-tile ping-pong kernel, 2x2 taps (conv_pp.hip)
    if (rc != 1) return rc;
    rc = dsn_dgrad_s2_ws_try(dy, w_s2, dx, p, br, stream);                          // the large stems: weights-stationary, gathered K
    if (rc != 1) return rc;
    if (dy->dtype == DSN_F32)
        return launch<float>(dy, w_s2, nullptr, nullptr, dx, nullptr, BnAcc{}, g, (hipStream_t)stream, nullptr, br);
    return launch<bf16_t>(dy, w_s2, nullptr, nullptr, dx, nullptr, BnAcc{}, g, (hipStream_t)stream, nullptr, br);
}

extern "C" int dsn_conv2d_dgrad_s2(const dsn_tensor* dy, const void* w_s2, const dsn_tensor* dx, const dsn_conv_params* p,
                                   void* stream) {
    return conv_dgrad_s2_impl(dy, w_s2, dx, p, nullptr, stream);
}

namespace {
int conv_dgrad_impl(const dsn_tensor* dy, const void* w, const dsn_tensor* dx, const dsn_conv_params* p,
                    const dsn_tensor* residual, void* stream, const dsn_bnred* br = nullptr) {
    int rc = check_common(dy, w, dx, p);
    if (rc) return rc;
    if ((rc = bnred_check(br, dx))) return rc;
    const int ho = (dx->h + 2 * p->pad - p->dil * (p->kh - 1) - 1) / p->stride + 1;
    const int wo = (dx->w + 2 * p->pad - p->dil * (p->kw - 1) - 1) / p->stride + 1;
    DSN_CHECK_ARG(ho == dy->h && wo == dy->w, "conv dgrad: dy is %dx%d, expected %dx%d", dy->h, dy->w, ho, wo);
    if (residual)
        DSN_CHECK_ARG(tensor_ok(residual) && residual->dtype == dx->dtype && residual->n == dx->n && residual->h == dx->h &&
                          residual->w == dx->w && residual->c == dx->c && p->stride == 1,
                      "conv dgrad: residual must have dx's shape (stride-1 convolutions only)");
    rc = dsn_conv1x1_pp_try(dy, w, nullptr, residual, dx, p, 1, nullptr, stream, br);       // long-K 1x1 on big tiles (conv_pp.hip)
    if (rc != 1) return rc;
    rc = dsn_conv1x1_ws_try(dy, w, nullptr, residual, dx, p, 1, nullptr, stream, br);
    if (rc != 1) return rc;
    rc = dsn_conv3x3_pp_try(dy, w, nullptr, residual, dx, p, 1, nullptr, stream, br);       // big-tile ping-pong kernel (conv_pp.hip)
    if (rc != 1) return rc;
    rc = dsn_conv3x3_ws_try(dy, w, nullptr, residual, dx, p, 1, nullptr, stream, br);
    if (rc != 1) return rc;
    rc = dsn_conv3x3_halo_try(dy, w, nullptr, residual, dx, p, 1, nullptr, stream, br);
    if (rc != 1) return rc;
    rc = dsn_conv1x1_dma_try(dy, w, nullptr, residual, dx, p, 1, nullptr, stream, br);
    if (rc != 1) return rc;
    Geom g{};
    g.M = (int32_t)npix(dx); g.Hd = dx->h; g.Wd = dx->w;
    g.Hs = dy->h; g.Ws = dy->w; g.Cs = dy->c; g.Cd = dx->c;
    g.KH = p->kh; g.KW = p->kw; g.Ktot = p->kh * p->kw * dy->c;
    g.a = 1; g.b = p->pad; g.d = -p->dil; g.q = p->stride;
    g.act = DSN_ACT_NONE; g.accumulate = p->accumulate; g.is_dgrad = 1;
    g.sld = dy->ldc; g.dld = dx->ldc; g.rld = residual ? residual->ldc : 0;
    if (dy->dtype == DSN_F32)
        return launch<float>(dy, w, nullptr, residual, dx, nullptr, BnAcc{}, g, (hipStream_t)stream, nullptr, br);
    return launch<bf16_t>(dy, w, nullptr, residual, dx, nullptr, BnAcc{}, g, (hipStream_t)stream, nullptr, br);
}
}  // namespace

// Input gradient whose launch writes the FINAL value of dz for one or two BatchNorm blocks upstream: their backward sums are
// formed in the epilogue (include/desenet_hip.h: dsn_bnred).  residual may be NULL.  DSN_EUNSUPPORTED when the layer cannot take
// the vectorised store: run dsn_conv2d_dgrad[_res] and dsn_bn_act_bwd_reduce instead.
extern "C" int dsn_conv2d_dgrad_bnred(const dsn_tensor* dy, const void* w, const dsn_tensor* dx, const dsn_conv_params* p,
                                      const dsn_tensor* residual, const dsn_bnred* red, void* stream) {
    DSN_CHECK_ARG(red && red->nseg > 0, "conv2d_dgrad_bnred: needs at least one segment");
    return conv_dgrad_impl(dy, w, dx, p, residual, stream, red);
}
extern "C" int dsn_conv2d_dgrad_s2_bnred(const dsn_tensor* dy, const void* w_s2, const dsn_tensor* dx, const dsn_conv_params* p,
                                         const dsn_bnred* red, void* stream) {
    DSN_CHECK_ARG(red && red->nseg > 0, "conv2d_dgrad_s2_bnred: needs at least one segment");
    return conv_dgrad_s2_impl(dy, w_s2, dx, p, red, stream);
}

extern "C" int dsn_conv2d_dgrad(const dsn_tensor* dy, const void* w, const dsn_tensor* dx, const dsn_conv_params* p,
                                void* stream) {
    return conv_dgrad_impl(dy, w, dx, p, nullptr, stream);
}

// dx (+)= conv_transpose(dy, w) + residual: the Bottleneck shortcut's gradient (common.py:111, `x + cv2(cv1(x))`) rides in the
// epilogue of cv1's input gradient instead of a separate copy / add pass.
extern "C" int dsn_conv2d_dgrad_res(const dsn_tensor* dy, const void* w, const dsn_tensor* dx, const dsn_conv_params* p,
                                    const dsn_tensor* residual, void* stream) {
    return conv_dgrad_impl(dy, w, dx, p, residual, stream);
}
