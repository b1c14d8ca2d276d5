// Detect head decode (yolo.py:255-277) and non-max suppression (general.py:659-750 + torchvision.ops.nms).
//
// This translation unit is compiled with -ffp-contract=off: the NMS selection must be BIT-EXACT against the fp32 CPU
// oracle, so `a1 + a2 - w*h`, `x - w/2`, `obj*cls` must round exactly like separate IEEE fp32 operations.
//
// NMS pipeline per image (one launch each, all images in parallel):
//   1. nms_candidates : every prediction row (x class) that passes the confidence tests appends ONE 64-bit key
//                       (~bits(conf) << 32 | row*nc + cls) through a per-image atomic counter.  Keys are unique, and sorting
//                       them ascending == stable descending-confidence order of the reference's row-major candidate list,
//                       so the arrival order of the atomics is irrelevant (deterministic result).
//   2. sort           : bitonic network, hand-written: one 1024-thread workgroup per image in LDS for <= 8192 keys; four
//                       8192-key tiles per image + two cross-tile merge launches up to 32768; above that (val setting: up to
//                       151,200 keys) a radix select keeps the 30000 best first (stage 2 below).
//   3. nms_greedy     : one 256-thread workgroup per image walks the sorted keys (capped at 30000) in chunks of 64: every
//                       candidate is tested against the boxes kept so far (4 waves split the kept list), then wave 0
//                       resolves the chunk internally with ballots; stops as soon as max_det boxes are kept.
#include "common.h"

namespace {

constexpr int MAX_WH = 4096;
constexpr int MAX_NMS = 30000;
constexpr int SORT_THREADS = 1024;
constexpr int SORT_LDS_KEYS = 8192;   // 64 KiB of LDS
// LDS position of key i of a tile: one key of padding per 32, so that the strided 8-byte accesses of the sort (lane stride 8, 16, ...
// keys in the late phases of every merge) spread over all banks
__device__ __forceinline__ int skew(int i) { return i + (i >> 5); }
constexpr int SORT_LDS_SLOTS = SORT_LDS_KEYS + SORT_LDS_KEYS / 32;
constexpr int MAX_DET_CAP = 4096;

#define GRID_STRIDE(i, total) \
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (total); i += (int64_t)gridDim.x * blockDim.x)

// ---- Detect ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void detect_decode_kernel(const T* __restrict__ t, int64_t tld, float* __restrict__ raw,
                                     float* __restrict__ pred, int64_t pred_rows, int64_t row_off, int N, int ny, int nx,
                                     int na, int no, float stride, const float* __restrict__ anchors) {
    const int C = na * no;
    const int64_t total = (int64_t)N * ny * nx * C;
    GRID_STRIDE(i, total) {
        const int ch = (int)(i % C);
        int64_t p = i / C;
        const int x = (int)(p % nx);
        int64_t q = p / nx;
        const int y = (int)(q % ny);
        const int n = (int)(q / ny);
        const int a = ch / no, o = ch - a * no;
        const float v = to_f32<T>(t[p * tld + ch]);
        const int64_t cell = ((int64_t)a * ny + y) * nx + x;
        raw[(((int64_t)n * na * ny * nx) + cell) * no + o] = v;
        if (pred) {
            const float s = 1.0f / (1.0f + expf(-v));
            float r;
            if (o == 0) r = (s * 2.0f - 0.5f + (float)x) * stride;
            else if (o == 1) r = (s * 2.0f - 0.5f + (float)y) * stride;
            else if (o == 2) { const float u = s * 2.0f; r = u * u * anchors[2 * a]; }
            else if (o == 3) { const float u = s * 2.0f; r = u * u * anchors[2 * a + 1]; }
            else r = s;
            pred[((int64_t)n * pred_rows + row_off + cell) * no + o] = r;
        }
    }
}

template <typename T>
__global__ void detect_raw_bwd_kernel(const float* __restrict__ draw, T* __restrict__ dt, int64_t tld, int N, int ny,
                                      int nx, int na, int no, int Cw) {
    const int C = na * no;
    // channels C .. Cw-1 (row padding the caller asked to clear) become zero
    const int64_t total = (int64_t)N * ny * nx * Cw;
    GRID_STRIDE(i, total) {
        const int ch = (int)(i % Cw);
        int64_t p = i / Cw;
        const int x = (int)(p % nx);
        int64_t q = p / nx;
        const int y = (int)(q % ny);
        const int n = (int)(q / ny);
        const int a = ch / no, o = ch - a * no;
        dt[p * tld + ch] = from_f32<T>(ch < C ? draw[((((int64_t)n * na + a) * ny + y) * nx + x) * no + o] : 0.f);
    }
}

// ---- all Detect levels per launch (yolo.py:258-276 runs the three heads one after the other) ---------------------------------
constexpr int DET_MAXL = 4;
struct DetectLevels {
    const void* t[DET_MAXL];       // head conv outputs (NHWC, na*no channels)
    int64_t tld[DET_MAXL];
    float* raw[DET_MAXL];          // [N][na][ny][nx][no] fp32 (forward: written; backward: the incoming gradient)
    int32_t ny[DET_MAXL], nx[DET_MAXL];
    int64_t row_off[DET_MAXL];     // first row of the level in pred
    float stride[DET_MAXL];
    int32_t cw[DET_MAXL];          // backward: channels written per pixel (>= na*no: zero-filled row padding)
    float* part[DET_MAXL];         // backward: per-block per-channel partial sums [gridDim.x][cw] (bias gradient) or NULL
    float* bias_grad[DET_MAXL];
    int32_t nl, N, na, no;
};

// I: index type -- uint32_t whenever every level's element count fits (64-bit divisions cost ~4x as many instructions, and this
// kernel runs five of them per element)
template <typename T, typename I>
__global__ __launch_bounds__(256) void detect_decode_multi_kernel(const DetectLevels L, float* __restrict__ pred, int64_t pred_rows,
                                                                  const float* __restrict__ anchors) {
    const int l = blockIdx.y;
    const int ny = L.ny[l], nx = L.nx[l], na = L.na, no = L.no, C = na * no;
    const T* __restrict__ t = (const T*)L.t[l];
    float* __restrict__ raw = L.raw[l];
    const int64_t tld = L.tld[l];
    const I total = (I)L.N * (I)(ny * nx) * (I)C, HW = (I)(ny * nx);
    const float stride = L.stride[l];
    for (I i = (I)blockIdx.x * 256 + threadIdx.x; i < total; i += (I)gridDim.x * 256) {
        const I p = i / (I)C;
        const int ch = (int)(i - p * (I)C);
        const I n = p / HW;
        const int cellp = (int)(p - n * HW);
        const int y = cellp / nx, x = cellp - y * nx;
        const int a = ch / no, o = ch - a * no;
        const float v = to_f32<T>(t[(int64_t)p * tld + ch]);
        const int64_t cell = (int64_t)a * (int64_t)HW + cellp;
        raw[(((int64_t)n * na * (int64_t)HW) + cell) * no + o] = v;
        if (pred) {
            const float s = 1.0f / (1.0f + expf(-v));
            float r;
            if (o == 0) r = (s * 2.0f - 0.5f + (float)x) * stride;
            else if (o == 1) r = (s * 2.0f - 0.5f + (float)y) * stride;
            else if (o == 2) { const float u = s * 2.0f; r = u * u * anchors[(l * na + a) * 2]; }
            else if (o == 3) { const float u = s * 2.0f; r = u * u * anchors[(l * na + a) * 2 + 1]; }
            else r = s;
            pred[((int64_t)n * pred_rows + L.row_off[l] + cell) * no + o] = r;
        }
    }
}

// backward of the permute for every level + the heads' bias gradients: a thread keeps ONE channel (the grid stride is a
// multiple of the row width); the block folds its threads' sums in LDS and writes ONE partial row, a finalize kernel adds the
// rows in order (an atomic per thread on the channel's accumulator serialised 6400 deep: 0.25 ms)
template <typename T>
__global__ __launch_bounds__(256) void detect_raw_bwd_multi_kernel(const DetectLevels L) {
    const int l = blockIdx.y;
    const int ny = L.ny[l], nx = L.nx[l], na = L.na, no = L.no, C = na * no, Cw = L.cw[l];
    const float* __restrict__ draw = L.raw[l];
    T* __restrict__ dt = (T*)L.t[l];
    const int64_t tld = L.tld[l];
    __shared__ float sh[64];
    if (threadIdx.x < 64) sh[threadIdx.x] = 0.f;
    __syncthreads();
    // 32-bit index arithmetic, one division per element (the host checks N * ny * nx < 2^31): the 64-bit i / Cw, p % nx, q % ny,
    // q / ny chain this loop used to run per element cost more than the memory traffic
    const unsigned i0 = blockIdx.x * 256u + threadIdx.x;
    const unsigned pstep = gridDim.x * 256u / (unsigned)Cw;           // whole pixels per sweep: the channel of a thread is fixed
    const unsigned pfirst = i0 / (unsigned)Cw;
    const int ch = (int)(i0 - pfirst * (unsigned)Cw);
    const int a = ch / no, o = ch - a * no;
    const unsigned HW = (unsigned)(ny * nx), P = (unsigned)L.N * HW;
    float s = 0.f;
    if (pfirst < pstep)
        for (unsigned p = pfirst; p < P; p += pstep) {
            const unsigned n = p / HW, q = p - n * HW;
            const float v = ch < C ? draw[(((int64_t)n * na + a) * HW + q) * no + o] : 0.f;
            dt[(int64_t)p * tld + ch] = from_f32<T>(v);
            s += v;
        }
    if (L.part[l]) {
        if (ch < C && pfirst < pstep) atomicAdd(&sh[ch], s);
        __syncthreads();
        if ((int)threadIdx.x < C) L.part[l][(int64_t)blockIdx.x * C + threadIdx.x] = sh[threadIdx.x];
    }
}

// bias_grad[l][c] += sum over blocks of part[l][block][c] (fixed order; one wave per channel)
__global__ __launch_bounds__(64) void detect_bias_finalize_kernel(const DetectLevels L, int nblocks) {
    const int l = blockIdx.y, c = blockIdx.x, C = L.na * L.no;
    if (c >= C) return;
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 64) s += (double)L.part[l][(int64_t)b * C + c];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if (threadIdx.x == 0) L.bias_grad[l][c] += (float)s;
}

// ---- NMS stage 1: candidates ----------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t make_key(float conf, uint32_t idx) {
    return ((uint64_t)(~__float_as_uint(conf)) << 32) | idx;
}

// One returning atomic per 256-thread BLOCK, not per candidate: the lanes that keep a candidate are counted with a ballot, the four
// waves' counts meet in LDS, ONE lane reserves the block's slots and every lane takes base + (kept lanes before it).  (One atomicAdd
// per candidate on the image's counter serialised at ~100 ns each: 2.6 ms for 16 x 25200 rows when every row survives the
// threshold; one per wave, 6300 of them, still cost 77 us.)  Uniform call sites only; blockIdx.y = image.
__device__ __forceinline__ int block_slot(bool want, int32_t* counter, int* s_cnt) {
    const uint64_t mask = __ballot(want);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_cnt[wave] = __popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        const int total = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        s_cnt[4] = total ? atomicAdd(counter, total) : 0;
    }
    __syncthreads();
    int base = s_cnt[4];
    for (int w = 0; w < wave; ++w) base += s_cnt[w];
    const int slot = want ? base + __popcll(mask & ((1ull << lane) - 1ull)) : -1;
    __syncthreads();
    return slot;
}

__global__ __launch_bounds__(256) void nms_candidates_kernel(const float* __restrict__ pred, int bs, int n, int nc,
                                                             float conf_thres, int multi_label, uint64_t classes_mask,
                                                             uint64_t* __restrict__ keys, int64_t cap,
                                                             int32_t* __restrict__ counts) {
    __shared__ int s_cnt[5];
    const int no = 5 + nc;
    const int b = blockIdx.y;
    uint64_t* kb = keys + (int64_t)b * cap;
    for (int base = blockIdx.x * 256; base < n; base += gridDim.x * 256) {       // uniform trip count: ballots see whole waves
        const int row = base + (int)threadIdx.x;
        const bool in = row < n;
        const float* r = pred + ((int64_t)b * n + (in ? row : 0)) * no;
        const float obj = in ? r[4] : 0.f;
        const bool live = in && (obj > conf_thres);
        if (multi_label) {
            for (int j = 0; j < nc; ++j) {
                const float conf = live ? r[5 + j] * obj : 0.f;
                const bool want = live && conf > conf_thres && (classes_mask == 0 || ((classes_mask >> j) & 1));
                const int slot = block_slot(want, &counts[b], s_cnt);
                if (want) kb[slot] = make_key(conf, (uint32_t)(row * nc + j));
            }
        } else {
            float best = 0.f;
            int bj = 0;
            if (live) {
                best = r[5] * obj;
                for (int j = 1; j < nc; ++j) {
                    const float conf = r[5 + j] * obj;
                    if (conf > best) { best = conf; bj = j; }
                }
            }
            const bool want = live && best > conf_thres && (classes_mask == 0 || ((classes_mask >> bj) & 1));
            const int slot = block_slot(want, &counts[b], s_cnt);
            if (want) kb[slot] = make_key(best, (uint32_t)(row * nc + bj));
        }
    }
}

// ---- NMS stage 2: sort of the candidate keys (ascending = confidence descending, candidate order on ties) ---------------------------
// All hand-written (round 4; rounds 2-3 sent more than 8192 slots per image through rocPRIM's device sort: 18 launches, 129 us for
// 16 x 25200 candidates).  Keys are unique, so every correct sort gives the same order: the selection stays bit-exact.
//   cnt <= 8192          one workgroup per image: bitonic network in 64 KB of LDS (nms_tile_sort_kernel, tile 0)
//   8192 < cnt <= 32768  the 32768-key bitonic network over FOUR 8192-key tiles per image, all of its compare-exchange phases of
//                        distance < 8192 inside LDS; only three phases cross tiles (distance 8192 of the 16384-merge, distances
//                        16384 and 8192 of the 32768-merge), and a tile computes its side of those straight from the source buffer
//                        (the second one needs the partner's value AFTER the first: recomputed from four source keys), so the
//                        whole sort is three launches of 4 x images workgroups, ping-ponging between two key buffers:
//                        nms_tile_sort_kernel (in place) -> nms_merge16k_kernel (A -> B) -> nms_merge32k_kernel (B -> A).
//                        Keys past cnt read as ~0 (they sort to the end), so nothing is padded in memory.
//   cnt > 32768          (validation settings: conf 0.001 + multi_label, up to 151200 candidates) only the max_nms = 30000 best
//                        are ever used (general.py:707-708): nms_select_kernel finds the 30000th smallest key by an 8-pass radix
//                        SELECT (256-bin LDS histograms, one workgroup per image), compacts the keys <= it in place and sets the
//                        image's count to 30000; then as above.
__device__ __forceinline__ void cmpswap(uint64_t& a, uint64_t& b, bool up) {
    if ((a > b) == up) { const uint64_t t = a; a = b; b = t; }
}
// Compare-exchange phases of distance jstart .. 1 of the merge of size `size` inside an LDS tile of n keys whose first key is global
// index `base`.  The network is bound by LDS traffic (128 KB per phase of an 8192-key tile), so THREE consecutive phases (distances
// j, j/2, j/4) are done per LDS round trip: a thread takes the 8 keys lo + r j/4 (r = 0 .. 7) of one 2j-key block -- every partner of
// those three phases is among them -- into registers; 13 phases of the 8192-merge are 5 round trips.  Leftover phases: two (4 keys)
// or one at a time.
template <int NK> __device__ __forceinline__ void phases_in_regs(uint64_t* s, int tid, int n, int size, int j, int base) {
    // NK = 8: distances j, j/2, j/4; NK = 4: j, j/2; NK = 2: j.  Key r of a thread lies at lo + r * (2j / NK).
    const int step = (2 * j) / NK, ls = 31 - __clz(step);      // (powers of two: no integer division in the loop)
    for (int t = tid; t < n / NK; t += SORT_THREADS) {
        const int lo = ((t >> ls) * (2 * j)) | (t & (step - 1));
        const bool up = ((base + lo) & size) == 0;
        uint64_t v[NK];
#pragma unroll
        for (int r = 0; r < NK; ++r) v[r] = s[skew(lo + r * step)];
#pragma unroll
        for (int d = NK / 2; d > 0; d >>= 1)
#pragma unroll
            for (int r = 0; r < NK; ++r)
                if ((r & d) == 0) cmpswap(v[r], v[r | d], up);
#pragma unroll
        for (int r = 0; r < NK; ++r) s[skew(lo + r * step)] = v[r];
    }
    __syncthreads();
}
__device__ __forceinline__ void tile_phases(uint64_t* s, int tid, int n, int size, int jstart, int base) {
    int j = jstart;
    while (j >= 4) { phases_in_regs<8>(s, tid, n, size, j, base); j >>= 3; }
    if (j == 2) phases_in_regs<4>(s, tid, n, size, 2, base);
    else if (j == 1) phases_in_regs<2>(s, tid, n, size, 1, base);
}

__global__ __launch_bounds__(SORT_THREADS) void nms_tile_sort_kernel(uint64_t* __restrict__ keys, int64_t cap,
                                                                     const int32_t* __restrict__ counts) {
    __shared__ __attribute__((aligned(16))) uint64_t s[SORT_LDS_SLOTS];
    const int b = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
    uint64_t* k = keys + (int64_t)b * cap;
    const int cnt = counts[b] < cap ? counts[b] : (int)cap;
    if (cnt <= 1 || (cnt <= SORT_LDS_KEYS && tile > 0)) return;
    int np = SORT_LDS_KEYS;                         // keys this tile sorts: the whole (power-of-two padded) list when it fits
    if (cnt <= SORT_LDS_KEYS) { np = 1; while (np < cnt) np <<= 1; }
    const int base = tile * SORT_LDS_KEYS;
    for (int i = tid; i < np; i += SORT_THREADS) s[skew(i)] = base + i < cnt ? k[base + i] : ~0ull;
    __syncthreads();
    for (int size = 2; size <= np; size <<= 1) tile_phases(s, tid, np, size, size >> 1, base);
    const int nst = cnt <= SORT_LDS_KEYS ? cnt : np;       // (the tiles of a larger list are stored whole: the merges read them)
    for (int i = tid; i < nst; i += SORT_THREADS) k[base + i] = s[skew(i)];
}

// merge of size 16384 (tile pairs): phase of distance 8192 from the source buffer, the rest in LDS.  src -> dst.
__global__ __launch_bounds__(SORT_THREADS) void nms_merge16k_kernel(const uint64_t* __restrict__ src, uint64_t* __restrict__ dst,
                                                                    int64_t cap, const int32_t* __restrict__ counts) {
    __shared__ __attribute__((aligned(16))) uint64_t s[SORT_LDS_SLOTS];
    const int b = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
    if (counts[b] <= SORT_LDS_KEYS) return;
    const uint64_t* k = src + (int64_t)b * cap;
    const int base = tile * SORT_LDS_KEYS;
    for (int i = tid; i < SORT_LDS_KEYS; i += SORT_THREADS) {
        const int g = base + i;
        const uint64_t a = k[g], p = k[g ^ SORT_LDS_KEYS];
        const bool take_min = ((g & SORT_LDS_KEYS) == 0) == ((g & (2 * SORT_LDS_KEYS)) == 0);     // lower half of an ascending pair
        s[skew(i)] = take_min ? (a < p ? a : p) : (a > p ? a : p);
    }
    __syncthreads();
    tile_phases(s, tid, SORT_LDS_KEYS, 2 * SORT_LDS_KEYS, SORT_LDS_KEYS >> 1, base);
    uint64_t* o = dst + (int64_t)b * cap;
    for (int i = tid; i < SORT_LDS_KEYS; i += SORT_THREADS) o[base + i] = s[skew(i)];
}

// merge of size 32768 (all four tiles, ascending): phases of distance 16384 and 8192 from the source buffer, the rest in LDS.
__global__ __launch_bounds__(SORT_THREADS) void nms_merge32k_kernel(const uint64_t* __restrict__ src, uint64_t* __restrict__ dst,
                                                                    int64_t cap, const int32_t* __restrict__ counts) {
    __shared__ __attribute__((aligned(16))) uint64_t s[SORT_LDS_SLOTS];
    const int b = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
    if (counts[b] <= 2 * SORT_LDS_KEYS) return;
    const uint64_t* k = src + (int64_t)b * cap;
    const int base = tile * SORT_LDS_KEYS;
    for (int i = tid; i < SORT_LDS_KEYS; i += SORT_THREADS) {
        const int g = base + i, h = g ^ SORT_LDS_KEYS;
        const uint64_t a0 = k[g], a1 = k[g ^ (2 * SORT_LDS_KEYS)], b0 = k[h], b1 = k[h ^ (2 * SORT_LDS_KEYS)];
        const bool low16 = (g & (2 * SORT_LDS_KEYS)) == 0;                    // (the same for g and its distance-8192 partner h)
        const uint64_t x = low16 ? (a0 < a1 ? a0 : a1) : (a0 > a1 ? a0 : a1);  // this key after the distance-16384 phase
        const uint64_t y = low16 ? (b0 < b1 ? b0 : b1) : (b0 > b1 ? b0 : b1);  // its partner after that phase
        s[skew(i)] = (g & SORT_LDS_KEYS) == 0 ? (x < y ? x : y) : (x > y ? x : y);
    }
    __syncthreads();
    tile_phases(s, tid, SORT_LDS_KEYS, 4 * SORT_LDS_KEYS, SORT_LDS_KEYS >> 1, base);
    uint64_t* o = dst + (int64_t)b * cap;
    for (int i = tid; i < SORT_LDS_KEYS; i += SORT_THREADS) o[base + i] = s[skew(i)];
}

// More than 32768 candidates: keep the MAX_NMS smallest keys (radix select on the 64-bit key, most significant byte first), in place.
__global__ __launch_bounds__(SORT_THREADS) void nms_select_kernel(uint64_t* __restrict__ keys, int64_t cap, int32_t* __restrict__ counts) {
    __shared__ unsigned hist[256];
    __shared__ unsigned long long prefix_s;
    __shared__ int want_s, base_s, wcnt[SORT_THREADS / 64];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cnt = counts[b] < cap ? counts[b] : (int)cap;
    if (cnt <= 4 * SORT_LDS_KEYS) return;
    uint64_t* k = keys + (int64_t)b * cap;
    if (tid == 0) { prefix_s = 0ull; want_s = MAX_NMS; }
    for (int pass = 0; pass < 8; ++pass) {
        const int shift = 56 - 8 * pass;
        if (tid < 256) hist[tid] = 0u;
        __syncthreads();
        const unsigned long long prefix = prefix_s;
        for (int i = tid; i < cnt; i += SORT_THREADS) {
            const uint64_t v = k[i];
            if (pass == 0 || (v >> (shift + 8)) == prefix) atomicAdd(&hist[(unsigned)(v >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {             // the digit in which the want-th smallest key of this prefix lies
            int want = want_s, d = 0;
            for (; d < 255; ++d) {
                if ((int)hist[d] >= want) break;
                want -= (int)hist[d];
            }
            want_s = want;
            prefix_s = (prefix << 8) | (unsigned long long)d;
        }
        __syncthreads();
    }
    const uint64_t kth = prefix_s;              // exactly MAX_NMS keys are <= kth (keys are unique)
    if (tid == 0) base_s = 0;
    __syncthreads();
    // in-place compaction, chunk by chunk in order: the write position never passes the read position
    for (int c0 = 0; c0 < cnt; c0 += SORT_THREADS) {
        const int i = c0 + tid;
        const uint64_t v = i < cnt ? k[i] : ~0ull;
        const bool keep = i < cnt && v <= kth;
        const unsigned long long m = __ballot(keep);
        if (lane == 0) wcnt[wave] = __popcll(m);
        __syncthreads();
        int off = base_s;
        for (int w = 0; w < wave; ++w) off += wcnt[w];
        if (keep) k[off + __popcll(m & ((1ull << lane) - 1ull))] = v;
        __syncthreads();
        if (tid == 0) {
            int t = 0;
            for (int w = 0; w < SORT_THREADS / 64; ++w) t += wcnt[w];
            base_s += t;
        }
        __syncthreads();
    }
    if (tid == 0) counts[b] = MAX_NMS;
}

// ---- NMS stage 3: greedy suppression ----------------------------------------------------------------------------------
struct Box { float x1, y1, x2, y2, area; };

__device__ __forceinline__ bool iou_gt(const Box& a, const Box& b, float thr) {
    const float xx1 = fmaxf(a.x1, b.x1), yy1 = fmaxf(a.y1, b.y1);
    const float xx2 = fminf(a.x2, b.x2), yy2 = fminf(a.y2, b.y2);
    const float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
    const float inter = w * h;
    // inter == 0: ovr is 0 (or NaN for two empty boxes), never > thr >= 0 -- the common case skips the IEEE division
    if (!(inter > 0.f)) return false;
    const float ovr = inter / (a.area + b.area - inter);
    return ovr > thr;
}

// Stage 3a (all CUs): the sorted candidates of every image as rows [x1, y1, x2, y2, conf, cls] (general.py:688-699: xywh2xyxy, conf =
// obj * cls) -- the key -> row -> prediction chain is two dependent trips to memory, which the serial sweep below used to pay once
// per 64 candidates.
constexpr int CAND_W = 6;
__global__ __launch_bounds__(256) void nms_gather_kernel(const float* __restrict__ pred, int n, int nc,
                                                         const uint64_t* __restrict__ keys, const uint64_t* __restrict__ keys_b,
                                                         int64_t cap, const int32_t* __restrict__ counts,
                                                         float* __restrict__ cand, int64_t cand_cap) {
    const int b = blockIdx.y;
    const int no = 5 + nc;
    int cnt = counts[b];
    // where the sorted keys lie: the 16384-merge leaves them in the second buffer, everything else in the first (stage 2)
    const uint64_t* k = ((cnt > SORT_LDS_KEYS && cnt <= 2 * SORT_LDS_KEYS) ? keys_b : keys) + (int64_t)b * cap;
    if (cnt > MAX_NMS) cnt = MAX_NMS;
    float* cb = cand + (int64_t)b * cand_cap * CAND_W;
    for (int ci = blockIdx.x * 256 + threadIdx.x; ci < cnt; ci += gridDim.x * 256) {
        const uint32_t idx = (uint32_t)(k[ci] & 0xFFFFFFFFu);
        const int row = idx / nc, cls = idx - row * nc;
        const float* r = pred + ((int64_t)b * n + row) * no;
        const float hw = r[2] / 2.0f, hh = r[3] / 2.0f;
        float* o = cb + (int64_t)ci * CAND_W;
        o[0] = r[0] - hw; o[1] = r[1] - hh; o[2] = r[0] + hw; o[3] = r[1] + hh;
        o[4] = r[5 + cls] * r[4];
        o[5] = (float)cls;
    }
}

// Stage 3b: greedy suppression, one workgroup of NW waves per image, 64 candidates (one per lane, every wave holds a copy) per
// round.  Per round: (a) the waves split the list of boxes kept so far (max_det 1000 at detect.py's settings: 63 IoU evaluations
// per wave with 16 waves); (b) the 64 x 64 comparisons INSIDE the chunk are split NW ways as well (4 earlier candidates per wave:
// one wave doing all 64 was the longest part of a round, 3 us); (c) wave 0 sweeps the chunk with the two bit masks, jumping from
// kept box to kept box (s_ff1) instead of visiting all 64 positions.  The next chunk's rows are requested before (a) and the
// barriers wait for LDS only, so no round waits on memory.  Same comparisons and the same order of decisions as the serial
// reference (torchvision.ops.nms semantics, general.py:714).
constexpr int GREEDY_WAVES = 16;
__global__ __launch_bounds__(GREEDY_WAVES * 64) void nms_greedy_kernel(const float* __restrict__ cand, int64_t cand_cap,
                                                                       const int32_t* __restrict__ counts, float iou_thres,
                                                                       int agnostic, int max_det, float* __restrict__ out,
                                                                       int32_t* __restrict__ out_count) {
    static_assert(GREEDY_WAVES == 16, "4 in-chunk candidates per wave, 16 nibbles per lane");
    extern __shared__ __attribute__((aligned(16))) float kept[];   // [max_det rounded up to 4][5]
    __shared__ unsigned long long dead_s[2][GREEDY_WAVES];
    __shared__ __attribute__((aligned(16))) unsigned char sup_s[2][64][GREEDY_WAVES];      // [buffer][lane][wave]: 4 bits each
    __shared__ int nkept_s;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, part = tid >> 6;
    int cnt = counts[b];
    if (cnt > MAX_NMS) cnt = MAX_NMS;
    const float* cb = cand + (int64_t)b * cand_cap * CAND_W;
    if (tid == 0) nkept_s = 0;
    __syncthreads();
    float* ob = out + (int64_t)b * max_det * 6;
    auto fetch = [&](int base, float (&raw)[6]) {
        const int ci = base + lane;
#pragma unroll
        for (int e = 0; e < 6; ++e) raw[e] = 0.f;
        if (ci < cnt) {
            const float2* r = reinterpret_cast<const float2*>(cb + (int64_t)ci * CAND_W);
            const float2 v0 = r[0], v1 = r[1], v2 = r[2];
            raw[0] = v0.x; raw[1] = v0.y; raw[2] = v1.x; raw[3] = v1.y; raw[4] = v2.x; raw[5] = v2.y;
        }
    };
    auto lds_barrier = [&]() {           // (workgroup traffic is LDS only: leave the prefetch in flight)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    auto box_of = [&](const float (&r)[6], bool valid) {
        Box me{0.f, 0.f, 0.f, 0.f, 0.f};
        if (valid) {
            const float off = r[5] * (agnostic ? 0.f : (float)MAX_WH);
            me.x1 = r[0] + off; me.y1 = r[1] + off; me.x2 = r[2] + off; me.y2 = r[3] + off;
            me.area = (me.x2 - me.x1) * (me.y2 - me.y1);
        }
        return me;
    };
    // my candidate against kept boxes [q_begin, q_end): this wave is number `w` of `nw` that share them, four boxes per step
    // (twenty consecutive floats = five broadcast ds_read_b128; the first step starts at a multiple of four and masks the front)
    auto vs_kept = [&](const Box& me, int q_begin, int q_end, int w, int nw, bool dead) {
        for (int q0 = (q_begin & ~3) + 4 * w; q0 < q_end; q0 += 4 * nw) {
            float kb[20];
#pragma unroll
            for (int v = 0; v < 5; ++v) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(&kept[q0 * 5 + 4 * v]);
                kb[4 * v] = t[0]; kb[4 * v + 1] = t[1]; kb[4 * v + 2] = t[2]; kb[4 * v + 3] = t[3];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (q0 + u >= q_begin && q0 + u < q_end &&
                    iou_gt(Box{kb[5 * u], kb[5 * u + 1], kb[5 * u + 2], kb[5 * u + 3], kb[5 * u + 4]}, me, iou_thres))
                    dead = true;
        }
        return dead;
    };
    // inside a chunk: bit jj = candidate 4 * part + jj (earlier, higher score) overlaps me
    auto in_chunk = [&](const Box& me) {
        unsigned bits = 0u;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int j = part * 4 + jj;
            Box o;
            o.x1 = __shfl(me.x1, j); o.y1 = __shfl(me.y1, j); o.x2 = __shfl(me.x2, j); o.y2 = __shfl(me.y2, j);
            o.area = __shfl(me.area, j);
            if (j < lane && iou_gt(o, me, iou_thres)) bits |= 1u << jj;
        }
        return bits;
    };
    // PIPELINE (round 4): while wave 0 sweeps chunk c (serial: up to 64 decisions), the other fifteen waves already test chunk c + 1
    // against the boxes kept BEFORE chunk c and among themselves; after the sweep all waves test chunk c + 1 against the (<= 64)
    // boxes chunk c added.  Same comparisons and the same order of decisions as the serial reference (torchvision.ops.nms semantics,
    // general.py:714): a candidate is dropped iff it overlaps a box kept earlier.
    float raw[6], nxt[6], nn[6];
    fetch(0, raw);
    fetch(64, nxt);
    int cur = 0;
    {   // chunk 0: nothing kept yet
        const Box me0 = box_of(raw, lane < cnt);
        const unsigned long long dm = __ballot(!(lane < cnt));
        if (lane == 0) dead_s[0][part] = dm;
        sup_s[0][lane][part] = (unsigned char)in_chunk(me0);
    }
    for (int base = 0; base < cnt; base += 64) {
        lds_barrier();                                    // chunk `base` is prepared (dead_s / sup_s [cur]); nkept_s is final
        const int nk0 = nkept_s;
        if (nk0 >= max_det) break;
        fetch(base + 128, nn);
        const Box me2 = box_of(nxt, base + 64 + lane < cnt);
        bool dead2 = !(base + 64 + lane < cnt);
        sup_s[cur ^ 1][lane][part] = (unsigned char)in_chunk(me2);
        if (part == 0) {
            const Box me = box_of(raw, base + lane < cnt);
            unsigned long long dm_all = 0ull;
#pragma unroll
            for (int q = 0; q < GREEDY_WAVES; ++q) dm_all |= dead_s[cur][q];
            const u32x4 sv = *reinterpret_cast<const u32x4*>(&sup_s[cur][lane][0]);
            unsigned long long sup = 0ull;
#pragma unroll
            for (int q = 0; q < GREEDY_WAVES; ++q)
                sup |= (unsigned long long)((sv[q >> 2] >> (8 * (q & 3))) & 15u) << (4 * q);
            const unsigned long long a0 = ~dm_all;
            // (readfirstlane returns int: widen through unsigned, or bit 31 of the low word smears over the high one)
            unsigned long long alive = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(a0 >> 32)) << 32) |
                                       (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a0);
            // the serial part only decides (scalar mask arithmetic + one ballot per kept box); the kept boxes are written afterwards
            // by their own lanes, in lane order = score order
            int nk = nk0;
            unsigned long long keepm = 0ull;
            while (alive != 0ull && nk < max_det) {          // wave-uniform
                const int j = __ffsll((long long)alive) - 1;
                keepm |= 1ull << j;
                ++nk;
                alive &= ~(1ull << j);
                alive &= ~__ballot((sup >> j) & 1ull);
            }
            if ((keepm >> lane) & 1ull) {
                const int q = nk0 + __popcll(keepm & ((1ull << lane) - 1ull));
                kept[q * 5] = me.x1; kept[q * 5 + 1] = me.y1; kept[q * 5 + 2] = me.x2; kept[q * 5 + 3] = me.y2;
                kept[q * 5 + 4] = me.area;
#pragma unroll
                for (int e = 0; e < 6; ++e) ob[q * 6 + e] = raw[e];
            }
            if (lane == 0) nkept_s = nk;
        } else {
            dead2 = vs_kept(me2, 0, nk0, part - 1, GREEDY_WAVES - 1, dead2);      // against everything kept before this chunk
        }
        lds_barrier();                                    // the sweep of chunk `base` is done
        const int nk1 = nkept_s;
        dead2 = vs_kept(me2, nk0, nk1, part, GREEDY_WAVES, dead2);                // against what this chunk added
        const unsigned long long dm2 = __ballot(dead2);
        if (lane == 0) dead_s[cur ^ 1][part] = dm2;
        cur ^= 1;
#pragma unroll
        for (int e = 0; e < 6; ++e) { raw[e] = nxt[e]; nxt[e] = nn[e]; }
    }
    __syncthreads();
    if (tid == 0) out_count[b] = nkept_s;
}

inline int ew_grid(int64_t total) {
    int64_t b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}
inline int64_t next_pow2(int64_t v) {
    int64_t p = 1;
    while (p < v) p <<= 1;
    return p;
}
inline int64_t key_cap(int32_t n, int32_t nc, int32_t multi_label) {
    return next_pow2((int64_t)n * (multi_label ? nc : 1));
}


// ---- evaluation arithmetic (scripts/val.py:101-122, core/utils/metrics.py:247-269,350-388) -----------------------------------
// IoU matrix of xyxy boxes, same operation order as the reference's box_iou (this file is compiled with -ffp-contract=off):
// area = (x2-x1)*(y2-y1); inter = clamp(min(x2)-max(x1), 0) * clamp(min(y2)-max(y1), 0); iou = inter / (a1 + a2 - inter)
__global__ void box_iou_kernel(const float* __restrict__ b1, const float* __restrict__ b2, float* __restrict__ out, int N, int M) {
    const int64_t total = (int64_t)N * M;
    GRID_STRIDE(i, total) {
        const int n = (int)(i / M), m = (int)(i - (int64_t)n * M);
        const float ax1 = b1[n * 4], ay1 = b1[n * 4 + 1], ax2 = b1[n * 4 + 2], ay2 = b1[n * 4 + 3];
        const float bx1 = b2[m * 4], by1 = b2[m * 4 + 1], bx2 = b2[m * 4 + 2], by2 = b2[m * 4 + 3];
        const float a1 = (ax2 - ax1) * (ay2 - ay1), a2 = (bx2 - bx1) * (by2 - by1);
        const float w = fmaxf(fminf(ax2, bx2) - fmaxf(ax1, bx1), 0.f), h = fmaxf(fminf(ay2, by2) - fmaxf(ay1, by1), 0.f);
        const float inter = w * h;
        out[i] = inter / (a1 + a2 - inter);
    }
}

// Segmentation counters over a batch of NCHW logits: predict = first arg-max over the class axis; out (int64):
// [0] pixels with target > 0 and predict == target, [1] pixels with target > 0, then for the nclass-1 bins of
// np.histogram(range=(1, nclass)) -- value v in [1, nclass], v == nclass falling into the last bin -- [2 + b] intersection
// (predict == target == v), [2 + nb + b] prediction area, [2 + 2nb + b] label area.
constexpr int SEG_MAXBINS = 64;
__global__ __launch_bounds__(256) void seg_eval_counts_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target,
                                                              int C, int64_t HW, int64_t total, int nclass,
                                                              unsigned long long* __restrict__ out) {
    __shared__ unsigned int h[2 + 3 * SEG_MAXBINS];
    const int nb = nclass - 1;
    for (int i = threadIdx.x; i < 2 + 3 * nb; i += 256) h[i] = 0u;
    __syncthreads();
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t n = i / HW, px = i - n * HW;
        const float* l = logits + n * C * HW + px;
        float best = l[0];
        int pred = 0;
        for (int c = 1; c < C; ++c) {
            const float v = l[c * HW];
            if (v > best) { best = v; pred = c; }
        }
        const int64_t t = target[i];
        if (t > 0) {
            atomicAdd(&h[1], 1u);
            if (pred == t) atomicAdd(&h[0], 1u);
        }
        if (nb > 0) {
            if (pred >= 1 && pred <= nclass) {
                const int b = pred - 1 < nb ? pred - 1 : nb - 1;
                atomicAdd(&h[2 + nb + b], 1u);
                if (pred == t) atomicAdd(&h[2 + b], 1u);
            }
            if (t >= 1 && t <= nclass) {
                const int b = (int)(t - 1 < nb ? t - 1 : nb - 1);
                atomicAdd(&h[2 + 2 * nb + b], 1u);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 + 3 * nb; i += 256)
        if (h[i]) atomicAdd(&out[i], (unsigned long long)h[i]);
}

// ---- evaluation resampling of the NCHW fp32 seg logits ---------------------------------------------------------------------
// F.interpolate(pred, size, mode='bilinear', align_corners=False) (val.py:47), ATen's arithmetic: scale = in / out (fp32),
// src = max(scale * (dst + 0.5) - 0.5, 0), i0 = min(floor(src), in - 1), i1 = i0 + (i0 < in - 1), l = src - i0, and
// out = (1 - ly) * ((1 - lx) * v00 + lx * v01) + ly * ((1 - lx) * v10 + lx * v11).  align_corners != 0: src = dst * (in-1)/(out-1).
__global__ __launch_bounds__(256) void resize_bilinear_nchw_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                   int64_t planes, int hi, int wi, int ho, int wo, float ry,
                                                                   float rx, int ac) {
    const int64_t total = planes * ho * wo;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ox = (int)(i % wo);
        const int64_t t = i / wo;
        const int oy = (int)(t % ho);
        const int64_t pl = t / ho;
        float sy = ac ? ry * oy : ry * (oy + 0.5f) - 0.5f;
        float sx = ac ? rx * ox : rx * (ox + 0.5f) - 0.5f;
        if (!ac) { sy = sy < 0.f ? 0.f : sy; sx = sx < 0.f ? 0.f : sx; }
        int y0 = (int)floorf(sy), x0 = (int)floorf(sx);
        y0 = y0 < hi - 1 ? y0 : hi - 1;
        x0 = x0 < wi - 1 ? x0 : wi - 1;
        const int y1 = y0 + (y0 < hi - 1 ? 1 : 0), x1 = x0 + (x0 < wi - 1 ? 1 : 0);
        float ly = sy - y0, lx = sx - x0;
        ly = ly < 0.f ? 0.f : (ly > 1.f ? 1.f : ly);
        lx = lx < 0.f ? 0.f : (lx > 1.f ? 1.f : lx);
        const float* p = x + pl * hi * wi;
        const float top = (1.f - lx) * p[(int64_t)y0 * wi + x0] + lx * p[(int64_t)y0 * wi + x1];
        const float bot = (1.f - lx) * p[(int64_t)y1 * wi + x0] + lx * p[(int64_t)y1 * wi + x1];
        y[i] = (1.f - ly) * top + ly * bot;
    }
}

// segoutput_to_target (plots.py:222-229): out[n][oy][ox] = (float) first arg-max over classes of logits[n][:][sy][sx] with
// ATen's legacy 'nearest' source index: dst if out == in, dst >> 1 if out == 2*in, else min(floor(dst * (float)in / out), in-1).
__device__ __forceinline__ int nearest_src(int dst, int in, int out, float scale) {
    if (out == in) return dst;
    if (out == 2 * in) return dst >> 1;
    const int s = (int)floorf(dst * scale);
    return s < in - 1 ? s : in - 1;
}
__global__ __launch_bounds__(256) void seg_argmax_nearest_kernel(const float* __restrict__ logits, float* __restrict__ out,
                                                                 int N, int C, int H, int W, int ho, int wo, float sy,
                                                                 float sx) {
    const int64_t total = (int64_t)N * ho * wo, HW = (int64_t)H * W;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ox = (int)(i % wo);
        const int64_t t = i / wo;
        const int oy = (int)(t % ho);
        const int64_t n = t / ho;
        const float* l = logits + n * C * HW + (int64_t)nearest_src(oy, H, ho, sy) * W + nearest_src(ox, W, wo, sx);
        float best = l[0];
        int pred = 0;
        for (int c = 1; c < C; ++c) {
            const float v = l[c * HW];
            if (v > best) { best = v; pred = c; }
        }
        out[i] = (float)pred;
    }
}

}  // namespace

extern "C" int dsn_resize_bilinear_nchw(const float* x, float* y, int64_t planes, int32_t hi, int32_t wi, int32_t ho,
                                        int32_t wo, int32_t align_corners, void* stream) {
    DSN_CHECK_ARG(x && y && planes > 0 && hi > 0 && wi > 0 && ho > 0 && wo > 0, "resize_bilinear_nchw: bad args");
    const float ry = align_corners ? (ho > 1 ? (float)(hi - 1) / (float)(ho - 1) : 0.f) : (float)hi / (float)ho;
    const float rx = align_corners ? (wo > 1 ? (float)(wi - 1) / (float)(wo - 1) : 0.f) : (float)wi / (float)wo;
    const int64_t total = planes * ho * wo;
    hipLaunchKernelGGL(resize_bilinear_nchw_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x, y, planes, hi,
                       wi, ho, wo, ry, rx, align_corners);
    DSN_LAUNCH_CHECK("resize_bilinear_nchw");
    return DSN_OK;
}

extern "C" int dsn_seg_argmax_nearest(const float* logits, float* out, int32_t n, int32_t c, int32_t h, int32_t w, int32_t ho,
                                      int32_t wo, void* stream) {
    DSN_CHECK_ARG(logits && out && n > 0 && c > 0 && h > 0 && w > 0 && ho > 0 && wo > 0, "seg_argmax_nearest: bad args");
    const int64_t total = (int64_t)n * ho * wo;
    hipLaunchKernelGGL(seg_argmax_nearest_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, logits, out, n, c, h,
                       w, ho, wo, (float)h / (float)ho, (float)w / (float)wo);
    DSN_LAUNCH_CHECK("seg_argmax_nearest");
    return DSN_OK;
}

extern "C" int dsn_detect_decode(const dsn_tensor* t, float* raw, float* pred, int64_t pred_rows, int64_t row_off,
                                 int32_t na, int32_t no, float stride, const float* anchors_px, void* stream) {
    DSN_CHECK_ARG(tensor_ok(t) && raw && na > 0 && no > 5 && t->c == na * no, "detect_decode: invalid arguments");
    DSN_CHECK_ARG(!pred || (anchors_px && row_off >= 0 && row_off + (int64_t)na * t->h * t->w <= pred_rows),
                  "detect_decode: pred slice out of range");
    const int64_t total = npix(t) * t->c;
    DSN_DISPATCH_DTYPE(t->dtype, T,
                       hipLaunchKernelGGL(detect_decode_kernel<T>, dim3(ew_grid(total)), dim3(256), 0,
                                          (hipStream_t)stream, (const T*)t->ptr, t->ldc, raw, pred, pred_rows, row_off,
                                          t->n, t->h, t->w, na, no, stride, anchors_px));
    DSN_LAUNCH_CHECK("detect_decode");
    return DSN_OK;
}

extern "C" int dsn_detect_raw_bwd(const float* draw, const dsn_tensor* dt, int32_t na, int32_t no, int32_t zero_pad_to,
                                  void* stream) {
    DSN_CHECK_ARG(tensor_ok(dt) && draw && na > 0 && no > 0 && dt->c == na * no, "detect_raw_bwd: invalid arguments");
    DSN_CHECK_ARG(zero_pad_to == 0 || (zero_pad_to >= dt->c && zero_pad_to <= dt->ldc), "detect_raw_bwd: bad zero_pad_to");
    const int cw = zero_pad_to > dt->c ? zero_pad_to : dt->c;
    DSN_DISPATCH_DTYPE(dt->dtype, T,
                       hipLaunchKernelGGL(detect_raw_bwd_kernel<T>, dim3(ew_grid(npix(dt) * cw)), dim3(256), 0,
                                          (hipStream_t)stream, draw, (T*)dt->ptr, dt->ldc, dt->n, dt->h, dt->w, na, no, cw));
    DSN_LAUNCH_CHECK("detect_raw_bwd");
    return DSN_OK;
}

// forward of all levels: ts[l] head outputs, raws[l] fp32 [N][na][ny][nx][no]; pred (may be NULL) [N][pred_rows][no] with level
// l starting at row_offs[l]; anchors_px: device [nl][na][2]
extern "C" int dsn_detect_decode_multi(const dsn_tensor* ts, float* const* raws, int32_t nl, float* pred, int64_t pred_rows,
                                       const int64_t* row_offs, int32_t na, int32_t no, const float* strides,
                                       const float* anchors_px, void* stream) {
    DSN_CHECK_ARG(ts && raws && nl >= 1 && nl <= DET_MAXL && na > 0 && no > 5 && strides && row_offs, "detect_decode_multi: invalid arguments");
    DSN_CHECK_ARG(!pred || anchors_px, "detect_decode_multi: pred needs anchors");
    DetectLevels L{};
    L.nl = nl; L.N = ts[0].n; L.na = na; L.no = no;
    int64_t most = 0;
    for (int l = 0; l < nl; ++l) {
        DSN_CHECK_ARG(tensor_ok(&ts[l]) && raws[l] && ts[l].c == na * no && ts[l].n == L.N && ts[l].dtype == ts[0].dtype,
                      "detect_decode_multi: level %d is malformed", l);
        DSN_CHECK_ARG(!pred || (row_offs[l] >= 0 && row_offs[l] + (int64_t)na * ts[l].h * ts[l].w <= pred_rows),
                      "detect_decode_multi: pred slice of level %d out of range", l);
        L.t[l] = ts[l].ptr; L.tld[l] = ts[l].ldc; L.raw[l] = raws[l]; L.ny[l] = ts[l].h; L.nx[l] = ts[l].w;
        L.row_off[l] = row_offs[l]; L.stride[l] = strides[l];
        const int64_t tot = npix(&ts[l]) * ts[l].c;
        most = tot > most ? tot : most;
    }
    if (most + 256ll * ew_grid(most) < (1ll << 32)) {
        DSN_DISPATCH_DTYPE(ts[0].dtype, T,
                           hipLaunchKernelGGL((detect_decode_multi_kernel<T, uint32_t>), dim3(ew_grid(most), nl), dim3(256), 0,
                                              (hipStream_t)stream, L, pred, pred_rows, anchors_px));
    } else {
        DSN_DISPATCH_DTYPE(ts[0].dtype, T,
                           hipLaunchKernelGGL((detect_decode_multi_kernel<T, int64_t>), dim3(ew_grid(most), nl), dim3(256), 0,
                                              (hipStream_t)stream, L, pred, pred_rows, anchors_px));
    }
    DSN_LAUNCH_CHECK("detect_decode_multi");
    return DSN_OK;
}

// backward of all levels: dts[l] (+ zero-filled row padding up to zero_pad_to[l]) from draws[l]; bias_grads[l] (may be NULL as
// a whole) += per-channel sums.  workspace: nl * 512 * na*no floats (per-block partial rows; no initialisation needed).
extern "C" int dsn_detect_raw_bwd_multi(const float* const* draws, const dsn_tensor* dts, int32_t nl, int32_t na, int32_t no,
                                        const int32_t* zero_pad_to, float* const* bias_grads, void* workspace,
                                        int64_t workspace_bytes, void* stream) {
    DSN_CHECK_ARG(draws && dts && nl >= 1 && nl <= DET_MAXL && na > 0 && no > 0, "detect_raw_bwd_multi: invalid arguments");
    const int C = na * no;
    DSN_CHECK_ARG(C <= 64, "detect_raw_bwd_multi: at most 64 head channels");
    constexpr int MAXB = 512;
    if (bias_grads && (!workspace || workspace_bytes < (int64_t)nl * MAXB * C * 4)) DSN_FAIL(DSN_EWORKSPACE, "detect_raw_bwd_multi: workspace too small");
    DetectLevels L{};
    L.nl = nl; L.N = dts[0].n; L.na = na; L.no = no;
    int64_t most = 0;
    for (int l = 0; l < nl; ++l) {
        DSN_CHECK_ARG(tensor_ok(&dts[l]) && draws[l] && dts[l].c == C && dts[l].n == L.N && dts[l].dtype == dts[0].dtype,
                      "detect_raw_bwd_multi: level %d is malformed", l);
        const int zp = zero_pad_to ? zero_pad_to[l] : 0;
        DSN_CHECK_ARG(zp == 0 || (zp >= C && zp <= dts[l].ldc), "detect_raw_bwd_multi: bad zero_pad_to for level %d", l);
        L.t[l] = dts[l].ptr; L.tld[l] = dts[l].ldc; L.raw[l] = (float*)draws[l]; L.ny[l] = dts[l].h; L.nx[l] = dts[l].w;
        L.cw[l] = zp > C ? zp : C;
        L.part[l] = bias_grads ? (float*)workspace + (int64_t)l * MAXB * C : nullptr;
        L.bias_grad[l] = bias_grads ? bias_grads[l] : nullptr;
        if (bias_grads) DSN_CHECK_ARG(bias_grads[l], "detect_raw_bwd_multi: null bias gradient for level %d", l);
        DSN_CHECK_ARG(npix(&dts[l]) < (1ll << 31), "detect_raw_bwd_multi: level %d has 2^31 or more pixels", l);
        const int64_t tot = npix(&dts[l]) * L.cw[l];
        most = tot > most ? tot : most;
    }
    hipStream_t st = (hipStream_t)stream;
    int64_t blocks = (most + 256 * 8 - 1) / (256 * 8);           // ~8 elements per thread
    blocks = blocks < 1 ? 1 : (blocks > MAXB ? MAXB : blocks);
    DSN_DISPATCH_DTYPE(dts[0].dtype, T,
                       hipLaunchKernelGGL(detect_raw_bwd_multi_kernel<T>, dim3((unsigned)blocks, nl), dim3(256), 0, st, L));
    if (bias_grads) hipLaunchKernelGGL(detect_bias_finalize_kernel, dim3(C, nl), dim3(64), 0, st, L, (int)blocks);
    DSN_LAUNCH_CHECK("detect_raw_bwd_multi");
    return DSN_OK;
}

// ---- the three head convolutions + permute of a TRAINING forward as one launch (round 3) -------------------------------------------
// yolo.py:258-276 runs, per level, a biased 1x1 convolution to na*no channels and a view/permute to [N][na][ny][nx][no].  As
// separate launches that was 3 implicit-GEMM kernels (7-11 us each: 33 output channels leave every tile 3/4 empty) + the permute
// launch (14 us) = 39 us of a 4 ms step for 0.76 GFLOP.  Here a block owns 64 pixels x all head channels (<= 64): the operands are
// read straight from global memory in MFMA fragment order (16-byte loads; the weights, <= 64 KB per level, stay in L2), the
// result goes through the SAME bf16 rounding the convolution's output had and lands in raw[] permuted.
struct DetHeadLevels {
    const void* x[DET_MAXL];
    const void* w[DET_MAXL];       // [na*no][K] bf16 (dsn_pack_weight_fwd)
    const float* bias[DET_MAXL];   // may be NULL
    float* raw[DET_MAXL];
    int64_t xld[DET_MAXL];
    int32_t ny[DET_MAXL], nx[DET_MAXL], K[DET_MAXL];
    int32_t blk0[DET_MAXL + 1];
    int32_t nl, N, na, no;
    uint32_t no_magic;             // floor(2^32 / no) + 1: __umulhi(v, no_magic) == v / no for v < 2^16
};
constexpr int DH_MI = 4;                    // 16-pixel fragments per block (64 pixels)
constexpr int DH_PX = 16 * DH_MI;
constexpr int DH_PITCH = DH_PX + 4;         // floats per channel row of the partial sums: 4 * 68 = 16 (mod 64) keeps the 4 k-groups in 4 bank quarters

// A block owns 64 pixels x all head channels; its 4 waves split the K axis (wave w takes the 32-wide k-steps w, w + 4, ...), so
// the 80 x 80 level (K = 128, most of the pixels) is ONE step per wave = one round trip, with few enough registers and LDS that
// (nearly) every block of the launch is resident at once -- the launch is bound by the latency chain of a block times the number
// of block rounds, not by bytes or flops (measured: 12 us for that level alone with 32-pixel blocks at 3 blocks per CU).  The
// four partial tiles are folded through LDS in a fixed order ((w0 + w2) + (w1 + w3)) and written [anchor][pixel][no]-major:
// consecutive threads, consecutive addresses.  No hardware divisions per element (they alone cost 17 us in the first version).
template <int NI>
__global__ __launch_bounds__(256) void detect_head_fwd_kernel(const DetHeadLevels L) {
    __shared__ float sP[2][NI * 16 * DH_PITCH];
    __shared__ float sB[NI * 16];
    int l = 0;
    while (l + 1 < L.nl && (int)blockIdx.x >= L.blk0[l + 1]) ++l;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, kg = lane >> 4;
    const int HW = L.ny[l] * L.nx[l], K = L.K[l], no = L.no, C = L.na * no;
    const int M = L.N * HW;                                   // (the host checks N * ny * nx < 2^31)
    const int p0 = ((int)blockIdx.x - L.blk0[l]) * DH_PX;
    const bf16_t* __restrict__ x = (const bf16_t*)L.x[l];
    const bf16_t* __restrict__ w = (const bf16_t*)L.w[l];
    // (the bias goes through LDS: a global load per output element made the epilogue a chain of L2 round trips)
    if ((int)threadIdx.x < NI * 16) sB[threadIdx.x] = (L.bias[l] && (int)threadIdx.x < C) ? L.bias[l][threadIdx.x] : 0.f;
    const bf16_t* xp[DH_MI];
    const bf16_t* wp[NI];
    bool wv[NI];
#pragma unroll
    for (int mi = 0; mi < DH_MI; ++mi) {
        int p = p0 + mi * 16 + r;
        p = p < M ? p : M - 1;                       // (rows past the end compute a duplicate that is never stored)
        xp[mi] = x + (int64_t)p * L.xld[l];
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int c = ni * 16 + r;
        wv[ni] = c < C;
        wp[ni] = w + (int64_t)(wv[ni] ? c : C - 1) * K;
    }
    f32x4 acc[NI][DH_MI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < DH_MI; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int kb = wave * 32; kb < K; kb += 128) {
        const int k = kb + kg * 8;                            // (K % 8 == 0: a lane's 8 values are all inside or all outside)
        const bool kv = k < K;
        const int ko = kv ? k : 0;
        u32x4 xa[DH_MI], wa[NI];
#pragma unroll
        for (int mi = 0; mi < DH_MI; ++mi) {
            xa[mi] = *reinterpret_cast<const u32x4*>(xp[mi] + ko);
            if (!kv) xa[mi] = u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            wa[ni] = *reinterpret_cast<const u32x4*>(wp[ni] + ko);
            if (!wv[ni]) wa[ni] = u32x4{0u, 0u, 0u, 0u};
        }
        // D[channel][pixel] += W[channel][k] * X[pixel][k]: a lane ends up with 4 consecutive channels of one pixel
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int mi = 0; mi < DH_MI; ++mi)
                acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wa[ni]), __builtin_bit_cast(bf16x8, xa[mi]),
                                                                      acc[ni][mi], 0, 0, 0);
    }
    float* sMine = sP[wave & 1];
    if (wave < 2) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int mi = 0; mi < DH_MI; ++mi)
#pragma unroll
                for (int i = 0; i < 4; ++i) sMine[(ni * 16 + kg * 4 + i) * DH_PITCH + mi * 16 + r] = acc[ni][mi][i];
    }
    __syncthreads();
    if (wave >= 2) {                                           // (each lane adds into the very elements its counterpart wrote)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int mi = 0; mi < DH_MI; ++mi)
#pragma unroll
                for (int i = 0; i < 4; ++i) sMine[(ni * 16 + kg * 4 + i) * DH_PITCH + mi * 16 + r] += acc[ni][mi][i];
    }
    __syncthreads();
    float* __restrict__ raw = L.raw[l];
    const int per_a = DH_PX * no, total = L.na * per_a;
    const int n0 = p0 / HW, q0 = p0 - n0 * HW;
    for (int e = threadIdx.x; e < total; e += 256) {
        const int a = (int)__umulhi((unsigned)(e / DH_PX), L.no_magic), rem = e - a * per_a;       // e / (DH_PX * no)
        const int pix = (int)__umulhi((unsigned)rem, L.no_magic), o = rem - pix * no;              // rem / no, exact for rem < 2^16
        if (p0 + pix >= M) continue;
        int n = n0, q = q0 + pix;
        while (q >= HW) { q -= HW; ++n; }
        const int c = a * no + o;
        const int at = c * DH_PITCH + pix;
        const float v = (sP[0][at] + sP[1][at]) + sB[c];
        raw[(((int64_t)n * L.na + a) * HW + q) * no + o] = to_f32<bf16_t>(from_f32<bf16_t>(v));
    }
}

extern "C" int dsn_detect_head_fwd_supported(int32_t dtype, int32_t na, int32_t no, int32_t k_min_multiple) {
    return dtype == DSN_BF16 && na > 0 && no >= 2 && na * no <= 64 && k_min_multiple > 0 && k_min_multiple % 8 == 0;
}

// xs[l]: bf16 head inputs (channels % 8 == 0, 16-byte rows), ws[l]: packed forward weights [na*no][K], biases[l]: fp32 or NULL,
// raws[l]: fp32 [N][na][ny][nx][no].  DSN_EUNSUPPORTED (nothing launched) outside dsn_detect_head_fwd_supported.
extern "C" int dsn_detect_head_fwd_multi(const dsn_tensor* xs, const void* const* ws, const float* const* biases, float* const* raws,
                                         int32_t nl, int32_t na, int32_t no, void* stream) {
    DSN_CHECK_ARG(xs && ws && raws && nl >= 1 && nl <= DET_MAXL && na > 0 && no > 0, "detect_head_fwd_multi: invalid arguments");
    const int C = na * no;
    DetHeadLevels L{};
    L.nl = nl; L.N = xs[0].n; L.na = na; L.no = no;
    L.no_magic = (uint32_t)((1ull << 32) / (uint64_t)no) + 1u;
    int blk = 0;
    double flops = 0, bytes = 0;
    for (int l = 0; l < nl; ++l) {
        DSN_CHECK_ARG(tensor_ok(&xs[l]) && ws[l] && raws[l] && xs[l].n == L.N && npix(&xs[l]) < (1ll << 31) - DH_PX,
                      "detect_head_fwd_multi: level %d is malformed", l);
        if (!dsn_detect_head_fwd_supported(xs[l].dtype, na, no, xs[l].c) || xs[l].ldc % 8 != 0 ||
            ((uintptr_t)xs[l].ptr | (uintptr_t)ws[l]) % 16 != 0)
            return DSN_EUNSUPPORTED;
        flops += 2.0 * npix(&xs[l]) * C * xs[l].c;
        bytes += (double)npix(&xs[l]) * (xs[l].c * 2.0 + C * 4.0) + (double)C * xs[l].c * 2.0;
    }
    // block ranges by DECREASING K: a block of the deepest level walks the most k-steps one after the other, so those start first
    // and the short-chain blocks of the large maps fill the tail of the launch
    int order[DET_MAXL];
    for (int l = 0; l < nl; ++l) order[l] = l;
    for (int i = 1; i < nl; ++i)
        for (int j = i; j > 0 && xs[order[j]].c > xs[order[j - 1]].c; --j) { const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
    for (int s = 0; s < nl; ++s) {
        const int l = order[s];
        L.x[s] = xs[l].ptr; L.w[s] = ws[l]; L.bias[s] = biases ? biases[l] : nullptr; L.raw[s] = raws[l];
        L.xld[s] = xs[l].ldc; L.ny[s] = xs[l].h; L.nx[s] = xs[l].w; L.K[s] = xs[l].c;
        L.blk0[s] = blk;
        blk += (int)((npix(&xs[l]) + DH_PX - 1) / DH_PX);
    }
    L.blk0[nl] = blk;
    hipStream_t st = (hipStream_t)stream;
    char layer[64] = "";
    if (dsn_prof_on()) {
        int ksum = 0;
        for (int l = 0; l < nl; ++l) ksum += xs[l].c;
        snprintf(layer, sizeof layer, "k1s1 %d->%d detect-heads x%d n%d", ksum, C, nl, L.N);
    }
    ProfScope prof("detect_head_fwd_kernel/bf16/64x64/fwd", layer, flops, bytes, st);
    if (C <= 16) hipLaunchKernelGGL(detect_head_fwd_kernel<1>, dim3(blk), dim3(256), 0, st, L);
    else if (C <= 32) hipLaunchKernelGGL(detect_head_fwd_kernel<2>, dim3(blk), dim3(256), 0, st, L);
    else if (C <= 48) hipLaunchKernelGGL(detect_head_fwd_kernel<3>, dim3(blk), dim3(256), 0, st, L);
    else hipLaunchKernelGGL(detect_head_fwd_kernel<4>, dim3(blk), dim3(256), 0, st, L);
    DSN_LAUNCH_CHECK("detect_head_fwd_multi");
    return DSN_OK;
}

extern "C" int64_t dsn_nms_workspace_bytes(int32_t bs, int32_t n, int32_t nc, int32_t multi_label) {
    if (bs <= 0 || n <= 0 || nc <= 0) return 0;
    const int64_t cap = key_cap(n, nc, multi_label && nc > 1);
    int64_t bytes = (int64_t)bs * cap * 8 + (int64_t)((bs * 4 + 255) / 256) * 256;
    if (cap > SORT_LDS_KEYS) bytes += (int64_t)bs * cap * 8 + 256;                    // second key buffer (tile merges ping-pong)
    bytes = (bytes + 255) / 256 * 256;
    bytes += (int64_t)bs * (cap < MAX_NMS ? cap : MAX_NMS) * CAND_W * 4;              // gathered candidate rows
    return bytes;
}

extern "C" int dsn_nms(const float* pred, int32_t bs, int32_t n, int32_t nc, float conf_thres, float iou_thres,
                       int32_t multi_label, int32_t agnostic, uint64_t classes_mask, int32_t max_det, float* out,
                       int32_t* out_count, void* workspace, int64_t workspace_bytes, void* stream) {
    DSN_CHECK_ARG(pred && out && out_count && workspace && bs > 0 && n > 0 && nc > 0 && nc <= 64,
                  "nms: invalid arguments");
    DSN_CHECK_ARG(conf_thres >= 0.f && conf_thres <= 1.f, "nms: invalid confidence threshold %f", conf_thres);
    DSN_CHECK_ARG(iou_thres >= 0.f && iou_thres <= 1.f, "nms: invalid IoU threshold %f", iou_thres);
    DSN_CHECK_ARG(max_det > 0 && max_det <= MAX_DET_CAP, "nms: max_det must be in 1..%d", MAX_DET_CAP);
    DSN_CHECK_ARG((int64_t)n * nc < (1ll << 31), "nms: too many candidates");
    multi_label = multi_label && nc > 1;   // general.py:679
    const int64_t workspace_bytes_needed = dsn_nms_workspace_bytes(bs, n, nc, multi_label);
    if (workspace_bytes < workspace_bytes_needed) DSN_FAIL(DSN_EWORKSPACE, "nms: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int64_t cap = key_cap(n, nc, multi_label);
    int32_t* counts = (int32_t*)workspace;
    uint64_t* keys = (uint64_t*)((char*)workspace + (int64_t)((bs * 4 + 255) / 256) * 256);
    dsn_fill_u32(counts, 0u, bs, st);
    const int cb = (n + 255) / 256;
    hipLaunchKernelGGL(nms_candidates_kernel, dim3(cb < 128 ? cb : 128, bs), dim3(256), 0, st, pred, bs, n, nc, conf_thres,
                       multi_label, classes_mask, keys, cap, counts);
    DSN_LAUNCH_CHECK("nms candidates");
    uint64_t* keys2 = cap > SORT_LDS_KEYS ? (uint64_t*)(((uintptr_t)(keys + (int64_t)bs * cap) + 255) / 256 * 256) : keys;
    if (cap > 4 * SORT_LDS_KEYS) hipLaunchKernelGGL(nms_select_kernel, dim3(bs), dim3(SORT_THREADS), 0, st, keys, cap, counts);
    const int tiles = cap > SORT_LDS_KEYS ? (int)((cap < 4 * SORT_LDS_KEYS ? cap : 4 * SORT_LDS_KEYS) / SORT_LDS_KEYS) : 1;
    hipLaunchKernelGGL(nms_tile_sort_kernel, dim3(tiles, bs), dim3(SORT_THREADS), 0, st, keys, cap, counts);
    if (tiles >= 2) hipLaunchKernelGGL(nms_merge16k_kernel, dim3(tiles, bs), dim3(SORT_THREADS), 0, st, keys, keys2, cap, counts);
    if (tiles >= 4) hipLaunchKernelGGL(nms_merge32k_kernel, dim3(tiles, bs), dim3(SORT_THREADS), 0, st, keys2, keys, cap, counts);
    DSN_LAUNCH_CHECK("nms sort");
    const int64_t cand_cap = cap < MAX_NMS ? cap : MAX_NMS;
    float* cand = (float*)((char*)workspace + workspace_bytes_needed - (int64_t)bs * cand_cap * CAND_W * 4);
    const int gb = (int)((cand_cap + 255) / 256);
    hipLaunchKernelGGL(nms_gather_kernel, dim3(gb < 128 ? gb : 128, bs), dim3(256), 0, st, pred, n, nc, keys, keys2, cap, counts,
                       cand, cand_cap);
    DSN_LAUNCH_CHECK("nms gather");
    hipLaunchKernelGGL(nms_greedy_kernel, dim3(bs), dim3(GREEDY_WAVES * 64), (size_t)((max_det + 3) / 4 * 4) * 5 * sizeof(float), st, cand, cand_cap,
                       counts, iou_thres, agnostic, max_det, out, out_count);
    DSN_LAUNCH_CHECK("nms greedy");
    return DSN_OK;
}

extern "C" int dsn_box_iou(const float* boxes1, int32_t n, const float* boxes2, int32_t m, float* out, void* stream) {
    DSN_CHECK_ARG(n >= 0 && m >= 0 && (n == 0 || boxes1) && (m == 0 || boxes2) && ((int64_t)n * m == 0 || out),
                  "box_iou: invalid arguments");
    if ((int64_t)n * m == 0) return DSN_OK;
    hipLaunchKernelGGL(box_iou_kernel, dim3(ew_grid((int64_t)n * m)), dim3(256), 0, (hipStream_t)stream, boxes1, boxes2, out, n, m);
    DSN_LAUNCH_CHECK("box_iou");
    return DSN_OK;
}

// out: int64[2 + 3*(nclass-1)], overwritten.
extern "C" int dsn_seg_eval_counts(const float* logits, const int64_t* target, int32_t n, int32_t c, int32_t h, int32_t w,
                                   int32_t nclass, int64_t* out, void* stream) {
    DSN_CHECK_ARG(logits && target && out && n > 0 && c > 0 && h > 0 && w > 0 && nclass >= 1 && nclass - 1 <= SEG_MAXBINS,
                  "seg_eval_counts: invalid arguments (at most %d classes)", SEG_MAXBINS + 1);
    hipStream_t st = (hipStream_t)stream;
    const int words = 2 * (2 + 3 * (nclass - 1));
    dsn_fill_u32(out, 0u, words, st);
    const int64_t HW = (int64_t)h * w, total = (int64_t)n * HW;
    int64_t b = (total + 255) / 256;
    // (few, fat blocks: every block ends with one global atomic per counter, and same-address atomics serialise at ~30 ns)
    hipLaunchKernelGGL(seg_eval_counts_kernel, dim3((unsigned)(b > 512 ? 512 : b)), dim3(256), 0, st, logits, target, c, HW, total,
                       nclass, (unsigned long long*)out);
    DSN_LAUNCH_CHECK("seg_eval_counts");
    return DSN_OK;
}
