// Training losses of the hot path as HIP kernels (no host synchronisation, fixed launch shapes -> hipGraph-capturable):
//   detection : ComputeLoss.__call__ + build_targets (core/utils/loss.py:117-223) with CIoU (core/utils/metrics.py:202-244)
//   segmentation : nn.CrossEntropyLoss(ignore_index) over N x C x H x W logits (core/utils/loss.py:242-243)
// Forward AND the gradient w.r.t. the network outputs are produced together (the training step always needs both).
//
// Detection, per pyramid level (one launch each, candidates = 5 offsets x na anchors x nt targets, id = o*(na*nt) + a*nt + t,
// i.e. the reference's candidate order):
//   det_match   : anchor-ratio test, neighbour-cell offsets, CIoU of the decoded box against the target with forward-mode
//                 dual numbers (4 partials: the raw x, y, w, h logits), class BCE; per-candidate records; the objectness
//                 target of a cell is owned by the LAST matching candidate in reference order (atomicMax on the id), which
//                 is what the reference's sequential CPU scatter `tobj[b,a,gj,gi] = iou` leaves behind.
//   det_obj     : BCE-with-logits of every cell's objectness against clamp(iou_owner, 0); writes the whole dense gradient
//                 tensor (zeros + objectness term), block partial sums of the loss.
//   det_scatter : adds the box / class gradients of the active candidates into the dense gradient (float atomics: the
//                 only place -- a handful of adds per step, duplicates are rare).
// det_finalize folds the three levels into (lbox + lobj + lcls) * bs and the three reported items.
#include "common.h"

namespace {

constexpr int LT = 256;            // threads per loss block
constexpr int MAX_NC = 32;

// ---- forward-mode dual numbers over the 4 box logits ------------------------------------------------------------------------
struct D4 {
    float v, d[4];
};
__device__ __forceinline__ D4 dconst(float v) { return D4{v, {0.f, 0.f, 0.f, 0.f}}; }
__device__ __forceinline__ D4 dvar(float v, int i) { D4 r = dconst(v); r.d[i] = 1.f; return r; }
__device__ __forceinline__ D4 operator+(const D4& a, const D4& b) { D4 r; r.v = a.v + b.v; for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
__device__ __forceinline__ D4 operator-(const D4& a, const D4& b) { D4 r; r.v = a.v - b.v; for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
__device__ __forceinline__ D4 operator*(const D4& a, const D4& b) { D4 r; r.v = a.v * b.v; for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
__device__ __forceinline__ D4 operator/(const D4& a, const D4& b) {
    D4 r; const float inv = 1.f / b.v; r.v = a.v * inv;
    for (int i = 0; i < 4; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv;
    return r;
}
__device__ __forceinline__ D4 operator+(const D4& a, float b) { D4 r = a; r.v += b; return r; }
__device__ __forceinline__ D4 operator-(const D4& a, float b) { D4 r = a; r.v -= b; return r; }
__device__ __forceinline__ D4 operator*(const D4& a, float b) { D4 r; r.v = a.v * b; for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] * b; return r; }
__device__ __forceinline__ D4 dmin(const D4& a, float b) { return a.v < b ? a : dconst(b); }
__device__ __forceinline__ D4 dmax(const D4& a, float b) { return a.v > b ? a : dconst(b); }
__device__ __forceinline__ D4 dclamp0(const D4& a) { return a.v > 0.f ? a : dconst(0.f); }
__device__ __forceinline__ D4 dsigmoid(const D4& a) {
    const float s = 1.f / (1.f + expf(-a.v));
    D4 r; r.v = s;
    for (int i = 0; i < 4; ++i) r.d[i] = s * (1.f - s) * a.d[i];
    return r;
}
__device__ __forceinline__ D4 datan(const D4& a) {
    D4 r; r.v = atanf(a.v);
    const float g = 1.f / (1.f + a.v * a.v);
    for (int i = 0; i < 4; ++i) r.d[i] = g * a.d[i];
    return r;
}

// BCE with logits, pos_weight pw: loss = (1 - t)*x + (1 + (pw-1)*t) * softplus(-x);  dloss/dx = sigma*(1 + (pw-1) t) - pw t
__device__ __forceinline__ float bce_logits(float x, float t, float pw, float* dx) {
    const float lw = 1.f + (pw - 1.f) * t;
    const float sp = fmaxf(-x, 0.f) + log1pf(expf(-fabsf(x)));     // softplus(-x), stable
    const float s = 1.f / (1.f + expf(-x));
    *dx = (1.f - t) - lw * (1.f - s);
    return (1.f - t) * x + lw * sp;
}

// FocalLoss wrapped around the same criterion (loss.py:36-61, 106-110; alpha = 0.25 as the reference constructs it):
//   loss = bce * w,  w = af * (1 - p_t)^gamma,  p_t = t p + (1 - t)(1 - p),  af = t alpha + (1 - t)(1 - alpha),  p = sigmoid(x)
//   d loss / dx = dbce/dx * w + bce * af * gamma (1 - p_t)^(gamma - 1) * (-(2t - 1) p (1 - p))
// gamma <= 0: the plain criterion (bit-identical to bce_logits).
__device__ __forceinline__ float bce_focal(float x, float t, float pw, float gamma, float* dx) {
    float d0;
    const float l0 = bce_logits(x, t, pw, &d0);
    if (!(gamma > 0.f)) { *dx = d0; return l0; }
    const float alpha = 0.25f;
    const float p = 1.f / (1.f + expf(-x));
    const float pt = t * p + (1.f - t) * (1.f - p);
    const float af = t * alpha + (1.f - t) * (1.f - alpha);
    const float om = fmaxf(1.f - pt, 0.f);
    const float m = powf(om, gamma);
    const float dm = om > 0.f ? gamma * powf(om, gamma - 1.f) * (-(2.f * t - 1.f) * p * (1.f - p)) : 0.f;
    *dx = af * (d0 * m + l0 * dm);
    return l0 * af * m;
}

struct DetParams {
    int32_t bs, na, no, nc, ny, nx, nt;
    float anchor_t, cls_pw, obj_pw, cp, cn, fl_gamma;
    float anchors[6];              // this level, grid units, [na<=3][2]
};

struct Cand {                      // one (offset, anchor, target) candidate
    int32_t cell;                  // ((b*na + a)*ny + gj)*nx + gi, or -1 when inactive
    int32_t cls;
    float iou;
    float dbox[4];                 // d(1 - ciou)/d(raw x,y,w,h)
};

// All pyramid levels run in ONE launch per stage (blockIdx.y = level): 5 launches per step instead of 14.
constexpr int MAX_MB = 16;         // match blocks per level
struct DetLevel {
    const float* p;                // raw predictions of the level
    float* dp;                     // their gradient
    Cand* cands;
    float* dcls;
    int32_t* owner;
    float* partial;                // [<= 1024] objectness block sums
    float* mpart;                  // [MAX_MB][3] match block sums (n active, sum(1 - ciou), sum class BCE)
    float* acc;                    // [4]: n active, sum(1 - ciou), sum class BCE, sum objectness BCE
    DetParams q;
    float obj_coef;                // h_obj * bs / cells (* the level's balance when that is a host constant)
    int32_t ob, mb, sb;            // blocks of the obj / match / scatter stages
};
// bal_dev != NULL: the per-level objectness weights live on the device (autobalance updates them every step, loss.py:158-164)
struct DetLevels { DetLevel l[5]; int32_t nl; float box_coef, cls_coef; const float* bal_dev; };

__global__ __launch_bounds__(LT) void det_clear_kernel(const DetLevels L) {
    const DetLevel& v = L.l[blockIdx.y];
    const int64_t ncell = (int64_t)v.q.bs * v.q.na * v.q.ny * v.q.nx;
    for (int64_t i = blockIdx.x * (int64_t)LT + threadIdx.x; i < ncell; i += (int64_t)gridDim.x * LT) v.owner[i] = -1;
}

__global__ __launch_bounds__(LT) void det_match_kernel(const DetLevels L, const float* __restrict__ targets) {
    const DetLevel& v = L.l[blockIdx.y];
    if ((int)blockIdx.x >= v.mb) return;
    const float* __restrict__ p = v.p;
    const DetParams q = v.q;
    Cand* __restrict__ cands = v.cands;
    float* __restrict__ dcls = v.dcls;
    int32_t* __restrict__ owner = v.owner;
    __shared__ float red[3][LT];
    const int ncand = 5 * q.na * q.nt;
    float n_act = 0.f, s_box = 0.f, s_cls = 0.f;
    for (int id = blockIdx.x * LT + threadIdx.x; id < ncand; id += v.mb * LT) {
        const int o = id / (q.na * q.nt), rem = id - o * (q.na * q.nt);
        const int a = rem / q.nt, t = rem - a * q.nt;
        const float* tg = targets + (int64_t)t * 6;
        const float gx = tg[2] * (float)q.nx, gy = tg[3] * (float)q.ny;
        const float gw = tg[4] * (float)q.nx, gh = tg[5] * (float)q.ny;
        const float aw = q.anchors[2 * a], ah = q.anchors[2 * a + 1];
        const float rw = gw / aw, rh = gh / ah;
        bool act = fmaxf(fmaxf(rw, 1.f / rw), fmaxf(rh, 1.f / rh)) < q.anchor_t;
        float ox = 0.f, oy = 0.f;
        if (o == 1) { act = act && (fmodf(gx, 1.f) < 0.5f) && (gx > 1.f); ox = 0.5f; }
        else if (o == 2) { act = act && (fmodf(gy, 1.f) < 0.5f) && (gy > 1.f); oy = 0.5f; }
        else if (o == 3) { const float ix = (float)q.nx - gx; act = act && (fmodf(ix, 1.f) < 0.5f) && (ix > 1.f); ox = -0.5f; }
        else if (o == 4) { const float iy = (float)q.ny - gy; act = act && (fmodf(iy, 1.f) < 0.5f) && (iy > 1.f); oy = -0.5f; }
        // a label row that names an image outside the batch or a class outside [0, nc) is skipped, never used as an index
        // (also rejects NaNs); rows padded with w = h = 0 (a fixed-capacity label buffer under hipGraph replay) fail the
        // anchor-ratio test above: 1 / 0 = inf >= anchor_t
        act = act && tg[0] >= 0.f && tg[0] < (float)q.bs && tg[1] >= 0.f && tg[1] < (float)q.nc;
        Cand c;
        c.cell = -1; c.cls = 0; c.iou = 0.f;
        c.dbox[0] = c.dbox[1] = c.dbox[2] = c.dbox[3] = 0.f;
        if (act) {
            const int b = (int)tg[0];
            const int ci = (int)(gx - ox), cj = (int)(gy - oy);            // .long() truncation
            const int gi = min(max(ci, 0), q.nx - 1), gj = min(max(cj, 0), q.ny - 1);
            const float tx = gx - (float)ci, ty = gy - (float)cj;          // target box in cell units (un-clamped cell)
            c.cell = ((b * q.na + a) * q.ny + gj) * q.nx + gi;
            c.cls = (int)tg[1];
            const float* ps = p + (int64_t)c.cell * q.no;
            // predicted box: pxy = 2*sigmoid - 0.5, pwh = (2*sigmoid)^2 * anchor
            const D4 sx = dsigmoid(dvar(ps[0], 0)), sy = dsigmoid(dvar(ps[1], 1));
            const D4 sw = dsigmoid(dvar(ps[2], 2)), sh = dsigmoid(dvar(ps[3], 3));
            const D4 px = sx * 2.f - 0.5f, py = sy * 2.f - 0.5f;
            const D4 pw = (sw * 2.f) * (sw * 2.f) * aw, ph = (sh * 2.f) * (sh * 2.f) * ah;
            // CIoU (metrics.py:202-244, x1y1x2y2=False), box1 = prediction, box2 = target
            const float eps = 1e-7f;
            const D4 b1x1 = px - pw * 0.5f, b1x2 = px + pw * 0.5f, b1y1 = py - ph * 0.5f, b1y2 = py + ph * 0.5f;
            const float b2x1 = tx - gw / 2.f, b2x2 = tx + gw / 2.f, b2y1 = ty - gh / 2.f, b2y2 = ty + gh / 2.f;
            const D4 iw = dclamp0(dmin(b1x2, b2x2) - dmax(b1x1, b2x1));
            const D4 ih = dclamp0(dmin(b1y2, b2y2) - dmax(b1y1, b2y1));
            const D4 inter = iw * ih;
            const D4 w1 = b1x2 - b1x1, h1 = b1y2 - b1y1 + eps;
            const float w2 = b2x2 - b2x1, h2 = b2y2 - b2y1 + eps;
            const D4 uni = w1 * h1 + (w2 * h2) - inter + eps;
            const D4 iou = inter / uni;
            const D4 cw = dmax(b1x2, b2x2) - dmin(b1x1, b2x1);
            const D4 chh = dmax(b1y2, b2y2) - dmin(b1y1, b2y1);
            const D4 c2 = cw * cw + chh * chh + eps;
            const D4 dxx = (b1x1 + b1x2) * -1.f + (b2x1 + b2x2), dyy = (b1y1 + b1y2) * -1.f + (b2y1 + b2y2);
            const D4 rho2 = (dxx * dxx + dyy * dyy) * 0.25f;
            const D4 at = datan(w1 / h1) * -1.f + atanf(w2 / h2);
            const D4 v = at * at * 0.4052847345693511f;                   // 4 / pi^2
            const float alpha = v.v / (v.v - iou.v + (1.f + eps));        // no_grad in the reference
            const D4 ciou = iou - (rho2 / c2 + v * alpha);
            c.iou = ciou.v;
            for (int i = 0; i < 4; ++i) c.dbox[i] = -ciou.d[i];
            n_act += 1.f;
            s_box += 1.f - ciou.v;
            // class BCE over nc logits (target cp at the class, cn elsewhere); gradient per candidate kept unscaled
            if (q.nc > 1) {
                for (int k = 0; k < q.nc; ++k) {
                    float dx;
                    s_cls += bce_focal(ps[5 + k], k == c.cls ? q.cp : q.cn, q.cls_pw, q.fl_gamma, &dx);
                    dcls[(int64_t)id * q.nc + k] = dx;
                }
            }
            atomicMax(&owner[c.cell], id);
        }
        cands[id] = c;
    }
    red[0][threadIdx.x] = n_act; red[1][threadIdx.x] = s_box; red[2][threadIdx.x] = s_cls;
    __syncthreads();
    for (int s = LT / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s)
            for (int k = 0; k < 3; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) { v.mpart[blockIdx.x * 3] = red[0][0]; v.mpart[blockIdx.x * 3 + 1] = red[1][0]; v.mpart[blockIdx.x * 3 + 2] = red[2][0]; }
}

// objectness BCE over every cell + dense gradient initialisation; partial[blockIdx] = sum of the block's cells
__global__ __launch_bounds__(LT) void det_obj_kernel(const DetLevels L) {
    const DetLevel& v = L.l[blockIdx.y];
    if ((int)blockIdx.x >= v.ob) return;
    const float* __restrict__ p = v.p;
    const DetParams q = v.q;
    const Cand* __restrict__ cands = v.cands;
    const int32_t* __restrict__ owner = v.owner;
    float* __restrict__ dp = v.dp;
    const float obj_coef = L.bal_dev ? v.obj_coef * L.bal_dev[blockIdx.y] : v.obj_coef;
    float* __restrict__ partial = v.partial;
    __shared__ float red[LT];
    const int64_t ncell = (int64_t)q.bs * q.na * q.ny * q.nx;
    const int ncand = 5 * q.na * q.nt;
    float s = 0.f;
    for (int64_t cell = blockIdx.x * (int64_t)LT + threadIdx.x; cell < ncell; cell += (int64_t)v.ob * LT) {
        const int own = owner[cell];
        const float tobj = (own >= 0 && own < ncand) ? fmaxf(cands[own].iou, 0.f) : 0.f;   // (a stale table can never index out of range)
        float dx;
        s += bce_focal(p[cell * q.no + 4], tobj, q.obj_pw, q.fl_gamma, &dx);
        float* d = dp + cell * q.no;
        for (int k = 0; k < q.no; ++k) d[k] = 0.f;
        d[4] = dx * obj_coef;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int t = LT / 2; t > 0; t >>= 1) {
        if (threadIdx.x < t) red[threadIdx.x] += red[threadIdx.x + t];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// add the (scaled) box / class gradients of active candidates; block 0 of a level also folds its partial sums into acc
__global__ __launch_bounds__(LT) void det_scatter_kernel(const DetLevels L) {
    const DetLevel& v = L.l[blockIdx.y];
    if ((int)blockIdx.x >= v.sb) return;
    const DetParams q = v.q;
    const Cand* __restrict__ cands = v.cands;
    const float* __restrict__ dcls = v.dcls;
    float* __restrict__ dp = v.dp;
    float n = 0.f, sbox = 0.f, scls = 0.f;
    if (q.nt > 0)
        for (int b = 0; b < v.mb; ++b) { n += v.mpart[b * 3]; sbox += v.mpart[b * 3 + 1]; scls += v.mpart[b * 3 + 2]; }   // fixed order
    const int ncand = 5 * q.na * q.nt;
    const float kb = n > 0.f ? L.box_coef / n : 0.f;
    const float kc = (n > 0.f && q.nc > 1) ? L.cls_coef / (n * (float)q.nc) : 0.f;
    for (int id = blockIdx.x * LT + threadIdx.x; id < ncand; id += v.sb * LT) {
        const Cand c = cands[id];
        if (c.cell < 0) continue;
        float* d = dp + (int64_t)c.cell * q.no;
        for (int i = 0; i < 4; ++i) atomicAdd(&d[i], c.dbox[i] * kb);
        if (q.nc > 1)
            for (int k = 0; k < q.nc; ++k) atomicAdd(&d[5 + k], dcls[(int64_t)id * q.nc + k] * kc);
    }
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        double s = 0.0;
        for (int i = threadIdx.x; i < v.ob; i += 64) s += (double)v.partial[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
        if (threadIdx.x == 0) { v.acc[0] = n; v.acc[1] = sbox; v.acc[2] = scls; v.acc[3] = (float)s; }
    }
}

// acc: [nl][4];  out[0] = (lbox + lobj + lcls) * bs, out[1..3] = lbox, lobj, lcls (already multiplied by their gains)
struct DetMeta { float ncells[5], balance[5]; };
__global__ void det_finalize_kernel(const float* __restrict__ acc, int nl, DetMeta meta, float h_box, float h_obj,
                                    float h_cls, int nc, int bs, float* __restrict__ out, float* __restrict__ bal_dev,
                                    int autobalance, int ssi) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float lbox = 0.f, lobj = 0.f, lcls = 0.f;
    float nb[5];
    for (int i = 0; i < nl; ++i) {
        const float n = acc[i * 4];
        if (n > 0.f) {
            lbox += acc[i * 4 + 1] / n;
            if (nc > 1) lcls += acc[i * 4 + 2] / (n * (float)nc);
        }
        const float bal = bal_dev ? bal_dev[i] : meta.balance[i];
        const float obji = acc[i * 4 + 3] / meta.ncells[i];
        lobj += obji * bal;
        nb[i] = bal * 0.9999f + 0.0001f / obji;           // loss.py:160 (applied below when autobalance)
    }
    if (autobalance && bal_dev) {                         // loss.py:158-164: running rescale, normalised by the stride-16 level
        const float norm = nb[ssi];
        for (int i = 0; i < nl; ++i) bal_dev[i] = nb[i] / norm;
    }
    lbox *= h_box; lobj *= h_obj; lcls *= h_cls;
    out[0] = (lbox + lobj + lcls) * (float)bs;
    out[1] = lbox; out[2] = lobj; out[3] = lcls;
}

typedef long i64x2 __attribute__((ext_vector_type(2)));
// ---- segmentation cross entropy ---------------------------------------------------------------------------------------------
// pass 1: per-block partial (sum of -log softmax[target], number of valid pixels)
// (four consecutive pixels per thread when HW % 4 == 0: 16-byte loads of each class plane and of the int64 targets -- the
// one-pixel-per-lane form ran at 1.5 TB/s on the 8 x 2 x 640 x 640 logits)
template <int V>
__global__ __launch_bounds__(LT) void seg_ce_fwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target,
                                                        int N, int C, int64_t HW, int ignore, float* __restrict__ partial) {
    __shared__ float red[2][LT];
    float s = 0.f, cnt = 0.f;
    const int64_t total = (int64_t)N * HW / V;
    for (int64_t q = blockIdx.x * (int64_t)LT + threadIdx.x; q < total; q += (int64_t)gridDim.x * LT) {
        const int64_t i = q * V;
        const int64_t n = i / HW, px = i - n * HW;
        const float* l = logits + n * C * HW + px;
        int64_t tv[V];
        float m[V], z[V], lt[V];
        if (V == 4) {
            const i64x2 t0 = *reinterpret_cast<const i64x2*>(target + i), t1 = *reinterpret_cast<const i64x2*>(target + i + 2);
            tv[0] = t0[0]; tv[1] = t0[1]; tv[2] = t1[0]; tv[3] = t1[1];
        } else {
            tv[0] = target[i];
        }
#pragma unroll
        for (int k = 0; k < V; ++k) { m[k] = -INFINITY; z[k] = 0.f; lt[k] = 0.f; }
        for (int c = 0; c < C; ++c) {
            float v[V];
            if (V == 4) { const f32x4 t = *reinterpret_cast<const f32x4*>(l + c * HW); v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3]; }
            else v[0] = l[c * HW];
#pragma unroll
            for (int k = 0; k < V; ++k) { m[k] = fmaxf(m[k], v[k]); if (c == tv[k]) lt[k] = v[k]; }
        }
        for (int c = 0; c < C; ++c) {
            float v[V];
            if (V == 4) { const f32x4 t = *reinterpret_cast<const f32x4*>(l + c * HW); v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3]; }
            else v[0] = l[c * HW];
#pragma unroll
            for (int k = 0; k < V; ++k) z[k] += expf(v[k] - m[k]);
        }
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const int64_t t = tv[k];
            if (t == ignore || t < 0 || t >= C) continue;
            s += logf(z[k]) + m[k] - lt[k];
            cnt += 1.f;
        }
    }
    red[0][threadIdx.x] = s; red[1][threadIdx.x] = cnt;
    __syncthreads();
    for (int t = LT / 2; t > 0; t >>= 1) {
        if (threadIdx.x < t) { red[0][threadIdx.x] += red[0][threadIdx.x + t]; red[1][threadIdx.x] += red[1][threadIdx.x + t]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = red[0][0]; partial[2 * blockIdx.x + 1] = red[1][0]; }
}
// out[0] = mean loss, out[1] = 1 / valid count (0 when no pixel is valid); one wave folds the block partials
__global__ __launch_bounds__(64) void seg_ce_finalize_kernel(const float* __restrict__ partial, int n, float* __restrict__ out) {
    double s = 0.0, c = 0.0;
    for (int i = threadIdx.x; i < n; i += 64) { s += (double)partial[2 * i]; c += (double)partial[2 * i + 1]; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { s += __shfl_down(s, off); c += __shfl_down(c, off); }
    if (threadIdx.x == 0) {
        out[0] = c > 0.0 ? (float)(s / c) : 0.f;      // torch returns nan for an all-ignored target; 0 keeps the step finite
        out[1] = c > 0.0 ? (float)(1.0 / c) : 0.f;
    }
}
// pass 2: dlogits = (softmax - onehot) / count   (0 for ignored pixels)
template <int V>
__global__ __launch_bounds__(LT) void seg_ce_bwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target,
                                                        int N, int C, int64_t HW, int ignore, const float* __restrict__ fin,
                                                        float* __restrict__ dlogits) {
    const float inv = fin[1];
    const int64_t total = (int64_t)N * HW / V;
    for (int64_t q = blockIdx.x * (int64_t)LT + threadIdx.x; q < total; q += (int64_t)gridDim.x * LT) {
        const int64_t i = q * V;
        const int64_t n = i / HW, px = i - n * HW;
        const float* l = logits + n * C * HW + px;
        float* d = dlogits + n * C * HW + px;
        int64_t tv[V];
        float m[V], z[V];
        if (V == 4) {
            const i64x2 t0 = *reinterpret_cast<const i64x2*>(target + i), t1 = *reinterpret_cast<const i64x2*>(target + i + 2);
            tv[0] = t0[0]; tv[1] = t0[1]; tv[2] = t1[0]; tv[3] = t1[1];
        } else {
            tv[0] = target[i];
        }
#pragma unroll
        for (int k = 0; k < V; ++k) { m[k] = -INFINITY; z[k] = 0.f; }
        for (int c = 0; c < C; ++c) {
            float v[V];
            if (V == 4) { const f32x4 t = *reinterpret_cast<const f32x4*>(l + c * HW); v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3]; }
            else v[0] = l[c * HW];
#pragma unroll
            for (int k = 0; k < V; ++k) m[k] = fmaxf(m[k], v[k]);
        }
        for (int c = 0; c < C; ++c) {
            float v[V];
            if (V == 4) { const f32x4 t = *reinterpret_cast<const f32x4*>(l + c * HW); v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3]; }
            else v[0] = l[c * HW];
#pragma unroll
            for (int k = 0; k < V; ++k) z[k] += expf(v[k] - m[k]);
        }
        float iz[V];
        bool ok[V];
#pragma unroll
        for (int k = 0; k < V; ++k) { iz[k] = 1.f / z[k]; ok[k] = !(tv[k] == ignore || tv[k] < 0 || tv[k] >= C); }
        for (int c = 0; c < C; ++c) {
            float v[V], o[V];
            if (V == 4) { const f32x4 t = *reinterpret_cast<const f32x4*>(l + c * HW); v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3]; }
            else v[0] = l[c * HW];
#pragma unroll
            for (int k = 0; k < V; ++k) o[k] = ok[k] ? (expf(v[k] - m[k]) * iz[k] - (c == tv[k] ? 1.f : 0.f)) * inv : 0.f;
            if (V == 4) *reinterpret_cast<f32x4*>(d + c * HW) = f32x4{o[0], o[1], o[2], o[3]};
            else d[c * HW] = o[0];
        }
    }
}

// ---- x8 bilinear (align_corners) + cross entropy, forward AND backward, without the full-resolution logits ---------------------
// SegMaskPSP ends in Conv2d(c_hid, n_segcls, 1) at 1/8 resolution followed by nn.Upsample(scale 8, bilinear, align_corners=True)
// (yolo.py:182-183), and training feeds that N x C x H x W fp32 tensor straight into nn.CrossEntropyLoss (loss.py:242-243):
// 26 MB written, read twice, and 26 MB of gradient written and read back -- five launches, ~100 us.  Here ONE kernel walks the
// high-resolution pixels, interpolates the C logits on the fly from an LDS copy of the 2-3 low-resolution rows it needs (the
// arithmetic of bilinear_kernel: same coordinates, same weights, same order of operations), evaluates the loss AND its gradient
// (softmax - onehot), and folds the gradient through the interpolation weights into the low-resolution grid (LDS atomics per
// block, then one global atomic per touched element); a second small kernel divides by the number of valid pixels, rounds to the
// storage type and restores the accumulator's zeros.  fp32 atomics: the summation order of the gradient is not fixed (1e-7).
struct Lerp2 { int i0, i1; float l0, l1; };
__device__ __forceinline__ Lerp2 lerp_ac(int o, float scale, int in) {     // = resample.hip: lerp_coord
    const float sc = scale * (float)o;
    Lerp2 r;
    r.i0 = (int)sc;
    if (r.i0 > in - 1) r.i0 = in - 1;
    r.i1 = r.i0 + (r.i0 < in - 1 ? 1 : 0);
    r.l1 = sc - (float)r.i0;
    r.l0 = 1.f - r.l1;
    return r;
}
constexpr int SCU_ROWS = 4;        // low-resolution rows a band of <= 8 output rows may touch (checked on the host)
// One block = one band of `RB` output rows of one image.  Stage 1: every thread evaluates its pixels (interpolation from an LDS
// copy of the low-resolution rows, softmax, loss) and parks the gradient w.r.t. the INTERPOLATED logits in LDS.  Stage 2 folds it
// back through the separable interpolation weights as two GATHERS (first along x, then along y): no atomics inside the block,
// a fixed summation order; only the hand-over of the (<= 4 x wl x C) band result to the global accumulator uses atomics, because
// neighbouring bands share low-resolution rows.
// NT threads per block: 512 halves the serial per-thread pixel loop (20 -> 10 pixels of exp / log each at 640 x 640); the grid is one
// round of blocks either way (LDS allows three per CU)
constexpr int SCU_NT = 512;
template <typename T, int C>
__global__ __launch_bounds__(SCU_NT) void seg_ce_up_kernel(const T* __restrict__ lg, int64_t ld, int hl, int wl, int H, int W, int RB,
                                                        float sh, float sw, const int64_t* __restrict__ target, int ignore,
                                                        float* __restrict__ gacc, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];
    float* s_gh = s_dyn;                            // [RB][W][C]
    float* s_tx = s_gh + (size_t)RB * W * C;        // [RB][wl][C]
    float* s_lg = s_tx + (size_t)RB * wl * C;       // [SCU_ROWS][wl][C]
    __shared__ float red[2][SCU_NT];
    const int nb = (H + RB - 1) / RB;
    const int n = blockIdx.x / nb, band = blockIdx.x - n * nb;
    const int Y0 = band * RB;
    const int rows = (Y0 + RB <= H) ? RB : H - Y0;
    const int ylo0 = lerp_ac(Y0, sh, hl).i0;
    for (int i = threadIdx.x; i < SCU_ROWS * wl * C; i += SCU_NT) {
        const int c = i % C, x = (i / C) % wl, r = i / (C * wl);
        const int yl = ylo0 + r;
        s_lg[i] = yl < hl ? to_f32<T>(lg[(((int64_t)n * hl + yl) * wl + x) * ld + c]) : 0.f;
    }
    __syncthreads();
    float s = 0.f, cnt = 0.f;
    const int64_t* tg = target + ((int64_t)n * H + Y0) * W;
    for (int p = threadIdx.x; p < rows * W; p += SCU_NT) {
        const int yr = p / W, X = p - yr * W;
        const int64_t t = tg[p];
        float gh[C];
#pragma unroll
        for (int c = 0; c < C; ++c) gh[c] = 0.f;
        if (!(t == ignore || t < 0 || t >= C)) {
            const Lerp2 a = lerp_ac(Y0 + yr, sh, hl), b = lerp_ac(X, sw, wl);
            const float* l0 = s_lg + (size_t)(a.i0 - ylo0) * wl * C;
            const float* l1 = s_lg + (size_t)(a.i1 - ylo0) * wl * C;
            float v[C], m = -INFINITY, lt = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                v[c] = a.l0 * (b.l0 * l0[b.i0 * C + c] + b.l1 * l0[b.i1 * C + c]) + a.l1 * (b.l0 * l1[b.i0 * C + c] + b.l1 * l1[b.i1 * C + c]);
                m = fmaxf(m, v[c]);
                if (c == t) lt = v[c];
            }
            float z = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) z += expf(v[c] - m);
            s += logf(z) + m - lt;
            cnt += 1.f;
            const float iz = 1.f / z;
#pragma unroll
            for (int c = 0; c < C; ++c) gh[c] = expf(v[c] - m) * iz - (c == t ? 1.f : 0.f);
        }
#pragma unroll
        for (int c = 0; c < C; ++c) s_gh[(size_t)p * C + c] = gh[c];
    }
    red[0][threadIdx.x] = s; red[1][threadIdx.x] = cnt;
    __syncthreads();
    // stage 2a: along x.  Low-resolution column xl receives l0 from the pixels whose left neighbour it is and l1 from those whose
    // right neighbour it is: sw * X in [xl - 1, xl + 1)
    const float isw = 1.f / sw;
    for (int i = threadIdx.x; i < rows * wl * C; i += SCU_NT) {
        const int c = i % C, xl = (i / C) % wl, yr = i / (C * wl);
        int xlo = (int)((float)(xl - 1) * isw) - 1, xhi = (int)((float)(xl + 1) * isw) + 1;
        xlo = xlo < 0 ? 0 : xlo;
        xhi = xhi > W - 1 ? W - 1 : xhi;
        float acc = 0.f;
        const float* row = s_gh + (size_t)yr * W * C + c;
        for (int X = xlo; X <= xhi; ++X) {
            const Lerp2 b = lerp_ac(X, sw, wl);
            const float wgt = (b.i0 == xl ? b.l0 : 0.f) + (b.i1 == xl ? b.l1 : 0.f);
            acc += wgt * row[(size_t)X * C];
        }
        s_tx[i] = acc;
    }
    for (int t = SCU_NT / 2; t > 0; t >>= 1) {
        if ((int)threadIdx.x < t) { red[0][threadIdx.x] += red[0][threadIdx.x + t]; red[1][threadIdx.x] += red[1][threadIdx.x + t]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = red[0][0]; partial[2 * blockIdx.x + 1] = red[1][0]; }
    // stage 2b: along y, then the band's contribution to the (zero-initialised) global accumulator
    for (int i = threadIdx.x; i < SCU_ROWS * wl * C; i += SCU_NT) {
        const int r = i / (C * wl), yl = ylo0 + r;
        if (yl >= hl) continue;
        float acc = 0.f;
        for (int yr = 0; yr < rows; ++yr) {
            const Lerp2 a = lerp_ac(Y0 + yr, sh, hl);
            const float wgt = (a.i0 == yl ? a.l0 : 0.f) + (a.i1 == yl ? a.l1 : 0.f);
            acc += wgt * s_tx[(size_t)yr * wl * C + (i - r * wl * C)];
        }
        if (acc != 0.f) unsafeAtomicAdd(&gacc[(int64_t)n * hl * wl * C + (int64_t)yl * wl * C + (i - r * wl * C)], acc);
    }
}
// out[0] = mean CE, out[1] = 1 / valid; dlog (storage type, NHWC rows of `ld` elements, padding lanes zero) = gain * gacc / valid;
// gacc is reset to zero (the accumulator is a "zero once" workspace).
template <typename T>
__global__ __launch_bounds__(256) void seg_ce_up_finalize_kernel(const float* __restrict__ partial, int np, float* __restrict__ gacc,
                                                                 int64_t npx, int C, T* __restrict__ dlog, int64_t ld, float gain,
                                                                 float* __restrict__ out) {
    __shared__ double rs[2][256];
    double s = 0.0, c = 0.0;
    for (int i = threadIdx.x; i < np; i += 256) { s += (double)partial[2 * i]; c += (double)partial[2 * i + 1]; }
    rs[0][threadIdx.x] = s; rs[1][threadIdx.x] = c;
    __syncthreads();
    for (int t = 128; t > 0; t >>= 1) {
        if ((int)threadIdx.x < t) { rs[0][threadIdx.x] += rs[0][threadIdx.x + t]; rs[1][threadIdx.x] += rs[1][threadIdx.x + t]; }
        __syncthreads();
    }
    const double cs = rs[1][0];
    const float inv = cs > 0.0 ? (float)(1.0 / cs) : 0.f;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        out[0] = cs > 0.0 ? (float)(rs[0][0] / cs) : 0.f;
        out[1] = inv;
    }
    for (int64_t px = blockIdx.x * 256ll + threadIdx.x; px < npx; px += (int64_t)gridDim.x * 256) {
        for (int k = 0; k < (int)ld; ++k) {
            float v = 0.f;
            if (k < C) { v = gacc[px * C + k] * inv * gain; gacc[px * C + k] = 0.f; }
            dlog[px * ld + k] = from_f32<T>(v);
        }
    }
}

inline int lgrid(int64_t total, int cap) {
    int64_t b = (total + LT - 1) / LT;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

extern "C" int64_t dsn_det_loss_workspace_bytes(int32_t nl, int32_t na, int32_t nt, int32_t nc, int64_t max_cells) {
    const int64_t ncand = (int64_t)5 * na * (nt > 0 ? nt : 1);
    const int64_t per_level = ncand * (int64_t)sizeof(Cand) + ncand * nc * 4 + max_cells * 4 + 1024 * 4;
    return nl * per_level + 256;
}

// p[i]: fp32 [bs, na, ny_i, nx_i, no] contiguous; dp[i]: same shape (gradient of out[0] w.r.t. p[i]); targets fp32 [nt,6]
// (image, class, x, y, w, h normalised); anchors: [nl][na][2] HOST floats in grid units; balance: [nl] HOST floats.
// out (device): [0] = (lbox + lobj + lcls) * bs, [1..3] = lbox, lobj, lcls.
extern "C" int dsn_det_loss(const float* const* p, float* const* dp, const int32_t* ny, const int32_t* nx, int32_t nl,
                            int32_t bs, int32_t na, int32_t nc, const float* targets, int32_t nt, const float* anchors,
                            const float* balance, float h_box, float h_obj, float h_cls, float cls_pw, float obj_pw,
                            float anchor_t, float cp, float cn, float* out, void* workspace, int64_t workspace_bytes,
                            void* stream) {
    return dsn_det_loss_opt(p, dp, ny, nx, nl, bs, na, nc, targets, nt, anchors, balance, h_box, h_obj, h_cls, cls_pw, obj_pw,
                            anchor_t, cp, cn, 0.f, nullptr, 0, 0, out, workspace, workspace_bytes, stream);
}

// The same with the options scripts/train.py leaves off: fl_gamma > 0 wraps both BCE criteria in the reference's FocalLoss
// (loss.py:106-110); balance_dev != NULL: [nl] DEVICE floats used instead of `balance` (which may then be NULL), and with
// autobalance != 0 updated after the losses are formed (loss.py:158-164; ssi: index of the stride-16 level, loss.py:113) -- the
// reference calls `.item()` per level there; here the state never leaves the device, so the step stays capturable.
extern "C" int dsn_det_loss_opt(const float* const* p, float* const* dp, const int32_t* ny, const int32_t* nx, int32_t nl,
                                int32_t bs, int32_t na, int32_t nc, const float* targets, int32_t nt, const float* anchors,
                                const float* balance, float h_box, float h_obj, float h_cls, float cls_pw, float obj_pw,
                                float anchor_t, float cp, float cn, float fl_gamma, float* balance_dev, int32_t autobalance,
                                int32_t ssi, float* out, void* workspace, int64_t workspace_bytes, void* stream) {
    DSN_CHECK_ARG(p && dp && ny && nx && out && workspace && anchors && (balance || balance_dev) && nl > 0 && nl <= 5 && na > 0 &&
                      na <= 3,
                  "det_loss: bad arguments");
    DSN_CHECK_ARG(fl_gamma >= 0.f && (!autobalance || (balance_dev && ssi >= 0 && ssi < nl)), "det_loss: bad focal / autobalance options");
    DSN_CHECK_ARG(nc >= 1 && nc <= MAX_NC && bs > 0 && nt >= 0 && (nt == 0 || targets), "det_loss: bad sizes");
    int64_t max_cells = 0;
    for (int i = 0; i < nl; ++i) {
        const int64_t c = (int64_t)bs * na * ny[i] * nx[i];
        max_cells = c > max_cells ? c : max_cells;
    }
    if (workspace_bytes < dsn_det_loss_workspace_bytes(nl, na, nt, nc, max_cells))
        DSN_FAIL(DSN_EWORKSPACE, "det_loss: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int no = nc + 5;
    const int64_t ncand = (int64_t)5 * na * (nt > 0 ? nt : 1);
    char* w = (char*)workspace;
    float* acc = (float*)w;                 // [nl][4]
    w += 256;
    DetMeta meta{};
    DetLevels L{};
    L.nl = nl; L.box_coef = h_box * (float)bs; L.cls_coef = h_cls * (float)bs; L.bal_dev = balance_dev;
    int max_ob = 1, max_mb = 1, max_sb = 1, max_cb = 1;
    for (int i = 0; i < nl; ++i) {
        meta.ncells[i] = (float)((int64_t)bs * na * ny[i] * nx[i]);
        meta.balance[i] = balance_dev ? 1.f : balance[i];
        DetLevel& v = L.l[i];
        DetParams& q = v.q;
        q.bs = bs; q.na = na; q.no = no; q.nc = nc; q.ny = ny[i]; q.nx = nx[i]; q.nt = nt;
        q.anchor_t = anchor_t; q.cls_pw = cls_pw; q.obj_pw = obj_pw; q.cp = cp; q.cn = cn; q.fl_gamma = fl_gamma;
        for (int k = 0; k < 2 * na; ++k) q.anchors[k] = anchors[i * na * 2 + k];
        v.p = p[i]; v.dp = dp[i]; v.acc = acc + i * 4;
        v.cands = (Cand*)w;                              w += ncand * sizeof(Cand);
        v.dcls = (float*)w;                              w += ncand * nc * 4;
        v.owner = (int32_t*)w;                           w += max_cells * 4;
        v.partial = (float*)w;                           w += 1024 * 4;      // [0, 960): objectness block sums, [960, 1008): match sums
        const int64_t ncell = (int64_t)bs * na * ny[i] * nx[i];
        v.ob = lgrid(ncell, 960);
        v.mpart = v.partial + 960;                       // 64 floats at the tail of the 1024-float block: MAX_MB*3 = 48
        v.mb = nt > 0 ? lgrid(ncand, MAX_MB) : 0;
        v.sb = nt > 0 ? lgrid(ncand, 64) : 1;
        v.obj_coef = h_obj * (balance_dev ? 1.f : balance[i]) * (float)bs / (float)ncell;
        max_ob = v.ob > max_ob ? v.ob : max_ob;
        max_mb = v.mb > max_mb ? v.mb : max_mb;
        max_sb = v.sb > max_sb ? v.sb : max_sb;
        const int cb = lgrid(ncell, 64);
        max_cb = cb > max_cb ? cb : max_cb;
    }
    // (no memset nodes: see common.h dsn_fill_u32) owner = -1 for every level, then match -> obj -> scatter, levels in gridDim.y
    hipLaunchKernelGGL(det_clear_kernel, dim3(max_cb, nl), dim3(LT), 0, st, L);
    if (nt > 0) hipLaunchKernelGGL(det_match_kernel, dim3(max_mb, nl), dim3(LT), 0, st, L, targets);
    hipLaunchKernelGGL(det_obj_kernel, dim3(max_ob, nl), dim3(LT), 0, st, L);
    hipLaunchKernelGGL(det_scatter_kernel, dim3(max_sb, nl), dim3(LT), 0, st, L);
    DSN_LAUNCH_CHECK("det_loss");
    hipLaunchKernelGGL(det_finalize_kernel, dim3(1), dim3(1), 0, st, acc, nl, meta, h_box, h_obj, h_cls, nc, bs, out, balance_dev,
                       autobalance, ssi);
    DSN_LAUNCH_CHECK("det_loss finalize");
    return DSN_OK;
}

extern "C" int64_t dsn_seg_ce_workspace_bytes(void) { return (int64_t)(2 * 1024 + 4) * sizeof(float); }

// Fused nn.Upsample(scale_factor, 'bilinear', align_corners=True) + CrossEntropyLoss(ignore_index), loss and gradient (see
// seg_ce_up_kernel).  logits: the LOW-resolution classifier output (NHWC, c classes); target: [n, H, W] int64; dlogits: gradient
// w.r.t. `logits` (same n, h, w; c channels + zeroed row padding up to its ldc), scaled by `gain`.
// workspace: dsn_seg_ce_up_workspace_bytes(n, h, w, c, H) bytes whose first n*h*w*c floats are ZERO on entry (the call restores
// them).  DSN_EUNSUPPORTED unless c == 2 (DeSeNet: se_nc 2), w <= 160 and the scale factor is >= 3.
extern "C" int64_t dsn_seg_ce_up_workspace_bytes(int32_t n, int32_t h, int32_t w, int32_t c, int32_t H) {
    const int64_t nb = (int64_t)n * ((H + 3) / 4);          // (bands of 4 or 8 output rows)
    return ((int64_t)n * h * w * c + 2 * nb + 64) * (int64_t)sizeof(float);
}
extern "C" int dsn_seg_ce_up(const dsn_tensor* logits, const int64_t* target, int32_t H, int32_t W, int32_t ignore_index,
                             float gain, float* out, const dsn_tensor* dlogits, void* workspace, int64_t workspace_bytes,
                             void* stream) {
    DSN_CHECK_ARG(tensor_ok(logits) && tensor_ok(dlogits) && target && out && workspace && H > 1 && W > 1, "seg_ce_up: bad arguments");
    DSN_CHECK_ARG(dlogits->n == logits->n && dlogits->h == logits->h && dlogits->w == logits->w && dlogits->c == logits->c &&
                      dlogits->dtype == logits->dtype, "seg_ce_up: gradient tensor must have the logits' shape");
    const int n = logits->n, hl = logits->h, wl = logits->w, C = logits->c;
    // a band of RB output rows may touch floor(scale * (RB - 1)) + 3 low-resolution rows at most; LDS: RB rows of the output width
    const int RB = W <= 640 ? 8 : 4;
    const size_t lds = ((size_t)RB * W * C + (size_t)RB * wl * C + (size_t)SCU_ROWS * wl * C) * sizeof(float);
    if (C != 2 || lds > 96 * 1024 || (float)(hl - 1) / (float)(H - 1) * (float)(RB - 1) + 3.f > (float)SCU_ROWS)
        DSN_FAIL(DSN_EUNSUPPORTED, "seg_ce_up: %d classes, width %d, %dx%d -> %dx%d needs the unfused path", C, wl, hl, wl, H, W);
    if (workspace_bytes < dsn_seg_ce_up_workspace_bytes(n, hl, wl, C, H)) DSN_FAIL(DSN_EWORKSPACE, "seg_ce_up: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float* gacc = (float*)workspace;
    float* partial = gacc + (int64_t)n * hl * wl * C;
    const int nb = n * ((H + RB - 1) / RB);
    const float sh = (float)(hl - 1) / (float)(H - 1), sw = (float)(wl - 1) / (float)(W - 1);
    const int64_t npx = (int64_t)n * hl * wl;
    DSN_DISPATCH_DTYPE(logits->dtype, T, {
        DSN_LDS_ATTR((seg_ce_up_kernel<T, 2>), 96 * 1024);
        hipLaunchKernelGGL((seg_ce_up_kernel<T, 2>), dim3(nb), dim3(SCU_NT), lds, st, (const T*)logits->ptr, logits->ldc, hl, wl, H, W, RB,
                           sh, sw, target, ignore_index, gacc, partial);
        hipLaunchKernelGGL(seg_ce_up_finalize_kernel<T>, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, st, partial, nb, gacc, npx, C,
                           (T*)dlogits->ptr, dlogits->ldc, gain, out);
    });
    DSN_LAUNCH_CHECK("seg_ce_up");
    return DSN_OK;
}

// logits: contiguous NCHW fp32; target: int64 [N,H,W]; out (device) [0] = mean CE over valid pixels, [1] = 1/valid count;
// dlogits (may be NULL): d(out[0]) / d(logits), same layout as logits.
extern "C" int dsn_seg_ce(const float* logits, const int64_t* target, int32_t n, int32_t c, int32_t h, int32_t w,
                          int32_t ignore_index, float* out, float* dlogits, void* workspace, int64_t workspace_bytes,
                          void* stream) {
    DSN_CHECK_ARG(logits && target && out && workspace && n > 0 && c > 0 && c <= 1024 && h > 0 && w > 0,
                  "seg_ce: bad arguments");
    if (workspace_bytes < dsn_seg_ce_workspace_bytes()) DSN_FAIL(DSN_EWORKSPACE, "seg_ce: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int64_t HW = (int64_t)h * w;
    const int nb = lgrid((int64_t)n * HW, 1024);
    float* partial = (float*)workspace;
    const bool v4 = HW % 4 == 0 && ((uintptr_t)logits % 16) == 0 && ((uintptr_t)target % 16) == 0 &&
                    (!dlogits || ((uintptr_t)dlogits % 16) == 0);
    if (v4)
        hipLaunchKernelGGL(seg_ce_fwd_kernel<4>, dim3(nb), dim3(LT), 0, st, logits, target, n, c, HW, ignore_index, partial);
    else
        hipLaunchKernelGGL(seg_ce_fwd_kernel<1>, dim3(nb), dim3(LT), 0, st, logits, target, n, c, HW, ignore_index, partial);
    hipLaunchKernelGGL(seg_ce_finalize_kernel, dim3(1), dim3(64), 0, st, partial, nb, out);
    if (dlogits && v4)
        hipLaunchKernelGGL(seg_ce_bwd_kernel<4>, dim3(lgrid((int64_t)n * HW / 4, 8192)), dim3(LT), 0, st, logits, target, n, c, HW,
                           ignore_index, out, dlogits);
    else if (dlogits)
        hipLaunchKernelGGL(seg_ce_bwd_kernel<1>, dim3(lgrid((int64_t)n * HW, 8192)), dim3(LT), 0, st, logits, target, n, c, HW,
                           ignore_index, out, dlogits);
    DSN_LAUNCH_CHECK("seg_ce");
    return DSN_OK;
}
