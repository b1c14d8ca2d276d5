"""Randomised sweep of the BatchNorm sums of the ping-pong kernels: forward partial sums against the kernels replaced, backward sums
(one or two segments) against the stand-alone reduction over the dz just written; and the kernel-row weight gradients against the
other weight-gradient kinds.  Both epilogue forms, 128- / 256-channel tiles, two blocks per CU."""
import sys, random
import torch
sys.path.insert(0, ".")
import desenet_amd
from desenet_amd import _lib, hip_ops as ops
from desenet_amd.hip_ops import ACT_SILU, ACT_NONE
dt = torch.bfloat16
desenet_amd.set_compute_dtype(dt)
L = _lib.lib()
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100
def fold(a, c): return a.view(torch.float64)[:8 * 2 * c].view(8, 2, c).sum(0)
def rnd(shape, scale=1.0): return torch.randn(shape, device="cuda") * scale
bad = 0
ONLY = int(sys.argv[3]) if len(sys.argv) > 3 else -1
for it in range(N):
    torch.manual_seed(it)
    kind = rng.choice(["k3", "k1", "s2", "wgrad"])
    n = rng.choice([1, 2, 3])
    h, w = rng.randint(3, 60), rng.randint(3, 60)
    m3, m1, d = rng.choice([2, 3, 4, 5]), rng.choice([2, 3]), rng.choice([0, 1])
    if ONLY >= 0 and it != ONLY:
        # (keep the random stream aligned: draw what the case would have drawn)
        if kind == "wgrad": rng.randint(1, 3); rng.randint(1, 3); rng.random()
        else:
            if kind == "k3": rng.randint(1, 4); rng.randint(2, 40)
            elif kind == "k1": rng.randint(1, 16); rng.randint(2, 40)
            else: rng.randint(4, 32); rng.randint(1, 4); rng.randint(2, 40); rng.randint(2, 40)
        continue
    try:
        if kind == "wgrad":
            ci, co = 128 * rng.randint(1, 3), 128 * rng.randint(1, 3)
            x, gy = ops.as_act(rnd((n, ci, h, w)).to(dt)), ops.as_act(rnd((n, co, h, w)).to(dt))
            acc = rng.random() < 0.5
            base = rnd((co, ci, 3, 3))
            got = []
            for md in (2, 0):
                L.dsn_wgrad_pp_mode(md)
                g = base.clone() if acc else torch.zeros_like(base)
                ops.conv2d_wgrad(x, gy, g, ci, ops.conv_params(3, 1, 1, 1, accumulate=acc), oihw=True)
                torch.cuda.synchronize()
                got.append(g)
            e = float((got[0] - got[1]).abs().max()) / (float(got[1].abs().max()) + 1e-9)
            ok = e <= 2e-4
            info = f"err {e:.2e}"
        else:
            k = 3 if kind != "k1" else 1
            if kind == "k3": ci, co = 64 * rng.randint(1, 4), 8 * rng.randint(2, 40)
            elif kind == "k1": ci, co = 32 * rng.randint(1, 16), 8 * rng.randint(2, 40)
            else: ci, co = 8 * rng.randint(4, 32), 64 * rng.randint(1, 4); h, w = 2 * rng.randint(2, 40), 2 * rng.randint(2, 40)
            L.dsn_pp_dir(d)
            # ---- backward sums of the data gradient
            if kind == "s2":
                conv = torch.nn.Conv2d(ci, co, 3, 2, 1, bias=False).cuda()
                bank = ops.WeightBank([conv], [ci], dt, "cuda"); bank.pack()
                gd = ops.as_act(rnd((n, co, h // 2, w // 2)).to(dt))
            else:
                wt = rnd((co, ci, k, k), 0.05)
                wd = ops.pack_weight_dgrad(wt, dt)
                gd = ops.as_act(rnd((n, co, h, w)).to(dt))
            nseg = rng.choice([1, 2]) if ci >= 16 else 1
            cut = (ci // 16) * 8 if nseg == 2 else ci
            segs_c = [(0, cut), (cut, ci)] if nseg == 2 else [(0, ci)]
            segments, refs = [], []
            for j, (c0, c1) in enumerate(segs_c):
                c = c1 - c0
                yseg = ops.as_act(rnd((n, c, h, w)).to(dt))
                st = torch.stack([torch.rand(c, device="cuda") + 0.5, torch.rand(c, device="cuda") - 0.5, torch.randn(c, device="cuda") * 0.1, torch.rand(c, device="cuda") + 0.5])
                a, _ = ops.bn_acc(c, "cuda")
                segments.append((c0, c1, yseg, st[0], st[1], st[2], st[3], rng.choice([ACT_SILU, ACT_NONE]), a, c, 0))
                refs.append((yseg, st, a, c, segments[-1][7]))
            L.dsn_pp_mode(m3 if kind != "s2" else rng.choice([2, 3])); L.dsn_pp1_mode(m1)
            dx = ops.new_act(n, ci, h, w, dt, "cuda")
            res = ops.as_act(rnd((n, ci, h, w)).to(dt)) if (kind != "s2" and rng.random() < 0.5) else None
            ops.profile_enable(True)
            if kind == "s2":
                ops.conv2d_dgrad_s2(gd, bank.dgrad_s2[0], dx, ops.conv_params(3, 2, 1, 1), red=ops.bnred(segments))
            else:
                ops.conv2d_dgrad(gd, wd, dx, ops.conv_params(k, 1, k // 2, 1), residual=res, red=ops.bnred(segments))
            torch.cuda.synchronize()
            labs = list(ops.profile_collect()); ops.profile_enable(False)
            ok, info = True, f" nseg {nseg} res {res is not None} acts {[r[4] for r in refs]}"
            for (c0, c1), (yseg, st, a, c, act) in zip(segs_c, refs):
                ws, _ = ops.bn_acc(c, "cuda")
                ops.bn_act_bwd_reduce(dx[:, c0:c1], yseg, st[0], st[1], st[2], st[3], act, ws)
                torch.cuda.synchronize()
                want, got = fold(ws, c), fold(a, c)
                e = float((got - want).abs().max()) / (float(want.abs().max()) + 1e-30)
                if e > 2e-5 * (h * w * n) ** 0.5:
                    ok = False
                info += f" bwd-sums err {e:.2e} [{labs[0][:40] if labs else ''}]"
            # ---- forward sums
            if kind != "s2":
                wf = ops.pack_weight_fwd(wt, dt)
                x = ops.as_act(rnd((n, ci, h, w)).to(dt))
                sums = []
                for md in ((m3, m1), (0, 0)):
                    L.dsn_pp_mode(md[0]); L.dsn_pp1_mode(md[1])
                    y = ops.new_act(n, co, h, w, dt, "cuda")
                    acc_, _ = ops.conv2d_fwd_acc(x, wf, y, ops.conv_params(k, 1, k // 2, 1))
                    torch.cuda.synchronize()
                    sums.append((y.clone(), fold(acc_, co)))
                if not torch.equal(sums[0][0], sums[1][0]):
                    ok = False; info += " fwd y differs"
                e = float((sums[0][1] - sums[1][1]).abs().max()) / (float(sums[1][1].abs().max()) + 1e-30)
                if e > 2e-5 * (h * w * n) ** 0.5:
                    ok = False
                info += f" fwd-sums err {e:.2e}"
        if not ok:
            bad += 1
            print("BAD", it, kind, (n, ci, co, h, w), (m3, m1, d), info, flush=True)
    except Exception as e:
        bad += 1
        print("ERROR", it, kind, (n, ci, co, h, w), (m3, m1, d), repr(e)[:300], flush=True)
L.dsn_pp_mode(1); L.dsn_pp1_mode(1); L.dsn_pp_dir(0); L.dsn_wgrad_pp_mode(1)
print("cases", N, "bad", bad)
