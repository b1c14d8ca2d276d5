"""Randomised sweep of the ping-pong kernels (3x3, 1x1, 2x2 form of the stride-2 data gradient; both epilogue forms; 128- and
256-channel tiles; two blocks per CU) against the kernels they replace: outputs must be BIT-IDENTICAL (same k order).  Random map
sizes (ragged patches, single rows / columns), channel counts (partial tiles, channel-slice operands), epilogue options."""
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
N = int(sys.argv[2]) if len(sys.argv) > 2 else 150
bad = 0
def rnd(shape, scale=1.0):
    return torch.randn(shape, device="cuda") * scale
ONLY = int(sys.argv[3]) if len(sys.argv) > 3 else -1
for it in range(N):
    kind = rng.choice(["k3", "k3", "k1", "k1", "s2"])
    n = rng.choice([1, 1, 2, 3, 5])
    h, w = rng.randint(1, 70), rng.randint(1, 70)
    if kind == "k3":
        ci, co = 64 * rng.randint(1, 6), 8 * rng.randint(1, 48)
    elif kind == "k1":
        ci, co = 32 * rng.randint(1, 20), 8 * rng.randint(1, 48)
    else:
        ci, co = 8 * rng.randint(4, 40), 64 * rng.randint(1, 5)      # conv ci -> co, stride 2: dy has co channels (whole slabs)
        h, w = rng.randint(2, 90), rng.randint(2, 90)
    pad_s, pad_d = rng.choice([0, 0, 8, 64]), rng.choice([0, 0, 8, 24])
    act = rng.choice([ACT_NONE, ACT_SILU])
    use_bias, use_res, accum = rng.random() < 0.4, rng.random() < 0.4, rng.random() < 0.3
    direction = rng.choice(["fwd", "dgrad"])
    torch.manual_seed(it)
    skip = ONLY >= 0 and it != ONLY
    modes = [(rng.choice([2, 3, 4, 5]), rng.choice([2, 3]), rng.choice([0, 1]))]
    if skip:
        continue
    try:
        if kind in ("k3", "k1"):
            k = 3 if kind == "k3" else 1
            wt = rnd((co, ci, k, k), 0.05)
            src_c, dst_c = (ci, co) if direction == "fwd" else (co, ci)
            sw = ops.as_act(rnd((n, src_c + pad_s, h, w)).to(dt)); src = sw[:, pad_s:]
            base = ops.as_act(rnd((n, dst_c + pad_d, h, w)).to(dt))
            res = ops.as_act(rnd((n, dst_c, h, w)).to(dt)) if use_res else None
            bias = rnd((dst_c,)) if (use_bias and direction == "fwd") else None
            wp = ops.pack_weight_fwd(wt, dt) if direction == "fwd" else ops.pack_weight_dgrad(wt, dt)
            outs = []
            for (m3, m1, d) in modes + [(0, 0, 0)]:
                L.dsn_pp_mode(m3); L.dsn_pp1_mode(m1); L.dsn_pp_dir(d)
                o = base.clone()
                if direction == "fwd":
                    ops.conv2d_fwd(src, wp, bias, res, o[:, pad_d:], ops.conv_params(k, 1, k // 2, 1, act=act, accumulate=False))
                else:
                    ops.conv2d_dgrad(src, wp, o[:, pad_d:], ops.conv_params(k, 1, k // 2, 1, accumulate=accum), residual=res)
                torch.cuda.synchronize()
                outs.append(o)
            ok = torch.equal(outs[0], outs[1])
        else:
            conv = torch.nn.Conv2d(ci, co, 3, 2, 1, bias=False).cuda()
            bank = ops.WeightBank([conv], [ci], dt, "cuda"); bank.pack()
            ho, wo = ops.conv_out_hw(h, w, 3, 2, 1, 1)
            gd = ops.as_act(rnd((n, co, ho, wo)).to(dt))
            base = ops.as_act(rnd((n, ci + pad_d, h, w)).to(dt))
            outs = []
            for (m3, m1, d) in modes + [(0, 0, 0)]:
                L.dsn_pp_mode(m3 if m3 in (2, 3) else 2); L.dsn_pp_dir(d)
                o = base.clone()
                ops.conv2d_dgrad_s2(gd, bank.dgrad_s2[0], o[:, pad_d:], ops.conv_params(3, 2, 1, 1, accumulate=accum))
                torch.cuda.synchronize()
                outs.append(o)
            ok = float((outs[0].float() - outs[1].float()).abs().max()) <= 1e-2 * float(outs[1].float().abs().max()) + 1e-6   # (other k order)
        if not ok:
            bad += 1
            d = (outs[0].float() - outs[1].float()).abs()
            print("   max |diff|", float(d.max()), "of", float(outs[1].float().abs().max()), "elements differing", int((d > 0).sum()), "of", d.numel())
            print("MISMATCH", it, kind, direction, (n, ci, co, h, w), "pads", pad_s, pad_d, "act", act, "bias", use_bias, "res", use_res, "acc", accum, modes, flush=True)
    except Exception as e:
        bad += 1
        print("ERROR", it, kind, direction, (n, ci, co, h, w), modes, repr(e)[:200], flush=True)
L.dsn_pp_mode(1); L.dsn_pp1_mode(1); L.dsn_pp_dir(0)
print("cases", N, "bad", bad)
