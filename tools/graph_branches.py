#!/usr/bin/env python3
"""Do parallel branches of a captured hipGraph overlap on this stack?  Two chains of N dependent tiny kernels captured (a) on
one stream, (b) on two streams forked and joined inside the capture; and the same with medium (3.3 MB) kernels."""
import sys, time, torch


def timed(fn, reps=5):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def main():
    n = 200
    for name, numel in (("tiny (64 floats)", 64), ("medium (1.6 M floats)", 1 << 21)):
        a = torch.zeros(numel, device="cuda"); b = torch.zeros(numel, device="cuda")
        s2 = torch.cuda.Stream()
        g1 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1):
            for _ in range(n):
                a.add_(1.0)
            for _ in range(n):
                b.add_(1.0)
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2):
            cur = torch.cuda.current_stream()
            s2.wait_stream(cur)
            with torch.cuda.stream(s2):
                for _ in range(n):
                    b.add_(1.0)
            for _ in range(n):
                a.add_(1.0)
            cur.wait_stream(s2)
        t1, t2 = timed(g1.replay), timed(g2.replay)
        print(f"{name:24s} one chain of {2*n}: {t1*1e3:7.1f} us   two parallel chains of {n}: {t2*1e3:7.1f} us", flush=True)


def main_convs():
    """The same question with the library's own latency-bound kernels: two independent chains of conv -> BN/act launches."""
    sys.path.insert(0, ".")
    from desenet_amd import hip_ops as ops
    dt = torch.bfloat16
    n = 40

    def chain(c, hw):
        x = ops.new_act(8, c, hw, hw, dt, "cuda"); x.normal_()
        y = ops.new_act(8, c, hw, hw, dt, "cuda")
        w = ops.pack_weight_fwd(torch.randn(c, c, 1, 1, device="cuda") * 0.05, dt)
        p = ops.conv_params(1, 1, 0, 1)
        sc = torch.ones(c, device="cuda"); sh = torch.zeros(c, device="cuda")

        def run():
            for _ in range(n):
                ops.conv2d_fwd(x, w, None, None, y, p)
                ops.bn_act_fwd(y, sc, sh, ops.ACT_SILU, None, x)
        return run

    for name, (ca, ha), (cb, hb) in (("128ch@40 | 128ch@40", (128, 40), (128, 40)), ("64ch@80 | 256ch@20", (64, 80), (256, 20)),
                                     ("128ch@80 | 128ch@80", (128, 80), (128, 80))):
        ra, rb = chain(ca, ha), chain(cb, hb)
        ra(); rb(); torch.cuda.synchronize()
        s2 = torch.cuda.Stream()
        g1 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1):
            ra(); rb()
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2):
            cur = torch.cuda.current_stream()
            s2.wait_stream(cur)
            with torch.cuda.stream(s2):
                rb()
            ra()
            cur.wait_stream(s2)
        ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(ga):
            ra()
        with torch.cuda.graph(gb):
            rb()
        s3 = torch.cuda.Stream()

        def two_graphs():
            cur = torch.cuda.current_stream()
            s3.wait_stream(cur)
            with torch.cuda.stream(s3):
                gb.replay()
            ga.replay()
            cur.wait_stream(s3)
        t1, t2, t3 = timed(g1.replay), timed(g2.replay), timed(two_graphs)
        print(f"conv+bn chains {name:22s} serial in one graph: {t1*1e3:7.1f} us   two branches of one graph: {t2*1e3:7.1f} us   "
              f"two graphs on two streams: {t3*1e3:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
    main_convs()
