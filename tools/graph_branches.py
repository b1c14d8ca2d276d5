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


if __name__ == "__main__":
    main()
