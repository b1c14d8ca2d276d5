"""In-kernel clock of conv_pp.hip's main loop (diagnostic build with -DDSN_PP_STAMP, see pp_clock.sh): s_memtime / s_memrealtime
deltas around the main loop of every block, after >= 2 s of back-to-back launches on random data.  Prints the median clock, the
cycles per barrier interval and the share of those cycles the 16 MFMAs of an interval need (16 x 16 = 256)."""
import ctypes as C, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from desenet_amd import hip_ops as ops, _lib
L = _lib.lib()
L.dsn_pp_stamp_read.restype = C.c_int
dt = torch.bfloat16
import os
K1 = os.environ.get("PP_CLOCK_1X1") == "1"
CASES_1X1 = [("1x1 512->512 @4x80x80 (BN 256)", 4, 512, 80, 80, 512, 3), ("1x1 512->512 @4x80x80 (BN 128)", 4, 512, 80, 80, 512, 2),
             ("1x1 512->256 @4x160x160 (BN 256)", 4, 512, 160, 160, 256, 3), ("1x1 1024->1024 @4x40x40 (BN 128)", 4, 1024, 40, 40, 1024, 2),
             ("1x1 256->512 @4x160x160 (BN 128)", 4, 256, 160, 160, 512, 2)]
for (name, n, ci, h, w, co, mode) in CASES_1X1 if K1 else [("256->256 @4x80x80 (200 blocks)", 4, 256, 80, 80, 256, 5), ("128->128 @4x160x160 (400 blocks, 1/CU)", 4, 128, 160, 160, 128, 5),
                                     ("128->128 @4x160x160 (400 blocks, 2/CU)", 4, 128, 160, 160, 128, 4), ("512->128 @4x160x160", 4, 512, 160, 160, 128, 5)]:
    (L.dsn_pp1_mode if K1 else L.dsn_pp_mode)(mode)
    k = 1 if K1 else 3
    x = ops.new_act(n, ci, h, w, dt, "cuda"); x.normal_()
    wt = torch.randn(co, ci, k, k, device="cuda") * 0.05
    wf = ops.pack_weight_fwd(wt, dt)
    y = ops.new_act(n, co, h, w, dt, "cuda")
    p = ops.conv_params(k, 1, k // 2, 1)
    t0 = time.time()
    it = 0
    while time.time() - t0 < 2.5:
        for _ in range(50):
            ops.conv2d_fwd(x, wf, None, None, y, p)
        torch.cuda.synchronize(); it += 50
    bn = 256 if (K1 and mode == 3) else 128
    blocks = (n * h * w // 256 if K1 else n * (h // 16) * (w // 16)) * (co // bn)
    nb = min(blocks, 4096)
    buf = (C.c_ulonglong * (6 * nb))()
    assert L.dsn_pp_stamp_read(buf, nb) == 0
    a = np.array(buf[:], dtype=np.float64).reshape(nb, 6)
    clk = a[:, 0] / a[:, 1] * 100.0          # MHz
    phases = (ci // 32) * (bn // 128) if K1 else (ci // 64) * 18
    cyc = a[:, 0] / (2 * phases + 1)
    print(f"{name}: {it} launches, clock median {np.median(clk):.0f} MHz (min {clk.min():.0f}, max {clk.max():.0f}); main loop "
          f"{np.median(a[:, 0]):.0f} cycles = {np.median(cyc):.0f} per barrier interval ({256 / np.median(cyc) * 100:.0f} % MFMA issue), "
          f"{np.median(a[:, 1]) / 100:.2f} us", flush=True)
    e = (a[:, 2] - a[:, 2].min()) / 100
    end = e + (a[:, 3] + a[:, 1] + a[:, 4]) / 100
    print(f"    block entry after the first: median {np.median(e):.2f} us, p90 {np.percentile(e, 90):.2f}, max {e.max():.2f}; prologue (entry -> loop) median "
          f"{np.median(a[:, 3]) / 100:.2f} us (max {a[:, 3].max() / 100:.2f}); epilogue (loop end -> stores done) median {np.median(a[:, 4]) / 100:.2f} us "
          f"(max {a[:, 4].max() / 100:.2f}); last block done {end.max():.2f} us after the first entry", flush=True)
L.dsn_pp_mode(1)
L.dsn_pp1_mode(1)
