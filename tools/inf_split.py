import sys, time, torch
sys.path.insert(0, ".")
import bench, desenet_amd
from desenet_amd.core.utils.general import non_max_suppression
from desenet_amd.graph import GraphedInference
from desenet_amd.synth import synth_images
dev = torch.device("cuda", 0)
desenet_amd.set_compute_dtype(torch.float32)
m = bench.build_model(dev).eval().fuse()
x = (synth_images(16, 640, 2) * 255).round().to(torch.uint8).to(dev)
g = GraphedInference(m, x)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
(pred, _), seg = g()
print("forward graph  %.2f ms" % t(lambda: g()))
print("NMS only       %.2f ms" % t(lambda: non_max_suppression(pred, 0.25, 0.45, max_det=1000)))
print("both           %.2f ms" % t(lambda: non_max_suppression(g()[0][0], 0.25, 0.45, max_det=1000)))
out = non_max_suppression(pred, 0.25, 0.45, max_det=1000)
print("detections per image:", [o.shape[0] for o in out])
print("NMS val settings (conf .001, IoU .6, multi_label, max_det 300)  %.2f ms" %
      t(lambda: non_max_suppression(pred, 0.001, 0.6, multi_label=True, max_det=300), n=5))
