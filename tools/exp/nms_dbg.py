import sys, ast, numpy as np, torch
sys.path.insert(0, ".")
from tests.util import golden
from desenet_amd import hip_ops as ops
g = golden("nms")
case = "default"
kw = ast.literal_eval(str(g[f"{case}/kw"]))
print(kw)
pred = torch.from_numpy(g[f"{case}/pred"]).cuda()
print(pred.shape)
out, cnt = ops.nms(pred, kw["conf_thres"], kw["iou_thres"], kw.get("multi_label", False), kw.get("agnostic", False), kw.get("classes"), kw["max_det"])
cnt = cnt.cpu().tolist()
print(cnt, [int(v) for v in g[f"{case}/n"]])
for i, c in enumerate(cnt):
    a = out[i, :c].cpu().numpy(); d = g[f"{case}/out{i}"]
    bad = np.where((a != d).any(axis=1))[0]
    print("image", i, "rows", c, "first bad", bad[:5])
    if len(bad):
        r = bad[0]
        print(a[max(r-1,0):r+3]); print(d[max(r-1,0):r+3])
        # is a[r] somewhere in d?
        m = np.where((d == a[r]).all(axis=1))[0]; print("actual row found in desired at", m)
        m = np.where((a == d[r]).all(axis=1))[0]; print("desired row found in actual at", m)
