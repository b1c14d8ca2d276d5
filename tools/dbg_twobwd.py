import sys, copy, torch
sys.path.insert(0, '.')
import desenet_amd
from desenet_amd.core.models.yolo import Model
from desenet_amd.core.utils.loss import ComputeLoss, SegmentationLosses
from desenet_amd.core.utils.hyp import scale_hyp
from desenet_amd.synth import synth_images, synth_targets, synthetic_checkpoint
m = Model("desenet_s.yaml", ch=3, nc=6); sd = m.state_dict(); synthetic_checkpoint(sd); m.load_state_dict(sd); m = m.cuda()
x = synth_images(2, 128, 21).cuda(); det_t, seg_t = synth_targets(2, 128, 21); det_t, seg_t = det_t.cuda(), seg_t.cuda()
def run(mode):
    mm = copy.deepcopy(m).train(); mm.hyp = scale_hyp(6, 128)
    dp, sp = mm(x)
    dl, _ = ComputeLoss(mm)(dp, det_t); sl = SegmentationLosses()(sp, seg_t)
    if mode == 'A': (dl * 0.14 + sl).backward()
    elif mode == 'B': (dl * 0.14).backward(retain_graph=True); (sl * 1.0).backward()
    elif mode == 'D': (dl * 0.14).backward()
    elif mode == 'S': (sl * 1.0).backward()
    return {k: p.grad.clone() for k, p in mm.named_parameters() if p.grad is not None}
def cmp(a, b, tag):
    errs = sorted(((((a[k]-b[k]).abs().max()/(b[k].abs().max()+1e-30)).item(), k) for k in a), reverse=True)
    print(tag, errs[:6])
A, A2, B, D, S = run('A'), run('A'), run('B'), run('D'), run('S')
cmp(A2, A, 'A vs A')
cmp(B, A, 'B vs A')
DS = {k: D[k] + S[k] for k in D}
cmp(DS, A, 'D+S vs A')
cmp(B, DS, 'B vs D+S')
