import sys; sys.path.insert(0,'/root/repo')
import torch, yaml, numpy as np
from oracle import desenet_ref as R, loss_ref
from desenet_amd.synth import synthetic_checkpoint, synth_images, synth_targets
cfg=yaml.safe_load(open('/root/repo/desenet_amd/cfg/desenet_s.yaml'))
sd0=R.make_state_dict(cfg); synthetic_checkpoint(sd0)
isp=lambda k: 'running' not in k and 'num_batches' not in k and 'anchor' not in k
def run(bs,size,seed,bf16):
    sd={k:v.clone().requires_grad_(isp(k)) for k,v in sd0.items()}
    x=synth_images(bs,size,seed); dt,st=synth_targets(bs,size,seed)
    with torch.autocast('cpu',dtype=torch.bfloat16,enabled=bf16):
        raws,seg,_=R.forward(cfg,sd,x,training=True)
    raws=[r.float() for r in raws]; seg=seg.float()
    total,dl,items,sl=loss_ref.step_loss(raws,seg,dt,st,sd['model.25.anchors'],6,size)
    total.backward()
    g={k:v.grad for k,v in sd.items() if isp(k) and v.grad is not None}
    return raws,seg,dl,sl,g
def rel(a,b): return ((a.double()-b.double()).abs().max()/(b.double().abs().max()+1e-12)).item()
for bs,size,seed in [(2,128,21),(1,640,3)]:
    r32=run(bs,size,seed,False); r16=run(bs,size,seed,True)
    print(size,'raw rel',[rel(a,b) for a,b in zip(r16[0],r32[0])],'seg',rel(r16[1],r32[1]),'det_loss',r16[2].item(),r32[2].item(),'seg_loss',r16[3].item(),r32[3].item())
    n32=np.sqrt(sum((v.double()**2).sum().item() for v in r32[4].values())); n16=np.sqrt(sum((v.double()**2).sum().item() for v in r16[4].values()))
    print('  grad l2',n16,n32, 'per-param rel err median/max', np.median([rel(r16[4][k],r32[4][k]) for k in r32[4]]), max(rel(r16[4][k],r32[4][k]) for k in r32[4]))
