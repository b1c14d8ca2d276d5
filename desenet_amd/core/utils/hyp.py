"""Training hyper-parameters of the hot path: the values of the reference's core/hyp/scratch.yaml that the losses read,
and their per-run scaling (scripts/train.py:258-260)."""

SCRATCH = dict(lr0=0.01, lrf=0.2, momentum=0.937, weight_decay=0.0005, warmup_epochs=3.0, warmup_momentum=0.8,
               warmup_bias_lr=0.1, box=0.05, cls=0.5, cls_pw=1.0, obj=0.7, obj_pw=1.0, iou_t=0.20, anchor_t=4.0,
               fl_gamma=0.0)
DETGAIN, SEGGAIN = 0.14, 1.0   # scripts/train.py:285


def scale_hyp(de_nc: int, imgsz: int, nl: int = 3, hyp=SCRATCH, label_smoothing: float = 0.0):
    h = dict(hyp)
    h["box"] *= 3.0 / nl
    h["cls"] *= de_nc / 80.0 * 3.0 / nl
    h["obj"] *= (imgsz / 640) ** 2 * 3.0 / nl
    h["label_smoothing"] = label_smoothing
    return h
