"""pylib/Criterion.py of the reference, on the GPU."""
from ._dev import lib, check, ptr, stream, dev, to_dev, torch


def weighted_L2(pred, gt, weight):
    """((pred - gt)^2 * weight).sum() / numel  (pylib/Criterion.py:12-18).  `weight` may be a scalar
    1 (the inline loss of stack-hg.py:156-159) or a tensor of pred's shape.  Returns a 0-d GPU tensor."""
    p = to_dev(pred, torch.float32)
    g = to_dev(gt, torch.float32)
    w = None
    if isinstance(weight, torch.Tensor) and weight.numel() > 1:
        w = to_dev(weight.expand_as(p) if weight.shape != p.shape else weight, torch.float32)
    scale = 1.0
    if w is None:
        scale = float(weight) if not isinstance(weight, torch.Tensor) else float(weight.reshape(-1)[0])
    loss = torch.zeros(1, dtype=torch.float32, device=dev())
    check(lib().pa_weighted_l2(ptr(p), ptr(g), ptr(w), p.numel(), ptr(loss), stream()), 'pa_weighted_l2')
    return loss[0] * scale
