"""pylib/HumanAcc.py of the reference, on the GPU (only approx_PCKh is named by the hot path)."""
from ._dev import lib, check, ptr, stream, dev, to_dev, torch


def approx_PCKh(pred, target, idxs, res):
    """pylib/HumanAcc.py:7-44.  pred/target: b x n x 2; valid where target > 0; normaliser res/10 with
    the reference's Python-2 INTEGER division (64 -> 6); threshold 0.5; mean over joints with >=1 valid."""
    p = to_dev(pred, torch.float32)
    t = to_dev(target, torch.float32)
    assert p.shape == t.shape
    B, J = p.shape[0], p.shape[1]
    norm = torch.full((B,), float(int(res) // 10), dtype=torch.float32, device=dev())
    ix = torch.as_tensor(list(idxs), dtype=torch.int32, device=dev())
    acc = torch.zeros(len(idxs) + 1, dtype=torch.float32, device=dev())
    check(lib().pa_pck(ptr(p), ptr(t), ptr(norm), 0.0, ptr(ix), len(idxs), 0.5, None, B, J, ptr(acc), None, None, stream()), 'pa_pck')
    return float(acc[0])
