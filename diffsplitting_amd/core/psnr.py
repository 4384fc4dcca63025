"""The quality metrics the reference reports (core/psnr.py:44-82), on whatever device
the tensors live on: PSNR and the range-invariant PSNR of grayscale image batches (B,H,W)."""
import torch


def _flat(x):
    return x.reshape(x.shape[0], -1).to(torch.float32)


def _psnr(gt, pred, rng):
    mse = torch.mean((gt - pred) ** 2, dim=1)
    return 20 * torch.log10(rng / torch.sqrt(mse))


def PSNR(gt, pred, range_=None):
    assert gt.dim() == 3, "Images must be in shape: (batch,H,W)"
    gt, pred = _flat(torch.as_tensor(gt)), _flat(torch.as_tensor(pred))
    if range_ is None:
        range_ = gt.max(dim=1).values - gt.min(dim=1).values
    return _psnr(gt, pred, range_)


def RangeInvariantPsnr(gt, pred):
    """Rescales the prediction (least squares against the standardised ground truth)
    before computing PSNR (core/psnr.py:70-82)."""
    assert gt.dim() == 3, "Images must be in shape: (batch,H,W)"
    gt, pred = _flat(torch.as_tensor(gt)), _flat(torch.as_tensor(pred))
    std = gt.std(dim=1, keepdim=True)
    ra = (gt.max(dim=1).values - gt.min(dim=1).values) / std[:, 0]
    g = (gt - gt.mean(dim=1, keepdim=True)) / std
    g = g - g.mean(dim=1, keepdim=True)
    p = pred - pred.mean(dim=1, keepdim=True)
    alpha = (g * p).sum(dim=1, keepdim=True) / (p * p).sum(dim=1, keepdim=True)
    return _psnr(g, alpha * p, ra)
