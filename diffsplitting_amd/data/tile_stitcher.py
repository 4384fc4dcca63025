"""data/tile_stitcher.py of the reference: ``stitch_predictions(predictions, idx_manager)``
-> (N,H,W,C), pasted by the HIP stitch kernel (dsx_stitch)."""
import numpy as np
import torch


def stitch_predictions(predictions, idx_manager):
    """``predictions``: (T,C,ph,pw) numpy array or tensor.  Returns the same kind."""
    was_numpy = isinstance(predictions, np.ndarray)
    t = torch.from_numpy(np.ascontiguousarray(predictions)) if was_numpy else predictions
    dtype = t.dtype
    out = idx_manager.plan.stitch(t.to(device="cuda", dtype=torch.float32))
    out = out.to(dtype)
    return out.cpu().numpy() if was_numpy else out
