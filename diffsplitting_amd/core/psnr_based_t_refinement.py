"""TimePredictor-driven refinement (core/psnr_based_t_refinement.py of the reference; notebook
EvaluateJointIndi.ipynb cells 42-62), batched on the MI355X:

  1. the time classifier predicts the mixing time of every tile            (one batched TimePredictor forward)
  2. indi_1 / indi_2 turn the input into the two channel estimates, each tile started at ITS predicted time
     (one batched loop per sampler with per-sample step tables; the reference loops over the batch with batch 1)
  3. optional MMSE: the mean over ``mmse_count`` independent repeats         (cells 60-62)
  4. the mixing time is re-estimated by scanning t in [0, 1) for the best RangeInvariantPsnr of
     t * ch1 + (1 - t) * ch2 against the input                             (:41-57)

Function names and return values follow the reference module; the reference's own file cannot be imported anywhere
(it imports the external ``disentangle`` package, :10) and calls ``p_sample_loop`` — the old name of ``inference``.
"""
import numpy as np
import torch

from .psnr import RangeInvariantPsnr


@torch.no_grad()
def get_time_prediction_from_classifier(inp, time_classifier):
    return time_classifier(inp.cuda())                                # :14-17


@torch.no_grad()
def get_channel_estimates(inp, indi_1, indi_2, time_classifier, num_timesteps=1, mmse_count=1, as_numpy=True):
    """:20-39.  for classifier input = t * c1 + (1 - t) * c2; indi_1 (target c1 at time 0) starts at 1 - t."""
    inp = inp.cuda().float()
    pred_t_2 = get_time_prediction_from_classifier(inp, time_classifier)
    pred_t_1 = 1 - pred_t_2
    acc1 = acc2 = None
    for _ in range(int(mmse_count)):
        indi_1.inference(inp, continuous=False, num_timesteps=num_timesteps, t_float_start=pred_t_1)
        ch1 = indi_1.last_full_batch.clone()
        indi_2.inference(inp, continuous=False, num_timesteps=num_timesteps, t_float_start=pred_t_2)
        ch2 = indi_2.last_full_batch.clone()
        acc1 = ch1 if acc1 is None else acc1 + ch1
        acc2 = ch2 if acc2 is None else acc2 + ch2
    pred1, pred2 = acc1 / mmse_count, acc2 / mmse_count               # MMSE estimate: mean over the repeats
    if as_numpy:
        return pred1.cpu().numpy(), pred2.cpu().numpy()
    return pred1, pred2


@torch.no_grad()
def estimate_time_using_PSNR(inp, indi_1, indi_2, time_classifier, num_timesteps=1, mmse_count=1):
    """:41-57.  inp: (B, 1, H, W) normalised input.  Returns (per_sample_t, concensus_t)."""
    pred1, pred2 = get_channel_estimates(inp, indi_1, indi_2, time_classifier, num_timesteps, mmse_count, as_numpy=False)
    gt = inp.cuda().float()[:, 0]
    t_list = np.arange(0, 1.0, 0.05)
    psnr_list = []
    for t in t_list:
        pred = pred1 * float(t) + pred2 * float(1 - t)
        psnr_list.append(RangeInvariantPsnr(gt, pred[:, 0]))
    psnr_matrix = torch.stack(psnr_list).cpu()
    per_sample_t = t_list[psnr_matrix.argmax(dim=0).numpy()]
    concensus_t = t_list[int(psnr_matrix.mean(dim=1).argmax())]
    return per_sample_t, concensus_t
