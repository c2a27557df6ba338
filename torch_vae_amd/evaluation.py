"""Forward-only consumer of the hot path: mirror of the reference's ``evaluation.evaluate``
(evaluation.py:12-113).  Same signature and result keys; the model runs in eval mode (BatchNorm with
running statistics) through the HIP kernels; MSE / MAE are reduced on the device instead of with
sklearn on the host (same definitions, converted to percentages as the reference does)."""
from __future__ import annotations

import torch


def evaluate(dataloader, model, device, partition_name="Val", verbosity=1):
    model.eval()
    n_seen = 0
    se = torch.zeros((), device=device, dtype=torch.float64)
    ae = torch.zeros((), device=device, dtype=torch.float64)
    n_elem = 0
    stim_min, stim_max, rec_min, rec_max = float("inf"), float("-inf"), float("inf"), float("-inf")
    n_samples = len(dataloader.dataset) if hasattr(dataloader, "dataset") else None
    for stimuli, _ in dataloader:
        stimuli = stimuli.to(device)
        if n_samples is not None and n_seen + stimuli.shape[0] > n_samples:
            stimuli = stimuli[: n_samples - n_seen]      # trim DistributedSampler padding (evaluation.py:88-95)
            if stimuli.shape[0] == 0:
                break
        with torch.no_grad():
            output = model(stimuli)
        rec = output["output"]
        d = (rec - stimuli).double()
        se += (d * d).sum()
        ae += d.abs().sum()
        n_elem += d.numel()
        n_seen += stimuli.shape[0]
        stim_min, stim_max = min(stim_min, float(stimuli.min())), max(stim_max, float(stimuli.max()))
        rec_min, rec_max = min(rec_min, float(rec.min())), max(rec_max, float(rec.max()))
    if verbosity >= 1:
        print(f"input has range  [{stim_min:.03f}, {stim_max:.03f}]")
        print(f"output has range [{rec_min:.03f}, {rec_max:.03f}]")
    results = {"count": n_seen}
    # F.cross_entropy(reconstruction, stimuli) over the single channel C=1 is identically 0 (evaluation.py:66)
    results["cross-entropy"] = 0.0
    results["mse"] = 100.0 * float(se) / max(n_elem, 1)
    results["mae"] = 100.0 * float(ae) / max(n_elem, 1)
    if verbosity >= 1:
        print(f"\n{partition_name} evaluation results:")
        for k, v in results.items():
            if "count" in k:
                print(f"  {k + ' ':.<21s}{v:7d}")
            elif "entropy" in k:
                print(f"  {k + ' ':.<24s} {v:9.5f} nat")
            else:
                print(f"  {k + ' ':.<24s} {v:6.2f} %")
    return results
