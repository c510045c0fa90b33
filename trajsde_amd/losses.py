"""Loss values of the reference's SDE config, computed on the device tensors the forward returns
(losses/L2.py:10-27 winner-takes-all min-ADE regression; losses/diff_BCE.py:11-16 BCE on the encoder's diffusion
outputs, real = 0, perturbed = 1).  These classes are what the YAML's `losses:` entries resolve to: they give the
loss *values* (validation-time reporting, tests); `training_step` takes their names and weights and differentiates
the same losses inside the HIP backward entry points (trajsde_decoder_l2_backward / trajsde_decoder_nll_backward,
trajsde_encoder_backward).
"""
import torch
import torch.nn.functional as F


class L2:
    def __init__(self, reduction: str = "mean") -> None:
        if reduction != "mean":
            raise ValueError(f"{reduction} is not a valid value for reduction")
        self.reduction = reduction

    def __call__(self, data, output) -> torch.Tensor:
        target = data["y"]                                               # already rotated by forward (MODEL:83-84)
        loc = output["loc"][..., :2]
        reg_mask = output["reg_mask"]
        l2 = torch.norm(target.unsqueeze(0) - loc, p=2, dim=-1)         # [K, N, T]
        ade = (l2 * reg_mask.unsqueeze(0)).mean(-1)                      # masked steps count as zero, mean over all T
        best = ade.argmin(0)                                             # winner per actor
        min_l2 = l2.gather(0, best.view(1, -1, 1).expand(1, -1, l2.size(-1))).squeeze(0)
        n = reg_mask.sum()
        return (min_l2 * reg_mask).sum() / n if int(n) > 0 else min_l2.sum() * 0


class DiffBCE:
    def __init__(self, reduction: str = "mean") -> None:
        self.reduction = reduction

    def __call__(self, data, output) -> torch.Tensor:
        return (F.binary_cross_entropy(output["diff_in"], output["label_in"], reduction=self.reduction) +
                F.binary_cross_entropy(output["diff_out"], output["label_out"], reduction=self.reduction))


class LaplaceNLLLoss:
    """losses/laplace_nll_loss.py:18-47: Laplace negative log-likelihood of the winner-takes-all mode (the mode with the
    smallest masked mean L2), scale clamped at `eps`.  No shipped configuration names it (CFG:78-83 use L2 + DiffBCE), so
    this class gives its VALUE on the forward's device tensors; `training_step` differentiates it through
    trajsde_decoder_nll_backward (the scale head is trained under it, unlike under L2)."""

    def __init__(self, eps: float = 1e-6, reduction: str = "mean") -> None:
        if reduction != "mean":
            raise ValueError("{} is not a valid value for reduction".format(reduction))
        self.eps, self.reduction = float(eps), reduction

    def __call__(self, data, output) -> torch.Tensor:
        target = data["y"]
        loc, scale = output["loc"].chunk(2, dim=-1)
        reg_mask = output["reg_mask"]
        diff = torch.norm(target.unsqueeze(0) - loc, dim=-1) * reg_mask.unsqueeze(0)
        best = diff.mean(-1).argmin(0)
        idx = torch.arange(best.size(0), device=best.device)
        loc, scale = loc[best, idx], scale[best, idx].clamp(min=self.eps)
        nll = torch.log(2 * scale) + torch.abs(target - loc) / scale
        return nll[reg_mask].mean()
