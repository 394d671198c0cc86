"""Token masks for masked pre-training: same functions and meaning as vit/tokens.py.

A mask is a bool tensor [B, L]; True = the token is kept (unmasked), False = it is dropped.  These are index
manipulations ahead of the encoder blocks (one gather of the patch tokens and one of the RoPE rows per
step); the blocks themselves see an ordinary, shorter sequence.
"""
from typing import Optional, Sequence, Union

import torch
import torch.nn.functional as F
from torch import Tensor

__all__ = ["mask_is_ragged", "apply_mask", "unapply_mask", "create_mask", "generate_non_overlapping_mask"]


def mask_is_ragged(mask: Tensor) -> bool:
    """True when the samples keep different numbers of tokens (vit/tokens.py:9-21)."""
    kept = mask.sum(dim=-1)
    return bool((kept != kept[0]).any())


def _kept_positions(mask: Tensor, width: int) -> Tensor:
    """[B, width] positions of the kept tokens of every sample in ascending order; when a sample keeps fewer
    than ``width`` tokens the tail holds positions of dropped tokens (the caller overwrites those slots)."""
    # stable descending sort of the 0/1 mask: kept positions first, each group in its original order
    order = torch.argsort(mask.to(torch.uint8), dim=1, descending=True, stable=True)
    return order[:, :width]


def apply_mask(mask: Tensor, x: Tensor, fill_value: Union[float, Tensor, None] = None,
               padding_value: Union[float, Tensor] = 0) -> Tensor:
    """vit/tokens.py:58-94.  x [B, L, D], mask [B, L].
    fill_value given: same shape out, dropped tokens replaced by it.  Otherwise dropped tokens are removed:
    [B, L', D] with L' the largest kept count; shorter samples are padded at the end with ``padding_value``."""
    if x.shape[:-1] != mask.shape:
        raise ValueError(f"Mask and input must match in all dimensions except the last: {x.shape} != {mask.shape}")
    B, L, D = x.shape
    if fill_value is not None:
        fill = fill_value.type_as(x) if isinstance(fill_value, Tensor) else fill_value
        return torch.where(mask.view(B, L, 1), x, fill)
    kept = mask.sum(dim=-1)
    width = int(kept.max()) if B > 0 else 0
    pos = _kept_positions(mask, width)
    out = torch.gather(x, 1, pos.unsqueeze(-1).expand(B, width, D))
    if bool((kept != width).any()):
        valid = torch.arange(width, device=x.device).view(1, width) < kept.view(B, 1)
        pad = padding_value.type_as(x) if isinstance(padding_value, Tensor) else padding_value
        out = torch.where(valid.unsqueeze(-1), out, pad)
    return out


def unapply_mask(mask: Tensor, x: Tensor) -> Tensor:
    """Inverse of apply_mask for a non-ragged mask: kept tokens back at their positions, zeros elsewhere
    (vit/tokens.py:97-104)."""
    assert not mask_is_ragged(mask), "Cannot unapply ragged mask"
    B, L = mask.shape
    D = x.shape[-1]
    width = x.shape[1]
    pos = _kept_positions(mask, width)
    out = x.new_zeros((B, L, D))
    return out.scatter(1, pos.unsqueeze(-1).expand(B, width, D), x)


def create_mask(size: Sequence[int], mask_ratio: float, batch_size: int = 1, scale: int = 1, roll: bool = False,
                device: torch.device = torch.device("cpu")) -> Tensor:
    """Random mask over a token grid: exactly round(L * mask_ratio) tokens dropped per sample, chosen
    independently per sample (vit/tokens.py:107-175).  scale > 1 draws the mask on the grid shrunk by ``scale`` and
    enlarges it, so tokens are dropped in scale x scale blocks; ``roll`` shifts half of those samples by scale // 2.
    The random calls follow the reference's order, so a seeded generator draws the reference's mask (pinned by
    tests/golden/g16_masks.pt on the CPU generator)."""
    if not 0 < mask_ratio < 1.0:
        raise ValueError(f"Invalid `mask_ratio` {mask_ratio}")
    if scale < 1:
        raise ValueError(f"Invalid `scale` {scale}")
    if any(dim <= scale for dim in size):
        raise ValueError(f"Invalid `size` {size} for `scale` {scale}. Size must be greater than scale")

    if scale > 1:
        small = tuple(s // scale for s in size)
        coarse = create_mask(small, mask_ratio, batch_size, scale=1, device=device)
        grid = coarse.view(batch_size, 1, *small).float()
        mask = F.interpolate(grid, scale_factor=scale, mode="nearest").view(batch_size, -1).bool()
        if roll:
            shifted = mask.roll(scale // 2, dims=1)
            pick = torch.rand(batch_size, device=device) < 0.5
            mask = torch.where(pick.view(-1, 1), shifted, mask)
        return mask

    total = 1
    for s in size:
        total *= int(s)
    dropped = int(torch.tensor(total * mask_ratio).round().item())
    # The random calls, in the reference's order, so that a seeded generator yields the reference's mask: one permutation
    # of the positions shared by all samples, then a [B, L] block of uniforms whose per-sample ranking picks which entries
    # of that permutation are dropped.
    order = torch.randperm(total, device=device)
    ranks = torch.argsort(torch.rand(batch_size, total, device=device, dtype=torch.float32), dim=-1)[:, :dropped]
    drop_pos = order[ranks]
    mask = torch.ones(batch_size, total, dtype=torch.bool, device=device)
    mask.scatter_(1, drop_pos, False)
    return mask


def generate_non_overlapping_mask(mask1: Tensor, p1: float, p2: float) -> Tensor:
    """Second selection of int(L * p2) positions per sample, none of them where ``mask1`` is True
    (vit/tokens.py:178-218)."""
    B, L = mask1.shape
    first, second = int(L * p1), int(L * p2)
    if first + second > L:
        raise ValueError("Cannot satisfy non-overlap constraint.")
    scores = torch.rand(B, L, device=mask1.device).masked_fill(mask1, 2.0)   # positions of mask1 sort last
    chosen = torch.argsort(scores, dim=1)[:, :second]
    mask2 = torch.zeros_like(mask1)
    mask2.scatter_(1, chosen, True)
    return mask2
