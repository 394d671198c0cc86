"""ViTFeatures: zero-copy views over the encoder output [B, S, D] laid out as cls | register | patch tokens
(reference: vit/vit.py:155-251)."""
from __future__ import annotations

from typing import Callable, Iterator, Optional, Sequence

import torch
from torch import Tensor


class ViTFeatures:
    def __init__(self, dense_features: Tensor, num_register_tokens: int, num_cls_tokens: int,
                 tokenized_size: Optional[Sequence[int]] = None):
        self._dense = dense_features
        self._nreg = num_register_tokens
        self._ncls = num_cls_tokens
        self._grid = None if tokenized_size is None else tuple(tokenized_size)

    def __repr__(self) -> str:
        return (f"{type(self).__name__}(cls_tokens={tuple(self.cls_tokens.shape)}, "
                f"register_tokens={tuple(self.register_tokens.shape)}, visual_tokens={tuple(self.visual_tokens.shape)})")

    def __iter__(self) -> Iterator[Tensor]:
        return iter((self.cls_tokens, self.register_tokens, self.visual_tokens))

    dense_features = property(lambda self: self._dense)
    num_register_tokens = property(lambda self: self._nreg)
    num_cls_tokens = property(lambda self: self._ncls)
    tokenized_size = property(lambda self: self._grid)

    @property
    def cls_tokens(self) -> Tensor:
        return self._dense[..., : self._ncls, :]

    @property
    def register_tokens(self) -> Tensor:
        return self._dense[..., self._ncls : self._ncls + self._nreg, :]

    @property
    def visual_tokens(self) -> Tensor:
        return self._dense[..., self._ncls + self._nreg :, :]

    @property
    def visual_tokens_as_grid(self) -> Tensor:
        if self._grid is None:
            raise ValueError("tokenized_size is not set, cannot reshape to grid")
        v = self.visual_tokens
        return v.view(v.shape[0], *self._grid, v.shape[-1])

    def apply(self, func: Callable[[Tensor], Tensor]) -> "ViTFeatures":
        return type(self)(func(self._dense), self._nreg, self._ncls, self._grid)

    @classmethod
    def from_separate_features(cls, cls_tokens: Tensor, register_tokens: Tensor, visual_tokens: Tensor,
                               tokenized_size: Optional[Sequence[int]] = None) -> "ViTFeatures":
        return cls(torch.cat([cls_tokens, register_tokens, visual_tokens], dim=1), register_tokens.shape[1],
                   cls_tokens.shape[1], tokenized_size)
