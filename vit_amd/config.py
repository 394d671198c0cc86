"""ViTConfig: the reference's configuration dataclass, field for field (vit/vit.py:67-152), including the
YAML round trip and the validation errors.  ``rope_normalize_coords`` is kept although the reference never
forwards it to the RoPE module (vit/vit.py:284-292): the model always uses "separate"."""
from __future__ import annotations

from dataclasses import dataclass, field
from pathlib import Path
from typing import Any, Optional, Sequence, Union

import torch
import yaml

from .norm import NORM_TYPE_CHOICES

_DTYPES = {"float32": torch.float32, "float16": torch.float16, "bfloat16": torch.bfloat16, "float64": torch.float64}


def _dtype_from_string(value):
    return _DTYPES.get(value) if isinstance(value, str) else value


@dataclass(frozen=True)
class ViTConfig:
    in_channels: int
    patch_size: Sequence[int]
    img_size: Sequence[int]
    depth: int
    hidden_size: int
    ffn_hidden_size: int
    num_attention_heads: int
    hidden_dropout: float = 0.1
    attention_dropout: float = 0.1
    attention_bias: bool = True
    mlp_bias: bool = True
    activation: str = "srelu"
    drop_path_rate: float = 0.0
    num_register_tokens: int = 0
    num_cls_tokens: int = 0
    pos_enc: str = "rope"
    layer_scale: Optional[float] = None
    glu_limit: Optional[float] = None
    glu_extra_bias: Optional[float] = None
    rope_normalize_coords: str = "separate"
    rope_base: float = 100
    rope_shift_coords: Optional[float] = None
    rope_jitter_coords: Optional[float] = None
    rope_rescale_coords: Optional[float] = None
    mlp_requires_grad: bool = True
    self_attention_requires_grad: bool = True
    activation_checkpointing: bool = False
    dtype: torch.dtype = torch.bfloat16
    norm_type: str = "rmsnorm"
    qk_normalization: bool = False
    patch_embed_normalization: bool = False
    heads: dict = field(default_factory=dict)

    def __post_init__(self) -> None:
        if self.hidden_size % self.num_attention_heads != 0:
            raise ValueError(f"hidden_size ({self.hidden_size}) must be divisible by "
                             f"num_attention_heads ({self.num_attention_heads})")
        if self.norm_type not in NORM_TYPE_CHOICES:
            raise ValueError(f"Unsupported norm_type: {self.norm_type}")
        if self.pos_enc == "fourier" and self.hidden_size % 2 != 0:
            raise ValueError(f"hidden_size ({self.hidden_size}) must be even when using Fourier positional encoding")

    def instantiate(self, device: Optional[torch.device] = None):
        from .vit import ViT

        return ViT(self, device=device)

    @classmethod
    def from_yaml(cls, path: Union[str, Path]):
        if isinstance(path, Path):
            if not path.is_file():
                raise FileNotFoundError(f"File not found: {path}")
            data = yaml.full_load(path.read_text())
        elif isinstance(path, str) and path.endswith(".yaml"):
            return cls.from_yaml(Path(path))
        else:
            data = yaml.full_load(path)
        if isinstance(data, cls):
            return data
        data = dict(data)
        data["dtype"] = _dtype_from_string(data.get("dtype", torch.bfloat16))
        return cls(**data)

    def to_yaml(self) -> str:
        data: dict[str, Any] = dict(self.__dict__)
        if isinstance(data.get("dtype"), torch.dtype):
            data["dtype"] = str(data["dtype"]).replace("torch.", "")
        return yaml.dump(data)


def _construct(loader, node):
    values = loader.construct_mapping(node, deep=True)
    values["dtype"] = _dtype_from_string(values.get("dtype", torch.bfloat16))
    return ViTConfig(**values)


for _tag in ("tag:yaml.org,2002:python/object:vit.vit.ViTConfig", "tag:yaml.org,2002:python/object:vit.ViTConfig",
             "tag:yaml.org,2002:python/object:vit_amd.ViTConfig"):
    for _loader in (yaml.SafeLoader, yaml.FullLoader, yaml.UnsafeLoader):
        _loader.add_constructor(_tag, _construct)
