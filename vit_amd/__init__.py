"""vit_amd: MI355X-native (gfx950) ViT encoder forward/backward behind the TidalPaladin/vit Python API.

    from vit_amd import ViTConfig
    model = ViTConfig(...).instantiate(device="cuda")      # same fields, module tree and state_dict keys
    tokens = model(x).visual_tokens                        # every kernel in forward/backward is libvitk.so

Importing the package does not load the extension; the first kernel call does, and raises if it is missing."""
from .config import ViTConfig
from .features import ViTFeatures
from .head import AttentivePoolHead, AttentivePoolHeadConfig, Head, HeadConfig
from .vit import ViT

__all__ = ["ViT", "ViTConfig", "ViTFeatures", "Head", "HeadConfig", "AttentivePoolHead", "AttentivePoolHeadConfig"]
__version__ = "0.1.0"
