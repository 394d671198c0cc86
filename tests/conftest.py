import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this process")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name: str) -> dict:
    return torch.load(os.path.join(GOLDEN, name + ".pt"), weights_only=True)


def golden_names() -> list[str]:
    """Encoder fixtures (g07_ops holds op-level vectors, g12_* the head fixtures, g13_* the decoder-layer fixtures, g16_masks
    the token masks)."""
    return sorted(f[:-3] for f in os.listdir(GOLDEN)
                  if f.endswith(".pt") and f not in ("g07_ops.pt", "g16_masks.pt") and not f.startswith(("g12_", "g13_")))


def layer_golden_names() -> list[str]:
    """Decoder / cross-attention layer fixtures."""
    return sorted(f[:-3] for f in os.listdir(GOLDEN) if f.endswith(".pt") and f.startswith("g13_"))


def head_golden_names() -> list[str]:
    return sorted(f[:-3] for f in os.listdir(GOLDEN) if f.endswith(".pt") and f.startswith("g12_"))
