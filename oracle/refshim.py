"""TEST INFRASTRUCTURE, container-only: make /root/reference importable on Python 3.10 without torchao.

Nothing in the product (vit_amd/) imports this file.  It exists so that the golden-vector generator
and the restatement checker can run the live reference in the build container.  It refuses to work
when /root/reference is absent (for example on the GPU box).

What it does, all in memory:
  * typing.Self / datetime.UTC aliases (the reference needs Python >= 3.11: vit/vit.py:4,
    benchmark/component_benchmark.py:14);
  * an empty stand-in module named ``torchao`` so that the top-level imports at vit/attention.py:7 and
    vit/fused.py:8 succeed.  torchao is only used by ``apply_quantization`` (int8 inference), which is
    outside the encoder hot path; the stand-in raises if it is ever called, so no arithmetic on the path
    is provided by it;
  * disables TorchDynamo exactly as the reference's own tests do (tests/conftest.py:9-13), so the
    ``@torch.compile`` decorators are inert and eager ATen semantics are the specification.
"""
import datetime
import os
import sys
import types
import typing

REFERENCE_ROOT = "/root/reference"


def install() -> None:
    if not os.path.isdir(os.path.join(REFERENCE_ROOT, "vit")):
        raise RuntimeError(f"{REFERENCE_ROOT} is not present: the live reference can only run in the build container")
    os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")
    if not hasattr(typing, "Self"):
        import typing_extensions

        typing.Self = typing_extensions.Self  # type: ignore[attr-defined]
    if not hasattr(datetime, "UTC"):
        datetime.UTC = datetime.timezone.utc  # type: ignore[attr-defined]
    if "torchao" not in sys.modules:

        def _absent(*_a, **_k):
            raise NotImplementedError("torchao is not installed; quantization is outside the hot path")

        ao = types.ModuleType("torchao")
        q = types.ModuleType("torchao.quantization")
        q.quantize_ = _absent
        q.Int8WeightOnlyConfig = type("Int8WeightOnlyConfig", (), {})
        q.Int8DynamicActivationInt8WeightConfig = type("Int8DynamicActivationInt8WeightConfig", (), {})
        d = types.ModuleType("torchao.dtypes")
        d.AffineQuantizedTensor = type("AffineQuantizedTensor", (), {})
        ao.quantization = q
        ao.dtypes = d
        sys.modules["torchao"] = ao
        sys.modules["torchao.quantization"] = q
        sys.modules["torchao.dtypes"] = d
    import torch

    torch._dynamo.config.disable = True
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def import_reference():
    install()
    import vit as ref  # noqa: PLC0415

    return ref
