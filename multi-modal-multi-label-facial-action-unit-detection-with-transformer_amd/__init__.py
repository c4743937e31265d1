"""MI355X-native implementation of the aural-visual transformer hot path of
ColinWine/Multi-modal-Multi-label-Facial-Action-Unit-Detection-with-Transformer.

The compute lives in ``lib/libavformer_hip.so`` (hand-written HIP for gfx950, C ABI in
``include/avformer_hip.h``); this package is the host-side mirror of the reference's
``Transformer`` / head / loss / model-registry surface.
"""
from . import _build, _lib, audio, checkpoint, dp, graphs, metrics, ops, optim  # noqa: F401
from ._lib import get_f32_arithmetic, set_f32_arithmetic  # noqa: F401
from .heads import AU_former, ResFormerTokens, TFormer, VA_former, former_AU_head, tformer_AU_head  # noqa: F401
from .loss import AULoss  # noqa: F401
from .models import (MODEL_REGISTRY, AudioFormer, SyntheticAVFormer, TwoStreamAuralVisualFormer,  # noqa: F401
                     VisualFormer, build_model)
from .transformer import Transformer  # noqa: F401

__all__ = ["Transformer", "AU_former", "VA_former", "tformer_AU_head", "former_AU_head", "TFormer", "AULoss",
           "ResFormerTokens", "TwoStreamAuralVisualFormer", "SyntheticAVFormer", "AudioFormer", "VisualFormer", "MODEL_REGISTRY",
           "build_model", "ops", "optim", "set_f32_arithmetic", "get_f32_arithmetic"]
