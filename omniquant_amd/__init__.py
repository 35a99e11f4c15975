"""omniquant_amd -- MI355X-native (gfx950) OmniQuant calibration hot path.

Drop-in surface (same names / signatures / state-dict keys as the reference):
    UniformAffineQuantizer  <- quantize/quantizer.py
    QuantLinear             <- quantize/int_linear.py
    QuantMatMul             <- quantize/int_matmul.py
    OmniLayerNorm, OmniLlamaRMSNorm <- quantize/omni_norm.py
    QuantLlamaDecoderLayer / QuantOPTDecoderLayer <- models/int_{llama,opt}_layer.py
    omniquant()             <- quantize/omniquant.py
All compute goes through hand-written HIP kernels behind the C ABI in include/oq_hip.h; there is no CPU or
eager fallback (a missing library raises).
"""
from ._capi import OQError, load as load_library  # noqa: F401
from .quantizer import UniformAffineQuantizer, round_ste  # noqa: F401
from .linear import QuantLinear  # noqa: F401
from .matmul import QuantMatMul  # noqa: F401
from .norm import OmniLayerNorm, OmniLlamaRMSNorm  # noqa: F401

__version__ = "0.1.0"
