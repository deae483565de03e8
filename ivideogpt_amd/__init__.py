"""ivideogpt_amd -- MI355X-native (gfx950) prediction engine for iVideoGPT models.

Mirror of the reference's inference API backed by libivg.so (hand-written HIP).
There is no CPU / eager-PyTorch compute path: using a model without the built library, or on a
non-GPU device, raises.
"""
from .vq_model import CompressiveVQModel, DetokenizeCache  # noqa: F401
from .transformer import HeadModelWithAction, LlamaForCausalLM  # noqa: F401
from . import switches, weights  # noqa: F401

__all__ = ["CompressiveVQModel", "DetokenizeCache", "HeadModelWithAction", "LlamaForCausalLM", "switches", "weights"]
