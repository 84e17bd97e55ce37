"""drakegpt_amd -- DrakeGPT's transformer training hot path on MI355X (gfx950) HIP kernels.

Importing this package loads libdrakegpt_hip.so and fails loudly if it has not been built
(`python -m drakegpt_amd.build`); there is no CPU or eager-PyTorch fallback."""
from . import _lib  # noqa: F401  (raises if the HIP library is missing)
from .model import (MODEL_CLASSES, BigramLM, BlocksLM, MultiHeadAttentionLM, ResidualBlocksLM,  # noqa: F401
                    SingleHeadAttentionLM, TransformerLM, model_params)
from .model_component import (Block, FeedForward, FeedForward2, FeedForward3, Head, Head2,  # noqa: F401
                              MultiHeadAttention, MultiHeadAttention2, MultiHeadAttention3, ResidualBlock,
                              ResidualBlock2)

__version__ = "0.1.0"
