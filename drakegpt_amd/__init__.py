"""drakegpt_amd -- DrakeGPT's transformer training hot path on MI355X (gfx950) HIP kernels.

The public names (the reference's model / component classes) are resolved lazily so that
`python -m drakegpt_amd.build` can (re)build the library; the first access to any of them loads
libdrakegpt_hip.so and fails loudly if it is missing or stale -- there is no CPU or eager-PyTorch
fallback."""
import importlib

__version__ = "0.1.0"

_MODEL = ("MODEL_CLASSES", "BigramLM", "BlocksLM", "MultiHeadAttentionLM", "ResidualBlocksLM", "SingleHeadAttentionLM",
          "TransformerLM", "model_params")
_COMPONENT = ("Block", "FeedForward", "FeedForward2", "FeedForward3", "Head", "Head2", "MultiHeadAttention",
              "MultiHeadAttention2", "MultiHeadAttention3", "ResidualBlock", "ResidualBlock2")
__all__ = list(_MODEL + _COMPONENT)


def __getattr__(name):
    if name in _MODEL:
        return getattr(importlib.import_module(".model", __name__), name)
    if name in _COMPONENT:
        return getattr(importlib.import_module(".model_component", __name__), name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
