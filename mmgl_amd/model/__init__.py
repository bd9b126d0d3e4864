"""Same exports as the reference's model/__init__.py:1-2."""
from .modelling_self_attention import SelfAttentionModel
from .modelling_cross_attention import CrossAttentionModel

__all__ = ["SelfAttentionModel", "CrossAttentionModel"]
