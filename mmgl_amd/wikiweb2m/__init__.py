"""Same exports as the reference's wikiweb2m/__init__.py."""
from .data import WikiWeb2M, load_wikiweb2m

__all__ = ["WikiWeb2M", "load_wikiweb2m"]
