from .cider import Cider

__all__ = ["Cider"]
