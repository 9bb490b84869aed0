from .gpsa import GPSA
from .vgpsa import VariationalGPSA

__all__ = ["GPSA", "VariationalGPSA"]
