"""alias of the reference's ``gpsa.plotting`` (gpsa/plotting/__init__.py:1-6): names only, see callbacks.py"""
from gpsa.plotting.callbacks import (  # noqa: F401
    callback_oned,
    callback_twod,
    callback_twod_aligned_only,
    callback_twod_multimodal,
)
