"""Mirror of src/nsbench/models/__init__.py for the hot-path models."""
from .fno import FNOContextModule, FNOModule, TFNO2DModule  # noqa: F401
from .fourcastnet import AFNONet, FourCastNet  # noqa: F401
from .swin_transformer import SwinTransformer  # noqa: F401

__all__ = ["FNOContextModule", "FNOModule", "TFNO2DModule", "AFNONet", "FourCastNet", "SwinTransformer"]
