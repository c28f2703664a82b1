"""Mirror of src/dlwpbench/models/__init__.py for the hot-path models."""
from .fno import FNO2DModule, TFNO2DModule  # noqa: F401
from .fourcastnet import AFNONet, FourCastNet, FourCastNetv2, SFNONet  # noqa: F401
from .panguweather import PanguWeather  # noqa: F401
from .sfno import SFNO2DModule  # noqa: F401
from .swin_transformer import SwinTransformer  # noqa: F401

__all__ = ["FNO2DModule", "TFNO2DModule", "SFNO2DModule", "AFNONet", "FourCastNet", "FourCastNetv2", "SFNONet", "PanguWeather",
           "SwinTransformer"]
