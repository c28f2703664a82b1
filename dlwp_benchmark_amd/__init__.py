"""dlwp_benchmark_amd — MI355X-native rollout training path for dlwp-benchmark backbones.

Only the hot path of SURVEY.md §8 lives here: HIP kernels + C ABI (csrc/, include/dlwpmi.h)
and the thin Python host side that mirrors the reference's nn.Module surface.
"""
from . import lib  # noqa: F401  (raises loudly on use if libdlwpmi.so is missing)

__all__ = ["lib"]
