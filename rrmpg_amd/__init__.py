"""rrmpg_amd -- MI355X-native ensemble rainfall-runoff engine.

Keeps the Python surface of kratzert/RRMPG (models.<Model>.simulate/.fit,
tools.monte_carlo) and runs the per-timestep model recurrences as hand-written
HIP kernels for gfx950 behind the C-ABI in include/rrhip.h.
"""

__version__ = "0.1.0"
