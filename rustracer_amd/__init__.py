"""rustracer_amd — MI355X (gfx950) implementation of rustracer's per-pixel trace/shade path.

Host-side mirror of the reference API (Scene / Raytracer / Canvas) over librustracer_hip.so
(C ABI: include/rustracer_hip.h).  See DESIGN.md and INTEGRATION.md.
"""
from ._lib import RtError, RenderParams, Material  # noqa: F401
from .scene import Scene, dragon_standin_mesh, DRAGON_STANDIN_FULL, DRAGON_STANDIN_SMALL  # noqa: F401
from .raytracer import Config, RenderConfig, RenderQuality, RenderingMode, SharedPixelBuffer, Raytracer, RaytracerInner, make_params  # noqa: F401
from .canvas import Canvas  # noqa: F401
