"""Raytracer — host-side mirror of src/raytracer.rs (Raytracer, RaytracerInner, SharedPixelBuffer,
RenderConfig, RenderingMode) over the HIP library.  The bodies of `do_render` / `debug_render`
(src/raytracer.rs:167-221, 353-391) are one call into librustracer_hip.so; everything else keeps
the reference's names, defaults and state machine.
"""
import ctypes as C
import threading

import numpy as np

from . import _lib
from ._lib import check, lib, ptr
from .canvas import Canvas

CAMERA_SPEED = 2.0                      # src/raytracer.rs:18
RUSSIAN_ROULETTE_PROBABILITY = 0.7      # src/raytracer.rs:11 (compiled into the kernels)
DEFAULT_SEED = 0x5EED                   # BASELINE.md; the reference is time-seeded (src/common.rs:137)


class RenderingMode:
    Debug = _lib.RT_MODE_DEBUG
    Full = _lib.RT_MODE_FULL


class RenderQuality:
    """Render quality presets (src/gui.rs:35-85): name -> (samples per pixel, light samples).  The panel is out of scope; the table is what
    BASELINE's configs are quoted on (16 spp -> 8, 64 -> 16, 256 -> 32)."""
    PRESETS = {"preview": (1, 1), "low": (4, 4), "medium": (16, 8), "high": (64, 16), "ultra": (256, 32)}
    NAMES = {"preview": "Preview (1 spp)", "low": "Low (4 spp)", "medium": "Medium (16 spp)", "high": "High (64 spp)", "ultra": "Ultra (256 spp)"}

    @staticmethod
    def get(name):
        try:
            return RenderQuality.PRESETS[name]
        except KeyError:
            raise ValueError("--preset takes %s, not %r" % (" | ".join(RenderQuality.PRESETS), name))


class Config:
    """src/main.rs:13-32 with the same defaults."""

    def __init__(self, screen_width=600, screen_height=600, fov_degrees=90.0, samples_per_pixel=4, light_samples=4, bounces=50,
                 debug=False, high_dpi=False, image_mode=False, single_threaded=False, seed=DEFAULT_SEED,
                 direct_estimator=_lib.RT_DIRECT_IMPORTANCE):
        self.screen_width = screen_width
        self.screen_height = screen_height
        self.fov = float(np.float32(fov_degrees) * (np.float32(np.pi) / np.float32(180.0)))   # f32::to_radians, main.rs:95
        self.origin = (0.0, 0.0, 0.0)
        self.samples_per_pixel = samples_per_pixel
        self.light_samples = light_samples
        self.bounces = bounces
        self.debug = debug
        self.high_dpi = high_dpi
        self.image_mode = image_mode
        self.single_threaded = single_threaded   # accepted for CLI compatibility; the GPU path has no such mode
        self.seed = seed
        self.direct_estimator = direct_estimator


class RenderConfig:
    """src/raytracer.rs:88-96, 136-148"""

    def __init__(self, config):
        self.screen_width = config.screen_width
        self.screen_height = config.screen_height
        self.fov = config.fov
        self.samples_per_pixel = config.samples_per_pixel
        self.light_samples = config.light_samples
        self.bounces = config.bounces
        self.single_threaded = config.single_threaded
        self.seed = config.seed
        self.direct_estimator = config.direct_estimator


class SharedPixelBuffer:
    """src/raytracer.rs:22-67: RGBA8, row-major, (y*width + x)*4, top row first."""

    def __init__(self, width, height):
        self.width, self.height = width, height
        self.data = np.zeros(width * height * 4, np.uint8)

    def set_pixel(self, x, y, rgb8):
        index = (y * self.width + x) * 4
        if index + 3 < self.data.shape[0]:
            self.data[index:index + 3] = rgb8
            self.data[index + 3] = 255

    def snapshot(self, dst):
        assert dst.shape[0] == self.data.shape[0]
        np.copyto(dst, self.data)

    def clear(self):
        self.data[:] = 0
        self.data[3::4] = 255


def make_params(render_config, camera_position, mode, tile=(32, 32), shard=(0, 1)):
    """ScreenParams::from_config (src/raytracer.rs:110-124) + camera + settings -> rt_render_params."""
    p = _lib.RenderParams()
    lib().rt_screen_params(render_config.screen_width, render_config.screen_height, C.c_float(render_config.fov), C.byref(p))
    p.cam[0], p.cam[1], p.cam[2] = camera_position
    p.spp = render_config.samples_per_pixel
    p.light_samples = render_config.light_samples
    p.bounces = render_config.bounces
    p.mode = mode
    p.direct_estimator = render_config.direct_estimator
    p.seed = render_config.seed
    p.tile_w, p.tile_h = tile
    p.shard_index, p.shard_count = shard
    return p


class RaytracerInner:
    """src/raytracer.rs:69-84.  Public fields keep their names; atomics become ctypes words the
    library writes (render_progress) or polls (interrupt)."""

    def __init__(self, config, scene):
        self.config = RenderConfig(config)
        self._config_lock = threading.Lock()
        self.canvas = Canvas(config.screen_width, config.screen_height, config.high_dpi, config.image_mode)
        self.scene = scene
        self._scene_lock = threading.RLock()           # scene RwLock, src/raytracer.rs:72
        self.camera_position = [0.0, 0.0, 0.0]
        self._camera_lock = threading.Lock()
        self.rendering_mode = RenderingMode.Debug       # src/raytracer.rs:445
        self._interrupt = C.c_uint32(0)
        self.is_rendering = False
        self._progress = C.c_uint32(0)
        self.pixel_buffer = SharedPixelBuffer(config.screen_width, config.screen_height)
        self.last_render_interrupted = False

    @property
    def render_progress(self):
        return int(self._progress.value)

    def render(self):
        """src/raytracer.rs:159-165 -> do_render / debug_render."""
        self._interrupt.value = 0
        self.is_rendering = True
        self._progress.value = 0
        with self._camera_lock:
            cam = tuple(self.camera_position)
        with self._config_lock:
            params = make_params(self.config, cam, self.rendering_mode)
        with self._scene_lock:                           # held for the whole render, src/raytracer.rs:176
            rc = lib().rt_render(self.scene.handle, C.byref(params), ptr(self.pixel_buffer.data), C.byref(self._progress),
                                 C.byref(self._interrupt))
        self.is_rendering = False
        self.last_render_interrupted = rc == _lib.RT_ERR_INTERRUPTED
        if rc not in (_lib.RT_OK, _lib.RT_ERR_INTERRUPTED):
            check(rc)
        if rc == _lib.RT_OK:
            self._progress.value = 100

    def debug_render(self):
        mode, self.rendering_mode = self.rendering_mode, RenderingMode.Debug
        try:
            self.render()
        finally:
            self.rendering_mode = mode

    def test(self, i, j):
        """src/raytracer.rs:419-428: print one pixel's debug colour."""
        saved = self.pixel_buffer.data.copy()
        self.debug_render()
        px = self.pixel_buffer.data[(j * self.pixel_buffer.width + i) * 4:(j * self.pixel_buffer.width + i) * 4 + 3].copy()
        self.pixel_buffer.data[:] = saved
        print(tuple(int(v) for v in px))
        return px


class Raytracer:
    """src/raytracer.rs:431-532"""

    def __init__(self, config, scene):
        self.inner = RaytracerInner(config, scene)
        self._threads = []

    @classmethod
    def new(cls, config, scene):
        return cls(config, scene)

    def start(self):
        if not self.inner.canvas.image_mode:
            self.render(False)
        self.inner.canvas.start(self.inner)

    def move_camera(self, direction):
        with self.inner._camera_lock:
            c = self.inner.camera_position
            f = np.float32
            self.inner.camera_position = [float(f(c[k]) + f(direction[k]) * f(CAMERA_SPEED)) for k in range(3)]

    def reset_camera(self):
        with self.inner._camera_lock:
            self.inner.camera_position = [0.0, 0.0, 0.0]

    def toggle_rendering_mode(self):
        self.interrupt_render()
        self.inner.rendering_mode = RenderingMode.Full if self.inner.rendering_mode == RenderingMode.Debug else RenderingMode.Debug

    def interrupt_render(self):
        self.inner._interrupt.value = 1

    def update_render_settings(self, samples_per_pixel, light_samples, max_bounces):
        with self.inner._config_lock:
            self.inner.config.samples_per_pixel = samples_per_pixel
            self.inner.config.light_samples = light_samples
            self.inner.config.bounces = max_bounces

    def set_scene(self, scene):
        self.interrupt_render()
        with self.inner._scene_lock:                     # blocks until the in-flight render has returned
            self.inner.scene = scene

    def render(self, wait_for_completion):
        if wait_for_completion:
            self.inner.render()
        else:
            t = threading.Thread(target=self.inner.render, daemon=True)
            self._threads.append(t)
            t.start()

    def wait(self):
        for t in self._threads:
            t.join()
        self._threads = []
