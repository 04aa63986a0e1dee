"""Canvas — headless half of src/canvas.rs (image mode, src/canvas.rs:41-110).

The SDL2/OpenGL/egui window (src/canvas.rs:113-449) is out of scope; `start` with
image_mode=False raises.
"""
import os
import struct
import time
import zlib

import numpy as np


def encode_png_rgb8(rgb):
    """RGB8 PNG (color type 2, bit depth 8), as png::Encoder is configured at src/canvas.rs:53-56."""
    h, w, _ = rgb.shape
    raw = np.concatenate([np.zeros((h, 1), np.uint8), np.ascontiguousarray(rgb, np.uint8).reshape(h, w * 3)], axis=1).tobytes()

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 6))
            + chunk(b"IEND", b""))


class Canvas:
    def __init__(self, width, height, high_dpi=False, image_mode=True):
        self.width, self.height, self.high_dpi, self.image_mode = width, height, high_dpi, image_mode
        self.last_path = None

    @staticmethod
    def rgba_to_rgb(rgba, width, height):
        """src/canvas.rs:68-77"""
        return np.ascontiguousarray(np.asarray(rgba, np.uint8).reshape(height, width, 4)[:, :, :3])

    def save_canvas(self, pixel_buffer_rgb):
        """src/canvas.rs:41-65"""
        os.makedirs("./dump", exist_ok=True)
        path_string = "./dump/%d.png" % int(time.time())
        print("Saving with filename %s" % path_string)
        with open(path_string, "wb") as f:
            f.write(encode_png_rgb8(pixel_buffer_rgb))
        print("Image saved successfully!")
        self.last_path = path_string
        return path_string

    def start(self, raytracer_inner):
        if self.image_mode:
            return self.start_image_mode(raytracer_inner)
        raise NotImplementedError("interactive SDL2 window is out of scope; run with image_mode=True (-i)")

    def start_image_mode(self, raytracer_inner):
        """src/canvas.rs:88-110"""
        from .raytracer import RenderingMode
        raytracer_inner.rendering_mode = RenderingMode.Full
        print("Rendering image...")
        start = time.perf_counter()
        raytracer_inner.render()
        print("Render time: %.3fs" % (time.perf_counter() - start))
        rgba = np.zeros(self.width * self.height * 4, np.uint8)
        raytracer_inner.pixel_buffer.snapshot(rgba)
        return self.save_canvas(self.rgba_to_rgb(rgba, self.width, self.height))
