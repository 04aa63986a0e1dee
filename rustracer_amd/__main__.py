"""`python -m rustracer_amd` — the reference CLI (src/main.rs:35-132): same flags, defaults and
SCENE environment variable.  Only image mode (-i) renders; the window is out of scope."""
import argparse
import os
import sys
import time

from . import Config, Raytracer, RenderQuality, Scene
from ._lib import RT_DIRECT_HEMISPHERE, RT_DIRECT_IMPORTANCE, check, lib


def main(argv=None):
    ap = argparse.ArgumentParser(prog="rustracer", add_help=False)
    ap.add_argument("--help", action="help")
    ap.add_argument("-s", type=int, default=4, help="Sets how many samples per pixel to do")
    ap.add_argument("-l", type=int, default=4, help="Sets how many light samples to do for one-bounce radiance")
    ap.add_argument("-b", type=int, default=50, help="Sets the max amount of bounces to simulate")
    ap.add_argument("-w", type=int, default=600, help="Screen width")
    ap.add_argument("-h", type=int, default=600, help="Screen height")
    ap.add_argument("-d", dest="debug", action="store_true", help="Debug mode, where only intersections are shown")
    ap.add_argument("--high-dpi", action="store_true")
    ap.add_argument("-i", "--image-mode", action="store_true", help="dump a png instead of opening a window")
    ap.add_argument("--single-threaded", action="store_true")
    ap.add_argument("--seed", type=lambda v: int(v, 0), default=0x5EED, help="RNG seed (the reference is time-seeded)")
    ap.add_argument("--hemisphere", action="store_true", help="hemisphere direct-lighting estimator (README.md:9)")
    ap.add_argument("--obj-dir", default="obj")
    ap.add_argument("--preset", choices=list(RenderQuality.PRESETS), help="the (spp, light samples) pairs of the reference's quality panel (src/gui.rs:56-74)")
    a = ap.parse_args(argv)
    if a.preset:
        a.s, a.l = RenderQuality.get(a.preset)
    config = Config(screen_width=a.w, screen_height=a.h, samples_per_pixel=a.s, light_samples=a.l, bounces=a.b, debug=a.debug,
                    high_dpi=a.high_dpi, image_mode=a.image_mode, single_threaded=a.single_threaded, seed=a.seed,
                    direct_estimator=RT_DIRECT_HEMISPHERE if a.hemisphere else RT_DIRECT_IMPORTANCE)
    check(lib().rt_init(int(os.environ.get("LOCAL_RANK", "0"))))
    scene_name = os.environ.get("SCENE", "specular")
    load_start = time.perf_counter()
    scene = Scene.by_name(scene_name, a.obj_dir)
    print("Scene '%s' loaded in %.3fs" % (scene_name, time.perf_counter() - load_start))
    if not a.image_mode:
        print("interactive window is out of scope for this build; pass -i / --image-mode", file=sys.stderr)
        return 2
    Raytracer.new(config, scene).start()
    return 0


if __name__ == "__main__":
    sys.exit(main())
