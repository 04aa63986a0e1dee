"""The C++ host mirror of the reference binary (rustracer_amd/host: Scene / Raytracer / Canvas / Config over the C ABI).

CPU part: command-line handling (src/main.rs:35-105) and the PNG writer (src/canvas.rs:41-65), no GPU touched.
GPU part: `SCENE=<name> rustracer -i ...` renders headless exactly like the reference's image mode
(src/canvas.rs:88-110), and the PNG it leaves in ./dump equals the oracle's render of the same scene, byte for byte."""
import os
import re
import struct
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "rustracer_amd", "rustracer")


def read_png_rgb8(path):
    b = open(path, "rb").read()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, hdr = 8, b"", None
    while pos < len(b):
        n, = struct.unpack(">I", b[pos:pos + 4])
        tag, data = b[pos + 4:pos + 8], b[pos + 8:pos + 8 + n]
        crc, = struct.unpack(">I", b[pos + 8 + n:pos + 12 + n])
        assert zlib.crc32(tag + data) & 0xFFFFFFFF == crc, tag
        if tag == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", data)
        elif tag == b"IDAT":
            idat += data
        pos += 12 + n
    w, h, depth, ctype = hdr[:4]
    assert (depth, ctype) == (8, 2)                       # png::BitDepth::Eight, ColorType::RGB (canvas.rs:54-55)
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + 3 * w)
    assert (raw[:, 0] == 0).all()                         # filter type 0 on every scanline
    return raw[:, 1:].reshape(h, w, 3)


def test_executable_is_built():
    assert os.path.exists(EXE), "build it with `make -C rustracer_amd/host` (or __graft_entry__.build())"


def test_png_writer(tmp_path):
    out = tmp_path / "g.png"
    subprocess.check_call([EXE, "--png-selftest", str(out)])
    img = read_png_rgb8(str(out))
    assert img.shape == (21, 37, 3)
    y, x = np.mgrid[0:21, 0:37]
    assert np.array_equal(img[:, :, 0], (x * 7).astype(np.uint8)) and np.array_equal(img[:, :, 1], (y * 12).astype(np.uint8))
    assert np.array_equal(img[:, :, 2], (x ^ y).astype(np.uint8))


def test_command_line_errors_and_help():
    r = subprocess.run([EXE, "--help"], capture_output=True, text=True)
    assert r.returncode == 0 and "-s <s>" in r.stdout and "--image-mode" in r.stdout and "-h <h>        Screen height" in r.stdout
    r = subprocess.run([EXE, "-q"], capture_output=True, text=True)
    assert r.returncode == 2 and "wasn't expected" in r.stderr
    r = subprocess.run([EXE, "-i", "-s"], capture_output=True, text=True)
    assert r.returncode == 2 and "requires a value" in r.stderr
    r = subprocess.run([EXE, "-i", "--preset", "best"], capture_output=True, text=True)
    assert r.returncode == 2 and "preview | low | medium | high | ultra" in r.stderr


def test_quality_presets_are_the_reference_panels():
    """The (samples per pixel, light samples) pairs of the reference's quality panel (src/gui.rs:56-74), which BASELINE's configs are quoted
    on: 16 spp -> 8 light samples, 64 -> 16, 256 -> 32.  Python mirror and `--preset` of both command lines."""
    from rustracer_amd.raytracer import RenderQuality
    assert RenderQuality.PRESETS == {"preview": (1, 1), "low": (4, 4), "medium": (16, 8), "high": (64, 16), "ultra": (256, 32)}
    assert RenderQuality.NAMES["ultra"] == "Ultra (256 spp)"
    with pytest.raises(ValueError):
        RenderQuality.get("best")
    r = subprocess.run([EXE, "-i", "-w", "abc"], capture_output=True, text=True)
    assert r.returncode == 2 and "invalid value" in r.stderr
    r = subprocess.run([EXE, "-w", "10"], capture_output=True, text=True)      # no -i: the window is out of scope
    assert r.returncode == 2 and "image-mode" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("scene,args,okw", [
    ("teapot", ["-w", "64", "-h", "48", "-s", "3", "-l", "4", "-b", "5"], dict(width=64, height=48, spp=3, light_samples=4, bounces=5)),
    ("specular", ["-w", "40", "-h", "40"], dict(width=40, height=40, spp=4, light_samples=4, bounces=50)),     # defaults of src/main.rs:13-18
    ("diffuse", ["-w", "36", "-h", "28", "-b", "3", "--preset", "medium"], dict(width=36, height=28, spp=16, light_samples=8, bounces=3)),   # gui.rs:56-74
    ("nonsense", ["-w", "32", "-h", "32", "-s", "2", "-b", "2", "--seed", "7"], dict(width=32, height=32, spp=2, light_samples=4, bounces=2, seed=7)),
])
def test_image_mode_matches_oracle(gpu, teapot_obj, tmp_path, scene, args, okw):
    from oracle import binding as ob
    env = dict(os.environ, SCENE=scene)
    r = subprocess.run([EXE, "-i", "--obj-dir", os.path.dirname(teapot_obj)] + args, cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().split("\n")
    assert re.fullmatch(r"Scene '%s' loaded in \d+\.\d{3}s" % scene, lines[0])          # main.rs:125-129
    assert lines[1] == "Rendering image..." and re.fullmatch(r"Render time: \d+\.\d{3}s", lines[2])   # canvas.rs:98-102
    m = re.fullmatch(r"Saving with filename (\./dump/\d+\.png)", lines[3])                # canvas.rs:47-48
    assert m and lines[4] == "Image saved successfully!"
    if scene == "nonsense":
        assert "Unknown scene 'nonsense', using specular" in r.stderr                   # main.rs:120-123
    img = read_png_rgb8(str(tmp_path / m.group(1)))
    osc = ob.OracleScene.named("specular" if scene == "nonsense" else scene, teapot_obj)
    ref = osc.render(ob.make_params(**okw))["rgba"]
    assert np.array_equal(img, ref[:, :, :3])


@pytest.mark.gpu
def test_image_mode_as_one_rank_of_a_multi_gpu_job(gpu, teapot_obj, tmp_path):
    """One process per GPU (WORLD_SIZE / RANK / RT_COMM_ID_FILE): the binary makes the RCCL communicator from the id file, renders its
    shard, gathers over RCCL (rt_render_comm) and rank 0 writes the PNG.  Run here as a job of one rank: same image as without."""
    from oracle import binding as ob
    env = dict(os.environ, SCENE="teapot", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", RT_COMM_ID_FILE=str(tmp_path / "rccl.id"), RT_JOB_ID="job-a")
    (tmp_path / "rccl.id").write_bytes(b"\0" * 136)           # a file left by an earlier job: rank 0 replaces it, and removes its own once the communicator is up
    r = subprocess.run([EXE, "-i", "--obj-dir", os.path.dirname(teapot_obj), "-w", "70", "-h", "50", "-s", "2", "-l", "2", "-b", "3"], cwd=str(tmp_path), env=env,
                       capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stderr
    m = re.search(r"Saving with filename (\./dump/\d+\.png)", r.stdout)
    assert m and not os.path.exists(tmp_path / "rccl.id")
    img = read_png_rgb8(str(tmp_path / m.group(1)))
    ref = ob.OracleScene.named("teapot", teapot_obj).render(ob.make_params(width=70, height=50, spp=2, light_samples=2, bounces=3))["rgba"]
    assert np.array_equal(img, ref[:, :, :3])


@pytest.mark.gpu
def test_debug_flag_prints_centre_pixel(gpu, tmp_path):
    r = subprocess.run([EXE, "-i", "-d", "-w", "33", "-h", "33", "-s", "1", "-l", "1", "-b", "1"], cwd=str(tmp_path), env=dict(os.environ, SCENE="specular"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    # centre ray of a Cornell scene hits the back wall at t = 68: 0.7 * (1 - 68/100)
    v = np.float32(0.7) * (np.float32(1.0) - np.float32(68.0) / np.float32(100.0))
    assert "Spectrum { r: %g, g: %g, b: %g }" % (v, v, v) in r.stdout


@pytest.mark.gpu
def test_cxx_control_surface(gpu, teapot_obj):
    """Raytracer::{render, toggle_rendering_mode, move_camera, reset_camera, update_render_settings, interrupt_render, set_scene}
    through the C++ mirror (rustracer_amd/host/selftest.cpp)."""
    r = subprocess.run([os.path.join(ROOT, "rustracer_amd", "host_selftest"), os.path.dirname(teapot_obj)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "host selftest ok" in r.stdout, r.stderr + r.stdout
