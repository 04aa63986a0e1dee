// rustracer.hpp — C++ host side above the C ABI (include/rustracer_hip.h): the reference's
// Scene / Raytracer / RaytracerInner / SharedPixelBuffer / Canvas / Config with the same names,
// argument meaning, defaults and state machine (citations are path:line under the reference tree).
// The bodies of do_render / debug_render (src/raytracer.rs:167-221, 353-391) are one call into
// librustracer_hip.so; there is no CPU path here.
//
// The reference is Rust; no Rust toolchain exists in this image, so this C++ mirror is the tested
// host (INTEGRATION.md shows the Rust `-sys` binding a maintainer would add instead).
#pragma once
#include <stdint.h>

#include <atomic>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rustracer_hip.h"

namespace rustracer {

struct Point { float x, y, z; };      // src/scene/geo.rs:11-13 (f32 triples)
struct Vector { float x, y, z; };

// src/main.rs:13-32
const uint32_t DEFAULT_SCREEN_WIDTH = 600, DEFAULT_SCREEN_HEIGHT = 600;
const float DEFAULT_FOV_DEGREES = 90.0f;
const uint32_t DEFAULT_SAMPLES_PER_PIXEL = 4, DEFAULT_LIGHT_SAMPLES = 4, DEFAULT_MAX_BOUNCES = 50;
const uint64_t DEFAULT_SEED = 0x5EED;   // the reference is time-seeded (src/common.rs:137); this build is reproducible

// Render quality presets (src/gui.rs:35-85): the (samples per pixel, light samples) pairs the reference's panel offers; BASELINE's configs
// are quoted on them (16 spp -> 8, 64 -> 16, 256 -> 32).  The panel itself is out of scope; `--preset <name>` sets the pair from the command line.
enum class RenderQuality { Preview, Low, Medium, High, Ultra };
const char* render_quality_name(RenderQuality q);        // "Preview (1 spp)" ... "Ultra (256 spp)", gui.rs:46-54
uint32_t render_quality_samples_per_pixel(RenderQuality q);   // 1, 4, 16, 64, 256, gui.rs:56-64
uint32_t render_quality_light_samples(RenderQuality q);       // 1, 4, 8, 16, 32, gui.rs:66-74
bool render_quality_from_string(const std::string& s, RenderQuality* out);   // "preview" | "low" | "medium" | "high" | "ultra"

struct Config {
    uint32_t screen_width = DEFAULT_SCREEN_WIDTH, screen_height = DEFAULT_SCREEN_HEIGHT;
    float fov;   // radians
    Point origin = {0.0f, 0.0f, 0.0f};
    uint32_t samples_per_pixel = DEFAULT_SAMPLES_PER_PIXEL, light_samples = DEFAULT_LIGHT_SAMPLES, bounces = DEFAULT_MAX_BOUNCES;
    bool debug = false, high_dpi = false, image_mode = false, single_threaded = false;
    // not in the reference:
    uint64_t seed = DEFAULT_SEED;
    uint32_t direct_estimator = RT_DIRECT_IMPORTANCE;
    std::string obj_dir = "obj";
    Config();
    // src/main.rs:35-105 (clap): -s -l -b -w -h <n>, -d, --high-dpi, -i/--image-mode, --single-threaded; plus --seed, --hemisphere, --obj-dir, --preset.
    // Throws std::runtime_error on a malformed command line (clap exits with a usage error).
    static Config from_args(int argc, char** argv);
};

// Raised where the reference panics (unwrap on a missing OBJ file, src/scene/mod.rs:136) or where the library reports an error.
struct Error : public std::runtime_error {
    int code;
    Error(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};

// src/scene/mod.rs:20-24, 108-401.  The object list, the BVH and every ray query live behind the handle.
class Scene {
  public:
    static Scene new_dragon(const std::string& obj_path = "obj/dragon.obj");    // mod.rs:191 (the stand-in mesh when the file is absent)
    static Scene new_teapot(const std::string& obj_path = "obj/teapot.obj");    // mod.rs:229
    static Scene new_specular();                                                // mod.rs:333
    static Scene new_diffuse();                                                 // mod.rs:368
    static Scene new_triangle();                                                // mod.rs:108
    static Scene new_sphere_on_plane();                                         // BASELINE config 1 (not a reference constructor)
    static Scene new_specular_teapot(const std::string& obj_path = "obj/teapot.obj");   // BASELINE config 5 (composed)
    static Scene by_name(const std::string& name, const std::string& obj_dir);  // SCENE dispatch, src/main.rs:112-124
    Scene(Scene&& o) noexcept : h_(o.h_) { o.h_ = nullptr; }
    Scene& operator=(Scene&& o) noexcept;
    Scene(const Scene&) = delete;
    Scene& operator=(const Scene&) = delete;
    ~Scene();
    rt_scene* handle() const { return h_; }
    std::vector<uint32_t> light_indexes() const;                                // mod.rs:495
    rt_scene_info info() const;

  private:
    explicit Scene(rt_scene* h) : h_(h) {}
    static Scene named(const char* name, const char* obj_path);
    rt_scene* h_;
};

// src/raytracer.rs:22-67: RGBA8, row-major, (y*width + x)*4, top row first; relaxed atomics, readers never lock.
class SharedPixelBuffer {
  public:
    SharedPixelBuffer(uint32_t width, uint32_t height);
    void set_pixel(uint32_t x, uint32_t y, uint8_t r, uint8_t g, uint8_t b);
    void snapshot(std::vector<uint8_t>& dst) const;
    void clear();
    uint32_t width, height;
    uint8_t* raw() { return reinterpret_cast<uint8_t*>(data_.data()); }   // what rt_render fills (one byte per atomic)
    size_t len() const { return data_.size(); }

  private:
    std::vector<std::atomic<uint8_t>> data_;
};

enum class RenderingMode { Debug, Full };   // src/raytracer.rs:151-154

struct RenderConfig {                       // src/raytracer.rs:88-96, 136-148
    uint32_t screen_width, screen_height;
    float fov;
    uint32_t samples_per_pixel, light_samples, bounces;
    bool single_threaded;
    uint64_t seed;
    uint32_t direct_estimator;
    explicit RenderConfig(const Config& c);
};

class RaytracerInner;

// src/canvas.rs:24-110 (headless half; the SDL2/OpenGL/egui window, canvas.rs:113-449, is out of scope)
class Canvas {
  public:
    Canvas(uint32_t width, uint32_t height, bool high_dpi, bool image_mode) : image_mode(image_mode), width_(width), height_(height), high_dpi_(high_dpi) {}
    void start(const std::shared_ptr<RaytracerInner>& raytracer_inner);
    bool image_mode;
    std::string last_path;   // the file save_canvas wrote
    bool save = true;        // multi-GPU: every rank holds the gathered frame, rank 0 writes it
    bool high_dpi() const { return high_dpi_; }   // only the window uses it (src/canvas.rs:113-449)

  private:
    void start_image_mode(const std::shared_ptr<RaytracerInner>& raytracer_inner);
    void save_canvas(const std::vector<uint8_t>& pixel_buffer_rgb);
    static std::vector<uint8_t> rgba_to_rgb(const std::vector<uint8_t>& rgba, uint32_t width, uint32_t height);
    uint32_t width_, height_;
    bool high_dpi_;
};

// src/raytracer.rs:69-84, 156-428
class RaytracerInner {
  public:
    RaytracerInner(const Config& config, Scene scene);
    // public fields of the reference (raytracer.rs:69-84)
    Canvas canvas;
    std::mutex camera_mutex;
    Point camera_position = {0.0f, 0.0f, 0.0f};
    std::mutex mode_mutex;
    RenderingMode rendering_mode = RenderingMode::Debug;   // raytracer.rs:445
    std::atomic<bool> is_rendering{false};
    std::atomic<uint32_t> render_progress{0};              // 0..100
    std::shared_ptr<SharedPixelBuffer> pixel_buffer;

    void render();          // raytracer.rs:159-165 -> do_render / debug_render by rendering_mode
    void debug_render();    // raytracer.rs:353
    void test(uint32_t i, uint32_t j);   // raytracer.rs:419-428: prints one pixel's debug colour
    bool last_render_interrupted = false;
    uint32_t shard_index = 0, shard_count = 1;   // multi-GPU: this process's share of the screen tiles (Raytracer::set_shard)
    rt_comm* comm = nullptr;                      // RCCL communicator of the job (not owned)

    // used by Raytracer
    std::mutex config_mutex;
    RenderConfig config;
    std::shared_mutex scene_lock;   // scene RwLock, raytracer.rs:72: read-held for a whole render, write-taken by set_scene
    Scene scene;
    std::atomic<uint32_t> interrupt{0};

  private:
    void render_mode(RenderingMode mode);
};

// src/raytracer.rs:431-532
class Raytracer {
  public:
    static Raytracer new_(const Config& config, Scene scene);
    void set_shard(uint32_t rank, uint32_t world, rt_comm* comm);
    void start();
    void move_camera(Vector direction);           // x CAMERA_SPEED, raytracer.rs:18, 477-481
    void reset_camera();
    void toggle_rendering_mode();
    void interrupt_render();
    void update_render_settings(uint32_t samples_per_pixel, uint32_t light_samples, uint32_t max_bounces);
    void set_scene(Scene scene);
    void render(bool wait_for_completion);        // false: on a new OS thread (raytracer.rs:527)
    void wait();                                  // joins background renders (the reference detaches them)
    ~Raytracer();
    Raytracer(Raytracer&&) = default;
    std::shared_ptr<RaytracerInner> inner;

  private:
    Raytracer() = default;
    std::vector<std::thread> threads_;
};

// RGB8 PNG (colour type 2, bit depth 8), as png::Encoder is configured at src/canvas.rs:53-56; stored deflate blocks.
std::vector<uint8_t> encode_png_rgb8(const std::vector<uint8_t>& rgb, uint32_t width, uint32_t height);

}  // namespace rustracer
