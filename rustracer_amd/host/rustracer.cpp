// rustracer.cpp — implementation of the C++ host mirror (rustracer.hpp) over librustracer_hip.so.
#include "rustracer.hpp"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>

#include <chrono>

namespace rustracer {

namespace {

const float CAMERA_SPEED = 2.0f;   // src/raytracer.rs:18

void check(int rc) {
    if (rc != RT_OK) throw Error(rc, std::string("rustracer_hip error ") + std::to_string(rc) + ": " + rt_last_error());
}

bool file_exists(const std::string& p) {
    struct stat st;
    return stat(p.c_str(), &st) == 0;
}

uint32_t parse_u32(const char* flag, const char* text) {
    char* end = nullptr;
    const unsigned long long v = strtoull(text, &end, 10);
    if (end == text || *end != '\0' || v > 0xFFFFFFFFull) throw std::runtime_error(std::string("invalid value '") + text + "' for '" + flag + "'");
    return (uint32_t)v;
}

}  // namespace

// ---- Config (src/main.rs:20-105) ----
Config::Config() {
    // f32::to_radians(DEFAULT_FOV_DEGREES), src/main.rs:95: degrees * (PI / 180) in f32
    const float pi = 3.14159265358979323846f;
    fov = DEFAULT_FOV_DEGREES * (pi / 180.0f);
}

const char* render_quality_name(RenderQuality q) {
    switch (q) {
        case RenderQuality::Preview: return "Preview (1 spp)";
        case RenderQuality::Low: return "Low (4 spp)";
        case RenderQuality::Medium: return "Medium (16 spp)";
        case RenderQuality::High: return "High (64 spp)";
        default: return "Ultra (256 spp)";
    }
}
uint32_t render_quality_samples_per_pixel(RenderQuality q) {
    switch (q) {
        case RenderQuality::Preview: return 1;
        case RenderQuality::Low: return 4;
        case RenderQuality::Medium: return 16;
        case RenderQuality::High: return 64;
        default: return 256;
    }
}
uint32_t render_quality_light_samples(RenderQuality q) {
    switch (q) {
        case RenderQuality::Preview: return 1;
        case RenderQuality::Low: return 4;
        case RenderQuality::Medium: return 8;
        case RenderQuality::High: return 16;
        default: return 32;
    }
}
bool render_quality_from_string(const std::string& s, RenderQuality* out) {
    const struct { const char* n; RenderQuality q; } t[] = {{"preview", RenderQuality::Preview}, {"low", RenderQuality::Low}, {"medium", RenderQuality::Medium},
                                                            {"high", RenderQuality::High}, {"ultra", RenderQuality::Ultra}};
    for (const auto& e : t)
        if (s == e.n) { *out = e.q; return true; }
    return false;
}

Config Config::from_args(int argc, char** argv) {
    Config c;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto value = [&](const char* flag) -> const char* {
            if (i + 1 >= argc) throw std::runtime_error(std::string("the argument '") + flag + " <n>' requires a value");
            return argv[++i];
        };
        if (a == "-s") c.samples_per_pixel = parse_u32("-s", value("-s"));
        else if (a == "-l") c.light_samples = parse_u32("-l", value("-l"));
        else if (a == "-b") c.bounces = parse_u32("-b", value("-b"));
        else if (a == "-w") c.screen_width = parse_u32("-w", value("-w"));
        else if (a == "-h") c.screen_height = parse_u32("-h", value("-h"));
        else if (a == "-d") c.debug = true;
        else if (a == "--high-dpi") c.high_dpi = true;
        else if (a == "-i" || a == "--image-mode") c.image_mode = true;
        else if (a == "--single-threaded") c.single_threaded = true;   // accepted; the GPU path has no such mode
        else if (a == "--seed") c.seed = strtoull(value("--seed"), nullptr, 0);
        else if (a == "--hemisphere") c.direct_estimator = RT_DIRECT_HEMISPHERE;
        else if (a == "--obj-dir") c.obj_dir = value("--obj-dir");
        else if (a == "--preset") {
            RenderQuality q;
            const std::string v = value("--preset");
            if (!render_quality_from_string(v, &q)) throw std::runtime_error("--preset takes preview | low | medium | high | ultra, not '" + v + "'");
            c.samples_per_pixel = render_quality_samples_per_pixel(q);     // what GuiAction::ChangeQuality does, src/gui.rs / canvas.rs:392-441
            c.light_samples = render_quality_light_samples(q);
        }
        else throw std::runtime_error("found argument '" + a + "' which wasn't expected");
    }
    return c;
}

// ---- Scene ----
Scene Scene::named(const char* name, const char* obj_path) {
    rt_scene* h = nullptr;
    check(rt_scene_new_named(name, obj_path, &h));
    return Scene(h);
}
Scene Scene::new_dragon(const std::string& obj_path) {
    if (file_exists(obj_path)) return named("dragon", obj_path.c_str());
    // obj/dragon.obj is not in the reference mount (.MISSING_LARGE_BLOBS:1): the deterministic dragon-class stand-in takes its
    // place, with the same scale and offset (src/scene/mod.rs:200-209)
    const uint32_t nu = 2178, nv = 200;   // 871,200 triangles
    std::vector<float> vp((size_t)nu * nv * 3), vn((size_t)nu * nv * 3);
    std::vector<uint32_t> f((size_t)2 * nu * nv * 3);
    check(rt_make_dragon_standin(nu, nv, vp.data(), vn.data(), f.data()));
    rt_scene* h = nullptr;
    check(rt_scene_new_dragon_mesh(vp.data(), vn.data(), nu * nv, f.data(), 2 * nu * nv, &h));
    return Scene(h);
}
Scene Scene::new_teapot(const std::string& obj_path) { return named("teapot", obj_path.c_str()); }
Scene Scene::new_specular() { return named("specular", nullptr); }
Scene Scene::new_diffuse() { return named("diffuse", nullptr); }
Scene Scene::new_triangle() { return named("triangle", nullptr); }
Scene Scene::new_sphere_on_plane() { return named("sphere_on_plane", nullptr); }
Scene Scene::new_specular_teapot(const std::string& obj_path) { return named("specular_teapot", obj_path.c_str()); }

Scene Scene::by_name(const std::string& name, const std::string& obj_dir) {
    if (name == "dragon") return new_dragon(obj_dir + "/dragon.obj");
    if (name == "teapot") return new_teapot(obj_dir + "/teapot.obj");
    if (name == "specular") return new_specular();
    if (name == "diffuse") return new_diffuse();
    if (name == "triangle") return new_triangle();
    if (name == "sphere_on_plane") return new_sphere_on_plane();
    if (name == "specular_teapot") return new_specular_teapot(obj_dir + "/teapot.obj");
    fprintf(stderr, "Unknown scene '%s', using specular\n", name.c_str());   // src/main.rs:120-123
    return new_specular();
}
Scene& Scene::operator=(Scene&& o) noexcept {
    if (this != &o) {
        if (h_) rt_scene_destroy(h_);
        h_ = o.h_;
        o.h_ = nullptr;
    }
    return *this;
}
Scene::~Scene() {
    if (h_) rt_scene_destroy(h_);
}
rt_scene_info Scene::info() const {
    rt_scene_info i;
    check(rt_scene_info_get(h_, &i));
    return i;
}
std::vector<uint32_t> Scene::light_indexes() const {
    std::vector<uint32_t> out(info().n_lights);
    check(rt_scene_lights(h_, out.data(), (uint32_t)out.size()));
    return out;
}

// ---- SharedPixelBuffer (src/raytracer.rs:22-67) ----
SharedPixelBuffer::SharedPixelBuffer(uint32_t w, uint32_t h) : width(w), height(h), data_((size_t)w * h * 4) {
    static_assert(sizeof(std::atomic<uint8_t>) == 1, "one byte per pixel component");
    for (size_t i = 0; i < data_.size(); ++i) data_[i].store(0, std::memory_order_relaxed);
}
void SharedPixelBuffer::set_pixel(uint32_t x, uint32_t y, uint8_t r, uint8_t g, uint8_t b) {
    const size_t index = ((size_t)y * width + x) * 4;
    if (index + 3 < data_.size()) {
        data_[index].store(r, std::memory_order_relaxed);
        data_[index + 1].store(g, std::memory_order_relaxed);
        data_[index + 2].store(b, std::memory_order_relaxed);
        data_[index + 3].store(255, std::memory_order_relaxed);
    }
}
void SharedPixelBuffer::snapshot(std::vector<uint8_t>& dst) const {
    if (dst.size() != data_.size()) throw std::runtime_error("snapshot: destination length differs from the pixel buffer");   // debug_assert_eq!, raytracer.rs:56
    for (size_t i = 0; i < data_.size(); ++i) dst[i] = data_[i].load(std::memory_order_relaxed);
}
void SharedPixelBuffer::clear() {
    for (size_t i = 0; i < data_.size(); ++i) data_[i].store((i & 3) == 3 ? 255 : 0, std::memory_order_relaxed);
}

RenderConfig::RenderConfig(const Config& c)
    : screen_width(c.screen_width), screen_height(c.screen_height), fov(c.fov), samples_per_pixel(c.samples_per_pixel),
      light_samples(c.light_samples), bounces(c.bounces), single_threaded(c.single_threaded), seed(c.seed), direct_estimator(c.direct_estimator) {}

// ---- RaytracerInner ----
RaytracerInner::RaytracerInner(const Config& c, Scene s)
    : canvas(c.screen_width, c.screen_height, c.high_dpi, c.image_mode),
      pixel_buffer(std::make_shared<SharedPixelBuffer>(c.screen_width, c.screen_height)), config(c), scene(std::move(s)) {}

void RaytracerInner::render() {
    RenderingMode mode;
    {
        std::lock_guard<std::mutex> g(mode_mutex);
        mode = rendering_mode;
    }
    render_mode(mode);
}
void RaytracerInner::debug_render() { render_mode(RenderingMode::Debug); }

void RaytracerInner::render_mode(RenderingMode mode) {
    // do_render / debug_render, src/raytracer.rs:167-221, 353-391
    interrupt.store(0);
    is_rendering.store(true);
    if (mode == RenderingMode::Full) render_progress.store(0);
    Point cam;
    {
        std::lock_guard<std::mutex> g(camera_mutex);
        cam = camera_position;
    }
    rt_render_params p;
    memset(&p, 0, sizeof p);
    {
        std::lock_guard<std::mutex> g(config_mutex);
        rt_screen_params(config.screen_width, config.screen_height, config.fov, &p);
        p.spp = config.samples_per_pixel;
        p.light_samples = config.light_samples;
        p.bounces = config.bounces;
        p.seed = config.seed;
        p.direct_estimator = config.direct_estimator;
    }
    p.cam[0] = cam.x; p.cam[1] = cam.y; p.cam[2] = cam.z;
    p.mode = mode == RenderingMode::Full ? RT_MODE_FULL : RT_MODE_DEBUG;
    p.tile_w = 32; p.tile_h = 32; p.shard_index = shard_index; p.shard_count = shard_count;   // one process per GPU: this rank's screen tiles
    int rc;
    {
        std::shared_lock<std::shared_mutex> g(scene_lock);   // read-held for the whole render, raytracer.rs:176
        uint32_t progress_word = 0;
        // the library writes progress through a plain word; mirror it into the atomic the canvas polls
        rc = rt_render_comm(scene.handle(), &p, comm, pixel_buffer->raw(), mode == RenderingMode::Full ? reinterpret_cast<volatile uint32_t*>(&render_progress) : &progress_word,
                       reinterpret_cast<const volatile uint32_t*>(&interrupt));
    }
    is_rendering.store(false);
    last_render_interrupted = rc == RT_ERR_INTERRUPTED;
    if (rc != RT_OK && rc != RT_ERR_INTERRUPTED) check(rc);
    if (mode == RenderingMode::Full) render_progress.store(100);   // raytracer.rs:220 (also after an interrupt)
}

void RaytracerInner::test(uint32_t i, uint32_t j) {
    // src/raytracer.rs:419-428: debug_render_helper for one pixel, printed in Spectrum's Debug form
    Point cam;
    {
        std::lock_guard<std::mutex> g(camera_mutex);
        cam = camera_position;
    }
    rt_render_params p;
    memset(&p, 0, sizeof p);
    {
        std::lock_guard<std::mutex> g(config_mutex);
        rt_screen_params(config.screen_width, config.screen_height, config.fov, &p);
    }
    // ScreenParams::screen_to_world, raytracer.rs:127-133 (single f32 operations; this file is built with -ffp-contract=off)
    const float iw = ((float)i + 0.5f) * p.inv_w, jh = ((float)j + 0.5f) * p.inv_h;
    const float xi = (p.start + iw * p.total) * p.aspect_ratio, yi = -p.start - jh * p.total, zi = -p.z;
    const float n = sqrtf((xi * xi + yi * yi) + zi * zi);
    const float o[3] = {cam.x, cam.y, cam.z}, d[3] = {xi / n, yi / n, zi / n};
    uint32_t id = 0xFFFFFFFFu;
    float t = 0.0f;
    {
        std::shared_lock<std::shared_mutex> g(scene_lock);
        check(rt_intersect_rays(scene.handle(), o, d, 1, &id, &t));
    }
    float v = 0.0f;
    if (id != 0xFFFFFFFFu) {
        const float max_distance = 100.0f;
        const float dm = (t != t) ? max_distance : (t < max_distance ? t : max_distance);   // f32::min
        v = 0.7f * (1.0f - (dm / max_distance));
    }
    printf("Spectrum { r: %g, g: %g, b: %g }\n", v, v, v);
}

// ---- Raytracer (src/raytracer.rs:431-532) ----
Raytracer Raytracer::new_(const Config& config, Scene scene) {
    Raytracer r;
    r.inner = std::make_shared<RaytracerInner>(config, std::move(scene));
    return r;
}
void Raytracer::set_shard(uint32_t rank, uint32_t world, rt_comm* comm) {
    // no reference counterpart: this process renders the tiles t % world == rank and every frame ends with one RCCL gather
    if (world == 0 || rank >= world || (world > 1 && !comm)) throw Error(RT_ERR_INVALID, "bad shard description");
    inner->shard_index = rank;
    inner->shard_count = world;
    inner->comm = comm;
}
void Raytracer::start() {
    if (!inner->canvas.image_mode) render(false);   // raytracer.rs:469-471
    inner->canvas.start(inner);
}
void Raytracer::move_camera(Vector direction) {
    std::lock_guard<std::mutex> g(inner->camera_mutex);
    Point& c = inner->camera_position;
    c.x = c.x + direction.x * CAMERA_SPEED;
    c.y = c.y + direction.y * CAMERA_SPEED;
    c.z = c.z + direction.z * CAMERA_SPEED;
}
void Raytracer::reset_camera() {
    std::lock_guard<std::mutex> g(inner->camera_mutex);
    inner->camera_position = Point{0.0f, 0.0f, 0.0f};
}
void Raytracer::toggle_rendering_mode() {
    interrupt_render();
    std::lock_guard<std::mutex> g(inner->mode_mutex);
    inner->rendering_mode = inner->rendering_mode == RenderingMode::Debug ? RenderingMode::Full : RenderingMode::Debug;
}
void Raytracer::interrupt_render() { inner->interrupt.store(1); }
void Raytracer::update_render_settings(uint32_t samples_per_pixel, uint32_t light_samples, uint32_t max_bounces) {
    std::lock_guard<std::mutex> g(inner->config_mutex);
    inner->config.samples_per_pixel = samples_per_pixel;
    inner->config.light_samples = light_samples;
    inner->config.bounces = max_bounces;
}
void Raytracer::set_scene(Scene scene) {
    interrupt_render();
    std::unique_lock<std::shared_mutex> g(inner->scene_lock);   // blocks until the in-flight render has seen the interrupt and returned
    inner->scene = std::move(scene);
}
void Raytracer::render(bool wait_for_completion) {
    std::shared_ptr<RaytracerInner> local_self = inner;
    if (wait_for_completion) {
        local_self->render();
    } else {
        threads_.emplace_back([local_self]() {
            try {
                local_self->render();
            } catch (const std::exception& e) {
                fprintf(stderr, "render thread: %s\n", e.what());
            }
        });
    }
}
void Raytracer::wait() {
    for (std::thread& t : threads_)
        if (t.joinable()) t.join();
    threads_.clear();
}
Raytracer::~Raytracer() {
    if (inner) wait();
}

// ---- Canvas (src/canvas.rs:41-110) ----
void Canvas::start(const std::shared_ptr<RaytracerInner>& raytracer_inner) {
    if (image_mode) start_image_mode(raytracer_inner);
    else throw Error(RT_ERR_UNSUPPORTED, "the interactive SDL2 window (src/canvas.rs:113-449) is out of scope for this build; run with -i / --image-mode");
}
void Canvas::start_image_mode(const std::shared_ptr<RaytracerInner>& raytracer_inner) {
    {
        std::lock_guard<std::mutex> g(raytracer_inner->mode_mutex);
        raytracer_inner->rendering_mode = RenderingMode::Full;
    }
    printf("Rendering image...\n");
    const auto start = std::chrono::steady_clock::now();
    raytracer_inner->render();
    const double elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
    printf("Render time: %.3fs\n", elapsed);
    std::vector<uint8_t> rgba((size_t)width_ * height_ * 4);
    raytracer_inner->pixel_buffer->snapshot(rgba);
    if (save) save_canvas(rgba_to_rgb(rgba, width_, height_));
}
std::vector<uint8_t> Canvas::rgba_to_rgb(const std::vector<uint8_t>& rgba, uint32_t width, uint32_t height) {
    const size_t pixel_count = (size_t)width * height;
    std::vector<uint8_t> rgb;
    rgb.reserve(pixel_count * 3);
    for (size_t i = 0; i < pixel_count; ++i) {
        rgb.push_back(rgba[i * 4]);
        rgb.push_back(rgba[i * 4 + 1]);
        rgb.push_back(rgba[i * 4 + 2]);
    }
    return rgb;
}
void Canvas::save_canvas(const std::vector<uint8_t>& pixel_buffer_rgb) {
    mkdir("./dump", 0777);   // create_dir_all(...).ok()
    const std::string path_string = "./dump/" + std::to_string((long long)time(nullptr)) + ".png";
    printf("Saving with filename %s\n", path_string.c_str());
    FILE* f = fopen(path_string.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot create " + path_string);   // File::create(path).unwrap()
    const std::vector<uint8_t> png = encode_png_rgb8(pixel_buffer_rgb, width_, height_);
    const bool ok = fwrite(png.data(), 1, png.size(), f) == png.size();
    fclose(f);
    if (!ok) throw std::runtime_error("Failed to write canvas to png");
    printf("Image saved successfully!\n");
    last_path = path_string;
}

// ---- PNG ----
namespace {
uint32_t crc32_update(uint32_t crc, const uint8_t* p, size_t n) {
    static uint32_t table[256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        ready = true;
    }
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xFF] ^ (crc >> 8);
    return crc;
}
void put_u32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}
void put_chunk(std::vector<uint8_t>& out, const char tag[4], const std::vector<uint8_t>& data) {
    put_u32(out, (uint32_t)data.size());
    const size_t at = out.size();
    out.insert(out.end(), tag, tag + 4);
    out.insert(out.end(), data.begin(), data.end());
    put_u32(out, crc32_update(0xFFFFFFFFu, out.data() + at, out.size() - at) ^ 0xFFFFFFFFu);
}
}  // namespace

std::vector<uint8_t> encode_png_rgb8(const std::vector<uint8_t>& rgb, uint32_t width, uint32_t height) {
    if (rgb.size() != (size_t)width * height * 3) throw std::runtime_error("encode_png_rgb8: wrong buffer length");
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n'};
    std::vector<uint8_t> ihdr;
    put_u32(ihdr, width); put_u32(ihdr, height);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    put_chunk(out, "IHDR", ihdr);
    // scanlines with filter type 0, wrapped in a zlib stream of stored deflate blocks
    std::vector<uint8_t> raw;
    raw.reserve((size_t)height * (1 + (size_t)width * 3));
    for (uint32_t y = 0; y < height; ++y) {
        raw.push_back(0);
        raw.insert(raw.end(), rgb.begin() + (size_t)y * width * 3, rgb.begin() + (size_t)(y + 1) * width * 3);
    }
    std::vector<uint8_t> z = {0x78, 0x01};
    uint32_t a = 1, b = 0;   // Adler-32
    for (size_t i = 0; i < raw.size(); ++i) { a = (a + raw[i]) % 65521u; b = (b + a) % 65521u; }
    size_t pos = 0;
    do {
        const size_t n = raw.size() - pos < 65535 ? raw.size() - pos : 65535;
        z.push_back(pos + n == raw.size() ? 1 : 0);
        z.push_back((uint8_t)(n & 0xFF)); z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)(~n & 0xFF)); z.push_back((uint8_t)((~n >> 8) & 0xFF));
        z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
        pos += n;
    } while (pos < raw.size());
    put_u32(z, (b << 16) | a);
    put_chunk(out, "IDAT", z);
    put_chunk(out, "IEND", std::vector<uint8_t>());
    return out;
}

}  // namespace rustracer
