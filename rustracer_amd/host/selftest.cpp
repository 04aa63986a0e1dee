// selftest.cpp — exercises the control surface of the C++ host mirror on a GPU (src/raytracer.rs:431-532,
// src/canvas.rs:241-289 call sites): mode toggle, camera moves, settings update, background render with
// interrupt, scene swap under the scene lock.  Prints "host selftest ok" and exits 0, or the failed check.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <thread>

#include "rustracer.hpp"

using namespace rustracer;

#define CHECK(cond)                                                     \
    do {                                                                \
        if (!(cond)) {                                                  \
            fprintf(stderr, "selftest failed: %s (line %d)\n", #cond, __LINE__); \
            return 1;                                                   \
        }                                                               \
    } while (0)

int main(int argc, char** argv) {
    const std::string obj_dir = argc > 1 ? argv[1] : "obj";
    if (rt_init(0)) { fprintf(stderr, "rt_init: %s\n", rt_last_error()); return 1; }
    try {
        Config c;
        c.screen_width = 96; c.screen_height = 64; c.samples_per_pixel = 2; c.light_samples = 2; c.bounces = 3; c.image_mode = true;
        Raytracer rt = Raytracer::new_(c, Scene::new_specular());
        CHECK(rt.inner->rendering_mode == RenderingMode::Debug);             // raytracer.rs:445
        CHECK(rt.inner->scene.light_indexes() == std::vector<uint32_t>({10}));   // mod.rs:91-97 on new_specular
        std::vector<uint8_t> snap(rt.inner->pixel_buffer->len());

        // debug render: greys, alpha 255; progress untouched (debug_render never stores it, raytracer.rs:353-391)
        rt.render(true);
        rt.inner->pixel_buffer->snapshot(snap);
        CHECK(snap[3] == 255 && snap[0] == snap[1] && snap[1] == snap[2]);
        CHECK(rt.inner->render_progress.load() == 0 && !rt.inner->is_rendering.load());
        const std::vector<uint8_t> debug_frame = snap;

        // full render
        rt.toggle_rendering_mode();
        CHECK(rt.inner->rendering_mode == RenderingMode::Full);
        rt.render(true);
        CHECK(rt.inner->render_progress.load() == 100 && !rt.inner->last_render_interrupted);
        rt.inner->pixel_buffer->snapshot(snap);
        const std::vector<uint8_t> full_frame = snap;
        CHECK(full_frame != debug_frame);

        // camera: x CAMERA_SPEED (raytracer.rs:18, 479); a moved camera changes the image, reset restores it
        rt.move_camera(Vector{0.0f, 0.0f, -1.0f});
        CHECK(rt.inner->camera_position.z == -2.0f);
        rt.render(true);
        rt.inner->pixel_buffer->snapshot(snap);
        CHECK(snap != full_frame);
        rt.reset_camera();
        rt.render(true);
        rt.inner->pixel_buffer->snapshot(snap);
        CHECK(snap == full_frame);                                           // deterministic streams: same settings, same pixels

        // update_render_settings changes spp / light samples / bounces only (raytracer.rs:501-511)
        rt.update_render_settings(1, 1, 1);
        rt.render(true);
        rt.inner->pixel_buffer->snapshot(snap);
        CHECK(snap != full_frame);

        // background render + interrupt (canvas.rs:241-289: every key action interrupts first)
        rt.update_render_settings(64, 16, 8);
        rt.inner->render_progress.store(55);                                 // sentinel: do_render stores 0 first and 100 last
        rt.render(false);
        for (int k = 0; k < 5000 && !rt.inner->is_rendering.load() && rt.inner->render_progress.load() == 55; ++k)   // bounded: the render may already be over
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        rt.interrupt_render();
        rt.wait();
        CHECK(!rt.inner->is_rendering.load());
        CHECK(rt.inner->render_progress.load() == 100);                      // raytracer.rs:220 runs after an interrupt too

        // set_scene: interrupts, takes the scene write lock, swaps (raytracer.rs:514-518)
        rt.update_render_settings(2, 2, 3);
        rt.render(false);
        rt.set_scene(Scene::new_teapot(obj_dir + "/teapot.obj"));
        rt.wait();
        CHECK(rt.inner->scene.info().n_tri == 1034);                         // 10 walls + 1024 teapot triangles (mod.rs:229-264)
        rt.render(true);
        rt.inner->pixel_buffer->snapshot(snap);
        CHECK(snap != full_frame);
        rt.inner->pixel_buffer->clear();
        rt.inner->pixel_buffer->snapshot(snap);
        CHECK(snap[0] == 0 && snap[3] == 255);                               // clear(): alpha 255, rest 0 (raytracer.rs:61-66)
        rt.inner->pixel_buffer->set_pixel(5, 7, 1, 2, 3);
        rt.inner->pixel_buffer->snapshot(snap);
        CHECK(snap[(7 * 96 + 5) * 4 + 2] == 3);                              // (y*width + x)*4 (raytracer.rs:44)

        // a missing OBJ is an error value here, a panic in the reference (mod.rs:136)
        bool threw = false;
        try { Scene::new_teapot("/nonexistent/teapot.obj"); } catch (const Error& e) { threw = e.code == RT_ERR_INVALID; }
        CHECK(threw);
    } catch (const std::exception& e) {
        fprintf(stderr, "selftest: %s\n", e.what());
        return 1;
    }
    printf("host selftest ok\n");
    return 0;
}
