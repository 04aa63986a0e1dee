// main.cpp — the reference binary's entry point (src/main.rs:108-132) over the C++ host mirror:
// same flags, defaults, SCENE environment variable and stdout lines.  Only image mode (-i) renders;
// the window is out of scope.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <string>

#include "rustracer.hpp"

using namespace rustracer;

static int usage(FILE* f) {
    fprintf(f,
            "rustracer\n\nUSAGE:\n    rustracer [FLAGS] [OPTIONS]\n\nFLAGS:\n"
            "    -d                     Debug mode, where only intersections are shown\n"
            "        --high-dpi         Basically for handling that OSX does display scaling weird\n"
            "    -i, --image-mode       Mode that doesn't display anything to screen, but rather just dumps it to a png\n"
            "        --single-threaded  Mode that runs without parallelization\n"
            "        --hemisphere       Hemisphere direct-lighting estimator (README.md:9)\n"
            "        --help             Prints help information\n\nOPTIONS:\n"
            "    -b <b>        Sets the max amount of bounces to simulate\n"
            "    -h <h>        Screen height\n"
            "    -l <l>        Sets how many light samples to do for one-bounce radiance\n"
            "    -s <s>        Sets how many samples per pixel to do\n"
            "    -w <w>        Screen width\n"
            "        --seed <n>     RNG seed (the reference is time-seeded)\n"
            "        --obj-dir <d>  Directory holding teapot.obj / dragon.obj (default: obj)\n\n"
            "ENVIRONMENT:\n    SCENE   dragon | teapot | specular (default) | diffuse | triangle | sphere_on_plane | specular_teapot\n");
    return f == stdout ? 0 : 2;
}

int main(int argc, char** argv) {
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--help")) return usage(stdout);
        if (!strcmp(argv[i], "--png-selftest") && i + 1 < argc) {
            // host-only check of the PNG writer (no GPU): a 37x21 gradient
            const uint32_t w = 37, h = 21;
            std::vector<uint8_t> rgb((size_t)w * h * 3);
            for (uint32_t y = 0; y < h; ++y)
                for (uint32_t x = 0; x < w; ++x) {
                    rgb[((size_t)y * w + x) * 3] = (uint8_t)(x * 7);
                    rgb[((size_t)y * w + x) * 3 + 1] = (uint8_t)(y * 12);
                    rgb[((size_t)y * w + x) * 3 + 2] = (uint8_t)(x ^ y);
                }
            const std::vector<uint8_t> png = encode_png_rgb8(rgb, w, h);
            FILE* f = fopen(argv[i + 1], "wb");
            if (!f || fwrite(png.data(), 1, png.size(), f) != png.size()) return 1;
            fclose(f);
            return 0;
        }
    }
    try {
        // parse args
        const Config config = Config::from_args(argc, argv);
        if (!config.image_mode) {
            fprintf(stderr, "the interactive window is out of scope for this build; pass -i / --image-mode\n");
            return 2;
        }
        int device = 0;
        if (const char* lr = getenv("LOCAL_RANK")) device = atoi(lr);
        if (int rc = rt_init(device)) {
            fprintf(stderr, "rt_init: %s\n", rt_last_error());
            return rc;
        }
        const char* env = getenv("SCENE");
        const std::string scene_name = env ? env : "specular";
        const auto load_start = std::chrono::steady_clock::now();
        Scene scene = Scene::by_name(scene_name, config.obj_dir);
        printf("Scene '%s' loaded in %.3fs\n", scene_name.c_str(), std::chrono::duration<double>(std::chrono::steady_clock::now() - load_start).count());
        Raytracer raytracer = Raytracer::new_(config, std::move(scene));
        if (config.debug) {
            // `-d` is parsed but unused by the reference (Config.debug, src/main.rs:28); here it prints the centre pixel's debug colour
            raytracer.inner->test(config.screen_width / 2, config.screen_height / 2);
        }
        raytracer.start();
        return 0;
    } catch (const Error& e) {
        fprintf(stderr, "%s\n", e.what());
        return e.code ? e.code : 1;
    } catch (const std::exception& e) {
        fprintf(stderr, "error: %s\n\n", e.what());
        usage(stderr);
        return 2;
    }
}
