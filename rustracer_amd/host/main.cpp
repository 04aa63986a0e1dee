// main.cpp — the reference binary's entry point (src/main.rs:108-132) over the C++ host mirror:
// same flags, defaults, SCENE environment variable and stdout lines.  Only image mode (-i) renders;
// the window is out of scope.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>

#include <chrono>
#include <string>
#include <thread>

#include "rustracer.hpp"

using namespace rustracer;

static void check(int rc) {
    if (rc != RT_OK) throw Error(rc, rt_last_error());
}

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
            "        --obj-dir <d>  Directory holding teapot.obj / dragon.obj (default: obj)\n"
            "        --preset <q>   preview | low | medium | high | ultra: the (spp, light samples) pairs of the reference's quality panel\n\n"
            "ENVIRONMENT:\n    SCENE   dragon | teapot | specular (default) | diffuse | triangle | sphere_on_plane | specular_teapot\n"
            "    WORLD_SIZE, RANK, LOCAL_RANK, RT_COMM_ID_FILE   one process per GPU: rank r renders the screen tiles t %% world == r, the tiles are\n"
            "            gathered over RCCL and rank 0 writes the PNG; RT_COMM_ID_FILE is a path all ranks share (rank 0 writes the RCCL id there)\n");
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
        // one process per GPU (no reference counterpart)
        const int world = getenv("WORLD_SIZE") ? atoi(getenv("WORLD_SIZE")) : 1, rank = getenv("RANK") ? atoi(getenv("RANK")) : 0;
        rt_comm* comm = nullptr;
        if (world > 1 || getenv("RT_COMM_ID_FILE")) {
            const char* path = getenv("RT_COMM_ID_FILE");
            if (!path) throw Error(RT_ERR_INVALID, "WORLD_SIZE > 1 needs RT_COMM_ID_FILE (a path every rank can read)");
            // The file holds an 8-byte job tag and the 128-byte id.  The tag comes from the launcher (RT_JOB_ID, else MASTER_PORT: the same on
            // every rank of a job, different between jobs), so a file left by another job is not taken for this one's; without either a file
            // older than a minute before this process started is refused.  Rank 0 removes any old file first and the new one once every rank
            // holds the communicator (rt_comm_init_rank is collective: all ranks have read the id by then).
            uint8_t id[128];
            uint64_t tag = 0;
            const char* job = getenv("RT_JOB_ID") ? getenv("RT_JOB_ID") : getenv("MASTER_PORT");
            for (const char* q = job; q && *q; ++q) tag = (tag ^ (uint8_t)*q) * 0x100000001B3ull;
            if (job) tag |= 1ull << 63;
            const time_t started = time(nullptr);
            if (rank == 0) {
                (void)remove(path);
                check(rt_comm_unique_id(id));
                const std::string tmp = std::string(path) + ".tmp";
                FILE* f = fopen(tmp.c_str(), "wb");
                if (!f || fwrite(&tag, 1, 8, f) != 8 || fwrite(id, 1, 128, f) != 128) throw Error(RT_ERR_INVALID, "cannot write RT_COMM_ID_FILE");
                fclose(f);
                if (rename(tmp.c_str(), path)) throw Error(RT_ERR_INVALID, "cannot publish RT_COMM_ID_FILE");
            } else {
                bool got = false;
                for (int tries = 0; tries < 6000 && !got; ++tries) {   // up to a minute
                    if (FILE* f = fopen(path, "rb")) {
                        uint64_t seen = 0;
                        struct stat sb;
                        const bool fresh = job || (fstat(fileno(f), &sb) == 0 && sb.st_mtime + 60 >= started);
                        got = fread(&seen, 1, 8, f) == 8 && fread(id, 1, 128, f) == 128 && seen == tag && fresh;
                        fclose(f);
                    }
                    if (!got) std::this_thread::sleep_for(std::chrono::milliseconds(10));
                }
                if (!got) throw Error(RT_ERR_INVALID, "RT_COMM_ID_FILE did not appear (or holds another job's id)");
            }
            check(rt_comm_init_rank(id, world, rank, &comm));
            if (rank == 0) (void)remove(path);
            raytracer.set_shard((uint32_t)rank, (uint32_t)world, comm);
            raytracer.inner->canvas.save = rank == 0;
        }
        if (config.debug) {
            // `-d` is parsed but unused by the reference (Config.debug, src/main.rs:28); here it prints the centre pixel's debug colour
            raytracer.inner->test(config.screen_width / 2, config.screen_height / 2);
        }
        raytracer.start();
        if (comm) rt_comm_destroy(comm);
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
