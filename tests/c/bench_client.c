/* tests/c/bench_client.c — bench.py's schedule from plain C, on the HIP runtime libpt_hip.so links by itself.
 *
 * Every number bench.py prints is measured in a Python process in which torch has already mapped ITS bundled HIP runtime (7.0.2 here); a Java or C host of the
 * library gets ROCm's own libamdhip64.so instead.  This client is that host: no Python, no torch, the library linked the ordinary way.  It loads
 * the SSBO contents a Python script dumped (binding_<n>.bin, sky.bin — the reference's Main would hand over its direct buffers), creates ONE context of
 * `streams` wavefront streams on GPU 0 (pt_create_multi), and runs the schedule of bench.py: a step = pt_next_image + one asynchronous batch of `fps` frames
 * (u_frameCount = 1.., seeds (1234 + 7919 f) mod 10000); the image of a step is gathered (pt_gather_image) two steps later; untimed set-up step, `warmup` steps,
 * then `steps` steps + the gathers still owed + pt_synchronize inside the clock.  Prints Msamples/s and an FNV-1a hash of the last image (the test compares it
 * with the same schedule driven through the Python wrappers).
 *
 * usage: bench_client <dir with the .bin files> W H skyW skyH spp_per_frame fps steps warmup streams
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "pt_api.h"

static void* slurp(const char* dir, const char* name, long* bytes) {
    char p[1024];
    snprintf(p, sizeof p, "%s/%s", dir, name);
    FILE* f = fopen(p, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", p); exit(2); }
    fseek(f, 0, SEEK_END); *bytes = ftell(f); fseek(f, 0, SEEK_SET);
    void* b = malloc(*bytes > 0 ? (size_t)*bytes : 1);
    if (fread(b, 1, (size_t)*bytes, f) != (size_t)*bytes) { fprintf(stderr, "short read of %s\n", p); exit(2); }
    fclose(f);
    return b;
}

#define REQUIRE(call, what) do { if ((call) != PT_OK) { fprintf(stderr, "bench client: %s failed (%s)\n", what, pt_last_error()); return 1; } } while (0)

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

int main(int argc, char** argv) {
    if (argc < 11) { fprintf(stderr, "usage: bench_client dir W H skyW skyH spp_per_frame fps steps warmup streams\n"); return 2; }
    const char* dir = argv[1];
    const int W = atoi(argv[2]), H = atoi(argv[3]), skyW = atoi(argv[4]), skyH = atoi(argv[5]), sres = atoi(argv[6]), fps = atoi(argv[7]), steps = atoi(argv[8]), warmup = atoi(argv[9]);
    int streams = atoi(argv[10]);
    if (streams < 1 || streams > 8 || fps < 1 || fps > 4096) return 2;
    setenv("GPU_MAX_HW_QUEUES", "8", 0);          /* the streams of a GPU on distinct hardware queues (read once by the HIP runtime; pt_create_multi sets it too) */
    int devices[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    pt_ctx* ctx = NULL;
    if (streams > 1) REQUIRE(pt_create_multi(&ctx, devices, streams, W, H), "pt_create_multi");
    else REQUIRE(pt_create(&ctx, 0, W, H, 0, 1), "pt_create");
    static const int bindings[] = {0, 1, 2, 3, 4, 5, 7, 10, 11, 12, 13, 14};
    for (size_t k = 0; k < sizeof bindings / sizeof bindings[0]; k++) {
        char name[64]; long bytes;
        snprintf(name, sizeof name, "binding_%d.bin", bindings[k]);
        void* b = slurp(dir, name, &bytes);
        REQUIRE(pt_set_buffer(ctx, bindings[k], b, (size_t)bytes), name);
        free(b);
    }
    { long bytes; void* sky = slurp(dir, "sky.bin", &bytes); REQUIRE(pt_set_texture(ctx, 0, skyW, skyH, (const uint8_t*)sky), "pt_set_texture"); free(sky); }
    int32_t* seeds = (int32_t*)malloc(sizeof(int32_t) * (size_t)fps);
    for (int f = 1; f <= fps; f++) seeds[f - 1] = (1234 + 7919 * f) % 10000;
    void* img = NULL;
    int in_flight = 0;
    /* one step of bench.py's pipeline (shard.StepPipeline, lag 2) */
#define STEP() do { REQUIRE(pt_next_image(ctx), "pt_next_image"); REQUIRE(pt_render_batch_async(ctx, 1, fps, seeds), "pt_render_batch_async"); \
                    if (in_flight == 2) REQUIRE(pt_gather_image(ctx, 2, &img), "pt_gather_image"); else in_flight++; } while (0)
#define DRAIN() do { while (in_flight > 0) { in_flight--; REQUIRE(pt_gather_image(ctx, in_flight, &img), "pt_gather_image (drain)"); } } while (0)
    STEP(); DRAIN(); REQUIRE(pt_synchronize(ctx), "pt_synchronize");                     /* untimed: pool and ring allocation */
    for (int k = 0; k < warmup; k++) STEP();
    DRAIN(); REQUIRE(pt_synchronize(ctx), "pt_synchronize");
    const double t0 = now();
    for (int k = 0; k < steps; k++) STEP();
    DRAIN(); REQUIRE(pt_synchronize(ctx), "pt_synchronize"); REQUIRE(pt_stream_wait(ctx), "pt_stream_wait");
    const double dt = now() - t0;
    /* the last image, read back through the runtime the library itself uses */
    void* rt = dlopen("libamdhip64.so", RTLD_NOW | RTLD_NOLOAD);
    if (!rt) rt = dlopen("libamdhip64.so", RTLD_NOW);
    int (*hipMemcpy_)(void*, const void*, size_t, int) = rt ? (int (*)(void*, const void*, size_t, int))dlsym(rt, "hipMemcpy") : NULL;
    int (*hipRuntimeGetVersion_)(int*) = rt ? (int (*)(int*))dlsym(rt, "hipRuntimeGetVersion") : NULL;
    if (!hipMemcpy_ || !img) { fprintf(stderr, "no hipMemcpy / no image\n"); return 1; }
    const size_t fbytes = (size_t)W * H * 16;
    float* host = (float*)malloc(fbytes);
    if (hipMemcpy_(host, img, fbytes, 2 /* hipMemcpyDeviceToHost */) != 0) { fprintf(stderr, "hipMemcpy failed\n"); return 1; }
    uint64_t hsh = 1469598103934665603ull;
    const unsigned char* p = (const unsigned char*)host;
    for (size_t k = 0; k < fbytes; k++) { hsh ^= p[k]; hsh *= 1099511628211ull; }
    int ver = 0; if (hipRuntimeGetVersion_) hipRuntimeGetVersion_(&ver);
    Dl_info info; const char* where = (hipMemcpy_ && dladdr((void*)hipMemcpy_, &info) && info.dli_fname) ? info.dli_fname : "?";
    const double samples = (double)W * H * sres * fps * steps;
    printf("{\"host\": \"plain C, no torch\", \"hip_runtime\": \"%s\", \"hip_runtime_version\": %d, \"streams\": %d, \"width\": %d, \"height\": %d, \"spp_per_step\": %d, \"steps\": %d, \"warmup\": %d, "
           "\"ms_per_step\": %.3f, \"value\": %.3f, \"unit\": \"Msamples/s\", \"count\": %.1f, \"fnv1a\": \"%016llx\"}\n",
           where, ver, streams, W, H, sres * fps, steps, warmup, dt / steps * 1e3, samples / dt / 1e6, (double)host[3], (unsigned long long)hsh);
    REQUIRE(pt_destroy(ctx), "pt_destroy");
    free(host); free(seeds);
    return 0;
}
