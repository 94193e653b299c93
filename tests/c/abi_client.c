/* A plain-C client of the two C ABIs (include/pt_scene.h, include/pt_api.h): what a compiled host (the reference's Java via JNI, or any
 * C/C++ program) does — build a scene with the reference's DSL calls, pack the SSBO images, hand them to the renderer, draw frames,
 * read FRAME back.  Used by tests/test_abi.py: compiled as C everywhere (the headers must be valid C), run on the GPU box, where its
 * checksum is compared with the Python path's.  usage: abi_client <libdir> <W> <H> <frames> */
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pt_api.h"
#include "pt_scene.h"

#define SYM(lib, name) (*(void**)(&name##_) = dlsym(lib, #name)); if (!name##_) { fprintf(stderr, "missing symbol %s\n", #name); return 2; }

static pts_scene* (*pts_create_)(void);
static void (*pts_destroy_)(pts_scene*);
static const char* (*pts_last_error_)(void);
static int (*pts_add_material_)(pts_scene*, const char*);
static int (*pts_set_last_mtl_)(pts_scene*, const char*, const double*, int);
static int (*pts_add_object_text_)(pts_scene*, const char*, size_t, int, const double[3], const double[3], const double[3], const char*);
static int (*pts_add_ellipsoid_)(pts_scene*, const double[3], const double[3], const double[3], float, int);
static int (*pts_pack_)(pts_scene*);
static int (*pts_get_buffer_)(pts_scene*, int, const void**, size_t*);
static int (*pt_create_)(pt_ctx**, int, int, int, int, int);
static int (*pt_destroy_)(pt_ctx*);
static const char* (*pt_last_error_)(void);
static int (*pt_set_buffer_)(pt_ctx*, int, const void*, size_t);
static int (*pt_set_texture_)(pt_ctx*, int, int, int, const uint8_t*);
static int (*pt_reset_frame_)(pt_ctx*);
static int (*pt_render_)(pt_ctx*, int, int);
static int (*pt_read_frame_)(pt_ctx*, float*);
static int (*pt_write_frame_)(pt_ctx*, const float*);
/* the rest of the boundary (every non-debug symbol of pt_api.h is driven below) */
static int (*pt_create_multi_)(pt_ctx**, const int*, int, int, int);
static int (*pt_create_multi_part_)(pt_ctx**, const int*, int, int, int, int, int);
static int (*pt_stream_wait_)(pt_ctx*);
static int (*pt_render_batch_)(pt_ctx*, int, int, const int32_t*);
static int (*pt_render_batch_async_)(pt_ctx*, int, int, const int32_t*);
static int (*pt_next_image_)(pt_ctx*);
static int (*pt_finish_image_)(pt_ctx*, int);
static int (*pt_image_device_)(pt_ctx*, int, void**, size_t*);
static int (*pt_gather_image_)(pt_ctx*, int, void**);
static int (*pt_synchronize_)(pt_ctx*);
static int (*pt_read_display_)(pt_ctx*, int, int, uint8_t*);
static int (*pt_save_png_)(pt_ctx*, int, int, const char*);
static int (*pt_frame_device_)(pt_ctx*, void**, size_t*);
static int (*pt_shard_slots_)(int, int, int, size_t*);
static int (*pt_shard_map_)(int, int, int, int, int32_t*, size_t);
static int (*pt_unshard_)(pt_ctx*, const void*, void*);
static int (*pt_set_stream_)(pt_ctx*, void*);
static int (*pt_build_bvh_)(int, const double*, int64_t, int32_t*, double*, int32_t*, int32_t*, int32_t*, int32_t*);
static int (*pt_get_counters_)(pt_ctx*, uint64_t*, int);
static int (*pt_reset_counters_)(pt_ctx*);
/* HIP runtime entry points for the device buffers pt_unshard works on (the client itself has no device code) */
static int (*hipMalloc_)(void**, size_t);
static int (*hipFree_)(void*);
static int (*hipMemcpy_)(void*, const void*, size_t, int);

#define REQUIRE(cond, what) do { if (!(cond)) { fprintf(stderr, "ABI client: %s failed (%s)\n", what, pt_last_error_()); return 1; } } while (0)

static int upload(pt_ctx* ctx, pts_scene* sc, int W, int H) {
    static const int bindings[] = {3, 5, 7, 10, 11, 12, 13, 14};
    for (unsigned k = 0; k < sizeof bindings / sizeof *bindings; k++) {
        const void* data; size_t bytes;
        pts_get_buffer_(sc, bindings[k], &data, &bytes);
        if (pt_set_buffer_(ctx, bindings[k], data, bytes)) return 1;
    }
    const float cam[3] = {0.0f, 1.0f, -3.0f}, rot[3] = {0.0f, 0.0f, 0.0f}, mouse[3] = {-1e6f, -1e6f, 0.0f};
    const float params[12] = {1.5f, 1.0f, (float)W, (float)H / (float)W, 8.0f, 4.0f, 0.0f, 0.001f, 1.0f, 1.0f, 0.0f, 1.0f};      /* dispatch.java:191-205 */
    const uint8_t sky[4] = {150, 180, 230, 255};
    return pt_set_buffer_(ctx, 0, cam, sizeof cam) || pt_set_buffer_(ctx, 1, rot, sizeof rot) || pt_set_buffer_(ctx, 2, mouse, sizeof mouse) ||
           pt_set_buffer_(ctx, 4, params, sizeof params) || pt_set_texture_(ctx, 0, 1, 1, sky);
}

static const char* QUAD =
    "o floor\nvn 0 1 0\nv -2 0 -2\nv 2 0 -2\nv 2 0 2\nv -2 0 2\nf 1//1 2//1 3//1\nf 1//1 3//1 4//1\n"
    "o lamp\nusemtl lamp\nvn 0 -1 0\nv -0.5 2 -0.5\nv 0.5 2 -0.5\nv 0.5 2 0.5\nv -0.5 2 0.5\nf 5//2 7//2 6//2\nf 5//2 8//2 7//2\n";

int main(int argc, char** argv) {
    if (argc < 5) { fprintf(stderr, "usage: abi_client <libdir> <W> <H> <frames>\n"); return 2; }
    char path[1024];
    int W = atoi(argv[2]), H = atoi(argv[3]), frames = atoi(argv[4]);
    snprintf(path, sizeof path, "%s/libpt_host.so", argv[1]);
    void* host = dlopen(path, RTLD_NOW);
    snprintf(path, sizeof path, "%s/libpt_hip.so", argv[1]);
    void* hip = dlopen(path, RTLD_NOW);
    if (!host || !hip) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    SYM(host, pts_create) SYM(host, pts_destroy) SYM(host, pts_last_error) SYM(host, pts_add_material) SYM(host, pts_set_last_mtl)
    SYM(host, pts_add_object_text) SYM(host, pts_add_ellipsoid) SYM(host, pts_pack) SYM(host, pts_get_buffer)
    SYM(hip, pt_create) SYM(hip, pt_destroy) SYM(hip, pt_last_error) SYM(hip, pt_set_buffer) SYM(hip, pt_set_texture) SYM(hip, pt_reset_frame)
    SYM(hip, pt_render) SYM(hip, pt_read_frame) SYM(hip, pt_write_frame)
    SYM(hip, pt_create_multi) SYM(hip, pt_create_multi_part) SYM(hip, pt_stream_wait) SYM(hip, pt_render_batch) SYM(hip, pt_render_batch_async) SYM(hip, pt_next_image) SYM(hip, pt_finish_image) SYM(hip, pt_image_device)
    SYM(hip, pt_gather_image) SYM(hip, pt_synchronize) SYM(hip, pt_read_display) SYM(hip, pt_save_png) SYM(hip, pt_frame_device) SYM(hip, pt_shard_slots) SYM(hip, pt_shard_map)
    SYM(hip, pt_unshard) SYM(hip, pt_set_stream) SYM(hip, pt_build_bvh) SYM(hip, pt_get_counters) SYM(hip, pt_reset_counters)
    void* rt = dlopen("libamdhip64.so", RTLD_NOW);
    if (!rt) { fprintf(stderr, "dlopen libamdhip64.so: %s\n", dlerror()); return 2; }
    SYM(rt, hipMalloc) SYM(rt, hipFree) SYM(rt, hipMemcpy)

    /* scene.addMaterial / setLastMtl / addObject / addEllipsoid (dispatch.java:223-266) */
    pts_scene* sc = pts_create_();
    const double kd[3] = {0.8, 0.8, 0.8}, one = 1.0, ke[3] = {12, 12, 12}, zero[3] = {0, 0, 0}, unit[3] = {1, 1, 1}, pm = 1.0, pr = 0.2;
    pts_add_material_(sc, "default"); pts_set_last_mtl_(sc, "Kd", kd, 3); pts_set_last_mtl_(sc, "Pr", &one, 1);
    pts_add_material_(sc, "lamp"); pts_set_last_mtl_(sc, "Ke", ke, 3);
    pts_add_material_(sc, "metal"); pts_set_last_mtl_(sc, "Pm", &pm, 1); pts_set_last_mtl_(sc, "Pr", &pr, 1);
    if (pts_add_object_text_(sc, QUAD, strlen(QUAD), 0, unit, zero, zero, "")) { fprintf(stderr, "scene: %s\n", pts_last_error_()); return 1; }
    const double c0[3] = {0.0, 0.5, 0.0};
    pts_add_ellipsoid_(sc, c0, unit, zero, 0.5f, 2);
    if (pts_pack_(sc)) { fprintf(stderr, "pack: %s\n", pts_last_error_()); return 1; }

    pt_ctx* ctx = NULL;
    if (pt_create_(&ctx, 0, W, H, 0, 1)) { fprintf(stderr, "pt_create: %s\n", pt_last_error_()); return 1; }
    if (upload(ctx, sc, W, H)) { fprintf(stderr, "upload: %s\n", pt_last_error_()); return 1; }
    pt_reset_frame_(ctx);
    pt_reset_counters_(ctx);
    for (int f = 1; f <= frames; f++)
        if (pt_render_(ctx, f, (1234 + 7919 * f) % 10000)) { fprintf(stderr, "pt_render: %s\n", pt_last_error_()); return 1; }
    float* frame = (float*)malloc((size_t)W * H * 16);
    if (pt_read_frame_(ctx, frame)) { fprintf(stderr, "pt_read_frame: %s\n", pt_last_error_()); return 1; }
    /* FNV-1a over the raw bits: the test compares it with the same scene driven through the Python wrappers */
    uint64_t hsh = 1469598103934665603ull;
    const unsigned char* p = (const unsigned char*)frame;
    for (size_t k = 0; k < (size_t)W * H * 16; k++) { hsh ^= p[k]; hsh *= 1099511628211ull; }
    double sum = 0; for (size_t k = 0; k < (size_t)W * H; k++) sum += frame[4 * k] + frame[4 * k + 1] + frame[4 * k + 2];
    /* ---- the rest of the boundary, each result compared with the frame above (same frames, same seeds: same bits) ---- */
    const size_t fbytes = (size_t)W * H * 16;
    float* other = (float*)malloc(fbytes);
    int32_t* seeds = (int32_t*)malloc(sizeof(int32_t) * (size_t)frames);
    for (int f = 1; f <= frames; f++) seeds[f - 1] = (1234 + 7919 * f) % 10000;
    uint64_t cnt[PT_CNT_N];
    REQUIRE(pt_get_counters_(ctx, cnt, PT_CNT_N) == 0 && cnt[PT_CNT_ITERATIONS] > 0 && cnt[PT_CNT_EXTEND_LAUNCHES] == cnt[PT_CNT_ITERATIONS], "pt_get_counters");
    /* pt_render_batch == the same pt_render calls */
    REQUIRE(pt_reset_frame_(ctx) == 0 && pt_render_batch_(ctx, 1, frames, seeds) == 0 && pt_synchronize_(ctx) == 0 && pt_read_frame_(ctx, other) == 0, "pt_render_batch");
    REQUIRE(memcmp(frame, other, fbytes) == 0, "pt_render_batch == frame-at-a-time");
    /* pt_write_frame: the first frame's image written back, the remaining frames on top == all frames in one go */
    if (frames >= 2) {
        REQUIRE(pt_reset_frame_(ctx) == 0 && pt_render_(ctx, 1, seeds[0]) == 0 && pt_read_frame_(ctx, other) == 0, "first frame alone");
        REQUIRE(pt_reset_frame_(ctx) == 0 && pt_write_frame_(ctx, other) == 0 && pt_render_batch_(ctx, 2, frames - 1, seeds + 1) == 0 && pt_read_frame_(ctx, other) == 0, "pt_write_frame");
        REQUIRE(memcmp(frame, other, fbytes) == 0, "pt_write_frame + the remaining frames == all frames");
        REQUIRE(pt_write_frame_(ctx, NULL) == PT_ERR_ARG, "pt_write_frame(NULL) is refused");
    }
    /* overlapped form: new image, asynchronous batches, finish, read */
    void* dptr = NULL; size_t nslots = 0;
    REQUIRE(pt_next_image_(ctx) == 0 && pt_render_batch_async_(ctx, 1, 1, seeds) == 0 && (frames < 2 || pt_render_batch_async_(ctx, 2, frames - 1, seeds + 1) == 0) &&
            pt_finish_image_(ctx, 0) == 0 && pt_synchronize_(ctx) == 0 && pt_image_device_(ctx, 0, &dptr, &nslots) == 0 && dptr && nslots == (size_t)W * H, "overlapped batches");
    REQUIRE(hipMemcpy_(other, dptr, fbytes, 2 /* hipMemcpyDeviceToHost */) == 0 && memcmp(frame, other, fbytes) == 0, "pt_image_device holds the same image");
    void* whole = NULL;
    REQUIRE(pt_gather_image_(ctx, 0, &whole) == 0 && whole == dptr, "pt_gather_image on one GPU is the image itself");
    void* fdev = NULL; size_t fn = 0;
    REQUIRE(pt_frame_device_(ctx, &fdev, &fn) == 0 && fdev == dptr && fn == nslots, "pt_frame_device");
    /* screenshot bytes */
    uint8_t* rgb = (uint8_t*)malloc((size_t)W * H * 3);
    REQUIRE(pt_read_display_(ctx, frames, 1, rgb) == 0, "pt_read_display");
    unsigned long nz = 0; for (size_t k = 0; k < (size_t)W * H * 3; k++) nz += rgb[k] != 0;
    REQUIRE(nz > (unsigned long)W * H, "pt_read_display shows a picture");
    {   /* the screenshot file: signature, IHDR with the image size, an IDAT and an IEND */
        char path[256]; snprintf(path, sizeof path, "/tmp/abi_client_%dx%d_%d.png", W, H, frames);
        REQUIRE(pt_save_png_(ctx, frames, 1, path) == 0, "pt_save_png");
        FILE* pf = fopen(path, "rb"); REQUIRE(pf != NULL, "pt_save_png wrote a file");
        unsigned char hd[24]; size_t got = fread(hd, 1, sizeof hd, pf); fclose(pf); remove(path);
        REQUIRE(got == sizeof hd && memcmp(hd, "\x89PNG\r\n\x1a\n", 8) == 0 && memcmp(hd + 12, "IHDR", 4) == 0, "PNG signature and IHDR");
        REQUIRE(((hd[16] << 24) | (hd[17] << 16) | (hd[18] << 8) | hd[19]) == W && ((hd[20] << 24) | (hd[21] << 16) | (hd[22] << 8) | hd[23]) == H, "PNG size");
    }
    REQUIRE(pt_set_stream_(ctx, NULL) == 0, "pt_set_stream(NULL) = the context's own stream");
    /* tile shards: maps partition the image; two shard contexts + pt_unshard rebuild the frame */
    size_t sslots = 0;
    REQUIRE(pt_shard_slots_(W, H, 2, &sslots) == 0 && sslots >= (size_t)W * H / 2, "pt_shard_slots");
    int32_t* map = (int32_t*)malloc(sizeof(int32_t) * sslots * 2);
    REQUIRE(pt_shard_map_(W, H, 0, 2, map, sslots) == 0 && pt_shard_map_(W, H, 1, 2, map + sslots, sslots) == 0, "pt_shard_map");
    { unsigned char* seen = (unsigned char*)calloc((size_t)W * H, 1); size_t covered = 0;
      for (size_t k = 0; k < 2 * sslots; k++) if (map[k] >= 0) { REQUIRE(map[k] < W * H && !seen[map[k]], "shard maps are disjoint"); seen[map[k]] = 1; covered++; }
      REQUIRE(covered == (size_t)W * H, "shard maps cover the image"); free(seen); }
    pt_ctx* sh[2] = {NULL, NULL};
    void *gathered = NULL, *full = NULL;
    REQUIRE(hipMalloc_(&gathered, 2 * sslots * 16) == 0 && hipMalloc_(&full, fbytes) == 0, "hipMalloc");
    for (int r = 0; r < 2; r++) {
        void* p = NULL; size_t n = 0;
        REQUIRE(pt_create_(&sh[r], 0, W, H, r, 2) == 0 && upload(sh[r], sc, W, H) == 0 && pt_reset_frame_(sh[r]) == 0 && pt_render_batch_(sh[r], 1, frames, seeds) == 0 &&
                pt_synchronize_(sh[r]) == 0 && pt_frame_device_(sh[r], &p, &n) == 0 && n == sslots, "shard context");
        REQUIRE(hipMemcpy_((char*)gathered + (size_t)r * sslots * 16, p, sslots * 16, 3 /* hipMemcpyDeviceToDevice */) == 0, "copy of a shard accumulator");
    }
    REQUIRE(pt_unshard_(sh[0], gathered, full) == 0 && pt_synchronize_(sh[0]) == 0 && hipMemcpy_(other, full, fbytes, 2) == 0, "pt_unshard");
    REQUIRE(memcmp(frame, other, fbytes) == 0, "two shards + pt_unshard == the frame");
    pt_destroy_(sh[0]); pt_destroy_(sh[1]); hipFree_(gathered); hipFree_(full);
    /* ONE context for several devices (here both shards on device 0): the gather happens inside pt_read_frame */
    pt_ctx* multi = NULL; const int devs[2] = {0, 0};
    REQUIRE(pt_create_multi_(&multi, devs, 2, W, H) == 0 && upload(multi, sc, W, H) == 0 && pt_reset_frame_(multi) == 0, "pt_create_multi");
    for (int f = 1; f <= frames; f++) REQUIRE(pt_render_(multi, f, seeds[f - 1]) == 0, "pt_render on the multi-GPU context");
    REQUIRE(pt_read_frame_(multi, other) == 0 && memcmp(frame, other, fbytes) == 0, "multi-GPU context == one GPU");
    REQUIRE(pt_gather_image_(multi, 0, &whole) == 0 && whole != NULL && pt_synchronize_(multi) == 0, "pt_gather_image on the multi-GPU context");
    REQUIRE(hipMemcpy_(other, whole, fbytes, 2) == 0 && memcmp(frame, other, fbytes) == 0, "gathered image == one GPU");
    pt_destroy_(multi);
    /* two groups of two streams each hold shards 0-1 and 2-3 of 4 (one process per GPU, two streams per GPU): each gathers its packed
     * block, the blocks side by side are what the host layer's collective delivers, pt_unshard rebuilds the image */
    { pt_ctx* part[2] = {NULL, NULL}; size_t s4 = 0; void *g4 = NULL, *f4 = NULL;
      REQUIRE(pt_shard_slots_(W, H, 4, &s4) == 0 && hipMalloc_(&g4, 4 * s4 * 16) == 0 && hipMalloc_(&f4, fbytes) == 0, "buffers for four shards");
      for (int p = 0; p < 2; p++) {
          void* block = NULL;
          REQUIRE(pt_create_multi_part_(&part[p], devs, 2, W, H, 2 * p, 4) == 0 && upload(part[p], sc, W, H) == 0 && pt_reset_frame_(part[p]) == 0 &&
                  pt_render_batch_(part[p], 1, frames, seeds) == 0 && pt_gather_image_(part[p], 0, &block) == 0 && block && pt_stream_wait_(part[p]) == 0, "pt_create_multi_part");
          REQUIRE(hipMemcpy_((char*)g4 + (size_t)p * 2 * s4 * 16, block, 2 * s4 * 16, 3) == 0, "copy of a packed block");
      }
      REQUIRE(pt_unshard_(part[0], g4, f4) == 0 && pt_stream_wait_(part[0]) == 0 && hipMemcpy_(other, f4, fbytes, 2) == 0, "pt_unshard of four shards");
      REQUIRE(memcmp(frame, other, fbytes) == 0, "two part groups + pt_unshard == the frame");
      pt_destroy_(part[0]); pt_destroy_(part[1]); hipFree_(g4); hipFree_(f4); }
    /* the reference's BVH builder on the GPU: two triangles -> a root and two leaves */
    { const double tri9[18] = {0, 0, 0, 1, 1, 0, 1.0 / 3, 1.0 / 3, 0, 2, 0, 0, 3, 1, 0, 7.0 / 3, 1.0 / 3, 0};
      int32_t nn = 0, links[8], leaf[8], lt[2], depth = 0; double bounds[24];
      REQUIRE(pt_build_bvh_(0, tri9, 2, &nn, bounds, links, leaf, lt, &depth) == 0 && nn == 3 && links[0] == 1 && links[1] == 2 && links[2] == -1, "pt_build_bvh"); }
    printf("ABI_CLIENT_OK %dx%d frames %d fnv1a %016llx mean %.6f alpha %.1f all-entry-points\n", W, H, frames, (unsigned long long)hsh, sum / (3.0 * W * H * frames), frame[3]);
    free(frame); free(other); free(seeds); free(rgb); free(map); pt_destroy_(ctx); pts_destroy_(sc);
    return 0;
}
