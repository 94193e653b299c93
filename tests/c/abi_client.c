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
    SYM(hip, pt_render) SYM(hip, pt_read_frame)

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
    static const int bindings[] = {3, 5, 7, 10, 11, 12, 13, 14};
    for (unsigned k = 0; k < sizeof bindings / sizeof *bindings; k++) {
        const void* data; size_t bytes;
        pts_get_buffer_(sc, bindings[k], &data, &bytes);
        if (pt_set_buffer_(ctx, bindings[k], data, bytes)) { fprintf(stderr, "binding %d: %s\n", bindings[k], pt_last_error_()); return 1; }
    }
    const float cam[3] = {0.0f, 1.0f, -3.0f}, rot[3] = {0.0f, 0.0f, 0.0f}, mouse[3] = {-1e6f, -1e6f, 0.0f};
    const float params[12] = {1.5f, 1.0f, (float)W, (float)H / (float)W, 8.0f, 4.0f, 0.0f, 0.001f, 1.0f, 1.0f, 0.0f, 1.0f};      /* dispatch.java:191-205 */
    const uint8_t sky[4] = {150, 180, 230, 255};
    pt_set_buffer_(ctx, 0, cam, sizeof cam); pt_set_buffer_(ctx, 1, rot, sizeof rot); pt_set_buffer_(ctx, 2, mouse, sizeof mouse);
    pt_set_buffer_(ctx, 4, params, sizeof params); pt_set_texture_(ctx, 0, 1, 1, sky);
    pt_reset_frame_(ctx);
    for (int f = 1; f <= frames; f++)
        if (pt_render_(ctx, f, (1234 + 7919 * f) % 10000)) { fprintf(stderr, "pt_render: %s\n", pt_last_error_()); return 1; }
    float* frame = (float*)malloc((size_t)W * H * 16);
    if (pt_read_frame_(ctx, frame)) { fprintf(stderr, "pt_read_frame: %s\n", pt_last_error_()); return 1; }
    /* FNV-1a over the raw bits: the test compares it with the same scene driven through the Python wrappers */
    uint64_t hsh = 1469598103934665603ull;
    const unsigned char* p = (const unsigned char*)frame;
    for (size_t k = 0; k < (size_t)W * H * 16; k++) { hsh ^= p[k]; hsh *= 1099511628211ull; }
    double sum = 0; for (size_t k = 0; k < (size_t)W * H; k++) sum += frame[4 * k] + frame[4 * k + 1] + frame[4 * k + 2];
    printf("ABI_CLIENT_OK %dx%d frames %d fnv1a %016llx mean %.6f alpha %.1f\n", W, H, frames, (unsigned long long)hsh, sum / (3.0 * W * H * frames), frame[3]);
    free(frame); pt_destroy_(ctx); pts_destroy_(sc);
    return 0;
}
