/* include/pt_api.h — C ABI of the MI355X render path (libpt_hip.so).
 *
 * The reference has no function-level plugin API: its "render call" is the OpenGL state the
 * fragment shader consumes plus one glDrawArrays per frame
 * (/root/reference/src/Main/dispatch.java:693-705).  This ABI accepts exactly that state:
 * SSBO contents by binding point, texture 0, the two per-frame uniforms, and hands back the
 * FRAME accumulation image.  Each entry point cites the reference interface it replaces.
 *
 * Conventions mirrored from the reference: buffers are caller-owned host memory copied at call
 * time (glBufferData semantics, e.g. dispatch.java:210); single-threaded per context (one GL
 * thread, :168); errors are returned as negative codes with a message in pt_last_error()
 * (the reference throws RuntimeException from its check helpers, :1853-1865).
 * There is no CPU fallback: every function fails with PT_ERR_NO_DEVICE when no gfx950 device
 * is usable.
 */
#ifndef PT_API_H
#define PT_API_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct pt_ctx pt_ctx;

enum {
    PT_OK = 0,
    PT_ERR_ARG = -1,        /* bad argument / unknown binding */
    PT_ERR_NO_DEVICE = -2,  /* no usable HIP device */
    PT_ERR_HIP = -3,        /* a HIP runtime call failed */
    PT_ERR_SCENE = -4,      /* scene buffers inconsistent (index out of range, BVH deeper than the
                               reference's int stack[64] (frag.glsl:465), ...) */
    PT_ERR_UNSUPPORTED = -5 /* a request outside what this implementation's encodings hold (e.g. directDiffuse with subsurface
                               materials over more than 65536 BVHs) */
};

/* SSBO binding points of frag.glsl:14-77 accepted by pt_set_buffer */
enum {
    PT_BIND_ORIGIN = 0,      /* vec3 ORIGIN              dispatch.java:554-560, 628-631 */
    PT_BIND_ROTATION = 1,    /* vec3 ROTATION            :562-568, 632-635 */
    PT_BIND_MOUSE = 2,       /* vec3 MOUSE_POS           :570-574, 636-643 */
    PT_BIND_TRIANGLES = 3,   /* 40 f32 per triangle      :386-424 */
    PT_BIND_PARAMS = 4,      /* 12 f32 Parameters        :191-211 */
    PT_BIND_IMPLICITS = 5,   /* ImpData, must be [0]     :429-456 */
    PT_BIND_ELLIPSOIDS = 7,  /* EllipData                :460-487 */
    PT_BIND_BVHDATA = 10,    /* 8 f32 per node           :496-504 */
    PT_BIND_BVHTREE = 11,    /* 3 i32 per node           :505-513 */
    PT_BIND_LEAFTRIS = 12,   /* i32 triangle ids         :515-523 */
    PT_BIND_OBJINDICES = 13, /* [count, root ids]        :525-534 */
    PT_BIND_MATERIALS = 14   /* [48.0, 48 f32/material]  :270-329 */
};

/* Creates a render context on HIP device `device` for a width x height FRAME image
 * (glTexStorage2D(GL_RGBA32F, res, res*screenHratio), dispatch.java:186-189).
 * shard_rank / shard_count select the tile shard this context renders (1 GPU: 0 / 1): the image
 * is cut into 32x8-pixel tiles dealt round-robin to the ranks. */
int pt_create(pt_ctx** out, int device, int width, int height, int shard_rank, int shard_count);
/* ONE context made of several wavefront streams (SURVEY.md §8(b) "pt_create(backend, n_devices, W, H)", §8(e)): what the reference's
 * single thread owning the single GL context (dispatch.java:168, :593-713) can drive.  devices[] names the HIP device of every stream:
 *   {0,1,...,7}      all GPUs of the node, one stream each;
 *   {0,0}            TWO independent streams on one GPU — each with its own path pool and HIP stream, unsynchronised, so the intersect
 *                    kernel of one (issue/latency bound) overlaps the shading kernel of the other (HBM bound) and the launch tails fill:
 *                    11-24 % faster than one stream (round 3: C2 +11 %, C3 +14 %, C5 +19 %, C4 +24 %; profiles/r03_n_bench_*.json, r03_h_memory_pipe.txt (3));
 *   {0,0,1,1,...}    both (a device's entries adjacent, every device the same number of times).
 * Every entry point below works on it: uploads replicate the scene, the render calls render every stream's tile shard concurrently
 * (one host thread per stream inside the library), and pt_read_frame / pt_read_display / pt_gather_image perform the ONE collective
 * of an image — device copies between the streams of a device, ONE RCCL gather (ncclGather, single process, ncclCommInitAll) across
 * devices, un-tiling on devices[0].  Results are bit-identical to a one-stream context. */
int pt_create_multi(pt_ctx** out, const int* devices, int n_devices, int width, int height);
/* The same for a PART of the image: the group's n streams are tile shards first_shard .. first_shard+n-1 of total_shards (one process
 * per GPU, each with several streams on its GPU).  pt_gather_image then delivers the group's packed block — n * pt_shard_slots(W, H,
 * total_shards) accumulators, shard-major — for the host layer's gather across processes; pt_unshard un-tiles the gathered blocks. */
int pt_create_multi_part(pt_ctx** out, const int* devices, int n_devices, int width, int height, int first_shard, int total_shards);
int pt_destroy(pt_ctx* ctx);
const char* pt_last_error(void);

/* glBufferData / glBufferSubData on the SSBO bound at `binding` (copy at call time). */
int pt_set_buffer(pt_ctx* ctx, int binding, const void* data, size_t bytes);
/* Texture upload + bindless handle slot `index` (dispatch.java:334-378): RGBA8, LINEAR, REPEAT, row 0 first as stbi_load
 * delivers it.  Index 0 is the sky (frag.glsl:235-242); the others are what materials' map_* fields name (mapMtl,
 * frag.glsl:210-225, and the raw-texel normal of :827). */
int pt_set_texture(pt_ctx* ctx, int index, int width, int height, const uint8_t* rgba8);

/* resetTexture(FRAME)                                             dispatch.java:732-735 */
int pt_reset_frame(pt_ctx* ctx);

/* One frame: glUniform1i(u_frameCount), glUniform1i(u_seed), glDrawArrays(GL_TRIANGLES,0,6)
 *                                                                  dispatch.java:697-705
 * On return the frame is accumulated in FRAME (in stream order on the context's stream).
 * Limits of this implementation where the shader's are its int range (PT_ERR_ARG / PT_ERR_UNSUPPORTED / PT_ERR_SCENE otherwise):
 * SAMPLE_RES <= 2047, MAX_BOUNCES <= 4095, pixels * frames of a batch < 2^31, BVH depth <= 64 (the shader's own `int stack[64]`,
 * frag.glsl:465), texture indices <= 4095.  Implicit surfaces (binding 5) are accepted and, as in the reference — rayImplicit returns 1e30 before anything
 * else, frag.glsl:385-386 — never hit. */
int pt_render(pt_ctx* ctx, int frame_count, int seed);
/* n_frames consecutive frames (u_frameCount = first_frame .. first_frame+n_frames-1, u_seed =
 * seeds[i]) rendered as ONE wavefront batch; FRAME is accumulated in frame order, so the result
 * is bit-identical to n_frames pt_render calls.  Camera/params must not change inside a batch. */
int pt_render_batch(pt_ctx* ctx, int first_frame, int n_frames, const int32_t* seeds);

/* ---- overlapped batches (no reference counterpart: the GL driver pipelines the reference's draw calls by itself) ----
 * pt_render_batch_async submits a batch like pt_render_batch, gets the GPU going and returns while jobs are still waiting to
 * be handed to path slots (at most ~14 M of them) and paths are in flight; they finish underneath the next batch, so the
 * path pool never drains between batches (and, fed frame by frame, grows with the backlog).  Batches overlap as long as the frame inputs (bindings 0, 1, 2, 4) and the scene are unchanged; a change
 * finishes the running work first.  Frames are still added to the FRAME image in u_frameCount order, bit-identical to the
 * synchronous calls.  Every other entry point that reads or modifies results (pt_read_frame, pt_read_display,
 * pt_synchronize, pt_reset_frame, pt_get_counters, pt_set_option, ...) completes all submitted batches first. */
int pt_render_batch_async(pt_ctx* ctx, int first_frame, int n_frames, const int32_t* seeds);
/* Start a new, zeroed FRAME image for the batches submitted from now on (a ring of four images); batches already submitted
 * still land in the image they were submitted for.  The asynchronous counterpart of pt_reset_frame. */
int pt_next_image(pt_ctx* ctx);
/* Complete (stream-ordered) every batch submitted for the FRAME image `age` pt_next_image calls ago (0 = current, up to 3).
 * A path lives for up to SAMPLE_RES * MAX_BOUNCES iterations, so an image is cheapest to finish one or two images later. */
int pt_finish_image(pt_ctx* ctx, int age);
/* Device pointer of the FRAME image `age` pt_next_image calls ago (0 = current), layout as pt_frame_device (one-device contexts). */
int pt_image_device(pt_ctx* ctx, int age, void** dev_ptr, size_t* n_pixels);
/* The whole FRAME image `age` pt_next_image calls ago as width*height RGBA32F in device memory (row 0 = bottom): completes that
 * image like pt_finish_image and, on a pt_create_multi context, performs its ONE collective (device copies + RCCL gather of the shard
 * accumulators on devices[0] + un-tiling kernel; stream-ordered there, not synchronised; the buffer is reused by the next gather).
 * On a one-stream context it is that context's own image; on a pt_create_multi_part context the group's packed block.
 * The buffer belongs to the context and is written again by the next pt_gather_image / pt_read_frame: a caller that hands it to an
 * asynchronous consumer (a collective on another stream) waits for that consumer before the next call. */
int pt_gather_image(pt_ctx* ctx, int age, void** full_dev);

/* glFinish() (dispatch.java:598) */
int pt_synchronize(pt_ctx* ctx);
/* Waits for what is already enqueued on the context's HIP stream(s) — an accumulation, a gather, an un-tiling — WITHOUT completing
 * the batches still in flight (pt_synchronize does that): what a host layer calls between pt_gather_image and a collective of its own. */
int pt_stream_wait(pt_ctx* ctx);

/* glReadPixels-like read-back of the FRAME image as width*height RGBA32F, row 0 = bottom
 * (rgb = running sum, a = frame count; frag.glsl:924-933).  Synchronises.  A multi-GPU context
 * (pt_create_multi) gathers the shards first (one RCCL gather on devices[0]).  A context that is one
 * shard of a process-per-GPU run (pt_create with shard_count > 1) writes only its own pixels: the
 * gather is then the host layer's single RCCL collective on pt_frame_device(). */
int pt_read_frame(pt_ctx* ctx, float* rgba_out);

/* glTexSubImage2D on FRAME (no call site in the reference: its accumulation dies with the window; SURVEY.md §5 "checkpoint / resume: re-upload sum + count"):
 * writes width*height RGBA32F (layout of pt_read_frame) into the current FRAME image.  FRAME is the only state the path tracer carries from frame to frame
 * (frag.glsl:924-933), so N frames + pt_read_frame, then — in this or a new context with the same scene — pt_write_frame + frames N+1 .. N+M equal N+M frames
 * rendered in one go, bit for bit.  A multi-stream context distributes the image over its shards; a single shard of several takes its own pixels.  Synchronises. */
int pt_write_frame(pt_ctx* ctx, const float* rgba_in);

/* The reference's screenshot path: display colour = FRAME.rgb / frame_count (frag.glsl:932) through an
 * UNORM8 framebuffer (clamp, *255, round to nearest), glReadPixels(GL_RGB, GL_UNSIGNED_BYTE) (dispatch.java:813), the
 * signed-byte packing of :819-822 when java_bytes != 0 (a channel >= 128 borrows 1 from the channel above it), vertical flip
 * (:828-833).  rgb_out: width*height*3 bytes, top row first.  Needs the whole image: a one-GPU or a multi-GPU context.  Synchronises. */
int pt_read_display(pt_ctx* ctx, int frame_count, int java_bytes, uint8_t* rgb_out);
/* ... and the file functions.screenshot writes (dispatch.java:840-848: ImageIO.write(imageOut, "PNG", file)): the pixels of pt_read_display
 * as an 8-bit RGB PNG at `path` (the directory must exist; the reference creates "screenshots/").  The pixel values are the reference's; the
 * file's bytes are not ImageIO's (a PNG encoder is free in its compression: this one stores the scanlines uncompressed), and the Java2D
 * bilinear AffineTransformOp the reference flips with (:828-833) copies rows exactly except for JRE-specific edge handling, which is not
 * restated.  Synchronises. */
int pt_save_png(pt_ctx* ctx, int frame_count, int java_bytes, const char* path);

/* Device-resident accumulator of this shard: n_pixels RGBA32F in shard-local pixel order
 * (shard_count == 1: plain row-major FRAME).  Valid until pt_destroy. */
int pt_frame_device(pt_ctx* ctx, void** dev_ptr, size_t* n_pixels);
/* Number of local pixel slots every shard of this image carries (max over ranks; the packed
 * buffers are padded to it so that an all-gather has equal counts). */
int pt_shard_slots(int width, int height, int shard_count, size_t* n_slots);
/* global pixel index (y*width+x) of every local slot of shard `rank`, -1 for padding */
int pt_shard_map(int width, int height, int shard_rank, int shard_count, int32_t* pixel_index_out, size_t n_slots);
/* Scatter the all-gathered packed accumulators (shard_count * n_slots RGBA32F, device memory) into
 * a full width*height RGBA32F image (device memory) on the context's stream. */
int pt_unshard(pt_ctx* ctx, const void* gathered_dev, void* full_dev);

/* Launch everything on this HIP stream (e.g. torch's current stream) instead of the context's own. */
int pt_set_stream(pt_ctx* ctx, void* hip_stream);

/* ---- the reference's BVH builder on the GPU: BVH(int triIndicesStart, int triIndicesEnd) dispatch.java:1630-1646,
 * splitTEST :1647-1721, testSplitOnTEST / cost :1722-1752, in double precision, same tree as the Java recursion (and as libpt_host.so).
 * tri9: per triangle of ONE object 9 doubles = triangle.min, triangle.max, triangle.centroid as the triangle constructor computes them
 * (:1237-1255).  Outputs are caller-allocated for the worst case of 2*n_tris nodes; node k is the node with the k-th id in creation
 * (DFS pre-order) order, ids local to the object:
 *   node_bounds 6 doubles (min, max) | node_links 2 ints (left, right; -1,-1 for a leaf) | node_leaf 2 ints ([start,end) into leaf_tris;
 *   0,0 for an inner node) | leaf_tris n_tris ints: triangle indices (0-based within the object) in leaf order (= flattenBVH's order)
 * PT_ERR_SCENE when the root cannot be split (the reference throws at :1644) or a coordinate is NaN.
 * Needs no context: plug it into the scene producer with pts_set_bvh_builder (include/pt_scene.h). */
int pt_build_bvh(int device, const double* tri9, int64_t n_tris, int32_t* n_nodes, double* node_bounds, int32_t* node_links,
                 int32_t* node_leaf, int32_t* leaf_tris, int32_t* max_depth);

/* Statistics since the last pt_reset_counters (PT_CNT_* order).  Node/triangle/hit-update counts
 * are only collected when option 1 is set (they slow the intersect kernel down). */
enum { PT_CNT_SEGMENTS = 0, PT_CNT_NODES, PT_CNT_TRITESTS, PT_CNT_HITUPD, PT_CNT_SAMPLES, PT_CNT_BOXTESTS,
       PT_CNT_ITERATIONS, PT_CNT_EXTEND_LAUNCHES, PT_CNT_N };
int pt_get_counters(pt_ctx* ctx, uint64_t* out, int n);
int pt_reset_counters(pt_ctx* ctx);

/* Tuning knobs, per-kernel timers and the parity probes of the tests are not part of the boundary: include/pt_debug.h. */

#ifdef __cplusplus
}
#endif
#endif
