package Main;

import java.nio.Buffer;

/**
 * JNI face of libpt_hip.so (include/pt_api.h) for the reference's Java host.
 *
 * One method per C-ABI entry point; direct NIO buffers are passed exactly where the reference hands them to
 * glBufferData / glBufferSubData (dispatch.java:208-574, 628-643).  Errors surface as RuntimeException, like the
 * reference's check helpers (dispatch.java:1853-1865).
 * NOT compiled in this repository's environment (no JDK); see INTEGRATION.md.
 */
public final class PtNative {
    static { System.loadLibrary("pt_jni"); }     // pt_jni.c, linked against libpt_hip.so

    private PtNative() {}

    /** one GPU (shardRank 0 of shardCount 1), or one tile shard of a process-per-GPU run */
    public static native long create(int device, int width, int height, int shardRank, int shardCount);
    /** ONE context for several GPUs of the node (pt_create_multi): what dispatch.java's single thread drives; every method below
     *  works on it, readFrame / readDisplay / gatherImage perform the one RCCL gather of an image */
    public static native long createMulti(int[] devices, int width, int height);
    /** the same for a part of the image: the streams are tile shards firstShard.. of totalShards (one JVM per GPU; pt_create_multi_part) */
    public static native long createMultiPart(int[] devices, int width, int height, int firstShard, int totalShards);
    public static native void destroy(long ctx);
    /** glBufferData(GL_SHADER_STORAGE_BUFFER, buf) + glBindBufferBase(binding): copy at call time */
    public static native void setBuffer(long ctx, int binding, Buffer directBuffer, long bytes);
    /** stbi_load + glTextureSubImage2D + bindless handle slot `index` (dispatch.java:334-378) */
    public static native void setTexture(long ctx, int index, int width, int height, Buffer rgba8);
    /** resetTexture(FRAME) (dispatch.java:732-735) */
    public static native void resetFrame(long ctx);
    /** glUniform1i(u_frameCount), glUniform1i(u_seed), glDrawArrays(GL_TRIANGLES,0,6) (dispatch.java:697-705) */
    public static native void render(long ctx, int frameCount, int seed);
    public static native void renderBatch(long ctx, int firstFrame, int[] seeds);
    /** the same batch left in flight, as the GL driver leaves draw calls until glFinish (pt_render_batch_async) */
    public static native void renderBatchAsync(long ctx, int firstFrame, int[] seeds);
    /** one frame left in flight (pt_render_batch_async with one seed) */
    public static native void renderAsync(long ctx, int frameCount, int seed);
    /** a new, zeroed FRAME image for the batches submitted from now on (ring of four; pt_next_image) */
    public static native void nextImage(long ctx);
    /** complete every batch submitted for the image `age` nextImage() calls ago (pt_finish_image) */
    public static native void finishImage(long ctx, int age);
    /** device address of that image's accumulator (one-GPU contexts; pt_image_device); slots[0] receives the pixel-slot count */
    public static native long imageDevice(long ctx, int age, long[] slots);
    /** device address of the WHOLE image `age` images ago; on a multi-GPU context: the one RCCL gather + un-tiling (pt_gather_image) */
    public static native long gatherImage(long ctx, int age);
    /** glFinish() */
    public static native void synchronize(long ctx);
    /** wait for what is enqueued on the context's streams (a gather, an un-tiling) without completing batches in flight (pt_stream_wait) */
    public static native void streamWait(long ctx);
    /** glReadPixels of the RGBA32F FRAME image into a direct FloatBuffer of width*height*4 floats */
    public static native void readFrame(long ctx, Buffer rgbaOut);
    /** the inverse (pt_write_frame): a saved FRAME image — running sum + count, the path tracer's only persistent state — written back, so an
     *  accumulation goes on where it stopped: N frames, readFrame, writeFrame, frames N+1.. = one uninterrupted run, bit for bit */
    public static native void writeFrame(long ctx, Buffer rgbaIn);
    /** functions.screenshot's pixels (dispatch.java:804-833): width*height*3 bytes, top row first; javaBytes = keep its signed-byte packing */
    public static native void readDisplay(long ctx, int frameCount, boolean javaBytes, Buffer rgbOut);
    /** what functions.screenshot writes (dispatch.java:804-851): those pixels as an 8-bit RGB PNG at `path` (pt_save_png) */
    public static native void savePng(long ctx, int frameCount, boolean javaBytes, String path);
    /** PT_CNT_* statistics since resetCounters: segments, nodes, triangle tests, hit updates, samples, box tests, iterations, launches */
    public static native long[] getCounters(long ctx);
    public static native void resetCounters(long ctx);
}
