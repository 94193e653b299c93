/* pt_jni.c — the only file that includes jni.h: maps Main.PtNative onto the C ABI of include/pt_api.h.
 *
 * build (on a box with a JDK; none exists in this repository's environment):
 *   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../../include \
 *       -o libpt_jni.so pt_jni.c -L.. -lpt_hip -Wl,-rpath,'$ORIGIN/..'
 *
 * java.nio direct buffers -> GetDirectBufferAddress (zero copy on the Java side; the library copies at
 * call time, glBufferData semantics); error codes -> RuntimeException with pt_last_error().
 */
#include <jni.h>
#include <stdint.h>
#include <stdio.h>
#include "pt_api.h"

static void throw_rt(JNIEnv* env, const char* where) {
    char msg[512];
    snprintf(msg, sizeof msg, "%s: %s", where, pt_last_error());
    (*env)->ThrowNew(env, (*env)->FindClass(env, "java/lang/RuntimeException"), msg);
}
/* address of a direct NIO buffer, or NULL with an IllegalArgumentException pending: heap buffers and null references never reach
 * the library.  (GetDirectBufferCapacity counts ELEMENTS of the buffer's type, so sizes are checked on the Java side, PtNative.java.) */
static void* direct(JNIEnv* env, jobject buf, const char* where) {
    void* p = buf ? (*env)->GetDirectBufferAddress(env, buf) : NULL;
    if (!p) {
        char msg[160];
        snprintf(msg, sizeof msg, "%s needs a direct NIO buffer (BufferUtils.create*Buffer / ByteBuffer.allocateDirect)", where);
        (*env)->ThrowNew(env, (*env)->FindClass(env, "java/lang/IllegalArgumentException"), msg);
    }
    return p;
}
#define CHECK(call, where) do { if ((call) != PT_OK) { throw_rt(env, where); return; } } while (0)
#define CTX(h) ((pt_ctx*)(intptr_t)(h))

JNIEXPORT jlong JNICALL Java_Main_PtNative_create(JNIEnv* env, jclass c, jint device, jint w, jint h, jint rank, jint count) {
    pt_ctx* ctx = NULL;
    if (pt_create(&ctx, device, w, h, rank, count) != PT_OK) { throw_rt(env, "pt_create"); return 0; }
    return (jlong)(intptr_t)ctx;
}
JNIEXPORT jlong JNICALL Java_Main_PtNative_createMulti(JNIEnv* env, jclass c, jintArray devices, jint w, jint h) {
    pt_ctx* ctx = NULL;
    jsize n = (*env)->GetArrayLength(env, devices);
    jint* d = (*env)->GetIntArrayElements(env, devices, NULL);
    int rc = pt_create_multi(&ctx, (const int*)d, (int)n, w, h);
    (*env)->ReleaseIntArrayElements(env, devices, d, JNI_ABORT);
    if (rc != PT_OK) { throw_rt(env, "pt_create_multi"); return 0; }
    return (jlong)(intptr_t)ctx;
}
JNIEXPORT jlong JNICALL Java_Main_PtNative_createMultiPart(JNIEnv* env, jclass c, jintArray devices, jint w, jint h, jint first, jint total) {
    pt_ctx* ctx = NULL;
    jsize n = (*env)->GetArrayLength(env, devices);
    jint* d = (*env)->GetIntArrayElements(env, devices, NULL);
    int rc = pt_create_multi_part(&ctx, (const int*)d, (int)n, w, h, first, total);
    (*env)->ReleaseIntArrayElements(env, devices, d, JNI_ABORT);
    if (rc != PT_OK) { throw_rt(env, "pt_create_multi_part"); return 0; }
    return (jlong)(intptr_t)ctx;
}
JNIEXPORT void JNICALL Java_Main_PtNative_streamWait(JNIEnv* env, jclass c, jlong h) { CHECK(pt_stream_wait(CTX(h)), "pt_stream_wait"); }
JNIEXPORT void JNICALL Java_Main_PtNative_destroy(JNIEnv* env, jclass c, jlong h) { pt_destroy(CTX(h)); }
JNIEXPORT void JNICALL Java_Main_PtNative_setBuffer(JNIEnv* env, jclass c, jlong h, jint binding, jobject buf, jlong bytes) {
    void* p = direct(env, buf, "setBuffer");
    if (!p) return;
    CHECK(pt_set_buffer(CTX(h), binding, p, (size_t)bytes), "pt_set_buffer");
}
JNIEXPORT void JNICALL Java_Main_PtNative_setTexture(JNIEnv* env, jclass c, jlong h, jint index, jint w, jint ht, jobject buf) {
    void* p = direct(env, buf, "setTexture");
    if (!p) return;
    CHECK(pt_set_texture(CTX(h), index, w, ht, (const uint8_t*)p), "pt_set_texture");
}
JNIEXPORT void JNICALL Java_Main_PtNative_resetFrame(JNIEnv* env, jclass c, jlong h) { CHECK(pt_reset_frame(CTX(h)), "pt_reset_frame"); }
JNIEXPORT void JNICALL Java_Main_PtNative_render(JNIEnv* env, jclass c, jlong h, jint frameCount, jint seed) { CHECK(pt_render(CTX(h), frameCount, seed), "pt_render"); }
JNIEXPORT void JNICALL Java_Main_PtNative_renderBatch(JNIEnv* env, jclass c, jlong h, jint first, jintArray seeds) {
    jsize n = (*env)->GetArrayLength(env, seeds);
    jint* s = (*env)->GetIntArrayElements(env, seeds, NULL);
    int rc = pt_render_batch(CTX(h), first, (int)n, (const int32_t*)s);
    (*env)->ReleaseIntArrayElements(env, seeds, s, JNI_ABORT);
    if (rc != PT_OK) throw_rt(env, "pt_render_batch");
}
JNIEXPORT void JNICALL Java_Main_PtNative_renderBatchAsync(JNIEnv* env, jclass c, jlong h, jint first, jintArray seeds) {
    jsize n = (*env)->GetArrayLength(env, seeds);
    jint* s = (*env)->GetIntArrayElements(env, seeds, NULL);
    int rc = pt_render_batch_async(CTX(h), first, (int)n, (const int32_t*)s);
    (*env)->ReleaseIntArrayElements(env, seeds, s, JNI_ABORT);
    if (rc != PT_OK) throw_rt(env, "pt_render_batch_async");
}
JNIEXPORT void JNICALL Java_Main_PtNative_nextImage(JNIEnv* env, jclass c, jlong h) { CHECK(pt_next_image(CTX(h)), "pt_next_image"); }
JNIEXPORT void JNICALL Java_Main_PtNative_finishImage(JNIEnv* env, jclass c, jlong h, jint age) { CHECK(pt_finish_image(CTX(h), age), "pt_finish_image"); }
JNIEXPORT jlong JNICALL Java_Main_PtNative_imageDevice(JNIEnv* env, jclass c, jlong h, jint age, jlongArray slots) {
    void* p = NULL; size_t n = 0;
    if (pt_image_device(CTX(h), age, &p, &n) != PT_OK) { throw_rt(env, "pt_image_device"); return 0; }
    if (slots && (*env)->GetArrayLength(env, slots) > 0) { jlong v = (jlong)n; (*env)->SetLongArrayRegion(env, slots, 0, 1, &v); }
    return (jlong)(intptr_t)p;
}
JNIEXPORT jlong JNICALL Java_Main_PtNative_gatherImage(JNIEnv* env, jclass c, jlong h, jint age) {
    void* p = NULL;
    if (pt_gather_image(CTX(h), age, &p) != PT_OK) { throw_rt(env, "pt_gather_image"); return 0; }
    return (jlong)(intptr_t)p;
}
JNIEXPORT jlongArray JNICALL Java_Main_PtNative_getCounters(JNIEnv* env, jclass c, jlong h) {
    uint64_t cnt[PT_CNT_N];
    if (pt_get_counters(CTX(h), cnt, PT_CNT_N) != PT_OK) { throw_rt(env, "pt_get_counters"); return NULL; }
    jlong v[PT_CNT_N];
    for (int k = 0; k < PT_CNT_N; k++) v[k] = (jlong)cnt[k];
    jlongArray out = (*env)->NewLongArray(env, PT_CNT_N);
    if (out) (*env)->SetLongArrayRegion(env, out, 0, PT_CNT_N, v);
    return out;
}
JNIEXPORT void JNICALL Java_Main_PtNative_resetCounters(JNIEnv* env, jclass c, jlong h) { CHECK(pt_reset_counters(CTX(h)), "pt_reset_counters"); }
JNIEXPORT void JNICALL Java_Main_PtNative_renderAsync(JNIEnv* env, jclass c, jlong h, jint frameCount, jint seed) {
    int32_t s = (int32_t)seed;
    CHECK(pt_render_batch_async(CTX(h), frameCount, 1, &s), "pt_render_batch_async");
}
JNIEXPORT void JNICALL Java_Main_PtNative_readDisplay(JNIEnv* env, jclass c, jlong h, jint frameCount, jboolean javaBytes, jobject out) {
    void* p = direct(env, out, "readDisplay");
    if (!p) return;
    CHECK(pt_read_display(CTX(h), frameCount, javaBytes ? 1 : 0, (uint8_t*)p), "pt_read_display");
}
JNIEXPORT void JNICALL Java_Main_PtNative_savePng(JNIEnv* env, jclass c, jlong h, jint frameCount, jboolean javaBytes, jstring path) {
    const char* p = path ? (*env)->GetStringUTFChars(env, path, NULL) : NULL;
    if (!p) { (*env)->ThrowNew(env, (*env)->FindClass(env, "java/lang/IllegalArgumentException"), "savePng needs a path"); return; }
    int rc = pt_save_png(CTX(h), frameCount, javaBytes ? 1 : 0, p);
    (*env)->ReleaseStringUTFChars(env, path, p);
    if (rc != PT_OK) throw_rt(env, "pt_save_png");
}
JNIEXPORT void JNICALL Java_Main_PtNative_synchronize(JNIEnv* env, jclass c, jlong h) { CHECK(pt_synchronize(CTX(h)), "pt_synchronize"); }
JNIEXPORT void JNICALL Java_Main_PtNative_readFrame(JNIEnv* env, jclass c, jlong h, jobject out) {
    void* p = direct(env, out, "readFrame");
    if (!p) return;
    CHECK(pt_read_frame(CTX(h), (float*)p), "pt_read_frame");
}
JNIEXPORT void JNICALL Java_Main_PtNative_writeFrame(JNIEnv* env, jclass c, jlong h, jobject in) {
    void* p = direct(env, in, "writeFrame");
    if (!p) return;
    CHECK(pt_write_frame(CTX(h), (const float*)p), "pt_write_frame");
}
