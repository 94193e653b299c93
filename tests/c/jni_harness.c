/* TEST-ONLY: runs pathtracer-0_amd/java/pt_jni.c WITHOUT a JVM.  The shim is compiled against tests/c/jni_standin/jni.h (the JNI types and the eleven JNIEnv
 * functions it calls, with the specification's prototypes) and linked with this file, which supplies a JNIEnv whose functions work on plain C objects —
 * a direct buffer is (address, capacity), an int[] is (length, data), a pending exception is (class name, message) — and then calls the
 * Java_Main_PtNative_* entry points in the order the reference's Main would (dispatch.java:208-574 uploads, :693-705 draws, :804-851 screenshot),
 * error paths included.  Inputs: one binary file per binding, written by tests/test_abi.py; output: the FRAME image as a file.
 *
 * What it proves: every line of the shim executes, marshals its arguments the way the C ABI expects them and turns error codes into exceptions.
 * What it does not: anything about a real JVM (the real JNINativeInterface_ has some 230 slots in the specification's order; a maintainer builds
 * pt_jni.c against the JDK's jni.h). */
#include <jni.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct _jobject { void* addr; jlong capacity; };                 /* a java.nio.Buffer: addr == NULL: a heap buffer (not direct) */
struct _jclass { const char* name; };
struct _jstring { const char* utf; };
struct _jintArray { jsize n; jint* data; };
struct _jlongArray { jsize n; jlong* data; };

static struct { int pending; char cls[96]; char msg[600]; } exc;
static int array_elements_out;                                   /* Get...ArrayElements without its Release */

static jclass FindClass_(JNIEnv* env, const char* name) { (void)env; static struct _jclass c[4]; static int k; c[k & 3].name = name; return &c[k++ & 3]; }
static jint ThrowNew_(JNIEnv* env, jclass clazz, const char* msg) {
    (void)env; exc.pending = 1; snprintf(exc.cls, sizeof exc.cls, "%s", clazz->name); snprintf(exc.msg, sizeof exc.msg, "%s", msg); return 0;
}
static jsize GetArrayLength_(JNIEnv* env, jarray a) { (void)env; return *(jsize*)a; }          /* both array structs start with their length */
static jint* GetIntArrayElements_(JNIEnv* env, jintArray a, jboolean* isCopy) { (void)env; if (isCopy) *isCopy = 0; array_elements_out++; return a->data; }
static void ReleaseIntArrayElements_(JNIEnv* env, jintArray a, jint* e, jint mode) { (void)env; (void)a; (void)e; (void)mode; array_elements_out--; }
static jlongArray NewLongArray_(JNIEnv* env, jsize len) {
    (void)env; jlongArray a = malloc(sizeof *a); a->n = len; a->data = calloc((size_t)len, sizeof(jlong)); return a;
}
static void SetLongArrayRegion_(JNIEnv* env, jlongArray a, jsize start, jsize len, const jlong* buf) {
    (void)env; if (start < 0 || start + len > a->n) { fprintf(stderr, "SetLongArrayRegion out of bounds\n"); exit(3); } memcpy(a->data + start, buf, (size_t)len * sizeof(jlong));
}
static const char* GetStringUTFChars_(JNIEnv* env, jstring s, jboolean* isCopy) { (void)env; if (isCopy) *isCopy = 0; array_elements_out++; return s->utf; }
static void ReleaseStringUTFChars_(JNIEnv* env, jstring s, const char* c) { (void)env; (void)s; (void)c; array_elements_out--; }
static void* GetDirectBufferAddress_(JNIEnv* env, jobject b) { (void)env; return b->addr; }
static jlong GetDirectBufferCapacity_(JNIEnv* env, jobject b) { (void)env; return b->addr ? b->capacity : -1; }

static const struct JNINativeInterface_ table = {FindClass_, ThrowNew_, GetArrayLength_, GetIntArrayElements_, ReleaseIntArrayElements_, NewLongArray_, SetLongArrayRegion_,
                                                 GetStringUTFChars_, ReleaseStringUTFChars_, GetDirectBufferAddress_, GetDirectBufferCapacity_};

/* the natives of Main.PtNative (pt_jni.c) */
jlong Java_Main_PtNative_create(JNIEnv*, jclass, jint, jint, jint, jint, jint);
jlong Java_Main_PtNative_createMulti(JNIEnv*, jclass, jintArray, jint, jint);
void Java_Main_PtNative_destroy(JNIEnv*, jclass, jlong);
void Java_Main_PtNative_setBuffer(JNIEnv*, jclass, jlong, jint, jobject, jlong);
void Java_Main_PtNative_setTexture(JNIEnv*, jclass, jlong, jint, jint, jint, jobject);
void Java_Main_PtNative_resetFrame(JNIEnv*, jclass, jlong);
void Java_Main_PtNative_render(JNIEnv*, jclass, jlong, jint, jint);
void Java_Main_PtNative_renderBatch(JNIEnv*, jclass, jlong, jint, jintArray);
void Java_Main_PtNative_renderBatchAsync(JNIEnv*, jclass, jlong, jint, jintArray);
void Java_Main_PtNative_renderAsync(JNIEnv*, jclass, jlong, jint, jint);
void Java_Main_PtNative_nextImage(JNIEnv*, jclass, jlong);
void Java_Main_PtNative_finishImage(JNIEnv*, jclass, jlong, jint);
jlong Java_Main_PtNative_imageDevice(JNIEnv*, jclass, jlong, jint, jlongArray);
jlong Java_Main_PtNative_gatherImage(JNIEnv*, jclass, jlong, jint);
void Java_Main_PtNative_synchronize(JNIEnv*, jclass, jlong);
void Java_Main_PtNative_streamWait(JNIEnv*, jclass, jlong);
void Java_Main_PtNative_readFrame(JNIEnv*, jclass, jlong, jobject);
void Java_Main_PtNative_writeFrame(JNIEnv*, jclass, jlong, jobject);
void Java_Main_PtNative_readDisplay(JNIEnv*, jclass, jlong, jint, jboolean, jobject);
void Java_Main_PtNative_savePng(JNIEnv*, jclass, jlong, jint, jboolean, jstring);
jlongArray Java_Main_PtNative_getCounters(JNIEnv*, jclass, jlong);
void Java_Main_PtNative_resetCounters(JNIEnv*, jclass, jlong);

static void* slurp(const char* dir, const char* name, long* bytes) {
    char p[1024]; snprintf(p, sizeof p, "%s/%s", dir, name);
    FILE* f = fopen(p, "rb"); if (!f) { fprintf(stderr, "cannot open %s\n", p); exit(2); }
    fseek(f, 0, SEEK_END); *bytes = ftell(f); fseek(f, 0, SEEK_SET);
    void* b = malloc((size_t)*bytes + 16); if (fread(b, 1, (size_t)*bytes, f) != (size_t)*bytes) exit(2); fclose(f); return b;
}
#define NO_EXCEPTION(what) do { if (exc.pending) { fprintf(stderr, "unexpected %s from %s: %s\n", exc.cls, what, exc.msg); return 1; } } while (0)
#define EXPECT_EXCEPTION(cls_, what) do { if (!exc.pending || strcmp(exc.cls, cls_)) { fprintf(stderr, "%s: expected %s, got %s\n", what, cls_, exc.pending ? exc.cls : "nothing"); return 1; } \
                                           printf("%s -> %s: %s\n", what, exc.cls, exc.msg); exc.pending = 0; } while (0)

/* usage: jni_harness <dir with binding_<n>.bin, sky.bin> W H skyW skyH two_streams(0/1) */
int main(int argc, char** argv) {
    if (argc < 7) return 2;
    const char* dir = argv[1]; int W = atoi(argv[2]), H = atoi(argv[3]), skyW = atoi(argv[4]), skyH = atoi(argv[5]), two = atoi(argv[6]);
    const struct JNINativeInterface_* envp = &table; JNIEnv* env = &envp;
    struct _jclass self = {"Main/PtNative"}; jclass cls = &self;
    jlong ctx;
    if (two) { jint dev[2] = {0, 0}; struct _jintArray d = {2, dev}; ctx = Java_Main_PtNative_createMulti(env, cls, &d, W, H); }
    else ctx = Java_Main_PtNative_create(env, cls, 0, W, H, 0, 1);
    NO_EXCEPTION("create");
    /* a heap buffer never reaches the library */
    { struct _jobject heap = {NULL, 12}; Java_Main_PtNative_setBuffer(env, cls, ctx, 0, &heap, 12); EXPECT_EXCEPTION("java/lang/IllegalArgumentException", "setBuffer(heap buffer)"); }
    /* render before any upload: the library's error code becomes a RuntimeException carrying pt_last_error() */
    Java_Main_PtNative_render(env, cls, ctx, 1, 1); EXPECT_EXCEPTION("java/lang/RuntimeException", "render without a scene");
    static const int bindings[] = {0, 1, 2, 3, 4, 5, 7, 10, 11, 12, 13, 14};
    for (unsigned k = 0; k < sizeof bindings / sizeof bindings[0]; k++) {
        char name[64]; snprintf(name, sizeof name, "binding_%d.bin", bindings[k]);
        long bytes; void* data = slurp(dir, name, &bytes);
        struct _jobject buf = {data, bytes / 4};                  /* (capacity counts elements, as GetDirectBufferCapacity does: the shim must not use it for sizes) */
        Java_Main_PtNative_setBuffer(env, cls, ctx, bindings[k], &buf, (jlong)bytes); NO_EXCEPTION(name);
        memset(data, 0xab, (size_t)bytes); free(data);            /* glBufferData semantics: the caller's buffer may be reused at once */
    }
    { long bytes; void* sky = slurp(dir, "sky.bin", &bytes); struct _jobject buf = {sky, bytes};
      Java_Main_PtNative_setTexture(env, cls, ctx, 0, skyW, skyH, &buf); NO_EXCEPTION("setTexture"); free(sky); }
    Java_Main_PtNative_resetFrame(env, cls, ctx); NO_EXCEPTION("resetFrame");
    Java_Main_PtNative_resetCounters(env, cls, ctx); NO_EXCEPTION("resetCounters");
    /* frames 1..6: one draw, a batch of three, two draws left in flight */
    Java_Main_PtNative_render(env, cls, ctx, 1, 9153); NO_EXCEPTION("render");
    { jint seeds[3] = {7072, 4991, 2910}; struct _jintArray s = {3, seeds}; Java_Main_PtNative_renderBatch(env, cls, ctx, 2, &s); NO_EXCEPTION("renderBatch"); }
    Java_Main_PtNative_renderAsync(env, cls, ctx, 5, 829); NO_EXCEPTION("renderAsync");
    { jint seeds[1] = {8748}; struct _jintArray s = {1, seeds}; Java_Main_PtNative_renderBatchAsync(env, cls, ctx, 6, &s); NO_EXCEPTION("renderBatchAsync"); }
    Java_Main_PtNative_synchronize(env, cls, ctx); NO_EXCEPTION("synchronize");
    size_t nf = (size_t)W * H * 4;
    float* frame = malloc(nf * sizeof(float));
    { struct _jobject out = {frame, (jlong)nf}; Java_Main_PtNative_readFrame(env, cls, ctx, &out); NO_EXCEPTION("readFrame"); }
    { char p[1024]; snprintf(p, sizeof p, "%s/frame.bin", dir); FILE* f = fopen(p, "wb"); fwrite(frame, sizeof(float), nf, f); fclose(f); }
    unsigned char* rgb = malloc((size_t)W * H * 3);
    { struct _jobject out = {rgb, (jlong)W * H * 3}; Java_Main_PtNative_readDisplay(env, cls, ctx, 6, 1, &out); NO_EXCEPTION("readDisplay"); }
    { char p[1024]; snprintf(p, sizeof p, "%s/display.bin", dir); FILE* f = fopen(p, "wb"); fwrite(rgb, 1, (size_t)W * H * 3, f); fclose(f); }
    { char p[1024]; snprintf(p, sizeof p, "%s/shot.png", dir); struct _jstring s = {p}; Java_Main_PtNative_savePng(env, cls, ctx, 6, 0, &s); NO_EXCEPTION("savePng"); }
    Java_Main_PtNative_savePng(env, cls, ctx, 6, 0, NULL); EXPECT_EXCEPTION("java/lang/IllegalArgumentException", "savePng(null)");
    { jlongArray c = Java_Main_PtNative_getCounters(env, cls, ctx); NO_EXCEPTION("getCounters");
      printf("counters:"); for (jsize k = 0; k < c->n; k++) printf(" %lld", (long long)c->data[k]); printf("\n"); }
    /* the image ring: a second image, finished one nextImage() later; its device address and slot count */
    Java_Main_PtNative_nextImage(env, cls, ctx); NO_EXCEPTION("nextImage");
    Java_Main_PtNative_renderAsync(env, cls, ctx, 1, 9153); NO_EXCEPTION("renderAsync (second image)");
    Java_Main_PtNative_nextImage(env, cls, ctx); NO_EXCEPTION("nextImage");
    Java_Main_PtNative_finishImage(env, cls, ctx, 1); NO_EXCEPTION("finishImage");
    { jlong p = Java_Main_PtNative_gatherImage(env, cls, ctx, 1); NO_EXCEPTION("gatherImage"); if (!p) { fprintf(stderr, "gatherImage returned 0\n"); return 1; } }
    Java_Main_PtNative_streamWait(env, cls, ctx); NO_EXCEPTION("streamWait");
    if (!two) { jlong n[1] = {0}; struct _jlongArray slots = {1, n}; jlong p = Java_Main_PtNative_imageDevice(env, cls, ctx, 1, &slots); NO_EXCEPTION("imageDevice");
                if (!p || n[0] < (jlong)W * H) { fprintf(stderr, "imageDevice: %lld slots\n", (long long)n[0]); return 1; } }
    else { Java_Main_PtNative_imageDevice(env, cls, ctx, 1, NULL); EXPECT_EXCEPTION("java/lang/RuntimeException", "imageDevice on a several-stream context"); }
    /* resume: the image of frames 1..6 written back into a fresh image, frame 7 on top of it */
    Java_Main_PtNative_nextImage(env, cls, ctx); NO_EXCEPTION("nextImage");
    { struct _jobject in = {frame, (jlong)nf}; Java_Main_PtNative_writeFrame(env, cls, ctx, &in); NO_EXCEPTION("writeFrame"); }
    Java_Main_PtNative_render(env, cls, ctx, 7, 6667); NO_EXCEPTION("render (frame 7 after writeFrame)");
    { struct _jobject out = {frame, (jlong)nf}; Java_Main_PtNative_readFrame(env, cls, ctx, &out); NO_EXCEPTION("readFrame"); }
    { char p[1024]; snprintf(p, sizeof p, "%s/frame7.bin", dir); FILE* f = fopen(p, "wb"); fwrite(frame, sizeof(float), nf, f); fclose(f); }
    { struct _jobject heap = {NULL, 0}; Java_Main_PtNative_writeFrame(env, cls, ctx, &heap); EXPECT_EXCEPTION("java/lang/IllegalArgumentException", "writeFrame(heap buffer)"); }
    Java_Main_PtNative_destroy(env, cls, ctx); NO_EXCEPTION("destroy");
    if (array_elements_out) { fprintf(stderr, "%d Get...Elements / GetStringUTFChars without their Release\n", array_elements_out); return 1; }
    printf("JNI_HARNESS_OK\n");
    return 0;
}
