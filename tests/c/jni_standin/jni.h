/* TEST-ONLY stand-in for <jni.h>: the types and the few JNIEnv functions pathtracer-0_amd/java/pt_jni.c uses, with the signatures the JNI
 * specification gives them (Java Native Interface Specification, ch. 4 "JNI Functions"; the jclass / jstring / j<type>Array types are distinct
 * struct pointers here so that a swapped argument is a compile error, which the real C header — where they are all jobject — would not catch).
 * tests/test_abi.py::test_jni_shim_type_checks compiles the shim against it with -fsyntax-only -Wall -Werror.  It proves that the file is
 * valid C and that every call matches these prototypes; it pins NOTHING about a real JVM, is never linked, and is no part of the product:
 * a maintainer builds pt_jni.c against the JDK's own jni.h. */
#ifndef PT_TEST_JNI_STANDIN_H
#define PT_TEST_JNI_STANDIN_H
#include <stdint.h>

typedef int32_t jint;
typedef int64_t jlong;
typedef uint8_t jboolean;
typedef jint jsize;

struct _jobject;      typedef struct _jobject* jobject;
struct _jclass;       typedef struct _jclass* jclass;
struct _jstring;      typedef struct _jstring* jstring;
struct _jintArray;    typedef struct _jintArray* jintArray;
struct _jlongArray;   typedef struct _jlongArray* jlongArray;
/* GetArrayLength takes any array: the shim passes jintArray and jlongArray; a union-free C stand-in needs one parameter type */
typedef void* jarray;

#define JNI_ABORT 2
#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL

struct JNINativeInterface_;
typedef const struct JNINativeInterface_* JNIEnv;
struct JNINativeInterface_ {
    jclass (*FindClass)(JNIEnv* env, const char* name);
    jint (*ThrowNew)(JNIEnv* env, jclass clazz, const char* msg);
    jsize (*GetArrayLength)(JNIEnv* env, jarray array);
    jint* (*GetIntArrayElements)(JNIEnv* env, jintArray array, jboolean* isCopy);
    void (*ReleaseIntArrayElements)(JNIEnv* env, jintArray array, jint* elems, jint mode);
    jlongArray (*NewLongArray)(JNIEnv* env, jsize len);
    void (*SetLongArrayRegion)(JNIEnv* env, jlongArray array, jsize start, jsize len, const jlong* buf);
    const char* (*GetStringUTFChars)(JNIEnv* env, jstring str, jboolean* isCopy);
    void (*ReleaseStringUTFChars)(JNIEnv* env, jstring str, const char* chars);
    void* (*GetDirectBufferAddress)(JNIEnv* env, jobject buf);
    jlong (*GetDirectBufferCapacity)(JNIEnv* env, jobject buf);
};
#endif
