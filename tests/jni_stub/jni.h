/* jni.h -- SYNTAX-CHECK STUB, not a JNI implementation and not part of any build of the product.
 *
 * The authoring image has no JDK, so java/jni/sdpgpu_jni.c (the shim a maintainer of the Java reference would
 * compile against the real <jni.h>) could never be compiled here.  tests/test_jni_syntax.py runs
 * `gcc -fsyntax-only -Wall -Werror` on it against THIS header, which declares just the JNI 1.6 types and the
 * JNINativeInterface_ members the shim uses, with their standard signatures (Java Native Interface Specification,
 * chapter 4).  It catches missing includes, wrong argument counts and typos; it proves nothing about behaviour. */
#ifndef SDP_TEST_JNI_STUB_H
#define SDP_TEST_JNI_STUB_H
#include <stdint.h>

typedef int32_t jint;
typedef int64_t jlong;
typedef int8_t jbyte;
typedef uint8_t jboolean;
typedef double jdouble;
typedef jint jsize;
struct _jobject;
typedef struct _jobject* jobject;
typedef jobject jclass;
typedef jobject jstring;
typedef jobject jarray;
typedef jarray jobjectArray;
typedef jarray jintArray;
typedef jarray jlongArray;
typedef jarray jbyteArray;
typedef jarray jdoubleArray;
#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_ABORT 2

struct JNINativeInterface_;
typedef const struct JNINativeInterface_* JNIEnv;
struct JNINativeInterface_ {
  jclass (*FindClass)(JNIEnv*, const char*);
  jint (*ThrowNew)(JNIEnv*, jclass, const char*);
  void (*DeleteLocalRef)(JNIEnv*, jobject);
  const char* (*GetStringUTFChars)(JNIEnv*, jstring, jboolean*);
  void (*ReleaseStringUTFChars)(JNIEnv*, jstring, const char*);
  jstring (*NewStringUTF)(JNIEnv*, const char*);
  jsize (*GetArrayLength)(JNIEnv*, jarray);
  jobjectArray (*NewObjectArray)(JNIEnv*, jsize, jclass, jobject);
  void (*SetObjectArrayElement)(JNIEnv*, jobjectArray, jsize, jobject);
  jbyteArray (*NewByteArray)(JNIEnv*, jsize);
  jlongArray (*NewLongArray)(JNIEnv*, jsize);
  jdoubleArray (*NewDoubleArray)(JNIEnv*, jsize);
  jint* (*GetIntArrayElements)(JNIEnv*, jintArray, jboolean*);
  jdouble* (*GetDoubleArrayElements)(JNIEnv*, jdoubleArray, jboolean*);
  void (*ReleaseIntArrayElements)(JNIEnv*, jintArray, jint*, jint);
  void (*ReleaseDoubleArrayElements)(JNIEnv*, jdoubleArray, jdouble*, jint);
  void (*GetByteArrayRegion)(JNIEnv*, jbyteArray, jsize, jsize, jbyte*);
  void (*GetIntArrayRegion)(JNIEnv*, jintArray, jsize, jsize, jint*);
  void (*GetLongArrayRegion)(JNIEnv*, jlongArray, jsize, jsize, jlong*);
  void (*GetDoubleArrayRegion)(JNIEnv*, jdoubleArray, jsize, jsize, jdouble*);
  void (*SetByteArrayRegion)(JNIEnv*, jbyteArray, jsize, jsize, const jbyte*);
  void (*SetLongArrayRegion)(JNIEnv*, jlongArray, jsize, jsize, const jlong*);
  void (*SetDoubleArrayRegion)(JNIEnv*, jdoubleArray, jsize, jsize, const jdouble*);
  void* (*GetPrimitiveArrayCritical)(JNIEnv*, jarray, jboolean*);
  void (*ReleasePrimitiveArrayCritical)(JNIEnv*, jarray, void*, jint);
};
#endif
