"""java/jni/sdpgpu_jni.c cannot be built here (no JDK, no jni.h) -- but it can at least be PARSED: gcc -fsyntax-only
against tests/jni_stub/jni.h (standard JNI signatures of the functions the shim uses) with -Wall -Werror, so a missing
include (malloc without <stdlib.h>), a wrong argument count or an ABI entry point that no longer exists is caught
here instead of by the maintainer who first compiles it.  Also checks that every `native` method of SdpGpu.java has its
Java_sdp_gpu_SdpGpu_<name> definition in the shim and vice versa."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_jni_shim_parses_against_stub_header():
    r = subprocess.run(["gcc", "-std=c11", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-Wno-unused-parameter",
                        "-I", os.path.join(ROOT, "tests", "jni_stub"), "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "java", "jni", "sdpgpu_jni.c")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_every_native_method_has_a_shim_function():
    java = open(os.path.join(ROOT, "java", "sdp", "gpu", "SdpGpu.java")).read()
    shim = open(os.path.join(ROOT, "java", "jni", "sdpgpu_jni.c")).read()
    natives = set(re.findall(r"public static native [\w\[\]]+ (\w+)\(", java))
    defined = set(re.findall(r"Java_sdp_gpu_SdpGpu_(\w+)\(", shim))
    assert natives and natives == defined, (sorted(natives - defined), sorted(defined - natives))
