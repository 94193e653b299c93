#!/bin/bash
# one-off environment probe of a GPU box: JDK for the JNI shim, HIP runtimes
echo "== jdk"; for t in javac java jar; do command -v $t || echo "$t: not found"; done
find / -name jni.h -not -path '*/proc/*' 2>/dev/null | head -5; echo "jni.h search done"
find / \( -name 'libjvm.so' -o -name '*.jar' \) -not -path '/proc/*' 2>/dev/null | head -5; echo "libjvm/jar search done"
echo "== hip runtimes on disk"; find / -name 'libamdhip64.so*' -not -path '/proc/*' 2>/dev/null
echo "== cpu"; nproc
for m in torch-first lib-first lib-only; do python3 scripts/hip_runtime_probe.py $m 2>&1 | grep -v amdgpu.ids; done
