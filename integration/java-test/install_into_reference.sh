#!/bin/sh
# Copies the parity-pinning kit into a checkout of GassiusODude/spectral_analyzer and prints the command that runs it.
#   sh integration/java-test/install_into_reference.sh /path/to/spectral_analyzer
# Nothing of the reference is modified: one test class and its resources are ADDED under src/test.
set -e
REF=${1:?usage: install_into_reference.sh <reference checkout>}
HERE=$(cd "$(dirname "$0")" && pwd)
test -f "$REF/build.gradle" || { echo "$REF does not look like the reference (no build.gradle)"; exit 1; }
mkdir -p "$REF/src/test/java/net/kcundercover/spectral_analyzer" "$REF/src/test/resources/specgpu-fixtures" "$REF/src/test/resources/specgpu-jdsp-probe"
cp "$HERE/SpectralServiceParityTest.java" "$HERE/JdspSemanticsProbe.java" "$REF/src/test/java/net/kcundercover/spectral_analyzer/"
cp "$HERE"/fixtures/* "$REF/src/test/resources/specgpu-fixtures/"
cp "$HERE"/jdsp-probe/* "$REF/src/test/resources/specgpu-jdsp-probe/"
echo "installed.  Unmodified reference against the committed expectations (<= 4 ulp):"
echo "  (cd $REF && ./gradlew test --tests '*SpectralServiceParityTest' -i)"
echo "What JDSP v1.3.1 returns for the probe signals (Welch PSD, down-converter), recorded under build/specgpu-jdsp-observed/:"
echo "  (cd $REF && ./gradlew test --tests '*JdspSemanticsProbe' -i)   then, in this repository:  python tools/fit_jdsp.py $REF/build/specgpu-jdsp-observed"
echo "Drop-in classes + JNI + GPU against the same expectations (fp64 tolerance of tests/test_gpu_parity.py):"
echo "  cp $HERE/../java/net/kcundercover/spectral_analyzer/services/SpectralService.java $REF/src/main/java/net/kcundercover/spectral_analyzer/services/"
echo "  (cd $REF && ./gradlew test --tests '*SpectralServiceParityTest' -i -Dspecgpu.dropin=true -Djava.library.path=$HERE/../../spectral_analyzer_amd/lib)"
echo "  (Gradle forwards -D to the test JVM only if build.gradle says so: add  test { systemProperties System.properties.findAll { it.key.toString().startsWith('specgpu') }; jvmArgs \"-Djava.library.path=\${System.getProperty('java.library.path')}\" }  for the second run)"
