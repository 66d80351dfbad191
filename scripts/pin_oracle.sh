#!/bin/bash
# pin_oracle.sh — pin this repository's oracle (and, with a GPU, the HIP library) against a REAL JVector, literal value by literal value.
#
#   JAVA_HOME=/path/to/jdk-22 scripts/pin_oracle.sh /path/to/jvector-checkout          # needs JDK >= 22 and Maven (network for Maven deps)
#   scripts/pin_oracle.sh --dry-run                                                      # no JDK: the same checker on an oracle-made file
#
# What it does: copies jvector-native-hip/ (the Panama shim module + GoldenDump.java) next to jvector-native in the checkout, adds it
# to the root pom's <modules>, compiles, runs GoldenDump under the reference's SCALAR provider (no -Djvector.vectorization_provider:
# DefaultVectorizationProvider — the arithmetic oracle/jv_oracle.c restates), which writes tests/golden/ref/jvector_goldens.bin, and
# then runs tests/test_reference_goldens.py: the oracle against the reference's literals on the CPU, and — when a GPU is present —
# libjvector_hip.so against them through the C ABI.  The image this engine is built in has no JDK: until someone runs
# this script once and commits the file, parity is pinned at the format / layout / known-answer level only.
set -euo pipefail
HERE=$(cd "$(dirname "$0")/.." && pwd)
OUT=$HERE/tests/golden/ref/jvector_goldens.bin
if [ "${1:-}" = "--dry-run" ]; then
    cd "$HERE"
    echo "[pin_oracle] dry run: the golden container format and checker on records the oracle produces (no JDK needed)"
    exec python -m pytest tests/test_reference_goldens.py -q -m "not gpu" -k "oracle_made_file"
fi
JV=${1:?usage: JAVA_HOME=... scripts/pin_oracle.sh /path/to/jvector-checkout | --dry-run}
[ -f "$JV/pom.xml" ] && [ -d "$JV/jvector-native" ] || { echo "[pin_oracle] $JV does not look like a JVector checkout (pom.xml + jvector-native/)"; exit 2; }
command -v mvn >/dev/null || { echo "[pin_oracle] Maven (mvn) is not on PATH"; exit 2; }
"${JAVA_HOME:+$JAVA_HOME/bin/}java" -version 2>&1 | head -1
rm -rf "$JV/jvector-native-hip"
cp -r "$HERE/jvector-native-hip" "$JV/jvector-native-hip"
grep -q "<module>jvector-native-hip</module>" "$JV/pom.xml" || sed -i 's#<module>jvector-native</module>#<module>jvector-native</module>\n        <module>jvector-native-hip</module>#' "$JV/pom.xml"
mkdir -p "$(dirname "$OUT")"
( cd "$JV" && mvn -q -pl jvector-native-hip -am test-compile && mvn -q -pl jvector-native-hip exec:java -Dexec.args="$OUT" )
ls -la "$OUT"
cd "$HERE"
python -m pytest tests/test_reference_goldens.py -q -m "not gpu"
if python -c "import torch,sys; sys.exit(0 if torch.cuda.is_available() else 1)" 2>/dev/null; then
    python -m pytest tests/test_reference_goldens.py -q -m gpu
else
    echo "[pin_oracle] no GPU here: run 'python -m pytest tests/test_reference_goldens.py -m gpu' on an MI355X to pin the HIP library too"
fi
echo "[pin_oracle] done: commit $OUT — parity is pinned against the reference's literals from now on"
