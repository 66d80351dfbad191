#!/bin/bash
# Developer aid: an A/B library for one GPU session.  Recompiles ONE translation unit of jvector_amd/csrc with extra flags and links
# it with the default build's other objects into build/variants/libjvector_hip_<name>.so; JVECTOR_HIP_LIBRARY=<that file> makes
# jvector_amd load it instead of the default library (jvector_amd/_lib.py).  usage: scripts/build_variant.sh <name> <tu.hip> "<flags>"
set -euo pipefail
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; TU=$2; FLAGS=${3:-}
make -C $R/jvector_amd/csrc -j8 >/dev/null
D=$R/build/variants/$NAME; mkdir -p $D
BASE=$(basename ${TU%.*})
EXTRA=""; [ "$BASE" = "k_gsearch_ubr" ] && EXTRA="-fno-slp-vectorize -mllvm -enable-ipra=0"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-function $EXTRA $FLAGS \
  -Rpass-analysis=kernel-resource-usage -c $R/jvector_amd/csrc/$TU -o $D/$BASE.o 2> $D/resources.txt || { cat $D/resources.txt; exit 1; }
OBJS=$(ls $R/build/csrc/*.o | grep -v "/$BASE.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $D/$BASE.o -ldl -o $R/build/variants/libjvector_hip_$NAME.so
grep -E 'Function Name|VGPRs:|ScratchSize' $D/resources.txt | paste - - - | sed 's/remark: [^ ]* *//g; s/\[-Rpass-analysis=kernel-resource-usage\]//g' | grep -E "${4:-ubr_kernelILi2ELi6ELb0}" || true
echo "built build/variants/libjvector_hip_$NAME.so"
