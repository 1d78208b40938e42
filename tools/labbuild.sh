#!/bin/bash
# Lab variant of libggcn_hip.so that differs from the product build in a few translation units only: those are compiled
# with the extra flags (or from another git revision), everything else is linked from csrc/build/*.o (run `make` first).
#   tools/labbuild.sh NAME "-DFLAG ..." file1.hip [file2.hip ...]        current sources + flags
#   tools/labbuild.sh NAME@REV "" file1.hip ...                          those files as of git revision REV
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
CS=$R/ed-gated-gcn_amd/csrc
SPEC=$1; FLAGS=$2; shift 2
NAME=${SPEC%@*}; REV=""; [[ "$SPEC" == *@* ]] && REV=${SPEC#*@}
OUT=$R/tools/_lab; mkdir -p $OUT/obj_$NAME
SRC=$CS
if [ -n "$REV" ]; then
  SRC=$OUT/src_$NAME/ed-gated-gcn_amd/csrc; mkdir -p $OUT/src_$NAME
  git -C $R archive $REV ed-gated-gcn_amd/csrc include | tar -x -C $OUT/src_$NAME
fi
OBJS=""
for f in $CS/build/*.o; do
  b=$(basename $f .o); skip=0
  for s in "$@"; do [ "$b.hip" == "$s" ] && skip=1; done
  [ $skip == 0 ] && OBJS="$OBJS $f"
done
PIDS=""
for s in "$@"; do
  rm -f $OUT/obj_$NAME/${s%.hip}.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FLAGS -c $SRC/$s -o $OUT/obj_$NAME/${s%.hip}.o &
  PIDS="$PIDS $!"
done
for p in $PIDS; do wait $p || { echo "labbuild: a translation unit failed to compile"; exit 1; }; done
for s in "$@"; do OBJS="$OBJS $OUT/obj_$NAME/${s%.hip}.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libggcn_$NAME.so $OBJS 2>&1 | grep -v hip-link || true
echo built $NAME
