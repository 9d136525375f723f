#!/bin/bash
# Builds libdiqt_hip.so (gfx950) next to this script.  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -Wno-unused-variable"
OBJS=""
for f in conv_mfma elementwise bgemm conv_direct; do
  if [ ! -f $f.o ] || [ $f.hip -nt $f.o ] || [ common.h -nt $f.o ] || [ ../../include/diqt.h -nt $f.o ]; then
    $HIPCC $FLAGS -c $f.hip -o $f.o &
  fi
  OBJS="$OBJS $f.o"
done
if [ ! -f lib.o ] || [ lib.cpp -nt lib.o ] || [ common.h -nt lib.o ]; then
  $HIPCC $FLAGS -x hip -c lib.cpp -o lib.o &
fi
wait
$HIPCC --offload-arch=gfx950 -shared -fPIC -o libdiqt_hip.so $OBJS lib.o
echo "built $(pwd)/libdiqt_hip.so"
