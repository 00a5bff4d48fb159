#!/bin/sh
# Build libfsm_hip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles
# without a GPU present.
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/../libfsm_hip.so
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
cd "$HERE"
gcc -std=c99 -O2 -fPIC -Wall -Wextra -c shim.c -o shim.o
g++ -std=c++17 -O2 -fPIC -Wall -Wextra -c plan.cpp -o plan.o
g++ -std=c++17 -O2 -fPIC -Wall -Wextra -c strings.cpp -o strings.o
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -c fsm_hip.hip -o fsm_hip.o ${HIPCC_EXTRA}
$HIPCC --offload-arch=gfx950 -shared -fPIC fsm_hip.o plan.o strings.o shim.o -o "$OUT" -ldl
echo "built $OUT"
