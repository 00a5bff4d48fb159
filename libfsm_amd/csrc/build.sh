#!/bin/sh
# Build libfsm_hip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles
# without a GPU present.  The kernel translation units compile in parallel.
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/../libfsm_hip.so
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
HF="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall ${HIPCC_EXTRA}"
cd "$HERE"
pids=""
for u in fsm_hip node multi file kern_tiny kern_lds kern_comb kern_glob kern_glob16; do
	$HIPCC $HF -c $u.hip -o $u.o &
	pids="$pids $!"
done
gcc -std=c99 -O2 -fPIC -Wall -Wextra -c shim.c -o shim.o
g++ -std=c++17 -O2 -fPIC -Wall -Wextra -c plan.cpp -o plan.o
g++ -std=c++17 -O2 -fPIC -Wall -Wextra -c strings.cpp -o strings.o
for p in $pids; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC fsm_hip.o node.o multi.o file.o kern_tiny.o kern_lds.o kern_comb.o kern_glob.o kern_glob16.o plan.o strings.o shim.o -o "$OUT" -ldl -lpthread
echo "built $OUT"
