#!/bin/sh
# integration/retest/build.sh -- `retest -l hip` and `reperf -l hip`: the reference's own retest(1) and reperf(1)
# with IMPL_HIP added to their shared runner; impl_hip.patch is the whole change.
#
# Nothing of the reference is stored in this repository: runner.c, runner.h, main.c and reperf.c are copied from
# $FSM_REF (default /root/reference) into integration/_build/ (git-ignored, travels to the GPU box like
# every other build output), impl_hip.patch is applied there, and the result is compiled against the
# reference archive oracle/build_ref.sh produced (oracle/_ref/libfsmre.a) and libfsm_hip.so.
#   integration/_build/retest            the patched retest: `retest -l hip tests/retest/*.tst`
#   integration/_build/reperf            the patched reperf: `reperf -l hip reperf/boost.scr`
# Without the reference tree (the GPU box) the prebuilt binary is kept.
set -e
R=${FSM_REF:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(cd "$HERE/../.." && pwd)
OUT=$ROOT/integration/_build
if [ ! -d "$R/src/retest" ]; then
	echo "integration/retest/build.sh: reference tree not found at $R; keeping prebuilt $OUT/retest" >&2
	exit 0
fi
if [ ! -f "$ROOT/oracle/_ref/libfsmre.a" ]; then
	sh "$ROOT/oracle/build_ref.sh"
fi
rm -rf "$OUT/src"
mkdir -p "$OUT/src/retest"
cp "$R/src/retest/runner.c" "$R/src/retest/runner.h" "$R/src/retest/main.c" "$R/src/retest/reperf.c" "$OUT/src/retest/"
(cd "$OUT" && patch -p1 -s < "$HERE/impl_hip.patch")
gcc -std=gnu99 -O2 -DNDEBUG -I"$R/include" -I"$R/src" -I"$ROOT/include" \
	"$OUT/src/retest/main.c" "$OUT/src/retest/runner.c" \
	-Wl,--whole-archive "$ROOT/oracle/_ref/libfsmre.a" -Wl,--no-whole-archive \
	-rdynamic -L"$ROOT/libfsm_amd" -lfsm_hip -Wl,-rpath,'$ORIGIN/../../libfsm_amd' -Wl,-rpath-link,/opt/rocm/lib -ldl \
	-o "$OUT/retest"
# reperf(1), the reference's timing driver over the same runner: `reperf -l hip reperf/boost.scr`
gcc -std=gnu99 -O2 -DNDEBUG -I"$R/include" -I"$R/src" -I"$ROOT/include" \
	"$OUT/src/retest/reperf.c" "$OUT/src/retest/runner.c" \
	-Wl,--whole-archive "$ROOT/oracle/_ref/libfsmre.a" -Wl,--no-whole-archive \
	-rdynamic -L"$ROOT/libfsm_amd" -lfsm_hip -Wl,-rpath,'$ORIGIN/../../libfsm_amd' -Wl,-rpath-link,/opt/rocm/lib -ldl -lm \
	-o "$OUT/reperf"
# --whole-archive + -rdynamic: libfsm is linked statically here, and libfsm_hip.so's shim finds libfsm's
# public functions with dlsym(RTLD_DEFAULT, ...): the executable has to contain and export all of them
# (a plain static link drops the ones retest itself never calls, e.g. fsm_walk_edges).
rm -rf "$OUT/src"   # the patched copies were only needed for the compile
echo "built $OUT/retest $OUT/reperf"
