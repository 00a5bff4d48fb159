#!/bin/sh
# integration/reftests/build.sh -- the reference's own C test programs (tests/endids: 16, tests/re_strings: 4, tests/eager_output: 22),
# compiled where they lie with every fsm_exec() call routed to the HIP path (gcc -include exec_via_hip.h) and linked
# against the reference archive + libfsm_hip.so.  Their own assert()s are the check: exit status 0 = the test passes
# with the GPU doing the matching.  Outputs: integration/_build/reftests/<program>; nothing of the reference is stored.
set -e
R=${FSM_REF:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(cd "$HERE/../.." && pwd)
OUT=$ROOT/integration/_build/reftests
if [ ! -d "$R/tests/endids" ]; then
	echo "integration/reftests/build.sh: reference tree not found at $R; keeping prebuilt $OUT" >&2
	exit 0
fi
if [ ! -f "$ROOT/oracle/_ref/libfsmre.a" ]; then
	sh "$ROOT/oracle/build_ref.sh"
fi
mkdir -p "$OUT"
CF="-std=gnu99 -O1 -UNDEBUG -I$R/include -I$R/src -I$R/src/adt -I$ROOT/include -I$HERE"
LF="-Wl,--whole-archive $ROOT/oracle/_ref/libfsmre.a -Wl,--no-whole-archive -s -rdynamic -L$ROOT/libfsm_amd -lfsm_hip -Wl,-rpath,\$ORIGIN/../../../libfsm_amd -Wl,-rpath-link,/opt/rocm/lib -ldl"
gcc $CF -c "$HERE/exec_via_hip.c" -o "$OUT/exec_via_hip.o"
n=0
for f in "$R"/tests/endids/endids*.c; do
	b=$(basename "$f" .c)
	gcc $CF -include exec_via_hip.h "$f" "$R/tests/endids/utils.c" "$OUT/exec_via_hip.o" $LF -o "$OUT/$b"
	n=$((n + 1))
done
for f in "$R"/tests/re_strings/re_strings*.c; do
	b=$(basename "$f" .c)
	gcc $CF -include exec_via_hip.h "$f" "$R/tests/re_strings/testutil.c" "$OUT/exec_via_hip.o" $LF -o "$OUT/$b"
	n=$((n + 1))
done
for f in "$R"/tests/eager_output/eager_output*.c; do
	b=$(basename "$f" .c)
	gcc $CF -include exec_via_hip.h "$f" "$R/tests/eager_output/utils.c" "$OUT/exec_via_hip.o" $LF -o "$OUT/$b"
	n=$((n + 1))
done
rm -f "$OUT/exec_via_hip.o"
echo "built $n reference test programs in $OUT"
