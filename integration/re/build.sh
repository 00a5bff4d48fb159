#!/bin/sh
# integration/re/build.sh -- `re -H`: the reference's own re(1) with a HIP execution mode next to -M (its DFAVM mode).
# hip_exec.patch is the whole change to src/re/main.c: -H compiles the DFA's table for the GPU (fsm_hip_compile) and
# matches ALL string arguments in one launch (fsm_hip_exec_batch_offsets; end states are the ids fsm_exec returns, so
# -z keeps working), files (-x) through fsm_hip_match_file.  As in integration/retest: main.c is copied from $FSM_REF
# into the git-ignored integration/_build/, patched, compiled against the reference archive and libfsm_hip.so, and the
# copy is deleted; nothing of the reference is stored here.
#   integration/_build/re               `re -H -r pcre '^ab+c$' abc abbc xyz`
set -e
R=${FSM_REF:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(cd "$HERE/../.." && pwd)
OUT=$ROOT/integration/_build
if [ ! -d "$R/src/re" ]; then
	echo "integration/re/build.sh: reference tree not found at $R; keeping prebuilt $OUT/re" >&2
	exit 0
fi
if [ ! -f "$ROOT/oracle/_ref/libfsmre.a" ]; then
	sh "$ROOT/oracle/build_ref.sh"
fi
rm -rf "$OUT/src"
mkdir -p "$OUT/src/re"
cp "$R/src/re/main.c" "$OUT/src/re/"
(cd "$OUT" && patch -p1 -s < "$HERE/hip_exec.patch")
gcc -std=c99 -O2 -DNDEBUG -D_XOPEN_SOURCE=700 \
	-I"$R/include" -I"$R/src" -I"$R/src/libfsm" -I"$R/src/libre" -I"$ROOT/include" \
	"$OUT/src/re/main.c" \
	-Wl,--whole-archive "$ROOT/oracle/_ref/libfsmre.a" -Wl,--no-whole-archive \
	-rdynamic -L"$ROOT/libfsm_amd" -lfsm_hip -Wl,-rpath,'$ORIGIN/../../libfsm_amd' -Wl,-rpath-link,/opt/rocm/lib -ldl \
	-o "$OUT/re"
rm -rf "$OUT/src"
echo "built $OUT/re"
