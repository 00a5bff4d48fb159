#!/bin/sh
# integration/fsm/build.sh -- `fsm -H`: the reference's own fsm(1) with its text arguments matched on the GPU.
# hip_exec.patch is the whole change to src/fsm/main.c: -H compiles the automaton read from stdin (after the -t / -m
# transformations) for the GPU and matches ALL text arguments in one launch (fsm_hip_exec_batch_offsets), a file (-x)
# through fsm_hip_match_file; an automaton fsm_exec would refuse (not a DFA) is refused by fsm_hip_compile the same way.
# As in integration/re: main.c is copied from $FSM_REF into the git-ignored integration/_build/, patched, compiled
# (main.c alone, as the reference's Makefile does: wordgen.c is commented out there) against the reference archive
# and libfsm_hip.so, and the copy is deleted; nothing of the reference is stored here.
#   integration/_build/fsm               `fsm -H abc abd < dfa.fsm`
set -e
R=${FSM_REF:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(cd "$HERE/../.." && pwd)
OUT=$ROOT/integration/_build
if [ ! -d "$R/src/fsm" ]; then
	echo "integration/fsm/build.sh: reference tree not found at $R; keeping prebuilt $OUT/fsm" >&2
	exit 0
fi
if [ ! -f "$ROOT/oracle/_ref/libfsmre.a" ]; then
	sh "$ROOT/oracle/build_ref.sh"
fi
rm -rf "$OUT/src"
mkdir -p "$OUT/src/fsm"
cp "$R/src/fsm/main.c" "$OUT/src/fsm/"
(cd "$OUT" && patch -p1 -s < "$HERE/hip_exec.patch")
gcc -std=c99 -O2 -DNDEBUG -D_XOPEN_SOURCE=700 \
	-I"$R/include" -I"$R/src" -I"$R/src/fsm" -I"$ROOT/include" \
	"$OUT/src/fsm/main.c" \
	-Wl,--whole-archive "$ROOT/oracle/_ref/libfsmre.a" -Wl,--no-whole-archive \
	-s -rdynamic -L"$ROOT/libfsm_amd" -lfsm_hip -Wl,-rpath,'$ORIGIN/../../libfsm_amd' -Wl,-rpath-link,/opt/rocm/lib -ldl -lm \
	-o "$OUT/fsm"
rm -rf "$OUT/src"
echo "built $OUT/fsm"
