#!/bin/sh
# oracle/build_ref.sh -- TEST INFRASTRUCTURE, not product code.
#
# Compiles the *unmodified* reference libfsm + libre sources where they lie
# under $FSM_REF (default /root/reference) with plain gcc, into
#   oracle/_ref/libfsmre.a      static archive (all of adt, print, libfsm, libre)
#   oracle/_ref/libfsm_ref.so   the same objects as one shared library
#   oracle/_ref/re, retest      the reference CLIs (used only to cross-check)
# Nothing is copied into the repo: oracle/_ref/ is git-ignored build output.
# The reference's own build system (bmake + kmkf) is not used; every
# generated lexer/parser is checked in upstream so a flat gcc loop suffices.
# Per-file flags follow the reference Makefiles:
#   -DLX_HEADER='"lexer.h"'   src/libfsm/Makefile:61, src/libre/dialect/Makefile:26
#   -DDIALECT=<d>             src/libre/dialect/Makefile:31
#   -DPCRE_DIALECT=1          src/libre/dialect/Makefile:40
set -e
R=${FSM_REF:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/_ref
if [ ! -d "$R/src/libfsm" ]; then
	echo "build_ref.sh: reference tree not found at $R (expected on the GPU box); keeping prebuilt $OUT" >&2
	exit 0
fi
mkdir -p "$OUT/obj"
CF="-std=c99 -O2 -fPIC -D_POSIX_C_SOURCE=200809L -DNDEBUG -I$R/include -I$R/src -I$R/src/libfsm -I$R/src/libre"
JOBS=${JOBS:-$(nproc)}
LIST=$OUT/obj/cmds.txt
: > "$LIST"
objname() { echo "$OUT/obj/$(echo "${1#$R/src/}" | tr / _ | sed 's/\.c$/.o/')"; }
emit() { # src extra-flags
	o=$(objname "$1")
	if [ ! -f "$o" ] || [ "$1" -nt "$o" ]; then
		echo "gcc $CF $2 -c $1 -o $o" >> "$LIST"
	fi
}
for f in $R/src/adt/*.c $R/src/print/*.c $R/src/libfsm/*.c \
	 $R/src/libfsm/cost/*.c $R/src/libfsm/pred/*.c $R/src/libfsm/print/*.c \
	 $R/src/libfsm/walk/*.c $R/src/libfsm/vm/*.c; do
	case $f in
	*/libfsm/lexer.c) emit "$f" '-DLX_HEADER=\"lexer.h\"' ;;
	*) emit "$f" "" ;;
	esac
done
for f in $R/src/libre/*.c $R/src/libre/class/*.c $R/src/libre/print/*.c; do
	emit "$f" ""
done
for d in glob like literal native pcre sql; do
	for f in $R/src/libre/dialect/$d/*.c; do
		X='-DLX_HEADER=\"lexer.h\"'" -DDIALECT=$d"
		[ $d = pcre ] && X="$X -DPCRE_DIALECT=1"
		# object names must be unique per dialect
		emit "$f" "$X"
	done
done
if [ -s "$LIST" ]; then
	xargs -d '\n' -P "$JOBS" -I{} sh -c '{}' < "$LIST"
fi
rm -f "$OUT/libfsmre.a"
ar rcs "$OUT/libfsmre.a" "$OUT"/obj/*.o
# One shared object = reference objects + our C wrappers (ref_helper.c), linked
# -Bsymbolic: libfsm's re_comp() would otherwise be interposed by glibc's BSD
# re_comp() once loaded into a process such as python.
gcc -std=c99 -O2 -fPIC -D_POSIX_C_SOURCE=200809L -I$R/include -c "$HERE/ref_helper.c" -o "$OUT/ref_helper.o"
# fsm_exec_hoisted(): NOT the reference -- fsm_exec (src/libfsm/exec.c:85-167) with the per-call
# `fsm_all(fsm, fsm_isdfa)` block of :106-109 removed, derived here from a scratch copy so that nothing of
# the reference is stored in the repository.  One of bench.py's CPU baseline lines (SURVEY.md 8(d) line 2).
mkdir -p "$OUT/aux"
sed -e 's/^fsm_exec(/fsm_exec_hoisted(/' \
    -e '/if (!fsm_all(fsm, fsm_isdfa)) {/,/^\t}$/d' "$R/src/libfsm/exec.c" > "$OUT/aux/exec_hoisted.c"
grep -q 'fsm_exec_hoisted(' "$OUT/aux/exec_hoisted.c" && ! grep -q 'fsm_isdfa' "$OUT/aux/exec_hoisted.c"
gcc $CF -c "$OUT/aux/exec_hoisted.c" -o "$OUT/aux/exec_hoisted.o"
rm -f "$OUT/aux/exec_hoisted.c"
gcc -shared -Wl,-Bsymbolic -o "$OUT/libfsm_ref.so" -Wl,--whole-archive "$OUT/libfsmre.a" -Wl,--no-whole-archive "$OUT/ref_helper.o" "$OUT/aux/exec_hoisted.o" -lpthread
gcc $CF -D_XOPEN_SOURCE=700 $R/src/re/main.c "$OUT/libfsmre.a" -o "$OUT/re"
gcc -std=gnu99 -O2 -DNDEBUG -I$R/include -I$R/src $R/src/retest/main.c $R/src/retest/runner.c "$OUT/libfsmre.a" -ldl -o "$OUT/retest"
echo "built: $(ls "$OUT"/obj/*.o | wc -l) objects -> $OUT/libfsmre.a, libfsm_ref.so, re, retest"
