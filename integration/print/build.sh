#!/bin/sh
# integration/print/build.sh -- FSM_PRINT_HIP: one more output language of fsm_print(), so that the reference's own
# rx(1) and re(1) write a DFA in the form libfsm_hip loads:   rx -l hip patterns > t.fsmhip ; hipgrep t.fsmhip < log
# print_hip.patch: the enumerator (include/fsm/print.h), the case in fsm_print()'s language switch -- on the print_ir tier,
# next to fsm_print_c / fsm_print_ir (src/libfsm/print.c:308-338, :347-369) -- and one row in the -l tables of
# src/rx/main.c and src/re/main.c.  The printer itself is print_hip_ir.c (ours): an ir_print_f that expands libfsm's
# codegen IR (print/ir.h) to rows the way the VM compiler's dfa_table does (vm/ir.c:649-750) and writes the FSMHIP form.
# irflat_check (test infrastructure) compares that expansion with the shim's fsm_walk_edges flattening, file by file.  As in
# integration/retest: the four files are copied from $FSM_REF into the git-ignored integration/_build/, patched,
# compiled against the reference archive (minus its own print.o) and libfsm_hip.so, and the copies are deleted;
# nothing of the reference is stored here.
#   integration/_build/print/rx    integration/_build/print/re    integration/_build/print/irflat_check
set -e
R=${FSM_REF:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(cd "$HERE/../.." && pwd)
OUT=$ROOT/integration/_build
if [ ! -d "$R/src/rx" ]; then
	echo "integration/print/build.sh: reference tree not found at $R; keeping prebuilt $OUT/print/" >&2
	exit 0
fi
if [ ! -f "$ROOT/oracle/_ref/libfsmre.a" ]; then
	sh "$ROOT/oracle/build_ref.sh"
fi
W=$OUT/print_src
rm -rf "$W"
mkdir -p "$W/include/fsm" "$W/src/libfsm" "$W/src/rx" "$W/src/re" "$OUT/print"
cp "$R/include/fsm/print.h" "$W/include/fsm/"
cp "$R/src/libfsm/print.c" "$W/src/libfsm/"
cp "$R/src/rx/main.c" "$W/src/rx/"
cp "$R/src/re/main.c" "$W/src/re/"
(cd "$W" && patch -p1 -s < "$HERE/print_hip.patch")
# the patched print.h first on the include path; print.c keeps its own directory's "print.h" / "internal.h"
INC="-I$W/include -I$R/include -I$R/src -I$R/src/libfsm -I$R/src/libre -I$ROOT/include"
gcc -std=c99 -O2 -DNDEBUG -D_POSIX_C_SOURCE=200809L $INC -c "$W/src/libfsm/print.c" -o "$W/print_hip.o"
gcc -std=c99 -O2 -Wall -Wextra -D_POSIX_C_SOURCE=200809L $INC -c "$HERE/print_hip_ir.c" -o "$W/print_hip_ir.o"
cp "$ROOT/oracle/_ref/libfsmre.a" "$W/libfsmre.a"
ar d "$W/libfsmre.a" libfsm_print.o
LINK="$W/print_hip.o $W/print_hip_ir.o -Wl,--whole-archive $W/libfsmre.a -Wl,--no-whole-archive -rdynamic -L$ROOT/libfsm_amd -lfsm_hip -Wl,-rpath,\$ORIGIN/../../../libfsm_amd -Wl,-rpath-link,/opt/rocm/lib -ldl"
gcc -std=c99 -O2 -DNDEBUG -D_XOPEN_SOURCE=700 $INC "$W/src/rx/main.c" $LINK -o "$OUT/print/rx"
gcc -std=c99 -O2 -DNDEBUG -D_XOPEN_SOURCE=700 $INC "$W/src/re/main.c" $LINK -o "$OUT/print/re"
gcc -std=c99 -O2 -Wall -Wextra -D_POSIX_C_SOURCE=200809L $INC "$HERE/irflat_check.c" $LINK -o "$OUT/print/irflat_check"
rm -rf "$W"
echo "built $OUT/print/rx $OUT/print/re $OUT/print/irflat_check"
