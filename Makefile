# Convenience targets; the driver uses __graft_entry__.build() / pytest / bench.py directly.
PY ?= python

all: lib oracle

lib:
	sh libfsm_amd/csrc/build.sh

oracle:
	$(PY) -c "from oracle import pyoracle; pyoracle.build_oracle(True); pyoracle.build_ref()"

golden:            # needs /root/reference
	$(PY) tests/golden/make_golden.py

check:
	$(PY) -m pytest tests -q -m "not gpu"

check-gpu:         # on an MI355X
	$(PY) -m pytest tests -q -m gpu

bench:
	$(PY) bench.py

# plain-C callers of the boundary; LIBFSM = where the host's libfsm / libre live (here: the oracle's build of the reference)
LIBFSM ?= oracle/_ref
examples: lib
	$(CC) -std=c99 -Wall -Iinclude examples/hipgrep.c -o examples/hipgrep -Llibfsm_amd -lfsm_hip -Wl,-rpath,$(CURDIR)/libfsm_amd -Wl,-rpath-link,/opt/rocm/lib
	$(CC) -std=c99 -Wall -Iinclude examples/retest_hip.c -o examples/retest_hip -L$(LIBFSM) -lfsm_ref -Llibfsm_amd -lfsm_hip \
		-Wl,-rpath,$(CURDIR)/$(LIBFSM) -Wl,-rpath,$(CURDIR)/libfsm_amd -Wl,-rpath-link,/opt/rocm/lib

.PHONY: all lib oracle golden check check-gpu bench examples
