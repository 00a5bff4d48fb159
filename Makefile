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

.PHONY: all lib oracle golden check check-gpu bench
