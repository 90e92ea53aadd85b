# Convenience targets; the driver uses __graft_entry__.build() / smoke(), bench.py and pytest directly.
PY ?= python

build:            ## HIP library (gfx950), C++ host layer, oracle, host tests, examples
	$(PY) -c "import __graft_entry__ as g; g.build()"

test-cpu: build   ## oracle vs the reference's known answers, host logic, C-ABI symbols
	$(PY) -m pytest tests -q -m "not gpu"

test-gpu: build   ## parity of the HIP path against the oracle through the C ABI (needs an MI355X)
	$(PY) -m pytest tests -q -m gpu

smoke: build
	$(PY) -c "import __graft_entry__ as g; g.smoke()"

bench: build
	$(PY) bench.py

clean:
	rm -f lcqpow_amd/*.so oracle/*.so tests/cpp/host_tests examples/bin/*

.PHONY: build test-cpu test-gpu smoke bench clean
