# Convenience targets (the driver calls __graft_entry__.build()/smoke(), pytest and bench.py directly).
PY ?= python

build:            ## hipcc --offload-arch=gfx950 -> safe-policy-optimization_amd/safepo/_lib/libsafepo_hip.so (no GPU needed)
	$(PY) -c "import __graft_entry__ as g; g.build()"

test-cpu: build   ## oracle vs reference goldens, ABI symbols, host logic, 2-rank gloo
	$(PY) -m pytest tests -x -q -m "not gpu"

test-gpu: build   ## parity through the C ABI on an MI355X
	$(PY) -m pytest tests -x -q -m gpu

smoke: build
	$(PY) -c "import __graft_entry__ as g; g.smoke()"

bench: build      ## one JSON line: env-steps/s, roofline, cpu_baseline
	$(PY) bench.py --gpus 1

golden:           ## regenerate tests/golden/*.npz by running the unmodified reference (needs /root/reference)
	$(PY) oracle/make_golden.py

.PHONY: build test-cpu test-gpu smoke bench golden
