# Convenience targets; the driver's entry points are __graft_entry__.py (build, smoke), bench.py and pytest.
PY ?= python

.PHONY: build test test-gpu bench smoke clean sanitize
build:            ## libpfv_hip.so (hipcc, gfx950) + the CPU oracle
	$(PY) -c "import __graft_entry__ as g; g.build()"
test: build       ## CPU suite: oracle, host logic, C-ABI symbols, kernels on the CPU emulator, gloo sharding
	$(PY) -m pytest tests -q -m "not gpu"
test-gpu: build   ## parity suite on a real MI355X
	$(PY) -m pytest tests -q -m gpu
smoke: build
	$(PY) __graft_entry__.py smoke
bench: build
	$(PY) bench.py
clean:
	rm -f pretty-fast-video_amd/libpfv_hip.so oracle/libpfv_oracle.so tests/hipemu/libpfv_emu*.so
sanitize:         ## ASan + UBSan and TSan passes over the host half on the CPU emulator build (tools/sanitize.sh; logs under profiles/)
	bash tools/sanitize.sh all
