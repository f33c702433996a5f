# Convenience targets (the driver calls __graft_entry__.build()/smoke(), pytest and bench.py directly).
PY ?= python
TAG ?= r03

build:            ## hipcc --offload-arch=gfx950 -> e3-invaraint-diffusion-model_amd/libe3d_hip.so
	$(PY) -c "import __graft_entry__ as g; g.build()"

test-cpu:         ## oracle vs reference fixtures, host logic, C-ABI exports, 2-rank gloo sharding
	$(PY) -m pytest tests -q -m "not gpu"

test-gpu:         ## parity of every kernel / model / sampler / gradient through the C-ABI (needs an MI355X)
	$(PY) -m pytest tests -q -m gpu

smoke:
	$(PY) -c "import __graft_entry__ as g; g.smoke()"

bench:            ## the headline line (one JSON object on stdout)
	$(PY) bench.py

profile:          ## rocprofv3 kernel stats + PMC traffic -> gpurun_out/prof_round, then profiles/
	bash tools/profile_round.sh && $(PY) tools/profile_collect.py $(TAG)

.PHONY: build test-cpu test-gpu smoke bench profile
