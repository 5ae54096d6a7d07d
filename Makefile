# Convenience targets (the driver uses __graft_entry__.build(), which runs the same two sub-makes).
.PHONY: build test-cpu test-gpu bench clean
build:
	$(MAKE) -C pixell.jl_amd/csrc
	$(MAKE) -C oracle
test-cpu: build
	python -m pytest tests -q -m "not gpu"
test-gpu: build
	python -m pytest tests -q -m gpu
bench: build
	python bench.py
clean:
	$(MAKE) -C pixell.jl_amd/csrc clean
	$(MAKE) -C oracle clean
