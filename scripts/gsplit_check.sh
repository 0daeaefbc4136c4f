#!/bin/bash
# GPU: the generic operand-split kernel — bit equality with the generated kernels, then per-launch times beside them.
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_flows.py -x -q -m gpu -k "generic_split or without_a_generated" 2>&1 | tail -15 | tee gpurun_out/r05/gsplit_pytest.txt
timeout 600 python scripts/static_shapes_bench.py 20 2>&1 | tee gpurun_out/r05/gsplit_shapes.jsonl | cut -c1-900
