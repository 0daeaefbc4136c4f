#!/bin/bash
# GPU: the new tests of this step (generic split kernel incl. NCSF, wavefront layer-wise inverse), the inverse-related goldens, then the side paths of the bench
mkdir -p gpurun_out/r05
timeout 1200 python -m pytest tests/test_gpu_flows.py -x -q -m gpu -k "generic_split or without_a_generated or wavefront_form or golden or polynomial or inverse" 2>&1 | tail -15 | tee gpurun_out/r05/check2_pytest.txt
timeout 900 python - <<'PY' 2>&1 | tail -60 | tee gpurun_out/r05/check2_side.txt
import json, bench, torch
out = bench.side_paths_report()
for k in ("sospf_d64", "bpf_d64", "generic_split_kernel"):
    print(k, json.dumps(out.get(k), indent=1))
PY
