#!/bin/bash
# short reads through both trace implementations: parity tests, then the whole-pipeline timing of the 1 M-read sample
set -e -o pipefail
for mode in lanes pipeline; do
  echo "=== KMAHIP_TRACE=$mode"
  KMAHIP_TRACE=$mode timeout -k 5 600 python -m pytest tests/test_trace_gpu.py tests/test_reference_binary_gpu.py tests/test_fuzz_gpu.py -x -q 2>&1 | tail -3
  KMAHIP_TRACE=$mode timeout -k 5 600 python tools/pipeline_time.py 1000000 2>&1 | grep -E "align_trace|assemble"
done
