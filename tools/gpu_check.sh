#!/bin/bash
# Run the GPU test suite (or "$@" as a pytest selection) under a timeout and print only the verdict lines.
mkdir -p gpurun_out/check
timeout ${TEST_TIMEOUT:-900} python -m pytest ${@:-tests} -q -m gpu -x > gpurun_out/check/pytest.log 2>&1
echo "pytest rc=$?"
grep -E "passed|failed|error|FAILED|ERROR|Error" gpurun_out/check/pytest.log | tail -15
