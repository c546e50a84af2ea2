#!/bin/bash
# Cheapest end-of-round check: the driver's smoke entry point and the newest GPU tests.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -12 ) > gpurun_out/smoke.log
( timeout 400 python -m pytest tests -m gpu -x -q -k "${KICP_TEST_FILTER:-long_queue or 16_lane or ties}" 2>&1 | tail -8 ) > gpurun_out/pytest_new.log
cat gpurun_out/smoke.log gpurun_out/pytest_new.log
