#!/bin/bash
# one-off GPU job (round 4): the parity tables of the HIP path on the final kernels, smoke()
cd "${GRAFT_REPO_ROOT:-.}"
python tools/parity_report.py hip > gpurun_out/r04n/parity_hip.md 2> gpurun_out/r04n/parity_hip.err; echo "parity rc=$?"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
