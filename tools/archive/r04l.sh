#!/bin/bash
# one-off GPU job (round 4): the chain fuzzer on the round's new coverage + the routine fuzzer
cd "${GRAFT_REPO_ROOT:-.}"
timeout 1500 python tests/gpu_fuzz_chain.py ${1:-400000} ${2:-1500} 2>&1 | grep -v " +host$\| SKIP \|^seed" | tail -30
timeout 600 python tests/gpu_fuzz.py ${3:-500000} ${4:-600} 2>&1 | tail -3
