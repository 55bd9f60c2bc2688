#!/bin/bash
# copies what tools/r06_profiles.sh left under gpurun_out/<TAG>/ into profiles/r06_* (the committed record of the shipped library)
set -eu
TAG=${1:?tag}; O=gpurun_out/$TAG
cp $O/stats_C3.md profiles/r06_kernel_stats_C3.md; cp $O/stats_C4.md profiles/r06_kernel_stats_C4.md
cp $O/timeline_C3.txt profiles/r06_timeline_C3.txt; cp $O/timeline_C4.txt profiles/r06_timeline_C4.txt
cp $O/pmc_C3.json profiles/r06_pmc_C3.json; cp $O/pmc_C4.json profiles/r06_pmc_C4.json
for c in C2 C4 C4R C5 C3_6250; do cp $O/bench_$c.json profiles/r06_bench_$c.json; done
cp $O/bench_default.json profiles/r06_bench_default.json
grep "^p=" $O/wide.txt > profiles/r06_wide.txt
cp $O/general_path.txt profiles/r06_general_path.txt
cp $O/library.sha256 profiles/r06_library.sha256
cp $O/wide_phases.txt profiles/r06_wide_phases.txt; cp $O/phases_C3.txt profiles/r06_phases_C3.txt
sha256sum deseq2_amd/libdeseq2_mi355x.so; cat profiles/r06_library.sha256
python - <<'PY'
import json
for c in ["default", "C2", "C4", "C4R", "C5", "C3_6250"]:
    d = json.loads(open("profiles/r06_bench_%s.json" % c).read().strip().splitlines()[-1])
    o = d.get("one_call_at_a_time")
    print(c, round(d["ms_per_step"], 3), "ms/step", round(d["value"] / 1e6, 3), "M genes/s; one at a time", o.get("ms_per_step") if isinstance(o, dict) else o,
          "; pmc_matches_library", d.get("pmc_matches_library"), "; parity", {k: d["parity"].get(k) for k in ("rows", "iter_equal", "max_rel")} if "parity" in d else None)
PY
