#!/bin/bash
# one rocprofv3 --pmc pass (SQ counters only) over tools/widebench.py for a few wide designs: how busy the VALU and the LDS are
# under the rolled kernels  ->  <out>/wide_pmc.txt   (counters in a run of their own, kernel trace only: the pool's rule)
cd "$(dirname "$0")/.."
R=$PWD; O=${1:-$R/gpurun_out/widepmc}; shift || true; case $O in /*) ;; *) O=$R/$O;; esac
CASES=${*:-48 31}
mkdir -p "$O"; rm -rf "$O/pmc"
(cd /tmp && TMPDIR=/tmp timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SALU --output-format csv -d "$O/pmc" -o p -- python "$R/tools/widebench.py" $CASES > "$O/run.log" 2>&1)
python - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
f = glob.glob(O + "/pmc/**/p_counter_collection.csv", recursive=True)
if not f:
    print("no counter file"); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); seen = set()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0].replace("void dsq::", "")
    if "_rolled_kernel" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (k, r["Dispatch_Id"])
    if key not in seen: seen.add(key); cnt[k] += 1
out = []
for k in sorted(acc, key=lambda z: -acc[z]["SQ_WAVE_CYCLES"]):
    a = acc[k]; w = a["SQ_WAVE_CYCLES"] or 1
    out.append("%-58s launches %2d  VALU insts/wave-cycle %.3f  VALU active %.3f  LDS insts per VALU inst %.2f  LDS active %.3f  bank-conflict cycles / LDS active %.2f  SALU per VALU %.2f" % (
        k, cnt[k], a["SQ_INSTS_VALU"] / w, a["SQ_ACTIVE_INST_VALU"] / w, a["SQ_INSTS_LDS"] / max(a["SQ_INSTS_VALU"], 1), a["SQ_ACTIVE_INST_LDS"] / w,
        a["SQ_LDS_BANK_CONFLICT"] / max(a["SQ_ACTIVE_INST_LDS"], 1), a["SQ_INSTS_SALU"] / max(a["SQ_INSTS_VALU"], 1)))
open(O + "/wide_pmc.txt", "w").write("\n".join(out) + "\n"); print("\n".join(out))
PY
rm -rf "$O/pmc"
