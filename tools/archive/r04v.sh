#!/bin/bash
# one-off GPU job (round 4): SQ counters of the general fit_beta (rows in registers) at p = 10, 20 000 x 200
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r04v; mkdir -p $O
export TMPDIR=/tmp
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA"; do
  d=$O/pmc; rm -rf $d
  (cd /tmp && CONTBENCH_ONLY=${1:-3} timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -o p -- python ${GRAFT_REPO_ROOT:-/root/repo}/tools/contbench.py > $O/run.log 2>&1)
  python - $d <<'PY'
import sys, glob, pandas as pd
f = glob.glob(sys.argv[1] + "/**/p_counter_collection.csv", recursive=True)[0]
df = pd.read_csv(f)
df = df[df["Kernel_Name"].str.contains("fit_beta_kernel|fit_disp_kernel")]
kt = pd.read_csv(f.replace("counter_collection", "kernel_trace"))
kt["ms"] = (kt["End_Timestamp"] - kt["Start_Timestamp"]) / 1e6
df = df.merge(kt[["Dispatch_Id", "ms"]], on="Dispatch_Id")
df["k"] = df["Kernel_Name"].str.slice(0, 60)
df = df[df["ms"] >= 0.5 * df.groupby("k")["ms"].transform("max")]
print(df.groupby(["k", "Counter_Name"])["Counter_Value"].mean().unstack().T.to_string())
print(df.groupby("k")[["ms", "VGPR_Count", "Scratch_Size", "LDS_Block_Size"]].mean().to_string())
PY
done
