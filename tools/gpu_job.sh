#!/bin/bash
# tools/gpu_job.sh TAG STEP [STEP ...] -- ONE parameterised GPU job (replaces the per-state scripts of round 2).
# Run on the GPU box through gpurun:  gpurun --timeout 900 -- 'bash tools/gpu_job.sh r03a tests bench:C3 bench:C4R:--steps=3'
# Outputs under gpurun_out/<TAG>/ (merged back); summaries worth keeping are copied to profiles/ by hand.
#   tests[:PYTEST_ARGS]        pytest -m gpu (extra args after ':' , e.g. tests:tests/test_gpu_fused.py)
#   bench:CFG[:ARGS...]        python bench.py --config CFG ARGS (':'-separated; without CPU baseline / hostpath unless +cpu / +host)
#   stats:CFG[:DEPTH]          rocprofv3 --kernel-trace --stats of bench.py --config CFG (5 steps, --pipeline DEPTH, default 1: one step
#                              at a time, so that the timeline reads step by step), summarised to <TAG>/stats_CFG.md
#   pmc:CFG                    five separate rocprofv3 --pmc passes (SQ counters, FETCH_SIZE, WRITE_SIZE, the f64 op mix, int / cvt / smem) -> <TAG>/pmc_CFG.json
#   py:SCRIPT[:ARGS...]        python SCRIPT ARGS
#   sh:SCRIPT[:ARGS...]        bash SCRIPT ARGS
#   env:NAME=VALUE             export for the following steps (DSQ_* tuning knobs)
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG; mkdir -p "$O"
cd "$R"
export TMPDIR=/tmp
for step in "$@"; do
  IFS=':' read -r -a F <<< "$step"
  kind=${F[0]}
  t0=$(date +%s)
  case $kind in
    env) export "${F[1]}"; echo "[gpu_job] export ${F[1]}";;
    tests)
      timeout 900 python -m pytest ${F[1]:-tests} -m gpu --maxfail=10 -q > "$O/tests.log" 2>&1; echo "[gpu_job] tests rc=$? $(tail -1 "$O/tests.log")";;
    bench)
      cfg=${F[1]}; extra=(); cpu=--no-cpu-baseline; host=--no-hostpath; var=--no-variants; name=$cfg
      for a in "${F[@]:2}"; do
        case $a in +cpu) cpu=;; +host) host=;; +var) var=;; name=*) name=${a#name=};; *) extra+=("$a");; esac
      done
      timeout 900 python bench.py --config "$cfg" $cpu $host $var "${extra[@]}" > "$O/bench_$name.json" 2> "$O/bench_$name.err"
      echo "[gpu_job] bench $name rc=$?"
      python - "$O/bench_$name.json" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("   ", j["config"]["name"], "gpus", j["n_gpus"], "genes/s", round(j["value"]), "ms/step", round(j["ms_per_step"],3), "roofline", round(j["roofline"]["frac"],5),
          {k:round(v["avg_ms"],3) for k,v in j["kernels"].items()}, "hostpath", j.get("hostpath_ms"), "fused-host", j.get("hostpath_fused_ms"),
          (j.get("hostpath_fused") or {}).get("with_assays_ms"), "cpu", (j.get("cpu_baseline") or {}).get("value"), "chain", j["config"]["chain"][:6], j["result_digest"][:12], "parity", j.get("parity"), "variants", [(v["seed"], v["size_factors"], round(v["ms_per_step"], 3)) for v in (j.get("variants") or [])])
except Exception as e:
    print("    FAILED", e)
PY
      ;;
    stats)
      cfg=${F[1]}; d=$O/prof_$cfg; rm -rf "$d"
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$d" -o trace -- python "$R/bench.py" --config "$cfg" --steps 5 --warmup 2 --pipeline ${F[2]:-1} --no-cpu-baseline --no-hostpath --no-variants --no-parity > "$O/stats_$cfg.log" 2>&1)
      python tools/timeline.py "$(find "$d" -name '*.db' | head -1)" > "$O/timeline_$cfg.txt" 2>&1
      python profiles/summarize_rocpd.py "$(find "$d" -name '*.db' | head -1)" > "$O/stats_$cfg.md" 2>> "$O/stats_$cfg.log"; echo "[gpu_job] stats $cfg rc=$?"; head -25 "$O/stats_$cfg.md"
      rm -rf "$d";;
    pmc)
      cfg=${F[1]}
      k=0
      for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" \
                 "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64" "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_SMEM"; do
        k=$((k+1)); d=$O/pmc${k}_$cfg; rm -rf "$d"
        (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$d" -o p -- python "$R/bench.py" --config "$cfg" --steps 2 --warmup 1 --pipeline 1 --no-cpu-baseline --no-hostpath --no-variants --no-parity > "$O/pmc${k}_$cfg.log" 2>&1)
      done
      P1=$(dirname "$(find "$O/pmc1_$cfg" -name p_counter_collection.csv | head -1)"); P2=$(dirname "$(find "$O/pmc2_$cfg" -name p_counter_collection.csv | head -1)"); P3=$(dirname "$(find "$O/pmc3_$cfg" -name p_counter_collection.csv | head -1)")
      export DSQ_PMC_EXTRA="$(dirname "$(find "$O/pmc4_$cfg" -name p_counter_collection.csv | head -1)"):$(dirname "$(find "$O/pmc5_$cfg" -name p_counter_collection.csv | head -1)")"
      NG=$(python -c "import bench; print(bench.CONFIGS['$cfg']['genes'])")
      python tools/pmc_summary.py "$P2" "$P3" "$P1" "$O/pmc_$cfg.json" "$NG" "rocprofv3 --pmc passes (SQ_*, FETCH_SIZE, WRITE_SIZE separately) of bench.py --config $cfg --steps 2 --warmup 1, state $TAG; means over the FULL-SIZE launches of each kernel (dispatches joined with the kernel trace of the same pass; a launch counts when it ran for >= half of the kernel's longest launch); FETCH_SIZE x2 (gfx950 note) + WRITE_SIZE; tools/gpu_job.sh pmc:$cfg" > "$O/pmc_$cfg.log" 2>&1
      echo "[gpu_job] pmc $cfg rc=$?"; tail -3 "$O/pmc_$cfg.log"
      rm -rf "$O/pmc1_$cfg" "$O/pmc2_$cfg" "$O/pmc3_$cfg" "$O/pmc4_$cfg" "$O/pmc5_$cfg";;
    sh)
      name=$(basename "${F[1]}" .sh)
      timeout 900 bash "${F[1]}" "${F[@]:2}" > "$O/$name.log" 2>&1; echo "[gpu_job] sh ${F[1]} rc=$?"; tail -20 "$O/$name.log";;
    py)
      name=$(basename "${F[1]}" .py)
      timeout 900 python "${F[1]}" "${F[@]:2}" > "$O/$name.log" 2>&1; echo "[gpu_job] py ${F[1]} rc=$?"; tail -15 "$O/$name.log";;
    *) echo "[gpu_job] unknown step $step";;
  esac
  echo "[gpu_job] $step took $(( $(date +%s) - t0 )) s"
done
