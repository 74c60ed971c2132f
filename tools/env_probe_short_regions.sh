R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/env20
mkdir -p $OUT
A="--steps 20 --warmup 5 --repeats 40 --no-cpu-baseline --no-pooled-only --no-secondary --no-secondary-shapes --no-kernel-breakdown --no-floor"
run() {
    name=$1; shift
    ( export "$@"; timeout -k 10 120 python3 $R/bench.py $A > $OUT/$name.json 2> $OUT/$name.err ) || echo "$name failed"
    python3 - "$name" "$OUT/$name.json" <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print("%-22s us/step %.2f  median %.2f  repeats %s  %s %s" % (sys.argv[1], r["ms_per_step"]*1e3, r["ms_per_step_median"]*1e3, {k: round(v*1e3,2) if k!="n" else v for k,v in r["ms_per_step_repeats"].items()}, r["launch_mode"][:12], r["launch_trial_us_per_step"]))
except Exception as e:
    print(sys.argv[1], "no result", e)
PY
}
run base X_NONE=1
run base2 X_NONE=1
run hsa_int0 HSA_ENABLE_INTERRUPT=0
run active_wait ROC_ACTIVE_WAIT_TIMEOUT=1000
run both HSA_ENABLE_INTERRUPT=0 ROC_ACTIVE_WAIT_TIMEOUT=1000
