#!/bin/bash
# Runs on the GPU box: the default bench step and the launch-boundary probe under HIP runtime environment knobs that touch
# the dispatch path (kernel-argument placement, end-of-kernel flush scope, graph packet capture).  Output: gpurun_out/env/*.json
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/env
mkdir -p $OUT
A="--steps 2000 --warmup 200 --no-cpu-baseline --no-pooled-only --no-secondary --no-kernel-breakdown"
run() {   # name, env assignments...
    name=$1; shift
    ( export "$@"; timeout -k 10 120 python3 $R/bench.py $A > $OUT/$name.json 2> $OUT/$name.err ) || echo "$name failed"
    python3 - "$name" "$OUT/$name.json" <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print("%-28s ms/step %.5f  trial %s" % (sys.argv[1], r["ms_per_step"], r["config"]["launch_trial_us_per_step"]))
except Exception as e:
    print(sys.argv[1], "no result", e)
PY
}
run base X_NONE=1
run dev_kernarg HIP_FORCE_DEV_KERNARG=1
run dev_kernarg0 HIP_FORCE_DEV_KERNARG=0
run opt_flush0 AMD_OPT_FLUSH=0
run sys_scope0 ROC_SYSTEM_SCOPE_SIGNAL=0
run graph_capture1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run graph_capture0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run kernarg_copy_opt0 DEBUG_HIP_KERNARG_COPY_OPT=0
run fgs_kernarg0 ROC_USE_FGS_KERNARG=0
run active_wait ROC_ACTIVE_WAIT_TIMEOUT=1000
run combo HIP_FORCE_DEV_KERNARG=1 ROC_SYSTEM_SCOPE_SIGNAL=0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
