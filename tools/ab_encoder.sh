#!/bin/bash
# Same-box A/B of how the STOCK encoder is configured: memory format, MIOpen NHWC, find mode.
#   bash tools/ab_encoder.sh <tag>          (on the GPU box through gpurun; results in gpurun_out/<tag>_*.json)
. tools/gpu_guard.sh
T=${1:-ab_enc}
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
A="--no-cpu-baseline --parity 0 --lpg-op 0 --f32-line 0 --eager-steps 5 --steps 20 --warmup 5"
run() {   # run <name> <limit> <args...>
    local n=$1 lim=$2; shift 2
    local t0=$(date +%s)
    run_to $lim python bench.py $A "$@" > $O/${T}_$n.json 2> $O/${T}_$n.err
    echo "$n rc=$? $(( $(date +%s) - t0 ))s: $(python - <<PY
import json
try:
    j = json.loads(open("$O/${T}_$n.json").read().strip().splitlines()[-1])
    print(j["value"], "img/s", j["ms_per_step"], "ms; eager", (j.get("eager") or {}).get("ms_per_step"), "; in-scope", j.get("hip_kernels_ms_per_step"), "ms", j["config"]["launch"][:30])
except Exception as e:
    print("no line:", e)
PY
)"
}
run nchw 300 --channels-last 0 --miopen-nhwc 0
run cl_nhwc 300 --channels-last 1 --miopen-nhwc 1
run cl_only 300 --channels-last 1 --miopen-nhwc 0
run nchw_envnhwc 300 --channels-last 0 --miopen-nhwc 1
run cl_nhwc_find 900 --channels-last 1 --miopen-nhwc 1 --cudnn-benchmark 1
run nchw_find 900 --channels-last 0 --miopen-nhwc 0 --cudnn-benchmark 1
cd /tmp
for n in cl_nhwc nchw; do
  [ $n = cl_nhwc ] && F="--channels-last 1 --miopen-nhwc 1" || F="--channels-last 0 --miopen-nhwc 0"
  run_to 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/${T}_prof_$n -o b -- python $R/bench.py --no-cpu-baseline --parity 0 --lpg-op 0 --f32-line 0 --eager-steps 0 --steps 10 --warmup 3 $F > /dev/null 2> $R/$O/${T}_rocprof_$n.err
  cp $(find /tmp/${T}_prof_$n -name '*kernel_stats.csv' | head -1) $R/$O/${T}_kernel_stats_$n.csv 2> /dev/null
done
cd $R
ls -la $O | grep ${T}_
