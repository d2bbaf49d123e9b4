mkdir -p gpurun_out/r02l
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02l
timeout 900 python -m pytest tests -m gpu -x -q --durations=5 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_bf16.json 2> $O/bench_bf16.err
timeout 200 python tools/kernel_probe.py --set wgrad --iters 10 > $O/wgrad.jsonl 2> $O/wgrad.err
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 10 > $O/stats.log 2>&1
cd $GRAFT_REPO_ROOT
rm -f $O/stats/bench_kernel_trace.csv
tail -8 $O/pytest_gpu.log; tail -1 $O/smoke.log; cut -c1-100 $O/wgrad.jsonl
python - <<PY
import json
for l in open("$O/bench_bf16.json"):
    if l.startswith("{"):
        d=json.loads(l); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline_lpg"]["frac"]); print(d["kernel_time_ms_per_step"]); print(d["hip_kernels_ms_per_step"], d["cpu_baseline"]["value"])
PY
head -12 $O/stats/bench_kernel_stats.csv | cut -c1-120
