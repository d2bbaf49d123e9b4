mkdir -p gpurun_out/r02i
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02i
timeout 900 python -m pytest tests -m gpu -x -q --durations=6 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_bf16.json 2> $O/bench_bf16.err
timeout 300 python bench.py --dtype f32 --no-cpu-baseline > $O/bench_f32.json 2> $O/bench_f32.err
timeout 300 python bench.py --height 416 --width 544 --batch 16 --dataset nyu --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err
timeout 300 python bench.py --encoder resnext101_bts --dtype f32 --no-cpu-baseline --steps 5 --warmup 2 > $O/bench_c4.json 2> $O/bench_c4.err
timeout 300 python bench.py --mode infer > $O/bench_infer.json 2> $O/bench_infer.err
tail -6 $O/pytest_gpu.log; tail -2 $O/smoke.log
for f in bf16 f32 c2 c4 infer; do python - <<PY
import json
try:
    for l in open("$O/bench_$f.json"):
        if l.startswith("{"):
            d=json.loads(l); print("$f", d["metric"][:50], d["value"], d.get("ms_per_step"), d.get("roofline",{}).get("frac"), d.get("cpu_baseline",{}) and d["cpu_baseline"].get("value"))
except Exception as e: print("$f ERR", e)
PY
done
