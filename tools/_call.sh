mkdir -p gpurun_out/r02b
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02b
timeout 60 tools/probes/tr_probe > $O/tr_probe.log 2>&1
timeout 600 python tools/parity_probe.py --cfg c3 --out $O/parity_probe_c3.json > $O/parity_probe_c3.log 2>&1
timeout 200 python bench.py --no-cpu-baseline --no-kernel-events --channels-last 1 > $O/bench_cl1.json 2> $O/bench_cl1.err
timeout 400 python bench.py --no-cpu-baseline --no-kernel-events --cudnn-benchmark 1 > $O/bench_cb1.json 2> $O/bench_cb1.err
tail -5 $O/parity_probe_c3.log; head -c 300 $O/bench_cl1.json; echo; head -c 300 $O/bench_cb1.json; tail -12 $O/tr_probe.log
