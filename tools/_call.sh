mkdir -p gpurun_out/r02g
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02g
B="--no-cpu-baseline --no-kernel-events --steps 20 --warmup 5"
timeout 200 python bench.py $B --graph 0 > $O/bench_eager.json 2> $O/bench_eager.err
timeout 200 python bench.py $B --force-dist 1 > $O/bench_split.json 2> $O/bench_split.err
timeout 200 python bench.py $B --force-dist 1 --reducer ddp > $O/bench_ddp1.json 2> $O/bench_ddp1.err
timeout 200 python bench.py $B --force-dist 1 --reducer bts > $O/bench_bts1.json 2> $O/bench_bts1.err
timeout 200 python -m pytest tests/test_gpu_4_model.py -q -k "rccl" > $O/pytest_rccl.log 2>&1
P="--graph 0 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events"
cd /tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -o b -- python $GRAFT_REPO_ROOT/bench.py $P > $O/pmc_f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -o b -- python $GRAFT_REPO_ROOT/bench.py $P > $O/pmc_w.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_traffic.py $O/pmc_f/b_counter_collection.csv $O/pmc_w/b_counter_collection.csv $O/pmc_traffic.json > $O/pmc_traffic.log 2>&1
rm -f $O/pmc_f/b_kernel_trace.csv $O/pmc_w/b_kernel_trace.csv
for f in eager split ddp1 bts1; do head -c 900 $O/bench_$f.json | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['config']['launch'], d['config']['final_loss'])
except Exception as e: print('$f ERR', e)
"; tail -2 $O/bench_$f.err; done
tail -3 $O/pytest_rccl.log; cat $O/pmc_traffic.log; du -sh $O
