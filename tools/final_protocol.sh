#!/bin/bash
# Round-end measurement set on the binary in bts_amd/lib (run on the GPU box through gpurun; everything lands in gpurun_out/).
#   bash tools/final_protocol.sh <tag> [deadline seconds]
# Steps run in order of what the bench line depends on (default bench line + launch table, the two traffic passes, kernel statistics,
# pytest -m gpu + smoke, then the other configurations / SQ pass / chunk trace); a step that would start after the deadline is skipped
# and says so, so that the call ends on its own inside the GPU-minutes that are left.
# pytest -m gpu + smoke, the bench lines of every BASELINE configuration (the default line carries the f32 child, the eager figure, parity
# and cpu_baseline), rocprofv3 kernel statistics of the bench command, the FETCH_SIZE / WRITE_SIZE passes behind profiles/pmc_traffic.json
# and one SQ pass (MFMA-busy / wait counters).  Since round 6 the three counter passes run on the DEFAULT kernel mix (PMC_ENV empty): rounds
# 3-5 saw rocprofv3 abort the FETCH_SIZE pass (HSA_STATUS_ERROR_INVALID_PACKET_FORMAT) at conv_halo_wide's / conv_igemm_res's first dispatch
# (profiles/r03_pmc_fetch_abort_with_halo_wide.log) and ran them with BTS_CONV_WIDE=0 BTS_RES=0; round 6 probed every counter request on
# single layers (tools/pmc_wide_probe.sh: all pass, profiles/r06_pmcprobe.txt) and then the whole bench step (gpurun r06 call 6: rc 0, 96
# conv_halo_wide + 48 conv_igemm_res rows).  PMC_ENV="BTS_CONV_WIDE=0 BTS_RES=0" bash tools/final_protocol.sh ... restores the old passes if a
# box aborts again (an aborted profiler leaves a process behind that holds the call until its limit).  Every step runs in its own
# process group with a hard kill (tools/gpu_guard.sh).
. tools/gpu_guard.sh
T=${1:-final}
DEADLINE=${2:-100000}
export PMC_ENV=${PMC_ENV:-}
T0=$(date +%s)
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
MD5=$(md5sum bts_amd/lib/libbts_amd.so | cut -d' ' -f1)
# step <limit> <cmd...>: run_to with the limit clipped to what is left before the deadline; skipped when nothing is left
step() {
    local lim=$1; shift
    local left=$(( DEADLINE - ( $(date +%s) - T0 ) ))
    if [ $left -lt 40 ]; then echo "SKIPPED (deadline): $*" | cut -c1-160; return 99; fi
    [ $lim -gt $left ] && lim=$left
    run_to $lim "$@"
}
A="--no-cpu-baseline --parity 0 --lpg-op 0 --f32-line 0 --eager-steps 0"
X="--no-cpu-baseline --lpg-op 0 --f32-line 0 --eager-steps 0"
P="--graph 0 --steps 3 --warmup 1 --no-kernel-events $A"
echo "$MD5  bts_amd/lib/libbts_amd.so" > $O/${T}_pytest_gpu.log
# 1. the bench line (f32 child, eager figure, parity, cpu_baseline) and the launch table
step 420 python bench.py --steps 20 --warmup 5 --dump-launches $O/${T}_launches.json > $O/${T}_bench_bf16.json 2> $O/${T}_bench_bf16.err
cut -c1-160 $O/${T}_bench_bf16.json; echo "t=$(( $(date +%s) - T0 ))"
# 2. FETCH_SIZE / WRITE_SIZE passes + the launch table of their configuration -> pmc_traffic.json
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  ( [ -n "$PMC_ENV" ] && export $PMC_ENV; step 150 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/${T}_pmc_$c -o b -- python $R/bench.py $P > $R/$O/${T}_pmc_$c.out 2> $R/$O/${T}_pmc_$c.err )
  echo "pmc $c rc=$? t=$(( $(date +%s) - T0 ))"
  cp $(find /tmp/${T}_pmc_$c -name '*counter_collection.csv' | head -1) $R/$O/${T}_pmc_$c.csv 2> /dev/null
done
cd $R
LT=$O/${T}_launches.json      # launch table of the counter passes' configuration: the default one unless PMC_ENV changes the kernel mix
if [ -n "$PMC_ENV" ]; then
  ( export $PMC_ENV; step 150 python bench.py $A --steps 5 --warmup 2 --dump-launches $O/${T}_launches_wide0.json > /dev/null 2>&1 )
  LT=$O/${T}_launches_wide0.json
fi
python tools/pmc_traffic.py $O/${T}_pmc_FETCH_SIZE.csv $O/${T}_pmc_WRITE_SIZE.csv $O/${T}_pmc_traffic.json $MD5 $LT > $O/${T}_pmc_traffic.txt 2>&1; tail -12 $O/${T}_pmc_traffic.txt
# 3. rocprofv3 kernel statistics of the bench command
cd /tmp
step 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/${T}_prof -o b -- python $R/bench.py $A --steps 10 --warmup 3 > $R/$O/${T}_bench_under_rocprof.json 2> $R/$O/${T}_rocprof.err
cp $(find /tmp/${T}_prof -name '*kernel_stats.csv' | head -1) $R/$O/${T}_bench_kernel_stats.csv 2> /dev/null
cd $R
echo "kernel stats t=$(( $(date +%s) - T0 ))"
# 4. the GPU tests and smoke()
rm -f $O/parity_bounds.jsonl
step 500 python -m pytest tests -m gpu -x -q >> $O/${T}_pytest_gpu.log 2>&1
cp $O/parity_bounds.jsonl $O/${T}_parity_bounds.jsonl 2> /dev/null      # measured distance of every tightened parity check to its bound
step 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> $O/${T}_pytest_gpu.log 2>&1
tail -3 $O/${T}_pytest_gpu.log; echo "tests t=$(( $(date +%s) - T0 ))"
# 5. the other BASELINE configurations, the SQ pass, the chunk timeline
step 200 python bench.py --height 416 --width 544 --batch 16 --dataset nyu $X > $O/${T}_bench_c2.json 2> /dev/null
step 300 python bench.py --mode infer --height 704 --width 1216 --batch 32 > $O/${T}_bench_infer.json 2> /dev/null
step 300 python bench.py --encoder resnext101_bts --dtype f32 $X > $O/${T}_bench_c4.json 2> /dev/null
for f in c2 c4 infer; do cut -c1-160 $O/${T}_bench_$f.json; done
cd /tmp
( [ -n "$PMC_ENV" ] && export $PMC_ENV; step 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d /tmp/${T}_pmc_sq -o b -- python $R/bench.py $P > $R/$O/${T}_pmc_sq.out 2> $R/$O/${T}_pmc_sq.err )
echo "pmc SQ rc=$?"
cp $(find /tmp/${T}_pmc_sq -name '*counter_collection.csv' | head -1) $R/$O/${T}_pmc_sq.csv 2> /dev/null
cd $R
python tools/pmc_sq.py $O/${T}_pmc_sq.csv $O/${T}_pmc_sq.json $MD5 $LT > $O/${T}_pmc_sq.txt 2>&1; tail -12 $O/${T}_pmc_sq.txt
# chunk timeline of conv_igemm_dma on the diagnostic build (tools/build_trace_lib.sh must have been run before the call: the .so travels)
[ -f bts_amd/lib/libbts_amd_trace.so ] && step 120 python tools/trace_igemm.py > $O/${T}_trace_igemm.jsonl 2> /dev/null
# the counter CSVs are large: keep the summaries, drop the raw files beyond the 64 MiB the call may bring back
gzip -9 -f $O/${T}_pmc_FETCH_SIZE.csv $O/${T}_pmc_WRITE_SIZE.csv $O/${T}_pmc_sq.csv 2> /dev/null
ls -la $O | grep ${T}_ | head -40
du -sh $O
echo "done t=$(( $(date +%s) - T0 ))"
# 6. the N > 1 code path on this one GPU: world size 1 over RCCL (flat parameter broadcast, hook-driven GradAllReducer, BufferSync, eager step) and
#    two self-spawned ranks over gloo sharing the GPU (plumbing evidence, not a scaling number)
step 200 python bench.py --force-dist 1 --backend nccl --steps 10 --warmup 3 --no-cpu-baseline --parity 0 --lpg-op 0 --f32-line 0 > $O/${T}_bench_world1_rccl.json 2> $O/${T}_bench_world1_rccl.err
cut -c1-200 $O/${T}_bench_world1_rccl.json
step 300 python bench.py --gpus 2 --backend gloo --steps 5 --warmup 2 --no-cpu-baseline --parity 0 --lpg-op 0 --f32-line 0 > $O/${T}_bench_selfspawn_2ranks_one_gpu_gloo.json 2> $O/${T}_bench_selfspawn_2ranks_one_gpu_gloo.err
cut -c1-200 $O/${T}_bench_selfspawn_2ranks_one_gpu_gloo.json
echo "dist t=$(( $(date +%s) - T0 ))"
