#!/bin/bash
# Record MIOpen's find results for the stock encoder at every BASELINE configuration (channels-last / NHWC) into
# gpurun_out/<tag>_miopen_db (copy the *.ufdb.txt into bts_amd/miopen_db/ afterwards), then re-run the default line in find mode
# `fast` on the recorded db to show what a fresh box gets from the shipped files.
#   bash tools/miopen_warm.sh <tag> [deadline seconds]
. tools/gpu_guard.sh
T=${1:-warm}
DEADLINE=${2:-3000}
T0=$(date +%s)
O=gpurun_out
DB=$PWD/$O/${T}_miopen_db
mkdir -p $DB
export TMPDIR=/tmp
cp bts_amd/miopen_db/*.ufdb.txt $DB/ 2> /dev/null
A="--no-cpu-baseline --parity 0 --lpg-op 0 --f32-line 0 --eager-steps 0 --steps 5 --warmup 2 --no-kernel-events --miopen-db $DB"
step() {
    local lim=$1; shift
    local left=$(( DEADLINE - ( $(date +%s) - T0 ) ))
    if [ $left -lt 60 ]; then echo "SKIPPED (deadline): $*" | cut -c1-120; return 99; fi
    [ $lim -gt $left ] && lim=$left
    run_to $lim "$@"
}
warm() {   # warm <name> <limit> <args...>
    local n=$1 lim=$2; shift 2
    local t0=$(date +%s)
    step $lim python bench.py $A --miopen-find-mode normal "$@" > $O/${T}_warm_$n.json 2> $O/${T}_warm_$n.err
    echo "warm $n rc=$? $(( $(date +%s) - t0 ))s: $(cut -c1-110 $O/${T}_warm_$n.json) | db lines: $(cat $DB/*.ufdb.txt 2> /dev/null | wc -l)"
}
warm c3_bf16 900
warm c3_f32 900 --dtype f32
warm c5_infer 600 --mode infer --height 704 --width 1216 --batch 32
warm c2_bf16 900 --height 416 --width 544 --batch 16 --dataset nyu
warm c4_f32 1000 --encoder resnext101_bts --dtype f32
ls -la $DB
# what a fresh box gets from the recorded db: find mode fast, nothing searched
t0=$(date +%s)
step 300 python bench.py --no-cpu-baseline --parity 0 --lpg-op 0 --f32-line 0 --eager-steps 5 --steps 20 --warmup 5 --miopen-db $DB > $O/${T}_fast_c3_bf16.json 2> $O/${T}_fast_c3_bf16.err
echo "fast c3_bf16 $(( $(date +%s) - t0 ))s: $(cut -c1-140 $O/${T}_fast_c3_bf16.json)"
t0=$(date +%s)
step 300 python bench.py --no-cpu-baseline --parity 0 --lpg-op 0 --f32-line 0 --eager-steps 5 --steps 20 --warmup 5 --miopen-db none --miopen-find-mode fast > $O/${T}_fast_nodb_c3_bf16.json 2> $O/${T}_fast_nodb_c3_bf16.err
echo "fast, no db c3_bf16 $(( $(date +%s) - t0 ))s: $(cut -c1-140 $O/${T}_fast_nodb_c3_bf16.json)"
du -sh $DB ~/.cache/miopen 2> /dev/null
echo "done t=$(( $(date +%s) - T0 ))"
