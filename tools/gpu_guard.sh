# source me: run_to <seconds> <cmd...> runs the command in its OWN session / process group and SIGKILLs the whole group at the limit,
# so that nothing it spawned (rocprofv3's target, a hung HIP process) can outlive the gpurun call and burn the GPU budget.
run_to() {
    local t=$1; shift
    setsid "$@" &
    local pid=$!
    ( sleep "$t"; kill -KILL -- -"$pid" 2>/dev/null ) &
    local w=$!
    wait "$pid"; local rc=$?
    kill "$w" 2>/dev/null; wait "$w" 2>/dev/null
    kill -KILL -- -"$pid" 2>/dev/null      # stragglers of a finished command
    return $rc
}
