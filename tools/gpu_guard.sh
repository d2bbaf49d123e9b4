# source me: run_to <seconds> <cmd...> runs the command in its OWN session / process group and SIGKILLs the whole group at the limit,
# so that nothing it spawned (rocprofv3's target, a hung HIP process) can outlive the gpurun call and burn the GPU budget.
run_to() {
    local t=$1; shift
    setsid "$@" &
    local pid=$!
    # the watchdog must not hold the caller's stdout / stderr: in `run_to 900 cmd | tail` its orphaned `sleep` would keep the pipe open
    # (and the gpurun call running, and charged) for the full limit after cmd has long finished -- two calls of round 6 lost 40 min each
    ( sleep "$t"; kill -KILL -- -"$pid" 2>/dev/null ) > /dev/null 2>&1 < /dev/null &
    local w=$!
    wait "$pid"; local rc=$?
    pkill -P "$w" 2>/dev/null; kill "$w" 2>/dev/null; wait "$w" 2>/dev/null
    kill -KILL -- -"$pid" 2>/dev/null      # stragglers of a finished command
    return $rc
}
