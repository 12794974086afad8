#!/bin/bash
# round 4: the frame-granular search at depth 1 and 2 (one batch: the search of a frame overlaps the pass over the next frames)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04o
export TMPDIR=/tmp
R="python tools/svc_rate.py"
run() { name=$1; shift; timeout -s KILL 150 env SVC_RATE_STAGE_MS=1 "$@" > gpurun_out/r04o/$name.json 2> gpurun_out/r04o/$name.err; echo "$name rc=$?"; tail -1 gpurun_out/r04o/$name.json | cut -c1-1500; grep "watchdog\|slow submit\|Error\|error" gpurun_out/r04o/$name.err | head -4 | cut -c1-400; }
run d1_frame RATE_SEARCH=frame $R 256 1 300
run d1_frame_nopush RATE_SEARCH=frame RATE_FLAGS=8 $R 256 1 300
run d1_batch RATE_SEARCH=batch $R 256 1 300
run d2_frame RATE_SEARCH=frame $R 256 2 300
run d2_frame_nopush RATE_SEARCH=frame RATE_FLAGS=8 $R 256 2 300
run d2_batch RATE_SEARCH=batch $R 256 2 300
run d3_frame RATE_SEARCH=frame $R 256 3 300
run d3_batch RATE_SEARCH=batch $R 256 3 300
run d1_frame_1024 RATE_SEARCH=frame $R 1024 1 80
run d1_batch_1024 RATE_SEARCH=batch $R 1024 1 80
run d2_frame_1024 RATE_SEARCH=frame $R 1024 2 80
run d2_batch_1024 RATE_SEARCH=batch $R 1024 2 80
run c3_d1_frame RATE_SEARCH=frame $R 128 1 300 0xF 2560 1440
run c3_d1_batch RATE_SEARCH=batch $R 128 1 300 0xF 2560 1440
