#!/bin/bash
# round 4: the streaming side of a pipeline with the search service against the batch-granular pipeline's when there is nothing to
# search (no marker stage): residency of the service, number of streams, prologue stream, priority; then the full workload by depth.
# (knobs: tools/svc_rate.py's own RATE_* variables -> smhv_pipeline_options)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04k
export TMPDIR=/tmp
R="python tools/svc_rate.py"
run() { name=$1; shift; timeout -s KILL 150 env SVC_RATE_STAGE_MS=1 "$@" > gpurun_out/r04k/$name.json 2> gpurun_out/r04k/$name.err; echo "$name rc=$?"; tail -1 gpurun_out/r04k/$name.json | cut -c1-1500; grep "watchdog\|slow submit\|Error\|error" gpurun_out/r04k/$name.err | head -4 | cut -c1-400; }
run nm_d12 $R 256 12 400 0xE
run nm_d12_wgs1 RATE_WGS=1 $R 256 12 400 0xE
run nm_d12_wgs1_ns4 RATE_WGS=1 RATE_STREAMS=4 RATE_FLAGS=4 $R 256 12 400 0xE
run nm_d12_ns4_nopro RATE_STREAMS=4 RATE_FLAGS=4 $R 256 12 400 0xE
run nm_d12_ns3_nopro RATE_STREAMS=3 RATE_FLAGS=4 $R 256 12 400 0xE
run nm_d12_noprio RATE_FLAGS=2 $R 256 12 400 0xE
run nm_batch_d4 RATE_SEARCH=batch $R 256 4 400 0xE
run nm_batch_d4_nopolicy RATE_SEARCH=batch RATE_POLICY=2 $R 256 4 400 0xE
run full_d12 $R 256 12 400
run full_d12_ns3_nopro RATE_STREAMS=3 RATE_FLAGS=4 $R 256 12 400
run full_d12_ns4_nopro RATE_STREAMS=4 RATE_FLAGS=4 $R 256 12 400
run full_d8 $R 256 8 400
run full_d4_frame RATE_SEARCH=frame $R 256 4 400
run full_batch_d4 RATE_SEARCH=batch $R 256 4 400
