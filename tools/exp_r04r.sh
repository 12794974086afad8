#!/bin/bash
# round 4: the reference's screenshots, warm and long runs: batch-granular against frame-granular search by depth
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04r
export TMPDIR=/tmp
run() { name=$1; shift; timeout -s KILL 200 env "$@" 2>&1 | grep -v amdgpu.ids | tail -4 > gpurun_out/r04r/$name.txt; echo "$name rc=$?"; grep "^GPU" gpurun_out/r04r/$name.txt; tail -1 gpurun_out/r04r/$name.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ', d['config'].get('search_service'))"; }
for D in 1 2 4 8 12 16; do run batch_d$D SAMPLES_SEARCH=batch python tools/bench_samples.py 128 $D; done
for D in 4 8 12 16; do run frame_d$D SAMPLES_SEARCH=frame python tools/bench_samples.py 128 $D; done
run batch256_d8 SAMPLES_SEARCH=batch python tools/bench_samples.py 256 8
run frame256_d8 SAMPLES_SEARCH=frame python tools/bench_samples.py 256 8
run full_batch_d4 SAMPLES_SEARCH=batch SMH_BENCH_STAGES=0xF python tools/bench_samples.py 128 4
run full_frame_d12 SAMPLES_SEARCH=frame SMH_BENCH_STAGES=0xF python tools/bench_samples.py 128 12
