#!/bin/bash
# Round 6: the line-search fuzz over frame shapes no test had run before (ultrawide, 16:10, 4:3, odd sizes), through the batch path
# (k_lsd_tile) and through the frame-granular service -- run ON THE GPU BOX.  usage: tools/fuzz_shapes_r06.sh [iterations=2] [frames=16] [seed=9100]
IT=${1:-2}; N=${2:-16}; SEED=${3:-9100}; bad=0
for s in 3440x1440 5120x1440 2560x1080 3840x1600 1920x1200 2560x1600 1680x1050 1440x900 1280x720 1366x768 4096x2160 2048x1152 3200x1800 2880x1620 1600x900 7680x4320; do
  for svc in 0 1; do
    [ "$s" = "7680x4320" ] && [ $svc = 1 ] && continue            # (no service at that size: the pipeline keeps the batch-granular search)
    if [ $svc = 1 ]; then out=$(FUZZ_SIZE=$s FUZZ_SERVICE=1 timeout 600 python tools/fuzz_lsd.py $IT $N $SEED 2>&1 | tail -1); else out=$(FUZZ_SIZE=$s timeout 600 python tools/fuzz_lsd.py $IT $N $SEED 2>&1 | tail -1); fi
    echo "$s service=$svc: $out"
    case "$out" in *"FUZZ OK"*) ;; *) bad=1;; esac
  done
done
rm -f gpurun_out/fuzz_bad_*.npy
[ $bad = 0 ] && echo "ALL SHAPES OK" || echo "SOME SHAPE FAILED"
