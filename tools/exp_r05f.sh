T=$1
for rep in 1 2 3; do
for lib in libsmh_vision_hip.so libsmh_vision_hip_w4g16.so libsmh_vision_hip_w3g16.so; do
  export SMH_VISION_HIP_LIB=$PWD/squad-mortar-helper_amd/$lib
  RATE_SEARCH=frame python tools/svc_rate.py 256 12 1000 2>&1 | tail -1 > gpurun_out/${T}_syn_${lib}_$rep.json
done
done
for rep in 1 2; do
for lib in libsmh_vision_hip.so libsmh_vision_hip_w4g16.so; do
  export SMH_VISION_HIP_LIB=$PWD/squad-mortar-helper_amd/$lib
  RATE_SEARCH=frame python tools/svc_rate.py 128 12 800 15 2560 1440 2>&1 | tail -1 > gpurun_out/${T}_syn1440_${lib}_$rep.json
  RATE_SEARCH=frame SVC_RATE_DISTINCT=256 python tools/svc_rate.py 1024 8 200 2>&1 | tail -1 > gpurun_out/${T}_syn1024_${lib}_$rep.json
done
done
