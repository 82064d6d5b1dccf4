#!/bin/bash
# blocking lone-query latency under the runtime's wait policies (deployment knobs, no code change)
O=gpurun_out/r03/lone_env
mkdir -p $O
echo "== default" | tee $O/log.txt
timeout -k 10 200 python tools/probes/lone_query_latency.py 2>&1 | tee -a $O/log.txt
echo "== HSA_ENABLE_INTERRUPT=0" | tee -a $O/log.txt
HSA_ENABLE_INTERRUPT=0 timeout -k 10 200 python tools/probes/lone_query_latency.py 2>&1 | tee -a $O/log.txt
echo "== HIP_FORCE_SPIN... GPU_MAX_HW_QUEUES=1" | tee -a $O/log.txt
GPU_MAX_HW_QUEUES=1 timeout -k 10 200 python tools/probes/lone_query_latency.py 2>&1 | tee -a $O/log.txt
