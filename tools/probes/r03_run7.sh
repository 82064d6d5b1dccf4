set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 1000 python -m pytest tests -m gpu -x -q -k "not config_t and not config_c3_full and not config_c4 and not config_c5 and not soak" > gpurun_out/r03/gputests7.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -4 gpurun_out/r03/gputests7.log
