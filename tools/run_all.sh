set -e
cd /root/repo
timeout -k 10 1000 python -u -m pytest tests -x -q -m gpu --timeout 600 --timeout-method thread > gpurun_out/gpu_all5.log 2>&1
echo "tests done"; tail -3 gpurun_out/gpu_all5.log
timeout -k 10 300 python bench.py > gpurun_out/r02f_bench.json 2> gpurun_out/r02f_bench.err
echo "bench done"
bash tools/profile.sh r02f
