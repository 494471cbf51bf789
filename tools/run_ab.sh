set -e
cd /root/repo
for i in 1 2; do
echo "== prev"; IPX_LIB=/root/repo/tools/_exp/libipx_prev.so timeout -k 10 120 python tools/bench_stage.py flow 257 3
echo "== new"; timeout -k 10 120 python tools/bench_stage.py flow 257 3
done
timeout -k 10 120 python tools/bench_stage.py detect 1000 3
timeout -k 10 600 python -u -m pytest tests/test_gpu_chunk.py tests/test_gpu_stages.py -x -q -m gpu --timeout 300 --timeout-method thread
