for p in 0 1 2 3; do echo "== IPX_FI_PRIO=$p"; IPX_FI_PRIO=$p timeout -k 10 100 python tools/bench_stage.py flow 257 3 2>&1 | grep -v amdgpu.ids | head -5 || exit 1; done
