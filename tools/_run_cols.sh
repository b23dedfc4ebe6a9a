set -e
: > gpurun_out/cols.log
run() { echo "== $*" >> gpurun_out/cols.log; env "$@" CG_ENGINE=1 timeout -k 10 100 python tools/engine_compare.py 4,128 64x64x64x512 4,64 4,96 5,48 64x64x64x64x16 5,32 101x101x101x101 2>&1 | grep -v amdgpu.ids >> gpurun_out/cols.log; }
run CG_X=0
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_slab.py -x -q --timeout 120 --timeout-method thread > gpurun_out/cols_t.log 2>&1
CG_COLS_LDS_A=1 timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_slab.py -x -q --timeout 120 --timeout-method thread -k "blocked or random or slab or virtual" > gpurun_out/cols_t2.log 2>&1
