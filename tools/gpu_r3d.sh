set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3d
timeout -k 10 900 python -u -m pytest tests/test_gpu_kernels.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -5
for st in 0 1; do
VITK_EPI_STREAM=$st timeout -k 10 300 python tools/bench_gemm.py 256 "fc2 dglu" 2>&1 | grep -v amdgpu.ids
VITK_EPI_STREAM=$st timeout -k 10 300 python tools/bench_gemm.py 256 "fc2 fwd" 2>&1 | grep -v amdgpu.ids
VITK_EPI_STREAM=$st timeout -k 10 300 python tools/bench_gemm.py 256 "out fwd" 2>&1 | grep -v amdgpu.ids
done
