set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -u -m pytest tests/test_gpu_kernels.py -x -q --timeout 120 --timeout-method thread -k "attention or attn" 2>&1 | tail -3
for lib in exp/libvitk_base.so vit_amd/libvitk.so; do
echo "== $lib"
VITK_LIB=$lib timeout -k 10 120 python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids
VITK_LIB=$lib timeout -k 10 120 python tools/bench_attn.py 8 5329 16 88 2>&1 | grep -v amdgpu.ids
VITK_LIB=$lib timeout -k 10 120 python tools/bench_attn.py 64 256 16 80 2>&1 | grep -v amdgpu.ids
VITK_LIB=$lib timeout -k 10 120 python tools/bench_attn.py 64 576 16 64 2>&1 | grep -v amdgpu.ids
done
