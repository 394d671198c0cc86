set -e
cd $GRAFT_REPO_ROOT
for lib in exp/libvitk_base.so vit_amd/libvitk.so exp/libvitk_silu.so; do
echo "== $lib"
VITK_LIB=$lib timeout -k 10 300 python tools/bench_gemm.py 256 "fc2 dglu" 2>&1 | grep -v "amdgpu.ids\|^sum"
VITK_LIB=$lib VITK_EPI_STREAM=0 timeout -k 10 300 python tools/bench_gemm.py 256 "fc2 dglu" 2>&1 | grep -v "amdgpu.ids\|^sum"
VITK_LIB=$lib timeout -k 10 300 python tools/bench_gemm.py 256 "fc1 glu" 2>&1 | grep -v "amdgpu.ids\|^sum"
done
