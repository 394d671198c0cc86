set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3c
for m in store glu dglu res out fc2; do
VITK_LIB=exp/libvitk_stamp.so timeout -k 10 120 python tools/stamp_gemm.py $m 2>&1 | grep -v amdgpu.ids >> gpurun_out/r3c/stamps.txt
done
cat gpurun_out/r3c/stamps.txt
