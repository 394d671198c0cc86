set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -u -m pytest tests/test_gpu_model.py -x -q --timeout 120 --timeout-method thread -k "handed or written_through" 
timeout -k 10 300 python tools/rehearse_ddp_one_rank.py --json gpurun_out/ddp_b14.json
timeout -k 10 300 python tools/rehearse_ddp_one_rank.py --workload vit_h14_224_droppath_ls_ckpt --json gpurun_out/ddp_h14.json
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_k.json
tail -c 1500 gpurun_out/bench_k.json
