set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_sharded_native.py tests/test_gpu_entrypoints.py -x -q 2>&1 | tail -5
python bench.py --steps 10 --warmup 2 > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err; echo "rc=$?"; cat gpurun_out/r02_bench_default.json
PXL_BENCH_SHARE_GPU=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 3 --warmup 1 --backend gloo --workload cfg3 > gpurun_out/r02_rehearsal_gloo.json 2> gpurun_out/r02_rehearsal_gloo.err; echo "gloo rehearsal rc=$?"; cat gpurun_out/r02_rehearsal_gloo.json | cut -c1-600
PXL_BENCH_SHARE_GPU=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --steps 3 --warmup 1 --backend nccl --workload cfg3 > gpurun_out/r02_rehearsal_nccl.json 2> gpurun_out/r02_rehearsal_nccl.err; echo "nccl-on-one-gpu rehearsal rc=$? (non-zero expected: RCCL refuses two ranks on one device, and gloo is not a fallback)"; grep "halo transport\|no halo" gpurun_out/r02_rehearsal_nccl.err | head -8
