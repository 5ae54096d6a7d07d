set -x
cd $GRAFT_REPO_ROOT
for wl in cfg3 cfg4 cfg5; do
PXL_BENCH_SHARE_GPU=1 PXL_BENCH_POINTS=2e8 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29631 bench.py --gpus 2 --steps 3 --warmup 1 --backend gloo --workload $wl > gpurun_out/r02_rehearsal2_$wl.json 2> gpurun_out/r02_rehearsal2_$wl.err; echo "$wl rc=$?"; cut -c1-700 gpurun_out/r02_rehearsal2_$wl.json; tail -2 gpurun_out/r02_rehearsal2_$wl.err | cut -c1-300
done
