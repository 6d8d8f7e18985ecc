cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python bench.py --force-dist --eager-collectives --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/dist_eager.json 2> gpurun_out/dist_eager.err
cut -c1-200 gpurun_out/dist_eager.json; echo
# the captured-collective path, several times: capture races with ProcessGroupNCCL's watchdog thread were seen here
for i in 1 2 3 4 5; do
  timeout 600 python bench.py --force-dist --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/dist_graph$i.json 2> gpurun_out/dist_graph$i.err
  echo "run $i rc=$? $(cut -c1-200 gpurun_out/dist_graph$i.json)"; grep -c "capture with collectives failed" gpurun_out/dist_graph$i.err
done
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --force-dist --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/dist_torchrun.json 2> gpurun_out/dist_torchrun.err
echo "torchrun rc=$? $(head -c 200 gpurun_out/dist_torchrun.json; echo; wc -l gpurun_out/dist_torchrun.json)"; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids\|socket.cpp\|^$" gpurun_out/dist_torchrun.err | tail -5
