cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python bench.py --force-dist --eager-collectives --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/dist28_eager.json 2> gpurun_out/dist28_eager.err
timeout 600 python bench.py --force-dist --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/dist28_graph.json 2> gpurun_out/dist28_graph.err
cut -c1-330 gpurun_out/dist28_eager.json; tail -3 gpurun_out/dist28_eager.err; cut -c1-330 gpurun_out/dist28_graph.json; tail -5 gpurun_out/dist28_graph.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --force-dist --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/dist28_torchrun.json 2> gpurun_out/dist28_torchrun.err
cut -c1-200 gpurun_out/dist28_torchrun.json; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids\|socket.cpp\|^$" gpurun_out/dist28_torchrun.err | tail -5
