cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests/test_gpu_chain_ipc.py -q -m gpu -x > gpurun_out/r3/ipc_tests.log 2>&1
tail -4 gpurun_out/r3/ipc_tests.log | cut -c1-300
timeout 300 python bench.py --force-dist --no-cpu-baseline --no-verify > gpurun_out/r3/bench_forcedist.json 2> gpurun_out/r3/bench_forcedist.err; tail -3 gpurun_out/r3/bench_forcedist.err
python -c "import json; d=json.load(open('gpurun_out/r3/bench_forcedist.json')); print(d['ms_per_step'], d['config']['path'], d['config']['launch'], d['n_gpus'])"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --no-cpu-baseline --no-verify > gpurun_out/r3/bench_torchrun1.json 2> gpurun_out/r3/bench_torchrun1.err; tail -2 gpurun_out/r3/bench_torchrun1.err
python -c "import json; d=json.load(open('gpurun_out/r3/bench_torchrun1.json')); print(d['ms_per_step'], d['config']['path'], d['n_gpus'])"
