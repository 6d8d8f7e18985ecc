cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
for n in 64 1024; do timeout 900 python tools/bench_prefill.py $n > gpurun_out/prefill_n$n.txt 2>&1; tail -6 gpurun_out/prefill_n$n.txt; done
