cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
for v in 0 1; do
echo "== HIP_FORCE_DEV_KERNARG=$v" >> gpurun_out/kernarg19.txt
HIP_FORCE_DEV_KERNARG=$v timeout 300 python tools/tune_quad.py 2>&1 | cut -c1-60 >> gpurun_out/kernarg19.txt
HIP_FORCE_DEV_KERNARG=$v timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | cut -c1-200 >> gpurun_out/kernarg19.txt
done
HIP_FORCE_DEV_KERNARG=1 timeout 600 python tools/stamps.py 2>&1 | grep -A4 "== down" >> gpurun_out/kernarg19.txt
cat gpurun_out/kernarg19.txt
