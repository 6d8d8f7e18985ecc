python -m pytest tests/test_gpu_stream.py -x -q 2>&1 | tail -5
for n in 1 8; do for t in 96; do echo "== NCLS=$n target=$t"; TMAC_STREAM_NCLS=$n TMAC_STREAM_VISIT_ITEMS=$t python tools/bench_stream.py 2>&1 | grep -v "Warn\|amdgpu.ids"; done; done
for n in 4 16; do echo "== NCLS=$n target=96"; TMAC_STREAM_NCLS=$n python tools/bench_stream.py 2>&1 | grep -v "Warn\|amdgpu.ids"; done
for t in 48 192 384; do echo "== NCLS=16 target=$t"; TMAC_STREAM_NCLS=16 TMAC_STREAM_VISIT_ITEMS=$t python tools/bench_stream.py 2>&1 | grep -v "Warn\|amdgpu.ids"; done
