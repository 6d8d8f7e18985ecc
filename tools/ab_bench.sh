#!/bin/bash
# ab_bench.sh "variant ..." [bench.py args]: bench.py once per library build (tmac_amd/lib/ko/libtmac_hip_<variant>.so; "new" = the tree's own build),
# one line per run: variant, ms_per_step, frac, stream_core us per GEMV, verified.  Interleave variants by naming them repeatedly.
vars=$1; shift
for v in $vars; do
  if [ "$v" = new ]; then unset TMAC_HIP_LIB; else export TMAC_HIP_LIB=$PWD/tmac_amd/lib/ko/libtmac_hip_$v.so; fi
  python bench.py --no-cpu-baseline --no-decoder-pattern "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('%-8s' % '$v', d['ms_per_step'], r['frac'], r.get('stream_core', {}).get('us_per_gemv'), (d.get('verified') or {}).get('ok'), (d.get('verified') or {}).get('max_rel_err'))"
done
