#!/bin/bash
# ab_stamps.sh "variant ..." [bench.py args]: bench.py --stamps once per library build; per call type the phases of the critical workgroup
vars=$1; shift
for v in $vars; do
  if [ "$v" = new ]; then unset TMAC_HIP_LIB; else export TMAC_HIP_LIB=$PWD/tmac_amd/lib/ko/libtmac_hip_$v.so; fi
  python bench.py --no-cpu-baseline --no-decoder-pattern --no-stream-core --stamps "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('%-8s' % '$v', d['ms_per_step'])
for k, p in r.get('per_call_from_stamps', {}).items(): print('   ', k, p)"
done
