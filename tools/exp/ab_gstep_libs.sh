for v in "" ${VARIANTS:-} ""; do
  if [ -z "$v" ]; then lib=$PWD/fenerf_amd/libfenerf_hip.so; else lib=$PWD/fenerf_amd/libexp_$v.so; fi
  echo -n "${v:-shipped}: "
  FENERF_LIB=$lib python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32 --no-sweep64 --no-gstep-b6 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); g=j['gstep']; r=g['roofline']
        print('gstep %.3f' % g['ms'], ' '.join('%s %.3f' % (k['name'], k['ms']) for k in r['per_kernel']), '| amp %.3f' % j['gstep_amp']['ms'])"
done
