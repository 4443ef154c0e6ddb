for v in "" ${VARIANTS:-} ""; do
  if [ -z "$v" ]; then lib=$PWD/fenerf_amd/libfenerf_hip.so; else lib=$PWD/fenerf_amd/libexp_$v.so; fi
  echo -n "${v:-shipped}: "
  FENERF_LIB=$lib python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-gstep --no-f32 --no-sweep64 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; print('ms/step %.4f kernel_ms %.4f cycles %.0f clock %.3f' % (j['ms_per_step'], r['kernel_ms'], r['cycles_per_launch'], r['clock_ghz_effective']))"
done
