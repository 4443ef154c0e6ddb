# A/B of the cache policy of the fire-and-forget tape / d(theta) stores (FENERF_ST_POLICY in fenerf_siren_f16w.hip / fenerf_siren_bwd16w.hip):
#   libexp_stpol1 "sc1 nt", 2 "sc0 sc1 nt", 3 "sc0 sc1", 4 "sc0 nt"; shipped = "nt".  Same box, interleaved, shipped first and last of each round.
for rep in 1 2; do
for v in "" stpol1 stpol2 stpol3 stpol4 ""; do
  if [ -z "$v" ]; then lib=$PWD/fenerf_amd/libfenerf_hip.so; else lib=$PWD/fenerf_amd/libexp_$v.so; fi
  [ -f $lib ] || continue
  echo -n "${v:-shipped}: "
  FENERF_LIB=$lib timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32 --no-sweep64 --no-gstep-ddp --no-gstep-b6 > /dev/null 2>&1
  python - <<PY
import json
g = json.load(open("bench_detail.json"))["gstep"]
print("gstep %.3f ms" % g["ms"], {k["name"]: round(k["ms"], 3) for k in g["roofline"]["per_kernel"]})
PY
done; done
