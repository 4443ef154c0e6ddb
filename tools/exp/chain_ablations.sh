# Timing ablations of the chain kernel (FENERF_EXP_CHAIN_ABLATE, fenerf_siren_bwd16w.hip; the variants compute WRONG gradients):
#   for v in 1 2 4 7; do hipcc ... -DFENERF_EXP_CHAIN_ABLATE=$v -c fenerf_siren_bwd16w_w.hip -o build_exp/w_$v.o; link as fenerf_amd/libexp_chain_ablate$v.so; done
# then on the GPU box: bash tools/exp/chain_ablations.sh   -> chain / step times per variant, shipped library first and last
for v in "" 1 2 4 7 ""; do
  if [ -z "$v" ]; then lib=$PWD/fenerf_amd/libfenerf_hip.so; else lib=$PWD/fenerf_amd/libexp_chain_ablate$v.so; fi
  [ -f $lib ] || continue
  echo -n "${v:-shipped}: "
  FENERF_LIB=$lib timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32 --no-sweep64 --no-gstep-ddp --no-gstep-b6 > /dev/null 2>&1
  python - <<PY
import json
g = json.load(open("bench_detail.json"))["gstep"]
print("gstep %.3f ms" % g["ms"], {k["name"]: round(k["ms"], 3) for k in g["roofline"]["per_kernel"]})
PY
done
