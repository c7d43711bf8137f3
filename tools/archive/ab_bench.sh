# Same-box comparison of the headline across library builds: tools/ab_bench.sh <tag> <tag> ...  ("" = libaircraft_hip.so)
for i in 1 2 3; do for t in "$@"; do
  s=${t:+_$t}
  AIRCRAFT_HIP_LIB=$PWD/aircraft_amd/libaircraft_hip$s.so python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null \
   | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('${t:-current}', d['value'], d['ms_per_step'])"
done; done
