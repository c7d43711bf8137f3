# same-box comparison of the cfg2 (3x64, MFMA) sensitivity step across library builds: tools/ab_cfg2.sh <tag> ...
for i in 1 2 3; do for t in "$@"; do
  s=${t:+_$t}
  for B in 256 4096; do
  AIRCRAFT_HIP_LIB=$PWD/aircraft_amd/libaircraft_hip$s.so python bench.py --hidden 64,64,64 --batch $B --steps 50 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null \
   | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('${t:-current}', $B, d['value'], d['ms_per_step'])"
  done
done; done
