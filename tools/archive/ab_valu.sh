# same-box comparison of the cfg2 "MFMA off" sensitivity step (3x64, use_mfma = 0) across library builds:
#   tools/ab_valu.sh <tag> ...      ("" = the product library, <tag> = aircraft_amd/libaircraft_hip_<tag>.so)
for i in 1 2 3; do for t in "$@"; do
  s=${t:+_$t}
  for B in 256 4096; do
  AIRCRAFT_HIP_LIB=$PWD/aircraft_amd/libaircraft_hip$s.so python bench.py --no-mfma --hidden 64,64,64 --batch $B --steps 50 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null \
   | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('${t:-current}', $B, '%.4g steps/s' % d['value'], '%.4f ms' % d['ms_per_step'], 'frac %.3f' % d['roofline']['frac'], d['roofline']['kernel'])"
  done
done; done
