# same-box comparison of the forward kernels (forward shooting step, rollout) across library builds: tools/ab_fwd.sh <tag> ...
for i in 1 2 3; do for t in "$@"; do
  s=${t:+_$t}
  AIRCRAFT_HIP_LIB=$PWD/aircraft_amd/libaircraft_hip$s.so python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null \
   | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('${t:-current}', d['ms_per_step'], d['alongside']['forward_shooting_ms'], d['alongside']['rollout_ms'])"
done; done
