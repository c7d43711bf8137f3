# Same-box A/B of bench_modes cases between the product library and flavours:  tools/ab_modes.sh "<cases>" <tag> [<tag> ...]
# (interleaved A B A B: clocks drift between boxes by a few per cent, much less inside one call)
export TMPDIR=/tmp
cases=$1; shift
for rep in 1 2; do
  for t in "" "$@"; do
    s=${t:+_$t}
    echo "== ${t:-product} (pass $rep)"
    AIRCRAFT_HIP_LIB=$PWD/aircraft_amd/libaircraft_hip$s.so python3 tools/bench_modes.py $cases 2>/dev/null | python3 -c "
import json, sys
for l in sys.stdin:
    d = json.loads(l); print(f\"{d['case']:34s} sens {d['sens_ms']*1e3:8.1f} us  {d['sens_steps_per_s']:.3e}/s\")"
  done
done
