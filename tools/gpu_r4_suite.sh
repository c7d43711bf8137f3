# Round 4: the whole GPU suite, then the differential fuzz sweep over 480 seeds with fuzz_diag figures for every miss.
export TMPDIR=/tmp
tag=${1:-r4s}
OUT=gpurun_out/$tag; mkdir -p $OUT
export AIRCRAFT_PARITY_REPORT=$PWD/$OUT/parity_report.jsonl
rm -f $AIRCRAFT_PARITY_REPORT
timeout -k 10 1000 python3 -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -x > $OUT/pytest_gpu.log 2>&1; rc=$?
tail -15 $OUT/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 700 python3 tools/fuzz_many.py 12 492 > $OUT/fuzz_480.txt 2>&1
grep -c "^FAIL" $OUT/fuzz_480.txt; tail -2 $OUT/fuzz_480.txt | cut -c1-600
seeds=$(grep "^FAIL" $OUT/fuzz_480.txt | awk '{print $2}' | tr '\n' ' ')
[ -n "$seeds" ] && timeout -k 10 300 python3 tools/fuzz_diag.py $seeds > $OUT/fuzz_diag.txt 2>&1; grep -v amdgpu $OUT/fuzz_diag.txt | tail -40
