# PMC passes for the bench kernel (separate runs; never combined with other trace domains than kernel-trace)
export TMPDIR=/tmp
OUT=gpurun_out/pmc_r1
mkdir -p $OUT
rocprofv3 -L > $OUT/counters.txt 2>&1 || true
grep -c . $OUT/counters.txt
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/$tag -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/$tag.json 2> $OUT/$tag.err || echo "pass $tag failed"
done
find $OUT -name "*counter_collection.csv" | head -20
