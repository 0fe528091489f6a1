O=$(pwd)/gpurun_out/r03p4; mkdir -p $O; R=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU --kernel-include-regex prefill_attn -f csv -d /tmp/pmc_pf1 -o pf -- python $R/tools/prefill_bench.py > $O/prefill_under_pmc1.json 2> $O/pmc1.err; echo "pmc1 rc=$?"
f=$(find /tmp/pmc_pf1 -name '*counter_collection.csv' | head -1); python $R/tools/pmc_group_summary.py $f prefill_attn $O/prefill_pmc_waits.json | tail -120
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_IDX_ACTIVE --kernel-include-regex prefill_attn -f csv -d /tmp/pmc_pf2 -o pf -- python $R/tools/prefill_bench.py > $O/prefill_under_pmc2.json 2> $O/pmc2.err; echo "pmc2 rc=$?"
f=$(find /tmp/pmc_pf2 -name '*counter_collection.csv' | head -1); python $R/tools/pmc_group_summary.py $f prefill_attn $O/prefill_pmc_active.json | grep -v '"SQ_\|counters\|}' | head -80
