cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out
BARGS="--steps 1 --warmup 0 --timesteps 200 --no-cpu-baseline --north-star-batch 0 --no-extra-shapes"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $o/pmc_fetch -- python3 bench.py $BARGS > /dev/null 2> $o/pmc_fetch.err; echo rc=$?
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $o/pmc_write -- python3 bench.py $BARGS > /dev/null 2> $o/pmc_write.err; echo rc=$?
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_LDS --output-format csv -d $o/pmc_sq -- python3 bench.py $BARGS > /dev/null 2> $o/pmc_sq.err; echo rc=$?
python3 tools/collect_traffic.py $o/pmc_fetch $o/pmc_write --sq $o/pmc_sq --out $o/kernel_traffic.json --command "python3 bench.py $BARGS" > $o/${TAG:-r03_z}_pmc_summary.json; echo rc=$?
tail -3 $o/pmc_fetch.err
rm -rf $o/pmc_fetch $o/pmc_write $o/pmc_sq
# rocprofv3 kernel stats of the headline command alone (no other record of the default line)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/kstats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --north-star-batch 0 --no-extra-shapes > /dev/null 2>&1; echo rc=$?
cp $(find $o/kstats -name "*kernel_stats.csv" | head -1) $o/${TAG:-r03_z}_kernel_stats_b64_T1000.csv; rm -rf $o/kstats
