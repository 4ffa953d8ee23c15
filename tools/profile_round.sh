set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r01b
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/pmc_write.log 2>&1
cd $R
python3 bench.py --no-cpu-baseline --all-rows-live > $O/bench_all_rows_live.json 2>$O/err.log
python3 bench.py --no-cpu-baseline --force-sharded > $O/bench_sharded_1rank.json 2>>$O/err.log
python3 bench.py --no-cpu-baseline --config cfg2 > $O/bench_cfg2.json 2>>$O/err.log
python3 bench.py --no-cpu-baseline --config cfg5_taobao --steps 6 --warmup 2 --batches 2 > $O/bench_cfg5_taobao.json 2>>$O/err.log
python3 bench.py --no-cpu-baseline --no-skip-masked > $O/bench_no_skip_masked.json 2>>$O/err.log
python3 tools/eval_bench.py > $O/eval_1p99.json 2>>$O/err.log
ls $O
