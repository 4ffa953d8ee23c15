# Round-6 profile session (run on the GPU box through gpurun; outputs under gpurun_out/$1):
#   cfg-3 as rounds 2 - 5 (kernel stats at the driver's protocol; PMC traffic -- FETCH_SIZE and WRITE_SIZE in SEPARATE runs, no other
#   trace domain beside --kernel-trace -- of the bench workload and of the low-duplication gather probe), kernel stats of cfg-2 and
#   the Tmall default shape, the launch sequences of one step of each (what bench.py's small_shapes legs quote their launch counts
#   from), the evaluation pass at 1 + 99, and the bench lines of record.
# Condensed afterwards in the build container (git is there): tools/summarize_profile.py stats / pmc_round.
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r06p}
mkdir -p $O
cd $R
python3 bench.py --steps 20 --warmup 5 > $O/bench_cfg3_driver_protocol.json 2>$O/err.log
python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline > $O/bench_cfg3_200.json 2>>$O/err.log
python3 tools/eval_bench.py > $O/eval_1p99.json 2>>$O/err.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side > $O/stats.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-side > $O/pmc_fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-side > $O/pmc_write.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/probe_fetch -- python3 $R/bench.py --gather-probe-only > $O/probe_fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/probe_write -- python3 $R/bench.py --gather-probe-only > $O/probe_write.log 2>&1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/probe_stats -- python3 $R/bench.py --gather-probe-only > $O/probe_stats.log 2>&1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg2 -- python3 $R/bench.py --config cfg2 --steps 400 --warmup 50 --no-cpu-baseline --no-side > $O/stats_cfg2.log 2>&1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_tmall_default -- python3 $R/bench.py --config tmall_default --steps 400 --warmup 50 --no-cpu-baseline --no-side > $O/stats_tmall_default.log 2>&1
cd $R
for c in cfg3 cfg2 tmall_default; do
  if [ $c = cfg3 ]; then a=""; else a="--config $c"; fi
  timeout -k 10 300 bash tools/kernel_sequence.sh ${1:-r06p}/seq_$c $a > $O/seq_$c.log 2>&1
done
# keep what travels back small: the per-dispatch CSVs only
find $O -name "*.db" -delete
find $O -name "*_kernel_trace.csv" -size +20M -delete
du -sh $O
