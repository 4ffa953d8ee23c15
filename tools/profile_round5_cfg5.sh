# kernel stats + PMC traffic of the cfg-5 Taobao-shaped step on one GPU (refresh the profile)
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r05c5}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --config cfg5_taobao --steps 8 --warmup 3 --batches 2 --no-cpu-baseline --no-side > $O/stats.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --config cfg5_taobao --steps 4 --warmup 2 --batches 2 --no-cpu-baseline --no-side > $O/pmc_fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --config cfg5_taobao --steps 4 --warmup 2 --batches 2 --no-cpu-baseline --no-side > $O/pmc_write.log 2>&1
find $O -name "*.db" -delete
find $O -name "*_kernel_trace.csv" -size +20M -delete
du -sh $O
