set -e
mkdir -p gpurun_out/r5
O=gpurun_out/r5
timeout -k 10 500 python -m pytest tests/test_gpu_bench_contract.py tests/test_gpu_dist.py tests/test_gpu_bad_ids.py tests/test_gpu_harness.py -x -q > $O/a_tests.log 2>&1 && tail -3 $O/a_tests.log
for gb in 128 256 512; do
  timeout -k 10 200 python bench.py --config cfg3 --force-sharded --global-batch $gb --steps 60 --warmup 10 --no-cpu-baseline --no-side > $O/a_cfg4rank_fs_b$gb.json 2> $O/a_cfg4rank_fs_b$gb.err && tail -c 2500 $O/a_cfg4rank_fs_b$gb.json
done
timeout -k 10 300 python bench.py --config cfg5_tmall --force-sharded --global-batch 512 --steps 40 --warmup 10 --no-cpu-baseline --no-side > $O/a_cfg5_fs_b512.json 2> $O/a_cfg5_fs_b512.err && tail -c 2500 $O/a_cfg5_fs_b512.json

