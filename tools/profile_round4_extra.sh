# Round-4 side benches (run on the GPU box through gpurun; outputs under gpurun_out/$1)
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r04x}
mkdir -p $O
cd $R
python3 bench.py --steps 20 --warmup 5 > $O/bench_cfg3_driver_protocol.json 2>$O/err.log &&
python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline > $O/bench_cfg3_200.json 2>>$O/err.log
for i in 1 2 3 4 5; do python3 bench.py --no-cpu-baseline --force-sharded --steps 40 --warmup 10 > $O/bench_sharded_1rank_$i.json 2>>$O/err.log; done
python3 bench.py --no-cpu-baseline --no-side --config cfg2 --steps 300 > $O/bench_cfg2.json 2>>$O/err.log
python3 bench.py --no-cpu-baseline --no-side --config tmall_default --steps 300 > $O/bench_tmall_default.json 2>>$O/err.log
python3 bench.py --no-cpu-baseline --no-side --config ccmr_default --steps 300 > $O/bench_ccmr_default.json 2>>$O/err.log
python3 bench.py --no-cpu-baseline --no-side --config taobao_default --steps 300 > $O/bench_taobao_default.json 2>>$O/err.log
python3 bench.py --no-cpu-baseline --no-side --config cfg5_taobao --steps 10 --warmup 3 --batches 2 > $O/bench_cfg5_taobao.json 2>>$O/err.log
python3 bench.py --no-cpu-baseline --no-side --config cfg5_tmall --steps 6 --warmup 2 --batches 2 > $O/bench_cfg5_tmall.json 2>>$O/err.log
SCORE_BENCH_DEVICE=0 SCORE_DIST_BACKEND=gloo python3 bench.py --gpus 2 --no-cpu-baseline --steps 40 > $O/bench_2ranks_gloo_one_gpu.json 2>>$O/err.log
python3 tools/eval_bench.py > $O/eval_1p99.json 2>>$O/err.log
ls -la $O
