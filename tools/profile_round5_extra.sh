# Round-5 bench lines (run on the GPU box through gpurun; outputs under gpurun_out/$1)
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r05x}
mkdir -p $O
cd $R
python3 bench.py --steps 20 --warmup 5 > $O/bench_cfg3_driver_protocol.json 2>$O/err.log &&
python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline > $O/bench_cfg3_200.json 2>>$O/err.log
for c in cfg2 tmall_default taobao_default ccmr_default; do
  python3 bench.py --no-cpu-baseline --no-side --config $c --steps 2000 --warmup 100 > $O/bench_$c.json 2>>$O/err.log
  python3 bench.py --no-cpu-baseline --no-side --config $c --steps 2000 --warmup 100 --debug-flags 512 > $O/bench_${c}_layered.json 2>>$O/err.log
done
python3 bench.py --no-cpu-baseline --no-side --config cfg5_taobao --steps 10 --warmup 3 --batches 2 > $O/bench_cfg5_taobao.json 2>>$O/err.log
python3 bench.py --no-cpu-baseline --no-side --config cfg5_tmall --steps 6 --warmup 2 --batches 2 > $O/bench_cfg5_tmall.json 2>>$O/err.log
for c in cfg2 tmall_default; do
  timeout -k 10 300 bash tools/kernel_sequence.sh ${1:-r05x}/seq_$c --config $c > $O/seq_$c.log 2>&1
done
python3 tools/host_calls.py tmall_default > $O/host_calls_tmall_default.txt 2>&1
python3 tools/host_calls.py cfg2 > $O/host_calls_cfg2.txt 2>&1
ls -la $O
# the per-phase times of the two per-sample kernels (a -DPS_PHASE_TIMING build: python tools/build_variant.py phase ps_fwd.hip,ps_bwd.hip -DPS_PHASE_TIMING)
if [ -f score_amd/lib/libscore_hip_phase.so ]; then
  for c in tmall_default cfg2 ccmr_default; do
    SCORE_HIP_LIB=score_amd/lib/libscore_hip_phase.so python3 tools/ps_phase_probe.py $c > $O/phases_$c.txt 2>&1
  done
  SCORE_HIP_LIB=score_amd/lib/libscore_hip_phase.so python3 tools/ps_phase_probe.py tmall_default 8 > $O/phases_tmall_default_b8.txt 2>&1
fi
