# The "first RCCL process of a fresh box is ~20 % slower" effect (profiles/README.md): three identical one-rank runs of the
# sharded bench back to back on a fresh box, RCCL's own log of each, and the stage table of each.
#   bash tools/rccl_first_process.sh <outdir-under-gpurun_out>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03_rccl}; mkdir -p $O; cd $R
for i in 1 2 3; do
  NCCL_DEBUG=INFO NCCL_DEBUG_FILE=$O/nccl_$i.log python3 bench.py --no-cpu-baseline --force-sharded --steps 200 > $O/run_$i.json 2> $O/run_$i.err
  python3 - $O/run_$i.json $i <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("run", sys.argv[2], round(j["value"]), "samples/s", round(j["ms_per_step"], 4), "ms", {k: (round(v, 3) if v else v) for k, v in j["stages_ms"].items()})
PY
done
python3 bench.py --no-cpu-baseline --no-side --steps 200 > $O/run_unsharded.json 2> $O/run_unsharded.err
python3 - $O/run_unsharded.json u <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("unsharded", round(j["value"]), "samples/s", round(j["ms_per_step"], 4), "ms", {k: (round(v, 3) if v else v) for k, v in j["stages_ms"].items()})
PY
ls -la ~/.cache 2>/dev/null | head; wc -l $O/nccl_*.log
diff <(sed 's/^[^ ]* //' $O/nccl_1.log | sed 's/[0-9]\{4,\}/N/g' | sort -u) <(sed 's/^[^ ]* //' $O/nccl_2.log | sed 's/[0-9]\{4,\}/N/g' | sort -u) | head -30
