# the previous round's tree against this tree on ONE box, interleaved:
#   (build container)  mkdir _r4tree && git archive <last commit of the previous round> | tar -x -C _r4tree && (cd _r4tree && python -c "from score_amd import build; build.build()")
#   (GPU box)          bash tools/ab_rounds.sh <out-under-gpurun_out>          -- _r4tree/ is git-ignored and travels with the snapshot; remove it afterwards
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O
{
for c in tmall_default cfg2 taobao_default ccmr_default; do
  echo "== $c (--steps 2000 --warmup 200)"
  bash $GRAFT_REPO_ROOT/tools/ab_trees.sh 2 _r4tree . -- --config $c --steps 2000 --warmup 200
done
echo "== cfg3 (--steps 200 --warmup 20)"
bash $GRAFT_REPO_ROOT/tools/ab_trees.sh 2 _r4tree . -- --steps 200 --warmup 20
} > $O/ab_rounds.txt 2>&1
cat $O/ab_rounds.txt | cut -c1-80
