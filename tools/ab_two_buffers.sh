cd $GRAFT_REPO_ROOT
for c in tmall_default cfg2 taobao_default ccmr_default; do
  bash tools/ab_args.sh 2 "--set plan_two_workspaces=False" "--set plan_two_workspaces=True" -- --config $c --steps 2000 --warmup 300 2>&1 | sed "s/^/$c /" | cut -c1-110
done
