cd $GRAFT_REPO_ROOT
for d in "" -DTNP_NOMFMA -DTNP_NOLOAD -DTNP_NOSTORE -DTNP_NOCSTORE "-DTNP_NOLOAD -DTNP_NOSTORE -DTNP_NOCSTORE" "-DTNP_NOMFMA -DTNP_NOSTORE" $TNP_EXTRA; do
  echo "== build: ${d:-full}"
  timeout -k 10 200 python3 tools/tnp_probe.py $d 2>&1 | grep "^K=\|rror" | cut -c1-150
done
