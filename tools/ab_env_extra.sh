# usage: bash tools/ab_env_extra.sh VAR a b CONFIG B PREC   -- alternate tools/extra_bench.py runs with VAR=a / VAR=b on one box
for r in 1 2; do for v in $2 $3; do
  echo "== $1=$v"; env $1=$v timeout -k 10 280 python tools/extra_bench.py $4 $5 $6 5 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'], d['final_loss'])" || exit 1
done; done
