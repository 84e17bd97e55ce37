# usage: bash tools/ab_env.sh VAR a b   -- alternate bench runs with VAR=a / VAR=b on one box (ms/step, tokens/s, final loss)
set -e
for r in 1 2; do for v in $2 $3; do
  echo "== $1=$v"; env $1=$v timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-extra | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'], d['final_loss'])"
done; done
