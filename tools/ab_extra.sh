# usage: bash tools/ab_extra.sh CONFIG B PREC -- ab/lib_old.so vs ab/lib_new.so alternated on one box through tools/extra_bench.py
for v in old new old new; do
  cp ab/lib_$v.so drakegpt_amd/lib/libdrakegpt_hip.so
  echo "== $v"; timeout -k 10 280 python tools/extra_bench.py $1 $2 $3 5 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'], d['final_loss'])" || exit 1
done
