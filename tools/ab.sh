set -e
for v in old new old new; do
  cp ab/lib_$v.so drakegpt_amd/lib/libdrakegpt_hip.so
  echo "== $v"; timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-extra | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])"
done
