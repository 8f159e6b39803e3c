cd $GRAFT_REPO_ROOT
for P in 130 132 134 136 138 140 142; do
  rm -f adt_str_amd/csrc/logmel.o
  make -C adt_str_amd/csrc EXTRA=-DADT_LOGMEL_PITCH=$P > /dev/null 2>&1
  echo -n "P=$P  "; timeout 300 python bench.py --workload logmel --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4),'ms', round(d['value']))"
done
