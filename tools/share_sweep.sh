cd $GRAFT_REPO_ROOT
for J in 8 16 32 64; do
  for SC in tail busy warm; do
    BS_SCAN_SHARE=$J python bench.py --scenario $SC --steps 300 --warmup 30 --no-cpu-baseline --no-pmc --no-extras | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('J=$J $SC', round(d['ms_per_step']*1000,2), {k:round(v*1000,2) for k,v in d['kernel_ms_per_step'].items()})"
  done
done
