export GPBO_LIB=$PWD/ab_libs/diag_sigma.so
for v in 0 2 0 2; do
  for n in 512 1024 4096; do
    m=1048576; [ $n = 4096 ] && m=524288
    GPBO_SIGMA_VARIANT=$v python bench.py --no-cpu-baseline --no-also --steps 4 --warmup 1 --n-obs $n --m-per-gpu $m 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant', $v, 'N', $n, 'step', round(d['ms_per_step'],3), 'sigma', d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
  done
done
