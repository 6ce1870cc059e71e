show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%s  value %.3e  kernel_ms %.4f (min %.4f)  frac %.3f region %.3f' % (r['kernel'], d['value'], r['kernel_ms_mean'], r['kernel_ms_min'], r['frac'], d['roofline_region']['frac']))"; }
for rep in 1 2; do for so in snowmocap_amd/csrc/ab/libsnowtri_coopA.so snowmocap_amd/csrc/ab/libsnowtri_coopB.so; do
  echo -n "$(basename $so): "; SNOWTRI_LIB=$PWD/$so python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra --no-per-frame --repeats 3 --large-frames 0 2>&1 | tail -1 | show
done; done
