cd $GRAFT_REPO_ROOT
for flags in "--no-cpu-baseline --no-secondary --warmup 5 --steps 20" "--warmup 5 --steps 20" ; do
  python bench.py $flags 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$flags', 'value %.2fM api %.2fM ms/step %.4f kernel_ms %.4f frac %.4f'%(d['value']/1e6, d['value_api']/1e6, d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])); print({k:(v.get('value'), v.get('ms')) for k,v in d.get('secondary',{}).items()})"
done
