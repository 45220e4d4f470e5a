import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('C3', round(d['ms_per_step']*1e3,2), 'k', round(d['roofline']['kernel_us'],2), end=' | ')
for k,v in d.get('regimes',{}).items(): print(k.split('_n')[0]+('H16' if 'H16' in k else ''), round(v['us_per_iteration'],2), 'k', round(v['gradient_kernel_us'],2), end=' | ')
print()
