import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
pk=d['roofline']['per_kernel']
print(sys.argv[1], d['value'], d['ms_per_step'], 'frac', d['roofline']['frac'])
print('  '+' '.join(f"{k.split('/')[-1]}={v['avg_us']:.0f}" for k,v in pk.items()))
