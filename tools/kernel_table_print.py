import json,sys
t=json.load(open(sys.argv[1]))
tab=t.get('kernels',t)
n=t.get('nprof',3)
rows=sorted(tab.items(), key=lambda kv:-kv[1]['ms'])
tot=sum(v['ms'] for k,v in rows)
for k,v in rows[:16]:
    print(f"{k[:62]:62s} n={v['launches']//n:4d} ms/step={v['ms']/n:7.3f} us/launch={v['ms']/v['launches']*1e3:7.1f} TF={v['flops']/(v['ms']*1e-3)/1e12 if v['ms'] else 0:6.0f}")
print('sum', tot/n)
