import json,sys
for l in sys.stdin:
    l=l.strip()
    if not l.startswith('{'): continue
    d=json.loads(l)
    print(d['tag'], d.get('opts'), ' '.join(f"{k}={d[k]['ms']:.4f}" for k in d if isinstance(d[k],dict) and 'ms' in d[k]))
