import json, sys
for l in open(sys.argv[1]):
    if not l.startswith('SEG'): print(l.strip()); continue
    d=json.loads(l[4:])
    print(d['segments'], 'nn_kernel %.3f (max %.3f)'%(d['nn_kernel_ms']['mean'], d['nn_kernel_ms']['max']), {k:(round(v['mean'],3) if isinstance(v,dict) and 'mean' in v else round(v,3)) for k,v in d.items() if k.endswith('_ms') and k!='nn_kernel_ms'})
