"""Compare two tools/bench_ops.py sweeps: python tools/cmp_ops.py old.jsonl new.jsonl"""
import json, sys
def load(f):
    d={}
    for l in open(f):
        try: r=json.loads(l)
        except ValueError: continue
        k=json.dumps({a:b for a,b in r.items() if a not in('us','tflops','gbps','tile','split','kernel','plan','best_tile','best_split')},sort_keys=True)
        if 'us' in r and (k not in d or r['us']<d[k]['us']): d[k]=r
    return d
a=load(sys.argv[1]); b=load(sys.argv[2])
ta=tb=0
for k in b:
    if k in a and 'us' in a[k]:
        ta+=a[k]['us'];tb+=b[k]['us']
        r=b[k]
        print('%-100s %8.1f -> %8.1f  %+4.0f%%  %6.0f TF/s %s'%(k[:100],a[k]['us'],b[k]['us'],100*(b[k]['us']/a[k]['us']-1), r.get('tflops',0), 'tile %s split %s'%(r.get('tile',r.get('best_tile','')),r.get('split',r.get('best_split','')))))
print('sum us %.1f -> %.1f'%(ta,tb))
