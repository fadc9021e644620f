import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in rows:
    k=r['Kernel_Name'].split('(')[0].replace('tsamd::(anonymous namespace)::','')[:40]
    agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in agg.items():
    d=len(set(r['Dispatch_Id'] for r in rows if r['Kernel_Name'].split('(')[0].replace('tsamd::(anonymous namespace)::','')[:40]==k))
    print(k, 'dispatches', d)
    for c,val in sorted(v.items()): print('   %-28s %14.0f per dispatch'%(c, val/d))
