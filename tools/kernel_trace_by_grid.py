import csv,sys,glob,collections
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
d=collections.defaultdict(lambda:[0,0])
for r in csv.DictReader(open(f)):
    k=(r['Kernel_Name'][:60],r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size'))
    d[k][0]+=1; d[k][1]+=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
tot=sum(v[1] for v in d.values())
for k,v in sorted(d.items(),key=lambda kv:-kv[1][1])[:28]:
    print(f"{k[0]:60s} grid={k[1]:>9} n={v[0]:6d} avg={v[1]/v[0]/1e3:8.1f}us tot={v[1]/1e6:8.1f}ms {100*v[1]/tot:4.1f}%")
