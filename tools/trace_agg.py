import csv, glob, collections, sys
f = glob.glob(sys.argv[1]+'/*/*kernel_trace.csv')[0]
keys = sys.argv[2].split(',')
agg = collections.defaultdict(lambda:[0,0])
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    for key in keys:
        if key in n:
            k=(n.split('dh')[1][:30] if 'dh' in n else n[:30], r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'])
            agg[k][0]+=1; agg[k][1]+=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1])[:int(sys.argv[3]) if len(sys.argv)>3 else 25]:
    print(k, v[0], round(v[1]/v[0]/1e3,1),'us', round(v[1]/1e6,2),'ms')
