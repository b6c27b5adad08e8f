import csv,glob,sys,collections
f=sorted(glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True))[0]
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name']
    if 'dgs::' not in k: continue
    name=k.split('(')[0].replace('void ','')+' grid='+r.get('Grid_Size', r.get('Grid_Size_X', '?'))
    acc[name].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in acc.items():
    v=sorted(v); print(f'{len(v):5d} calls  median {v[len(v)//2]:8.1f} us  min {v[0]:8.1f}  {k}')
