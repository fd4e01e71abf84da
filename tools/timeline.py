import csv,glob,sys
t=glob.glob(sys.argv[1]+'/*/*kernel_trace.csv')[0]
rows=sorted(csv.DictReader(open(t)), key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'focal_stream' in r['Kernel_Name']]
i0=idx[150]
base=int(rows[i0-4]['Start_Timestamp'])
prev_end=None
for r in rows[i0-4:i0+14]:
    s=int(r['Start_Timestamp'])-base; e=int(r['End_Timestamp'])-base
    gap = '' if prev_end is None else f"gap {(s-prev_end)/1e3:6.2f}"
    print(f"{s/1e3:9.2f} -> {e/1e3:9.2f} us  dur {(e-s)/1e3:7.2f} {gap:12s} {r['Kernel_Name'][:60]}")
    prev_end=e
