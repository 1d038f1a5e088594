import csv,glob,sys
f=glob.glob(sys.argv[1]+'/*/*_kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
kn=[r['Kernel_Name'] for r in rows]
starts=[i for i,k in enumerate(kn) if 'nhwc_to_pnc' in k]
def cls(k): return 'gemm' if 'seg_gemm' in k else 'ew'
for si in [int(a) for a in sys.argv[2:]]:
    i0,i1=starts[si],starts[si+1]
    seq=rows[i0:i1]
    t0=int(seq[0]['Start_Timestamp']); t1=max(int(r['End_Timestamp']) for r in seq)
    qs={}
    for r in seq:
        q=(r['Queue_Id'],r['Stream_Id']); qs.setdefault(q,[0,0]); qs[q][0]+=1; qs[q][1]+=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
    ev=[]
    for r in seq:
        c=cls(r['Kernel_Name']); ev.append((int(r['Start_Timestamp']),1,c)); ev.append((int(r['End_Timestamp']),-1,c))
    ev.sort()
    act={'gemm':0,'ew':0}; last=ev[0][0]; acc={}
    for t,d,c in ev:
        key=(min(act['gemm'],2),act['ew']>0)
        acc[key]=acc.get(key,0)+t-last; last=t
        act[c]+=d
    print('step',si,'span us',(t1-t0)/1e3,'n',len(seq),qs)
    for k,v in sorted(acc.items()): print('   #gemm active, ew active =',k, round(v/1e3,1),'us')
