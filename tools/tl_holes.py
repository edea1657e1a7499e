import sys, json
rows=[]
for line in open(sys.argv[1]):
    if line.startswith("#"): continue
    tag, units, t0, dur = line.split()
    rows.append((float(t0), float(dur), tag))
rows.sort()
print("launches", len(rows), "span ms", (rows[-1][0]+rows[-1][1]-rows[0][0])/1e3)
# long kernels
import collections
med=collections.defaultdict(list)
for t0,d,tag in rows: med[tag].append(d)
med={k:sorted(v)[len(v)//2] for k,v in med.items()}
longk=[(round(t0/1e3,2),tag,round(d/1e3,2),round(med[tag]/1e3,3)) for t0,d,tag in rows if d>3000 and d>4*med[tag]]
print("kernels > 3 ms and > 4x their median:", len(longk)); 
for r in longk[:60]: print("  ",r)
# idle holes
end=rows[0][0]; holes=[]
for t0,d,tag in rows:
    if t0-end>500: holes.append((round(end/1e3,2), round((t0-end)/1e3,2)))
    end=max(end,t0+d)
print("idle holes > 0.5 ms:", holes[:40])
