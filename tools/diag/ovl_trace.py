import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# take the last step-ish window: find overlaps of tn_group kernels with others
ov = 0; tot = 0; n = 0
ends = []
for i, r in enumerate(rows):
    if 'tn_group_big' in r['Kernel_Name']:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        o = 0
        for q in rows[max(0, i - 30): i + 30]:
            if q is r: continue
            qs, qe = int(q['Start_Timestamp']), int(q['End_Timestamp'])
            o += max(0, min(e, qe) - max(s, qs))
        ov += o; tot += e - s; n += 1
print(f"{n} group launches, mean {tot / max(n,1) / 1e3:.1f} us, overlapped with other kernels {ov / max(n,1) / 1e3:.1f} us each")
