"""print the per-dispatch timeline of the last train step in a rocprofv3 kernel-trace CSV (diagnostic)."""
import csv, sys, re
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'noise_' in r['Kernel_Name']]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2         # which step (default: one from the middle = the timed region)
seg = rows[idx[k]:idx[k + 1]]
t0 = int(seg[0]['Start_Timestamp']); tot = 0
def short(n):
    m = re.search(r'(\w+_kernel)', n)
    k = m.group(1) if m else n[:30]
    t = re.findall(r'Li(\d+)E', n)
    return k + ('<' + ','.join(t) + '>' if t else '')
for r in seg:
    dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += dur
    g = (int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), int(r['Grid_Size_Y']), int(r['Grid_Size_Z']))
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} q{r.get('Queue_Id','?'):>2s} {short(r['Kernel_Name']):44s} blocks={str(g):18s} vgpr={r['VGPR_Count']:>3s} lds={r['LDS_Block_Size']:>6s} {dur:8.1f} us")
print("sum of kernel time %.1f us, span %.1f us" % (tot, (int(seg[-1]['End_Timestamp']) - t0) / 1e3))
