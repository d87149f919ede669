"""Summarise the last training step of a rocprofv3 --kernel-trace CSV: python tools/trace_summary.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
pre = [i for i, r in enumerate(rows) if 'preprocess_kernel' in r['Kernel_Name']]
nper = int(sys.argv[2]) if len(sys.argv) > 2 else 4
step = rows[pre[-nper]:]
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in step)
print('last step: kernels', len(step), 'busy ms %.3f' % (busy / 1e6))
agg = defaultdict(lambda: [0, 0])
for r in step:
    n = r['Kernel_Name']
    key = n[:50]
    if 'conv_igemm' in n or 'conv_wgrad' in n:
        key = (n[:44], int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']))
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    agg[key][0] += d
    agg[key][1] += 1
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print(f"{v[0] / 1e3:9.1f} us  x{v[1]:4d}  avg {v[0] / v[1] / 1e3:8.1f}  {k}")
